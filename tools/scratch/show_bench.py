import json, sys
for p in [a for a in sys.argv[1:] if not a.startswith("--")]:
    txt = [l for l in open(p).read().split("\n") if l.startswith("{")]
    d = json.loads(txt[-1])
    print("==", p)
    for k in ("value", "ms_per_step", "bases_per_s", "plain_ms_per_step", "plain_records_per_s_per_gpu", "shuffle_ms_per_step", "wire_bytes_per_step",
              "wire_bytes_off_rank_per_step", "partials_per_step", "dist_phase_host_ms_per_step", "config", "launches_per_step",
              "gpu_kernel_ms_per_step_total", "step_frac_of_hbm_peak_algorithmic", "kernel_path_host_to_host", "end_to_end", "cpu_baseline"):
        if k in d:
            print(" ", k, d[k])
    for k in ("roofline", "roofline_coverage", "roofline_collapse"):
        if k in d:
            r = d[k]
            print(" ", k, {x: r.get(x) for x in ("kernel", "avg_launch_us", "launch_us_min", "launch_us_max", "achieved", "frac", "traffic", "frac_traffic", "coverage_call_ms_median", "frac_whole_call")})
    if "--kernels" in sys.argv or True:
        km = d.get("kernel_ms_per_step", {})
        print("  kernels:", ", ".join("%s %.2f" % (k.split("/")[1], v) for k, v in list(km.items())[:24]))
