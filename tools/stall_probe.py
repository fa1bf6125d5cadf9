"""Diagnosis: what the host's memory management does under a run of the command line whose library calls stall now and then
(NUMA balancing, huge-page compaction: /proc/vmstat deltas per run beside the run's phase lines)."""
import atexit, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import synth, synth_dev

def rd(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return "n/a (%s)" % e.__class__.__name__

for p in ("/proc/sys/kernel/numa_balancing", "/sys/kernel/mm/transparent_hugepage/enabled", "/sys/kernel/mm/transparent_hugepage/defrag",
          "/sys/kernel/mm/transparent_hugepage/khugepaged/defrag", "/sys/kernel/mm/transparent_hugepage/khugepaged/scan_sleep_millisecs"):
    print(p, "=", rd(p))
print(subprocess.run(["numactl", "-H"], capture_output=True, text=True).stdout[:600] if os.path.exists("/usr/bin/numactl") else "no numactl")
print("cpus allowed:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:40])

KEYS = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated", "pgmigrate_success", "thp_fault_alloc", "thp_fault_fallback", "thp_collapse_alloc",
        "compact_stall", "compact_success", "pgfault", "thp_split_page", "thp_split_pmd", "allocstall_normal", "pgscan_direct")
def vm():
    d = {}
    for l in open("/proc/vmstat"):
        k, v = l.split()
        if k in KEYS:
            d[k] = int(v)
    return d

files, reads, runs, seq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extra_env = dict(kv.split("=", 1) for kv in sys.argv[5:])
d = tempfile.mkdtemp(prefix="tbk_stall_", dir="/tmp")
atexit.register(shutil.rmtree, d, True)
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, os.environ.get("STALL_PROFILE", "c2"), device="cuda:0"))
paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=bool(seq))
del tile
os.sync()
out = os.path.join(d, "out.bam")
pause = float(os.environ.get("STALL_PAUSE", "0"))
for r in range(runs):
    if os.path.exists(out):
        os.remove(out)
    time.sleep(pause)
    a = vm()
    t = time.time()
    p = subprocess.run([os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush"), "-o", out] + os.environ.get("STALL_FLAGS", "").split() + paths, capture_output=True, text=True,
                       env={**os.environ, "TBK_TIMING": "1", "TBK_EXIT_TIMING": "1", **extra_env})
    t_end = time.time()
    dt = t_end - t
    ex = [l for l in p.stderr.split("\n") if l.startswith("exit timing")]
    after_exit = (t_end - float(ex[-1].rsplit(" ", 1)[1])) if ex else float("nan")
    b = vm()
    os.sync()
    print("run %d: %.3f s (%.3f after _exit) rc %d | %s" % (r, dt, after_exit, p.returncode, " ".join("%s+%d" % (k, b[k] - a[k]) for k in KEYS if b.get(k, 0) != a.get(k, 0))))
    for l in p.stderr.split("\n"):
        if l.startswith(("host path", "hybrid path", "representatives", "writer closed", "released", "collapse phases", "YD stage ms", "device writer stages")) and "D2H of 0.0" not in l and "pointer attr" not in l:
            print("     ", l[:260])
