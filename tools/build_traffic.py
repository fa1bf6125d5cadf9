#!/usr/bin/env python3
"""profiles/traffic_<workload>.json from the two PMC summaries (tools/pmc_summary.py of a --pmc FETCH_SIZE pass and of a
--pmc WRITE_SIZE pass of the same bench.py command): HBM bytes per launch and bench.py kernel name.
Usage: build_traffic.py <fetch_summary.csv> <write_summary.csv> <out.json> "<workload text>"
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is taken as reported (see profiles/README.md for why the x2 correction of
MI355X_MICROARCH.md for 16-byte streaming reads is not applied to these 4/8-byte-per-lane kernels)."""
import csv, json, re, sys, collections

ALIAS = {  # kernel symbol stem -> the name bench.py reports (TBK_LAUNCH name) where the two differ
    "yd_fill_w": "yd_fill", "yd_gcount_w": "yd_gcount", "w64_scatter": "rx_scatter", "col_recgroup_w": "col_recgroup",
    "wg_offsets_stream": "wg_offsets", "wg_offsets_edges": "wg_offsets_edges", "wg_finish_raw": "wg_finish", "wg_sample_raw": "wg_sample",
    "w64_emit_flat": "yd_scatter", "yd_lscatter": "yd_scatter", "yd_ltotal": "yd_lscan", "g2c_gather_key": "g2c_gather", "g2c_count_key": "g2c_count",
    "wg_finish_sparse": "wg_finish",
}
SCAN = {"EffKey": "col_effkey_scan", "SegMaxY": "yd_chain_scan", "SegMax": "cov_bundle_scan", "ShKey": "shard_eff_scan",
        "HeadNex": "yd_chain_number", "PmKey": "partial_emax_scan"}
TWO = {"SegMaxY": "yd_chains", "SegMax": "cov_bundles"}   # so_two_k: the two-stage scans (heads found and numbered in one pass)


def bench_name(sym):
    m = re.search(r"(\w+)_k\b", sym)
    if not m:
        return None
    stem = m.group(1)
    if stem == "so_two":
        t = re.search(r"::(SegMaxY|SegMax)\b", sym)
        return TWO.get(t.group(1)) if t else stem
    if stem in ("so_reduce", "so_spine", "so_down", "so_single"):
        t = re.search(r"::(EffKey|SegMaxY|SegMax|ShKey|HeadNex|PmKey)\b", sym)
        return SCAN.get(t.group(1)) if t else stem
    if stem == "w64_scatter" and "YdEmit" in sym:
        return "yd_scatter"
    return ALIAS.get(stem, stem)


def load(path, col):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            n = bench_name(r["kernel"])
            if n is None or not r.get(col):
                continue
            d = int(float(r["dispatches"]))
            acc[n][0] += float(r[col]) * 1024.0 * d
            acc[n][1] += d
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for n in sorted(set(fe) | set(wr)):
    f = fe[n][0] / fe[n][1] if n in fe and fe[n][1] else 0.0
    w = wr[n][0] / wr[n][1] if n in wr and wr[n][1] else 0.0
    out[n] = int(f + w)
json.dump({"workload": sys.argv[4] if len(sys.argv) > 4 else "", "unit": "bytes of HBM traffic per launch (FETCH_SIZE + WRITE_SIZE)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), summarised by tools/pmc_summary.py; FETCH_SIZE as "
                     "reported (4/8-byte-per-lane reads: the gfx950 x2 correction for 16-byte streaming reads does not apply)",
           "bytes_per_launch": out}, open(sys.argv[3], "w"), indent=1)
print(len(out), "kernels ->", sys.argv[3])
