"""Shared helpers of the parity tests (golden normaliser of SURVEY.md §4.4)."""
import os

import numpy as np

from tiebrush_amd import bamio, soa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sample_paths(name):
    return [os.path.join(GOLDEN, name, "%ss%d.bam" % (name, i)) for i in range(10)]


def compare_groups_to_golden_bam(res, tile, bams, gold):
    """res: dict with rep/yc/yx/yd in output order.  Golden normaliser: absent YC == 1,
    absent YX == 1, absent YD == 0 (bamio applies these defaults); identity = all fixed
    fields + qname + cigar + seq + qual."""
    assert res["n_groups"] == gold.n
    fo = tile.file_of()
    bad = []
    for k in range(gold.n):
        gi = int(res["rep"][k])
        f = int(fo[gi])
        idx = gi - int(tile.file_off[f])
        ok = bamio.record_identity(bams[f], idx) == bamio.record_identity(gold, k)
        gyc = gold.yc[k] if gold.has_yc[k] else 1.0
        ok = ok and float(np.float32(res["yc"][k])) == gyc and int(res["yx"][k]) == int(gold.yx[k]) \
            and int(res["yd"][k]) == int(gold.yd[k])
        if not ok:
            bad.append(k)
    return bad


def bedgraph_lines(c, names, fmt="int"):
    out = ["track type=bedGraph"]
    for i in range(c["n_intervals"]):
        v = c["iv_val"][i]
        vs = "%d" % int(v) if fmt == "int" else "%.3f" % v
        out.append("%s\t%d\t%d\t%s" % (names[c["iv_tid"][i]], c["iv_start"][i], c["iv_end"][i], vs))
    return out


def junction_lines(c, names, fmt="int"):
    out = ["track name=junctions"]
    for i in range(c["n_junctions"]):
        v = c["j_val"][i]
        vs = "%d" % int(v) if fmt == "int" else "%.3f" % v
        out.append("%s\t%d\t%d\tJUNC%08d\t%s\t%s" % (names[c["j_tid"][i]], c["j_start"][i], c["j_end"][i], i + 1, vs,
                                                     chr(c["j_strand"][i])))
    return out


def read_lines(path):
    with open(path) as fh:
        g = fh.read().split("\n")
    if g and g[-1] == "":
        g = g[:-1]
    return g


def tile_from_records(files):
    """files: list of lists of (tid, pos, flag, mapq, strand, nh, [(len, op) ...]) in file order -> SoATile"""
    allr = [r for f in files for r in f]
    n, k = len(allr), len(files)
    fo = np.zeros(k + 1, np.uint32)
    fo[1:] = np.cumsum([len(f) for f in files])
    cigs = [[(l << 4) | o for l, o in r[6]] for r in allr]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(c) for c in cigs])
    return soa.SoATile(
        n_files=k, file_off=fo, tbmerged=np.zeros(k, np.uint8), tid=np.array([r[0] for r in allr], np.int32),
        pos=np.array([r[1] for r in allr], np.int32), flag=np.array([r[2] for r in allr], np.uint16),
        mapq=np.array([r[3] for r in allr], np.uint8), strand=np.array([ord(r[4]) for r in allr], np.uint8),
        nh=np.array([r[5] for r in allr], np.int32), cig_off=off, cig=np.array([x for c in cigs for x in c], np.uint32))


def paired_end_like_files():
    """What a real coordinate-sorted BAM holds besides mapped reads: unmapped mates placed at their mate's position, unplaced
    reads (tid -1) behind everything, an input without any record — and a CIGAR that ends in an intron."""
    M, N, S = 0, 3, 4
    un = (-1, -1, 4, 0, ".", -(2**31), [])
    f0 = [(0, 100, 0, 60, "+", 1, [(50, M)]), (0, 100, 4 | 8, 0, ".", 1, []), (0, 100, 0, 60, "+", 1, [(30, M)]),
          (0, 100, 0, 60, "+", 1, [(50, M)]), (0, 120, 0, 60, "+", 1, [(30, M)]), (1, 9, 0, 60, ".", 1, [(24, M), (7, N), (2, S)]),
          (1, 40, 0, 60, ".", 1, [(15, M), (40, N), (10, M)]), (1, 41, 0, 60, "+", 1, [(10, M), (20, N), (30, M)]), un, un]
    f1 = []
    f2 = [(0, 100, 0, 60, "+", 1, [(30, M)]), (0, 100, 0, 60, "+", 1, [(50, M)]), (0, 500, 4 | 8, 0, ".", 1, []),
          (1, 41, 16, 60, "+", 1, [(10, M), (20, N), (30, M)]), (2, 5, 0, 60, "-", 1, [(40, M)]), un]
    f3 = [(0, 90, 0, 60, "-", 1, [(30, M)]), (2, 5, 0, 60, "-", 1, [(40, M)]), (2, 7, 0, 0, "-", 1, [(40, M)])]
    return [f0, f1, f2, f3]


def tbk_debug(monkeypatch, **kw):
    """test hooks of the library (struct TbkDebug, tiebrush_amd/csrc/tbk_internal.h) for the contexts and command lines of this test:
    merged into TBK_DEBUG ("key=value,..."); a value of None takes a key out again"""
    cur = dict(x.split("=", 1) for x in os.environ.get("TBK_DEBUG", "").split(",") if x)
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    monkeypatch.setenv("TBK_DEBUG", ",".join("%s=%s" % kv for kv in cur.items()))
