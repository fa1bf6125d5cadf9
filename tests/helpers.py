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
