"""ctypes binding of the CPU oracle (oracle/tb_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
— never by anything under tiebrush_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib(opt: str = "O2"):
    if opt not in _LIBS:
        name = "libtb_oracle.so" if opt == "O2" else "libtb_oracle_O0.so"
        path = os.path.join(os.environ.get("TB_ORACLE_BUILD_DIR") or os.path.join(_HERE, "_build"), name)   # (the override: tools/san_check.sh)
        if not os.path.exists(path):
            build()
        _LIBS[opt] = C.CDLL(path)
    return _LIBS[opt]


class Opts(C.Structure):
    _fields_ = [("strategy", C.c_int32), ("max_nh", C.c_int32), ("min_qual", C.c_int32), ("flags_mask", C.c_uint32),
                ("keep_supplementary", C.c_uint8), ("keep_secondary", C.c_uint8), ("keep_unmapped", C.c_uint8),
                ("collapse_same", C.c_uint8), ("store_frac", C.c_uint8)]


_P = C.c_void_p


class In(C.Structure):
    _fields_ = [("n_files", C.c_uint32), ("n_records", C.c_uint32), ("file_off", _P), ("tbmerged", _P), ("tid", _P),
                ("pos", _P), ("flag", _P), ("mapq", _P), ("strand", _P), ("nh", _P), ("cig_off", _P), ("cig", _P),
                ("yc_in", _P), ("yx_in", _P), ("yd_in", _P), ("md_off", _P), ("md", _P), ("md_has", _P),
                ("qn_off", _P), ("qn", _P)]


class Groups(C.Structure):
    _fields_ = [("cap", C.c_uint32), ("rep", _P), ("yc", _P), ("yx", _P), ("yd", _P), ("g_start", _P), ("g_end", _P),
                ("rec_group", _P), ("merge_order", _P), ("n_groups", C.c_uint32), ("n_passed", C.c_uint32)]


class CovIn(C.Structure):
    _fields_ = [("n_records", C.c_uint32), ("tid", _P), ("pos", _P), ("flag", _P), ("cig_off", _P), ("cig", _P),
                ("yc", _P), ("strand", _P), ("yx", _P)]


class CovOut(C.Structure):
    _fields_ = [("cap_intervals", C.c_uint32), ("iv_tid", _P), ("iv_start", _P), ("iv_end", _P), ("iv_val", _P),
                ("cap_junctions", C.c_uint32), ("j_tid", _P), ("j_start", _P), ("j_end", _P), ("j_strand", _P),
                ("j_val", _P), ("cap_sample", C.c_uint32), ("num_samples", C.c_int32), ("s_tid", _P), ("s_start", _P),
                ("s_end", _P), ("s_count", _P), ("s_heat", _P), ("n_intervals", C.c_uint32),
                ("n_junctions", C.c_uint32), ("n_sample", C.c_uint32), ("n_bases", C.c_uint64),
                ("span_bases", C.c_uint64)]


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def make_opts(strategy=0, max_nh=2**31 - 1, min_qual=-1, keep_supplementary=False, keep_secondary=False,
              keep_unmapped=False, collapse_same=False, store_frac=False, flags_mask=0) -> Opts:
    return Opts(strategy, max_nh, min_qual, flags_mask, int(keep_supplementary), int(keep_secondary),
                int(keep_unmapped), int(collapse_same), int(store_frac))


def collapse(tile, opt: str = "O2", want_rec_group=False, want_merge_order=False, **kw):
    """Run tbo_collapse on a tiebrush_amd.soa.SoATile; returns a dict of numpy arrays."""
    L = lib(opt)
    o = make_opts(**kw)
    n = tile.n_records
    keep = []

    def c(a, dt):
        if a is None:
            return None
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b.ctypes.data

    i = In(tile.n_files, n, c(tile.file_off, np.uint32), c(tile.tbmerged, np.uint8), c(tile.tid, np.int32),
           c(tile.pos, np.int32), c(tile.flag, np.uint16), c(tile.mapq, np.uint8), c(tile.strand, np.uint8),
           c(tile.nh, np.int32), c(tile.cig_off, np.uint32), c(tile.cig, np.uint32), c(tile.yc_in, np.float64),
           c(tile.yx_in, np.int64), c(tile.yd_in, np.int64), c(tile.md_off, np.uint32), c(tile.md, np.uint8),
           c(tile.md_has, np.uint8), c(tile.qn_off, np.uint32), c(tile.qn, np.uint8))
    cap = max(n, 1)
    rep = np.zeros(cap, np.uint32)
    yc = np.zeros(cap, np.float64)
    yx = np.zeros(cap, np.int64)
    yd = np.zeros(cap, np.int32)
    gs = np.zeros(cap, np.int32)
    ge = np.zeros(cap, np.int32)
    rg = np.zeros(max(n, 1), np.int32) if want_rec_group else None
    mo = np.zeros(max(n, 1), np.uint32) if want_merge_order else None
    g = Groups(cap, _ptr(rep), _ptr(yc), _ptr(yx), _ptr(yd), _ptr(gs), _ptr(ge), _ptr(rg), _ptr(mo), 0, 0)
    rc = L.tbo_collapse(C.byref(o), C.byref(i), C.byref(g))
    if rc != 0:
        raise RuntimeError("tbo_collapse failed: %d" % rc)
    m = g.n_groups
    res = dict(n_groups=m, n_passed=g.n_passed, rep=rep[:m].copy(), yc=yc[:m].copy(), yx=yx[:m].copy(),
               yd=yd[:m].copy(), g_start=gs[:m].copy(), g_end=ge[:m].copy())
    if want_rec_group:
        res["rec_group"] = rg[:n].copy()
    if want_merge_order:
        res["merge_order"] = mo[:n].copy()
    return res


def coverage(cin, want_cov=True, want_junc=True, num_samples=0, opt: str = "O2"):
    """Run tbo_coverage on a tiebrush_amd.soa.CovInput; returns a dict of numpy arrays."""
    L = lib(opt)
    n = cin.n_records
    keep = []

    def c(a, dt):
        if a is None:
            return None
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b.ctypes.data

    i = CovIn(n, c(cin.tid, np.int32), c(cin.pos, np.int32), c(cin.flag, np.uint16), c(cin.cig_off, np.uint32),
              c(cin.cig, np.uint32), c(cin.yc, np.float64), c(cin.strand, np.uint8), c(cin.yx, np.int64))
    ncig = int(cin.cig.shape[0])
    cap_iv = (2 * ncig + 2 * n + 16) if want_cov else 0
    cap_j = (ncig + 16) if want_junc else 0
    cap_s = (2 * ncig + 2 * n + 16) if num_samples > 0 else 0
    # the sample track can have one interval per base in the worst case
    if num_samples > 0:
        oplen = (cin.cig >> 4).astype(np.int64)
        cap_s = int(oplen[(cin.cig & 0xF) == 0].sum()) + 16
    iv = [np.zeros(max(cap_iv, 1), np.int32) for _ in range(3)] + [np.zeros(max(cap_iv, 1), np.float64)]
    jv = [np.zeros(max(cap_j, 1), np.int32) for _ in range(3)] + [np.zeros(max(cap_j, 1), np.uint8),
                                                                  np.zeros(max(cap_j, 1), np.float64)]
    sv = [np.zeros(max(cap_s, 1), np.int32) for _ in range(3)] + [np.zeros(max(cap_s, 1), np.int64),
                                                                  np.zeros(max(cap_s, 1), np.float32)]
    o = CovOut(cap_iv, *[_ptr(a) for a in iv], cap_j, *[_ptr(a) for a in jv], cap_s, num_samples,
               *[_ptr(a) for a in sv], 0, 0, 0, 0, 0)
    rc = L.tbo_coverage(C.byref(i), C.byref(o))
    if rc != 0:
        raise RuntimeError("tbo_coverage failed: %d" % rc)
    a, b, s = o.n_intervals, o.n_junctions, o.n_sample
    return dict(n_intervals=a, iv_tid=iv[0][:a].copy(), iv_start=iv[1][:a].copy(), iv_end=iv[2][:a].copy(),
                iv_val=iv[3][:a].copy(), n_junctions=b, j_tid=jv[0][:b].copy(), j_start=jv[1][:b].copy(),
                j_end=jv[2][:b].copy(), j_strand=jv[3][:b].copy(), j_val=jv[4][:b].copy(), n_sample=s,
                s_tid=sv[0][:s].copy(), s_start=sv[1][:s].copy(), s_end=sv[2][:s].copy(), s_count=sv[3][:s].copy(),
                s_heat=sv[4][:s].copy(), n_bases=o.n_bases, span_bases=o.span_bases)


def setup_coordinates(flag, pos, cig):
    L = lib()
    cig = np.ascontiguousarray(cig, dtype=np.uint32)
    s = C.c_int32()
    e = C.c_int32()
    ex = np.zeros(2 * (len(cig) + 1), np.int32)
    n = L.tbo_setup_coordinates(C.c_uint16(flag), C.c_int32(pos), C.c_void_p(cig.ctypes.data), C.c_uint32(len(cig)),
                                C.byref(s), C.byref(e), C.c_void_p(ex.ctypes.data), C.c_uint32(len(cig) + 1))
    return s.value, e.value, ex[:2 * n].reshape(-1, 2).copy()
