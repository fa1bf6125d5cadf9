"""ctypes binding of the C ABI in include/tbk.h (libtbk.so, hand-written HIP for gfx950).

There is deliberately no fallback: if the shared library is missing or fails to load the
import raises, so a GPU box can never silently run something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libtbk.so")

_P = C.c_void_p

TBK_MEM_HOST, TBK_MEM_DEVICE, TBK_MEM_KEPT = 0, 1, 2
PARTIAL_BUNDLES = 15
PARTIAL_CAND = 2 + 2 * PARTIAL_BUNDLES      # include/tbk.h: TBK_PARTIAL_CAND
PARTIAL_META = 64 + 4                         # TBK_PARTIAL_META
STRAT = {"cigar": 0, "full": 1, "clip": 2, "exon": 3}
STATUS = {0: "TBK_OK", -1: "TBK_EINVAL", -2: "TBK_ENOMEM", -3: "TBK_EHIP", -4: "TBK_E2BIG", -5: "TBK_EUNSUPPORTED",
          -6: "TBK_EUNSORTED", -7: "TBK_EFATALOP", -8: "TBK_ECOLLISION", -9: "TBK_ENODEVICE"}

# every symbol include/tbk.h declares
SYMBOLS = ["tbk_abi_version", "tbk_create", "tbk_destroy", "tbk_strerror", "tbk_last_error", "tbk_set_stream",
           "tbk_get_stream", "tbk_set_profiling", "tbk_set_debug", "tbk_kernel_times", "tbk_host_alloc", "tbk_host_free",
           "tbk_collapse_opts_default", "tbk_collapse_tile", "tbk_collapse_finish_yd", "tbk_coverage_tile", "tbk_sample_tile",
           "tbk_groups_to_cov_in", "tbk_bgzf_inflate", "tbk_bam_decode", "tbk_bam_records", "tbk_bam_release", "tbk_shard_prepare", "tbk_shard_probe_max", "tbk_shard_probe_next",
           "tbk_shard_pack", "tbk_shard_unpack", "tbk_partial_keys", "tbk_partial_pack", "tbk_partial_unpack", "tbk_partial_reduce", "tbk_unpack_tile", "tbk_tile_join", "tbk_reserve_tile", "tbk_bgzf_deflate", "tbk_bam_encode", "tbk_kept_results", "tbk_warmup", "tbk_partial_stage_keys", "tbk_partial_stage_cands", "tbk_partial_stage_pack",
           "tbk_partial_pack_md", "tbk_partial_unpack_md", "tbk_partial_reduce_md"]


class CollapseOpts(C.Structure):
    _fields_ = [("strategy", C.c_int32), ("max_nh", C.c_int32), ("min_qual", C.c_int32), ("flags_mask", C.c_uint32),
                ("keep_supplementary", C.c_uint8), ("keep_secondary", C.c_uint8), ("keep_unmapped", C.c_uint8),
                ("collapse_same", C.c_uint8), ("store_frac", C.c_uint8), ("defer_yd", C.c_uint8), ("keep_results", C.c_uint8), ("reserved", C.c_uint8 * 1)]


class SoaIn(C.Structure):
    _fields_ = [("mem", C.c_int32), ("n_files", C.c_uint32), ("n_records", C.c_uint32), ("n_cigar_ops", C.c_uint32),
                ("file_off", _P), ("tbmerged", _P), ("tid", _P), ("pos", _P), ("flag", _P), ("mapq", _P),
                ("strand", _P), ("nh", _P), ("cig_off", _P), ("cig", _P), ("yc_in", _P), ("yx_in", _P), ("yd_in", _P),
                ("md_off", _P), ("md", _P), ("md_has", _P), ("qname_hash", _P), ("prio_hi", _P), ("prio_lo", _P),
                ("qname_off", _P), ("qname", _P)]


class PackedIn(C.Structure):
    _fields_ = [("n_files", C.c_uint32), ("n_records", C.c_uint32), ("n_cigar_ops", C.c_uint32), ("n_tid_runs", C.c_uint32), ("file_off", _P),
                ("tid_run_end", _P), ("tid_run_tid", _P), ("pos", _P), ("meta", _P), ("ncig", _P), ("cig", _P), ("n_nh_esc", C.c_uint32),
                ("n_ncig_esc", C.c_uint32), ("nh_esc_idx", _P), ("nh_esc_val", _P), ("ncig_esc_idx", _P), ("ncig_esc_val", _P)]


class GroupsOut(C.Structure):
    _fields_ = [("mem", C.c_int32), ("cap_groups", C.c_uint32), ("rep", _P), ("yc", _P), ("yx", _P), ("yd", _P),
                ("g_start", _P), ("g_end", _P), ("rec_group", _P), ("rep_effend", _P), ("g_key", _P), ("n_groups", C.c_uint32),
                ("n_passed", C.c_uint32)]


class CovIn(C.Structure):
    _fields_ = [("mem", C.c_int32), ("n_records", C.c_uint32), ("n_cigar_ops", C.c_uint32), ("tid", _P), ("pos", _P),
                ("flag", _P), ("cig_off", _P), ("cig", _P), ("yc", _P), ("strand", _P), ("yx", _P)]


class CovOut(C.Structure):
    _fields_ = [("mem", C.c_int32), ("cap_intervals", C.c_uint32), ("iv_tid", _P), ("iv_start", _P), ("iv_end", _P),
                ("iv_val", _P), ("cap_junctions", C.c_uint32), ("j_tid", _P), ("j_start", _P), ("j_end", _P),
                ("j_strand", _P), ("j_val", _P), ("n_intervals", C.c_uint32), ("n_junctions", C.c_uint32),
                ("n_bases", C.c_uint64), ("span_bases", C.c_uint64)]


class SampleOut(C.Structure):
    _fields_ = [("mem", C.c_int32), ("cap_intervals", C.c_uint32), ("iv_tid", _P), ("iv_start", _P), ("iv_end", _P),
                ("iv_count", _P), ("iv_heat", _P), ("n_intervals", C.c_uint32)]


class EncIn(C.Structure):
    _fields_ = [("mem", C.c_int32), ("n", C.c_uint32), ("rep", _P), ("yc", _P), ("yx", _P), ("yd", _P), ("n_dev", C.c_uint32), ("n_host", C.c_uint32),
                ("host_blob", _P), ("host_off", _P), ("host_slot", _P), ("first", C.c_uint32), ("from_ctx", _P)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_uint32)]


_lib = None


def load():
    """Load libtbk.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64; load it
    # FIRST so that libtbk.so's NEEDED libamdhip64.so.7 binds to that already-loaded copy instead of
    # pulling in /opt/rocm's second runtime (two runtimes in one process cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:  # standalone use without torch: the system ROCm runtime is the only one
        pass
    L = C.CDLL(LIB_PATH)
    L.tbk_abi_version.restype = C.c_int
    L.tbk_create.argtypes = [C.c_int, C.POINTER(_P)]
    L.tbk_create.restype = C.c_int
    L.tbk_destroy.argtypes = [_P]
    L.tbk_destroy.restype = None
    L.tbk_strerror.argtypes = [C.c_int]
    L.tbk_strerror.restype = C.c_char_p
    L.tbk_last_error.argtypes = [_P]
    L.tbk_last_error.restype = C.c_char_p
    L.tbk_set_stream.argtypes = [_P, _P]
    L.tbk_get_stream.argtypes = [_P]
    L.tbk_get_stream.restype = _P
    L.tbk_set_profiling.argtypes = [_P, C.c_int]
    L.tbk_set_debug.argtypes = [_P, C.c_char_p]
    L.tbk_kernel_times.argtypes = [_P, C.POINTER(KernelTime), C.c_int]
    L.tbk_host_alloc.argtypes = [C.c_size_t, C.POINTER(_P)]
    L.tbk_host_free.argtypes = [_P]
    L.tbk_host_free.restype = None
    L.tbk_collapse_opts_default.argtypes = [C.POINTER(CollapseOpts)]
    L.tbk_collapse_opts_default.restype = None
    L.tbk_collapse_tile.argtypes = [_P, C.POINTER(CollapseOpts), C.POINTER(SoaIn), C.POINTER(GroupsOut)]
    L.tbk_collapse_finish_yd.argtypes = [_P]
    L.tbk_coverage_tile.argtypes = [_P, C.POINTER(CovIn), C.POINTER(CovOut)]
    L.tbk_sample_tile.argtypes = [_P, C.POINTER(CovIn), C.c_int32, C.POINTER(SampleOut)]
    L.tbk_groups_to_cov_in.argtypes = [_P, C.POINTER(SoaIn), C.POINTER(GroupsOut), C.POINTER(CovIn)]
    L.tbk_bgzf_inflate.argtypes = [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
    L.tbk_bam_decode.argtypes = [_P, C.c_uint32, _P, _P, _P, C.c_int, C.c_int, C.POINTER(SoaIn), _P]
    L.tbk_bam_records.argtypes = [_P, _P, C.c_uint32, C.c_int, _P, C.c_uint64, _P]
    L.tbk_bam_release.argtypes = [_P]
    L.tbk_bam_release.restype = None
    L.tbk_shard_prepare.argtypes = [_P, C.POINTER(CollapseOpts), C.POINTER(SoaIn), _P, _P, _P, _P]
    L.tbk_shard_probe_max.argtypes = [_P, _P, C.c_uint32, _P, _P, _P, C.c_uint32, _P]
    L.tbk_shard_probe_next.argtypes = [_P, _P, C.c_uint32, _P, _P, C.c_uint32, _P]
    L.tbk_shard_pack.argtypes = [_P, C.POINTER(SoaIn), _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P]
    L.tbk_shard_unpack.argtypes = [_P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P]
    L.tbk_partial_keys.argtypes = [_P, C.POINTER(SoaIn), C.POINTER(GroupsOut), _P, _P, C.POINTER(C.c_uint32)]
    L.tbk_partial_pack.argtypes = [_P, C.POINTER(CollapseOpts), C.POINTER(SoaIn), C.POINTER(GroupsOut), _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P]
    L.tbk_partial_reduce.argtypes = [_P, C.POINTER(CollapseOpts), _P, C.c_uint32, _P, C.c_uint32, _P, C.POINTER(GroupsOut), C.POINTER(CovIn)]
    L.tbk_partial_unpack.argtypes = [_P, _P, C.c_uint32] + [_P] * 12
    L.tbk_tile_join.argtypes = [_P, C.POINTER(SoaIn), C.POINTER(SoaIn), C.POINTER(SoaIn), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    L.tbk_tile_join.restype = C.c_int
    L.tbk_reserve_tile.argtypes = [_P, C.c_uint64, C.c_uint64]
    L.tbk_reserve_tile.restype = C.c_int
    L.tbk_unpack_tile.argtypes = [_P, C.POINTER(PackedIn), C.POINTER(SoaIn)]
    L.tbk_partial_stage_keys.argtypes = [_P, C.POINTER(SoaIn), C.POINTER(GroupsOut), _P, _P, C.c_uint32, C.c_int64, _P]
    L.tbk_partial_stage_cands.argtypes = [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P]
    L.tbk_partial_stage_pack.argtypes = [_P, C.POINTER(CollapseOpts), C.POINTER(SoaIn), C.POINTER(GroupsOut), _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P]
    L.tbk_partial_pack_md.argtypes = [_P, C.POINTER(SoaIn), C.POINTER(GroupsOut), _P, C.c_uint32, _P, _P, _P]
    L.tbk_partial_unpack_md.argtypes = [_P, _P, C.c_uint32, _P, _P]
    L.tbk_partial_reduce_md.argtypes = [_P, C.POINTER(CollapseOpts), _P, C.c_uint32, _P, C.c_uint32, _P, _P, C.POINTER(GroupsOut), C.POINTER(CovIn)]
    L.tbk_bgzf_deflate.argtypes = [_P, _P, C.c_uint64, C.c_int, _P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]
    L.tbk_bgzf_deflate.restype = C.c_int
    L.tbk_bam_encode.argtypes = [_P, C.POINTER(EncIn), _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tbk_bam_encode.restype = C.c_int
    L.tbk_kept_results.argtypes = [_P, C.c_uint32, C.c_uint32, _P, _P, _P, _P]
    L.tbk_kept_results.restype = C.c_int
    L.tbk_warmup.argtypes = [_P]
    L.tbk_warmup.restype = C.c_int
    _lib = L
    return L


# (TBK_HOST_LIB: the sanitizer builds of tools/san_check.sh)
HOST_LIB_PATH = os.environ.get("TBK_HOST_LIB") or os.path.join(_HERE, "_build", "libtbh.so")
HOST_SYMBOLS = ["tbh_abi_version", "tbh_last_error", "tbh_tag_deflate_part", "tbh_write_bam_parts", "tbh_is_tiebrush"]   # include/tbh_host.h
_host = None


def load_host():
    """libtbh.so (include/tbh_host.h): the host-side write path of the multi-rank command line; no GPU code"""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % HOST_LIB_PATH)
    H = C.CDLL(HOST_LIB_PATH)
    H.tbh_abi_version.restype = C.c_int
    H.tbh_last_error.restype = C.c_char_p
    H.tbh_tag_deflate_part.argtypes = [_P, _P, _P, C.c_uint32, _P, _P, _P, C.c_int, C.c_int, C.c_char_p]
    H.tbh_tag_deflate_part.restype = C.c_int
    H.tbh_write_bam_parts.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                      C.POINTER(C.c_char_p), C.c_int]
    H.tbh_write_bam_parts.restype = C.c_int
    H.tbh_is_tiebrush.argtypes = [C.c_char_p]
    H.tbh_is_tiebrush.restype = C.c_int
    _host = H
    return H


class TbkError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("%s (%d) %s" % (STATUS.get(status, "?"), status, detail))
