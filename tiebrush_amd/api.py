"""Host-side Python mirror of the hot-path boundary (`include/tbk.h`).

`Context.collapse` stands where the reference's tiebrush main loop stands
(/root/reference/src/tiebrush.cpp:570-592: TInputFiles::next + passes_options + addPData +
flushPData), `Context.coverage` where tiecov's loop stands (tiecov.cpp:435-513).  Arrays may be
numpy (TBK_MEM_HOST: staged by the library) or torch CUDA tensors (TBK_MEM_DEVICE: used in
place, resident in HBM).  torch is plumbing only: device memory + streams.
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from dataclasses import replace

import numpy as np

from . import _lib
from ._lib import TbkError
from .soa import CovInput, SoATile


def _torch():
    import torch
    return torch


def _is_torch(a):
    return a is not None and type(a).__module__.startswith("torch")


def _addr(a, dtype, keep, n=None):
    """Pointer of a numpy array / torch tensor after a dtype + contiguity check."""
    if a is None:
        return None
    if _is_torch(a):
        torch = _torch()
        want = {np.int32: torch.int32, np.uint32: torch.int32, np.uint16: torch.int16, np.uint8: torch.uint8,
                np.float64: torch.float64, np.int64: torch.int64, np.uint64: torch.int64, np.float32: torch.float32}[dtype]
        # unsigned types travel as their signed twins of the same width (torch has no uint32 storage maths we need)
        if a.dtype != want and a.element_size() != np.dtype(dtype).itemsize:
            raise TypeError("tensor dtype %s does not match %s" % (a.dtype, dtype))
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if n is not None and a.numel() < n:
            raise ValueError("tensor too small: %d < %d" % (a.numel(), n))
        keep.append(a)
        return a.data_ptr()
    b = np.ascontiguousarray(a, dtype=dtype)
    if n is not None and b.size < n:
        raise ValueError("array too small: %d < %d" % (b.size, n))
    keep.append(b)
    return b.ctypes.data


def to_device(obj, device="cuda:0"):
    """Copy the per-record arrays of a SoATile / CovInput into HBM (torch tensors)."""
    torch = _torch()

    def mv(a, dt):
        if a is None:
            return None
        b = np.ascontiguousarray(a, dtype=dt)
        sd = {np.uint32: np.int32, np.uint16: np.int16, np.uint64: np.int64}.get(dt, dt)
        return torch.from_numpy(b.view(sd)).to(device)

    if isinstance(obj, SoATile):
        r = replace(obj, tid=mv(obj.tid, np.int32), pos=mv(obj.pos, np.int32), flag=mv(obj.flag, np.uint16),
                       mapq=mv(obj.mapq, np.uint8), strand=mv(obj.strand, np.uint8), nh=mv(obj.nh, np.int32),
                       cig_off=mv(obj.cig_off, np.uint32), cig=mv(obj.cig, np.uint32), yc_in=mv(obj.yc_in, np.float64),
                       yx_in=mv(obj.yx_in, np.int64), yd_in=mv(obj.yd_in, np.int64), md_off=mv(obj.md_off, np.uint32),
                       md=mv(obj.md, np.uint8), md_has=mv(obj.md_has, np.uint8),
                       qname_hash=mv(obj.qname_hash, np.uint64), qn_off=mv(obj.qn_off, np.uint32), qn=mv(obj.qn, np.uint8),
                       prio_hi=mv(obj.prio_hi, np.uint64), prio_lo=mv(obj.prio_lo, np.uint64))
        torch.cuda.synchronize()
        return r
    if isinstance(obj, CovInput):
        r = replace(obj, tid=mv(obj.tid, np.int32), pos=mv(obj.pos, np.int32), flag=mv(obj.flag, np.uint16),
                       cig_off=mv(obj.cig_off, np.uint32), cig=mv(obj.cig, np.uint32), yc=mv(obj.yc, np.float64),
                       strand=mv(obj.strand, np.uint8), yx=mv(obj.yx, np.int64))
        torch.cuda.synchronize()
        return r
    raise TypeError(type(obj))


def _numel(a):
    return int(a.numel()) if _is_torch(a) else int(np.asarray(a).size)


class DeviceTile:
    """a tile living in context-owned device memory (tbk_unpack_tile): goes to collapse() like a SoATile"""

    def __init__(self, struct, n_files, keep):
        self.struct, self.n_files, self._keep = struct, n_files, keep
        self.n_records = int(struct.n_records)


class DeviceCovView:
    """tbk_cov_in view living in context-owned device memory (valid until the next call)."""

    def __init__(self, struct, n_records, n_cigar_ops):
        self.struct = struct
        self.n_records = n_records
        self.n_cigar_ops = n_cigar_ops


class _FollowDebug:
    """The library reads TBK_DEBUG once, when a context is created (tbk_create).  This binding is what the tests drive, and a test changes
    the variable between two calls on one context: before a call goes out, the context is told (tbk_set_debug) when the variable has
    changed since the last one."""

    def __init__(self, lib, ctx):
        self._lib, self._ctx = lib, ctx

    def __getattr__(self, name):
        self._ctx._follow_debug()
        return getattr(self._lib, name)


class Context:
    def __init__(self, device: int = 0):
        lib = _lib.load()
        h = C.c_void_p()
        self._debug_spec = os.environ.get("TBK_DEBUG", "")
        rc = lib.tbk_create(int(device), C.byref(h))
        if rc != 0:
            raise TbkError(rc, "tbk_create(device=%d)" % device)
        self._lib = lib
        self.L = _FollowDebug(lib, self)
        self.h = h
        self.device = device
        self._on_torch_stream = False

    def _follow_debug(self):
        spec = os.environ.get("TBK_DEBUG", "")
        if spec != self._debug_spec and getattr(self, "h", None):
            self._debug_spec = spec
            self._lib.tbk_set_debug(self.h, spec.encode())

    def set_debug(self, spec: str):
        """tbk_set_debug: test hooks / forced paths of this context ("key=value,..."; "" = the defaults)"""
        self._debug_spec = os.environ.get("TBK_DEBUG", "")      # (an explicit setting stays until the variable changes again)
        self._check(self._lib.tbk_set_debug(self.h, spec.encode()), "tbk_set_debug")

    def close(self):
        if getattr(self, "h", None):
            self._lib.tbk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise TbkError(rc, "%s: %s" % (what, (self.L.tbk_last_error(self.h) or b"").decode()))

    def use_torch_stream(self):
        """Launch on torch's current stream so that torch ops and our kernels order correctly."""
        s = _torch().cuda.current_stream(self.device).cuda_stream
        self._check(self.L.tbk_set_stream(self.h, C.c_void_p(s if s else 1)), "tbk_set_stream")   # (0 = the default stream: TBK_STREAM_DEFAULT)
        self._on_torch_stream = True

    def _order_after_torch(self, dev):
        """Device inputs were produced on torch's stream; the context launches on its own stream unless
        use_torch_stream() was called: make the producer visible first."""
        if dev and not self._on_torch_stream:
            _torch().cuda.current_stream(self.device).synchronize()   # (of THIS context's device: the current device is per thread)

    def last_message(self) -> str:
        return (self.L.tbk_last_error(self.h) or b"").decode()

    def stream_ptr(self):
        return self.L.tbk_get_stream(self.h)

    def set_profiling(self, on: bool):
        self._check(self.L.tbk_set_profiling(self.h, 1 if on else 0), "tbk_set_profiling")

    def kernel_times(self):
        arr = (_lib.KernelTime * 128)()
        n = self.L.tbk_kernel_times(self.h, arr, 128)
        return {arr[i].name.decode(): (float(arr[i].ms), int(arr[i].launches)) for i in range(min(n, 128))}

    def _alloc(self, device_mode, n, dtype):
        n = max(int(n), 1)
        if device_mode:
            torch = _torch()
            td = {np.int32: torch.int32, np.uint32: torch.int32, np.uint8: torch.uint8, np.float64: torch.float64,
                  np.int64: torch.int64, np.uint64: torch.int64, np.float32: torch.float32}[dtype]
            return torch.empty(n, dtype=td, device="cuda:%d" % self.device)
        return np.empty(n, dtype=dtype)

    @staticmethod
    def _trim(a, n, dtype):
        if _is_torch(a):
            return a[:n]
        return a[:n].view(dtype) if a.dtype != dtype else a[:n]

    # ---- collapse -----------------------------------------------------------------------------
    def make_opts(self, strategy="cigar", max_nh=2**31 - 1, min_qual=-1, keep_supplementary=False,
                  keep_secondary=False, keep_unmapped=False, collapse_same=False, store_frac=False, flags_mask=0,
                  defer_yd=False, keep_results=False):
        o = _lib.CollapseOpts()
        self.L.tbk_collapse_opts_default(C.byref(o))
        o.strategy = _lib.STRAT[strategy] if isinstance(strategy, str) else int(strategy)
        o.max_nh, o.min_qual, o.flags_mask = int(max_nh), int(min_qual), int(flags_mask)
        o.keep_supplementary, o.keep_secondary = int(keep_supplementary), int(keep_secondary)
        o.keep_unmapped, o.collapse_same, o.store_frac = int(keep_unmapped), int(collapse_same), int(store_frac)
        o.defer_yd = int(defer_yd)
        o.keep_results = int(keep_results)
        return o

    def finish_yd(self):
        """Wait for a deferred YD stage (collapse(..., defer_yd=True)); the `yd` array is complete afterwards."""
        self._check(self.L.tbk_collapse_finish_yd(self.h), "tbk_collapse_finish_yd")

    def unpack_tile(self, pt) -> DeviceTile:
        """tbk_unpack_tile: a soa.PackedTile (numpy, or torch tensors in pinned host memory) -> the tile on the device"""
        keep = []

        def addr(a, dt):
            if _is_torch(a):
                keep.append(a)
                return a.data_ptr() if a.numel() else None
            b = np.ascontiguousarray(a, dtype=dt)
            keep.append(b)
            return b.ctypes.data if b.size else None

        fo = np.ascontiguousarray(pt.file_off, dtype=np.uint32)
        re_, rt = np.ascontiguousarray(pt.tid_run_end, np.uint32), np.ascontiguousarray(pt.tid_run_tid, np.int32)
        keep += [fo, re_, rt]
        p = _lib.PackedIn(pt.n_files, _numel(pt.pos), _numel(pt.cig), len(re_), fo.ctypes.data, re_.ctypes.data, rt.ctypes.data, addr(pt.pos, np.int32),
                          addr(pt.meta, np.uint32), addr(pt.ncig, np.uint8), addr(pt.cig, np.uint32), _numel(pt.nh_esc_idx), _numel(pt.ncig_esc_idx),
                          addr(pt.nh_esc_idx, np.uint32), addr(pt.nh_esc_val, np.int32), addr(pt.ncig_esc_idx, np.uint32),
                          addr(pt.ncig_esc_val, np.uint32))
        s = _lib.SoaIn()
        self._check(self.L.tbk_unpack_tile(self.h, C.byref(p), C.byref(s)), "tbk_unpack_tile")
        return DeviceTile(s, pt.n_files, keep)

    def _soa_struct(self, tile: SoATile, keep):
        if isinstance(tile, DeviceTile):
            keep.append(tile)
            return tile.struct, True, tile.n_records
        dev = _is_torch(tile.tid)
        n = _numel(tile.tid)
        nc = _numel(tile.cig)
        fo = np.ascontiguousarray(tile.file_off, dtype=np.uint32)
        tb = np.ascontiguousarray(tile.tbmerged, dtype=np.uint8)
        keep += [fo, tb]
        s = _lib.SoaIn(
            _lib.TBK_MEM_DEVICE if dev else _lib.TBK_MEM_HOST, tile.n_files, n, nc, fo.ctypes.data, tb.ctypes.data,
            _addr(tile.tid, np.int32, keep, n), _addr(tile.pos, np.int32, keep, n), _addr(tile.flag, np.uint16, keep, n),
            _addr(tile.mapq, np.uint8, keep, n), _addr(tile.strand, np.uint8, keep, n), _addr(tile.nh, np.int32, keep, n),
            _addr(tile.cig_off, np.uint32, keep, n + 1), _addr(tile.cig, np.uint32, keep, nc),
            _addr(tile.yc_in, np.float64, keep, n), _addr(tile.yx_in, np.int64, keep, n),
            _addr(tile.yd_in, np.int64, keep, n), _addr(tile.md_off, np.uint32, keep, n + 1),
            _addr(tile.md, np.uint8, keep), _addr(tile.md_has, np.uint8, keep, n),
            _addr(tile.qname_hash, np.uint64, keep, n), _addr(tile.prio_hi, np.uint64, keep, n),
            _addr(tile.prio_lo, np.uint64, keep, n), _addr(tile.qn_off, np.uint32, keep, n + 1), _addr(tile.qn, np.uint8, keep))
        return s, dev, n

    def collapse(self, tile: SoATile, opts=None, want_coords=True, want_rec_group=False, want_effend=False, out=None,
                 raw=False, want_key=None, cap_groups=None, **kw):
        """Collapse one tile.  Returns a dict (rep, yc, yx, yd[, g_start, g_end, rec_group, g_key], n_groups,
        n_passed) in the reference's output order.  `out` may carry preallocated buffers to reuse.
        want_key (default: device tiles): tbk_groups_out.g_key — groups_to_cov_in then builds the tiecov input of the
        representatives from the keys instead of fetching every representative."""
        o = opts if opts is not None else self.make_opts(**kw)
        keep = []
        s, dev, n = self._soa_struct(tile, keep)
        self._order_after_torch(dev)
        # capacity of the group arrays: one group per record can never overflow (the default); a caller that knows its data
        # collapses (cap_groups, e.g. a quarter of the records) saves the memory, and a call that needs more says how many
        # (TBK_E2BIG with n_groups = the need) and is repeated once with that
        cap = max(n if cap_groups is None else min(int(cap_groups), n), 1)
        bufs = out if out is not None else {}
        if bufs.get("_cap_groups", 0) > cap:
            cap = bufs["_cap_groups"]
        res = self._collapse_once(o, s, dev, n, cap, bufs, keep, want_coords, want_rec_group, want_effend, want_key, raw)
        if isinstance(res, int):                    # TBK_E2BIG: res groups are needed
            bufs["_cap_groups"] = cap = min(n, res + res // 16 + 1)
            res = self._collapse_once(o, s, dev, n, cap, bufs, keep, want_coords, want_rec_group, want_effend, want_key, raw)
            assert not isinstance(res, int)
        return res

    def _collapse_once(self, o, s, dev, n, cap, bufs, keep, want_coords, want_rec_group, want_effend, want_key, raw):

        def buf(name, count, dt, wanted=True):
            if not wanted:
                return None
            if name not in bufs or _numel(bufs[name]) < count:
                bufs[name] = self._alloc(dev, count, dt)
            return bufs[name]

        rep, yc = buf("rep", cap, np.uint32), buf("yc", cap, np.float64)
        yx, yd = buf("yx", cap, np.int64), buf("yd", cap, np.int32)
        gs, ge = buf("g_start", cap, np.int32, want_coords), buf("g_end", cap, np.int32, want_coords)
        rg = buf("rec_group", max(n, 1), np.int32, want_rec_group)
        re_ = buf("rep_effend", cap, np.int32, want_effend)
        want_key = bool(dev) if want_key is None else want_key
        gk = buf("g_key", 2 * cap, np.uint64, want_key)
        g = _lib.GroupsOut(s.mem, cap, _addr(rep, np.uint32, keep), _addr(yc, np.float64, keep),
                           _addr(yx, np.int64, keep), _addr(yd, np.int32, keep), _addr(gs, np.int32, keep),
                           _addr(ge, np.int32, keep), _addr(rg, np.int32, keep), _addr(re_, np.int32, keep),
                           _addr(gk, np.uint64, keep), 0, 0)
        rc = self.L.tbk_collapse_tile(self.h, C.byref(o), C.byref(s), C.byref(g))
        if rc == -4 and int(g.n_groups) > cap:      # TBK_E2BIG on the group arrays: the call reported the need
            return int(g.n_groups)
        self._check(rc, "tbk_collapse_tile")
        m = int(g.n_groups)
        res = dict(n_groups=m, n_passed=int(g.n_passed), _bufs=bufs, _struct=g, _soa=s, _keep=keep)
        if raw:
            return res
        res.update(rep=self._trim(rep, m, np.uint32), yc=yc[:m], yx=yx[:m], yd=yd[:m])
        if want_coords:
            res.update(g_start=gs[:m], g_end=ge[:m])
        if want_rec_group:
            res["rec_group"] = rg[:n]
        if want_effend:
            res["rep_effend"] = re_[:m]
        if want_key:
            res["g_key"] = gk[:2 * m]
        return res

    def groups_to_cov_in(self, collapse_result) -> DeviceCovView:
        """Device-side chain tiebrush -> tiecov (tbk_groups_to_cov_in)."""
        v = _lib.CovIn()
        self._order_after_torch(True)
        self._check(self.L.tbk_groups_to_cov_in(self.h, C.byref(collapse_result["_soa"]),
                                                C.byref(collapse_result["_struct"]), C.byref(v)),
                    "tbk_groups_to_cov_in")
        return DeviceCovView(v, int(v.n_records), int(v.n_cigar_ops))

    # ---- BGZF on the device -------------------------------------------------------------------------
    def bgzf_inflate(self, comp: bytes) -> bytes:
        """tbk_bgzf_inflate: whole BGZF members -> their payload (host bytes in, host bytes out)"""
        src = np.frombuffer(comp, dtype=np.uint8)
        need = C.c_uint64(0)
        out = np.empty(max(1, 4 * len(src) + 65536), dtype=np.uint8)
        rc = self.L.tbk_bgzf_inflate(self.h, src.ctypes.data, len(src), out.ctypes.data, out.size, C.byref(need), _lib.TBK_MEM_HOST)
        if rc == -4:    # TBK_E2BIG: the call reported the size
            out = np.empty(int(need.value), dtype=np.uint8)
            rc = self.L.tbk_bgzf_inflate(self.h, src.ctypes.data, len(src), out.ctypes.data, out.size, C.byref(need), _lib.TBK_MEM_HOST)
        self._check(rc, "tbk_bgzf_inflate")
        return out[:int(need.value)].tobytes()

    def bgzf_deflate(self, payload: bytes, cuts=None) -> bytes:
        """tbk_bgzf_deflate: a byte run -> whole BGZF members (host bytes in, host bytes out); cuts = payload offsets of the member
        boundaries ([0, ..., len(payload)]), default one member per 0xff00 bytes"""
        src = np.frombuffer(payload, dtype=np.uint8)
        need = C.c_uint64(0)
        out = np.empty(len(src) + (len(src) // 0xff00 + 2 + (len(cuts) if cuts is not None else 0)) * 64 + 4096, dtype=np.uint8)
        cp, nm = None, 0
        if cuts is not None:
            cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
            cp, nm = cuts.ctypes.data, len(cuts) - 1
        rc = self.L.tbk_bgzf_deflate(self.h, src.ctypes.data if len(src) else None, len(src), _lib.TBK_MEM_HOST, cp, nm, out.ctypes.data, out.size,
                                     C.byref(need))
        if rc == -4:
            out = np.empty(int(need.value), dtype=np.uint8)
            rc = self.L.tbk_bgzf_deflate(self.h, src.ctypes.data, len(src), _lib.TBK_MEM_HOST, cp, nm, out.ctypes.data, out.size, C.byref(need))
        self._check(rc, "tbk_bgzf_deflate")
        return out[:int(need.value)].tobytes()

    def kept_results(self, first, n, tags_only=False):
        """tbk_kept_results: groups [first, first + n) of the results the last collapse(..., keep_results=True) left on the context"""
        rep, yc, yx, yd = np.empty(n, np.uint32), np.empty(n, np.float64), np.empty(n, np.int64), np.empty(n, np.int32)
        self._check(self.L.tbk_kept_results(self.h, first, n, None if tags_only else rep.ctypes.data, yc.ctypes.data, yx.ctypes.data, yd.ctypes.data),
                    "tbk_kept_results")
        return (None if tags_only else rep), yc, yx, yd

    def bam_encode(self, rep, yc, yx, yd, n_dev=0, host_records=None, kept_first=None, from_ctx=None):
        """tbk_bam_encode: the output records of a collapse -> (run of BGZF members, payload bytes).  rep < n_dev: records of the tile
        bam_decode left on this context; the others come from host_records = {group index: raw record bytes WITHOUT block_size}.
        kept_first: the len(rep) groups from that index on of the results the context kept (keep_results): rep / yc / yx / yd are not handed over;
        from_ctx: the Context that holds them (and the decoded tile), when it is not this one"""
        rep = np.ascontiguousarray(rep, dtype=np.uint32)
        n = len(rep)
        e = _lib.EncIn()
        if kept_first is None:
            yc = np.ascontiguousarray(yc, dtype=np.float64)
            yx = np.ascontiguousarray(yx, dtype=np.int64)
            yd = np.ascontiguousarray(yd, dtype=np.int32)
            e.mem, e.n, e.rep, e.yc, e.yx, e.yd, e.n_dev = _lib.TBK_MEM_HOST, n, rep.ctypes.data, yc.ctypes.data, yx.ctypes.data, yd.ctypes.data, n_dev
        else:
            e.mem, e.n, e.n_dev, e.first = _lib.TBK_MEM_KEPT, n, n_dev, int(kept_first)
        if from_ctx is not None:                  # another Context whose kept results / decoded tile this call reads (tbk_enc_in.from)
            e.from_ctx = from_ctx.h
        keep = []
        if host_records:
            slot = np.zeros(n, dtype=np.uint32)
            blobs, off = [], [0]
            for k, g in enumerate(sorted(host_records)):
                r = host_records[g]
                slot[g] = k
                blobs.append(len(r).to_bytes(4, "little") + r)
                off.append(off[-1] + 4 + len(r))
            blob = np.frombuffer(b"".join(blobs), dtype=np.uint8)
            offa = np.asarray(off, dtype=np.uint64)
            keep = [slot, blob, offa]
            e.n_host, e.host_blob, e.host_off, e.host_slot = len(blobs), blob.ctypes.data, offa.ctypes.data, slot.ctypes.data
        need, pay = C.c_uint64(0), C.c_uint64(0)
        out = np.empty(max(1 << 16, 64 * n), dtype=np.uint8)
        rc = self.L.tbk_bam_encode(self.h, C.byref(e), out.ctypes.data, out.size, C.byref(need), C.byref(pay))
        if rc == -4:
            out = np.empty(int(need.value), dtype=np.uint8)
            rc = self.L.tbk_bam_encode(self.h, C.byref(e), out.ctypes.data, out.size, C.byref(need), C.byref(pay))
        self._check(rc, "tbk_bam_encode")
        del keep
        return out[:int(need.value)].tobytes(), int(pay.value)

    def bam_decode(self, files, tbmerged=None, want_md=False, want_names=False):
        """tbk_bam_decode: list of whole BAM files (bytes) -> (SoaIn struct describing the device-resident tile, file_off).
        The struct can go straight to collapse_struct(); its arrays live in the context until bam_release()."""
        k = len(files)
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
        ptrs = (C.c_void_p * k)(*[b.ctypes.data for b in bufs])
        sizes = np.array([len(b) for b in bufs], dtype=np.uint64)
        tb = np.ascontiguousarray(tbmerged if tbmerged is not None else np.zeros(k, np.uint8), dtype=np.uint8)
        fo = np.zeros(k + 1, dtype=np.uint32)
        s = _lib.SoaIn()
        self._check(self.L.tbk_bam_decode(self.h, k, ptrs, sizes.ctypes.data, tb.ctypes.data, int(want_md), int(want_names), C.byref(s),
                                          fo.ctypes.data), "tbk_bam_decode")
        self._bam_keep = (bufs, tb, fo)
        return s, fo

    def bam_records(self, idx):
        """tbk_bam_records: (bytes of the packed records, offsets [n+1])"""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        n = len(idx)
        off = np.zeros(n + 1, dtype=np.uint64)
        out = np.empty(max(1, 512 * n), dtype=np.uint8)
        rc = self.L.tbk_bam_records(self.h, idx.ctypes.data, n, _lib.TBK_MEM_HOST, out.ctypes.data, out.size, off.ctypes.data)
        if rc == -4:
            out = np.empty(int(off[n]), dtype=np.uint8)
            rc = self.L.tbk_bam_records(self.h, idx.ctypes.data, n, _lib.TBK_MEM_HOST, out.ctypes.data, out.size, off.ctypes.data)
        self._check(rc, "tbk_bam_records")
        return out[:int(off[n])].tobytes(), off

    def bam_release(self):
        self.L.tbk_bam_release(self.h)

    def tile_join(self, dev_struct, host_tile: SoATile):
        """tbk_tile_join: the device tile of bam_decode + a host tile (plain inputs) -> (SoaIn struct of the joined device tile, file_off);
        its arrays live in the context until bam_release()"""
        keep = []
        hs, dev, _ = self._soa_struct(host_tile, keep)
        assert not dev
        k = int(dev_struct.n_files) + int(host_tile.n_files)
        fo = np.zeros(k + 1, dtype=np.uint32)
        tb = np.zeros(k, dtype=np.uint8)
        out = _lib.SoaIn()
        self._check(self.L.tbk_tile_join(self.h, C.byref(dev_struct), C.byref(hs), C.byref(out), fo.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         tb.ctypes.data_as(C.POINTER(C.c_uint8))), "tbk_tile_join")
        self._join_keep = (fo, tb)
        return out, fo

    def reserve_tile(self, n_records, n_cigar_ops):
        self._check(self.L.tbk_reserve_tile(self.h, int(n_records), int(n_cigar_ops)), "tbk_reserve_tile")

    def soa_to_numpy(self, s, fields=("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig")):
        """copy arrays of a device-resident SoaIn (bam_decode) to numpy (tests)"""
        import ctypes
        torch = _torch()
        n, nc = int(s.n_records), int(s.n_cigar_ops)
        spec = {"tid": (np.int32, n), "pos": (np.int32, n), "flag": (np.uint16, n), "mapq": (np.uint8, n), "strand": (np.uint8, n),
                "nh": (np.int32, n), "cig_off": (np.uint32, n + 1), "cig": (np.uint32, nc), "yc_in": (np.float64, n), "yx_in": (np.int64, n),
                "yd_in": (np.int64, n), "md_off": (np.uint32, n + 1), "md_has": (np.uint8, n), "qname_hash": (np.uint64, n),
                "qname_off": (np.uint32, n + 1)}
        out = {}
        hip = ctypes.CDLL("libamdhip64.so")
        for name in fields:
            dt, cnt = spec[name]
            a = np.empty(cnt, dtype=dt)
            ptr = getattr(s, name)
            if cnt and ptr:
                assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(ptr), C.c_size_t(a.nbytes), 2) == 0
            out[name] = a
        for name, offname in (("md", "md_off"), ("qname", "qname_off")):
            if offname in out and getattr(s, name):
                cnt = int(out[offname][-1])
                a = np.empty(cnt, dtype=np.uint8)
                if cnt:
                    assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(getattr(s, name)), C.c_size_t(cnt), 2) == 0
                out[name] = a
        return out

    def collapse_struct(self, s, n_files, **kw):
        """collapse a tile given as a raw SoaIn struct on the device (bam_decode); returns the usual dict (torch tensors)"""
        torch = _torch()
        o = self.make_opts(**kw)
        n = int(s.n_records)
        cap = max(n, 1)
        dev = "cuda:%d" % self.device
        rep = torch.empty(cap, dtype=torch.int32, device=dev)
        yc = torch.empty(cap, dtype=torch.float64, device=dev)
        yx = torch.empty(cap, dtype=torch.int64, device=dev)
        yd = torch.empty(cap, dtype=torch.int32, device=dev)
        gs = torch.empty(cap, dtype=torch.int32, device=dev)
        ge = torch.empty(cap, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        g = _lib.GroupsOut(_lib.TBK_MEM_DEVICE, cap, rep.data_ptr(), yc.data_ptr(), yx.data_ptr(), yd.data_ptr(), gs.data_ptr(), ge.data_ptr(),
                           None, None, None, 0, 0)
        self._check(self.L.tbk_collapse_tile(self.h, C.byref(o), C.byref(s), C.byref(g)), "tbk_collapse_tile")
        m = int(g.n_groups)
        return dict(n_groups=m, n_passed=int(g.n_passed), rep=rep[:m], yc=yc[:m], yx=yx[:m], yd=yd[:m], g_start=gs[:m], g_end=ge[:m])

    # ---- multi-GPU: shuffle, then collapse (device side of tiebrush_amd.dist) -------------------
    def _dev(self):
        return "cuda:%d" % self.device

    def shard_prepare(self, tile: SoATile, out=None, **filters):
        """tbk_shard_prepare: (key, emax, effend, pass) for the device-resident tile."""
        torch = _torch()
        keep = []
        s, dev, n = self._soa_struct(tile, keep)
        assert dev, "shard_prepare needs a device-resident tile"
        bufs = out if out is not None else {}
        for name, dt in (("key", torch.int64), ("emax", torch.int64), ("effend", torch.int32), ("pass", torch.uint8)):
            if name not in bufs or bufs[name].numel() < max(n, 1):
                bufs[name] = torch.empty(max(n, 1), dtype=dt, device=self._dev())
        o = self.make_opts(**filters)
        self._order_after_torch(True)
        self._check(self.L.tbk_shard_prepare(self.h, C.byref(o), C.byref(s), C.c_void_p(bufs["key"].data_ptr()),
                                             C.c_void_p(bufs["emax"].data_ptr()), C.c_void_p(bufs["effend"].data_ptr()),
                                             C.c_void_p(bufs["pass"].data_ptr())), "tbk_shard_prepare")
        return bufs["key"][:n], bufs["emax"][:n], bufs["effend"][:n], bufs["pass"][:n]

    def shard_probe_max(self, file_off, key, emax, cuts, m_out):
        fo = np.ascontiguousarray(file_off, dtype=np.uint32)
        self._order_after_torch(True)
        self._check(self.L.tbk_shard_probe_max(self.h, fo.ctypes.data, len(fo) - 1, C.c_void_p(key.data_ptr()), C.c_void_p(emax.data_ptr()),
                                               C.c_void_p(cuts.data_ptr()), int(cuts.numel()), C.c_void_p(m_out.data_ptr())),
                    "tbk_shard_probe_max")
        return m_out

    def shard_probe_next(self, file_off, key, m, nxt_out):
        fo = np.ascontiguousarray(file_off, dtype=np.uint32)
        self._order_after_torch(True)
        self._check(self.L.tbk_shard_probe_next(self.h, fo.ctypes.data, len(fo) - 1, C.c_void_p(key.data_ptr()), C.c_void_p(m.data_ptr()),
                                                int(m.numel()), C.c_void_p(nxt_out.data_ptr())), "tbk_shard_probe_next")
        return nxt_out

    def shard_pack(self, tile: SoATile, key, passm, effend, cuts, world, out=None):
        """tbk_shard_pack: (rows [n, 6] int32 — the first sum(tab[..., 1]) are filled —, cig words, src_idx [n] int64,
        tab [world, n_files, 5] int64)."""
        torch = _torch()
        keep = []
        s, dev, n = self._soa_struct(tile, keep)
        bufs = out if out is not None else {}
        nc = _numel(tile.cig)
        n_pass = n                                   # upper bound; the table tells how many rows were written
        if "rows" not in bufs or bufs["rows"].shape[0] < max(n_pass, 1):
            bufs["rows"] = torch.empty((max(n_pass, 1), 6), dtype=torch.int32, device=self._dev())
        if "cig" not in bufs or bufs["cig"].numel() < max(nc, 1):
            bufs["cig"] = torch.empty(max(nc, 1), dtype=torch.int32, device=self._dev())
        if "src" not in bufs or bufs["src"].numel() < max(n_pass, 1):
            bufs["src"] = torch.empty(max(n_pass, 1), dtype=torch.int64, device=self._dev())
        tab = torch.empty((world, tile.n_files, 5), dtype=torch.int64, device=self._dev())
        self._order_after_torch(True)
        self._check(self.L.tbk_shard_pack(self.h, C.byref(s), C.c_void_p(key.data_ptr()), C.c_void_p(passm.data_ptr()),
                                          C.c_void_p(effend.data_ptr()), C.c_void_p(cuts.data_ptr()) if world > 1 else None, int(world),
                                          C.c_void_p(bufs["rows"].data_ptr()), C.c_void_p(bufs["cig"].data_ptr()),
                                          C.c_void_p(bufs["src"].data_ptr()), C.c_void_p(tab.data_ptr())), "tbk_shard_pack")
        return bufs["rows"][:n_pass], bufs["cig"], bufs["src"][:n_pass], tab

    def shard_unpack(self, rows, file_off2, out=None):
        """tbk_shard_unpack: the SoA arrays (torch tensors) of the received rows; file_off2 = run boundaries (host)."""
        torch = _torch()
        n2 = int(rows.shape[0])
        fo = np.ascontiguousarray(file_off2, dtype=np.uint32)
        bufs = out if out is not None else {}
        spec = (("tid", torch.int32, n2), ("pos", torch.int32, n2), ("flag", torch.int16, n2), ("mapq", torch.uint8, n2),
                ("strand", torch.uint8, n2), ("nh", torch.int32, n2), ("cig_off", torch.int32, n2 + 1), ("prio_hi", torch.int64, n2),
                ("prio_lo", torch.int64, n2))
        for name, dt, cnt in spec:
            if name not in bufs or bufs[name].numel() < max(cnt, 1):
                bufs[name] = torch.empty(max(cnt, 1), dtype=dt, device=self._dev())
        self._order_after_torch(True)
        rows = rows.contiguous()
        self._check(self.L.tbk_shard_unpack(self.h, C.c_void_p(rows.data_ptr()) if n2 else None, n2, fo.ctypes.data, len(fo) - 1,
                                            *[C.c_void_p(bufs[name].data_ptr()) for name, _, _ in spec]), "tbk_shard_unpack")
        return {name: bufs[name][:cnt] for name, _, cnt in spec}

    # ---- multi-GPU: collapse locally, exchange group partials (device side of tiebrush_amd.dist.partials_collapse) ------
    def partial_keys(self, tile: SoATile, fin, out=None):
        """tbk_partial_keys: (key [ng], emax [ng], not_packable) of the local groups in `fin` (a collapse() result with
        want_coords=True)."""
        torch = _torch()
        ng = int(fin["n_groups"])
        bufs = out if out is not None else {}
        for name in ("pkey", "pemax"):
            if name not in bufs or bufs[name].numel() < max(ng, 1):
                bufs[name] = torch.empty(max(ng, 1), dtype=torch.int64, device=self._dev())
        bad = C.c_uint32(0)
        self._order_after_torch(True)
        self._check(self.L.tbk_partial_keys(self.h, C.byref(fin["_soa"]), C.byref(fin["_struct"]), C.c_void_p(bufs["pkey"].data_ptr()),
                                            C.c_void_p(bufs["pemax"].data_ptr()), C.byref(bad)), "tbk_partial_keys")
        return bufs["pkey"][:ng], bufs["pemax"][:ng], int(bad.value)

    def partial_pack(self, tile: SoATile, fin, key, cuts, world, first_fidx, out=None, opts=None, **kw):
        """tbk_partial_pack: (rows [ng, 12] int32, cig words, tab [world, 3] int64) of the local groups (fin needs want_effend=True
        and a final yd; opts / kw = the options the groups were collapsed with)."""
        torch = _torch()
        o = opts if opts is not None else self.make_opts(**kw)
        ng = int(fin["n_groups"])
        nc = _numel(tile.cig)
        bufs = out if out is not None else {}
        if "prows" not in bufs or bufs["prows"].shape[0] < max(ng, 1):
            bufs["prows"] = torch.empty((max(ng, 1), 12), dtype=torch.int32, device=self._dev())
        if "pcig" not in bufs or bufs["pcig"].numel() < max(nc, 1):
            bufs["pcig"] = torch.empty(max(nc, 1), dtype=torch.int32, device=self._dev())
        tab = torch.empty((world, 3), dtype=torch.int64, device=self._dev())
        self._order_after_torch(True)
        self._check(self.L.tbk_partial_pack(self.h, C.byref(o), C.byref(fin["_soa"]), C.byref(fin["_struct"]), C.c_void_p(key.data_ptr()) if ng else None,
                                            C.c_void_p(cuts.data_ptr()) if world > 1 else None, int(world), int(first_fidx),
                                            C.c_void_p(bufs["prows"].data_ptr()), C.c_void_p(bufs["pcig"].data_ptr()),
                                            C.c_void_p(tab.data_ptr())), "tbk_partial_pack")
        return bufs["prows"][:ng], bufs["pcig"], tab

    # ---- the sender's side without a wait (tbk_partial_stage_*): every call only queues kernels, on torch's current stream, so that
    # the collectives between the stages are ordered with them by the stream alone
    @contextlib.contextmanager
    def _queued_on_torch_stream(self):
        torch = _torch()
        was = self._on_torch_stream
        if not was:
            sp = torch.cuda.current_stream(self.device).cuda_stream
            self.L.tbk_set_stream(self.h, C.c_void_p(sp if sp else 1))      # (0 = the default stream: TBK_STREAM_DEFAULT)
            self._on_torch_stream = True
        try:
            yield
        finally:
            if not was:
                self.L.tbk_set_stream(self.h, None)
                self._on_torch_stream = False

    def partial_stage_keys(self, tile: SoATile, fin, first_fidx, carry=0, out=None):
        """tbk_partial_stage_keys: (key, emax, meta [PARTIAL_META] int64) — device tensors, nothing read back"""
        torch = _torch()
        ng = int(fin["n_groups"])
        bufs = out if out is not None else {}
        for name in ("pkey", "pemax"):
            if name not in bufs or bufs[name].numel() < max(ng, 1):
                bufs[name] = torch.empty(max(ng, 1), dtype=torch.int64, device=self._dev())
        meta = torch.empty(_lib.PARTIAL_META, dtype=torch.int64, device=self._dev())
        with self._queued_on_torch_stream():
            self._check(self.L.tbk_partial_stage_keys(self.h, C.byref(fin["_soa"]), C.byref(fin["_struct"]), C.c_void_p(bufs["pkey"].data_ptr()),
                                                      C.c_void_p(bufs["pemax"].data_ptr()), int(first_fidx), int(carry), C.c_void_p(meta.data_ptr())),
                        "tbk_partial_stage_keys")
        return bufs["pkey"][:ng], bufs["pemax"][:ng], meta

    def partial_stage_cands(self, key, emax, allmeta, world):
        """tbk_partial_stage_cands: (targets [world - 1], cands [world - 1, PARTIAL_CAND]) from the gathered meta rows"""
        torch = _torch()
        nc = max(world - 1, 1)
        targets = torch.empty(nc, dtype=torch.int64, device=self._dev())
        cands = torch.empty((nc, _lib.PARTIAL_CAND), dtype=torch.int64, device=self._dev())
        ng = int(key.numel())
        allmeta = allmeta.contiguous()
        with self._queued_on_torch_stream():
            self._check(self.L.tbk_partial_stage_cands(self.h, C.c_void_p(key.data_ptr()) if ng else None, C.c_void_p(emax.data_ptr()) if ng else None, ng,
                                                       C.c_void_p(allmeta.data_ptr()), int(world), C.c_void_p(targets.data_ptr()), C.c_void_p(cands.data_ptr())),
                        "tbk_partial_stage_cands")
        return targets, cands

    def partial_stage_pack(self, tile: SoATile, fin, key, meta, allcands, targets, world, first_fidx, out=None, opts=None, **kw):
        """tbk_partial_stage_pack: (rows [ng, 12] int32, cig words, tabx [world * 3 + 4] int64, cuts [world - 1]) — all on the device"""
        torch = _torch()
        o = opts if opts is not None else self.make_opts(**kw)
        ng = int(fin["n_groups"])
        nc = _numel(tile.cig)
        bufs = out if out is not None else {}
        if "prows" not in bufs or bufs["prows"].shape[0] < max(ng, 1):
            bufs["prows"] = torch.empty((max(ng, 1), 12), dtype=torch.int32, device=self._dev())
        if "pcig" not in bufs or bufs["pcig"].numel() < max(nc, 1):
            bufs["pcig"] = torch.empty(max(nc, 1), dtype=torch.int32, device=self._dev())
        tabx = torch.empty(world * 3 + 4, dtype=torch.int64, device=self._dev())
        cuts = torch.empty(max(world - 1, 1), dtype=torch.int64, device=self._dev())
        allcands = allcands.contiguous() if allcands is not None else None
        with self._queued_on_torch_stream():
            self._check(self.L.tbk_partial_stage_pack(self.h, C.byref(o), C.byref(fin["_soa"]), C.byref(fin["_struct"]), C.c_void_p(key.data_ptr()) if ng else None,
                                                      C.c_void_p(meta.data_ptr()), C.c_void_p(allcands.data_ptr()) if world > 1 else None,
                                                      C.c_void_p(targets.data_ptr()) if world > 1 else None, int(world), int(first_fidx),
                                                      C.c_void_p(cuts.data_ptr()), C.c_void_p(bufs["prows"].data_ptr()), C.c_void_p(bufs["pcig"].data_ptr()),
                                                      C.c_void_p(tabx.data_ptr())), "tbk_partial_stage_pack")
        return bufs["prows"][:ng], bufs["pcig"], tabx, cuts[:max(world - 1, 0)]

    def partial_unpack(self, rows, out=None):
        """tbk_partial_unpack: the SoA arrays (torch tensors) of the received partial rows."""
        torch = _torch()
        n2 = int(rows.shape[0])
        bufs = out if out is not None else {}
        spec = (("tid", torch.int32, n2), ("pos", torch.int32, n2), ("flag", torch.int16, n2), ("mapq", torch.uint8, n2),
                ("strand", torch.uint8, n2), ("nh", torch.int32, n2), ("cig_off", torch.int32, n2 + 1), ("yc_in", torch.float64, n2),
                ("yx_in", torch.int64, n2), ("yd_in", torch.int64, n2), ("prio_hi", torch.int64, n2), ("prio_lo", torch.int64, n2))
        for name, dt, cnt in spec:
            if name not in bufs or bufs[name].numel() < max(cnt, 1):
                bufs[name] = torch.empty(max(cnt, 1), dtype=dt, device=self._dev())
        self._order_after_torch(True)
        rows = rows.contiguous()
        self._check(self.L.tbk_partial_unpack(self.h, C.c_void_p(rows.data_ptr()) if n2 else None, n2,
                                              *[C.c_void_p(bufs[name].data_ptr()) for name, _, _ in spec]), "tbk_partial_unpack")
        return {name: bufs[name][:cnt] for name, _, cnt in spec}

    def partial_pack_md(self, tile: SoATile, fin, tab, world, rows):
        """tbk_partial_pack_md (-L): (MD bytes of the local groups' representatives in group order [uint8], bytes per destination [world]
        int64) — device tensors; the rows' word 11 is written.  tab: the [world, 3] table of the pack (tabx starts with it)."""
        torch = _torch()
        ng = int(fin["n_groups"])
        nmd = _numel(tile.md)
        md_out = torch.empty(max(nmd, 1), dtype=torch.uint8, device=self._dev())
        md_tab = torch.empty(world, dtype=torch.int64, device=self._dev())
        with self._queued_on_torch_stream():
            self._check(self.L.tbk_partial_pack_md(self.h, C.byref(fin["_soa"]), C.byref(fin["_struct"]), C.c_void_p(tab.data_ptr()), int(world),
                                                   C.c_void_p(rows.data_ptr()) if ng else None, C.c_void_p(md_out.data_ptr()), C.c_void_p(md_tab.data_ptr())),
                        "tbk_partial_pack_md")
        return md_out, md_tab

    def partial_unpack_md(self, rows):
        """tbk_partial_unpack_md: (md_off [n2 + 1] uint32 as int32 tensor, md_has [n2] uint8) of received rows"""
        torch = _torch()
        n2 = int(rows.shape[0])
        md_off = torch.empty(n2 + 1, dtype=torch.int32, device=self._dev())
        md_has = torch.empty(max(n2, 1), dtype=torch.uint8, device=self._dev())
        self._order_after_torch(True)
        rows = rows.contiguous()
        self._check(self.L.tbk_partial_unpack_md(self.h, C.c_void_p(rows.data_ptr()) if n2 else None, n2, C.c_void_p(md_off.data_ptr()),
                                                 C.c_void_p(md_has.data_ptr())), "tbk_partial_unpack_md")
        return md_off, md_has[:n2]

    def partial_reduce(self, rows, run_off, cig, out=None, want_view=True, opts=None, md=None, **kw):
        """tbk_partial_reduce: the owner's merge-reduce of the received partial rows (torch int32 [n2, 12]; run_off = host run
        boundaries [R + 1]; cig = the CIGAR words as received).  Returns the usual collapse dict (rep = ROW index of the
        representative) plus "view" (DeviceCovView of the reduced groups, valid until the next call)."""
        torch = _torch()
        o = opts if opts is not None else self.make_opts(**kw)
        n2 = int(rows.shape[0])
        ro = np.ascontiguousarray(run_off, dtype=np.uint32)
        bufs = out if out is not None else {}
        cap = max(n2, 1)
        spec = (("rep", torch.int32), ("yc", torch.float64), ("yx", torch.int64), ("yd", torch.int32), ("g_start", torch.int32),
                ("g_end", torch.int32))
        for name, dt in spec:
            if name not in bufs or bufs[name].numel() < cap:
                bufs[name] = torch.empty(cap, dtype=dt, device=self._dev())
        rows = rows.contiguous()
        g = _lib.GroupsOut(_lib.TBK_MEM_DEVICE, cap, *[bufs[name].data_ptr() for name, _ in spec], None, None, None, 0, 0)
        v = _lib.CovIn()
        self._order_after_torch(True)
        self._check(self.L.tbk_partial_reduce_md(self.h, C.byref(o), C.c_void_p(rows.data_ptr()) if n2 else None, n2, ro.ctypes.data, len(ro) - 1,
                                                 C.c_void_p(cig.data_ptr()) if cig is not None and cig.numel() else None,
                                                 C.c_void_p(md.data_ptr()) if md is not None and md.numel() else None, C.byref(g),
                                                 C.byref(v) if want_view else None), "tbk_partial_reduce")
        m = int(g.n_groups)
        res = dict(n_groups=m, n_passed=n2, _bufs=bufs, _struct=g, _keep=[rows, cig, ro, md])
        res.update({name: bufs[name][:m] for name, _ in spec})
        if want_view:
            res["view"] = DeviceCovView(v, int(v.n_records), int(v.n_cigar_ops))
        return res

    # ---- coverage -----------------------------------------------------------------------------
    def coverage(self, cin, want_cov=True, want_junc=True, cap_intervals=None, cap_junctions=None, out=None, raw=False):
        keep = []
        if isinstance(cin, DeviceCovView):
            s = cin.struct
            dev, n, nc = True, cin.n_records, cin.n_cigar_ops
        else:
            dev = _is_torch(cin.tid)
            n = _numel(cin.tid)
            nc = _numel(cin.cig)
            s = _lib.CovIn(_lib.TBK_MEM_DEVICE if dev else _lib.TBK_MEM_HOST, n, nc, _addr(cin.tid, np.int32, keep, n),
                           _addr(cin.pos, np.int32, keep, n), _addr(cin.flag, np.uint16, keep, n),
                           _addr(cin.cig_off, np.uint32, keep, n + 1), _addr(cin.cig, np.uint32, keep, nc),
                           _addr(cin.yc, np.float64, keep, n), _addr(cin.strand, np.uint8, keep, n), None)
        self._order_after_torch(dev)
        ci = (cap_intervals if cap_intervals is not None else 2 * nc + 2 * n + 16) if want_cov else 0
        cj = (cap_junctions if cap_junctions is not None else nc + 16) if want_junc else 0
        bufs = out if out is not None else {}

        def buf(name, count, dt):
            if count == 0:
                return None
            if name not in bufs or _numel(bufs[name]) < count:
                bufs[name] = self._alloc(dev, count, dt)
            return bufs[name]

        iv = [buf("iv_tid", ci, np.int32), buf("iv_start", ci, np.int32), buf("iv_end", ci, np.int32),
              buf("iv_val", ci, np.float64)]
        jv = [buf("j_tid", cj, np.int32), buf("j_start", cj, np.int32), buf("j_end", cj, np.int32),
              buf("j_strand", cj, np.uint8), buf("j_val", cj, np.float64)]
        o = _lib.CovOut(s.mem, ci, _addr(iv[0], np.int32, keep), _addr(iv[1], np.int32, keep),
                        _addr(iv[2], np.int32, keep), _addr(iv[3], np.float64, keep), cj, _addr(jv[0], np.int32, keep),
                        _addr(jv[1], np.int32, keep), _addr(jv[2], np.int32, keep), _addr(jv[3], np.uint8, keep),
                        _addr(jv[4], np.float64, keep), 0, 0, 0, 0)
        self._check(self.L.tbk_coverage_tile(self.h, C.byref(s), C.byref(o)), "tbk_coverage_tile")
        a, b = int(o.n_intervals), int(o.n_junctions)
        res = dict(n_intervals=a, n_junctions=b, n_bases=int(o.n_bases), span_bases=int(o.span_bases), _bufs=bufs)
        if raw:
            return res
        if want_cov:
            res.update(iv_tid=iv[0][:a], iv_start=iv[1][:a], iv_end=iv[2][:a], iv_val=iv[3][:a])
        if want_junc:
            res.update(j_tid=jv[0][:b], j_start=jv[1][:b], j_end=jv[2][:b], j_strand=jv[3][:b], j_val=jv[4][:b])
        return res

    def sample(self, cin: CovInput, num_samples: int, cap_intervals=None):
        keep = []
        dev = _is_torch(cin.tid)
        n = _numel(cin.tid)
        nc = _numel(cin.cig)
        s = _lib.CovIn(_lib.TBK_MEM_DEVICE if dev else _lib.TBK_MEM_HOST, n, nc, _addr(cin.tid, np.int32, keep, n),
                       _addr(cin.pos, np.int32, keep, n), _addr(cin.flag, np.uint16, keep, n),
                       _addr(cin.cig_off, np.uint32, keep, n + 1), _addr(cin.cig, np.uint32, keep, nc), None, None,
                       _addr(cin.yx, np.int64, keep, n))
        if cap_intervals is None:
            if dev:
                raise ValueError("cap_intervals is required for device inputs")
            c = np.asarray(cin.cig)
            cap_intervals = int((c >> 4)[(c & 0xF) == 0].sum()) + 16
        ci = cap_intervals
        iv = [self._alloc(dev, ci, np.int32) for _ in range(3)] + [self._alloc(dev, ci, np.int64),
                                                                    self._alloc(dev, ci, np.float32)]
        o = _lib.SampleOut(s.mem, ci, _addr(iv[0], np.int32, keep), _addr(iv[1], np.int32, keep),
                           _addr(iv[2], np.int32, keep), _addr(iv[3], np.int64, keep), _addr(iv[4], np.float32, keep), 0)
        self._check(self.L.tbk_sample_tile(self.h, C.byref(s), int(num_samples), C.byref(o)), "tbk_sample_tile")
        a = int(o.n_intervals)
        return dict(n_sample=a, s_tid=iv[0][:a], s_start=iv[1][:a], s_end=iv[2][:a], s_count=iv[3][:a], s_heat=iv[4][:a])


def to_numpy(d):
    """Bring every tensor value of a result dict back to numpy (tests / writers)."""
    out = {}
    for k, v in d.items():
        if k.startswith("_"):
            continue
        out[k] = v.cpu().numpy() if _is_torch(v) else v
    if "rep" in out and out["rep"].dtype == np.int32:
        out["rep"] = out["rep"].view(np.uint32)
    return out
