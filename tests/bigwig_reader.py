"""A small bigWig reader for the tests (independent of tiebrush_amd/csrc/host/bigwig.cpp): header, chromosome B+ tree,
R-tree walk, bedGraph / variableStep / fixedStep sections, zoom levels, total summary.  Format: Kent et al. 2010 (bbi files)."""
import struct
import zlib


class BigWig:
    def __init__(self, path):
        self.d = open(path, "rb").read()
        d = self.d
        (magic, self.version, self.n_zoom, self.tree_off, self.data_off, self.index_off, fc, dfc, asql, self.summary_off,
         self.ubuf, ext) = struct.unpack_from("<IHHQQQHHQQIQ", d, 0)
        assert magic == 0x888FFC26, hex(magic)
        assert struct.unpack_from("<I", d, len(d) - 4)[0] == 0x888FFC26
        self.zoom_headers = [struct.unpack_from("<IIQQ", d, 64 + 24 * i) for i in range(self.n_zoom)]
        self.summary = struct.unpack_from("<Qdddd", d, self.summary_off)   # bases covered, min, max, sum, sum of squares
        self.chroms = self._chrom_tree()

    def _chrom_tree(self):
        d = self.d
        magic, block, key, val, count, _ = struct.unpack_from("<IIIIQQ", d, self.tree_off)
        assert magic == 0x78CA8C91 and val == 8
        out = {}

        def walk(off):
            leaf, _, n = struct.unpack_from("<BBH", d, off)
            p = off + 4
            for _ in range(n):
                k = d[p:p + key].rstrip(b"\0").decode()
                if leaf:
                    cid, size = struct.unpack_from("<II", d, p + key)
                    out[cid] = (k, size)
                else:
                    walk(struct.unpack_from("<Q", d, p + key)[0])
                p += key + 8
        walk(self.tree_off + 32)
        assert len(out) == count
        return out

    def _blocks(self, index_off):
        d = self.d
        magic, block, count, c0, s0, c1, e1, end_off, per_slot, _ = struct.unpack_from("<IIQIIIIQII", d, index_off)
        assert magic == 0x2468ACE0
        out = []

        def walk(off):
            leaf, _, n = struct.unpack_from("<BBH", d, off)
            p = off + 4
            for _ in range(n):
                if leaf:
                    a, b, c, e, o, sz = struct.unpack_from("<IIIIQQ", d, p)
                    out.append((a, b, c, e, o, sz)); p += 32
                else:
                    a, b, c, e, o = struct.unpack_from("<IIIIQ", d, p)
                    walk(o); p += 24
        if count:
            walk(index_off + 48)
        assert len(out) == count
        return out

    def _payload(self, off, size):
        raw = self.d[off:off + size]
        return zlib.decompress(raw) if self.ubuf else raw

    def intervals(self):
        """-> list of (chrom name, start, end, value) in file order"""
        n_sections = struct.unpack_from("<Q", self.d, self.data_off)[0]
        blocks = self._blocks(self.index_off)
        assert len(blocks) == n_sections
        out = []
        for a, b, c, e, o, sz in blocks:
            s = self._payload(o, sz)
            assert len(s) <= self.ubuf
            cid, cs, ce, step, span, typ, _, n = struct.unpack_from("<IIIIIBBH", s, 0)
            assert (cid, cs, ce) == (a, b, e) and c == a
            name = self.chroms[cid][0]
            p = 24
            for i in range(n):
                if typ == 1:
                    st, en, v = struct.unpack_from("<IIf", s, p); p += 12
                elif typ == 2:
                    st, v = struct.unpack_from("<If", s, p); en = st + span; p += 8
                else:
                    v = struct.unpack_from("<f", s, p)[0]; st = cs + i * step; en = st + span; p += 4
                out.append((name, st, en, v))
        return out

    def zoom(self, level):
        """-> (reduction, list of (chrom id, start, end, valid, min, max, sum, sumsq))"""
        red, _, data_off, index_off = self.zoom_headers[level]
        n = struct.unpack_from("<I", self.d, data_off)[0]
        out = []
        for a, b, c, e, o, sz in self._blocks(index_off):
            s = self._payload(o, sz)
            for i in range(len(s) // 32):
                out.append(struct.unpack_from("<IIIIffff", s, 32 * i))
        assert len(out) == n
        return red, out
