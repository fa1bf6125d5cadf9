// tagwrite.cpp — see tagwrite.h
#include "tagwrite.h"

#include <string.h>

namespace tbh {

void append_tagged(const RecView& v, double yc, int64_t yx, int32_t yd, std::vector<uint8_t>& o, BamRec& rr) {
  // A record that carries none of the three tags yet (every record of a plain BAM input) takes them appended in the order the
  // reference sets them — YC:f, YX by value width, YD when > 0 (bam_aux_update_* appends a missing tag; GSam.h:300-305,
  // tiebrush.cpp:506-525): written straight into the run.  Anything else goes through BamRec.
  bool fresh = yx >= 0 && yx <= (int64_t)UINT32_MAX;
  for (const uint8_t* a = v.aux_begin(); fresh && a + 3 <= v.aux_end();) {
    const size_t sz = aux_field_size(a, v.aux_end());
    if (!sz) break;
    if (a[0] == 'Y' && (a[1] == 'C' || a[1] == 'X' || a[1] == 'D')) fresh = false;
    a += sz;
  }
  if (fresh) {
    uint8_t tg[24];
    size_t tn = 0;
    const float ycf = (float)yc;
    tg[tn++] = 'Y', tg[tn++] = 'C', tg[tn++] = 'f';
    memcpy(tg + tn, &ycf, 4);
    tn += 4;
    auto put_int = [&](char t1, uint32_t val) {  // bam_aux_update_int of a missing tag: C < 255, S < 65535, else I
      tg[tn++] = 'Y', tg[tn++] = (uint8_t)t1;
      const int w = val < UINT8_MAX ? 1 : (val < UINT16_MAX ? 2 : 4);
      tg[tn++] = (uint8_t)(w == 1 ? 'C' : (w == 2 ? 'S' : 'I'));
      for (int q = 0; q < w; ++q) tg[tn++] = (uint8_t)(val >> (8 * q));
    };
    put_int('X', (uint32_t)yx);
    if (yd > 0) put_int('D', (uint32_t)yd);
    const uint32_t bs = v.len + (uint32_t)tn;
    const size_t at = o.size();
    o.resize(at + 4 + bs);
    memcpy(o.data() + at, &bs, 4);
    memcpy(o.data() + at + 4, v.p, v.len);
    memcpy(o.data() + at + 4 + v.len, tg, tn);
    return;
  }
  rr.d.assign(v.p, v.p + v.len);
  rr.update_float("YC", (float)yc);
  rr.update_int("YX", yx);
  if (yd > 0)
    rr.update_int("YD", yd);
  else
    rr.del("YD");
  const uint32_t bs = (uint32_t)rr.d.size();
  const uint8_t le[4] = {(uint8_t)bs, (uint8_t)(bs >> 8), (uint8_t)(bs >> 16), (uint8_t)(bs >> 24)};
  o.insert(o.end(), le, le + 4);
  o.insert(o.end(), rr.d.begin(), rr.d.end());
}

}  // namespace tbh
