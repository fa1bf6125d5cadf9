// tagwrite.h — flushPData's tagging of one output record (/root/reference/src/tiebrush.cpp:506-525, GSam.h:300-305): the
// representative takes YC:f (always), YX:i (always, width by value) and YD:i (only when positive; an older YD is removed), then is
// framed for the BAM stream.  Shared by the single-GPU command line and the multi-rank writer (tbh_capi.cpp).
#pragma once
#include <stdint.h>

#include <vector>

#include "bam.h"

namespace tbh {

// appends `block_size | record + tags` of one group to `o`; `scratch` is a BamRec the caller keeps per thread
void append_tagged(const RecView& v, double yc, int64_t yx, int32_t yd, std::vector<uint8_t>& o, BamRec& scratch);

// the groups [g0, g1) of a run, tagged, framed and deflated into whole BGZF members appended to `members`; rec(g) hands out the
// representative of group g.  false when the deflate fails.
template <class RecOf>
bool tag_and_deflate(uint32_t g0, uint32_t g1, RecOf rec, const double* yc, const int64_t* yx, const int32_t* yd, int level, std::vector<uint8_t>& framed,
                     BamRec& scratch, std::vector<uint8_t>& members);

}  // namespace tbh

#include "bgzf.h"
namespace tbh {
template <class RecOf>
bool tag_and_deflate(uint32_t g0, uint32_t g1, RecOf rec, const double* yc, const int64_t* yx, const int32_t* yd, int level, std::vector<uint8_t>& framed,
                     BamRec& scratch, std::vector<uint8_t>& members) {
  framed.clear();
  for (uint32_t g = g0; g < g1; ++g) append_tagged(rec(g), yc[g], yx[g], yd[g], framed, scratch);
  // BGZF members are independent deflate streams: the slice compresses itself, the writer only appends
  return bgzf_deflate_members(framed.data(), framed.size(), level, members);
}
}  // namespace tbh
