// fastload.h — whole-input loader of the accelerated command lines: every input BAM read, inflated and decoded into the SoA tile
// of tbk_soa_in in two parallel passes.  What GSamReader::next() -> sam_read1() does record by record (GSam.h:506-516) for the
// case that the inputs fit in memory; anything else (SAM text, inputs larger than memory, -L / -A extras) keeps the streaming
// path of TInputFiles::next_tile / load_tile.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../../include/tbk.h"

namespace tbh {

// Uninitialised memory for the large, process-lifetime arrays of the command lines: 2 MiB-aligned and advised onto transparent
// huge pages when large (the inflated inputs and the tile are gigabytes: with 4 KiB pages most of a second goes into page faults
// and, at exit, into giving the pages back one by one).  Never freed by its users.
void* big_alloc(size_t bytes);
// Give one block back (a buffer that grows): its pages return to the system and big_release_all forgets it.
void big_free(void* p);
// Give the pages of every large big_alloc block back, `threads` workers side by side (madvise DONTNEED on slices): a process
// that is about to exit would otherwise return its gigabytes in one thread, page by page, while its caller waits.  The blocks
// must not be read afterwards.
void big_release_all(int threads);

struct FastTile {
  // per input
  struct In {
    std::string path;
    const uint8_t* comp = nullptr;  // the file as the page cache holds it (a private read-only mapping, gone once the members are inflated)
    size_t comp_n = 0;
    uint8_t* data = nullptr;     // its inflated stream (malloc, not initialised)
    size_t data_n = 0;
    uint64_t* rec_off = nullptr;    // offset in `data` of every kept record's block_size field
    size_t n_rec = 0;
    uint64_t n_cig = 0;
  };
  std::vector<In> in;
  // the tile (file-major; malloc'd, never value-initialised)
  std::vector<uint32_t> file_off;
  std::vector<uint8_t> tbmerged;
  int32_t *tid = nullptr, *pos = nullptr, *nh = nullptr;
  uint16_t* flag = nullptr;
  uint8_t *mapq = nullptr, *strand = nullptr;
  uint32_t *cig_off = nullptr, *cig = nullptr;
  double* yc_in = nullptr;
  int64_t *yx_in = nullptr, *yd_in = nullptr;
  size_t n = 0, n_cig = 0;
  double ms_read = 0, ms_inflate = 0, ms_index = 0, ms_soa = 0;
  size_t n_fused = 0;  // inputs whose record index came from the members' own walks
  ~FastTile();
  tbk_soa_in view() const;
  // the raw record (without its block_size field) behind tile index g
  const uint8_t* record(uint32_t g, uint32_t* len) const;
  // A caller that walks many records in no memory order (the output side: representatives in coordinate order, from every input) hides
  // the two dependent misses of record() by asking ahead: the index entry of g a few records early, then — the entry there — the record's
  // first line.
  void prefetch_index(uint32_t g) const {
    const size_t f = (size_t)(std::upper_bound(file_off.begin(), file_off.end(), g) - file_off.begin()) - 1;
    __builtin_prefetch(in[f].rec_off + (g - file_off[f]));
  }
  void prefetch_record(uint32_t g) const {
    const size_t f = (size_t)(std::upper_bound(file_off.begin(), file_off.end(), g) - file_off.begin()) - 1;
    __builtin_prefetch(in[f].data + in[f].rec_off[g - file_off[f]]);
  }
};

// Reads paths[f] (BGZF-compressed BAM each) with `threads` workers.  false + err on malformed input; *fits = false (and true
// returned) when the inflated inputs would not fit `mem_budget` bytes — nothing is loaded then.
bool fast_load(const std::vector<std::string>& paths, const std::vector<uint8_t>& tbmerged, int threads, size_t mem_budget, FastTile& t,
               bool* fits, std::string& err);

}  // namespace tbh
