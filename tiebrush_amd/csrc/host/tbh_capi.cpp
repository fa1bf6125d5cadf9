// tbh_capi.cpp — libtbh.so: the C ABI of include/tbh_host.h over the host codec (tagging + BGZF deflate of a rank's slice of the
// output, the output header + concatenation of the ranks' parts).  No GPU code.
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/tbh_host.h"
#include "GSam.h"
#include "bgzf.h"
#include "tagwrite.h"
#include "tmerge.h"

namespace {
thread_local std::string g_err;
int fail(const std::string& m) {
  g_err = m;
  return -1;
}
}  // namespace

extern "C" {

int tbh_abi_version(void) { return TBH_ABI_VERSION; }
const char* tbh_last_error(void) { return g_err.c_str(); }

int tbh_tag_deflate_part(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, const double* yc, const int64_t* yx,
                         const int32_t* yd, int level, int threads, const char* out_path) {
  if (!out_path || (n && (!blob || !rec_off || !rec_len || !yc || !yx || !yd))) return fail("tbh_tag_deflate_part: null argument");
  int nt = threads > 0 ? threads : tbh::cpu_budget();
  if (nt < 1) nt = 1;
  if (nt > 128) nt = 128;
  if (n < 4096) nt = 1;
  // slices of 16 K records taken by the workers as they come free; every slice deflates itself into its own run of members
  const uint32_t per = 16384, nsl = n ? (n + per - 1) / per : 0;
  std::vector<std::vector<uint8_t>> runs((size_t)nsl);
  std::atomic<uint32_t> next{0};
  std::atomic<bool> bad{false};
  auto rec = [&](uint32_t g) {
    tbh::RecView v;
    v.p = blob + rec_off[g];
    v.len = rec_len[g];
    return v;
  };
  auto worker = [&]() {
    std::vector<uint8_t> framed;
    tbh::BamRec scratch;
    for (;;) {
      const uint32_t sl = next.fetch_add(1);
      if (sl >= nsl || bad.load()) break;
      const uint32_t g0 = sl * per, g1 = g0 + per < n ? g0 + per : n;
      if (!tbh::tag_and_deflate(g0, g1, rec, yc, yx, yd, level, framed, scratch, runs[(size_t)sl])) bad.store(true);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(worker);
  worker();
  for (auto& x : th) x.join();
  if (bad.load()) return fail("tbh_tag_deflate_part: deflate failed");
  FILE* f = fopen(out_path, "wb");
  if (!f) return fail(std::string("tbh_tag_deflate_part: cannot open ") + out_path);
  bool ok = true;
  for (auto& r : runs) ok = ok && (r.empty() || fwrite(r.data(), 1, r.size(), f) == r.size());
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : fail(std::string("tbh_tag_deflate_part: write failed on ") + out_path);
}

int tbh_write_bam_parts(const char* out_path, const char* version, int cmd_argc, const char* const* cmd_argv, int n_files, const char* const* files,
                        int n_parts, const char* const* parts, int remove_parts) {
  if (!out_path || !version || n_files <= 0 || !files || n_parts < 0 || (n_parts && !parts)) return fail("tbh_write_bam_parts: bad argument");
  // the header exactly as the single-GPU command line builds it: TInputFiles opens every input (header + first record), merges the
  // @SQ tables, lists the samples in @CO lines and adds its @PG line (tmerge.cpp: addSam)
  TInputFiles in;
  std::vector<char*> av;
  for (int i = 0; i < cmd_argc; ++i) av.push_back(const_cast<char*>(cmd_argv[i]));
  in.setup(version, cmd_argc, av.data());
  for (int i = 0; i < n_files; ++i) in.addFile(tbh_realpath(files[i]).c_str());
  in.start();
  // written beside the target and renamed when complete: a part that cannot be read must not leave a well-formed but truncated BAM
  const std::string tmp = std::string(out_path) + ".tmp";
  std::string bad;
  {
    GSamWriter out(tmp.c_str(), in.header(), GSamFile_BAM);
    std::vector<uint8_t> buf((size_t)8 << 20);
    for (int p = 0; p < n_parts && bad.empty(); ++p) {
      FILE* f = fopen(parts[p], "rb");
      if (!f) {
        bad = std::string("tbh_write_bam_parts: cannot open ") + parts[p];
        break;
      }
      size_t got;
      while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) out.write_members(buf.data(), got);
      if (ferror(f)) bad = std::string("tbh_write_bam_parts: read failed on ") + parts[p];
      fclose(f);
    }
  }  // (closing the writer appends the EOF member)
  in.stop();
  if (bad.empty() && rename(tmp.c_str(), out_path) != 0) bad = std::string("tbh_write_bam_parts: cannot rename to ") + out_path;
  if (!bad.empty()) {
    (void)unlink(tmp.c_str());
    return fail(bad);
  }
  if (remove_parts)
    for (int p = 0; p < n_parts; ++p) (void)unlink(parts[p]);
  return 0;
}

int tbh_is_tiebrush(const char* path) {
  if (!path) return -1;
  tbh::BamFile bf;
  std::string err;
  if (!bf.open(path, err, 1)) {
    g_err = err;
    return -1;
  }
  return bf.hdr.is_tiebrush() ? 1 : 0;
}

}  // extern "C"
