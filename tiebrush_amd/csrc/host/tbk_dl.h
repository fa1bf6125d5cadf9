// tbk_dl.h — the entry points of include/tbk.h that the tiebrush command line uses, bound at run time.
// libtbk.so pulls in the HIP runtime (tens of megabytes of shared objects to map and relocate) and tbk_create brings the device
// up: ~0.3 s together.  Bound with dlopen on a helper thread, both happen WHILE the inputs are read and inflated instead of
// before main() starts.  The library is looked up next to the executable (the in-tree build), then on the loader's path.
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <string.h>
#include <unistd.h>

#include <string>

#include "../../../include/tbk.h"

struct TbkApi {
  decltype(&tbk_abi_version) abi_version = nullptr;
  decltype(&tbk_create) create = nullptr;
  decltype(&tbk_destroy) destroy = nullptr;
  decltype(&tbk_strerror) strerror_ = nullptr;
  decltype(&tbk_last_error) last_error = nullptr;
  decltype(&tbk_collapse_tile) collapse_tile = nullptr;
  decltype(&tbk_bam_decode) bam_decode = nullptr;
  decltype(&tbk_bam_records) bam_records = nullptr;
  decltype(&tbk_bam_release) bam_release = nullptr;
  decltype(&tbk_tile_join) tile_join = nullptr;
  decltype(&tbk_reserve_tile) reserve_tile = nullptr;
  decltype(&tbk_bam_encode) bam_encode = nullptr;
  decltype(&tbk_kept_results) kept_results = nullptr;
  decltype(&tbk_warmup) warmup = nullptr;
  decltype(&tbk_host_alloc) host_alloc = nullptr;
  decltype(&tbk_host_free) host_free = nullptr;
  decltype(&tbk_set_profiling) set_profiling = nullptr;
  decltype(&tbk_kernel_times) kernel_times = nullptr;
  std::string error;

  bool load() {
    std::string dir;
    char exe[PATH_MAX];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    if (n > 0) {
      exe[n] = 0;
      const char* slash = strrchr(exe, '/');
      if (slash) dir.assign(exe, (size_t)(slash - exe) + 1);
    }
    void* h = dlopen((dir + "libtbk.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libtbk.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      error = dlerror();
      return false;
    }
#define TBK_BIND(field, sym)                         \
  field = (decltype(field))dlsym(h, #sym);           \
  if (!field) {                                      \
    error = "libtbk.so lacks " #sym;                 \
    return false;                                    \
  }
    TBK_BIND(abi_version, tbk_abi_version)
    TBK_BIND(create, tbk_create)
    TBK_BIND(destroy, tbk_destroy)
    TBK_BIND(strerror_, tbk_strerror)
    TBK_BIND(last_error, tbk_last_error)
    TBK_BIND(collapse_tile, tbk_collapse_tile)
    TBK_BIND(bam_decode, tbk_bam_decode)
    TBK_BIND(bam_records, tbk_bam_records)
    TBK_BIND(bam_release, tbk_bam_release)
    TBK_BIND(tile_join, tbk_tile_join)
    TBK_BIND(reserve_tile, tbk_reserve_tile)
    TBK_BIND(bam_encode, tbk_bam_encode)
    TBK_BIND(kept_results, tbk_kept_results)
    TBK_BIND(warmup, tbk_warmup)
    TBK_BIND(host_alloc, tbk_host_alloc)
    TBK_BIND(host_free, tbk_host_free)
    TBK_BIND(set_profiling, tbk_set_profiling)
    TBK_BIND(kernel_times, tbk_kernel_times)
#undef TBK_BIND
    if (abi_version() != TBK_ABI_VERSION) {
      error = "libtbk.so has another ABI version";
      return false;
    }
    return true;
  }
};
