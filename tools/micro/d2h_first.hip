// d2h_first.hip — what the FIRST copy into a freshly page-locked host block costs (hipMemcpyAsync returns late: the call itself blocks).
// Variants: anonymous memory on huge pages + hipHostRegister, untouched / touched before the registration / touched after it;
// hipHostMalloc; plain pageable memory.  For each: the registration, the first D2H (call, wait), the second D2H.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_first tools/micro/d2h_first.hip && ./d2h_first [MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <chrono>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                     \
  do {                                                            \
    hipError_t e_ = (x);                                          \
    if (e_ != hipSuccess) {                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
      exit(1);                                                    \
    }                                                             \
  } while (0)

static void* thp(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, (size_t)2 << 20, (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) != 0) return nullptr;
  (void)madvise(p, n, MADV_HUGEPAGE);
  return p;
}

static void copies(const char* name, void* host, const void* dev, size_t n, hipStream_t st, double t_reg) {
  double t[6];
  for (int k = 0; k < 2; ++k) {
    const double a = now_ms();
    CK(hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st));
    const double b = now_ms();
    CK(hipStreamSynchronize(st));
    t[2 * k] = b - a, t[2 * k + 1] = now_ms() - b;
  }
  printf("%-44s register %6.2f | 1st D2H call %6.2f wait %6.2f | 2nd call %6.2f wait %6.2f   (%.1f MB)\n", name, t_reg, t[0], t[1], t[2], t[3], n / 1e6);
}

int main(int argc, char** argv) {
  const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 20) << 20;
  CK(hipSetDevice(0));
  void* dev;
  CK(hipMalloc(&dev, n));
  CK(hipMemset(dev, 1, n));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {  // warm the copy path
    void* w;
    CK(hipHostMalloc(&w, 1 << 20));
    CK(hipMemcpyAsync(w, dev, 1 << 20, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    CK(hipHostFree(w));
  }
  for (int rep = 0; rep < 2; ++rep) {
    {
      void* p = thp(n);
      double a = now_ms();
      CK(hipHostRegister(p, n, hipHostRegisterDefault));
      copies("huge pages, untouched, registered", p, dev, n, st, now_ms() - a);
      CK(hipHostUnregister(p));
      free(p);
    }
    {
      void* p = thp(n);
      memset(p, 0, n);
      double a = now_ms();
      CK(hipHostRegister(p, n, hipHostRegisterDefault));
      copies("huge pages, touched, then registered", p, dev, n, st, now_ms() - a);
      CK(hipHostUnregister(p));
      free(p);
    }
    {
      void* p = thp(n);
      double a = now_ms();
      CK(hipHostRegister(p, n, hipHostRegisterDefault));
      double r = now_ms() - a;
      a = now_ms();
      memset(p, 0, n);
      printf("   (memset after the registration: %.2f ms)\n", now_ms() - a);
      copies("huge pages, registered, then touched", p, dev, n, st, r);
      CK(hipHostUnregister(p));
      free(p);
    }
    {
      void* p = nullptr;
      double a = now_ms();
      CK(hipHostMalloc(&p, n));
      copies("hipHostMalloc", p, dev, n, st, now_ms() - a);
      CK(hipHostFree(p));
    }
    {
      void* p = thp(n);
      copies("huge pages, pageable (no registration)", p, dev, n, st, 0.0);
      free(p);
    }
    {
      void* p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      double a = now_ms();
      CK(hipHostRegister(p, n, hipHostRegisterDefault));
      copies("4 KB pages (mmap), untouched, registered", p, dev, n, st, now_ms() - a);
      CK(hipHostUnregister(p));
      munmap(p, n);
    }
  }
  return 0;
}
