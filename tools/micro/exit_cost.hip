// exit_cost — what a HIP process pays AFTER _exit (the parent's waitpid returns later than the child's last instruction): run by
// tools/micro/exit_cost.py with different amounts of device / pinned memory still allocated at the end.
//   exit_cost <device MB> <pinned MB> <touch: 0|1> [mode]   mode 0: hipHostMalloc, left at exit; 1: hipHostMalloc + hipHostFree before exit;
//   2: 2 MiB-aligned anonymous memory advised onto huge pages + hipHostRegister, left; 3: the same + hipHostUnregister before exit; 4: ... + munmap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
__global__ void touch_k(char* p, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096;
  if (i < n) p[i] = 1;
}
static double now() {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + ts.tv_nsec * 1e-9;
}
int main(int argc, char** argv) {
  const size_t dev_mb = argc > 1 ? (size_t)atol(argv[1]) : 0, pin_mb = argc > 2 ? (size_t)atol(argv[2]) : 0;
  const int touch = argc > 3 ? atoi(argv[3]) : 1;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const double t0 = now();
  char *d = nullptr, *h = nullptr;
  if (hipSetDevice(0) != hipSuccess) return 2;
  if (dev_mb && hipMalloc((void**)&d, dev_mb << 20) != hipSuccess) return 3;
  double t_pin = now();
  if (pin_mb && mode < 2 && hipHostMalloc((void**)&h, pin_mb << 20, hipHostMallocDefault) != hipSuccess) return 4;
  if (pin_mb && mode >= 2) {
    h = (char*)mmap(nullptr, (pin_mb << 20) + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (h == MAP_FAILED) return 5;
    h = (char*)(((uintptr_t)h + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    (void)madvise(h, pin_mb << 20, MADV_HUGEPAGE);
    for (size_t i = 0; i < (pin_mb << 20); i += 4096) h[i] = 1;
    if (hipHostRegister(h, pin_mb << 20, hipHostRegisterDefault) != hipSuccess) return 6;
  }
  t_pin = now() - t_pin;
  if (d && touch) {
    const size_t n = dev_mb << 20;
    touch_k<<<(unsigned)((n / 4096 + 255) / 256), 256>>>(d, n);
  }
  (void)hipDeviceSynchronize();
  if (h && touch)
    for (size_t i = 0; i < (pin_mb << 20); i += 4096) h[i] = 1;
  double t_free = now();
  if (h && mode == 1) (void)hipHostFree(h);
  int urc = 0;
  if (h && mode >= 3) urc = (int)hipHostUnregister(h);
  const double t_unreg = now() - t_free;
  if (h && mode == 4) (void)munmap(h, pin_mb << 20);
  t_free = now() - t_free;
  fprintf(stderr, "pin %.3f s, free %.3f s (unregister rc %d, %.3f s)\n", t_pin, t_free, urc, t_unreg);
  printf("%.6f %.6f\n", now() - t0, now());
  fflush(stdout);
  _exit(0);
}
