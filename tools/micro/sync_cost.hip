// micro-benchmark: what one "kernel -> read a device scalar on the host" round trip costs on this box, three ways
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void bump(unsigned long long* d, unsigned long long* h) {
  if (threadIdx.x == 0) {
    unsigned long long v = *d + 1;
    *d = v;
    if (h) *h = v;
  }
}
__global__ void work(float* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long *d, *hp, *hm;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  hipHostMalloc(&hp, 64, hipHostMallocDefault);
  hipHostMalloc(&hm, 64, hipHostMallocMapped);
  unsigned long long* hm_dev;
  hipHostGetDevicePointer((void**)&hm_dev, hm, 0);
  float* x;
  int n = 1 << 21;
  hipMalloc(&x, n * 4);
  hipMemset(x, 0, n * 4);
  const int R = 2000;
  auto t = [&](const char* name, auto f) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < R; ++i) f();
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    printf("%-46s %8.2f us/iter\n", name, us);
  };
  t("kernel only (async, no sync)", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); });
  t("2 kernels (async)", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); bump<<<1, 64, 0, s>>>(d, nullptr); });
  t("kernel + sync", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipStreamSynchronize(s); });
  t("kernel + memcpyAsync D2H + sync", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipMemcpyAsync(hp, d, 64, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
  t("kernel + 2 memcpyAsync D2H + sync", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipMemcpyAsync(hp, d, 32, hipMemcpyDeviceToHost, s); hipMemcpyAsync(hp + 4, d + 4, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
  t("kernel writing mapped host word + sync", [&] { bump<<<1, 64, 0, s>>>(d, hm_dev); hipStreamSynchronize(s); });
  t("kernel + spin on hipStreamQuery", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); while (hipStreamQuery(s) == hipErrorNotReady) {} });
  t("kernel + memcpyAsync D2H + spin hipStreamQuery", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipMemcpyAsync(hp, d, 64, hipMemcpyDeviceToHost, s); while (hipStreamQuery(s) == hipErrorNotReady) {} });
  t("kernel + mapped word + spin hipStreamQuery", [&] { bump<<<1, 64, 0, s>>>(d, hm_dev); while (hipStreamQuery(s) == hipErrorNotReady) {} });
  {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    t("kernel + eventRecord + spin hipEventQuery", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) {} });
    t("kernel + eventRecord + hipEventSynchronize", [&] { bump<<<1, 64, 0, s>>>(d, nullptr); hipEventRecord(ev, s); hipEventSynchronize(ev); });
  }
  t("work(2M) kernel + sync", [&] { work<<<n / 256, 256, 0, s>>>(x, n); hipStreamSynchronize(s); });
  t("work(2M) x2 async", [&] { work<<<n / 256, 256, 0, s>>>(x, n); work<<<n / 256, 256, 0, s>>>(x, n); });
  t("memsetAsync 128B", [&] { hipMemsetAsync(d, 0, 64, s); });
  volatile unsigned long long sink = *hm + *hp;
  (void)sink;
  return 0;
}
