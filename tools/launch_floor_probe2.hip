// Diagnostic (not product): cost of launching empty kernels of the prepare launch's shapes (1024-thread workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(double* out) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = 1.0;
}
__global__ __launch_bounds__(256) void k256(double* out) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = 1.0;
}
template <typename F>
void run(const char* what, F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < 2000; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.2f us per launch (back to back)\n", what, ms * 1e3 / 2000);
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 65536);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  run("512 x 1024 threads, 24 KB LDS", [&] { k<<<512, 1024, 24 * 1024>>>(out); });
  run("256 x 1024 threads, 24 KB LDS", [&] { k<<<256, 1024, 24 * 1024>>>(out); });
  run("256 x 1024 threads, 146 KB LDS", [&] { k<<<256, 1024, 146 * 1024>>>(out); });
  run("1024 x 1024 threads, 24 KB LDS", [&] { k<<<1024, 1024, 24 * 1024>>>(out); });
  run("2048 x 256 threads, 6 KB LDS", [&] { k256<<<2048, 256, 6 * 1024>>>(out); });
  run("8192 x 64 threads, 2 KB LDS", [&] { k256<<<8192, 64, 2 * 1024>>>(out); });
  run("128 x 256 threads", [&] { k256<<<128, 256, 0>>>(out); });
  return 0;
}
