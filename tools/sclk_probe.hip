// Diagnostic (not product): the shader clock a kernel actually runs at -- s_memtime (shader clock) against s_memrealtime
// (constant 100 MHz) -- for long and for short f64-MFMA kernels launched back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, unsigned long long* clk, int iters) {
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  d4 acc = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 512 * 512);
  unsigned long long* clk; hipMalloc(&clk, 16);
  unsigned long long h[2];
  for (int iters : {32768, 4096, 512, 512}) {
    const int reps = iters >= 4096 ? 3 : 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<512, 512>>>(out, clk, iters);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<<<512, 512>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("iters %6d: %.1f us per kernel; wave: %llu shader ticks over %llu ref ticks (100 MHz) = %.0f MHz; %.1f TFLOP/s\n", iters, ms * 1e3 / reps, h[0], h[1],
           (double)h[0] / h[1] * 100.0, (double)iters * 512 * 8 * 2048.0 * reps / ms / 1e9);
  }
  return 0;
}
