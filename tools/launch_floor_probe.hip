// Diagnostic (not product): what a launch of 512-thread workgroups with 78.8 KB of LDS costs before it computes anything --
// empty, with one round of global loads (fragments + three tiles into LDS), with two dependent rounds, and with a
// store -> barrier -> load-back round at the end (the stash of the product-fused sweep kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(const double* __restrict__ src, double* __restrict__ out, const int* __restrict__ sel, double* scratch) {
  extern __shared__ double lds[];
  const int t = threadIdx.x;
  double acc = 0.0;
  if (MODE >= 1) {
    int off = 0;
    if (MODE >= 2) off = sel[blockIdx.x & 63];                 // a dependent first round
    double fr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) fr[i] = src[off + i * 512 + t];
    const double2* s2 = reinterpret_cast<const double2*>(src + 32768 + (size_t)blockIdx.x * 3072);
    for (int k = 0; k < 3; ++k) reinterpret_cast<double2*>(lds)[k * 512 + t] = s2[k * 512 + t];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += fr[i];
    __syncthreads();
    acc += lds[(t * 7) & 3071];
  }
  if (MODE >= 3) {
    double2* st = reinterpret_cast<double2*>(scratch + (size_t)blockIdx.x * 6144);
    for (int k = 0; k < 6; ++k) st[k * 512 + t] = make_double2(acc, acc + k);
    __syncthreads();
    for (int k = 0; k < 6; ++k) { const double2 v = st[k * 512 + ((t + 64) & 511)]; acc += v.x * v.y; }
  }
  if (MODE == 0 ? t == 0 : true) out[(size_t)blockIdx.x * 512 + t] = acc;
}
template <int MODE>
void run(int grid, const double* src, double* out, const int* sel, double* scratch) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 78816);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) k<MODE><<<grid, 512, 78816>>>(src, out, sel, scratch);
  hipEventRecord(e0);
  for (int i = 0; i < 2000; ++i) k<MODE><<<grid, 512, 78816>>>(src, out, sel, scratch);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d grid %d: %.2f us per launch (back to back)\n", MODE, grid, ms * 1e3 / 2000);
}
int main() {
  double *src, *out, *scratch; int* sel;
  hipMalloc(&src, sizeof(double) * (32768 + 512 * 3072 + 65536)); hipMemset(src, 0, sizeof(double) * (32768 + 512 * 3072 + 65536));
  hipMalloc(&out, sizeof(double) * 512 * 512); hipMalloc(&scratch, sizeof(double) * 512 * 6144);
  hipMalloc(&sel, 256); hipMemset(sel, 0, 256);
  for (int grid : {256, 512}) { run<0>(grid, src, out, sel, scratch); run<1>(grid, src, out, sel, scratch); run<2>(grid, src, out, sel, scratch); run<3>(grid, src, out, sel, scratch); }
  return 0;
}
