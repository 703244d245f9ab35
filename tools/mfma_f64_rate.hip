// Microbenchmark (diagnostic): issue rate of v_mfma_f64_16x16x4_f64 on gfx950 with 1 / 2 / 4 / 8 independent
// accumulator chains, one or two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(double* out, int iters) {
  d4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int waves_per_simd) {
  const int iters = 4096 / CH * 8, blocks = 256, threads = 256 * waves_per_simd;
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<blocks, threads>>>(out, iters);
  hipEventRecord(e0);
  k<CH><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * CH * blocks * (threads / 64);
  printf("chains %d, waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", CH, waves_per_simd, ms,
         n_mfma * 2048 / ms / 1e9, ms * 1e6 / (n_mfma / (blocks * 4)));
  hipFree(out);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<2>(2); run<4>(2);
  return 0;
}
