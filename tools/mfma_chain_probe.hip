// Diagnostic (not product): v_mfma_f64_16x16x4_f64 over R accumulators visited in turn, CH consecutive (dependent) MFMAs per
// visit -- does the matrix pipe run a dependent run faster than independent accumulators (source C forwarded, not read)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int R, int CH>
__global__ void k(double* out, int iters) {
  d4 acc[R];
  for (int r = 0; r < R; ++r) acc[r] = d4{0, 0, 0, 0};
  double a[CH], b[CH];
  for (int c = 0; c < CH; ++c) { a[c] = threadIdx.x * 1e-3 + c; b[c] = 1.0 + threadIdx.x * 1e-4 * (c + 1); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], b[c], acc[r], 0, 0, 0);
  }
  double s = 0;
  for (int r = 0; r < R; ++r) s += acc[r].x + acc[r].y + acc[r].z + acc[r].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int R, int CH>
void run(int waves_per_simd) {
  const int iters = 16384 / (R * CH), blocks = 256, threads = 256 * waves_per_simd;
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<R, CH><<<blocks, threads>>>(out, iters);
  hipEventRecord(e0);
  k<R, CH><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * R * CH * blocks * (threads / 64);
  printf("accumulators %d, run of %d, waves/SIMD %d: %.1f TFLOP/s\n", R, CH, waves_per_simd, n * 2048 / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 2; ++w) { run<1, 1>(w); run<4, 1>(w); run<8, 1>(w); run<4, 2>(w); run<4, 4>(w); run<4, 8>(w); run<2, 16>(w); }
  return 0;
}
