// Diagnostic: (1) issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950, (2) its lane -> element maps, found with exact
// integer data: A[b][i][k], B[b][k][n] with distinct small integers, then which (b,i,n) each lane's D holds.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void rate(double* out, int iters) {
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  double s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void one(const double* a, const double* b, double* d) {
  d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  rate<<<256, 256>>>(out, iters);
  hipEventRecord(e0); rate<<<256, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * 256 * 4;
  printf("4x4x4_4b: %.3f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s (256 MAC each)\n", ms, ms * 1e6 / (n / 1024), n * 512 / ms / 1e9);
  // layout probe: hypothesis A: lane l -> block l/16, i = l%4 ... unknown; brute force over candidate maps
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // single-lane probes: set A = 1 on lane la only, B = 1 on lane lb only, see which D lanes become nonzero
  printf("nonzero D lanes for (A-lane, B-lane) probes:\n");
  for (int la : {0, 1, 4, 5, 16, 17, 20}) for (int lb : {0, 1, 4, 5, 16, 17, 20}) {
    for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = (i == lb); }
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    one<<<1, 64>>>(da, db, dd); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    printf("  A%-2d B%-2d ->", la, lb);
    for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf(" %d", i);
    printf("\n");
  }
  return 0;
}
