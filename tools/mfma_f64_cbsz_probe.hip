// Diagnostic (not product): does v_mfma_f64_4x4x4_f64 honour CBSZ / ABID (A-block broadcast) on gfx950, and at what rate
// does it issue against v_mfma_f64_16x16x4_f64?  One-hot probing as in mfma_f64_4x4x4_layout.hip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_cbsz_probe.hip -o gpurun_out/cbsz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(unsigned long long* out) {      // out[la * 64 + lb] = mask of result lanes that see a[la] * b[lb]
  const int lane = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) out[blockIdx.x] = m;
}

template <int CBSZ, int ABID>
void show() {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(4096), dim3(64), 0, 0, d);
  std::vector<unsigned long long> h(4096);
  hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
  printf("== cbsz %d abid %d ==\n", CBSZ, ABID);
  for (int la = 0; la < 64; ++la) {
    bool any = false;
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) any = true;
    if (!any) continue;
    printf("a lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb]) {
        printf(" b%d->d", lb);
        for (int l = 0; l < 64; ++l) if (h[la * 64 + lb] >> l & 1) printf("%d,", l);
      }
    printf("\n");
  }
  hipFree(d);
}

// rates: MODE 0 = 16x16x4 (one chain of 4 registers), 1 = four 4x4x4 with abid 0..3 (cbsz 2) into four chains,
// 2 = four 4x4x4 without broadcast into four chains
template <int MODE>
__global__ void rate(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  d4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    } else if (MODE == 1) {
      acc.x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.x, 2, 0, 0);
      acc.y = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.y, 2, 1, 0);
      acc.z = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.z, 2, 2, 0);
      acc.w = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.w, 2, 3, 0);
    } else {
      acc.x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.x, 0, 0, 0);
      acc.y = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.y, 0, 0, 0);
      acc.z = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.z, 0, 0, 0);
      acc.w = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.w, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE>
void run_rate(int waves_per_simd) {
  const int iters = 16384, blocks = 512, threads = 256 * waves_per_simd;
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e0);
  rate<MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)iters * blocks * (threads / 64) * 2048.0;     // a 16x16x4 product (or its four quarters) per iteration
  printf("mode %d (%s), waves/SIMD %d: %.3f ms, %.1f TFLOP/s\n", MODE, MODE == 0 ? "16x16x4" : (MODE == 1 ? "4 x 4x4x4 cbsz 2" : "4 x 4x4x4"),
         waves_per_simd, ms, flops / ms / 1e9);
  hipFree(out);
}

// do vector f64 FMAs and f64 MFMAs share a pipe?  WHO: 1 = even waves issue MFMAs (odd waves idle), 2 = odd waves issue v_fma_f64, 3 = both
template <int WHO, int MFMA_MODE>
__global__ void mix(double* out, int iters) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  d4 acc = {0, 0, 0, 0};
  double f[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  if ((wave & 1) == 0) {
    if (WHO & 1)
      for (int i = 0; i < iters; ++i) {
        if (MFMA_MODE == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        else {
          acc.x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.x, 2, 0, 0);
          acc.y = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.y, 2, 1, 0);
          acc.z = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.z, 2, 2, 0);
          acc.w = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc.w, 2, 3, 0);
        }
      }
  } else {
    if (WHO & 2)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 8; ++c) f[c] = __builtin_fma(f[c], b, a);          // 16 vector FMAs = 2048 flops per iteration, like one 16x16x4
      }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
}
template <int WHO, int MFMA_MODE>
void run_mix() {
  const int iters = 16384, blocks = 512, threads = 512;      // 8 waves: two per SIMD, one of each kind
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix<WHO, MFMA_MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e0);
  mix<WHO, MFMA_MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)iters * blocks * 4 * 2048.0 * ((WHO & 1 ? 1 : 0) + (WHO & 2 ? 1 : 0));
  printf("mix who %d (1 MFMA waves, 2 vector-FMA waves, 3 both) mfma %s: %.3f ms, %.1f TFLOP/s\n", WHO, MFMA_MODE ? "4x4x4" : "16x16x4", ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run_mix<1, 0>(); run_mix<2, 0>(); run_mix<3, 0>(); run_mix<1, 1>(); run_mix<3, 1>();
  show<0, 0>();
  show<2, 0>(); show<2, 1>(); show<2, 3>();
  show<1, 0>(); show<1, 2>();
  for (int w = 1; w <= 4; w *= 2) { run_rate<0>(w); run_rate<1>(w); run_rate<2>(w); }
  return 0;
}
