// Diagnostic (not product): the sustained f64 / f32 MFMA rate and the shader clock this GPU holds under that load.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(int iters, unsigned long long* ticks, double* sink) {
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long m0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0));
  double acc_out = 0.0;
  if (MODE == 0) {           // v_mfma_f64_16x16x4_f64, 8 independent accumulators
    d4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) acc_out += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else if (MODE == 1) {    // v_mfma_f32_16x16x4_f32
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) acc_out += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else if (MODE == 3) {    // v_mfma_f64_4x4x4_f64 (four 4x4 blocks), 8 independent accumulators
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) acc_out += c[i];
  } else {                   // vector f64 FMA, 8 independent chains
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = threadIdx.x + i;
    double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_fma(c[i], a, b);
    }
    for (int i = 0; i < 8; ++i) acc_out += c[i];
  }
  unsigned long long m1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1));
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = m1 - m0; ticks[2 * blockIdx.x + 1] = t1 - t0; }
  if (acc_out == 12345.678) sink[0] = acc_out;
}

template <int MODE>
static void run(const char* name, double flop_per_wave_iter, int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;       // 256 threads = 4 waves, one per SIMD
  unsigned long long* d_ticks; double* d_sink;
  hipMalloc(&d_ticks, blocks * 2 * sizeof(unsigned long long)); hipMalloc(&d_sink, 8);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(burn<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d_ticks, d_sink);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    std::vector<unsigned long long> t(blocks * 2);
    hipMemcpy(t.data(), d_ticks, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mt = 0, ct = 0; for (int i = 0; i < blocks; ++i) { mt += t[2 * i]; ct += t[2 * i + 1]; }
    mt /= blocks; ct /= blocks;
    const double flops = flop_per_wave_iter * 8.0 * iters * blocks * 4.0;
    printf("%-28s waves/SIMD %d: %.3f ms  %.1f TFLOP/s   s_memtime %.0f ticks (%.3f GHz)  readcyclecounter %.0f (%.3f GHz)\n", name,
           waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, mt, mt / (ms * 1e6), ct, ct / (ms * 1e6));
  }
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("mfma_f64_16x16x4", 2.0 * 16 * 16 * 4, w, 20000);
    run<3>("mfma_f64_4x4x4", 2.0 * 4 * 4 * 4 * 4, w, 20000);
    run<1>("mfma_f32_16x16x4", 2.0 * 16 * 16 * 4, w, 20000);
    run<2>("v_fma_f64", 2.0 * 64, w, 20000);
  }
  return 0;
}
