// Diagnostic for the planned MFMA form of the per-graph kernel: how fast can a workgroup per graph bring its 3 tables
// (32 KiB each, unique per graph) into registers in BOTH v_mfma_f64_4x4x4 operand layouts, and run 18 matrix-vector
// products of 16 instructions each on them?  Compares with the current layout (24 x 16-byte loads per thread).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_gemv_load_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WG = 256;
__global__ __launch_bounds__(WG, 3) void current_layout(const double* tables, double* out) {
  const double2* T = reinterpret_cast<const double2*>(tables + (size_t)blockIdx.x * 3 * 4096);
  double2 tab[3][8];
  for (int p = 0; p < 3; ++p)
    for (int k = 0; k < 8; ++k) tab[p][k] = T[p * 2048 + k * WG + threadIdx.x];
  double acc = 0.0;
  for (int p = 0; p < 3; ++p)
    for (int k = 0; k < 8; ++k) acc += tab[p][k].x + tab[p][k].y;
  out[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
template <int UPDATES>
__global__ __launch_bounds__(WG, 2) void both_layouts(const double* tables, double* out) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, k = lane >> 4;
  const double* T0 = tables + (size_t)blockIdx.x * 3 * 4096;
  double2 aTM[3][8];         // rows 16w + i, columns 8s + 2k, +1   (64-byte segments per row)
  double aMT[3][16];         // rows 4s + k, column 16w + i          (128-byte segments)
  for (int p = 0; p < 3; ++p) {
    const double* T = T0 + p * 4096;
    for (int s = 0; s < 8; ++s) aTM[p][s] = *reinterpret_cast<const double2*>(T + (16 * wave + i) * 64 + 8 * s + 2 * k);
    for (int s = 0; s < 16; ++s) aMT[p][s] = T[(4 * s + k) * 64 + 16 * wave + i];
  }
  double b = 1.0 + lane * 1e-3, acc = 0.0;
  for (int u = 0; u < UPDATES; ++u) {
    const int p = u % 3;
    double d0 = 0.0, d1 = 0.0;
    if (u & 1) {
      for (int s = 0; s < 16; s += 2) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(aMT[p][s], b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(aMT[p][s + 1], b, d1, 0, 0, 0);
      }
    } else {
      for (int s = 0; s < 8; ++s) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(aTM[p][s].x, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(aTM[p][s].y, b, d1, 0, 0, 0);
      }
    }
    acc += d0 + d1;
    b = acc * 1e-30 + 1.0;          // dependent chain like the real sweep
  }
  out[(size_t)blockIdx.x * WG + t] = acc;
}
int main() {
  const int B = 8192;
  double *tables, *out;
  hipMalloc(&tables, sizeof(double) * (size_t)B * 3 * 4096); hipMalloc(&out, sizeof(double) * (size_t)B * WG);
  hipMemset(tables, 0, sizeof(double) * (size_t)B * 3 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 50; ++i) current_layout<<<B, WG>>>(tables, out);
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) current_layout<<<B, WG>>>(tables, out); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("current layout, loads only:            %.3f ms per launch (%.2f TB/s)\n", ms / 20, B * 3 * 32768.0 / (ms / 20) / 1e9);
    for (int i = 0; i < 50; ++i) both_layouts<0><<<B, WG>>>(tables, out);
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) both_layouts<0><<<B, WG>>>(tables, out); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("both MFMA layouts, loads only:         %.3f ms per launch (%.2f TB/s of table bytes)\n", ms / 20, B * 3 * 32768.0 / (ms / 20) / 1e9);
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) both_layouts<18><<<B, WG>>>(tables, out); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("both MFMA layouts + 18 x 16 MFMA 4x4x4: %.3f ms per launch\n", ms / 20);
  }
  return 0;
}
