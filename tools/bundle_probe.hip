// Diagnostic (not product): the product-fused bundle of the shared-table sweep kernel as a synthetic loop -- eight waves, each
// member 8 tile reads (16 B per lane) feeding 16 dependent v_mfma_f64_16x16x4, 4 + 4 multiplications, a column sum, three LDS
// writes and one barrier -- timed per bundle with one and two workgroups per CU, and with parts taken out.
//   MODE bit 0: no barrier   bit 1: no MFMAs   bit 2: no LDS reads (operands constant)   bit 3: no tail (products, sums, writes)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double column_sum(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(const double* __restrict__ frag, double* __restrict__ out, int bundles) {
  extern __shared__ double lds[];
  double* tiles = lds;                     // 9 tiles of 1024 doubles
  double* tot = lds + 9 * 1024;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), half = wave >> 2, rb = wave & 3;
  const int gl = lane & 15, cq = lane >> 4;
  double fr0[16], fr1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { fr0[i] = frag[(half * 2) * 4096 + rb * 1024 + 64 * i + lane]; fr1[i] = frag[(half * 2 + 1) * 4096 + rb * 1024 + 64 * i + lane]; }
  for (int i = t; i < 9 * 1024; i += 512) tiles[i] = 1.0 / 64.0;
  for (int i = t; i < 9 * 64; i += 512) tot[i] = 0.25;
  __syncthreads();
  for (int b = 0; b < bundles; ++b) {
    const int S = __builtin_amdgcn_readfirstlane((b * 2 + half) % 6), dst = __builtin_amdgcn_readfirstlane((b * 2 + half + 3) % 6), cd = 6 + (b + half) % 3;
    const bool second = (b & 1) != 0;
    const double2* src = reinterpret_cast<const double2*>(tiles + S * 1024) + lane;
    const double2* tp = reinterpret_cast<const double2*>(tot + S * 64 + gl * 4);
    const double2* cs = reinterpret_cast<const double2*>(tiles + cd * 1024) + 128 * rb + lane;
    double2 qa0, qa1, qb0, qb1, ta, tb, c0, c1;
    if (!(MODE & 4)) { qa0 = src[0]; qa1 = src[64]; qb0 = src[128]; qb1 = src[192]; ta = tp[0]; tb = tp[1]; }
    else { qa0 = qa1 = qb0 = qb1 = make_double2(1.0 / 64, 1.0 / 64); ta = tb = make_double2(0.25, 0.25); }
    d4 acc = {0, 0, 0, 0};
    double s = 1.0;
#define Q(FR, H, V0, V1) if (!(MODE & 2)) { \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H)], V0.x, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 1], V0.y, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 2], V1.x, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 3], V1.y, acc, 0, 0, 0); } \
    else { acc.x += V0.x; acc.y += V0.y; acc.z += V1.x; acc.w += V1.y; }
#define BODY(FR) \
    Q(FR, 0, qa0, qa1) __builtin_amdgcn_sched_barrier(0); \
    if (!(MODE & 4)) { qa0 = src[256]; qa1 = src[320]; } \
    { const double total = (ta.x + ta.y) + (tb.x + tb.y); s = __builtin_amdgcn_rcp(total); } __builtin_amdgcn_sched_barrier(0); \
    Q(FR, 1, qb0, qb1) __builtin_amdgcn_sched_barrier(0); \
    if (!(MODE & 4)) { qb0 = src[384]; qb1 = src[448]; } __builtin_amdgcn_sched_barrier(0); \
    Q(FR, 2, qa0, qa1) __builtin_amdgcn_sched_barrier(0); \
    if (!(MODE & 4)) { c0 = cs[0]; c1 = cs[64]; } else { c0 = c1 = make_double2(1.0, 1.0); } __builtin_amdgcn_sched_barrier(0); \
    Q(FR, 3, qb0, qb1) __builtin_amdgcn_sched_barrier(0);
    if (!second) { BODY(fr0) } else { BODY(fr1) }
    if (!(MODE & 8)) {
      const double k0 = c0.x * s, k1 = c0.y * s, k2 = c1.x * s, k3 = c1.y * s;
      const double p0 = acc.x * k0, p1 = acc.y * k1, p2 = acc.z * k2, p3 = acc.w * k3;
      double2* o = reinterpret_cast<double2*>(tiles + dst * 1024) + 128 * rb + lane;
      o[0] = make_double2(p0, p1); o[64] = make_double2(p2, p3);
      const double colsum = column_sum((p0 + p1) + (p2 + p3));
      if (cq == 0) tot[dst * 64 + gl * 4 + rb] = colsum;
    } else {
      if (acc.x == 12345.0) tot[0] = acc.y + c0.x + s;
    }
    if (!(MODE & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __syncthreads();
  out[(size_t)blockIdx.x * 512 + t] = tiles[t] + tot[t & 63];
}
template <int MODE>
void run(int grid, const double* frag, double* out) {
  const int lds = 9 * 1024 * 8 + 9 * 64 * 8 + 512;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float t[2];
  for (int pass = 0; pass < 2; ++pass) {
    const int bundles = pass ? 1100 : 100;
    k<MODE><<<grid, 512, lds>>>(frag, out, bundles);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<grid, 512, lds>>>(frag, out, bundles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&t[pass], e0, e1);
  }
  const double us = (t[1] - t[0]) * 1e3 / 5 / 1000;
  printf("mode %2d (%s%s%s%s) grid %d: %.3f us per bundle = %.0f cycles at 2.4 GHz\n", MODE, MODE & 1 ? "no barrier " : "", MODE & 2 ? "no MFMA " : "", MODE & 4 ? "no LDS reads " : "",
         MODE & 8 ? "no tail" : "", grid, us, us * 2400);
}
int main() {
  double *frag, *out;
  hipMalloc(&frag, sizeof(double) * 4 * 4096); hipMemset(frag, 0, sizeof(double) * 4 * 4096);
  hipMalloc(&out, sizeof(double) * 512 * 512);
  for (int grid : {256, 512}) { run<0>(grid, frag, out); run<1>(grid, frag, out); run<2>(grid, frag, out); run<4>(grid, frag, out); run<8>(grid, frag, out); run<12>(grid, frag, out); run<13>(grid, frag, out); run<6>(grid, frag, out); }
  return 0;
}
