// X = 64, float64, per-graph tables: the "lean" scale-free sweep kernel (default path of BASELINE configs 2-4).
//
// Same algorithm and the same scale-free representation as sweep_x64_sf_kernel (mlbp_sweep.hip): one 256-thread
// workgroup owns one graph for all sweeps of a call, messages live in LDS carried with an exact power-of-two
// scale, the pairwise tables stay in registers, true normalisations (LBP.py:649-657) happen once after the last
// sweep, graphs the representation cannot hold are flagged for the exact kernel.  What differs is the cost of one
// update -- the old kernel spent ~275 wave instructions and ~10 dependent LDS round trips around 16 useful FMAs:
//
//   * the program is compiled on the host into 8-word micro-ops whose operands are LDS byte offsets, so an
//     update reads one descriptor and issues its message loads at once (no source lists, no slot arithmetic);
//   * a thread owns a 4 x 4 block of each table (rows 4R..4R+3, column pairs {2c,2c+1} and {32+2c,33+2c}), so BOTH
//     directions of an update are 16 FMAs followed by a short select-light reduction:
//        T.m   (reduce over columns): v_permlane16_swap (4 -> 2 values), one DPP step with a select (2 -> 1), two
//              plain DPP steps;
//        m^T.T (reduce over rows):    one DPP row_ror:8 step (the two column pairs sit in swapped registers in lanes
//              with bit 3 set, so no select is needed), v_permlane32_swap, then the four waves meet in LDS;
//     and the variable->factor product feeding an update is formed directly in the distribution the contraction
//     needs (4 columns or 4 rows per lane) -- no LDS round trip between product and contraction;
//   * every global load instruction of a table covers 128-byte contiguous pieces (8 lanes x 16 B).
// One barrier per bundle of (at most two independent) updates, as before.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "mlbp_device.h"
#include "mlbp_internal.h"

using mlbp::fail;
using namespace mlbp_dev;

// Cache policy (MLBP_LEAN_NT: bit 0 the pairwise tables by non-temporal loads -- per-graph tables are read once: 12.5 -> 13.0 k
// iterations/s on the default workload --, bit 1 write-back and marginals by non-temporal stores (no effect measured), bit 2 the
// unary rows by non-temporal loads)
#ifndef MLBP_LEAN_NT
#define MLBP_LEAN_NT 1
#endif
typedef double nt_d2 __attribute__((ext_vector_type(2)));
#if MLBP_LEAN_NT & 1
#define NT_LOAD2(p) ([&] { const nt_d2 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p)); return make_double2(v_.x, v_.y); }())
#else
#define NT_LOAD2(p) (*reinterpret_cast<const double2*>(p))
#endif
#if MLBP_LEAN_NT & 4
#define NT_LOAD1(p) __builtin_nontemporal_load(p)
#else
#define NT_LOAD1(p) (*(p))
#endif
#if MLBP_LEAN_NT & 2
#define NT_STORE2(p, v) { const double2 w_ = (v); nt_d2 x_; x_.x = w_.x; x_.y = w_.y; __builtin_nontemporal_store(x_, reinterpret_cast<nt_d2*>(p)); }
#define NT_STORE1(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NT_STORE2(p, v) (*(p) = (v))
#define NT_STORE1(p, v) (*(p) = (v))
#endif

namespace {

constexpr int WG = 256;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(MLBP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// micro-op word 0
constexpr int UOP_VAR = 1;          // bit 0: variable product only (stored, no contraction)
constexpr int UOP_MT = 2;           // bit 1: out = m^T . T (else T . m)
constexpr int UOP_PSLOT_SHIFT = 2;  // bits 2-4: pair slot (register-resident table; 0..7)
constexpr int UOP_NOP = 32;         // bit 5: empty second slot of a bundle
constexpr int UOP_STORE_VF = 64;    // bit 6: the variable->factor message is stored (word 5)
constexpr int UOP_NSRC_SHIFT = 8;   // bits 8-11: number of sources (1..4)
constexpr int UOP_CARRY_IN = 0x1000;   // a link of a long variable product: starts from the running product in registers
constexpr int UOP_CARRY_OUT = 0x2000;  // ... and hands it on instead of storing it
constexpr int GROUP_WORDS = 48;     // one group descriptor of a MULTI launch (see launch_lean_groups)
// micro-op words: 0 flags | 1-4 source byte offsets | 5 byte offset of the variable->factor message to store, or -1 |
// 6 destination byte offset | 7 unused.  A bundle = two micro-ops = 16 words = one s_load_dwordx16.

struct LeanDev {
  const int32_t* image;   // bundles [n_bundles][16] | hoist [4][HL][2] | cprod lists [n_cprod][16] | written [4][WL] | pad
  const int32_t* readout; // [n_vars][16]: count, base slot, varying slots... (build_lean_readout) or NULL
  uint8_t* bail;          // [B]
  int32_t n_bundles, HL, n_cprod, WL, n_ext, init, dense, keep;
};

// 16 consecutive words through the scalar data cache (s_load_dwordx16): wave-uniform program data lands in SGPRs.
struct Words16 { int32_t w[16]; };
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ Words16 sload16(const int32_t* p) {
  const v16i v = *(const v16i __attribute__((address_space(4)))*)(uintptr_t)p;
  Words16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = v[i];
  return r;
}

// Diagnostic build only (-DMLBP_LEAN_PROBE, tools/lean_probe.py): phase ablation by mask and shader-clock stamps of
// wave 0 of selected workgroups into a side buffer no other code reads.  Never defined in the shipped library.
#ifdef MLBP_LEAN_PROBE
__device__ int g_probe_mask = 0;
__device__ unsigned long long* g_probe_buf = nullptr;
#define PROBE_DECL const int probe_mask_ = __builtin_amdgcn_readfirstlane(g_probe_mask); unsigned long long pst_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int pn_ = 0;
#define PROBED(bit) (probe_mask_ & (1 << (bit)))
#define PSTAMP { unsigned long long _t; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); pst_[pn_++] = _t; }
#define PSTAMP_VM { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PSTAMP }
#define PFLUSH if (g_probe_buf && threadIdx.x == 0 && (blockIdx.x & 63) == 0) { unsigned long long _rt; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_rt) :: "memory"); for (int _i = 0; _i < 10; ++_i) g_probe_buf[(blockIdx.x >> 6) * 12 + _i] = pst_[_i]; g_probe_buf[(blockIdx.x >> 6) * 12 + 10] = _rt; }
#else
#define PROBE_DECL
#define PROBED(bit) 0
#define PSTAMP
#define PSTAMP_VM
#define PFLUSH
#endif

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3] (rows of 16 lanes); a' + b'
__device__ __forceinline__ double swapadd16(double a, double b) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// a' = [a.lo32, b.lo32], b' = [a.hi32, b.hi32] (halves of 32 lanes); a' + b'
__device__ __forceinline__ double swapadd32(double a, double b) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

struct LaneGeo {
  int tm0, tm1;   // byte offsets inside a message of the lane's column pairs held in table registers k = 0, 1
  int mt;         // byte offset of the lane's four rows
  int red_tm;     // byte offset (in a [64] array) of the row this lane reports after the T.m reduction
  int red_mt;     // byte offset (in a [4][64] array) of the column this lane reports after the m^T.T reduction
  bool up, st_tm, st_mt, wr_tm;
};

__device__ __forceinline__ double2 lds2(const char* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void mul2(double2& a, const double2 b) { a.x *= b.x; a.y *= b.y; }

// One table (T[r][k] = rows 4R + r, column pair k ^ b3) against one input vector; u = the micro-op's 8 words (scalars).
// u0 = flags (wave-uniform scalar); s1..s4, vf = source / store byte offsets (same in every lane, kept in VGPRs).
// A table held in registers, or -- the (NT+1)-th table of a 7-table graph -- in LDS as [r * 2 + k][thread] double2.
struct RegTable {
  const double2 (&T)[4][2];
  __device__ __forceinline__ double2 operator()(int r, int k) const { return T[r][k]; }
};
struct LdsTable {
  const double2* base;                      // this thread's column of the image
  __device__ __forceinline__ double2 operator()(int r, int k) const { return base[(r * 2 + k) * WG]; }
};

template <bool MT, typename TB>
__device__ __forceinline__ void contract(const TB T, char* wb, int u0, int s1, int s2, int s3, int s4, int vf,
                                         const LaneGeo& G, char* redA) {
  const int nsrc = (u0 >> UOP_NSRC_SHIFT) & 15;
  const bool has_vf = (u0 & UOP_STORE_VF) != 0;
  if (MT) {
    double2 ma = lds2(wb + s1 + G.mt), mb = lds2(wb + s1 + G.mt + 16);
    if (nsrc > 1) {
      const double2 a = lds2(wb + s2 + G.mt), b = lds2(wb + s2 + G.mt + 16);
      mul2(ma, a); mul2(mb, b);
      if (nsrc > 2) {
        const double2 a2 = lds2(wb + s3 + G.mt), b2 = lds2(wb + s3 + G.mt + 16);
        mul2(ma, a2); mul2(mb, b2);
        if (nsrc > 3) { const double2 a3 = lds2(wb + s4 + G.mt), b3 = lds2(wb + s4 + G.mt + 16); mul2(ma, a3); mul2(mb, b3); }
      }
    }
    if (has_vf && G.st_mt) {
      *reinterpret_cast<double2*>(wb + vf + G.mt) = ma;
      *reinterpret_cast<double2*>(wb + vf + G.mt + 16) = mb;
    }
    double a00, a01, a10, a11;
    { const double2 t0 = T(0, 0), t1 = T(0, 1); a00 = ma.x * t0.x; a01 = ma.x * t0.y; a10 = ma.x * t1.x; a11 = ma.x * t1.y; }
    { const double2 t0 = T(1, 0), t1 = T(1, 1); a00 += ma.y * t0.x; a01 += ma.y * t0.y; a10 += ma.y * t1.x; a11 += ma.y * t1.y; }
    { const double2 t0 = T(2, 0), t1 = T(2, 1); a00 += mb.x * t0.x; a01 += mb.x * t0.y; a10 += mb.x * t1.x; a11 += mb.x * t1.y; }
    { const double2 t0 = T(3, 0), t1 = T(3, 1); a00 += mb.y * t0.x; a01 += mb.y * t0.y; a10 += mb.y * t1.x; a11 += mb.y * t1.y; }
    // rows: lane bit 3 (partner l ^ 8 keeps the other column pair in its register 0), lane bit 5, then the waves
    a00 += dpp_mov<0x128>(a10);
    a01 += dpp_mov<0x128>(a11);
    *reinterpret_cast<double*>(redA + G.red_mt) = swapadd32(a00, a01);
  } else {
    double2 m0 = lds2(wb + s1 + G.tm0), m1 = lds2(wb + s1 + G.tm1);
    if (nsrc > 1) {
      const double2 a = lds2(wb + s2 + G.tm0), b = lds2(wb + s2 + G.tm1);
      mul2(m0, a); mul2(m1, b);
      if (nsrc > 2) {
        const double2 a2 = lds2(wb + s3 + G.tm0), b2 = lds2(wb + s3 + G.tm1);
        mul2(m0, a2); mul2(m1, b2);
        if (nsrc > 3) { const double2 a3 = lds2(wb + s4 + G.tm0), b3 = lds2(wb + s4 + G.tm1); mul2(m0, a3); mul2(m1, b3); }
      }
    }
    if (has_vf && G.st_tm) {
      *reinterpret_cast<double2*>(wb + vf + G.tm0) = m0;
      *reinterpret_cast<double2*>(wb + vf + G.tm1) = m1;
    }
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double2 t0 = T(r, 0), t1 = T(r, 1);
      v[r] = t0.x * m0.x;
      v[r] += t0.y * m0.y;
      v[r] += t1.x * m1.x;
      v[r] += t1.y * m1.y;
    }
    // columns: lane bit 4 (rows of 16 lanes exchange registers), lane bit 2 (select), lane bits 1 and 0 (plain)
    const double u0 = swapadd16(v[0], v[2]), u1 = swapadd16(v[1], v[3]);
    const double keep = G.up ? u1 : u0, send = G.up ? u0 : u1;
    double w = keep + dpp_mov<0x141>(send);
    w += dpp_mov<0xB1>(w);
    w += dpp_mov<0x4E>(w);
    *reinterpret_cast<double*>(redA + G.red_tm) = w;        // the four lanes of a quad hold (and store) the same row
  }
}

template <int NT, int NL>
__device__ __forceinline__ void front(const double2 (&tab)[NT][4][2], const double2* tl, char* wb, int u0, const int4& lo, const int4& hi,
                                      const LaneGeo& G, char* redA) {
  const int pslot = (u0 >> UOP_PSLOT_SHIFT) & 7;
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    if (p == pslot) {
      if (u0 & UOP_MT) contract<true>(RegTable{tab[p]}, wb, u0, lo.y, lo.z, lo.w, hi.x, hi.y, G, redA);
      else contract<false>(RegTable{tab[p]}, wb, u0, lo.y, lo.z, lo.w, hi.x, hi.y, G, redA);
    }
  }
  if (NL > 0 && pslot >= NT) {
    const LdsTable T = {tl + (size_t)(pslot - NT) * 8 * WG};
    if (u0 & UOP_MT) contract<true>(T, wb, u0, lo.y, lo.z, lo.w, hi.x, hi.y, G, redA);
    else contract<false>(T, wb, u0, lo.y, lo.z, lo.w, hi.x, hi.y, G, redA);
  }
}

// FactorGraph.get_unregularized_gradeint (LBP.py:301-320) for one graph, from what the workgroup still holds after its
// sweeps: tables in registers (4 x 4 blocks), messages in LDS.  Pairwise factor (LBP.py:543-569, 592-619): beliefs =
// normalize((c r^T) (.) T), gradient = phi[label cell] - sum_ij beliefs_ij phi_ij -- the scale of c and r cancels, so the
// scaled messages serve as they are, and the 64 x 64 belief matrix is never formed: each thread folds its 16 cells into
// Z and three feature sums.  Unary factor (LBP.py:535-541, 600-603): beliefs = au.normalize(table), which IS the factor's
// hoisted message whenever the table's total is positive (else zero); its expected features are accumulated per lane over
// the wave's factors and reduced once.  gst: [U] message slot, [U] total positive, [U] kind, [U] observed column, [U] label.
template <int NT>
__device__ __forceinline__ void gradient_epilogue(const SweepDev& d, const GradFusedDev& gf, const double2 (&tab)[NT][4][2],
                                                  const char* wb, const int32_t* gst, double* scratch, const LaneGeo& G, int g,
                                                  int R_, int c_, int b3_) {
  constexpr int FEE = 3, FED = 6;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // Pairwise factors.  sum_p E_p[k] = sum_ij (sum_p b_p[i][j]) phi[i][j][k] over the factors p that read one feature tensor, b_p =
  // the factor's normalised beliefs: so the beliefs of those factors are ADDED cell by cell first and the tensor -- 98 KB per
  // graph out of L2, interleaved [64][64][3] -- goes past once per graph instead of once per factor (2.4 GB per 8192-graph
  // launch of K3 graphs: what the epilogue was bound by).  Three steps, all on the thread's 4 x 4 blocks:
  //   1. W_p = (c r^T) (.) T_p in place of the table registers, and its total Z_p (wave sums, the four waves meet in LDS);
  //   2. V_t = sum over the factors reading tensor t (gap == 1: phi_en_en_w1, else phi_en_en; LBP.py:469-480) of W_p / Z_p
  //      (au.normalize: a total <= 0 gives zero beliefs);
  //   3. G_t[k] = sum_ij V_t[i][j] phi_t[i][j][k]: three accumulators per tensor, the thread's sixteen cells of the tensor
  //      loaded eight at a time.
  // thread p < P: the label cell's features of pairwise factor p -- two dependent loads, requested first and long finished when
  // thread 0 adds them up (they used to be the last thing the workgroup waited for, one thread, two round trips)
  double lf[FEE] = {0.0, 0.0, 0.0};
  bool lbad = false;
  if (t < NT && t < d.P) {
    const int l0 = gf.pair_label[((size_t)g * d.P + t) * 2], l1 = gf.pair_label[((size_t)g * d.P + t) * 2 + 1];
    if ((unsigned)l0 >= 64u || (unsigned)l1 >= 64u) lbad = true;
    else {
      const double* phl = (gf.pair_phi[t] ? gf.phi_en_en_w1 : gf.phi_en_en) + ((size_t)l0 * 64 + l1) * FEE;
#pragma unroll
      for (int q = 0; q < FEE; ++q) lf[q] = phl[q];
    }
  }
  int which = 0;                                                 // bit p: factor p reads phi_en_en_w1
  double2 W[NT][4][2];
  double* zs = scratch + 512;                                    // [4 waves][NT] (the per-wave results below use scratch[0 .. 4 PER))
  double* lfs = scratch + 544;                                   // [NT][FEE] the label features, for thread 0's final sum
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    double z = 0.0;
    if (p < d.P) {
      const int cs = as_const(gf.pair_c_slot)[p], rs = as_const(gf.pair_r_slot)[p];
      which |= (as_const(gf.pair_phi)[p] ? 1 : 0) << p;
      const double2 ca = lds2(wb + cs * 512 + G.mt), cb = lds2(wb + cs * 512 + G.mt + 16);
      const double2 r0 = lds2(wb + rs * 512 + G.tm0), r1 = lds2(wb + rs * 512 + G.tm1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double ci = r == 0 ? ca.x : (r == 1 ? ca.y : (r == 2 ? cb.x : cb.y));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double2 rj = k ? r1 : r0, T = tab[p][r][k];
          W[p][r][k] = make_double2((ci * rj.x) * T.x, (ci * rj.y) * T.y);
          z += W[p][r][k].x + W[p][r][k].y;
        }
      }
      z = wave_sum(z);
      if (lane == 0) zs[wave * NT + p] = z;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) W[p][r][0] = W[p][r][1] = make_double2(0.0, 0.0);
    }
  }
  if (lbad) atomicExch(d.status, 1);
  if (t < NT && t < d.P) {                                       // (parked in LDS here: kept in registers to the end they spill the tables)
#pragma unroll
    for (int q = 0; q < FEE; ++q) lfs[t * FEE + q] = lf[q];
  }
  lds_barrier();
  double2 V[2][4][2];
#pragma unroll
  for (int tz = 0; tz < 2; ++tz)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[tz][r][0] = V[tz][r][1] = make_double2(0.0, 0.0);
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    if (p < d.P) {
      const double Z = (zs[p] + zs[NT + p]) + (zs[2 * NT + p] + zs[3 * NT + p]);
      const double inv = Z > 0.0 ? 1.0 / Z : 0.0;                // au.normalize: zero sum -> zero beliefs
      const double i0 = ((which >> p) & 1) ? 0.0 : inv, i1 = ((which >> p) & 1) ? inv : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          V[0][r][k].x += W[p][r][k].x * i0; V[0][r][k].y += W[p][r][k].y * i0;
          V[1][r][k].x += W[p][r][k].x * i1; V[1][r][k].y += W[p][r][k].y * i1;
        }
    }
  }
  double gacc[2][FEE] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  const int all = (1 << (d.P < NT ? d.P : NT)) - 1;
#pragma unroll
  for (int tz = 0; tz < 2; ++tz) {
    if (tz ? (which & all) == 0 : (which & all) == all) continue;          // no factor reads this tensor (wave-uniform)
    // (global address space spelled out: behind a run-time choice the loads otherwise come out as FLAT ones, which wait on
    // the LDS counter as well)
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef const v2d __attribute__((address_space(1))) * gptr2;
    const double* phi = tz ? gf.phi_en_en_w1 : gf.phi_en_en;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = 32 * (k ? 1 - b3_ : b3_) + 2 * c_;
      v2d cell[4][3];                                            // cell (i, col): f0 f1 f2 | cell (i, col + 1): g0 g1 g2
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const gptr2 ph = (gptr2)(uintptr_t)(phi + ((size_t)(4 * R_ + r) * 64 + col) * FEE);   // 48 bytes, 16-aligned
        cell[r][0] = ph[0]; cell[r][1] = ph[1]; cell[r][2] = ph[2];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double w0 = V[tz][r][k].x, w1 = V[tz][r][k].y;
        gacc[tz][0] += w0 * cell[r][0].x + w1 * cell[r][1].y;
        gacc[tz][1] += w0 * cell[r][0].y + w1 * cell[r][2].x;
        gacc[tz][2] += w0 * cell[r][1].x + w1 * cell[r][2].y;
      }
    }
  }
  double gsum[FEE];                                              // expected features of all pairwise factors (this wave's share)
#pragma unroll
  for (int q = 0; q < FEE; ++q) gsum[q] = wave_sum(gacc[0][q] + gacc[1][q]);
  // unary factors: wave w takes factors w, w + 4, ...; six at a time so that their feature slabs are in flight together (the
  // tables are dead by now: the registers are there)
  double lab_ee[FEE] = {0.0, 0.0, 0.0}, lab_ed[FED] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // phi at the label (wave-uniform)
  double exp_ee[FEE] = {0.0, 0.0, 0.0}, exp_ed[FED] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // per-lane partial expected features
  bool bad = false;
  constexpr int UB = 6;                                          // (a K3 user graph's 24 unary factors: one round trip per wave)
  const int U = d.U;
  for (int u0 = wave; u0 < U; u0 += 4 * UB) {
    double b[UB], pv[UB][FED];
    int kd[UB], at[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = u0 + 4 * j;
      kd[j] = -1;
      if (u < U) {
        const int kind = __builtin_amdgcn_readfirstlane(gst[2 * U + u]), obs = __builtin_amdgcn_readfirstlane(gst[3 * U + u]);
        const int lab = __builtin_amdgcn_readfirstlane(gst[4 * U + u]);
        const int cols = kind == 2 ? gf.Vde : 64;
        if ((unsigned)obs >= (unsigned)cols || (unsigned)lab >= 64u || (unsigned)kind > 2u) { bad = true; continue; }
        kd[j] = kind;
        const int slot = __builtin_amdgcn_readfirstlane(gst[u]);
        b[j] = __builtin_amdgcn_readfirstlane(gst[U + u]) ? *reinterpret_cast<const double*>(wb + slot * 512 + lane * 8) : 0.0;
        if (kind == 2) {
          const double2* ph = reinterpret_cast<const double2*>(gf.phi_en_de_t + ((size_t)obs * 64 + lane) * FED);
          at[j] = (obs * 64 + lab) * FED;
          const double2 a0 = ph[0], a1 = ph[1], a2 = ph[2];
          pv[j][0] = a0.x; pv[j][1] = a0.y; pv[j][2] = a1.x; pv[j][3] = a1.y; pv[j][4] = a2.x; pv[j][5] = a2.y;
        } else {
          typedef const double __attribute__((address_space(1))) * gptr1;      // (global, not FLAT, behind the run-time choice of tensor)
          const gptr1 ph = (gptr1)(uintptr_t)((kind ? gf.phi_en_en_w1_t : gf.phi_en_en_t) + ((size_t)obs * 64 + lane) * FEE);
          at[j] = (obs * 64 + lab) * FEE;
#pragma unroll
          for (int q = 0; q < FEE; ++q) pv[j][q] = ph[q];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (kd[j] == 2) {
#pragma unroll
        for (int q = 0; q < FED; ++q) { lab_ed[q] += as_const_f64(gf.phi_en_de_t)[at[j] + q]; exp_ed[q] += b[j] * pv[j][q]; }
      } else if (kd[j] >= 0) {
        const double* base = kd[j] ? gf.phi_en_en_w1_t : gf.phi_en_en_t;
#pragma unroll
        for (int q = 0; q < FEE; ++q) { lab_ee[q] += as_const_f64(base)[at[j] + q]; exp_ee[q] += b[j] * pv[j][q]; }
      }
    }
  }
  if (bad && lane == 0) atomicExch(d.status, 1);
  // combine the four waves: per wave the FEE pairwise sums, then FEE + FED unary sums
  constexpr int PER = FEE + FEE + FED;
  {
    double ue[FEE], ud[FED];
#pragma unroll
    for (int q = 0; q < FEE; ++q) ue[q] = lab_ee[q] - wave_sum(exp_ee[q]);
#pragma unroll
    for (int q = 0; q < FED; ++q) ud[q] = lab_ed[q] - wave_sum(exp_ed[q]);
    if (lane == 0) {
      double* o = scratch + wave * PER;
#pragma unroll
      for (int q = 0; q < FEE; ++q) o[q] = gsum[q];
#pragma unroll
      for (int q = 0; q < FEE; ++q) o[FEE + q] = ue[q];
#pragma unroll
      for (int q = 0; q < FED; ++q) o[FEE + FEE + q] = ud[q];
    }
  }
  lds_barrier();
  if (t == 0) {
    double tot[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) tot[q] = (scratch[q] + scratch[PER + q]) + (scratch[2 * PER + q] + scratch[3 * PER + q]);
    double gee[FEE];
#pragma unroll
    for (int q = 0; q < FEE; ++q) gee[q] = tot[FEE + q] - tot[q];
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < d.P) {
#pragma unroll
        for (int q = 0; q < FEE; ++q) gee[q] += lfs[p * FEE + q];
      }
    }
#pragma unroll
    for (int q = 0; q < FEE; ++q) gf.grad_en_en[(size_t)g * FEE + q] = gee[q];
#pragma unroll
    for (int q = 0; q < FED; ++q) gf.grad_en_de[(size_t)g * FED + q] = tot[FEE + FEE + q];
  }
}

// The 64 partial results of one update (lane = state), rescaled by an exact power of two so that ONE normal element
// lands in [1, 2) (the first one; which one is immaterial -- the scale cancels in everything normalised later).
// `bad` is raised when the vector cannot be carried this way: a negative / non-finite entry, or no normal entry at all
// (the zero-sum -> uniform rule of LBP.py:655-657 would act).  No branch: a bad graph keeps computing (harmlessly) and
// is handed to the exact kernel when its sweeps are over.
__device__ __forceinline__ double rescale(double r, int& bad) {
  const unsigned key = mag_key(r);
  const unsigned long long normal = __ballot(key - KEY_MIN < KEY_BAD - KEY_MIN);
  const unsigned long long wrong = __ballot(key >= KEY_BAD);
  bad |= (wrong != 0) | (normal == 0);
  const int ref = __builtin_amdgcn_readlane((int)key, normal ? __builtin_ctzll(normal) : 0);
  return __builtin_ldexp(r, 1023 - ((ref >> 20) & 0x7FF));
}

// PADX: X = d.X < 64 states, vectors and tables zero-padded to 64 on the way in, the first X entries written back.
// MULTI: the launch holds several GROUPS of graphs -- each group its own program (topology and root sequence), tables,
// messages -- described by a device table; the workgroup looks its group up and from then on runs as if launched for
// that group alone.  This is what lets a minibatch of mixed sentence shapes, each with its own roots (LBP.py:223-225,
// train_mp.py:257-299), sweep in one launch.
// NL: tables NT .. NT+NL-1 of the graph live in LDS instead of registers (a 7-table chain then fits 242 VGPRs + 70 KB, so two
// workgroups share a CU, where 8 register-resident tables allow one).
// GRAD: FactorGraph.get_unregularized_gradeint (LBP.py:301-320) of the graph as an epilogue (gradient_epilogue): the tables
// are still in registers and the messages in LDS, so the per-graph gradient costs no second pass over the tables in HBM.
template <int NT, bool PADX, bool MULTI, int NL, bool GRAD>
// (GRAD instances: two workgroups per CU -- the epilogue keeps twelve accumulators and a thread's cells of a feature tensor beside
// the tables, which does not fit the 168 registers of three; the sweeps lose nothing at two per CU, profiles/r03h_lean_kernel_occupancy.txt)
__global__ __launch_bounds__(WG, (NT >= 7 ? 1 : (NT >= 4 ? 2 : 3))) void sweep_x64_lean_kernel(SweepDev d, LeanDev f, const int32_t* groups,
                                                                                             int n_groups, GradFusedDev gf) {
  int g = blockIdx.x;
  if (MULTI) {
    // groups[k * GROUP_WORDS] = first graph of group k (ascending): the last k with start <= g
    int lo = 0, hi = n_groups - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (as_const(groups)[(size_t)mid * GROUP_WORDS] <= g) lo = mid; else hi = mid - 1;
    }
    const int32_t* G = groups + (size_t)lo * GROUP_WORDS;
    const Words16 w0 = sload16(G), w1 = sload16(G + 16), w2 = sload16(G + 32);
    auto ptr = [](int32_t a, int32_t b) { return (uintptr_t)(uint32_t)a | ((uintptr_t)(uint32_t)b << 32); };
    g -= w0.w[0];
    f.image = (const int32_t*)ptr(w0.w[2], w0.w[3]); f.readout = (const int32_t*)ptr(w0.w[4], w0.w[5]);
    f.bail = (uint8_t*)ptr(w0.w[6], w0.w[7]); d.msgs = (double*)ptr(w0.w[8], w0.w[9]);
    d.marginals = (double*)ptr(w0.w[10], w0.w[11]); d.pair_tables = (const double*)ptr(w0.w[12], w0.w[13]);
    d.pair_tab = (const int32_t*)ptr(w0.w[14], w0.w[15]); d.unary_tables = (const double*)ptr(w1.w[0], w1.w[1]);
    d.unary_tab = (const int32_t*)ptr(w1.w[2], w1.w[3]); d.status = (int32_t*)ptr(w1.w[4], w1.w[5]);
    f.n_bundles = w1.w[6]; f.HL = w1.w[7]; f.n_cprod = w1.w[8]; f.WL = w1.w[9]; f.n_ext = w1.w[10];
    d.n_msgs = w1.w[11]; d.P = w1.w[12]; d.U = w1.w[13]; d.n_vars = w1.w[14]; d.n_pair_tables = w1.w[15];
    d.n_unary_tables = w2.w[0]; f.dense = w2.w[1];
  }
  extern __shared__ double lds[];
  double* work = lds;                                        // [n_msgs + n_ext][64] scaled messages
  double* red = lds + (size_t)(d.n_msgs + f.n_ext) * 64;     // [2 parities][2 bundle slots][4][64]
  int32_t* limg = reinterpret_cast<int32_t*>(red + 4 * 256);      // [n_bundles + 1][16] micro-ops
  double2* tl = reinterpret_cast<double2*>(limg + 16 * (f.n_bundles + 1));      // [NL][8][256] tables kept in LDS
  // GRAD: per unary factor -- message slot, table total positive?, kind, observed column, label
  int32_t* gst = reinterpret_cast<int32_t*>(tl + (size_t)NL * 8 * WG);

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int X = PADX ? d.X : 64;
  const double uniform = 1.0 / (double)X;
  const double uni_l = (!PADX || lane < X) ? uniform : 0.0;      // the uniform vector, lane = state
  double* gm = d.msgs + (size_t)g * d.n_msgs * X;
  const int32_t* img_hoist = f.image + 16 * (size_t)f.n_bundles;
  const int32_t* img_cprod = img_hoist + 8 * (size_t)f.HL;
  const int32_t* img_written = img_cprod + 16 * (size_t)f.n_cprod;

  LaneGeo G;
  const int c_ = (lane & 7) | ((lane >> 1) & 8), b3_ = (lane >> 3) & 1, R_ = 4 * wave + (b3_ | ((lane >> 5) << 1));
  {
    const int b5 = lane >> 5, b4 = (lane >> 4) & 1, b2 = (lane >> 2) & 1;
    G.tm0 = (32 * b3_ + 2 * c_) * 8;
    G.tm1 = (32 * (1 - b3_) + 2 * c_) * 8;
    G.mt = 4 * R_ * 8;
    G.red_tm = (4 * R_ + 2 * b4 + b2) * 8;
    G.red_mt = (wave * 64 + 32 * b3_ + 2 * c_ + b5) * 8;
    G.up = b2 != 0;
    G.st_tm = wave == 0 && (lane & 0x28) == 0;
    G.st_mt = (lane & 0x17) == 0;
    G.wr_tm = (lane & 3) == 0;
  }

  PROBE_DECL
  PSTAMP          // 0: start
  if (t == 0) f.bail[g] = 0;
  int g_kind = 0, g_obs = 0, g_lab = 0;           // GRAD: unary factor t's kind / observed column / label, parked until the
  if (GRAD && t < d.U) {                          // unary rows have landed (the loads retire in order ahead of them)
    g_kind = gf.unary_kind[t]; g_obs = gf.unary_obs[(size_t)g * d.U + t]; g_lab = gf.unary_label[(size_t)g * d.U + t];
  }
  // ---- phase A: every HBM load of the graph is issued before anything waits: unary rows first (they are needed
  //      first and vmcnt retires in order), then the tables, then the (dense) index check ----
  bool ok = true;
  constexpr int HB = 8;                       // unary rows per wave held in registers; more go round again below
  double ur[HB];
  int uslot[HB], ufac[HB];
  {
    const Words16 hl = sload16(img_hoist + wave * 2 * f.HL);      // this wave's (unary slot, message slot) pairs, -1 padded
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const int u = hl.w[2 * j];
      uslot[j] = hl.w[2 * j + 1];
      ufac[j] = u;
      ur[j] = 0.0;
      if (u >= 0) {
        const int row = f.dense ? g * d.U + u : as_const(d.unary_tab)[(size_t)g * d.U + u];
        if ((unsigned)row >= (unsigned)d.n_unary_tables) ok = false;
        else if (!PROBED(5) && (!PADX || lane < X)) ur[j] = NT_LOAD1(&d.unary_tables[(size_t)row * X + lane]);
      }
    }
  }
  double2 tab[NT][4][2];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    if (p < d.P) {
      const int ti = f.dense ? g * d.P + p : as_const(d.pair_tab)[(size_t)g * d.P + p];
      if ((unsigned)ti >= (unsigned)d.n_pair_tables) { ok = false; continue; }
      if (PROBED(4)) continue;
      if (!PADX) {
        const double* T = d.pair_tables + (size_t)ti * 4096 + (size_t)(4 * R_) * 64 + 2 * c_;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tab[p][r][0] = NT_LOAD2(T + r * 64 + 32 * b3_);
          tab[p][r][1] = NT_LOAD2(T + r * 64 + 32 * (1 - b3_));
        }
      } else {
        const double* T = d.pair_tables + (size_t)ti * X * X;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int row = 4 * R_ + r, col = 32 * (k ? 1 - b3_ : b3_) + 2 * c_;
            tab[p][r][k].x = (row < X && col < X) ? T[(size_t)row * X + col] : 0.0;
            tab[p][r][k].y = (row < X && col + 1 < X) ? T[(size_t)row * X + col + 1] : 0.0;
          }
      }
    }
  }
  // tables NT .. NT+NL-1: straight into LDS (LDS-DMA: no registers; lane l of a wave-instruction lands at base + 16 l, which
  // is exactly the [r * 2 + k][thread] image), waited for with the register tables before the sweeps start
#pragma unroll
  for (int p = NT; p < NT + NL; ++p) {
    if (p < d.P && !PADX) {
      const int ti = f.dense ? g * d.P + p : as_const(d.pair_tab)[(size_t)g * d.P + p];
      if ((unsigned)ti >= (unsigned)d.n_pair_tables) { ok = false; continue; }
      const double* T = d.pair_tables + (size_t)ti * 4096 + (size_t)(4 * R_) * 64 + 2 * c_;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          typedef __attribute__((address_space(3))) void* lds_ptr;
          typedef const __attribute__((address_space(1))) void* glb_ptr;
          double2* dst = tl + ((p - NT) * 8 + r * 2 + k) * WG + wave * 64;          // wave-uniform
          __builtin_amdgcn_global_load_lds((glb_ptr)(T + r * 64 + 32 * (k ? 1 - b3_ : b3_)), (lds_ptr)dst, 16, 0, 0);
        }
    }
  }
  // MLBP_SWEEP_DENSE_TABLES is a statement about the index arrays; it is checked off the critical path (the loads
  // queue behind the tables and are looked at after the sweeps): a false one sends the graph to the exact kernel,
  // which reads the arrays.
  bool dense_ok = true;
  if (f.dense)
    for (int i = t; i < d.P + d.U; i += WG)
      dense_ok &= i < d.P ? d.pair_tab[(size_t)g * d.P + i] == g * d.P + i : d.unary_tab[(size_t)g * d.U + (i - d.P)] == g * d.U + (i - d.P);
  // the micro-op image, requested behind the tables (it is first read when the sweeps start, i.e. when the tables are
  // there) and parked in registers until then
  constexpr int PW = 2;
  int32_t pre[PW];
#pragma unroll
  for (int q = 0; q < PW; ++q) pre[q] = (t + q * WG < 16 * (f.n_bundles + 1)) ? f.image[t + q * WG] : 0;
  unsigned bad_key = 0;
  {
    double2* dst = reinterpret_cast<double2*>(work);
    const int x0 = (2 * t) & 63;                                  // states of the double2 this thread writes (stride 256 keeps them)
    const double2 uni2 = PADX ? make_double2(x0 < X ? uniform : 0.0, x0 + 1 < X ? uniform : 0.0) : make_double2(uniform, uniform);
    if (f.init) {
      for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = uni2;
    } else if (!PADX) {
      const double2* src = reinterpret_cast<const double2*>(gm);
      for (int i = t; i < d.n_msgs * 32; i += WG) {
        const double2 v = src[i];
        bad_key = max(bad_key, max(mag_key(v.x), mag_key(v.y)));
        dst[i] = v;
      }
    } else {
      for (int i = t; i < d.n_msgs * 64; i += WG) {
        const int slot = i >> 6, x = i & 63;
        const double v = x < X ? gm[(size_t)slot * X + x] : 0.0;
        bad_key = max(bad_key, mag_key(v));
        work[i] = v;
      }
    }
    if (t < 32) dst[d.n_msgs * 32 + t] = uni2;                                           // ext slot 0: the uniform vector
    if (t >= 32 && t < 64) dst[(d.n_msgs + f.n_ext - 1) * 32 + (t - 32)] = make_double2(1.0, 1.0);   // last ext slot: ones
  }
  PSTAMP          // 1: every load issued
  lds_barrier();        // the fill above and the unary messages below write the same slots from different waves
  // hoisted unary messages (exact values: they are outputs); further rounds only when a wave has more than HB rows
#pragma unroll
  for (int j = 0; j < HB; ++j) {
    if (uslot[j] >= 0) {
      const double s = wave_sum(ur[j]);
      const double m = renorm(ur[j], s, uni_l, true);
      bad_key = max(bad_key, mag_key(m));
      work[uslot[j] * 64 + lane] = m;
      if (GRAD && lane == 0) { gst[ufac[j]] = uslot[j]; gst[d.U + ufac[j]] = s > 0.0 ? 1 : 0; }
    }
  }
  if (GRAD && t < d.U) { gst[2 * d.U + t] = g_kind; gst[3 * d.U + t] = g_obs; gst[4 * d.U + t] = g_lab; }
  for (int j = HB; j < f.HL; ++j) {
    const const_i32p hp = as_const(img_hoist + (wave * f.HL + j) * 2);
    const int u = hp[0], slot = hp[1];
    if (u < 0) break;
    const int row = f.dense ? g * d.U + u : as_const(d.unary_tab)[(size_t)g * d.U + u];
    double r = 0.0;
    if ((unsigned)row >= (unsigned)d.n_unary_tables) ok = false;
    else if (!PADX || lane < X) r = d.unary_tables[(size_t)row * X + lane];
    const double s = wave_sum(r);
    const double m = renorm(r, s, uni_l, true);
    bad_key = max(bad_key, mag_key(m));
    work[slot * 64 + lane] = m;
    if (GRAD && lane == 0) { gst[u] = slot; gst[d.U + u] = s > 0.0 ? 1 : 0; }
  }
  if (!__syncthreads_and(ok ? 1 : 0)) {       // an out-of-range table index: skip the graph, raise the status word
    if (t == 0) atomicExch(d.status, 1);
    // "skipped" means untouched -- except that a call which was asked to initialise leaves the graph initialised
    // (FactorGraph.initialize, LBP.py:211-216), not holding whatever the caller's buffer held
    if (f.init)
      for (int i = t; i < d.n_msgs * X; i += WG) gm[i] = uniform;
    return;
  }
  PSTAMP          // 2: unary messages normalised
  // constant products: uniform x the hoisted messages a variable multiplies in, in facset order (LBP.py:381-386);
  // lists are 16 words (count, 15 slots padded with the all-ones slot), wave k & 3 takes list k
  for (int k = wave; k < f.n_cprod; k += 4) {
    const Words16 cl = sload16(img_cprod + 16 * k);
    double acc = uni_l;
#pragma unroll
    for (int q0 = 1; q0 < 16; q0 += 5) {
      if (cl.w[0] >= q0) {
        double m[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) m[q] = work[cl.w[q0 + q] * 64 + lane];
#pragma unroll
        for (int q = 0; q < 5; ++q) acc *= m[q];
      }
    }
    bad_key = max(bad_key, mag_key(acc));               // a non-finite intermediate stays non-finite: nan_to_num territory
    work[(d.n_msgs + 1 + k) * 64 + lane] = acc;
  }
  if (__syncthreads_or(bad_key >= KEY_BAD ? 1 : 0)) {
    if (t == 0) f.bail[g] = 1;                    // bail codes: 1 prologue, 2 main loop, 3 final pass
    return;
  }
  PSTAMP          // 3: constant products
#ifdef MLBP_LEAN_PROBE
  PSTAMP_VM       // 4: tables arrived
#else
  PSTAMP
#endif

  // ---- main loop: identical in all four waves; one barrier per bundle.  The micro-ops sit in LDS; a bundle's 16 words
  //      are read (broadcast) one bundle ahead ----
  if (NL > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA tables have landed (the barrier below publishes them)
#pragma unroll
  for (int q = 0; q < PW; ++q)
    if (t + q * WG < 16 * (f.n_bundles + 1)) limg[t + q * WG] = pre[q];
  for (int i = t + PW * WG; i < 16 * (f.n_bundles + 1); i += WG) limg[i] = f.image[i];
  lds_barrier();
  char* wb = reinterpret_cast<char*>(work);
  int parity = 0, bad = 0;
  double carry = 1.0;
  const int4* li = reinterpret_cast<const int4*>(limg);
  int4 A0 = li[0], A1 = li[1], B0 = li[2], B1 = li[3];
  for (int k = 0; k < f.n_bundles && !PROBED(0); ++k) {
    const int4 nA0 = li[4 * k + 4], nA1 = li[4 * k + 5], nB0 = li[4 * k + 6], nB1 = li[4 * k + 7];     // the image is padded by one bundle
    const int fA = __builtin_amdgcn_readfirstlane(A0.x), fB = __builtin_amdgcn_readfirstlane(B0.x);
    if (fA & UOP_VAR) {
      // a lone variable update: the product, lane = state, the same in every wave; no contraction, no barrier
      // a product of more than four sources is a chain of links; the running product stays in a register between
      // them and only the last link stores (every wave stores the same value once: no read-modify-write of a slot
      // that another wave may still be writing)
      const int nsrc = (fA >> UOP_NSRC_SHIFT) & 15;
      double m = work[(A0.y >> 3) + lane];
      if (fA & UOP_CARRY_IN) m *= carry;
      if (nsrc > 1) m *= work[(A0.z >> 3) + lane];
      if (nsrc > 2) m *= work[(A0.w >> 3) + lane];
      if (nsrc > 3) m *= work[(A1.x >> 3) + lane];
      if (fA & UOP_CARRY_OUT) carry = m;
      else work[(A1.y >> 3) + lane] = m;
    } else {
      char* redP = reinterpret_cast<char*>(red) + parity * 4096;
      front<NT, NL>(tab, tl + t, wb, fA, A0, A1, G, redP);
      const bool two = !(fB & UOP_NOP);
      if (two) front<NT, NL>(tab, tl + t, wb, fB, B0, B1, G, redP + 2048);
      lds_barrier();
      const double* rd = reinterpret_cast<const double*>(redP);
      double rA = rd[lane];
      if (fA & UOP_MT) rA = (rA + rd[64 + lane]) + (rd[128 + lane] + rd[192 + lane]);
      work[(A1.z >> 3) + lane] = rescale(rA, bad);
      if (two) {
        double rB = rd[256 + lane];
        if (fB & UOP_MT) rB = (rB + rd[320 + lane]) + (rd[384 + lane] + rd[448 + lane]);
        work[(B1.z >> 3) + lane] = rescale(rB, bad);
      }
      parity ^= 1;
    }
    A0 = nA0; A1 = nA1; B0 = nB0; B1 = nB1;
  }
  if (bad) {                                      // the same in every wave: the inputs were identical
    if (t == 0) f.bail[g] = 2;
    return;
  }
  lds_barrier();
  PSTAMP          // 5: main loop
  // ---- read-out (VariableNode.get_marginal, LBP.py:392-400) straight from the scaled messages: the marginal is
  //      normalised, so the scales cancel; constant part = the variable's constant product ----
  double marg[2] = {0.0, 0.0};
  bool bad_out = !dense_ok;
  if (f.readout && !PROBED(3)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = wave + 4 * j;
      if (v < d.n_vars) {
        const Words16 rl = sload16(f.readout + 16 * v);
        double acc = work[rl.w[1] * 64 + lane];
#pragma unroll
        for (int q0 = 2; q0 < 16; q0 += 7) {
          if (rl.w[0] >= q0) {
            double m[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) m[q] = work[rl.w[q0 + q] * 64 + lane];
#pragma unroll
            for (int q = 0; q < 7; ++q) acc *= m[q];
          }
        }
        const unsigned key = wave_max_u32(mag_key(acc));
        bad_out |= key >= KEY_BAD || key < KEY_MIN;
        marg[j] = acc / wave_sum(acc);
      }
    }
    for (int v = wave + 8; v < d.n_vars; v += 4) {       // graphs with more than 8 variables: one at a time
      const Words16 rl = sload16(f.readout + 16 * v);
      double acc = work[rl.w[1] * 64 + lane];
      for (int q = 2; q <= rl.w[0]; ++q) acc *= work[as_const(f.readout + 16 * v)[q] * 64 + lane];
      const unsigned key = wave_max_u32(mag_key(acc));
      bad_out |= key >= KEY_BAD || key < KEY_MIN;
      if (!bad_out) d.marginals[((size_t)g * d.n_vars + v) * 64 + lane] = acc / wave_sum(acc);   // (redone by the exact kernel when flagged)
    }
  }
  lds_barrier();        // the normalisation below rewrites slots the read-out above has just read
  // ---- the deferred normalisations (only when the messages go back to memory): exactly the slots the program wrote,
  //      wave w takes its WL-entry list, four at a time ----
  if ((f.keep || GRAD) && !PROBED(1)) {      // (GRAD: the same validity check guards the messages the gradient reads)
    for (int j0 = 0; j0 < f.WL; j0 += 4) {
      const const_i32p wl = as_const(img_written + wave * f.WL + j0);
      int slot[4];
      double v[4], s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { slot[j] = wl[j]; v[j] = slot[j] >= 0 ? work[slot[j] * 64 + lane] : 1.0; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned key = wave_max_u32(mag_key(v[j]));
        bad_out |= key >= KEY_BAD || key < KEY_MIN;       // a variable product underflowed or vanished
        s[j] = wave_sum(v[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (slot[j] >= 0) work[slot[j] * 64 + lane] = v[j] / s[j];
    }
  }
  PSTAMP          // 6: read-out + final normalisation
  if (__syncthreads_or(bad_out ? 1 : 0)) {         // nothing of a flagged graph is written back
    if (t == 0) f.bail[g] = 3;
    return;
  }
  if (f.keep && !PADX) {
    const double2* src = reinterpret_cast<const double2*>(work);
    double2* dst = reinterpret_cast<double2*>(gm);
    for (int i = t; i < d.n_msgs * 32 && !PROBED(2); i += WG) NT_STORE2(dst + i, src[i]);
  } else if (f.keep) {
    for (int i = t; i < d.n_msgs * 64; i += WG)
      if ((i & 63) < X) gm[(size_t)(i >> 6) * X + (i & 63)] = work[i];
  }
  PSTAMP          // 7: write-back issued
  if (f.readout && !PROBED(3)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = wave + 4 * j;
      if (v < d.n_vars) NT_STORE1(&d.marginals[((size_t)g * d.n_vars + v) * 64 + lane], marg[j]);
    }
  }
  PSTAMP          // 8: marginals issued
  if (GRAD) {                 // (the verdict barrier above has published the final normalisation)
    gradient_epilogue<NT>(d, gf, tab, reinterpret_cast<const char*>(work), gst, reinterpret_cast<double*>(red), G, g, R_, c_, b3_);
  }
#ifdef MLBP_LEAN_PROBE
  PSTAMP_VM       // 9: stores drained
  PFLUSH
#endif
}

int ensure_lds(const void* fn, size_t bytes) {
  static std::vector<std::pair<const void*, size_t>> granted;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& g : granted)
    if (g.first == fn && g.second >= bytes) return MLBP_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  granted.push_back({fn, bytes});
  return MLBP_OK;
}

}  // namespace

namespace mlbp {

// FusedProgram -> micro-ops.  Every operand becomes an LDS byte offset (slot * 512); a variable product with more
// than four sources is split into a chain of variable-only micro-ops ("links") that hand the running product on in a
// register (UOP_CARRY_OUT / UOP_CARRY_IN); only the last link stores.
// Image: bundles [n_bundles][16] | per-wave hoist lists [4][HL][2] | constant-product lists [n_cprod][16] |
// per-wave written-slot lists [4][WL] | one bundle of padding (the loop prefetches one bundle past the end).
void build_lean_program(const FusedProgram& fp, int n_msgs, LeanProgram& out) {
  out = LeanProgram();
  if (fp.has_unary_fops) { out.why = "in-loop unary updates (not hoistable)"; return; }
  std::vector<int32_t> U;                            // micro-ops, 8 words each
  std::vector<char> second;                          // micro-op i is the second member of a bundle
  auto emit_var = [&](const int32_t* src, int n, int c) {      // work[c] = prod(src[0..n))
    int done = 0;
    do {
      int32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int k = 0;
      const bool first = done == 0;
      while (k < 4 && done < n) w[1 + k++] = src[done++] * 512;
      w[0] = UOP_VAR | (k << UOP_NSRC_SHIFT) | (first ? 0 : UOP_CARRY_IN) | (done < n ? UOP_CARRY_OUT : 0);
      w[5] = c * 512;
      U.insert(U.end(), w, w + 8);
      second.push_back(0);
    } while (done < n);
  };
  const int n_fops = (int)fp.fops.size() / 8;
  for (int i = 0; i < n_fops; ++i) {
    const int32_t* w = &fp.fops[8 * (size_t)i];
    const int kind = w[0] & 0xFF;
    if (kind == FOP_VAR) { emit_var(&fp.psrcs[w[1]], w[2], w[3]); continue; }
    // a bundle: pre-chains of both members first (they touch slots disjoint from the partner's), then the members
    const int members = (w[0] & FOP_BUNDLED) ? 2 : 1;
    for (int m = 0; m < members; ++m) {
      const int32_t* q = &fp.fops[8 * (size_t)(i + m)];
      const int kd = q[0] & 0xFF;
      if ((kd == FOP_VAR_PAIR_TM || kd == FOP_VAR_PAIR_MT) && q[2] > 4) emit_var(&fp.psrcs[q[1]], q[2], q[3]);
    }
    for (int m = 0; m < members; ++m) {
      const int32_t* q = &fp.fops[8 * (size_t)(i + m)];
      const int kd = q[0] & 0xFF;
      int32_t u[8] = {0, 0, 0, 0, 0, -1, 0, 0};
      if (kd == FOP_PAIR_TM || kd == FOP_PAIR_MT) {
        u[0] = (kd == FOP_PAIR_MT ? UOP_MT : 0) | (q[1] << UOP_PSLOT_SHIFT) | (1 << UOP_NSRC_SHIFT);
        u[1] = q[2] * 512;
        u[6] = q[3] * 512;
      } else {
        const bool chained = q[2] > 4;
        const int n = chained ? 1 : q[2];
        u[0] = (kd == FOP_VAR_PAIR_MT ? UOP_MT : 0) | (q[4] << UOP_PSLOT_SHIFT) | (n << UOP_NSRC_SHIFT);
        if (chained) u[1] = q[3] * 512;
        else for (int k = 0; k < n; ++k) u[1 + k] = fp.psrcs[q[1] + k] * 512;
        u[5] = chained ? -1 : q[3] * 512;           // the variable->factor message itself (dropped below when dead)
        u[6] = q[5] * 512;
      }
      U.insert(U.end(), u, u + 8);
      second.push_back(m == 1);
    }
    i += members - 1;
  }
  const int n_uops = (int)U.size() / 8;
  // a fused variable->factor message is stored only when something reads the slot before its next write, or when it
  // is the slot's final value (the messages are an output of the call)
  for (int i = 0; i < n_uops; ++i) {
    int32_t* u = &U[8 * (size_t)i];
    if ((u[0] & UOP_VAR) || u[5] < 0) continue;
    const int c = u[5];
    bool needed = true;
    for (int j = i + 1; j < n_uops; ++j) {
      const int32_t* v = &U[8 * (size_t)j];
      const int n = (v[0] >> UOP_NSRC_SHIFT) & 15;
      bool reads = false;
      for (int k = 0; k < n; ++k) reads |= v[1 + k] == c;
      if (reads) break;
      const bool writes = v[5] == c || (!(v[0] & UOP_VAR) && v[6] == c);
      if (writes) { needed = false; break; }
    }
    if (!needed) u[5] = -1;
  }
  for (int i = 0; i < n_uops; ++i) {
    int32_t* u = &U[8 * (size_t)i];
    if (!(u[0] & UOP_VAR) && u[5] >= 0) u[0] |= UOP_STORE_VF;
    if (u[5] < 0) u[5] = 0;
  }
  std::vector<int32_t>& I = out.image;
  const int32_t nop[8] = {UOP_NOP, 0, 0, 0, 0, -1, 0, 0};
  for (int i = 0; i < n_uops; ++i) {
    I.insert(I.end(), U.begin() + 8 * (size_t)i, U.begin() + 8 * (size_t)i + 8);
    if (i + 1 < n_uops && second[i + 1]) { ++i; I.insert(I.end(), U.begin() + 8 * (size_t)i, U.begin() + 8 * (size_t)i + 8); }
    else I.insert(I.end(), nop, nop + 8);
  }
  out.n_bundles = (int)I.size() / 16;
  // per-wave hoist lists: entry h goes to wave h & 3
  const int n_hoist = (int)fp.hoist.size() / 2;
  out.HL = std::max(8, ((n_hoist + 3) / 4 + 7) / 8 * 8);
  {
    std::vector<int32_t> hl(4 * (size_t)out.HL * 2, -1);
    for (int h = 0; h < n_hoist; ++h) {
      hl[((size_t)(h & 3) * out.HL + (h >> 2)) * 2] = fp.hoist[2 * h];
      hl[((size_t)(h & 3) * out.HL + (h >> 2)) * 2 + 1] = fp.hoist[2 * h + 1];
    }
    I.insert(I.end(), hl.begin(), hl.end());
  }
  // constant-product lists, 16 words each, padded with the all-ones ext slot
  out.n_cprod = fp.n_cprod;
  const int ones = n_msgs + 1 + fp.n_cprod;
  out.cprods.clear();
  for (size_t at = 0; at < fp.cpw.size();) {
    const int cnt = fp.cpw[at];
    if (cnt > 15) { out.why = "a constant product of more than 15 messages"; out.image.clear(); return; }
    int32_t l[16];
    l[0] = cnt;
    for (int q = 0; q < 15; ++q) l[1 + q] = q < cnt ? fp.cpw[at + 1 + q] : ones;
    I.insert(I.end(), l, l + 16);
    out.cprods.push_back(std::vector<int32_t>(fp.cpw.begin() + at + 1, fp.cpw.begin() + at + 1 + cnt));
    at += 1 + cnt;
  }
  out.hoisted.assign(n_msgs, 0);
  for (int h = 0; h < n_hoist; ++h) out.hoisted[fp.hoist[2 * h + 1]] = 1;
  // per-wave written-slot lists
  const int n_written = (int)fp.written.size();
  out.WL = std::max(4, ((n_written + 3) / 4 + 3) / 4 * 4);
  {
    std::vector<int32_t> wl(4 * (size_t)out.WL, -1);
    for (int i = 0; i < n_written; ++i) wl[(size_t)(i & 3) * out.WL + (i >> 2)] = fp.written[i];
    I.insert(I.end(), wl.begin(), wl.end());
  }
  for (int q = 0; q < 16; ++q) I.push_back(q == 0 || q == 8 ? UOP_NOP : 0);
  out.ok = true;
}

// Read-out lists of the lean kernel: per variable 16 words -- count (base included), base slot (the variable's constant
// product, or the uniform vector), then the varying incoming slots.  False when a variable has more than 15 entries.
bool build_lean_readout(const LeanProgram& lp, int n_msgs, int n_vars, const int32_t* in_off, const int32_t* in_slots,
                        std::vector<int32_t>& image) {
  image.assign(16 * (size_t)n_vars, n_msgs);
  for (int v = 0; v < n_vars; ++v) {
    std::vector<int32_t> consts, vars;
    for (int q = in_off[v]; q < in_off[v + 1]; ++q) (lp.hoisted[in_slots[q]] ? consts : vars).push_back(in_slots[q]);
    int base = n_msgs;                               // the uniform vector
    if (!consts.empty()) {
      size_t k = 0;
      for (; k < lp.cprods.size(); ++k)
        if (lp.cprods[k] == consts) break;
      if (k < lp.cprods.size()) base = n_msgs + 1 + (int)k;
      else { vars.insert(vars.begin(), consts.begin(), consts.end()); }      // no matching product: multiply them in
    }
    if (vars.size() > 14) return false;
    int32_t* l = &image[16 * (size_t)v];
    l[0] = 1 + (int)vars.size();
    l[1] = base;
    const int ones = n_msgs + 1 + lp.n_cprod;
    for (int q = 0; q < 14; ++q) l[2 + q] = q < (int)vars.size() ? vars[q] : ones;
  }
  return true;
}

#ifdef MLBP_LEAN_PROBE
extern "C" int mlbp_debug_lean_probe(int mask, void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_mask), &mask, sizeof(mask)));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_buf), &p, sizeof(p)));
  return MLBP_OK;
}
#endif

// Does the lean kernel apply to this (program, arguments) pair?  Fills the device-side descriptions when it does.
// grad: the call's gradient is fused into the launch (the caller has checked that it can be); a gradient request without
// it is served by the standalone kernel afterwards, from the messages in memory.
static int lean_plan(const mlbp_program* prog, const mlbp_sweep_args* a, bool grad, bool* ok, SweepDev* d, LeanDev* f, size_t* lds) {
  *ok = false;
  const LeanProgram& lp = prog->lean;
  if (!lp.ok || !prog->d_limage || a->X > 64 || a->X < 2 || !a->normalize_messages || prog->P < 1 || prog->P > 8) return MLBP_OK;
  const bool padx = a->X < 64;
  if (padx && prog->P > 4) return MLBP_OK;
  if (a->gradient && padx) return MLBP_OK;
  if (grad && (prog->P > 3 || prog->U > WG)) return fail(MLBP_EINVAL, "lean kernel: fused gradient needs P <= 3 and U <= %d", WG);
  if (a->flags & MLBP_SWEEP_APPROX_INFERENCE) return MLBP_OK;
  if (a->marginals && !prog->d_lreadout && !padx) return MLBP_OK;
  const int n_ext = 2 + lp.n_cprod;               // uniform, the constant products, ones
  *lds = ((size_t)(prog->n_msgs + n_ext) * 64 + 4 * 256) * sizeof(double) + 16 * (size_t)(lp.n_bundles + 1) * sizeof(int32_t);
  if (prog->P == 7 && !padx) *lds += 32 * 1024;   // the seventh table lives in LDS
  if (grad) *lds += 5 * (size_t)prog->U * sizeof(int32_t);
#ifdef MLBP_LEAN_EXTRA_LDS            // diagnostic build (tools/lean_occupancy.py): fewer workgroups per CU, same kernel
  *lds += MLBP_LEAN_EXTRA_LDS;
#endif
  if (*lds > 80 * 1024) return MLBP_OK;           // large graphs: the generic kernel's rules apply
  const bool dense = (a->flags & MLBP_SWEEP_DENSE_TABLES) != 0;
  if (dense && ((int64_t)a->B * prog->P > a->n_pair_tables || (int64_t)a->B * prog->U > a->n_unary_tables))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: MLBP_SWEEP_DENSE_TABLES needs B*P pair tables and B*U unary columns");
  mlbp_program* mp = const_cast<mlbp_program*>(prog);
  if (mp->bail_cap < a->B)
    if (int e = mlbp_program_reserve(mp, a->B)) return e;
  d->pair_tables = a->pair_tables; d->pair_tab = a->pair_tab;
  d->unary_tables = a->unary_tables; d->unary_tab = a->unary_tab;
  d->msgs = a->msgs;
  d->ops = nullptr; d->srcs = nullptr; d->sweeps = nullptr; d->pairseq = nullptr;
  d->status = prog->d_status;
  d->n_sweeps = prog->n_sweeps; d->n_msgs = prog->n_msgs; d->P = prog->P; d->U = prog->U; d->X = a->X;
  d->n_pair_tables = a->n_pair_tables; d->n_unary_tables = a->n_unary_tables;
  d->marginals = padx ? nullptr : a->marginals; d->readout = nullptr; d->n_vars = prog->n_vars;
  d->only = nullptr; d->fill_uniform = 0; d->approx_k = 0;
  f->image = prog->d_limage; f->readout = (a->marginals && !padx) ? prog->d_lreadout : nullptr; f->bail = mp->d_bail;
  f->n_bundles = lp.n_bundles; f->HL = lp.HL; f->n_cprod = lp.n_cprod; f->WL = lp.WL;
  f->n_ext = n_ext; f->init = a->init_messages; f->dense = dense ? 1 : 0;
  // the messages go back to memory unless the caller waives them and takes the fused read-outs instead (a gradient that
  // is NOT fused reads them from memory)
  const bool waived = (a->flags & MLBP_SWEEP_NO_MESSAGE_WRITEBACK) && !padx && (a->marginals || grad) && (!a->gradient || grad);
  f->keep = waived ? 0 : 1;                       // (small X: the read-out is a separate launch over the messages)
  *ok = true;
  return MLBP_OK;
}

typedef void (*lean_fn)(SweepDev, LeanDev, const int32_t*, int, GradFusedDev);

template <bool MULTI>
static lean_fn pick_lean(int P, bool padx) {
  if (padx) {
    switch (P) {
      case 1: return sweep_x64_lean_kernel<1, true, MULTI, 0, false>;
      case 2: return sweep_x64_lean_kernel<2, true, MULTI, 0, false>;
      case 3: return sweep_x64_lean_kernel<3, true, MULTI, 0, false>;
      default: return sweep_x64_lean_kernel<4, true, MULTI, 0, false>;
    }
  }
  switch (P) {
    case 1: return sweep_x64_lean_kernel<1, false, MULTI, 0, false>;
    case 2: return sweep_x64_lean_kernel<2, false, MULTI, 0, false>;
    case 3: return sweep_x64_lean_kernel<3, false, MULTI, 0, false>;
    case 4: return sweep_x64_lean_kernel<4, false, MULTI, 0, false>;
    case 5: case 6: return sweep_x64_lean_kernel<6, false, MULTI, 0, false>;
    case 7: return sweep_x64_lean_kernel<6, false, MULTI, 1, false>;     // six tables in registers, the seventh in LDS: two workgroups per CU
    default: return sweep_x64_lean_kernel<8, false, MULTI, 0, false>;    // part of the tables lives in the accumulator registers
  }
}
static lean_fn pick_lean_grad(int P) {
  switch (P) {
    case 1: return sweep_x64_lean_kernel<1, false, false, 0, true>;
    case 2: return sweep_x64_lean_kernel<2, false, false, 0, true>;
    default: return sweep_x64_lean_kernel<3, false, false, 0, true>;
  }
}

int launch_lean_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, const GradFusedDev* gf, void* stream, bool* launched) {
  *launched = false;
  SweepDev d;
  LeanDev f;
  size_t lds = 0;
  bool ok = false;
  if (int e = lean_plan(prog, a, gf != nullptr, &ok, &d, &f, &lds)) return e;
  if (!ok) return MLBP_OK;
  lean_fn k = gf ? pick_lean_grad(prog->P) : pick_lean<false>(prog->P, a->X < 64);
  if (int e = ensure_lds((const void*)k, lds)) return e;
  hipLaunchKernelGGL(k, dim3(a->B), dim3(WG), lds, (hipStream_t)stream, d, f, nullptr, 0, gf ? *gf : GradFusedDev{});
  HIP_TRY(hipGetLastError());
  *launched = true;
  return MLBP_OK;
}

// Several (program, arguments) groups in ONE launch of the lean kernel.  *launched stays false when some group does not
// qualify (the caller then runs the groups one by one).  The group table lives in a device buffer owned by the FIRST
// program of the call (like its redo flags: one stream at a time per program) and is uploaded only when its contents
// differ from the last call's, so a repeated call -- the trainer's every step -- is enqueue-only and can be captured
// into a HIP graph after one warm-up call.
int launch_lean_groups(const mlbp_program* const* progs, const mlbp_sweep_args* args, int n_groups, void* stream, bool* launched) {
  *launched = false;
  if (n_groups < 1) return MLBP_OK;
  std::vector<int32_t> table((size_t)n_groups * GROUP_WORDS, 0);
  size_t lds_max = 0;
  int p_max = 0, total = 0;
  SweepDev d0;
  LeanDev f0;
  for (int k = 0; k < n_groups; ++k) {
    if (!progs[k] || args[k].X != 64) return MLBP_OK;
    for (int j = 0; j < k; ++j)
      if (progs[j] == progs[k]) return MLBP_OK;      // two groups would share one set of redo flags
    SweepDev d;
    LeanDev f;
    size_t lds = 0;
    bool ok = false;
    if (int e = lean_plan(progs[k], &args[k], false, &ok, &d, &f, &lds)) return e;
    if (!ok) return MLBP_OK;
    if (k == 0) { d0 = d; f0 = f; }
    else if (f.init != f0.init || f.keep != f0.keep || (f.readout != nullptr) != (f0.readout != nullptr)) return MLBP_OK;
    lds_max = std::max(lds_max, lds + (progs[k]->P < 7 ? 32 * 1024 : 0));      // (a 7-table group may sit beside it: its LDS table)
    p_max = std::max(p_max, (int)progs[k]->P);
    int32_t* w = &table[(size_t)k * GROUP_WORDS];
    auto put = [&](int at, const void* p) { const uintptr_t v = (uintptr_t)p; w[at] = (int32_t)(uint32_t)v; w[at + 1] = (int32_t)(uint32_t)(v >> 32); };
    w[0] = total; w[1] = args[k].B;
    put(2, f.image); put(4, f.readout); put(6, f.bail); put(8, d.msgs); put(10, d.marginals); put(12, d.pair_tables);
    put(14, d.pair_tab); put(16, d.unary_tables); put(18, d.unary_tab); put(20, d.status);
    w[22] = f.n_bundles; w[23] = f.HL; w[24] = f.n_cprod; w[25] = f.WL; w[26] = f.n_ext; w[27] = d.n_msgs; w[28] = d.P; w[29] = d.U;
    w[30] = d.n_vars; w[31] = d.n_pair_tables; w[32] = d.n_unary_tables; w[33] = f.dense;
    total += args[k].B;
  }
  // one device copy per distinct table (stream-ordered upload on first sight, none afterwards): a captured graph keeps its own
  mlbp_program* owner = const_cast<mlbp_program*>(progs[0]);
  int32_t* d_gtable = nullptr;
  if (int e = group_table_device(owner->gtables, table, stream, &d_gtable)) return e;
  lean_fn k = pick_lean<true>(p_max, false);
  if (int e = ensure_lds((const void*)k, lds_max)) return e;
  hipLaunchKernelGGL(k, dim3(total), dim3(WG), lds_max, (hipStream_t)stream, d0, f0, d_gtable, n_groups, GradFusedDev{});
  HIP_TRY(hipGetLastError());
  *launched = true;
  return MLBP_OK;
}

}  // namespace mlbp
