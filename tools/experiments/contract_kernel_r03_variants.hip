// Round-3 snapshot of mlbp_gemm.hip with the overlap experiments that did not pay (MLBP_CONTRACT_EXPERIMENT = 1..4,
// MLBP_CONTRACT_STAGGER, MLBP_CONTRACT_16X16; DESIGN/LAB_NOTES 4.2b, profiles/r03e_contract_kernel_experiments.txt).  Not built into
// libmlbp.so; kept so that the measurements can be repeated (drop it over csrc/mlbp_gemm.hip of commit ce7d003).
// Shared pairwise tables at large state spaces (X = 128 .. 512): every factor->variable update of the WHOLE batch is
// one dense contraction on the matrix cores, written here (no library call).
//
// With one table behind factor p for every graph (the reference's layout, LBP.py:456-467; X = |V_en| there) the update
// of all B graphs is      OUT[X x B] = T[X x X] . M[X x B]      (or T^T . M)
// -- SURVEY.md section 8(d)'s "shared-table (GEMM/MFMA) variant" of BASELINE config 5.  At X = 64 the whole sweep
// fits one workgroup per 16 graphs (mlbp_shared.hip); at X = 512 a table is 2 MiB, so the sweep runs update by update
// over the batch, ONE launch of contract_kernel per update:
//
//   prologue   the input message of N_T = 16 or 32 graphs is formed and parked in LDS: either a stored slot, or --
//              fused -- the variable->factor product of VariableNode.update_message_to (LBP.py:377-389: uniform x the
//              listed incoming messages, nan_to_num after each product, renormalised), which is also stored;
//   main loop  four v_mfma_f64_4x4x4_f64 per 16 x 16 x 4 product (this GPU sustains 70 TFLOP/s on that form, 48 on the
//              single v_mfma_f64_16x16x4_f64; tools/mfma_peak.hip): wave w owns the row tiles w, w+4, ...; its A
//              fragments (table rows) come straight
//              from L2 as one coalesced 16-byte load per lane and two k-steps, out of a copy of the table laid out in
//              fragment order once per call (table_frag_kernel), with a register double buffer; the B fragments
//              (messages) are 8-byte LDS reads out of a [graph][state + 2] image (conflict-free);
//   epilogue   the accumulators go back through the same LDS image transposed, so that Message.renormalize
//              (LBP.py:649-657) sees whole columns and the result leaves in full 512-byte rows.
//
// float32 tables (MLBP_SWEEP_PAIR_TABLES_F32, the "batched f32 MFMA message contraction" of config 5): the same
// structure on v_mfma_f32_16x16x4_f32 -- table and message fragments in float32, products summed in float32 over 64
// states at a time, the 64-state partial sums added into float64 accumulators (tolerance study: DESIGN.md 4.2b).
//
// Same updates in the same order as every other path (the fused program of build_fused_program); only the summation
// order inside a contraction differs.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "mlbp_internal.h"

namespace mlbp {
namespace {

constexpr int WG = 256;

__device__ __forceinline__ double nan_to_num(double x) {
  if (x != x) return 0.0;
  if (x == __builtin_huge_val()) return DBL_MAX;
  if (x == -__builtin_huge_val()) return -DBL_MAX;
  return x;
}

__device__ __forceinline__ double wave_sum64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ double block_sum(double v, double* scratch /*[4]*/) {
  v = wave_sum64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// ---- the table in fragment order ----------------------------------------------------------------------------------
// A operand of the contraction: Aop[i][k] = T[i][k] (out = T . m) or T[k][i] (out = m^T . T), optionally times a feature
// plane (the gradient's T (.) phi_k).  MFMA 16x16x4 lane map: lane l holds A[i = l & 15][k = l >> 4].
//   float64: frag[rt][kp][l][e]  = Aop[16 rt + (l & 15)][8 kp + 4 e + (l >> 4)],  e = 0, 1      (one double2 per lane)
//   float32: frag[rt][kq][l][e]  = Aop[16 rt + (l & 15)][16 kq + 4 e + (l >> 4)], e = 0 .. 3    (one float4 per lane)
// XA >= X is the padded size the contraction runs at (a multiple of 128): rows and columns past X are zero, so that any
// vocabulary size (X = len(en_domain), train_mp.py:591-594) takes the matrix-core path.
template <typename TT, int E>
__global__ void table_frag_kernel(const TT* T, const double* plane, int X, int XA, int transpose, TT* frag) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one output element
  if (idx >= (size_t)XA * XA) return;
  const int e = idx % E, l = (idx / E) % 64;
  const size_t blk = idx / (E * 64);
  const int KB = XA / (4 * E);
  const int kb = blk % KB, rt = blk / KB;
  const int i = 16 * rt + (l & 15), k = 4 * E * kb + 4 * e + (l >> 4);
  double v = 0.0;
  if (i < X && k < X) {
    const size_t at = transpose ? (size_t)k * X + i : (size_t)i * X + k;
    v = (double)T[at];
    if (plane) v *= plane[at];
  }
  frag[idx] = (TT)v;
}

struct ContractDev {
  const void* frag;          // the table in fragment order (table_frag_kernel)
  const double* in;          // source messages: (b, slot, x) at in[b * in_ld + slot * X + x]
  double* out;               // results:         (b, slot, x) at out[b * out_ld + slot * X + x]
  int32_t src[8];            // source slots of the fused variable product, reference order (n_src <= 8)
  size_t in_ld, out_ld;
  int32_t n_src;             // 0: the input is slot in_slot as it stands
  int32_t in_slot, vf_slot, dst_slot;      // vf_slot: where the variable->factor message itself is stored, or -1
  int32_t B, normalize;
  int32_t X;                 // states (<= the 64 RT the kernel instance runs at; smaller: zero-padded operands, PADDED instances)
};

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename TT> struct Frag;
template <> struct Frag<double> {
  typedef double2 vec;                      // two k-steps per 16-byte load
  static constexpr int KSTEPS = 2;
  typedef double lds_t;
};
template <> struct Frag<float> {
  typedef float4 vec;                       // four k-steps per 16-byte load
  static constexpr int KSTEPS = 4;
  typedef float lds_t;
};

// The contraction runs at XA = 64 * RT states; N_T = 16 * NCT graphs per pass.  PADDED: the messages have d.X <= XA
// states (rows of d.X doubles in memory, any parity: 8-byte accesses), the operands are zero beyond.
// HALVES = 2: the workgroup takes 2 * N_T graphs as two passes over the table, and the passes' memory phases meet the other
// pass's matrix phase -- the second pass's source messages are requested before the first pass's main loop and land in
// registers under it; the first pass's results are stored (fire and forget) under the second pass's loop.  With ONE pass
// per workgroup every workgroup of the launch is in its HBM phase at the same time and the matrix cores wait (a third of an
// update at X = 512, tools/contract_probe.py); two co-resident workgroups cannot be staggered to the same effect.
template <typename TT, int RT, int NCT, int DEPTH, int NW, bool PADDED, int HALVES>
__global__ __launch_bounds__(64 * NW, (NW == 16 && HALVES == 1 && NCT == 1) ? 8 : 1) void contract_kernel(ContractDev d) {
  constexpr int XA = 64 * RT, NT_G = 16 * NCT, XP = XA + 2;
  const int X = PADDED ? d.X : XA;
  constexpr int RTW = 4 * RT / NW;                               // 16-row tiles per wave (wave w owns tiles w, w + NW, ...)
  static_assert((4 * RT) % NW == 0 && NT_G % NW == 0, "");
  constexpr int KS = Frag<TT>::KSTEPS, KB = XA / (4 * KS);      // 16-byte fragment blocks along k
  typedef typename Frag<TT>::vec avec;
  typedef typename Frag<TT>::lds_t mt_t;
  extern __shared__ double lds_raw[];
  mt_t* Mt = reinterpret_cast<mt_t*>(lds_raw);                   // [NT_G][XP] input messages (later: the results, float64)
  double* Ot = lds_raw;                                          // [NT_G][XP] float64 view for the epilogue
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double uniform = 1.0 / (double)X;

  // ---- prologue pieces: the input messages of a pass's graphs -> LDS.  A wave takes NT_G / NW consecutive graphs, up to
  //      four at a time, so that one round of memory latency covers four graphs; lane l holds states 2l, 2l+1 (+128 j) ----
  constexpr int GPW = NT_G / NW, GU = GPW < 4 ? GPW : 4;        // graphs per wave, and how many of them go through together
  constexpr int H = RT / 2 + (RT & 1);                         // double2 pieces per lane (odd RT: last half-used)
  constexpr int PF = 2;                                         // source messages requested together
  static_assert(RT % 2 == 0, "X must be a multiple of 128 here");
  static_assert(HALVES == 1 || GPW == GU, "the overlapped form keeps one batch of graphs per wave in registers");
  // states 2 (lane + 64 j), + 1 of a row of X doubles; beyond X: zero
  auto load2 = [&](const double* row, int j) {
    if (!PADDED) return reinterpret_cast<const double2*>(row)[lane + 64 * j];
    const int x0 = 2 * (lane + 64 * j);
    return make_double2(x0 < X ? row[x0] : 0.0, x0 + 1 < X ? row[x0 + 1] : 0.0);
  };
  auto uniform2 = [&](int j) {
    const int x0 = 2 * (lane + 64 * j);
    return (!PADDED) ? make_double2(uniform, uniform) : make_double2(x0 < X ? uniform : 0.0, x0 + 1 < X ? uniform : 0.0);
  };
  auto store2 = [&](double* row, int j, double2 val) {
    if (!PADDED) { reinterpret_cast<double2*>(row)[lane + 64 * j] = val; return; }
    const int x0 = 2 * (lane + 64 * j);
    if (x0 < X) row[x0] = val.x;
    if (x0 + 1 < X) row[x0 + 1] = val.y;
  };
  // graph u of the batch that starts at the wave's graph g4 of the pass whose first graph is b0
  auto graph_of = [&](int b0, int g4, int u) { return b0 + wave * GPW + g4 + u; };
  // requests sources q0, q0 + 1 (of the fused product; q0 = 0 with n_src = 0: the stored slot) of the batch's graphs
  auto request = [&](int b0, int g4, int q0, double2 (&m)[PF][GU][H]) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (f > 0 && q0 + f >= d.n_src) break;
      const size_t so = (size_t)(d.n_src == 0 ? d.in_slot : d.src[q0 + f]) * X;
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int b = graph_of(b0, g4, u);
        const double* gin = d.in + (size_t)(b < d.B ? b : 0) * d.in_ld + so;
#pragma unroll
        for (int j = 0; j < H; ++j) m[f][u][j] = load2(gin, j);
      }
    }
  };
  // v <- v x sources q0, q0 + 1 in the reference's order, nan_to_num after each product (LBP.py:377-389)
  auto multiply = [&](int q0, const double2 (&m)[PF][GU][H], double2 (&v)[GU][H]) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (q0 + f >= d.n_src) break;
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int j = 0; j < H; ++j) {
          double px = m[f][u][j].x * v[u][j].x, py = m[f][u][j].y * v[u][j].y;
          if (__builtin_expect(__any(!__builtin_isfinite(px) || !__builtin_isfinite(py)), 0)) { px = nan_to_num(px); py = nan_to_num(py); }
          v[u][j] = make_double2(px, py);
        }
    }
  };
  // the batch's messages, given their first PF sources in m (requested earlier): the remaining sources, the renormalisation,
  // the store of the variable->factor message when something later reads it, the LDS image
  auto finish_prologue = [&](int b0, int g4, double2 (&m)[PF][GU][H]) {
    double2 v[GU][H];
    if (d.n_src == 0) {
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int j = 0; j < H; ++j) v[u][j] = m[0][u][j];
    } else {
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int j = 0; j < H; ++j) v[u][j] = uniform2(j);
      multiply(0, m, v);
      for (int q = PF; q < d.n_src; q += PF) {
        request(b0, g4, q, m);
        multiply(q, m, v);
      }
      if (d.normalize) {
        double tot[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          double part = 0.0;
#pragma unroll
          for (int j = 0; j < H; ++j) part += v[u][j].x + v[u][j].y;
          tot[u] = wave_sum64(part);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
          for (int j = 0; j < H; ++j)
            v[u][j] = tot[u] > 0.0 ? make_double2(v[u][j].x / tot[u], v[u][j].y / tot[u]) : uniform2(j);
      }
      if (d.vf_slot >= 0) {
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (graph_of(b0, g4, u) < d.B) {
            double* o = d.out + (size_t)graph_of(b0, g4, u) * d.out_ld + (size_t)d.vf_slot * X;
#pragma unroll
            for (int j = 0; j < H; ++j) store2(o, j, v[u][j]);
          }
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      mt_t* row = Mt + (wave * GPW + g4 + u) * XP;
      const bool live = graph_of(b0, g4, u) < d.B;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        row[2 * lane + 128 * j] = live ? (mt_t)v[u][j].x : (mt_t)0;
        row[2 * lane + 128 * j + 1] = live ? (mt_t)v[u][j].y : (mt_t)0;
      }
    }
  };

  const avec* Af = reinterpret_cast<const avec*>(d.frag);
  const int gcol = lane & 15, krow = lane >> 4;
  constexpr int DIST = DEPTH / 2;
#ifdef MLBP_CONTRACT_NOLOOP          // diagnostic build (tools/contract_probe.py): prologue + epilogue only
  constexpr int KB_RUN = 4;
#else
  constexpr int KB_RUN = KB;
#endif
  static_assert(KB % 4 == 0 && (DEPTH == 2 || DEPTH == 4), "");

  // ---- main loop of one pass: acc = table x the LDS image ----
  auto main_loop = [&](double4_t (&acc)[RTW][NCT], auto&& at_step) {      // at_step(kb): called once per DEPTH steps, in front of them
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[r][c] = double4_t{0.0, 0.0, 0.0, 0.0};
    // DEPTH register sets of A fragments in rotation, each requested DEPTH / 2 whole steps before it is used (the loop is
    // unrolled by DEPTH so that no set is ever copied)
    avec a[DEPTH][RTW];
    auto load_a = [&](avec (&dst)[RTW], int kb) {
#pragma unroll
      for (int r = 0; r < RTW; ++r) dst[r] = Af[((size_t)(wave + NW * r) * KB + kb) * 64 + lane];
    };
#pragma unroll
    for (int s0 = 0; s0 < DIST; ++s0) load_a(a[s0], s0);
    if constexpr (sizeof(TT) == 8) {
      // The 16 x 16 x 4 product as FOUR v_mfma_f64_4x4x4 (4 blocks of 4 x 4 x 4): measured on this GPU the 16x16x4 form sustains
      // 47-49 TFLOP/s, the 4x4x4 form 70-71 (tools/mfma_peak.hip, profiles/r02k_mfma_peak.txt).  Operand lanes (probed,
      // profiles/r02k_mfma_f64_4x4x4_layout.txt): A_blk[i][k] at lane i + 4 blk + 16 k -- the 16x16x4 A fragment as it is,
      // block = rows 4 blk .. 4 blk + 3; B_blk[k][j] at lane j + 4 blk + 16 k -- four graphs 4 q + j per instruction, the
      // same in every block (LDS broadcast); D_blk[i][j] at lane j + 4 blk + 16 i, i.e. accumulator q of a lane holds
      // (row 4 ((lane >> 2) & 3) + (lane >> 4), graph 4 q + (lane & 3)).
      auto step = [&](const avec (&af)[RTW], int kb) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#ifdef MLBP_CONTRACT_16X16          // A/B build: the single-instruction form
          double bf[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) bf[c] = Mt[(16 * c + gcol) * XP + 8 * kb + 4 * e + krow];
#pragma unroll
          for (int r = 0; r < RTW; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
              acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(e ? af[r].y : af[r].x, bf[c], acc[r][c], 0, 0, 0);
#else
          double bf[NCT][4];
#pragma unroll
          for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[c][q] = Mt[(16 * c + 4 * q + (lane & 3)) * XP + 8 * kb + 4 * e + krow];
#pragma unroll
          for (int r = 0; r < RTW; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[r][c][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(e ? af[r].y : af[r].x, bf[c][q], acc[r][c][q], 0, 0, 0);
#endif
        }
      };
#pragma unroll 1
      for (int kb = 0; kb < KB_RUN; kb += DEPTH) {
        at_step(kb);
#pragma unroll
        for (int s0 = 0; s0 < DEPTH; ++s0) {
          if (kb + s0 + DIST < KB) load_a(a[(s0 + DIST) % DEPTH], kb + s0 + DIST);
          step(a[s0], kb + s0);
        }
      }
    } else {
      // float32 products, summed in float32 over 64 states (4 fragment blocks) at a time, then added in float64
      float4_t part[RTW][NCT];
      auto step = [&](const avec (&af4)[RTW], int kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float bf[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) bf[c] = Mt[(16 * c + gcol) * XP + 16 * kb + 4 * e + krow];
#pragma unroll
          for (int r = 0; r < RTW; ++r) {
            const float af = e == 0 ? af4[r].x : (e == 1 ? af4[r].y : (e == 2 ? af4[r].z : af4[r].w));
#pragma unroll
            for (int c = 0; c < NCT; ++c) part[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[c], part[r][c], 0, 0, 0);
          }
        }
      };
#pragma unroll 1
      for (int kb = 0; kb < KB_RUN; kb += 4) {
        at_step(kb);
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int c = 0; c < NCT; ++c) part[r][c] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s0 = 0; s0 < 4; ++s0) {
          if (kb + s0 + DIST < KB) load_a(a[(s0 + DIST) % DEPTH], kb + s0 + DIST);
          step(a[s0 % DEPTH], kb + s0);
        }
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r][c][i] += (double)part[r][c][i];
      }
    }
  };

  // ---- epilogue of one pass: accumulators -> LDS transposed ([graph][state], float64), renormalise, store whole rows
  //      (every wave has read its last message fragment when this starts: the image becomes the output).
  // result element (row, col) of a 16 x 16 tile: float64 (four 4x4x4 MFMAs): see the main loop; float32 MFMA:
  // col = lane & 15 (graph), row = 4 (lane >> 4) + i
  auto epilogue = [&](int b0, const double4_t (&acc)[RTW][NCT]) {
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifndef MLBP_CONTRACT_16X16
          if (sizeof(TT) == 8) {
            Ot[(16 * c + 4 * i + (lane & 3)) * XP + 16 * (wave + NW * r) + 4 * ((lane >> 2) & 3) + krow] = acc[r][c][i];
            continue;
          }
#endif
          const int row = sizeof(TT) == 8 ? krow + 4 * i : 4 * krow + i;
          Ot[(16 * c + gcol) * XP + 16 * (wave + NW * r) + row] = acc[r][c][i];
        }
    __syncthreads();
    for (int g4 = 0; g4 < GPW; g4 += GU) {
      double2 v[GU][H];
      double tot[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const double* row = Ot + (wave * GPW + g4 + u) * XP;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          v[u][j] = make_double2(row[2 * lane + 128 * j], row[2 * lane + 128 * j + 1]);
          part += v[u][j].x + v[u][j].y;
        }
        tot[u] = d.normalize ? wave_sum64(part) : 1.0;
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int b = graph_of(b0, g4, u);
        if (b < d.B) {
          double* o = d.out + (size_t)b * d.out_ld + (size_t)d.dst_slot * X;
#pragma unroll
          for (int j = 0; j < H; ++j) {
            double2 val = v[u][j];
            if (d.normalize) val = tot[u] > 0.0 ? make_double2(val.x / tot[u], val.y / tot[u]) : make_double2(uniform, uniform);
            store2(o, j, val);
          }
        }
      }
    }
  };

#ifdef MLBP_CONTRACT_STAGGER         // experiment: the second half of the grid starts late (units of 64 * 127 clocks)
  if (blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < MLBP_CONTRACT_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  double4_t acc[RTW][NCT];
  const int b_first = blockIdx.x * (NT_G * HALVES);
  {
    double2 m[PF][GU][H];
    for (int g4 = 0; g4 < GPW; g4 += GU) {
      request(b_first, g4, 0, m);
      finish_prologue(b_first, g4, m);
    }
  }
  __syncthreads();
  if constexpr (HALVES == 1) {
    main_loop(acc, [](int) {});
    __syncthreads();
    epilogue(b_first, acc);
  } else {
    const int b_second = b_first + NT_G;
    const bool second = b_second < d.B;                          // (uniform over the workgroup)
    double2 m[PF][GU][H];
#if defined(MLBP_CONTRACT_EXPERIMENT) && MLBP_CONTRACT_EXPERIMENT == 4
    // the request goes out INSIDE the first pass's loop, at a step that differs from wave to wave: the table fragments behind it
    // wait for it (loads return in order), but one wave at a time, with the SIMD's other waves on the matrix pipe
    const int at = (wave * 4 * (KB / 64 > 0 ? KB / 64 : 1)) % KB;
    main_loop(acc, [&](int kb) { if (second && kb == (at / DEPTH) * DEPTH) request(b_second, 0, 0, m); });
#else
    if (second) request(b_second, 0, 0, m);                      // in flight under the first pass's main loop
    main_loop(acc, [](int) {});
#endif
    __syncthreads();
    epilogue(b_first, acc);                                      // its stores drain under the second pass
    if (!second) return;
    __syncthreads();                                             // the image is free again
    finish_prologue(b_second, 0, m);
    __syncthreads();
    main_loop(acc, [](int) {});
    __syncthreads();
    epilogue(b_second, acc);
  }
}


template <typename TT>
size_t contract_lds_bytes(int X, int nct) {
  // the float32 image is reused as the float64 output image
  (void)sizeof(TT);
  return (size_t)16 * nct * (X + 2) * sizeof(double);
}

template <typename TT, int RT, bool PADDED>
int launch_contract_rt(const ContractDev& d, int nct, hipStream_t st) {
  const int XA = 64 * RT;
  const int ntg = 16 * nct;
  size_t lds_bytes = contract_lds_bytes<TT>(XA, nct);
  // 32 graphs per workgroup (float64, small tables, big batches): 4 waves, deep fragment prefetch; else 16 graphs per
  // workgroup and 8 waves (two workgroups = 4 waves per SIMD hide each other's stalls; measured 3-7 % over 4 waves)
  void (*k)(ContractDev) = nullptr;
  if constexpr (PADDED) k = contract_kernel<TT, RT, 1, 2, 8, true, 1>;
  else k = nct == 2 ? contract_kernel<TT, RT, 2, 4, 4, false, 1>
                    : (sizeof(TT) == 8 ? contract_kernel<TT, RT, 1, 2, 8, false, 1> : contract_kernel<TT, RT, 1, 4, 8, false, 1>);
  int threads = nct == 2 ? WG : 512, graphs_per_wg = ntg;
#ifdef MLBP_CONTRACT_EXPERIMENT      // tools/contract_probe.py -DMLBP_CONTRACT_EXPERIMENT=n: the round-3 forms that did not pay
  if constexpr (!PADDED && (4 * RT) % 16 == 0) {
    if (nct == 1 && d.B > 16) {
      threads = 1024;
      if (MLBP_CONTRACT_EXPERIMENT == 1 || MLBP_CONTRACT_EXPERIMENT == 4) {   // two passes per workgroup, the second's sources requested under the first's loop (4: inside it)
        k = contract_kernel<TT, RT, 1, 4, 16, false, 2>; graphs_per_wg = 32;
      } else if (MLBP_CONTRACT_EXPERIMENT == 2) {   // 16 graphs per 16-wave workgroup at 64 registers: 8 waves per SIMD
        k = contract_kernel<TT, RT, 1, 2, 16, false, 1>; graphs_per_wg = 16;
      } else {                                      // 32 graphs per 16-wave workgroup: half the table traffic per graph
        k = contract_kernel<TT, RT, 2, 2, 16, false, 1>; graphs_per_wg = 32; lds_bytes = contract_lds_bytes<TT>(XA, 2);
      }
    }
  }
#endif
  static std::mutex mu;
  static std::vector<const void*> granted;
  {
    std::lock_guard<std::mutex> lock(mu);
    bool have = false;
    for (const void* g : granted) have |= g == (const void*)k;
    if (!have) {
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess)
        return fail(MLBP_EHIP, "contract_kernel: cannot raise the dynamic LDS limit");
      granted.push_back((const void*)k);
    }
  }
  hipLaunchKernelGGL(k, dim3((d.B + graphs_per_wg - 1) / graphs_per_wg), dim3(threads), lds_bytes, st, d);
  return MLBP_OK;
}

// the size the contraction of X states runs at: the next multiple of 128
int padded_states(int X) { return (X + 127) / 128 * 128; }

// graphs per workgroup: 32 when the batch still fills the chip twice over and the accumulators fit, else 16
template <typename TT>
int launch_contract(ContractDev d, int X, hipStream_t st) {
  const bool f32 = sizeof(TT) == 4;
  const int XA = padded_states(X);
  d.X = X;
  if (XA != X) {
    if constexpr (sizeof(TT) == 8) {
      switch (XA) {
        case 128: return launch_contract_rt<TT, 2, true>(d, 1, st);
        case 256: return launch_contract_rt<TT, 4, true>(d, 1, st);
        case 384: return launch_contract_rt<TT, 6, true>(d, 1, st);
        case 512: return launch_contract_rt<TT, 8, true>(d, 1, st);
      }
    }
    return fail(MLBP_EUNSUPPORTED, "shared-table contraction: X = %d with float32 tables (256 or 512)", X);
  }
  int nct = (!f32 && X <= 256 && d.B >= 32 * 512) ? 2 : 1;     // small tables: fewer passes over the table per graph
  switch (X) {
    case 128: return launch_contract_rt<TT, 2, false>(d, nct, st);
    case 256: return launch_contract_rt<TT, 4, false>(d, nct, st);
    case 384: return launch_contract_rt<TT, 6, false>(d, nct, st);
    case 512: return launch_contract_rt<TT, 8, false>(d, nct, st);
  }
  return fail(MLBP_EUNSUPPORTED, "shared-table contraction: X = %d (65 .. 512)", X);
}

// ---- the small kernels around the contraction --------------------------------------------------------------------
// VariableNode.update_message_to (LBP.py:377-389) when it does NOT feed the next contraction: uniform times the listed
// incoming messages, nan_to_num after each product, renormalised when asked.
__global__ __launch_bounds__(WG) void variable_update_kernel(double* msgs, int n_msgs, int X, const int32_t* srcs, int b, int c,
                                                            int normalize) {
  __shared__ double scratch[4];
  extern __shared__ double raw[];
  double* gm = msgs + (size_t)blockIdx.x * n_msgs * X;
  const double uniform = 1.0 / (double)X;
  double part = 0.0;
  for (int j = threadIdx.x; j < X; j += WG) {
    double acc = uniform;
    for (int q = 0; q < b; ++q) acc = nan_to_num(gm[(size_t)srcs[q] * X + j] * acc);
    raw[j] = acc;
    part += acc;
  }
  const double total = normalize ? block_sum(part, scratch) : 0.0;
  for (int j = threadIdx.x; j < X; j += WG)
    gm[(size_t)c * X + j] = !normalize ? raw[j] : (total > 0.0 ? raw[j] / total : uniform);
}

// Unary factor -> variable (LBP.py:494-498): the factor's column, renormalised when asked.
__global__ __launch_bounds__(WG) void unary_update_kernel(double* msgs, int n_msgs, int X, const double* unary_tables,
                                                         const int32_t* unary_tab, int U, int n_unary_tables, int u, int c,
                                                         int normalize, int32_t* status) {
  __shared__ double scratch[4];
  const int ti = unary_tab[(size_t)blockIdx.x * U + u];
  if ((unsigned)ti >= (unsigned)n_unary_tables) {
    if (threadIdx.x == 0) atomicExch(status, 1);
    return;
  }
  const double* t = unary_tables + (size_t)ti * X;
  double* m = msgs + ((size_t)blockIdx.x * n_msgs + c) * X;
  double part = 0.0;
  for (int j = threadIdx.x; j < X; j += WG) part += t[j];
  const double total = normalize ? block_sum(part, scratch) : 0.0;
  const double uniform = 1.0 / (double)X;
  for (int j = threadIdx.x; j < X; j += WG) m[j] = !normalize ? t[j] : (total > 0.0 ? t[j] / total : uniform);
}

__global__ void fill_uniform_kernel(double* p, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// The caller's statement "every graph has the pair_tab row given on the host" is checked on the device; a false
// statement raises the program's status word to 2 (the results of this launch are then meaningless).
struct HostRow { int32_t v[16]; };
__global__ void check_shared_claim_kernel(const int32_t* pair_tab, int B, int P, int n_pair_tables, HostRow row, int32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P) return;
  const int want = row.v[i % P];
  if (pair_tab[i] != want || (unsigned)want >= (unsigned)n_pair_tables) atomicExch(status, 2);
}

// ---- pairwise part of the gradient (LBP.py:528-619, 301-320) ----
// S[b] = sum_i c[b][i] * Y[b][i]
__global__ __launch_bounds__(WG) void row_dot_kernel(const double* msgs, int n_msgs, int X, int c_slot, const double* Y, double* S) {
  __shared__ double scratch[4];
  const double* c = msgs + ((size_t)blockIdx.x * n_msgs + c_slot) * X;
  const double* y = Y + (size_t)blockIdx.x * X;
  double part = 0.0;
  for (int j = threadIdx.x; j < X; j += WG) part += c[j] * y[j];
  const double tot = block_sum(part, scratch);
  if (threadIdx.x == 0) S[blockIdx.x] = tot;
}
// grad_en_en[b][k] += phi[l0][l1][k] - S_k[b] / Z[b]   (au.normalize: zero-sum -> expectation 0)
__global__ void pair_gradient_combine_kernel(const double* S /*[4][B]*/, int B, int X, const int32_t* pair_label, int P, int p,
                                             const double* phi, double* grad_en_en, int32_t* status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int l0 = pair_label[((size_t)b * P + p) * 2], l1 = pair_label[((size_t)b * P + p) * 2 + 1];
  if ((unsigned)l0 >= (unsigned)X || (unsigned)l1 >= (unsigned)X) { atomicExch(status, 1); return; }
  const double Z = S[3 * (size_t)B + b];
  for (int k = 0; k < 3; ++k)
    grad_en_en[(size_t)b * 3 + k] += phi[((size_t)l0 * X + l1) * 3 + k] - (Z > 0.0 ? S[k * (size_t)B + b] / Z : 0.0);
}

// One device-wide scratch buffer per purpose, grown on demand (launches on different streams must not overlap; a
// stream-capturing caller runs one eager step first so that nothing is allocated under capture).
std::mutex g_scratch_mutex;
int ensure_scratch(void** p, size_t* cap, size_t need) {
  if (need <= *cap) return MLBP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  if (hipMalloc(p, need) != hipSuccess) return fail(MLBP_EHIP, "shared-table contraction: scratch allocation of %zu bytes failed", need);
  *cap = need;
  return MLBP_OK;
}

template <typename TT>
void launch_table_frag(const TT* T, const double* plane, int X, int transpose, TT* frag, hipStream_t st) {
  constexpr int E = sizeof(TT) == 8 ? 2 : 4;
  const int XA = padded_states(X);
  hipLaunchKernelGGL((table_frag_kernel<TT, E>), dim3((XA * XA + 255) / 256), dim3(256), 0, st, T, plane, X, XA, transpose, frag);
}

bool contract_supports(int X) { return X > 64 && X <= 512; }

}  // namespace

int gemm_path_ready() { return MLBP_OK; }          // hand-written: nothing to load
bool gemm_path_supports(int X) { return contract_supports(X); }

int launch_gemm_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream) {
  if (!contract_supports(a->X)) return fail(MLBP_EUNSUPPORTED, "shared-table gradient: X = %d", a->X);
  hipStream_t st = (hipStream_t)stream;
  const int B = a->B, X = a->X;
  // scratch: W fragments [X][X], Y [B][X], S [4][B]
  static void* scratch = nullptr;
  static size_t cap = 0;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  const size_t XA = (size_t)padded_states(X);
  if (int e = ensure_scratch(&scratch, &cap, (XA * XA + (size_t)B * X + 4 * (size_t)B) * sizeof(double))) return e;
  double* W = (double*)scratch; double* Y = W + XA * XA; double* S = Y + (size_t)B * X;
  // slots come from DEVICE arrays in the ABI (pair_c_slot / pair_r_slot / pair_phi): fetch the few ints once
  int32_t h_c[16], h_r[16], h_phi[16];
  if (a->P > 16) return fail(MLBP_EUNSUPPORTED, "shared-table gradient: at most 16 pairwise factors (got %d)", a->P);
  if (a->pair_slots_host) {                       // the caller's host copy: nothing to read back, the call only enqueues
    for (int p = 0; p < a->P; ++p) { h_c[p] = a->pair_slots_host[p]; h_r[p] = a->pair_slots_host[a->P + p]; h_phi[p] = a->pair_slots_host[2 * a->P + p]; }
  } else if (hipMemcpyAsync(h_c, a->pair_c_slot, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipMemcpyAsync(h_r, a->pair_r_slot, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipMemcpyAsync(h_phi, a->pair_phi, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipStreamSynchronize(st) != hipSuccess) {
    return fail(MLBP_EHIP, "shared-table gradient: reading the slot tables failed");
  }
  for (int p = 0; p < a->P; ++p) {
    if ((unsigned)h_c[p] >= (unsigned)a->n_msgs || (unsigned)h_r[p] >= (unsigned)a->n_msgs ||
        (unsigned)a->pair_tab_host[p] >= (unsigned)a->n_pair_tables)
      return fail(MLBP_EINVAL, "shared-table gradient: slot or table index of factor %d out of range", p);
    const double* T = a->pair_tables + (size_t)a->pair_tab_host[p] * X * X;
    const double* planes = h_phi[p] ? a->phi_en_en_w1_p : a->phi_en_en_p;
    for (int k = 0; k < 4; ++k) {
      // Y[b][:] = (T (.) phi_k) . r_b   (k = 3: T . r_b, the normaliser), r = msgs[:, r_slot, :]
      launch_table_frag<double>(T, k < 3 ? planes + (size_t)k * X * X : nullptr, X, 0, W, st);
      ContractDev d = {};
      d.frag = W; d.in = a->msgs; d.in_ld = (size_t)a->n_msgs * X; d.out = Y; d.out_ld = X;
      d.n_src = 0; d.in_slot = h_r[p]; d.vf_slot = -1; d.dst_slot = 0; d.B = B; d.normalize = 0;
      if (int e = launch_contract<double>(d, X, st)) return e;
      hipLaunchKernelGGL(row_dot_kernel, dim3(B), dim3(WG), 0, st, a->msgs, a->n_msgs, X, h_c[p], Y, S + (size_t)k * B);
    }
    hipLaunchKernelGGL(pair_gradient_combine_kernel, dim3((B + 255) / 256), dim3(256), 0, st, S, B, X, a->pair_label, a->P, p,
                       h_phi[p] ? a->phi_en_en_w1 : a->phi_en_en, a->grad_en_en, status);
  }
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "shared-table gradient: a launch failed");
  return MLBP_OK;
}

int launch_gemm_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream) {
  if (!a->pair_tab_host) return fail(MLBP_EINVAL, "shared-table contraction path: pair_tab_host (host int32 [P]) is required");
  if (prog->P > 16) return fail(MLBP_EUNSUPPORTED, "shared-table contraction path: at most 16 pairwise factors (got %d)", prog->P);
  if (!contract_supports(a->X)) return fail(MLBP_EUNSUPPORTED, "shared-table contraction path: X = %d", a->X);
  for (int p = 0; p < prog->P; ++p)
    if ((unsigned)a->pair_tab_host[p] >= (unsigned)a->n_pair_tables)
      return fail(MLBP_EINVAL, "pair_tab_host[%d] = %d out of [0,%d)", p, a->pair_tab_host[p], a->n_pair_tables);
  hipStream_t st = (hipStream_t)stream;
  const int B = a->B, X = a->X, n_msgs = prog->n_msgs, norm = a->normalize_messages ? 1 : 0;
  const bool f32 = (a->flags & MLBP_SWEEP_PAIR_TABLES_F32) != 0;
  const size_t elem = f32 ? sizeof(float) : sizeof(double);
  const FusedProgram& fp = prog->fused;
  // fragment-ordered copies of the distinct tables, both orientations: [P][2][X*X]
  static void* frag = nullptr;
  static size_t frag_cap = 0;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  const size_t XA = (size_t)padded_states(X);
  if ((a->flags & MLBP_SWEEP_PAIR_TABLES_F32) && XA != (size_t)X)
    return fail(MLBP_EUNSUPPORTED, "shared-table contraction path: float32 tables need X = 256 or 512");
  if (int e = ensure_scratch(&frag, &frag_cap, (size_t)prog->P * 2 * XA * XA * elem)) return e;
  {
    HostRow row = {};
    for (int p = 0; p < prog->P; ++p) row.v[p] = a->pair_tab_host[p];
    hipLaunchKernelGGL(check_shared_claim_kernel, dim3((B * prog->P + 255) / 256), dim3(256), 0, st, a->pair_tab, B, prog->P,
                       a->n_pair_tables, row, prog->d_status);
  }
  for (int p = 0; p < prog->P; ++p)
    for (int tr = 0; tr < 2; ++tr) {
      const size_t off = ((size_t)p * 2 + tr) * XA * XA;
      if (f32) launch_table_frag<float>(a->pair_tables_f32 + (size_t)a->pair_tab_host[p] * X * X, nullptr, X, tr, (float*)frag + off, st);
      else launch_table_frag<double>(a->pair_tables + (size_t)a->pair_tab_host[p] * X * X, nullptr, X, tr, (double*)frag + off, st);
    }
  if (a->init_messages)
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(1024), dim3(256), 0, st, a->msgs, (size_t)B * n_msgs * X, 1.0 / (double)X);
  // unary messages the program could hoist are constants (LBP.py:494-498): once per call
  for (size_t h = 0; h + 1 < fp.hoist.size(); h += 2)
    hipLaunchKernelGGL(unary_update_kernel, dim3(B), dim3(WG), 0, st, a->msgs, n_msgs, X, a->unary_tables, a->unary_tab, prog->U,
                       a->n_unary_tables, fp.hoist[h], fp.hoist[h + 1], norm, prog->d_status);
  const size_t ld = (size_t)n_msgs * X;
  const int n_fops = (int)fp.fops.size() / 8;
  // a fused variable->factor message goes to memory only when something reads the slot before its next write, or when
  // it is the slot's final value (the messages are an output of the call): in a 10-sweep call most are neither
  std::vector<char> store_vf(n_fops, 1);
  for (int i = 0; i < n_fops; ++i) {
    const int32_t* w = &fp.fops[8 * (size_t)i];
    const int kd = w[0] & 0xFF;
    if (kd != FOP_VAR_PAIR_TM && kd != FOP_VAR_PAIR_MT) continue;
    const int c = w[3];
    for (int j = i + 1; j < n_fops; ++j) {
      const int32_t* v = &fp.fops[8 * (size_t)j];
      const int kj = v[0] & 0xFF;
      bool reads = false, writes = false;
      if (kj == FOP_PAIR_TM || kj == FOP_PAIR_MT) { reads = v[2] == c; writes = v[3] == c; }
      else if (kj == FOP_UNARY) { writes = v[3] == c; }
      else {
        for (int q = 0; q < v[7]; ++q) reads |= fp.psrcs[v[6] + q] == c;
        writes = v[3] == c || (kj != FOP_VAR && v[5] == c);
      }
      if (reads) break;
      if (writes) { store_vf[i] = 0; break; }
    }
  }
  for (int i = 0; i < n_fops; ++i) {
    const int32_t* w = &fp.fops[8 * (size_t)i];
    const int kind = w[0] & 0xFF;
    if (kind == FOP_UNARY) {
      hipLaunchKernelGGL(unary_update_kernel, dim3(B), dim3(WG), 0, st, a->msgs, n_msgs, X, a->unary_tables, a->unary_tab, prog->U,
                         a->n_unary_tables, w[1], w[3], norm, prog->d_status);
      continue;
    }
    if (kind == FOP_VAR) {          // exact source list: psrcs[w[6] .. w[6] + w[7])
      hipLaunchKernelGGL(variable_update_kernel, dim3(B), dim3(WG), (size_t)X * sizeof(double), st, a->msgs, n_msgs, X,
                         (prog->d_fops + 8 * (size_t)prog->n_fops) + w[6], w[7], w[3], norm);
      continue;
    }
    ContractDev d = {};
    d.in = a->msgs; d.out = a->msgs; d.in_ld = ld; d.out_ld = ld; d.B = B; d.normalize = norm;
    int pslot, tm;
    if (kind == FOP_PAIR_TM || kind == FOP_PAIR_MT) {
      pslot = w[1]; tm = kind == FOP_PAIR_TM;
      d.n_src = 0; d.in_slot = w[2]; d.vf_slot = -1; d.dst_slot = w[3];
    } else {
      pslot = w[4]; tm = kind == FOP_VAR_PAIR_TM;
      d.n_src = w[7]; d.in_slot = 0; d.vf_slot = w[3]; d.dst_slot = w[5];
      if (d.n_src == 0 || d.n_src > 8) {      // no other factor (the message is the uniform vector), or a long list: its own launch
        hipLaunchKernelGGL(variable_update_kernel, dim3(B), dim3(WG), (size_t)X * sizeof(double), st, a->msgs, n_msgs, X,
                           (prog->d_fops + 8 * (size_t)prog->n_fops) + w[6], w[7], w[3], norm);
        d.n_src = 0; d.in_slot = w[3]; d.vf_slot = -1;
      } else {
        for (int q = 0; q < d.n_src; ++q) d.src[q] = fp.psrcs[w[6] + q];
        if (!store_vf[i]) d.vf_slot = -1;
      }
    }
    // out = T . m contracts over the table's columns: A = T; out = m^T . T over its rows: A = T^T
    const size_t off = ((size_t)pslot * 2 + (tm ? 0 : 1)) * XA * XA;
    d.frag = f32 ? (const void*)((const float*)frag + off) : (const void*)((const double*)frag + off);
    if (int e = f32 ? launch_contract<float>(d, X, st) : launch_contract<double>(d, X, st)) return e;
  }
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "shared-table contraction path: a launch failed");
  return MLBP_OK;
}

}  // namespace mlbp
