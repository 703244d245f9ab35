// Shared pairwise tables at large state spaces (64 < X <= 4096): every factor->variable update of the WHOLE batch is
// one dense contraction on the matrix cores, written here (no library call).
//
// With one table behind factor p for every graph (the reference's layout, LBP.py:456-467; X = |V_en| = len(en_domain)
// there, train_mp.py:591-594) the update of all B graphs is      OUT[X x B] = T[X x X] . M[X x B]      (or T^T . M)
// -- SURVEY.md section 8(d)'s "shared-table (GEMM/MFMA) variant" of BASELINE config 5.  At X = 64 the whole sweep
// fits one workgroup per 16 graphs (mlbp_shared.hip); at X = 512 a table is 2 MiB, so the sweep runs update by update
// over the batch, ONE launch per update.  A workgroup takes 16 graphs (32 for small tables and big batches):
//
//   prologue   the input message of its graphs is formed and parked in LDS: either a stored slot, or -- fused -- the
//              variable->factor product of VariableNode.update_message_to (LBP.py:377-389: uniform x the listed incoming
//              messages, nan_to_num after each product, renormalised), which is also stored when something reads it;
//   main loop  four v_mfma_f64_4x4x4_f64 per 16 x 16 x 4 product (this GPU sustains 70 TFLOP/s on that form, 48 on the
//              single v_mfma_f64_16x16x4_f64; tools/mfma_peak.hip): wave w owns the row tiles w, w + NW, ...; its A
//              fragments (table rows) come straight from L2 as one coalesced 16-byte load per lane and two k-steps, out of a
//              copy of the table laid out in fragment order once per call (table_frag_kernel), with a register double
//              buffer; the B fragments (messages) are 8-byte LDS reads out of a [graph][state + 2] image (conflict-free);
//   epilogue   the accumulators go back through the same LDS image transposed, so that Message.renormalize
//              (LBP.py:649-657) sees whole columns and the result leaves in full rows.
//
// contract_kernel        X <= 1024 (the contraction runs at XA = X rounded up to 128, operands zero beyond X): the image of
//                        16 graphs is 16 (XA + 2) doubles of LDS, at most 131 KiB.
// contract_chunked_kernel X > 1024: the image no longer fits, so the table is walked in RP x RP blocks of CH x CH states
//                        (CH <= 1024): for every block row the accumulators collect the block columns one after the other,
//                        the matching CH states of the input messages re-loaded into the image each time (from L2: the
//                        formed message was stored first); a block row's results leave unnormalised with their partial
//                        sums kept per graph, and the workgroup divides its own rows once the last block row is done.
//
// float32 tables (MLBP_SWEEP_PAIR_TABLES_F32, the "batched f32 MFMA message contraction" of config 5; X = 256 / 512): the
// same structure on v_mfma_f32_16x16x4_f32 -- table and message fragments in float32, products summed in float32 over 64
// states at a time, the 64-state partial sums added into float64 accumulators (tolerance study: DESIGN.md 4.2b).
//
// The pairwise part of the gradient (LBP.py:528-619) is the same contraction with T (.) phi_k as the table and a different
// epilogue: instead of storing Y = (T (.) phi_k) . r the workgroup takes the dot product with c on the spot (`dot` mode);
// the four tables of a factor (k = 0..2 and the normaliser) are one launch (blockIdx.y).
//
// Same updates in the same order as every other path (the fused program of build_fused_program); only the summation
// order inside a contraction differs.  The round-3 overlap experiments that did not pay live in
// tools/experiments/contract_kernel_r03_variants.hip.
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
// XA >= X is the padded size the contraction runs at: rows and columns past X are zero, so that any vocabulary size
// (X = len(en_domain), train_mp.py:591-594) takes the matrix-core path.  blockIdx.y = k selects the k-th plane (planes
// [n_planes][X][X]; k == n_planes or no planes: the table itself) and the k-th output set (XA * XA elements apart).
template <typename TT, int E>
__global__ void table_frag_kernel(const TT* T, const double* planes, int n_planes, int X, int XA, int transpose, TT* frag) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one output element
  if (idx >= (size_t)XA * XA) return;
  const int kset = blockIdx.y;
  const double* plane = (planes && kset < n_planes) ? planes + (size_t)kset * X * X : nullptr;
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
  frag[(size_t)kset * XA * XA + idx] = (TT)v;
}

struct ContractDev {
  const void* frag;          // the table in fragment order (table_frag_kernel); blockIdx.y selects set y, frag_set elements apart
  size_t frag_set;
  const double* in;          // source messages: (b, slot, x) at in[b * in_ld + slot * X + x]
  double* out;               // results:         (b, slot, x) at out[b * out_ld + slot * X + x]
  int32_t src[8];            // source slots of the fused variable product, reference order (n_src <= 8)
  size_t in_ld, out_ld;
  int32_t n_src;             // 0: the input is slot in_slot as it stands
  int32_t in_slot, vf_slot, dst_slot;      // vf_slot: where the variable->factor message itself is stored, or -1
  int32_t B, normalize;
  int32_t X;                 // states (<= the size the kernel instance runs at; smaller: zero-padded operands, PADDED instances)
  int32_t passes;            // chunked kernel: the table is passes x passes blocks of 64 RT states
  double* xbuf;              // chunked kernel: [B][X] the formed input message of a fused update whose vf_slot is -1
  int32_t dot_slot;          // >= 0: `dot` mode -- nothing is stored; dots[y * B + b] = sum_x in[b][dot_slot][x] * result[b][x]
  int32_t pad_;
  double* dots;
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

// acc += (row tiles rt0 + wave + NW r, r < RTW, of the fragment-ordered table; fragment blocks kb0 .. kb0 + kb_run of the
// kbt a row tile has) x (the LDS image Mt[16 NCT graphs][XP], whose state 0 is the first state of block kb0).
// float64: the 16 x 16 x 4 product as FOUR v_mfma_f64_4x4x4 (4 blocks of 4 x 4 x 4): measured on this GPU the 16x16x4 form
// sustains 47-49 TFLOP/s, the 4x4x4 form 70-71 (tools/mfma_peak.hip, profiles/r02k_mfma_peak.txt).  Operand lanes (probed,
// profiles/r02k_mfma_f64_4x4x4_layout.txt): A_blk[i][k] at lane i + 4 blk + 16 k -- the 16x16x4 A fragment as it is, block =
// rows 4 blk .. 4 blk + 3; B_blk[k][j] at lane j + 4 blk + 16 k -- four graphs 4 q + j per instruction, the same in every
// block (LDS broadcast); D_blk[i][j] at lane j + 4 blk + 16 i, i.e. accumulator q of a lane holds (row 4 ((lane >> 2) & 3) +
// (lane >> 4), graph 4 q + (lane & 3)).
// float32: products summed in float32 over 64 states (4 fragment blocks) at a time, then added in float64.
// DEPTH register sets of A fragments in rotation, each requested DEPTH / 2 whole steps before it is used (the loop is
// unrolled by DEPTH so that no set is ever copied).
template <typename TT, int RTW, int NCT, int DEPTH, int NW, int XP>
__device__ __forceinline__ void contract_tiles(double4_t (&acc)[RTW][NCT], const typename Frag<TT>::vec* Af, int rt0, int kb0, int kbt,
                                               int kb_run, const typename Frag<TT>::lds_t* Mt, int lane, int wave) {
  typedef typename Frag<TT>::vec avec;
  constexpr int DIST = DEPTH / 2;
  static_assert(DEPTH == 2 || DEPTH == 4, "");
  const int gcol = lane & 15, krow = lane >> 4;
  avec a[DEPTH][RTW];
  auto load_a = [&](avec (&dst)[RTW], int kb) {
#pragma unroll
    for (int r = 0; r < RTW; ++r) dst[r] = Af[((size_t)(rt0 + wave + NW * r) * kbt + kb0 + kb) * 64 + lane];
  };
#pragma unroll
  for (int s0 = 0; s0 < DIST; ++s0) load_a(a[s0], s0);
  if constexpr (sizeof(TT) == 8) {
    auto step = [&](const avec (&af)[RTW], int kb) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
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
      }
    };
#pragma unroll 1
    for (int kb = 0; kb < kb_run; kb += DEPTH) {
#pragma unroll
      for (int s0 = 0; s0 < DEPTH; ++s0) {
        if (kb + s0 + DIST < kb_run) load_a(a[(s0 + DIST) % DEPTH], kb + s0 + DIST);
        step(a[s0], kb + s0);
      }
    }
  } else {
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
    for (int kb = 0; kb < kb_run; kb += 4) {
#pragma unroll
      for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) part[r][c] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s0 = 0; s0 < 4; ++s0) {
        if (kb + s0 + DIST < kb_run) load_a(a[(s0 + DIST) % DEPTH], kb + s0 + DIST);
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
}

// accumulators -> the LDS image transposed ([graph][state of this block row], float64).  Result element (row, col) of a
// 16 x 16 tile: float64 (four 4x4x4 MFMAs): see contract_tiles; float32 MFMA: col = lane & 15 (graph), row = 4 (lane >> 4) + i
template <typename TT, int RTW, int NCT, int NW, int XP>
__device__ __forceinline__ void tiles_to_image(const double4_t (&acc)[RTW][NCT], double* Ot, int lane, int wave) {
  const int gcol = lane & 15, krow = lane >> 4;
#pragma unroll
  for (int r = 0; r < RTW; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (sizeof(TT) == 8) Ot[(16 * c + 4 * i + (lane & 3)) * XP + 16 * (wave + NW * r) + 4 * ((lane >> 2) & 3) + krow] = acc[r][c][i];
        else Ot[(16 * c + gcol) * XP + 16 * (wave + NW * r) + 4 * krow + i] = acc[r][c][i];
      }
}

// The contraction runs at XA = 64 * RT states; N_T = 16 * NCT graphs per workgroup.  PADDED: the messages have d.X <= XA
// states (rows of d.X doubles in memory, any parity: 8-byte accesses), the operands are zero beyond.
// Cache policy of the message traffic of the one-image kernel (MLBP_GEMM_NT: bit 0 source messages by non-temporal loads, bit 1
// results by non-temporal stores).  Every message of the batch passes through once per update (134 MB at X = 512, B = 8192) and
// the launch is one round of workgroups, all of them in the memory phase at the same time: X = 512 float32 tables 60.9 -> 58.5 us
// per update, float64 87.1 -> 85.9 (round 4; with the reciprocal-multiply normalisation: from 62.0 / 90.1).
#ifndef MLBP_GEMM_NT
#define MLBP_GEMM_NT 3
#endif
template <typename TT, int RT, int NCT, int DEPTH, int NW, bool PADDED>
__global__ __launch_bounds__(64 * NW, NW == 16 ? 4 : 1) void contract_kernel(ContractDev d) {
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

  // ---- prologue pieces: the input messages of the workgroup's graphs -> LDS.  A wave takes NT_G / NW consecutive graphs, up to
  //      four at a time, so that one round of memory latency covers four graphs; lane l holds states 2l, 2l+1 (+128 j) ----
  constexpr int GPW = NT_G / NW, GU = GPW < 4 ? GPW : 4;        // graphs per wave, and how many of them go through together
  constexpr int H = RT / 2;                                     // double2 pieces per lane
  constexpr int PF = RT > 8 ? 1 : 2;                            // source messages requested together
  static_assert(RT % 2 == 0, "X must be a multiple of 128 here");
  static_assert(KB % 4 == 0, "");
  // states 2 (lane + 64 j), + 1 of a row of X doubles; beyond X: zero
  typedef double nt_d2 __attribute__((ext_vector_type(2)));
  auto load2 = [&](const double* row, int j) {
    if (!PADDED) {
#if MLBP_GEMM_NT & 1
      const nt_d2 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(row) + lane + 64 * j);
      return make_double2(v_.x, v_.y);
#else
      return reinterpret_cast<const double2*>(row)[lane + 64 * j];
#endif
    }
    const int x0 = 2 * (lane + 64 * j);
    return make_double2(x0 < X ? row[x0] : 0.0, x0 + 1 < X ? row[x0 + 1] : 0.0);
  };
  auto uniform2 = [&](int j) {
    const int x0 = 2 * (lane + 64 * j);
    return (!PADDED) ? make_double2(uniform, uniform) : make_double2(x0 < X ? uniform : 0.0, x0 + 1 < X ? uniform : 0.0);
  };
  auto store2 = [&](double* row, int j, double2 val) {
    if (!PADDED) {
#if MLBP_GEMM_NT & 2
      nt_d2 x_; x_.x = val.x; x_.y = val.y;
      __builtin_nontemporal_store(x_, reinterpret_cast<nt_d2*>(row) + lane + 64 * j);
#else
      reinterpret_cast<double2*>(row)[lane + 64 * j] = val;
#endif
      return;
    }
    const int x0 = 2 * (lane + 64 * j);
    if (x0 < X) row[x0] = val.x;
    if (x0 + 1 < X) row[x0 + 1] = val.y;
  };
  const int b0 = blockIdx.x * NT_G;
  // graph u of the batch that starts at the wave's graph g4
  auto graph_of = [&](int g4, int u) { return b0 + wave * GPW + g4 + u; };
  // requests sources q0 .. q0 + PF - 1 (of the fused product; q0 = 0 with n_src = 0: the stored slot) of the batch's graphs
  auto request = [&](int g4, int q0, double2 (&m)[PF][GU][H]) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (f > 0 && q0 + f >= d.n_src) break;
      const size_t so = (size_t)(d.n_src == 0 ? d.in_slot : d.src[q0 + f]) * X;
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int b = graph_of(g4, u);
        const double* gin = d.in + (size_t)(b < d.B ? b : 0) * d.in_ld + so;
#pragma unroll
        for (int j = 0; j < H; ++j) m[f][u][j] = load2(gin, j);
      }
    }
  };
  // v <- v x sources q0 .. in the reference's order, nan_to_num after each product (LBP.py:377-389)
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
  // the batch's messages, given their first PF sources in m: the remaining sources, the renormalisation, the store of the
  // variable->factor message when something later reads it, the LDS image
  auto finish_prologue = [&](int g4, double2 (&m)[PF][GU][H]) {
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
        request(g4, q, m);
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
        // (one division per graph and a multiplication per state: the eight-to-sixteen divisions per lane this was are ~100
        // float64 instructions each side of a launch whose every wave is in this phase at the same time)
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          const double inv = 1.0 / tot[u];
#pragma unroll
          for (int j = 0; j < H; ++j)
            v[u][j] = tot[u] > 0.0 ? make_double2(v[u][j].x * inv, v[u][j].y * inv) : uniform2(j);
        }
      }
      if (d.vf_slot >= 0) {
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (graph_of(g4, u) < d.B) {
            double* o = d.out + (size_t)graph_of(g4, u) * d.out_ld + (size_t)d.vf_slot * X;
#pragma unroll
            for (int j = 0; j < H; ++j) store2(o, j, v[u][j]);
          }
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      mt_t* row = Mt + (wave * GPW + g4 + u) * XP;
      const bool live = graph_of(g4, u) < d.B;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        row[2 * lane + 128 * j] = live ? (mt_t)v[u][j].x : (mt_t)0;
        row[2 * lane + 128 * j + 1] = live ? (mt_t)v[u][j].y : (mt_t)0;
      }
    }
  };

#ifdef MLBP_CONTRACT_NOLOOP          // diagnostic build (tools/contract_probe.py): prologue + epilogue only
  constexpr int KB_RUN = 4;
#else
  constexpr int KB_RUN = KB;
#endif
  {
    double2 m[PF][GU][H];
    for (int g4 = 0; g4 < GPW; g4 += GU) {
      request(g4, 0, m);
      finish_prologue(g4, m);
    }
  }
  __syncthreads();
  double4_t acc[RTW][NCT];
#pragma unroll
  for (int r = 0; r < RTW; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[r][c] = double4_t{0.0, 0.0, 0.0, 0.0};
  const avec* Af = reinterpret_cast<const avec*>(d.frag) + (size_t)blockIdx.y * d.frag_set / (sizeof(avec) / sizeof(TT));
  contract_tiles<TT, RTW, NCT, DEPTH, NW, XP>(acc, Af, 0, 0, KB, KB_RUN, Mt, lane, wave);
  __syncthreads();
  // ---- epilogue: accumulators -> LDS transposed, renormalise, store whole rows (every wave has read its last message
  //      fragment: the image becomes the output) -- or, `dot` mode, the dot product with another stored message ----
  tiles_to_image<TT, RTW, NCT, NW, XP>(acc, Ot, lane, wave);
  __syncthreads();
  for (int g4 = 0; g4 < GPW; g4 += GU) {
    double2 v[GU][H];
    double tot[GU];
    if (d.dot_slot >= 0) {
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int b = graph_of(g4, u);
        const double* row = Ot + (wave * GPW + g4 + u) * XP;
        const double* c = d.in + (size_t)(b < d.B ? b : 0) * d.in_ld + (size_t)d.dot_slot * X;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 cv = load2(c, j);
          part += cv.x * row[2 * lane + 128 * j] + cv.y * row[2 * lane + 128 * j + 1];
        }
        tot[u] = wave_sum64(part);
      }
#pragma unroll
      for (int u = 0; u < GU; ++u)
        if (lane == 0 && graph_of(g4, u) < d.B) d.dots[(size_t)blockIdx.y * d.B + graph_of(g4, u)] = tot[u];
      continue;
    }
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
      const int b = graph_of(g4, u);
      if (b < d.B) {
        double* o = d.out + (size_t)b * d.out_ld + (size_t)d.dst_slot * X;
        const double inv = 1.0 / tot[u];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          double2 val = v[u][j];
          if (d.normalize) val = tot[u] > 0.0 ? make_double2(val.x * inv, val.y * inv) : make_double2(uniform, uniform);
          store2(o, j, val);
        }
      }
    }
  }
}

// X > 1024: the table as d.passes x d.passes blocks of CH = 64 RT states (see the head of the file).  16 graphs per workgroup,
// wave w owns graphs w GPW .. in the vector phases and the row tiles w, w + NW, ... of a block row in the matrix phase.
// Messages have d.X <= passes * CH states, any parity (8-byte accesses, like the PADDED instances above).
template <typename TT, int RT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 16 ? 4 : 1) void contract_chunked_kernel(ContractDev d) {
  constexpr int CH = 64 * RT, XP = CH + 2;
  constexpr int RTW = 4 * RT / NW, GPW = 16 / NW, H = RT / 2;
  static_assert((4 * RT) % NW == 0 && 16 % NW == 0 && RT % 2 == 0, "");
  constexpr int KS = Frag<TT>::KSTEPS, KBC = CH / (4 * KS);      // fragment blocks of one chunk along k
  static_assert(KBC % 4 == 0, "");
  typedef typename Frag<TT>::vec avec;
  typedef typename Frag<TT>::lds_t mt_t;
  extern __shared__ double lds_raw[];
  mt_t* Mt = reinterpret_cast<mt_t*>(lds_raw);
  double* Ot = lds_raw;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int X = d.X, RP = d.passes;
  const double uniform = 1.0 / (double)X;
  const int b0 = blockIdx.x * 16;
  // states x0 = first + 2 (lane + 64 j), x0 + 1 of a row of X doubles
  auto load2 = [&](const double* row, int first, int j) {
    const int x0 = first + 2 * (lane + 64 * j);
    return make_double2(x0 < X ? row[x0] : 0.0, x0 + 1 < X ? row[x0 + 1] : 0.0);
  };
  auto store2 = [&](double* row, int first, int j, double2 val) {
    const int x0 = first + 2 * (lane + 64 * j);
    if (x0 < X) row[x0] = val.x;
    if (x0 + 1 < X) row[x0 + 1] = val.y;
  };
  auto uniform2 = [&](int first, int j) {
    const int x0 = first + 2 * (lane + 64 * j);
    return make_double2(x0 < X ? uniform : 0.0, x0 + 1 < X ? uniform : 0.0);
  };
  // ---- phase 1: the input message of every graph, whole, in memory: a stored slot as it stands, or the fused variable product
  //      (LBP.py:377-389) formed chunk by chunk -- stored unnormalised with its total collected, then divided in place (the
  //      wave reads back its own stores) -- in the message's own slot when something reads it later, else in xbuf ----
  const double* xin[GPW];
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    const int b = b0 + wave * GPW + u, bc = b < d.B ? b : 0;
    if (d.n_src == 0) { xin[u] = d.in + (size_t)bc * d.in_ld + (size_t)d.in_slot * X; continue; }
    double* xo = d.vf_slot >= 0 ? d.out + (size_t)bc * d.out_ld + (size_t)d.vf_slot * X : d.xbuf + (size_t)bc * X;
    xin[u] = xo;
    if (b >= d.B) continue;
    double part = 0.0;
    for (int c = 0; c < RP; ++c) {
      double2 v[H];
#pragma unroll
      for (int j = 0; j < H; ++j) v[j] = uniform2(c * CH, j);
      for (int q = 0; q < d.n_src; ++q) {
        const double* s = d.in + (size_t)bc * d.in_ld + (size_t)d.src[q] * X;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 m = load2(s, c * CH, j);
          double px = m.x * v[j].x, py = m.y * v[j].y;
          if (__builtin_expect(__any(!__builtin_isfinite(px) || !__builtin_isfinite(py)), 0)) { px = nan_to_num(px); py = nan_to_num(py); }
          v[j] = make_double2(px, py);
        }
      }
#pragma unroll
      for (int j = 0; j < H; ++j) { part += v[j].x + v[j].y; store2(xo, c * CH, j, v[j]); }
    }
    if (d.normalize) {
      const double tot = wave_sum64(part), inv = 1.0 / tot;
      for (int c = 0; c < RP; ++c)
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 v = load2(xo, c * CH, j);
          store2(xo, c * CH, j, tot > 0.0 ? make_double2(v.x * inv, v.y * inv) : uniform2(c * CH, j));
        }
    }
  }
  const avec* Af = reinterpret_cast<const avec*>(d.frag) + (size_t)blockIdx.y * d.frag_set / (sizeof(avec) / sizeof(TT));
  const int kbt = RP * KBC;
  double tot[GPW];
#pragma unroll
  for (int u = 0; u < GPW; ++u) tot[u] = 0.0;
  for (int rp = 0; rp < RP; ++rp) {
    double4_t acc[RTW][1];
#pragma unroll
    for (int r = 0; r < RTW; ++r) acc[r][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < RP; ++kc) {
      __syncthreads();                                           // the image is free (and phase 1's stores are visible)
#pragma unroll
      for (int u = 0; u < GPW; ++u) {
        mt_t* row = Mt + (wave * GPW + u) * XP;
        const bool live = b0 + wave * GPW + u < d.B;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 v = live ? load2(xin[u], kc * CH, j) : make_double2(0.0, 0.0);
          row[2 * lane + 128 * j] = (mt_t)v.x;
          row[2 * lane + 128 * j + 1] = (mt_t)v.y;
        }
      }
      __syncthreads();
      contract_tiles<TT, RTW, 1, 2, NW, XP>(acc, Af, rp * (CH / 16), kc * KBC, kbt, KBC, Mt, lane, wave);
    }
    __syncthreads();
    tiles_to_image<TT, RTW, 1, NW, XP>(acc, Ot, lane, wave);
    __syncthreads();
    // this block row of the results: its share of the total (or of the dot product), stored unnormalised
#pragma unroll
    for (int u = 0; u < GPW; ++u) {
      const int b = b0 + wave * GPW + u;
      if (b >= d.B) continue;
      const double* row = Ot + (wave * GPW + u) * XP;
      if (d.dot_slot >= 0) {
        const double* c = d.in + (size_t)b * d.in_ld + (size_t)d.dot_slot * X;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 cv = load2(c, rp * CH, j);
          tot[u] += cv.x * row[2 * lane + 128 * j] + cv.y * row[2 * lane + 128 * j + 1];
        }
      } else {
        double* o = d.out + (size_t)b * d.out_ld + (size_t)d.dst_slot * X;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const double2 v = make_double2(row[2 * lane + 128 * j], row[2 * lane + 128 * j + 1]);
          tot[u] += v.x + v.y;
          store2(o, rp * CH, j, v);
        }
      }
    }
  }
  // ---- phase 3: Message.renormalize (LBP.py:649-657) over the rows the wave has just written ----
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    const int b = b0 + wave * GPW + u;
    if (b >= d.B) continue;
    const double total = wave_sum64(tot[u]);
    if (d.dot_slot >= 0) {
      if (lane == 0) d.dots[(size_t)blockIdx.y * d.B + b] = total;
      continue;
    }
    if (!d.normalize) continue;
    double* o = d.out + (size_t)b * d.out_ld + (size_t)d.dst_slot * X;
    const double inv = 1.0 / total;
    for (int c = 0; c < RP; ++c)
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const double2 v = load2(o, c * CH, j);
        store2(o, c * CH, j, total > 0.0 ? make_double2(v.x * inv, v.y * inv) : make_double2(uniform, uniform));
      }
  }
}

int grant_lds(const void* k) {
  static std::mutex mu;
  static std::vector<const void*> granted;
  std::lock_guard<std::mutex> lock(mu);
  for (const void* g : granted)
    if (g == k) return MLBP_OK;
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess)
    return fail(MLBP_EHIP, "contract_kernel: cannot raise the dynamic LDS limit");
  granted.push_back(k);
  return MLBP_OK;
}

template <typename TT, int RT, bool PADDED>
int launch_contract_rt(const ContractDev& d, int nct, int n_sets, hipStream_t st) {
  const int XA = 64 * RT;
  void (*k)(ContractDev) = nullptr;
  int threads = 512;
  if constexpr (RT > 8) {
    // 640 .. 1024 states: one workgroup per CU (its image is 82 .. 131 KiB); 16 waves when the row tiles divide among them
    static_assert(PADDED && sizeof(TT) == 8, "");
    if constexpr ((4 * RT) % 16 == 0) { k = contract_kernel<TT, RT, 1, 2, 16, true>; threads = 1024; }
    else k = contract_kernel<TT, RT, 1, 2, 8, true>;
    nct = 1;
  } else if constexpr (PADDED) {
    k = contract_kernel<TT, RT, 1, 2, 8, true>;
    nct = 1;
  } else {
    // 32 graphs per workgroup (float64, small tables, big batches): 4 waves, deep fragment prefetch; else 16 graphs per
    // workgroup and 8 waves (two workgroups = 4 waves per SIMD hide each other's stalls; measured 3-7 % over 4 waves)
    if constexpr (RT <= 4 && sizeof(TT) == 8) {
      if (nct == 2) { k = contract_kernel<TT, RT, 2, 4, 4, false>; threads = WG; }
    }
    if (!k) { nct = 1; k = sizeof(TT) == 8 ? contract_kernel<TT, RT, 1, 2, 8, false> : contract_kernel<TT, RT, 1, 4, 8, false>; }
  }
  if (int e = grant_lds((const void*)k)) return e;
  const size_t lds_bytes = (size_t)16 * nct * (XA + 2) * sizeof(double);      // (the float32 image is reused as the float64 output image)
  hipLaunchKernelGGL(k, dim3((d.B + 16 * nct - 1) / (16 * nct), n_sets), dim3(threads), lds_bytes, st, d);
  return MLBP_OK;
}

template <int RT>
int launch_chunked_rt(const ContractDev& d, int n_sets, hipStream_t st) {
  void (*k)(ContractDev) = nullptr;
  int threads = 512;
  if constexpr ((4 * RT) % 16 == 0) { k = contract_chunked_kernel<double, RT, 16>; threads = 1024; }
  else k = contract_chunked_kernel<double, RT, 8>;
  if (int e = grant_lds((const void*)k)) return e;
  hipLaunchKernelGGL(k, dim3((d.B + 15) / 16, n_sets), dim3(threads), (size_t)16 * (64 * RT + 2) * sizeof(double), st, d);
  return MLBP_OK;
}

// How the contraction of X states runs: in `passes` x `passes` blocks of `chunk` states (passes == 1: the one-image kernel
// at XA = chunk), XA = passes * chunk >= X.
struct ContractShape { int passes, chunk, XA; };
ContractShape contract_shape(int X) {
  ContractShape s;
  s.passes = (X + 1023) / 1024;
  const int per = (X + s.passes - 1) / s.passes;
  s.chunk = (per + 127) / 128 * 128;
  s.XA = s.passes * s.chunk;
  return s;
}
int padded_states(int X) { return contract_shape(X).XA; }

// n_sets: fragment sets contracted side by side (blockIdx.y; the gradient's four tables per factor)
template <typename TT>
int launch_contract(ContractDev d, int X, int n_sets, hipStream_t st) {
  const bool f32 = sizeof(TT) == 4;
  const ContractShape sh = contract_shape(X);
  d.X = X;
  d.passes = sh.passes;
  d.frag_set = (size_t)sh.XA * sh.XA;
  if (sh.passes > 1) {
    if constexpr (sizeof(TT) == 8) {
      switch (sh.chunk) {
        case 640: return launch_chunked_rt<10>(d, n_sets, st);
        case 768: return launch_chunked_rt<12>(d, n_sets, st);
        case 896: return launch_chunked_rt<14>(d, n_sets, st);
        case 1024: return launch_chunked_rt<16>(d, n_sets, st);
      }
    }
    return fail(MLBP_EUNSUPPORTED, "shared-table contraction: X = %d%s", X, f32 ? " with float32 tables (256 or 512)" : "");
  }
  const int XA = sh.XA;
  if (XA != X || XA > 512) {
    if constexpr (sizeof(TT) == 8) {
      switch (XA) {
        case 128: return launch_contract_rt<TT, 2, true>(d, 1, n_sets, st);
        case 256: return launch_contract_rt<TT, 4, true>(d, 1, n_sets, st);
        case 384: return launch_contract_rt<TT, 6, true>(d, 1, n_sets, st);
        case 512: return launch_contract_rt<TT, 8, true>(d, 1, n_sets, st);
        case 640: return launch_contract_rt<TT, 10, true>(d, 1, n_sets, st);
        case 768: return launch_contract_rt<TT, 12, true>(d, 1, n_sets, st);
        case 896: return launch_contract_rt<TT, 14, true>(d, 1, n_sets, st);
        case 1024: return launch_contract_rt<TT, 16, true>(d, 1, n_sets, st);
      }
    }
    return fail(MLBP_EUNSUPPORTED, "shared-table contraction: X = %d with float32 tables (256 or 512)", X);
  }
  // graphs per workgroup: 32 when the batch still fills the chip twice over and the accumulators fit, else 16
  const int nct = (!f32 && X <= 256 && d.B >= 32 * 512 && d.dot_slot < 0) ? 2 : 1;
  switch (X) {
    case 128: return launch_contract_rt<TT, 2, false>(d, nct, n_sets, st);
    case 256: return launch_contract_rt<TT, 4, false>(d, nct, n_sets, st);
    case 384: return launch_contract_rt<TT, 6, false>(d, nct, n_sets, st);
    case 512: return launch_contract_rt<TT, 8, false>(d, nct, n_sets, st);
  }
  return fail(MLBP_EUNSUPPORTED, "shared-table contraction: X = %d (65 .. 4096)", X);
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
// grad_en_en[b][k] += phi[l0][l1][k] - S_k[b] / Z[b]   (au.normalize: zero-sum -> expectation 0), S = dots [4][B]
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

template <typename TT>
void launch_table_frag(const TT* T, const double* planes, int n_planes, int n_sets, int X, int transpose, TT* frag, hipStream_t st) {
  constexpr int E = sizeof(TT) == 8 ? 2 : 4;
  const size_t XA = (size_t)padded_states(X);
  hipLaunchKernelGGL((table_frag_kernel<TT, E>), dim3((unsigned)((XA * XA + 255) / 256), n_sets), dim3(256), 0, st, T, planes, n_planes, X, (int)XA,
                     transpose, frag);
}

bool contract_supports(int X) { return X > 64 && X <= 4096; }

}  // namespace

int gemm_path_ready() { return MLBP_OK; }          // hand-written: nothing to load
bool gemm_path_supports(int X) { return contract_supports(X); }

// workspace of the pairwise gradient: the four weighted fragment sets of one factor [4][XA][XA], the dot products [4][B]
size_t gemm_gradient_workspace_bytes(const mlbp_gradient_args* a) {
  const size_t XA = (size_t)padded_states(a->X);
  return (4 * XA * XA + 4 * (size_t)a->B) * sizeof(double);
}

// Three launches per pairwise factor: its four tables T (.) phi_k (k = 0..2) and T in fragment order; ONE contraction launch over
// the four (blockIdx.y) whose epilogue takes S_k[b] = c_b . ((T (.) phi_k) . r_b) on the spot; the combine.
int launch_gemm_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream) {
  if (!contract_supports(a->X)) return fail(MLBP_EUNSUPPORTED, "shared-table gradient: X = %d", a->X);
  hipStream_t st = (hipStream_t)stream;
  const int B = a->B, X = a->X;
  const size_t XA = (size_t)padded_states(X);
  const size_t need = gemm_gradient_workspace_bytes(a);
  void* ws = a->workspace;
  if (ws) {
    if ((size_t)a->workspace_bytes < need) return fail(MLBP_EINVAL, "mlbp_gradient_f64: workspace of %zu bytes, %zu needed", (size_t)a->workspace_bytes, need);
  } else if (int e = fallback_scratch(SCRATCH_GEMM_GRADIENT, need, &ws)) return e;
  double* W = (double*)ws; double* S = W + 4 * XA * XA;
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
    // set k: (T (.) phi_k) for k < 3, T itself (the normaliser) for k = 3;  r = msgs[:, r_slot, :], c = msgs[:, c_slot, :]
    launch_table_frag<double>(T, planes, 3, 4, X, 0, W, st);
    ContractDev d = {};
    d.frag = W; d.in = a->msgs; d.in_ld = (size_t)a->n_msgs * X; d.out = nullptr; d.out_ld = 0;
    d.n_src = 0; d.in_slot = h_r[p]; d.vf_slot = -1; d.dst_slot = 0; d.B = B; d.normalize = 0;
    d.dot_slot = h_c[p]; d.dots = S;
    if (int e = launch_contract<double>(d, X, 4, st)) return e;
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
  const ContractShape sh = contract_shape(X);
  const size_t XA = (size_t)sh.XA;
  if (f32 && !(X == 256 || X == 512))
    return fail(MLBP_EUNSUPPORTED, "shared-table contraction path: float32 tables need X = 256 or 512");
  // fragment-ordered copies of the distinct tables, both orientations: [P][2][XA * XA] -- scratch the PROGRAM owns (calls with
  // different programs run on different streams; an outgrown block stays alive, so a captured graph stays valid)
  mlbp_program* mp = const_cast<mlbp_program*>(prog);
  if (int e = program_grow(mp, &mp->d_gfrag, &mp->gfrag_cap, (size_t)prog->P * 2 * XA * XA * elem)) return e;
  void* frag = mp->d_gfrag;
  double* xbuf = nullptr;
  if (sh.passes > 1) {                             // the chunked kernel parks a fused update's input message in memory
    if (int e = program_grow(mp, &mp->d_gxbuf, &mp->gxbuf_cap, (size_t)B * X * sizeof(double))) return e;
    xbuf = static_cast<double*>(mp->d_gxbuf);
  }
  {
    HostRow row = {};
    for (int p = 0; p < prog->P; ++p) row.v[p] = a->pair_tab_host[p];
    hipLaunchKernelGGL(check_shared_claim_kernel, dim3((B * prog->P + 255) / 256), dim3(256), 0, st, a->pair_tab, B, prog->P,
                       a->n_pair_tables, row, prog->d_status);
  }
  for (int p = 0; p < prog->P; ++p)
    for (int tr = 0; tr < 2; ++tr) {
      const size_t off = ((size_t)p * 2 + tr) * XA * XA;
      if (f32) launch_table_frag<float>(a->pair_tables_f32 + (size_t)a->pair_tab_host[p] * X * X, nullptr, 0, 1, X, tr, (float*)frag + off, st);
      else launch_table_frag<double>(a->pair_tables + (size_t)a->pair_tab_host[p] * X * X, nullptr, 0, 1, X, tr, (double*)frag + off, st);
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
    d.dot_slot = -1; d.xbuf = xbuf;
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
    if (int e = f32 ? launch_contract<float>(d, X, 1, st) : launch_contract<double>(d, X, 1, st)) return e;
  }
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "shared-table contraction path: a launch failed");
  return MLBP_OK;
}

}  // namespace mlbp
