// Batched sum-product sweeps for gfx950 (MI355X): the hot path of libmlbp.so.
//
// One 256-thread workgroup (4 wavefronts of 64) owns one factor graph for ALL sweeps of a call:
//   * the graph's messages (n_msgs x X f64) are staged into LDS once, updated there and written
//     back once -- the per-update traffic of the reference's `graph.messages` dict never reaches
//     HBM;
//   * pairwise potential tables are streamed HBM -> VGPR with 16-byte-per-lane loads (8 rows of
//     the 64x64 table = 4 KiB per workgroup load instruction, fully coalesced in BOTH message
//     directions), and the table of the NEXT pairwise update is prefetched into a second register
//     set while the current update reduces/normalises, so every workgroup keeps 32 KiB in flight;
//   * the op list is identical for every graph of the launch (wave-uniform control flow, scalar
//     loads).
// Updates inside a graph are strictly sequential (Gauss-Seidel order of LBP.py:227-243);
// parallelism comes from batch x |X| only.
//
// Algorithmic HBM bytes per pairwise update: (X*X + 2X) * 8.
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <map>
#include <vector>

#include "mlbp_internal.h"
#include "mlbp_device.h"

using mlbp::fail;
using namespace mlbp_dev;

namespace {

constexpr int WG = 256;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(MLBP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// Uniform check of the graph's table indices; an out-of-range index would be an out-of-bounds
// read, so the whole graph is skipped and the status word raised instead.
__device__ __forceinline__ bool tables_in_range(const SweepDev& d, int g) {
  bool ok = true;
  for (int i = threadIdx.x; i < d.P; i += WG)
    ok &= (unsigned)d.pair_tab[(size_t)g * d.P + i] < (unsigned)d.n_pair_tables;
  for (int i = threadIdx.x; i < d.U; i += WG)
    ok &= (unsigned)d.unary_tab[(size_t)g * d.U + i] < (unsigned)d.n_unary_tables;
  int all_ok = __syncthreads_and(ok ? 1 : 0);
  if (!all_ok && threadIdx.x == 0) atomicExch(d.status, 1);
  return all_ok != 0;
}

// ------------------------------------------------------------------------------------------------
// X = 64, float64, the EXACT kernel ("fused" program form, see build_fused_program): thread t holds rows 8k + (t >> 5),
// columns 2 (t & 31) .. +1 of a resident table (as double2 element k * 256 + t).  It normalises after every update like the
// reference and is the fix-up pass behind the fast kernels (mlbp_lean.hip, mlbp_shared.hip), the kernel for
// normalize_messages = False and for programs whose unary updates cannot be hoisted:
//   * unary factor->variable messages depend on nothing but their table, so when the program
//     allows it (every such slot is written before it is first read) they are all computed once,
//     in parallel over the 4 waves, before the first sweep, and the UNARY ops are dropped -- the
//     values are bit-identical to recomputing them every sweep as the reference does;
//   * a variable->factor update that feeds the next pairwise update is fused into it: every wave
//     forms the product redundantly (lane i = state i), so no barrier separates the two;
//   * every wave finishes every update redundantly (same inputs, same order => same bits) and
//     stores the result itself, so a wave only ever reads back its own LDS writes; the single
//     workgroup barrier per pairwise update orders the exchange of partial sums, which lives in a
//     double-buffered scratch;
//   * NT > 0: the graph's (at most NT) pairwise tables are loaded ONCE into registers and reused
//     by every sweep of the call (16 f64 per thread and table); NT == 0 streams with a one-table
//     register prefetch as before;
//   * init != 0 starts from uniform messages instead of reading them (FactorGraph.initialize fused).
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (-DMLBP_STAMPS, tools/stamp_profile.py): per-phase shader-clock sums of
// workgroup 0..15's wave 0 go to a side buffer that no other code reads.  Never defined in the
// shipped library.
#ifdef MLBP_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP_DECL unsigned long long _t0 = 0, _ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t0) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long _t1; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t1) :: "memory"); __builtin_amdgcn_sched_barrier(0); _ph[i] += _t1 - _t0; _t0 = _t1; }
#define STAMP_FLUSH if (g_stamp_buf && blockIdx.x < 64 && threadIdx.x == 0) { for (int _i = 0; _i < 8; ++_i) g_stamp_buf[blockIdx.x * 8 + _i] = _ph[_i]; }
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Diagnostic build only (-DMLBP_ABLATE, tools/ablate_profile.py): MLBP_ABLATE_MASK removes one phase
// of the fused kernel at a time (results become wrong; only the timing delta matters).
#ifdef MLBP_ABLATE
__device__ int g_ablate_mask = 0;
#define ABLATED(bit) (ablate_mask_ & (1 << (bit)))
#define ABLATE_DECL const int ablate_mask_ = __builtin_amdgcn_readfirstlane(g_ablate_mask);
#else
#define ABLATED(bit) 0
#define ABLATE_DECL
#endif

// VariableNode.get_marginal (LBP.py:392-400) for every variable of the graph, straight from the
// normalised messages in LDS: uniform x incoming messages in facset order, nan_to_num after each
// product, renormalise.  Wave w takes variables w, w+4, ...
__device__ __forceinline__ void marginals_from_lds_x64(const SweepDev& d, const double* msg, int g, int wave, int lane) {
  if (!d.marginals) return;
  const const_i32p off = as_const(d.readout), slots = as_const(d.readout + d.n_vars + 1);
  for (int v = wave; v < d.n_vars; v += 4) {
    double acc = 1.0 / 64.0;
    for (int q = off[v]; q < off[v + 1]; ++q) acc = mul_nan_to_num(msg[slots[q] * 64 + lane], acc);
    d.marginals[((size_t)g * d.n_vars + v) * 64 + lane] = renorm(acc, wave_sum(acc), 1.0 / 64.0, true);
  }
}

using mlbp::FOP_UNARY; using mlbp::FOP_PAIR_TM; using mlbp::FOP_PAIR_MT; using mlbp::FOP_VAR;
using mlbp::FOP_VAR_PAIR_TM; using mlbp::FOP_VAR_PAIR_MT; using mlbp::FOP_BUNDLED;

constexpr int LP_MAX_BLOCKS = 4096;
constexpr int LP_WG = 256;
__device__ double g_lp_partials[LP_MAX_BLOCKS];
__device__ unsigned g_lp_done[2] = {0, 0};

// FactorGraph.get_posterior_probs (LBP.py:247-259) behind the sweeps of the same call (train_mp.py:381-400 calls one after the
// other): the fix-up pass, which walks every graph's flag anyway, also takes the log-posteriors and their batch sum.
struct PosteriorDev {
  const int32_t* labels;   // [B][n_vars] or NULL (off)
  double* out;             // [B]
  double* sum_out;         // [1] or NULL
  unsigned generation;     // of the batch sum's arrival counter (see log_posterior_kernel)
  int32_t pad_;
};

struct FusedDev {
  PosteriorDev post;
  const uint8_t* only;     // when non-NULL: run only graphs with only[g] != 0 (fix-up pass)
  const int32_t* image;    // fused op headers, source lists, hoist list, constant-product lists (one block)
  const int32_t* fsweeps;  // [n_sweeps][2]
  int32_t n_fops, n_psrcs, n_hoist, n_cprod, n_cpw, n_ext, init;
  int32_t n_graphs;        // B
};
constexpr int FIXUP_GRAPHS_PER_WG = 64;

__device__ __forceinline__ void wg_barrier() {
  // LDS traffic only: wait for this wave's LDS ops, then the workgroup barrier.  Outstanding
  // global loads (table prefetch) stay in flight across it.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool MT>
__device__ __forceinline__ void pair_partials(const double2 (&T)[8], const double* m, double* red, int rg, int cp,
                                              int lane) {
  if (MT) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double mi = m[8 * k + rg];
      a0 += mi * T[k].x;
      a1 += mi * T[k].y;
    }
    reinterpret_cast<double2*>(red + rg * 64)[cp] = make_double2(a0, a1);
  } else {
    const double2 mj = reinterpret_cast<const double2*>(m)[cp];
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = T[k].x * mj.x + T[k].y * mj.y;
    // transposing butterfly over the 16 lanes of a DPP row: each step pairs lane l with its mirror
    // and halves the live values (8 -> 4 -> 2 -> 1); "up" lanes keep the upper half.
    {
      const bool up = lane & 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double send = up ? v[i] : v[i + 4], keep = up ? v[i + 4] : v[i];
        v[i] = keep + dpp_mov<0x140>(send);
      }
    }
    {
      const bool up = lane & 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double send = up ? v[i] : v[i + 2], keep = up ? v[i + 2] : v[i];
        v[i] = keep + dpp_mov<0x141>(send);
      }
    }
    {
      const bool up = lane & 2;
      const double send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
      v[0] = keep + dpp_mov<0x1B>(send);
    }
    v[0] += dpp_mov<0xB1>(v[0]);
    // lane holds row 8k+rg summed over the 32 columns of its 16-lane row; the other 32 columns
    // sit in the neighbouring row of the same half-wave: both go to LDS, added after the barrier.
    if ((lane & 1) == 0) {
      const int k = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
      red[((lane >> 4) & 1) * 64 + 8 * k + rg] = v[0];
    }
  }
}

// ---- gradient fused into the sweep epilogue (X = 64, F = (3, 6), tables resident) --------------------
// After the last sweep the workgroup still holds what FactorGraph.get_unregularized_gradeint needs
// (LBP.py:301-320): the pairwise tables in registers, the normalised messages in LDS -- and a unary
// factor's normalised message IS its belief vector au.normalize(table) whenever the table's total is
// positive (LBP.py:540 vs 494-498).  Only the shared feature tensors come from L2.  Saves the
// standalone gradient kernel's second pass over every table in HBM.
// umsg[u] = message slot of unary slot u's factor->variable message, upos[u] = 1 when its table total
// was positive (both filled in the prologue from the hoist list).  scratch: >= 4*24 doubles of LDS.
template <int NT>
__device__ __forceinline__ void gradient_epilogue_x64(const SweepDev& d, const GradFusedDev& gf, const double2 (&tab)[NT][8],
                                                      const double* msg, const int32_t* umsg, const int32_t* upos,
                                                      double* scratch, int g) {
  const int32_t* ukind = upos + d.U;      // staged by the prologue right behind upos: kind, observed
  const int32_t* uobs = ukind + d.U;      // column and label of every unary factor of this graph
  const int32_t* ulab = uobs + d.U;
  constexpr int FEE = 3, FED = 6;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, rg = t >> 5, cp = t & 31;
  double pacc[NT][FEE + 1];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
#pragma unroll
    for (int f = 0; f <= FEE; ++f) pacc[p][f] = 0.0;
    if (p < d.P) {
      const double* c = msg + gf.pair_c_slot[p] * 64;
      const double* r = msg + gf.pair_r_slot[p] * 64;
      // interleaved [64][64][F] features: measured faster here than the planar copies the standalone
      // gradient kernel prefers (0.517 vs 0.55 ms per fused launch, tools/time_train_step.py)
      const double* phi = gf.pair_phi[p] ? gf.phi_en_en_w1 : gf.phi_en_en;
      const double2 rj = reinterpret_cast<const double2*>(r)[cp];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double ci = c[8 * k + rg];
        const double w0 = (ci * rj.x) * tab[p][k].x, w1 = (ci * rj.y) * tab[p][k].y;
        pacc[p][0] += w0 + w1;
        const double* ph = phi + ((size_t)(8 * k + rg) * 64 + 2 * cp) * FEE;
#pragma unroll
        for (int f = 0; f < FEE; ++f) pacc[p][1 + f] += w0 * ph[f] + w1 * ph[FEE + f];
      }
#pragma unroll
      for (int f = 0; f <= FEE; ++f) pacc[p][f] = wave_sum(pacc[p][f]);
    }
  }
  // unary factors: wave w takes slots w, w+4, ...; belief = normalised message (zero when the table
  // total was <= 0).  kind / observed column / label were staged in LDS by the prologue; the feature
  // slabs of a whole batch of factors are requested before the first reduction.
  double uee[FEE] = {0.0, 0.0, 0.0}, ued[FED] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  bool bad = false;
  constexpr int UB = 3;
  for (int u0 = wave; u0 < d.U; u0 += 4 * UB) {
    double b[UB], pv[UB][FED];
    int kd[UB], at[UB];                       // kind and scalar offset of the label's feature row
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = u0 + 4 * j;
      kd[j] = -1;
      if (u < d.U) {
        const int kind = __builtin_amdgcn_readfirstlane(ukind[u]), obs = __builtin_amdgcn_readfirstlane(uobs[u]);
        const int lab = __builtin_amdgcn_readfirstlane(ulab[u]);
        const int cols = kind == 2 ? gf.Vde : 64;
        if ((unsigned)obs >= (unsigned)cols || (unsigned)lab >= 64u) { bad = true; continue; }
        kd[j] = kind;
        b[j] = upos[u] ? msg[umsg[u] * 64 + lane] : 0.0;
        if (kind == 2) {
          const double* ph = gf.phi_en_de_t + ((size_t)obs * 64 + lane) * FED;
          at[j] = (obs * 64 + lab) * FED;
#pragma unroll
          for (int f = 0; f < FED; ++f) pv[j][f] = ph[f];
        } else {
          const double* ph = (kind ? gf.phi_en_en_w1_t : gf.phi_en_en_t) + ((size_t)obs * 64 + lane) * FEE;
          at[j] = (obs * 64 + lab) * FEE;
#pragma unroll
          for (int f = 0; f < FEE; ++f) pv[j][f] = ph[f];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (kd[j] == 2) {
#pragma unroll
        for (int f = 0; f < FED; ++f) ued[f] += gf.phi_en_de_t[at[j] + f] - wave_sum(b[j] * pv[j][f]);
      } else if (kd[j] >= 0) {
        const double* base = kd[j] ? gf.phi_en_en_w1_t : gf.phi_en_en_t;
#pragma unroll
        for (int f = 0; f < FEE; ++f) uee[f] += base[at[j] + f] - wave_sum(b[j] * pv[j][f]);
      }
    }
  }
  if (bad && lane == 0) atomicExch(d.status, 1);
  // combine the four waves: per wave NT*(FEE+1) pair sums + FEE + FED unary sums
  constexpr int PER = NT * (FEE + 1) + FEE + FED;
  if (lane == 0) {
    double* o = scratch + wave * PER;
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
      for (int f = 0; f <= FEE; ++f) o[p * (FEE + 1) + f] = pacc[p][f];
#pragma unroll
    for (int f = 0; f < FEE; ++f) o[NT * (FEE + 1) + f] = uee[f];
#pragma unroll
    for (int f = 0; f < FED; ++f) o[NT * (FEE + 1) + FEE + f] = ued[f];
  }
  wg_barrier();
  if (t == 0) {
    double tot[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) tot[q] = (scratch[q] + scratch[PER + q]) + (scratch[2 * PER + q] + scratch[3 * PER + q]);
    double gee[FEE];
#pragma unroll
    for (int f = 0; f < FEE; ++f) gee[f] = tot[NT * (FEE + 1) + f];
    bool lbad = false;
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < d.P) {
        const int l0 = gf.pair_label[((size_t)g * d.P + p) * 2], l1 = gf.pair_label[((size_t)g * d.P + p) * 2 + 1];
        if ((unsigned)l0 >= 64u || (unsigned)l1 >= 64u) { lbad = true; continue; }
        const double* phi = gf.pair_phi[p] ? gf.phi_en_en_w1 : gf.phi_en_en;
        const double Z = tot[p * (FEE + 1)];
#pragma unroll
        for (int f = 0; f < FEE; ++f)
          gee[f] += phi[((size_t)l0 * 64 + l1) * FEE + f] - (Z > 0.0 ? tot[p * (FEE + 1) + 1 + f] / Z : 0.0);
      }
    }
    if (lbad) atomicExch(d.status, 1);
#pragma unroll
    for (int f = 0; f < FEE; ++f) gf.grad_en_en[(size_t)g * FEE + f] = gee[f];
#pragma unroll
    for (int f = 0; f < FED; ++f) gf.grad_en_de[(size_t)g * FED + f] = tot[NT * (FEE + 1) + FEE + f];
  }
}

// Register budget: the resident tables take 32 VGPRs each; the waves-per-SIMD floor keeps the
// allocator at 3 workgroups per CU with 3 resident tables (<= 168 VGPRs) and 4 otherwise (<= 128).
//
// LDS image of one workgroup (doubles unless noted):
//   msg   [n_msgs][64]      the graph's messages
//   ext   [n_ext][64]       slot n_msgs = the uniform vector; then one "constant product" per
//                           distinct set of hoisted unary messages a variable multiplies in
//   gin   [64]              input vector of the pairwise update in flight
//   red   [8][64]           partial sums of the contraction
//   prog  int32             fused op headers [n_fops][8], source lists, hoist / constant-product
//                           lists, the graph's table indices
template <bool NORM, int NT, bool GRAD>
__device__ __forceinline__ void sweep_x64_fused_body(const SweepDev& d, const FusedDev& f, const GradFusedDev& gf, const int g) {
  extern __shared__ double lds[];
  double* msg = lds;
  double* gin = lds + (size_t)(d.n_msgs + f.n_ext) * 64;
  double* red = gin + 64;
  int32_t* prog = reinterpret_cast<int32_t*>(red + 512);
  const int32_t* psrcs = prog + f.n_fops * 8;          // [n_psrcs]
  const int32_t* phoist = psrcs + f.n_psrcs;            // [n_hoist][2]
  const int32_t* pcp = phoist + 2 * f.n_hoist;          // [n_cpw] constant-product lists: count, slots...
  int32_t* tabidx = const_cast<int32_t*>(pcp) + f.n_cpw;  // [P + U] this graph's table indices
  int32_t* flags = tabidx + d.P + d.U;                  // [0] = vector wave already stored the v->f message
  int32_t* umsg = flags + 4;                            // [U] message slot of each unary factor's message
  int32_t* upos = umsg + d.U;                           // [U] 1 when that factor's table total was positive

  STAMP_DECL
  ABLATE_DECL
  STAMP_START
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int rg = t >> 5, cp = t & 31;
  double* gm = d.msgs + (size_t)g * d.n_msgs * 64;
  const double uniform = 1.0 / 64.0;
  const const_i32p c_fsweeps = as_const(f.fsweeps), c_pairseq = as_const(d.pairseq);

  // ---- phase A: table indices (range-checked), program image and messages into LDS ----
  bool ok = true;
  for (int i = t; i < d.P + d.U; i += WG) {
    const int v = i < d.P ? d.pair_tab[(size_t)g * d.P + i] : d.unary_tab[(size_t)g * d.U + (i - d.P)];
    ok &= (unsigned)v < (unsigned)(i < d.P ? d.n_pair_tables : d.n_unary_tables);
    tabidx[i] = v;
  }
  {
    const int n_img = f.n_fops * 8 + f.n_psrcs + 2 * f.n_hoist + f.n_cpw;
    for (int i = t; i < n_img; i += WG) prog[i] = f.image[i];
    double2* dst = reinterpret_cast<double2*>(msg);
    if (f.init) {
      for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = make_double2(uniform, uniform);
    } else {
      const double2* src = reinterpret_cast<const double2*>(gm);
      for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = src[i];
    }
    if (t < 32) dst[d.n_msgs * 32 + t] = make_double2(uniform, uniform);    // ext slot 0
    if (t == 0) flags[0] = 0;
    if (GRAD)
      for (int i = t; i < d.U; i += WG) {
        umsg[i] = -1; upos[i] = 0;
        upos[d.U + i] = gf.unary_kind[i];
        upos[2 * d.U + i] = gf.unary_obs[(size_t)g * d.U + i];
        upos[3 * d.U + i] = gf.unary_label[(size_t)g * d.U + i];
      }
  }
  if (!__syncthreads_and(ok ? 1 : 0)) {       // an out-of-range table index: skip the graph, raise the status word
    if (t == 0) atomicExch(d.status, 1);
    return;
  }

  // ---- phase B: every HBM load of the prologue in flight together ----
  constexpr int NR = NT > 0 ? NT : 1;
  double2 tab[NR][8];
  int pair_k = 0;
  if (NT > 0) {
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < d.P) {
        const int ti = __builtin_amdgcn_readfirstlane(tabidx[p]);
        const double2* T = reinterpret_cast<const double2*>(d.pair_tables + (size_t)ti * 4096);
#pragma unroll
        for (int k = 0; k < 8; ++k) tab[p][k] = T[k * WG + t];
      }
    }
  } else {
    const int s0 = c_pairseq[0];
    if (s0 >= 0) {
      const int ti = __builtin_amdgcn_readfirstlane(tabidx[s0]);
      const double2* T = reinterpret_cast<const double2*>(d.pair_tables + (size_t)ti * 4096);
#pragma unroll
      for (int k = 0; k < 8; ++k) tab[0][k] = T[k * WG + t];
    }
  }
  // hoisted unary messages: wave w takes entries w, w+4, ...; 8 loads in flight per wave
  constexpr int HB = 8;
  for (int h0 = wave; h0 < f.n_hoist; h0 += 4 * HB) {
    double r[HB];
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const int h = h0 + 4 * j;
      r[j] = (h < f.n_hoist) ? d.unary_tables[(size_t)tabidx[d.P + phoist[2 * h]] * 64 + lane] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const int h = h0 + 4 * j;
      if (h < f.n_hoist) {
        const double tot = wave_sum(r[j]);
        msg[phoist[2 * h + 1] * 64 + lane] = renorm(r[j], tot, uniform, NORM);
        if (GRAD && lane == 0) { umsg[phoist[2 * h]] = phoist[2 * h + 1]; upos[phoist[2 * h]] = tot > 0.0 ? 1 : 0; }
      }
    }
  }
  wg_barrier();
  // constant products: uniform times the hoisted messages a variable multiplies in, in facset
  // order with nan_to_num after each product (the constant prefix of LBP.py:381-386)
  {
    int at = 0;
    for (int k = 0; k < f.n_cprod; ++k) {
      const int cnt = pcp[at];
      if ((k & 3) == wave) {
        double acc = uniform;
        bool clean = true;
        for (int q = 0; q < cnt; ++q) {
          acc *= msg[pcp[at + 1 + q] * 64 + lane];
          clean = clean && __all(__builtin_isfinite(acc));
        }
        // a non-finite intermediate means nan_to_num would have acted inside the constant part:
        // poison the product so that every update using it takes the exact-order path
        msg[(d.n_msgs + 1 + k) * 64 + lane] = clean ? acc : __builtin_nan("");
      }
      at += 1 + cnt;
    }
  }
  wg_barrier();
  STAMP(0)   // prologue

  // Main loop.  Wave 0 is the graph's "vector wave": it alone runs the 64-element work on the
  // critical path of every update (variable product, gathering the partial sums, normalising the
  // factor->variable message); all four waves run the table contraction.  Two LDS-only barriers per
  // pairwise update hand the input vector to the contraction and the partial sums back.
  for (int s = 0; s < d.n_sweeps; ++s) {
    const int op0 = c_fsweeps[2 * s], nop = c_fsweeps[2 * s + 1];
    for (int o = op0; o < op0 + nop; ++o) {
      const int4 h0 = reinterpret_cast<const int4*>(prog)[2 * o];
      const int4 h1 = reinterpret_cast<const int4*>(prog)[2 * o + 1];
      const int kind = __builtin_amdgcn_readfirstlane(h0.x) & 0xFF;
      STAMP(1)   // op header
      if (ABLATED(5)) continue;
      if (kind == FOP_UNARY) {
        if (wave == 0) {
          const int us = __builtin_amdgcn_readfirstlane(h0.y), c = __builtin_amdgcn_readfirstlane(h0.w);
          const double r = d.unary_tables[(size_t)tabidx[d.P + us] * 64 + lane];
          msg[c * 64 + lane] = renorm(r, wave_sum(r), uniform, NORM);
        }
        continue;
      }
      int pslot, dst;
      const double* m;
      const bool from_var = (kind == FOP_VAR || kind == FOP_VAR_PAIR_TM || kind == FOP_VAR_PAIR_MT);
      const int c = __builtin_amdgcn_readfirstlane(h0.w);
      if (from_var) {
        if (wave == 0) {
          // fast form: (constant product) x (the varying messages), valid while every factor is
          // finite and non-negative and the product is not identically zero; then the product's
          // scale cancels in the normalised pairwise update, so the contraction takes it
          // unnormalised and wave 1 normalises and stores the variable->factor message meanwhile
          const int a = __builtin_amdgcn_readfirstlane(h0.y), n = ABLATED(0) ? 0 : __builtin_amdgcn_readfirstlane(h0.z);
          const int4 s4 = *reinterpret_cast<const int4*>(psrcs + a);
          double acc = ABLATED(0) ? uniform : msg[s4.x * 64 + lane];
          if (n > 1) acc *= msg[s4.y * 64 + lane];
          if (n > 2) acc *= msg[s4.z * 64 + lane];
          if (n > 3) acc *= msg[s4.w * 64 + lane];
          for (int q = 4; q < n; ++q) acc *= msg[psrcs[a + q] * 64 + lane];
          const bool fast = ABLATED(6) ? true : (__all(acc >= 0.0 && acc < __builtin_huge_val()) && __any(acc > 0.0));
          STAMP(2)   // variable product
#ifdef MLBP_STAMPS
          if (!fast) _ph[2] += (1ULL << 40);
          if (!__all(acc >= 0.0)) _ph[2] += (1ULL << 44);
          if (!__all(acc < __builtin_huge_val())) _ph[2] += (1ULL << 48);
          if (!__any(acc > 0.0)) _ph[2] += (1ULL << 52);
#endif
          if (ABLATED(7)) {
            asm volatile("" ::"v"(acc));
          } else if (fast && kind != FOP_VAR) {
            gin[lane] = acc;
            if (lane == 0) flags[0] = 0;
          } else {
            // exact reference order (LBP.py:381-389) from the raw messages
            const int ae = __builtin_amdgcn_readfirstlane(h1.z), ne = __builtin_amdgcn_readfirstlane(h1.w);
            double ex = uniform;
            for (int q = 0; q < ne; ++q) ex = mul_nan_to_num(msg[psrcs[ae + q] * 64 + lane], ex);
            const double mn = renorm(ex, wave_sum(ex), uniform, NORM);
            msg[c * 64 + lane] = mn;
            gin[lane] = mn;
            if (lane == 0) flags[0] = 1;
          }
          STAMP(3)   // (slow path only) normalisation
        }
        if (kind == FOP_VAR) continue;
        m = gin;
        pslot = __builtin_amdgcn_readfirstlane(h1.x);
        dst = __builtin_amdgcn_readfirstlane(h1.y);
      } else {
        pslot = __builtin_amdgcn_readfirstlane(h0.y);
        m = msg + __builtin_amdgcn_readfirstlane(h0.z) * 64;
        dst = c;
      }
      const bool mt = (kind == FOP_PAIR_MT || kind == FOP_VAR_PAIR_MT);
      if (!ABLATED(4)) wg_barrier();          // the input vector (and every earlier vector-wave store) is visible
      STAMP(5)   // barrier
      if (ABLATED(1)) {
      } else if (NT > 0) {
#pragma unroll
        for (int p = 0; p < NT; ++p) {
          if (p == pslot) {
            if (mt) pair_partials<true>(tab[p], m, red, rg, cp, lane);
            else pair_partials<false>(tab[p], m, red, rg, cp, lane);
          }
        }
      } else {
        double2 cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = tab[0][k];
        ++pair_k;
        const int sn = c_pairseq[pair_k];
        if (sn >= 0) {
          const int ti = __builtin_amdgcn_readfirstlane(tabidx[sn]);
          const double2* T = reinterpret_cast<const double2*>(d.pair_tables + (size_t)ti * 4096);
#pragma unroll
          for (int k = 0; k < 8; ++k) tab[0][k] = T[k * WG + t];
        }
        if (mt) pair_partials<true>(cur, m, red, rg, cp, lane);
        else pair_partials<false>(cur, m, red, rg, cp, lane);
      }
      if (!ABLATED(3) && from_var && wave == 1 && flags[0] == 0) {
        // off the critical path: normalise and store the variable->factor message (LBP.py:387-389)
        const double v = gin[lane];
        msg[c * 64 + lane] = renorm(v, wave_sum(v), uniform, NORM);
      }
      STAMP(4)   // partial sums
      if (!ABLATED(4)) wg_barrier();          // partial sums are in LDS
      STAMP(5)   // barrier
      if (wave == 0 && ABLATED(2)) {
        if (!ABLATED(8)) msg[dst * 64 + lane] = uniform;
      } else if (wave == 0) {
        double r;
        if (mt) {
          r = 0.0;
#pragma unroll
          for (int q = 0; q < 8; ++q) r += red[q * 64 + lane];
        } else {
          r = red[lane] + red[64 + lane];
        }
        STAMP(6)   // gather partials
        msg[dst * 64 + lane] = renorm(r, wave_sum(r), uniform, NORM);
        STAMP(7)   // normalise
      }
    }
  }
  STAMP_FLUSH
  wg_barrier();
  {
    const double2* src = reinterpret_cast<const double2*>(msg);
    double2* dst = reinterpret_cast<double2*>(gm);
    for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = src[i];
  }
  if (NORM) marginals_from_lds_x64(d, msg, g, wave, lane);
  if (GRAD) gradient_epilogue_x64<(NT > 0 ? NT : 1)>(d, gf, tab, msg, umsg, upos, red, g);       // (NT = 0 with GRAD: no pairwise factor at all -- a single predicted word --, the unary terms only: every pairwise step of the epilogue is behind p < d.P)
}

// Log-posteriors of the block's 64 graphs from the marginals in memory (the fast kernel's, or the ones this workgroup has just
// redone), and -- sum_out given -- their batch sum in a fixed order: lanes, then the blocks' partials by the last block to
// arrive (the two-counter protocol of log_posterior_kernel).
// The label thread t reads when the block's variables fit one wave each (n_vars <= WG / 64: wave v takes variable v of the
// block's 64 graphs), requested by the caller TOGETHER with the flags -- one round of memory latency instead of two; -1: none.
__device__ __forceinline__ int posterior_prefetch_label(const SweepDev& d, const PosteriorDev& p, int base, int n_graphs) {
  const int t = threadIdx.x, v = t >> 6, g = base + (t & 63);
  if (!p.labels || d.n_vars > WG / 64 || v >= d.n_vars || g >= n_graphs) return -1;
  return p.labels[(size_t)g * d.n_vars + v];
}
__device__ __forceinline__ void posterior_of_block(const SweepDev& d, const PosteriorDev& p, int base, int n_graphs, int my_label) {
  __shared__ double part[WG / 64];
  __shared__ double lpv[WG];
  __shared__ bool last;
  __syncthreads();                                   // (the marginals of graphs redone above are this workgroup's own stores)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double total = 0.0;
  const int g = base + t;
  if (d.n_vars <= WG / 64) {
    // thread (variable t >> 6, graph t & 63): label (here already) -> marginal entry -> log, all the block's entries in ONE round
    // of latency; then the graph's terms are added in variable order, as the loop below adds them
    double lp = 0.0;
    if (wave < d.n_vars && base + lane < n_graphs) {
      if ((unsigned)my_label >= 64u) atomicExch(d.status, 1);
      else {
        lp = log(d.marginals[((size_t)(base + lane) * d.n_vars + wave) * 64 + my_label]);
        if (lp == -__builtin_huge_val()) lp = -99.99;        // LBP.py:254-256
      }
    }
    lpv[t] = lp;
    __syncthreads();
    if (t < FIXUP_GRAPHS_PER_WG && g < n_graphs) {
      for (int v = 0; v < d.n_vars; ++v) total += lpv[v * 64 + t];
      p.out[g] = total;
    }
  } else if (t < FIXUP_GRAPHS_PER_WG && g < n_graphs) {
    for (int v = 0; v < d.n_vars; ++v) {
      const int lab = p.labels[(size_t)g * d.n_vars + v];
      if ((unsigned)lab >= 64u) { atomicExch(d.status, 1); continue; }
      const double lp = log(d.marginals[((size_t)g * d.n_vars + v) * 64 + lab]);
      total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
    }
    p.out[g] = total;
  }
  if (!p.sum_out) return;
  if (wave == 0) {
    const double ws = wave_sum(total);
    if (lane == 0) {
      g_lp_partials[blockIdx.x] = ws;
      __threadfence();
      if (blockIdx.x == 0) g_lp_done[(p.generation + 1) & 1] = 0;
      last = atomicAdd(&g_lp_done[p.generation & 1], 1u) == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double acc = 0.0;
  for (unsigned q = t; q < gridDim.x; q += WG) acc += __builtin_nontemporal_load(&g_lp_partials[q]);
  const double fs = wave_sum(acc);
  if (lane == 0) part[wave] = fs;
  __syncthreads();
  if (t == 0) {
    double v = part[0];
    for (int w = 1; w < WG / 64; ++w) v += part[w];
    *p.sum_out = v;
    g_lp_done[p.generation & 1] = 0;
  }
}

// One workgroup per graph -- or, as the fix-up pass behind a fast kernel (f.only), one workgroup per 64 graphs that
// walks their flags and redoes the (normally zero) flagged ones: 128 workgroups instead of 8192 that exit at once.
template <bool NORM, int NT, bool GRAD>
__global__ __launch_bounds__(WG, (NT >= 4 ? 2 : (NT == 3 ? 3 : 4))) void sweep_x64_fused_kernel(SweepDev d, FusedDev f, GradFusedDev gf) {
  // fix-up mode: all 64 flags in one load (lane i of every wave reads flag i; the ballot is the same in the four waves)
  int base = blockIdx.x;
  unsigned long long todo = 1;
  int my_label = -1;
  if (f.only) {
    base = blockIdx.x * FIXUP_GRAPHS_PER_WG;
    const int mine = base + (threadIdx.x & 63);
    my_label = posterior_prefetch_label(d, f.post, base, f.n_graphs);      // (requested with the flags)
    todo = __ballot(mine < f.n_graphs && f.only[mine] != 0);
  }
  while (todo) {
    const int i = __builtin_ctzll(todo);
    todo &= todo - 1;
    sweep_x64_fused_body<NORM, NT, GRAD>(d, f, gf, base + i);
    if (todo) __syncthreads();
  }
  if (f.only && f.post.labels) posterior_of_block(d, f.post, base, f.n_graphs, my_label);
}

// The fix-up pass for SEVERAL groups of graphs in one launch (mlbp_sweep_groups_f64 behind the shared-table kernels: a
// minibatch of mixed sentence shapes used to pay one near-empty launch per shape).  groups[k]: the group's descriptions as a
// single-group fix-up launch would get them, and its first workgroup; tables are streamed (NT = 0: any number of pairwise
// factors), the gradient of a redone graph comes from gradient_x64_kernel's flagged-only mode (mlbp_grad.hip).
struct FixupGroup { SweepDev d; FusedDev f; int32_t first_block, per_wg; };      // per_wg: graphs a workgroup looks at (64; 1 for a group that is redone as a whole)
__global__ __launch_bounds__(WG, 4) void sweep_x64_fixup_groups_kernel(const FixupGroup* groups, int n_groups) {
  int lo = 0, hi = n_groups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (groups[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FixupGroup& G = groups[lo];
  const int base = ((int)blockIdx.x - G.first_block) * G.per_wg;
  const int mine = base + (threadIdx.x & 63);
  unsigned long long todo = __ballot((int)(threadIdx.x & 63) < G.per_wg && mine < G.f.n_graphs && G.f.only[mine] != 0);
  if (!todo) return;
  const SweepDev d = G.d;
  const FusedDev f = G.f;
  const GradFusedDev gf = {};
  while (todo) {
    const int i = __builtin_ctzll(todo);
    todo &= todo - 1;
    sweep_x64_fused_body<true, 0, false>(d, f, gf, base + i);
    if (todo) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Any X: messages in LDS when they fit (LDSMSG) else in place in global memory (only this
// workgroup touches its graph's messages; __syncthreads orders the accesses).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* scratch /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

template <bool NORM, bool LDSMSG>
__global__ __launch_bounds__(WG) void sweep_generic_kernel(SweepDev d) {
  extern __shared__ double lds[];
  const int X = d.X;
  double* raw = lds;            // [X]
  double* scratch = lds + X;    // [4]
  double* lmsg = lds + X + 4;   // [n_msgs][X] when LDSMSG
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (d.only && !d.only[g]) return;
  if (!tables_in_range(d, g)) return;
  double* gm = d.msgs + (size_t)g * d.n_msgs * X;
  double* msg = LDSMSG ? lmsg : gm;
  if (d.fill_uniform) {
    for (int i = t; i < d.n_msgs * X; i += WG) msg[i] = 1.0 / (double)X;
  } else if (LDSMSG) {
    for (int i = t; i < d.n_msgs * X; i += WG) lmsg[i] = gm[i];
  }
  const int32_t* ptab = d.pair_tab + (size_t)g * d.P;
  const int32_t* utab = d.unary_tab + (size_t)g * d.U;
  const double uniform = 1.0 / (double)X;
  __syncthreads();
  for (int s = 0; s < d.n_sweeps; ++s) {
    const int op0 = d.sweeps[2 * s], nop = d.sweeps[2 * s + 1];
    for (int o = op0; o < op0 + nop; ++o) {
      const int kind = d.ops[4 * o], a = d.ops[4 * o + 1], b = d.ops[4 * o + 2], c = d.ops[4 * o + 3];
      if (kind == MLBP_OP_PAIR_TM) {
        const double* T = d.pair_tables + (size_t)ptab[a] * X * X;
        const double* m = msg + (size_t)b * X;
        for (int row = wave; row < X; row += 4) {
          const double* Tr = T + (size_t)row * X;
          double acc = 0.0;
          for (int j = lane; j < X; j += 64) acc += Tr[j] * m[j];
          acc = wave_sum(acc);
          if (lane == 0) raw[row] = acc;
        }
      } else if (kind == MLBP_OP_PAIR_MT) {
        const double* T = d.pair_tables + (size_t)ptab[a] * X * X;
        const double* m = msg + (size_t)b * X;
        for (int j = t; j < X; j += WG) {
          double acc = 0.0;
#pragma unroll 8
          for (int i = 0; i < X; ++i) acc += m[i] * T[(size_t)i * X + j];
          raw[j] = acc;
        }
      } else if (kind == MLBP_OP_VAR) {
        for (int j = t; j < X; j += WG) {
          double acc = uniform;
          for (int q = 0; q < b; ++q) acc = nan_to_num(msg[(size_t)d.srcs[a + q] * X + j] * acc);
          raw[j] = acc;
        }
      } else {
        const double* u = d.unary_tables + (size_t)utab[a] * X;
        for (int j = t; j < X; j += WG) raw[j] = u[j];
      }
      __syncthreads();
      double part = 0.0;
      for (int j = t; j < X; j += WG) part += raw[j];
      const double total = NORM ? block_sum(part, scratch) : 0.0;
      double* out = msg + (size_t)c * X;
      for (int j = t; j < X; j += WG) out[j] = renorm(raw[j], total, uniform, NORM);
      __syncthreads();
    }
  }
  if (LDSMSG) {
    for (int i = t; i < d.n_msgs * X; i += WG) gm[i] = lmsg[i];
  }
}

// ------------------------------------------------------------------------------------------------
// X in {128, 256, 512}, float64: every pairwise update streams a 128 KiB .. 2 MiB table, so the
// kernel is organised around that stream.  Wave w of the workgroup owns rows i = w, w+4, ...; a row
// is read by its wave as Q = X/128 fully coalesced 16-byte-per-lane loads (lane l, piece q holds
// columns 128q + 2l, +1), four rows in flight per wave (16 KiB * Q per workgroup).
//   out = T . m   (TM): the lane's 2Q message values stay in registers; per row one dot-product
//                       partial per lane, four rows reduced together by a DPP butterfly.
//   out = m^T . T (MT): the lane accumulates its 2Q columns over the wave's rows (m_i by LDS
//                       broadcast); the four waves' accumulators meet in LDS once at the end.
// Messages stay in global memory (L2 resident; 4 KiB against a 2 MiB table at X = 512).
// Algorithmic bytes per pairwise update (X*X + 2X) * 8 -- HBM-bound.
// ------------------------------------------------------------------------------------------------
// Table element type TT: double (V = 2 columns per lane and piece) or float (V = 4, the optional f32
// table mode of SURVEY.md section 8 / BASELINE config 5: half the bytes per update; products and sums stay
// in float64, so the only difference from the f64 path is the rounding of the table entries themselves).
// PAD: 0 = X is exactly 64 V Q; 1 = any even X <= 64 V Q (pieces of a row that reach past X are masked: a pair of
// columns lies wholly inside or wholly outside the row, and every row starts 16-byte aligned); 2 = any odd X (8-byte
// loads, masked per column).  Real vocabularies (train_mp.py:591-594: X = len(en_domain)) are not powers of two.
template <typename TT, int V, int PAD> struct WidePiece;
template <> struct WidePiece<double, 2, 0> {
  __device__ static __forceinline__ void load(const double* row, int idx, int, double (&o)[2]) {
    const double2 v = reinterpret_cast<const double2*>(row)[idx];
    o[0] = v.x; o[1] = v.y;
  }
};
template <> struct WidePiece<double, 2, 1> {
  __device__ static __forceinline__ void load(const double* row, int idx, int X, double (&o)[2]) {
    o[0] = 0.0; o[1] = 0.0;
    if (2 * idx < X) {
      const double2 v = reinterpret_cast<const double2*>(row)[idx];
      o[0] = v.x; o[1] = v.y;
    }
  }
};
template <> struct WidePiece<double, 2, 2> {
  __device__ static __forceinline__ void load(const double* row, int idx, int X, double (&o)[2]) {
    o[0] = 2 * idx < X ? row[2 * idx] : 0.0;
    o[1] = 2 * idx + 1 < X ? row[2 * idx + 1] : 0.0;
  }
};
template <> struct WidePiece<float, 4, 0> {
  __device__ static __forceinline__ void load(const float* row, int idx, int, double (&o)[4]) {
    const float4 v = reinterpret_cast<const float4*>(row)[idx];
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};

template <bool NORM, int Q, typename TT, int V, int PAD>
__global__ __launch_bounds__(WG) void sweep_wide_kernel(SweepDev d) {
  constexpr int XP = 64 * V * Q;           // padded width
  const int X = PAD ? d.X : XP;
  extern __shared__ double lds[];
  double* vin = lds;             // [XP] input message of the update in flight (zero beyond X)
  double* raw = lds + XP;        // [XP] un-normalised result
  double* part = lds + 2 * XP;   // [4][XP] per-wave column accumulators (MT)
  double* scratch = part + 4 * XP;  // [4]
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (!tables_in_range(d, g)) return;
  double* gm = d.msgs + (size_t)g * d.n_msgs * X;
  const int32_t* ptab = d.pair_tab + (size_t)g * d.P;
  const int32_t* utab = d.unary_tab + (size_t)g * d.U;
  const double uniform = 1.0 / (double)X;
  const const_i32p c_ops = as_const(d.ops), c_sweeps = as_const(d.sweeps), c_srcs = as_const(d.srcs);
  for (int s = 0; s < d.n_sweeps; ++s) {
    const int op0 = c_sweeps[2 * s], nop = c_sweeps[2 * s + 1];
    for (int o = op0; o < op0 + nop; ++o) {
      const int kind = c_ops[4 * o], a = c_ops[4 * o + 1], b = c_ops[4 * o + 2], c = c_ops[4 * o + 3];
      if (kind == MLBP_OP_PAIR_TM || kind == MLBP_OP_PAIR_MT) {
        const TT* T = reinterpret_cast<const TT*>(d.pair_tables) + (size_t)ptab[a] * X * X;
        const double* m = gm + (size_t)b * X;
        for (int j = t; j < XP; j += WG) vin[j] = j < X ? m[j] : 0.0;
        __syncthreads();
        if (d.approx_k > 0) {
          // use_approx_inference: keep the K largest entries of the message (rank by value, ties by index: the rule of
          // mlbp_topk_f64), zero the rest -- sum over the index set, as au.sparse_vec_mat_dot does
          double* keep = part;
          for (int j = t; j < X; j += WG) {
            const double x = vin[j];
            int rank = 0;
            for (int i = 0; i < X; ++i) {
              const double y = vin[i];
              rank += (y > x) || (y == x && i < j);
            }
            keep[j] = rank < d.approx_k ? x : 0.0;
          }
          __syncthreads();
          for (int j = t; j < X; j += WG) vin[j] = keep[j];
          __syncthreads();
        }
        if (kind == MLBP_OP_PAIR_TM) {
          double mj[Q][V];
#pragma unroll
          for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) mj[q][e] = vin[64 * V * q + V * lane + e];
          for (int i0 = wave; i0 < X; i0 += 16) {          // rows i0, i0+4, i0+8, i0+12 of this wave
            double r[4][Q][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int row = PAD ? min(i0 + 4 * u, X - 1) : i0 + 4 * u;     // past the table: re-read the last row, result dropped
#pragma unroll
              for (int q = 0; q < Q; ++q)
                WidePiece<TT, V, PAD>::load(T + (size_t)row * X, 64 * q + lane, X, r[u][q]);
            }
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              double acc = 0.0;
#pragma unroll
              for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int e = 0; e < V; ++e) acc += r[u][q][e] * mj[q][e];
              v[u] = acc;
            }
            // 4 values x 64 lanes -> 4 row sums: transposing butterfly inside each 16-lane row (4 -> 2
            // -> 1 live values), then plain sums inside the row, then the four rows through readlane
            {
              const bool up = lane & 8;
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const double send = up ? v[k] : v[k + 2], keep = up ? v[k + 2] : v[k];
                v[k] = keep + dpp_mov<0x140>(send);
              }
            }
            {
              const bool up = lane & 4;
              const double send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
              v[0] = keep + dpp_mov<0x141>(send);
            }
            v[0] += dpp_mov<0x1B>(v[0]);
            v[0] += dpp_mov<0xB1>(v[0]);
            // lanes with equal (bit3, bit2) hold the same row u = 2*bit3 + bit2, summed over 16 lanes
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int src = ((u >> 1) << 3) | ((u & 1) << 2);
              const double tot = (read_lane(v[0], src) + read_lane(v[0], 16 + src)) +
                                 (read_lane(v[0], 32 + src) + read_lane(v[0], 48 + src));
              if (lane == 0 && (!PAD || i0 + 4 * u < X)) raw[i0 + 4 * u] = tot;
            }
          }
        } else {
          double acc[Q][V];
#pragma unroll
          for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[q][e] = 0.0;
          for (int i0 = wave; i0 < X; i0 += 16) {
            double r[4][Q][V];
            double mi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int row = PAD ? min(i0 + 4 * u, X - 1) : i0 + 4 * u;
              mi[u] = (!PAD || i0 + 4 * u < X) ? vin[i0 + 4 * u] : 0.0;
#pragma unroll
              for (int q = 0; q < Q; ++q)
                WidePiece<TT, V, PAD>::load(T + (size_t)row * X, 64 * q + lane, X, r[u][q]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[q][e] += mi[u] * r[u][q][e];
          }
#pragma unroll
          for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) part[wave * XP + 64 * V * q + V * lane + e] = acc[q][e];
          __syncthreads();
          for (int j = t; j < X; j += WG) raw[j] = (part[j] + part[XP + j]) + (part[2 * XP + j] + part[3 * XP + j]);
        }
      } else if (kind == MLBP_OP_VAR) {
        for (int j = t; j < X; j += WG) {
          double acc = uniform;
          for (int q = 0; q < b; ++q) acc = nan_to_num(gm[(size_t)c_srcs[a + q] * X + j] * acc);
          raw[j] = acc;
        }
      } else {
        const double* u = d.unary_tables + (size_t)utab[a] * X;
        for (int j = t; j < X; j += WG) raw[j] = u[j];
      }
      __syncthreads();
      double psum = 0.0;
      for (int j = t; j < X; j += WG) psum += raw[j];
      const double total = NORM ? block_sum(psum, scratch) : 0.0;
      double* out = gm + (size_t)c * X;
      for (int j = t; j < X; j += WG) out[j] = renorm(raw[j], total, uniform, NORM);
      __syncthreads();
    }
  }
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// One workgroup per graph; marginal of every variable (LBP.py:392-400).
__global__ __launch_bounds__(WG) void marginals_kernel(const double* msgs, int n_msgs, int X, int n_vars,
                                                       const int32_t* in_off, const int32_t* in_slots,
                                                       int normalize, double* out) {
  __shared__ double scratch[4];
  const int g = blockIdx.x, t = threadIdx.x;
  const double* gm = msgs + (size_t)g * n_msgs * X;
  const double uniform = 1.0 / (double)X;
  for (int v = 0; v < n_vars; ++v) {
    const int s0 = in_off[v], s1 = in_off[v + 1];
    double part = 0.0;
    double* o = out + ((size_t)g * n_vars + v) * X;
    for (int j = t; j < X; j += WG) {
      double acc = uniform;
      for (int q = s0; q < s1; ++q) acc = nan_to_num(gm[(size_t)in_slots[q] * X + j] * acc);
      o[j] = acc;
      part += acc;
    }
    if (normalize) {
      const double total = block_sum(part, scratch);
      for (int j = t; j < X; j += WG) o[j] = renorm(o[j], total, uniform, true);
    }
    __syncthreads();
  }
}

// FactorGraph.get_posterior_probs for every graph, and (when sum_out is given) their sum over the batch in a fixed
// order: lanes by DPP, the block's waves in order, then the last block to finish adds the block partials in block order.
// "Last" by an arrival counter the last block re-arms itself.  There are TWO counters, taken in turn by the launch's
// generation number, and every launch also zeroes the one it does not use: a launch that was aborted halfway leaves its counter
// dirty, the next launch (other counter) cleans it, so it cannot wedge a later one -- without the memset that used to precede
// every launch (4.6 us of every bench step).

__global__ __launch_bounds__(LP_WG) void log_posterior_kernel(const double* marg, const int32_t* labels, int B, int n_vars, int X,
                                                              double* out, double* sum_out, int32_t* status, unsigned generation) {
  __shared__ double part[LP_WG / 64];
  __shared__ bool last;
  const int g = blockIdx.x * LP_WG + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double total = 0.0;
  if (g < B) {
    for (int v = 0; v < n_vars; ++v) {
      int lab = labels[(size_t)g * n_vars + v];
      if ((unsigned)lab >= (unsigned)X) { atomicExch(status, 1); continue; }
      double lp = log(marg[((size_t)g * n_vars + v) * X + lab]);
      total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
    }
    out[g] = total;
  }
  if (!sum_out) return;
  const double ws = wave_sum(total);
  if (lane == 0) part[wave] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = part[0];
    for (int w = 1; w < LP_WG / 64; ++w) v += part[w];
    g_lp_partials[blockIdx.x] = v;
    __threadfence();
    if (blockIdx.x == 0) g_lp_done[(generation + 1) & 1] = 0;
    last = atomicAdd(&g_lp_done[generation & 1], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double acc = 0.0;                                  // block-strided partial sums per thread, then lanes, then waves: fixed order
  for (unsigned q = threadIdx.x; q < gridDim.x; q += LP_WG) acc += __builtin_nontemporal_load(&g_lp_partials[q]);
  const double fs = wave_sum(acc);
  if (lane == 0) part[wave] = fs;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = part[0];
    for (int w = 1; w < LP_WG / 64; ++w) v += part[w];
    *sum_out = v;
    g_lp_done[generation & 1] = 0;                             // ready for the next launch of this parity (a replayed capture keeps its generation)
  }
}

// The same per-graph sum for several groups of graphs (each its own variable count) in one launch: thread i finds its group
// by binary search over the groups' first indices.
__global__ __launch_bounds__(LP_WG) void log_posterior_groups_kernel(const mlbp_posterior_group* groups, int n_groups, long long n_total, int X,
                                                                     double* out, int32_t* status) {
  const long long i = (long long)blockIdx.x * LP_WG + threadIdx.x;
  if (i >= n_total) return;
  int lo = 0, hi = n_groups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (groups[mid].start <= i) lo = mid; else hi = mid - 1;
  }
  const mlbp_posterior_group g = groups[lo];
  const long long b = i - g.start;
  if (b >= g.B) return;
  double total = 0.0;
  for (int v = 0; v < g.n_vars; ++v) {
    const int lab = g.labels[(size_t)b * g.n_vars + v];
    if ((unsigned)lab >= (unsigned)X) { atomicExch(status, 1); continue; }
    const double lp = log(g.marginals[((size_t)b * g.n_vars + v) * X + lab]);
    total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
  }
  out[i] = total;
}

// Fused program form (see sweep_x64_fused_kernel).  Input: the validated 4-word op list.
using mlbp::FusedProgram;

void build_fused_program(const int32_t* ops_in, const int32_t* srcs, const int32_t* sweeps, int n_sweeps, int n_msgs,
                         FusedProgram& out) {
  // 0. inside each sweep, sink every variable->factor update down to just before the pairwise update that
  //    consumes it when nothing in between writes one of its inputs or touches its output.  The up pass of a
  //    loopy schedule (LBP.py:227-233) emits "X7->F17, X4->F14, F17->X1, F14->X1": neither pair is adjacent, so
  //    without this no fusion happens.  Updates keep their inputs, hence their values; only the order of
  //    independent updates changes.
  int n_total = 0;
  for (int s = 0; s < n_sweeps; ++s) n_total = std::max(n_total, sweeps[2 * s] + sweeps[2 * s + 1]);
  std::vector<int32_t> ops_v(ops_in, ops_in + 4 * (size_t)n_total);
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    int32_t* q = ops_v.data() + 4 * (size_t)first;
    auto is_pair = [&](int i) { return q[4 * i] == MLBP_OP_PAIR_TM || q[4 * i] == MLBP_OP_PAIR_MT; };
    for (int i = 0; i < cnt; ++i) {
      if (q[4 * i] != MLBP_OP_VAR) continue;
      const int a = q[4 * i + 1], b = q[4 * i + 2], c = q[4 * i + 3];
      int j = i + 1;
      bool legal = true;
      for (; j < cnt && legal; ++j) {
        if (is_pair(j) && q[4 * j + 2] == c) break;                                  // the consumer
        const int w = q[4 * j + 3];
        if (w == c) legal = false;
        for (int k = a; k < a + b && legal; ++k) if (srcs[k] == w) legal = false;
        if (q[4 * j] == MLBP_OP_VAR)
          for (int k = q[4 * j + 1]; k < q[4 * j + 1] + q[4 * j + 2] && legal; ++k) if (srcs[k] == c) legal = false;
      }
      if (!legal || j >= cnt || j == i + 1) continue;
      const int32_t v[4] = {q[4 * i], a, b, c};
      for (int k = i; k < j - 1; ++k)
        for (int e = 0; e < 4; ++e) q[4 * k + e] = q[4 * (k + 1) + e];
      for (int e = 0; e < 4; ++e) q[4 * (j - 1) + e] = v[e];
      --i;                                                                           // the op that slid into place i
    }
  }
  const int32_t* ops = ops_v.data();
  // 1. may the unary messages be hoisted?  Every read of a unary factor's message slot must come
  //    after a UNARY op has written that slot (then the value read is always the same constant).
  std::vector<char> is_unary_dst(n_msgs, 0), written(n_msgs, 0);
  std::vector<int> unary_of(n_msgs, -1);
  bool hoistable = true;
  for (int s = 0; s < n_sweeps; ++s)
    for (int o = sweeps[2 * s]; o < sweeps[2 * s] + sweeps[2 * s + 1]; ++o)
      if (ops[4 * o] == MLBP_OP_UNARY) {
        int c = ops[4 * o + 3];
        if (is_unary_dst[c] && unary_of[c] != ops[4 * o + 1]) hoistable = false;  // two tables, one slot
        is_unary_dst[c] = 1;
        unary_of[c] = ops[4 * o + 1];
      }
  for (int s = 0; s < n_sweeps && hoistable; ++s)
    for (int o = sweeps[2 * s]; o < sweeps[2 * s] + sweeps[2 * s + 1] && hoistable; ++o) {
      const int kind = ops[4 * o], a = ops[4 * o + 1], b = ops[4 * o + 2], c = ops[4 * o + 3];
      if (kind == MLBP_OP_UNARY) {
        written[c] = 1;
      } else if (kind == MLBP_OP_VAR) {
        for (int q = a; q < a + b; ++q)
          if (is_unary_dst[srcs[q]] && !written[srcs[q]]) hoistable = false;
        if (is_unary_dst[c]) hoistable = false;
      } else {
        if ((is_unary_dst[b] && !written[b]) || is_unary_dst[c]) hoistable = false;
      }
    }
  if (hoistable)
    for (int c = 0; c < n_msgs; ++c)
      if (is_unary_dst[c]) { out.hoist.push_back(unary_of[c]); out.hoist.push_back(c); }
  // 2. per variable update: fast source list = [base slot, varying sources...] where the base is the
  //    uniform vector (ext slot 0) or the constant product of the hoisted sources (ext slot 1+k);
  //    exact source list = the original one.  Lists start on multiples of 4 words (int4 reads).
  std::vector<std::vector<int>> cprods;                      // distinct constant-source lists
  auto pad4 = [&]() { while (out.psrcs.size() % 4) out.psrcs.push_back(n_msgs); };
  auto var_lists = [&](int a, int b, int& fa, int& fn, int& ea, int& en) {
    std::vector<int> consts, vars;
    for (int q = a; q < a + b; ++q) (hoistable && is_unary_dst[srcs[q]] ? consts : vars).push_back(srcs[q]);
    int base = n_msgs;                                        // uniform
    if (!consts.empty()) {
      int k = 0;
      for (; k < (int)cprods.size(); ++k)
        if (cprods[k] == consts) break;
      if (k == (int)cprods.size()) cprods.push_back(consts);
      base = n_msgs + 1 + k;
    }
    pad4();
    fa = (int)out.psrcs.size();
    out.psrcs.push_back(base);
    for (int v : vars) out.psrcs.push_back(v);
    fn = 1 + (int)vars.size();
    pad4();
    ea = (int)out.psrcs.size();
    for (int q = a; q < a + b; ++q) out.psrcs.push_back(srcs[q]);
    en = b;
  };
  // 3. fuse "variable -> factor" into the pairwise update it feeds; drop hoisted unary ops.
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    const int f0 = (int)out.fops.size() / 8;
    for (int o = first; o < first + cnt; ++o) {
      const int kind = ops[4 * o], a = ops[4 * o + 1], b = ops[4 * o + 2], c = ops[4 * o + 3];
      if (kind == MLBP_OP_UNARY) {
        if (!hoistable) out.fops.insert(out.fops.end(), {FOP_UNARY, a, 0, c, 0, 0, 0, 0});
      } else if (kind == MLBP_OP_VAR) {
        int fa, fn, ea, en;
        var_lists(a, b, fa, fn, ea, en);
        const bool next_is_pair = o + 1 < first + cnt &&
                                  (ops[4 * (o + 1)] == MLBP_OP_PAIR_TM || ops[4 * (o + 1)] == MLBP_OP_PAIR_MT) &&
                                  ops[4 * (o + 1) + 2] == c;
        if (next_is_pair) {
          const int pk = ops[4 * (o + 1)];
          out.fops.insert(out.fops.end(), {pk == MLBP_OP_PAIR_TM ? FOP_VAR_PAIR_TM : FOP_VAR_PAIR_MT, fa, fn, c,
                                           ops[4 * (o + 1) + 1], ops[4 * (o + 1) + 3], ea, en});
          out.pairseq.push_back(ops[4 * (o + 1) + 1]);
          ++o;
        } else {
          out.fops.insert(out.fops.end(), {FOP_VAR, fa, fn, c, 0, 0, ea, en});
        }
      } else {
        out.fops.insert(out.fops.end(), {kind == MLBP_OP_PAIR_TM ? FOP_PAIR_TM : FOP_PAIR_MT, a, b, c, 0, 0, 0, 0});
        out.pairseq.push_back(a);
      }
    }
    out.fsweeps.push_back(f0);
    out.fsweeps.push_back((int)out.fops.size() / 8 - f0);
  }
  // 3b. drop lone variable->factor updates whose result is overwritten before anything reads it (the last two of a
  //     sweep when the next sweep's root differs: the new schedule recomputes those messages first).  Backward
  //     liveness over the whole call; every slot is live at the end (the messages are an output).
  {
    const int n = (int)out.fops.size() / 8;
    std::vector<char> live(n_msgs + 1 + (int)cprods.size(), 1), dead(n, 0);
    for (int i = n - 1; i >= 0; --i) {
      const int32_t* w = &out.fops[8 * (size_t)i];
      const int kd = w[0] & 0xFF;
      if (kd == FOP_VAR && !live[w[3]]) { dead[i] = 1; continue; }
      if (kd == FOP_UNARY) { live[w[3]] = 0; continue; }
      if (kd == FOP_PAIR_TM || kd == FOP_PAIR_MT) { live[w[3]] = 0; live[w[2]] = 1; continue; }
      live[w[3]] = 0;
      if (kd != FOP_VAR) live[w[5]] = 0;
      for (int q = 0; q < w[2]; ++q) live[out.psrcs[w[1] + q]] = 1;
      for (int q = 0; q < w[7]; ++q) live[out.psrcs[w[6] + q]] = 1;
    }
    std::vector<int32_t> kept;
    std::vector<int32_t> fs;
    for (size_t sw = 0; sw + 1 < out.fsweeps.size(); sw += 2) {
      const int f0 = (int)kept.size() / 8;
      for (int i = out.fsweeps[sw]; i < out.fsweeps[sw] + out.fsweeps[sw + 1]; ++i)
        if (!dead[i]) kept.insert(kept.end(), out.fops.begin() + 8 * (size_t)i, out.fops.begin() + 8 * (size_t)i + 8);
      fs.push_back(f0); fs.push_back((int)kept.size() / 8 - f0);
    }
    out.fops.swap(kept);
    out.fsweeps.swap(fs);
  }
  // 3c. bundles
  for (size_t sw = 0; sw + 1 < out.fsweeps.size(); sw += 2) {
    const int f0 = out.fsweeps[sw];
    // bundle adjacent pairwise updates that touch disjoint message slots
    {
      const int f1 = f0 + out.fsweeps[sw + 1];
      auto is_pair = [&](int i) { int kd = out.fops[8 * i] & 0xFF; return kd == FOP_PAIR_TM || kd == FOP_PAIR_MT || kd == FOP_VAR_PAIR_TM || kd == FOP_VAR_PAIR_MT; };
      auto sets = [&](int i, std::vector<int>& rd, std::vector<int>& wr) {
        const int32_t* w = &out.fops[8 * i];
        const int kd = w[0] & 0xFF;
        rd.clear(); wr.clear();
        if (kd == FOP_PAIR_TM || kd == FOP_PAIR_MT) { rd.push_back(w[2]); wr.push_back(w[3]); }
        else { for (int q = 0; q < w[7]; ++q) rd.push_back(out.psrcs[w[6] + q]); wr.push_back(w[3]); wr.push_back(w[5]); }
      };
      auto meets = [](const std::vector<int>& x, const std::vector<int>& y) {
        for (int u : x) for (int v : y) if (u == v) return true;
        return false;
      };
      std::vector<int> ra, wa, rb, wb;
      for (int i = f0; i + 1 < f1; ++i) {
        if (!is_pair(i) || !is_pair(i + 1)) continue;
        sets(i, ra, wa); sets(i + 1, rb, wb);
        if (meets(wa, rb) || meets(wb, ra) || meets(wa, wb)) continue;
        out.fops[8 * i] |= FOP_BUNDLED;
        ++i;                                       // bundles hold two updates
      }
    }
  }
  out.pairseq.push_back(-1);
  {
    std::vector<char> w(n_msgs, 0);
    for (size_t i = 0; i < out.fops.size(); i += 8) {
      const int kind = out.fops[i] & 0xFF;
      if (kind == FOP_UNARY) { out.has_unary_fops = true; continue; }
      w[out.fops[i + 3]] = 1;                                            // VAR dst / standalone PAIR dst
      if (kind == FOP_VAR_PAIR_TM || kind == FOP_VAR_PAIR_MT) w[out.fops[i + 5]] = 1;
    }
    for (int c = 0; c < n_msgs; ++c)
      if (w[c]) out.written.push_back(c);
  }
  for (int q = 0; q < 8; ++q) out.psrcs.push_back(n_msgs);   // tail padding for the int4 reads
  pad4();
  out.n_cprod = (int)cprods.size();
  for (auto& l : cprods) {
    out.cpw.push_back((int)l.size());
    for (int v : l) out.cpw.push_back(v);
  }
}

// mlbp_set_sweep_variant: 1 = the fast kernels with the exact kernel as fix-up (default), 3 = the exact / per-graph
// kernels on every graph (the tests' reference on the same inputs).
int g_sweep_variant = 1;
thread_local int g_last_kernel = -1;       // mlbp_last_sweep_kernel()
thread_local int g_last_fused_gradient = 0; // mlbp_last_sweep_fused_gradient()
}  // namespace

namespace mlbp {
bool exact_kernel_fuses_gradient(const mlbp_program* prog, const mlbp_sweep_args* a) {
  const mlbp_gradient_args* ga = a->gradient;
  if (!ga || a->X != 64 || !a->normalize_messages) return false;
  if (ga->B != a->B || ga->X != a->X || ga->P != prog->P || ga->U != prog->U || ga->n_msgs != prog->n_msgs || ga->msgs != a->msgs) return false;
  // (P = 0: a single predicted word, unary terms only -- fused for small batches, where the second launch is what costs: at 8192
  // graphs the per-graph gradient kernel behind the sweeps is the faster pair, 0.106 against 0.131 ms per trainer step)
  if (prog->P == 0 && a->B > 1024) return false;
  return ga->F_ee == 3 && ga->F_ed == 6 && prog->P >= 0 && prog->P <= 3 && prog->n_hoist == prog->U && prog->U <= WG &&
         ga->phi_en_en_t && ga->phi_en_en_w1_t && ga->phi_en_de_t && !(ga->flags & MLBP_GRADIENT_APPROX_BELIEFS);
}
}  // namespace mlbp

namespace {
thread_local bool g_lean_predone = false;  // set by mlbp_sweep_groups_f64 around the per-group fix-up calls
thread_local bool g_shared_predone = false;   // the same when the shared-table kernels ran the groups
int sweep_variant() { return g_sweep_variant; }

// hipFuncSetAttribute is a slow host call (~0.1 ms); remember the largest dynamic-LDS size already
// granted per kernel and only call again when a launch needs more.
int ensure_dynamic_lds(const void* fn, size_t bytes) {
  static std::vector<std::pair<const void*, size_t>> granted;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& g : granted)
    if (g.first == fn) {
      if (g.second >= bytes) return MLBP_OK;
      HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      g.second = bytes;
      return MLBP_OK;
    }
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  granted.push_back({fn, bytes});
  return MLBP_OK;
}

int check_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MLBP_ENODEVICE, "no HIP device visible: libmlbp.so has no CPU fallback");
  }
  return MLBP_OK;
}

// status words for the kernels that have no program attached
int32_t* g_status = nullptr;
int global_status(int32_t** out) {
  if (!g_status) {
    HIP_TRY(hipMalloc(&g_status, sizeof(int32_t)));
    HIP_TRY(hipMemset(g_status, 0, sizeof(int32_t)));
  }
  *out = g_status;
  return MLBP_OK;
}

}  // namespace

extern "C" {

int mlbp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// Range checks shared by mlbp_program_create and mlbp_program_plan; every later routine indexes freely.
static int validate_program(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs, const int32_t* sweeps,
                            int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U, int* max_srcs_out) {
  if (!ops || !sweeps || n_ops <= 0 || n_sweeps <= 0 || n_msgs <= 0 || P < 0 || U < 0 || n_srcs < 0 ||
      (n_srcs > 0 && !srcs))
    return fail(MLBP_EINVAL, "program: bad sizes or NULL arrays");
  int max_srcs = 0;
  for (int o = 0; o < n_ops; ++o) {
    const int kind = ops[4 * o], a = ops[4 * o + 1], b = ops[4 * o + 2], c = ops[4 * o + 3];
    if (c < 0 || c >= n_msgs) return fail(MLBP_EINVAL, "op %d: destination slot %d out of [0,%d)", o, c, n_msgs);
    switch (kind) {
      case MLBP_OP_UNARY:
        if (a < 0 || a >= U) return fail(MLBP_EINVAL, "op %d: unary slot %d out of [0,%d)", o, a, U);
        break;
      case MLBP_OP_PAIR_TM:
      case MLBP_OP_PAIR_MT:
        if (a < 0 || a >= P) return fail(MLBP_EINVAL, "op %d: pair slot %d out of [0,%d)", o, a, P);
        if (b < 0 || b >= n_msgs) return fail(MLBP_EINVAL, "op %d: source slot %d out of range", o, b);
        if (b == c) return fail(MLBP_EINVAL, "op %d: source and destination slot coincide", o);
        break;
      case MLBP_OP_VAR:
        if (a < 0 || b < 0 || (int64_t)a + b > n_srcs) return fail(MLBP_EINVAL, "op %d: srcs range [%d,%d) out of [0,%d)", o, a, a + b, n_srcs);
        for (int q = a; q < a + b; ++q)
          if (srcs[q] < 0 || srcs[q] >= n_msgs) return fail(MLBP_EINVAL, "op %d: source slot %d out of range", o, srcs[q]);
        if (b > max_srcs) max_srcs = b;
        break;
      default:
        return fail(MLBP_EINVAL, "op %d: unknown kind %d", o, kind);
    }
  }
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    if (first < 0 || cnt < 0 || (int64_t)first + cnt > n_ops)
      return fail(MLBP_EINVAL, "sweep %d: op range [%d,%d) out of [0,%d)", s, first, first + cnt, n_ops);
  }
  if (max_srcs_out) *max_srcs_out = max_srcs;
  return MLBP_OK;
}

/* Host only: what the program rewrites make of an op list (no device needed). */
int mlbp_program_plan(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs, const int32_t* sweeps,
                      int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U, int32_t* out8) {
  if (!out8) return fail(MLBP_EINVAL, "mlbp_program_plan: out8 is NULL");
  if (int e = validate_program(ops, n_ops, srcs, n_srcs, sweeps, n_sweeps, n_msgs, P, U, nullptr)) return e;
  FusedProgram fp;
  build_fused_program(ops, srcs, sweeps, n_sweeps, n_msgs, fp);
  int lone = 0, fused = 0, bundled = 0;
  for (size_t i = 0; i < fp.fops.size(); i += 8) {
    const int k = fp.fops[i] & 0xFF;
    lone += k == FOP_VAR;
    fused += k == FOP_VAR_PAIR_TM || k == FOP_VAR_PAIR_MT;
    bundled += (fp.fops[i] & FOP_BUNDLED) != 0;
  }
  mlbp::SharedProgram sp;
  mlbp::build_shared_program(fp, n_msgs, P, U, sp);
  out8[0] = (int)fp.fops.size() / 8; out8[1] = lone; out8[2] = fused; out8[3] = bundled;
  out8[4] = (sp.ok ? 1 : 0) | (sp.ok && sp.pf_ok ? 2 : 0) | (sp.ok && sp.pf_ok && sp.vf_direct ? 4 : 0) | (sp.ok && sp.p3_ok ? 8 : 0); out8[5] = sp.n_live; out8[6] = sp.n_ops; out8[7] = sp.n_live * (64 * 16 + 64) * 8;
  return MLBP_OK;
}

// MLBP_SWEEP_SKIP_UNCHANGED: the op list with every update dropped whose inputs -- and therefore whose result, bit for
// bit -- are what they were when the destination slot was last computed.  Value numbering over the whole call: a slot's
// value is named by (kind, table / factor, names of the source values); what the call starts from is opaque.  A root
// sequence re-walks messages that the previous sweep left final (the whole of a tree after its first sweep; the part of a
// loopy graph upstream of the first changed message), LBP.py:223-233 recomputes them, this list does not.  Returns the
// number of updates dropped; sweeps_out are ranges into ops_out (no sharing between equal roots any more).
static int drop_unchanged_updates(const int32_t* ops, const int32_t* srcs, const int32_t* sweeps, int n_sweeps, int n_msgs,
                                  std::vector<int32_t>& ops_out, std::vector<int32_t>& sweeps_out) {
  std::map<std::vector<int64_t>, int64_t> names;
  std::vector<int64_t> val(n_msgs);
  for (int c = 0; c < n_msgs; ++c) val[c] = -(int64_t)c - 1;
  int dropped = 0;
  std::vector<int64_t> key;
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    const int start = (int)ops_out.size() / 4;
    for (int o = first; o < first + cnt; ++o) {
      const int kind = ops[4 * o], a = ops[4 * o + 1], b = ops[4 * o + 2], c = ops[4 * o + 3];
      key.clear();
      key.push_back(kind);
      if (kind == MLBP_OP_VAR) {
        for (int q = a; q < a + b; ++q) key.push_back(val[srcs[q]]);
      } else if (kind == MLBP_OP_UNARY) {
        key.push_back(a);
      } else {
        key.push_back(a);
        key.push_back(val[b]);
      }
      auto it = names.find(key);
      const int64_t name = it != names.end() ? it->second : (int64_t)names.size();
      if (it == names.end()) names.emplace(key, name);
      if (val[c] == name) { ++dropped; continue; }
      val[c] = name;
      ops_out.insert(ops_out.end(), ops + 4 * o, ops + 4 * o + 4);
    }
    sweeps_out.push_back(start);
    sweeps_out.push_back((int)ops_out.size() / 4 - start);
  }
  return dropped;
}

static int create_program(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                          const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                          mlbp_program** out, bool with_pruned);

int mlbp_program_create(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                        const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                        mlbp_program** out) {
  return create_program(ops, n_ops, srcs, n_srcs, sweeps, n_sweeps, n_msgs, P, U, out, true);
}

static int create_program(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                          const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                          mlbp_program** out, bool with_pruned) {
  if (!out) return fail(MLBP_EINVAL, "out is NULL");
  *out = nullptr;
  int max_srcs = 0;
  if (int e = validate_program(ops, n_ops, srcs, n_srcs, sweeps, n_sweeps, n_msgs, P, U, &max_srcs)) return e;
  std::vector<int32_t> pairseq;
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    for (int o = first; o < first + cnt; ++o)
      if (ops[4 * o] == MLBP_OP_PAIR_TM || ops[4 * o] == MLBP_OP_PAIR_MT) pairseq.push_back(ops[4 * o + 1]);
  }
  if (int e = check_device()) return e;
  mlbp_program* p = new mlbp_program();
  p->n_ops = n_ops; p->n_srcs = n_srcs; p->n_sweeps = n_sweeps; p->n_msgs = n_msgs; p->P = P; p->U = U;
  p->n_pairseq = (int)pairseq.size();
  p->max_srcs = max_srcs;
  pairseq.push_back(-1);
  p->d_ops = p->d_srcs = p->d_sweeps = p->d_pairseq = p->d_status = nullptr;
  p->d_fops = p->d_fsweeps = p->d_fpairseq = nullptr;
  (void)hipGetDevice(&p->device);
  auto up = [&](int32_t** dst, const int32_t* src, size_t n) -> hipError_t {
    hipError_t e = hipMalloc(dst, (n ? n : 1) * sizeof(int32_t));
    if (e != hipSuccess) return e;
    return n ? hipMemcpy(*dst, src, n * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t e = up(&p->d_ops, ops, (size_t)n_ops * 4);
  std::vector<int32_t> srcs_padded(srcs ? srcs : nullptr, srcs ? srcs + n_srcs : nullptr);
  srcs_padded.resize(((size_t)n_srcs + 16 + 3) / 4 * 4, 0);   // the fused kernel reads sources 16 at a time
  if (e == hipSuccess) e = up(&p->d_srcs, srcs_padded.data(), srcs_padded.size());
  if (e == hipSuccess) e = up(&p->d_sweeps, sweeps, (size_t)n_sweeps * 2);
  if (e == hipSuccess) e = up(&p->d_pairseq, pairseq.data(), pairseq.size());
  int32_t zero = 0;
  if (e == hipSuccess) e = up(&p->d_status, &zero, 1);
  FusedProgram fp;
  build_fused_program(ops, srcs, sweeps, n_sweeps, n_msgs, fp);
  p->n_fops = (int)fp.fops.size() / 8;
  p->n_hoist = (int)fp.hoist.size() / 2;
  p->n_psrcs = (int)fp.psrcs.size();
  p->n_cprod = fp.n_cprod;
  p->n_cpw = (int)fp.cpw.size();
  std::vector<int32_t> image(fp.fops);
  image.insert(image.end(), fp.psrcs.begin(), fp.psrcs.end());
  image.insert(image.end(), fp.hoist.begin(), fp.hoist.end());
  image.insert(image.end(), fp.cpw.begin(), fp.cpw.end());
  image.insert(image.end(), fp.written.begin(), fp.written.end());
  p->n_written = (int)fp.written.size();
  p->sf_ok = !fp.has_unary_fops;
  p->d_bail = nullptr;
  p->bail_cap = 0;
  p->d_readout = nullptr;
  p->n_vars = 0;
  p->n_readout = 0;
  p->d_simage = p->d_sreadout = nullptr;
  p->d_tfrag = nullptr;
  p->h_ops.assign(ops, ops + 4 * (size_t)n_ops);
  p->h_sweeps.assign(sweeps, sweeps + 2 * (size_t)n_sweeps);
  p->n_sreadout = 0;
  mlbp::build_shared_program(fp, n_msgs, P, U, p->shared);
  if (p->shared.ok && e == hipSuccess) e = up(&p->d_simage, p->shared.image.data(), p->shared.image.size());
  p->fused = fp;
  mlbp::build_lean_program(fp, n_msgs, p->lean);
  if (p->lean.ok && e == hipSuccess) e = up(&p->d_limage, p->lean.image.data(), p->lean.image.size());
  if (e == hipSuccess) e = up(&p->d_fops, image.data(), image.size());
  if (e == hipSuccess) e = up(&p->d_fsweeps, fp.fsweeps.data(), fp.fsweeps.size());
  if (e == hipSuccess) e = up(&p->d_fpairseq, fp.pairseq.data(), fp.pairseq.size());
  if (e != hipSuccess) {
    mlbp_program_destroy(p);
    return fail(MLBP_EHIP, "mlbp_program_create: device upload failed: %s", hipGetErrorString(e));
  }
  if (with_pruned) {
    std::vector<int32_t> ops2, sweeps2;
    p->n_dropped = drop_unchanged_updates(ops, srcs, sweeps, n_sweeps, n_msgs, ops2, sweeps2);
    if (p->n_dropped > 0) {
      if (int rc = create_program(ops2.data(), (int)ops2.size() / 4, srcs, n_srcs, sweeps2.data(), n_sweeps, n_msgs, P, U, &p->pruned, false)) {
        mlbp_program_destroy(p);
        return rc;
      }
      p->pruned->is_twin = true;
    }
  }
  *out = p;
  return MLBP_OK;
}

// the program a call runs: the pruned twin under MLBP_SWEEP_SKIP_UNCHANGED (a fused gradient reads the final messages and
// the resident tables only, so it follows either list)
static const mlbp_program* effective_program(const mlbp_program* p, const mlbp_sweep_args* a) {
  const bool pruned = p && a && (a->flags & MLBP_SWEEP_SKIP_UNCHANGED) && p->pruned;
  if (p && a && !p->is_twin) const_cast<mlbp_program*>(p)->last_was_pruned = pruned;
  return pruned ? p->pruned : p;
}

int mlbp_program_destroy(mlbp_program* p) {
  if (!p) return MLBP_OK;
  if (p->pruned) (void)mlbp_program_destroy(p->pruned);
  (void)hipFree(p->d_ops); (void)hipFree(p->d_srcs); (void)hipFree(p->d_sweeps);
  (void)hipFree(p->d_pairseq); (void)hipFree(p->d_status);
  (void)hipFree(p->d_fops); (void)hipFree(p->d_fsweeps); (void)hipFree(p->d_fpairseq); (void)hipFree(p->d_bail); (void)hipFree(p->d_readout);
  (void)hipFree(p->d_limage); (void)hipFree(p->d_lreadout); (void)hipFree(p->d_simage); (void)hipFree(p->d_sreadout); (void)hipFree(p->d_tfrag); (void)hipFree(p->d_spill); (void)hipFree(p->d_ptiles); (void)hipFree(p->d_wfrag); (void)hipFree(p->d_header);
  if (p->side_stream) (void)hipStreamDestroy((hipStream_t)p->side_stream);
  if (p->ev_fork) (void)hipEventDestroy((hipEvent_t)p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy((hipEvent_t)p->ev_join);
  (void)hipFree(p->d_gfrag); (void)hipFree(p->d_gxbuf); (void)hipFree(p->d_gwork);
  for (void* q : p->retired) (void)hipFree(q);
  mlbp::group_tables_free(p->gtables); mlbp::group_tables_free(p->stables);
  delete p;
  return MLBP_OK;
}

// mlbp_gradient_f64 behind the sweeps of a call: a gradient that was given no workspace gets scratch the PROGRAM owns
// (grown by new blocks only), so that the call is safe on its own stream and inside a captured graph
static int gradient_behind_sweeps(const mlbp_program* prog, const mlbp_gradient_args* ga, void* stream) {
  mlbp_gradient_args g = *ga;
  if (!g.workspace) {
    const int64_t need = mlbp_gradient_workspace_bytes(&g);
    if (need > 0) {
      mlbp_program* mp = const_cast<mlbp_program*>(prog);
      if (int e = mlbp::program_grow(mp, &mp->d_gwork, &mp->gwork_cap, (size_t)need)) return e;
      g.workspace = mp->d_gwork;
      g.workspace_bytes = (int64_t)mp->gwork_cap;
    }
  }
  return mlbp_gradient_f64(&g, stream);
}

static unsigned next_lp_generation() {
  static std::atomic<unsigned> generation{0};
  return generation.fetch_add(1u) + 1u;
}

// mlbp_sweep_args.posterior by its own launch (no fix-up pass took it)
static int posterior_behind_sweeps(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream) {
  const mlbp_posterior_args* pa = a->posterior;
  if (!pa) return MLBP_OK;
  return mlbp_log_posterior_sum_f64(a->marginals, pa->labels, a->B, prog->n_vars, a->X, pa->out, pa->sum_out, stream);
}

int mlbp_sweep_f64(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream) {
  g_last_fused_gradient = 0;
  if (!prog || !a) return fail(MLBP_EINVAL, "mlbp_sweep_f64: NULL program or args");
  prog = effective_program(prog, a);
  if (a->B <= 0 || a->X <= 0) return fail(MLBP_EINVAL, "mlbp_sweep_f64: B=%d X=%d", a->B, a->X);
  if (!a->msgs) return fail(MLBP_EINVAL, "mlbp_sweep_f64: msgs is NULL");
  if (prog->P > 0 && (!((a->flags & MLBP_SWEEP_PAIR_TABLES_F32) ? (const void*)a->pair_tables_f32 : (const void*)a->pair_tables) ||
                      !a->pair_tab || a->n_pair_tables <= 0))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: program has %d pairwise factors but no pair tables", prog->P);
  if (prog->U > 0 && (!a->unary_tables || !a->unary_tab || a->n_unary_tables <= 0))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: program has %d unary factors but no unary tables", prog->U);
  if (a->X > 4096) return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: X=%d > 4096", a->X);
  const bool approx = (a->flags & MLBP_SWEEP_APPROX_INFERENCE) != 0;
  if (approx && a->X < MLBP_APPROX_K)      // np.argpartition(-vec, K - 1) in the reference: "kth(=99) out of bounds"
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: approximate inference keeps the %d largest entries; kth(=%d) out of bounds (%d)",
                MLBP_APPROX_K, MLBP_APPROX_K - 1, a->X);
  if (approx && !(a->X > 64 && a->X <= 1024 && a->normalize_messages && !(a->flags & MLBP_SWEEP_PAIR_TABLES_F32)))
    return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: batched approximate inference needs 100 <= X <= 1024, normalised messages, float64 tables");
  if (int e = check_device()) return e;
  {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != prog->device)
      return fail(MLBP_EINVAL, "mlbp_sweep_f64: the program was created on device %d, the calling thread's current device is %d",
                  prog->device, dev);
  }
  SweepDev d;
  d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab;
  d.unary_tables = a->unary_tables; d.unary_tab = a->unary_tab;
  d.msgs = a->msgs;
  d.ops = prog->d_ops; d.srcs = prog->d_srcs; d.sweeps = prog->d_sweeps; d.pairseq = prog->d_pairseq;
  d.status = prog->d_status;
  d.n_sweeps = prog->n_sweeps; d.n_msgs = prog->n_msgs; d.P = prog->P; d.U = prog->U; d.X = a->X;
  d.n_pair_tables = a->n_pair_tables; d.n_unary_tables = a->n_unary_tables;
  d.marginals = nullptr; d.readout = prog->d_readout; d.n_vars = prog->n_vars;
  d.only = nullptr; d.fill_uniform = 0; d.approx_k = 0;
  if (a->marginals && !prog->d_readout)
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: marginals requested but mlbp_program_set_readout was not called");
  if (a->posterior && (!a->marginals || !a->posterior->labels || !a->posterior->out))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: posterior needs marginals, labels and an output array");
  hipStream_t st = (hipStream_t)stream;
  const bool norm = a->normalize_messages != 0;
  const size_t LDS_MAX = 160 * 1024;
  const int variant = sweep_variant();
  if (a->X == 64) {
    const int n_ext = 1 + prog->n_cprod;
    const size_t img_words = (size_t)prog->n_fops * 8 + prog->n_psrcs + 2 * prog->n_hoist + prog->n_cpw + prog->n_written;
    // the exact kernel keeps the tables in registers when the graph has at most 3 of them, else it streams them
    const int nt = (prog->P >= 1 && prog->P <= 3) ? prog->P : 0;
    const size_t lds = ((size_t)(prog->n_msgs + n_ext) * 64 + 64 + 512) * sizeof(double) +
                       (img_words + prog->P + 6 * prog->U + 8) * sizeof(int32_t);
    if (lds <= LDS_MAX) {
      mlbp_program* mp = const_cast<mlbp_program*>(prog);
      const bool fast = variant == 1;               // variant 3: the exact kernel on every graph
      if (norm) d.marginals = a->marginals;         // read-out fused into the kernels' epilogue
      // the gradient runs as the sweep kernels' epilogue when the tables are on chip in BOTH the fast and the exact kernel
      GradFusedDev gf = {};
      const mlbp_gradient_args* ga = a->gradient;
      bool grad_fused = false;
      if (ga) {
        if (ga->B != a->B || ga->X != a->X || ga->P != prog->P || ga->U != prog->U || ga->n_msgs != prog->n_msgs || ga->msgs != a->msgs)
          return fail(MLBP_EINVAL, "mlbp_sweep_f64: gradient arguments do not describe the same batch");
        grad_fused = mlbp::exact_kernel_fuses_gradient(prog, a) && !g_lean_predone;
        if (grad_fused) {
          gf.pair_c_slot = ga->pair_c_slot; gf.pair_r_slot = ga->pair_r_slot; gf.pair_phi = ga->pair_phi; gf.pair_label = ga->pair_label;
          gf.unary_kind = ga->unary_kind; gf.unary_obs = ga->unary_obs; gf.unary_label = ga->unary_label;
          gf.phi_en_en = ga->phi_en_en; gf.phi_en_en_w1 = ga->phi_en_en_w1;
          gf.phi_en_en_t = ga->phi_en_en_t; gf.phi_en_en_w1_t = ga->phi_en_en_w1_t; gf.phi_en_de_t = ga->phi_en_de_t;
          gf.grad_en_en = ga->grad_en_en; gf.grad_en_de = ga->grad_en_de; gf.Vde = ga->Vde; gf.enabled = 1;
        }
      }
      bool shared_done = false;                // shared-table batches: 16 graphs per workgroup on the matrix cores
      if (g_shared_predone) shared_done = true;  // mlbp_sweep_groups_f64 ran the shared-table kernels for this group already
      else if (fast && !g_lean_predone)
        if (int e = mlbp::launch_shared_sweep(prog, a, stream, &shared_done)) return e;
      // (the shared-table kernel ran the gradient as its epilogue: the fix-up pass keeps its own for the graphs it redoes -- or,
      // with more than three pairwise factors, where the exact kernel streams its tables and has no epilogue, the per-graph
      // gradient kernel follows on the flagged graphs only)
      bool grad_flagged_fixup = false;
      if (shared_done) {
        const bool sg = ga && mlbp::shared_gradient_fused(prog, a);
        if (sg && !grad_fused) grad_flagged_fixup = true;
        if (!sg || !grad_fused) { grad_fused = false; gf = GradFusedDev{}; }
      }
      bool lean_done = false;                  // default path: the lean scale-free kernel (mlbp_lean.hip), up to 8 resident tables
      if (g_lean_predone) lean_done = true;     // mlbp_sweep_groups_f64 ran the lean kernel for this group already
      else if (fast && norm && prog->sf_ok && prog->P >= 1 && prog->P <= 8 && !shared_done)
        if (int e = mlbp::launch_lean_sweep(prog, a, grad_fused ? &gf : nullptr, stream, &lean_done)) return e;
      g_last_kernel = shared_done ? MLBP_KERNEL_SHARED_MFMA : (lean_done ? MLBP_KERNEL_LEAN : MLBP_KERNEL_EXACT);
      g_last_fused_gradient = (grad_fused || grad_flagged_fixup) ? 1 : 0;
      FusedDev f;
      f.post = PosteriorDev{};
      f.only = (shared_done || lean_done) ? mp->d_bail : nullptr;     // after a fast pass: flagged graphs only
      // get_posterior_probs of the call: taken by the fix-up pass (it visits every graph's flag anyway); without a fix-up pass
      // -- the exact kernel on every graph -- by its own launch below
      const mlbp_posterior_args* pa = a->posterior;
      const bool post_in_fixup = pa && f.only && norm && (a->B + FIXUP_GRAPHS_PER_WG - 1) / FIXUP_GRAPHS_PER_WG <= LP_MAX_BLOCKS;
      if (post_in_fixup) {
        f.post.labels = pa->labels; f.post.out = pa->out; f.post.sum_out = pa->sum_out;
        f.post.generation = pa->sum_out ? next_lp_generation() : 0u;
      }
      f.image = prog->d_fops; f.fsweeps = prog->d_fsweeps;
      f.n_fops = prog->n_fops; f.n_psrcs = prog->n_psrcs; f.n_hoist = prog->n_hoist;
      f.n_cprod = prog->n_cprod; f.n_cpw = prog->n_cpw; f.n_ext = n_ext;
      f.init = a->init_messages;
      f.n_graphs = a->B;
      d.pairseq = prog->d_fpairseq;
      void (*k)(SweepDev, FusedDev, GradFusedDev) = nullptr;
#define MLBP_PICK(N) k = grad_fused ? sweep_x64_fused_kernel<true, N, true> : (norm ? sweep_x64_fused_kernel<true, N, false> : sweep_x64_fused_kernel<false, N, false>)
      switch (nt) {
        case 1: MLBP_PICK(1); break;
        case 2: MLBP_PICK(2); break;
        case 3: MLBP_PICK(3); break;
        default: MLBP_PICK(0); break;
      }
#undef MLBP_PICK
      if (int e = ensure_dynamic_lds((const void*)k, lds)) return e;
      hipLaunchKernelGGL(k, dim3(f.only ? (a->B + FIXUP_GRAPHS_PER_WG - 1) / FIXUP_GRAPHS_PER_WG : a->B), dim3(WG), lds, st, d, f, gf);
      HIP_TRY(hipGetLastError());
      if (ga && grad_flagged_fixup) {
        if (int e = mlbp::gradient_flagged_only(ga, mp->d_bail, stream)) return e;
      } else if (ga && !grad_fused) {
        if (int e = gradient_behind_sweeps(prog, ga, stream)) return e;
      }
      if (a->marginals && !norm)
        if (int e = mlbp_marginals_f64(a->msgs, a->B, prog->n_msgs, a->X, prog->n_vars, prog->d_readout,
                                       prog->d_readout + prog->n_vars + 1, 0, a->marginals, stream)) return e;
      if (pa && !post_in_fixup) return posterior_behind_sweeps(prog, a, stream);
      return MLBP_OK;
    }
  }
  const bool small_lean_candidate = a->X < 64 && a->X >= 2 && norm && prog->sf_ok && prog->P >= 1 && prog->P <= 4 && variant == 1 &&
                                    !a->gradient && prog->lean.ok && prog->d_limage;
  if (a->init_messages && !small_lean_candidate) {
    int e = mlbp_init_messages_f64(a->msgs, (int64_t)a->B * prog->n_msgs, a->X, stream);
    if (e) return e;
  }
  if ((a->flags & MLBP_SWEEP_SHARED_PAIR_TABLES) && a->pair_tab_host && mlbp::gemm_path_supports(a->X) && !approx &&
      prog->P >= 1 && prog->P <= 16 && variant == 1) {
    // shared tables at a large state space: every contraction is one MFMA launch over the whole batch
    if ((a->flags & MLBP_SWEEP_PAIR_TABLES_F32) && a->gradient)
      return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: no gradient with float32 pairwise tables");
    const int eg = mlbp::launch_gemm_sweep(prog, a, stream);
    if (eg != MLBP_EUNSUPPORTED) {                 // unsupported shape: the per-graph kernels below
      if (eg) return eg;
      g_last_kernel = MLBP_KERNEL_SHARED_GEMM;
      if (a->marginals)
        if (int e = mlbp_marginals_f64(a->msgs, a->B, prog->n_msgs, a->X, prog->n_vars, prog->d_readout,
                                       prog->d_readout + prog->n_vars + 1, norm ? 1 : 0, a->marginals, stream)) return e;
      if (a->gradient)
        if (int e = gradient_behind_sweeps(prog, a->gradient, stream)) return e;
      return posterior_behind_sweeps(prog, a, stream);
    }
  }
  const bool f32_tables = (a->flags & MLBP_SWEEP_PAIR_TABLES_F32) != 0;
  if (f32_tables && !(a->X == 256 || a->X == 512))
    return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: float32 pairwise tables need X = 256 or 512 (got %d)", a->X);
  if (f32_tables && a->gradient)
    return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: no gradient with float32 pairwise tables");
  // large state spaces: the wide kernel for X = 128 / 256 / 512 exactly, and (normalised messages, float64 tables) for
  // any X in (64, 1024] with the last pieces of each row masked
  d.approx_k = approx ? MLBP_APPROX_K : 0;
  const bool wide_exact = a->X == 128 || a->X == 256 || a->X == 512;
  const bool wide_padded = !wide_exact && !f32_tables && norm && a->X > 64 && a->X <= 1024;
  if (wide_exact || wide_padded) {
    g_last_kernel = MLBP_KERNEL_WIDE;
    void (*kw)(SweepDev) = nullptr;
    int xp = a->X;
    if (f32_tables) {
      d.pair_tables = reinterpret_cast<const double*>(a->pair_tables_f32);
      if (a->X == 256) kw = norm ? sweep_wide_kernel<true, 1, float, 4, 0> : sweep_wide_kernel<false, 1, float, 4, 0>;
      else kw = norm ? sweep_wide_kernel<true, 2, float, 4, 0> : sweep_wide_kernel<false, 2, float, 4, 0>;
    } else if (a->X == 128) kw = norm ? sweep_wide_kernel<true, 1, double, 2, 0> : sweep_wide_kernel<false, 1, double, 2, 0>;
    else if (a->X == 256) kw = norm ? sweep_wide_kernel<true, 2, double, 2, 0> : sweep_wide_kernel<false, 2, double, 2, 0>;
    else if (a->X == 512) kw = norm ? sweep_wide_kernel<true, 4, double, 2, 0> : sweep_wide_kernel<false, 4, double, 2, 0>;
    else {
      const int q = (a->X + 127) / 128;              // 128-column pieces per row
      const bool odd = (a->X & 1) != 0;
#define MLBP_WIDE_PAD(QQ) (kw = odd ? sweep_wide_kernel<true, QQ, double, 2, 2> : sweep_wide_kernel<true, QQ, double, 2, 1>, xp = 128 * QQ)
      if (q <= 1) MLBP_WIDE_PAD(1);
      else if (q <= 2) MLBP_WIDE_PAD(2);
      else if (q <= 3) MLBP_WIDE_PAD(3);
      else if (q <= 4) MLBP_WIDE_PAD(4);
      else if (q <= 6) MLBP_WIDE_PAD(6);
      else MLBP_WIDE_PAD(8);
#undef MLBP_WIDE_PAD
    }
    const size_t ldsw = ((size_t)6 * xp + 4) * sizeof(double);
    hipLaunchKernelGGL(kw, dim3(a->B), dim3(WG), ldsw, st, d);
    HIP_TRY(hipGetLastError());
    if (a->marginals)
      if (int e = mlbp_marginals_f64(a->msgs, a->B, prog->n_msgs, a->X, prog->n_vars, prog->d_readout,
                                     prog->d_readout + prog->n_vars + 1, norm ? 1 : 0, a->marginals, stream)) return e;
    if (a->gradient)
      if (int e = gradient_behind_sweeps(prog, a->gradient, stream)) return e;
    return posterior_behind_sweeps(prog, a, stream);
  }
  // small state spaces (X < 64): the lean X = 64 kernel on zero-padded vectors and tables; the graphs it flags are redone
  // by the generic kernel below in its fix-up mode
  bool lean_small = false;
  if (a->X < 64 && a->X >= 2 && norm && prog->sf_ok && prog->P >= 1 && prog->P <= 4 && variant == 1 && !a->gradient) {
    if (int e = mlbp::launch_lean_sweep(prog, a, nullptr, stream, &lean_small)) return e;
    if (lean_small) {
      d.only = const_cast<mlbp_program*>(prog)->d_bail;
      d.fill_uniform = a->init_messages;
    } else if (small_lean_candidate && a->init_messages) {       // the lean kernel declined after all: initialise here
      if (int e = mlbp_init_messages_f64(a->msgs, (int64_t)a->B * prog->n_msgs, a->X, stream)) return e;
    }
  }
  g_last_kernel = lean_small ? MLBP_KERNEL_LEAN : MLBP_KERNEL_GENERIC;
  size_t base = ((size_t)a->X + 4) * sizeof(double);
  size_t with_msgs = base + (size_t)prog->n_msgs * a->X * sizeof(double);
  if (with_msgs <= 64 * 1024) {
    auto k = norm ? sweep_generic_kernel<true, true> : sweep_generic_kernel<false, true>;
    hipLaunchKernelGGL(k, dim3(a->B), dim3(WG), with_msgs, st, d);
  } else {
    auto k = norm ? sweep_generic_kernel<true, false> : sweep_generic_kernel<false, false>;
    hipLaunchKernelGGL(k, dim3(a->B), dim3(WG), base, st, d);
  }
  HIP_TRY(hipGetLastError());
  if (a->marginals)
    if (int e = mlbp_marginals_f64(a->msgs, a->B, prog->n_msgs, a->X, prog->n_vars, prog->d_readout,
                                   prog->d_readout + prog->n_vars + 1, norm ? 1 : 0, a->marginals, stream)) return e;
  if (a->gradient)
    if (int e = gradient_behind_sweeps(prog, a->gradient, stream)) return e;
  return posterior_behind_sweeps(prog, a, stream);
}

// Behind launch_shared_groups: every group's fix-up pass in ONE launch (and, when the call carries gradients, one launch of
// the per-graph gradient kernel over the flagged graphs of all groups).  *done false: some group needs the per-group path
// (messages kept -- the unary write-back ran already, but marginals without normalisation, an LDS image too large, a gradient
// the shared-table kernel did not produce).
static int finish_shared_groups(const mlbp_program* const* progs, const mlbp_sweep_args* args, int n_groups, void* stream, bool* done) {
  *done = false;
  const size_t LDS_MAX = 160 * 1024;
  std::vector<FixupGroup> table(n_groups);
  std::vector<mlbp_gradient_args> grads;
  std::vector<const uint8_t*> grad_flags;
  size_t lds_max = 0;
  int blocks = 0;
  bool any_grad = false;
  for (int k = 0; k < n_groups; ++k) {
    const mlbp_program* prog = progs[k];
    const mlbp_sweep_args* a = &args[k];
    if (a->X != 64 || !a->normalize_messages || !a->init_messages) return MLBP_OK;
    if (a->marginals && !prog->d_readout) return MLBP_OK;
    const int n_ext = 1 + prog->n_cprod;
    const size_t img_words = (size_t)prog->n_fops * 8 + prog->n_psrcs + 2 * prog->n_hoist + prog->n_cpw + prog->n_written;
    const size_t lds = ((size_t)(prog->n_msgs + n_ext) * 64 + 64 + 512) * sizeof(double) + (img_words + prog->P + 6 * prog->U + 8) * sizeof(int32_t);
    if (lds > LDS_MAX) return MLBP_OK;
    lds_max = std::max(lds_max, lds);
    // (a group without pairwise factors -- one predicted word -- never ran the shared-table kernels: launch_shared_groups flagged
    // all of its graphs, so this pass and the flagged graphs' gradient below ARE its sweep call)
    if (a->gradient && prog->P > 0 && !mlbp::shared_gradient_fused(prog, a)) return MLBP_OK;
    mlbp_program* mp = const_cast<mlbp_program*>(prog);
    FixupGroup& G = table[k];
    memset(&G, 0, sizeof(G));
    SweepDev& d = G.d;
    d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab; d.unary_tables = a->unary_tables; d.unary_tab = a->unary_tab;
    d.msgs = a->msgs; d.ops = prog->d_ops; d.srcs = prog->d_srcs; d.sweeps = prog->d_sweeps; d.pairseq = prog->d_fpairseq;
    d.status = prog->d_status;
    d.n_sweeps = prog->n_sweeps; d.n_msgs = prog->n_msgs; d.P = prog->P; d.U = prog->U; d.X = a->X;
    d.n_pair_tables = a->n_pair_tables; d.n_unary_tables = a->n_unary_tables;
    d.marginals = a->marginals; d.readout = prog->d_readout; d.n_vars = prog->n_vars;
    FusedDev& f = G.f;
    f.only = mp->d_bail; f.image = prog->d_fops; f.fsweeps = prog->d_fsweeps;
    f.n_fops = prog->n_fops; f.n_psrcs = prog->n_psrcs; f.n_hoist = prog->n_hoist; f.n_cprod = prog->n_cprod; f.n_cpw = prog->n_cpw;
    f.n_ext = n_ext; f.init = a->init_messages; f.n_graphs = a->B;
    G.first_block = blocks;
    G.per_wg = prog->P == 0 ? 1 : FIXUP_GRAPHS_PER_WG;          // (a pairwise-free group is all flagged: one workgroup per graph, not 64 graphs in a row)
    blocks += (a->B + G.per_wg - 1) / G.per_wg;
    if (a->gradient) { any_grad = true; grads.push_back(*a->gradient); grad_flags.push_back(mp->d_bail); }
  }
  if (any_grad && (int)grads.size() != n_groups) return MLBP_OK;        // (all groups or none carry a gradient)
  // the table as 32-bit words in the first program's group-table cache (one device copy per distinct contents)
  static_assert(sizeof(FixupGroup) % 4 == 0, "");
  std::vector<int32_t> words(sizeof(FixupGroup) / 4 * (size_t)n_groups + 1);
  memcpy(words.data(), table.data(), sizeof(FixupGroup) * (size_t)n_groups);
  words.back() = 0x46495855;                                                  // (keeps this table apart from the sweep kernels' own)
  mlbp_program* owner = const_cast<mlbp_program*>(progs[0]);
  int32_t* d_table = nullptr;
  if (int e = mlbp::group_table_device(owner->stables, words, stream, &d_table)) return e;
  if (int e = ensure_dynamic_lds((const void*)sweep_x64_fixup_groups_kernel, lds_max)) return e;
  hipLaunchKernelGGL(sweep_x64_fixup_groups_kernel, dim3(blocks), dim3(WG), lds_max, (hipStream_t)stream,
                     reinterpret_cast<const FixupGroup*>(d_table), n_groups);
  HIP_TRY(hipGetLastError());
  if (any_grad)
    if (int e = mlbp::gradient_flagged_groups(grads.data(), grad_flags.data(), n_groups, owner, stream)) return e;
  g_last_kernel = MLBP_KERNEL_SHARED_MFMA;
  g_last_fused_gradient = any_grad ? 1 : 0;
  *done = true;
  return MLBP_OK;
}

int mlbp_sweep_groups_f64(const mlbp_program* const* progs, const mlbp_sweep_args* args, int32_t n_groups, void* stream) {
  if (!progs || !args || n_groups < 1) return fail(MLBP_EINVAL, "mlbp_sweep_groups_f64: bad arguments");
  bool one_launch = false;
  std::vector<const mlbp_program*> eff(n_groups);
  for (int k = 0; k < n_groups; ++k) {
    if (!progs[k]) return fail(MLBP_EINVAL, "mlbp_sweep_groups_f64: NULL program");
    eff[k] = effective_program(progs[k], &args[k]);
  }
  progs = eff.data();
  bool shared_launch = false;
  if (sweep_variant() == 1)
    if (int e = mlbp::launch_shared_groups(progs, args, n_groups, stream, &shared_launch)) return e;
  if (sweep_variant() == 1 && !shared_launch)
    if (int e = mlbp::launch_lean_groups(progs, args, n_groups, stream, &one_launch)) return e;
  // the shared-table kernels have run every group, gradient included: ONE fix-up launch for the flagged graphs of all groups and
  // one more for their gradients (a mixed minibatch used to pay both per group)
  if (shared_launch) {
    bool done = false;
    if (int e = finish_shared_groups(progs, args, n_groups, stream, &done)) return e;
    if (done) return MLBP_OK;
  }
  // the fast kernel has run every group (one_launch): what is left per group is the fix-up pass over the graphs it
  // flagged; otherwise the groups run one after the other exactly as separate calls would
  int rc = MLBP_OK;
  g_lean_predone = one_launch;
  g_shared_predone = shared_launch;
  for (int k = 0; k < n_groups && rc == MLBP_OK; ++k) rc = mlbp_sweep_f64(progs[k], &args[k], stream);
  g_lean_predone = false;
  g_shared_predone = false;
  return rc;
}

#ifdef MLBP_ABLATE
int mlbp_debug_set_ablate_mask(int mask) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ablate_mask), &mask, sizeof(mask)));
  return MLBP_OK;
}
#endif

#ifdef MLBP_STAMPS
int mlbp_debug_set_stamp_buffer(void* dev_ptr) {
  unsigned long long* p = (unsigned long long*)dev_ptr;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)));
  return MLBP_OK;
}
#endif

}  // extern "C"

namespace mlbp {
int program_grow(mlbp_program* prog, void** p, size_t* cap, size_t bytes, bool zero) {
  if (bytes <= *cap && *p) return MLBP_OK;
  void* fresh = nullptr;
  if (hipMalloc(&fresh, bytes ? bytes : 1) != hipSuccess) return fail(MLBP_EHIP, "program scratch: allocation of %zu bytes failed", bytes);
  if (zero && hipMemset(fresh, 0, bytes) != hipSuccess) { (void)hipFree(fresh); return fail(MLBP_EHIP, "program scratch: memset failed"); }
  if (*p) prog->retired.push_back(*p);     // a captured graph may still name it: freed with the program
  *p = fresh;
  *cap = bytes;
  return MLBP_OK;
}

int fallback_scratch(int purpose, size_t bytes, void** out) {
  struct Block { void* p = nullptr; size_t cap = 0; };
  static std::mutex mu;
  static std::vector<std::vector<Block>> per_device;      // [device][purpose]; replaced blocks are never freed (process lifetime)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MLBP_EHIP, "hipGetDevice failed");
  std::lock_guard<std::mutex> lock(mu);
  if ((int)per_device.size() <= dev) per_device.resize(dev + 1, std::vector<Block>(SCRATCH_PURPOSES));
  Block& b = per_device[dev][purpose];
  if (bytes > b.cap) {
    void* fresh = nullptr;
    const size_t want = bytes > 2 * b.cap ? bytes : 2 * b.cap;
    if (hipMalloc(&fresh, want) != hipSuccess) return fail(MLBP_EHIP, "scratch allocation of %zu bytes failed", want);
    b.p = fresh; b.cap = want;
  }
  *out = b.p;
  return MLBP_OK;
}

int group_table_device(GroupTables& gt, const std::vector<int32_t>& table, void* stream, int32_t** out) {
  for (auto& e : gt.entries)
    if (e.words == table) { *out = e.dev; return MLBP_OK; }
  if (getenv("MLBP_DEBUG_GROUP_TABLE")) {
    fprintf(stderr, "group table miss: %zu words, %zu cached\n", table.size(), gt.entries.size());
    for (auto& e : gt.entries)
      if (e.words.size() == table.size())
        for (size_t i = 0; i < table.size(); ++i)
          if (e.words[i] != table[i]) { fprintf(stderr, "  first difference to a cached table at word %zu: %d vs %d\n", i, e.words[i], table[i]); break; }
  }
  GroupTables::Entry* slot = nullptr;
  if (gt.entries.size() < (size_t)GroupTables::MAX) { gt.entries.emplace_back(); slot = &gt.entries.back(); }
  else { slot = &gt.entries[gt.next_evict]; gt.next_evict = (gt.next_evict + 1) % GroupTables::MAX; }
  if (table.size() > slot->cap_words) {
    int32_t* fresh = nullptr;
    if (hipMalloc(&fresh, table.size() * sizeof(int32_t)) != hipSuccess) return fail(MLBP_EHIP, "group table allocation failed");
    (void)hipFree(slot->dev);               // (only a recycled slot has one: its table is being replaced anyway)
    slot->dev = fresh; slot->cap_words = table.size();
  }
  slot->words = table;
  if (hipMemcpyAsync(slot->dev, slot->words.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return fail(MLBP_EHIP, "group table upload failed");
  *out = slot->dev;
  return MLBP_OK;
}

void group_tables_free(GroupTables& gt) {
  for (auto& e : gt.entries) (void)hipFree(e.dev);
  gt.entries.clear();
}
}  // namespace mlbp

extern "C" {

int mlbp_program_reserve(mlbp_program* p, int32_t max_graphs) {
  if (!p || max_graphs <= 0) return fail(MLBP_EINVAL, "mlbp_program_reserve: bad arguments");
  if (p->pruned)
    if (int e = mlbp_program_reserve(p->pruned, max_graphs)) return e;
  if (p->bail_cap >= max_graphs) return MLBP_OK;
  size_t cap = (size_t)p->bail_cap;                        // (cleared: mlbp_program_exact_count before any fast-path launch reads 0)
  if (int e = mlbp::program_grow(p, reinterpret_cast<void**>(&p->d_bail), &cap, (size_t)max_graphs, true)) return e;
  p->bail_cap = max_graphs;
  return MLBP_OK;
}

int mlbp_program_set_readout(mlbp_program* p, int32_t n_vars, const int32_t* in_off, const int32_t* in_slots) {
  if (!p || n_vars <= 0 || !in_off || !in_slots) return fail(MLBP_EINVAL, "mlbp_program_set_readout: bad arguments");
  if (in_off[0] != 0) return fail(MLBP_EINVAL, "in_off[0] must be 0");
  for (int v = 0; v < n_vars; ++v)
    if (in_off[v + 1] < in_off[v]) return fail(MLBP_EINVAL, "in_off must be non-decreasing");
  const int n_in = in_off[n_vars];
  for (int q = 0; q < n_in; ++q)
    if (in_slots[q] < 0 || in_slots[q] >= p->n_msgs) return fail(MLBP_EINVAL, "in_slots[%d] = %d out of [0,%d)", q, in_slots[q], p->n_msgs);
  if (p->pruned)
    if (int e = mlbp_program_set_readout(p->pruned, n_vars, in_off, in_slots)) return e;
  std::vector<int32_t> img(in_off, in_off + n_vars + 1);
  img.insert(img.end(), in_slots, in_slots + n_in);
  img.push_back(0);
  (void)hipFree(p->d_readout);
  p->d_readout = nullptr;
  HIP_TRY(hipMalloc(&p->d_readout, img.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(p->d_readout, img.data(), img.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  p->n_vars = n_vars;
  p->n_readout = (int)img.size();
  (void)hipFree(p->d_lreadout);
  p->d_lreadout = nullptr;
  std::vector<int32_t> limg;
  if (p->lean.ok && mlbp::build_lean_readout(p->lean, p->n_msgs, n_vars, in_off, in_slots, limg)) {
    HIP_TRY(hipMalloc(&p->d_lreadout, limg.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p->d_lreadout, limg.data(), limg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  (void)hipFree(p->d_sreadout);
  p->d_sreadout = nullptr;
  p->n_sreadout = 0;
  std::vector<int32_t> simg;
  p->sreadout_all_based = false; p->sreadout_all_tiled = false;
  if (p->shared.ok && mlbp::build_shared_readout(p->shared, p->n_msgs, n_vars, in_off, in_slots, simg)) {
    p->sreadout_all_based = true; p->sreadout_all_tiled = true;
    for (int v = 0; v < n_vars; ++v) {
      p->sreadout_all_based &= simg[simg[v]] >= 0;
      p->sreadout_all_tiled &= simg[simg[v] + 1] >= 1;       // (the three-source product-fused read-out stages a variable's rows in its first message tile)
      for (int u = 0; u < v; ++u) p->sreadout_all_based &= simg[simg[u]] != simg[simg[v]];      // (and its own: the read-out stages a variable's rows there)
    }
    HIP_TRY(hipMalloc(&p->d_sreadout, simg.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p->d_sreadout, simg.data(), simg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    p->n_sreadout = (int)simg.size();
  }
  return MLBP_OK;
}

int mlbp_set_sweep_variant(int32_t variant) {
  const bool known = variant == 1 || variant == 3;
  if (!known) return fail(MLBP_EINVAL, "unknown sweep variant %d", variant);
  g_sweep_variant = variant;
  return MLBP_OK;
}

int mlbp_last_sweep_kernel(void) { return g_last_kernel; }
int mlbp_last_sweep_fused_gradient(void) { return g_last_fused_gradient; }

int mlbp_program_exact_count(const mlbp_program* prog, int32_t B) {
  // Synchronising: how many of the first B graphs of the last default-variant launch were handed
  // to the exact kernel (0 when the scale-free kernel was not used).
  if (!prog || B < 0) return fail(MLBP_EINVAL, "mlbp_program_exact_count: bad arguments");
  if (prog->last_was_pruned && prog->pruned) prog = prog->pruned;
  if (!prog->d_bail || B == 0) return 0;
  if (B > prog->bail_cap) B = prog->bail_cap;
  std::vector<unsigned char> h((size_t)B);
  HIP_TRY(hipMemcpy(h.data(), prog->d_bail, (size_t)B, hipMemcpyDeviceToHost));
  int n = 0, hist[4] = {0, 0, 0, 0};
  for (unsigned char c : h) { n += c ? 1 : 0; hist[c & 3]++; }
  fail(0, "exact-kernel graphs by reason: prologue %d, main loop %d, final pass %d (shared-table kernel: 2 = degenerate total, 4 -> counted under 0 = tables not shared)", hist[1], hist[2], hist[3]);
  return n;
}

int mlbp_program_status(const mlbp_program* prog) {
  // Synchronising read of the status word: 0 = clean, 1 = a kernel skipped a graph because a
  // table index was out of range.  Resets the word.
  if (!prog) return fail(MLBP_EINVAL, "NULL program");
  int32_t v = 0, zero = 0;
  HIP_TRY(hipMemcpy(&v, prog->d_status, sizeof(v), hipMemcpyDeviceToHost));
  if (v) HIP_TRY(hipMemcpy(prog->d_status, &zero, sizeof(zero), hipMemcpyHostToDevice));
  if (prog->pruned) {
    const int w = mlbp_program_status(prog->pruned);
    if (w < 0) return w;
    v |= w;
  }
  return v;
}

int mlbp_program_skippable_updates(const mlbp_program* prog) {
  if (!prog) return fail(MLBP_EINVAL, "NULL program");
  return prog->n_dropped;
}

int mlbp_init_messages_f64(double* msgs, int64_t n_rows, int32_t X, void* stream) {
  if (!msgs || n_rows <= 0 || X <= 0) return fail(MLBP_EINVAL, "mlbp_init_messages_f64: bad arguments");
  if (int e = check_device()) return e;
  int64_t n = n_rows * X;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, msgs, n, 1.0 / (double)X);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_marginals_f64(const double* msgs, int32_t B, int32_t n_msgs, int32_t X, int32_t n_vars,
                       const int32_t* in_off, const int32_t* in_slots, int32_t normalize_messages,
                       double* out, void* stream) {
  if (!msgs || !in_off || !in_slots || !out || B <= 0 || n_msgs <= 0 || X <= 0 || n_vars <= 0)
    return fail(MLBP_EINVAL, "mlbp_marginals_f64: bad arguments");
  if (int e = check_device()) return e;
  hipLaunchKernelGGL(marginals_kernel, dim3(B), dim3(WG), 0, (hipStream_t)stream, msgs, n_msgs, X, n_vars,
                     in_off, in_slots, normalize_messages, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_log_posterior_sum_f64(const double* marginals, const int32_t* labels, int32_t B, int32_t n_vars,
                               int32_t X, double* out, double* sum_out, void* stream) {
  if (!marginals || !labels || !out || B <= 0 || n_vars <= 0 || X <= 0)
    return fail(MLBP_EINVAL, "mlbp_log_posterior_f64: bad arguments");
  if (int e = check_device()) return e;
  int32_t* status = nullptr;
  if (int e = global_status(&status)) return e;
  const int blocks = (B + LP_WG - 1) / LP_WG;
  if (sum_out && blocks > LP_MAX_BLOCKS)
    return fail(MLBP_EUNSUPPORTED, "mlbp_log_posterior_sum_f64: at most %d graphs with sum_out", LP_MAX_BLOCKS * LP_WG);
  // the block partials live in one device-wide scratch array: launches on DIFFERENT streams must not overlap.  Every
  // launch has its own generation number (the arrival word restarts with it: see the kernel)
  const unsigned gen = sum_out ? next_lp_generation() : 0u;
  hipLaunchKernelGGL(log_posterior_kernel, dim3(blocks), dim3(LP_WG), 0, (hipStream_t)stream, marginals, labels, B, n_vars, X, out,
                     sum_out, status, gen);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_log_posterior_groups_f64(const mlbp_posterior_group* groups, int32_t n_groups, int64_t n_total, int32_t X, double* out,
                                  void* stream) {
  if (!groups || !out || n_groups <= 0 || n_total <= 0 || X <= 0) return fail(MLBP_EINVAL, "mlbp_log_posterior_groups_f64: bad arguments");
  if (int e = check_device()) return e;
  int32_t* status = nullptr;
  if (int e = global_status(&status)) return e;
  hipLaunchKernelGGL(log_posterior_groups_kernel, dim3((unsigned)((n_total + LP_WG - 1) / LP_WG)), dim3(LP_WG), 0, (hipStream_t)stream, groups,
                     n_groups, (long long)n_total, X, out, status);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_log_posterior_f64(const double* marginals, const int32_t* labels, int32_t B, int32_t n_vars,
                           int32_t X, double* out, void* stream) {
  return mlbp_log_posterior_sum_f64(marginals, labels, B, n_vars, X, out, nullptr, stream);
}

}  // extern "C"
