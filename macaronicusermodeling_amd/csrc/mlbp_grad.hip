// Factor beliefs and the log-linear gradient, batched over graphs (gfx950).
//
// Reference path: FactorNode.get_factor_beliefs (LBP.py:528-574), cell_gradient (LBP.py:615-619),
// get_gradient (LBP.py:592-613) and FactorGraph.get_unregularized_gradeint (LBP.py:301-320).
// The reference materialises three X*X temporaries per pairwise factor (outer product, product
// with the table, normalised beliefs) and then contracts with the (X,X,F) feature tensor.  Here one
// pass over the table accumulates Z = sum c_i r_j T_ij and S_k = sum c_i r_j T_ij phi_ijk, and
//     grad_k = phi[l0][l1][k] - S_k / Z            (zero beliefs when Z <= 0, like au.normalize)
// so the belief matrix never exists in memory.  Per pairwise factor the HBM traffic is the table
// (X*X*8 bytes, unique per graph) -- the feature tensors are shared by the whole batch and stay in
// L2.  One workgroup per graph; wave-level reductions; results per graph, summed over the batch by
// mlbp_sum_rows_f64 (the device half of train_mp.py's accumulate callback).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "mlbp_internal.h"

using mlbp::fail;

namespace {

constexpr int WG = 256;
constexpr int FMAX = 8;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(MLBP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);   // one v_mov_b32_dpp each (update_dpp adds a copy)
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

struct GradDev {
  mlbp_gradient_args a;
  int32_t* status;
  int32_t skip_pairs;      // the pairwise factors are handled by the shared-table (MFMA) kernel
  const uint8_t* only;     // non-NULL (X = 64 kernel): only the graphs whose flag byte is set (the fix-up behind a fused gradient)
};

// sums[0..n) over the workgroup; result valid in every thread.  scratch: [4][FMAX+1] doubles.
template <int N>
__device__ __forceinline__ void block_sums(double (&v)[N], double* scratch) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) scratch[(threadIdx.x >> 6) * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = (scratch[k] + scratch[N + k]) + (scratch[2 * N + k] + scratch[3 * N + k]);
}

template <int F>
__device__ void pair_gradient(const GradDev& d, int g, int p, double* scratch, double* out) {
  const mlbp_gradient_args& a = d.a;
  const int X = a.X;
  const int tab = a.pair_tab[(size_t)g * a.P + p];
  const int l0 = a.pair_label[((size_t)g * a.P + p) * 2], l1 = a.pair_label[((size_t)g * a.P + p) * 2 + 1];
  const bool ok = (unsigned)tab < (unsigned)a.n_pair_tables && (unsigned)l0 < (unsigned)X && (unsigned)l1 < (unsigned)X;
  if (!ok) {
    if (threadIdx.x == 0) atomicExch(d.status, 1);
    return;
  }
  const double* T = a.pair_tables + (size_t)tab * X * X;
  const double* phi = a.pair_phi[p] ? a.phi_en_en_w1 : a.phi_en_en;
  const double* c = a.msgs + ((size_t)g * a.n_msgs + a.pair_c_slot[p]) * X;
  const double* r = a.msgs + ((size_t)g * a.n_msgs + a.pair_r_slot[p]) * X;
  if (a.flags & MLBP_GRADIENT_APPROX_BELIEFS) {
    // use_approx_beliefs (LBP.py:554-563; au.sparse_dot / sparse_pointwise_multiply / sparse_normalize): beliefs live on
    // the block of the K largest entries of c times the K largest of r -- the other entries of both vectors are dropped
    extern __shared__ double kept[];                 // [2][X]
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * X; j += WG) {
      const double* v = j < X ? c : r;
      const int jj = j < X ? j : j - X;
      const double x = v[jj];
      int rank = 0;
      for (int i = 0; i < X; ++i) {
        const double y = v[i];
        rank += (y > x) || (y == x && i < jj);
      }
      kept[j] = rank < MLBP_APPROX_K ? x : 0.0;
    }
    __syncthreads();
    c = kept; r = kept + X;
  }
  double acc[F + 1];
#pragma unroll
  for (int k = 0; k <= F; ++k) acc[k] = 0.0;
  for (int e = threadIdx.x; e < X * X; e += WG) {
    const int i = e / X, j = e - i * X;
    const double w = (c[i] * r[j]) * T[e];     // (c.r) then * T: the reference's order (LBP.py:566-568)
    acc[0] += w;
#pragma unroll
    for (int k = 0; k < F; ++k) acc[1 + k] += w * phi[(size_t)e * F + k];
  }
  block_sums<F + 1>(acc, scratch);
  const double Z = acc[0];
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const double expect = Z > 0.0 ? acc[1 + k] / Z : 0.0;
    out[k] += phi[((size_t)l0 * X + l1) * F + k] - expect;
  }
}

// One wave per unary factor: beliefs = au.normalize(table) (LBP.py:540), gradient
// g^T . phi[:, observed_dim, :] (LBP.py:600-603).
template <int F>
__device__ void unary_gradient(const GradDev& d, int g, int u, const double* phi, int cols, double* out) {
  const mlbp_gradient_args& a = d.a;
  const int X = a.X;
  const int lane = threadIdx.x & 63;
  const int tab = a.unary_tab[(size_t)g * a.U + u];
  const int obs = a.unary_obs[(size_t)g * a.U + u];
  const int lab = a.unary_label[(size_t)g * a.U + u];
  const bool ok = (unsigned)tab < (unsigned)a.n_unary_tables && (unsigned)obs < (unsigned)cols && (unsigned)lab < (unsigned)X;
  if (!ok) {
    if (lane == 0) atomicExch(d.status, 1);
    return;
  }
  const double* t = a.unary_tables + (size_t)tab * X;
  double acc[F + 1];
#pragma unroll
  for (int k = 0; k <= F; ++k) acc[k] = 0.0;
  for (int x = lane; x < X; x += 64) {
    const double w = t[x];
    acc[0] += w;
#pragma unroll
    for (int k = 0; k < F; ++k) acc[1 + k] += w * phi[((size_t)x * cols + obs) * F + k];
  }
#pragma unroll
  for (int k = 0; k <= F; ++k) acc[k] = wave_sum(acc[k]);
  const double Z = acc[0];
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const double expect = Z > 0.0 ? acc[1 + k] / Z : 0.0;
    out[k] += phi[((size_t)lab * cols + obs) * F + k] - expect;
  }
}

template <int FEE, int FED>
__global__ __launch_bounds__(WG) void gradient_kernel(GradDev d) {
  __shared__ double scratch[4 * (FMAX + 1)];
  __shared__ double wave_out[4][2 * FMAX];
  const mlbp_gradient_args& a = d.a;
  const int g = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // pairwise factors: whole workgroup per factor (all are en_en, LBP.py:305-315)
  double gee[FEE];
#pragma unroll
  for (int k = 0; k < FEE; ++k) gee[k] = 0.0;
  for (int p = 0; p < a.P && !d.skip_pairs; ++p) pair_gradient<FEE>(d, g, p, scratch, gee);
  // unary factors: one wave each
  double uee[FEE], ued[FED];
#pragma unroll
  for (int k = 0; k < FEE; ++k) uee[k] = 0.0;
#pragma unroll
  for (int k = 0; k < FED; ++k) ued[k] = 0.0;
  for (int u = wave; u < a.U; u += 4) {
    const int kind = a.unary_kind[u];
    if (kind == 2) unary_gradient<FED>(d, g, u, a.phi_en_de, a.Vde, ued);
    else unary_gradient<FEE>(d, g, u, kind ? a.phi_en_en_w1 : a.phi_en_en, a.X, uee);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < FEE; ++k) wave_out[wave][k] = uee[k];
#pragma unroll
    for (int k = 0; k < FED; ++k) wave_out[wave][FMAX + k] = ued[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < FEE; ++k)
      a.grad_en_en[(size_t)g * FEE + k] = gee[k] + ((wave_out[0][k] + wave_out[1][k]) + (wave_out[2][k] + wave_out[3][k]));
#pragma unroll
    for (int k = 0; k < FED; ++k)
      a.grad_en_de[(size_t)g * FED + k] =
          (wave_out[0][FMAX + k] + wave_out[1][FMAX + k]) + (wave_out[2][FMAX + k] + wave_out[3][FMAX + k]);
  }
}

// ---- X = 64 specialisation ----------------------------------------------------------------------------
// Same thread <-> table-element map as the sweep kernel (thread t: rows 8k + (t>>5), columns
// 2(t&31), +1; every load instruction of the workgroup covers 4 KiB of the table).  The feature
// tensor is read in the matching order: for one row and column pair the 2F feature values are
// contiguous (2F*8 bytes per lane, consecutive lanes consecutive), so phi streams from L2 coalesced
// while T streams from HBM.  Unary factors: one wave each, the table row (512 B) and -- when the
// caller supplies the transposed feature tensors [column][x][F] -- a contiguous 64*F*8-byte slab of
// phi; without them the strided column gather of the generic kernel is used.
template <int F>
__device__ __forceinline__ void pair_gradient_x64(const GradDev& d, int g, int p, double* scratch, double* out) {
  const mlbp_gradient_args& a = d.a;
  const int t = threadIdx.x, rg = t >> 5, cp = t & 31;
  const int tab = a.pair_tab[(size_t)g * a.P + p];
  const int l0 = a.pair_label[((size_t)g * a.P + p) * 2], l1 = a.pair_label[((size_t)g * a.P + p) * 2 + 1];
  const bool ok = (unsigned)tab < (unsigned)a.n_pair_tables && (unsigned)l0 < 64u && (unsigned)l1 < 64u;
  if (!ok) {
    if (t == 0) atomicExch(d.status, 1);
    return;
  }
  const double2* T = reinterpret_cast<const double2*>(a.pair_tables + (size_t)tab * 4096);
  const double* phi = a.pair_phi[p] ? a.phi_en_en_w1 : a.phi_en_en;
  const double* c = a.msgs + ((size_t)g * a.n_msgs + a.pair_c_slot[p]) * 64;
  const double* r = a.msgs + ((size_t)g * a.n_msgs + a.pair_r_slot[p]) * 64;
  double2 tv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) tv[k] = T[k * WG + t];
  const double2 rj = reinterpret_cast<const double2*>(r)[cp];
  const double* planar = a.pair_phi[p] ? a.phi_en_en_w1_p : a.phi_en_en_p;     // [F][64][64] or NULL
  double acc[F + 1];
#pragma unroll
  for (int k = 0; k <= F; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = 8 * k + rg;
    const double ci = c[i];
    const double w0 = (ci * rj.x) * tv[k].x, w1 = (ci * rj.y) * tv[k].y;     // (c.r) * T, LBP.py:566-568
    acc[0] += w0 + w1;
    if (planar) {                                                            // table-like 16-byte-per-lane reads
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const double2 q = reinterpret_cast<const double2*>(planar)[f * 2048 + k * WG + t];
        acc[1 + f] += w0 * q.x + w1 * q.y;
      }
    } else {
      const double* ph = phi + ((size_t)i * 64 + 2 * cp) * F;                // 2F contiguous doubles
#pragma unroll
      for (int f = 0; f < F; ++f) acc[1 + f] += w0 * ph[f] + w1 * ph[F + f];
    }
  }
  block_sums<F + 1>(acc, scratch);
  const double Z = acc[0];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const double expect = Z > 0.0 ? acc[1 + f] / Z : 0.0;
    out[f] += phi[((size_t)l0 * 64 + l1) * F + f] - expect;
  }
}

template <int F>
__device__ __forceinline__ void unary_gradient_x64(const GradDev& d, int g, int u, const double* phi, const double* phi_t,
                                                   int cols, double* out) {
  const mlbp_gradient_args& a = d.a;
  const int lane = threadIdx.x & 63;
  const int tab = a.unary_tab[(size_t)g * a.U + u];
  const int obs = a.unary_obs[(size_t)g * a.U + u];
  const int lab = a.unary_label[(size_t)g * a.U + u];
  const bool ok = (unsigned)tab < (unsigned)a.n_unary_tables && (unsigned)obs < (unsigned)cols && (unsigned)lab < 64u;
  if (!ok) {
    if (lane == 0) atomicExch(d.status, 1);
    return;
  }
  const double w = a.unary_tables[(size_t)tab * 64 + lane];
  const double* ph = phi_t ? phi_t + ((size_t)obs * 64 + lane) * F : phi + ((size_t)lane * cols + obs) * F;
  double acc[F + 1];
  acc[0] = w;
#pragma unroll
  for (int f = 0; f < F; ++f) acc[1 + f] = w * ph[f];
#pragma unroll
  for (int f = 0; f <= F; ++f) acc[f] = wave_sum(acc[f]);
  const double Z = acc[0];
  const double* pl = phi_t ? phi_t + ((size_t)obs * 64 + lab) * F : phi + ((size_t)lab * cols + obs) * F;
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] += pl[f] - (Z > 0.0 ? acc[1 + f] / Z : 0.0);
}

template <int FEE, int FED>
__device__ __forceinline__ void gradient_x64_body(const GradDev& d, const int g);

template <int FEE, int FED>
__global__ __launch_bounds__(WG) void gradient_x64_kernel(GradDev d) {
  const int g = blockIdx.x;
  if (d.only && !d.only[g]) return;
  gradient_x64_body<FEE, FED>(d, g);
}

// Several groups of graphs in one launch, flagged graphs only (the fix-up behind mlbp_sweep_groups_f64's fused gradients):
// groups[k] = the group's description, first[k] = its first block (ascending; one block per GRADIENT_GROUPS_GB graphs: the block
// reads their flags together and runs the per-graph body for the flagged ones -- with one block per graph a launch over a few
// thousand graphs of a few hundred groups, nothing flagged, spent 17 us on its blocks' group searches).
constexpr int GRADIENT_GROUPS_GB = 16;
struct GradGroup { GradDev d; int32_t first, per_block; };      // per_block: graphs a block looks at (GRADIENT_GROUPS_GB; 1 for a group that is flagged as a whole)
__global__ __launch_bounds__(WG) void gradient_x64_groups_kernel(const GradGroup* groups, int n_groups) {
  int lo = 0, hi = n_groups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (groups[mid].first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int per = groups[lo].per_block;
  const int g0 = ((int)blockIdx.x - groups[lo].first) * per;
  const uint8_t* only = groups[lo].d.only;
  const int B = groups[lo].d.a.B;
  if (g0 >= B || !only) return;
  const int lane = threadIdx.x & 63;
  const bool mine = lane < per && g0 + lane < B && only[g0 + lane] != 0;
  unsigned long long todo = __ballot(mine);                     // (the same in every wave of the block)
  if (!todo) return;
  const GradDev d = groups[lo].d;
  while (todo) {
    const int j = __builtin_ctzll(todo);
    todo &= todo - 1;
    gradient_x64_body<3, 6>(d, g0 + j);
    __syncthreads();                                             // (the body's shared scratch, before the next graph uses it)
  }
}

template <int FEE, int FED>
__device__ __forceinline__ void gradient_x64_body(const GradDev& d, const int g) {
  __shared__ double scratch[4 * (FMAX + 1)];
  __shared__ double wave_out[4][2 * FMAX];
  const mlbp_gradient_args& a = d.a;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double gee[FEE];
#pragma unroll
  for (int k = 0; k < FEE; ++k) gee[k] = 0.0;
  for (int p = 0; p < a.P && !d.skip_pairs; ++p) pair_gradient_x64<FEE>(d, g, p, scratch, gee);
  double uee[FEE], ued[FED];
#pragma unroll
  for (int k = 0; k < FEE; ++k) uee[k] = 0.0;
#pragma unroll
  for (int k = 0; k < FED; ++k) ued[k] = 0.0;
  for (int u = wave; u < a.U; u += 4) {
    const int kind = a.unary_kind[u];
    if (kind == 2) unary_gradient_x64<FED>(d, g, u, a.phi_en_de, a.phi_en_de_t, a.Vde, ued);
    else unary_gradient_x64<FEE>(d, g, u, kind ? a.phi_en_en_w1 : a.phi_en_en, kind ? a.phi_en_en_w1_t : a.phi_en_en_t, 64, uee);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < FEE; ++k) wave_out[wave][k] = uee[k];
#pragma unroll
    for (int k = 0; k < FED; ++k) wave_out[wave][FMAX + k] = ued[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < FEE; ++k)
      a.grad_en_en[(size_t)g * FEE + k] = gee[k] + ((wave_out[0][k] + wave_out[1][k]) + (wave_out[2][k] + wave_out[3][k]));
#pragma unroll
    for (int k = 0; k < FED; ++k)
      a.grad_en_de[(size_t)g * FED + k] =
          (wave_out[0][FMAX + k] + wave_out[1][FMAX + k]) + (wave_out[2][FMAX + k] + wave_out[3][FMAX + k]);
  }
}

// E[row][k] = sum_x normalize(row)[x] * phi_t[obs][x][k]: one wave per table row (X = 64), see mlbp.h.
__global__ __launch_bounds__(WG) void unary_expectations_kernel(const double* tables, int n_rows, const int32_t* row_kind,
                                                                const int32_t* row_obs, const double* t_ee, const double* t_w1,
                                                                const double* t_ed, int F_ee, int F_ed, int Vde, double* out,
                                                                int32_t* status) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int kind = row_kind[row], obs = row_obs[row];
  const int F = kind == 2 ? F_ed : F_ee;
  if ((unsigned)kind > 2u || (unsigned)obs >= (unsigned)(kind == 2 ? Vde : 64)) {
    if (lane == 0) atomicExch(status, 1);
    return;
  }
  const double* ph = (kind == 2 ? t_ed : (kind ? t_w1 : t_ee)) + ((size_t)obs * 64 + lane) * F;
  const double w = tables[(size_t)row * 64 + lane];
  const double Z = wave_sum(w);
  for (int f = 0; f < F; ++f) {
    const double e = wave_sum(w * ph[f]);
    if (lane == 0) out[(size_t)row * 8 + f] = Z > 0.0 ? e / Z : 0.0;          // au.normalize: zero-sum -> 0
  }
}

// beliefs of every pairwise factor, materialised: out[b][p][i][j]
__global__ __launch_bounds__(WG) void pair_beliefs_kernel(const double* msgs, int n_msgs, int X, int P,
                                                          const double* tables, const int32_t* pair_tab,
                                                          int n_tables, const int32_t* c_slot, const int32_t* r_slot,
                                                          double* out, int32_t* status) {
  __shared__ double scratch[4];
  const int g = blockIdx.x / P, p = blockIdx.x % P;
  const int tab = pair_tab[(size_t)g * P + p];
  if ((unsigned)tab >= (unsigned)n_tables) {
    if (threadIdx.x == 0) atomicExch(status, 1);
    return;
  }
  const double* T = tables + (size_t)tab * X * X;
  const double* c = msgs + ((size_t)g * n_msgs + c_slot[p]) * X;
  const double* r = msgs + ((size_t)g * n_msgs + r_slot[p]) * X;
  double* o = out + (size_t)blockIdx.x * X * X;
  double part = 0.0;
  for (int e = threadIdx.x; e < X * X; e += WG) {
    const int i = e / X, j = e - i * X;
    const double w = (c[i] * r[j]) * T[e];
    o[e] = w;
    part += w;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = part;
  __syncthreads();
  const double Z = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  for (int e = threadIdx.x; e < X * X; e += WG) o[e] = Z > 0.0 ? o[e] / Z : 0.0;
}

// out[j] = sum_b in[b][j], deterministic (fixed order): SUM_PARTS workgroups each reduce a contiguous slice of
// the rows to one partial per column, the last one to finish (device-wide counter) adds the partials in
// slice order.  One launch, no host round trip, the same bits on every run.
constexpr int SUM_PARTS = 64;
__device__ double g_sum_partials[SUM_PARTS * 64];
__device__ unsigned g_sum_done = 0;

// up to three [rows][cols_i] arrays read as one [rows][sum cols] matrix; with `marg` the third array's single column is not
// read but computed: row b's log-posterior sum_v log marg[b][v][labels[b][v]] (LBP.py:247-259), also stored to lp_out if given
struct SumCat {
  const double* in[3]; int cols[3];
  const double* marg; const int32_t* labels; int n_vars, X; double* lp_out; int32_t* status;
  const int32_t* key; const int32_t* key_value;       // optional selection: only the rows with key[b] == *key_value (both DEVICE)
};

__global__ __launch_bounds__(WG) void sum_rows_kernel(SumCat cat, int64_t rows, int append_count, double* out) {
  // every thread sums ALL columns of its rows (fixed order), the columns then meet once: lanes by DPP, waves through
  // LDS -- one barrier per launch instead of nine per column
  __shared__ double part[WG / 64][64];
  __shared__ bool last;
  const int64_t per = (rows + SUM_PARTS - 1) / SUM_PARTS;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kv = cat.key ? *cat.key_value : 0;
  int jout = 0;
  double selected = 0.0;                                       // rows of this thread that passed the selection (first pass counts)
  for (int a = 0; a < 3; ++a) {
    const double* in = cat.in[a];
    const int cols = cat.cols[a];
    for (int j0 = 0; j0 < cols; j0 += 8) {                 // eight columns at a time in registers
      const int nj = cols - j0 < 8 ? cols - j0 : 8;
      double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (a == 2 && cat.marg) {
        for (int64_t b = r0 + threadIdx.x; b < r1; b += WG) {
          double total = 0.0;
          for (int v = 0; v < cat.n_vars; ++v) {
            const int lab = cat.labels[b * cat.n_vars + v];
            if ((unsigned)lab >= (unsigned)cat.X) { atomicExch(cat.status, 1); continue; }
            const double lp = log(cat.marg[(b * cat.n_vars + v) * cat.X + lab]);
            total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
          }
          if (cat.lp_out) cat.lp_out[b] = total;
          acc[0] += total;
        }
      } else
      for (int64_t b = r0 + threadIdx.x; b < r1; b += WG) {
        if (cat.key && cat.key[b] != kv) continue;
        if (a == 0 && j0 == 0) selected += 1.0;
        const double* row = in + b * cols + j0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nj) acc[j] += row[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nj) {
          const double v = wave_sum(acc[j]);
          if (lane == 0) part[wave][jout + j] = v;
        }
      jout += nj;
    }
  }
  if (cat.key && append_count) {                               // the count column of a selection: the rows that passed
    const double v = wave_sum(selected);
    if (lane == 0) part[wave][jout] = v;
    jout += 1;
  }
  __syncthreads();
  if (threadIdx.x < jout) {
    double v = part[0][threadIdx.x];
    for (int w = 1; w < WG / 64; ++w) v += part[w][threadIdx.x];
    g_sum_partials[blockIdx.x * 64 + threadIdx.x] = v;
    __threadfence();                                         // visible device-wide before this workgroup counts itself done
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(&g_sum_done, 1u) == SUM_PARTS - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < jout) {
    double acc = 0.0;
    for (int q = 0; q < SUM_PARTS; ++q) acc += __builtin_nontemporal_load(&g_sum_partials[q * 64 + threadIdx.x]);
    out[threadIdx.x] = acc;
  }
  if (threadIdx.x == 0 && append_count && !cat.key) out[jout] = (double)rows;
  if (threadIdx.x == 0) g_sum_done = 0;                       // ready for the next (stream-ordered) launch
}


// mlbp_step_statistics_f64 at the trainer's feature counts: sum_rows_kernel's arithmetic in its order (so: its bits), with a
// row's nine gradient entries, its labels and the marginals they select requested together -- the generic kernel walks the
// three arrays one after the other, three rounds of memory latency in a launch that is little else.
template <int C0, int C1>
__global__ __launch_bounds__(WG) void step_statistics_kernel(SumCat cat, int64_t rows, double* out) {
  constexpr int NC = C0 + C1 + 1;
  __shared__ double part[WG / 64][NC];
  __shared__ bool last;
  const int64_t per = (rows + SUM_PARTS - 1) / SUM_PARTS;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) acc[j] = 0.0;
  for (int64_t b = r0 + threadIdx.x; b < r1; b += WG) {
    double v[C0 + C1];
#pragma unroll
    for (int j = 0; j < C0; ++j) v[j] = cat.in[0][b * C0 + j];
#pragma unroll
    for (int j = 0; j < C1; ++j) v[C0 + j] = cat.in[1][b * C1 + j];
    double total = 0.0;
    if (cat.n_vars <= 8) {
      // the row's labels together, then the marginal entries they select together: two rounds of memory latency, not two per
      // variable (the sum below is in variable order, as the loop's)
      int lab[8];
      double mv[8];
#pragma unroll
      for (int vi = 0; vi < 8; ++vi) lab[vi] = vi < cat.n_vars ? cat.labels[b * cat.n_vars + vi] : 0;
#pragma unroll
      for (int vi = 0; vi < 8; ++vi) {
        const bool ok = vi < cat.n_vars && (unsigned)lab[vi] < (unsigned)cat.X;
        mv[vi] = ok ? cat.marg[(b * cat.n_vars + vi) * cat.X + lab[vi]] : 1.0;
      }
#pragma unroll
      for (int vi = 0; vi < 8; ++vi) {
        if (vi >= cat.n_vars) continue;
        if ((unsigned)lab[vi] >= (unsigned)cat.X) { atomicExch(cat.status, 1); continue; }
        const double lp = log(mv[vi]);
        total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
      }
    } else
    for (int vi = 0; vi < cat.n_vars; ++vi) {
      const int lab = cat.labels[b * cat.n_vars + vi];
      if ((unsigned)lab >= (unsigned)cat.X) { atomicExch(cat.status, 1); continue; }
      const double lp = log(cat.marg[(b * cat.n_vars + vi) * cat.X + lab]);
      total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
    }
    if (cat.lp_out) cat.lp_out[b] = total;
#pragma unroll
    for (int j = 0; j < C0 + C1; ++j) acc[j] += v[j];
    acc[NC - 1] += total;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const double s = wave_sum(acc[j]);
    if (lane == 0) part[wave][j] = s;
  }
  __syncthreads();
  if (threadIdx.x < NC) {
    double v = part[0][threadIdx.x];
    for (int w = 1; w < WG / 64; ++w) v += part[w][threadIdx.x];
    g_sum_partials[blockIdx.x * 64 + threadIdx.x] = v;
    __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(&g_sum_done, 1u) == SUM_PARTS - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < NC) {
    double p[SUM_PARTS];
#pragma unroll
    for (int q = 0; q < SUM_PARTS; ++q) p[q] = __builtin_nontemporal_load(&g_sum_partials[q * 64 + threadIdx.x]);
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < SUM_PARTS; ++q) a += p[q];
    out[threadIdx.x] = a;
  }
  if (threadIdx.x == 0) { out[NC] = (double)rows; g_sum_done = 0; }
}

// out[s][j] = sum over the rows b with seg_id[b] == s of in[b][j]: one workgroup per segment, fixed order -- every thread adds
// the matching rows of its stride with ALL columns of a 16-column chunk in registers (one pass over the rows per chunk, where the
// first version made one pass and an eight-barrier tree per column), then lanes by DPP, the four waves in order: the per-domain
// half of batch_sgd_accumulate (train_mp.py:413-415), and the minibatch selection of the trainer's resident shard.
__global__ __launch_bounds__(WG) void segment_sum_rows_kernel(const double* in, int64_t rows, int cols, const int32_t* seg_id,
                                                              double* out) {
  __shared__ double part[WG / 64][16];
  const int seg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j0 = 0; j0 < cols; j0 += 16) {
    const int nj = cols - j0 < 16 ? cols - j0 : 16;
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    for (int64_t b = threadIdx.x; b < rows; b += WG) {
      if (seg_id[b] != seg) continue;
      const double* row = in + b * cols + j0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < nj) acc[j] += row[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nj) {
        const double v = wave_sum(acc[j]);
        if (lane == 0) part[wave][j] = v;
      }
    __syncthreads();
    if (threadIdx.x < nj) {
      double v = part[0][threadIdx.x];
      for (int w = 1; w < WG / 64; ++w) v += part[w][threadIdx.x];
      out[(size_t)seg * cols + j0 + threadIdx.x] = v;
    }
    __syncthreads();
  }
}

// ---- per-instance sparse feature planes (train_mp.py:178-217) ---------------------------------------
// The reference rewrites three planes of phi_en_de per instance ('correct', 'full_history',
// 'hit_history'); they are non-zero in a handful of cells.  Only cells in the observed column of one
// of the instance's predicted words reach the graph, so each affected (instance, en_de factor) gets a
// PRIVATE copy of its table row:  t[x] = base[x] * exp(sum_items theta[k] * val)   (one wave per row).
__global__ __launch_bounds__(64) void patch_tables_kernel(const double* base_tables, const int32_t* base_row,
                                                          const int32_t* item_off, const int32_t* item_x,
                                                          const int32_t* item_k, const double* item_val,
                                                          const double* theta, int X, double* out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const double* b = base_tables + (size_t)base_row[r] * X;
  double* o = out + (size_t)r * X;
  for (int x = lane; x < X; x += 64) {
    double e = 0.0;
    for (int q = item_off[r]; q < item_off[r + 1]; ++q)
      if (item_x[q] == x) e += theta[item_k[q]] * item_val[q];
    o[x] = b[x] * exp(e);
  }
}

// Their gradient share: for every item (x, k, val) of a private row
//   grad_en_de[graph][k] += val * ( [x == label] - t[x] / sum(t) )       (LBP.py:600-603 on those cells)
__global__ __launch_bounds__(64) void patch_gradient_kernel(const double* priv_tables, const int32_t* item_off,
                                                            const int32_t* item_x, const int32_t* item_k,
                                                            const double* item_val, const int32_t* row_graph,
                                                            const int32_t* row_label, int n_rows, int X, int F, double* grad) {
  // One wave per RUN of consecutive rows of one graph (the wave of the run's first row walks it, the others leave): a
  // graph whose rows are contiguous -- the trainer's are -- is added to by one lane in row order, the same bits every launch.
  int r = blockIdx.x;
  const int lane = threadIdx.x, graph = row_graph[r];
  if (r > 0 && row_graph[r - 1] == graph) return;
  double* g = grad + (size_t)graph * F;
  for (; r < n_rows && row_graph[r] == graph; ++r) {
    const double* t = priv_tables + (size_t)r * X;
    double part = 0.0;
    for (int x = lane; x < X; x += 64) part += t[x];
    const double Z = wave_sum(part);
    if (lane == 0) {
      const int lab = row_label[r];
      for (int q = item_off[r]; q < item_off[r + 1]; ++q) {
        const int x = item_x[q];
        const double belief = Z > 0.0 ? t[x] / Z : 0.0;
        atomicAdd(&g[item_k[q]], item_val[q] * ((x == lab ? 1.0 : 0.0) - belief));   // (atomic: a graph split into several runs)
      }
    }
  }
}

int32_t* g_status = nullptr;
int status_word(int32_t** out) {
  if (!g_status) {
    HIP_TRY(hipMalloc(&g_status, sizeof(int32_t)));
    HIP_TRY(hipMemset(g_status, 0, sizeof(int32_t)));
  }
  *out = g_status;
  return MLBP_OK;
}

int need_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MLBP_ENODEVICE, "no HIP device visible: libmlbp.so has no CPU fallback");
  }
  return MLBP_OK;
}

}  // namespace

// which shared-table form a gradient call takes (a function of the arguments alone)
static bool takes_shared_x64(const mlbp_gradient_args* a) {
  return a->X == 64 && !(a->flags & MLBP_GRADIENT_APPROX_BELIEFS) && (a->flags & MLBP_GRADIENT_SHARED_PAIR_TABLES) && a->F_ee == 3 && a->P > 0 &&
         a->n_pair_tables <= 32 && a->phi_en_en_p && a->phi_en_en_w1_p;
}
static bool takes_gemm_pairs(const mlbp_gradient_args* a) {
  return a->X != 64 && !(a->flags & MLBP_GRADIENT_APPROX_BELIEFS) && (a->flags & MLBP_GRADIENT_SHARED_PAIR_TABLES) && a->pair_tab_host &&
         mlbp::gemm_path_supports(a->X) && a->F_ee == 3 && a->P > 0 && a->P <= 16 && a->phi_en_en_p && a->phi_en_en_w1_p;
}

namespace mlbp {
// The gradient of the graphs whose flag byte is set, by the per-graph X = 64 kernel (every factor of the graph, from the
// messages in memory): the fix-up behind a shared-table sweep whose epilogue produced the gradient of all the others.
int gradient_flagged_only(const mlbp_gradient_args* a, const uint8_t* flags, void* stream) {
  if (!a || a->X != 64 || a->F_ee != 3 || a->F_ed != 6) return fail(MLBP_EINVAL, "gradient_flagged_only: X = 64, F = (3, 6)");
  GradDev d;
  d.a = *a;
  d.skip_pairs = 0;
  d.only = flags;
  if (int e = status_word(&d.status)) return e;
  hipLaunchKernelGGL((gradient_x64_kernel<3, 6>), dim3(a->B), dim3(WG), 0, (hipStream_t)stream, d);
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "gradient fix-up launch failed");
  return MLBP_OK;
}
int gradient_flagged_groups(const mlbp_gradient_args* args, const uint8_t* const* flags, int n_groups, mlbp_program* owner, void* stream) {
  std::vector<GradGroup> table(n_groups);
  int blocks = 0;
  int32_t* status = nullptr;
  if (int e = status_word(&status)) return e;
  for (int k = 0; k < n_groups; ++k) {
    if (args[k].X != 64 || args[k].F_ee != 3 || args[k].F_ed != 6) return fail(MLBP_EINVAL, "gradient_flagged_groups: X = 64, F = (3, 6)");
    memset(&table[k], 0, sizeof(GradGroup));
    table[k].d.a = args[k]; table[k].d.status = status; table[k].d.skip_pairs = 0; table[k].d.only = flags[k];
    table[k].first = blocks;
    table[k].per_block = args[k].P == 0 ? 1 : GRADIENT_GROUPS_GB;      // (no pairwise factor: the group is flagged as a whole, mlbp_sweep_groups_f64)
    blocks += (args[k].B + table[k].per_block - 1) / table[k].per_block;
  }
  static_assert(sizeof(GradGroup) % 4 == 0, "");
  std::vector<int32_t> words(sizeof(GradGroup) / 4 * (size_t)n_groups + 1);
  memcpy(words.data(), table.data(), sizeof(GradGroup) * (size_t)n_groups);
  words.back() = 0x47524144;
  int32_t* d_table = nullptr;
  if (int e = group_table_device(owner->stables, words, stream, &d_table)) return e;
  hipLaunchKernelGGL(gradient_x64_groups_kernel, dim3(blocks), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const GradGroup*>(d_table), n_groups);
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "gradient fix-up launch failed");
  return MLBP_OK;
}
}  // namespace mlbp

extern "C" {

int64_t mlbp_gradient_workspace_bytes(const mlbp_gradient_args* a) {
  if (!a) return 0;
  if (takes_shared_x64(a)) return (int64_t)mlbp::shared_gradient_workspace_bytes(a);
  if (takes_gemm_pairs(a)) return (int64_t)mlbp::gemm_gradient_workspace_bytes(a);
  return 0;
}

int mlbp_gradient_f64(const mlbp_gradient_args* a, void* stream) {
  if (!a) return fail(MLBP_EINVAL, "mlbp_gradient_f64: NULL args");
  if (a->B <= 0 || a->X <= 0 || a->n_msgs <= 0 || a->P < 0 || a->U < 0)
    return fail(MLBP_EINVAL, "mlbp_gradient_f64: bad sizes");
  if (!((a->F_ee == 3 && a->F_ed == 6) || (a->F_ee == 2 && a->F_ed == 2) || (a->F_ee == 1 && a->F_ed == 1)))
    return fail(MLBP_EUNSUPPORTED, "mlbp_gradient_f64: feature counts (%d, %d) not instantiated (3/6, 2/2, 1/1)", a->F_ee, a->F_ed);
  if (!a->msgs || !a->grad_en_en || !a->grad_en_de) return fail(MLBP_EINVAL, "mlbp_gradient_f64: NULL buffers");
  if (a->P > 0 && (!a->pair_tables || !a->pair_tab || !a->pair_c_slot || !a->pair_r_slot || !a->pair_phi || !a->pair_label ||
                   !a->phi_en_en || !a->phi_en_en_w1))
    return fail(MLBP_EINVAL, "mlbp_gradient_f64: pairwise inputs missing");
  if (a->U > 0 && (!a->unary_tables || !a->unary_tab || !a->unary_kind || !a->unary_obs || !a->unary_label ||
                   !a->phi_en_en || !a->phi_en_en_w1 || !a->phi_en_de || a->Vde <= 0))
    return fail(MLBP_EINVAL, "mlbp_gradient_f64: unary inputs missing");
  if (int e = need_device()) return e;
  GradDev d;
  d.a = *a;
  d.skip_pairs = 0;
  d.only = nullptr;
  if (int e = status_word(&d.status)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (a->X == 64 && (a->flags & MLBP_GRADIENT_APPROX_BELIEFS))
    return fail(MLBP_EINVAL, "mlbp_gradient_f64: approximate beliefs keep the %d largest entries; kth(=%d) out of bounds (64)", MLBP_APPROX_K, MLBP_APPROX_K - 1);
  if (a->X == 64) {
    // shared pairwise tables: the pairwise factors of 16 graphs at a time on the matrix cores, after the unary part
    const bool shared = takes_shared_x64(a);
    d.skip_pairs = shared ? 1 : 0;
    if (shared && a->unary_expect && a->F_ed == 6)               // unary part by gather inside the pair kernel
      return mlbp::launch_shared_pair_gradient(a, d.status, stream);
    if (a->F_ee == 3) hipLaunchKernelGGL((gradient_x64_kernel<3, 6>), dim3(a->B), dim3(WG), 0, st, d);
    else if (a->F_ee == 2) hipLaunchKernelGGL((gradient_x64_kernel<2, 2>), dim3(a->B), dim3(WG), 0, st, d);
    else hipLaunchKernelGGL((gradient_x64_kernel<1, 1>), dim3(a->B), dim3(WG), 0, st, d);
    HIP_TRY(hipGetLastError());
    if (shared) return mlbp::launch_shared_pair_gradient(a, d.status, stream);
    return MLBP_OK;
  }
  const bool approx = (a->flags & MLBP_GRADIENT_APPROX_BELIEFS) != 0;
  if (approx && a->X < MLBP_APPROX_K)
    return fail(MLBP_EINVAL, "mlbp_gradient_f64: approximate beliefs keep the %d largest entries; kth(=%d) out of bounds (%d)",
                MLBP_APPROX_K, MLBP_APPROX_K - 1, a->X);
  const size_t dyn = approx ? 2 * (size_t)a->X * sizeof(double) : 0;
  // shared pairwise tables at a large state space: pairwise part as MFMA contractions over the whole batch (mlbp_gemm.hip)
  const bool gemm_pairs = takes_gemm_pairs(a);
  d.skip_pairs = gemm_pairs ? 1 : 0;
  if (a->F_ee == 3) hipLaunchKernelGGL((gradient_kernel<3, 6>), dim3(a->B), dim3(WG), dyn, st, d);
  else if (a->F_ee == 2) hipLaunchKernelGGL((gradient_kernel<2, 2>), dim3(a->B), dim3(WG), dyn, st, d);
  else hipLaunchKernelGGL((gradient_kernel<1, 1>), dim3(a->B), dim3(WG), dyn, st, d);
  HIP_TRY(hipGetLastError());
  if (gemm_pairs) return mlbp::launch_gemm_pair_gradient(a, d.status, stream);
  return MLBP_OK;
}

int mlbp_unary_expectations_f64(const double* unary_tables, int32_t n_rows, int32_t X, const int32_t* row_kind,
                                const int32_t* row_obs, const double* phi_en_en_t, const double* phi_en_en_w1_t,
                                const double* phi_en_de_t, int32_t F_ee, int32_t F_ed, int32_t Vde, double* out,
                                void* stream) {
  if (!unary_tables || !row_kind || !row_obs || !phi_en_en_t || !phi_en_en_w1_t || !phi_en_de_t || !out || n_rows <= 0 ||
      F_ee <= 0 || F_ee > 8 || F_ed <= 0 || F_ed > 8 || Vde <= 0)
    return fail(MLBP_EINVAL, "mlbp_unary_expectations_f64: bad arguments");
  if (X != 64) return fail(MLBP_EUNSUPPORTED, "mlbp_unary_expectations_f64: X = %d (only 64)", X);
  if (int e = need_device()) return e;
  int32_t* status = nullptr;
  if (int e = status_word(&status)) return e;
  hipLaunchKernelGGL(unary_expectations_kernel, dim3((n_rows + 3) / 4), dim3(WG), 0, (hipStream_t)stream, unary_tables, n_rows,
                     row_kind, row_obs, phi_en_en_t, phi_en_en_w1_t, phi_en_de_t, F_ee, F_ed, Vde, out, status);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_pair_beliefs_f64(const double* msgs, int32_t B, int32_t n_msgs, int32_t X, int32_t P,
                          const double* pair_tables, const int32_t* pair_tab, int32_t n_pair_tables,
                          const int32_t* c_slot, const int32_t* r_slot, double* out, void* stream) {
  if (!msgs || !pair_tables || !pair_tab || !c_slot || !r_slot || !out || B <= 0 || P <= 0 || X <= 0 || n_msgs <= 0)
    return fail(MLBP_EINVAL, "mlbp_pair_beliefs_f64: bad arguments");
  if (int e = need_device()) return e;
  int32_t* status = nullptr;
  if (int e = status_word(&status)) return e;
  hipLaunchKernelGGL(pair_beliefs_kernel, dim3(B * P), dim3(WG), 0, (hipStream_t)stream, msgs, n_msgs, X, P, pair_tables,
                     pair_tab, n_pair_tables, c_slot, r_slot, out, status);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_sum_rows_cat_f64(const double* in0, int32_t cols0, const double* in1, int32_t cols1, const double* in2, int32_t cols2,
                          int64_t rows, int32_t append_count, double* out, void* stream) {
  if (!in0 || !out || rows <= 0 || cols0 <= 0 || cols1 < 0 || cols2 < 0 || (cols1 > 0 && !in1) || (cols2 > 0 && !in2))
    return fail(MLBP_EINVAL, "mlbp_sum_rows_cat_f64: bad arguments");
  if (int e = need_device()) return e;
  if ((int64_t)cols0 + cols1 + cols2 > 64)
    return fail(MLBP_EUNSUPPORTED, "mlbp_sum_rows: at most 64 columns (got %d)", cols0 + cols1 + cols2);
  return mlbp_select_sum_rows_cat_f64(in0, cols0, in1, cols1, in2, cols2, rows, nullptr, nullptr, append_count, out, stream);
}

int mlbp_select_sum_rows_cat_f64(const double* in0, int32_t cols0, const double* in1, int32_t cols1, const double* in2, int32_t cols2,
                                 int64_t rows, const int32_t* key, const int32_t* key_value, int32_t append_count, double* out, void* stream) {
  if (!in0 || !out || rows <= 0 || cols0 <= 0 || cols1 < 0 || cols2 < 0 || (cols1 > 0 && !in1) || (cols2 > 0 && !in2) || ((key != nullptr) != (key_value != nullptr)))
    return fail(MLBP_EINVAL, "mlbp_select_sum_rows_cat_f64: bad arguments");
  if (int e = need_device()) return e;
  if ((int64_t)cols0 + cols1 + cols2 + (append_count ? 1 : 0) > 64)
    return fail(MLBP_EUNSUPPORTED, "mlbp_sum_rows: at most 64 columns, the appended count included (got %d)", cols0 + cols1 + cols2);
  SumCat cat = {{in0, in1, in2}, {cols0, cols1, cols2}, nullptr, nullptr, 0, 0, nullptr, nullptr, key, key_value};
  // the partials live in one device-wide scratch array: launches on DIFFERENT streams must not overlap
  hipLaunchKernelGGL(sum_rows_kernel, dim3(SUM_PARTS), dim3(WG), 0, (hipStream_t)stream, cat, rows, append_count ? 1 : 0, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_step_statistics_f64(const double* grad_en_en, int32_t F_ee, const double* grad_en_de, int32_t F_ed, const double* marginals,
                             const int32_t* labels, int32_t n_vars, int32_t X, int64_t B, double* lp_out, double* out, void* stream) {
  if (!grad_en_en || !grad_en_de || !marginals || !labels || !out || B <= 0 || F_ee <= 0 || F_ed <= 0 || n_vars <= 0 || X <= 0)
    return fail(MLBP_EINVAL, "mlbp_step_statistics_f64: bad arguments");
  if (F_ee + F_ed + 1 > 64) return fail(MLBP_EUNSUPPORTED, "mlbp_step_statistics_f64: at most 63 feature columns (got %d)", F_ee + F_ed);
  if (int e = need_device()) return e;
  SumCat cat = {{grad_en_en, grad_en_de, marginals}, {F_ee, F_ed, 1}, marginals, labels, n_vars, X, lp_out, nullptr, nullptr, nullptr};
  if (int e = status_word(&cat.status)) return e;
  // (the partials live in one device-wide scratch array, as for mlbp_sum_rows_cat_f64)
  if (F_ee == 3 && F_ed == 6) hipLaunchKernelGGL((step_statistics_kernel<3, 6>), dim3(SUM_PARTS), dim3(WG), 0, (hipStream_t)stream, cat, B, out);
  else hipLaunchKernelGGL(sum_rows_kernel, dim3(SUM_PARTS), dim3(WG), 0, (hipStream_t)stream, cat, B, 1, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_sum_rows_f64(const double* in, int64_t rows, int32_t cols, double* out, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return fail(MLBP_EINVAL, "mlbp_sum_rows_f64: bad arguments");
  return mlbp_sum_rows_cat_f64(in, cols, nullptr, 0, nullptr, 0, rows, 0, out, stream);
}

int mlbp_segment_sum_rows_f64(const double* in, int64_t rows, int32_t cols, const int32_t* seg_id, int32_t n_seg, double* out,
                              void* stream) {
  if (!in || !out || !seg_id || rows <= 0 || cols <= 0 || n_seg <= 0)
    return fail(MLBP_EINVAL, "mlbp_segment_sum_rows_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(n_seg), dim3(WG), 0, (hipStream_t)stream, in, rows, cols, seg_id, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_patch_unary_tables_f64(const double* base_tables, const int32_t* base_row, const int32_t* item_off,
                                const int32_t* item_x, const int32_t* item_k, const double* item_val,
                                const double* theta, int32_t n_rows, int32_t X, double* out, void* stream) {
  if (!base_tables || !base_row || !item_off || !item_x || !item_k || !item_val || !theta || !out || n_rows <= 0 || X <= 0)
    return fail(MLBP_EINVAL, "mlbp_patch_unary_tables_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(patch_tables_kernel, dim3(n_rows), dim3(64), 0, (hipStream_t)stream, base_tables, base_row, item_off,
                     item_x, item_k, item_val, theta, X, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_patch_gradient_f64(const double* priv_tables, const int32_t* item_off, const int32_t* item_x,
                            const int32_t* item_k, const double* item_val, const int32_t* row_graph,
                            const int32_t* row_label, int32_t n_rows, int32_t X, int32_t F, double* grad, void* stream) {
  if (!priv_tables || !item_off || !item_x || !item_k || !item_val || !row_graph || !row_label || !grad || n_rows <= 0 ||
      X <= 0 || F <= 0)
    return fail(MLBP_EINVAL, "mlbp_patch_gradient_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(patch_gradient_kernel, dim3(n_rows), dim3(64), 0, (hipStream_t)stream, priv_tables, item_off, item_x,
                     item_k, item_val, row_graph, row_label, n_rows, X, F, grad);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_gradient_status(void) {
  // Synchronising read-and-reset: 1 when a gradient / belief kernel skipped a factor because a table,
  // label or observed-column index was out of range.
  if (!g_status) return 0;
  int32_t v = 0, zero = 0;
  HIP_TRY(hipMemcpy(&v, g_status, sizeof(v), hipMemcpyDeviceToHost));
  if (v) HIP_TRY(hipMemcpy(g_status, &zero, sizeof(zero), hipMemcpyHostToDevice));
  return v;
}

}  // extern "C"
