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

#include <cfloat>
#include <vector>

#include "mlbp_internal.h"

using mlbp::fail;

namespace {

constexpr int WG = 256;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(MLBP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

__device__ __forceinline__ double nan_to_num(double x) {
  // np.nan_to_num (LBP.py:729): NaN -> 0, +inf -> DBL_MAX, -inf -> -DBL_MAX
  if (x != x) return 0.0;
  if (x == __builtin_huge_val()) return DBL_MAX;
  if (x == -__builtin_huge_val()) return -DBL_MAX;
  return x;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Message.renormalize (LBP.py:649-657): positive total -> v / total, else uniform.
__device__ __forceinline__ double renorm(double v, double total, double uniform, bool normalize) {
  if (!normalize) return v;
  return total > 0.0 ? v / total : uniform;
}

struct SweepDev {
  const double* pair_tables;
  const int32_t* pair_tab;
  const double* unary_tables;
  const int32_t* unary_tab;
  double* msgs;
  const int32_t* ops;
  const int32_t* srcs;
  const int32_t* sweeps;
  const int32_t* pairseq;
  int32_t* status;
  int32_t n_sweeps, n_msgs, P, U, X, n_pair_tables, n_unary_tables;
};

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
// X = 64, float64: the BASELINE configs 2-4.
// thread t: row group rg = t >> 5 (0..7), column pair cp = t & 31 (columns 2cp, 2cp+1).
// Table element (as double2) k*256 + t  is row 8k+rg, columns 2cp..2cp+1.
// ------------------------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(WG) void sweep_x64_kernel(SweepDev d) {
  extern __shared__ double lds[];
  double* msg = lds;                       // [n_msgs][64]
  double* red = lds + (size_t)d.n_msgs * 64;  // [8][64] partial sums / raw results

  const int g = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int rg = t >> 5, cp = t & 31;
  if (!tables_in_range(d, g)) return;

  double* gm = d.msgs + (size_t)g * d.n_msgs * 64;
  {
    const double2* src = reinterpret_cast<const double2*>(gm);
    double2* dst = reinterpret_cast<double2*>(msg);
    for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = src[i];
  }
  const int32_t* ptab = d.pair_tab + (size_t)g * d.P;
  const int32_t* utab = d.unary_tab + (size_t)g * d.U;

  double2 nxt[8];
  int pair_k = 0;  // index into pairseq of the next pair op to execute
  {
    int s0 = d.pairseq[0];
    if (s0 >= 0) {
      const double2* T = reinterpret_cast<const double2*>(d.pair_tables + (size_t)ptab[s0] * 4096);
#pragma unroll
      for (int k = 0; k < 8; ++k) nxt[k] = T[k * WG + t];
    }
  }
  __syncthreads();

  const double uniform = 1.0 / 64.0;
  for (int s = 0; s < d.n_sweeps; ++s) {
    const int op0 = d.sweeps[2 * s], nop = d.sweeps[2 * s + 1];
    for (int o = op0; o < op0 + nop; ++o) {
      const int kind = d.ops[4 * o], a = d.ops[4 * o + 1], b = d.ops[4 * o + 2], c = d.ops[4 * o + 3];
      if (kind == MLBP_OP_PAIR_TM || kind == MLBP_OP_PAIR_MT) {
        double2 cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
        ++pair_k;
        {
          int sn = d.pairseq[pair_k];
          if (sn >= 0) {
            const double2* T = reinterpret_cast<const double2*>(d.pair_tables + (size_t)ptab[sn] * 4096);
#pragma unroll
            for (int k = 0; k < 8; ++k) nxt[k] = T[k * WG + t];
          }
        }
        const double* m = msg + b * 64;
        if (kind == MLBP_OP_PAIR_MT) {
          // out_j = sum_i m_i T[i][j]: accumulate my 8 rows, then add the 8 row groups through LDS
          double a0 = 0.0, a1 = 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            double mi = m[8 * k + rg];
            a0 += mi * cur[k].x;
            a1 += mi * cur[k].y;
          }
          reinterpret_cast<double2*>(red + rg * 64)[cp] = make_double2(a0, a1);
        } else {
          // out_i = sum_j T[i][j] m_j: 8 row partials per lane, transposing butterfly over the
          // 32 lanes that share a row group (halves the live values at each of the first 3 steps)
          const double2 mj = reinterpret_cast<const double2*>(m)[cp];
          double v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = cur[k].x * mj.x + cur[k].y * mj.y;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool up = lane & 16;
            double send = up ? v[i] : v[i + 4], keep = up ? v[i + 4] : v[i];
            v[i] = keep + __shfl_xor(send, 16, 64);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            bool up = lane & 8;
            double send = up ? v[i] : v[i + 2], keep = up ? v[i + 2] : v[i];
            v[i] = keep + __shfl_xor(send, 8, 64);
          }
          {
            bool up = lane & 4;
            double send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
            v[0] = keep + __shfl_xor(send, 4, 64);
          }
          v[0] += __shfl_xor(v[0], 2, 64);
          v[0] += __shfl_xor(v[0], 1, 64);
          if ((lane & 3) == 0) {
            int k = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            red[8 * k + rg] = v[0];
          }
        }
        __syncthreads();
        if (t < 64) {
          double r;
          if (kind == MLBP_OP_PAIR_MT) {
            r = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) r += red[q * 64 + t];
          } else {
            r = red[t];
          }
          msg[c * 64 + t] = renorm(r, wave_sum(r), uniform, NORM);
        }
        __syncthreads();
      } else if (kind == MLBP_OP_VAR) {
        if (t < 64) {
          double acc = uniform;
          for (int q = 0; q < b; ++q) acc = nan_to_num(msg[d.srcs[a + q] * 64 + t] * acc);
          msg[c * 64 + t] = renorm(acc, wave_sum(acc), uniform, NORM);
        }
        __syncthreads();
      } else {  // MLBP_OP_UNARY
        if (t < 64) {
          double r = d.unary_tables[(size_t)utab[a] * 64 + t];
          msg[c * 64 + t] = renorm(r, wave_sum(r), uniform, NORM);
        }
        __syncthreads();
      }
    }
  }
  {
    const double2* src = reinterpret_cast<const double2*>(msg);
    double2* dst = reinterpret_cast<double2*>(gm);
    for (int i = t; i < d.n_msgs * 32; i += WG) dst[i] = src[i];
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
  if (!tables_in_range(d, g)) return;
  double* gm = d.msgs + (size_t)g * d.n_msgs * X;
  double* msg = LDSMSG ? lmsg : gm;
  if (LDSMSG) {
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

__global__ void log_posterior_kernel(const double* marg, const int32_t* labels, int B, int n_vars, int X,
                                     double* out, int32_t* status) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B) return;
  double total = 0.0;
  for (int v = 0; v < n_vars; ++v) {
    int lab = labels[(size_t)g * n_vars + v];
    if ((unsigned)lab >= (unsigned)X) { atomicExch(status, 1); continue; }
    double lp = log(marg[((size_t)g * n_vars + v) * X + lab]);
    total += (lp == -__builtin_huge_val()) ? -99.99 : lp;  // LBP.py:254-256
  }
  out[g] = total;
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

int mlbp_program_create(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                        const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                        mlbp_program** out) {
  if (!out) return fail(MLBP_EINVAL, "out is NULL");
  *out = nullptr;
  if (!ops || !sweeps || n_ops <= 0 || n_sweeps <= 0 || n_msgs <= 0 || P < 0 || U < 0 || n_srcs < 0 ||
      (n_srcs > 0 && !srcs))
    return fail(MLBP_EINVAL, "mlbp_program_create: bad sizes or NULL arrays");
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
  std::vector<int32_t> pairseq;
  for (int s = 0; s < n_sweeps; ++s) {
    const int first = sweeps[2 * s], cnt = sweeps[2 * s + 1];
    if (first < 0 || cnt < 0 || (int64_t)first + cnt > n_ops)
      return fail(MLBP_EINVAL, "sweep %d: op range [%d,%d) out of [0,%d)", s, first, first + cnt, n_ops);
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
  (void)hipGetDevice(&p->device);
  auto up = [&](int32_t** dst, const int32_t* src, size_t n) -> hipError_t {
    hipError_t e = hipMalloc(dst, (n ? n : 1) * sizeof(int32_t));
    if (e != hipSuccess) return e;
    return n ? hipMemcpy(*dst, src, n * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t e = up(&p->d_ops, ops, (size_t)n_ops * 4);
  if (e == hipSuccess) e = up(&p->d_srcs, srcs, (size_t)n_srcs);
  if (e == hipSuccess) e = up(&p->d_sweeps, sweeps, (size_t)n_sweeps * 2);
  if (e == hipSuccess) e = up(&p->d_pairseq, pairseq.data(), pairseq.size());
  int32_t zero = 0;
  if (e == hipSuccess) e = up(&p->d_status, &zero, 1);
  if (e != hipSuccess) {
    mlbp_program_destroy(p);
    return fail(MLBP_EHIP, "mlbp_program_create: device upload failed: %s", hipGetErrorString(e));
  }
  *out = p;
  return MLBP_OK;
}

int mlbp_program_destroy(mlbp_program* p) {
  if (!p) return MLBP_OK;
  (void)hipFree(p->d_ops); (void)hipFree(p->d_srcs); (void)hipFree(p->d_sweeps);
  (void)hipFree(p->d_pairseq); (void)hipFree(p->d_status);
  delete p;
  return MLBP_OK;
}

int mlbp_sweep_f64(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream) {
  if (!prog || !a) return fail(MLBP_EINVAL, "mlbp_sweep_f64: NULL program or args");
  if (a->B <= 0 || a->X <= 0) return fail(MLBP_EINVAL, "mlbp_sweep_f64: B=%d X=%d", a->B, a->X);
  if (!a->msgs) return fail(MLBP_EINVAL, "mlbp_sweep_f64: msgs is NULL");
  if (prog->P > 0 && (!a->pair_tables || !a->pair_tab || a->n_pair_tables <= 0))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: program has %d pairwise factors but no pair tables", prog->P);
  if (prog->U > 0 && (!a->unary_tables || !a->unary_tab || a->n_unary_tables <= 0))
    return fail(MLBP_EINVAL, "mlbp_sweep_f64: program has %d unary factors but no unary tables", prog->U);
  if (a->X > 4096) return fail(MLBP_EUNSUPPORTED, "mlbp_sweep_f64: X=%d > 4096", a->X);
  if (int e = check_device()) return e;
  SweepDev d;
  d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab;
  d.unary_tables = a->unary_tables; d.unary_tab = a->unary_tab;
  d.msgs = a->msgs;
  d.ops = prog->d_ops; d.srcs = prog->d_srcs; d.sweeps = prog->d_sweeps; d.pairseq = prog->d_pairseq;
  d.status = prog->d_status;
  d.n_sweeps = prog->n_sweeps; d.n_msgs = prog->n_msgs; d.P = prog->P; d.U = prog->U; d.X = a->X;
  d.n_pair_tables = a->n_pair_tables; d.n_unary_tables = a->n_unary_tables;
  hipStream_t st = (hipStream_t)stream;
  const bool norm = a->normalize_messages != 0;
  const size_t LDS_MAX = 160 * 1024;
  if (a->X == 64) {
    size_t lds = ((size_t)prog->n_msgs * 64 + 8 * 64) * sizeof(double);
    if (lds <= LDS_MAX) {
      auto k = norm ? sweep_x64_kernel<true> : sweep_x64_kernel<false>;
      HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k, dim3(a->B), dim3(WG), lds, st, d);
      HIP_TRY(hipGetLastError());
      return MLBP_OK;
    }
  }
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
  return MLBP_OK;
}

int mlbp_program_status(const mlbp_program* prog) {
  // Synchronising read of the status word: 0 = clean, 1 = a kernel skipped a graph because a
  // table index was out of range.  Resets the word.
  if (!prog) return fail(MLBP_EINVAL, "NULL program");
  int32_t v = 0, zero = 0;
  HIP_TRY(hipMemcpy(&v, prog->d_status, sizeof(v), hipMemcpyDeviceToHost));
  if (v) HIP_TRY(hipMemcpy(prog->d_status, &zero, sizeof(zero), hipMemcpyHostToDevice));
  return v;
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

int mlbp_log_posterior_f64(const double* marginals, const int32_t* labels, int32_t B, int32_t n_vars,
                           int32_t X, double* out, void* stream) {
  if (!marginals || !labels || !out || B <= 0 || n_vars <= 0 || X <= 0)
    return fail(MLBP_EINVAL, "mlbp_log_posterior_f64: bad arguments");
  if (int e = check_device()) return e;
  int32_t* status = nullptr;
  if (int e = global_status(&status)) return e;
  hipLaunchKernelGGL(log_posterior_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, marginals,
                     labels, B, n_vars, X, out, status);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

}  // extern "C"
