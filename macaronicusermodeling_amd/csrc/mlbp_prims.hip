// Array primitives of the `array_utils.c_array_utils` surface as batched gfx950 kernels.
// These back the drop-in `au.*` functions one call at a time (the fused sweep in mlbp_sweep.hip is
// the performance path); they are written for correctness on arbitrary strides first.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cfloat>

#include "mlbp_internal.h"

using mlbp::fail;

namespace {

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(MLBP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

int need_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MLBP_ENODEVICE, "no HIP device visible: libmlbp.so has no CPU fallback");
  }
  return MLBP_OK;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// C[b][i][j] = sum_k A[b][i][k] * B[b][k][j].  One wavefront per output element: lanes stride over
// k (coalesced when the contraction axis is the contiguous one, as in T.m), then a wave reduction.
__global__ __launch_bounds__(256) void dense_dot_kernel(int batch, int M, int K, int N, const double* A,
                                                        int64_t ab, int64_t ar, int64_t ac, const double* B,
                                                        int64_t bb, int64_t br, int64_t bc, double* C,
                                                        int64_t cb, int64_t cr) {
  const int lane = threadIdx.x & 63;
  const int64_t n_out = (int64_t)batch * M * N;
  for (int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); o < n_out; o += (int64_t)gridDim.x * 4) {
    const int j = (int)(o % N);
    const int i = (int)((o / N) % M);
    const int b = (int)(o / ((int64_t)M * N));
    const double* a = A + b * ab + i * ar;
    const double* bp = B + b * bb + j * bc;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) acc += a[k * ac] * bp[k * br];
    acc = wave_sum(acc);
    if (lane == 0) C[b * cb + i * cr + j] = acc;
  }
}

// K == 1 (outer product c . r, LBP.py:566) and other thin contractions: one thread per output.
__global__ void dense_dot_thin_kernel(int batch, int M, int K, int N, const double* A, int64_t ab, int64_t ar,
                                      int64_t ac, const double* B, int64_t bb, int64_t br, int64_t bc,
                                      double* C, int64_t cb, int64_t cr) {
  const int64_t n_out = (int64_t)batch * M * N;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(o % N);
    const int i = (int)((o / N) % M);
    const int b = (int)(o / ((int64_t)M * N));
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += A[b * ab + i * ar + k * ac] * B[b * bb + k * br + j * bc];
    C[b * cb + i * cr + j] = acc;
  }
}

__global__ void pointwise_multiply_kernel(const double* a, const double* b, double* out, int64_t n, int nan2num) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double x = a[i] * b[i];
    if (nan2num) {
      if (x != x) x = 0.0;
      else if (x == __builtin_huge_val()) x = DBL_MAX;
      else if (x == -__builtin_huge_val()) x = -DBL_MAX;
    }
    out[i] = x;
  }
}

// One workgroup per vector.
__global__ __launch_bounds__(256) void normalize_kernel(const double* in, double* out, int64_t n, int mode,
                                                        int32_t* positive) {
  __shared__ double part[4];
  const double* x = in + (int64_t)blockIdx.x * n;
  double* y = out + (int64_t)blockIdx.x * n;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  const double total = part[0] + part[1] + part[2] + part[3];
  const bool pos = total > 0.0;
  const double fill = mode == MLBP_NORM_UNIFORM ? 1.0 / (double)n : 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) y[i] = pos ? x[i] / total : fill;
  if (positive && threadIdx.x == 0) positive[blockIdx.x] = pos ? 1 : 0;
}

// Rank-select of the K largest entries of v[0..n) (ties: lower index first).  One workgroup;
// rank_i = #{j : v_j > v_i or (v_j == v_i and j < i)}; entries with rank < K are written to
// idx[rank], i.e. in descending value order.  O(n^2 / 256) compares per thread -- n is a vocabulary
// size (hundreds to a few thousand).  NaNs compare false everywhere and rank first among equals;
// the reference's np.argpartition leaves their place unspecified.
__global__ __launch_bounds__(256) void topk_kernel(const double* v0, int64_t stride, int n, int K, int32_t* idx0,
                                                   int64_t row_stride) {
  extern __shared__ double sv[];
  const double* v = v0 + (int64_t)blockIdx.x * row_stride;      // one workgroup per row
  int32_t* idx = idx0 + (int64_t)blockIdx.x * K;
  for (int i = threadIdx.x; i < n; i += 256) sv[i] = v[(int64_t)i * stride];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const double x = sv[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double y = sv[j];
      rank += (y > x) || (y == x && j < i);
    }
    if (rank < K) idx[rank] = i;
  }
}

// au.sparse_vec_mat_dot (c_array_utils.pyx:193-205) after the top-K selection:
//   vec_is_row = 0: out_i = sum_{k in idx} mat[i][k] * vec[k]      (mat[:, idx] . vec[idx])
//   vec_is_row = 1: out_j = sum_{k in idx} vec[k] * mat[k][j]      (vec[0, idx] . mat[idx, :])
__global__ void gather_dot_kernel(const double* vec, int64_t vstride, const double* mat, int64_t mrow, int64_t mcol,
                                  int n_out, const int32_t* idx, int K, int vec_is_row, double* out) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  double acc = 0.0;
  for (int q = 0; q < K; ++q) {
    const int k = idx[q];
    const double m = vec_is_row ? mat[(int64_t)k * mrow + (int64_t)o * mcol] : mat[(int64_t)o * mrow + (int64_t)k * mcol];
    acc += vec_is_row ? vec[(int64_t)k * vstride] * m : m * vec[(int64_t)k * vstride];
  }
  out[o] = acc;
}

__global__ void zero_kernel(double* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

// Block (cidx x ridx) operations on an n_rows x n_cols row-major matrix.
//   mode 0: out[ci][rj] = c[ci] * r[rj]                 (au.sparse_dot, pyx:128)
//   mode 1: out[ci][rj] = a[ci][rj] * b[ci][rj]         (au.sparse_pointwise_multiply, pyx:113)
//   mode 2: out[ci][rj] = a[ci][rj] / *total            (au.sparse_normalize, pyx:25)
__global__ void block_op_kernel(int mode, const double* a, const double* b, const double* total, int n_cols,
                                const int32_t* cidx, int Kc, const int32_t* ridx, int Kr, double* out) {
  const int n = Kc * Kr;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int i = cidx[e / Kr], j = ridx[e % Kr];
    const int64_t at = (int64_t)i * n_cols + j;
    double v;
    if (mode == 0) v = a[i] * b[j];
    else if (mode == 1) v = a[at] * b[at];
    else v = a[at] / *total;
    out[at] = v;
  }
}

// total = sum of the (cidx x ridx) block; one workgroup (au.sparse_normalize, pyx:24).
__global__ __launch_bounds__(256) void block_sum_kernel(const double* a, int n_cols, const int32_t* cidx, int Kc,
                                                        const int32_t* ridx, int Kr, double* total) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int e = threadIdx.x; e < Kc * Kr; e += 256) acc += a[(int64_t)cidx[e / Kr] * n_cols + ridx[e % Kr]];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *total = part[0] + part[1] + part[2] + part[3];
}

// pot[i][j] = exp(sum_k phi[i][j][k] * theta[k])  (train_mp.py:220-255), written in both layouts:
// row-major [rows][cols] for pairwise tables and transposed [cols][rows] so that a unary factor's
// table (a COLUMN of the pot, LBP.py:702-703) is one contiguous row for the sweep kernel.
__global__ void potentials_kernel(const double* phi, const double* theta, int rows, int cols, int F, double* pot,
                                  double* pot_t) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < F; ++k) acc += phi[e * F + k] * theta[k];
    const double v = exp(acc);
    if (pot) pot[e] = v;
    if (pot_t) {
      const int64_t i = e / cols, j = e - i * cols;
      pot_t[j * rows + i] = v;
    }
  }
}

struct PotentialsJobs { mlbp_potentials_job j[MLBP_POTENTIALS_MAX_JOBS]; };
// blockIdx.y = job, blockIdx.z = repetition (parameter vector); the arithmetic of potentials_kernel
__global__ void potentials_multi_kernel(PotentialsJobs js) {
  const mlbp_potentials_job& J = js.j[blockIdx.y];
  const int64_t n = (int64_t)J.rows * J.cols;
  const double* theta = J.theta + (int64_t)blockIdx.z * J.theta_stride;
  double* pot = J.pot ? J.pot + (int64_t)blockIdx.z * J.pot_stride : nullptr;
  double* pot_t = J.pot_t ? J.pot_t + (int64_t)blockIdx.z * J.pot_t_stride : nullptr;
  if (J.expect && J.rows == 64) {
    // one wave per column j, lane = row i: the column is a unary factor's table (LBP.py:702-703), so its total and its expected
    // features come out of the same registers (unary_expectations_kernel's arithmetic: wave sums of v and v * phi_k)
    double* ex = J.expect + (int64_t)blockIdx.z * J.expect_stride;
    const int lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < J.cols; j += gridDim.x * 4) {
      const int64_t e = (int64_t)lane * J.cols + j;
      const double* ph = J.phi + e * J.F;
      double acc = 0.0;
      for (int k = 0; k < J.F; ++k) acc += ph[k] * theta[k];
      const double v = exp(acc);
      if (pot) pot[e] = v;
      if (pot_t) pot_t[(int64_t)j * 64 + lane] = v;
      const double Z = wave_sum(v);
      for (int k = 0; k < J.F && k < 8; ++k) {          // (the features are read again, from L1: kept in an array indexed at run time they live in scratch)
        const double s = wave_sum(v * ph[k]);
        if (lane == 0) ex[(int64_t)j * 8 + k] = Z > 0.0 ? s / Z : 0.0;          // au.normalize: zero-sum -> 0
      }
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < J.F; ++k) acc += J.phi[e * J.F + k] * theta[k];
    const double v = exp(acc);
    if (pot) pot[e] = v;
    if (pot_t) {
      const int64_t i = e / J.cols, j = e - i * J.cols;
      pot_t[j * J.rows + i] = v;
    }
  }
}

__global__ void log_kernel(const double* in, double* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = log(in[i]);
}

// out = onehot(cell) - beliefs  (FactorNode.cell_gradient, LBP.py:615-619).
__global__ void observed_minus_kernel(const double* beliefs, int64_t n, int64_t cell, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (i == cell ? 1.0 : 0.0) - beliefs[i];
}

}  // namespace

extern "C" {

int mlbp_dense_dot_f64(int32_t batch, int32_t M, int32_t K, int32_t N, const double* A, int64_t a_batch,
                       int64_t a_row, int64_t a_col, const double* B, int64_t b_batch, int64_t b_row,
                       int64_t b_col, double* C, int64_t c_batch, int64_t c_row, void* stream) {
  if (!A || !B || !C || batch <= 0 || M <= 0 || K <= 0 || N <= 0)
    return fail(MLBP_EINVAL, "mlbp_dense_dot_f64: bad arguments (batch=%d M=%d K=%d N=%d)", batch, M, K, N);
  if (int e = need_device()) return e;
  const int64_t n_out = (int64_t)batch * M * N;
  if (K < 32) {
    int blocks = (int)((n_out + 255) / 256 < 8192 ? (n_out + 255) / 256 : 8192);
    hipLaunchKernelGGL(dense_dot_thin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, batch, M, K, N, A,
                       a_batch, a_row, a_col, B, b_batch, b_row, b_col, C, c_batch, c_row);
  } else {
    int blocks = (int)((n_out + 3) / 4 < 16384 ? (n_out + 3) / 4 : 16384);
    hipLaunchKernelGGL(dense_dot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, batch, M, K, N, A, a_batch,
                       a_row, a_col, B, b_batch, b_row, b_col, C, c_batch, c_row);
  }
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_pointwise_multiply_f64(const double* a, const double* b, double* out, int64_t n, int32_t nan_to_num,
                                void* stream) {
  if (!a || !b || !out || n <= 0) return fail(MLBP_EINVAL, "mlbp_pointwise_multiply_f64: bad arguments");
  if (int e = need_device()) return e;
  int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(pointwise_multiply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n,
                     nan_to_num);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_normalize_f64(const double* in, double* out, int32_t batch, int64_t n, int32_t mode, int32_t* positive,
                       void* stream) {
  if (!in || !out || batch <= 0 || n <= 0 || (mode != MLBP_NORM_ZERO && mode != MLBP_NORM_UNIFORM))
    return fail(MLBP_EINVAL, "mlbp_normalize_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(normalize_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, in, out, n, mode, positive);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_topk_f64(const double* v, int64_t stride, int32_t n, int32_t K, int32_t* idx, void* stream) {
  if (!v || !idx || n <= 0 || K <= 0) return fail(MLBP_EINVAL, "mlbp_topk_f64: bad arguments");
  if (K > n) return fail(MLBP_EINVAL, "kth(=%d) out of bounds (%d)", K - 1, n);   // np.argpartition's ValueError text
  if (n > 16384) return fail(MLBP_EUNSUPPORTED, "mlbp_topk_f64: n=%d > 16384", n);
  if (int e = need_device()) return e;
  size_t lds = (size_t)n * sizeof(double);
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, v, stride, n, K, idx, (int64_t)0);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_topk_rows_f64(const double* v, int64_t rows, int32_t n, int32_t K, int32_t* idx, void* stream) {
  if (!v || !idx || rows <= 0 || n <= 0 || K <= 0) return fail(MLBP_EINVAL, "mlbp_topk_rows_f64: bad arguments");
  if (K > n) return fail(MLBP_EINVAL, "kth(=%d) out of bounds (%d)", n - K, n);
  if (n > 16384) return fail(MLBP_EUNSUPPORTED, "mlbp_topk_rows_f64: n=%d > 16384", n);
  if (rows > 0x7fffffff) return fail(MLBP_EINVAL, "mlbp_topk_rows_f64: too many rows");
  if (int e = need_device()) return e;
  size_t lds = (size_t)n * sizeof(double);
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(topk_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, v, (int64_t)1, n, K, idx, (int64_t)n);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_sparse_vec_mat_dot_f64(const double* vec, int64_t vstride, const double* mat, int64_t m_row, int64_t m_col,
                                int32_t n_out, const int32_t* idx, int32_t K, int32_t vec_is_row, double* out,
                                void* stream) {
  if (!vec || !mat || !idx || !out || n_out <= 0 || K <= 0)
    return fail(MLBP_EINVAL, "mlbp_sparse_vec_mat_dot_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(gather_dot_kernel, dim3((n_out + 127) / 128), dim3(128), 0, (hipStream_t)stream, vec, vstride, mat,
                     m_row, m_col, n_out, idx, K, vec_is_row, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

static int launch_block_op(int mode, const double* a, const double* b, const double* total, int n_cols,
                           const int32_t* cidx, int Kc, const int32_t* ridx, int Kr, double* out, void* stream) {
  int n = Kc * Kr;
  int blocks = (n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096;
  hipLaunchKernelGGL(block_op_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, a, b, total, n_cols, cidx,
                     Kc, ridx, Kr, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

static int launch_zero(double* p, int64_t n, void* stream) {
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_sparse_dot_f64(const double* c, const double* r, int32_t n, const int32_t* cidx, const int32_t* ridx,
                        int32_t K, double* out, void* stream) {
  if (!c || !r || !cidx || !ridx || !out || n <= 0 || K <= 0) return fail(MLBP_EINVAL, "mlbp_sparse_dot_f64: bad arguments");
  if (int e = need_device()) return e;
  if (int e = launch_zero(out, (int64_t)n * n, stream)) return e;
  return launch_block_op(0, c, r, nullptr, n, cidx, K, ridx, K, out, stream);
}

int mlbp_sparse_pointwise_multiply_f64(const double* sparse_m, const double* dense_m, int32_t n_rows, int32_t n_cols,
                                       const int32_t* cidx, int32_t Kc, const int32_t* ridx, int32_t Kr, double* out,
                                       void* stream) {
  if (!sparse_m || !dense_m || !cidx || !ridx || !out || n_rows <= 0 || n_cols <= 0 || Kc <= 0 || Kr <= 0)
    return fail(MLBP_EINVAL, "mlbp_sparse_pointwise_multiply_f64: bad arguments");
  if (int e = need_device()) return e;
  if (int e = launch_zero(out, (int64_t)n_rows * n_cols, stream)) return e;
  return launch_block_op(1, sparse_m, dense_m, nullptr, n_cols, cidx, Kc, ridx, Kr, out, stream);
}

int mlbp_sparse_normalize_f64(double* m, int32_t n_cols, const int32_t* cidx, int32_t Kc, const int32_t* ridx,
                              int32_t Kr, double* scratch1, void* stream) {
  if (!m || !cidx || !ridx || !scratch1 || n_cols <= 0 || Kc <= 0 || Kr <= 0)
    return fail(MLBP_EINVAL, "mlbp_sparse_normalize_f64: bad arguments");
  if (int e = need_device()) return e;
  hipLaunchKernelGGL(block_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, m, n_cols, cidx, Kc, ridx, Kr, scratch1);
  HIP_TRY(hipGetLastError());
  return launch_block_op(2, m, nullptr, scratch1, n_cols, cidx, Kc, ridx, Kr, m, stream);
}

int mlbp_potentials_multi_f64(const mlbp_potentials_job* jobs, int32_t n_jobs, int32_t n_rep, void* stream) {
  if (!jobs || n_jobs < 1 || n_jobs > MLBP_POTENTIALS_MAX_JOBS || n_rep < 1 || n_rep > 65535)
    return fail(MLBP_EINVAL, "mlbp_potentials_multi_f64: 1..%d jobs, 1..65535 repetitions", MLBP_POTENTIALS_MAX_JOBS);
  PotentialsJobs js;
  int64_t n_max = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const mlbp_potentials_job& J = jobs[j];
    if (!J.phi || !J.theta || J.rows <= 0 || J.cols <= 0 || J.F <= 0 || (!J.pot && !J.pot_t))
      return fail(MLBP_EINVAL, "mlbp_potentials_multi_f64: bad job %d", j);
    if (J.expect && (J.rows != 64 || J.F > 8))
      return fail(MLBP_EUNSUPPORTED, "mlbp_potentials_multi_f64: job %d: expectations need 64 rows and at most 8 features", j);
    js.j[j] = J;
    n_max = std::max<int64_t>(n_max, (int64_t)J.rows * J.cols);
  }
  for (int j = n_jobs; j < MLBP_POTENTIALS_MAX_JOBS; ++j) js.j[j] = jobs[0];
  if (int e = need_device()) return e;
  const int blocks = (int)std::min<int64_t>((n_max + 255) / 256, 4096);
  hipLaunchKernelGGL(potentials_multi_kernel, dim3(blocks, n_jobs, n_rep), dim3(256), 0, (hipStream_t)stream, js);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_potentials_f64(const double* phi, const double* theta, int32_t rows, int32_t cols, int32_t F, double* pot,
                        double* pot_t, void* stream) {
  if (!phi || !theta || rows <= 0 || cols <= 0 || F <= 0 || (!pot && !pot_t))
    return fail(MLBP_EINVAL, "mlbp_potentials_f64: bad arguments");
  if (int e = need_device()) return e;
  const int64_t n = (int64_t)rows * cols;
  int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(potentials_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, phi, theta, rows, cols, F, pot,
                     pot_t);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_log_f64(const double* in, double* out, int64_t n, void* stream) {
  if (!in || !out || n <= 0) return fail(MLBP_EINVAL, "mlbp_log_f64: bad arguments");
  if (int e = need_device()) return e;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(log_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

int mlbp_observed_minus_f64(const double* beliefs, int64_t n, int64_t cell, double* out, void* stream) {
  if (!beliefs || !out || n <= 0 || cell < 0 || cell >= n) return fail(MLBP_EINVAL, "mlbp_observed_minus_f64: bad arguments");
  if (int e = need_device()) return e;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(observed_minus_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, beliefs, n, cell, out);
  HIP_TRY(hipGetLastError());
  return MLBP_OK;
}

}  // extern "C"
