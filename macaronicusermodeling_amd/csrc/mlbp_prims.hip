// Array primitives of the `array_utils.c_array_utils` surface as batched gfx950 kernels.
// These back the drop-in `au.*` functions one call at a time (the fused sweep in mlbp_sweep.hip is
// the performance path); they are written for correctness on arbitrary strides first.
#include <hip/hip_runtime.h>

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

}  // extern "C"
