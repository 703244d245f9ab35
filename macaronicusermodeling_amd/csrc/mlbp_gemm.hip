// Shared pairwise tables at large state spaces (X >= 128): update by update over the whole batch.
//
// With one table behind factor p for every graph (the reference's layout, LBP.py:456-467; X = |V_en| there) the
// factor->variable update of ALL B graphs is one plain dense product
//     OUT[X x B] = T[X x X] . M[X x B]        (or T^T . M)
// -- SURVEY.md section 8(d)'s "shared-table (GEMM/MFMA) variant" of config 5.  At X = 64 the whole sweep fits one
// workgroup per 16 graphs (mlbp_shared.hip); at X = 512 a table is 2 MiB and a message tile of 16 graphs 64 KiB,
// so the sweep is run op by op instead: the contraction is a library DGEMM straight on the strided message
// buffer (rocBLAS; column-major views of msgs[:, slot, :] with leading dimension n_msgs * X, no copies), the rest
// -- Message.renormalize, the variable->factor products with nan_to_num, the unary messages -- are the small
// kernels below.  Same updates in the same order as every other path; only the summation order inside the
// contraction differs.
//
// rocBLAS is bound at first use with dlopen (no link-time dependency: the library and every other path work
// without it; this path then returns MLBP_EUNSUPPORTED).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cfloat>
#include <mutex>
#include <vector>

#include "mlbp_internal.h"

namespace mlbp {
namespace {

constexpr int WG = 256;

// ---- the four rocBLAS entry points this path needs (ABI as in rocblas/internal/rocblas-functions.h) ----
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_dgemm_t)(rb_handle, int, int, int, int, int, const double*, const double*, int, const double*, int,
                          const double*, double*, int);
constexpr int RB_OP_NONE = 111, RB_OP_TRANSPOSE = 112;      // rocblas_operation_none / _transpose

struct RocBlas {
  rb_create_t create = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_dgemm_t dgemm = nullptr;
  rb_handle handle = nullptr;
  bool tried = false, ok = false;
};
RocBlas g_rb;
std::mutex g_rb_mutex;

int rocblas_ready() {
  std::lock_guard<std::mutex> lock(g_rb_mutex);
  if (!g_rb.tried) {
    g_rb.tried = true;
    void* h = nullptr;
    for (const char* name : {"librocblas.so", "librocblas.so.5", "librocblas.so.4"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (h) {
      g_rb.create = (rb_create_t)dlsym(h, "rocblas_create_handle");
      g_rb.set_stream = (rb_set_stream_t)dlsym(h, "rocblas_set_stream");
      g_rb.dgemm = (rb_dgemm_t)dlsym(h, "rocblas_dgemm");
      g_rb.ok = g_rb.create && g_rb.set_stream && g_rb.dgemm && g_rb.create(&g_rb.handle) == 0;
    }
  }
  return g_rb.ok ? MLBP_OK : fail(MLBP_EUNSUPPORTED, "shared-table GEMM path: rocBLAS could not be loaded");
}

__device__ __forceinline__ double nan_to_num(double x) {
  if (x != x) return 0.0;
  if (x == __builtin_huge_val()) return DBL_MAX;
  if (x == -__builtin_huge_val()) return -DBL_MAX;
  return x;
}

__device__ __forceinline__ double block_sum(double v, double* scratch /*[4]*/) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// Message.renormalize (LBP.py:649-657) of slot `c` of every graph, in place: total > 0 -> v / total, else uniform.
__global__ __launch_bounds__(WG) void renormalize_slot_kernel(double* msgs, int n_msgs, int X, int c) {
  __shared__ double scratch[4];
  double* m = msgs + ((size_t)blockIdx.x * n_msgs + c) * X;
  double part = 0.0;
  for (int j = threadIdx.x; j < X; j += WG) part += m[j];
  const double total = block_sum(part, scratch);
  const double uniform = 1.0 / (double)X;
  for (int j = threadIdx.x; j < X; j += WG) m[j] = total > 0.0 ? m[j] / total : uniform;
}

// VariableNode.update_message_to (LBP.py:377-389): uniform times the listed incoming messages, nan_to_num after
// each product, renormalised when asked.
__global__ __launch_bounds__(WG) void variable_update_kernel(double* msgs, int n_msgs, int X, const int32_t* srcs, int a, int b,
                                                            int c, int normalize) {
  __shared__ double scratch[4];
  extern __shared__ double raw[];
  double* gm = msgs + (size_t)blockIdx.x * n_msgs * X;
  const double uniform = 1.0 / (double)X;
  double part = 0.0;
  for (int j = threadIdx.x; j < X; j += WG) {
    double acc = uniform;
    for (int q = 0; q < b; ++q) acc = nan_to_num(gm[(size_t)srcs[a + q] * X + j] * acc);
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

// ---- pairwise part of the gradient (LBP.py:528-619, 301-320) as DGEMMs ----
// W = T (.) phi_k (k = 0..2) or T (k = 3), row-major like T
__global__ void weight_table_kernel(const double* T, const double* phi_plane, int n, double* W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) W[i] = phi_plane ? T[i] * phi_plane[i] : T[i];
}
// S[k][b] = sum_i c[b][i] * Y[i][b]   (Y column-major X x B: Y[b * X + i])
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

}  // namespace

int gemm_path_ready() { return rocblas_ready(); }

int launch_gemm_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream) {
  if (int e = rocblas_ready()) return e;
  hipStream_t st = (hipStream_t)stream;
  const int B = a->B, X = a->X, ld = a->n_msgs * X;
  // scratch: W [X][X], Y [B][X] (column-major X x B), S [4][B]; one device-wide buffer, grown on demand (launches on
  // different streams must not overlap; a stream-capturing caller runs one eager step first)
  static double* scratch = nullptr;
  static size_t cap = 0;
  std::lock_guard<std::mutex> lock(g_rb_mutex);
  const size_t need = (size_t)X * X + (size_t)B * X + 4 * (size_t)B;
  if (need > cap) {
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr; cap = 0;
    if (hipMalloc(&scratch, need * sizeof(double)) != hipSuccess) return fail(MLBP_EHIP, "gradient GEMM scratch allocation failed");
    cap = need;
  }
  double* W = scratch; double* Y = W + (size_t)X * X; double* S = Y + (size_t)B * X;
  if (g_rb.set_stream(g_rb.handle, st) != 0) return fail(MLBP_EHIP, "rocblas_set_stream failed");
  const double one = 1.0, zero = 0.0;
  // slots come from DEVICE arrays in the ABI (pair_c_slot / pair_r_slot / pair_phi): fetch the few ints once
  int32_t h_c[16], h_r[16], h_phi[16];
  if (hipMemcpyAsync(h_c, a->pair_c_slot, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(h_r, a->pair_r_slot, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(h_phi, a->pair_phi, sizeof(int32_t) * a->P, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return fail(MLBP_EHIP, "gradient GEMM path: reading the slot tables failed");
  for (int p = 0; p < a->P; ++p) {
    if ((unsigned)h_c[p] >= (unsigned)a->n_msgs || (unsigned)h_r[p] >= (unsigned)a->n_msgs ||
        (unsigned)a->pair_tab_host[p] >= (unsigned)a->n_pair_tables)
      return fail(MLBP_EINVAL, "gradient GEMM path: slot or table index of factor %d out of range", p);
    const double* T = a->pair_tables + (size_t)a->pair_tab_host[p] * X * X;
    const double* planes = h_phi[p] ? a->phi_en_en_w1_p : a->phi_en_en_p;
    for (int k = 0; k < 4; ++k) {
      const double* A = T;
      if (k < 3) {
        hipLaunchKernelGGL(weight_table_kernel, dim3((X * X + 255) / 256), dim3(256), 0, st, T, planes + (size_t)k * X * X, X * X, W);
        A = W;
      }
      // Y[:, b] = A . r_b  (A row-major = A^T column-major -> op transpose), r = msgs[:, r_slot, :]
      const int rc = g_rb.dgemm(g_rb.handle, RB_OP_TRANSPOSE, RB_OP_NONE, X, B, X, &one, A, X, a->msgs + (size_t)h_r[p] * X, ld, &zero, Y, X);
      if (rc != 0) return fail(MLBP_EHIP, "rocblas_dgemm failed with status %d", rc);
      hipLaunchKernelGGL(row_dot_kernel, dim3(B), dim3(WG), 0, st, a->msgs, a->n_msgs, X, h_c[p], Y, S + (size_t)k * B);
    }
    hipLaunchKernelGGL(pair_gradient_combine_kernel, dim3((B + 255) / 256), dim3(256), 0, st, S, B, X, a->pair_label, a->P, p,
                       h_phi[p] ? a->phi_en_en_w1 : a->phi_en_en, a->grad_en_en, status);
  }
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "gradient GEMM path: a launch failed");
  return MLBP_OK;
}

int launch_gemm_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream) {
  if (!a->pair_tab_host) return fail(MLBP_EINVAL, "shared-table GEMM path: pair_tab_host (host int32 [P]) is required");
  if (prog->P > 16) return fail(MLBP_EUNSUPPORTED, "shared-table GEMM path: at most 16 pairwise factors (got %d)", prog->P);
  for (int p = 0; p < prog->P; ++p)
    if ((unsigned)a->pair_tab_host[p] >= (unsigned)a->n_pair_tables)
      return fail(MLBP_EINVAL, "pair_tab_host[%d] = %d out of [0,%d)", p, a->pair_tab_host[p], a->n_pair_tables);
  if (int e = rocblas_ready()) return e;
  hipStream_t st = (hipStream_t)stream;
  const int B = a->B, X = a->X, n_msgs = prog->n_msgs, norm = a->normalize_messages ? 1 : 0;
  const int ld = n_msgs * X;
  {
    HostRow row = {};
    for (int p = 0; p < prog->P; ++p) row.v[p] = a->pair_tab_host[p];
    hipLaunchKernelGGL(check_shared_claim_kernel, dim3((B * prog->P + 255) / 256), dim3(256), 0, st, a->pair_tab, B, prog->P,
                       a->n_pair_tables, row, prog->d_status);
  }
  if (a->init_messages)
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(1024), dim3(256), 0, st, a->msgs, (size_t)B * n_msgs * X, 1.0 / (double)X);
  std::lock_guard<std::mutex> lock(g_rb_mutex);          // one handle: its stream is set per launch sequence
  if (g_rb.set_stream(g_rb.handle, st) != 0) return fail(MLBP_EHIP, "rocblas_set_stream failed");
  const double one = 1.0, zero = 0.0;
  // unary messages are constants (LBP.py:494-498): when nothing else ever writes their slots (the same test that
  // lets the X = 64 kernels hoist them) each is computed once per call, not once per sweep
  const bool unary_once = prog->n_hoist == prog->U;
  std::vector<char> unary_done(prog->n_msgs, 0);
  for (int s = 0; s < prog->n_sweeps; ++s) {
    const int first = prog->h_sweeps[2 * s], cnt = prog->h_sweeps[2 * s + 1];
    for (int o = first; o < first + cnt; ++o) {
      const int kind = prog->h_ops[4 * o], x = prog->h_ops[4 * o + 1], y = prog->h_ops[4 * o + 2], c = prog->h_ops[4 * o + 3];
      if (kind == MLBP_OP_PAIR_TM || kind == MLBP_OP_PAIR_MT) {
        // column-major views: C = msgs[:, c, :]^T (X x B, ld), Bm = msgs[:, y, :]^T; the row-major table is T^T
        // column-major, so T . m needs op(A) = transpose and m^T . T none
        const double* T = a->pair_tables + (size_t)a->pair_tab_host[x] * X * X;
        const int rc = g_rb.dgemm(g_rb.handle, kind == MLBP_OP_PAIR_TM ? RB_OP_TRANSPOSE : RB_OP_NONE, RB_OP_NONE, X, B, X, &one, T, X,
                                  a->msgs + (size_t)y * X, ld, &zero, a->msgs + (size_t)c * X, ld);
        if (rc != 0) return fail(MLBP_EHIP, "rocblas_dgemm failed with status %d", rc);
        if (norm) hipLaunchKernelGGL(renormalize_slot_kernel, dim3(B), dim3(WG), 0, st, a->msgs, n_msgs, X, c);
      } else if (kind == MLBP_OP_VAR) {
        hipLaunchKernelGGL(variable_update_kernel, dim3(B), dim3(WG), (size_t)X * sizeof(double), st, a->msgs, n_msgs, X,
                           prog->d_srcs, x, y, c, norm);
      } else if (!(unary_once && unary_done[c])) {
        unary_done[c] = 1;
        hipLaunchKernelGGL(unary_update_kernel, dim3(B), dim3(WG), 0, st, a->msgs, n_msgs, X, a->unary_tables, a->unary_tab,
                           prog->U, a->n_unary_tables, x, c, norm, prog->d_status);
      }
    }
  }
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "shared-table GEMM path: a launch failed");
  return MLBP_OK;
}

}  // namespace mlbp
