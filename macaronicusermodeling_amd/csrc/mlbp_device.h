// Device helpers shared by the HIP translation units of libmlbp.so (gfx950 only).  Not part of the ABI.
#ifndef MLBP_DEVICE_H
#define MLBP_DEVICE_H

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace mlbp_dev {

__device__ __forceinline__ double nan_to_num(double x) {
  // np.nan_to_num (LBP.py:729): NaN -> 0, +inf -> DBL_MAX, -inf -> -DBL_MAX
  if (x != x) return 0.0;
  if (x == __builtin_huge_val()) return DBL_MAX;
  if (x == -__builtin_huge_val()) return -DBL_MAX;
  return x;
}

// 64-bit DPP move: lane l receives the value of the lane selected by CTRL inside its row of 16.
// 0xB1 = quad_perm[1,0,3,2] (l^1), 0x4E = quad_perm[2,3,0,1] (l^2), 0x1B = quad_perm[3,2,1,0] (3-l),
// 0x141 = row_half_mirror (7-l within 8), 0x140 = row_mirror (15-l within 16).  VALU speed: no LDS
// crossbar round trip as with ds_bpermute (__shfl_xor).
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

// Sum over the 64 lanes, the same bits in every lane: four DPP steps give each row of 16 its
// sum (every pairing adds the same two operands in both partners, so the row agrees bitwise), then
// the four row sums are combined through scalar registers.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// acc * m followed by nan_to_num; the three compares only run when some lane of the wave saw a
// non-finite product (wave-uniform branch).
__device__ __forceinline__ double mul_nan_to_num(double m, double acc) {
  double p = m * acc;
  if (__builtin_expect(__any(!__builtin_isfinite(p)), 0)) p = nan_to_num(p);
  return p;
}

// Message.renormalize (LBP.py:649-657): positive total -> v / total, else uniform.
__device__ __forceinline__ double renorm(double v, double total, double uniform, bool normalize) {
  if (!normalize) return v;
  return total > 0.0 ? v / total : uniform;
}

// Program data (op headers, source lists, sweep table) is read-only for the whole launch and
// wave-uniform.  Reading it through the CONSTANT address space lets the compiler use scalar loads
// (s_load -> SGPRs, lgkmcnt) instead of per-lane vector loads that queue behind the table stream
// on vmcnt; a plain `const int32_t*` is not enough because the kernel also stores to global memory.
typedef const int32_t __attribute__((address_space(4))) * const_i32p;
__device__ __forceinline__ const_i32p as_const(const int32_t* p) {
  return (const_i32p)(uintptr_t)p;
}

typedef const double __attribute__((address_space(4))) * const_f64p;
__device__ __forceinline__ const_f64p as_const_f64(const double* p) {
  return (const_f64p)(uintptr_t)p;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), e = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, e));
}



__device__ __forceinline__ unsigned mag_key(double x) { return (unsigned)__double2hiint(x); }
constexpr unsigned KEY_BAD = 0x7FF00000u;   // and above: negative or non-finite
constexpr unsigned KEY_MIN = 0x00100000u;   // below: zero or subnormal maximum


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
  // optional fused read-out of the variable marginals (LBP.py:392-400) from the on-chip messages
  double* marginals;           // [B][n_vars][X] or NULL
  const int32_t* readout;      // device: in_off [n_vars+1] then in_slots
  int32_t n_vars;
  // fix-up mode of the generic kernel behind a fast kernel: only graphs with only[g] != 0 are computed, from uniform
  // messages when fill_uniform is set (FactorGraph.initialize fused)
  const uint8_t* only;
  int32_t fill_uniform;
  int32_t approx_k;            // > 0: use_approx_inference (LBP.py:506-507, 515-516): only the approx_k largest entries of the
                               // incoming message enter a pairwise update (au.sparse_vec_mat_dot, c_array_utils.pyx:193-205)
};

// Gradient fused into a sweep launch (X = 64, F = (3, 6), tables still on chip): what FactorGraph.get_unregularized_gradeint
// (LBP.py:301-320) needs besides the messages and tables the workgroup already holds.
struct GradFusedDev {
  const int32_t* pair_c_slot; const int32_t* pair_r_slot; const int32_t* pair_phi; const int32_t* pair_label;
  const int32_t* unary_kind; const int32_t* unary_obs; const int32_t* unary_label;
  const double* phi_en_en; const double* phi_en_en_w1;
  const double* phi_en_en_t; const double* phi_en_en_w1_t; const double* phi_en_de_t;
  double* grad_en_en; double* grad_en_de;
  int32_t Vde, enabled;
};

}  // namespace mlbp_dev

#endif
