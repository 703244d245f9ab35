// Internal declarations shared by the host (mlbp_host.cpp) and device (mlbp_*.hip) translation
// units of libmlbp.so.  Not part of the ABI.
#ifndef MLBP_INTERNAL_H
#define MLBP_INTERNAL_H

#include "../../include/mlbp.h"

namespace mlbp {
// Records a printf-style message for mlbp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace mlbp

// Device-resident, validated op list (see mlbp_program_create).
struct mlbp_program {
  int32_t n_ops, n_srcs, n_sweeps, n_msgs, P, U;
  int32_t n_pairseq;      // pair ops over all sweeps, in execution order
  int32_t max_srcs;       // largest MLBP_OP_VAR fan-in
  int32_t* d_ops;         // [n_ops][4]
  int32_t* d_srcs;        // [n_srcs]
  int32_t* d_sweeps;      // [n_sweeps][2]
  int32_t* d_pairseq;     // [n_pairseq + 1] pair slot of the k-th executed pair op (-1 terminated)
  int32_t* d_status;      // [1] set non-zero by a kernel that met an out-of-range table index
  // fused form used by the X = 64 kernel (build_fused_program in mlbp_sweep.hip)
  int32_t n_fops, n_hoist, n_psrcs, n_cprod, n_cpw, n_written;
  bool sf_ok;             // the scale-free kernel applies (no in-loop unary ops)
  unsigned char* d_bail;  // [bail_cap] per-graph "redo with the exact kernel" flags
  int32_t bail_cap;
  int32_t* d_readout;     // in_off [n_vars+1] then in_slots (mlbp_program_set_readout), or NULL
  int32_t n_vars, n_readout;
  int32_t* d_fops;        // one block: op headers [n_fops][8], source lists [n_psrcs], hoist list
                          // [n_hoist][2], constant-product lists [n_cpw]
  int32_t* d_fsweeps;     // [n_sweeps][2]
  int32_t* d_fpairseq;    // pair slot of the k-th executed pairwise update of the fused form (-1 terminated)
  int device;
};

#endif
