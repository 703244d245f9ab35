// Internal declarations shared by the host (mlbp_host.cpp) and device (mlbp_*.hip) translation
// units of libmlbp.so.  Not part of the ABI.
#ifndef MLBP_INTERNAL_H
#define MLBP_INTERNAL_H

#include <vector>

#include "../../include/mlbp.h"

namespace mlbp_dev { struct GradFusedDev; }

struct mlbp_program;

namespace mlbp {
// Records a printf-style message for mlbp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Fused program form of the X = 64 kernels (build_fused_program in mlbp_sweep.hip): 8-word op headers.
enum { FOP_UNARY = 0, FOP_PAIR_TM = 1, FOP_PAIR_MT = 2, FOP_VAR = 3, FOP_VAR_PAIR_TM = 4, FOP_VAR_PAIR_MT = 5,
       FOP_BUNDLED = 0x100 /* flag: the next update touches disjoint slots and may share this one's barrier */ };
struct FusedProgram {
  std::vector<int32_t> fops, psrcs, fsweeps, hoist, cpw, pairseq, written;
  int n_cprod = 0;
  bool has_unary_fops = false;
};

// Micro-op form of the lean X = 64 kernel (mlbp_lean.hip): operands are LDS byte offsets.
struct LeanProgram {
  bool ok = false;
  const char* why = "";
  int n_bundles = 0, HL = 0, n_cprod = 0, WL = 0;
  std::vector<int32_t> image;          // bundles [n_bundles][16] | hoist [4][HL][2] | cprod lists [n_cprod][16] | written [4][WL] | padding
  std::vector<char> hoisted;           // [n_msgs] the slot holds a hoisted (constant) unary message
  std::vector<std::vector<int32_t>> cprods;   // hoisted message slots of constant product k
};
void build_lean_program(const FusedProgram& fp, int n_msgs, LeanProgram& out);
bool build_lean_readout(const LeanProgram& lp, int n_msgs, int n_vars, const int32_t* in_off, const int32_t* in_slots,
                        std::vector<int32_t>& image);
// Enqueues the lean scale-free kernel when it applies (sets *launched); flagged graphs are left in prog->d_bail.
// gf != NULL: the call's gradient runs as the kernel's epilogue (P <= 3, every unary message hoisted, F = (3, 6)).
int launch_lean_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, const mlbp_dev::GradFusedDev* gf, void* stream, bool* launched);
// The same for several (program, arguments) groups in one launch; *launched false = some group does not qualify.
int launch_lean_groups(const mlbp_program* const* progs, const mlbp_sweep_args* args, int n_groups, void* stream, bool* launched);

// Shared-table (MFMA) form, mlbp_shared.hip: 16 graphs per workgroup, messages kept as [state][graph]
// tiles in LDS, only the "live" slots (read or written inside the sweeps) resident.
struct SharedProgram {
  bool ok = false;
  const char* why = "";               // when !ok: what rules the program out
  int n_ops = 0, n_live = 0, n_cpw = 0, n_back = 0, n_fill = 0, n_init = 0, n_bundles = 0;
  int off_ent = 0, off_back = 0, off_fill = 0, off_init = 0, off_ptile = 0, off_written = 0;
  int max_sources = 1;                 // most tiles any variable update multiplies
  // product-fused form (degree <= 2 variables: K2, K3, chains, rings): the producer of a factor->variable message stores it
  // already multiplied by the destination variable's constant product, so a contraction reads ONE tile and multiplies nothing
  bool pf_ok = false;
  int off_pfb = 0, off_stash = 0, off_pinit = 0, n_stash = 0, n_pinit = 0, off_vftile = 0;
  bool vf_direct = false;             // the gradient epilogue may read the final variable->factor messages from the message tiles
  // ... its three-source variant (K4 cliques: messages stored as sqrt(c) (.) m, constant products in memory; build_shared_program)
  bool p3_ok = false;
  int n_lds = 0, sqrt_mask = 0, off_map3 = 0, off_kind3 = 0, off_back3 = 0;
  std::vector<int32_t> sweeps;         // {first op, count} of the transformed op list (one sequence: sweep boundaries mean nothing here)
  std::vector<int32_t> image;          // bundles [n_bundles + 1][2][16] | cprod entries [n_cpw] | write-back pairs [n_back][2] | fill slots [n_fill] | uniform tiles [n_init]
  std::vector<int32_t> live_of_slot;   // [n_msgs + 1 + n_cprod] LDS tile of a slot (ext slots included) or -1
  std::vector<int32_t> hoisted;        // [n_msgs] unary factor whose constant message the slot holds, or -1
  std::vector<char> written;           // [n_msgs] some update of the program writes the slot
  std::vector<std::vector<int32_t>> cprods;   // hoisted message slots of constant product k
};
void build_shared_program(const FusedProgram& fp, int n_msgs, int P, int U, SharedProgram& out);
// Builds the device read-out image of the shared form from the per-variable incoming-slot lists;
// false when some variable's constant part matches no constant product.
bool build_shared_readout(const SharedProgram& sp, int n_msgs, int n_vars, const int32_t* in_off, const int32_t* in_slots,
                          std::vector<int32_t>& image);
// Enqueues the shared-table kernel when it applies (sets *launched); flagged graphs are left in
// prog->d_bail for the exact kernel.
int launch_shared_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream, bool* launched);
// Whether that launch also runs a->gradient (as the sweep kernel's epilogue); and whether the exact X = 64 kernel can (it
// redoes flagged graphs, gradient included).  Both are functions of the arguments alone.
bool shared_gradient_fused(const mlbp_program* prog, const mlbp_sweep_args* a);
bool exact_kernel_fuses_gradient(const mlbp_program* prog, const mlbp_sweep_args* a);
// mlbp_grad.hip: the per-graph gradient kernel on the graphs whose flag byte is set (fix-up behind a fused gradient the exact
// kernel cannot redo: more than three pairwise factors)
int gradient_flagged_only(const mlbp_gradient_args* a, const uint8_t* flags, void* stream);
// The same for n_groups groups in ONE launch (table cached with `owner`'s group tables).
int gradient_flagged_groups(const mlbp_gradient_args* args, const uint8_t* const* flags, int n_groups, mlbp_program* owner, void* stream);
// The same for several (program, arguments) groups in one launch sequence; *launched false = some group does not qualify.
int launch_shared_groups(const mlbp_program* const* progs, const mlbp_sweep_args* args, int n_groups, void* stream, bool* launched);
// Pairwise part of the gradient for shared tables (X = 64, F_ee = 3), ADDED to a->grad_en_en.
int launch_shared_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream);
// Shared tables at X = 128 .. 512: the sweeps update by update over the whole batch, every contraction one launch of the
// hand-written MFMA kernel of mlbp_gemm.hip (float64 tables, or float32 with MLBP_SWEEP_PAIR_TABLES_F32).
int launch_gemm_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream);
int gemm_path_ready();      // always MLBP_OK (kept for the callers' sake)
bool gemm_path_supports(int X);
// Pairwise part of the gradient for shared tables at X >= 128 (F_ee = 3), ADDED to a->grad_en_en.
int launch_gemm_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream);
// Bytes of mlbp_gradient_args.workspace that call needs (0: none): the weighted table fragments and the per-graph sums.
size_t gemm_gradient_workspace_bytes(const mlbp_gradient_args* a);
size_t shared_gradient_workspace_bytes(const mlbp_gradient_args* a);

// Program-owned device scratch grows by allocating a NEW block; the old one stays alive until mlbp_program_destroy -- a HIP
// graph captured earlier may still name it (a free on growth was a use-after-free on replay).  *cap in bytes.
int program_grow(mlbp_program* prog, void** p, size_t* cap, size_t bytes, bool zero = false);
// A process-wide arena for the calls that carry no program and were given no workspace (mlbp_gradient_f64 with shared tables):
// one block per (device, purpose), grown the same way and never freed.  Calls that use it on different streams must not overlap.
int fallback_scratch(int purpose, size_t bytes, void** out);
enum { SCRATCH_GEMM_GRADIENT = 0, SCRATCH_SHARED_GRADIENT = 1, SCRATCH_PURPOSES = 2 };
}  // namespace mlbp

namespace mlbp {
// Device copies of the group tables of mlbp_sweep_groups_f64, ONE PER DISTINCT CONTENTS: a call whose table equals an earlier
// one reuses that copy without an upload, a call with other buffers or groups gets its own -- so a HIP graph captured on one
// table keeps replaying against it.  At most MAX tables per program; beyond that the oldest copy is overwritten
// (a graph captured on it must be re-captured: include/mlbp.h says so).
struct GroupTables {
  enum { MAX = 64 };
  struct Entry { std::vector<int32_t> words; int32_t* dev = nullptr; size_t cap_words = 0; };
  std::vector<Entry> entries;
  size_t next_evict = 0;
};
int group_table_device(GroupTables& gt, const std::vector<int32_t>& table, void* stream, int32_t** out);
void group_tables_free(GroupTables& gt);
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
  mlbp::FusedProgram fused;      // host copy (the update-by-update contraction path walks it)
  mlbp::LeanProgram lean;
  mlbp_program* pruned = nullptr;   // the same program without the updates MLBP_SWEEP_SKIP_UNCHANGED drops, or NULL (none to drop)
  int32_t n_dropped = 0;
  bool is_twin = false;             // this IS some program's pruned twin
  bool last_was_pruned = false;     // the last sweep call ran the twin (mlbp_program_exact_count reads its flags)
  int32_t* d_limage = nullptr;   // LeanProgram::image
  int32_t* d_lreadout = nullptr; // build_lean_readout
  mlbp::GroupTables gtables;     // group tables of mlbp_sweep_groups_f64 calls that name this program first (launch_lean_groups)
  mlbp::GroupTables stables;     // the same for the shared-table kernels (launch_shared_groups)
  // shared-table form (mlbp_shared.hip)
  mlbp::SharedProgram shared;
  int32_t* d_simage;      // SharedProgram::image
  int32_t* d_sreadout;    // per variable: base tile, count, live tiles (4-word aligned lists) or NULL
  int32_t n_sreadout;
  bool sreadout_all_tiled = false;   // ... and at least one message tile
  bool sreadout_all_based = false;   // every variable of the read-out has a constant product (the product-fused read-out needs it)
  double* d_tfrag;        // [32][2][4096] table fragments in MFMA operand order (lazily allocated)
  double* d_spill = nullptr;   // message tiles of the shared-table kernel that do not fit LDS (lazily allocated)
  double* d_wfrag = nullptr;   // the gradient epilogue's weighted table fragments [n_pair_tables][2][4][4096] (lazily allocated)
  size_t wfrag_cap = 0;          // in BYTES (program_grow); spill_cap / ptiles_cap / stable_cap / gtable_cap likewise
  int32_t* d_header = nullptr; // the prepare launch's per-group headers for the sweep kernel [groups][8] (lazily allocated)
  size_t header_cap = 0;       // in bytes
  // grouped launches (mlbp_sweep_groups_f64) with more than one form of the shared-table kernel: the product-fused groups' sweep
  // launch runs on a side stream beside the others' (fork / join by events behind the prepare launch); created on first use
  void* side_stream = nullptr; void* ev_fork = nullptr; void* ev_join = nullptr;
  double* d_ptiles = nullptr;  // constant-product tiles of the shared-table kernel [groups][n_cprod][1024] (lazily allocated)
  size_t ptiles_cap = 0;       // in doubles
  size_t spill_cap = 0;        // in doubles
  std::vector<int32_t> h_ops, h_sweeps;   // host copies of the validated op list (the op-by-op GEMM path walks them)
  // large-state shared-table path (mlbp_gemm.hip): fragment-ordered table copies [P][2][XA^2], the formed input messages of
  // the chunked contraction [B][X], and the gradient's workspace when the caller gave none
  void* d_gfrag = nullptr; size_t gfrag_cap = 0;
  void* d_gxbuf = nullptr; size_t gxbuf_cap = 0;
  void* d_gwork = nullptr; size_t gwork_cap = 0;
  std::vector<void*> retired;    // blocks program_grow replaced: freed with the program
};

#endif
