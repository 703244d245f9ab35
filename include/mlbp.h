/*
 * mlbp.h -- C ABI of libmlbp.so: the MI355X-native (gfx950) loopy-belief-propagation hot path.
 *
 * The reference (arendu-zz/MacaronicUserModeling) has no C ABI: its hot path sits behind two
 * Python modules, `LBP.py` (FactorGraph / VariableNode / FactorNode) and the Cython extension
 * `array_utils.c_array_utils`.  The entry points below are what a binding for that path would
 * bind; each cites the reference interface it replaces (paths relative to the reference root).
 * The Python mirror of the reference API (macaronicusermodeling_amd/LBP.py and
 * macaronicusermodeling_amd/array_utils/c_array_utils.py) calls ONLY these functions, through
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MLBP_E* code otherwise; the text of the last
 *     error on the calling thread is available from mlbp_last_error();
 *   - pointers documented "device" are HIP device addresses (e.g. torch tensor.data_ptr());
 *     pointers documented "host" are ordinary process memory; the library never frees or
 *     reallocates caller memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); device functions only
 *     enqueue work and do not synchronise unless stated;
 *   - all floating-point data is IEEE float64, the reference's dtype (LBP.py:4); integer data is
 *     int32 unless stated; matrices are row-major;
 *   - no CPU fallback exists: a device function on a machine without a gfx950 device fails with
 *     MLBP_ENODEVICE.
 *
 * Concurrency (what the deployment model -- one process per GPU, train_mp.py:634 -- needs, stated exactly)
 *   - host functions and mlbp_last_error() are thread-safe; the sweep-kernel diagnostic mlbp_last_sweep_kernel() is
 *     per calling thread;
 *   - a mlbp_program belongs to the device that was current when it was created (mlbp_sweep_f64 checks this) and owns
 *     per-program device state (status word, per-graph redo flags): calls that use the SAME program must be enqueued
 *     on one stream, or be ordered by the caller;
 *   - calls with DIFFERENT programs may run on different streams concurrently (every scratch buffer of the sweep paths --
 *     redo flags, fragment copies, spill areas, formed messages, group tables -- belongs to a program), with these exceptions
 *     that share process-wide scratch and must not overlap across streams: mlbp_log_posterior_sum_f64 /
 *     mlbp_step_statistics_f64 (their block partials) and a STANDALONE mlbp_gradient_f64 on shared tables that was given no
 *     mlbp_gradient_args.workspace; a mlbp_sweep_groups_f64 call counts as a use of EVERY program it names (its group table
 *     lives with the first one);
 *   - scratch is allocated -- and a group table uploaded -- at the first call that needs it or needs more of it: that call is
 *     not enqueue-only; mlbp_program_reserve and one eager step with the same arguments before stream capture move all of it
 *     up front.  Scratch only ever GROWS by a new block; a block that was outgrown stays alive until mlbp_program_destroy (the
 *     process-wide fallback blocks: until exit), so a HIP graph captured earlier keeps replaying against valid memory whatever
 *     later calls ask for.  A program keeps one device copy per DISTINCT group table (up to 64; beyond that the oldest copy is
 *     overwritten and a graph captured on it must be captured again);
 *   - mlbp_set_sweep_variant and the status words behind mlbp_gradient_status are process-wide.
 */
#ifndef MLBP_H
#define MLBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLBP_VERSION_MAJOR 0
#define MLBP_VERSION_MINOR 1

enum {
  MLBP_OK = 0,
  MLBP_EINVAL = -1,    /* bad argument (shape, index out of range, NULL pointer)                 */
  MLBP_EHIP = -2,      /* a HIP runtime call failed                                              */
  MLBP_ENODEVICE = -3, /* no gfx950 device visible                                               */
  MLBP_ENOMEM = -4,    /* output buffer capacity too small (host functions)                      */
  MLBP_EUNSUPPORTED = -5
};

/* ---------------------------------------------------------------------------------------------
 * library info
 * ------------------------------------------------------------------------------------------- */
int mlbp_version(void);                 /* major*100 + minor                                     */
const char* mlbp_arch(void);            /* "gfx950": the only code object in the library         */
const char* mlbp_last_error(void);      /* thread-local, never NULL                              */
int mlbp_device_count(void);            /* number of visible HIP devices, 0 if none (no error)   */

/* ---------------------------------------------------------------------------------------------
 * HOST integer logic (bit-exact with the reference; runs without a GPU)
 *
 * A topology describes one factor graph after FactorGraph.initialize() has sorted the factors by
 * id (LBP.py:195-196).  Variables and factors are referred to by dense indices:
 *   variable index v in [0, n_vars), factor index f in [0, n_factors) (id-sorted position).
 * A *node* is encoded as: variable v -> v;  factor f -> n_vars + f.
 * ------------------------------------------------------------------------------------------- */
typedef struct mlbp_topology {
  int32_t n_vars;
  int32_t n_factors;
  const int32_t* fac_nvars;   /* host [n_factors]   1 (unary) or 2 (pairwise); LBP.py:446-447     */
  const int32_t* fac_var;     /* host [2*n_factors] variable index per VARSET position, -1 pad    */
  const int32_t* fac_dim;     /* host [2*n_factors] table axis of that variable (var_id2dim)      */
  const int32_t* var_fac_off; /* host [n_vars+1]    CSR offsets into var_fac                      */
  const int32_t* var_fac;     /* host [..]          factor indices in FACSET (creation) order     */
} mlbp_topology;

/* FactorGraph.has_loops with an explicit root instead of the random.sample draw (LBP.py:174-190).
 * Returns 1 / 0, or a negative error code. */
int mlbp_has_loops(const mlbp_topology* t, int32_t root_var);

/* FactorGraph.get_message_schedule (LBP.py:155-172).  Writes (child node, parent node) pairs in
 * discovery order to pairs[2*i], pairs[2*i+1]; returns the number of pairs or MLBP_ENOMEM when
 * cap_pairs is too small. */
int mlbp_message_schedule(const mlbp_topology* t, int32_t root_var, int32_t* pairs, int32_t cap_pairs);

/* Message-slot numbering used by every device function: factors in index order; a unary factor
 * owns one slot (factor->variable); a pairwise factor owns, for each varset position k,
 * slot (variable->factor) then slot (factor->variable) (the creation order of LBP.py:211-216).
 * f2v[2*f+k] / v2f[2*f+k] receive the slot or -1.  pair_slot[f] / unary_slot[f] receive the
 * factor's position among the pairwise / unary factors (or -1).  Returns n_msgs. */
int mlbp_message_slots(const mlbp_topology* t, int32_t* f2v, int32_t* v2f, int32_t* pair_slot,
                       int32_t* unary_slot);

/* Operation list ("program") of one sweep of FactorGraph.treelike_inference rooted at root_var
 * (LBP.py:223-243): the up pass over the reversed schedule then the down pass, updates whose
 * destination is a unary factor dropped (LBP.py:228, 237).  Each op is 4 int32 words
 * {kind, a, b, c}:
 *   MLBP_OP_UNARY    a = unary slot                    c = dst message slot   (LBP.py:494-498)
 *   MLBP_OP_PAIR_TM  a = pair slot  b = src msg slot   c = dst   out = T . m  (LBP.py:509)
 *   MLBP_OP_PAIR_MT  a = pair slot  b = src msg slot   c = dst   out = m^T . T (LBP.py:518)
 *   MLBP_OP_VAR      a = offset into srcs, b = count   c = dst   product of srcs (LBP.py:381-389)
 * Returns the op count, or MLBP_ENOMEM; *n_srcs receives the number of srcs entries written. */
enum { MLBP_OP_UNARY = 0, MLBP_OP_PAIR_TM = 1, MLBP_OP_PAIR_MT = 2, MLBP_OP_VAR = 3 };
int mlbp_compile_sweep(const mlbp_topology* t, int32_t root_var, int32_t* ops, int32_t cap_ops,
                       int32_t* srcs, int32_t cap_srcs, int32_t* n_srcs);

/* ---------------------------------------------------------------------------------------------
 * DEVICE: programs (validated, device-resident op lists)
 * ------------------------------------------------------------------------------------------- */
typedef struct mlbp_program mlbp_program;

/* Validates every slot index of the host op list against (n_msgs, P, U), copies ops / srcs /
 * sweep table to the device and returns an opaque handle.  sweeps[2*s], sweeps[2*s+1] = (first
 * op, op count) of sweep s, so sweeps with equal roots can share their ops. */
int mlbp_program_create(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                        const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                        mlbp_program** out);
int mlbp_program_destroy(mlbp_program* p);

/* Pre-allocates the per-graph flag bytes the default X = 64 path needs for batches of up to
 * max_graphs graphs, so that mlbp_sweep_f64 itself performs no allocation (stream capture).  Without
 * it the first launch with a larger batch allocates. */
int mlbp_program_reserve(mlbp_program* p, int32_t max_graphs);

/* HOST only (no device needed): what the program rewrites make of an op list.  out8 = { updates of the fused
 * form, of them lone variable->factor updates, fused variable+pairwise updates, bundles (two updates under
 * one barrier); shared-table form: bit 0 applicable, bit 1 in its product-fused form (every variable update multiplies at
 * most one constant product and one message: the message's producer stores the product), bit 2 the fused gradient reads
 * the final variable->factor messages from that form's tiles, bit 3 the three-source variant of that form (variables with three
 * pairwise factors, K4 cliques: sqrt(c) (.) message stored); its resident message tiles, its updates, LDS bytes of its
 * tiles for 16 graphs }.  Same validation and error codes as mlbp_program_create. */
int mlbp_program_plan(const int32_t* ops, int32_t n_ops, const int32_t* srcs, int32_t n_srcs,
                      const int32_t* sweeps, int32_t n_sweeps, int32_t n_msgs, int32_t P, int32_t U,
                      int32_t* out8);

/* Attaches the variable read-out tables to a program so that mlbp_sweep_f64 can write the
 * variable marginals straight from the on-chip messages (mlbp_sweep_args.marginals): in_off
 * [n_vars+1] / in_slots are HOST arrays, variable v multiplies the messages in slots
 * in_slots[in_off[v] .. in_off[v+1]) in that (facset) order (LBP.py:394-396). */
int mlbp_program_set_readout(mlbp_program* p, int32_t n_vars, const int32_t* in_off, const int32_t* in_slots);

/* Synchronising: number of graphs (among the first B) that the last default-variant launch of this
 * program handed from the scale-free kernel to the exact kernel (degenerate inputs: zero-sum
 * messages, negative or non-finite entries).  Diagnostic; results are exact either way. */
int mlbp_program_exact_count(const mlbp_program* p, int32_t B);
/* The number of updates of the program's root sequence that MLBP_SWEEP_SKIP_UNCHANGED drops (0: the flag changes nothing). */
int mlbp_program_skippable_updates(const mlbp_program* p);

/* Synchronising read-and-reset of the program's device status word: 0 = clean; 1 = a kernel skipped
 * a graph because one of its table indices lay outside [0, n_*_tables) (instead of reading out of
 * bounds); negative = error code. */
int mlbp_program_status(const mlbp_program* p);

/* ---------------------------------------------------------------------------------------------
 * DEVICE: the batched sweep (the hot path)
 *
 * Runs program->n_sweeps sweeps of sum-product message passing on B independent graphs that
 * share one topology / root sequence.  One workgroup owns one graph for the whole call; messages
 * live on-chip between updates; pairwise tables are streamed from HBM.
 * Replaces: FactorGraph.treelike_inference (LBP.py:218-245) with VariableNode.update_message_to
 * (LBP.py:377-389), FactorNode.update_message_to (LBP.py:490-526), Message.renormalize
 * (LBP.py:649-657), au.dense_dot / au.normalize / au.pointwise_multiply
 * (c_array_utils.pyx:90-91, 29-40, 12-16).
 * ------------------------------------------------------------------------------------------- */
struct mlbp_gradient_args;   /* defined below */

/* FactorGraph.get_posterior_probs of every graph behind the sweeps of the same call (train_mp.py:381-400 calls them back to
 * back): out[b] = sum_v log(marginals[b][v][labels[b][v]]), -inf replaced by -99.99 (LBP.py:247-259); *sum_out = sum_b out[b]
 * in a fixed order.  On the fast X = 64 paths the fix-up launch takes it (no launch of its own); elsewhere it is
 * mlbp_log_posterior_sum_f64 enqueued behind the sweeps. */
typedef struct mlbp_posterior_args {
  const int32_t* labels;   /* device [B][n_vars] */
  double* out;             /* device [B]         */
  double* sum_out;         /* device [1] or NULL */
} mlbp_posterior_args;

typedef struct mlbp_sweep_args {
  int32_t B;                  /* graphs                                                           */
  int32_t X;                  /* states per variable (len(v.domain))                              */
  int32_t n_pair_tables;      /* tables in pair_tables                                            */
  int32_t n_unary_tables;     /* columns in unary_tables                                          */
  const double* pair_tables;  /* device [n_pair_tables][X][X]                                     */
  const int32_t* pair_tab;    /* device [B][P]  table index of pair slot p of graph b             */
  const double* unary_tables; /* device [n_unary_tables][X]                                       */
  const int32_t* unary_tab;   /* device [B][U]                                                    */
  double* msgs;               /* device [B][n_msgs][X]  in/out                                    */
  int32_t normalize_messages; /* FactorGraph.normalize_messages (LBP.py:41)                       */
  int32_t init_messages;      /* non-zero: start from uniform 1/X messages (FactorGraph.initialize,
                                 LBP.py:211-216, fused into the launch) instead of reading msgs      */
  double* marginals;          /* device [B][n_vars][X] or NULL: VariableNode.get_marginal of every
                                 variable after the last sweep (LBP.py:392-400), read out in the same
                                 launch; needs mlbp_program_set_readout                              */
  const struct mlbp_gradient_args* gradient;
                              /* HOST pointer or NULL: when set, the per-graph gradients of
                                 FactorGraph.get_unregularized_gradeint (LBP.py:301-320) are written
                                 after the last sweep -- inside the same launch (tables still in
                                 registers, messages in LDS) when X = 64, F = (3,6), at most 3
                                 pairwise factors and the transposed feature tensors are given
                                 (shared pairwise tables: the matrix-core kernel's epilogue for the
                                 pairwise factors -- it needs the planar tensors and unary_expect --,
                                 the unary factors' terms in the launch in front of it);
                                 otherwise by mlbp_gradient_f64 enqueued behind the sweeps            */
  int32_t flags;              /* MLBP_SWEEP_* bits, 0 = none                                        */
  const int32_t* pair_tab_host;
                              /* HOST int32 [P] or NULL: with MLBP_SWEEP_SHARED_PAIR_TABLES, the pair_tab row
                                 every graph has.  Needed for X > 64 (any X up to 4096: the reference's X is its vocabulary,
                                 train_mp.py:591-594), where the sweeps then run op by op
                                 over the whole batch with one hand-written MFMA contraction launch per factor->variable update;
                                 the statement is checked on the device, a false one raises
                                 mlbp_program_status to 2                                            */
  const float* pair_tables_f32;
                              /* device [n_pair_tables][X][X] float32, read instead of pair_tables
                                 when MLBP_SWEEP_PAIR_TABLES_F32 is set                             */
  const mlbp_posterior_args* posterior;
                              /* HOST pointer or NULL: the log-posteriors (and their batch sum) of the call;
                                 needs `marginals`                                                   */
} mlbp_sweep_args;

/* flags: the caller states that pair_tab[b][p] is the same for every graph b (the reference's own
 * layout: one pot_en_en / pot_en_en_w1 array behind all pairwise factors, LBP.py:456-467, and one
 * theta per minibatch, train_mp.py:178-255).  With X = 64, init_messages and normalize_messages set,
 * at most two distinct tables and a working set that fits LDS, 16 graphs then share a workgroup
 * and the factor->variable updates run as float64 MFMA contractions T[64x64] . M[64x16]
 * (mlbp_shared.hip).  The statement is checked on the device: groups of graphs for which it does not
 * hold, and degenerate graphs, are computed by the exact kernel instead. */
#define MLBP_SWEEP_SHARED_PAIR_TABLES 1
/* flags: with MLBP_SWEEP_SHARED_PAIR_TABLES and a fused read-out (marginals != NULL) the messages
 * need not be written back to msgs; when a gradient is requested too, only the variable->factor
 * messages it reads are written back (contents of the other slots are then undefined). */
#define MLBP_SWEEP_NO_MESSAGE_WRITEBACK 2
/* flags: the pairwise tables are float32 (pair_tables_f32): the optional large-state mode of BASELINE
 * config 5 -- half the HBM bytes per update.  X = 256 or 512; messages, products and sums stay float64, so
 * results equal the float64 path run on the float32-rounded tables; no gradient in this mode. */
#define MLBP_SWEEP_PAIR_TABLES_F32 4
/* flags: the caller states that the index arrays are the identity, pair_tab[b][p] == b*P + p and unary_tab[b][u] ==
 * b*U + u (one private table per (graph, factor), stored in graph order -- FactorGraphBatch's default layout).  The
 * fast X = 64 kernel then addresses the tables directly instead of waiting for an index load before it can issue
 * the table loads.  The arrays must still be valid: the statement is checked on the device off the critical path,
 * and a graph for which it does not hold is computed by the exact kernel through the arrays. */
#define MLBP_SWEEP_DENSE_TABLES 8
/* flags: FactorGraph.use_approx_inference (LBP.py:506-507, 515-516): a pairwise factor->variable update uses only the
 * MLBP_APPROX_K = 100 largest entries of the incoming message (au.sparse_vec_mat_dot, c_array_utils.pyx:193-205).
 * Batched: the selection runs on the device inside the sweep launch (rank by value, ties by lower index).
 * 100 <= X <= 1024; smaller X fails like the reference's argpartition ("kth out of bounds"). */
#define MLBP_SWEEP_APPROX_INFERENCE 16
#define MLBP_APPROX_K 100
/* flags: run the program WITHOUT the updates whose inputs are bit for bit what they were when their destination was
 * last computed (mlbp_program_skippable_updates of them).  A root sequence (LBP.py:223-233) recomputes such messages
 * -- all of a tree's after its first sweep, the ones upstream of the first changed message when the root of a loopy
 * graph moves -- and gets the same values again.  On the default X <= 64 kernel (MLBP_KERNEL_LEAN) dropping them changes
 * no output bit (tests/test_gpu_sweep.py compares bitwise); the other kernels fuse a variable->factor product into the
 * contraction that follows it when the two are adjacent in the list, so a shorter list can move their results by a
 * rounding error (1e-12 in the tests).  Off by default: the default executes every update of the reference's schedule.
 * A gradient fused into the call reads the final messages and follows either list. */
#define MLBP_SWEEP_SKIP_UNCHANGED 32

int mlbp_sweep_f64(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream);

/* A minibatch of mixed graphs in one go: n_groups (program, arguments) pairs -- every group its own topology, root
 * sequence (the reference draws a fresh root per instance and sweep, LBP.py:223-225, and builds a different K_n per
 * instance, train_mp.py:257-299), tables and message buffer.  Equivalent to calling mlbp_sweep_f64 on the groups one
 * after the other, in fewer launches:
 *   - every group states MLBP_SWEEP_SHARED_PAIR_TABLES and qualifies for the shared-table kernel (X = 64, normalised and
 *     initialised messages, distinct programs): ONE prepare launch and one sweep launch per FORM of the kernel present
 *     among the groups (product-fused / its three-source variant / general; MLBP_KERNEL_SHARED_MFMA behind a group table: a
 *     workgroup looks its group up by block index) cover ALL groups -- with more than one form the product-fused launch runs on
 *     a side stream of progs[0], forked behind the prepare launch and joined before the call returns to `stream` (a stream
 *     capture records two parallel kernel nodes); groups that name the same tables and feature tensors share one set of
 *     fragment copies (written once per launch, owned by the first such group's program); a group WITHOUT pairwise factors (one
 *     predicted word) may be among them: all its graphs are flagged and redone by the fix-up launch; what follows per group is its fix-up
 *     pass over flagged graphs, its unary write-back when messages are kept, and its gradient when args[k].gradient is
 *     set -- a minibatch of mixed sentence shapes over the two shared pots (train_mp.py:220-299);
 *   - otherwise, when every group qualifies for the lean X = 64 kernel (float64 tables, normalised messages, at most 8
 *     pairwise factors, the same init / write-back / read-out choices, distinct programs) that kernel runs ALL groups in
 *     a single launch, followed by one small fix-up launch (and the gradient, if any) per group;
 *   - otherwise group by group.
 * progs / args are HOST arrays.  The group table is device memory owned by progs[0], uploaded only when its contents
 * differ from the previous call's (a stream capture of a repeated call records no copy). */
int mlbp_sweep_groups_f64(const mlbp_program* const* progs, const mlbp_sweep_args* args, int32_t n_groups, void* stream);

/* Kernel selector for the parity tests (process-wide; not needed in normal use):
 *   1     default: the fast kernels -- for X <= 64 the lean scale-free kernel (messages carried with an exact
 *         power-of-two scale, true normalisation deferred to the end of the call, mlbp_lean.hip), for shared tables the
 *         matrix-core kernels -- each followed by the exact kernel on the graphs it flagged as degenerate;
 *   3     the exact kernel (normalises after every update like the reference) and the per-graph streaming kernels on
 *         every graph: the tests' second opinion on the same inputs.
 * Both compute every update from the same inputs as the reference's order does (the X = 64 program form may reorder
 * independent updates and skips updates whose result is overwritten unread); they agree to rounding.  Other values
 * (the kernel generations retired in round 3) return MLBP_EINVAL. */
int mlbp_set_sweep_variant(int32_t variant);

/* Diagnostic: which kernel family the calling thread's last mlbp_sweep_f64 enqueued first (the exact
 * kernel may follow it for flagged graphs); -1 before the first call.  (0 and 1 were the first-generation and the first
 * scale-free kernel, retired.) */
#define MLBP_KERNEL_EXACT 2
#define MLBP_KERNEL_SHARED_MFMA 3
#define MLBP_KERNEL_WIDE 4
#define MLBP_KERNEL_GENERIC 5
#define MLBP_KERNEL_SHARED_GEMM 6
#define MLBP_KERNEL_LEAN 7            /* scale-free X <= 64 kernel, micro-op form (mlbp_lean.hip): the default */
int mlbp_last_sweep_kernel(void);
/* Diagnostic: 1 when that call's gradient (mlbp_sweep_args.gradient) ran as the epilogue of the sweep kernels themselves --
 * tables and final messages still on chip: MLBP_KERNEL_LEAN and MLBP_KERNEL_SHARED_MFMA with F = (3, 6) and at most three
 * pairwise factors -- 0 when a separate gradient launch followed the sweeps (or none was asked for). */
int mlbp_last_sweep_fused_gradient(void);

/* Fills msgs[B][n_msgs][X] with 1/X: FactorGraph.initialize (LBP.py:211-216). */
int mlbp_init_messages_f64(double* msgs, int64_t n_rows, int32_t X, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DEVICE: read-outs
 * ------------------------------------------------------------------------------------------- */
/* VariableNode.get_marginal for every variable of every graph (LBP.py:392-400):
 * out[b][v] = renormalize(uniform * prod_k msgs[b][in_slots[in_off[v]+k]]), nan_to_num after each
 * product.  in_off / in_slots are DEVICE int32 arrays ([n_vars+1], [..]) whose contents the
 * caller guarantees to lie in [0, n_msgs). */
int mlbp_marginals_f64(const double* msgs, int32_t B, int32_t n_msgs, int32_t X, int32_t n_vars,
                       const int32_t* in_off, const int32_t* in_slots, int32_t normalize_messages,
                       double* out, void* stream);

/* FactorGraph.get_posterior_probs (LBP.py:247-259): out[b] = sum_v log(marg[b][v][label[b][v]]),
 * -inf replaced by -99.99.  labels is a DEVICE int32 [B][n_vars] array. */
int mlbp_log_posterior_f64(const double* marginals, const int32_t* labels, int32_t B, int32_t n_vars,
                           int32_t X, double* out, void* stream);
/* The same, and *sum_out = sum_b out[b] in a fixed order (device pointer; NULL = no sum): the batch total of
 * train_mp.py:405-411 without a second reduction launch. */
int mlbp_log_posterior_sum_f64(const double* marginals, const int32_t* labels, int32_t B, int32_t n_vars,
                               int32_t X, double* out, double* sum_out, void* stream);

/* FactorGraph.get_posterior_probs (LBP.py:247-259) for SEVERAL groups of graphs in one launch -- a minibatch of mixed sentence
 * shapes, every group its own variable count (train_mp.py:257-299 builds a different K_n per instance): out[start_k + b] is the
 * log-posterior of graph b of group k.  `groups` is a DEVICE array of n_groups records sorted by `start`, the groups' ranges
 * [start_k, start_k + B_k) tile [0, n_total).  A label outside [0, X) is skipped and raises mlbp_gradient_status. */
typedef struct mlbp_posterior_group {
  const double* marginals;    /* device [B][n_vars][X] */
  const int32_t* labels;      /* device [B][n_vars]    */
  int32_t n_vars, B;
  int64_t start;
} mlbp_posterior_group;
int mlbp_log_posterior_groups_f64(const mlbp_posterior_group* groups, int32_t n_groups, int64_t n_total, int32_t X, double* out,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * DEVICE: factor beliefs and the log-linear gradient, batched over graphs
 * ------------------------------------------------------------------------------------------- */
/* FactorNode.get_factor_beliefs for every pairwise factor (LBP.py:543-569):
 * out[b][p] = normalise((c r^T) * T) with c / r the messages from the dim-0 / dim-1 variable
 * (device int32 c_slot[P], r_slot[P] give their message slots); all-zero when the total is <= 0
 * (au.normalize).  out: device [B][P][X][X]. */
int mlbp_pair_beliefs_f64(const double* msgs, int32_t B, int32_t n_msgs, int32_t X, int32_t P,
                          const double* pair_tables, const int32_t* pair_tab, int32_t n_pair_tables,
                          const int32_t* c_slot, const int32_t* r_slot, double* out, void* stream);

/* FactorGraph.get_unregularized_gradeint (LBP.py:301-320) for every graph of a batch, fused with
 * FactorNode.get_gradient / cell_gradient / get_factor_beliefs (LBP.py:592-619, 528-574) so that no
 * belief matrix is materialised:
 *   pairwise en_en factor p: grad_k += phi[l0][l1][k] - sum_ij b_ij phi[i][j][k]
 *   unary factor u:          grad_k += phi[lab][obs][k] - sum_x b_x phi[x][obs][k],  b = normalise(table)
 * phi selection follows FactorNode.get_phi (LBP.py:469-480): kind 0 = phi_en_en (gap > 1),
 * 1 = phi_en_en_w1 (gap == 1), 2 = phi_en_de.  All index arrays are DEVICE int32; per-graph indices
 * are range-checked in the kernel (a bad index skips the factor and raises mlbp_gradient_status). */
typedef struct mlbp_gradient_args {
  int32_t B, X, n_msgs, P, U;
  int32_t F_ee, F_ed;           /* feature counts: (3, 6) as train_mp.py:520-523; (2,2), (1,1) also built */
  int32_t Vde;                  /* columns of phi_en_de / pot_en_de                                 */
  int32_t n_pair_tables, n_unary_tables;
  const double* msgs;           /* [B][n_msgs][X]                                                   */
  const double* pair_tables;    /* [n_pair_tables][X][X]                                            */
  const int32_t* pair_tab;      /* [B][P]                                                           */
  const int32_t* pair_c_slot;   /* [P] message slot  (dim-0 variable -> factor)                     */
  const int32_t* pair_r_slot;   /* [P] message slot  (dim-1 variable -> factor)                     */
  const int32_t* pair_phi;      /* [P] 0 / 1                                                        */
  const int32_t* pair_label;    /* [B][P][2] supervised label index of the dim-0 / dim-1 variable   */
  const double* unary_tables;   /* [n_unary_tables][X]                                              */
  const int32_t* unary_tab;     /* [B][U]                                                           */
  const int32_t* unary_kind;    /* [U] 0 / 1 / 2                                                    */
  const int32_t* unary_obs;     /* [B][U] observed column (PotentialTable.observed_dim)             */
  const int32_t* unary_label;   /* [B][U] supervised label index of the factor's variable           */
  const double* phi_en_en;      /* [X][X][F_ee]                                                     */
  const double* phi_en_en_w1;   /* [X][X][F_ee]                                                     */
  const double* phi_en_de;      /* [X][Vde][F_ed]                                                   */
  const double* phi_en_en_t;    /* optional (may be NULL): the same tensors with the first two axes  */
  const double* phi_en_en_w1_t; /* swapped, [column][x][F], so that a unary factor's feature column  */
  const double* phi_en_de_t;    /* phi[:, observed_dim, :] (LBP.py:602) is one contiguous slab        */
  const double* phi_en_en_p;    /* optional (may be NULL): feature-major ("planar") copies [F][X][X]  */
  const double* phi_en_en_w1_p; /* of the two en_en tensors; a pairwise factor then reads each feature */
                                /* plane with the same 16-byte-per-lane pattern as its table           */
  double* grad_en_en;           /* out [B][F_ee]                                                    */
  double* grad_en_de;           /* out [B][F_ed]                                                    */
  int32_t flags;                /* MLBP_GRADIENT_* bits                                              */
  const int32_t* pair_tab_host; /* HOST int32 [P] or NULL: with MLBP_GRADIENT_SHARED_PAIR_TABLES and X >= 128 the
                                   pairwise factors become four MFMA contractions each over the whole batch         */
  const double* unary_expect;   /* optional [n_unary_tables][8] from mlbp_unary_expectations_f64: with
                                   MLBP_GRADIENT_SHARED_PAIR_TABLES the unary factors then cost one
                                   gather per factor instead of a reduction over the states           */
  const int32_t* pair_slots_host; /* HOST int32 [3][P] or NULL: the contents of pair_c_slot | pair_r_slot | pair_phi.  The X >= 128
                                   path walks the factors on the host; given this copy it only enqueues (and can be captured
                                   into a HIP graph), without it the call reads the three device arrays back and synchronises */
  void* workspace;              /* optional DEVICE scratch of workspace_bytes bytes, owned by the caller: the shared-table paths keep
                                   their weighted table fragments and per-graph sums there (mlbp_gradient_workspace_bytes says how
                                   much).  NULL: a process-wide block, grown on demand and never freed -- calls that fall back on it
                                   must not overlap across streams.  Inside mlbp_sweep_f64 (mlbp_sweep_args.gradient) a NULL
                                   workspace is replaced by scratch the PROGRAM owns                                        */
  int64_t workspace_bytes;
} mlbp_gradient_args;
/* Bytes of workspace mlbp_gradient_f64(a) would use (0: that call needs none).  Host only. */
int64_t mlbp_gradient_workspace_bytes(const mlbp_gradient_args* a);
/* flags: pair_tab[b][p] is the same for every graph b (see MLBP_SWEEP_SHARED_PAIR_TABLES).  With X = 64,
 * F_ee = 3 and the planar feature copies given, the pairwise factors of 16 graphs at a time are then
 * contracted on the matrix cores ((T (.) phi_k) . r, four contractions per factor).  Groups of 16 graphs for
 * which the claim does not hold are computed one graph at a time inside the same kernel (correct, slower). */
#define MLBP_GRADIENT_SHARED_PAIR_TABLES 1
/* flags: FactorGraph.use_approx_beliefs (LBP.py:554-563): a pairwise factor's beliefs live on the block of the
 * MLBP_APPROX_K largest entries of its two incoming messages (au.sparse_dot, sparse_pointwise_multiply,
 * sparse_normalize, c_array_utils.pyx:108-129, 23-26).  X >= MLBP_APPROX_K. */
#define MLBP_GRADIENT_APPROX_BELIEFS 2
int mlbp_gradient_f64(const mlbp_gradient_args* a, void* stream);

/* A unary factor's belief is normalize(table) (LBP.py:540) whatever the messages say, so its expected
 * features  E[row][k] = sum_x b_x phi[x][obs][k]  depend on the table ROW only; when many factors share
 * rows (the reference's layout: rows are columns of the shared pots) they are worth computing once per
 * row.  row_kind / row_obs: DEVICE int32 [n_rows], the phi selector (0 / 1 / 2) and observed column every
 * user of that row has; phi_*_t as in mlbp_gradient_args ([column][x][F]); out: [n_rows][8] (first F used).
 * Rows with an out-of-range kind or column are skipped and raise mlbp_gradient_status. */
int mlbp_unary_expectations_f64(const double* unary_tables, int32_t n_rows, int32_t X, const int32_t* row_kind,
                                const int32_t* row_obs, const double* phi_en_en_t, const double* phi_en_en_w1_t,
                                const double* phi_en_de_t, int32_t F_ee, int32_t F_ed, int32_t Vde, double* out,
                                void* stream);
int mlbp_gradient_status(void); /* synchronising read-and-reset: 1 = some factor was skipped        */

/* Per-instance sparse feature planes (train_mp.py:178-217: 'correct', 'full_history', 'hit_history'
 * are written into phi_en_de per instance and are non-zero in a few cells).  Row r of `out` becomes a
 * private copy of base table row base_row[r] with
 *     out[r][x] = base[x] * exp( sum over items q of row r with item_x[q] == x of theta[item_k[q]] * item_val[q] )
 * items of row r are item_off[r] .. item_off[r+1].  All index arrays are DEVICE arrays the caller has
 * range-checked (base_row < rows of base_tables, item_x < X, item_k < length of theta). */
int mlbp_patch_unary_tables_f64(const double* base_tables, const int32_t* base_row, const int32_t* item_off,
                                const int32_t* item_x, const int32_t* item_k, const double* item_val,
                                const double* theta, int32_t n_rows, int32_t X, double* out, void* stream);

/* Gradient share of those cells, ADDED to grad[row_graph[r]][item_k]:
 *     val * ( [item_x == row_label[r]] - t[item_x] / sum(t) ),   t = private row r
 * (FactorNode.get_gradient on the patched cells, LBP.py:600-603).  grad: device [B][F]. */
int mlbp_patch_gradient_f64(const double* priv_tables, const int32_t* item_off, const int32_t* item_x,
                            const int32_t* item_k, const double* item_val, const int32_t* row_graph,
                            const int32_t* row_label, int32_t n_rows, int32_t X, int32_t F, double* grad, void* stream);

/* out[j] = sum over rows of in[rows][cols], fixed summation order (bitwise reproducible): the
 * device half of batch_sgd_accumulate (train_mp.py:405-424). */
int mlbp_sum_rows_f64(const double* in, int64_t rows, int32_t cols, double* out, void* stream);
/* The same over up to three arrays [rows][cols_i] read side by side (out has cols0 + cols1 + cols2 entries, plus one
 * more holding `rows` when append_count is set; in1 / in2 may be NULL with cols 0): the trainer's fused statistics
 * [sum grad_en_en | sum grad_en_de | sum log-posterior | count] without assembling a matrix. */
int mlbp_sum_rows_cat_f64(const double* in0, int32_t cols0, const double* in1, int32_t cols1, const double* in2,
                          int32_t cols2, int64_t rows, int32_t append_count, double* out, void* stream);

/* The same sums over the rows with key[b] == *key_value only (key: DEVICE int32 [rows]; key_value: DEVICE int32 [1], read by the
 * launch -- so one captured launch serves every value); with append_count the extra entry holds the NUMBER OF SELECTED rows.
 * key == NULL: mlbp_sum_rows_cat_f64.  The trainer's minibatches of a resident shard: key = the minibatch each instance belongs
 * to this epoch, *key_value = the current minibatch (train_mp.py:631-649 updates once per instance; a minibatch of k is its
 * k instances' steps summed, 405-424). */
int mlbp_select_sum_rows_cat_f64(const double* in0, int32_t cols0, const double* in1, int32_t cols1, const double* in2,
                                 int32_t cols2, int64_t rows, const int32_t* key, const int32_t* key_value,
                                 int32_t append_count, double* out, void* stream);

/* One optimisation step's batch statistics in ONE launch: out = [sum_b grad_en_en[b][:] | sum_b grad_en_de[b][:] |
 * sum_b log-posterior(b) | B], the log-posterior of FactorGraph.get_posterior_probs (LBP.py:247-259: sum_v
 * log marginals[b][v][labels[b][v]], -inf replaced by -99.99) computed on the fly -- what mlbp_log_posterior_f64 followed
 * by mlbp_sum_rows_cat_f64 produce, the same bits (train_mp.py:405-424's accumulation).  lp_out: optional DEVICE [B], receives the
 * per-graph log-posteriors.  A label outside [0, X) is skipped and raises mlbp_gradient_status. */
int mlbp_step_statistics_f64(const double* grad_en_en, int32_t F_ee, const double* grad_en_de, int32_t F_ed, const double* marginals,
                             const int32_t* labels, int32_t n_vars, int32_t X, int64_t B, double* lp_out, double* out, void* stream);

/* out[s][j] = sum of in[b][j] over the rows with seg_id[b] == s (DEVICE int32 [rows], values in [0, n_seg)), fixed
 * order: the per-domain sums of batch_sgd_accumulate under --user_adapt / --experience_adapt (train_mp.py:413-415). */
int mlbp_segment_sum_rows_f64(const double* in, int64_t rows, int32_t cols, const int32_t* seg_id, int32_t n_seg,
                              double* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DEVICE: array primitives (the c_array_utils surface), batched over `batch` independent items
 * ------------------------------------------------------------------------------------------- */
/* au.dense_dot (c_array_utils.pyx:90-91): C[b] = A[b] (M x K) . B[b] (K x N), arbitrary element
 * strides so transposed views (msg.m.T, LBP.py:518) are accepted. Strides are in ELEMENTS. */
int mlbp_dense_dot_f64(int32_t batch, int32_t M, int32_t K, int32_t N,
                       const double* A, int64_t a_batch, int64_t a_row, int64_t a_col,
                       const double* B, int64_t b_batch, int64_t b_row, int64_t b_col,
                       double* C, int64_t c_batch, int64_t c_row, void* stream);

/* au.pointwise_multiply / au.dense_pointwise_multiply (c_array_utils.pyx:12-16, 93-94), with the
 * optional nan_to_num of LBP.pointwise_multiply (LBP.py:728-729). Contiguous n elements. */
int mlbp_pointwise_multiply_f64(const double* a, const double* b, double* out, int64_t n,
                                int32_t nan_to_num, void* stream);

/* au.normalize (c_array_utils.pyx:29-40) and Message.renormalize (LBP.py:649-657) over `batch`
 * contiguous vectors of n elements: total > 0 -> x / total; otherwise
 *   mode MLBP_NORM_ZERO    -> zeros   (au.normalize's m1.fill(0))
 *   mode MLBP_NORM_UNIFORM -> 1/n     (Message.renormalize)
 * positive[b] (device int32, may be NULL) receives 1 when the total was > 0. in may equal out. */
enum { MLBP_NORM_ZERO = 0, MLBP_NORM_UNIFORM = 1 };
int mlbp_normalize_f64(const double* in, double* out, int32_t batch, int64_t n, int32_t mode,
                       int32_t* positive, void* stream);

/* Top-K selection used by the reference's approximate ("sparse") primitives, K = 100
 * (c_array_utils.pyx:118,194; np.argpartition(-v, K-1)[:K]).  Writes the indices of the K largest
 * entries of the strided vector v to idx (device int32 [K]) in descending value order, ties broken
 * by lower index (the reference leaves tie order unspecified).  K > n fails with MLBP_EINVAL and the
 * text of NumPy's error, "kth(=K-1) out of bounds (n)". */
int mlbp_topk_f64(const double* v, int64_t stride, int32_t n, int32_t K, int32_t* idx, void* stream);

/* The same selection for every row of a contiguous [rows][n] matrix (idx: device int32 [rows][K]):
 * VariableNode.get_max_vocab for a batch of marginals (LBP.py:402-411). */
int mlbp_topk_rows_f64(const double* v, int64_t rows, int32_t n, int32_t K, int32_t* idx, void* stream);

/* au.sparse_vec_mat_dot (c_array_utils.pyx:193-205) given the selected indices:
 *   vec_is_row = 0: out[i] = sum_q mat[i][idx[q]] * vec[idx[q]]     (mat[:, idx] . vec[idx])
 *   vec_is_row = 1: out[j] = sum_q vec[idx[q]] * mat[idx[q]][j]     (vec[0, idx] . mat[idx, :])
 * mat strides in elements. */
int mlbp_sparse_vec_mat_dot_f64(const double* vec, int64_t vstride, const double* mat, int64_t m_row, int64_t m_col,
                                int32_t n_out, const int32_t* idx, int32_t K, int32_t vec_is_row, double* out,
                                void* stream);

/* au.sparse_dot (c_array_utils.pyx:117-129): out (n x n, fully written) = zeros with the
 * (cidx x ridx) block set to c[i] * r[j]. */
int mlbp_sparse_dot_f64(const double* c, const double* r, int32_t n, const int32_t* cidx, const int32_t* ridx,
                        int32_t K, double* out, void* stream);

/* au.sparse_pointwise_multiply (c_array_utils.pyx:108-114): out = zeros, block = sparse_m * dense_m. */
int mlbp_sparse_pointwise_multiply_f64(const double* sparse_m, const double* dense_m, int32_t n_rows, int32_t n_cols,
                                       const int32_t* cidx, int32_t Kc, const int32_t* ridx, int32_t Kr, double* out,
                                       void* stream);

/* au.sparse_normalize (c_array_utils.pyx:23-26): divides the block by its own sum IN PLACE (no
 * zero-sum guard, like the reference).  scratch1: device double[1]. */
int mlbp_sparse_normalize_f64(double* m, int32_t n_cols, const int32_t* cidx, int32_t Kc, const int32_t* ridx,
                              int32_t Kr, double* scratch1, void* stream);

/* Potential construction, the step just before the path (train_mp.py:220-255):
 * pot[i][j] = exp(sum_k phi[i][j][k] * theta[k]).  pot (row-major [rows][cols]) and/or pot_t (its
 * transpose [cols][rows], so a unary factor's column PotentialTable.slice_potentials takes,
 * LBP.py:702-703, is one contiguous row) may be NULL.  theta: device [F]. */
int mlbp_potentials_f64(const double* phi, const double* theta, int32_t rows, int32_t cols, int32_t F, double* pot,
                        double* pot_t, void* stream);

/* The same for up to MLBP_POTENTIALS_MAX_JOBS feature sets and n_rep parameter vectors in ONE launch: what
 * create_factor_graph does per instance with (phi_en_en, phi_en_en_w1, phi_en_de) and the instance's theta -- the global
 * one, or under --user_adapt / --experience_adapt the domain's (train_mp.py:220-255).  Job j, repetition r:
 * theta = jobs[j].theta + r * theta_stride, outputs at pot + r * pot_stride and pot_t + r * pot_t_stride (strides in
 * doubles; pot / pot_t may be NULL).  `jobs` is a HOST array (copied by value into the launch). */
#define MLBP_POTENTIALS_MAX_JOBS 4
typedef struct mlbp_potentials_job {
  const double* phi;       /* [rows][cols][F] */
  const double* theta;     /* [n_rep] vectors of F, theta_stride apart */
  double* pot;             /* [n_rep] x [rows][cols], or NULL */
  double* pot_t;           /* [n_rep] x [cols][rows], or NULL */
  int64_t theta_stride, pot_stride, pot_t_stride;
  int32_t rows, cols, F, reserved;
  double* expect;          /* [n_rep] x [cols][8], or NULL (rows == 64 only): expect[j][k] = sum_i b_i phi[i][j][k] with
                              b = au.normalize(pot[:, j]) -- the expected features of a unary factor observed at column j
                              (LBP.py:540, 600-603), i.e. what mlbp_unary_expectations_f64 computes for row j of pot_t, here
                              from the values the launch has in registers anyway (one launch less per optimisation step) */
  int64_t expect_stride;   /* doubles between two repetitions' expect arrays */
} mlbp_potentials_job;
int mlbp_potentials_multi_f64(const mlbp_potentials_job* jobs, int32_t n_jobs, int32_t n_rep, void* stream);

/* Elementwise natural log (np.log at LBP.py:139, 252, 408, 411: log-marginal read-outs). */
int mlbp_log_f64(const double* in, double* out, int64_t n, void* stream);

/* FactorNode.cell_gradient (LBP.py:615-619): out = onehot(cell) - beliefs over n contiguous cells. */
int mlbp_observed_minus_f64(const double* beliefs, int64_t n, int64_t cell, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLBP_H */
