// Host-side integer logic of libmlbp.so: topology checks, loop test, BFS message schedule, message
// slot numbering and the sweep compiler.  Everything here must be bit-exact with the reference
// (LBP.py:155-190, 192-245); it is pinned against tests/golden/schedules.npz.
//
// No HIP calls in this file: these entry points work on a machine without a GPU.
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "mlbp_internal.h"

namespace mlbp {

static thread_local std::string g_last_error = "";

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// Validates a topology; every later routine may index without further checks.
static int check_topology(const mlbp_topology* t) {
  if (!t) return fail(MLBP_EINVAL, "topology is NULL");
  if (t->n_vars <= 0 || t->n_factors <= 0)
    return fail(MLBP_EINVAL, "topology needs at least one variable and one factor (LBP.py:193-194)");
  if (!t->fac_nvars || !t->fac_var || !t->fac_dim || !t->var_fac_off || !t->var_fac)
    return fail(MLBP_EINVAL, "topology has a NULL array");
  for (int f = 0; f < t->n_factors; ++f) {
    int nv = t->fac_nvars[f];
    if (nv != 1 && nv != 2)
      return fail(MLBP_EINVAL, "factor %d has %d variables; only unary and pairwise factors are supported (LBP.py:446-447)", f, nv);
    for (int k = 0; k < nv; ++k) {
      int v = t->fac_var[2 * f + k], d = t->fac_dim[2 * f + k];
      if (v < 0 || v >= t->n_vars) return fail(MLBP_EINVAL, "factor %d: variable index %d out of range", f, v);
      if (d < 0 || d >= nv) return fail(MLBP_EINVAL, "factor %d: table axis %d out of range", f, d);
    }
    if (nv == 2) {
      if (t->fac_var[2 * f] == t->fac_var[2 * f + 1])
        return fail(MLBP_EINVAL, "factor %d joins a variable to itself (LBP.py:444)", f);
      if (t->fac_dim[2 * f] == t->fac_dim[2 * f + 1])
        return fail(MLBP_EINVAL, "factor %d maps both variables to one table axis", f);
    }
  }
  if (t->var_fac_off[0] != 0) return fail(MLBP_EINVAL, "var_fac_off[0] must be 0");
  for (int v = 0; v < t->n_vars; ++v) {
    if (t->var_fac_off[v + 1] < t->var_fac_off[v]) return fail(MLBP_EINVAL, "var_fac_off not monotone");
    for (int i = t->var_fac_off[v]; i < t->var_fac_off[v + 1]; ++i) {
      int f = t->var_fac[i];
      if (f < 0 || f >= t->n_factors) return fail(MLBP_EINVAL, "variable %d: factor index %d out of range", v, f);
      bool member = false;
      for (int k = 0; k < t->fac_nvars[f]; ++k) member |= (t->fac_var[2 * f + k] == v);
      if (!member) return fail(MLBP_EINVAL, "variable %d lists factor %d which does not contain it", v, f);
    }
  }
  return MLBP_OK;
}

static inline bool is_var(const mlbp_topology* t, int node) { return node < t->n_vars; }

// Neighbours of a node in the order the reference iterates them: facset (creation order) for a
// variable, varset order for a factor (LBP.py:165-169, 185-187).
static void neighbours(const mlbp_topology* t, int node, std::vector<int>& out) {
  out.clear();
  if (is_var(t, node)) {
    for (int i = t->var_fac_off[node]; i < t->var_fac_off[node + 1]; ++i) out.push_back(t->n_vars + t->var_fac[i]);
  } else {
    int f = node - t->n_vars;
    for (int k = 0; k < t->fac_nvars[f]; ++k) out.push_back(t->fac_var[2 * f + k]);
  }
}

// LBP.py:155-172.  FIFO with duplicates; expansion on first dequeue; a neighbour that has not
// been EXPANDED yet (it may already be queued) yields a pair and is queued again.
static void bfs_schedule(const mlbp_topology* t, int root, std::vector<int>& pairs) {
  const int n_nodes = t->n_vars + t->n_factors;
  std::vector<char> expanded(n_nodes, 0);
  std::vector<int> fifo, nb;
  size_t head = 0;
  fifo.push_back(root);
  pairs.clear();
  while (head < fifo.size()) {
    int n = fifo[head++];
    if (expanded[n]) continue;
    expanded[n] = 1;
    neighbours(t, n, nb);
    for (int m : nb)
      if (!expanded[m]) { pairs.push_back(m); pairs.push_back(n); }
    for (int m : nb)
      if (!expanded[m]) fifo.push_back(m);
  }
}

struct Slots {
  std::vector<int> f2v, v2f, pair_slot, unary_slot;
  int n_msgs = 0, P = 0, U = 0;
};

static void number_slots(const mlbp_topology* t, Slots& s) {
  s.f2v.assign(2 * t->n_factors, -1);
  s.v2f.assign(2 * t->n_factors, -1);
  s.pair_slot.assign(t->n_factors, -1);
  s.unary_slot.assign(t->n_factors, -1);
  for (int f = 0; f < t->n_factors; ++f) {
    if (t->fac_nvars[f] == 1) {
      s.f2v[2 * f] = s.n_msgs++;
      s.unary_slot[f] = s.U++;
    } else {
      for (int k = 0; k < 2; ++k) {
        s.v2f[2 * f + k] = s.n_msgs++;
        s.f2v[2 * f + k] = s.n_msgs++;
      }
      s.pair_slot[f] = s.P++;
    }
  }
}

static int varset_pos(const mlbp_topology* t, int f, int v) {
  for (int k = 0; k < t->fac_nvars[f]; ++k)
    if (t->fac_var[2 * f + k] == v) return k;
  return -1;
}

struct OpSink {
  std::vector<int> ops, srcs;
};

// One message update frm -> to, as the reference dispatches it (LBP.py:227-243).
static void emit(const mlbp_topology* t, const Slots& s, int frm, int to, OpSink& out) {
  if (!is_var(t, to) && t->fac_nvars[to - t->n_vars] < 2) return;  // destination is a unary factor
  if (is_var(t, frm)) {
    // VariableNode.update_message_to(factor): every OTHER factor of the facset, in order.
    int f = to - t->n_vars, v = frm;
    int first = (int)out.srcs.size(), count = 0;
    for (int i = t->var_fac_off[v]; i < t->var_fac_off[v + 1]; ++i) {
      int g = t->var_fac[i];
      if (g == f) continue;
      out.srcs.push_back(s.f2v[2 * g + varset_pos(t, g, v)]);
      ++count;
    }
    int dst = s.v2f[2 * f + varset_pos(t, f, v)];
    out.ops.insert(out.ops.end(), {MLBP_OP_VAR, first, count, dst});
  } else {
    int f = frm - t->n_vars, v = to;
    int k = varset_pos(t, f, v);
    if (t->fac_nvars[f] == 1) {
      out.ops.insert(out.ops.end(), {MLBP_OP_UNARY, s.unary_slot[f], 0, s.f2v[2 * f]});
    } else {
      int ko = 1 - k;                       // the other variable
      int src = s.v2f[2 * f + ko];
      int kind = (t->fac_dim[2 * f + ko] == 1) ? MLBP_OP_PAIR_TM : MLBP_OP_PAIR_MT;  // LBP.py:503-518
      out.ops.insert(out.ops.end(), {kind, s.pair_slot[f], src, s.f2v[2 * f + k]});
    }
  }
}

}  // namespace mlbp

using namespace mlbp;

extern "C" {

int mlbp_version(void) { return MLBP_VERSION_MAJOR * 100 + MLBP_VERSION_MINOR; }
const char* mlbp_arch(void) { return "gfx950"; }
const char* mlbp_last_error(void) { return g_last_error.c_str(); }

int mlbp_has_loops(const mlbp_topology* t, int32_t root) {
  if (int e = check_topology(t)) return e;
  if (root < 0 || root >= t->n_vars) return fail(MLBP_EINVAL, "root variable %d out of range", root);
  // LBP.py:174-190: LIFO of (node, arrival node); revisiting any node means a cycle.
  const int n_nodes = t->n_vars + t->n_factors;
  std::vector<char> seen(n_nodes, 0);
  std::vector<std::pair<int, int>> stack;
  std::vector<int> nb;
  stack.push_back({root, -1});
  while (!stack.empty()) {
    auto [n, parent] = stack.back();
    stack.pop_back();
    if (seen[n]) return 1;
    seen[n] = 1;
    neighbours(t, n, nb);
    for (int m : nb)
      if (m != parent) stack.push_back({m, n});
  }
  return 0;
}

int mlbp_message_schedule(const mlbp_topology* t, int32_t root, int32_t* pairs, int32_t cap_pairs) {
  if (int e = check_topology(t)) return e;
  if (root < 0 || root >= t->n_vars) return fail(MLBP_EINVAL, "root variable %d out of range", root);
  std::vector<int> p;
  bfs_schedule(t, root, p);
  int n = (int)p.size() / 2;
  if (n > cap_pairs || (!pairs && n > 0)) return fail(MLBP_ENOMEM, "schedule has %d pairs, capacity %d", n, cap_pairs);
  for (size_t i = 0; i < p.size(); ++i) pairs[i] = p[i];
  return n;
}

int mlbp_message_slots(const mlbp_topology* t, int32_t* f2v, int32_t* v2f, int32_t* pair_slot,
                       int32_t* unary_slot) {
  if (int e = check_topology(t)) return e;
  Slots s;
  number_slots(t, s);
  for (int i = 0; i < 2 * t->n_factors; ++i) {
    if (f2v) f2v[i] = s.f2v[i];
    if (v2f) v2f[i] = s.v2f[i];
  }
  for (int f = 0; f < t->n_factors; ++f) {
    if (pair_slot) pair_slot[f] = s.pair_slot[f];
    if (unary_slot) unary_slot[f] = s.unary_slot[f];
  }
  return s.n_msgs;
}

int mlbp_compile_sweep(const mlbp_topology* t, int32_t root, int32_t* ops, int32_t cap_ops,
                       int32_t* srcs, int32_t cap_srcs, int32_t* n_srcs) {
  if (int e = check_topology(t)) return e;
  if (root < 0 || root >= t->n_vars) return fail(MLBP_EINVAL, "root variable %d out of range", root);
  Slots s;
  number_slots(t, s);
  std::vector<int> sched;
  bfs_schedule(t, root, sched);
  OpSink out;
  const int n = (int)sched.size() / 2;
  for (int i = n - 1; i >= 0; --i) emit(t, s, sched[2 * i], sched[2 * i + 1], out);  // child -> parent
  for (int i = 0; i < n; ++i) emit(t, s, sched[2 * i + 1], sched[2 * i], out);       // parent -> child
  const int n_ops = (int)out.ops.size() / 4;
  if (n_ops > cap_ops || (int)out.srcs.size() > cap_srcs)
    return fail(MLBP_ENOMEM, "sweep needs %d ops / %d srcs, capacity %d / %d", n_ops, (int)out.srcs.size(), cap_ops, cap_srcs);
  for (size_t i = 0; i < out.ops.size(); ++i) ops[i] = out.ops[i];
  for (size_t i = 0; i < out.srcs.size(); ++i) srcs[i] = out.srcs[i];
  if (n_srcs) *n_srcs = (int)out.srcs.size();
  return n_ops;
}

}  // extern "C"
