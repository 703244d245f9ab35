// Shared-table sweeps on the gfx950 matrix cores (X = 64, float64).
//
// When every graph of the batch reads the SAME pairwise table for factor p -- the reference's own
// layout: one pot_en_en / pot_en_en_w1 array per FactorGraph, shared by all of its pairwise factors
// (LBP.py:456-467, 695-710) and, at one theta, by every instance of a minibatch (train_mp.py:178-255)
// -- the factor->variable update of G graphs is a dense contraction  OUT[64 x G] = T[64 x 64] . M[64 x G]
// (or T^T . M), i.e. exactly the case SURVEY.md section 8(d) prices against the MFMA peak instead of HBM.
//
// One 256-thread workgroup owns G = 16 graphs for all sweeps of the call:
//   * wave w holds rows 16w..16w+15 of every distinct table in BOTH orientations as
//     v_mfma_f64_16x16x4_f64 A-fragments in registers (16 doubles per table and orientation) for the
//     whole launch -- the tables are read once per workgroup, from L2;
//   * messages live in LDS as [state][graph] tiles (8 KiB): a tile read 64 lanes wide IS the B operand
//     of k-step s (lane l = state 4s + (l >> 4), graph l & 15), and the D fragment of wave w (lane l,
//     register r = state 16w + (l >> 4) + 4r, graph l & 15) is stored straight back into that layout;
//   * a factor->variable message is kept UNNORMALISED with its four per-wave partial column sums
//     beside it; readers multiply by the reciprocal of the total, so an update needs ONE barrier
//     (the scale of the variable->factor input cancels in normalise(T.m), LBP.py:509-524, 649-657);
//   * only the slots the sweeps read or write are resident ("live" tiles): unary messages are constants
//     (LBP.py:494-498), written back once in the prologue and folded into one product per variable.
// Degenerate graphs (zero / non-finite totals, where Message.renormalize and nan_to_num take their
// special branches) are flagged per graph and redone by the exact kernel, like the scale-free path.
//
// Flops per pairwise update per graph: 2 * 64 * 64 = 8192 (shared-table mode).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include "mlbp_internal.h"

namespace mlbp {

// ------------------------------------------------------------------------------------------------
// host: live-tile form of the fused program
// ------------------------------------------------------------------------------------------------
void build_shared_program(const FusedProgram& fp, int n_msgs, int P, int U, SharedProgram& out) {
  out = SharedProgram();
  const int n_hoist = (int)fp.hoist.size() / 2;
  out.why = "unary messages are not all constant, or no / too many pairwise factors";
  if (fp.has_unary_fops || n_hoist != U || P < 1 || P > 16 || U > 64) return;
  const int n_all = n_msgs + 1 + fp.n_cprod;
  out.hoisted.assign(n_msgs, -1);
  for (int h = 0; h < n_hoist; ++h) out.hoisted[fp.hoist[2 * h + 1]] = fp.hoist[2 * h];
  out.why = "a unary message is folded into no variable update";
  {
    std::vector<char> in_list(n_msgs, 0);
    size_t at = 0;
    for (int k = 0; k < fp.n_cprod; ++k) {
      const int cnt = fp.cpw[at];
      out.cprods.emplace_back(fp.cpw.begin() + at + 1, fp.cpw.begin() + at + 1 + cnt);
      for (int c : out.cprods.back()) {
        if (c < 0 || c >= n_msgs || out.hoisted[c] < 0) return;
        in_list[c] = 1;
      }
      at += 1 + cnt;
    }
    for (int c = 0; c < n_msgs; ++c)
      if (out.hoisted[c] >= 0 && !in_list[c]) return;     // a unary message no variable update folds in
  }
  // Sweep boundaries mean nothing to this kernel (it runs the updates in order), so the whole call is one
  // sequence.  Two rewrites keep the variable->factor messages out of LDS:
  //   1. a pairwise update whose input message c was produced by a variable->factor update whose own inputs
  //      have not changed since RECOMPUTES c in registers (fused pair) instead of reading a stored tile --
  //      the up pass of a loopy schedule (LBP.py:227-233) emits "X7->F17, X4->F14, F17->X1, F14->X1", and the
  //      sweep that follows may read X4->F14 once more;
  //   2. a lone variable->factor update whose output is rewritten later and not read before that is dropped.
  // Both leave every stored value exactly what the original order computes.
  std::vector<int32_t> fops;                                   // transformed op list, 8 words each
  {
    std::vector<std::vector<int32_t>> seq;
    for (size_t sw = 0; sw + 1 < fp.fsweeps.size(); sw += 2)
      for (int i = fp.fsweeps[sw]; i < fp.fsweeps[sw] + fp.fsweeps[sw + 1]; ++i) {
        seq.emplace_back(fp.fops.begin() + 8 * (size_t)i, fp.fops.begin() + 8 * (size_t)i + 8);
        seq.back()[0] &= 0xFF;
      }
    auto is_pair = [](const std::vector<int32_t>& w) { return w[0] == FOP_PAIR_TM || w[0] == FOP_PAIR_MT; };
    auto writes = [&](const std::vector<int32_t>& w, int slot) {
      if (is_pair(w) || w[0] == FOP_VAR) return w[3] == slot;
      return w[3] == slot || w[5] == slot;
    };
    for (size_t j = 0; j < seq.size(); ++j) {
      if (!is_pair(seq[j])) continue;
      const int c = seq[j][2];
      int i = (int)j - 1;
      while (i >= 0 && !writes(seq[i], c)) --i;
      if (i < 0 || is_pair(seq[i]) || seq[i][3] != c) continue;               // never written, or not by a variable update
      const std::vector<int32_t> v = seq[i];
      bool legal = true;
      for (size_t k = i + 1; k < j && legal; ++k) {
        for (int q = 0; q < v[2] && legal; ++q) if (writes(seq[k], fp.psrcs[v[1] + q])) legal = false;
        for (int q = 0; q < v[7] && legal; ++q) if (writes(seq[k], fp.psrcs[v[6] + q])) legal = false;
      }
      if (!legal) continue;
      const std::vector<int32_t> pr = seq[j];
      seq[j] = {pr[0] == FOP_PAIR_TM ? FOP_VAR_PAIR_TM : FOP_VAR_PAIR_MT, v[1], v[2], c, pr[1], pr[3], v[6], v[7]};
    }
    for (size_t i = 0; i < seq.size();) {
      if (seq[i][0] != FOP_VAR) { ++i; continue; }
      const int c = seq[i][3];
      bool dead = false;
      for (size_t k = i + 1; k < seq.size(); ++k) {
        if (is_pair(seq[k]) && seq[k][2] == c) break;                          // still read from its tile
        if (writes(seq[k], c)) { dead = true; break; }
      }
      if (dead) seq.erase(seq.begin() + i);
      else ++i;
    }
    out.sweeps.push_back(0);
    out.sweeps.push_back((int)seq.size());
    for (auto& w : seq) fops.insert(fops.end(), w.begin(), w.end());
  }
  const int n_ops = (int)fops.size() / 8;
  out.why = "unsupported update kind or slot use";
  std::vector<char> live(n_all, 0), written(n_msgs, 0);
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* w = &fops[8 * i];
    const int kind = w[0] & 0xFF;
    if (kind == FOP_PAIR_TM || kind == FOP_PAIR_MT) {
      if (written[w[2]]) live[w[2]] = 1;        // else: still the initial uniform message, no tile needed
      else fops[8 * i + 2] = -1;
      live[w[3]] = 1; written[w[3]] = 1;
    } else if (kind == FOP_VAR || kind == FOP_VAR_PAIR_TM || kind == FOP_VAR_PAIR_MT) {
      for (int q = 0; q < w[2]; ++q) live[fp.psrcs[w[1] + q]] = 1;
      written[w[3]] = 1;
      if (kind != FOP_VAR) { live[w[5]] = 1; written[w[5]] = 1; }
    } else {
      return;
    }
  }
  for (int c = 0; c < n_msgs; ++c)
    if (out.hoisted[c] >= 0 && (live[c] || written[c])) return;
  out.written = written;
  // tile numbering: constant products and factor->variable messages first, stored variable->factor messages
  // (each read once, by a later pairwise update) last -- when LDS cannot hold every tile the tail lives in
  // global memory (launcher: n_res resident tiles)
  out.live_of_slot.assign(n_all, -1);
  {
    std::vector<char> is_vf(n_all, 0);
    for (int i = 0; i < n_ops; ++i)
      if ((fops[8 * i] & 0xFF) >= FOP_VAR) is_vf[fops[8 * i + 3]] = 1;
    for (int pass = 0; pass < 2; ++pass)
      for (int s = 0; s < n_all; ++s)
        if (live[s] && (is_vf[s] ? 1 : 0) == pass) out.live_of_slot[s] = out.n_live++;
  }
  // ops + source lists
  std::vector<int32_t> ops((size_t)n_ops * 8, 0), lists;
  std::vector<int> last_var_write(n_msgs, -1);
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* w = &fops[8 * i];
    int32_t* so = &ops[8 * (size_t)i];
    const int kind = w[0] & 0xFF;
    so[0] = kind;
    if (kind == FOP_PAIR_TM || kind == FOP_PAIR_MT) {
      so[1] = w[2] < 0 ? -1 : out.live_of_slot[w[2]]; so[4] = w[1]; so[5] = out.live_of_slot[w[3]];
    } else {
      while (lists.size() % 4) lists.push_back(0);
      so[1] = (int)lists.size(); so[2] = w[2];
      for (int q = 0; q < w[2]; ++q) lists.push_back(out.live_of_slot[fp.psrcs[w[1] + q]]);
      so[3] = out.live_of_slot[w[3]]; so[6] = w[3];
      if (kind != FOP_VAR) { so[4] = w[4]; so[5] = out.live_of_slot[w[5]]; }
      if (so[3] < 0) last_var_write[w[3]] = i;
    }
  }
  for (int c = 0; c < n_msgs; ++c)
    if (last_var_write[c] >= 0) ops[8 * (size_t)last_var_write[c]] |= 0x200;     // write this v->f message out here
  while (lists.size() % 4) lists.push_back(0);
  // constant products, flattened: {unary factor, message slot, tile, 1 = first | 2 = last of its product}
  std::vector<int32_t> ent;
  for (int k = 0; k < fp.n_cprod; ++k) {
    const int tile = out.live_of_slot[n_msgs + 1 + k];
    if (tile < 0 || out.cprods[k].empty()) return;
    for (size_t q = 0; q < out.cprods[k].size(); ++q) {
      const int c = out.cprods[k][q];
      ent.insert(ent.end(), {out.hoisted[c], c, tile, (q == 0 ? 1 : 0) | (q + 1 == out.cprods[k].size() ? 2 : 0)});
    }
  }
  std::vector<char> is_vf(n_msgs, 0);                          // slots some variable->factor update writes
  for (int i = 0; i < n_ops; ++i)
    if ((fops[8 * i] & 0xFF) >= FOP_VAR) is_vf[fops[8 * i + 3]] = 1;
  std::vector<int32_t> back, fill;
  for (int c = 0; c < n_msgs; ++c) {
    if (out.hoisted[c] >= 0) continue;
    if (written[c] && out.live_of_slot[c] >= 0) { back.push_back(out.live_of_slot[c]); back.push_back(c | (is_vf[c] ? 0x40000000 : 0)); }
    else if (!written[c]) fill.push_back(c);
  }
  out.n_ops = n_ops; out.n_lists = (int)lists.size(); out.n_cpw = (int)ent.size();
  out.n_back = (int)back.size() / 2; out.n_fill = (int)fill.size();
  out.image = ops;
  out.image.insert(out.image.end(), lists.begin(), lists.end());
  out.image.insert(out.image.end(), ent.begin(), ent.end());
  out.image.insert(out.image.end(), back.begin(), back.end());
  out.image.insert(out.image.end(), fill.begin(), fill.end());
  out.off_sweeps = (int)out.image.size();
  out.image.insert(out.image.end(), out.sweeps.begin(), out.sweeps.end());
  out.image.push_back(0);
  out.why = "";
  out.ok = true;
}

bool build_shared_readout(const SharedProgram& sp, int n_msgs, int n_vars, const int32_t* in_off, const int32_t* in_slots,
                          std::vector<int32_t>& image) {
  // layout: offset of variable v's list [n_vars], then per variable {base tile or -1, n, tiles...}
  image.assign(n_vars, 0);
  while (image.size() % 4) image.push_back(0);
  for (int v = 0; v < n_vars; ++v) {
    std::vector<int32_t> consts, tiles;
    for (int q = in_off[v]; q < in_off[v + 1]; ++q) {
      const int c = in_slots[q];
      if (sp.hoisted[c] >= 0) consts.push_back(c);
      else if (sp.live_of_slot[c] >= 0) tiles.push_back(sp.live_of_slot[c]);
      else if (sp.written[c]) return false;            // written but not resident: not an incoming message we can read
      // else: never read and never written inside the sweeps -> still uniform, cancels in the normalisation
    }
    int base = -1;
    if (!consts.empty()) {
      for (size_t k = 0; k < sp.cprods.size(); ++k)
        if (sp.cprods[k] == consts) base = sp.live_of_slot[n_msgs + 1 + (int)k];
      if (base < 0) return false;
    }
    image[v] = (int)image.size();
    image.push_back(base);
    image.push_back((int)tiles.size());
    image.insert(image.end(), tiles.begin(), tiles.end());
    while (image.size() % 4) image.push_back(0);
  }
  return true;
}

namespace {

constexpr int WG = 256;
constexpr int G = 16;                 // graphs per workgroup = the N of v_mfma_f64_16x16x4_f64
constexpr int TILE = 64 * G;          // doubles per message tile

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {      // all 64 lanes, same bits everywhere
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
// Sum over the four lanes l, l^16, l^32, l^48 (= the four k-rows of one graph column), the same bits in all
// four: v_permlane16_swap / v_permlane32_swap (gfx950) exchange whole rows of 16 / halves of 32 lanes in the
// VALU -- with both operands equal the two results are the value of the even and of the odd partner row.
__device__ __forceinline__ double column_sum(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}
// A total a reader may divide by: finite, positive and far from the ends of the exponent range.
__device__ __forceinline__ bool total_ok(double t) { return t >= 1e-280 && t <= 1e280; }

typedef double double4_t __attribute__((ext_vector_type(4)));

struct SharedDev {
  const double* pair_tables;
  const int32_t* pair_tab;
  const double* unary_tables;
  const int32_t* unary_tab;
  double* msgs;                 // [B][n_msgs][64] or NULL (no write-back)
  double* marginals;            // [B][n_vars][64] or NULL
  int32_t* status;
  uint8_t* bail;
  const int32_t* image;
  const int32_t* fsweeps;
  const int32_t* readout;
  int32_t B, n_sweeps, n_msgs, P, U, n_pair_tables, n_unary_tables, n_vars;
  int32_t n_ops, n_live, n_lists, n_cpw, n_back, n_fill, n_readout;
  int32_t vf_only;              // write back only the variable->factor messages (what the gradient reads)
  int32_t n_res;                // tiles [0, n_res) live in LDS, the rest in `spill`
  double* spill;                // [workgroups][n_live - n_res][64][16] or NULL
  const double* tfrag;          // [n_pair_tables][2][4096] A fragments of every table (only when there are <= FRAG_TABLES), or NULL
};

#ifdef MLBP_STAMPS
__device__ unsigned long long* g_sh_stamp = nullptr;
__device__ int g_sh_ablate = 0;
#define ABL(bit) (abl_ & (1 << (bit)))
#define ABL_DECL const int abl_ = __builtin_amdgcn_readfirstlane(g_sh_ablate);
#define STAMP_DECL unsigned long long _t0 = 0, _ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t0) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#ifdef MLBP_STAMPS_LIGHT      // only the workgroup's lifetime (start -> last stamp): the shader clock the kernel runs at, undisturbed
#define STAMPV(i) if ((i) == 6) { unsigned long long _t1; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t1) :: "memory"); _ph[6] += _t1 - _t0; _t0 = _t1; }
#define STAMP(i)
#else
#define STAMPV(i) { unsigned long long _t1; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t1) :: "memory"); __builtin_amdgcn_sched_barrier(0); _ph[i] += _t1 - _t0; _t0 = _t1; }
#define STAMP(i) { unsigned long long _t1; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t1) :: "memory"); __builtin_amdgcn_sched_barrier(0); _ph[i] += _t1 - _t0; _t0 = _t1; }
#endif
#define STAMP_FLUSH if (g_sh_stamp && blockIdx.x < 64 && threadIdx.x == 0) { for (int _i = 0; _i < 8; ++_i) g_sh_stamp[blockIdx.x * 8 + _i] = _ph[_i]; }
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMPV(i)
#define ABL(bit) 0
#define ABL_DECL
#define STAMP_FLUSH
#endif

#define MLBP_MFMA16(A)                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 16; s_ += 2) {                            \
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s_], b[s_], acc0, 0, 0, 0);       \
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s_ + 1], b[s_ + 1], acc1, 0, 0, 0); \
  }

// out[table][0 | 1][wave w][k-step s][lane] = the A fragments of every table for T.m and m^T.T: element
// (16w + (lane & 15), 4s + (lane >> 4)) of T resp. T^T.
constexpr int FRAG_TABLES = 32;                                  // shared-table batches have a handful of tables
__global__ __launch_bounds__(WG) void table_fragments_kernel(const double* pair_tables, double* out) {
  const int ti = blockIdx.x >> 1, mt = blockIdx.x & 1;
  const double* T = pair_tables + (size_t)ti * 4096;
  double* o = out + ((size_t)ti * 2 + mt) * 4096;
  for (int e = threadIdx.x; e < 4096; e += WG) {
    const int lane = e & 63, s = (e >> 6) & 15, w = e >> 10;
    const int i = 16 * w + (lane & 15), k = 4 * s + (lane >> 4);
    o[e] = mt ? T[k * 64 + i] : T[i * 64 + k];
  }
}

template <int NTAB, bool SPILL>
__global__ __launch_bounds__(WG, 2) void sweep_x64_shared_kernel(SharedDev d) {
  extern __shared__ double lds[];
  double* tiles = lds;                                           // [n_res][64 states][16 graphs]
  double* tot = tiles + (size_t)d.n_res * TILE;                  // [n_live][4 waves][16 graphs] partial column sums
  double* spill = SPILL ? d.spill + (size_t)blockIdx.x * (d.n_live - d.n_res) * TILE : nullptr;
  // tile t: LDS when resident, else this workgroup's slice of the global spill area (same [state][graph] layout;
  // __syncthreads orders the workgroup's global accesses as it does the LDS ones)
  // (SPILL is a template parameter so that the all-resident instance keeps plain LDS instructions)
  auto TP = [&](int tile) -> double* {
    if (!SPILL) return tiles + (size_t)tile * TILE;
    return tile < d.n_res ? tiles + (size_t)tile * TILE : spill + (size_t)(tile - d.n_res) * TILE;
  };
  int32_t* img = reinterpret_cast<int32_t*>(tot + (size_t)d.n_live * 64);
  const int32_t* lists = img + d.n_ops * 8;
  const int32_t* ent = lists + d.n_lists;
  const int32_t* back = ent + d.n_cpw;
  const int32_t* fill = back + 2 * d.n_back;
  int32_t* rd = const_cast<int32_t*>(fill) + d.n_fill;           // read-out image
  int32_t* utab = rd + d.n_readout;                              // [16][U]
  int32_t* ptab = utab + G * d.U;                                // [P] table of factor p
  int32_t* preg = ptab + d.P;                                    // [P] register set of factor p
  int32_t* dist = preg + d.P;                                    // [NTAB] distinct tables, then {count, overflow}
  int32_t* gflag = dist + NTAB + 2;                              // [16] prologue verdict per graph

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int gl = lane & 15, cq = lane >> 4;                      // B/D operand: graph column, k-row
  const int g0 = blockIdx.x * G;
  const int gi = g0 + gl;
  const bool gvalid = gi < d.B;
  const int gc = gvalid ? gi : d.B - 1;                          // tail columns replay the last graph, outputs masked
  const double uniform = 1.0 / 64.0;

  STAMP_DECL
  ABL_DECL
  STAMP_START
  // ---- phase A: indices (range-checked), program image, uniform tiles ----
  bool ok = true, same = true;
  for (int i = t; i < G * d.U; i += WG) {
    const int gg = i / d.U, u = i - gg * d.U;
    const int v = d.unary_tab[(size_t)min(g0 + gg, d.B - 1) * d.U + u];
    ok &= (unsigned)v < (unsigned)d.n_unary_tables;
    utab[i] = v;
  }
  for (int i = t; i < G * d.P; i += WG) {
    const int gg = i / d.P, p = i - gg * d.P;
    const int v = d.pair_tab[(size_t)min(g0 + gg, d.B - 1) * d.P + p];
    ok &= (unsigned)v < (unsigned)d.n_pair_tables;
    same &= v == d.pair_tab[(size_t)g0 * d.P + p];
    if (gg == 0) ptab[p] = v;
  }
  {
    const int n_img = d.n_ops * 8 + d.n_lists + d.n_cpw + 2 * d.n_back + d.n_fill;
    for (int i = t; i < n_img; i += WG) img[i] = d.image[i];
    if (d.marginals)
      for (int i = t; i < d.n_readout; i += WG) rd[i] = d.readout[i];
    double2* dst = reinterpret_cast<double2*>(tiles);
    for (int i = t; i < d.n_res * (TILE / 2); i += WG) dst[i] = make_double2(uniform, uniform);
    if (SPILL)
      for (int i = t; i < (d.n_live - d.n_res) * (TILE / 2); i += WG) reinterpret_cast<double2*>(spill)[i] = make_double2(uniform, uniform);
    for (int i = t; i < d.n_live * 64; i += WG) tot[i] = 0.25;    // four partials of a total of 1
    if (t < G) gflag[t] = 0;
  }
  if (!__syncthreads_and(ok ? 1 : 0)) {
    if (t == 0) atomicExch(d.status, 1);
    return;
  }
  if (t == 0) {
    int nd = 0, over = 0;
    for (int p = 0; p < d.P; ++p) {
      int r = 0;
      while (r < nd && dist[r] != ptab[p]) ++r;
      if (r == nd) {
        if (nd < NTAB) dist[nd++] = ptab[p];
        else { over = 1; r = 0; }
      }
      preg[p] = r;
    }
    dist[NTAB] = nd; dist[NTAB + 1] = over;
  }
  if (!__syncthreads_and(same ? 1 : 0) || dist[NTAB + 1]) {      // not a shared-table batch: exact kernel takes all 16
    if (t < G && g0 + t < d.B) d.bail[g0 + t] = 4;
    return;
  }

  STAMPV(0)
  // ---- phase B: A fragments of every distinct table, both orientations (lane: row l&15, k l>>4) ----
  double aTM[NTAB][16], aMT[NTAB][16];
  const int nd = __builtin_amdgcn_readfirstlane(dist[NTAB]);
#pragma unroll
  for (int r = 0; r < NTAB; ++r) {
    const int ti = __builtin_amdgcn_readfirstlane(dist[r < nd ? r : 0]);
    if (d.tfrag) {
      // copies in operand order (table_fragments_kernel): one contiguous 512-byte read per fragment; read
      // straight from the row-major table the same fragments are 16 rows x 32 bytes per instruction and kept
      // the CU's address unit busy for ~20 us per workgroup
      const double* F = d.tfrag + (size_t)ti * 2 * 4096 + wave * 1024 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        aTM[r][s] = F[64 * s];
        aMT[r][s] = F[4096 + 64 * s];
      }
    } else {
      const double* T = d.pair_tables + (size_t)ti * 4096;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        aTM[r][s] = T[(16 * wave + gl) * 64 + 4 * s + cq];       // (T.m)[x]  : A[x][y] = T[x][y]
        aMT[r][s] = T[(4 * s + cq) * 64 + 16 * wave + gl];       // (m^T.T)[x]: A[x][y] = T[y][x]
      }
    }
  }

  STAMP(1)
  // ---- phase C: unary messages (constants) -> one product tile per variable ----
  // wave w takes graphs 4w..4w+3 with lanes = states.  Loads only (the write-back of these constants is a
  // separate streaming kernel): every load is unconditional (clamped index) so that the rows of the NEXT
  // batch stay in flight while the current one is reduced -- the only HBM reads of the kernel that miss L2.
  {
    // Four rotating buffers of HB rows: three batches are always in flight behind the one being reduced (a
    // batch costs ~0.4 us of arithmetic against ~2.5 us of load latency).  Every fetch is unconditional -- the
    // cursor clamps at the last batch -- so the wait before a batch is a fixed vmcnt and never drains the queue.
    constexpr int HB = 8;
    const int E = d.n_cpw / 4;
    const int cpg = (E + HB - 1) / HB, nb = 4 * cpg;             // batches per graph, batches of this wave
    double cur = 1.0;
    unsigned key = 0;                                            // largest high word met in the running product's rows
    // high word of 1e280: a row entry above it as an unsigned integer is negative, not finite, or too large to multiply on
    constexpr unsigned KEY_LIMIT = 0x7A11A0FCu;
    int fj = 0, fe = 0, pj = 0, pe = 0;                          // fetch / process cursors: graph 4w + j, first entry
    // entry e of the list lives in lane e: the per-row scalars come from v_readlane instead of a chain of
    // dependent LDS reads per row (three round trips per row were most of this phase)
    const int el = min(lane, E > 0 ? E - 1 : 0);
    const int ent_u = E > 0 ? ent[4 * el] : 0, ent_tile = E > 0 ? ent[4 * el + 2] : 0, ent_flags = E > 0 ? ent[4 * el + 3] : 0;
    int row_of = 0;                                              // table row of entry `lane` of the graph being fetched
    auto fetch = [&](double (&r)[HB]) {
      if (fe == 0) row_of = utab[(4 * wave + fj) * d.U + ent_u];
#pragma unroll
      for (int j = 0; j < HB; ++j)
        r[j] = d.unary_tables[(size_t)__builtin_amdgcn_readlane(row_of, min(fe + j, E - 1)) * 64 + lane];
      if (fe + HB < E) fe += HB;
      else if (fj < 3) { fe = 0; ++fj; }
    };
    auto process = [&](const double (&row)[HB]) {
      const int gg = 4 * wave + pj;
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        if (pe + j < E) {
          // The scale of a unary message cancels in everything downstream (only its normalised form is ever
          // stored, by unary_writeback_kernel), so the raw columns are multiplied and the PRODUCT is
          // normalised once (hardware reciprocal: only the magnitude matters).  A column Message.renormalize
          // would replace by the uniform vector (total <= 0, LBP.py:655-657) zeroes the product, and an entry that
          // is negative, not finite or huge shows in the high words: either sends the graph to the exact kernel,
          // decided once per product instead of once per row.
          const int flags = __builtin_amdgcn_readlane(ent_flags, pe + j);
          if (flags & 1) { cur = 1.0; key = 0; }
          const double r = row[j];
          key = max(key, (unsigned)__double2hiint(r));
          cur *= r;
          if (flags & 2) {
            const double s = ABL(2) ? 64.0 : wave_sum(cur);
            const bool bad = !total_ok(s) || __any(key > KEY_LIMIT);
            if (!ABL(1)) TP(__builtin_amdgcn_readlane(ent_tile, pe + j))[lane * G + gg] = cur * __builtin_amdgcn_rcp(s);
            if (bad) gflag[gg] = 1;
          }
        }
      }
      pe += HB;
      if (pe >= E) { pe = 0; ++pj; }
    };
    if (E > 0) {
      double r0[HB], r1[HB], r2[HB], r3[HB];
      fetch(r0); fetch(r1); fetch(r2);
      for (int i = 0; i < nb; i += 4) {
        fetch(r3); process(r0);
        fetch(r0); if (i + 1 < nb) process(r1);
        fetch(r1); if (i + 2 < nb) process(r2);
        fetch(r2); if (i + 3 < nb) process(r3);
      }
    }
  }
  if (d.msgs && !d.vf_only)                                       // slots the sweeps never touch stay uniform
    for (int i = t; i < d.n_fill * G * 64; i += WG) {
      const int x = i & 63, gg = (i >> 6) & (G - 1), k = i >> 10;
      if (g0 + gg < d.B) d.msgs[((size_t)(g0 + gg) * d.n_msgs + fill[k]) * 64 + x] = uniform;
    }
  __syncthreads();
  bool bad = gflag[gl] != 0;
  STAMPV(2)

  // ---- main loop: the same in all four waves; one barrier per update ----
  for (int sw = 0; sw < d.n_sweeps; ++sw) {
    const int op0 = d.fsweeps[2 * sw], op1 = op0 + d.fsweeps[2 * sw + 1];
    for (int o = op0; o < op1; ++o) {
      const int4 h0 = reinterpret_cast<const int4*>(img)[2 * o];
      const int4 h1 = reinterpret_cast<const int4*>(img)[2 * o + 1];
      const int kind = __builtin_amdgcn_readfirstlane(h0.x);
      const int k8 = kind & 0xFF;
      const int pslot = __builtin_amdgcn_readfirstlane(h1.x), dst = __builtin_amdgcn_readfirstlane(h1.y);
      double b[16];
      if (k8 >= FOP_VAR) {
        // variable -> factor (LBP.py:377-389): constant product (or uniform) times the other incoming messages
        const int a = __builtin_amdgcn_readfirstlane(h0.y), n = __builtin_amdgcn_readfirstlane(h0.z);
        {
          const double* src = TP(__builtin_amdgcn_readfirstlane(lists[a])) + lane;
#pragma unroll
          for (int s = 0; s < 16; ++s) b[s] = src[64 * s];
        }
        for (int q = 1; q < n && !ABL(7); ++q) {
          const int tl = __builtin_amdgcn_readfirstlane(lists[a + q]);
          const double* tp = tot + tl * 64 + gl;
          const double total = (tp[0] + tp[16]) + (tp[32] + tp[48]);
          bad |= !total_ok(total);
          // the scale of this product cancels downstream (every stored message is normalised by its own
          // total), the factor only keeps the magnitudes in range: the hardware reciprocal is enough
          const double inv = __builtin_amdgcn_rcp(total);
          const double* src = TP(tl) + lane;
#pragma unroll
          for (int s = 0; s < 16; ++s) b[s] *= src[64 * s] * inv;
        }
        STAMP(3)
        const int ct = __builtin_amdgcn_readfirstlane(h0.w);
        const bool out_now = (kind & 0x200) && d.msgs && !ABL(4);
        if (ct >= 0 || out_now || k8 == FOP_VAR) {               // the message itself is wanted: its total too
          const double part = (((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]))) +
                              (((b[8] + b[9]) + (b[10] + b[11])) + ((b[12] + b[13]) + (b[14] + b[15])));
          const double tb = ABL(5) ? 4.0 * part : column_sum(part);
          bad |= !total_ok(tb);
          if (ct >= 0) {                                         // read again later: keep it as a tile
#pragma unroll
            for (int s = 0; s < 16; ++s)
              if ((s >> 2) == wave) TP(ct)[64 * s + lane] = b[s];
            if (cq == 0) tot[ct * 64 + wave * 16 + gl] = 0.25 * tb;
          } else if (out_now && gvalid && !bad) {                // last value of this slot: straight to HBM
            double* out = d.msgs + ((size_t)gc * d.n_msgs + __builtin_amdgcn_readfirstlane(h1.z)) * 64 + cq;
            const double itb = 1.0 / tb;
#pragma unroll
            for (int s = 0; s < 16; ++s)
              if ((s >> 2) == wave) out[4 * s] = b[s] * itb;
          }
        }
        if (k8 == FOP_VAR) {
          __syncthreads();
          continue;
        }
      } else {
        const int tl = __builtin_amdgcn_readfirstlane(h0.y);
        if (tl >= 0) {
          const double* tp = tot + tl * 64 + gl;
          const double total = (tp[0] + tp[16]) + (tp[32] + tp[48]);
          bad |= !total_ok(total);
          const double inv = __builtin_amdgcn_rcp(total);
          const double* src = TP(tl) + lane;
#pragma unroll
          for (int s = 0; s < 16; ++s) b[s] = src[64 * s] * inv;
        } else {                                                 // a message nothing has updated yet (LBP.py:211-216)
#pragma unroll
          for (int s = 0; s < 16; ++s) b[s] = uniform;
        }
      }
      STAMP(7)
      // factor -> variable (LBP.py:500-524): 16 x v_mfma_f64_16x16x4_f64 against the resident fragments
      const bool mt = (k8 == FOP_PAIR_MT || k8 == FOP_VAR_PAIR_MT);
      const int sel = __builtin_amdgcn_readfirstlane(preg[pslot]) * 2 + (mt ? 1 : 0);
      double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      if (ABL(6)) { acc0.x = b[0]; acc0.y = b[5]; acc1.z = b[10]; acc1.w = b[15]; }
      else if (sel == 0) { MLBP_MFMA16(aTM[0]) }
      else if (sel == 1) { MLBP_MFMA16(aMT[0]) }
      else if (NTAB > 1 && sel == 2) { MLBP_MFMA16(aTM[NTAB > 1 ? 1 : 0]) }
      else if (NTAB > 1 && sel == 3) { MLBP_MFMA16(aMT[NTAB > 1 ? 1 : 0]) }
      else if (NTAB > 2 && sel == 4) { MLBP_MFMA16(aTM[NTAB > 2 ? 2 : 0]) }
      else if (NTAB > 2) { MLBP_MFMA16(aMT[NTAB > 2 ? 2 : 0]) }
      const double4_t acc = acc0 + acc1;
      double* out = TP(dst) + (16 * wave + cq) * G + gl;             // D: state 16w + (l>>4) + 4r
      out[0] = acc.x; out[4 * G] = acc.y; out[8 * G] = acc.z; out[12 * G] = acc.w;
      const double part = column_sum((acc.x + acc.y) + (acc.z + acc.w));
      if (cq == 0) tot[dst * 64 + wave * 16 + gl] = part;
      STAMP(4)
      __syncthreads();
      STAMP(5)
    }
  }

  // ---- epilogue: marginals, message write-back, verdicts ----
  // a bad total met only here (the last update's result) must reach the verdict of every wave
  if (d.marginals) {
    for (int v = wave; v < d.n_vars; v += 4) {
      const int at = rd[v];
      const int base = rd[at], n = rd[at + 1];
      double m[16];
      if (base >= 0) {
        const double* src = TP(base) + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] = src[64 * s];
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] = uniform;
      }
      for (int q = 0; q < n; ++q) {
        const int tl = rd[at + 2 + q];
        const double* tp = tot + tl * 64 + gl;
        const double total = (tp[0] + tp[16]) + (tp[32] + tp[48]);
        bad |= !total_ok(total);
        const double inv = 1.0 / total;
        const double* src = TP(tl) + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] *= src[64 * s] * inv;
      }
      double part = 0.0;
#pragma unroll
      for (int s = 0; s < 16; ++s) part += m[s];
      const double tm = column_sum(part);
      bad |= !total_ok(tm);
      if (gvalid && !bad) {
        double* out = d.marginals + ((size_t)gc * d.n_vars + v) * 64 + cq;
        const double itm = 1.0 / tm;
#pragma unroll
        for (int s = 0; s < 16; ++s) out[4 * s] = m[s] * itm;
      }
    }
  }
  for (int i = wave; i < d.n_back; i += 4) {
    const int tl = back[2 * i], slot = back[2 * i + 1] & 0x3FFFFFFF;
    const bool is_vf = (back[2 * i + 1] & 0x40000000) != 0;
    const double* tp = tot + tl * 64 + gl;
    const double total = (tp[0] + tp[16]) + (tp[32] + tp[48]);
    bad |= !total_ok(total);
    if (d.msgs && (!d.vf_only || is_vf) && gvalid && !bad) {
      const double* src = TP(tl) + lane;
      double* out = d.msgs + ((size_t)gc * d.n_msgs + slot) * 64 + cq;
      const double inv = 1.0 / total;
#pragma unroll
      for (int s = 0; s < 16; ++s) out[4 * s] = src[64 * s] * inv;
    }
  }
  STAMPV(6)
  STAMP_FLUSH
  if (bad && gvalid) d.bail[gi] = 2;                             // any wave that saw it says so (idempotent)
}

// Unary factor -> variable messages are constants (LBP.py:494-498): msgs[g][slot] = renormalize(column).  One
// wave per (graph, unary factor); pure streaming, enqueued behind the sweep kernel when the caller wants the
// message buffer filled.
__global__ __launch_bounds__(WG) void unary_writeback_kernel(const double* unary_tables, const int32_t* unary_tab,
                                                             const int32_t* ent, int E, int B, int U, int n_unary_tables,
                                                             int n_msgs, double* msgs) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * E) return;
  const int g = (int)(row / E), e = (int)(row % E);
  const int ti = unary_tab[(size_t)g * U + ent[4 * e]];
  if ((unsigned)ti >= (unsigned)n_unary_tables) return;       // the sweep kernel raises the status word for this
  const double v = unary_tables[(size_t)ti * 64 + lane];
  const double s = wave_sum(v);
  msgs[((size_t)g * n_msgs + ent[4 * e + 1]) * 64 + lane] = s > 0.0 ? v * (1.0 / s) : 1.0 / 64.0;
}

// ------------------------------------------------------------------------------------------------
// Pairwise part of FactorGraph.get_unregularized_gradeint (LBP.py:301-320, 528-619) for shared tables:
//   grad_k += phi[l0][l1][k] - (sum_ij c_i r_j T_ij phi_ijk) / (sum_ij c_i r_j T_ij)
// For 16 graphs at once  sum_j T_ij phi_ijk r_j  is the contraction (T (.) phi_k)[64x64] . R[64x16]: four of
// them per factor (k = 0..2 and the normaliser), then a dot with c along the rows.  c / r are the STORED
// variable->factor messages (the reference reads graph.messages, not a recomputation), straight from msgs.
// ------------------------------------------------------------------------------------------------
struct PairGradDev {
  const double* msgs; const double* pair_tables; const int32_t* pair_tab;
  const int32_t* c_slot; const int32_t* r_slot; const int32_t* pair_phi; const int32_t* pair_label;
  const double* phi[2];         // interleaved [64][64][3]: the label term
  const double* phi_p[2];       // planar [3][64][64]
  const double* wfrag;          // [table][which][4][4096] A fragments of T (.) phi_k and T (pair_weight_fragments_kernel)
  double* grad_en_en;           // [B][3], ADDED to (assigned when the unary part is done here too)
  int32_t* status;
  int32_t B, n_msgs, P, n_pair_tables;
  // unary factors by gather: phi[label][obs][:] - E[row][:]  (E from mlbp_unary_expectations_f64), or NULL
  const double* unary_expect; const int32_t* unary_tab; const int32_t* unary_kind; const int32_t* unary_obs;
  const int32_t* unary_label; const double* phi_ed; double* grad_en_de;
  int32_t U, n_unary_tables, Vde;
};

constexpr int PG_MAXP = 3;      // pairwise factors per pass: their two message tiles each stay in LDS (48 KiB)
constexpr int PG_MAXW = 32;     // tables the fragment scratch holds (x 2 feature tensors x 4 planes x 32 KiB = 8 MiB)

// W[table][which][k] = T (.) phi_which,k (k = 0..2) and T itself (k = 3), written in the order the MFMA A operand
// is read: [...][wave w][k-step s][lane] = element (row 16w + (lane & 15), column 4s + (lane >> 4)).  Every fragment
// load of the gradient kernel is then one contiguous 512-byte read (the direct form -- 16 rows x 32 bytes per
// instruction -- kept the CU's address unit busy for longer than the MFMAs took).
__global__ __launch_bounds__(WG) void pair_weight_fragments_kernel(const double* pair_tables, const double* phi_p0,
                                                                   const double* phi_p1, double* wfrag) {
  const int ti = blockIdx.x >> 3, which = (blockIdx.x >> 2) & 1, k = blockIdx.x & 3;
  const double* T = pair_tables + (size_t)ti * 4096;
  const double* ph = (which ? phi_p1 : phi_p0) + (size_t)k * 4096;
  double* out = wfrag + (((size_t)ti * 2 + which) * 4 + k) * 4096;
  for (int e = threadIdx.x; e < 4096; e += WG) {
    const int lane = e & 63, s = (e >> 6) & 15, w = e >> 10;
    const int idx = (16 * w + (lane & 15)) * 64 + 4 * s + (lane >> 4);
    out[e] = k < 3 ? T[idx] * ph[idx] : T[idx];
  }
}

__global__ __launch_bounds__(WG) void gradient_shared_pairs_kernel(PairGradDev d) {
  __shared__ double tile[PG_MAXP][2][64 * G];                  // [factor][r | c][state][graph]
  __shared__ double red[PG_MAXP][4][G][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int gl = lane & 15, cq = lane >> 4;
  const int g0 = blockIdx.x * G;
  const int gi = g0 + gl, gc = gi < d.B ? gi : d.B - 1;
  double out[3] = {0.0, 0.0, 0.0};                             // thread t < 16: graph g0 + t
  for (int p0 = 0; p0 < d.P; p0 += PG_MAXP) {
    const int np = d.P - p0 < PG_MAXP ? d.P - p0 : PG_MAXP;
    // stage the stored variable->factor messages: wave w reads whole 512-byte rows of graphs 4w..4w+3 and
    // writes them transposed (lane = state)
    for (int i = 0; i < np * 2 * 4; ++i) {
      const int pp = i >> 3, rc = (i >> 2) & 1, gg = 4 * wave + (i & 3);
      const int ggc = g0 + gg < d.B ? g0 + gg : d.B - 1;
      const int slot = rc ? d.c_slot[p0 + pp] : d.r_slot[p0 + pp];
      tile[pp][rc][lane * G + gg] = d.msgs[((size_t)ggc * d.n_msgs + slot) * 64 + lane];
    }
    __syncthreads();
    for (int pp = 0; pp < np; ++pp) {
      const int p = p0 + pp;
      const int ti = d.pair_tab[(size_t)g0 * d.P + p];
      const int tg = d.pair_tab[(size_t)gc * d.P + p];
      const int l0 = d.pair_label[((size_t)gc * d.P + p) * 2], l1 = d.pair_label[((size_t)gc * d.P + p) * 2 + 1];
      const bool valid = (unsigned)tg < (unsigned)d.n_pair_tables && (unsigned)l0 < 64u && (unsigned)l1 < 64u;
      if (!__all(valid)) {                                     // a bad index: skip the factor, tell the caller
        if (t == 0) atomicExch(d.status, 1);
        if (cq == 0) for (int k = 0; k < 4; ++k) red[pp][wave][gl][k] = 0.0;
        continue;
      }
      if (!__all(tg == ti)) {
        // the 16 graphs do not share this factor's table (e.g. a group straddling two domains of the trainer):
        // one graph at a time, every thread 16 cells of its table -- slow, correct, rare
        const double* php = d.phi_p[d.pair_phi[p] ? 1 : 0];
        for (int gg = 0; gg < G; ++gg) {
          const double* Tg = d.pair_tables + (size_t)d.pair_tab[(size_t)(g0 + gg < d.B ? g0 + gg : d.B - 1) * d.P + p] * 4096;
          double a[4] = {0.0, 0.0, 0.0, 0.0};
          for (int e = t; e < 4096; e += WG) {
            const double w = (tile[pp][1][(e >> 6) * G + gg] * tile[pp][0][(e & 63) * G + gg]) * Tg[e];
            a[3] += w;
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] += w * php[k * 4096 + e];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double v = wave_sum(a[k]);
            if (lane == 0) red[pp][wave][gg][k] = v;
          }
        }
        continue;
      }
      const double* W = d.wfrag + ((size_t)ti * 2 + (d.pair_phi[p] ? 1 : 0)) * 4 * 4096 + wave * 1024 + lane;
      const double* rt = tile[pp][0] + lane;                   // B operand: state 4s + (l >> 4), graph l & 15
      double4_t acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double b = rt[64 * s];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(W[k * 4096 + 64 * s], b, acc[k], 0, 0, 0);
      }
      const double* ct = tile[pp][1] + (16 * wave + cq) * G + gl;  // D rows 16w + cq + 4r
      const double c0 = ct[0], c1 = ct[4 * G], c2 = ct[8 * G], c3 = ct[12 * G];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double part = column_sum((c0 * acc[k].x + c1 * acc[k].y) + (c2 * acc[k].z + c3 * acc[k].w));
        if (cq == 0) red[pp][wave][gl][k] = part;
      }
    }
    __syncthreads();
    if (t < G && g0 + t < d.B) {
      const int gg = g0 + t;
      for (int pp = 0; pp < np; ++pp) {
        const int p = p0 + pp;
        const double (*rp)[G][4] = red[pp];
        const double Z = (rp[0][t][0 + 3] + rp[1][t][3]) + (rp[2][t][3] + rp[3][t][3]);
        const int m0 = d.pair_label[((size_t)gg * d.P + p) * 2], m1 = d.pair_label[((size_t)gg * d.P + p) * 2 + 1];
        if ((unsigned)m0 >= 64u || (unsigned)m1 >= 64u) continue;
        const int which = d.pair_phi[p] ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double S = (rp[0][t][k] + rp[1][t][k]) + (rp[2][t][k] + rp[3][t][k]);
          out[k] += d.phi[which][((size_t)m0 * 64 + m1) * 3 + k] - (Z > 0.0 ? S / Z : 0.0);     // au.normalize: zero-sum -> 0
        }
      }
    }
    __syncthreads();
  }
  if (d.unary_expect) {
    // unary factors: thread (graph t & 15, lane group t >> 4) takes factors u = t>>4, t>>4 + 16, ...; every term is
    // two short gathers (phi at the label, E of the table row), so the 16 groups hide each other's latency
    __shared__ double ured[16][G][9];
    const int ug = t & 15, uj = t >> 4, gg = g0 + ug;
    double acc[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (gg < d.B)
      for (int u = uj; u < d.U; u += 16) {
        const int kind = d.unary_kind[u];
        const int row = d.unary_tab[(size_t)gg * d.U + u], obs = d.unary_obs[(size_t)gg * d.U + u];
        const int lab = d.unary_label[(size_t)gg * d.U + u];
        const int cols = kind == 2 ? d.Vde : 64;
        if ((unsigned)row >= (unsigned)d.n_unary_tables || (unsigned)obs >= (unsigned)cols || (unsigned)lab >= 64u || (unsigned)kind > 2u) {
          atomicExch(d.status, 1);
          continue;
        }
        const double* E = d.unary_expect + (size_t)row * 8;
        if (kind == 2) {
          const double* pl = d.phi_ed + ((size_t)lab * cols + obs) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[3 + k] += pl[k] - E[k];
        } else {
          const double* pl = d.phi[kind] + ((size_t)lab * 64 + obs) * 3;
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] += pl[k] - E[k];
        }
      }
#pragma unroll
    for (int k = 0; k < 9; ++k) ured[uj][ug][k] = acc[k];
    __syncthreads();
    if (t < G && g0 + t < d.B) {
      double tot9[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        double v = 0.0;
        for (int j = 0; j < 16; ++j) v += ured[j][t][k];
        tot9[k] = v;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) d.grad_en_en[(size_t)(g0 + t) * 3 + k] = out[k] + tot9[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) d.grad_en_de[(size_t)(g0 + t) * 6 + k] = tot9[3 + k];
    }
  } else if (t < G && g0 + t < d.B) {
#pragma unroll
    for (int k = 0; k < 3; ++k) d.grad_en_en[(size_t)(g0 + t) * 3 + k] += out[k];
  }
}

std::mutex g_attr_mutex;

template <int NTAB, bool SPILL>
int launch(const SharedDev& d, size_t lds, hipStream_t st) {
  static size_t granted = 0;
  {
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (lds > granted) {
      hipError_t e = hipFuncSetAttribute((const void*)sweep_x64_shared_kernel<NTAB, SPILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(MLBP_EHIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      granted = lds;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sweep_x64_shared_kernel<NTAB, SPILL>, WG, lds) == hipSuccess)
        fail(MLBP_OK, "shared-table kernel <%d%s>: %zu bytes of LDS per workgroup, %d workgroups per CU", NTAB, SPILL ? ", spilling" : "", lds, per_cu);
    }
  }
  hipLaunchKernelGGL((sweep_x64_shared_kernel<NTAB, SPILL>), dim3((d.B + G - 1) / G), dim3(WG), lds, st, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MLBP_EHIP, "shared-table sweep launch failed: %s", hipGetErrorString(e));
  return MLBP_OK;
}

}  // namespace

int launch_shared_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream, bool* launched) {
  *launched = false;
  const SharedProgram& sp = prog->shared;
  if (!(a->flags & MLBP_SWEEP_SHARED_PAIR_TABLES)) return MLBP_OK;
  if (a->X != 64 || !a->normalize_messages || !a->init_messages)
    return fail(MLBP_OK, "shared-table kernel not used: needs X = 64, normalize_messages and init_messages");
  if (!sp.ok || !prog->d_simage) return fail(MLBP_OK, "shared-table kernel not used: %s", sp.why);
  if (a->marginals && !prog->d_sreadout)
    return fail(MLBP_OK, "shared-table kernel not used: a variable's constant messages match no folded product");
  const int n_readout = a->marginals ? prog->n_sreadout : 0;
  const int ntab = prog->P >= 2 ? 2 : 1;
  const size_t words = (size_t)sp.off_sweeps + 1 + n_readout + (size_t)G * prog->U + 2 * prog->P + 2 * ntab + 3 + G + 8;
  // resident tiles: as many as fit HALF the CU's LDS, so that two workgroups share a CU (constant products and
  // factor->variable messages come first in the numbering); the rest spill to global memory.  Measured on K4 user
  // graphs (21 tiles): 8 resident + 13 spilled with two workgroups per CU 0.216 ms, 16 resident + 5 spilled with
  // one 0.266 ms.
  const size_t fixed = (size_t)sp.n_live * 64 * sizeof(double) + words * sizeof(int32_t);
  int n_res = sp.n_live;
  while (n_res > 0 && fixed + (size_t)n_res * TILE * sizeof(double) > 80 * 1024) --n_res;
  const size_t lds = fixed + (size_t)n_res * TILE * sizeof(double);
  if (n_res < 1 || sp.n_live - n_res > 16 || lds > 160 * 1024)   // too much would spill: the per-graph kernels do better
    return fail(MLBP_OK, "shared-table kernel not used: %d live message tiles, %d fit LDS", sp.n_live, n_res);
  mlbp_program* mp = const_cast<mlbp_program*>(prog);
  const size_t spill_doubles = (size_t)((a->B + G - 1) / G) * (sp.n_live - n_res) * TILE;
  if (spill_doubles > mp->spill_cap) {             // first use at this size (a stream-capturing caller warms up first)
    (void)hipFree(mp->d_spill);
    mp->d_spill = nullptr; mp->spill_cap = 0;
    if (hipMalloc(&mp->d_spill, spill_doubles * sizeof(double)) != hipSuccess) return fail(MLBP_EHIP, "tile spill allocation failed");
    mp->spill_cap = spill_doubles;
  }
  if (mp->bail_cap < a->B)
    if (int e = mlbp_program_reserve(mp, a->B)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(mp->d_bail, 0, (size_t)a->B, st) != hipSuccess) return fail(MLBP_EHIP, "hipMemsetAsync failed");
  SharedDev d;
  d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab; d.unary_tables = a->unary_tables; d.unary_tab = a->unary_tab;
  d.msgs = ((a->flags & MLBP_SWEEP_NO_MESSAGE_WRITEBACK) && !a->gradient) ? nullptr : a->msgs;
  d.vf_only = ((a->flags & MLBP_SWEEP_NO_MESSAGE_WRITEBACK) && a->gradient) ? 1 : 0;
  d.marginals = a->marginals; d.status = prog->d_status; d.bail = mp->d_bail;
  d.image = prog->d_simage; d.fsweeps = prog->d_simage + sp.off_sweeps; d.readout = prog->d_sreadout;
  d.B = a->B; d.n_sweeps = (int)sp.sweeps.size() / 2; d.n_msgs = prog->n_msgs; d.P = prog->P; d.U = prog->U;
  d.n_pair_tables = a->n_pair_tables; d.n_unary_tables = a->n_unary_tables; d.n_vars = prog->n_vars;
  d.n_ops = sp.n_ops; d.n_live = sp.n_live; d.n_lists = sp.n_lists; d.n_cpw = sp.n_cpw; d.n_back = sp.n_back;
  d.n_fill = sp.n_fill; d.n_readout = n_readout;
  d.n_res = n_res; d.spill = n_res < sp.n_live ? mp->d_spill : nullptr;
  d.tfrag = nullptr;
  if (a->n_pair_tables <= FRAG_TABLES) {
    if (!mp->d_tfrag) {                            // first use (a stream-capturing caller warms up or reserves first)
      if (hipMalloc(&mp->d_tfrag, sizeof(double) * FRAG_TABLES * 2 * 4096) != hipSuccess)
        return fail(MLBP_EHIP, "fragment scratch allocation failed");
    }
    d.tfrag = mp->d_tfrag;
    hipLaunchKernelGGL(table_fragments_kernel, dim3(a->n_pair_tables * 2), dim3(WG), 0, st, a->pair_tables, mp->d_tfrag);
  }
  int e = d.spill ? (ntab == 2 ? launch<2, true>(d, lds, st) : launch<1, true>(d, lds, st))
                  : (ntab == 2 ? launch<2, false>(d, lds, st) : launch<1, false>(d, lds, st));
  if (e) return e;
  if (d.msgs && !d.vf_only && sp.n_cpw > 0) {
    const int E = sp.n_cpw / 4;
    const long long rows = (long long)a->B * E;
    hipLaunchKernelGGL(unary_writeback_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(WG), 0, st, a->unary_tables, a->unary_tab,
                       prog->d_simage + sp.n_ops * 8 + sp.n_lists, E, a->B, prog->U, a->n_unary_tables, prog->n_msgs, a->msgs);
    if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "unary write-back launch failed");
  }
  *launched = true;
  return MLBP_OK;
}

int launch_shared_pair_gradient(const mlbp_gradient_args* a, int32_t* status, void* stream) {
  PairGradDev d;
  d.msgs = a->msgs; d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab;
  d.c_slot = a->pair_c_slot; d.r_slot = a->pair_r_slot; d.pair_phi = a->pair_phi; d.pair_label = a->pair_label;
  d.phi[0] = a->phi_en_en; d.phi[1] = a->phi_en_en_w1; d.phi_p[0] = a->phi_en_en_p; d.phi_p[1] = a->phi_en_en_w1_p;
  d.grad_en_en = a->grad_en_en; d.status = status;
  d.B = a->B; d.n_msgs = a->n_msgs; d.P = a->P; d.n_pair_tables = a->n_pair_tables;
  d.unary_expect = (a->F_ed == 6 && a->U > 0) ? a->unary_expect : nullptr;
  d.unary_tab = a->unary_tab; d.unary_kind = a->unary_kind; d.unary_obs = a->unary_obs; d.unary_label = a->unary_label;
  d.phi_ed = a->phi_en_de; d.grad_en_de = a->grad_en_de; d.U = a->U; d.n_unary_tables = a->n_unary_tables; d.Vde = a->Vde;
  // fragment scratch: one device-wide buffer (launches on different streams must not overlap, like mlbp_sum_rows_f64)
  static double* wfrag = nullptr;
  static std::mutex wmutex;
  {
    std::lock_guard<std::mutex> lock(wmutex);
    if (!wfrag && hipMalloc(&wfrag, sizeof(double) * PG_MAXW * 2 * 4 * 4096) != hipSuccess)
      return fail(MLBP_EHIP, "shared-table pair gradient: scratch allocation failed");
  }
  d.wfrag = wfrag;
  hipLaunchKernelGGL(pair_weight_fragments_kernel, dim3(a->n_pair_tables * 8), dim3(WG), 0, (hipStream_t)stream, a->pair_tables,
                     a->phi_en_en_p, a->phi_en_en_w1_p, wfrag);
  hipLaunchKernelGGL(gradient_shared_pairs_kernel, dim3((a->B + G - 1) / G), dim3(WG), 0, (hipStream_t)stream, d);
  if (hipGetLastError() != hipSuccess) return fail(MLBP_EHIP, "shared-table pair gradient launch failed");
  return MLBP_OK;
}

#ifdef MLBP_STAMPS
extern "C" int mlbp_debug_set_shared_stamp_buffer(void* dev_ptr, int ablate_mask) {
  unsigned long long* p = (unsigned long long*)dev_ptr;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sh_ablate), &ablate_mask, sizeof(int)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sh_stamp), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace mlbp
