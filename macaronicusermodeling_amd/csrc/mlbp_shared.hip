// Shared-table sweeps on the gfx950 matrix cores (X = 64, float64).
//
// When every graph of the batch reads the SAME pairwise table for factor p -- the reference's own
// layout: one pot_en_en / pot_en_en_w1 array per FactorGraph, shared by all of its pairwise factors
// (LBP.py:456-467, 695-710) and, at one theta, by every instance of a minibatch (train_mp.py:178-255)
// -- the factor->variable update of G graphs is a dense contraction  OUT[64 x G] = T[64 x 64] . M[64 x G]
// (or T^T . M), i.e. exactly the case SURVEY.md section 8(d) prices against the MFMA peak instead of HBM.
//
// One 512-thread workgroup owns G = 16 graphs for all sweeps of the call (two halves of four waves; a bundle's two
// independent updates run side by side, one per half, under one barrier):
//   * wave (half, row block w) holds rows 16w..16w+15 of two of the four (table, orientation) pairs as
//     v_mfma_f64_16x16x4_f64 A-fragments in registers (16 doubles per pair) for the whole launch -- the tables
//     are read once per workgroup, from L2;
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
// Where every variable has at most two pairwise factors (K2, K3, chains, rings) the kernel runs its PRODUCT-FUSED
// form (template parameter PF; build_shared_program, "product-fused form"): the producer of a message stores
// c (.) message, a contraction reads one tile straight into the matrix cores, read-out and gradient from LDS.
//
// Flops per pairwise update per graph: 2 * 64 * 64 = 8192 (shared-table mode).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
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
  // Members: one record per update, MW words --
  //   [0] flags: 1 contraction, 2 m^T.T (else T.m), 4 the variable->factor product is kept as tile [3], 8 it is this
  //       slot's last value and goes straight to memory (message slot [4]), 16 variable update only; bits 8-11 = number of
  //       source tiles   [1] pair slot   [2] destination tile   [3] product tile or -1   [4] message slot of the product
  //   [8..15] source tiles (the first one -1: a message nothing has updated yet, i.e. the uniform vector)
  // (packed into 8 words for the device, pack_member below).
  // Bundles: two members that touch disjoint tiles share one barrier; the kernel runs them on different halves of its
  // eight waves when their (table, orientation) pairs live in different halves (it decides: the tables are device data).
  constexpr int MW = 16;
  std::vector<int32_t> mem((size_t)n_ops * MW, 0);
  std::vector<int> last_var_write(n_msgs, -1);
  std::vector<char> pair_kind(n_ops, 0);                        // a factor update that reads a STORED variable->factor message
  out.why = "more than 254 live tiles or 65535 message slots";
  if (out.n_live > 254 || n_msgs > 65535) return;
  out.why = "a variable update multiplies more than 8 tiles";
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* w = &fops[8 * i];
    int32_t* m = &mem[(size_t)i * MW];
    const int kind = w[0] & 0xFF;
    for (int q = 0; q < 8; ++q) m[8 + q] = -1;
    m[2] = m[3] = -1;
    if (kind == FOP_PAIR_TM || kind == FOP_PAIR_MT) {
      m[0] = 1 | (kind == FOP_PAIR_MT ? 2 : 0) | (1 << 8);
      pair_kind[i] = 1;
      m[1] = w[1]; m[2] = out.live_of_slot[w[3]];
      m[8] = w[2] < 0 ? -1 : out.live_of_slot[w[2]];
    } else {
      if (w[2] > 8 || w[2] < 1) return;
      out.max_sources = std::max(out.max_sources, (int)w[2]);
      for (int q = 0; q < w[2]; ++q) m[8 + q] = out.live_of_slot[fp.psrcs[w[1] + q]];
      m[0] = (w[2] << 8) | (kind == FOP_VAR ? 16 : (1 | (kind == FOP_VAR_PAIR_MT ? 2 : 0)));
      m[3] = out.live_of_slot[w[3]]; m[4] = w[3];
      if (m[3] >= 0) m[0] |= 4;
      else last_var_write[w[3]] = i;
      if (kind != FOP_VAR) { m[1] = w[4]; m[2] = out.live_of_slot[w[5]]; }
    }
  }
  for (int c = 0; c < n_msgs; ++c)
    if (last_var_write[c] >= 0) mem[(size_t)last_var_write[c] * MW] |= 8;        // write this v->f message out here
  // tiles read before anything in the program has written them start as the uniform vector (FactorGraph.initialize)
  std::vector<int32_t> init_tiles;
  {
    std::vector<char> have(out.n_live, 0);
    for (int k = 0; k < fp.n_cprod; ++k) have[out.live_of_slot[n_msgs + 1 + k]] = 1;        // written by the prologue
    for (int i = 0; i < n_ops; ++i) {
      const int32_t* m = &mem[(size_t)i * MW];
      const int n = (m[0] >> 8) & 15;
      for (int q = 0; q < n; ++q)
        if (m[8 + q] >= 0 && !have[m[8 + q]]) { have[m[8 + q]] = 1; init_tiles.push_back(m[8 + q]); }
      if (m[2] >= 0) have[m[2]] = 1;
      if (m[3] >= 0) have[m[3]] = 1;
    }
  }
  std::vector<int32_t> bundles;
  const int32_t nop[MW] = {0, 0, -1, -1, 0, 0, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1};
  // device form, 4 words: [0] flags | nsrc << 8 | pair slot << 16   [1] destination tile | product tile << 8 (0xFF = none) |
  // message slot of the product << 16   [2] source tiles 0-3, [3] 4-7, one byte each (0xFF = none)
  auto pack_member = [&](const int32_t* m) {
    int32_t w[4] = {0, 0, 0, 0};
    w[0] = m[0] | (m[1] << 16);
    w[1] = (m[2] & 0xFF) | ((m[3] & 0xFF) << 8) | (m[4] << 16);
    for (int q = 0; q < 8; ++q) w[2 + (q >> 2)] |= (m[8 + q] & 0xFF) << (8 * (q & 3));
    bundles.insert(bundles.end(), w, w + 4);
  };
  auto disjoint = [&](const int32_t* x, const int32_t* y) {
    // y reads nothing x writes, and writes nothing x reads or writes
    auto writes = [](const int32_t* m, int tile) { return tile >= 0 && (m[2] == tile || m[3] == tile); };
    for (int q = 0; q < 8; ++q) if (writes(x, y[8 + q]) || writes(y, x[8 + q])) return false;
    return !(writes(x, y[2]) || writes(x, y[3]));
  };
  std::vector<int> pairing;                                      // per bundle: its members' indices (the second -1: alone)
  for (int i = 0; i < n_ops;) {
    const int32_t* x = &mem[(size_t)i * MW];
    pack_member(x);
    if (i + 1 < n_ops && disjoint(x, &mem[(size_t)(i + 1) * MW])) {
      pack_member(&mem[(size_t)(i + 1) * MW]);
      pairing.push_back(i); pairing.push_back(i + 1);
      i += 2;
    } else {
      pack_member(nop);
      pairing.push_back(i); pairing.push_back(-1);
      i += 1;
    }
  }
  out.n_bundles = (int)bundles.size() / 8;
  if (getenv("MLBP_DEBUG_SHARED_PROGRAM")) {                     // diagnostic: the member records, one line each
    for (int i = 0; i < n_ops; ++i) {
      const int32_t* m = &mem[(size_t)i * MW];
      fprintf(stderr, "member %2d: flags 0x%03x pair %d dst %d ptile %d pslot %d src", i, m[0], m[1], m[2], m[3], m[4]);
      for (int q = 0; q < 8; ++q) if (m[8 + q] >= 0 || q == 0) fprintf(stderr, " %d", m[8 + q]);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "n_live %d n_bundles %d n_cprod %d\n", out.n_live, out.n_bundles, fp.n_cprod);
  }
  pack_member(nop); pack_member(nop);                            // the kernel prefetches one bundle past the end
  // constant products, flattened: {unary factor, message slot, tile, 1 = first | 2 = last of its product}
  std::vector<int32_t> ent;
  out.why = "unsupported update kind or slot use";
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
  out.n_ops = n_ops; out.n_cpw = (int)ent.size();
  out.n_back = (int)back.size() / 2; out.n_fill = (int)fill.size(); out.n_init = (int)init_tiles.size();
  // device image: bundles [n_bundles + 1][2][4] | cprod entries | write-back pairs | fill slots | uniform tiles | product tiles | written bits
  out.image = bundles;
  out.off_ent = (int)out.image.size();
  out.image.insert(out.image.end(), ent.begin(), ent.end());
  out.off_back = (int)out.image.size();
  out.image.insert(out.image.end(), back.begin(), back.end());
  out.off_fill = (int)out.image.size();
  out.image.insert(out.image.end(), fill.begin(), fill.end());
  out.off_init = (int)out.image.size();
  out.image.insert(out.image.end(), init_tiles.begin(), init_tiles.end());
  out.off_ptile = (int)out.image.size();                         // tile of constant product k
  for (int k = 0; k < fp.n_cprod; ++k) out.image.push_back(out.live_of_slot[n_msgs + 1 + k]);
  out.off_written = (int)out.image.size();                       // bit c: some update of the program writes slot c
  for (int c0 = 0; c0 < n_msgs; c0 += 32) {
    uint32_t w = 0;
    for (int c = c0; c < n_msgs && c < c0 + 32; ++c) w |= (written[c] ? 1u : 0u) << (c - c0);
    out.image.push_back((int32_t)w);
  }
  for (int q = 0; q < 16; ++q) out.image.push_back(0);
  // ---- product-fused form.  When every variable update of the program multiplies at most one constant product and one
  // factor->variable message (variables with at most two pairwise factors: K2, K3, chains, rings), the message F->X is only
  // ever read as the product  c_X (.) m_{F->X}  (LBP.py:377-389).  The PRODUCER then stores that product -- its 16 rows of the
  // D fragment times its 16 rows of c_X: four multiplications -- and every contraction reads ONE tile straight into the matrix
  // cores: no products, half the tile reads, and (float64 vector operations share the matrix cores' pipe) some fifty
  // operations per update off the dependent chain.  The raw result of the LAST update of each slot goes to a scratch tile
  // in memory for the read-out.  Member record, 4 words:
  //   [0] flags | pair slot << 16: 1 contraction, 2 m^T.T, 8 the input S, normalised, is this variable->factor slot's last value:
  //       to memory, 16 no contraction, 32 last update of the destination slot: raw result to stash [2] >> 8 (when the call writes
  //       the messages back), 64 store the product
  //   [1] destination tile | S tile << 8 (0xFF: the uniform vector) | message slot of S << 16   [2] constant-product tile the
  //       result is multiplied by (0xFF none) | stash index << 8
  {
    std::vector<char> is_c(out.n_live, 0);
    std::vector<int> prod_of_tile(out.n_live, -1);
    for (int k = 0; k < fp.n_cprod; ++k) { is_c[out.live_of_slot[n_msgs + 1 + k]] = 1; prod_of_tile[out.live_of_slot[n_msgs + 1 + k]] = k; }
    bool okp = out.max_sources <= 2;
    std::vector<int> cp(out.n_live, -2);                          // constant product a message tile is read with: -2 never read, -1 none
    std::vector<int> s_of(n_ops, 0xFF), last_writer(out.n_live, -1);
    for (int i = 0; i < n_ops && okp; ++i) {
      const int32_t* m = &mem[(size_t)i * MW];
      const int n = (m[0] >> 8) & 15;
      if (m[0] & 4) okp = false;                                  // a variable->factor message kept as a tile of its own
      int c = -1, mt = -1, nc = 0, nm = 0;
      for (int q = 0; q < n; ++q) {
        const int tl = m[8 + q];
        if (tl < 0) continue;
        if (is_c[tl]) { c = tl; ++nc; } else { mt = tl; ++nm; }
      }
      if (nc > 1 || nm > 1) okp = false;
      if (nm) {
        if (cp[mt] == -2) cp[mt] = c; else if (cp[mt] != c) okp = false;
        s_of[i] = mt;
      } else if (nc) {
        s_of[i] = c;
      }
      if ((m[0] & 1) && m[2] >= 0) { if (is_c[m[2]]) okp = false; last_writer[m[2]] = i; }
    }
    // every stored product needs its constant product (a variable without unary factors: the general form), and a member whose
    // input is still the uniform vector reads a tile nothing touches before a later bundle writes it, filled by the prologue
    for (int tl = 0; tl < out.n_live && okp; ++tl)
      if (!is_c[tl] && cp[tl] == -1) okp = false;
    std::vector<int32_t> uinit;                                   // tiles the prologue fills with the uniform vector
    if (okp) {
      std::vector<int> bundle_of(n_ops, 0), first_touch(out.n_live, n_ops + 1);
      for (size_t b = 0; b < pairing.size(); b += 2) { bundle_of[pairing[b]] = (int)b / 2; if (pairing[b + 1] >= 0) bundle_of[pairing[b + 1]] = (int)b / 2; }
      for (int i = n_ops - 1; i >= 0; --i) {
        const int32_t* m = &mem[(size_t)i * MW];
        if ((m[0] & 1) && m[2] >= 0) first_touch[m[2]] = bundle_of[i];
        if (s_of[i] != 0xFF) first_touch[s_of[i]] = bundle_of[i];
      }
      for (int q : init_tiles) first_touch[q] = -1;              // (holds c (.) uniform from the start)
      for (int i = 0; i < n_ops && okp; ++i) {
        if (s_of[i] != 0xFF) continue;
        int pick = -1;
        for (int tl = 0; tl < out.n_live && pick < 0; ++tl)
          if (!is_c[tl] && first_touch[tl] > bundle_of[i]) pick = tl;
        if (pick < 0) { okp = false; break; }
        s_of[i] = pick;
        if (std::find(uinit.begin(), uinit.end(), pick) == uinit.end()) uinit.push_back(pick);
      }
    }
    out.pf_ok = okp;
    if (okp) {
      std::vector<int32_t> stash(out.n_live, -1), pfb, pinit;
      for (int tl : uinit) { pinit.push_back(tl); pinit.push_back(-1); }
      for (int tl = 0; tl < out.n_live; ++tl) if (last_writer[tl] >= 0) stash[tl] = out.n_stash++;
      auto pack_pf = [&](int i) {
        int32_t w[4] = {0, 0xFFFF, 0xFF, 0};
        if (i >= 0) {
          const int32_t* m = &mem[(size_t)i * MW];
          const int dst = (m[0] & 1) ? m[2] : -1;
          w[0] = (m[0] & (1 | 2 | 8 | 16)) | (m[1] << 16);
          if (dst >= 0 && last_writer[dst] == i) w[0] |= 32;
          if (dst >= 0) w[0] |= 64;                              // (a slot nothing reads as an input is kept for the read-out: the raw result, c absent)
          w[1] = (dst & 0xFF) | ((s_of[i] & 0xFF) << 8) | (m[4] << 16);
          w[2] = ((dst >= 0 && cp[dst] >= 0 ? cp[dst] : 0xFF) & 0xFF) | ((dst >= 0 ? stash[dst] : 0) << 8);
        }
        pfb.insert(pfb.end(), w, w + 4);
      };
      for (size_t b = 0; b < pairing.size(); b += 2) { pack_pf(pairing[b]); pack_pf(pairing[b + 1]); }
      pack_pf(-1); pack_pf(-1);
      // message tiles read before the program writes them: c (.) uniform, i.e. a copy of the constant product (or uniform)
      for (int q = 0; q < (int)init_tiles.size(); ++q) {
        const int tl = init_tiles[q];
        if (is_c[tl]) continue;
        pinit.push_back(tl); pinit.push_back(cp[tl] >= 0 ? prod_of_tile[cp[tl]] : -1);
      }
      out.n_pinit = (int)pinit.size() / 2;
      // The gradient epilogue takes a factor's two variable->factor messages straight from LDS when every such slot's last value
      // (the input S of its flag-8 member) is a message tile no later member rewrites: vftile[slot] = that tile, -1 = never updated
      // (uniform), and the form is off (vf_direct false) when some slot has no such tile or its S is a constant-product tile (the
      // read-out stages the marginals there).
      std::vector<int32_t> vftile(n_msgs, -1);
      out.vf_direct = true;
      for (int i = 0; i < n_ops; ++i) {
        const int32_t* m = &mem[(size_t)i * MW];
        if (!(m[0] & 8)) continue;
        const int tl = s_of[i];
        bool intact = tl != 0xFF && !is_c[tl];
        for (int j = i; j < n_ops && intact; ++j) {               // (member i itself included: its own result must land elsewhere)
          const int32_t* mj = &mem[(size_t)j * MW];
          if ((mj[0] & 1) && mj[2] == tl) intact = false;
        }
        if (!intact) { out.vf_direct = false; break; }
        vftile[m[4]] = tl;
      }
      for (int c = 0; c < n_msgs && out.vf_direct; ++c)
        if (out.hoisted[c] < 0 && is_vf[c] && vftile[c] < 0) out.vf_direct = false;      // (a slot some variable update writes but no flag-8 member hands out)
      out.off_pfb = (int)out.image.size();
      out.image.insert(out.image.end(), pfb.begin(), pfb.end());
      out.off_stash = (int)out.image.size();
      out.image.insert(out.image.end(), stash.begin(), stash.end());
      for (int tl = 0; tl < out.n_live; ++tl) out.image.push_back(cp[tl] >= 0 ? 1 : 0);       // [n_live] behind it: the tile holds c (.) message (else the message)
      out.off_pinit = (int)out.image.size();
      out.image.insert(out.image.end(), pinit.begin(), pinit.end());
      out.off_vftile = (int)out.image.size();
      out.image.insert(out.image.end(), vftile.begin(), vftile.end());
      for (int q = 0; q < 16; ++q) out.image.push_back(0);
      if (getenv("MLBP_DEBUG_SHARED_PROGRAM")) {
        for (size_t i = 0; i + 3 < pfb.size(); i += 4) fprintf(stderr, "pf member: flags 0x%02x pair %d dst %d S %d slot %d c %d stash %d\n", pfb[i] & 0xFF,
                                                                pfb[i] >> 16, pfb[i + 1] & 0xFF, (pfb[i + 1] >> 8) & 0xFF, pfb[i + 1] >> 16, pfb[i + 2] & 0xFF, pfb[i + 2] >> 8);
        fprintf(stderr, "pf: n_stash %d n_pinit %d\n", out.n_stash, out.n_pinit);
      }
    }
  }
  // ---- product-fused form, variables with THREE pairwise factors (K4 cliques: every variable update multiplies the constant
  // product c and two factor->variable messages).  The producer of a message into such a variable stores  sqrt(c) (.) m  (the
  // prepare kernel writes sqrt(c) for these products): the input of a contraction is then the PRODUCT OF TWO TILES,
  // sqrt(c) m_a (.) sqrt(c) m_b = c (.) m_a (.) m_b -- two tile reads and one multiplication per element instead of three reads
  // and two, and 12 message tiles + the stored variable->factor messages instead of 21 tiles: everything stays in LDS (one
  // workgroup per CU; the constant products themselves stay in memory: a producer asks for its sixteen rows of them in front of
  // its matrix instructions).  A variable with two pairwise factors in the same program keeps the c (.) m form above (the
  // exponent is per constant product: sqrt_mask).  Stored variable->factor messages (members with flag 4, read by a later
  // plain factor update) are raw tiles.  Record, 4 words:
  //   [0] flags | pair slot << 16: as above, and 4 the input product is kept as tile [3] >> 8, 0x100 a second input tile [3] & 0xFF
  //   [1] destination | input tile << 8 | message slot of the input product << 16      (tiles: LDS indices, the constant products left out)
  //   [2] constant PRODUCT INDEX the result is multiplied by (0xFF none) | stash index << 8      [3] second input | kept tile << 8
  // Sections: map3 [n_live] LDS index of a tile (0x100 | product index for a constant product), kind3 [n_lds] 0 raw / 1 c (.) m /
  // 2 sqrt(c) (.) m, | 0x100 some update writes it; stash [n_lds]; back3 [n_back][2] the write-back list in LDS indices.
  if (!out.pf_ok && out.max_sources == 3 && !getenv("MLBP_SHARED_NO_P3")) {
    const int NL = out.n_live;
    std::vector<char> is_c(NL, 0);
    std::vector<int> prod_of_tile(NL, -1);
    for (int k = 0; k < fp.n_cprod; ++k) { is_c[out.live_of_slot[n_msgs + 1 + k]] = 1; prod_of_tile[out.live_of_slot[n_msgs + 1 + k]] = k; }
    bool ok3 = true;
    std::vector<int> kind(NL, -2), cp(NL, -1), last_writer(NL, -1), ckind(NL, 0);
    std::vector<int> s1(n_ops, 0xFF), s2(n_ops, 0xFF);
    for (int i = 0; i < n_ops && ok3; ++i) {
      const int32_t* m = &mem[(size_t)i * MW];
      const int n = (m[0] >> 8) & 15;
      if (pair_kind[i]) {
        const int tl = m[8];
        if (tl >= 0) {
          if (is_c[tl] || (kind[tl] != -2 && kind[tl] != 0)) ok3 = false;
          else { kind[tl] = 0; s1[i] = tl; }
        }
      } else {
        int c = -1, nc = 0, nm = 0, mt[2] = {-1, -1};
        for (int q = 0; q < n; ++q) {
          const int tl = m[8 + q];
          if (tl < 0) { ok3 = false; break; }
          if (is_c[tl]) { c = tl; ++nc; }
          else if (nm < 2) mt[nm++] = tl;
          else ok3 = false;
        }
        if (nc != 1 || nm < 1) ok3 = false;
        for (int q = 0; q < nm && ok3; ++q) {
          if (kind[mt[q]] == -2) { kind[mt[q]] = nm; cp[mt[q]] = c; }
          else if (kind[mt[q]] != nm || cp[mt[q]] != c) ok3 = false;
        }
        if (ok3) {
          if (ckind[c] == 0) ckind[c] = nm; else if (ckind[c] != nm) ok3 = false;
          s1[i] = mt[0]; s2[i] = nm == 2 ? mt[1] : 0xFF;
        }
        if (ok3 && (m[0] & 4)) {
          const int K = m[3];
          if (K < 0 || is_c[K] || (kind[K] != -2 && kind[K] != 0)) ok3 = false; else kind[K] = 0;
        }
      }
      if (ok3 && (m[0] & 1) && m[2] >= 0) { if (is_c[m[2]]) ok3 = false; last_writer[m[2]] = i; }
    }
    std::vector<int32_t> uinit;
    if (ok3) {
      std::vector<int> bundle_of(n_ops, 0), first_touch(NL, n_ops + 1);
      for (size_t b = 0; b < pairing.size(); b += 2) { bundle_of[pairing[b]] = (int)b / 2; if (pairing[b + 1] >= 0) bundle_of[pairing[b + 1]] = (int)b / 2; }
      for (int i = n_ops - 1; i >= 0; --i) {
        const int32_t* m = &mem[(size_t)i * MW];
        if ((m[0] & 1) && m[2] >= 0) first_touch[m[2]] = bundle_of[i];
        if (m[0] & 4) first_touch[m[3]] = bundle_of[i];
        if (s1[i] != 0xFF) first_touch[s1[i]] = bundle_of[i];
        if (s2[i] != 0xFF) first_touch[s2[i]] = bundle_of[i];
      }
      for (int q : init_tiles) first_touch[q] = -1;
      for (int i = 0; i < n_ops && ok3; ++i) {
        if (s1[i] != 0xFF) continue;                              // (only a plain factor update whose input nothing has written yet)
        int pick = -1;
        for (int tl = 0; tl < NL && pick < 0; ++tl)
          if (!is_c[tl] && first_touch[tl] > bundle_of[i]) pick = tl;
        if (pick < 0) { ok3 = false; break; }
        s1[i] = pick;
        if (std::find(uinit.begin(), uinit.end(), pick) == uinit.end()) uinit.push_back(pick);
      }
    }
    out.p3_ok = ok3;
    if (ok3) {
      std::vector<int> lds_of(NL, -1);
      out.n_lds = 0;
      for (int tl = 0; tl < NL; ++tl) if (!is_c[tl]) lds_of[tl] = out.n_lds++;
      out.sqrt_mask = 0;
      for (int tl = 0; tl < NL; ++tl) if (is_c[tl] && ckind[tl] == 2) out.sqrt_mask |= 1 << prod_of_tile[tl];
      std::vector<int32_t> stash(out.n_lds, -1), pfb, pinit, map3(NL, 0), kind3(out.n_lds, 0), back3;
      out.n_stash = 0;
      for (int tl = 0; tl < NL; ++tl) {
        map3[tl] = is_c[tl] ? (0x100 | prod_of_tile[tl]) : lds_of[tl];
        if (is_c[tl]) continue;
        if (last_writer[tl] >= 0) stash[lds_of[tl]] = out.n_stash++;
        kind3[lds_of[tl]] = std::max(kind[tl], 0) | (last_writer[tl] >= 0 ? 0x100 : 0);
      }
      for (int tl : uinit) { pinit.push_back(lds_of[tl]); pinit.push_back(-1); }
      for (int q = 0; q < (int)init_tiles.size(); ++q) {
        const int tl = init_tiles[q];
        if (is_c[tl]) continue;
        pinit.push_back(lds_of[tl]); pinit.push_back(cp[tl] >= 0 ? prod_of_tile[cp[tl]] : -1);
      }
      auto pack3 = [&](int i) {
        int32_t w[4] = {0, 0xFFFF, 0xFF, 0xFFFF};
        if (i >= 0) {
          const int32_t* m = &mem[(size_t)i * MW];
          const int dst = (m[0] & 1) ? m[2] : -1;
          w[0] = (m[0] & (1 | 2 | 4 | 8 | 16)) | (m[1] << 16);
          if (dst >= 0 && last_writer[dst] == i) w[0] |= 32;
          if (dst >= 0) w[0] |= 64;
          if (s2[i] != 0xFF) w[0] |= 0x100;
          w[1] = ((dst >= 0 ? lds_of[dst] : 0xFF) & 0xFF) | ((lds_of[s1[i]] & 0xFF) << 8) | (m[4] << 16);
          w[2] = ((dst >= 0 && cp[dst] >= 0 ? prod_of_tile[cp[dst]] : 0xFF) & 0xFF) | ((dst >= 0 ? stash[lds_of[dst]] : 0) << 8);
          w[3] = ((s2[i] != 0xFF ? lds_of[s2[i]] : 0xFF) & 0xFF) | ((((m[0] & 4) ? lds_of[m[3]] : 0xFF) & 0xFF) << 8);
        }
        pfb.insert(pfb.end(), w, w + 4);
      };
      for (size_t b = 0; b < pairing.size(); b += 2) { pack3(pairing[b]); pack3(pairing[b + 1]); }
      pack3(-1); pack3(-1);
      for (size_t q = 0; q + 1 < back.size(); q += 2) { back3.push_back(lds_of[back[q]]); back3.push_back(back[q + 1]); }
      out.n_pinit = (int)pinit.size() / 2;
      out.vf_direct = false;
      out.off_pfb = (int)out.image.size();
      out.image.insert(out.image.end(), pfb.begin(), pfb.end());
      out.off_stash = (int)out.image.size();
      out.image.insert(out.image.end(), stash.begin(), stash.end());
      out.off_pinit = (int)out.image.size();
      out.image.insert(out.image.end(), pinit.begin(), pinit.end());
      out.off_map3 = (int)out.image.size();
      out.image.insert(out.image.end(), map3.begin(), map3.end());
      out.off_kind3 = (int)out.image.size();
      out.image.insert(out.image.end(), kind3.begin(), kind3.end());
      out.off_back3 = (int)out.image.size();
      out.image.insert(out.image.end(), back3.begin(), back3.end());
      for (int q = 0; q < 16; ++q) out.image.push_back(0);
      if (getenv("MLBP_DEBUG_SHARED_PROGRAM")) {
        for (size_t i = 0; i + 3 < pfb.size(); i += 4)
          fprintf(stderr, "p3 member: flags 0x%03x pair %d dst %d S %d S2 %d keep %d slot %d cprod %d stash %d\n", pfb[i] & 0xFFF, pfb[i] >> 16, pfb[i + 1] & 0xFF,
                  (pfb[i + 1] >> 8) & 0xFF, pfb[i + 3] & 0xFF, (pfb[i + 3] >> 8) & 0xFF, pfb[i + 1] >> 16, pfb[i + 2] & 0xFF, pfb[i + 2] >> 8);
        fprintf(stderr, "p3: n_lds %d n_stash %d n_pinit %d sqrt_mask 0x%x\n", out.n_lds, out.n_stash, out.n_pinit, out.sqrt_mask);
      }
    }
  }
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

constexpr int WG = 256;                // the gradient kernels and the small helpers
constexpr int SWG = 512;               // the sweep kernel: eight waves
constexpr int G = 16;                 // graphs per workgroup = the N of v_mfma_f64_16x16x4_f64
constexpr int TILE = 64 * G;          // doubles per message tile
constexpr int MW = 4;                 // words per packed member record (build_shared_program); a bundle = 8 words

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
typedef const int32_t __attribute__((address_space(4))) * const_i32p;
__device__ __forceinline__ const_i32p as_const(const int32_t* p) { return (const_i32p)(uintptr_t)p; }

// The gradient as the sweep kernel's epilogue (FactorGraph.get_unregularized_gradeint, LBP.py:301-320, beliefs fused in): what
// gradient_shared_pairs_kernel reads, minus the message buffer -- the final messages are the workgroup's own.
// Group tables (mlbp_sweep_groups_f64): the group of block `block` -- starts ascending, starts[n] = the grid size -- and its
// description, both through the scalar data cache: the description then lives in SGPRs exactly like a kernel argument
// (a plain struct copy from global memory lands in VGPRs and from there in scratch).
__device__ __forceinline__ int find_group(const int32_t* starts, int n, int block) {
  const const_i32p s = as_const(starts);
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s[mid] <= block) lo = mid; else hi = mid - 1;
  }
  return __builtin_amdgcn_readfirstlane(lo);
}
template <typename T>
__device__ __forceinline__ void load_uniform(T& dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "copied as 32-bit words");
  const const_i32p w = as_const(reinterpret_cast<const int32_t*>(src));
  int32_t* o = reinterpret_cast<int32_t*>(&dst);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) o[i] = w[i];
}

struct SharedGradDev {
  const int32_t* c_slot; const int32_t* r_slot; const int32_t* pair_phi; const int32_t* pair_label;
  const double* phi[2];         // interleaved [64][64][3]: the label term
  const double* wfrag;          // [table][which][4][4096] A fragments of T (.) phi_k and T (shared_prepare_kernel)
  const int32_t* plane_flags;   // [which][4]: feature plane k of tensor `which` is 0 general | 1 all zeros | 2 all ones (same launch)
  double* grad_en_en;           // [B][3]: the unary factors' terms are there (shared_prepare_kernel), the pairwise ones are added
  int32_t enabled, pad_;
};

struct SharedDev {
  const double* pair_tables;
  const int32_t* pair_tab;
  const double* unary_tables;
  const int32_t* unary_tab;
  double* msgs;                 // [B][n_msgs][64] or NULL (no write-back)
  double* marginals;            // [B][n_vars][64] or NULL
  int32_t* status;
  uint8_t* bail;
  const int32_t* image;         // SharedProgram::image (global memory; the bundles are read through the scalar cache)
  const int32_t* readout;
  int32_t B, n_msgs, P, U, n_pair_tables, n_unary_tables, n_vars;
  int32_t n_bundles, n_live, n_cprod, n_back, n_fill, n_init, off_back, off_fill, off_init, off_ptile;
  const double* ptiles;         // [groups][n_cprod][1024] constant products as tiles (shared_prepare_kernel)
  int32_t vf_only;              // write back only the variable->factor messages (what the gradient reads)
  int32_t n_res;                // tiles [0, n_res) live in LDS, the rest in `spill`
  double* spill;                // [workgroups][n_live - n_res][64][16] or NULL
  const double* tfrag;          // [n_pair_tables][2][4096] A fragments of every table (only when there are <= FRAG_TABLES), or NULL
  int32_t off_written, pad_;
  // product-fused form (SharedProgram::pf_ok): its bundle records, the stash index of every tile, the tiles that start as a copy of
  // a constant product; stash [workgroups][n_stash][64][16]: the raw result of the last update of every slot
  int32_t off_pfb, off_stash, off_pinit, n_pinit, n_stash, pad2_;
  double* stash;
  const int32_t* header;        // [groups][HDR] shared_prepare_kernel's verdict on each group of 16 graphs and its fragment sets
  int32_t off_vftile, vf_direct;  // product-fused + gradient: the epilogue reads the final variable->factor messages from the message tiles
  int32_t lds_bytes, pad3_;     // the workgroup's dynamic LDS
  // three-source product-fused form (SharedProgram::p3_ok): tile maps, the write-back list in LDS indices, which constant products
  // the prepare kernel stored as square roots
  int32_t off_map3, off_kind3, off_back3, sqrt_mask;
  SharedGradDev gr;
};

#define MLBP_MFMA16(A)                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 16; s_ += 2) {                            \
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s_], b[s_], acc0, 0, 0, 0);       \
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s_ + 1], b[s_ + 1], acc1, 0, 0, 0); \
  }

#ifdef MLBP_STAMPS
__device__ unsigned long long* g_sh_stamp = nullptr;
__device__ int g_sh_ablate = 0;      // timing experiments of tools/stamp_shared.py (results become wrong): 1 no MFMAs, 2 no tile reads, 4 no result stores,
                                     // 8 / 16 prepare kernel without its row loads / copy-out, 512 no marginal read-out, 1024 no fragment loads, 2048 no main loop
#define ABL(bit) (abl_ & (bit))
#define ABL_DECL const int abl_ = __builtin_amdgcn_readfirstlane(g_sh_ablate);
#ifdef MLBP_STAMPS_LIGHT      // ablations only: no clock reads in the kernel
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#define STAMP_FLUSH_GRAD
#else
#define STAMP_DECL unsigned long long _t0 = 0, _ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t0) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long _t1; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t1) :: "memory"); __builtin_amdgcn_sched_barrier(0); _ph[i] += _t1 - _t0; _t0 = _t1; }
#define STAMP_FLUSH if (g_sh_stamp && blockIdx.x < 64 && (threadIdx.x & 63) == 0) { for (int _i = 0; _i < 8; ++_i) g_sh_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12 + _i] = _ph[_i]; }
#define STAMP_FLUSH_GRAD if (g_sh_stamp && blockIdx.x < 64 && (threadIdx.x & 63) == 0) { for (int _i = 8; _i < 12; ++_i) g_sh_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12 + _i] = _ph[_i]; }
#endif
#else
#define ABL(bit) 0
#define ABL_DECL
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#define STAMP_FLUSH_GRAD
#endif

// Message tiles: 64 states x 16 graphs as [k-step pair sp][lane][2] doubles -- lane l = (state & 3) * 16 + graph, k-step
// s = state >> 2 -- so that one 16-byte LDS read per lane is two k-steps of the MFMA B operand (the D fragment of row block
// rb, registers r = 0..3 = k-steps 4 rb + r, goes back as two 16-byte writes).  8-byte accesses in ds_read2 form run at
// half the LDS rate (MI355X_MICROARCH.md, LDS table) and were 60 % of the CU's LDS cycles.
__device__ __forceinline__ int tile_index(int state, int graph) {
  const int s = state >> 2, cq = state & 3;
  return (((s >> 1) * 64 + cq * 16 + graph) << 1) | (s & 1);
}

// Which half of the workgroup holds (table rank, orientation) pair `sel` = 2 * rank + mt, under partition `mode`:
//   mode 0: {T0.m, m.T1 | m.T0, T1.m}   mode 1: {T0.m, m.T0 | T1.m, m.T1}   mode 2: {T0.m, T1.m | m.T0, m.T1}
// and its position (0 / 1) among the two that half holds.
__device__ __forceinline__ int half_of(int mode, int sel) { return ((mode == 0 ? 0x6 : (mode == 1 ? 0xC : 0xA)) >> sel) & 1; }
__device__ __forceinline__ int index_in_half(int mode, int sel) { return ((mode == 0 ? 0xC : (mode == 1 ? 0xA : 0xC)) >> sel) & 1; }

// Everything of a shared-table sweep that does not depend on the sweeps, as ONE streaming launch in front of it:
//   one wave per graph: the constant products of the graph -- per variable the product of its unary factors' table columns
//     (LBP.py:494-498, 381-386), normalised -- written as its column of its group's message tiles in `ptiles`
//     [group][product][1024], and the per-graph verdict in bail[] (0 = clean, 1 = the exact kernel must redo the graph),
//     which also clears the flags;
//   and, spread over the first blocks, fragment-ordered copies of every pairwise table: out[table][0 | 1][row block w]
//     [k-step s][lane] = element (16 w + (lane & 15), 4 s + (lane >> 4)) of T resp. T^T (the MFMA A operand).
// The dependent chain (row index -> rows -> product) that cost the sweep kernel a fifth of its single round runs here at
// full occupancy instead: 8192 independent waves, two batches of rows in flight each.
constexpr int FRAG_TABLES = 32;                                  // shared-table batches have a handful of tables
constexpr int PWG = 1024, PGB = 16;                              // threads / graphs per block of the prepare kernel: one group of 16 graphs
struct PrepareDev {
  const double* pair_tables; double* tfrag; int32_t n_frag_tables;
  int32_t n_wfrag_tables;                                        // > 0: also the gradient's weighted fragments, T (.) phi_k and T
  double* wfrag; const double* phi_p0; const double* phi_p1;     // wfrag [table][which][4][4096]; phi planar [3][64][64]
  int32_t* plane_flags;                                          // [2][4] (behind wfrag): constant feature planes, see SharedGradDev
  // ... and the unary factors' gradient terms, phi[label][obs][:] - E[table row][:] (E from mlbp_unary_expectations_f64), summed
  // per graph into grad_en_en [B][3] / grad_en_de [B][6] (assigned: the sweep kernel's epilogue adds the pairwise terms)
  const double* unary_expect; const int32_t* unary_kind; const int32_t* unary_obs; const int32_t* unary_label;
  const double* phi_i0; const double* phi_i1; const double* phi_ed;        // interleaved [64][64][3] x 2, [64][Vde][6]
  double* grad_en_en; double* grad_en_de; int32_t Vde, grad_on;
  const double* unary_tables; const int32_t* unary_tab; const int32_t* ent;      // ent: [E][4] unary factor, slot, tile, first | last flags
  double* ptiles; uint8_t* bail; int32_t* status;
  int32_t B, U, n_unary_tables, E, n_cprod, n_groups;
  // ... and the group's HEADER for the sweep kernel, header[group][HDR] = {verdict: 0 run | 1 a table index out of range | 2 the 16
  // graphs do not name the same tables (or more than two distinct ones), first table, second table (= the first: one), bit p =
  // pairwise factor p reads the second, partition of the fragment sets over the halves}: what the sweep kernel used to work out
  // from the same data at the start of its single round, one dependent round of memory latency in front of its fragment loads
  const int32_t* pair_tab; const int32_t* image; int32_t* header;
  int32_t P, n_pair_tables, n_bundles, sqrt_mask;                 // sqrt_mask bit k: constant product k is stored as its square root (three-source product-fused form)
};
constexpr int HDR = 8;
// MULTI: several groups of graphs (mlbp_sweep_groups_f64: every group its own program, tables, messages) in one launch;
// gtab[k] = the group's PrepareDev, gstart[k] = its first block (ascending; gstart[n_groups] = the grid size).
// (Measured and not kept, round 4: a STAGED form -- one 1024-thread workgroup per CU copies the unary tables, 96 KB for the trainer's
// pots, into LDS once and takes every n-th group of 16 graphs, rows out of LDS, next group's indices prefetched, two sets of
// tiles -- 14.6 us against 13.0: sixteen waves per CU walk their chains of round trips one group after the other, thirty-two
// overlap them.)
template <bool MULTI>
__global__ __launch_bounds__(PWG, 8) void shared_prepare_kernel(PrepareDev d, const PrepareDev* gtab, const int32_t* gstart, int n_groups) {
  extern __shared__ double ptile_lds[];                          // [n_cprod][1024] the group's product tiles, assembled here and stored as whole lines
  ABL_DECL
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int block = blockIdx.x, n_blocks = gridDim.x;
  const PrepareDev launch_job = d;
  if (MULTI) {
    const int lo = find_group(gstart, n_groups, block);
    load_uniform(d, gtab + lo);
    block -= as_const(gstart)[lo];
    n_blocks = as_const(gstart)[lo + 1] - as_const(gstart)[lo];
  }
  // (MULTI: the launch's shared job first -- the by-value description, spread over ALL blocks of the launch: the groups of a
  // minibatch read the same pots, and a table's copies are written once per launch, not once per group -- then the group's own)
  auto fragment_jobs = [&](const PrepareDev& d, const int block, const int n_blocks) {
    for (int q = block; q < (ABL(16384) ? 0 : 2 * d.n_frag_tables); q += n_blocks) {
      const int ti = q >> 1, mt = q & 1;
      const double* T = d.pair_tables + (size_t)ti * 4096;
      double* o = d.tfrag + ((size_t)ti * 2 + mt) * 4096;
      for (int e = t; e < 4096; e += PWG) {
        const int l = e & 63, sk = (e >> 6) & 15, w = e >> 10;
        const int i = 16 * w + (l & 15), k = 4 * sk + (l >> 4);
        o[e] = mt ? T[k * 64 + i] : T[i * 64 + k];
      }
    }
    // the gradient epilogue's A operands (what pair_weight_fragments_kernel writes): quarter jobs (one row block each) on the
    // blocks behind those, which start as early and so finish inside the launch's main body
    for (int q = block - 2 * d.n_frag_tables; q < 32 * d.n_wfrag_tables; q += n_blocks) {
      if (q < 0) continue;
      const int ti = q >> 5, which = (q >> 4) & 1, k = (q >> 2) & 3, w = q & 3;
      const double* T = d.pair_tables + (size_t)ti * 4096;
      const double* ph = (which ? d.phi_p1 : d.phi_p0) + (size_t)(k < 3 ? k : 0) * 4096;
      double* o = d.wfrag + (((size_t)ti * 2 + which) * 4 + k) * 4096 + w * 1024;
      for (int e = t; e < 1024; e += PWG) {
        const int l = e & 63, sk = e >> 6;
        const int idx = (16 * w + (l & 15)) * 64 + 4 * sk + (l >> 4);
        o[e] = k < 3 ? T[idx] * ph[idx] : T[idx];
      }
    }
    // ... and which feature planes are constant: the reference's tensors are [pmi, 0, 1] and [pmi, pmi_w1, 1]
    // (train_mp.py:600-606: a zero plane and the bias), and for those the epilogue needs no contraction -- the expected
    // feature is 0 resp. the belief's total.  One job per plane, on the blocks behind the fragment jobs.
    for (int q = block - 2 * d.n_frag_tables - 32 * d.n_wfrag_tables; q < (d.n_wfrag_tables > 0 ? 6 : 0); q += n_blocks) {
      if (q < 0) continue;
      const int which = q / 3, k = q - 3 * which;
      const double* ph = (which ? d.phi_p1 : d.phi_p0) + (size_t)k * 4096;
      const double c = ph[0];
      bool same = c == 0.0 || c == 1.0;
      for (int e = t; e < 4096 && same; e += PWG) same = ph[e] == c;
      const int all = __syncthreads_and(same ? 1 : 0);
      if (t == 0) d.plane_flags[which * 4 + k] = all ? (c == 0.0 ? 1 : 2) : 0;
    }
  };
  if (MULTI) fragment_jobs(launch_job, (int)blockIdx.x, (int)gridDim.x);
  fragment_jobs(d, block, n_blocks);
  // one workgroup = one group of 16 graphs, one wave per graph: a wave writes its column of the group's tiles into LDS, and the
  // workgroup stores each tile as whole cache lines (a wave's column alone is 16 bytes in each of 32 lines per tile: the
  // partial-line stores of 16 different waves into the same lines were what this launch spent most of its time on)
  __shared__ int hdr_ok[PGB], hdr_same[PGB];
  // (the first 64 entries of the constant-product list: program data -- requested up front)
  const int ent_u0 = d.ent[4 * min(lane, d.E - 1)], ent_flags0 = d.ent[4 * min(lane, d.E - 1) + 3];
  const int g = block * PGB + wave;
  const bool on = g < d.B;
  const int col = wave;
  // the group's header: wave w compares graph w's table indices with the group's first graph's (lane p: pairwise factor p) ...
  // (one register -- lanes 0..P-1: this graph's row, lanes 16..16+P-1: the group's first graph's; P <= 16 -- requested here, looked
  // at behind the products: nothing below waits for it, and the kernel's 64 registers hold it without a spill)
  int pair_rows = 0;
  if (d.header && (lane & 15) < d.P && lane < 32 && !ABL(32768))
    pair_rows = d.pair_tab[(size_t)(lane < 16 ? min(g, d.B - 1) : block * PGB) * d.P + (lane & 15)];
  // lane u holds the table row of the graph's unary factor u (U <= 64: build_shared_program), lane e entry e of the
  // constant-product list; both loads are independent, the entry's row then comes through the lane crossbar
  int my_row = (on && lane < d.U && !ABL(4096)) ? d.unary_tab[(size_t)g * d.U + lane] : 0;
  if (on) {
  constexpr int RB = 8;
  constexpr unsigned KEY_LIMIT = 0x7A11A0FCu;                    // high word of 1e280
  double cur = 1.0;
  unsigned key = 0;
  int k_out = 0;
  bool flagged = false, open_product = false;
  if ((unsigned)my_row >= (unsigned)d.n_unary_tables) { flagged = true; my_row = 0; }    // the exact kernel meets it again and reports it
  // the unary factors' gradient terms (LBP.py:592-619 with the belief's expectation taken once per table row): eight lanes
  // per factor -- lane 8 j + k takes feature k of factor 8 c + j -- so that a factor's row of expected features and its
  // label's feature row are one or two cache lines per FACTOR for the address unit, not per feature (one lane per factor
  // asked for three times as many lines as the table rows below).  Requested here, summed behind the products.
  double u_ee = 0.0, u_ed = 0.0;                                 // lane k (after the sums): feature k of grad_en_en / grad_en_de
  if (d.grad_on && d.unary_expect) {
    int my_obs = 0, my_lab = 0, my_kind = 0;
    if (lane < d.U) { my_obs = d.unary_obs[(size_t)g * d.U + lane]; my_lab = d.unary_label[(size_t)g * d.U + lane]; my_kind = d.unary_kind[lane]; }
    const int k = lane & 7;
    for (int c = 0; c < d.U; c += 8) {
      const int u = c + (lane >> 3);
      const int kind = __shfl(my_kind, u), row = __shfl(my_row, u), obs = __shfl(my_obs, u), lab = __shfl(my_lab, u);
      if (u >= d.U) continue;
      const int cols = kind == 2 ? d.Vde : 64, nf = kind == 2 ? 6 : 3;
      if ((unsigned)obs >= (unsigned)cols || (unsigned)lab >= 64u || (unsigned)kind > 2u) { atomicExch(d.status, 1); continue; }
      if (k >= nf) continue;
      const double* pl = kind == 2 ? d.phi_ed + ((size_t)lab * cols + obs) * 6 : (kind ? d.phi_i1 : d.phi_i0) + ((size_t)lab * 64 + obs) * 3;
      const double v = pl[k] - d.unary_expect[(size_t)row * 8 + k];
      if (kind == 2) u_ed += v; else u_ee += v;
    }
  }
  for (int c0 = 0; c0 < (ABL(8192) ? 0 : d.E); c0 += 64) {        // (more than 64 entries: a chunk at a time)
    const int el = min(c0 + lane, d.E - 1), n_here = min(64, d.E - c0);
    const int ent_u = c0 == 0 ? ent_u0 : d.ent[4 * el], ent_flags = c0 == 0 ? ent_flags0 : d.ent[4 * el + 3];
    const int row = __shfl(my_row, ent_u);
    // The entries of one product are consecutive; the products' ends are the set bits of `ends`.  Batches of up to RB rows of ONE
    // product, two batches in flight (the next one -- of this product or the first of the next -- is requested before this one is
    // multiplied), every test on an entry a SCALAR one: the flag-driven form of this loop was some twenty vector instructions per
    // row and sixteen copies of the product's finish, and the launch was bound by instruction issue (eight waves per SIMD).
    const unsigned long long real_ends = __ballot((ent_flags & 2) != 0 && lane < n_here);
    unsigned long long ends = real_ends | (1ull << (n_here - 1));        // (a product that runs on into the next chunk of 64 entries pauses at the chunk's end)
    int e_next = 0, p_last = -1;                                 // next entry to hand out; last entry of the product being handed out
    auto next_batch = [&](int& b0, int& b1) {                    // false: no entries left
      if (e_next > p_last) {
        if (!ends) return false;
        p_last = __builtin_ctzll(ends);
        ends &= ends - 1;
      }
      b0 = e_next; b1 = min(e_next + RB - 1, p_last); e_next = b1 + 1;
      return true;
    };
    auto fetch = [&](double (&r)[RB], int b0, int b1) {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int rj = __builtin_amdgcn_readlane(row, min(b0 + j, b1));
        r[j] = ABL(8) ? 0.5 : d.unary_tables[(size_t)rj * 64 + lane];
      }
    };
    auto reduce = [&](const double (&r)[RB], int b0, int b1, bool first, bool last) {
      if (first) { cur = 1.0; key = 0; }
#pragma unroll
      for (int j = 0; j < RB; ++j)
        if (b0 + j <= b1) { key = max(key, (unsigned)__double2hiint(r[j])); cur *= r[j]; }
      if (last) {
        // The scale of a unary message cancels in everything downstream (only its normalised form is ever stored, by
        // unary_writeback_kernel), so the raw columns are multiplied and the PRODUCT is normalised once.  A column
        // Message.renormalize would replace by the uniform vector (total <= 0, LBP.py:655-657) zeroes the product, and an entry
        // that is negative, not finite or huge shows in the high words: either sends the graph to the exact kernel, decided once
        // per product.
        const double sum = wave_sum(cur);
        flagged |= !total_ok(sum) || __any(key > KEY_LIMIT);
        // (1 / sum by the hardware reciprocal and one Newton step, 2^-54: the product-fused sweep kernel takes this tile's total to
        // be 1 when it writes the tile out as a message)
        double inv_sum = __builtin_amdgcn_rcp(sum);
        inv_sum = __builtin_fma(__builtin_fma(-sum, inv_sum, 1.0), inv_sum, inv_sum);
        if (!ABL(65536)) ptile_lds[(size_t)k_out * TILE + tile_index(lane, col)] = cur * inv_sum;
        ++k_out;
      }
    };
    double ra[RB], rc[RB];
    int a0 = 0, a1 = 0, c0_ = 0, c1_ = 0;
    bool have_a = next_batch(a0, a1), have_c = false;
    bool a_first = !open_product;
    if (have_a) fetch(ra, a0, a1);
    while (have_a) {
      const bool a_last = ((real_ends >> a1) & 1) != 0;
      have_c = next_batch(c0_, c1_);
      if (have_c) fetch(rc, c0_, c1_);
      reduce(ra, a0, a1, a_first, a_last);
      open_product = !a_last;
      if (!have_c) break;
      const bool c_first = a_last, c_last = ((real_ends >> c1_) & 1) != 0;
      have_a = next_batch(a0, a1);
      if (have_a) fetch(ra, a0, a1);
      reduce(rc, c0_, c1_, c_first, c_last);
      a_first = c_last;
      open_product = !c_last;
    }
  }
  if (__any(flagged)) flagged = true;
  if (lane == 0) d.bail[g] = flagged ? 1 : 0;
  if (d.grad_on) {                                               // lanes k, k + 8, ..., k + 56 meet: xor 8, 16, 32
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) { u_ee += __shfl_xor(u_ee, m); u_ed += __shfl_xor(u_ed, m); }
    if (lane < 3) d.grad_en_en[(size_t)g * 3 + lane] = u_ee;
    if (lane < 6) d.grad_en_de[(size_t)g * 6 + lane] = u_ed;
  }
  }   // if (on)
  int first_row = 0;
  if (d.header) {
    first_row = __shfl(pair_rows, (lane & 15) + 16);
    const bool mine_lane = lane < 16 && lane < d.P;
    const int all_in_range = __all(!mine_lane || (unsigned)pair_rows < (unsigned)d.n_pair_tables), all_same = __all(!mine_lane || pair_rows == first_row);
    if (lane == 0) { hdr_ok[wave] = all_in_range ? 1 : 0; hdr_same[wave] = all_same ? 1 : 0; }
  }
  __syncthreads();
  if (!ABL(16)) {
    const double2* src = reinterpret_cast<const double2*>(ptile_lds);
    double2* dst = reinterpret_cast<double2*>(d.ptiles + (size_t)block * d.n_cprod * TILE);
    if (d.sqrt_mask == 0) {
      for (int i = t; i < d.n_cprod * (TILE / 2); i += PWG) dst[i] = src[i];        // (columns of graphs beyond B: whatever LDS held; never read as results)
    } else {
      // a variable with three pairwise factors: the sweep kernel stores sqrt(c) (.) message (build_shared_program, three-source
      // product-fused form) and wants sqrt(c) here -- taken on the way out, not in the products' loop
      for (int i = t; i < d.n_cprod * (TILE / 2); i += PWG) {
        double2 v = src[i];
        if ((d.sqrt_mask >> (i / (TILE / 2))) & 1) { v.x = sqrt(v.x); v.y = sqrt(v.y); }
        dst[i] = v;
      }
    }
  }
  if (d.header && wave == PWG / 64 - 1 && !ABL(32768)) {
    // ... and wave 0 finishes it: the distinct tables, which factor reads which, and the partition that splits most bundles
    const bool ok = __all(lane >= PGB || hdr_ok[lane & (PGB - 1)] != 0), same = __all(lane >= PGB || hdr_same[lane & (PGB - 1)] != 0);
    const int d0 = __builtin_amdgcn_readfirstlane(first_row);
    const unsigned long long in_p = d.P >= 64 ? ~0ull : ((1ull << d.P) - 1);
    const unsigned long long second = __ballot(first_row != d0) & in_p;
    const int d1 = second ? __shfl(first_row, __builtin_ctzll(second)) : d0;
    const bool over = (__ballot(first_row != d0 && first_row != d1) & in_p) != 0;
    const int rankmask = (int)second;
    int mode = 1;
    if (d1 != d0) {
      int c0 = 0, c1 = 0, c2 = 0;
      for (int b0 = 0; b0 < d.n_bundles; b0 += 64) {
        const int i = b0 + lane;
        bool p0 = false, p1 = false, p2 = false;
        if (i < d.n_bundles) {
          const int fa = d.image[(size_t)i * 2 * MW], fb = d.image[(size_t)i * 2 * MW + MW];
          if ((fa & 1) && (fb & 1)) {
            const int sa = 2 * ((rankmask >> ((fa >> 16) & 15)) & 1) + ((fa >> 1) & 1);
            const int sb = 2 * ((rankmask >> ((fb >> 16) & 15)) & 1) + ((fb >> 1) & 1);
            p0 = half_of(0, sa) != half_of(0, sb); p1 = half_of(1, sa) != half_of(1, sb); p2 = half_of(2, sa) != half_of(2, sb);
          }
        }
        c0 += __popcll(__ballot(p0)); c1 += __popcll(__ballot(p1)); c2 += __popcll(__ballot(p2));
      }
      mode = (c0 >= c1 && c0 >= c2) ? 0 : (c1 >= c2 ? 1 : 2);
    }
    if (lane < HDR) {
      const int w = lane == 0 ? (!ok ? 1 : ((!same || over) ? 2 : 0)) : (lane == 1 ? d0 : (lane == 2 ? d1 : (lane == 3 ? rankmask : (lane == 4 ? mode : 0))));
      d.header[(size_t)block * HDR + lane] = w;
    }
  }
}

// 16 consecutive words through the scalar data cache (s_load_dwordx16): wave-uniform program data lands in SGPRs.
struct Words16 { int32_t w[16]; };
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ Words16 sload16(const int32_t* p) {
  const v16i v = *(const v16i __attribute__((address_space(4)))*)(uintptr_t)p;
  Words16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = v[i];
  return r;
}

// The gradient epilogue of the sweep kernel (below).  It forms every lane-derived value again from the thread id: values shared with
// the code in front of the main loop would be held in registers across it, and the loop's fragments would go to scratch inside the
// MFMA sequences.  (As a real function call it paid the ABI's callee-saved registers and spills of its own: 70 -> 95 us.)
// DIRECT (product-fused kernel, SharedProgram::vf_direct): a factor's two variable->factor messages are message tiles the sweeps
// left in LDS -- c (.) message, unnormalised: any positive per-graph scale cancels in S / Z -- so nothing is written to the message
// buffer during the sweeps and nothing read back here; partial sums and per-graph terms live in the totals' rows (spent) and the
// spare LDS behind them.
template <bool DIRECT, int NPMAX, typename Dev>
__device__ __forceinline__ void shared_gradient_epilogue(Dev& d, const int wg) {
  extern __shared__ double lds[];
  double* tiles = lds;
  double* tot = tiles + (size_t)d.n_res * TILE;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = wave >> 2, rb = wave & 3;
  const int g0 = wg * G;
  const double uniform = 1.0 / 64.0;
  const const_i32p img = as_const(d.image);
  const const_i32p row0 = as_const(d.pair_tab + (size_t)g0 * d.P);
  double fr0[16], fr1[16];
  STAMP_DECL
  ABL_DECL
  STAMP_START
  if (DIRECT && t == 0) *reinterpret_cast<double2*>(reinterpret_cast<char*>(lds) + d.lds_bytes - 16) = make_double2(uniform, uniform);
  __syncthreads();
  // factors per pass -- memory form: two tiles each, one tile of partial sums; direct form: what the totals' rows and the spare LDS
  // behind them hold (per factor 256 partial sums + 48 per-graph terms)
  const int NP = DIRECT ? min(NPMAX, (int)((d.lds_bytes - 16 - (size_t)d.n_res * TILE * sizeof(double)) / ((256 + 48) * sizeof(double)))) : min(NPMAX, (d.n_res - 1) >> 1);
  double* red = DIRECT ? tot : tiles + (size_t)2 * NP * TILE;    // [factor][k][row block][graph]
  double* pc = DIRECT ? tot + (size_t)NP * 256 : tot;            // [factor][graph][3] per-graph terms of a pass (the totals are spent)
  // (direct form: the uniform vector's 16 bytes at the very end of the workgroup's LDS -- the sweeps' constants lie inside `red`)
  const double2* const uni2 = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(lds) + d.lds_bytes - 16);
  double out3[3] = {0.0, 0.0, 0.0};                              // thread t < 16: graph g0 + t
  for (int p0 = 0; p0 < d.P; p0 += NP) {
    const int np = min(NP, d.P - p0);
    // thread (factor t >> 4, graph t & 15) of the first np * 16: the factor's labels ...
    const int lp = p0 + (t >> 4), lg = g0 + (t & 15);
    const bool l_on = t < np * G && lg < d.B;
    int m0 = 0, m1 = 0;
    if (l_on) { m0 = d.gr.pair_label[((size_t)lg * d.P + lp) * 2]; m1 = d.gr.pair_label[((size_t)lg * d.P + lp) * 2 + 1]; }
    // ... the messages: wave w takes rows i = w + 8 j of the pass's np * 32 (factor i >> 5, side (i >> 4) & 1, graph i & 15), one
    // 512-byte row each.  The slots (and whether the program ever writes them) come through the scalar cache -- a vector
    // load here would be one more round of memory latency in front of the rows --, then all the rows are requested
    // before the first is used.  The half's first A fragment goes out ahead of them.
    const const_i32p cs = as_const(d.gr.c_slot), rs = as_const(d.gr.r_slot), pf = as_const(d.gr.plane_flags);
    int slots[2 * NPMAX];                                                 // tile q = 2 * factor + side: its message slot, or -1
#pragma unroll
    for (int q = 0; q < 2 * NPMAX; ++q) slots[q] = q < 2 * np ? ((q & 1) ? cs[p0 + (q >> 1)] : rs[p0 + (q >> 1)]) : -1;
#pragma unroll
    for (int q = 0; q < 2 * NPMAX; ++q) {
      const bool in = (unsigned)slots[q] < (unsigned)d.n_msgs;
      const int w = img[d.off_written + (in ? slots[q] >> 5 : 0)];
      if (!in || !((w >> (slots[q] & 31)) & 1)) slots[q] = -1;
    }
    // direct form: tile q's LDS tile (or -1: the uniform vector, read from the constants with stride 0)
    int vt[2 * NPMAX];
#pragma unroll
    for (int q = 0; q < 2 * NPMAX; ++q) vt[q] = (DIRECT && slots[q] >= 0) ? img[d.off_vftile + slots[q]] : -1;
    const int n_items = 2 * np;
    auto key = [&](int j) { const int pp = j % np; return row0[p0 + pp] * 2 + (as_const(d.gr.pair_phi)[p0 + pp] ? 1 : 0); };
    auto fetchw = [&](double (&fr)[16], int j) {
      const int k = 2 * (j / np) + half;
      const double* W = d.gr.wfrag + ((size_t)key(j) * 4 + k) * 4096 + rb * 1024 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) fr[s] = ABL(256) ? 0.5 : W[64 * s];
    };
    auto run_end = [&](int j) {                                  // first item behind the run that starts at j
      int e = j + 1;
      while (e < n_items && e % np != 0 && key(e) == key(e - 1)) ++e;
      return e;
    };
    // an item whose feature plane is all zeros or all ones needs no contraction (the per-graph terms below use 0 resp. Z)
    auto needed = [&](int j) { const int k = 2 * (j / np) + half; return k == 3 || pf[(key(j) & 1) * 4 + k] == 0; };
    auto next_run = [&](int j) {                                  // start of the first needed run at or behind j
      while (j < n_items && !needed(j)) j = run_end(j);
      return j;
    };
    const int j_first = next_run(0);
    if (j_first < n_items) fetchw(fr0, j_first);
    double sv[4 * NPMAX];
    if constexpr (!DIRECT) {
#pragma unroll
    for (int j = 0; j < 4 * NPMAX; ++j) {                                // row i = wave + 8 j: tile i >> 4 = j >> 1, graph wave + 8 (j & 1)
      sv[j] = uniform;                                            // never updated: still uniform (LBP.py:211-216)
      if (j < 4 * np && slots[j >> 1] >= 0 && !ABL(32)) {
        const int ggc = min(g0 + wave + 8 * (j & 1), d.B - 1);
        sv[j] = d.msgs[((size_t)ggc * d.n_msgs + slots[j >> 1]) * 64 + lane];
      }
    }
    }
    // ... and the label's feature row (the labels have arrived, the messages are still on their way)
    double lf[3] = {0.0, 0.0, 0.0};
    const bool l_ok = l_on && (unsigned)m0 < 64u && (unsigned)m1 < 64u;
    if (l_on && !l_ok) atomicExch(d.status, 1);
    if (l_ok) {
      const double* ph = d.gr.phi[d.gr.pair_phi[lp] ? 1 : 0] + ((size_t)m0 * 64 + m1) * 3;
      lf[0] = ph[0]; lf[1] = ph[1]; lf[2] = ph[2];
    }
    if constexpr (!DIRECT) {
#pragma unroll
    for (int j = 0; j < 4 * NPMAX; ++j)
      if (j < 4 * np) {
        tiles[(size_t)(j >> 1) * TILE + tile_index(lane, wave + 8 * (j & 1))] = sv[j];
      }
    __syncthreads();
    }
    STAMP(8)
    // tile q = 2 * factor + side of this pass: where it is read from (16 bytes per lane and k-step pair), and the strides of a read
    auto tile_ptr = [&](int q) -> const double2* {
      if constexpr (DIRECT) {
        int tl = -1;
#pragma unroll
        for (int i = 0; i < 2 * NPMAX; ++i) if (i == q) tl = vt[i];
        return tl >= 0 ? reinterpret_cast<const double2*>(tiles + (size_t)tl * TILE) + lane : uni2;
      } else {
        return reinterpret_cast<const double2*>(tiles + (size_t)q * TILE) + lane;
      }
    };
    auto tile_on = [&](int q) -> bool {                          // false: the uniform constants (every read the same 16 bytes)
      if constexpr (DIRECT) {
        int tl = -1;
#pragma unroll
        for (int i = 0; i < 2 * NPMAX; ++i) if (i == q) tl = vt[i];
        return tl >= 0;
      } else {
        return true;
      }
    };
    // this half's items in k-major order: j = kk * np + pp  ->  factor p0 + pp, feature k = 2 kk + half.  Consecutive
    // factors that read the same (table, feature tensor) share one A fragment: it is fetched once per run (a K3 user
    // graph's three factors are one run: 2 fetches per wave instead of 6), the next run's while this run multiplies; two
    // items of a run go through the matrix pipe interleaved (two independent accumulation chains).
    auto finish = [&](const double4_t& acc, int pp, int k) {
      const bool on_c = tile_on(2 * pp + 1);
      const double2* ct = tile_ptr(2 * pp + 1) + (on_c ? 128 * rb : 0);
      const double2 c0 = ct[0], c1 = ct[on_c ? 64 : 0];
      const double part = column_sum((c0.x * acc.x + c0.y * acc.y) + (c1.x * acc.z + c1.y * acc.w));
      if ((lane >> 4) == 0) red[((pp * 4 + k) * 4 + rb) * G + (lane & 15)] = part;
    };
    auto item = [&](const double (&fr)[16], int j) {
      const int pp = j % np, k = 2 * (j / np) + half;
      const double2* rt = tile_ptr(2 * pp);
      const int rs_ = tile_on(2 * pp) ? 128 : 0, ro_ = tile_on(2 * pp) ? 64 : 0;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const double2 v0 = rt[rs_ * h], v1 = rt[rs_ * h + ro_];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h], v0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 1], v0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 2], v1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 3], v1.y, acc, 0, 0, 0);
      }
      finish(acc, pp, k);
    };
    auto item2 = [&](const double (&fr)[16], int j) {            // items j and j + 1: the same fragment, the next factor
      const int pp = j % np, k = 2 * (j / np) + half;
      const double2* rt = tile_ptr(2 * pp);
      const double2* rt2 = tile_ptr(2 * pp + 2);
      const int rs_ = tile_on(2 * pp) ? 128 : 0, ro_ = tile_on(2 * pp) ? 64 : 0, rs2 = tile_on(2 * pp + 2) ? 128 : 0, ro2 = tile_on(2 * pp + 2) ? 64 : 0;
      double4_t acc = {0.0, 0.0, 0.0, 0.0}, bcc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const double2 v0 = rt[rs_ * h], v1 = rt[rs_ * h + ro_], w0 = rt2[rs2 * h], w1 = rt2[rs2 * h + ro2];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h], v0.x, acc, 0, 0, 0);
        bcc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h], w0.x, bcc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 1], v0.y, acc, 0, 0, 0);
        bcc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 1], w0.y, bcc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 2], v1.x, acc, 0, 0, 0);
        bcc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 2], w1.x, bcc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 3], v1.y, acc, 0, 0, 0);
        bcc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[4 * h + 3], w1.y, bcc, 0, 0, 0);
      }
      finish(acc, pp, k);
      finish(bcc, pp + 1, k);
    };
    for (int j = ABL(64) ? n_items : j_first; j < n_items;) {
      int e = run_end(j), nx = next_run(e);
      if (nx < n_items) fetchw(fr1, nx);
#pragma unroll 1
      for (; j + 1 < e; j += 2) item2(fr0, j);
      if (j < e) item(fr0, j);
      j = nx;
      if (j >= n_items) break;
      e = run_end(j); nx = next_run(e);
      if (nx < n_items) fetchw(fr0, nx);
#pragma unroll 1
      for (; j + 1 < e; j += 2) item2(fr1, j);
      if (j < e) item(fr1, j);
      j = nx;
    }
    __syncthreads();
    STAMP(9)
    if (t < np * G) {                                            // label features minus expected features, per (factor, graph)
      const double* rp = red + (size_t)(t >> 4) * 16 * G + (t & 15);
      const double Z = (rp[(12 + 0) * G] + rp[(12 + 1) * G]) + (rp[(12 + 2) * G] + rp[(12 + 3) * G]);
      const int wh = d.gr.pair_phi[p0 + (t >> 4)] ? 1 : 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int flag = d.gr.plane_flags[wh * 4 + k];
        const double S = (rp[(4 * k + 0) * G] + rp[(4 * k + 1) * G]) + (rp[(4 * k + 2) * G] + rp[(4 * k + 3) * G]);
        // expected feature: S / Z; a zero plane 0, the bias plane the normalised belief's total, 1  (au.normalize: zero-sum -> 0)
        const double ex = Z > 0.0 ? (flag == 0 ? S / Z : (flag == 1 ? 0.0 : 1.0)) : 0.0;
        pc[t * 3 + k] = l_ok ? lf[k] - ex : 0.0;
      }
    }
    __syncthreads();
    if (t < G)
      for (int pp = 0; pp < np; ++pp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out3[k] += pc[(pp * G + t) * 3 + k];
      }
    __syncthreads();
    STAMP(10)
  }
  // the unary factors' terms are in the output already (shared_prepare_kernel: they do not depend on the sweeps)
  if (t < G && g0 + t < d.B) {
#pragma unroll
    for (int k = 0; k < 3; ++k) d.gr.grad_en_en[(size_t)(g0 + t) * 3 + k] += out3[k];
  }
  STAMP(11)
  STAMP_FLUSH_GRAD
}

// One workgroup = 16 graphs x 8 waves.  Wave w: half h = w >> 2, row block r = w & 3.  Each half keeps TWO of the four
// (table, orientation) fragment sets in registers (64 VGPRs instead of 128: four waves per SIMD instead of two), and a
// bundle's two independent updates run side by side, one per half, under one barrier -- the dependent chain of a
// three-sweep K3 program is 9 bundles instead of 18 updates, and every SIMD has four waves to interleave matrix work
// with tile reads.  Which sets a half holds (one of three partitions) is chosen on the device from the program and the
// batch's table indices so that as many bundles as possible split.
// MULTI: the launch holds several GROUPS of graphs, each with its own program and buffers (mlbp_sweep_groups_f64): gtab[k] is
// group k's SharedDev, gstart[k] its first workgroup; the workgroup looks its group up and runs as if launched for it alone.
// (the body: `d` is the kernel argument, or -- MULTI -- a reference into the group table through the scalar data cache, so
// that in both forms the description sits in SGPRs / is fetched by scalar loads where it is used)
template <int NTAB, bool SPILL, bool WIDE, bool GRAD, bool PF, bool P3, typename Dev>
__device__ __forceinline__ void sweep_x64_shared_body(Dev& d, const int wg) {
  static_assert(!PF || (!SPILL && !WIDE), "the product-fused form keeps every tile in LDS and reads one tile per update");
  static_assert(!P3 || PF, "the three-source form is a product-fused form");
  extern __shared__ double lds[];
  double* tiles = lds;                                           // [n_res][64 states][16 graphs]
  double* tot = tiles + (size_t)d.n_res * TILE;                  // [n_live][16 graphs][4 row blocks] partial column sums
  double2* dummy = reinterpret_cast<double2*>(tot + (size_t)d.n_live * 64);          // {1/64, 1/64}, {1, 1}, {0.25, 0.25} x 2: absent sources
  int32_t* limg = reinterpret_cast<int32_t*>(dummy + 4);                                    // [n_bundles + 1][8] the bundles (read one ahead, broadcast)
  double* spill = (SPILL && d.spill) ? d.spill + (size_t)wg * (d.n_live - d.n_res) * TILE : nullptr;
  // tile t: LDS when resident, else this workgroup's slice of the global spill area (same [state][graph] layout;
  // __syncthreads orders the workgroup's global accesses as it does the LDS ones)
  auto TP = [&](int tile) -> double* {
    if (!SPILL) return tiles + (size_t)tile * TILE;
    return tile < d.n_res ? tiles + (size_t)tile * TILE : spill + (size_t)(tile - d.n_res) * TILE;
  };
  const int t = threadIdx.x, lane = t & 63, lane_ = lane, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = wave >> 2, rb = wave & 3;
  const int gl = lane & 15, cq = lane >> 4;                      // B/D operand: graph column, k-row
  const int g0 = wg * G;
  const int gi = g0 + gl;
  const bool gvalid = gi < d.B;
  const int gc = gvalid ? gi : d.B - 1;                          // tail columns replay the last graph, outputs masked
  const double uniform = 1.0 / 64.0;
  const const_i32p img = as_const(d.image);

  STAMP_DECL
  ABL_DECL
  STAMP_START
  // ---- prologue: ONE round of loads.  shared_prepare_kernel has left the group's header (verdict on the 16 graphs' table
  //      indices, the distinct tables, which factor reads which, the partition of the fragment sets over the halves -- read through
  //      the scalar cache), the constant products as tiles and the per-graph verdicts; the bundle records, the product tiles
  //      and the verdicts are requested before the header is waited for, the fragments as soon as it is here ----
  const const_i32p hd = as_const(d.header + (size_t)wg * HDR);
  const int verdict = hd[0], d0 = hd[1], d1 = hd[2], rankmask = hd[3];
  const int mode = hd[4];
  // the bundle records go to LDS RESOLVED: slot [bundle][half] = the member that half runs (flag bit 7: its fragments are the half's
  // second set), so the main loop reads one record and decides nothing.  Two members whose fragment sets one half holds (two
  // tables, a bundle the partition does not split): that half's record carries CHAIN and the other slot the second member parked
  // (its flag byte moved to bits 24-31) -- the half runs both, one after the other.
  constexpr int CHAIN = 1 << 30;
  {
    const bool single_ = d1 == d0;
    const int32_t* gb = d.image + (PF ? d.off_pfb : 0);
    int4* lo = reinterpret_cast<int4*>(limg);
    for (int k = t; k < d.n_bundles + 1; k += SWG) {
      const int32_t* b = gb + 2 * MW * (size_t)k;
      int4 A = make_int4(b[0], b[1], b[2], b[3]), B = make_int4(b[4], b[5], b[6], b[7]);
      const int sa = 2 * ((rankmask >> ((A.x >> 16) & 15)) & 1) + ((A.x >> 1) & 1), sb = 2 * ((rankmask >> ((B.x >> 16) & 15)) & 1) + ((B.x >> 1) & 1);
      int hA = ((A.x & 1) && !single_ && !P3) ? half_of(mode, sa) : -1, hB = ((B.x & 1) && !single_ && !P3) ? half_of(mode, sb) : -1;
      if (hA < 0) hA = hB >= 0 ? 1 - hB : 0;                     // a member that can run anywhere goes to the idle half
      if (hB < 0) hB = 1 - hA;
      const bool runA = (A.x & 0x7F) != 0, runB = (B.x & 0x7F) != 0;
      if (P3) {
        // (three-source form: every wave keeps all four fragment sets -- 256 registers a wave --, any member runs on either half and
        // a bundle always splits; bit 7 = the member's table is the second one)
        if (A.x & 1) A.x |= (sa >> 1) << 7;
        if (B.x & 1) B.x |= (sb >> 1) << 7;
      } else {
      if (A.x & 1) A.x |= index_in_half(mode, sa) << 7;
      if (B.x & 1) B.x |= index_in_half(mode, sb) << 7;
      }
      const int4 nop4 = make_int4(0, 0, 0, 0);
      int4 s0 = nop4, s1 = nop4;
      if (!(runA && runB && hA == hB)) {
        if (runA) { if (hA == 0) s0 = A; else s1 = A; }
        if (runB) { if (hB == 0) s0 = B; else s1 = B; }
      } else {                                                   // both on half hA
        int4 parked = B;
        parked.x = (int)(((unsigned)B.x & 0xFFu) << 24) | (B.x & 0x00FFFF00);
        A.x |= CHAIN;
        if (hA == 0) { s0 = A; s1 = parked; } else { s1 = A; s0 = parked; }
      }
      lo[2 * k] = s0; lo[2 * k + 1] = s1;
    }
  }
  if (t < 4) dummy[t] = t == 0 ? make_double2(1.0 / 64.0, 1.0 / 64.0) : (t == 1 ? make_double2(1.0, 1.0) : make_double2(0.25, 0.25));
  bool bad = d.bail[gc] != 0;                                    // the prepare kernel's verdict on this lane's graph
  const double2* psrc = reinterpret_cast<const double2*>(d.ptiles + (size_t)wg * d.n_cprod * TILE);
  double2 pv[8];                                                 // (n_cprod <= 8: shared_plan)
  int ptile[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ptile[k] = k < d.n_cprod ? img[d.off_ptile + k] : 0;
    pv[k] = k < d.n_cprod ? psrc[(size_t)k * (TILE / 2) + t] : make_double2(0.0, 0.0);       // SWG threads x 16 bytes = one tile
  }
  // which fragment sets this half keeps, then the fragments (lane: row l & 15, k l >> 4).  One table: BOTH halves keep its two
  // orientations (partition 1 with T1 = T0), so any member runs on either half and a bundle always splits
  const bool ok = verdict == 0 && !(NTAB == 1 && d1 != d0);
  double fr0[16], fr1[16];
  double fr2[P3 ? 16 : 1], fr3[P3 ? 16 : 1];                     // (three-source form: the second table's two orientations; fr0 / fr1: the first's)
  if constexpr (P3) {
    if (ok && d.tfrag && d1 != d0 && NTAB > 1) {
      const double* F = d.tfrag + (size_t)d1 * 2 * 4096 + rb * 1024 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) { fr2[s] = F[64 * s]; fr3[s] = F[4096 + 64 * s]; }
    } else if (ok && d1 != d0 && NTAB > 1) {
      const double* T = d.pair_tables + (size_t)d1 * 4096;
#pragma unroll
      for (int s = 0; s < 16; ++s) { fr2[s] = T[(16 * rb + gl) * 64 + 4 * s + cq]; fr3[s] = T[(4 * s + cq) * 64 + 16 * rb + gl]; }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) { fr2[s] = 0.0; fr3[s] = 0.0; }
    }
  }
#pragma unroll
  for (int idx = 0; idx < 2; ++idx) {
    int sel = 0;
    for (int q = 0; q < 4; ++q)
      if (half_of(mode, q) == half && index_in_half(mode, q) == idx) sel = q;
    if (P3) sel = idx;                                           // (first table, orientation idx)
    const int ti = (sel >> 1) ? d1 : d0, mt = sel & 1;
    double (&fr)[16] = idx ? fr1 : fr0;
    if (ok && d.tfrag) {
      // copies in operand order (shared_prepare_kernel): one contiguous 512-byte read per fragment
      const double* F = d.tfrag + ((size_t)ti * 2 + mt) * 4096 + rb * 1024 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) fr[s] = ABL(1024) ? 0.5 : F[64 * s];
    } else if (ok) {
      const double* T = d.pair_tables + (size_t)ti * 4096;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        fr[s] = mt ? T[(4 * s + cq) * 64 + 16 * rb + gl]          // (m^T.T)[x]: A[x][y] = T[y][x]
                   : T[(16 * rb + gl) * 64 + 4 * s + cq];         // (T.m)[x]  : A[x][y] = T[x][y]
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) fr[s] = 0.0;
    }
  }
  if (verdict == 1) {                                            // a table index out of range
    if (t == 0) atomicExch(d.status, 1);
    return;
  }
  if (!ok) {                                                     // not a shared-table batch: the exact kernel takes all 16
    if (t < G && g0 + t < d.B) d.bail[g0 + t] = 4;
    return;
  }
  if (!P3) {                                                     // (three-source form: the constant products stay in memory)
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < d.n_cprod) {
      reinterpret_cast<double2*>(TP(ptile[k]))[t] = pv[k];
      if (t < 64) tot[ptile[k] * 64 + t] = 0.25;                 // four partials of a total of 1
    }
  }
  // tiles the program reads before writing them start as the uniform vector (FactorGraph.initialize, LBP.py:211-216)
  if (PF) {
    // (product-fused: a message tile holds c (.) message; before the first update that is c (.) uniform -- a copy of c)
    for (int k = 0; k < d.n_pinit; ++k) {
      const int tile = img[d.off_pinit + 2 * k], prod = img[d.off_pinit + 2 * k + 1];
      double2 v = make_double2(uniform, uniform);
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q == prod) v = pv[q];
      reinterpret_cast<double2*>(TP(tile))[t] = v;
      if (t < 64) tot[tile * 64 + t] = 0.25;
    }
  } else
  for (int k = 0; k < d.n_init; ++k) {
    const int tile = img[d.off_init + k];
    reinterpret_cast<double2*>(TP(tile))[t] = make_double2(uniform, uniform);
    if (t < 64) tot[tile * 64 + t] = 0.25;
  }
  if (d.msgs && !d.vf_only)                                       // slots the sweeps never touch stay uniform
    for (int i = t; i < d.n_fill * G * 64; i += SWG) {
      const int x = i & 63, gg = (i >> 6) & (G - 1), k = i >> 10;
      if (g0 + gg < d.B) d.msgs[((size_t)(g0 + gg) * d.n_msgs + d.image[d.off_fill + k]) * 64 + x] = uniform;
    }
  STAMP(0)
  __syncthreads();
  if (SPILL) __syncthreads();                                    // (spilled tiles were written through global memory)
  STAMP(1)

  // ---- main loop: one barrier per bundle; a bundle's members go to the halves that hold their fragments ----
  // One member, run by the four waves of a half.  The K = 64 contraction goes in four quarters of four k-steps: a
  // quarter's tile reads and products issue while the previous quarter's MFMAs run, and only 4 + 4 message registers are
  // live beside the 64 fragment registers (the budget is 128 at four waves per SIMD).
  // The next bundle's descriptor is requested (scalar load) once this member's LDS reads are over -- scalar loads
  // and LDS reads share one counter, so a request in front of them would be waited for with them.
  // WIDE = false (programs whose variable updates multiply at most two tiles -- a constant product and one message: K2, K3,
  // chains, rings): every LDS read of a quarter is issued together, and quarter h + 1's reads are issued BEFORE quarter h's
  // MFMAs, so the only exposed LDS latency of a member is its first one.  WIDE = true: any number of sources, read
  // one after the other (the straight-line form for three sources does not fit the 128 registers).
  auto run = [&](const int flags, const int w1, const int w3, const int w4) {
    // (lane-derived values are formed again per member from a laundered copy of the lane id: kept across the loop they cost
    // registers the fragments need -- the compiler otherwise spills FRAGMENTS to scratch inside the MFMA sequence)
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int gl = lane & 15, cq = lane >> 4;
    const int nsrc = (flags >> 8) & 15;
    const bool want = (flags & (4 | 8 | 16)) != 0, mm = (flags & 1) != 0;
    const bool second_set = (flags & 0x80) != 0;                 // (resolved by the prologue)
    const int s0 = w3 & 0xFF, s1 = (w3 >> 8) & 0xFF;
    const bool has0 = s0 != 0xFF && !ABL(2), has1 = nsrc > 1 && !ABL(2);
    // an absent source reads a 16-byte constant instead (every lane the same address, stride 0): {1/64, 1/64} for a first
    // source nothing has updated yet (LBP.py:211-216), {1, 1} for a second one -- so that every read below is
    // unconditional (conditional reads double the live registers)
    const double2* src0 = has0 ? reinterpret_cast<const double2*>(TP(s0)) + lane : dummy;
    const double2* src1 = has1 ? reinterpret_cast<const double2*>(TP(s1)) + lane : dummy + 1;
    const int st0 = has0 ? 128 : 0, of0 = has0 ? 64 : 0, st1 = has1 ? 128 : 0, of1 = has1 ? 64 : 0;
    double2 r0a, r0b, r1a, r1b;
    double scale = 1.0;
    if (!WIDE) {
      // all first reads of the member at once: the sources' partial totals and their first quarter.  The scale of the
      // product cancels downstream (every stored message is normalised by its own total); the reciprocals only keep the
      // magnitudes in range: the hardware reciprocal is enough
      const double2* tp0 = has0 ? reinterpret_cast<const double2*>(tot + s0 * 64 + gl * 4) : dummy + 2;
      const double2* tp1 = has1 ? reinterpret_cast<const double2*>(tot + s1 * 64 + gl * 4) : dummy + 2;
      const double2 ta = tp0[0], tb = tp0[1], tc = tp1[0], td = tp1[1];
      r0a = src0[0]; r0b = src0[of0];
      r1a = src1[0]; r1b = src1[of1];
      const double t0 = (ta.x + ta.y) + (tb.x + tb.y), t1 = (tc.x + tc.y) + (td.x + td.y);
      bad |= (int)!total_ok(t0) | (int)!total_ok(t1);
      scale = __builtin_amdgcn_rcp(t0) * __builtin_amdgcn_rcp(t1);
    } else {
#pragma unroll 1
      for (int q = 0; q < nsrc; ++q) {
        const int tl = ((q < 4 ? w3 : w4) >> (8 * (q & 3))) & 0xFF;
        if (tl != 0xFF) {
          const double* tp = tot + tl * 64 + gl * 4;
          const double total = (tp[0] + tp[1]) + (tp[2] + tp[3]);
          bad |= !total_ok(total);
          scale *= __builtin_amdgcn_rcp(total);
        }
      }
    }
    STAMP(6)
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                                // quarter h = states 16 h .. 16 h + 15 = k-steps 4 h .. 4 h + 3
      double b[4];
      if (!WIDE) {
        // variable -> factor (LBP.py:377-389): constant product (or uniform) times the other incoming message
        b[0] = (r0a.x * scale) * r1a.x; b[1] = (r0a.y * scale) * r1a.y; b[2] = (r0b.x * scale) * r1b.x; b[3] = (r0b.y * scale) * r1b.y;
        __builtin_amdgcn_sched_barrier(0);
        if (h < 3) {                                             // the next quarter's reads go out ahead of this quarter's MFMAs
          r0a = src0[st0 * (h + 1)]; r0b = src0[st0 * (h + 1) + of0];
          r1a = src1[st1 * (h + 1)]; r1b = src1[st1 * (h + 1) + of1];
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        {
          const double2 v0 = src0[st0 * h], v1 = src0[st0 * h + of0];
          b[0] = v0.x * scale; b[1] = v0.y * scale; b[2] = v1.x * scale; b[3] = v1.y * scale;
        }
#pragma unroll 1
        for (int q = 1; q < nsrc && !ABL(2); ++q) {
          const int tl = ((q < 4 ? w3 : w4) >> (8 * (q & 3))) & 0xFF;
          const double2* sq = reinterpret_cast<const double2*>(TP(tl)) + lane + 128 * h;
          const double2 u0 = sq[0], u1 = sq[64];
          b[0] *= u0.x; b[1] *= u0.y; b[2] *= u1.x; b[3] *= u1.y;
        }
      }
      if (want) {                                                // the message itself is wanted: its total, and this wave's quarter
        part += (b[0] + b[1]) + (b[2] + b[3]);
        if (rb == h && (flags & 4)) {                            // read again later: keep it as a tile
          double2* o = reinterpret_cast<double2*>(TP((w1 >> 8) & 0xFF)) + 128 * rb + lane;
          o[0] = make_double2(b[0], b[1]); o[64] = make_double2(b[2], b[3]);
        }
      }
      STAMP(2)
      // factor -> variable (LBP.py:500-524): v_mfma_f64_16x16x4_f64 against the resident fragments
      if (mm && ABL(1)) { acc.x += b[0]; acc.y += b[1]; acc.z += b[2]; acc.w += b[3]; }
      else if (mm) {
        if (!second_set) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr0[4 * h + i], b[i], acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr1[4 * h + i], b[i], acc, 0, 0, 0);
        }
      }
      STAMP(7)
    }
    if (want) {
      const double tb = column_sum(part);
      bad |= !total_ok(tb);
      if (flags & 4) {
        if (cq == 0) tot[((w1 >> 8) & 0xFF) * 64 + gl * 4 + rb] = 0.25 * tb;
      } else if ((flags & 8) && d.msgs && g0 + gl < d.B && !bad) {      // last value of this slot: straight to HBM, normalised --
        double kq[4];                                            // this wave's quarter is formed once more now that the total is known
        {
          const double2 v0 = src0[st0 * rb], v1 = src0[st0 * rb + of0];
          kq[0] = v0.x * scale; kq[1] = v0.y * scale; kq[2] = v1.x * scale; kq[3] = v1.y * scale;
        }
#pragma unroll 1
        for (int q = 1; q < nsrc; ++q) {
          const double2* sq = reinterpret_cast<const double2*>(TP(((q < 4 ? w3 : w4) >> (8 * (q & 3))) & 0xFF)) + lane + 128 * rb;
          const double2 v0 = sq[0], v1 = sq[64];
          kq[0] *= v0.x; kq[1] *= v0.y; kq[2] *= v1.x; kq[3] *= v1.y;
        }
        const int gcv = g0 + gl;
        double* out = d.msgs + ((size_t)gcv * d.n_msgs + ((w1 >> 16) & 0xFFFF)) * 64 + 16 * rb + cq;
        const double itb = 1.0 / tb;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * j] = kq[j] * itb;
      }
    }
    if (!mm) return;
    const int dst = w1 & 0xFF;
    double2* out = reinterpret_cast<double2*>(TP(dst)) + 128 * rb + lane;        // D: state 16 rb + (l >> 4) + 4 r = k-step 4 rb + r
    if (!ABL(4)) { out[0] = make_double2(acc.x, acc.y); out[64] = make_double2(acc.z, acc.w); }
    const double colsum = column_sum((acc.x + acc.y) + (acc.z + acc.w));
    if (cq == 0) tot[dst * 64 + gl * 4 + rb] = colsum;
    STAMP(3)
  };
  // Product-fused member (SharedProgram::pf_ok; record layout: build_shared_program).  The input S -- c (.) message, stored by
  // the message's producer -- goes from LDS straight into the matrix cores; the result is multiplied by this wave's 16 rows of
  // the destination's constant product and by 1 / total(S) (any positive per-graph scale cancels downstream; this one keeps
  // the magnitudes where they are) and stored with its partial column sums.  The reciprocal and the scaled c rows are formed
  // BETWEEN the quarters' MFMAs: float64 vector operations run on the matrix cores' pipe, so behind an MFMA they cost their
  // own few cycles and in front of the first one a whole wait.
  double2* const stash_g = PF && d.stash && d.msgs && !d.vf_only ? reinterpret_cast<double2*>(d.stash + (size_t)wg * d.n_stash * TILE) : nullptr;
  auto run_pf = [&](const int flags, const int w1, const int w2) {
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int gl = lane & 15, cq = lane >> 4;
    const bool mm = (flags & 1) != 0;
    const bool second_set = (flags & 0x80) != 0;                 // (resolved by the prologue)
    const int S = (w1 >> 8) & 0xFF, cd = w2 & 0xFF;
    // (the host gives every member a real S tile -- an input that is still the uniform vector reads a tile the prologue filled --
    // and every stored product a c tile: each address below is one base register plus immediate offsets)
    const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)S * TILE) + lane;
    const double2* tp = reinterpret_cast<const double2*>(tot + S * 64 + gl * 4);
    const bool noC = cd == 0xFF;                                 // (a slot only the read-out uses: the message itself is stored -- c = {1, 1} from the constants)
    const double2* cs = noC ? dummy + 1 : reinterpret_cast<const double2*>(tiles + (size_t)cd * TILE) + 128 * rb + lane;
    const int cof = noC ? 0 : 64;
    double2 qa0 = src[0], qa1 = src[64], qb0 = src[128], qb1 = src[192], ta = tp[0], tb = tp[1], c0, c1;
    STAMP(6)
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double s = 1.0;
    double k0, k1, k2, k3;
    // (the totals arrive behind the first two quarters and are spent behind quarter 0's MFMAs, the c rows are requested in front of
    // quarter 3's: their latency passes under MFMAs and the live registers stay under the 128 of four waves per SIMD)
#define MLBP_PF_QUARTER(FR, H, V0, V1)                                                         \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H)], V0.x, acc, 0, 0, 0);               \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 1], V0.y, acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 2], V1.x, acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * (H) + 3], V1.y, acc, 0, 0, 0);
#define MLBP_PF_TOTAL                                                                          \
    { const double total = (ta.x + ta.y) + (tb.x + tb.y); bad |= !total_ok(total); s = __builtin_amdgcn_rcp(total); }
#define MLBP_PF_BODY(FR)                                                                       \
    MLBP_PF_QUARTER(FR, 0, qa0, qa1)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    qa0 = src[256]; qa1 = src[320];                                                            \
    MLBP_PF_TOTAL                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    MLBP_PF_QUARTER(FR, 1, qb0, qb1)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    qb0 = src[384]; qb1 = src[448];                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    MLBP_PF_QUARTER(FR, 2, qa0, qa1)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    c0 = cs[0]; c1 = cs[cof];                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    MLBP_PF_QUARTER(FR, 3, qb0, qb1)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    k0 = c0.x * s; k1 = c0.y * s; k2 = c1.x * s; k3 = c1.y * s;
    if (mm && !ABL(1)) {
      if (!second_set) { MLBP_PF_BODY(fr0) } else { MLBP_PF_BODY(fr1) }
    } else {
      MLBP_PF_TOTAL
    }
#undef MLBP_PF_BODY
#undef MLBP_PF_TOTAL
#undef MLBP_PF_QUARTER
    STAMP(7)
    if ((flags & 8) && d.msgs && !(GRAD && d.vf_direct && d.vf_only) && g0 + gl < d.B && !bad) {        // S, normalised, is the last value of a variable->factor slot: to memory
      // (not when only the gradient epilogue wanted it and reads the tiles themselves: vf_direct)
      const double2 v0 = src[128 * rb], v1 = src[128 * rb + 64], t0 = tp[0], t1 = tp[1];      // this wave's rows once more (states 16 rb + cq + 4 r = k-steps 4 rb + r)
      double* out = d.msgs + ((size_t)(g0 + gl) * d.n_msgs + ((w1 >> 16) & 0xFFFF)) * 64 + 16 * rb + cq;
      const double it = 1.0 / ((t0.x + t0.y) + (t1.x + t1.y));
      out[0] = v0.x * it; out[4] = v0.y * it; out[8] = v1.x * it; out[12] = v1.y * it;
    }
    if (!mm) return;
    const int dst = w1 & 0xFF;
    if (flags & 64) {
      const double p0 = acc.x * k0, p1 = acc.y * k1, p2 = acc.z * k2, p3 = acc.w * k3;
      double2* out = reinterpret_cast<double2*>(tiles + (size_t)dst * TILE) + 128 * rb + lane;
      if (!ABL(4)) { out[0] = make_double2(p0, p1); out[64] = make_double2(p2, p3); }
      const double colsum = column_sum((p0 + p1) + (p2 + p3));
      if (cq == 0) tot[dst * 64 + gl * 4 + rb] = colsum;
    }
    if ((flags & 32) && stash_g) {                               // the slot's last update: the raw result for the message write-back (any scale: the epilogue normalises)
      double2* out = stash_g + (size_t)(w2 >> 8) * (TILE / 2) + 128 * rb + lane;
      out[0] = make_double2(acc.x, acc.y); out[64] = make_double2(acc.z, acc.w);
    }
    STAMP(3)
  };
  // Three-source product-fused member (SharedProgram::p3_ok; record layout: build_shared_program).  The input is the product of
  // one or two tiles (sqrt(c) (.) m_a times sqrt(c) (.) m_b), formed a quarter ahead of the matrix instructions that take it; the
  // destination's rows of ITS constant product (or its square root) come from memory, requested first and needed last.  One
  // workgroup per CU: 256 registers a wave, nothing here is squeezed.
  auto run_p3 = [&](const int flags, const int w1, const int w2, const int w3) {
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int gl = lane & 15, cq = lane >> 4;
    const bool mm = (flags & 1) != 0, second_table = (flags & 0x80) != 0, two = (flags & 0x100) != 0;     // (bit 7: resolved by the prologue)
    const bool want = (flags & (4 | 8)) != 0;
    const int S = (w1 >> 8) & 0xFF, S2 = w3 & 0xFF, K = (w3 >> 8) & 0xFF, ck = w2 & 0xFF;
    const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)S * TILE) + lane;
    const double2* src2 = two ? reinterpret_cast<const double2*>(tiles + (size_t)S2 * TILE) + lane : dummy + 1;      // (absent: {1, 1}, stride 0)
    const int st2 = two ? 128 : 0, of2 = two ? 64 : 0;
    const double2* tp = reinterpret_cast<const double2*>(tot + S * 64 + gl * 4);
    const double2* tp2 = two ? reinterpret_cast<const double2*>(tot + S2 * 64 + gl * 4) : dummy + 2;               // (absent: a total of 1)
    double2 c0 = make_double2(1.0, 1.0), c1 = c0;
    if (mm && ck != 0xFF) {
      const double2* cg = psrc + (size_t)ck * (TILE / 2) + 128 * rb + lane;
      c0 = cg[0]; c1 = cg[64];
    }
    const double2 ta = tp[0], tb = tp[1], tc = tp2[0], td = tp2[1];
    double2 q0 = src[0], q1 = src[64], r0 = src2[0], r1 = src2[of2];
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const double b0 = q0.x * r0.x, b1 = q0.y * r0.y, b2 = q1.x * r1.x, b3 = q1.y * r1.y;
      __builtin_amdgcn_sched_barrier(0);
      if (h < 3) {                                               // the next quarter's reads go out ahead of this quarter's MFMAs
        q0 = src[128 * (h + 1)]; q1 = src[128 * (h + 1) + 64];
        r0 = src2[st2 * (h + 1)]; r1 = src2[st2 * (h + 1) + of2];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (want) {
        part += (b0 + b1) + (b2 + b3);
        if ((flags & 4) && rb == h) {                            // read by a later factor update: kept as a tile (raw; its total below)
          double2* o = reinterpret_cast<double2*>(tiles + (size_t)K * TILE) + 128 * rb + lane;
          o[0] = make_double2(b0, b1); o[64] = make_double2(b2, b3);
        }
      }
      if (mm && !ABL(1)) {
#define MLBP_P3_QUARTER(FR)                                                                    \
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * h], b0, acc, 0, 0, 0);             \
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * h + 1], b1, acc, 0, 0, 0);         \
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * h + 2], b2, acc, 0, 0, 0);         \
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(FR[4 * h + 3], b3, acc, 0, 0, 0);
        if constexpr (P3) {
          if (!second_table) { if (!(flags & 2)) { MLBP_P3_QUARTER(fr0) } else { MLBP_P3_QUARTER(fr1) } }
          else               { if (!(flags & 2)) { MLBP_P3_QUARTER(fr2) } else { MLBP_P3_QUARTER(fr3) } }
        }
#undef MLBP_P3_QUARTER
      }
    }
    const double t0 = (ta.x + ta.y) + (tb.x + tb.y), t1 = (tc.x + tc.y) + (td.x + td.y);
    bad |= (int)!total_ok(t0) | (int)!total_ok(t1);
    if (want) {
      const double tprod = column_sum(part);
      bad |= !total_ok(tprod);
      if (flags & 4) {
        if (cq == 0) tot[K * 64 + gl * 4 + rb] = 0.25 * tprod;
      } else if (d.msgs && g0 + gl < d.B && !bad) {              // last value of this variable->factor slot: to memory, normalised
        const double2 v0 = src[128 * rb], v1 = src[128 * rb + 64], u0 = src2[st2 * rb], u1 = src2[st2 * rb + of2];
        double* out = d.msgs + ((size_t)(g0 + gl) * d.n_msgs + ((w1 >> 16) & 0xFFFF)) * 64 + 16 * rb + cq;
        const double it = 1.0 / tprod;
        out[0] = (v0.x * u0.x) * it; out[4] = (v0.y * u0.y) * it; out[8] = (v1.x * u1.x) * it; out[12] = (v1.y * u1.y) * it;
      }
    }
    if (!mm) return;
    const int dst = w1 & 0xFF;
    if (flags & 64) {
      // (any positive per-graph scale cancels downstream; 1 / (the inputs' totals) keeps the magnitudes where they are)
      const double sc = __builtin_amdgcn_rcp(t0) * __builtin_amdgcn_rcp(t1);
      const double p0 = acc.x * (c0.x * sc), p1 = acc.y * (c0.y * sc), p2 = acc.z * (c1.x * sc), p3 = acc.w * (c1.y * sc);
      double2* out = reinterpret_cast<double2*>(tiles + (size_t)dst * TILE) + 128 * rb + lane;
      out[0] = make_double2(p0, p1); out[64] = make_double2(p2, p3);
      const double colsum = column_sum((p0 + p1) + (p2 + p3));
      if (cq == 0) tot[dst * 64 + gl * 4 + rb] = colsum;
    }
    if ((flags & 32) && stash_g) {
      double2* out = stash_g + (size_t)(w2 >> 8) * (TILE / 2) + 128 * rb + lane;
      out[0] = make_double2(acc.x, acc.y); out[64] = make_double2(acc.z, acc.w);
    }
  };
  // the bundles sit in LDS; a bundle's 16 words are read (broadcast) one bundle ahead -- LDS reads return in order, so they
  // cost the tile reads behind them nothing (a scalar load from memory would be waited for with them: one counter)
  const int4* li = reinterpret_cast<const int4*>(limg) + half;  // this half's slot of bundle k: li[2 k]
  int4 nx = li[0];
  for (int k = 0; k < (ABL(2048) ? 0 : d.n_bundles); ++k) {
    int cf = __builtin_amdgcn_readfirstlane(nx.x), c1 = __builtin_amdgcn_readfirstlane(nx.y), c2 = __builtin_amdgcn_readfirstlane(nx.z),
        c3 = __builtin_amdgcn_readfirstlane(nx.w);
    const bool chain = (cf & CHAIN) != 0;
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {                          // (one copy of the member code; the second round only behind CHAIN)
      if ((cf & 0x7F) != 0) {
        if constexpr (P3) run_p3(cf, c1, c2, c3);
        else if constexpr (PF) run_pf(cf, c1, c2);
        else run(cf, c1, c2, c3);
      }
      if (rep == 1 || !chain) break;
      const int4 o = reinterpret_cast<const int4*>(limg)[2 * k + 1 - half];      // the member parked in the other half's slot
      const int ox = __builtin_amdgcn_readfirstlane(o.x);
      cf = (int)((unsigned)ox >> 24) | (ox & 0x00FFFF00);
      c1 = __builtin_amdgcn_readfirstlane(o.y); c2 = __builtin_amdgcn_readfirstlane(o.z); c3 = __builtin_amdgcn_readfirstlane(o.w);
    }
    // the next bundle's words are requested here, behind the member (its registers are free again) and in front of the
    // barrier (they arrive while the workgroup meets)
    __builtin_amdgcn_sched_barrier(0);
    nx = li[2 * k + 2];                                          // the image is padded by one bundle
    if constexpr (PF) {
      // LDS traffic only: the stores to memory a member may have issued (stash, last variable->factor values) are read after the
      // loop, behind a full __syncthreads -- the fence of one here would wait for them in every bundle
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      __syncthreads();
    }
    STAMP(4)
  }

  // ---- epilogue: marginals, message write-back, verdicts ----
  // (every lane-derived value is formed again from a laundered thread id: what the epilogues need would otherwise be computed in
  // front of the loop and held in registers across it -- registers the fragments need)
  auto tail = [&]() {
  int t_again = threadIdx.x;
  asm volatile("" : "+v"(t_again));
  const int t = t_again, lane = t & 63, gl = lane & 15, cq = lane >> 4, gi = g0 + gl;
  const bool gvalid = gi < d.B;
  const int gc = gvalid ? gi : d.B - 1;
  // a bad total met only here (the last update's result) must reach the verdict of every wave
  if constexpr (P3) {
    // three-source form: a tile holds C (.) message with C = c or sqrt(c) (kind3: 1 / 2; 0: the message itself), C in memory.  With
    // n_p updated tiles of that kind the marginal c (.) prod m is  C^(e - n_p) (.) prod tiles  (e = 1 / 2: C = c / sqrt(c)); a negative
    // power is taken out of the FIRST tile (no underflow of the product), where C is 0 so is the marginal.
    if (stash_g) __syncthreads();
    const const_i32p rd = as_const(d.readout);
    double* meet = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + d.lds_bytes) - 64 * d.n_vars;      // [variable][quarter][graph] partial sums
    if (d.marginals && !ABL(512)) {
      constexpr int JW = 4;                                      // (4 n_vars <= 32 jobs: n_vars <= 8, shared_plan)
      const int n_jobs = 4 * d.n_vars;
      double m[JW][4];
#pragma unroll
      for (int q = 0; q < JW; ++q) {
        const int j = wave + q * (SWG / 64);
        if (j >= n_jobs) continue;
        const int v = j >> 2, h = j & 3;
        const int at = rd[v];
        const int base = rd[at], n = rd[at + 1];
        const int ck = img[d.off_map3 + base] & 0xFF;            // (every variable of the read-out has a constant product: shared_plan)
        const double2* cg = psrc + (size_t)ck * (TILE / 2) + 128 * h + lane;
        const double2 c0 = cg[0], c1 = cg[64];
        const double cc[4] = {c0.x, c0.y, c1.x, c1.y};
        int n_p = 0;
        for (int k = 0; k < n; ++k) {
          const int kd = img[d.off_kind3 + img[d.off_map3 + rd[at + 2 + k]]];
          if ((kd & 0x100) && (kd & 0xFF)) ++n_p;
        }
        const int pw = (((d.sqrt_mask >> ck) & 1) ? 2 : 1) - n_p;
        m[q][0] = m[q][1] = m[q][2] = m[q][3] = 1.0;
        bool first = true;
        for (int k = 0; k < n; ++k) {
          const int tl = img[d.off_map3 + rd[at + 2 + k]];
          const int kd = img[d.off_kind3 + tl];
          if (!(kd & 0x100)) continue;                           // never updated: still uniform, cancels in the normalisation
          const double2* tp = reinterpret_cast<const double2*>(tot + tl * 64 + gl * 4);
          const double2 ta = tp[0], tb = tp[1];
          const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)tl * TILE) + 128 * h + lane;
          const double2 x0 = src[0], x1 = src[64];
          const double total = (ta.x + ta.y) + (tb.x + tb.y);
          bad |= !total_ok(total);
          const double inv = __builtin_amdgcn_rcp(total);        // (any positive scale: it cancels below)
          m[q][0] *= x0.x * inv; m[q][1] *= x0.y * inv; m[q][2] *= x1.x * inv; m[q][3] *= x1.y * inv;
          if ((kd & 0xFF) && first && pw < 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                        // m / C by the hardware reciprocal and one Newton step (3e-17 relative)
              double rc = __builtin_amdgcn_rcp(cc[r]);
              rc = __builtin_fma(__builtin_fma(-cc[r], rc, 1.0), rc, rc);
              m[q][r] = cc[r] > 1e-290 ? m[q][r] * rc : 0.0;
            }
          }
          if (kd & 0xFF) first = false;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (pw >= 1) m[q][r] *= cc[r];
          if (pw >= 2) m[q][r] *= cc[r];
        }
        const double part = column_sum((m[q][0] + m[q][1]) + (m[q][2] + m[q][3]));
        if (cq == 0) meet[v * 64 + h * G + gl] = part;
      }
      __syncthreads();
      // the normalised values leave as whole 512-byte rows: staged in the variable's FIRST message tile (spent: every job holds its
      // values), state s of graph gl at s ^ gl
#pragma unroll
      for (int q = 0; q < JW; ++q) {
        const int j = wave + q * (SWG / 64);
        if (j >= n_jobs) continue;
        const int v = j >> 2, h = j & 3;
        const double* rp = meet + v * 64 + gl;
        const double tm = (rp[0] + rp[G]) + (rp[2 * G] + rp[3 * G]);
        bad |= !total_ok(tm);
        const double itm = 1.0 / tm;
        double* o = tiles + (size_t)img[d.off_map3 + rd[rd[v] + 2]] * TILE + gl * 64;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(16 * h + cq + 4 * r) ^ gl] = m[q][r] * itm;
      }
      __syncthreads();
    }
    // write-back: stored variable->factor messages from their tiles (what the gradient reads; NOT the tiles the marginals were staged
    // in: those are factor->variable tiles), factor->variable messages from the raw results the members stashed
    if (d.msgs)
      for (int i = wave; i < d.n_back; i += SWG / 64) {
        const int tl = img[d.off_back3 + 2 * i], sl = img[d.off_back3 + 2 * i + 1];
        const bool is_vf = (sl & 0x40000000) != 0;
        if (is_vf) {
          const double* tp = tot + tl * 64 + gl * 4;
          const double total = (tp[0] + tp[1]) + (tp[2] + tp[3]);
          bad |= !total_ok(total);
          if (gvalid && !bad) {
            const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)tl * TILE) + lane;
            double* out = d.msgs + ((size_t)gc * d.n_msgs + (sl & 0x3FFFFFFF)) * 64 + cq;
            const double inv = 1.0 / total;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) { const double2 v = src[64 * sp]; out[8 * sp] = v.x * inv; out[8 * sp + 4] = v.y * inv; }
          }
          continue;
        }
        const int si = img[d.off_stash + tl];
        if (si < 0 || !stash_g || d.vf_only) continue;
        const double2* src = stash_g + (size_t)si * (TILE / 2) + lane;
        double2 x[8];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) x[sp] = src[64 * sp];
        double part = 0.0;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) part += x[sp].x + x[sp].y;
        const double total = column_sum(part);
        bad |= !total_ok(total);
        if (gvalid && !bad) {
          double* out = d.msgs + ((size_t)gc * d.n_msgs + (sl & 0x3FFFFFFF)) * 64 + cq;
          const double inv = 1.0 / total;
#pragma unroll
          for (int sp = 0; sp < 8; ++sp) { out[8 * sp] = x[sp].x * inv; out[8 * sp + 4] = x[sp].y * inv; }
        }
      }
  } else if constexpr (PF) {
    // product-fused: a tile holds c (.) message (or, when no update reads it, the message).  With the variable's constant
    // product c and P_a = c (.) m_a, P_b = c (.) m_b the marginal c (.) m_a (.) m_b is P_a (.) (P_b / c) -- where c is 0 so is the
    // marginal --: everything from LDS.  (c (.) m keeps nothing of m where c is 0, so the message WRITE-BACK reads the raw
    // results the members stashed in memory: the workgroup's own stores, visible behind the fence of a full barrier.)
    if (stash_g) __syncthreads();
    const const_i32p rd = as_const(d.readout);
    if (d.marginals && !ABL(512)) {
      // jobs (variable v, quarter h of its states: k-steps 4 h .. 4 h + 3) over the eight waves; a job's values stay in registers
      // while the quarters' column sums meet in LDS (the totals row of the variable's c tile is free: 4 x 16 doubles)
      constexpr int JW = 4;                                      // jobs a wave may hold (4 n_vars <= 32 jobs: n_vars <= 8; beyond, the rest in a second pass)
      const int n_jobs = 4 * d.n_vars;
      for (int j0 = 0; j0 < n_jobs; j0 += JW * (SWG / 64)) {
        double m[JW][4];
#pragma unroll
        for (int q = 0; q < JW; ++q) {
          const int j = j0 + wave + q * (SWG / 64);
          if (j >= n_jobs) continue;
          const int v = j >> 2, h = j & 3;
          const int at = rd[v];
          const int base = rd[at], n = rd[at + 1];
          int n_p = 0, n_t = 0;
          for (int k = 0; k < n; ++k) {
            const int tl = rd[at + 2 + k];
            if (img[d.off_stash + tl] >= 0) { ++n_t; n_p += img[d.off_stash + d.n_live + tl]; }
          }
          double2 c0 = make_double2(uniform, uniform), c1 = c0;
          if (base >= 0) {
            const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)base * TILE) + 128 * h + lane;
            c0 = src[0]; c1 = src[64];
          }
          const bool from_c = n_t == 0 || n_p == 0;              // c itself (or uniform) times the messages
          m[q][0] = from_c ? c0.x : 1.0; m[q][1] = from_c ? c0.y : 1.0; m[q][2] = from_c ? c1.x : 1.0; m[q][3] = from_c ? c1.y : 1.0;
          int p_left = n_p;
          for (int k = 0; k < n; ++k) {
            const int tl = rd[at + 2 + k];
            if (img[d.off_stash + tl] < 0) continue;             // never updated: still uniform, cancels in the normalisation
            const bool is_p = img[d.off_stash + d.n_live + tl] != 0;
            const double2* tp = reinterpret_cast<const double2*>(tot + tl * 64 + gl * 4);
            const double2 ta = tp[0], tb = tp[1];
            const double2* src = reinterpret_cast<const double2*>(tiles + (size_t)tl * TILE) + 128 * h + lane;
            const double2 x0 = src[0], x1 = src[64];
            const double total = (ta.x + ta.y) + (tb.x + tb.y);
            bad |= !total_ok(total);
            const double inv = __builtin_amdgcn_rcp(total);      // (any positive scale: it cancels below)
            m[q][0] *= x0.x * inv; m[q][1] *= x0.y * inv; m[q][2] *= x1.x * inv; m[q][3] *= x1.y * inv;
            if (is_p && --p_left > 0) {                          // another c (.) message follows: take this one's c out first (no underflow of the pair)
              const double cc[4] = {c0.x, c0.y, c1.x, c1.y};
#pragma unroll
              for (int r = 0; r < 4; ++r) {                      // m / c by the hardware reciprocal and one Newton step (3e-17 relative)
                double rc = __builtin_amdgcn_rcp(cc[r]);
                rc = __builtin_fma(__builtin_fma(-cc[r], rc, 1.0), rc, rc);
                m[q][r] = cc[r] > 1e-290 ? m[q][r] * rc : 0.0;
              }
            }
          }
          const double part = column_sum((m[q][0] + m[q][1]) + (m[q][2] + m[q][3]));
          // the meeting place: the totals row of the variable's c tile (nothing reads a c tile's totals behind the main loop; the
          // launcher takes this form only when every variable of the read-out has a constant product)
          if (cq == 0) tot[base * 64 + h * G + gl] = part;
        }
        __syncthreads();
        // the normalised values go through LDS once more (the tiles are spent: every job has read its own), so that they leave as
        // whole 512-byte rows, one per (graph, variable): written from this layout they were 32-byte pieces of sixteen rows per
        // store, and the launch spent 6 us draining 12.6 MB of them
        // (variable v's sixteen rows in ITS constant product's tile -- 16 x 64 doubles, spent: every job has read its own --, so
        // that the message tiles stay what the gradient epilogue reads)
#pragma unroll
        for (int q = 0; q < JW; ++q) {
          const int j = j0 + wave + q * (SWG / 64);
          if (j >= n_jobs) continue;
          const int v = j >> 2, h = j & 3;
          const double* rp = tot + rd[rd[v]] * 64 + gl;
          const double tm = (rp[0] + rp[G]) + (rp[2 * G] + rp[3 * G]);
          bad |= !total_ok(tm);
          const double itm = 1.0 / tm;
          double* o = tiles + (size_t)rd[rd[v]] * TILE + gl * 64;  // (state s of graph gl at s ^ gl: the sixteen graphs of a store on sixteen banks)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[(16 * h + cq + 4 * r) ^ gl] = m[q][r] * itm;
        }
        __syncthreads();
        if (j0 + JW * (SWG / 64) < n_jobs) __syncthreads();
      }
    }
    if (d.msgs && !d.vf_only && stash_g)
      for (int i = wave; i < d.n_back; i += SWG / 64) {
        const int tl = img[d.off_back + 2 * i], sl = img[d.off_back + 2 * i + 1];
        const int si = img[d.off_stash + tl];
        if (si < 0 || (sl & 0x40000000)) continue;               // (this form keeps no variable->factor message as a tile)
        const double2* src = stash_g + (size_t)si * (TILE / 2) + lane;
        double2 x[8];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) x[sp] = src[64 * sp];
        double part = 0.0;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) part += x[sp].x + x[sp].y;
        const double total = column_sum(part);
        bad |= !total_ok(total);
        if (gvalid && !bad) {
          double* out = d.msgs + ((size_t)gc * d.n_msgs + (sl & 0x3FFFFFFF)) * 64 + cq;
          const double inv = 1.0 / total;
#pragma unroll
          for (int sp = 0; sp < 8; ++sp) { out[8 * sp] = x[sp].x * inv; out[8 * sp + 4] = x[sp].y * inv; }
        }
      }
  } else {
  if (d.marginals) {
    const const_i32p rd = as_const(d.readout);
    for (int v = wave; v < d.n_vars; v += SWG / 64) {
      const int at = rd[v];
      const int base = rd[at], n = rd[at + 1];
      double m[16];
      if (base >= 0) {
        const double2* src = reinterpret_cast<const double2*>(TP(base)) + lane;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) { const double2 v = src[64 * sp]; m[2 * sp] = v.x; m[2 * sp + 1] = v.y; }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] = uniform;
      }
      for (int q = 0; q < n; ++q) {
        const int tl = rd[at + 2 + q];
        const double* tp = tot + tl * 64 + gl * 4;
        const double total = (tp[0] + tp[1]) + (tp[2] + tp[3]);
        bad |= !total_ok(total);
        const double inv = 1.0 / total;
        const double2* src = reinterpret_cast<const double2*>(TP(tl)) + lane;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) { const double2 v = src[64 * sp]; m[2 * sp] *= v.x * inv; m[2 * sp + 1] *= v.y * inv; }
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
        for (int s = 0; s < 16; ++s) __builtin_nontemporal_store(m[s] * itm, &out[4 * s]);     // (written once, read by another launch: past L2)
      }
    }
  }
  for (int i = wave; i < d.n_back; i += SWG / 64) {
    const int tl = img[d.off_back + 2 * i], sl = img[d.off_back + 2 * i + 1];
    const int slot = sl & 0x3FFFFFFF;
    const bool is_vf = (sl & 0x40000000) != 0;
    const double* tp = tot + tl * 64 + gl * 4;
    const double total = (tp[0] + tp[1]) + (tp[2] + tp[3]);
    bad |= !total_ok(total);
    if (d.msgs && (!d.vf_only || is_vf) && gvalid && !bad) {
      const double2* src = reinterpret_cast<const double2*>(TP(tl)) + lane;
      double* out = d.msgs + ((size_t)gc * d.n_msgs + slot) * 64 + cq;
      const double inv = 1.0 / total;
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) { const double2 v = src[64 * sp]; out[8 * sp] = v.x * inv; out[8 * sp + 4] = v.y * inv; }
    }
  }
  }   // !PF
  if (bad && gvalid) d.bail[gi] = 2;                             // any wave that saw it says so (idempotent)
  STAMP(5)
  if constexpr (PF) {
    // the gradient epilogue in front of the marginals' way out: it reads the message tiles (intact: the marginals are staged in
    // the constant products' tiles) and nothing it waits for is behind 12.6 MB of stores
    if constexpr (GRAD) {
      if (d.gr.enabled && d.vf_direct) shared_gradient_epilogue<true, 4>(d, wg);
    }
    if (d.marginals && !ABL(512)) {
      const const_i32p rd = as_const(d.readout);
      const unsigned long long flagged = __ballot(bad);
      for (int row = wave; row < d.n_vars * G; row += SWG / 64) {
        const int g = row & (G - 1), v = row >> 4;
        if (g0 + g < d.B && !((flagged >> g) & 0x0001000100010001ull))
          __builtin_nontemporal_store(tiles[(size_t)(P3 ? img[d.off_map3 + rd[rd[v] + 2]] : rd[rd[v]]) * TILE + g * 64 + (lane ^ g)],
                                      &d.marginals[((size_t)(g0 + g) * d.n_vars + v) * 64 + lane]);
      }
    }
  }
  if (!GRAD || !d.gr.enabled) {                                  // (GRAD is a template parameter: the sweeps-only instances do not carry the epilogue's registers)
    STAMP_FLUSH
    return;
  }
  // ---- gradient epilogue (LBP.py:301-320 over beliefs of LBP.py:528-574): per pairwise factor and feature k
  //      E_k = m_c^T (T (.) phi_k) m_r and Z = m_c^T T m_r on the matrix cores, from the factor's two stored variable->factor
  //      messages -- read back from the message buffer this workgroup has just written (its own stores, behind a barrier) into
  //      the tiles the sweeps no longer need.  Pairwise terms: the operations of gradient_shared_pairs_kernel in its order.
  //      Every wave of the workgroup waits for every load here, so the gathers are issued a phase AHEAD of their use: a
  //      pass's labels and first fragment before its messages, its label features behind them.  (The unary factors'
  //      terms do not depend on the sweeps: shared_prepare_kernel has written them, this adds to them.) ----
  STAMP_FLUSH
  if (!(PF && d.vf_direct)) shared_gradient_epilogue<false, P3 ? 6 : 4>(d, wg);      // (three-source form: 256 registers a wave, six factors -- K4 -- in one pass)
  };
  tail();
}

typedef const SharedDev __attribute__((address_space(4))) SharedDevConst;
template <int NTAB, bool SPILL, bool WIDE, bool MULTI, bool GRAD, bool PF = false, bool P3 = false>
__global__ __launch_bounds__(SWG, P3 ? 2 : 4) void sweep_x64_shared_kernel(SharedDev d, const SharedDev* gtab, const int32_t* gstart, int n_groups) {
  if (MULTI) {
    const int lo = find_group(gstart, n_groups, blockIdx.x);
    SharedDevConst& dg = *(SharedDevConst*)(uintptr_t)(gtab + lo);
    sweep_x64_shared_body<NTAB, SPILL, WIDE, GRAD, PF, P3>(dg, (int)blockIdx.x - as_const(gstart)[lo]);
  } else {
    // the description is the first kernel argument: read where it is used, through the scalar cache, out of the kernel
    // argument segment -- preloaded as a by-value struct its 80 words crowd the scalar registers of the main loop
    (void)d;
    SharedDevConst& dk = *(SharedDevConst*)__builtin_amdgcn_kernarg_segment_ptr();
    sweep_x64_shared_body<NTAB, SPILL, WIDE, GRAD, PF, P3>(dk, (int)blockIdx.x);
  }
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

typedef void (*sweep_fn)(SharedDev, const SharedDev*, const int32_t*, int);

// the gradient-epilogue instances that exist: all-resident two-source, and spilling + wide
template <int T, bool S, bool W, bool M>
sweep_fn grad_instance() {
  if constexpr (S == W) return (sweep_fn)sweep_x64_shared_kernel<T, S, W, M, true>;
  else return nullptr;
}

// the instance for (two tables?, spilled tiles?, more than two sources?, groups?); raises its dynamic LDS limit once
int pick_sweep_kernel(bool two, bool spill, bool wide, bool multi, bool grad, bool pf, size_t lds, sweep_fn* out, bool p3 = false) {
  sweep_fn k = nullptr;
  if (spill || wide) pf = false;
  if (p3) {                                                      // three-source product-fused form
#ifdef MLBP_STAMPS
    return fail(MLBP_EUNSUPPORTED, "stamps build: no three-source instance");
#else
#define MLBP_P3(T, M) (grad ? (sweep_fn)sweep_x64_shared_kernel<T, false, false, M, true, true, true> : (sweep_fn)sweep_x64_shared_kernel<T, false, false, M, false, true, true>)
    k = two ? (multi ? MLBP_P3(2, true) : MLBP_P3(2, false)) : (multi ? MLBP_P3(1, true) : MLBP_P3(1, false));
#undef MLBP_P3
#endif
  } else {
#ifdef MLBP_STAMPS      // the diagnostic build instantiates the all-resident two-source kernels only
  if (spill || wide || multi) return fail(MLBP_EUNSUPPORTED, "stamps build: no spilling / wide / grouped instance");
  if (pf)
    k = two ? (grad ? sweep_x64_shared_kernel<2, false, false, false, true, true> : sweep_x64_shared_kernel<2, false, false, false, false, true>)
            : (grad ? sweep_x64_shared_kernel<1, false, false, false, true, true> : sweep_x64_shared_kernel<1, false, false, false, false, true>);
  else
  k = two ? (grad ? sweep_x64_shared_kernel<2, false, false, false, true> : sweep_x64_shared_kernel<2, false, false, false, false>)
          : (grad ? sweep_x64_shared_kernel<1, false, false, false, true> : sweep_x64_shared_kernel<1, false, false, false, false>);
#else
  if (pf) {
#define MLBP_PK(T, M) (grad ? (sweep_fn)sweep_x64_shared_kernel<T, false, false, M, true, true> : (sweep_fn)sweep_x64_shared_kernel<T, false, false, M, false, true>)
    k = two ? (multi ? MLBP_PK(2, true) : MLBP_PK(2, false)) : (multi ? MLBP_PK(1, true) : MLBP_PK(1, false));
#undef MLBP_PK
  } else {
  // (the gradient epilogue comes in two forms: every tile in LDS and two-source updates -- K2, K3 --, or spilled tiles AND
  // wide updates -- K4 and larger cliques, train_mp.py:272-282: a variable with three pairwise factors has three neighbours, and
  // four variables' tiles do not fit.  A launch of mixed groups takes the second.)
  if (grad && spill != wide) { spill = true; wide = true; }
#define MLBP_SK(T, S, W, M) (grad ? grad_instance<T, S, W, M>() : (sweep_fn)sweep_x64_shared_kernel<T, S, W, M, false>)
#define MLBP_SK_M(T, S, W) (multi ? MLBP_SK(T, S, W, true) : MLBP_SK(T, S, W, false))
#define MLBP_SK_W(T, S) (wide ? MLBP_SK_M(T, S, true) : MLBP_SK_M(T, S, false))
#define MLBP_SK_S(T) (spill ? MLBP_SK_W(T, true) : MLBP_SK_W(T, false))
  k = two ? MLBP_SK_S(2) : MLBP_SK_S(1);
#undef MLBP_SK_S
#undef MLBP_SK_W
#undef MLBP_SK_M
#undef MLBP_SK
  }
#endif
  }
  static std::vector<std::pair<const void*, size_t>> granted;
  {
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    bool have = false;
    for (auto& g : granted) have |= g.first == (const void*)k && g.second >= lds;
    if (!have) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(MLBP_EHIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      granted.push_back({(const void*)k, lds});
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, SWG, lds) == hipSuccess)
        fail(MLBP_OK, "shared-table kernel <%d%s%s%s%s%s>: %zu bytes of LDS per workgroup, %d workgroups per CU", two ? 2 : 1,
             spill ? ", spilling" : "", wide ? ", wide" : "", multi ? ", groups" : "", grad ? ", gradient" : "", p3 ? ", product-fused (three sources)" : (pf ? ", product-fused" : ""), lds, per_cu);
    }
  }
  *out = k;
  return MLBP_OK;
}

// resident tiles: as many as fit HALF the CU's LDS, so that two workgroups share a CU; *lds: the workgroup's LDS bytes
int resident_tiles(const SharedProgram& sp, size_t* lds) {
  const size_t fixed = (size_t)sp.n_live * 64 * sizeof(double) + 64 + 8 * (size_t)(sp.n_bundles + 1) * sizeof(int32_t) + 64;
  int n_res = sp.n_live;
  while (n_res > 0 && fixed + (size_t)n_res * 64 * 16 * sizeof(double) > 80 * 1024) --n_res;
  if (lds) *lds = fixed + (size_t)n_res * 64 * 16 * sizeof(double);
  return n_res;
}

}  // namespace

// True when launch_shared_sweep / launch_shared_groups run the call's gradient (a->gradient) as the sweep kernel's epilogue:
// shared tables with their planar feature tensors, the unary part by gather, and an exact kernel that can produce the
// gradient of the graphs it redoes.  A function of the arguments alone: the dispatcher asks it again after the launch.
bool shared_gradient_fused(const mlbp_program* prog, const mlbp_sweep_args* a) {
  const mlbp_gradient_args* ga = a->gradient;
  if (!ga || a->X != 64 || !a->normalize_messages) return false;
  if (ga->B != a->B || ga->X != a->X || ga->P != prog->P || ga->U != prog->U || ga->n_msgs != prog->n_msgs || ga->msgs != a->msgs) return false;
  if (!(ga->F_ee == 3 && ga->F_ed == 6) || (ga->flags & MLBP_GRADIENT_APPROX_BELIEFS)) return false;
  // graphs the sweep kernel flags get their gradient from the exact kernel's own epilogue (at most three pairwise factors), or
  // -- larger cliques -- from the per-graph gradient kernel run on the flagged graphs only; both read the transposed tensors
  if (!(ga->phi_en_en_t && ga->phi_en_en_w1_t && ga->phi_en_de_t)) return false;
  if (prog->P <= 3 && !exact_kernel_fuses_gradient(prog, a)) return false;
  return (ga->flags & MLBP_GRADIENT_SHARED_PAIR_TABLES) && prog->P >= 1 && prog->P <= 16 && a->n_pair_tables <= FRAG_TABLES && ga->phi_en_en_p &&
         ga->phi_en_en_w1_p && ga->phi_en_en && ga->phi_en_en_w1 && (prog->U == 0 || (ga->unary_expect && ga->phi_en_de)) && a->msgs &&
         prog->shared.ok && prog->shared.n_live >= 3 && resident_tiles(prog->shared, nullptr) >= 3;
}

namespace {

// What a shared-table sweep of (prog, a) needs: the two device descriptions, the LDS size, the grid sizes.  *ok false: the
// kernel does not apply (mlbp_last_error says why).  Allocates the program's scratch on first use.
struct SharedPlan { SharedDev d; PrepareDev q; size_t lds; int n_wg, n_prep_blocks; bool wide, spill, pf, p3; };
int shared_plan(const mlbp_program* prog, const mlbp_sweep_args* a, bool* ok, SharedPlan* out, bool allow_p3 = true) {
  *ok = false;
  memset(out, 0, sizeof(*out));
  const SharedProgram& sp = prog->shared;
  if (!(a->flags & MLBP_SWEEP_SHARED_PAIR_TABLES)) return MLBP_OK;
  if (a->X != 64 || !a->normalize_messages || !a->init_messages)
    return fail(MLBP_OK, "shared-table kernel not used: needs X = 64, normalize_messages and init_messages");
  if (!sp.ok || !prog->d_simage) return fail(MLBP_OK, "shared-table kernel not used: %s", sp.why);
  if (a->marginals && !prog->d_sreadout)
    return fail(MLBP_OK, "shared-table kernel not used: a variable's constant messages match no folded product");
  // resident tiles: as many as fit HALF the CU's LDS, so that two workgroups share a CU (constant products and
  // factor->variable messages come first in the numbering); the rest spill to global memory.  Measured on K4 user
  // graphs (21 tiles): 8 resident + 13 spilled with two workgroups per CU 0.216 ms, 16 resident + 5 spilled with
  // one 0.266 ms.
  size_t lds = 0;
  int n_res = resident_tiles(sp, &lds);
  const int n_cprod = (int)sp.cprods.size();
  // the three-source product-fused form (K4 cliques): every message tile and every stored variable->factor message in LDS, the
  // constant products in memory, one workgroup per CU
  const size_t lds3 = (size_t)sp.n_lds * (TILE + 64) * sizeof(double) + 64 + 8 * (size_t)(sp.n_bundles + 1) * sizeof(int32_t) + 64 +
                      (size_t)prog->n_vars * 64 * sizeof(double);
  const bool p3 = allow_p3 && sp.p3_ok && lds3 <= 160 * 1024 && n_cprod >= 1 && n_cprod <= 8 && prog->n_vars <= 8 &&
                  (!a->marginals || (prog->sreadout_all_based && prog->sreadout_all_tiled));
  if (p3) { n_res = sp.n_lds; lds = lds3; }
  // (K5 / K6 cliques -- 27 and 41 spilled tiles -- still run 4-5 x faster here than on the per-graph kernels, which re-read the
  // shared tables per graph: 0.41 against 1.82 ms and 0.79 against 3.74 ms per trainer step of 8192 instances; the limit used to
  // be 16 spilled tiles, a figure from the time the comparison was with per-graph TABLES)
  constexpr int MAX_SPILLED_TILES = 96;
  if (n_res < 1 || (!p3 && sp.n_live - n_res > MAX_SPILLED_TILES) || lds > 160 * 1024)
    return fail(MLBP_OK, "shared-table kernel not used: %d live message tiles, %d fit LDS", sp.n_live, n_res);
  if (n_cprod < 1 || n_cprod > 8) return fail(MLBP_OK, "shared-table kernel not used: %d constant products (1..8)", n_cprod);
  mlbp_program* mp = const_cast<mlbp_program*>(prog);
  const int n_groups = (a->B + G - 1) / G;
  // the product-fused form (diagnostic switch: MLBP_SHARED_NO_PF in the environment keeps the general form)
  static const bool no_pf = getenv("MLBP_SHARED_NO_PF") != nullptr;
  const bool pf = sp.pf_ok && !no_pf && n_res == sp.n_live && sp.max_sources <= 2 && (!a->marginals || prog->sreadout_all_based) && prog->n_vars <= 8;     // (the read-out stages 8 variables' rows in the spent tiles)
  const size_t spill_doubles = (pf || p3) ? (size_t)n_groups * sp.n_stash * TILE : (size_t)n_groups * (sp.n_live - n_res) * TILE;
  // (first use at this size allocates -- a stream-capturing caller warms up or reserves first; a block that is outgrown stays
  // alive with the program: program_grow)
  if (spill_doubles > 0)
    if (int e = program_grow(mp, reinterpret_cast<void**>(&mp->d_spill), &mp->spill_cap, spill_doubles * sizeof(double))) return e;
  const size_t ptile_doubles = (size_t)n_groups * n_cprod * TILE;
  if (int e = program_grow(mp, reinterpret_cast<void**>(&mp->d_ptiles), &mp->ptiles_cap, ptile_doubles * sizeof(double))) return e;
  if (int e = program_grow(mp, reinterpret_cast<void**>(&mp->d_header), &mp->header_cap, (size_t)n_groups * HDR * sizeof(int32_t))) return e;
  if (mp->bail_cap < a->B)
    if (int e = mlbp_program_reserve(mp, a->B)) return e;
  SharedDev& d = out->d;
  d.pair_tables = a->pair_tables; d.pair_tab = a->pair_tab; d.unary_tables = a->unary_tables; d.unary_tab = a->unary_tab;
  d.msgs = ((a->flags & MLBP_SWEEP_NO_MESSAGE_WRITEBACK) && !a->gradient) ? nullptr : a->msgs;
  d.vf_only = ((a->flags & MLBP_SWEEP_NO_MESSAGE_WRITEBACK) && a->gradient) ? 1 : 0;
  d.marginals = a->marginals; d.status = prog->d_status; d.bail = mp->d_bail;
  d.image = prog->d_simage; d.readout = prog->d_sreadout;
  d.B = a->B; d.n_msgs = prog->n_msgs; d.P = prog->P; d.U = prog->U;
  d.n_pair_tables = a->n_pair_tables; d.n_unary_tables = a->n_unary_tables; d.n_vars = prog->n_vars;
  d.n_bundles = sp.n_bundles; d.n_live = p3 ? sp.n_lds : sp.n_live; d.n_cprod = n_cprod; d.n_back = sp.n_back;
  d.n_fill = sp.n_fill; d.n_init = sp.n_init;
  d.off_back = sp.off_back; d.off_fill = sp.off_fill; d.off_init = sp.off_init; d.off_ptile = sp.off_ptile;
  d.ptiles = mp->d_ptiles;
  d.n_res = n_res; d.spill = (!p3 && n_res < sp.n_live) ? mp->d_spill : nullptr;
  d.off_map3 = sp.off_map3; d.off_kind3 = sp.off_kind3; d.off_back3 = sp.off_back3; d.sqrt_mask = p3 ? sp.sqrt_mask : 0;
  d.off_written = sp.off_written;
  d.off_pfb = sp.off_pfb; d.off_stash = sp.off_stash; d.off_pinit = sp.off_pinit; d.n_pinit = sp.n_pinit; d.n_stash = sp.n_stash;
  d.stash = (pf || p3) ? mp->d_spill : nullptr;
  d.header = mp->d_header;
  d.off_vftile = sp.off_vftile;
  d.vf_direct = 0;                                 // (set below when the gradient is this launch's epilogue)
  // the product-fused form takes the whole half of the CU's LDS: the spare bytes behind the totals are the gradient epilogue's
  if (pf) lds = std::max(lds, (size_t)79 * 1024 + 512);
  d.lds_bytes = (int32_t)lds;
  out->pf = pf; out->p3 = p3;
  d.tfrag = nullptr;
  if (a->n_pair_tables <= FRAG_TABLES) {
    if (!mp->d_tfrag) {                            // first use (a stream-capturing caller warms up or reserves first)
      if (hipMalloc(&mp->d_tfrag, sizeof(double) * FRAG_TABLES * 2 * 4096) != hipSuccess)
        return fail(MLBP_EHIP, "fragment scratch allocation failed");
    }
    d.tfrag = mp->d_tfrag;
  }
  // one launch in front of the sweeps: constant products as tiles, the per-graph flags (cleared or raised), table fragments
  PrepareDev& q = out->q;
  q.pair_tables = a->pair_tables; q.tfrag = d.tfrag ? mp->d_tfrag : nullptr; q.n_frag_tables = d.tfrag ? a->n_pair_tables : 0;
  q.unary_tables = a->unary_tables; q.unary_tab = a->unary_tab; q.ent = prog->d_simage + sp.off_ent;
  q.ptiles = mp->d_ptiles; q.bail = mp->d_bail; q.status = prog->d_status;
  q.B = a->B; q.U = prog->U; q.n_unary_tables = a->n_unary_tables; q.E = sp.n_cpw / 4; q.n_cprod = n_cprod; q.n_groups = n_groups;
  q.pair_tab = a->pair_tab; q.image = prog->d_simage; q.header = mp->d_header; q.P = prog->P; q.n_pair_tables = a->n_pair_tables; q.n_bundles = sp.n_bundles;
  q.sqrt_mask = p3 ? sp.sqrt_mask : 0;
  // the gradient as the sweep kernel's epilogue (the prepare launch also writes its weighted table fragments)
  if (shared_gradient_fused(prog, a)) {
    const mlbp_gradient_args* ga = a->gradient;
    const size_t need = (size_t)a->n_pair_tables * 8 * 4096 + 8;      // + the eight plane flags (as doubles' worth of bytes)
    if (int e = program_grow(mp, reinterpret_cast<void**>(&mp->d_wfrag), &mp->wfrag_cap, need * sizeof(double))) return e;
    q.n_wfrag_tables = a->n_pair_tables; q.wfrag = mp->d_wfrag; q.phi_p0 = ga->phi_en_en_p; q.phi_p1 = ga->phi_en_en_w1_p;
    q.plane_flags = reinterpret_cast<int32_t*>(mp->d_wfrag + (size_t)a->n_pair_tables * 8 * 4096);
    SharedGradDev& gr = d.gr;
    gr.c_slot = ga->pair_c_slot; gr.r_slot = ga->pair_r_slot; gr.pair_phi = ga->pair_phi; gr.pair_label = ga->pair_label;
    gr.phi[0] = ga->phi_en_en; gr.phi[1] = ga->phi_en_en_w1; gr.wfrag = mp->d_wfrag; gr.plane_flags = q.plane_flags;
    gr.grad_en_en = ga->grad_en_en; gr.enabled = 1;
    q.unary_expect = prog->U > 0 ? ga->unary_expect : nullptr;
    q.unary_kind = ga->unary_kind; q.unary_obs = ga->unary_obs; q.unary_label = ga->unary_label;
    q.phi_i0 = ga->phi_en_en; q.phi_i1 = ga->phi_en_en_w1; q.phi_ed = ga->phi_en_de;
    q.grad_en_en = ga->grad_en_en; q.grad_en_de = ga->grad_en_de; q.Vde = ga->Vde; q.grad_on = 1;
    d.msgs = a->msgs;                               // the epilogue reads the stored variable->factor messages back
    d.vf_direct = (pf && sp.vf_direct && a->marginals) ? 1 : 0;      // ... or, product-fused, takes them from the message tiles
  }
  out->lds = lds; out->n_wg = n_groups; out->n_prep_blocks = (a->B + PGB - 1) / PGB;       // (= n_groups)
  out->wide = !p3 && sp.max_sources > 2; out->spill = d.spill != nullptr;
  *ok = true;
  return MLBP_OK;
}

// The verdict on the launch just issued (and only on it: launch_begin() drops what an earlier runtime call of this thread -- the
// caller's, another library's -- may have left in the thread's last-error slot), with the runtime's own words.
inline void launch_begin() { (void)hipGetLastError(); }
inline int launch_verdict(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MLBP_OK : fail(MLBP_EHIP, "%s launch failed: %s", what, hipGetErrorString(e));
}

// unary factor -> variable messages of the call, written back behind the sweeps when the caller wants the message buffer
int enqueue_unary_writeback(const mlbp_program* prog, const mlbp_sweep_args* a, const SharedDev& d, hipStream_t st) {
  const SharedProgram& sp = prog->shared;
  if (!(d.msgs && !d.vf_only && sp.n_cpw > 0)) return MLBP_OK;
  const int E = sp.n_cpw / 4;
  const long long rows = (long long)a->B * E;
  hipLaunchKernelGGL(unary_writeback_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(WG), 0, st, a->unary_tables, a->unary_tab,
                     prog->d_simage + sp.off_ent, E, a->B, prog->U, a->n_unary_tables, prog->n_msgs, a->msgs);
  if (int e = launch_verdict("unary write-back")) return e;
  return MLBP_OK;
}

}  // namespace

int launch_shared_sweep(const mlbp_program* prog, const mlbp_sweep_args* a, void* stream, bool* launched) {
  *launched = false;
  SharedPlan pl;
  bool ok = false;
  if (int e = shared_plan(prog, a, &ok, &pl)) return e;
  if (!ok) return MLBP_OK;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  hipLaunchKernelGGL(shared_prepare_kernel<false>, dim3(pl.n_prep_blocks), dim3(PWG), (size_t)pl.q.n_cprod * TILE * sizeof(double), st, pl.q, nullptr, nullptr, 0);
  if (int e = launch_verdict("shared-table prepare")) return e;
  sweep_fn k = nullptr;
  if (int e = pick_sweep_kernel(prog->P >= 2, pl.spill, pl.wide, false, pl.d.gr.enabled != 0, pl.pf, pl.lds, &k, pl.p3)) return e;
  launch_begin();
  hipLaunchKernelGGL(k, dim3(pl.n_wg), dim3(SWG), pl.lds, st, pl.d, nullptr, nullptr, 0);
  if (int e = launch_verdict("shared-table sweep")) return e;
  if (int e = enqueue_unary_writeback(prog, a, pl.d, st)) return e;
  *launched = true;
  return MLBP_OK;
}

// Several (program, arguments) groups in ONE launch sequence (prepare + sweeps) of the shared-table kernels -- a minibatch of
// mixed sentence shapes whose pairwise factors all read the two shared pots (train_mp.py:220-255, 257-299).  *launched stays
// false when some group does not qualify.  The group tables live in a device buffer owned by the first program and are
// uploaded only when their contents change (like the lean kernel's).
int launch_shared_groups(const mlbp_program* const* progs, const mlbp_sweep_args* args, int n_groups, void* stream, bool* launched) {
  *launched = false;
  if (n_groups < 1) return MLBP_OK;
  std::vector<SharedPlan> plans(n_groups);
  int max_cprod = 1, n_planned = 0;
  for (int k = 0; k < n_groups; ++k) {
    if (!progs[k]) return MLBP_OK;
    for (int j = 0; j < k; ++j)
      if (progs[j] == progs[k]) return MLBP_OK;      // two groups would share one set of redo flags and scratch
    if (progs[k]->P == 0) {
      // a sentence shape with ONE predicted word: no pairwise factor, nothing for the matrix cores.  Every graph of such a group is
      // handed to the fix-up pass of the call (the exact kernel over all groups' flagged graphs, and the flagged graphs' gradient:
      // finish_shared_groups) -- its launches are shared with every other group instead of two launches per such shape
      const mlbp_sweep_args* a = &args[k];
      if (a->X != 64 || !a->normalize_messages || !a->init_messages) return MLBP_OK;
      memset(&plans[k], 0, sizeof(plans[k]));
      continue;
    }
    bool ok = false;
    if (int e = shared_plan(progs[k], &args[k], &ok, &plans[k])) return e;
    if (!ok) return MLBP_OK;
    ++n_planned;
    max_cprod = std::max(max_cprod, (int)plans[k].q.n_cprod);
  }
  if (n_planned == 0) return MLBP_OK;                            // (nothing for these kernels: the per-group path)
  for (int k = 0; k < n_groups; ++k)
    if (progs[k]->P == 0) {
      mlbp_program* mp = const_cast<mlbp_program*>(progs[k]);
      if (mp->bail_cap < args[k].B)
        if (int e = mlbp_program_reserve(mp, args[k].B)) return e;
      if (hipMemsetAsync(mp->d_bail, 1, (size_t)args[k].B, (hipStream_t)stream) != hipSuccess) return fail(MLBP_EHIP, "flagging a pairwise-free group failed");
    }
  // Fragment copies of the pairwise tables (and the gradient's weighted ones): written ONCE per distinct set of inputs -- the
  // groups of a minibatch read the same pots; with one small group per sentence shape (hundreds of them) every group writing its
  // own 640 KB was the launch: 272 us of a 0.69 ms epoch.  The first set is the launch's shared job (the kernel's by-value
  // description, spread over all blocks); the other groups read the owner's copies.
  PrepareDev launch_job;
  memset(&launch_job, 0, sizeof(launch_job));
  {
    std::vector<int> owners;
    for (int k = 0; k < n_groups; ++k) {
      PrepareDev& q = plans[k].q;
      if (q.n_frag_tables == 0 && q.n_wfrag_tables == 0) continue;
      int own = -1;
      for (int j : owners) {
        const PrepareDev& o = plans[j].q;
        if (o.pair_tables == q.pair_tables && o.n_frag_tables == q.n_frag_tables && o.n_wfrag_tables == q.n_wfrag_tables &&
            (q.n_wfrag_tables == 0 || (o.phi_p0 == q.phi_p0 && o.phi_p1 == q.phi_p1))) { own = j; break; }
      }
      if (own < 0) { owners.push_back(k); continue; }
      if (q.n_frag_tables > 0) plans[k].d.tfrag = plans[own].d.tfrag;
      if (q.n_wfrag_tables > 0) { plans[k].d.gr.wfrag = plans[own].d.gr.wfrag; plans[k].d.gr.plane_flags = plans[own].d.gr.plane_flags; }
      q.n_frag_tables = 0; q.n_wfrag_tables = 0;
    }
    if (!owners.empty()) {
      launch_job = plans[owners[0]].q;
      plans[owners[0]].q.n_frag_tables = 0; plans[owners[0]].q.n_wfrag_tables = 0;
    }
  }
  // The groups run in up to three sweep launches, one per FORM of the kernel: product-fused (K2, K3, chains, rings), its
  // three-source variant (K4), general (larger cliques, spilled tiles) -- a K3 group of a minibatch that also holds a K4 sentence
  // keeps its own, faster instance (and its gradient epilogue that reads the message tiles).  One prepare launch in front of all.
  // table image: [SharedDev x n][PrepareDev x n][prepare starts n + 1][sweep starts of class 0 | 1 | 2, each (its groups) + 1], as 32-bit
  // words; the groups in class order
  auto cls = [&](int k) { return progs[k]->P == 0 ? 3 : (plans[k].pf ? 0 : (plans[k].p3 ? 1 : 2)); };      // (3: not in these launches)
  std::vector<int> order;
  int first[4] = {0, 0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    first[c] = (int)order.size();
    for (int k = 0; k < n_groups; ++k) if (cls(k) == c) order.push_back(k);
  }
  first[3] = (int)order.size();
  const int n_tab = (int)order.size();                           // groups in the table (the pairwise-free ones are not)
  const size_t w_sd = sizeof(SharedDev) / 4, w_pd = sizeof(PrepareDev) / 4;
  static_assert(sizeof(SharedDev) % 8 == 0 && sizeof(PrepareDev) % 8 == 0, "group tables are copied as words");
  std::vector<int32_t> table((w_sd + w_pd) * n_tab + (n_tab + 1) + (n_tab + 3));
  int32_t* pstarts = table.data() + (w_sd + w_pd) * n_tab;
  int32_t* sstarts = pstarts + n_tab + 1;                     // class c: sstarts[first[c] + c .. first[c + 1] + c]
  int pb = 0, grid[3] = {0, 0, 0};
  for (int j = 0; j < n_tab; ++j) {
    const int k = order[j];
    memcpy(table.data() + w_sd * j, &plans[k].d, sizeof(SharedDev));
    memcpy(table.data() + w_sd * n_tab + w_pd * j, &plans[k].q, sizeof(PrepareDev));
    pstarts[j] = pb; pb += plans[k].n_prep_blocks;
  }
  pstarts[n_tab] = pb;
  for (int c = 0; c < 3; ++c) {
    int wg = 0;
    for (int j = first[c]; j < first[c + 1]; ++j) { sstarts[j + c] = wg; wg += plans[order[j]].n_wg; }
    sstarts[first[c + 1] + c] = wg;
    grid[c] = wg;
  }
  mlbp_program* owner = const_cast<mlbp_program*>(progs[0]);
  hipStream_t st = (hipStream_t)stream;
  int32_t* d_stable = nullptr;               // one device copy per distinct table: a captured graph keeps replaying against its own
  if (int e = group_table_device(owner->stables, table, stream, &d_stable)) return e;
  const SharedDev* d_sd = reinterpret_cast<const SharedDev*>(d_stable);
  const PrepareDev* d_pd = reinterpret_cast<const PrepareDev*>(d_stable + w_sd * n_tab);
  const int32_t* d_pstarts = d_stable + (w_sd + w_pd) * n_tab;
  const int32_t* d_sstarts = d_pstarts + n_tab + 1;
  launch_begin();
  hipLaunchKernelGGL(shared_prepare_kernel<true>, dim3(pb), dim3(PWG), (size_t)max_cprod * TILE * sizeof(double), st, launch_job, d_pd, d_pstarts, n_tab);
  if (int e = launch_verdict("shared-table prepare")) return e;
  // Two or more forms present: the product-fused groups' launch goes to a side stream of the owner program, forked behind the
  // prepare launch and joined in front of whatever follows -- the launches are independent, and the three-source form (one
  // workgroup per CU, its last round partly empty) leaves CUs the product-fused workgroups fill.  (In a captured graph: two
  // parallel kernel nodes.)  MLBP_SHARED_GROUPS_SERIAL=1 in the environment keeps one stream.
  static const bool serial = getenv("MLBP_SHARED_GROUPS_SERIAL") != nullptr;
  const bool fork = !serial && first[1] > first[0] && first[3] > first[1];
  hipStream_t side = st;
  if (fork) {
    if (!owner->side_stream) {
      hipStream_t s2 = nullptr; hipEvent_t e1 = nullptr, e2 = nullptr;
      if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess)
        return fail(MLBP_EHIP, "side stream for grouped launches: creation failed");
      owner->side_stream = s2; owner->ev_fork = e1; owner->ev_join = e2;
    }
    side = (hipStream_t)owner->side_stream;
    if (hipEventRecord((hipEvent_t)owner->ev_fork, st) != hipSuccess || hipStreamWaitEvent(side, (hipEvent_t)owner->ev_fork, 0) != hipSuccess)
      return fail(MLBP_EHIP, "side stream for grouped launches: fork failed");
  }
  for (int cc = 0; cc < 3; ++cc) {
    const int c = fork ? (cc + 1) % 3 : cc;                      // (forked: the long launches first, the product-fused one beside them)
    const int n = first[c + 1] - first[c];
    if (n == 0) continue;
    size_t lds = 0;
    bool wide = false, spill = false, two = false, grad = false;
    for (int j = first[c]; j < first[c + 1]; ++j) {
      const SharedPlan& pl = plans[order[j]];
      lds = std::max(lds, pl.lds);
      wide |= pl.wide; spill |= pl.spill; two |= progs[order[j]]->P >= 2; grad |= pl.d.gr.enabled != 0;
    }
    // (a general launch with the gradient epilogue runs the all-resident / two-source instance, or -- as soon as one group spills
    // tiles or has wider updates -- the spilling, wide one: pick_sweep_kernel)
    sweep_fn k = nullptr;
    if (int e = pick_sweep_kernel(two, spill, wide, true, grad, c == 0, lds, &k, c == 1)) return e;
    launch_begin();
    hipLaunchKernelGGL(k, dim3(grid[c]), dim3(SWG), lds, (fork && c == 0) ? side : st, plans[order[first[c]]].d, d_sd + first[c], d_sstarts + first[c] + c, n);
    if (int e = launch_verdict("shared-table sweep")) return e;
  }
  if (fork) {
    if (hipEventRecord((hipEvent_t)owner->ev_join, side) != hipSuccess || hipStreamWaitEvent(st, (hipEvent_t)owner->ev_join, 0) != hipSuccess)
      return fail(MLBP_EHIP, "side stream for grouped launches: join failed");
  }
  for (int g = 0; g < n_groups; ++g)
    if (int e = enqueue_unary_writeback(progs[g], &args[g], plans[g].d, st)) return e;
  *launched = true;
  return MLBP_OK;
}

size_t shared_gradient_workspace_bytes(const mlbp_gradient_args* a) { return sizeof(double) * (size_t)a->n_pair_tables * 2 * 4 * 4096; }

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
  // fragment scratch: the caller's workspace (mlbp_gradient_args.workspace), else the process-wide fallback block (calls that
  // use it on different streams must not overlap; never freed, so a captured graph stays valid)
  if (a->n_pair_tables > PG_MAXW) return fail(MLBP_EUNSUPPORTED, "shared-table pair gradient: at most %d pairwise tables", PG_MAXW);
  const size_t need = shared_gradient_workspace_bytes(a);
  double* wfrag = nullptr;
  if (a->workspace) {
    if (a->workspace_bytes < need) return fail(MLBP_EINVAL, "mlbp_gradient_f64: workspace of %zu bytes, %zu needed", (size_t)a->workspace_bytes, need);
    wfrag = static_cast<double*>(a->workspace);
  } else {
    void* blk = nullptr;
    if (int e = fallback_scratch(SCRATCH_SHARED_GRADIENT, need, &blk)) return e;
    wfrag = static_cast<double*>(blk);
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
