// pf_astar.h -- one-wavefront-per-agent A* connector (K2a / K2b).
//
// Reference semantics (bit-exact pop order on the total order (f, g, (r,c))):
//   VARIANT 0  AStarSolver.solve, astar.py:33-101  (closed set, decrease-key, cap 3RC)
//   VARIANT 1  MPA._a_star,       MPA.py:106-151   (no closed set, stale (f,g) entries
//              stay, popped nodes may be re-pushed, g from g_score, cap 2RC)
//
// Mapping.  The open list lives in LDS as 64 bins x S slots (SoA f/g/cell).
// Lane b caches the lexicographic minimum of bin b in registers, so a pop is
// one DPP min-reduction over the 64 cached f keys (ties resolved on g then
// (r,c) by ballot), and only the winning bin is rescanned: one coalesced LDS
// read of its <= S entries + one more reduction.  The 8 neighbours are
// expanded by lanes 0..7 in parallel (the order of relaxations within one pop
// cannot change the resulting open *set*), each with ONE 16-byte load of the
// neighbour's record from this agent's HBM-resident scratch; lane 8 fetches the
// popped cell's own record and lane 9 the static move mask in the same batch,
// so every pop pays a single memory round trip.  Improved neighbours are pushed
// round-robin into distinct bins through an 8-entry LDS staging area.
//
// VARIANT 0 uses lazy deletion: a decrease-key pushes a second entry; the
// superseded one has a larger g than the record (or the cell is closed) when
// it surfaces and is dropped without counting a step, which reproduces the
// reference's in-place replace + heapify exactly (at most one *valid* entry
// per node exists, and it always sorts before its stale twin).
#pragma once
#include "pf_device.h"

namespace pf {

struct Open {
  double* lf;  // [64*S] bin-major
  double* lg;
  int* lc;     // packed (r<<16)|c
  double* sf;  // [8] push staging
  double* sg;
  int* sc;
  int S;
};

struct Slot {
  Rec* rec;
  uint32_t tag;       // solve epoch
  uint32_t avoid_ep;  // eval epoch (24 bit)
};

struct AStat {
  unsigned long long pops, pushes, nbr, stale;
  int max_open;
};

PF_DEV bool ent_lt(double f1, double g1, int c1, double f2, double g2, int c2) {
  if (f1 != f2) return f1 < f2;
  if (g1 != g2) return g1 < g2;
  return c1 < c2;
}

// Among lanes in `tie` (all holding the same f), the one with the smallest (g, cell).
PF_DEV int resolve_tie(unsigned long long tie, double g, int c) {
  int w = __builtin_ctzll(tie);
  double bg = bcast_d(g, w);
  int bc = bcast_i(c, w);
  tie &= tie - 1;
  while (tie) {
    int l = __builtin_ctzll(tie);
    tie &= tie - 1;
    double lg_ = bcast_d(g, l);
    int lc_ = bcast_i(c, l);
    if (lg_ < bg || (lg_ == bg && lc_ < bc)) { w = l; bg = lg_; bc = lc_; }
  }
  return w;
}

// Mark cells[0..n) of a path as "avoid" for this slot's current eval epoch.
PF_DEV void mark_avoid(const Slot& s, const int* cells, int n, int lane) {
  for (int i = lane; i < n; i += 64) s.rec[cells[i]].meta = s.avoid_ep << PF_AVOID_SHIFT;
}

// Returns status (PF_ST_*).  On PF_ST_OK, out[0..out_n) holds the path cells
// (r*C+c) start..target.  out_cap is the room available at `out`.
template <int VARIANT>
__device__ int astar(const Grid& G, Slot& s, const Open& O, int start, int target, int* out, int out_cap,
                     int& out_n, AStat& st, int lane) {
  out_n = 0;
  const int C = G.C;
  const int sr = row_of(G, start), sc_ = start - sr * C;
  const int tr = row_of(G, target), tc = target - tr * C;
  if (VARIANT == 1 && start == target) {                    // MPA.py:107
    if (out_cap < 1) return 3;
    if (lane == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  if (G.occ[start] == 1 || G.occ[target] == 1) return 1;     // astar.py:37-39 / MPA.py:109-111 (cells are in bounds)
  if (VARIANT == 0 && start == target) {                    // astar.py:41
    if (out_cap < 1) return 3;
    if (lane == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  s.tag += 1;
  const uint32_t tag = s.tag;
  const uint32_t avm = s.avoid_ep;
  Rec* rec = s.rec;
  const int S = O.S;

  // per-lane constants: lanes 0..7 own one move each
  const int d = lane & 7;
  const int ddr = HM_DR[d], ddc = HM_DC[d];
  const double cost = d < 4 ? 1.0 : PF_SQRT2;

  // per-lane bin state
  double mf = PF_INF, mg = 0.0;
  int mc = 0, ms = 0, cnt = 0;
  int rr = 1;          // next bin for pushes
  int n_open = 0;

  // seed: (h(start), 0, start) into bin 0; record g(start) = 0
  {
    long dr0 = sr - tr, dc0 = sc_ - tc;
    double h0 = __builtin_sqrt((double)(dr0 * dr0 + dc0 * dc0));
    if (lane == 0) {
      O.lf[0] = h0; O.lg[0] = 0.0; O.lc[0] = (sr << 16) | sc_;
      mf = h0; mg = 0.0; mc = (sr << 16) | sc_; ms = 0; cnt = 1;
      Rec r0 = rec[start];
      Rec w; w.g = 0.0; w.tag = tag; w.meta = (r0.meta & ~0xFFu) | (VARIANT == 1 ? PF_M_INOPEN : 0u);
      rec[start] = w;
    }
    n_open = 1; st.pushes += 1;
  }
  const long long max_steps = (long long)G.R * C * (VARIANT == 0 ? 3 : 2);   // astar.py:58 / MPA.py:118
  long long steps = 0;
  int status = 1;

  for (;;) {
    if (steps >= max_steps) { status = n_open > 0 ? 2 : 1; break; }
    // ---- pop: global min over the 64 cached bin minima ----
    double fmin = wave_min_d(mf);
    if (fmin == PF_INF) { status = 1; break; }               // open list empty
    unsigned long long tie = __ballot(mf == fmin);
    int w = __builtin_ctzll(tie);
    if (tie & (tie - 1)) w = resolve_tie(tie, mg, mc);
    const double pg = bcast_d(mg, w);
    const int pprc = bcast_i(mc, w);
    const int pslot = bcast_i(ms, w);
    const int pr = pprc >> 16, pc = pprc & 0xFFFF;
    const int cur = pr * C + pc;
    // ---- issue this pop's memory batch (one round trip) ----
    const int nr = pr + ddr, nc = pc + ddc;
    const bool inb = lane < 8 && nr >= 0 && nr < G.R && nc >= 0 && nc < C;
    const int nidx = nr * C + nc;
    Rec rn; rn.g = 0.0; rn.tag = 0; rn.meta = 0;
    unsigned mmask = 0;
    if (inb) rn = rec[nidx];
    else if (lane == 8) rn = rec[cur];
    else if (lane == 9) mmask = G.mm[cur];
    // ---- remove the popped entry from bin w and rescan it (LDS, overlaps the loads) ----
    const int wcnt = bcast_i(cnt, w) - 1;
    if (lane == w) {
      cnt = wcnt;
      if (pslot != wcnt) {
        const int a = w * S + pslot, b = w * S + wcnt;
        O.lf[a] = O.lf[b]; O.lg[a] = O.lg[b]; O.lc[a] = O.lc[b];
      }
    }
    n_open -= 1;
    {
      double vf = PF_INF, vg = 0.0; int vc = 0;
      if (lane < wcnt) { vf = O.lf[w * S + lane]; vg = O.lg[w * S + lane]; vc = O.lc[w * S + lane]; }
      double rmin = wave_min_d(vf);
      int j = 0;
      if (rmin != PF_INF) {
        unsigned long long t2 = __ballot(vf == rmin);
        j = __builtin_ctzll(t2);
        if (t2 & (t2 - 1)) j = resolve_tie(t2, vg, vc);
      }
      const double jg = bcast_d(vg, j);
      const int jc = bcast_i(vc, j);
      if (lane == w) { mf = rmin; mg = jg; mc = jc; ms = j; }
    }
    // ---- the popped cell's own record ----
    const double cur_g = bcast_d(rn.g, 8);
    const uint32_t cur_tag = (uint32_t)bcast_i((int)rn.tag, 8);
    const uint32_t cur_meta = (uint32_t)bcast_i((int)rn.meta, 8);
    double base_g;
    if (VARIANT == 0) {
      const bool valid = cur_tag == tag && cur_g == pg && !(cur_meta & PF_M_CLOSED);
      if (!valid) { st.stale += 1; continue; }               // superseded entry: not a reference pop
      base_g = pg;                                           // astar.py:85 uses the popped g
    } else {
      base_g = cur_g;                                        // MPA.py:135 uses g_score[current]
    }
    steps += 1;
    if (cur == target) { status = 0; break; }                // astar.py:64 / MPA.py:123
    if (lane == 8) {                                         // astar.py:74 closed.add / leave the open list
      uint32_t m2 = VARIANT == 0 ? (cur_meta | PF_M_CLOSED) : (cur_meta & ~PF_M_INOPEN);
      rec[cur].meta = m2;
    }
    // ---- relax the 8 neighbours in parallel ----
    const unsigned M = (unsigned)bcast_i((int)mmask, 9);
    const bool rvalid = rn.tag == tag;
    const bool avoided = (rn.meta >> PF_AVOID_SHIFT) == avm;
    bool ok = inb && ((M >> d) & 1u);
    if (VARIANT == 0) ok = ok && !(rvalid && (rn.meta & PF_M_CLOSED)) && !(avoided && nidx != start && nidx != target);
    else ok = ok && !avoided;
    const double tent = base_g + cost;
    const bool better = ok && (!rvalid || tent < rn.g);      // astar.py:87 / MPA.py:137
    const bool was_open = VARIANT == 1 && rvalid && (rn.meta & PF_M_INOPEN);
    const bool push = better && !was_open;                   // MPA.py:141-150: an open node keeps its old entry
    st.nbr += (unsigned long long)__builtin_popcountll(__ballot(ok));
    double fnew = 0.0;
    if (better) {
      long dr1 = nr - tr, dc1 = nc - tc;
      fnew = tent + __builtin_sqrt((double)(dr1 * dr1 + dc1 * dc1));   // astar.py:90 / MPA.py:140
      Rec wv; wv.g = tent; wv.tag = tag;
      wv.meta = (rn.meta & ~0xFFu) | (unsigned)d | ((VARIANT == 1 && (was_open || push)) ? PF_M_INOPEN : 0u);
      rec[nidx] = wv;
    }
    // ---- pushes: k-th pushing lane -> bin (rr + k) & 63 via LDS staging ----
    const unsigned long long pm = __ballot(push);
    const int np = __builtin_popcountll(pm);
    if (np) {
      if (push) {
        const int k = __builtin_popcountll(pm & ((1ull << lane) - 1ull));
        O.sf[k] = fnew; O.sg[k] = tent; O.sc[k] = (nr << 16) | nc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int k2 = (lane - rr) & 63;
      bool ovf = false;
      if (k2 < np) {
        if (cnt >= S) ovf = true;
        else {
          const double ef = O.sf[k2], eg = O.sg[k2]; const int ec = O.sc[k2];
          const int a = lane * S + cnt;
          O.lf[a] = ef; O.lg[a] = eg; O.lc[a] = ec;
          if (ent_lt(ef, eg, ec, mf, mg, mc) || mf == PF_INF) { mf = ef; mg = eg; mc = ec; ms = cnt; }
          cnt += 1;
        }
      }
      __builtin_amdgcn_wave_barrier();
      unsigned long long om = __ballot(ovf);
      while (om) {                                           // designated bin full: any bin with room takes it
        const int l = __builtin_ctzll(om);
        om &= om - 1;
        const int k3 = (l - rr) & 63;
        const unsigned long long freem = __ballot(cnt < S);
        if (!freem) { status = 3; break; }                   // all 64*S slots used: caller retries with a larger S
        if (lane == __builtin_ctzll(freem)) {
          const double ef = O.sf[k3], eg = O.sg[k3]; const int ec = O.sc[k3];
          const int a = lane * S + cnt;
          O.lf[a] = ef; O.lg[a] = eg; O.lc[a] = ec;
          if (ent_lt(ef, eg, ec, mf, mg, mc) || mf == PF_INF) { mf = ef; mg = eg; mc = ec; ms = cnt; }
          cnt += 1;
        }
      }
      if (status == 3) break;
      rr = (rr + np) & 63;
      n_open += np; st.pushes += np;
      if (n_open > st.max_open) st.max_open = n_open;
    }
  }
  st.pops += (unsigned long long)steps;
  if (status != 0) return status;

  // ---- walk parents target -> start (astar.py:65-69 / MPA.py:124-130), then reverse in place ----
  int n = 0, cell = target;
  const int guard = G.R * C;
  while (cell != start) {
    if (n >= out_cap - 1 || n > guard) return 3;
    if (lane == 0) out[n] = cell;
    const unsigned m = rec[cell].meta & PF_M_PARENT;
    cell -= HM_DR[m] * C + HM_DC[m];
    n += 1;
  }
  if (lane == 0) out[n] = start;
  n += 1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = lane; i < n / 2; i += 64) {
    int a = out[i], b = out[n - 1 - i];
    out[i] = b; out[n - 1 - i] = a;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  out_n = n;
  return 0;
}

}  // namespace pf
