// pf_astar.h -- one-wavefront-per-agent A* connector (K2a / K2b).
//
// Reference semantics (bit-exact pop order on the total order (f, g, (r,c))):
//   VARIANT 0  AStarSolver.solve, astar.py:33-101  (closed set, decrease-key, cap 3RC)
//   VARIANT 1  MPA._a_star,       MPA.py:106-151   (no closed set, an open node keeps its
//              old (f,g) entry when improved, popped nodes may be re-pushed,
//              g from g_score, cap 2RC)
//
// Mapping (MI355X, wave64).  The open list lives in LDS as 64 bins x S slots
// (SoA f/g/cell; a free slot holds f = +inf).  Lane b owns bin b: it keeps the
// bin's occupancy bitmask and its lexicographic minimum in registers.
//   pop     = min-reduction of the 64 cached f keys done as two passes of
//             v_min_u32 with the DPP modifier (hi word, then lo word among the
//             hi-ties) + ballot; ties on f are resolved on (g, (r,c)).  Only
//             the winning bin is rescanned: one LDS read of its S slots by S
//             lanes + one more (row-local when S == 16) reduction.
//   expand  = the 8 neighbours are relaxed by 8 lanes in parallel (relaxations
//             within one pop touch distinct nodes, so their order cannot change
//             the open *set*).  Each lane issues ONE 16-byte load of its
//             neighbour's record from this agent's HBM scratch; a ninth lane
//             loads the popped cell's own record, whose tag word also carries
//             the cell's static move mask (bounds / obstacle / corner-cut rule,
//             helper.py:38-52) -- one memory round trip per pop.
//   push    = the relaxing lanes rotate every pop (lane = (rr + move) & 63), so
//             the lane that found an improvement inserts into ITS OWN bin: no
//             cross-lane traffic, and pushes spread evenly over the bins.
//   decrease-key (VARIANT 0, astar.py:96-100) = the record stores the entry's
//             (bin, slot); the relaxing lane rewrites (f, g) in place and the
//             owning lane refreshes its cached minimum.  Slots never move
//             (tombstones), so positions stay valid.
#pragma once
#include "pf_device.h"

namespace pf {

#ifndef PF_S
#define PF_S 16 /* LDS slots per bin (compile-time: LDS addresses fold to constants, fewer SGPRs) */
#endif
struct Open {
  double* lf;  // [64*S] bin-major; +inf == free slot
  double* lg;
  int* lc;     // packed (r<<16)|c
  // tier 2: PF_T2 more slots per bin in this agent's HBM scratch, used only when a bin's LDS slots are
  // full, so a search never has to restart with a larger LDS footprint
  double* of;  // [64*PF_T2]
  double* og;
  int* oc;
  char* sx;    // PF_SPEC_LDS bytes of LDS for the speculative loop's row exchange (pf_astar4.h)
};
#ifndef PF_LOOP
#define PF_LOOP 2 /* pop loop: 2 sorted window over a bucket pool (pf_astar_sw.h), 1 four-wide speculative pops over lane-owned
                     bins (pf_astar4.h), 0 one pop per trip over lane-owned bins; 0/1 are kept for A/B builds */
#endif

struct Slot {
  Rec* rec;
  const uint8_t* mm;  // move masks the records were initialised from (for the wrap wipe)
  uint32_t tag;       // solve epoch (24 bit)
  uint32_t avoid_ep;  // eval epoch (14 bit)
};

struct AStat {
  unsigned long long pops, pushes, nbr, deckey;
  int max_open;
  unsigned spills;   // open-list entries that went through the spill list (full bucket / beyond the circular range)
};
// diagnostic build only (-DPF_STAMPS): shader-clock time per section of the pop loop, never in the product .so
#ifdef PF_STAMPS
__device__ unsigned long long g_stamps[16];
#define PF_T(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define PF_ACC(i, a, b) st_acc[i] += (b) - (a);
#else
#define PF_T(var)
#define PF_ACC(i, a, b)
#endif

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

// (re)initialise a slot: every record carries its cell's static move mask, epoch 0
PF_DEV void slot_wipe(Slot& s, int RC, int lane) {
  for (int i = lane; i < RC; i += 64) { Rec z; z.g = 0.0; z.tagmm = s.mm[i]; z.meta = 0; s.rec[i] = z; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  s.avoid_ep = 1; s.tag = 1;
}

// helper-order move deltas from the move index without a table lookup
PF_DEV int move_dr(int d) { return (int)((0x0A25u >> (2 * d)) & 3u) - 1; }   // {0,0,1,-1,1,1,-1,-1} + 1 packed
PF_DEV int move_dc(int d) { return (int)((0x2252u >> (2 * d)) & 3u) - 1; }   // {1,-1,0,0,1,-1,1,-1} + 1 packed

}  // namespace pf
#include "pf_astar4.h"
#include "pf_astar_sw.h"
namespace pf {

// ---------------------------------------------------------------------------
// Bounded pocket check.  When the goal (or the start) is sealed inside a small pocket by the avoid set, the
// reference's A* pops the ENTIRE reachable region (~190 k pops on G512) before returning [] -- these cases
// are ~1 % of MPA rebuilds but set the tail of every batch.  The search graph is undirected (free cells;
// the corner-cut rule of helper.py:45-49 looks at the same two cells in both directions), and a cell may be
// entered iff it is not in the avoid set (VARIANT 0 exempts start/target, astar.py:55-56).  So a flood from
// `from` that exhausts its frontier without meeting `to` proves A* would fail; the flood is capped at
// PF_FLOOD_K cells (it answers "unknown" beyond that and the real search runs).  8 frontier cells x 8 moves
// per step on the 64 lanes; visited set = LDS hash table carved from the (not yet used) open-list area.
// Returns 0 reachable, 1 proven unreachable, 2 unknown.
// ---------------------------------------------------------------------------
#define PF_FLOOD_K 512
#define PF_FLOOD_TAB 2048
PF_DEV int pocket_flood(const Grid& G, const Slot& s, int* lds, int from, int to, int exempt, int lane) {
  int* tab = lds;                    // [PF_FLOOD_TAB] visited cells (open addressing), -1 = empty
  int* queue = lds + PF_FLOOD_TAB;   // [PF_FLOOD_K]
  for (int i = lane; i < PF_FLOOD_TAB; i += 64) tab[i] = -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  if (lane == 0) { queue[0] = from; tab[(unsigned)(from * 0x9E3779B1u) >> 21] = from; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  const int C = G.C;
  const uint32_t avm = s.avoid_ep;
  const int q = lane >> 3, m = lane & 7;
  const int ddr = move_dr(m), ddc = move_dc(m);
  int head = 0, tail = 1;
  while (head < tail) {
    const bool live = head + q < tail;
    int cell = live ? queue[head + q] : 0;
    head = head + 8 < tail ? head + 8 : tail;
    bool cand = false; int n = 0;
    if (live) {
      const unsigned mm = G.mm[cell];
      if ((mm >> m) & 1u) {
        const int r = row_of(G, cell), c = cell - r * C;
        n = (r + ddr) * C + (c + ddc);
        const bool avoided = (s.rec[n].meta >> PF_AVOID_SHIFT) == avm;
        cand = !avoided || n == to || n == exempt;
      }
    }
    if (__ballot(cand && n == to)) return 0;
    // claim unseen cells in the hash table
    bool fresh = false;
    if (cand) {
      unsigned h = (unsigned)(n * 0x9E3779B1u) >> 21;
      for (;;) {
        const int old = atomicCAS(&tab[h], -1, n);
        if (old == -1) { fresh = true; break; }
        if (old == n) break;
        h = (h + 1) & (PF_FLOOD_TAB - 1);
      }
    }
    const unsigned long long fm = __ballot(fresh);
    const int nf = __builtin_popcountll(fm);
    if (tail + nf > PF_FLOOD_K) return 2;
    if (fresh) queue[tail + __builtin_popcountll(fm & ((1ull << lane) - 1ull))] = n;
    tail += nf;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
  }
  return 1;
}

// Returns status (PF_ST_*).  On PF_ST_OK, out[0..out_n) holds the path cells
// (r*C+c) start..target.  out_cap is the room available at `out`.
template <int VARIANT>
__device__ __forceinline__ int astar(const Grid& G, Slot& s, const Open& O, int start, int target, int* out, int out_cap,
                     int& out_n, AStat& st, int lane) {
  // VARIANT 0 AStarSolver.solve (astar.py:33-101), 1 MPA._a_star (MPA.py:106-151), 2 DijkstraSolver.solve
  // (dijkstra.py:32-97: the loop of variant 0 with heap entries (g, node), i.e. h == 0 and key (g, g, node))
  constexpr int SEM = VARIANT == 1 ? 1 : 0;
  static_assert(VARIANT != 2 || PF_LOOP == 2, "the Dijkstra variant is built on the sorted-window loop");
  out_n = 0;
  // every argument is the same in all 64 lanes, but callers often compute them with vector instructions (a cell
  // drawn by the lane-replicated RNG, a value loaded from a path): tell the compiler, so the per-search constants
  // and everything derived from them live in scalar registers and scalar instructions
  start = first_i(start); target = first_i(target); out_cap = first_i(out_cap);
  const int C = G.C;
  const int sr = row_of(G, start), sc_ = start - sr * C;
  const int tr = row_of(G, target), tc = target - tr * C;
  if (SEM == 1 && start == target) {                    // MPA.py:107
    if (out_cap < 1) return 3;
    if (lane == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  if (G.occ[start] == 1 || G.occ[target] == 1) return 1;     // astar.py:37-39 / MPA.py:109-111 (cells are in bounds)
  if (SEM == 0 && start == target) {                    // astar.py:41
    if (out_cap < 1) return 3;
    if (lane == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  // Moves are symmetric (a diagonal and its reverse are gated by the same two orthogonal cells), so the free
  // cells split into components no search can leave, whatever its avoid set.  A goal in another component
  // makes both references pop the start's whole component and return [] (astar.py:103 / MPA.py:151).
  if (G.comp && G.comp[start] != G.comp[target]) return 1;
  const uint32_t avm = (uint32_t)first_i((int)s.avoid_ep);
  Rec* rec = s.rec;
  // MPA._a_star drops avoid nodes from every neighbour list (MPA.py:82) with no exemption for the goal, so a
  // goal inside the avoid set can never be pushed: the reference then pops the whole reachable region and
  // returns [] (MPA.py:151).  Same result, none of the work.  (On G512 this case is 44 % of all pops of an
  // MPA sweep: the Brownian target often lands on the predator's own prefix.)
  if (SEM == 1 && (rec[target].meta >> PF_AVOID_SHIFT) == avm) return 1;
  // small-pocket proof of unreachability, from both ends (see pocket_flood)
  {
    const int ex = SEM == 0 ? start : -1;               // VARIANT 0: start/target may sit in the avoid set
    if (pocket_flood(G, s, (int*)O.lf, target, start, ex, lane) == 1) return 1;
    if (pocket_flood(G, s, (int*)O.lf, start, target, SEM == 0 ? target : -1, lane) == 1) return 1;
  }
  s.tag = (uint32_t)first_i((int)s.tag) + 1;
  const uint32_t tag = s.tag;
  constexpr int S = PF_S;
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);

  // all slots of this lane's bin free
  for (int j = 0; j < S; ++j) O.lf[lane * S + j] = PF_INF;
  unsigned long long occ = 0, occ2 = 0;   // tier-1 (LDS) / tier-2 (HBM) slot occupancy of this lane's bin
  bool any_t2 = false;                    // uniform: some bin of this search has spilled to tier 2
  double mf = PF_INF, mg = 0.0;
  int mc = 0, ms = 0;
  int rr = 1;          // lane of move 0 for the current pop (rotates)
  int n_open = 0;

  // seed: (h(start), 0, start) into bin 0; record g(start) = 0
  double h0_seed;
  {
    long dr0 = sr - tr, dc0 = sc_ - tc;
    double h0 = VARIANT == 2 ? 0.0 : __builtin_sqrt((double)(dr0 * dr0 + dc0 * dc0));
    h0_seed = h0;
    if (lane == 0) {
      O.lf[0] = h0; O.lg[0] = 0.0; O.lc[0] = (sr << 16) | sc_;
      mf = h0; mg = 0.0; mc = (sr << 16) | sc_; ms = 0; occ = 1;
      Rec r0 = rec[start];
      Rec w; w.g = 0.0; w.tagmm = (tag << PF_TAG_SHIFT) | (r0.tagmm & 0xFFu);
      w.meta = (r0.meta & PF_AVOID_KEEP) | (SEM == 1 ? PF_M_INOPEN : 0u);   // position (0,0)
      rec[start] = w;
    }
    n_open = 1;
  }
  const int max_steps = G.R * C * (SEM == 0 ? 3 : 2);   // astar.py:58 / MPA.py:118 (R,C <= 4096)
#if PF_LOOP != 0
  (void)occ; (void)occ2; (void)any_t2; (void)mf; (void)mg; (void)mc; (void)ms; (void)rr; (void)n_open; (void)full; (void)h0_seed;
#if PF_LOOP == 2
  const int status4 = pop_loop_sw<VARIANT>(G, rec, O, tag, avm, start, target, tr, tc, max_steps, h0_seed, (sr << 16) | sc_, st, lane);
#else
  const int status4 = pop_loop4<VARIANT>(G, rec, O, tag, avm, start, target, tr, tc, max_steps, st, lane);
#endif
  if (status4 != 0) return status4;
#else
  int steps = 0;
  unsigned nbr32 = 0, push32 = 1, dk32 = 0;
  int status = 1;

#ifdef PF_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
    PF_T(t0)
    if (steps >= max_steps) { status = n_open > 0 ? 2 : 1; break; }
    // ---- pop: argmin over the 64 cached bin minima ----
    unsigned kh, kl;
    unsigned long long tie = argmin_mask_d<false>(mf, kh, kl);
    int w = __builtin_ctzll(tie);
    if (kh == PF_INF_HI) { status = 1; break; }              // every bin minimum is +inf: open list empty
    PF_T(t1)
    if (tie & (tie - 1)) { w = resolve_tie(tie, mg, mc); }
    PF_T(t2)
    const double pg = bcast_d(mg, w);
    const int pprc = bcast_i(mc, w);
    const int pslot = bcast_i(ms, w);
    const int pr = pprc >> 16, pc = pprc & 0xFFFF;
    const int cur = pr * C + pc;
    // ---- this pop's memory batch: one load instruction, one round trip ----
    const int d = (lane - rr) & 63;                           // 0..7 = move index, 8 = the popped cell itself
    const int ddr = move_dr(d & 7), ddc = move_dc(d & 7);
    const int nr = pr + ddr, nc = pc + ddc;
    const bool inb = d < 8 && nr >= 0 && nr < G.R && nc >= 0 && nc < C;
    const int nidx = nr * C + nc;
    Rec rn; rn.g = 0.0; rn.tagmm = 0; rn.meta = 0;
    if (inb || d == 8) rn = rec[inb ? nidx : cur];
    // the heuristic of each neighbour does not depend on the load: computed in its shadow (astar.py:90 / MPA.py:140)
    const long hdr = nr - tr, hdc = nc - tc;
    const double hn = __builtin_sqrt((double)(hdr * hdr + hdc * hdc));
    PF_T(t3)
    // ---- free the popped slot and rescan bin w (LDS; overlaps the load) ----
    if (lane == w) {
      if (pslot < S) { occ &= ~(1ull << pslot); O.lf[w * S + pslot] = PF_INF; }
      else occ2 &= ~(1ull << (pslot - S));
    }
    n_open -= 1;
    {
      double vf = PF_INF, vg = 0.0; int vc = 0;
      if (lane < S) { vf = O.lf[w * S + lane]; vg = O.lg[w * S + lane]; vc = O.lc[w * S + lane]; }
      unsigned rh, rl;
      const unsigned long long t2 = S <= 16 ? (argmin_mask_d<true>(vf, rh, rl) & 0xFFFFull) : argmin_mask_d<false>(vf, rh, rl);
      int j = __builtin_ctzll(t2);
      if (t2 & (t2 - 1)) j = resolve_tie(t2, vg, vc);
      // every lane re-reads the winning slot (uniform LDS address = broadcast read) instead of 5 v_readlane
      double jf = __hiloint2double((int)rh, (int)rl), jg = O.lg[w * S + j];
      int jc = O.lc[w * S + j];
      if (any_t2) {
        const unsigned o2lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)occ2, w);
        const unsigned o2hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(occ2 >> 32), w);
        const unsigned long long o2 = ((unsigned long long)o2hi << 32) | o2lo;
        if (o2) {                                             // the bin also has tier-2 entries: scan them too
          double uf = PF_INF, ug = 0.0; int uc = 0;
          if ((o2 >> lane) & 1ull) { uf = O.of[w * PF_T2 + lane]; ug = O.og[w * PF_T2 + lane]; uc = O.oc[w * PF_T2 + lane]; }
          unsigned uh, ul;
          const unsigned long long t3 = argmin_mask_d<false>(uf, uh, ul);
          int j3 = __builtin_ctzll(t3);
          if (t3 & (t3 - 1)) j3 = resolve_tie(t3, ug, uc);
          const double kf = bcast_d(uf, j3), kg = bcast_d(ug, j3);
          const int kc = bcast_i(uc, j3);
          if (jf == PF_INF || ent_lt(kf, kg, kc, jf, jg, jc)) { jf = kf; jg = kg; jc = kc; j = S + j3; }
        }
      }
      if (lane == w) { mf = jf; mg = jg; mc = jc; ms = j; }   // jf == +inf when the bin is now empty
    }
    PF_T(t4)
    // ---- the popped cell's own record (lane rr+8) ----
    const int lcur = (rr + 8) & 63;
    const double cur_g = bcast_d(rn.g, lcur);
    PF_T(t5)
    const uint32_t cur_tagmm = (uint32_t)bcast_i((int)rn.tagmm, lcur);
    const uint32_t cur_meta = (uint32_t)bcast_i((int)rn.meta, lcur);
    const double base_g = SEM == 0 ? pg : cur_g;          // astar.py:85 popped g / MPA.py:135 g_score[current]
    steps += 1;
    if (cur == target) { status = 0; break; }                // astar.py:64 / MPA.py:123
    if (lane == lcur)                                         // astar.py:74 closed.add / leave the open list
      rec[cur].meta = SEM == 0 ? (cur_meta | PF_M_CLOSED) : (cur_meta & ~PF_M_INOPEN);
    // ---- relax the 8 neighbours in parallel ----
    const unsigned M = cur_tagmm & 0xFFu;
    const bool rvalid = (rn.tagmm >> PF_TAG_SHIFT) == tag;
    const bool avoided = (rn.meta >> PF_AVOID_SHIFT) == avm;
    const bool closed = rvalid && (rn.meta & PF_M_CLOSED);
    bool ok = inb && ((M >> (d & 7)) & 1u);
    if (SEM == 0) ok = ok && !closed && !(avoided && nidx != start && nidx != target);
    else ok = ok && !avoided;
    const double tent = base_g + (d < 4 ? 1.0 : PF_SQRT2);
    const bool better = ok && (!rvalid || tent < rn.g);      // astar.py:87 / MPA.py:137
    // a valid, unclosed record has a live open entry in VARIANT 0; VARIANT 1 tracks it with a flag
    const bool in_open = SEM == 0 ? rvalid : (rvalid && (rn.meta & PF_M_INOPEN));
    const bool push = better && !in_open;
    const bool deckey = SEM == 0 && better && in_open;   // astar.py:96-100
    nbr32 += (unsigned)__builtin_popcountll(__ballot(ok));
    double fnew = 0.0;
    unsigned pos = (rn.meta >> PF_POS_SHIFT) & PF_POS_MASK;
    bool ovf = false;
    if (better) {
      fnew = tent + hn;                                        // astar.py:90 / MPA.py:140
      if (push) {
        const int prc = (nr << 16) | nc;
        int slot = -1;
        if (occ != full) {
          slot = __builtin_ctzll(~occ);
          occ |= 1ull << slot;
          const int a = lane * S + slot;
          O.lf[a] = fnew; O.lg[a] = tent; O.lc[a] = prc;
        } else if (occ2 != ~0ull) {                           // LDS slots of this bin are full: spill to HBM tier 2
          const int j2 = __builtin_ctzll(~occ2);
          occ2 |= 1ull << j2;
          const int a = lane * PF_T2 + j2;
          O.of[a] = fnew; O.og[a] = tent; O.oc[a] = prc;
          slot = S + j2;
        } else ovf = true;
        if (slot >= 0) {
          if (mf == PF_INF || ent_lt(fnew, tent, prc, mf, mg, mc)) { mf = fnew; mg = tent; mc = prc; ms = slot; }
          pos = ((unsigned)lane << 7) | (unsigned)slot;
        }
      } else if (deckey) {
        const int b = (int)(pos >> 7), sl = (int)(pos & 127u);
        if (sl < S) { O.lf[b * S + sl] = fnew; O.lg[b * S + sl] = tent; }
        else { O.of[b * PF_T2 + sl - S] = fnew; O.og[b * PF_T2 + sl - S] = tent; }
      }
      if (!ovf) {
        Rec wv; wv.g = tent; wv.tagmm = (tag << PF_TAG_SHIFT) | (rn.tagmm & 0xFFu);
        wv.meta = (rn.meta & PF_AVOID_KEEP) | (pos << PF_POS_SHIFT) | (unsigned)(d & 7) | (SEM == 1 ? PF_M_INOPEN : 0u);
        rec[nidx] = wv;
      }
    }
    PF_T(t6)
    const int np = __builtin_popcountll(__ballot(push));
    // ---- decrease-key: the owning lane refreshes its cached minimum ----
    if (SEM == 0) {
      unsigned long long dm = __ballot(deckey);
      dk32 += (unsigned)__builtin_popcountll(dm);
      while (dm) {
        const int l = __builtin_ctzll(dm); dm &= dm - 1;
        const unsigned p2 = (unsigned)bcast_i((int)pos, l);
        const double f2 = bcast_d(fnew, l), g2 = bcast_d(tent, l);
        const int c2 = (bcast_i(nr, l) << 16) | bcast_i(nc, l);
        if (lane == (int)(p2 >> 7) && ((int)(p2 & 127u) == ms || ent_lt(f2, g2, c2, mf, mg, mc))) { mf = f2; mg = g2; mc = c2; ms = (int)(p2 & 127u); }
      }
    }
    // ---- own bin full in both tiers: hand the entry to any lane with room ----
    unsigned long long om = __ballot(ovf);
    while (om) {
      const int l = __builtin_ctzll(om); om &= om - 1;
      const unsigned long long freem = __ballot(occ != full || occ2 != ~0ull);
      if (!freem) { status = 3; break; }                     // all 64*(S+PF_T2) slots used
      const int t = __builtin_ctzll(freem);
      const double f2 = bcast_d(fnew, l), g2 = bcast_d(tent, l);
      const int r2 = bcast_i(nr, l), c2 = bcast_i(nc, l), dd = bcast_i(d, l);
      const uint32_t tm2 = (uint32_t)bcast_i((int)rn.tagmm, l), me2 = (uint32_t)bcast_i((int)rn.meta, l);
      if (lane == t) {
        const int prc2 = (r2 << 16) | c2;
        int slot;
        if (occ != full) {
          slot = __builtin_ctzll(~occ); occ |= 1ull << slot;
          const int a = lane * S + slot;
          O.lf[a] = f2; O.lg[a] = g2; O.lc[a] = prc2;
        } else {
          const int j2 = __builtin_ctzll(~occ2); occ2 |= 1ull << j2;
          const int a = lane * PF_T2 + j2;
          O.of[a] = f2; O.og[a] = g2; O.oc[a] = prc2;
          slot = S + j2;
        }
        if (mf == PF_INF || ent_lt(f2, g2, prc2, mf, mg, mc)) { mf = f2; mg = g2; mc = prc2; ms = slot; }
        Rec wv; wv.g = g2; wv.tagmm = (tag << PF_TAG_SHIFT) | (tm2 & 0xFFu);
        wv.meta = (me2 & PF_AVOID_KEEP) | ((((unsigned)lane << 7) | (unsigned)slot) << PF_POS_SHIFT) | (unsigned)(dd & 7) |
                  (SEM == 1 ? PF_M_INOPEN : 0u);
        rec[r2 * C + c2] = wv;
      }
    }
    if (status == 3) break;
    if (!any_t2 && __ballot(occ2 != 0)) any_t2 = true;       // after every insertion path of this pop
    rr = (rr + 9) & 63;
    n_open += np; push32 += (unsigned)np;
    PF_T(t7)
    PF_ACC(0, t0, t1) PF_ACC(1, t1, t2) PF_ACC(2, t2, t3) PF_ACC(3, t3, t4) PF_ACC(4, t4, t5) PF_ACC(5, t5, t6) PF_ACC(6, t6, t7)
#ifdef PF_STAMPS
    st_acc[7] += (tie & (tie - 1)) ? 1 : 0;
#endif
    if (n_open > st.max_open) st.max_open = n_open;
  }
#ifdef PF_STAMPS
  if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
#endif
  st.pops += (unsigned long long)steps; st.pushes += push32; st.nbr += nbr32; st.deckey += dk32;
  if (status != 0) return status;
#endif

  // ---- walk parents target -> start (astar.py:65-69 / MPA.py:124-130), then reverse in place ----
  int n = 0, cell = target;
  const int guard = G.R * C;
  while (cell != start) {
    if (n >= out_cap - 1 || n > guard) return 3;
    if (lane == 0) out[n] = cell;
    const unsigned m = rec[cell].meta & PF_M_PARENT;
    cell -= move_dr((int)m) * C + move_dc((int)m);
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
