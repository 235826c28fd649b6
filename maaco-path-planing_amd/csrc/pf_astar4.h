// pf_astar4.h -- speculative four-wide pop loop of the one-wavefront A* (included by pf_astar.h).
//
// A pop uses 9 of the 64 lanes and ~350 dependent instructions, so a lone long search (the tail of every
// batch) runs at the issue rate of ONE wave.  This loop pops up to four entries per trip through the same
// instruction stream and commits exactly the prefix the sequential reference would have produced:
//
//   * the 64 lane-owned bins form four 16-lane rows; each row's cached minimum is a CANDIDATE (row-local
//     DPP reduction, all four rows at once);
//   * row p relaxes the candidate of row (p + sh) & 3 on its own lanes (pushes land in row p's bins; sh
//     rotates so entries spread over the rows) while every row rescans the bin its own candidate leaves;
//   * all relax work happens in registers.  Then, in candidate order (f, g, cell), candidate b COMMITS iff
//     every earlier candidate a committed and nothing a did can come before b or touch what b reads:
//         min f left in a's row  >  f_b      (else that entry, not b, is the next pop)
//         min f pushed/decreased by a  >  f_b
//         Chebyshev distance(a, b) > 2        (disjoint 3x3 neighbourhoods: b's loads saw no write of a)
//         f_a != f_b, a is not the target
//     The first candidate is the true global minimum and always commits, so every trip makes progress; a
//     candidate that does not commit is untouched (its slot, its bin's cached minimum, the records) and is
//     simply a candidate again next trip.
//   * only committed candidates write: slot release + cached minimum of the popped bin, LDS/HBM pushes,
//     16-byte record stores, decrease-keys.
// Pop order, parents, g values, pop/push counts are those of the sequential loop bit for bit (the GPU parity
// tests compare paths AND pop counts with the CPU oracle).  Measured on G512 the four rows commit ~2 pops a
// trip (CPU simulation of this rule: 1.96).
#pragma once

namespace pf {

struct __attribute__((aligned(16))) SpecX {   // per-row exchange record in LDS (64 B)
  double f, g;                 // the row's candidate (its minimum entry); f == +inf: the row is empty
  int rc, sl;                  // packed cell (r << 16 | c); slot | owner lane << 8, or -1
  unsigned long long rm2;      // bits of the smallest f left in this row once its candidate is gone
  unsigned long long mp;       // bits of the smallest f this row pushed / decreased while relaxing (indexed by RELAX row)
  unsigned long long pad[3];
};
#define PF_SPEC_LDS 256
// Lanes of the wave talk through LDS here.  The hardware executes one wave's LDS instructions in order, so no
// wait is needed, but the compiler must not move a lane's load above another lane's (to it unrelated) store:
// a compiler-only barrier at each hand-over point.
#define PF_LDS_ORDER() asm volatile("" ::: "memory")

PF_DEV unsigned row_umin(unsigned v) {   // every lane gets the minimum of its 16-lane row
  v = dpp_umin<0x121, 0xF>(v);
  v = dpp_umin<0x122, 0xF>(v);
  v = dpp_umin<0x124, 0xF>(v);
  v = dpp_umin<0x128, 0xF>(v);
  return v;
}
PF_DEV unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }
PF_DEV bool ent_lt_nb(double f1, double g1, int c1, double f2, double g2, int c2) {   // branch-free (f, g, cell) order
  return (f1 < f2) | ((f1 == f2) & ((g1 < g2) | ((g1 == g2) & (c1 < c2))));
}
// one winner bit per row out of a ballot that may hold several f-equal lanes per row
PF_DEV unsigned long long one_per_row(unsigned long long m, double g, int c) {
  unsigned long long fixed = 0;
  for (int r = 0; r < 4; ++r) {
    const unsigned long long fld = m & (0xFFFFull << (16 * r));
    if (fld) fixed |= 1ull << ((fld & (fld - 1)) ? resolve_tie(fld, g, c) : __builtin_ctzll(fld));
  }
  return fixed;
}

// The pop loop.  On entry the open list holds the start entry (bin 0, slot 0).  Returns the status
// (0 target popped, 1 exhausted, 2 step cap, 3 open list overflow).
template <int VARIANT>
__device__ __forceinline__ int pop_loop4(const Grid& G, Rec* rec, const Open& O, uint32_t tag, uint32_t avm, int start, int target,
                                         int tr, int tc, int max_steps, AStat& st, int lane) {
  constexpr int SEM = VARIANT == 1 ? 1 : 0;
  constexpr int S = PF_S;
  static_assert(S == 16, "the four-wide loop scans a bin with one 16-lane row");
  constexpr unsigned full = 0xFFFFu;
  const unsigned long long INFB = 0x7FF0000000000000ull;
  const int C = G.C;
  const int row = lane >> 4, i = lane & 15, rb = lane & 48;
  SpecX* X = (SpecX*)O.sx;
  const int trc = (tr << 16) | tc;

  unsigned occ = 0;                // LDS slot occupancy of this lane's bin
  unsigned long long occ2 = 0;     // tier-2 (HBM) slot occupancy
  double mf = PF_INF, mg = 0.0;    // cached minimum of this lane's bin
  int mc = 0, ms = 0;
  if (lane == 0) { occ = 1; mf = O.lf[0]; mg = 0.0; mc = O.lc[0]; ms = 0; }
  int rr = 1, sh = 0;
  int n_open = 1, steps = 0, status = 1;
  unsigned nbr32 = 0, push32 = 1, dk32 = 0;

  for (;;) {
    if (steps >= max_steps) { status = n_open > 0 ? 2 : 1; break; }
    // ---- S1: each row's minimum bin = its candidate ----
    const unsigned fh = (unsigned)__double2hiint(mf), fl = (unsigned)__double2loint(mf);
    const unsigned rh = row_umin(fh);
    const unsigned rl = row_umin(fh == rh ? fl : 0xFFFFFFFFu);
    unsigned long long wm = __ballot(fh == rh && fl == rl && rh != PF_INF_HI);
    if (wm == 0) { status = 1; break; }                       // every bin is empty
    unsigned f16 = (unsigned)(wm >> rb) & 0xFFFFu;
    if (__ballot((f16 & (f16 - 1u)) != 0u)) {                 // f-tie between bins of one row: (g, cell) decides
      wm = one_per_row(wm, mg, mc);
      f16 = (unsigned)(wm >> rb) & 0xFFFFu;
    }
    const int wl = f16 ? __builtin_ctz(f16) : 0;              // sub-lane of my row's winner (lane 0 of an empty row: mf = +inf)
    const bool iswin = i == wl;
    if (iswin) {
      X[row].f = mf; X[row].g = mg; X[row].rc = mc; X[row].sl = f16 ? (ms | (lane << 8)) : -1;
      X[row].rm2 = INFB; X[row].mp = INFB;
    }
    PF_LDS_ORDER();
    // ---- S2: fetch the candidate this row relaxes; one batch of loads ----
    const int q = (row + sh) & 3;
    const double cg = X[q].g;
    const int crc = X[q].rc, csl = X[q].sl;
    const int osl = X[row].sl;
    const int w = rb + wl;                                    // owner lane of the bin my row's candidate leaves
    double vf = O.lf[w * S + i], vg = O.lg[w * S + i];
    int vc = O.lc[w * S + i];
    const bool has = csl >= 0;
    const int pr = crc >> 16, pc = crc & 0xFFFF;
    const int cur = pr * C + pc;
    const int d = (i - rr) & 15;                              // 0..7 = move index, 8 = the popped cell itself
    const int ddr = move_dr(d & 7), ddc = move_dc(d & 7);
    const int nr = pr + ddr, nc = pc + ddc;
    const bool inb = has && d < 8 && nr >= 0 && nr < G.R && nc >= 0 && nc < C;
    const bool self = has && d == 8;
    const int nidx = nr * C + nc;
    Rec rn; rn.g = 0.0; rn.tagmm = 0; rn.meta = 0;
    unsigned M = 0; double cur_g = 0.0;
    if (inb || self) rn = rec[inb ? nidx : cur];
    if (inb) { M = G.mm[cur]; if (SEM == 1) cur_g = rec[cur].g; }   // row-uniform addresses: one line each
    const long hdr = nr - tr, hdc = nc - tc;
    double hn = __builtin_sqrt((double)(hdr * hdr + hdc * hdc));       // astar.py:90 / MPA.py:140
    asm volatile("" : "+v"(hn));                              // keep the square root in the shadow of the loads
    // ---- S3: what my row's candidate leaves behind in its bin ----
    const bool hasown = osl >= 0;
    const int pslot = osl & 0xFF;
    if (i == pslot) vf = PF_INF;
    const unsigned vh = (unsigned)__double2hiint(vf), vl = (unsigned)__double2loint(vf);
    const unsigned h2 = row_umin(vh);
    const unsigned l2 = row_umin(vh == h2 ? vl : 0xFFFFFFFFu);
    unsigned long long m2 = __ballot(vh == h2 && vl == l2);
    unsigned f2 = (unsigned)(m2 >> rb) & 0xFFFFu;
    if (__ballot((f2 & (f2 - 1u)) != 0u && h2 != PF_INF_HI)) {
      m2 = one_per_row(m2 & __ballot(h2 != PF_INF_HI), vg, vc) | (m2 & __ballot(h2 == PF_INF_HI));
      f2 = (unsigned)(m2 >> rb) & 0xFFFFu;
    }
    int jslot = __builtin_ctz(f2);
    double jf = __hiloint2double((int)h2, (int)l2), jg = 0.0;
    int jc = 0;
    if (iswin) { jg = O.lg[w * S + jslot]; jc = O.lc[w * S + jslot]; }
    {
      unsigned long long t2m = __ballot(iswin && hasown && occ2 != 0ull);   // rare: the bin also has tier-2 entries
      while (t2m) {
        const int l = __builtin_ctzll(t2m); t2m &= t2m - 1;
        const unsigned o2lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)occ2, l);
        const unsigned o2hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(occ2 >> 32), l);
        const unsigned long long o2 = ((unsigned long long)o2hi << 32) | o2lo;
        const int ps = bcast_i(osl, l) & 0xFF;
        double uf = PF_INF, ug = 0.0; int uc = 0;
        if (((o2 >> lane) & 1ull) && lane + S != ps) { uf = O.of[l * PF_T2 + lane]; ug = O.og[l * PF_T2 + lane]; uc = O.oc[l * PF_T2 + lane]; }
        unsigned uh, ul;
        const unsigned long long t3 = argmin_mask_d<false>(uf, uh, ul);
        int j3 = __builtin_ctzll(t3);
        if (t3 & (t3 - 1)) j3 = resolve_tie(t3, ug, uc);
        const double kf = bcast_d(uf, j3), kg = bcast_d(ug, j3);
        const int kc = bcast_i(uc, j3);
        if (lane == l && (jf == PF_INF || ent_lt_nb(kf, kg, kc, jf, jg, jc))) { jf = kf; jg = kg; jc = kc; jslot = S + j3; }
      }
    }
    if (hasown) {
      const double x = iswin ? jf : mf;
      if (x != PF_INF) __hip_atomic_fetch_min(&X[row].rm2, dbits(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef PF_DBG
    if (blockIdx.x == 0 && steps == 2 && row == 3)
      printf("  lane %d i %d wl %d iswin %d osl %x pslot %d vf %.6f jf %.6f jslot %d mf %.6f ms %d occ %x f16 %x f2 %x\n", lane, i, wl, (int)iswin, osl, pslot, vf, jf, jslot, mf, ms, occ, f16, f2);
#endif
    // ---- S4: relax the 8 neighbours of my row's candidate, in registers ----
    const uint32_t cur_meta = rn.meta;                         // meaningful in the self lane
    const double base_g = SEM == 0 ? cg : cur_g;           // astar.py:85 popped g / MPA.py:135 g_score[current]
    const bool rvalid = (rn.tagmm >> PF_TAG_SHIFT) == tag;
    const bool avoided = (rn.meta >> PF_AVOID_SHIFT) == avm;
    const bool closed = rvalid && (rn.meta & PF_M_CLOSED);
    const bool goal_here = cur == target;                      // popping the target ends the search before any relaxation
    bool ok = inb && ((M >> (d & 7)) & 1u) && !goal_here;
    if (SEM == 0) ok = ok && !closed && !(avoided && nidx != start && nidx != target);
    else ok = ok && !avoided;
    const double tent = base_g + (d < 4 ? 1.0 : PF_SQRT2);
    const bool better = ok && (!rvalid || tent < rn.g);       // astar.py:87 / MPA.py:137
    const bool in_open = SEM == 0 ? rvalid : (rvalid && (rn.meta & PF_M_INOPEN));
    const bool push = better && !in_open;
    const bool deckey = SEM == 0 && better && in_open;    // astar.py:96-100
    const double fnew = tent + hn;                             // astar.py:90 / MPA.py:140
    if (push || deckey) __hip_atomic_fetch_min(&X[row].mp, dbits(fnew), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // ---- S5: which candidates commit ----
    PF_LDS_ORDER();
    unsigned CM, TM;
    int ncommit;
    {
      const int a = (lane >> 2) & 3, b = lane & 3;
      const double fa = X[a].f, fb = X[b].f, ga = X[a].g, gb = X[b].g;
      const int rca = X[a].rc, rcb = X[b].rc;
      const unsigned long long rm2a = X[a].rm2, mpa = X[(a - sh) & 3].mp;
      const bool exa = fa != PF_INF, exb = fb != PF_INF;
      const bool bef = (a != b) & exa & exb & ent_lt_nb(fa, ga, rca, fb, gb, rcb);
      const int dr_ = (rca >> 16) - (rcb >> 16), dc_ = (rca & 0xFFFF) - (rcb & 0xFFFF);
      const bool nearby = (unsigned)(dr_ + 2) <= 4u && (unsigned)(dc_ + 2) <= 4u;
      const unsigned long long fbb = dbits(fb);
      const bool inv = bef & ((rm2a <= fbb) | (mpa <= fbb) | nearby | (fa == fb) | (rca == trc));
      const unsigned MB = (unsigned)__ballot(bef) & 0xFFFFu;
      const unsigned MI = (unsigned)__ballot(inv) & 0xFFFFu;
      const unsigned ME = (unsigned)__ballot(exb) & 0xFu;     // lanes 0..3: a == 0, b == lane
      TM = (unsigned)__ballot(exb && rcb == trc) & 0xFu;
      const unsigned viol = (MI | (MI >> 4) | (MI >> 8) | (MI >> 12)) & 0xFu;
      int rstar = max_steps - steps;                           // the step cap counts pops (astar.py:58 / MPA.py:118)
      if (rstar > 4) rstar = 4;
      int rank[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rank[k] = __builtin_popcount((MB >> k) & 0x1111u);
        if ((viol >> k) & 1u) rstar = rank[k] < rstar ? rank[k] : rstar;
      }
      CM = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (((ME >> k) & 1u) && rank[k] < rstar) CM |= 1u << k;
      ncommit = __builtin_popcount(CM);
    }
#ifdef PF_DBG
    if (lane == 0 && blockIdx.x == 0) {
      printf("it steps=%d sh=%d rr=%d CM=%x TM=%x |", steps, sh, rr, CM, TM);
      for (int k = 0; k < 4; ++k) printf(" [%d] f=%.6f g=%.4f rc=(%d,%d) rm2=%.6f mp=%.6f |", k, X[k].f, X[k].g, X[k].rc >> 16, X[k].rc & 0xFFFF,
                                         __longlong_as_double((long long)X[k].rm2), __longlong_as_double((long long)X[k].mp));
      printf("\n");
    }
#endif
    // ---- S6: committed candidates take effect ----
    const bool cpop = (CM >> row) & 1u;
    const bool crel = ((CM >> q) & 1u) && !goal_here;
    if (iswin && cpop) {                                       // release the slot; the bin's new minimum is the rescan result
      if (pslot < S) { occ &= ~(1u << pslot); O.lf[lane * S + pslot] = PF_INF; }
      else occ2 &= ~(1ull << (pslot - S));
      mf = jf; mg = jg; mc = jc; ms = jslot;
    }
    if (self && crel)                                          // astar.py:74 closed.add / leave the open list
      rec[cur].meta = SEM == 0 ? (cur_meta | PF_M_CLOSED) : (cur_meta & ~PF_M_INOPEN);
    nbr32 += (unsigned)__builtin_popcountll(__ballot(ok && crel));
    unsigned pos = (rn.meta >> PF_POS_SHIFT) & PF_POS_MASK;
    bool ovf = false;
    if (better && crel) {
      if (push) {
        const int prc = (nr << 16) | nc;
        int slot = -1;
        if (occ != full) {
          slot = __builtin_ctz(~occ);
          occ |= 1u << slot;
          const int a_ = lane * S + slot;
          O.lf[a_] = fnew; O.lg[a_] = tent; O.lc[a_] = prc;
        } else if (occ2 != ~0ull) {                           // LDS slots of this bin are full: spill to HBM tier 2
          const int j2 = __builtin_ctzll(~occ2);
          occ2 |= 1ull << j2;
          const int a_ = lane * PF_T2 + j2;
          O.of[a_] = fnew; O.og[a_] = tent; O.oc[a_] = prc;
          slot = S + j2;
        } else ovf = true;
        if (slot >= 0) {
          if ((mf == PF_INF) | ent_lt_nb(fnew, tent, prc, mf, mg, mc)) { mf = fnew; mg = tent; mc = prc; ms = slot; }
          pos = ((unsigned)lane << 7) | (unsigned)slot;
        }
      } else if (deckey) {
        const int b_ = (int)(pos >> 7), sl_ = (int)(pos & 127u);
        if (sl_ < S) { O.lf[b_ * S + sl_] = fnew; O.lg[b_ * S + sl_] = tent; }
        else { O.of[b_ * PF_T2 + sl_ - S] = fnew; O.og[b_ * PF_T2 + sl_ - S] = tent; }
      }
      if (!ovf) {
        Rec wv; wv.g = tent; wv.tagmm = (tag << PF_TAG_SHIFT) | (rn.tagmm & 0xFFu);
        wv.meta = (rn.meta & PF_AVOID_KEEP) | (pos << PF_POS_SHIFT) | (unsigned)(d & 7) | (SEM == 1 ? PF_M_INOPEN : 0u);
        rec[nidx] = wv;
      }
    }
    const int np = __builtin_popcountll(__ballot(push && crel));
    if (SEM == 0) {                                        // decrease-key: the owning lane refreshes its cached minimum
      unsigned long long dm = __ballot(deckey && crel);
      dk32 += (unsigned)__builtin_popcountll(dm);
      while (dm) {
        const int l = __builtin_ctzll(dm); dm &= dm - 1;
        const unsigned p2 = (unsigned)bcast_i((int)pos, l);
        const double f2_ = bcast_d(fnew, l), g2_ = bcast_d(tent, l);
        const int c2_ = (bcast_i(nr, l) << 16) | bcast_i(nc, l);
        if (lane == (int)(p2 >> 7) && ((int)(p2 & 127u) == ms || ent_lt_nb(f2_, g2_, c2_, mf, mg, mc))) { mf = f2_; mg = g2_; mc = c2_; ms = (int)(p2 & 127u); }
      }
    }
    {                                                          // own bin full in both tiers: hand the entry to any lane with room
      unsigned long long om = __ballot(ovf);
      while (om) {
        const int l = __builtin_ctzll(om); om &= om - 1;
        const unsigned long long freem = __ballot(occ != full || occ2 != ~0ull);
        if (!freem) { status = 3; break; }                    // all 64*(S+PF_T2) slots used
        const int t = __builtin_ctzll(freem);
        const double f2_ = bcast_d(fnew, l), g2_ = bcast_d(tent, l);
        const int r2 = bcast_i(nr, l), c2 = bcast_i(nc, l), dd = bcast_i(d, l);
        const uint32_t tm2 = (uint32_t)bcast_i((int)rn.tagmm, l), me2 = (uint32_t)bcast_i((int)rn.meta, l);
        if (lane == t) {
          const int prc2 = (r2 << 16) | c2;
          int slot;
          if (occ != full) {
            slot = __builtin_ctz(~occ); occ |= 1u << slot;
            const int a_ = lane * S + slot;
            O.lf[a_] = f2_; O.lg[a_] = g2_; O.lc[a_] = prc2;
          } else {
            const int j2 = __builtin_ctzll(~occ2); occ2 |= 1ull << j2;
            const int a_ = lane * PF_T2 + j2;
            O.of[a_] = f2_; O.og[a_] = g2_; O.oc[a_] = prc2;
            slot = S + j2;
          }
          if ((mf == PF_INF) | ent_lt_nb(f2_, g2_, prc2, mf, mg, mc)) { mf = f2_; mg = g2_; mc = prc2; ms = slot; }
          Rec wv; wv.g = g2_; wv.tagmm = (tag << PF_TAG_SHIFT) | (tm2 & 0xFFu);
          wv.meta = (me2 & PF_AVOID_KEEP) | ((((unsigned)lane << 7) | (unsigned)slot) << PF_POS_SHIFT) | (unsigned)(dd & 7) |
                    (SEM == 1 ? PF_M_INOPEN : 0u);
          rec[r2 * C + c2] = wv;
        }
      }
      if (status == 3) break;
    }
    steps += ncommit;
    n_open += np - ncommit; push32 += (unsigned)np;
    if (CM & TM) { status = 0; break; }                       // astar.py:64 / MPA.py:123: the target was popped
    rr = (rr + 9) & 15; sh = (sh + 1) & 3;
#ifdef PF_TRIPS
    st.max_open += 1;                                          // diagnostic build: trips instead of the open-list high-water mark
#else
    if (n_open > st.max_open) st.max_open = n_open;
#endif
    PF_LDS_ORDER();
  }
  st.pops += (unsigned long long)steps; st.pushes += push32; st.nbr += nbr32; st.deckey += dk32;
  return status;
}

}  // namespace pf
