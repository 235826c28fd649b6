// pf_astar16.h -- FOUR agents per wavefront: each 16-lane DPP row runs one A* search.
//
// Why.  With one agent per wave (pf_astar.h) a pop costs ~345 instructions of which at most 9 lanes do
// useful work; the chip tops out at ~0.5 G pops/s because the SIMDs are instruction-issue bound
// (profiles/r01_b_*).  The reductions of the pop are DPP row operations anyway, and a gfx950 DPP row is 16
// lanes -- so here every row is an independent agent ("row-uniform" values live in VGPRs, identical in the 16
// lanes of a row) and each instruction serves four searches.  Rows of one wave run in lockstep: a wave
// fetches four queue neighbours (the queue is sorted longest-expected-first, so their work is similar) and
// finishes when the longest of the four does.
//
// Differences to the 64-lane version, nothing semantic:
//   * 16 bins x PF_S16 (=32) LDS slots per agent (10 KiB; 40 KiB per wave, 4 waves per CU = 16 agents/CU),
//     tier-2 spill 16 x 64 in HBM; an agent that still overflows reports PF_ST_OVERFLOW and is re-run by the
//     64-lane kernel (capacity 5120).
//   * broadcasts inside a row are ds_bpermute (per-lane source index), row reductions are 4 DPP steps and
//     leave the result in every lane of the row (no readlane).
//   * the 9 relaxing lanes rotate inside the row: lane (rr + m) & 15.
#pragma once
#include "pf_astar.h"

namespace pf {

#define PF_S16 32
#define PF_LDS16 (16 * PF_S16 * 20) /* bytes of LDS per agent */

PF_DEV int rlane() { return lane_id() & 15; }
PF_DEV int rbase() { return lane_id() & 48; }
PF_DEV int rgrp() { return lane_id() >> 4; }
PF_DEV unsigned rballot(bool p) { return (unsigned)(__ballot(p) >> rbase()) & 0xFFFFu; }
PF_DEV int rbcast_i(int v, int k) { return __builtin_amdgcn_ds_bpermute((rbase() + k) << 2, v); }
PF_DEV double rbcast_d(double v, int k) {
  const int lo = rbcast_i(__double2loint(v), k), hi = rbcast_i(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
PF_DEV unsigned rmin_u32(unsigned v) {   // every lane of the row ends with the row minimum
  v = dpp_umin<0x121, 0xF>(v);
  v = dpp_umin<0x122, 0xF>(v);
  v = dpp_umin<0x124, 0xF>(v);
  v = dpp_umin<0x128, 0xF>(v);
  return v;
}
// 16-bit mask of the row's lanes holding the minimum of a non-negative double key
PF_DEV unsigned rargmin_d(double key, unsigned& mh, unsigned& ml) {
  const unsigned hi = (unsigned)__double2hiint(key), lo = (unsigned)__double2loint(key);
  mh = rmin_u32(hi);
  ml = rmin_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return rballot(hi == mh && lo == ml);
}
PF_DEV int rresolve_tie(unsigned tie, double g, int c) {
  int w = __builtin_ctz(tie);
  double bg = rbcast_d(g, w);
  int bc = rbcast_i(c, w);
  tie &= tie - 1;
  while (tie) {
    const int l = __builtin_ctz(tie);
    tie &= tie - 1;
    const double lg_ = rbcast_d(g, l);
    const int lc_ = rbcast_i(c, l);
    if (lg_ < bg || (lg_ == bg && lc_ < bc)) { w = l; bg = lg_; bc = lc_; }
  }
  return w;
}

struct Open16 {
  double* lf;  // [16*PF_S16] this row's bins
  double* lg;
  int* lc;
  double* of;  // tier 2 [16*PF_T2]
  double* og;
  int* oc;
};
PF_DEV Open16 make_open16(char* smem, char* tier2_slot) {
  Open16 O;
  char* b = smem + rgrp() * PF_LDS16;
  O.lf = (double*)b; O.lg = O.lf + 16 * PF_S16; O.lc = (int*)(O.lg + 16 * PF_S16);
  O.of = (double*)tier2_slot; O.og = O.of + 16 * PF_T2; O.oc = (int*)(O.og + 16 * PF_T2);
  return O;
}

PF_DEV void mark_avoid16(const Slot& s, const int* cells, int n) {
  for (int i = rlane(); i < n; i += 16) s.rec[cells[i]].meta = s.avoid_ep << PF_AVOID_SHIFT;
}
PF_DEV void slot_wipe16(Slot& s, int RC) {
  for (int i = rlane(); i < RC; i += 16) { Rec z; z.g = 0.0; z.tagmm = s.mm[i]; z.meta = 0; s.rec[i] = z; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  s.avoid_ep = 1; s.tag = 1;
}

// bounded pocket check (see pocket_flood): 2 frontier cells x 8 moves per step on the row's 16 lanes
#define PF_FLOOD_K16 256
PF_DEV int pocket_flood16(const Grid& G, const Slot& s, int* lds, int from, int to, int exempt) {
  int* tab = lds;                    // [PF_FLOOD_TAB]
  int* queue = lds + PF_FLOOD_TAB;   // [PF_FLOOD_K16]
  const int L = rlane();
  for (int i = L; i < PF_FLOOD_TAB; i += 16) tab[i] = -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (L == 0) { queue[0] = from; tab[(unsigned)(from * 0x9E3779B1u) >> 21] = from; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  const int C = G.C;
  const uint32_t avm = s.avoid_ep;
  const int q = L >> 3, m = L & 7;
  const int ddr = move_dr(m), ddc = move_dc(m);
  int head = 0, tail = 1, res = 1;
  while (head < tail) {
    const bool live = head + q < tail;
    const int cell = live ? queue[head + q] : 0;
    head = head + 2 < tail ? head + 2 : tail;
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
    if (rballot(cand && n == to)) { res = 0; break; }
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
    const unsigned fm = rballot(fresh);
    const int nf = __builtin_popcount(fm);
    if (tail + nf > PF_FLOOD_K16) { res = 2; break; }
    if (fresh) queue[tail + __builtin_popcount(fm & ((1u << L) - 1u))] = n;
    tail += nf;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  return res;
}

// One search per 16-lane row.  Same contract as astar<VARIANT>() of pf_astar.h; every argument and the result
// are row-uniform.  `out` etc. are per-row pointers.
template <int VARIANT>
__device__ int astar16(const Grid& G, Slot& s, const Open16& O, int start, int target, int* out, int out_cap,
                       int& out_n, AStat& st) {
  out_n = 0;
  const int L = rlane();
  const int C = G.C;
  const int sr = row_of(G, start), sc_ = start - sr * C;
  const int tr = row_of(G, target), tc = target - tr * C;
  if (VARIANT == 1 && start == target) {                    // MPA.py:107
    if (out_cap < 1) return 3;
    if (L == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  if (G.occ[start] == 1 || G.occ[target] == 1) return 1;     // astar.py:37-39 / MPA.py:109-111
  if (VARIANT == 0 && start == target) {                    // astar.py:41
    if (out_cap < 1) return 3;
    if (L == 0) out[0] = start;
    out_n = 1;
    return 0;
  }
  if (G.comp && G.comp[start] != G.comp[target]) return 1;   // different static components (see astar<>)
  const uint32_t avm = s.avoid_ep;
  Rec* rec = s.rec;
  if (VARIANT == 1 && (rec[target].meta >> PF_AVOID_SHIFT) == avm) return 1;   // goal inside the avoid set (see astar<>)
  {
    const int ex = VARIANT == 0 ? start : -1;
    if (pocket_flood16(G, s, (int*)O.lf, target, start, ex) == 1) return 1;
    if (pocket_flood16(G, s, (int*)O.lf, start, target, VARIANT == 0 ? target : -1) == 1) return 1;
  }
  s.tag += 1;
  const uint32_t tag = s.tag;
  constexpr int S = PF_S16;

  unsigned occ = 0;                 // 32 LDS slots of this lane's bin
  unsigned long long occ2 = 0;      // 64 tier-2 slots
  bool any_t2 = false;
  double mf = PF_INF, mg = 0.0;
  int mc = 0, ms = 0;
  int rr = 1;
  int n_open = 1;
  {
    const long dr0 = sr - tr, dc0 = sc_ - tc;
    const double h0 = __builtin_sqrt((double)(dr0 * dr0 + dc0 * dc0));
    if (L == 0) {
      O.lf[0] = h0; O.lg[0] = 0.0; O.lc[0] = (sr << 16) | sc_;
      mf = h0; mg = 0.0; mc = (sr << 16) | sc_; ms = 0; occ = 1;
      const Rec r0 = rec[start];
      Rec w; w.g = 0.0; w.tagmm = (tag << PF_TAG_SHIFT) | (r0.tagmm & 0xFFu);
      w.meta = (r0.meta & PF_AVOID_KEEP) | (VARIANT == 1 ? PF_M_INOPEN : 0u);
      rec[start] = w;
    }
  }
  const int max_steps = G.R * C * (VARIANT == 0 ? 3 : 2);
  int steps = 0;
  unsigned nbr32 = 0, push32 = 1, dk32 = 0;
  int status = 1;

  for (;;) {
    if (steps >= max_steps) { status = n_open > 0 ? 2 : 1; break; }
    // ---- pop: argmin over the row's 16 cached bin minima ----
    unsigned kh, kl;
    const unsigned tie = rargmin_d(mf, kh, kl);
    if (kh == PF_INF_HI) { status = 1; break; }
    int w = __builtin_ctz(tie);
    if (tie & (tie - 1)) w = rresolve_tie(tie, mg, mc);
    const double pg = rbcast_d(mg, w);
    const int pprc = rbcast_i(mc, w);
    const int pslot = rbcast_i(ms, w);
    const int pr = pprc >> 16, pc = pprc & 0xFFFF;
    const int cur = pr * C + pc;
    // ---- one load per lane: 8 neighbours + the popped cell ----
    const int d = (L - rr) & 15;
    const int ddr = move_dr(d & 7), ddc = move_dc(d & 7);
    const int nr = pr + ddr, nc = pc + ddc;
    const bool inb = d < 8 && nr >= 0 && nr < G.R && nc >= 0 && nc < C;
    const int nidx = nr * C + nc;
    Rec rn; rn.g = 0.0; rn.tagmm = 0; rn.meta = 0;
    if (inb || d == 8) rn = rec[inb ? nidx : cur];
    const long hdr = nr - tr, hdc = nc - tc;
    const double hn = __builtin_sqrt((double)(hdr * hdr + hdc * hdc));
    // ---- free the popped slot, rescan bin w (each lane looks at slots L and L+16) ----
    if (L == w) {
      if (pslot < S) occ &= ~(1u << pslot);
      else occ2 &= ~(1ull << (pslot - S));
    }
    n_open -= 1;
    {
      const unsigned wocc = (unsigned)rbcast_i((int)occ, w);
      double vf = PF_INF, vg = 0.0; int vc = 0, vs = L;
      if ((wocc >> L) & 1u) { vf = O.lf[w * S + L]; vg = O.lg[w * S + L]; vc = O.lc[w * S + L]; }
      if ((wocc >> (L + 16)) & 1u) {
        const double f2 = O.lf[w * S + L + 16], g2 = O.lg[w * S + L + 16];
        const int c2 = O.lc[w * S + L + 16];
        if (vf == PF_INF || ent_lt(f2, g2, c2, vf, vg, vc)) { vf = f2; vg = g2; vc = c2; vs = L + 16; }
      }
      unsigned rh, rl;
      const unsigned t2 = rargmin_d(vf, rh, rl);
      int j = __builtin_ctz(t2);
      if (t2 & (t2 - 1)) j = rresolve_tie(t2, vg, vc);
      double jf = __hiloint2double((int)rh, (int)rl), jg = rbcast_d(vg, j);
      int jc = rbcast_i(vc, j), js = rbcast_i(vs, j);
      if (any_t2) {
        const unsigned o2lo = (unsigned)rbcast_i((int)(unsigned)occ2, w), o2hi = (unsigned)rbcast_i((int)(unsigned)(occ2 >> 32), w);
        const unsigned long long o2 = ((unsigned long long)o2hi << 32) | o2lo;
        if (o2) {
          // 64 tier-2 slots, 4 per lane
          double uf = PF_INF, ug = 0.0; int uc = 0, us = 0;
          for (int k = 0; k < 4; ++k) {
            const int sl = L + 16 * k;
            if ((o2 >> sl) & 1ull) {
              const double f2 = O.of[w * PF_T2 + sl], g2 = O.og[w * PF_T2 + sl];
              const int c2 = O.oc[w * PF_T2 + sl];
              if (uf == PF_INF || ent_lt(f2, g2, c2, uf, ug, uc)) { uf = f2; ug = g2; uc = c2; us = sl; }
            }
          }
          unsigned uh, ul;
          const unsigned t3 = rargmin_d(uf, uh, ul);
          int j3 = __builtin_ctz(t3);
          if (t3 & (t3 - 1)) j3 = rresolve_tie(t3, ug, uc);
          const double kf = rbcast_d(uf, j3), kg = rbcast_d(ug, j3);
          const int kc = rbcast_i(uc, j3), ks = rbcast_i(us, j3);
          if (jf == PF_INF || ent_lt(kf, kg, kc, jf, jg, jc)) { jf = kf; jg = kg; jc = kc; js = S + ks; }
        }
      }
      if (L == w) { mf = jf; mg = jg; mc = jc; ms = js; }
    }
    // ---- the popped cell's record (lane rr+8 of the row) ----
    const int lcur = (rr + 8) & 15;
    const double cur_g = rbcast_d(rn.g, lcur);
    const uint32_t cur_tagmm = (uint32_t)rbcast_i((int)rn.tagmm, lcur);
    const uint32_t cur_meta = (uint32_t)rbcast_i((int)rn.meta, lcur);
    const double base_g = VARIANT == 0 ? pg : cur_g;
    steps += 1;
    if (cur == target) { status = 0; break; }
    if (L == lcur) rec[cur].meta = VARIANT == 0 ? (cur_meta | PF_M_CLOSED) : (cur_meta & ~PF_M_INOPEN);
    // ---- relax ----
    const unsigned M = cur_tagmm & 0xFFu;
    const bool rvalid = (rn.tagmm >> PF_TAG_SHIFT) == tag;
    const bool avoided = (rn.meta >> PF_AVOID_SHIFT) == avm;
    const bool closed = rvalid && (rn.meta & PF_M_CLOSED);
    bool ok = inb && ((M >> (d & 7)) & 1u);
    if (VARIANT == 0) ok = ok && !closed && !(avoided && nidx != start && nidx != target);
    else ok = ok && !avoided;
    const double tent = base_g + (d < 4 ? 1.0 : PF_SQRT2);
    const bool better = ok && (!rvalid || tent < rn.g);
    const bool in_open = VARIANT == 0 ? rvalid : (rvalid && (rn.meta & PF_M_INOPEN));
    const bool push = better && !in_open;
    const bool deckey = VARIANT == 0 && better && in_open;
    nbr32 += (unsigned)__builtin_popcount(rballot(ok));
    double fnew = 0.0;
    unsigned pos = (rn.meta >> PF_POS_SHIFT) & PF_POS_MASK;
    bool ovf = false;
    if (better) {
      fnew = tent + hn;
      if (push) {
        const int prc = (nr << 16) | nc;
        int slot = -1;
        if (occ != 0xFFFFFFFFu) {
          slot = __builtin_ctz(~occ);
          occ |= 1u << slot;
          const int a = L * S + slot;
          O.lf[a] = fnew; O.lg[a] = tent; O.lc[a] = prc;
        } else if (occ2 != ~0ull) {
          const int j2 = __builtin_ctzll(~occ2);
          occ2 |= 1ull << j2;
          const int a = L * PF_T2 + j2;
          O.of[a] = fnew; O.og[a] = tent; O.oc[a] = prc;
          slot = S + j2;
        } else ovf = true;
        if (slot >= 0) {
          if (mf == PF_INF || ent_lt(fnew, tent, prc, mf, mg, mc)) { mf = fnew; mg = tent; mc = prc; ms = slot; }
          pos = ((unsigned)L << 7) | (unsigned)slot;
        }
      } else if (deckey) {
        const int b = (int)(pos >> 7), sl = (int)(pos & 127u);
        if (sl < S) { O.lf[b * S + sl] = fnew; O.lg[b * S + sl] = tent; }
        else { O.of[b * PF_T2 + sl - S] = fnew; O.og[b * PF_T2 + sl - S] = tent; }
      }
      if (!ovf) {
        Rec wv; wv.g = tent; wv.tagmm = (tag << PF_TAG_SHIFT) | (rn.tagmm & 0xFFu);
        wv.meta = (rn.meta & PF_AVOID_KEEP) | (pos << PF_POS_SHIFT) | (unsigned)(d & 7) | (VARIANT == 1 ? PF_M_INOPEN : 0u);
        rec[nidx] = wv;
      }
    }
    const int np = __builtin_popcount(rballot(push));
    if (VARIANT == 0) {
      unsigned dm = rballot(deckey);
      dk32 += (unsigned)__builtin_popcount(dm);
      while (dm) {
        const int l = __builtin_ctz(dm); dm &= dm - 1;
        const unsigned p2 = (unsigned)rbcast_i((int)pos, l);
        const double f2 = rbcast_d(fnew, l), g2 = rbcast_d(tent, l);
        const int c2 = (rbcast_i(nr, l) << 16) | rbcast_i(nc, l);
        if (L == (int)(p2 >> 7) && ((int)(p2 & 127u) == ms || ent_lt(f2, g2, c2, mf, mg, mc))) { mf = f2; mg = g2; mc = c2; ms = (int)(p2 & 127u); }
      }
    }
    unsigned om = rballot(ovf);
    while (om) {
      const int l = __builtin_ctz(om); om &= om - 1;
      const unsigned freem = rballot(occ != 0xFFFFFFFFu || occ2 != ~0ull);
      if (!freem) { status = 3; break; }
      const int t = __builtin_ctz(freem);
      const double f2 = rbcast_d(fnew, l), g2 = rbcast_d(tent, l);
      const int r2 = rbcast_i(nr, l), c2 = rbcast_i(nc, l), dd = rbcast_i(d, l);
      const uint32_t tm2 = (uint32_t)rbcast_i((int)rn.tagmm, l), me2 = (uint32_t)rbcast_i((int)rn.meta, l);
      if (L == t) {
        const int prc2 = (r2 << 16) | c2;
        int slot;
        if (occ != 0xFFFFFFFFu) {
          slot = __builtin_ctz(~occ); occ |= 1u << slot;
          const int a = L * S + slot;
          O.lf[a] = f2; O.lg[a] = g2; O.lc[a] = prc2;
        } else {
          const int j2 = __builtin_ctzll(~occ2); occ2 |= 1ull << j2;
          const int a = L * PF_T2 + j2;
          O.of[a] = f2; O.og[a] = g2; O.oc[a] = prc2;
          slot = S + j2;
        }
        if (mf == PF_INF || ent_lt(f2, g2, prc2, mf, mg, mc)) { mf = f2; mg = g2; mc = prc2; ms = slot; }
        Rec wv; wv.g = g2; wv.tagmm = (tag << PF_TAG_SHIFT) | (tm2 & 0xFFu);
        wv.meta = (me2 & PF_AVOID_KEEP) | ((((unsigned)L << 7) | (unsigned)slot) << PF_POS_SHIFT) | (unsigned)(dd & 7) |
                  (VARIANT == 1 ? PF_M_INOPEN : 0u);
        rec[r2 * C + c2] = wv;
      }
    }
    if (status == 3) break;
    if (!any_t2 && rballot(occ2 != 0)) any_t2 = true;
    rr = (rr + 9) & 15;
    n_open += np; push32 += (unsigned)np;
    if (n_open > st.max_open) st.max_open = n_open;
  }
  st.pops += (unsigned long long)steps; st.pushes += push32; st.nbr += nbr32; st.deckey += dk32;
  if (status != 0) return status;

  int n = 0, cell = target;
  const int guard = G.R * C;
  while (cell != start) {
    if (n >= out_cap - 1 || n > guard) return 3;
    if (L == 0) out[n] = cell;
    const unsigned m = rec[cell].meta & PF_M_PARENT;
    cell -= move_dr((int)m) * C + move_dc((int)m);
    n += 1;
  }
  if (L == 0) out[n] = start;
  n += 1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = L; i < n / 2; i += 16) {
    const int a = out[i], b = out[n - 1 - i];
    out[i] = b; out[n - 1 - i] = a;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  out_n = n;
  return 0;
}

}  // namespace pf

// ---------------------------------------------------------------------------
// row-scoped scoring (same arithmetic and order as score_path of pf_score.h)
// ---------------------------------------------------------------------------
#include "pf_score.h"
namespace pf {
PF_DEV void score_path16(const Grid& G, const ScoreP& P, const int* cells, int Ln, double* out) {
  if (Ln <= 0) { out[0] = PF_INF; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0; out[4] = PF_INF; return; }
  const int L = rlane();
  const int C = G.C;
  double length = 0.0, safety = 0.0;
  int turns = 0, ncut = 0;
  for (int base = 0; base < Ln; base += 16) {
    const int i = base + L;
    int r0 = 0, c0 = 0, r1 = 0, c1 = 0, r2 = 0, c2 = 0;
    const bool v0 = i < Ln, v1 = i + 1 < Ln, v2 = i + 2 < Ln;
    int cell0 = 0;
    if (v0) { cell0 = cells[i]; r0 = row_of(G, cell0); c0 = cell0 - r0 * C; }
    if (v1) { const int x = cells[i + 1]; r1 = row_of(G, x); c1 = x - r1 * C; }
    if (v2) { const int x = cells[i + 2]; r2 = row_of(G, x); c2 = x - r2 * C; }
    const int dr = r1 - r0, dc = c1 - c0;
    double cost = 0.0;
    bool cut = false, isdiag = false, isgen = false;
    if (v1) {
      const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
      if (adr + adc == 1) cost = 1.0;
      else if (adr == 1 && adc == 1) {
        cost = PF_SQRT2; isdiag = true;
        cut = P.restrict_policy && (G.occ[r1 * C + c0] == 1 || G.occ[r0 * C + c1] == 1);
      } else { cost = __builtin_sqrt((double)((long)dr * dr + (long)dc * dc)); isgen = true; }
    }
    const bool turn = v2 && (dr != r2 - r1 || dc != c2 - c1);
    double pen = 0.0;
    if (P.variant == 0 && v0) pen = P.pen[G.d2near[cell0]];
    turns += __builtin_popcount(rballot(turn));
    ncut += __builtin_popcount(rballot(cut));
    int nsteps = Ln - 1 - base; if (nsteps > 16) nsteps = 16;
    const unsigned dmask = rballot(isdiag), gmask = rballot(isgen);
    if (!gmask) {
      for (int j = 0; j < nsteps; ++j) length = length + (((dmask >> j) & 1u) ? PF_SQRT2 : 1.0);
    } else {
      for (int j = 0; j < nsteps; ++j) length = length + rbcast_d(cost, j);
    }
    unsigned nz = rballot(pen != 0.0);
    while (nz) { const int j = __builtin_ctz(nz); nz &= nz - 1; safety = safety + rbcast_d(pen, j); }
  }
  if (P.variant == 0) safety = safety / (double)Ln;
  double diag = 0.0;
  if (Ln >= 2) for (int k = 0; k < ncut; ++k) diag += P.diag_pen;
  out[0] = length; out[1] = (double)turns; out[2] = safety; out[3] = diag;
  out[4] = length + P.w_turn * (double)turns + P.w_safe * safety + diag;
}
}  // namespace pf
