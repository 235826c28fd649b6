// pf_astar_pr.h -- two wavefronts per search: the POP wave and the POOL wave (included by pf_astar.h after pf_astar_sw.h).
//
// A lone wave issues one dependent instruction per ~5 clocks, and a trip of the sorted-window loop is ~880 of them
// (DESIGN.md 4.2): the search is bound by its own instruction stream while half of the chip idles.  What does NOT feed the
// next trip's heads can leave that stream: the bucket pool.  In this mode a search runs on a 128-thread workgroup:
//
//   wave 0 (P, "pop")   the trips of pop_loop_sw -- loads, replay, commit rule, record stores, window inserts -- and nothing
//                       of the pool: a push that does not belong to its window is 16 bytes written to a ring in LDS;
//   wave 1 (R, "pool")  owns the HBM bucket pool and a sorted window of its own (the SAME SwWin / SwPool code: sw_add,
//                       sw_refill, sw_early_refill, respill): it drains the ring into them, keeps its window topped up from the
//                       buckets in the background (the gather + sort_runs + respill that cost the single wave ~750 clocks of
//                       every trip), and hands the head of its window over when P's runs low.
//
// Invariant (the one the single-wave window/pool pair already has, one level up): every entry held by R is at or above P's
// limit; every entry in P's window is below it.  P sends R exactly the pushes (and window evictions) at or above its limit;
// a hand-over gives P the m smallest entries R holds and moves P's limit to the smallest key R keeps (or to R's own limit).
// So P's window is always the global head of the open list, in the reference's order: the pop sequence is unchanged.
//
// Protocol (LDS, all counters monotone over the life of the kernel; one producer and one consumer each, no atomics):
//   ring[256] x (g, cell)   P writes entries then publishes TAIL; R reads TAIL, the entries, publishes HEAD.
//   take                    P writes REQ_TAIL / WANT then REQ = seq (at the END of a trip, when fewer than 7 heads will be
//                           left, so that R's answer is ready when the next trip starts); R, once it has consumed the ring
//                           up to REQ_TAIL (refilling first if its window is empty), writes up to WANT entries + the new
//                           limit + GIVEN, then ACK = seq.
//   search start / stop     P writes the goal and the first bucket, then CMD = seq << 2 | RUN; ... CMD = seq << 2 | STOP and
//                           waits for STOPACK = seq (R reports its spill count / overflow there); CMD = EXIT ends R.
// One wave's LDS instructions execute in order, so "data, then flag" needs no fence; the flags are read with relaxed atomic
// loads (never cached in a register) and every wait is bounded: a wait that runs out reports PF_ST_OVERFLOW, never a hang.
#pragma once

namespace pf {

#define PF_PR_RING_OFF 8192                 /* [8192, 12288): free while a search runs (pocket_flood's scratch before it) */
#define PF_PR_RINGF_OFF 4096                /* [4096, 6144): the entries' f, between the bucket counts and the sort's staging area */
#define PF_PR_CTL_OFF (PF_SX_OFF + 256)     /* 64 ints */
#define PF_PR_HAND_OFF (PF_PR_CTL_OFF + 256)   /* 64 x (f, g) then 64 x cell */
#define PF_PR_LDS_BYTES (PF_PR_HAND_OFF + 64 * 16 + 64 * 4)
static_assert(PF_PR_RING_OFF + PF_PR_RING_N * 16 <= PF_GEO_OFF, "ring must end below the replay table");
static_assert(PF_PR_RING_OFF >= PF_SORT_LDS + 2048, "ring must start above the sort's staging area");
static_assert(PF_PR_RINGF_OFF >= 4 * (PF_SW_NBK + 1) && PF_PR_RINGF_OFF + PF_PR_RING_N * 8 <= PF_SORT_LDS, "the f ring lies between the bucket counts and the staging area");

enum { PR_CMD = 0, PR_TAIL, PR_HEAD, PR_REQ, PR_REQ_TAIL, PR_WANT, PR_ACK, PR_GIVEN, PR_FLAGS, PR_STOPACK, PR_TR, PR_TC, PR_BCUR0,
       PR_HZERO, PR_SPILLS, PR_RSV, PR_LIM0 /* .. PR_LIM0 + 4: lf lo/hi, lg lo/hi, lc */ };
enum { PR_RUN = 1, PR_STOP = 2, PR_EXIT = 3 };
#ifndef PF_PR_SPIN_MAX
#define PF_PR_SPIN_MAX (1 << 22)
#endif
#ifndef PF_PR_TOPUP
#define PF_PR_TOPUP 40     /* R tops its window up from the buckets in the background while it holds fewer entries than this */
#endif

// (every lane reads the same word: tell the compiler the value is wave-uniform, so the waits are scalar loops and branches)
PF_DEV int pr_ld(const int* p) { return first_i(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
PF_DEV void pr_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

PF_DEV PrLink pr_link(char* lds) {
  PrLink L;
  L.ctl = (int*)(lds + PF_PR_CTL_OFF); L.ring = (PoolEnt*)(lds + PF_PR_RING_OFF); L.ring_f = (double*)(lds + PF_PR_RINGF_OFF);
  L.hand_fg = (SortFG*)(lds + PF_PR_HAND_OFF); L.hand_c = (int*)(lds + PF_PR_HAND_OFF + 64 * 16);
  L.tail = (unsigned)pr_ld(L.ctl + PR_TAIL); L.pub = L.tail; L.req = pr_ld(L.ctl + PR_REQ); L.sseq = pr_ld(L.ctl + PR_CMD) >> 2;
  L.pending = false;
  return L;
}
// workgroup start-up (wave 0, before the barrier that releases R)
PF_DEV void pr_init_ctl(char* lds, int lane) {
  int* ctl = (int*)(lds + PF_PR_CTL_OFF);
  ctl[lane] = 0;
}
PF_DEV void pr_publish(PrLink& L, int lane) {
  if (L.tail != L.pub) {
    PF_LDS_ORDER();
    if (lane == 0) pr_st(L.ctl + PR_TAIL, (int)L.tail);
    L.pub = L.tail;
  }
}
// room for a whole trip's pushes (<= 63 + evictions)?  false: R never caught up (reported as overflow by the caller)
PF_DEV bool pr_wait_room(PrLink& L) {
#pragma unroll 1
  for (int spin = 0; spin < PF_PR_SPIN_MAX; ++spin) {
    const unsigned head = (unsigned)pr_ld(L.ctl + PR_HEAD);
    if (L.tail - head <= PF_PR_RING_N - 128) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
PF_DEV void pr_search_start(PrLink& L, int tr, int tc, int bcur0, bool hzero, int lane) {
  L.sseq += 1; L.pending = false;
  if (lane == 0) {
    L.ctl[PR_TR] = tr; L.ctl[PR_TC] = tc; L.ctl[PR_BCUR0] = bcur0; L.ctl[PR_HZERO] = hzero ? 1 : 0;
    PF_LDS_ORDER();
    pr_st(L.ctl + PR_CMD, (L.sseq << 2) | PR_RUN);
  }
}
// returns false if R did not answer; adds R's spill count, reports R's overflow flag
PF_DEV bool pr_search_stop(PrLink& L, unsigned& spills, bool& overflow, int lane) {
  pr_publish(L, lane);
  if (lane == 0) pr_st(L.ctl + PR_CMD, (L.sseq << 2) | PR_STOP);
  bool ok = false;
#pragma unroll 1
  for (int spin = 0; spin < PF_PR_SPIN_MAX; ++spin) {
    if (pr_ld(L.ctl + PR_STOPACK) == L.sseq) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  PF_LDS_ORDER();
  spills += (unsigned)pr_ld(L.ctl + PR_SPILLS);
  overflow = !ok || (pr_ld(L.ctl + PR_FLAGS) & 1);
  return ok;
}
PF_DEV void pr_exit(char* lds, int lane) {
  if (lane == 0) pr_st((int*)(lds + PF_PR_CTL_OFF) + PR_CMD, PR_EXIT);
}
PF_DEV void pr_request(PrLink& L, int want, int lane) {
  pr_publish(L, lane);
  L.req += 1; L.pending = true;
  if (lane == 0) {
    L.ctl[PR_REQ_TAIL] = (int)L.tail; L.ctl[PR_WANT] = want;
    PF_LDS_ORDER();
    pr_st(L.ctl + PR_REQ, L.req);
  }
}
// Complete the pending take: P's remaining entries move to lanes [0, rem), R's follow.  Returns 0, or 3 (no answer / R overflowed).
PF_DEV int pr_take(PrLink& L, SwWin& W, int lane) {
  const int rem = W.wn - W.wp;
  if (W.wp > 0) {
    const int so = W.wp + lane < 64 ? W.wp + lane : 63;
    const double of_ = bperm_d(so, W.wf), og_ = bperm_d(so, W.wg); const int oc_ = bperm_i(so, W.wc);
    W.wf = of_; W.wg = og_; W.wc = oc_;
  }
  bool ok = false;
#pragma unroll 1
  for (int spin = 0; spin < PF_PR_SPIN_MAX; ++spin) {
    if (pr_ld(L.ctl + PR_ACK) == L.req) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  L.pending = false;
  if (!ok) return 3;
  PF_LDS_ORDER();
  const int m = pr_ld(L.ctl + PR_GIVEN);
  if (pr_ld(L.ctl + PR_FLAGS) & 1) return 3;
  if (lane >= rem) { W.wf = PF_INF; W.wg = 0.0; W.wc = 0; }
  if (lane >= rem && lane < rem + m) { const SortFG v = L.hand_fg[lane - rem]; W.wf = v.f; W.wg = v.g; W.wc = L.hand_c[lane - rem]; }
  W.wp = 0; W.wn = rem + m; W.n_pool -= m;
  W.lf = __hiloint2double(first_i(L.ctl[PR_LIM0 + 1]), first_i(L.ctl[PR_LIM0])); W.lg = __hiloint2double(first_i(L.ctl[PR_LIM0 + 3]), first_i(L.ctl[PR_LIM0 + 2]));
  W.lc = first_i(L.ctl[PR_LIM0 + 4]);
  PF_LDS_ORDER();
  return 0;
}

// ---- the pool wave ----
template <int VARIANT>
__device__ __noinline__ void pool_wave(char* lds, char* tier2_slot, const Rec* rec, int C, int lane) {
  constexpr int SEM = VARIANT == 1 ? 1 : 0;
  constexpr int NBK = PF_SW_NBK, CAP = PF_SW_CAP;
  int* ctl = (int*)(lds + PF_PR_CTL_OFF);
  const PoolEnt* ring = (const PoolEnt*)(lds + PF_PR_RING_OFF);
  const double* ring_f = (const double*)(lds + PF_PR_RINGF_OFF);
  SortFG* hand_fg = (SortFG*)(lds + PF_PR_HAND_OFF); int* hand_c = (int*)(lds + PF_PR_HAND_OFF + 64 * 16);
  Open O; O.lf = (double*)lds; O.sx = lds + PF_SX_OFF; O.of = (double*)tier2_slot;
  unsigned head = 0;
  int seen = 0;
  for (;;) {
    const int cmd = pr_ld(ctl + PR_CMD);
    if (cmd == PR_EXIT) return;
    if ((cmd & 3) != PR_RUN || (cmd >> 2) == seen) { __builtin_amdgcn_s_sleep(2); continue; }
    seen = cmd >> 2;
    PF_LDS_ORDER();
    // ---- a new search: empty pool, empty window, the limit at the first bucket boundary ----
    SwPool P;
    P.cnt = (int*)O.lf; P.be = (PoolEnt*)O.of; P.se = P.be + (NBK + 1) * CAP;
    P.tr = first_i(ctl[PR_TR]); P.tc = first_i(ctl[PR_TC]); P.hzero = first_i(ctl[PR_HZERO]) != 0;
    for (int k = lane; k <= NBK; k += 64) P.cnt[k] = 0;
    SwWin W;
    W.wf = PF_INF; W.wg = 0.0; W.wc = 0; W.wp = 0; W.wn = 0;
    W.bcur = first_i(ctl[PR_BCUR0]);
    W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0;
    W.n_pool = 0; W.n_spill = 0;
    unsigned spills = 0;
    int flags = 0;
    int take_done = pr_ld(ctl + PR_REQ);            // (a request left unanswered by the previous search is void)
    if (lane == 0) { ctl[PR_FLAGS] = 0; }
    PF_LDS_ORDER();
#ifdef PF_STAMPS
    unsigned long long rs[6] = {0, 0, 0, 0, 0, 0};   // clocks draining, entries drained, clocks in background refills, background refills, clocks serving, refills at serve time
#define PR_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#else
#define PR_T(v)
#endif
    int stop = 0;
    while (!stop) {
      // (CMD, TAIL, HEAD, REQ are one 16-byte line: ONE LDS read per poll)
      typedef int pr_i4 __attribute__((ext_vector_type(4)));
      const pr_i4 q4 = *(const volatile pr_i4*)ctl;
      // 1. drain the ring
      const unsigned tail = (unsigned)first_i(q4.y);
      if (head != tail) {
        PF_LDS_ORDER();
        PR_T(td0)
        const int m = (int)(tail - head) < 64 ? (int)(tail - head) : 64;
        double ef = 0.0, eg = 0.0; int ec = 0;
        if (lane < m) { const unsigned at = (head + (unsigned)lane) & (PF_PR_RING_N - 1); const PoolEnt v = ring[at]; eg = v.g; ec = v.c; ef = ring_f[at]; }
        PF_LDS_ORDER();
        head += (unsigned)m;
        if (lane == 0) pr_st(ctl + PR_HEAD, (int)head);     // (the entries are in registers: the slots are free again)
        // Entries at or above my limit go to their buckets all at once (one LDS atomic for the slot and one 16-byte store per
        // lane, as the single wave's push section does); the few below it are inserted into my window one by one.  An
        // eviction during those inserts only LOWERS the limit, so the lanes already sent to the pool stay correctly placed.
        const bool inw = lane < m && key_lt(ef, eg, ec, W.lf, W.lg, W.lc);
        const bool top = lane < m && !inw;
        {
          const int pba = (int)(ef * PF_SW_Q);
          const int pb = pba < W.bcur ? NBK : (pba & (NBK - 1));
          const bool inrange = pba - W.bcur < NBK;
          int pat = 0;
          if (top && inrange) pat = __hip_atomic_fetch_add(&P.cnt[pb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const bool fits = inrange && pat < CAP;
          if (top && fits) ent_put(P.be + pb * CAP + pat, eg, ec);
          const unsigned long long sm = __ballot(top && !fits);
          if (sm) {
            if (top && !fits) {
              if (inrange) __hip_atomic_fetch_add(&P.cnt[pb], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              const int at = W.n_spill + __builtin_popcountll(sm & ((1ull << lane) - 1ull));
              if (at < PF_SW_SPILL) ent_put(P.se + at, eg, ec);
            }
            W.n_spill += __builtin_popcountll(sm); spills += (unsigned)__builtin_popcountll(sm);
            if (W.n_spill > PF_SW_SPILL) flags |= 1;
          }
          W.n_pool += __builtin_popcountll(__ballot(top));
        }
        for (unsigned long long im = __ballot(inw); im && !(flags & 1); im &= im - 1) {
          const int k = __builtin_ctzll(im);
          const int ns0 = W.n_spill;
          if (!sw_add(P, W, bcast_d(ef, k), bcast_d(eg, k), bcast_i(ec, k), lane)) flags |= 1;
          spills += (unsigned)(W.n_spill > ns0 ? W.n_spill - ns0 : 0);
        }
        if (flags & 1) { if (lane == 0) pr_st(ctl + PR_FLAGS, flags); }
#ifdef PF_STAMPS
        rs[0] += __builtin_amdgcn_s_memtime() - td0; rs[1] += m;
#endif
        continue;
      }
      // 2. a take request, once the ring is consumed up to where P stood when it asked
      const int req = first_i(q4.w);
      if (req != take_done) {
        PF_LDS_ORDER();
        if ((unsigned)first_i(ctl[PR_REQ_TAIL]) != head) continue;
        PR_T(ts0)
        while (W.wp == W.wn && W.n_pool > 0 && !(flags & 1)) {
#ifdef PF_STAMPS
          rs[5] += 1;
#endif
          const int rr = sw_refill<SEM, false>(P, W, O, rec, C, lane);
          if (rr == 3) flags |= 1;
          if (rr == 1) break;
        }
        const int want = first_i(ctl[PR_WANT]);
        const int have = W.wn - W.wp;
        const int m = want < have ? want : have;
        if (lane >= W.wp && lane < W.wp + m) { SortFG v; v.f = W.wf; v.g = W.wg; hand_fg[lane - W.wp] = v; hand_c[lane - W.wp] = W.wc; }
        double nf = W.lf, ng = W.lg; int nc = W.lc;
        if (have > m) { nf = bcast_d(W.wf, W.wp + m); ng = bcast_d(W.wg, W.wp + m); nc = bcast_i(W.wc, W.wp + m); }
        W.wp += m;
        if (lane == 0) {
          ctl[PR_LIM0] = __double2loint(nf); ctl[PR_LIM0 + 1] = __double2hiint(nf);
          ctl[PR_LIM0 + 2] = __double2loint(ng); ctl[PR_LIM0 + 3] = __double2hiint(ng); ctl[PR_LIM0 + 4] = nc;
          ctl[PR_GIVEN] = m; ctl[PR_FLAGS] = flags;
          PF_LDS_ORDER();
          pr_st(ctl + PR_ACK, req);
        }
        take_done = req;
        PF_LDS_ORDER();
#ifdef PF_STAMPS
        rs[4] += __builtin_amdgcn_s_memtime() - ts0;
#endif
        continue;
      }
      // 3. idle: top the window up from the buckets (this is the work taken off P's trips), else look for STOP
      if (!(flags & 1) && W.n_pool > 0 && W.wn - W.wp < PF_PR_TOPUP) {
        const int before = W.n_pool;
        PR_T(tb0)
        if (W.wp == W.wn) { if (sw_refill<SEM, false>(P, W, O, rec, C, lane) == 3) flags |= 1; }
        else sw_early_refill<SEM>(P, W, O, rec, C, lane, PF_PR_TOPUP);
        if (flags & 1) { if (lane == 0) pr_st(ctl + PR_FLAGS, flags); }
#ifdef PF_STAMPS
        rs[2] += __builtin_amdgcn_s_memtime() - tb0; rs[3] += W.n_pool != before;
#endif
        if (W.n_pool != before) continue;             // progress: look at the ring again first
      }
      if (first_i(q4.x) != ((seen << 2) | PR_RUN)) stop = 1;
    }
#ifdef PF_STAMPS
    if (lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[16 + i], rs[i]);
#endif
    // ---- STOP (or EXIT): report, acknowledge ----
    // Entries P published but this search no longer needs are skipped: the next search starts from P's tail.  TAIL is final here
    // (P published it BEFORE it wrote CMD = STOP) and must be read -- and HEAD stored -- BEFORE the acknowledgement: STOPACK
    // releases P, which may start the next search, push to the ring and publish a new TAIL at once; a TAIL read after that would
    // skip the new search's first entries.
    head = (unsigned)pr_ld(ctl + PR_TAIL);
    if (lane == 0) {
      pr_st(ctl + PR_HEAD, (int)head);
      ctl[PR_SPILLS] = (int)spills; ctl[PR_FLAGS] = flags;
      PF_LDS_ORDER();
      pr_st(ctl + PR_STOPACK, seen);
    }
    PF_LDS_ORDER();
  }
}

}  // namespace pf
