// pf_settle.h -- the closed-set connector (AStarSolver.solve astar.py:33-101, DijkstraSolver.solve dijkstra.py:32-97)
// WITHOUT the sequential pop order: 64 nodes are expanded per trip, one per lane.
//
// Why this is exact (CPU study scripts/settle/).  Let g be the least fixpoint of
//     g(x) = min over expanded neighbours p of fl(g(p) + c(p, x)),   g(start) = 0,
// where "expanded" = key(p) < key(goal), key(x) = (fl(g(x) + h(x)), g(x), cell(x)).  Any label-correcting schedule
// reaches it (fl(. + c) is monotone).  Call x REGULAR if some neighbour p with fl(g(p) + c) == g(x) has key(p) < key(x).
// THEOREM: if every node with key below the goal's is regular, the reference's loop pops exactly those nodes, in key
// order, closes each with g(x), and came_from[x] is the regular parent with the smallest key.  (Induction over the pops:
// the k+1-th smallest key has its argmin parent among the first k, so it holds its final label and is the heap minimum;
// every other open node holds a label >= its final one, hence a larger key.)  Irregular nodes exist -- double rounding can
// give a node on a ray towards the goal a key one ulp BELOW its only parent's -- in 5-15 % of the searches; the pass below
// detects every one of them and the search is then handed to the sequential engine (pf_astar_sw.h).  Nothing is assumed.
//
// Mapping.  Labels live in a per-slot array of 64-bit words, epoch-coded so that (a) a new search never clears it and
// (b) ONE atomicMin both compares and updates: word = code << 57 | key57, code = 127 - epoch (newer searches have
// SMALLER codes, so any stale word loses against a fresh one), key57 = the label's bit pattern minus 1022 << 52 (labels
// are 0 or >= 1 and < 2^24: 57 bits, order preserving).  Avoid cells are written as key 0 with the current code: no
// relaxation can ever beat them, which is exactly "excluded from every neighbour list" (astar.py:51-56,80).  The open
// list is the bucket pool of the sequential engine (floor(64 f), 256 circular buckets in HBM, counts in LDS) used as a
// plain bag: order inside a bucket is irrelevant for a fixpoint.  Every successful relaxation is also logged in a
// touched list; afterwards one pass over it checks regularity and writes the parent move of every node, and the path is
// read off the parents.
#pragma once

namespace pf {

#define PF_ST_SEQ 5                 /* internal: not certified -> the caller runs the sequential engine */
#define PF_SETTLE_CAP 1024          /* entries per bucket (16 B each) */
// the band structure of this engine is its own (the pop loop's bucket width is tuned separately, pf_astar_sw.h)
#ifndef PF_ST_Q
#define PF_ST_Q 64.0
#endif
#define PF_ST_NBK 256
#define PF_LAB_KEYMASK ((1ull << 57) - 1ull)
#ifndef PF_ST_WIDE
#define PF_ST_WIDE 1                /* nodes per lane and trip (64 x this many nodes per trip) */
#endif
#define PF_ST_MARK_OFF 1024         /* byte offset in the wave's LDS of the band marks (64 x PF_ST_WIDE ints; the bucket counts end at 1024) */
#define PF_ST_WIN_OFF 2048          /* ... of the trip's winner list: PF_ST_WIN_CAP cells (int), then as many labels (double) */
#define PF_ST_WIN_CAP (PF_ST_WIDE == 1 ? 512 : 768)   /* 64 nodes x 8 moves can never overflow it; a wider build hands the search back if a trip ever does */
static_assert(PF_ST_MARK_OFF + 256 * PF_ST_WIDE <= PF_ST_WIN_OFF && PF_ST_WIN_OFF + 12 * PF_ST_WIN_CAP <= PF_GEO_OFF, "settle LDS layout");
static_assert(PF_ST_MARK_OFF >= 4 * PF_ST_NBK && PF_ST_MARK_OFF + 256 * PF_ST_WIDE <= PF_GEO_OFF, "band marks between the bucket counts and the replay table");
#define PF_PAR_IRREG 0x7Fu            /* par[x]: x is irregular (no earlier argmin parent) */
#define PF_PAR_SEEN 0x80u             /* par[x] bit 7: reached by the ancestor walk of this search */

PF_DEV unsigned long long lab_enc(double g, unsigned code) {
  const unsigned long long b = dbits(g);
  return ((unsigned long long)code << 57) | (b ? b - (1022ull << 52) : 0ull);
}
PF_DEV double lab_dec(unsigned long long v) {
  const unsigned long long k = v & PF_LAB_KEYMASK;
  return k ? __builtin_bit_cast(double, k + (1022ull << 52)) : 0.0;
}
PF_DEV int opposite_move(int m) { return m < 4 ? (m ^ 1) : 11 - m; }   // helper.py:30-36 order: 0<->1 2<->3 4<->7 5<->6

// Irregular nodes exist.  What the sequential loop does around one differs from the fixpoint only DOWNSTREAM of it: a node
// all of whose argmin parents (offer == label) behave as in the theorem receives its label from the same parent at the same
// point of the pop order (every other offer it ever gets is above its label: labels of the sequential loop are never below
// the fixpoint's), so by induction over the key order the theorem holds on the set of nodes none of whose ancestors in the
// argmin-parent DAG is irregular.  The goal's path only needs the goal to be in that set: walk the DAG backwards from the
// goal (g strictly falls along its edges) and hand the search back only if an irregular node is met.  DESIGN.md 4.3.
// par[x] = move from the smallest-key regular parent (0..7) or PF_PAR_IRREG, bit 7 = reached by this walk.
// Returns true when no ancestor of the goal is irregular.  noinline + scalar arguments: the walk is cold (2-7 % of the
// searches), and inlined its mere presence cost the sequential pop loop of the same kernel 15 % (k_decode_batch 108 -> 125 ms
// with the engine switched off: register allocation / code layout).
template <int VARIANT>
__device__ __attribute__((noinline)) bool cone_walk(const uint8_t* gmm, int C, uint64_t magicC, const unsigned long long* lab, unsigned char* par,
                                                    int* queue, int qcap, int start, int target, int tr, int tc, unsigned code, double F, int lane) {
  const unsigned long long blocked = (unsigned long long)code << 57;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if ((par[target] & 0x7Fu) == PF_PAR_IRREG) return false;               // the goal itself
  int qh = 0, qt = 1;
  if (lane == 0) { queue[0] = target; par[target] |= PF_PAR_SEEN; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  bool tainted = false;
  while (qh < qt) {
    const bool in = qh + lane < qt;
    const int x = in ? queue[qh + lane] : start;
    qh = qh + 64 < qt ? qh + 64 : qt;
    const double gx = lab_dec(lab[x]);
    const int r = (int)(((uint64_t)(uint32_t)x * magicC) >> 40), c = x - r * C;
    const unsigned mm = (in && x != start) ? gmm[x] : 0u;
    unsigned long long vp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { vp[k] = ~0ull; if ((mm >> k) & 1u) vp[k] = lab[x + move_dr(k) * C + move_dc(k)]; }
    // the eight claims of a node go out together (one memory round trip per round of the walk, not eight)
    unsigned old8[8]; bool arg8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int p = x + move_dr(k) * C + move_dc(k);
      const bool live = ((mm >> k) & 1u) && (vp[k] >> 57) == code && (vp[k] != blocked || p == start) && p != target;
      const double gp = lab_dec(vp[k]);
      const int pr = r + move_dr(k), pc = c + move_dc(k);
      long dr_ = pr - tr, dc_ = pc - tc;
      const double fp_ = VARIANT == 2 ? gp : gp + __builtin_sqrt((double)(dr_ * dr_ + dc_ * dc_));
      arg8[k] = live && gp + (k < 4 ? 1.0 : PF_SQRT2) == gx && (fp_ < F || fp_ == F) && p != start;   // an expanded argmin parent (earlier or not)
      old8[k] = 0u;
      if (arg8[k]) {
        // claim p: its byte of the aligned word (the other bytes belong to neighbouring cells)
        const unsigned long long ad = (unsigned long long)(par + p);
        old8[k] = __hip_atomic_fetch_or((unsigned*)(ad & ~3ull), (unsigned)PF_PAR_SEEN << (8u * (unsigned)(ad & 3ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int p = x + move_dr(k) * C + move_dc(k);
      const unsigned pv = (old8[k] >> (8u * (unsigned)((unsigned long long)(par + p) & 3ull))) & 0xFFu;
      if (arg8[k] && (pv & 0x7Fu) == PF_PAR_IRREG) tainted = true;    // an irregular ancestor
      const bool fresh = arg8[k] && !(pv & PF_PAR_SEEN);
      const unsigned long long fm = __ballot(fresh);
      if (fm) {
        const int at = qt + __builtin_popcountll(fm & ((1ull << lane) - 1ull));
        if (fresh) { if (at < qcap) queue[at] = p; else tainted = true; }
        qt += __builtin_popcountll(fm);
      }
    }
    if (__ballot(tainted)) return false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

template <int VARIANT>
__device__ __forceinline__ int settle_impl(const Grid& G, const Open& O, const SettleMem& M, int start, int target, int tr, int tc,
                                           const int* av_list, int av_n, int* out, int out_cap, int& out_n, AStat& st, int lane) {
  constexpr int NBK = PF_ST_NBK, CAP = PF_SETTLE_CAP;
  const int C = G.C, RC = G.R * G.C;
  int* cnt = (int*)O.lf;                                            // LDS [NBK] entries per bucket
  double* eg = O.of;                                                // HBM [NBK][CAP] label of the entry
  int* ec = (int*)(eg + (size_t)NBK * CAP);                         // HBM [NBK][CAP] its cell
  for (int k = lane; k < NBK; k += 64) cnt[k] = 0;
  // ---- epoch: newer searches use smaller codes; wipe the label array before the codes run out ----
  unsigned ep = M.epoch[0] + 1u;
  if (ep >= 127u) {
    for (int i = lane; i < RC; i += 64) M.lab[i] = ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ep = 1u;
  }
  if (lane == 0) M.epoch[0] = ep;
  const unsigned code = 127u - ep;
  const unsigned long long blocked = (unsigned long long)code << 57;          // key 0 under the current code
  for (int i = lane; i < av_n; i += 64) { const int c = av_list[i]; if (c != start && c != target) M.lab[c] = blocked; }   // astar.py:51-56
  if (lane == 0) M.lab[start] = blocked;                                       // g(start) = 0 (the same word: told apart by the cell)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  PF_LDS_ORDER();
  const int sr = row_of(G, start), sc = start - sr * C;
  long dr0 = sr - tr, dc0 = sc - tc;
  const double h0 = VARIANT == 2 ? 0.0 : __builtin_sqrt((double)(dr0 * dr0 + dc0 * dc0));
  int bcur = (int)(h0 * PF_ST_Q);                                    // first bucket (absolute) that may hold entries
  if (lane == 0) { eg[(size_t)(bcur & (NBK - 1)) * CAP] = 0.0; ec[(size_t)(bcur & (NBK - 1)) * CAP] = start; cnt[bcur & (NBK - 1)] = 1; }
  PF_LDS_ORDER();
  int nt = 0;                                                        // touched entries
  unsigned exp_l = 0, nbr_l = 0, push_l = 0;
  bool fail = false;
  // Latency is what a trip costs (one wave, dependent memory round trips), so each trip is three of them and no more:
  // (1) the entries, (2) ONE batch with the entry's own label, its move mask and all eight neighbour labels, (3) ONE batch
  // with the atomics that survive the pre-check.  The goal's label (the region bound F) is requested at the top and
  // consumed at the bottom: a bound that is one trip old only ever over-expands, which the certificate below catches.
  double F = PF_INF;
  int trips = 0;
  for (;;) {
    trips += 1;
    const unsigned long long vt = __hip_atomic_load(&M.lab[target], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (at L2, where the atomics land)
    // ---- next non-empty bucket ----
    PF_LDS_ORDER();
    if (cnt[bcur & (NBK - 1)] == 0) {
      int b0 = -1;
      for (int base = 0; base < NBK; base += 64) {
        const int c_ = cnt[(bcur + base + lane) & (NBK - 1)];
        const unsigned long long nz = __ballot(c_ > 0);
        if (nz) { b0 = bcur + base + __builtin_ctzll(nz); break; }
      }
      if (b0 < 0) break;                                             // open list exhausted
      bcur = b0;
    }
    if (F != PF_INF && (double)bcur > F * PF_ST_Q) break;            // every remaining entry has f above the goal's
    // ---- K entries per lane: the whole buckets from bcur on that fit 64 K slots (a fixpoint does not care about the order, and
    // one 1/64-wide band alone rarely holds that many nodes), or 64 K entries of the first one when it is larger.  A trip is
    // three dependent memory round trips whatever its width (r03: ~3/4 of its time on an idle chip), so K nodes per lane cost
    // K times the arithmetic but the same latency. ----
    constexpr int K = PF_ST_WIDE;
    int eidx[K];                                                     // my entries' indices in the pool
#pragma unroll
    for (int u = 0; u < K; ++u) eidx[u] = -1;
    {
      const int lim = F == PF_INF ? 0x7FFFFFFF : (int)(F * PF_ST_Q);  // last band that can hold a node of the region
      const int cb = bcur + lane <= lim ? cnt[(bcur + lane) & (NBK - 1)] : 0;   // lane k: size of the k-th band from bcur
      const int c0 = bcast_i(cb, 0);
      if (c0 > 64 * K) {
#pragma unroll
        for (int u = 0; u < K; ++u) eidx[u] = (bcur & (NBK - 1)) * CAP + c0 - 64 * K + 64 * u + lane;
        PF_LDS_ORDER();
        if (lane == 0) cnt[bcur & (NBK - 1)] = c0 - 64 * K;
      } else {
        const int incl = wave_incl_sum(cb);
        const int k = __builtin_popcountll(__ballot(incl <= 64 * K));   // (sizes are >= 0: the bands that fit are a prefix; k >= 1)
        const int total = bcast_i(incl, k - 1);
        int* mark = (int*)((char*)O.lf + PF_ST_MARK_OFF);               // [64 K]: where, among the taken entries, each band starts
#pragma unroll
        for (int u = 0; u < K; ++u) mark[64 * u + lane] = 0;
        PF_LDS_ORDER();
        if (lane < k && cb > 0) mark[incl - cb] = lane;
        PF_LDS_ORDER();
        int carry = 0;
#pragma unroll
        for (int u = 0; u < K; ++u) {
          int kk = wave_incl_max(mark[64 * u + lane]);                  // my entry's band, counted from bcur
          kk = kk > carry ? kk : carry;
          carry = bcast_i(kk, 63);
          const int j = 64 * u + lane - bperm_i(kk, incl - cb);         // ... and its index in that band
          if (64 * u + lane < total) eidx[u] = ((bcur + kk) & (NBK - 1)) * CAP + j;
        }
        PF_LDS_ORDER();
        if (lane < k) cnt[(bcur + lane) & (NBK - 1)] = 0;
      }
      PF_LDS_ORDER();
    }
    // (1) the entries (read where the stores of earlier trips landed: never a stale L1 line of a recycled slot)
    double g[K]; int cell[K]; bool have[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      g[u] = 0.0; cell[u] = start; have[u] = eidx[u] >= 0;
      if (have[u]) {
        g[u] = __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)eg + eidx[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        cell[u] = __hip_atomic_load(ec + eidx[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // (2) own label + move mask + the eight neighbour labels, one batch (addresses clamped: the move mask rejects what the clamp invents)
    unsigned long long own[K], vn[K][8]; unsigned mm[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      own[u] = 0; mm[u] = 0;
      if (have[u]) {
        own[u] = __hip_atomic_load(&M.lab[cell[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a live entry must never be mistaken for a superseded one
        mm[u] = G.mm[cell[u]];
#pragma unroll
        for (int k = 0; k < 8; ++k) { int n = cell[u] + move_dr(k) * C + move_dc(k); n = n < 0 ? 0 : (n >= RC ? RC - 1 : n); vn[u][k] = M.lab[n]; }
      }
    }
    int rr[K], cc[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      rr[u] = row_of(G, cell[u]); cc[u] = cell[u] - rr[u] * C;
      long hdr = rr[u] - tr, hdc = cc[u] - tc;
      const double f = VARIANT == 2 ? g[u] : g[u] + __builtin_sqrt((double)(hdr * hdr + hdc * hdc));
      have[u] = have[u] && cell[u] != target && own[u] == lab_enc(g[u], code) && !(own[u] == blocked && cell[u] != start) && f <= F;   // superseded / goal / outside the region
      if (have[u]) exp_l += 1;
    }
    // (3) the atomics, all in flight together; a label only ever falls, so a pre-check against the (possibly stale) loaded
    // value never drops a relaxation that would have won
    unsigned long long old[K][8];
    unsigned wonm[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = have[u] && ((mm[u] >> k) & 1u);
        const unsigned long long nv = lab_enc(g[u] + (k < 4 ? 1.0 : PF_SQRT2), code);      // astar.py:84-85
        old[u][k] = 0ull;
        if (ok) { nbr_l += 1; if (nv < vn[u][k]) old[u][k] = __hip_atomic_fetch_min(&M.lab[cell[u] + move_dr(k) * C + move_dc(k)], nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // :87
      }
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
      wonm[u] = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) wonm[u] |= (old[u][k] > lab_enc(g[u] + (k < 4 ? 1.0 : PF_SQRT2), code)) ? (1u << k) : 0u;
    }
    // ---- pushes of the winners (LDS slot atomics + fire-and-forget stores) and the touched log ----
    // A node wins 1-1.5 of its eight relaxations.  Instead of eight wave-wide rounds (one per move, each with its ballot, sqrt,
    // bucket atomic and stores whenever ANY lane won that move: ~60 instructions a round, half of the trip's arithmetic), the
    // winners are written to a list in LDS -- a lane appends its own, a prefix sum gives it its place -- and then handled one per
    // lane: ceil(winners / 64) rounds, usually one or two.
    {
      int wcnt = 0;
#pragma unroll
      for (int u = 0; u < K; ++u) wcnt += __builtin_popcount(wonm[u]);
      const int incl = wave_incl_sum(wcnt);
      const int total = bcast_i(incl, 63);
      int* ln = (int*)((char*)O.lf + PF_ST_WIN_OFF);
      double* lt = (double*)((char*)O.lf + PF_ST_WIN_OFF + 4 * PF_ST_WIN_CAP);
      if (total > PF_ST_WIN_CAP) { fail = true; break; }
      int wi = incl - wcnt;
#pragma unroll
      for (int u = 0; u < K; ++u)
        for (unsigned m = wonm[u]; m; m &= m - 1) {
          const int k = __builtin_ctz(m);
          ln[wi] = cell[u] + move_dr(k) * C + move_dc(k); lt[wi] = g[u] + (k < 4 ? 1.0 : PF_SQRT2);
          wi += 1;
        }
      PF_LDS_ORDER();
      for (int c0 = 0; c0 < total; c0 += 64) {
        const bool won = c0 + lane < total;
        int n = start; double t = 0.0;
        if (won) { n = ln[c0 + lane]; t = lt[c0 + lane]; }
        const int nr = row_of(G, n), nc = n - nr * C;
        long dr_ = nr - tr, dc_ = nc - tc;
        const double fn = VARIANT == 2 ? t : t + __builtin_sqrt((double)(dr_ * dr_ + dc_ * dc_));   // :90
        int ba = (int)(fn * PF_ST_Q); ba = ba < bcur ? bcur : ba;       // (an ulp below the current band: it is processed with it)
        if (won) {
          push_l += 1;
          const int at = __hip_atomic_fetch_add(&cnt[ba & (NBK - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (at >= CAP || ba - bcur >= NBK) fail = true;
          else { eg[(size_t)(ba & (NBK - 1)) * CAP + at] = t; ec[(size_t)(ba & (NBK - 1)) * CAP + at] = n; }
          const int tt = nt + c0 + lane;
          if (tt < M.touched_cap) M.touched[tt] = n; else fail = true;
        }
      }
      nt += total;
      PF_LDS_ORDER();
    }
    if (__ballot(fail)) { fail = true; break; }
    F = (vt >> 57) == code ? lab_dec(vt) : PF_INF;                   // (h(goal) = 0: f = g) -- the bound the NEXT trip works with
  }
  st.pops += (unsigned long long)wave_sum_i((int)exp_l); st.pushes += 1u + (unsigned)wave_sum_i((int)push_l);
  st.nbr += (unsigned)wave_sum_i((int)nbr_l);
  st.max_open = trips;                                                // (a search answered here reports its trips where the sequential loop reports its open-list high-water mark)
  if (fail) return PF_ST_SEQ;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  // the passes below read the labels the atomics left in L2
  const unsigned long long vt = M.lab[target];
  if ((vt >> 57) != code || vt == blocked) return 1;                 // the goal was never reached: astar.py:101 -> []
  F = lab_dec(vt);
  // ---- regularity + parents, one touched node per lane ----
  // par[x] = the move from its smallest-key regular parent (0..7), or PF_PAR_IRREG when x is irregular
  bool bad = false;
  for (int i0 = 0; i0 < nt; i0 += 64) {
    const bool in = i0 + lane < nt;
    const int x = in ? M.touched[i0 + lane] : start;
    const unsigned long long vx = M.lab[x];
    const double gx = lab_dec(vx);
    const int r = row_of(G, x), c = x - r * C;
    long hdr = r - tr, hdc = c - tc;
    const double fx = VARIANT == 2 ? gx : gx + __builtin_sqrt((double)(hdr * hdr + hdc * hdc));
    const bool reg = in && x != start && (x == target || fx < F || fx == F);   // key(x) < key(goal): g(x) < g(goal) when the f tie
    const unsigned mm = reg ? G.mm[x] : 0u;
    unsigned long long vp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { vp[k] = ~0ull; if ((mm >> k) & 1u) vp[k] = M.lab[x + move_dr(k) * C + move_dc(k)]; }
    int best = -1; double bf = 0.0, bg = 0.0; int bc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int p = x + move_dr(k) * C + move_dc(k);                // moves are symmetric: p -> x is the reverse of move k
      const bool live = ((mm >> k) & 1u) && (vp[k] >> 57) == code && (vp[k] != blocked || p == start) && p != target;
      const double gp = lab_dec(vp[k]);
      const bool arg = live && gp + (k < 4 ? 1.0 : PF_SQRT2) == gx;  // an argmin parent: its offer IS the label
      const int pr = r + move_dr(k), pc = c + move_dc(k);
      long dr_ = pr - tr, dc_ = pc - tc;
      const double fp_ = VARIANT == 2 ? gp : gp + __builtin_sqrt((double)(dr_ * dr_ + dc_ * dc_));
      const bool earlier = arg && (fp_ < F || fp_ == F) && (fp_ < fx || fp_ == fx);   // expanded, and popped before x (g(p) < g(x))
      if (earlier && (best < 0 || key_lt(fp_, gp, p, bf, bg, bc))) { best = k; bf = fp_; bg = gp; bc = p; }
    }
    if (reg) { if (best < 0) { bad = true; M.par[x] = PF_PAR_IRREG; } else M.par[x] = (unsigned char)opposite_move(best); }
  }
  if (__ballot(bad)) {
    // Irregular nodes exist: certify the goal's ancestor cone instead of the whole region (cone_walk below).
    if (!cone_walk<VARIANT>(G.mm, C, G.magicC, M.lab, M.par, M.touched, M.touched_cap, start, target, tr, tc, code, F, lane)) return PF_ST_SEQ;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // ---- the path: parents from the goal (astar.py:65-69), then reverse in place ----
  int n = 0, cell = target;
  const int guard = RC;
  while (cell != start) {
    if (n >= out_cap - 1 || n > guard) return 3;
    if (lane == 0) out[n] = cell;
    const int mv = M.par[cell] & 7;
    cell -= move_dr(mv) * C + move_dc(mv);
    n += 1;
  }
  if (lane == 0) out[n] = start;
  n += 1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = lane; i < n / 2; i += 64) { const int a = out[i], b = out[n - 1 - i]; out[i] = b; out[n - 1 - i] = a; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  out_n = n;
  return 0;
}

// The engine as seen by the search wrapper: a REAL call.  It runs once per search; inlined into the kernels, its code and
// registers were part of the function that also holds the sequential pop loop, the kernel's hot code, which paid for it in
// register allocation and layout.  Arguments are scalars (a struct passed by reference would be forced into scratch memory in
// the caller); the wave's LDS is reached through the dynamic-LDS symbol, which every kernel here places at offset 0.
// Returns status | out_n << 8; the work counters come back through the 3 words at PF_SX_OFF.
extern __shared__ __attribute__((aligned(16))) char pf_dyn_lds[];
template <int VARIANT>
__device__ __attribute__((noinline)) long long settle_call(const uint8_t* gmm, int R, int C, uint64_t magicC, char* pool, unsigned long long* lab,
                                                          int* touched, unsigned char* par, unsigned* epoch, int touched_cap, int start,
                                                          int target, int tr, int tc, const int* av_list, int av_n, int* out, int out_cap, int lane) {
  Grid G; G.occ = nullptr; G.mm = gmm; G.d2near = nullptr; G.comp = nullptr; G.R = R; G.C = C; G.magicC = magicC; G.step_cap = 0;
  Open O; O.lf = (double*)pf_dyn_lds; O.sx = pf_dyn_lds + PF_SX_OFF; O.of = (double*)pool;
  SettleMem M; M.lab = lab; M.touched = touched; M.par = par; M.epoch = epoch; M.touched_cap = touched_cap; M.astar_too = true;
  AStat st = {0, 0, 0, 0, 0, 0};
  int out_n = 0;
  const int rs = settle_impl<VARIANT>(G, O, M, start, target, tr, tc, av_list, av_n, out, out_cap, out_n, st, lane);
  PF_LDS_ORDER();
  if (lane == 0) { unsigned* w = (unsigned*)(pf_dyn_lds + PF_SX_OFF); w[0] = (unsigned)st.pops; w[1] = (unsigned)st.pushes; w[2] = (unsigned)st.nbr; w[3] = (unsigned)st.max_open; }
  PF_LDS_ORDER();
  return (long long)rs | ((long long)out_n << 8);
}
template <int VARIANT>
__device__ __forceinline__ int settle(const Grid& G, const Open& O, const SettleMem& M, int start, int target, int tr, int tc,
                                      const int* av_list, int av_n, int* out, int out_cap, int& out_n, AStat& st, int lane) {
  const long long r = settle_call<VARIANT>(G.mm, G.R, G.C, G.magicC, (char*)O.of, M.lab, M.touched, M.par, M.epoch, M.touched_cap, start, target,
                                           tr, tc, av_list, av_n, out, out_cap, lane);
  PF_LDS_ORDER();
  const unsigned* w = (const unsigned*)O.sx;
  st.pops += w[0]; st.pushes += w[1]; st.nbr += w[2];
  if ((int)(r & 0xFF) != PF_ST_SEQ && (int)w[3] > st.max_open) st.max_open = (int)w[3];
  PF_LDS_ORDER();
  out_n = (int)(r >> 8);
  return (int)(r & 0xFF);
}

}  // namespace pf
