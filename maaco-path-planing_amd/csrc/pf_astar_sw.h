// pf_astar_sw.h -- sorted-window pop loop of the one-wavefront A* (included by pf_astar.h).
//
// The open list is a monotone bucket queue with a sorted window (the round-1 designs it replaced -- lane-owned LDS
// bins with an argmin + rescan per pop, and four speculative pops over those bins -- are in the history):
//
//   pool    entries live unsorted in HBM buckets keyed by floor(f * Q) (Q = 256, 1024 circular buckets: a push is at
//           most 2*sqrt(2) above the pop that made it, 725 buckets); an append is one LDS atomic for the slot
//           index and ONE fire-and-forget 16-byte store of (g, cell) -- nothing waits for it;
//   window  when the window runs dry the next non-empty buckets (<= 64 entries) are loaded one entry per lane and
//           sorted on the full key (f, g, cell): the buckets already are in key order, so each entry only has to be
//           ranked inside its own bucket (sort_runs; a bitonic network over the lanes is kept for buckets larger than
//           the window); lane k then holds the k-th next pop.  A pop is five v_readlane; no reduction, no rescan;
//   limit   every pool entry is >= every window entry.  A push whose key is below the window's limit is
//           inserted in place (one ballot for the position, one wave shift); when the window is full its largest
//           entry goes back to the pool and becomes the limit;
//   decrease-key (VARIANT 0, astar.py:96-100) = push the new entry; the old one is recognised when popped (its g
//           no longer equals the record's g, or the cell is closed) and dropped without being counted -- the same
//           pop sequence as the reference's in-place update + heapify, whose order depends only on the keys.
// The bucket function is monotone in f and equal f share a bucket, so taking whole buckets in order and sorting
// them on (f, g, cell) reproduces the reference's total order exactly (tests compare paths and pop counts).
#pragma once

namespace pf {

// Lanes of the wave talk through LDS here.  The hardware executes one wave's LDS instructions in order, so no
// wait is needed, but the compiler must not move a lane's load above another lane's (to it unrelated) store:
// a compiler-only barrier at each hand-over point.
#define PF_LDS_ORDER() asm volatile("" ::: "memory")
PF_DEV unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }

// Bucket geometry.  r04: 1/256-wide buckets (1024 circular ones of 256 entries) halve the runs the refill has to order once more --
// A/B on one box against 1/128 x 512 x 512: mpa512 190.9 -> 194.1 k evals/s, astar1024 55.6 -> 57.9 k solves/s, ga512 equal (r02
// found no gain beyond 1/128, when the trip still carried what r04 took out of it).  PF_SW_NBK must cover 2*sqrt(2)*Q + 1 buckets
// and its counts must fit below the sort's staging area; a -DPF_TWO_WAVE build keeps the r03 geometry (its rings live between them).
#if defined(PF_TWO_WAVE) && !defined(PF_SW_Q)
#define PF_SW_Q 128.0
#define PF_SW_NBK 512
#define PF_SW_CAP 512
#endif
#ifndef PF_SW_Q
#define PF_SW_Q 256.0     /* buckets per unit of f (r01 64, r02 128: mpa512 160.1 -> 166.4 k evals/s) */
#endif
#ifndef PF_EARLY_REFILL
#define PF_EARLY_REFILL 1   /* 0: refill only when the window is empty (A/B builds: 123-127 k against 131.5 k evals/s) */
#endif
#ifndef PF_EARLY_BELOW
#define PF_EARLY_BELOW 7    /* the early refill runs when fewer live entries than this are left (7 = the heads of one trip; 14 measured 1 % slower) */
#endif
#ifndef PF_RUN_SORT
#define PF_RUN_SORT 1       /* 0: the bitonic network for every refill (A/B builds) */
#endif
#ifndef PF_SW_NBK
#define PF_SW_NBK 1024    /* circular buckets, a power of two (a stress build with a quarter of it sends far keys through the spill list) */
#endif
#ifndef PF_SW_CAP
#define PF_SW_CAP 256    /* entries per bucket (a stress build with -DPF_SW_CAP=8 drives everything through the spill list) */
#endif

PF_DEV bool key_lt(double f1, double g1, int c1, double f2, double g2, int c2) {   // branch-free (f, g, cell) order
  return (f1 < f2) | ((f1 == f2) & ((g1 < g2) | ((g1 == g2) & (c1 < c2))));
}
PF_DEV pf_u64 key_lt_m(double f1, double g1, int c1, double f2, double g2, int c2) {   // ... as a wave mask (B / PL: pf_device.h)
  return B(f1 < f2) | (B(f1 == f2) & (B(g1 < g2) | (B(g1 == g2) & B(c1 < c2))));
}
PF_DEV int bperm_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
PF_DEV double bperm_d(int src_lane, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// lane i <- lane i-1 (lane 0 keeps its value) / lane i <- lane i+1 (lane 63 keeps its value)
PF_DEV int wave_up_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }     // wave_shr:1
PF_DEV int wave_down_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false); }   // wave_shl:1
PF_DEV double wave_up_d(double v) { return __hiloint2double(wave_up_i(__double2hiint(v)), wave_up_i(__double2loint(v))); }
PF_DEV double wave_down_d(double v) { return __hiloint2double(wave_down_i(__double2hiint(v)), wave_down_i(__double2loint(v))); }

// value of lane ^ J: inside a 16-lane row by DPP (no LDS crossbar round trip), across rows by ds_bpermute
template <int J>
PF_DEV int xor_lane_i(int v, int lane) {
  if (J == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  if (J == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);        // quad_perm [2,3,0,1]
  if (J == 4) {
    const int up = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0xF, false);         // row_shl:4  lane i <- i + 4
    const int dn = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false);         // row_shr:4  lane i <- i - 4
    return (lane & 4) ? dn : up;
  }
  if (J == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);       // row_ror:8
  return __builtin_amdgcn_ds_bpermute((lane ^ J) << 2, v);
}
// one compare-exchange stage of a bitonic network: partner = lane ^ J, ascending where `up`
template <int J>
PF_DEV void cmpx(double& f, double& g, int& c, int lane, bool up) {
  const double pf_ = __hiloint2double(xor_lane_i<J>(__double2hiint(f), lane), xor_lane_i<J>(__double2loint(f), lane));
  const double pg_ = __hiloint2double(xor_lane_i<J>(__double2hiint(g), lane), xor_lane_i<J>(__double2loint(g), lane));
  const int pc_ = xor_lane_i<J>(c, lane);
  const bool lower = (lane & J) == 0;
  // live keys are distinct (one entry per (cell, g)), so "mine < partner's" is the complement of "partner's < mine";
  // between two identical padding entries (f = +inf) either answer leaves the lanes unchanged
  const bool take = key_lt(pf_, pg_, pc_, f, g, c) == (lower == up);
  if (take) { f = pf_; g = pg_; c = pc_; }
}
// bitonic merge of blocks of K lanes (compile-time strides: the lane masks fold to constants)
template <int K>
PF_DEV void merge_block(double& f, double& g, int& c, int lane) {
  const bool up = (lane & K) == 0;
  if (K >= 64) cmpx<32>(f, g, c, lane, up);
  if (K >= 32) cmpx<16>(f, g, c, lane, up);
  if (K >= 16) cmpx<8>(f, g, c, lane, up);
  if (K >= 8) cmpx<4>(f, g, c, lane, up);
  if (K >= 4) cmpx<2>(f, g, c, lane, up);
  cmpx<1>(f, g, c, lane, up);
}
// sort the first n2 (power of two) lanes ascending; unused lanes must hold f = +inf
PF_DEV void sort_lanes(double& f, double& g, int& c, int lane, int n2) {
  if (n2 >= 2) merge_block<2>(f, g, c, lane);
  if (n2 >= 4) merge_block<4>(f, g, c, lane);
  if (n2 >= 8) merge_block<8>(f, g, c, lane);
  if (n2 >= 16) merge_block<16>(f, g, c, lane);
  if (n2 >= 32) merge_block<32>(f, g, c, lane);
  if (n2 >= 64) merge_block<64>(f, g, c, lane);
}
// 64 lanes holding a bitonic sequence -> ascending
PF_DEV void merge_lanes(double& f, double& g, int& c, int lane) { merge_block<64>(f, g, c, lane); }

// Sort <= 64 entries that already come as RUNS in key order (one run per f bucket, the buckets ascending): an entry's
// place is the number of live entries in the earlier runs plus the number of smaller keys in its own run.  The entries are
// staged in LDS and every lane walks its own run (the lanes of a run read the same address: a broadcast) -- as many rounds
// as the longest run holds entries (3.5 buckets a refill, the longest 26 entries on average on G512), each two LDS reads
// and one key compare, against the 21 compare-exchange stages of the bitonic network, which move five dwords across the
// lanes each (measured: ~4 950 clocks per 64-lane sort, ~630 clocks per trip).  `start` = first lane of my run, `m` = its
// length (0 for a lane without an entry); lanes without a live entry (f = +inf: none, or dropped as superseded) end up
// behind the live ones, as the network leaves them.
// acc + ((f1, g1, c1) < (f2, g2, c2) in the lanes of `en`), for keys of real entries: f, g >= 0 or +inf and c >= 0, so the
// doubles order like their bit patterns and the key is one 160-bit unsigned number -- "<" is the borrow of a subtraction:
// five 32-bit subtracts instead of four fp64 compares, an integer compare and their mask logic.
PF_DEV int count_key_lt(int acc, double f1, double g1, int c1, double f2, double g2, int c2, unsigned long long en) {
  int t;
  asm volatile("v_sub_co_u32 %[t], vcc, %[c1], %[c2]\n\t"
               "v_subb_co_u32 %[t], vcc, %[gl1], %[gl2], vcc\n\t"
               "v_subb_co_u32 %[t], vcc, %[gh1], %[gh2], vcc\n\t"
               "v_subb_co_u32 %[t], vcc, %[fl1], %[fl2], vcc\n\t"
               "v_subb_co_u32 %[t], vcc, %[fh1], %[fh2], vcc\n\t"
               "s_and_b64 vcc, vcc, %[en]\n\t"
               "v_addc_co_u32 %[acc], vcc, 0, %[acc], vcc"
               : [t] "=&v"(t), [acc] "+v"(acc)
               : [c1] "v"(c1), [c2] "v"(c2), [gl1] "v"(__double2loint(g1)), [gl2] "v"(__double2loint(g2)), [gh1] "v"(__double2hiint(g1)),
                 [gh2] "v"(__double2hiint(g2)), [fl1] "v"(__double2loint(f1)), [fl2] "v"(__double2loint(f2)), [fh1] "v"(__double2hiint(f1)),
                 [fh2] "v"(__double2hiint(f2)), [en] "s"(en)
               : "vcc", "scc");                                  // (s_and_b64 writes SCC: a scalar compare must not be kept live across this)
  return acc;
}
#ifndef PF_SORT_UNROLL
#define PF_SORT_UNROLL 8
#endif
#ifndef PF_SORT_LDS
#define PF_SORT_LDS 6144    /* staging: 64 x 32 B in the wave's LDS, [6144, 8192): free while the pop loop runs (pf_astar.h) */
#endif
struct __attribute__((aligned(16))) SortFG { double f, g; };
PF_DEV void sort_runs(char* lds, double& f, double& g, int& c, int start, int m, int lane) {
  char* base = lds + PF_SORT_LDS;
  PF_LDS_ORDER();
  { SortFG v; v.f = f; v.g = g; *(SortFG*)(base + lane * 32) = v; *(int*)(base + lane * 32 + 16) = c; }
  PF_LDS_ORDER();
  int below = 0;
  const char* run = base + start * 32;
  // PF_SORT_UNROLL entries a round, read unconditionally (their LDS reads go out back to back; past my run's end they return
  // other runs' entries or bytes behind the staging area -- inside the wave's LDS, ignored by the t + u < m test)
  for (int t = 0;; t += PF_SORT_UNROLL) {
    const pf_u64 m0 = B(t < m);                                    // (the round's first mask is the loop test)
    if (!m0) break;
    SortFG e[PF_SORT_UNROLL]; int ec[PF_SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < PF_SORT_UNROLL; ++u) { e[u] = *(const SortFG*)(run + (t + u) * 32); ec[u] = *(const int*)(run + (t + u) * 32 + 16); }
#pragma unroll
    for (int u = 0; u < PF_SORT_UNROLL; ++u) below = count_key_lt(below, e[u].f, e[u].g, ec[u], f, g, c, u == 0 ? m0 : B(t + u < m));
  }
  PF_LDS_ORDER();
  const bool live = f != PF_INF;
  const unsigned long long lm = __ballot(live);
  const int nlive = __builtin_popcountll(lm);
  const int pos = live ? __builtin_popcountll(lm & ((1ull << start) - 1ull)) + below
                       : nlive + __builtin_popcountll(~lm & ((1ull << lane) - 1ull));
  // every lane sends its entry to lane `pos` (a permutation of the 64 lanes)
  const int a = pos << 2;
  const int flo = __builtin_amdgcn_ds_permute(a, __double2loint(f)), fhi = __builtin_amdgcn_ds_permute(a, __double2hiint(f));
  const int glo = __builtin_amdgcn_ds_permute(a, __double2loint(g)), ghi = __builtin_amdgcn_ds_permute(a, __double2hiint(g));
  c = __builtin_amdgcn_ds_permute(a, c);
  f = __hiloint2double(fhi, flo); g = __hiloint2double(ghi, glo);
}

// inclusive prefix sum / running maximum over the lanes (row scan by row_shr, then row_bcast:15 / :31)
PF_DEV int wave_incl_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return v;
}
PF_DEV int wave_incl_max(int v) {                                // v >= 0
  int t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
  return v;
}

// Source-lane table of the replay (see pop_loop_sw): [earlier head e][offset code of (my head - head e), 25 = far]
// [my sub] -> bpermute address of the lane of head e that handles my cell.
#define PF_GEO_OFF 12288                       /* byte offset in the wave's LDS, past pocket_flood's scratch */
struct GeoTab { unsigned char v[6 * 26 * 16]; };
constexpr GeoTab make_geo() {
  GeoTab t{};
  const int mdr[8] = {0, 0, 1, -1, 1, 1, -1, -1}, mdc[8] = {1, -1, 0, 0, 1, -1, 1, -1};   // helper.py:30-36
  for (int e = 0; e < 6; ++e)
    for (int code = 0; code < 26; ++code)
      for (int sub = 0; sub < 16; ++sub) {
        int v = 63 * 4;
        if (code < 25 && sub <= 8) {
          const int r = code / 5 - 2 + (sub < 8 ? mdr[sub] : 0), c = code % 5 - 2 + (sub < 8 ? mdc[sub] : 0);
          if (r == 0 && c == 0) v = e * 4 + 1;                   // head e itself: its self lane, "pop" flag
          else if (r >= -1 && r <= 1 && c >= -1 && c <= 1)
            for (int m = 0; m < 8; ++m) if (mdr[m] == r && mdc[m] == c) v = (7 + 8 * e + m) * 4;
        }
        t.v[(e * 26 + code) * 16 + sub] = (unsigned char)v;
      }
  return t;
}
static __device__ const GeoTab PF_GEO = make_geo();
PF_DEV void geo_to_lds(char* smem, int lane) {
  const unsigned* src = (const unsigned*)PF_GEO.v;
  unsigned* dst = (unsigned*)(smem + PF_GEO_OFF);
  for (int i = lane; i < (int)(sizeof(GeoTab) / 4); i += 64) dst[i] = src[i];
}

// lanes 0..48 as head pairs (e, h) = (lane / 7, lane % 7): the lanes with e < h
constexpr unsigned long long make_pairs_eh() { unsigned long long m = 0; for (int l = 0; l < 49; ++l) if (l / 7 < l % 7) m |= 1ull << l; return m; }
#define PF_SW_SPILL 16384
static_assert((double)PF_SW_NBK >= 2.8285 * PF_SW_Q + 1.0, "a push lies at most 2*sqrt(2) above the pop that made it: the circular range must cover that");
static_assert(4 * (PF_SW_NBK + 1) <= PF_SORT_LDS, "bucket counts and the sort's staging area share the wave's LDS");
static_assert(((size_t)(PF_SW_NBK + 1) * PF_SW_CAP + PF_SW_SPILL) * 16 <= (size_t)PF_POOL_STRIDE, "bucket pool + spill list fit the slot's HBM scratch");
// A pool entry is 16 bytes: (g, packed cell).  Its f is not stored: f = g + h(cell) is the very fp64 operation that
// produced it when the entry was pushed (astar.py:90 / MPA.py:140), so reading an entry back recomputes it bit for bit
// -- one 16-byte store per push and one 16-byte load per refilled entry instead of three scattered ones each
// (sector traffic: VERDICT r01 item 3).
struct __attribute__((aligned(16))) PoolEnt { double g; int c; int pad; };
struct SwPool {
  int* cnt;      // LDS [NBK + 1] entries per bucket; bucket NBK is the FRONT bucket: entries above the window's limit
                 // but below every regular bucket (window evictions, late low pushes of MPA._a_star's stale pops)
  PoolEnt* be;   // HBM [(NBK + 1)*CAP]
  PoolEnt* se;   // HBM [PF_SW_SPILL] spill list: entries whose bucket was full (a plateau of near-equal f on open maps)
                 //   or beyond the circular range; every refill offers them to the window / their bucket again
  int tr, tc;    // the goal (for h)
  bool hzero;    // Dijkstra: f = g
};
PF_DEV void ent_put(PoolEnt* e, double g, int c) { PoolEnt v; v.g = g; v.c = c; v.pad = 0; *e = v; }
PF_DEV void ent_get(const SwPool& P, const PoolEnt* e, double& f, double& g, int& c) {
  const PoolEnt v = *e;
  g = v.g; c = v.c;
  const int dr = (v.c >> 16) - P.tr, dc = (v.c & 0xFFFF) - P.tc;
  f = P.hzero ? v.g : v.g + __builtin_sqrt((double)(__mul24(dr, dr) + __mul24(dc, dc)));   // (|dr|, |dc| < 4096)
}
// append one entry (uniform values): to the front bucket when it sorts before every regular bucket (f below the
// boundary of bucket bcur), else to its f bucket, else (bucket full / out of range) to the spill list;
// false only when the spill list is full too
PF_DEV bool pool_put1(const SwPool& P, double f, double g, int c, int bcur, int& n_spill, int lane) {
  const int ba = (int)(f * PF_SW_Q);
  const int b = ba < bcur ? PF_SW_NBK : (ba & (PF_SW_NBK - 1));
  PF_LDS_ORDER();
  const int n = P.cnt[b];
  if (n >= PF_SW_CAP || ba - bcur >= PF_SW_NBK) {
    if (n_spill >= PF_SW_SPILL) return false;
    if (lane == 0) ent_put(P.se + n_spill, g, c);
    n_spill += 1;
    return true;
  }
  if (lane == 0) { ent_put(P.be + b * PF_SW_CAP + n, g, c); P.cnt[b] = n + 1; }
  PF_LDS_ORDER();
  return true;
}
// ---- two-wave mode (pf_astar_pr.h): the pop wave's end of the link to the pool wave.  Declared here because the window code
// below sends its evictions through it.
struct PrLink {
  int* ctl; PoolEnt* ring; double* ring_f; struct SortFG* hand_fg; int* hand_c;
  unsigned tail;       // entries written to the ring so far (uniform)
  unsigned pub;        // ... and published
  int req;             // last take request
  int sseq;            // search sequence number
  bool pending;        // a take request is out
};
#define PF_PR_RING_N 256
PF_DEV void pr_ring_put1(PrLink& L, double f, double g, int c, int lane) {   // one entry (uniform values); published with the trip's other pushes
  if (lane == 0) { ent_put(L.ring + (L.tail & (PF_PR_RING_N - 1)), g, c); L.ring_f[L.tail & (PF_PR_RING_N - 1)] = f; }
  L.tail += 1;
}
// the window (one entry per lane, sorted in [wp, wn)) and the bookkeeping that goes with it
struct SwWin {
  double wf, wg; int wc;        // this lane's entry
  int wp, wn;                   // live lanes
  double lf, lg; int lc;        // keys below this limit belong to the window; every pool entry is at or above it
  int bcur;                     // first regular bucket (absolute index) not yet taken
  int n_pool, n_spill;          // entries outside the window (spilled ones included) / in the spill list
};
// insert a key that is below the limit; a full window returns its largest entry to the pool, which becomes the limit
template <bool PR = false>
PF_DEV bool win_insert(const SwPool& P, SwWin& W, double kf, double kg, int kc, int lane, PrLink* L = nullptr) {
  const int p = W.wp + __builtin_popcountll(B(lane >= W.wp) & B(lane < W.wn) & key_lt_m(W.wf, W.wg, W.wc, kf, kg, kc));   // first live lane not below the key
  if (W.wn < 64) {
    const double sf = wave_up_d(W.wf), sg = wave_up_d(W.wg); const int sc = wave_up_i(W.wc);
    if (lane > p && lane <= W.wn) { W.wf = sf; W.wg = sg; W.wc = sc; }
    if (lane == p) { W.wf = kf; W.wg = kg; W.wc = kc; }
    W.wn += 1;
  } else if (W.wp > 0) {
    const double sf = wave_down_d(W.wf), sg = wave_down_d(W.wg); const int sc = wave_down_i(W.wc);
    if (lane >= W.wp - 1 && lane < p - 1) { W.wf = sf; W.wg = sg; W.wc = sc; }
    if (lane == p - 1) { W.wf = kf; W.wg = kg; W.wc = kc; }
    W.wp -= 1;
  } else {
    // 64 live entries: the largest key (the new one, or lane 63's) returns to the pool and becomes the limit
    double ef = kf, eg = kg; int ec = kc;
    if (p < 64) {
      ef = bcast_d(W.wf, 63); eg = bcast_d(W.wg, 63); ec = bcast_i(W.wc, 63);
      const double sf = wave_up_d(W.wf), sg = wave_up_d(W.wg); const int sc = wave_up_i(W.wc);
      if (lane > p) { W.wf = sf; W.wg = sg; W.wc = sc; }
      if (lane == p) { W.wf = kf; W.wg = kg; W.wc = kc; }
    }
    if (PR) pr_ring_put1(*L, ef, eg, ec, lane);                      // two-wave mode: the pool wave takes it
    else if (!pool_put1(P, ef, eg, ec, W.bcur, W.n_spill, lane)) return false;
    W.n_pool += 1;
    W.lf = ef; W.lg = eg; W.lc = ec;
  }
  return true;
}
// key below the limit -> window, else -> pool (front bucket / f bucket / spill list)
template <bool PR = false>
PF_DEV bool sw_add(const SwPool& P, SwWin& W, double kf, double kg, int kc, int lane, PrLink* L = nullptr) {
  if (key_lt(kf, kg, kc, W.lf, W.lg, W.lc)) return win_insert<PR>(P, W, kf, kg, kc, lane, L);
  if (PR) pr_ring_put1(*L, kf, kg, kc, lane);
  else if (!pool_put1(P, kf, kg, kc, W.bcur, W.n_spill, lane)) return false;
  W.n_pool += 1;
  return true;
}
// After a refill: offer every spilled entry to the window (key below the new limit) or to its bucket (room again,
// or inside the circular range now); what still does not fit stays spilled.
template <bool PLAT>
PF_DEV bool respill(const SwPool& P, SwWin& W, int lane) {
  const int n = W.n_spill;
  W.n_spill = 0; W.n_pool -= n;
  for (int base = 0; base < n; base += 64) {
    const int m = n - base < 64 ? n - base : 64;
    double ef = 0.0, eg = 0.0; int ec = 0;
    if (lane < m) ent_get(P, P.se + base + lane, ef, eg, ec);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       // read this chunk before re-spilled entries overwrite it
    if (PLAT) {
    // An entry that is neither below the limit nor offered room by its bucket stays spilled: on a plateau that is nearly
    // all of them, so they are put back 64 at a time; only the others go through sw_add one by one.
    PF_LDS_ORDER();
    const int ba = (int)(ef * PF_SW_Q);
    const bool inr = ba - W.bcur < PF_SW_NBK;
    const int bb = ba < W.bcur ? PF_SW_NBK : (ba & (PF_SW_NBK - 1));
    const bool stays = lane < m && !key_lt(ef, eg, ec, W.lf, W.lg, W.lc) && (!inr || P.cnt[bb] >= PF_SW_CAP);
    const unsigned long long sm = __ballot(stays);
    if (stays) {
      const int at = W.n_spill + __builtin_popcountll(sm & ((1ull << lane) - 1ull));   // (<= base + lane: in place)
      ent_put(P.se + at, eg, ec);
    }
    W.n_spill += __builtin_popcountll(sm); W.n_pool += __builtin_popcountll(sm);
    for (unsigned long long rest = __ballot(lane < m && !stays); rest; rest &= rest - 1) {
      const int k = __builtin_ctzll(rest);
      if (!sw_add(P, W, bcast_d(ef, k), bcast_d(eg, k), bcast_i(ec, k), lane)) return false;
    }
    } else {
    for (int k = 0; k < m; ++k)
      if (!sw_add(P, W, bcast_d(ef, k), bcast_d(eg, k), bcast_i(ec, k), lane)) return false;
    }
  }
  return true;
}
// A bucket larger than the window: leave its 64 smallest entries sorted in the lanes, compact the rest in place.
// The same for a bucket of many windows' worth (a plateau of equal f on an open map holds thousands of entries): the
// sort-and-merge pass above costs a full sort per 64 entries of the bucket, every refill.  Here a pivot is drawn from a
// sorted sample of the bucket, one cheap pass counts the keys below it (retried with a lower pivot while more than 64), a
// second pass moves those keys to LDS and compacts the others in place; the <= 64 selected keys ARE the bucket's smallest.
// Returns how many were taken (>= 1), sorted in the lanes; 0 = no usable pivot (the caller falls back).
// PLAT (template parameter of the pop loop): cheaper refills on plateaus of equal f (open maps): pivot selection for
// buckets of many windows' worth and bulk re-spilling.  Exact either way (empty 1024^2 closed-set batch 284 -> 169 ms).  Its
// mere presence costs the hot loop 3-4 % (register allocation / code layout), so it lives in SEPARATELY compiled kernels
// that only sparse maps are dispatched to (pathfit.hip: plateau_map); the MPA sweep never carries it.
#ifndef PF_SELECT_MIN
#define PF_SELECT_MIN 256   /* bucket size from which the pivot selection replaces the sort-and-merge pass */
#endif
#define PF_SEL_LDS PF_SORT_LDS   /* staging area of the pivot selection (64 x 20 B): the run sort's */
PF_DEV int take_smallest_select(const SwPool& P, char* lds, int bi, int c0, double& wf, double& wg, int& wc, int lane) {
  constexpr int CAP = PF_SW_CAP;
  double* sf = (double*)(lds + PF_SEL_LDS); double* sg = sf + 64; int* sc = (int*)(sg + 64);
  double pf_, pg_; int pc_;
  ent_get(P, P.be + bi * CAP + lane, pf_, pg_, pc_);                    // sample: the first 64 entries
  sort_lanes(pf_, pg_, pc_, lane, 64);
  int r = (64 * 64) / c0; r = r < 1 ? 1 : (r > 63 ? 63 : r);           // about 64 keys of the bucket lie below the sample's r-th
  double vf = 0.0, vg = 0.0; int vc = 0, cnt = 0;
  for (;;) {
    vf = bcast_d(pf_, r); vg = bcast_d(pg_, r); vc = bcast_i(pc_, r);
    cnt = 0;
    for (int rd = 0; rd < c0; rd += 64) {
      const bool in = rd + lane < c0;
      double ef = PF_INF, eg = 0.0; int ec = 0;
      if (in) ent_get(P, P.be + bi * CAP + rd + lane, ef, eg, ec);
      cnt += __builtin_popcountll(__ballot(in && key_lt(ef, eg, ec, vf, vg, vc)));
    }
    if (cnt <= 64) break;                                               // (cnt >= r >= 1: the sample keys below the pivot)
    if (r == 1) return 0;
    r >>= 1;
  }
  int wr = 0, sel = 0;
  for (int rd = 0; rd < c0; rd += 64) {
    const bool in = rd + lane < c0;
    double ef = PF_INF, eg = 0.0; int ec = 0;
    if (in) ent_get(P, P.be + bi * CAP + rd + lane, ef, eg, ec);
    const bool below = in && key_lt(ef, eg, ec, vf, vg, vc);
    const unsigned long long bm = __ballot(below), km = __ballot(in && !below);
    const unsigned long long lo = (1ull << lane) - 1ull;
    if (below) { const int at = sel + __builtin_popcountll(bm & lo); sf[at] = ef; sg[at] = eg; sc[at] = ec; }
    if (in && !below) { const int at = wr + __builtin_popcountll(km & lo); ent_put(P.be + bi * CAP + at, eg, ec); }
    sel += __builtin_popcountll(bm); wr += __builtin_popcountll(km);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // write-back before the next chunk is read
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  PF_LDS_ORDER();
  wf = PF_INF; wg = 0.0; wc = 0;
  if (lane < sel) { wf = sf[lane]; wg = sg[lane]; wc = sc[lane]; }
  PF_LDS_ORDER();
  int n2 = 1; while (n2 < sel) n2 <<= 1;
  sort_lanes(wf, wg, wc, lane, n2);
  if (lane == 0) P.cnt[bi] = wr;
  return sel;
}

PF_DEV void take_smallest64(const SwPool& P, int bi, int c0, double& wf, double& wg, int& wc, int lane) {
  constexpr int CAP = PF_SW_CAP;
  ent_get(P, P.be + bi * CAP + lane, wf, wg, wc);
  sort_lanes(wf, wg, wc, lane, 64);
  int wr = 0;
  for (int rd = 64; rd < c0; rd += 64) {
    const int m = c0 - rd < 64 ? c0 - rd : 64;
    double cf = PF_INF, cg = 0.0; int cc = 0;
    if (lane < m) ent_get(P, P.be + bi * CAP + rd + lane, cf, cg, cc);
    sort_lanes(cf, cg, cc, lane, 64);
    const double rf = bperm_d(63 - lane, cf), rg = bperm_d(63 - lane, cg);   // chunk reversed: window ++ reversed chunk is bitonic
    const int rc_ = bperm_i(63 - lane, cc);
    const bool sw = key_lt(rf, rg, rc_, wf, wg, wc);
    const double hf = sw ? wf : rf, hg = sw ? wg : rg;                       // the larger of each pair goes back
    const int hc = sw ? wc : rc_;
    if (sw) { wf = rf; wg = rg; wc = rc_; }
    merge_lanes(wf, wg, wc, lane);
    const unsigned long long fin = __ballot(hf != PF_INF);
    if (hf != PF_INF) {
      const int at = wr + __builtin_popcountll(fin & ((1ull << lane) - 1ull));
      ent_put(P.be + bi * CAP + at, hg, hc);
    }
    wr += __builtin_popcountll(fin);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                   // write-back before the next chunk is read
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  if (lane == 0) P.cnt[bi] = wr;
}

#ifdef PF_STAMPS
#define SW_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define SW_ACC(i, a, b) sw_acc[i] += (b) - (a);
#else
#define SW_T(var)
#define SW_ACC(i, a, b)
#endif
// VARIANT 0, at refill: an entry whose cell is closed or whose g is no longer the record's g was superseded by a
// decrease-key (astar.py:96-100 rewrites it in place) and can never be a pop of the reference -- g only falls and a
// closed cell stays closed, so it is dropped here instead of taking a head slot and a batch of loads later.
// Marks dropped lanes f = +inf (they sort to the end); returns the number of live entries.
PF_DEV int drop_superseded(const Rec* rec, int C, double& wf, double wg, int wc) {
  bool live = wf != PF_INF;
  if (live) {
    const Rec r = rec[(wc >> 16) * C + (wc & 0xFFFF)];
    if ((r.meta & PF_M_CLOSED) || r.g != wg) { wf = PF_INF; live = false; }
  }
  return __builtin_popcountll(B(wf != PF_INF));
}

// ---- early refill: fewer than `below` entries left in the window.  With the front bucket and the spill list empty, every pool
// entry lies in a regular bucket at or above the limit, so the next whole buckets, sorted, simply continue the
// window (its remaining keys are below the limit): the trips keep seven heads instead of running the window dry.
template <int SEM>
PF_DEV void sw_early_refill(const SwPool& P, SwWin& W, const Open& O, const Rec* rec, int C, int lane, int below) {
  constexpr int NBK = PF_SW_NBK, CAP = PF_SW_CAP;
  if (W.wn - W.wp > 0 && W.wn - W.wp < below && W.n_pool > 0 && W.n_spill == 0) {
    const int rem = W.wn - W.wp;
    const int c_ = P.cnt[(W.bcur + lane) & (NBK - 1)];
    const unsigned long long nz = __ballot(c_ > 0);
    if (nz && P.cnt[NBK] == 0) {
      const int b0 = W.bcur + __builtin_ctzll(nz);
      const int cb = P.cnt[(b0 + lane) & (NBK - 1)];
      const int incl = wave_incl_sum(cb);
      const int k = __builtin_popcountll(__ballot(incl <= 64 - rem));
      if (k > 0) {
        const int total = bcast_i(incl, k - 1);
        int* mark = (int*)O.sx;
        mark[lane] = 0;
        PF_LDS_ORDER();
        if (lane < k && cb > 0) mark[incl - cb] = lane;
        PF_LDS_ORDER();
        const int kk = wave_incl_max(mark[lane]);
        const int j = lane - bperm_i(kk, incl - cb);
        double nf = PF_INF, ng = 0.0; int nc = 0;
        if (lane < total) {
          const int bi = (b0 + kk) & (NBK - 1);
          ent_get(P, P.be + bi * CAP + j, nf, ng, nc);
        }
        if (lane < k) P.cnt[(b0 + lane) & (NBK - 1)] = 0;
        int live = total;
        if (SEM == 0) live = drop_superseded(rec, C, nf, ng, nc);
#if !PF_RUN_SORT
        int n2 = 1; while (n2 < total) n2 <<= 1;
#endif
#if PF_RUN_SORT
        {
          const int msz = bperm_i(kk, cb);                      // (every lane takes part: a masked-off source lane would read as 0)
          sort_runs((char*)O.lf, nf, ng, nc, lane < total ? lane - j : 0, lane < total ? msz : 0, lane);
        }
#else
        sort_lanes(nf, ng, nc, lane, n2);
#endif
        // lanes 0..rem-1 <- the old window, then the new entries
        const int so = W.wp + lane < 64 ? W.wp + lane : 63, sn = lane >= rem ? lane - rem : 0;
        const double of_ = bperm_d(so, W.wf), og_ = bperm_d(so, W.wg); const int oc_ = bperm_i(so, W.wc);
        const double mf_ = bperm_d(sn, nf), mg_ = bperm_d(sn, ng); const int mc_ = bperm_i(sn, nc);
        W.wf = lane < rem ? of_ : mf_; W.wg = lane < rem ? og_ : mg_; W.wc = lane < rem ? oc_ : mc_;
        W.wp = 0; W.wn = rem + live; W.n_pool -= total;
        W.bcur = b0 + k;
        W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0;
        PF_LDS_ORDER();
      }
    }
  }
}

// ---- refill of an EMPTY window: the front bucket if it holds anything, else the next non-empty buckets (<= 64 entries); sorted.
// Returns 0 (the window holds entries again), 1 (nothing is left anywhere: the search has failed), 3 (scratch overflow) or
// 4 (everything taken was superseded / only the spill list was re-offered: call again).
template <int SEM, bool PLAT>
PF_DEV int sw_refill(const SwPool& P, SwWin& W, const Open& O, const Rec* rec, int C, int lane) {
  constexpr int NBK = PF_SW_NBK, CAP = PF_SW_CAP;
  {
    if (W.n_pool == 0) return 1;
    const int cF = P.cnt[NBK];
    W.wf = PF_INF; W.wg = 0.0; W.wc = 0;
    if (cF > 0) {
      if (cF <= 64) {
        if (lane < cF) ent_get(P, P.be + NBK * CAP + lane, W.wf, W.wg, W.wc);
        if (lane == 0) P.cnt[NBK] = 0;
        int live = cF;
        if (SEM == 0) live = drop_superseded(rec, C, W.wf, W.wg, W.wc);
#if PF_RUN_SORT
        sort_runs((char*)O.lf, W.wf, W.wg, W.wc, 0, lane < cF ? cF : 0, lane);      // one run: the front bucket
#else
        int n2 = 1; while (n2 < cF) n2 <<= 1;
        sort_lanes(W.wf, W.wg, W.wc, lane, n2);
#endif
        W.wp = 0; W.wn = live; W.n_pool -= cF;
        W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0;
      } else {
        int nt = PLAT && cF >= PF_SELECT_MIN ? take_smallest_select(P, (char*)O.lf, NBK, cF, W.wf, W.wg, W.wc, lane) : 0;
        if (nt == 0) { take_smallest64(P, NBK, cF, W.wf, W.wg, W.wc, lane); nt = 64; }
        W.wp = 0; W.wn = nt; W.n_pool -= nt;
        W.lf = bcast_d(W.wf, nt - 1); W.lg = bcast_d(W.wg, nt - 1); W.lc = bcast_i(W.wc, nt - 1);   // the rest of the front bucket is above this key
      }
    } else {
      int b0 = -1;
      for (int base = 0; base < NBK; base += 64) {
        const int c_ = P.cnt[(W.bcur + base + lane) & (NBK - 1)];
        const unsigned long long nz = __ballot(c_ > 0);
        if (nz) { b0 = W.bcur + base + __builtin_ctzll(nz); break; }
      }
      if (b0 < 0) {                                  // only spilled entries are left (their buckets emptied since)
        if (W.n_spill == 0) return 3;   // (n_pool > 0 with nothing anywhere: never loop silently)
        // window, front bucket and every regular bucket are empty: the circular range may start at the smallest
        // spilled key, which brings the entries that were beyond it back in range
        unsigned minb = 0xFFFFFFFFu;
        for (int base = 0; base < W.n_spill; base += 64)
          if (base + lane < W.n_spill) { double f_, g_; int c_; ent_get(P, P.se + base + lane, f_, g_, c_); const unsigned b_ = (unsigned)(int)(f_ * PF_SW_Q); minb = b_ < minb ? b_ : minb; }
        minb = wave_min_u32(minb);
        if ((int)minb > W.bcur) { W.bcur = (int)minb; W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0; }
        if (!respill<PLAT>(P, W, lane)) return 3;
        return 4;
      }
      const int cb = P.cnt[(b0 + lane) & (NBK - 1)]; // lane k: size of the k-th bucket from b0 (wraps onto empty ones)
      const int c0 = bcast_i(cb, 0);
      if (c0 <= 64) {
        // take buckets b0 .. b0+k-1, as many as fit the window: lane j loads the j-th entry of their concatenation
        // (prefix sums of the sizes; each bucket marks where it starts, a running maximum spreads the mark)
        const int incl = wave_incl_sum(cb);
        const int k = __builtin_popcountll(__ballot(incl <= 64));    // (sizes are >= 0: the lanes that fit are a prefix)
        const int total = bcast_i(incl, k - 1);
        int* mark = (int*)O.sx;
        mark[lane] = 0;
        PF_LDS_ORDER();
        if (lane < k && cb > 0) mark[incl - cb] = lane;
        PF_LDS_ORDER();
        const int kk = wave_incl_max(mark[lane]);                      // my entry's bucket, counted from b0
        const int j = lane - bperm_i(kk, incl - cb);                    // ... and its index in that bucket
        if (lane < total) {
          const int bi = (b0 + kk) & (NBK - 1);
          ent_get(P, P.be + bi * CAP + j, W.wf, W.wg, W.wc);
        }
        if (lane < k) P.cnt[(b0 + lane) & (NBK - 1)] = 0;
        int live = total;
        if (SEM == 0) live = drop_superseded(rec, C, W.wf, W.wg, W.wc);
#if PF_RUN_SORT
        {
          const int msz = bperm_i(kk, cb);                      // (every lane takes part: a masked-off source lane would read as 0)
          sort_runs((char*)O.lf, W.wf, W.wg, W.wc, lane < total ? lane - j : 0, lane < total ? msz : 0, lane);
        }
#else
        int n2 = 1; while (n2 < total) n2 <<= 1;
        sort_lanes(W.wf, W.wg, W.wc, lane, n2);
#endif
        W.wp = 0; W.wn = live; W.n_pool -= total;
        W.bcur = b0 + k;
        W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0;
      } else {
        int nt = PLAT && c0 >= PF_SELECT_MIN ? take_smallest_select(P, (char*)O.lf, b0 & (NBK - 1), c0, W.wf, W.wg, W.wc, lane) : 0;
        if (nt == 0) { take_smallest64(P, b0 & (NBK - 1), c0, W.wf, W.wg, W.wc, lane); nt = 64; }
        W.wp = 0; W.wn = nt; W.n_pool -= nt;
        W.bcur = b0;
        W.lf = bcast_d(W.wf, nt - 1); W.lg = bcast_d(W.wg, nt - 1); W.lc = bcast_i(W.wc, nt - 1);   // the rest of the bucket is above the window's last key
      }
    }
    PF_LDS_ORDER();
    if (W.n_spill > 0 && !respill<PLAT>(P, W, lane)) return 3;
    if (W.wn == 0) return 4;                                    // everything taken was superseded: take the next buckets
  }
  return 0;
}

// forward declarations of the pop wave's side of the two-wave protocol (pf_astar_pr.h)
PF_DEV void pr_search_start(PrLink& L, int tr, int tc, int bcur0, bool hzero, int lane);
PF_DEV bool pr_search_stop(PrLink& L, unsigned& spills, bool& overflow, int lane);
PF_DEV void pr_request(PrLink& L, int want, int lane);
PF_DEV int pr_take(PrLink& L, SwWin& W, int lane);
PF_DEV void pr_publish(PrLink& L, int lane);
PF_DEV bool pr_wait_room(PrLink& L);
PF_DEV int pr_ld(const int* p);

template <int VARIANT, bool PLAT, bool PR = false>
__device__ __forceinline__ int pop_loop_sw(const Grid& G, Rec* rec, const Open& O, uint32_t tag, uint32_t avm, int start, int target,
                                           int tr, int tc, int max_steps, double h0, int src, AStat& st, int lane, PrLink* L = nullptr) {
  constexpr int SEM = VARIANT == 1 ? 1 : 0;                   // 0: closed set + decrease-key (A*, Dijkstra), 1: MPA._a_star
  constexpr int NBK = PF_SW_NBK, CAP = PF_SW_CAP;
  const int C = G.C, RC = G.R * G.C;
  SwPool P;
  P.cnt = (int*)O.lf; P.be = (PoolEnt*)O.of; P.se = P.be + (NBK + 1) * CAP;
  P.tr = tr; P.tc = tc; P.hzero = VARIANT == 2;
  if (!PR) { for (int k = lane; k <= NBK; k += 64) P.cnt[k] = 0; }   // (two-wave mode: the pool wave owns the buckets)
  PF_LDS_ORDER();

  SwWin W;                                                   // the window: lane k in [wp, wn) holds the (k - wp)-th next pop
  W.wf = PF_INF; W.wg = 0.0; W.wc = 0;
  if (lane == 0) { W.wf = h0; W.wg = 0.0; W.wc = src; }
  W.wp = 0; W.wn = 1;
  W.bcur = (int)(h0 * PF_SW_Q) + 1;                           // first bucket (absolute index) not yet taken into the window
  W.lf = (double)W.bcur * (1.0 / PF_SW_Q); W.lg = -PF_INF; W.lc = 0;   // keys below (lf, lg, lc) belong to the window
  W.n_pool = 0; W.n_spill = 0;
  if (PR) pr_search_start(*L, tr, tc, W.bcur, VARIANT == 2, lane);
  int steps = 0, status = 1;
  unsigned nbr_s = 0, push_s = 0, dk_s = 0;                  // event counts of the wave (uniform)
  int n_max = 1;
#ifdef PF_TRIPS
  unsigned tr_short = 0, tr_full = 0, tr_viol = 0, tr_near = 0, tr_pot = 0;
#endif

  // per-lane roles: lane h < 7 reads head h's own record (sub = 8: the "self" lane, so ballots over the self lanes are
  // 7-bit head masks as they stand); lanes 7 + 8h .. 14 + 8h relax move `sub` of head h; lane 63 is idle (group 7)
  const int grp0 = lane < 7 ? lane : (lane - 7) >> 3, sub0 = lane < 7 ? 8 : ((lane - 7) & 7);
  const int d = sub0 & 7;
  const int ddr = move_dr(d), ddc = move_dc(d);
  const int doff = sub0 < 8 ? ddr * C + ddc : 0;
  const double cost = d < 4 ? 1.0 : PF_SQRT2;
  const int trc = (tr << 16) | tc;

#ifdef PF_STAMPS
  unsigned long long sw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long sw_er[6] = {0, 0, 0, 0, 0, 0};           // early refills: count, entries, largest run, buckets, clocks waiting for the entries, clocks sorting
#endif
  for (;;) {
    SW_T(t0)
    unsigned rhead = 0;
    if (PR) {
      // two-wave mode: the window is refilled from the pool wave's window (requested at the end of the previous trip)
      if (!L->pending && W.wp == W.wn) {
        if (W.n_pool == 0) { status = 1; break; }
        pr_request(*L, 64, lane);
#ifdef PF_STAMPS
        sw_cnt[3] += 1;
#endif
      }
      if (L->pending) {
#ifdef PF_STAMPS
        const int wn0_ = W.wn - W.wp; const unsigned long long tq0_ = __builtin_amdgcn_s_memtime();
#endif
        if (pr_take(*L, W, lane) != 0) { status = 3; break; }
#ifdef PF_STAMPS
        sw_cnt[0] += 1; sw_cnt[1] += (W.wn - W.wp) - wn0_; sw_cnt[2] += __builtin_amdgcn_s_memtime() - tq0_;
#endif
        if (W.wp == W.wn) { status = W.n_pool == 0 ? 1 : 3; break; }
      }
      rhead = (unsigned)pr_ld(L->ctl + 2 /* PR_HEAD */);        // (used by the push section: its latency is covered by the trip)
    } else {
      if (PF_EARLY_REFILL) sw_early_refill<SEM>(P, W, O, rec, C, lane, PF_EARLY_BELOW);
      if (W.wp == W.wn) {
        const int rr_ = sw_refill<SEM, PLAT>(P, W, O, rec, C, lane);
        if (rr_ == 1) { status = 1; break; }
        if (rr_ == 3) { status = 3; break; }
        if (rr_ == 4) continue;
      }
    }
    SW_T(t1)
    // ---- pop: up to seven heads of the window at once, nine lanes each ----
    // The window is sorted, so the next pops are known.  Lane group h relaxes head h in registers; the groups replay
    // each other's effects on the cells they share (below), and head h takes effect (and counts as a pop) iff every
    // earlier head did, no earlier head pushed a key at or below head h's f, no earlier head improved head h's own
    // cell, and no earlier head was the target: exactly the pops, in order, that the sequential loop would make.  A
    // head that does not qualify simply stays in the window.
    constexpr int NH = 7;
    static_assert(NH == 7, "the lane <-> (group, sub) and lane <-> (head pair) maps below are written for 7 x 9 lanes");
    const int nh = W.wn - W.wp < NH ? W.wn - W.wp : NH;
    // The lane-role predicates (grp == h, sub < 8, ...) are recomputed where they are used (one v_cmp each): as
    // loop invariants they sat in ~25 SGPR pairs, overflowed the scalar register file and came back through two
    // v_readlane each, >100 instructions per trip.
    int grp = grp0, sub = sub0, lane_t = lane;
    asm volatile("" : "+v"(grp), "+v"(sub), "+v"(lane_t));
    // my group's head cell: ONE crossbar permute from the window lane that holds it (seven v_readlane + six selects -- ~40 wave
    // instructions with their scalar index arithmetic -- bought the load addresses ~70 clocks of LDS latency at 4x the issue slots)
    const int hsrc = W.wp + grp < 64 ? W.wp + grp : 63;            // my group's head lives in this window lane
    const int prc = bperm_i(hsrc, W.wc);                        // (r << 16 | c)
    const pf_u64 mHave = B(grp < nh);                           // (lane 63 is group 7: never; as a wave mask, see below)
    const bool have = PL(mHave);
    const int pr = prc >> 16, pc = prc & 0xFFFF;
    const int cur = pr * C + pc;
    // ---- one batch of loads: 8 neighbour records, the cell's own record, its move mask (and g, MPA variant) ----
    int nidx = cur + doff;
    nidx = nidx < 0 ? 0 : (nidx >= RC ? RC - 1 : nidx);        // the move mask rejects what the clamp invents
    Rec rn; rn.g = 0.0; rn.tagmm = 0; rn.meta = 0;
    double cur_g = 0.0;
    // ONE vector-memory instruction per trip: the head's move mask is the low byte of its own record (every record carries its
    // cell's static mask: k_slot_init, and every store keeps it), which the self lane loads anyway -- the separate byte load from
    // the mask table was a second fully divergent access (address processing for 63 lanes, seven more sectors) for nothing.
    if (have) rn = rec[nidx];                                   // (MPA variant: g_score[current] comes from the self lane's record too)
#if defined(PF_STAMPS) && defined(PF_WAIT_EARLY)
    { SW_T(ti_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SW_T(tj_) sw_cnt[0] += tj_ - ti_; }   // diagnostic: the bare load latency
#endif
    // -- everything below is in the shadow of the loads --
    const double pg = bperm_d(hsrc, W.wg);
    const unsigned long long f_last = dbits(bcast_d(W.wf, W.wp + nh - 1));   // f of the trip's last head (uniform) for the push test below
    // lanes 0..48 look at the head pair (e, h) = (lane / 7, lane % 7): too close to be independent?
    const int pe = (lane_t * 37) >> 8, ph = lane_t - 7 * pe;
    const int rce = bperm_i(W.wp + pe < 64 ? W.wp + pe : 63, W.wc), rch = bperm_i(W.wp + ph < 64 ? W.wp + ph : 63, W.wc);
    const int nr = pr + ddr, nc = pc + ddc;
    const int hdr = nr - tr, hdc = nc - tc;                    // |.| < 2^15: the squares fit 32 bits
    double hn = VARIANT == 2 ? 0.0 : __builtin_sqrt((double)(__mul24(hdr, hdr) + __mul24(hdc, hdc)));   // astar.py:90 / MPA.py:140 / dijkstra.py:89 (24-bit multiplies run at full rate)
    asm volatile("" : "+v"(hn));                               // computed in the shadow of the loads
    // Heads within 2 cells of each other touch common records.  Instead of stopping the trip there, every lane
    // replays, in head order, what the earlier heads of this trip do to ITS cell: for an earlier head e the lane that
    // handles the same cell is fixed by geometry (cell - head e in [-1,1]^2 picks e's move lane, or e's self lane = the
    // pop of that cell).  The pair lanes (e, h) = (lane / 7, lane % 7) classify head h - head e once (25 offsets +
    // "far"); each lane then reads its source lane from a table in LDS, [e][offset][sub] -> bpermute address (bit 0:
    // head e IS my cell; 63*4: none).  All in the shadow of the loads.
    const int Dr_ = (rch >> 16) - (rce >> 16), Dc_ = (rch & 0xFFFF) - (rce & 0xFFFF);
    // (predicates as wave masks from here on: B(compare) / PL(mask), pf_device.h -- every ballot below is a scalar AND)
    constexpr pf_u64 PAIRS_EH = make_pairs_eh();                // lane < 49 && e < h
    const pf_u64 nearg = PAIRS_EH & B(ph < nh) & B((unsigned)(Dr_ + 2) <= 4u) & B((unsigned)(Dc_ + 2) <= 4u);   // bit 7e+h
    const int pcode = PL(nearg) ? (Dr_ + 2) * 5 + (Dc_ + 2) : 25;
    const unsigned char* geo = (const unsigned char*)O.lf + PF_GEO_OFF;
    int fsrc[NH - 1];
    {
#pragma unroll
      for (int e = 0; e < NH - 1; ++e) fsrc[e] = 63 * 4;        // no counterpart anywhere ...
      if (nearg) {                                              // ... unless SOME pair of heads is near (one wave-uniform test for the whole block)
        int pc_[NH - 1];                                        // the six permutes go out together, then the six table reads:
#pragma unroll                                                  // two LDS round trips in all instead of two per row
        for (int e = 0; e < NH - 1; ++e) pc_[e] = bperm_i(7 * e + grp, pcode);   // (groups <= e read a pair lane with pe >= ph: "far")
#pragma unroll
        for (int e = 0; e < NH - 1; ++e) fsrc[e] = geo[e * (26 * 16) + pc_[e] * 16 + sub];   // (skipping single rows without a near pair measured slower)
      }
#pragma unroll
      for (int e = 0; e < NH - 1; ++e) asm volatile("" : "+v"(fsrc[e]));
    }
    SW_T(t2)
    // ---- relax the 8 neighbours of each head in registers ----
    const uint32_t cur_meta = rn.meta;
#ifdef PF_STAMPS
    { unsigned tmp_ = cur_meta; asm volatile("" : "+v"(tmp_)); }   // the loads have arrived
#endif
    SW_T(t3)                          // meaningful in the self lanes (sub == 8)
    // VARIANT 0: an entry superseded by a decrease-key (astar.py:96-100 rewrites it in place) is not a pop of the
    // reference: its head is consumed without effect and without being counted
    constexpr pf_u64 SELF7 = 0x7Full, MOVES = 0x7FFFFFFFFFFFFF80ull;   // lanes 0..6: the self lanes (sub == 8); lanes 7..62: the move lanes (sub < 8)
    pf_u64 stm = 0ull;
    if (SEM == 0) stm = SELF7 & mHave & (B((cur_meta & PF_M_CLOSED) != 0u) | B(rn.g != pg));
    const unsigned M = (unsigned)bperm_i(grp, (int)rn.tagmm) & 0xFFu;   // my head's self lane is lane `grp`: its record's mask byte
    if (SEM == 1) cur_g = bperm_d(grp, rn.g);
    const double base_g = SEM == 0 ? pg : cur_g;           // astar.py:85 popped g / MPA.py:135 g_score[current]
    const pf_u64 mValid = B((rn.tagmm >> PF_TAG_SHIFT) == tag);
    const pf_u64 mAvoid = B((rn.meta >> PF_AVOID_SHIFT) == avm);
    pf_u64 mOk = mHave & MOVES & B(((M >> d) & 1u) != 0u) & B(cur != target);
    if (SEM == 0) mOk &= ~B(((stm >> grp) & 1ull) != 0ull) & ~(mValid & B((rn.meta & PF_M_CLOSED) != 0u)) & ~(mAvoid & B(nidx != start) & B(nidx != target));
    else mOk &= ~mAvoid;
    const bool ok = PL(mOk);
    const bool rvalid = PL(mValid);
    const double tent = base_g + cost;
    const unsigned stale7 = SEM == 0 ? (unsigned)stm & 0x7Fu : 0u;   // the self lanes are lanes 0..6
    // ---- replay the earlier heads' effects on my cell (see above), in head order ----
    const double g0 = rvalid ? rn.g : PF_INF;
    double gmin = g0;                                           // g_score of my cell as head `grp` will find it
    pf_u64 mClsd = 0ull;                                        // (closed-set variants) an earlier head of this trip pops my cell
    int pred = 63 * 4;                                          // the last earlier lane that writes my cell's record (bit 0: by popping it)
    {
      // The replay arithmetic runs for every row: a lane with no counterpart in head e reads lane 63 (+inf, no
      // event) -- or skips the read when head e has no near successor at all -- and the lanes of a superseded head
      // offer +inf by themselves; only its pop has to be switched off by hand.
      const double tent_ok = ok ? tent : PF_INF;               // what my relaxation offers my cell (decided statically)
      const unsigned long long b_ = dbits(tent_ok);
      double tf[NH - 1];
      if (nearg) {                                              // (wave-uniform) no near pair at all: no lane has a counterpart, nothing to replay
#pragma unroll
      for (int e = 0; e < NH - 1; ++e) {
        tf[e] = PF_INF;
        if ((nearg >> (7 * e)) & 0x7Full) {                       // (wave-uniform: with the chip busy the LDS crossbar is worth sparing)
          const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(fsrc[e], (int)(unsigned)b_);
          const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(fsrc[e], (int)(unsigned)(b_ >> 32));
          tf[e] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
      }
#pragma unroll
      for (int e = 0; e < NH - 1; ++e) {
        // (wave-uniform) head e has no near successor: every lane's source is "none", nothing below changes anything.  Skipped in
        // the closed-set variants only -- A/B on one box: ga512 +1.7 %, but the MPA variant's leaner row lost 3 % to the branches
        if (SEM == 0 && !((nearg >> (7 * e)) & 0x7Full)) continue;
        pf_u64 mPop = B((fsrc[e] & 1) != 0);                               // head e pops my cell ...
        if (SEM == 0 && ((stale7 >> e) & 1u)) mPop = 0ull;                  // ... unless it is a superseded entry (uniform)
        const pf_u64 mRel = SEM == 0 ? (B(tf[e] < gmin) & ~mClsd) : B(tf[e] < gmin);   // head e's lane improves my cell
        gmin = PL(mRel) ? tf[e] : gmin;
        if (SEM == 0) mClsd |= mPop;
        pred = PL(mRel | mPop) ? fsrc[e] : pred;
      }
      }
    }
    const pf_u64 mImp = B(gmin < g0);                           // some earlier head of this trip improved my cell
    // in the open list: after the last event on my cell -- an improvement puts it there, a pop takes it out (MPA.py:
    // 122/147); the closed-set variants test "has an entry" (astar.py:92)
    const pf_u64 mNoPred = SEM == 1 ? B(pred == 63 * 4) : 0ull;  // (MPA variant: used twice) nothing earlier in this trip writes my cell's record
    pf_u64 mInop;
    if (SEM == 0) mInop = mValid | mImp;
    else mInop = (mNoPred & mValid & B((rn.meta & PF_M_INOPEN) != 0u)) | (B((pred & 1) == 0) & ~mNoPred);
    const pf_u64 mOkd = SEM == 0 ? (mOk & ~mClsd) : mOk;             // astar.py:83 closed set, incl. this trip's earlier pops
    const pf_u64 mBetter = mOkd & B(tent < gmin);               // astar.py:87 / MPA.py:137
    const pf_u64 mPush0 = SEM == 0 ? mBetter : (mBetter & ~mInop);
    const double fnew = tent + hn;                             // astar.py:90 / MPA.py:140
    const int nrc = (nr << 16) | nc;
    // ---- which heads take effect ----
    // bit h: an earlier group pushes a key at or below head h's f.  The heads' f are sorted, so a push violates the heads
    // from some h on: the first one is 1 + max(#heads with f below the key, own group), and the mask starts at the
    // smallest of those over the pushing lanes.
    unsigned viol = 0;
    {
      const unsigned long long fb = dbits(fnew);                // non-negative doubles order like their bit patterns
      // A push can only violate a head that exists, lies behind its own group and has f >= the key -- so at the very least the
      // LAST head: no push at or below f_last from a group before the last one means no violation at all, and the six compares
      // (with the twelve v_readlane that fetch the heads' f) run only in the trips that may lose a head (one in ten).
      if (mPush0 & B(grp < nh - 1) & B(fb <= f_last)) {
        int c_ = 0;
#pragma unroll
        for (int h = 1; h < NH; ++h) c_ += dbits(bcast_d(W.wf, W.wp + (h < nh ? h : nh - 1))) < fb ? 1 : 0;   // (past the last head: its f again, so the list stays sorted)
        const int hv = (c_ > grp ? c_ : grp) + 1;
        const unsigned hmin = wave_min_u32(PL(mPush0) ? (unsigned)hv : (unsigned)NH);
        viol = (0x7Fu << hmin) & 0x7Fu;
      }
    }
    // Branch-free form of the sequential rule.  A superseded head (closed-set variants) is transparent: consumed, no
    // effect, not a pop.  A real head fails if an earlier group pushed at or below its f or it is near an earlier
    // REAL head; since a head can only take effect when every earlier real head did, testing nearness against all
    // earlier real heads (not just the committed ones) changes nothing.  Everything below the first failing real head
    // is consumed; the target, or the step cap, cuts that prefix short.
    const unsigned tgtm = (unsigned)(SELF7 & B(rch == trc));                  // bit h: head h is the target (lanes 0..6 are the pairs (0, h))
    // an earlier head improved head h's OWN cell: its relaxations would start from another g (MPA.py:135) or its entry
    // is superseded (astar.py:96-100) -- the one effect that is not replayed; the trip stops there
    const unsigned c1 = (unsigned)mImp & 0x7Eu;                 // (the self lanes are lanes 0..6)
    const unsigned exist7 = (1u << nh) - 1u;
    const int allowed = max_steps - steps;                       // the cap counts real pops (astar.py:58 / MPA.py:118)
    int first = nh; unsigned real = exist7;                      // the usual trip: nothing in the way, every head takes effect ...
    if (((viol | c1 | tgtm | stale7) != 0u) | (allowed < NH)) {  // ... decided by ONE scalar test instead of the dependent chain below
      first = __builtin_ctz((((viol | c1) & ~stale7) | ~exist7) | 0x80u);   // first failing real head (or nh)
      real = ((1u << first) - 1u) & ~stale7;                     // the real pops below it
      if (real & tgtm) { first = __builtin_ctz(real & tgtm) + 1; real &= (1u << first) - 1u; }   // astar.py:64 / MPA.py:123
      if (__builtin_popcount(real) > allowed) {
        unsigned r_ = real; int keep = allowed > 0 ? allowed : 0;
        while (keep-- > 0) r_ &= r_ - 1;                         // drop the `allowed` lowest real heads ...
        first = __builtin_ctz(r_);                               // ... the next one is where the cap stops the loop
        real &= (1u << first) - 1u;
      }
    }
    const int consumed = first;
    const bool hit = (real & tgtm) != 0;
    const unsigned E = real & ~tgtm;                            // heads whose relaxation takes effect (nothing is relaxed at the target)
    steps += __builtin_popcount(real);
    SW_T(t4)
    if (consumed == 0) { status = 2; break; }                   // only the step cap can stop head 0
    W.wp += consumed;
    const pf_u64 mEff = B(((E >> grp) & 1u) != 0u);
    const pf_u64 mBE = mBetter & mEff;
    // a record written twice in this trip keeps the LAST write: a lane that writes tells the previous writer of its
    // cell (forward permute; lanes nobody addresses read 0) to keep quiet -- no two lanes store to one address
    bool keep = true;
    if (mBE & ~(SEM == 1 ? mNoPred : B(pred == 63 * 4)))        // (rare: two improvements of one cell in one trip)
      keep = __builtin_amdgcn_ds_permute(PL(mBE) ? (pred & ~3) : 63 * 4, 1) == 0;
    if (PL(mEff & SELF7) && keep)                                // astar.py:74 closed.add / leave the open list
      rec[cur].meta = SEM == 0 ? (cur_meta | PF_M_CLOSED) : (cur_meta & ~PF_M_INOPEN);
    const pf_u64 mPush = mPush0 & mEff;
    nbr_s += (unsigned)__builtin_popcountll(mOkd & mEff);       // (event counts of the wave: one scalar popcount each)
    if (SEM == 0) dk_s += (unsigned)__builtin_popcountll(mBE & mInop);
    if (PL(mBE) && keep) {
      Rec wv; wv.g = tent; wv.tagmm = (tag << PF_TAG_SHIFT) | (rn.tagmm & 0xFFu);
      wv.meta = (rn.meta & PF_AVOID_KEEP) | (unsigned)d | (SEM == 1 ? PF_M_INOPEN : 0u);
      rec[nidx] = wv;
    }
    // ---- pushes: below the limit -> into the window, else -> pool bucket ----
    // The slot index of a pool append comes from an LDS atomic; it is requested here and used after the window
    // inserts, so its latency is covered by them.
    const pf_u64 mTow = mPush & key_lt_m(fnew, tent, nrc, W.lf, W.lg, W.lc);
    const pf_u64 mTop = mPush & ~mTow;
    const int pba = (int)(fnew * PF_SW_Q);
    const int pb = pba < W.bcur ? NBK : (pba & (NBK - 1));       // below every regular bucket: the front bucket
    const pf_u64 mInr = B(pba - W.bcur < NBK);                 // inside the circular bucket range (front bucket: always)
    int pat = 0;
    const unsigned long long pm = mPush, im0 = mTow;
    if (PR) {
      // two-wave mode: 16 bytes into the ring, in lane order; the pool wave buckets them
      if (L->tail - rhead > PF_PR_RING_N - 128) { if (!pr_wait_room(*L)) { status = 3; break; } }
      const unsigned long long tm = pm & ~im0;
      if (PL(mTop)) {
        const unsigned at = (L->tail + (unsigned)__builtin_popcountll(tm & ((1ull << lane) - 1ull))) & (PF_PR_RING_N - 1);
        ent_put(L->ring + at, tent, nrc); L->ring_f[at] = fnew;     // (f travels too: the pool wave need not take the square root again)
      }
      L->tail += (unsigned)__builtin_popcountll(tm);
      // Published now, not at the end of the trip: the pool wave buckets them while I store records.  And if fewer than seven
      // heads will be left for the next trip, the take is requested here already (no eviction can follow: the window is
      // nearly empty), so that the answer is ready when the trip starts.
      if (W.n_pool + __builtin_popcountll(tm) > 0 && (W.wn - W.wp) + __builtin_popcountll(im0) < 7)
        pr_request(*L, 64 - (W.wn - W.wp) - __builtin_popcountll(im0), lane);
      else pr_publish(*L, lane);
    } else {
      if (PL(mTop & mInr)) pat = __hip_atomic_fetch_add(&P.cnt[pb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    W.n_pool += __builtin_popcountll(pm & ~im0);
    push_s += (unsigned)__builtin_popcountll(SEM == 0 ? (mPush & ~mInop) : mPush);   // heappush calls of the reference
    SW_T(t5)
    unsigned long long im = im0;
    const unsigned long long dkm = SEM == 0 ? (mPush & mInop) : 0ull;   // decrease-keys among the pushes
    while (im) {
      const int l = __builtin_ctzll(im); im &= im - 1;
      const int kc = bcast_i(nrc, l);
      if (SEM == 0 && ((dkm >> l) & 1ull)) {
        // decrease-key (astar.py:96-100): if the entry it supersedes is in the window, take it out now instead of
        // dropping it when it reaches the head (it would cost a head slot and a batch of loads)
        const unsigned long long qm = B(lane >= W.wp) & B(lane < W.wn) & B(W.wc == kc);
        if (qm) {
          const int q = __builtin_ctzll(qm);
          const double sf = wave_down_d(W.wf), sg = wave_down_d(W.wg); const int sc = wave_down_i(W.wc);
          if (lane >= q && lane < W.wn - 1) { W.wf = sf; W.wg = sg; W.wc = sc; }
          W.wn -= 1;
        }
      }
      // (an eviction inside this loop may have lowered the limit below this key: sw_add re-checks)
      if (!sw_add<PR>(P, W, bcast_d(fnew, l), bcast_d(tent, l), kc, lane, L)) { status = 3; break; }
    }
    if (status == 3) break;
    SW_T(t6)
    if (!PR) {
      const pf_u64 mFits = mInr & B(pat < CAP);
      if (PL(mTop & mFits)) ent_put(P.be + pb * CAP + pat, tent, nrc);
      const unsigned long long sm = mTop & ~mFits;              // bucket full or beyond the circular range: spill list
      if (sm) {
        if (PL(sm)) {
          if (PL(mInr)) __hip_atomic_fetch_add(&P.cnt[pb], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // a number past the end: give it back
          const int at = W.n_spill + __builtin_popcountll(sm & ((1ull << lane) - 1ull));
          if (at < PF_SW_SPILL) ent_put(P.se + at, tent, nrc);
        }
        W.n_spill += __builtin_popcountll(sm);
        st.spills += (unsigned)__builtin_popcountll(sm);
        if (W.n_spill > PF_SW_SPILL) { status = 3; break; }     // never silent
      }
    }
    if (hit) { status = 0; break; }
    if (PR) pr_publish(*L, lane);                               // (evictions of the insert loop, if any)
    PF_LDS_ORDER();
    SW_T(t7)
    SW_ACC(0, t0, t1) SW_ACC(1, t1, t2) SW_ACC(2, t2, t3) SW_ACC(3, t3, t4) SW_ACC(4, t4, t5) SW_ACC(5, t5, t6) SW_ACC(6, t6, t7)
#ifdef PF_STAMPS
    sw_acc[7] += 1;
#endif
#ifdef PF_TRIPS
    n_max += 1;                                                 // diagnostic build: trips instead of the open-list high-water mark
    {                                                           // ... and why each trip stopped where it did
      const int f0 = __builtin_ctz((((viol | c1) & ~stale7) | ~exist7) | 0x80u);
      if (f0 >= nh) { if (nh < NH) tr_short += 1; else tr_full += 1; }
      else if ((viol >> f0) & 1u) tr_viol += 1; else tr_near += 1;      // tr_near: an earlier head improved the head's own cell
      // how far the trip could go if nearness were resolved by forwarding and only an earlier head improving
      // head h's own cell stopped it (decisions taken from the loaded state: an estimate)
      tr_pot += (unsigned)f0;
    }
#else
    const int n_open = W.n_pool + (W.wn - W.wp);
    if (n_open > n_max) n_max = n_open;
#endif
  }
#ifdef PF_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) { atomicAdd(&g_stamps[i], sw_acc[i]); atomicAdd(&g_stamps[8 + i], sw_cnt[i]); }
    for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[16 + i], sw_er[i]);
  }
#endif
  if (PR) {
    bool ovf = false;
    pr_search_stop(*L, st.spills, ovf, lane);
    if (ovf) status = 3;                                        // the pool wave lost entries (or never answered): never silent
  }
  if (n_max > st.max_open) st.max_open = n_max;
  st.pops += (unsigned long long)steps; st.pushes += 1u + push_s;
#ifdef PF_TRIPS
  st.nbr += tr_viol; st.deckey += tr_near; st.spills += tr_pot; (void)tr_full; (void)tr_short; (void)nbr_s; (void)dk_s;
#else
  st.nbr += nbr_s; st.deckey += dk_s;
#endif
  return status;
}

}  // namespace pf
