// pf_astar.h -- one-wavefront-per-search A* connector (K2a / K2b / Dijkstra).
//
// Reference semantics (bit-exact pop order on the total order (f, g, (r,c))):
//   VARIANT 0  AStarSolver.solve, astar.py:33-101  (closed set, decrease-key, cap 3RC)
//   VARIANT 1  MPA._a_star,       MPA.py:106-151   (no closed set, an open node keeps its old (f,g) entry when
//              improved, popped nodes may be re-pushed, g from g_score, cap 2RC)
//   VARIANT 2  DijkstraSolver.solve, dijkstra.py:32-97 (variant 0 with h == 0)
//
// Mapping (MI355X, wave64): per-cell records in this slot's HBM scratch (epoch stamped, never cleared), the open
// list as a sorted 64-entry register window over an HBM bucket pool (pf_astar_sw.h), up to seven pops per trip with
// the lanes replaying each other's effects.  The wrapper below handles the trivial cases, the exact
// work-avoidance proofs (components, avoided goal, sealed pockets) and the parent walk.
#pragma once
#include "pf_device.h"

namespace pf {

// LDS of one search wave (bytes): [0, 4 * (PF_SW_NBK + 1)) bucket counts; [0, 10240) doubles as the pocket flood's scratch before the
// search starts; [6144, 8192) staging area of the refill's run sort (and, in the plateau kernels, of the pivot selection); [PF_GEO_OFF, +2496) replay source-lane table;
// [PF_SX_OFF, +256) prefix-sum marks of the refill
#define PF_SX_OFF 14848
#define PF_LDS_BYTES (PF_SX_OFF + 256)
struct Open {
  double* lf;  // the wave's LDS (bucket counts first)
  char* sx;    // 256 bytes of LDS for the refill's bucket marks
  double* of;  // this slot's HBM bucket pool (PF_POOL_STRIDE bytes)
};

struct SettleMem {
  unsigned long long* lab;          // [RC] epoch-coded labels of this slot
  int* touched;                     // [2 RC] cells whose label was lowered, in order (duplicates allowed)
  unsigned char* par;               // [RC] parent move of every certified node (move index FROM the parent TO the node)
  unsigned* epoch;                  // this slot's label epoch (1..126), in HBM between launches
  int touched_cap;
  bool astar_too;                   // run it for VARIANT 0 as well (default: Dijkstra only, which it always certifies)
};
struct Slot {
  SettleMem sm;       // parallel closed-set engine's scratch (pf_settle.h); sm.lab == nullptr: not available
  Rec* rec;
  const uint8_t* mm;  // move masks the records were initialised from (for the wrap wipe)
  uint32_t tag;       // solve epoch (24 bit)
  uint32_t avoid_ep;  // eval epoch (14 bit)
};

struct AStat {
  unsigned long long pops, pushes, nbr, deckey;
  int max_open;
  unsigned spills;   // open-list entries that went through the spill list (full bucket / beyond the circular range)
  unsigned settled, sequential;   // closed-set searches answered by the parallel engine / handed on to the sequential one
};
// diagnostic build only (-DPF_STAMPS): shader-clock time per section of the pop loop, never in the product .so
#ifdef PF_STAMPS
__device__ unsigned long long g_stamps[24];
#endif

// Mark cells[0..n) of a path as "avoid" for this slot's current eval epoch.
PF_DEV void mark_avoid(const Slot& s, const int* cells, int n, int lane) {
  for (int i = lane; i < n; i += 64) s.rec[cells[i]].meta = s.avoid_ep << PF_AVOID_SHIFT;
}

// the same for a caller-supplied list (pf_astar_batch): indices outside the grid are skipped, never dereferenced
PF_DEV void mark_avoid_checked(const Slot& s, const int* cells, int n, int RC, int lane) {
  for (int i = lane; i < n; i += 64) { const int c = cells[i]; if ((unsigned)c < (unsigned)RC) s.rec[c].meta = s.avoid_ep << PF_AVOID_SHIFT; }
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
#include "pf_astar_sw.h"
#ifdef PF_TWO_WAVE
#include "pf_astar_pr.h"     // two wavefronts per search (off by default and measured 0.90x: compiled only on request)
#else
namespace pf {               // PR is false in every instantiation of this build: the pop wave's link functions are never called
PF_DEV void pr_search_start(PrLink&, int, int, int, bool, int) {}
PF_DEV bool pr_search_stop(PrLink&, unsigned&, bool&, int) { return true; }
PF_DEV void pr_request(PrLink&, int, int) {}
PF_DEV int pr_take(PrLink&, SwWin&, int) { return 0; }
PF_DEV void pr_publish(PrLink&, int) {}
PF_DEV bool pr_wait_room(PrLink&) { return true; }
PF_DEV int pr_ld(const int*) { return 0; }
}  // namespace pf
#endif
#include "pf_settle.h"
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
// (One wave runs the flood.  Its LDS instructions execute in order, so the hand-over points between the lanes need the
// stores to have been ISSUED, not a workgroup barrier -- which, in the two-wave workgroups of pf_astar_pr.h, the pool wave
// would never join.)
#define PF_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
PF_DEV int pocket_flood(const Grid& G, const Slot& s, int* lds, int from, int to, int exempt, int lane) {
  int* tab = lds;                    // [PF_FLOOD_TAB] visited cells (open addressing), -1 = empty
  int* queue = lds + PF_FLOOD_TAB;   // [PF_FLOOD_K]
  for (int i = lane; i < PF_FLOOD_TAB; i += 64) tab[i] = -1;
  PF_WAVE_SYNC();
  if (lane == 0) { queue[0] = from; tab[(unsigned)(from * 0x9E3779B1u) >> 21] = from; }
  PF_WAVE_SYNC();
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
    PF_WAVE_SYNC();
  }
  return 1;
}

// Returns status (PF_ST_*).  On PF_ST_OK, out[0..out_n) holds the path cells
// (r*C+c) start..target.  out_cap is the room available at `out`.
// av_list / av_n: the cells of this search's avoid set as a list (the same cells mark_avoid has stamped into the
// records); only the closed-set variants use it, to re-mark them under the parallel engine's label epoch.
template <int VARIANT, bool PLAT = false, bool PR = false>
__device__ __forceinline__ int astar(const Grid& G, Slot& s, const Open& O, int start, int target, int* out, int out_cap,
                     int& out_n, AStat& st, int lane, const int* av_list = nullptr, int av_n = 0, PrLink* L = nullptr) {
  // VARIANT 0 AStarSolver.solve (astar.py:33-101), 1 MPA._a_star (MPA.py:106-151), 2 DijkstraSolver.solve
  // (dijkstra.py:32-97: the loop of variant 0 with heap entries (g, node), i.e. h == 0 and key (g, g, node))
  constexpr int SEM = VARIANT == 1 ? 1 : 0;
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
  // The closed-set variants first try the 64-nodes-per-trip engine (pf_settle.h); it returns PF_ST_SEQ when it cannot
  // certify that the sequential loop would have produced the same labels and parents, and the search then runs below.
  if (SEM == 0 && s.sm.lab && G.step_cap == 0 && (VARIANT == 2 || s.sm.astar_too)) {
    const int rs = settle<VARIANT>(G, O, s.sm, start, target, tr, tc, av_list, av_n, out, out_cap, out_n, st, lane);
    if (rs != PF_ST_SEQ) { st.settled += 1; return rs; }
    st.sequential += 1;
  }
  s.tag = (uint32_t)first_i((int)s.tag) + 1;
  const uint32_t tag = s.tag;
  // seed: g(start) = 0 in the records; the window starts as the single entry (h(start), 0, start)
  long dr0 = sr - tr, dc0 = sc_ - tc;
  const double h0 = VARIANT == 2 ? 0.0 : __builtin_sqrt((double)(dr0 * dr0 + dc0 * dc0));   // astar.py:45 / MPA.py:113 / dijkstra.py:45
  if (lane == 0) {
    Rec r0 = rec[start];
    Rec w; w.g = 0.0; w.tagmm = (tag << PF_TAG_SHIFT) | (r0.tagmm & 0xFFu);
    w.meta = (r0.meta & PF_AVOID_KEEP) | (SEM == 1 ? PF_M_INOPEN : 0u);
    rec[start] = w;
  }
  long long cap_steps = (long long)G.R * C * (SEM == 0 ? 3 : 2);   // astar.py:58 / MPA.py:118 (R,C <= 4096)
  if (G.step_cap > 0 && G.step_cap < cap_steps) cap_steps = G.step_cap;   // test hook, see pf_set_option("astar_step_cap")
  const int max_steps = (int)cap_steps;
  const int status4 = pop_loop_sw<VARIANT, PLAT, PR>(G, rec, O, tag, avm, start, target, tr, tc, max_steps, h0, (sr << 16) | sc_, st, lane, L);
  if (status4 != 0) return status4;

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
