// pathfit.hip -- kernels K0..K7 and the C-ABI of libpathfit.so (gfx950 / MI355X).
// See include/pathfit.h for the boundary and DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <queue>
#include <string>
#include <vector>

#include "../../include/pathfit.h"
#include "pf_astar.h"
#include "pf_device.h"
#include "pf_score.h"

using namespace pf;

#define DOM_MAACO 1
#define DOM_MPA 2
#define DOM_PSO 3
#define DOM_MPA_FADS 5
#define DOM_GA 4
#define DOM_GA_SELECT 7

// ===========================================================================
// device-side launch parameter blocks
// ===========================================================================
struct DevCounters {
  unsigned long long pops, pushes, nbr, path_cells, steps, candidates, deckey, overflow, pruned, settled, sequential;
  unsigned long long done;   // agents of the running batch that have finished (k_decode_batch: the tail policy of the parallel engine)
};

struct Common {
  Grid G;
  Rec* rec;              // [nslots][RC]
  char* tier2;           // [nslots][PF_POOL_STRIDE] open-list HBM scratch (bucket pool / tier-2 overflow)
  uint32_t* slot_state;  // [nslots][2] = {tag, avoid_ep}
  int* work;             // dynamic work counter
  const int* queue;      // optional work order (longest-expected-first), else identity
  DevCounters* cnt;
  int S;                 // LDS bin capacity
  int retry;             // only agents whose status == 3
  // parallel closed-set engine (pf_settle.h): per-slot label / touched / parent arrays, or null (off)
  unsigned long long* st_lab; int* st_touched; unsigned char* st_par; unsigned* st_epoch;
  bool st_astar;         // also for the A* variant (else Dijkstra only)
  int st_top;            // ... or only for the items at queue positions below this (the longest-expected ones: the batch ends on
                         // them, and the engine shortens a search's chain at the price of more traffic per node)
  int st_tail;           // ... and, in a decode batch, for every search that STARTS once at most this many agents are unfinished: the
                         // chip is mostly idle by then and what is left are the long chains the batch ends on
};

PF_DEV Open make_open(char* smem, int /*S*/, char* tier2) {
  Open O;
  O.lf = (double*)smem;
  O.sx = smem + PF_SX_OFF;
  static_assert(PF_GEO_OFF >= (PF_FLOOD_TAB + PF_FLOOD_K) * 4 && PF_GEO_OFF + sizeof(GeoTab) <= PF_SX_OFF, "LDS layout");
  geo_to_lds(smem, lane_id());                    // the replay's source-lane table (pf_astar_sw.h), once per wave
  O.of = (double*)(tier2 + (size_t)blockIdx.x * PF_POOL_STRIDE);
  return O;
}
static size_t open_bytes(int /*S*/) { return (size_t)PF_LDS_BYTES; }

PF_DEV Slot slot_load(const Common& c, int RC) {
  Slot s;
  s.rec = c.rec + (size_t)blockIdx.x * RC;
  s.mm = c.G.mm;
  s.tag = c.slot_state[2 * blockIdx.x];
  s.avoid_ep = c.slot_state[2 * blockIdx.x + 1];
  s.sm.lab = c.st_lab ? c.st_lab + (size_t)blockIdx.x * RC : nullptr;
  s.sm.touched = c.st_lab ? c.st_touched + (size_t)blockIdx.x * 2 * RC : nullptr;
  s.sm.par = c.st_lab ? c.st_par + (size_t)blockIdx.x * RC : nullptr;
  s.sm.epoch = c.st_lab ? c.st_epoch + blockIdx.x : nullptr;
  s.sm.touched_cap = 2 * RC;
  s.sm.astar_too = c.st_astar;
  return s;
}
PF_DEV void slot_store(const Common& c, const Slot& s, int lane) {
  if (lane == 0) { c.slot_state[2 * blockIdx.x] = s.tag; c.slot_state[2 * blockIdx.x + 1] = s.avoid_ep; }
}
// new agent evaluation: fresh avoid epoch; wipe the slot before an epoch can wrap.  One evaluation runs at most
// PF_MAX_SEARCHES_PER_EVAL searches (each takes a fresh 24-bit tag): pf_decode_batch rejects W beyond it, the other
// callers run one or two.
#define PF_MAX_SEARCHES_PER_EVAL 0x8000
PF_DEV void slot_begin_eval(Slot& s, int RC, int lane) {
  s.avoid_ep += 1;
  if (s.avoid_ep >= 0x3FF0u || s.tag >= 0xFFFFFFu - 2u * PF_MAX_SEARCHES_PER_EVAL) slot_wipe(s, RC, lane);
}
PF_DEV int next_work(int* work, int lane) {
  int a = 0;
  if (lane == 0) a = atomicAdd(work, 1);
  return first_i(a);
}
// dynamic work distribution in longest-expected-first order (the batch finishes when its longest search
// does, so long searches must start first); returns -1 when the queue is drained
PF_DEV int next_agent(const Common& c, int n, int lane, int* pos = nullptr) {
  const int w = next_work(c.work, lane);
  if (w >= n) return -1;
  if (pos) *pos = w;
  return c.queue ? c.queue[w] : w;
}
PF_DEV void flush_counters(DevCounters* c, const AStat& st, unsigned long long cells, unsigned long long ovf, int lane) {
  if (lane == 0) {
    atomicAdd(&c->pops, st.pops); atomicAdd(&c->pushes, st.pushes); atomicAdd(&c->nbr, st.nbr);
    atomicAdd(&c->deckey, st.deckey); atomicAdd(&c->path_cells, cells); atomicAdd(&c->overflow, ovf & 0xFFFFFFFFull);
    if (st.spills) atomicAdd(&c->candidates, (unsigned long long)st.spills);   // A* kernels: spilled open-list entries
    if (st.settled) atomicAdd(&c->settled, (unsigned long long)st.settled);
    if (st.sequential) atomicAdd(&c->sequential, (unsigned long long)st.sequential);
    if (ovf >> 32) atomicAdd(&c->pruned, ovf >> 32);           // MPA items count pruned rebuilds in the upper half
  }
}

// ===========================================================================
// K0: grid preparation
// ===========================================================================
// static move masks (helper.py:38-52 order) for restrict on/off, and the clipped
// squared distance to the nearest obstacle inside a window of radius `rad` (7: every min_safe_distance up to 7; 15: up to
// 15.9, the largest distance whose square fits the u8 table -- helper.py:67-80 only ever needs obstacles closer than
// min_safe_distance).
__global__ void k_grid_prep(const uint8_t* occ, int R, int C, uint8_t* mm_r1, uint8_t* mm_r0, uint8_t* d2near, int rad) {
  int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= R * C) return;
  int r = cell / C, c = cell % C;
  unsigned m1 = 0, m0 = 0;
  for (int m = 0; m < 8; ++m) {
    int nr = r + HM_DR[m], nc = c + HM_DC[m];
    if (nr < 0 || nr >= R || nc < 0 || nc >= C || occ[nr * C + nc] == 1) continue;
    m0 |= 1u << m;
    if (m >= 4 && (occ[nr * C + c] == 1 || occ[r * C + nc] == 1)) continue;   // corner cells are in bounds here
    m1 |= 1u << m;
  }
  mm_r1[cell] = (uint8_t)m1; mm_r0[cell] = (uint8_t)m0;
  int best = 255;
  for (int dr = -rad; dr <= rad; ++dr) for (int dc = -rad; dc <= rad; ++dc) {
    int rr = r + dr, cc = c + dc;
    if (rr < 0 || rr >= R || cc < 0 || cc >= C || occ[rr * C + cc] != 1) continue;
    int d2 = dr * dr + dc * dc;
    if (d2 < best) best = d2;
  }
  d2near[cell] = (uint8_t)best;
}

// Exact squared distance to the nearest obstacle within a window of radius W, for any W (helper.py:67-80 accepts any
// min_safe_distance): two separable integer passes.  Rows: the distance to the nearest obstacle in the same row (|dc| <= W,
// else BIG); columns: d2[r][c] = min over |dr| <= W of dr^2 + g[r + dr][c]^2.  Every obstacle at Euclidean distance <= W lies
// inside both windows, so every d2 <= W^2 is exact; cells with nothing that close hold `cap`.
#define PF_EDT_BIG 0x3FFFFFFF
__global__ void k_edt_rows(const uint8_t* occ, int R, int C, int W, int* g) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= R * C) return;
  const int r = cell / C, c = cell - r * C;
  int best = PF_EDT_BIG;
  for (int d = 0; d <= W && best == PF_EDT_BIG; ++d) {
    if ((c - d >= 0 && occ[r * C + c - d] == 1) || (c + d < C && occ[r * C + c + d] == 1)) best = d;
  }
  g[cell] = best;
}
__global__ void k_edt_cols(const int* g, int R, int C, int W, int cap, int* d2) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= R * C) return;
  const int r = cell / C, c = cell - r * C;
  long best = cap;
  const int r0 = r - W < 0 ? 0 : r - W, r1 = r + W > R - 1 ? R - 1 : r + W;
  for (int rr = r0; rr <= r1; ++rr) {
    const int gv = g[rr * C + c];
    if (gv == PF_EDT_BIG) continue;
    const long v = (long)(rr - r) * (rr - r) + (long)gv * gv;
    best = v < best ? v : best;
  }
  d2[cell] = (int)best;
}

// search scratch initialisation: every record carries its cell's static move mask
__global__ void k_slot_init(Rec* rec, const uint8_t* mm, int RC, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) { Rec z; z.g = 0.0; z.tagmm = mm[i % (size_t)RC]; z.meta = 0; rec[i] = z; }
}

// ===========================================================================
// work estimates (longest-expected-first scheduling)
// ===========================================================================
PF_DEV float cell_dist(const Grid& G, int a, int b) {
  const int ar = a / G.C, ac = a - ar * G.C, br = b / G.C, bc = b - br * G.C;
  const float dr = (float)(ar - br), dc = (float)(ac - bc);
  return sqrtf(dr * dr + dc * dc);
}
__global__ void k_plan_astar(Grid G, int n, const int* start, const int* target, float* est) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < n) est[a] = cell_dist(G, start[a], target[a]);
}
__global__ void k_plan_decode(Grid G, int n, int W, const int* wp_cells, const double* wp_pos, int start, int target, float* est) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  int cur = start; float e = 0.f;
  for (int k = 0; k <= W; ++k) {
    int goal = target;
    if (k < W) {
      if (wp_cells) goal = wp_cells[(size_t)a * W + k];
      else {
        long r = (long)__builtin_rint(wp_pos[((size_t)a * W + k) * 2]), c = (long)__builtin_rint(wp_pos[((size_t)a * W + k) * 2 + 1]);
        r = r < 0 ? 0 : (r > G.R - 1 ? G.R - 1 : r); c = c < 0 ? 0 : (c > G.C - 1 ? G.C - 1 : c);
        goal = (int)(r * G.C + c);
      }
    }
    e += cell_dist(G, cur, goal); cur = goal;
  }
  est[a] = e;
}

// ===========================================================================
// K2: A* connector batch
// ===========================================================================
struct AstarArgs {
  Common c;
  int n, path_cap;
  const int* start; const int* target;
  const long long* avoid_off; const int* avoid_cells;
  int* cells; int* len; int* status; long long* counters;
};

template <int VARIANT, bool PLAT, bool PR = false>
__global__ __launch_bounds__(PR ? 128 : 64) __attribute__((amdgpu_waves_per_eu(PR ? 3 : 1, 8))) void k_astar_batch(AstarArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  const int RC = p.c.G.R * p.c.G.C;
#ifdef PF_TWO_WAVE
  if (PR) {                                                       // two wavefronts per search (pf_astar_pr.h): wave 1 owns the bucket pool
    const int wave = (int)(threadIdx.x >> 6);
    if (wave == 0) pr_init_ctl(smem, lane);
    __syncthreads();
    if (wave == 1) {
      pool_wave<VARIANT>(smem, p.c.tier2 + (size_t)blockIdx.x * PF_POOL_STRIDE, p.c.rec + (size_t)blockIdx.x * RC, p.c.G.C, lane);
      return;
    }
  }
  PrLink link = {};
  if (PR) link = pr_link(smem);                                   // (single-wave kernels never touch the link's LDS block: it lies beyond their allocation)
#else
  static_assert(!PR, "two-wave searches need -DPF_TWO_WAVE");
  PrLink link = {};
#endif
  PrLink* const L = &link;
  Open O = make_open(smem, p.c.S, p.c.tier2);
  Slot s = slot_load(p.c, RC);
  AStat tot = {0, 0, 0, 0, 0, 0};
  unsigned long long cells = 0, ovf = 0;
  for (;;) {
    int qpos = 0;
    const int a = next_agent(p.c, p.n, lane, &qpos);
    if (a < 0) break;
    if (p.c.retry && p.status[a] != 3) continue;
    s.sm.astar_too = p.c.st_astar || (p.c.queue && qpos < p.c.st_top);
    slot_begin_eval(s, RC, lane);
    if (p.avoid_off) {
      const long long b = p.avoid_off[a], e = p.avoid_off[a + 1];
      mark_avoid_checked(s, p.avoid_cells + b, (int)(e - b), RC, lane);
    }
    AStat st = {0, 0, 0, 0, 0, 0};
    int n = 0;
    // cell indices come straight from the caller: an index outside the grid is "not a valid node" (astar.py:37-39 /
    // MPA.py:109-111 -> []), never an address
    const int sa = p.start[a], ta = p.target[a];
    const long long ab = p.avoid_off ? p.avoid_off[a] : 0, ae = p.avoid_off ? p.avoid_off[a + 1] : 0;
    const int rc = ((unsigned)sa >= (unsigned)RC || (unsigned)ta >= (unsigned)RC) ? 1 :
                   astar<VARIANT, PLAT, PR>(p.c.G, s, O, sa, ta, p.cells + (size_t)a * p.path_cap, p.path_cap, n, st, lane,
                                  p.avoid_off ? p.avoid_cells + ab : nullptr, (int)(ae - ab), L);
    if (lane == 0) {
      p.len[a] = rc == 0 ? n : 0;
      p.status[a] = rc;
      if (p.counters) {
        p.counters[4 * a] = (long long)st.pops; p.counters[4 * a + 1] = (long long)st.pushes;
        p.counters[4 * a + 2] = st.max_open; p.counters[4 * a + 3] = (long long)st.nbr;
      }
    }
    tot.pops += st.pops; tot.pushes += st.pushes; tot.nbr += st.nbr; tot.deckey += st.deckey; tot.spills += st.spills;
    tot.settled += st.settled; tot.sequential += st.sequential;
    cells += rc == 0 ? n : 0; ovf += rc == 3;
  }
#ifdef PF_TWO_WAVE
  if (PR) pr_exit(smem, lane);
#endif
  slot_store(p.c, s, lane);
  flush_counters(p.c.cnt, tot, cells, ovf, lane);
}

// ===========================================================================
// K1: scoring batch
// ===========================================================================
struct ScoreArgs {
  Grid G; ScoreP sp; int n, path_cap;
  const int* cells; const int* len; double* stats;
};
__global__ __launch_bounds__(64) void k_score_batch(ScoreArgs p) {
  const int lane = lane_id();
  for (int a = blockIdx.x; a < p.n; a += gridDim.x) {
    double out[5];
    score_path(p.G, p.sp, p.cells + (size_t)a * p.path_cap, p.len[a], lane, out);
    if (lane < 5) p.stats[(size_t)a * 5 + lane] = out[lane];
  }
}

// ===========================================================================
// K3: chained waypoint decode (+ K1)
// ===========================================================================
struct DecodeArgs {
  Common c; ScoreP sp; int do_score;
  int n, W, path_cap, start, target;
  const int* wp_cells; const double* wp_pos;
  int* cells; int* len; int* status; double* stats;
};
template <bool PLAT>
__global__ __launch_bounds__(64) void k_decode_batch(DecodeArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  const Grid& G = p.c.G;
  const int RC = G.R * G.C;
  Open O = make_open(smem, p.c.S, p.c.tier2);
  Slot s = slot_load(p.c, RC);
  AStat tot = {0, 0, 0, 0, 0, 0};
  unsigned long long cells = 0, ovf = 0;
  for (;;) {
    int qpos = 0;
    const int a = next_agent(p.c, p.n, lane, &qpos);
    if (a < 0) break;
    if (p.c.retry && p.status[a] != 3) continue;
    s.sm.astar_too = p.c.st_astar || (p.c.queue && qpos < p.c.st_top);
    slot_begin_eval(s, RC, lane);
    int* out = p.cells + (size_t)a * p.path_cap;
    int n = 1, cur = p.start, rc = 0;
    // Exact short cut.  A waypoint on an obstacle makes its segment's connector return [] at once (astar.py:37-39), and an empty
    // segment makes the whole decode return [] (ga_solver.py:74 / pso.py:77) -- whatever the segments before it found, and they
    // have no effect outside the call.  So the answer is known before the first search: PSO positions round onto obstacles
    // 27 % of the time per waypoint (~80 % of a swarm on G512), and the reference spends their earlier segments for nothing.
    for (int k = 0; k < p.W && rc == 0; ++k) {
      int goal;
      if (p.wp_cells) goal = p.wp_cells[(size_t)a * p.W + k];
      else {
        const double x = p.wp_pos[((size_t)a * p.W + k) * 2], y = p.wp_pos[((size_t)a * p.W + k) * 2 + 1];
        long r = (long)__builtin_rint(x), c = (long)__builtin_rint(y);
        r = r < 0 ? 0 : (r > G.R - 1 ? G.R - 1 : r);
        c = c < 0 ? 0 : (c > G.C - 1 ? G.C - 1 : c);
        goal = (int)(r * G.C + c);
      }
      if ((unsigned)goal >= (unsigned)RC || G.occ[goal] == 1) rc = 1;
    }
    if (lane == 0) { out[0] = p.start; s.rec[p.start].meta = s.avoid_ep << PF_AVOID_SHIFT; }   // ga_solver.py:63-65
    for (int k = 0; k <= p.W && rc == 0; ++k) {
      int goal = p.target;
      if (k < p.W) {
        if (p.wp_cells) goal = p.wp_cells[(size_t)a * p.W + k];
        else {                                                   // pso.py:61,69-70: round-half-even then clamp
          double x = p.wp_pos[((size_t)a * p.W + k) * 2], y = p.wp_pos[((size_t)a * p.W + k) * 2 + 1];
          long r = (long)__builtin_rint(x), c = (long)__builtin_rint(y);
          r = r < 0 ? 0 : (r > G.R - 1 ? G.R - 1 : r);
          c = c < 0 ? 0 : (c > G.C - 1 ? G.C - 1 : c);
          goal = (int)(r * G.C + c);
        }
      }
      int m = 0;
      if (p.c.st_tail > 0 && !s.sm.astar_too) {                  // the batch's tail: few agents left, the chip mostly idle -> shorten the chain
        const unsigned long long dn = __hip_atomic_load(&p.c.cnt->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((long long)p.n - (long long)first_u64(dn) <= (long long)p.c.st_tail) s.sm.astar_too = true;
      }
      rc = astar<0, PLAT>(G, s, O, cur, goal, out + n - 1, p.path_cap - (n - 1), m, tot, lane, out, n);   // ga_solver.py:68-72 (avoid = the cells visited so far)
      if (rc != 0) break;                                        // :74 / :85 -> []
      mark_avoid(s, out + n, m - 1, lane);                       // :76 nodes_in_path_so_far.update
      n += m - 1;
      cur = goal;
    }
    // ga_solver.py:90-93 (drop consecutive duplicates) is a no-op here: a segment's tail never starts with its head
    if (rc != 0) n = 0;
    double sc[5];
    if (p.do_score) score_path(G, p.sp, out, n, lane, sc);
    if (lane == 0) { p.len[a] = n; p.status[a] = rc; atomicAdd(&p.c.cnt->done, 1ull); }
    if (p.do_score && lane < 5) p.stats[(size_t)a * 5 + lane] = sc[lane];
    cells += n; ovf += rc == 3;
  }
  slot_store(p.c, s, lane);
  flush_counters(p.c.cnt, tot, cells, ovf, lane);
}

// ===========================================================================
// K6: PSO update + pbest
// ===========================================================================
struct PsoArgs {
  int n, W, R, C; double w, c1, c2, max_vel;
  double* pos; double* vel; const double* pbest; const double* gbest;
  unsigned long long seed, iter, agent0;
  double* pos_keep; double* vel_keep;   // non-null: the pre-update values are left here (the asynchronous sweep's roll-back copy)
};
// one thread per (particle, waypoint): the counter RNG is random access, so
// waypoint d starts at draw 4*d of the particle's stream (pso.py:186-190 order).
__global__ void k_pso_update(PsoArgs p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.n * p.W) return;
  const int part = t / p.W, d = t - part * p.W;
  Rng g; g.init(p.seed, DOM_PSO, p.iter, p.agent0 + (unsigned long long)part);
  g.set_ctr(4ull * (unsigned long long)d);
  for (int ax = 0; ax < 2; ++ax) {
    const size_t i = ((size_t)part * p.W + d) * 2 + ax;
    const double hi = ax == 0 ? (double)(p.R - 1) : (double)(p.C - 1);
    const double r1 = g.random(), r2 = g.random();
    const double x = p.pos[i], v0 = p.vel[i];
    if (p.pos_keep) { p.pos_keep[i] = x; p.vel_keep[i] = v0; }
    double v = p.w * v0 + p.c1 * r1 * (p.pbest[i] - x) + p.c2 * r2 * (p.gbest[d * 2 + ax] - x);
    v = fmin(fmax(v, -p.max_vel), p.max_vel);                     // np.clip pso.py:192-193
    double nx = x + v;
    nx = fmin(fmax(nx, 0.0), hi);                                 // np.clip pso.py:201-202
    p.vel[i] = v; p.pos[i] = nx;
  }
}
__global__ void k_pso_pbest(int n, int W, const double* pos, const double* stats, const int* len, double* pbest,
                            double* pbest_fit, int* improved) {
  const int a = blockIdx.x;
  if (a >= n) return;
  const bool better = len[a] > 0 && stats[(size_t)a * 5 + 4] < pbest_fit[a];   // pso.py:210,216
  for (int i = threadIdx.x; i < W * 2; i += blockDim.x)
    if (better) pbest[(size_t)a * W * 2 + i] = pos[(size_t)a * W * 2 + i];
  __syncthreads();
  if (threadIdx.x == 0) { improved[a] = better; if (better) pbest_fit[a] = stats[(size_t)a * 5 + 4]; }
}

// pbest paths stay in HBM too: rows of the particles k_pso_pbest marked as improved are copied over (pso.py:218-219)
__global__ __launch_bounds__(64) void k_pso_pbest_paths(int n, int path_cap, const int* cells, const int* len, const int* improved,
                                                        int* pb_cells, int* pb_len) {
  const int a = blockIdx.x;
  if (a >= n || !improved[a]) return;
  const int L = len[a];
  for (int i = threadIdx.x; i < L; i += 64) pb_cells[(size_t)a * path_cap + i] = cells[(size_t)a * path_cap + i];
  if (threadIdx.x == 0) pb_len[a] = L;
}
// One round of the asynchronous sweep committed in ONE launch (it was pbest + pbest paths + two / four device-to-device copies
// + two small reads, each behind a stream synchronisation).  Block a = particle a of the evaluated batch [0, m):
//   a <  k   final: pso.py:216-220 -- fitness below its pbest (strict) -> position, fitness and path row become the pbest;
//   a == j   (j < k, or -1) the round's gbest improver: pso.py:222-229 -- its position, five stats and path row (length first)
//            go to the gbest buffers;
//   a >= k   evaluated on a gbest that has moved since: position and velocity roll back to the values k_pso_update kept.
__global__ __launch_bounds__(64) void k_pso_commit(int m, int W, int path_cap, int k, int j, double* pos, double* vel,
                                                   const double* pos_keep, const double* vel_keep, const double* stats, const int* len,
                                                   const int* cells, double* pbest, double* pbest_fit, int* pb_cells, int* pb_len,
                                                   double* gb, double* gstats, int* gpath) {
  const int a = blockIdx.x, t = threadIdx.x;
  if (a >= m) return;
  const size_t w0 = (size_t)a * W * 2;
  if (a >= k) {                                                   // roll back
    for (int i = t; i < W * 2; i += 64) { pos[w0 + i] = pos_keep[w0 + i]; vel[w0 + i] = vel_keep[w0 + i]; }
    return;
  }
  const int L = len[a];
  const double fit = stats[(size_t)a * 5 + 4];
  if (a == j) {                                                   // the gbest moves here (decided by k_pso_scan against the OLD pbest: read before it changes)
    for (int i = t; i < W * 2; i += 64) gb[i] = pos[w0 + i];
    if (t < 5) gstats[t] = stats[(size_t)a * 5 + t];
    for (int i = t; i < L; i += 64) gpath[1 + i] = cells[(size_t)a * path_cap + i];
    if (t == 0) gpath[0] = L;
  }
  const bool better = L > 0 && fit < pbest_fit[a];               // pso.py:210,216
  if (!better) return;
  for (int i = t; i < W * 2; i += 64) pbest[w0 + i] = pos[w0 + i];
  for (int i = t; i < L; i += 64) pb_cells[(size_t)a * path_cap + i] = cells[(size_t)a * path_cap + i];
  __syncthreads();                                                // (every thread has read pbest_fit[a] before it changes)
  if (t == 0) { pbest_fit[a] = fit; pb_len[a] = L; }
}
// pbest -> gbest scan over particles [0, n) of one evaluated batch (pso.py:216-229), one block.  A particle improves
// the gbest when its path is feasible and its fitness is below both its own pbest (:216) and the gbest (:222).
// Asynchronous mode: the FIRST improver (everything before it is final, everything after it has to be re-evaluated
// with the moved gbest); synchronous mode: the first particle with the smallest improving fitness.
// out[0] = index or -1, out[1] = number of particles with status 3 (scratch / path capacity overflow).
__global__ __launch_bounds__(256) void k_pso_scan(int n, const double* stats, const int* len, const int* status, const double* pbf,
                                                  double gbest_fit, int sync_mode, int* out, double* out_fit) {
  __shared__ unsigned long long best[256];
  __shared__ int ovf[256];
  unsigned long long b = ~0ull; int o = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    o += status[i] == 3;
    const double f = stats[(size_t)i * 5 + 4];
    if (len[i] > 0 && f < pbf[i] && f < gbest_fit) {
      if (!sync_mode) { const unsigned long long k = (unsigned long long)i; b = k < b ? k : b; }
      else {
        // order by (fitness, index): non-negative doubles order like their bit patterns; ties on the 64-bit fitness are
        // broken in a second pass below
        const unsigned long long k = (unsigned long long)__double_as_longlong(f);
        b = k < b ? k : b;
      }
    }
  }
  best[threadIdx.x] = b; ovf[threadIdx.x] = o;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { if (best[threadIdx.x + st] < best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + st]; ovf[threadIdx.x] += ovf[threadIdx.x + st]; }
    __syncthreads();
  }
  const unsigned long long w = best[0];
  const int novf = ovf[0];
  __syncthreads();
  if (sync_mode && w != ~0ull) {                                 // the first particle holding that fitness
    unsigned long long bi = ~0ull;
    for (int i = threadIdx.x; i < n; i += 256) {
      const double f = stats[(size_t)i * 5 + 4];
      if (len[i] > 0 && f < pbf[i] && f < gbest_fit && (unsigned long long)__double_as_longlong(f) == w) { bi = (unsigned long long)i < bi ? (unsigned long long)i : bi; }
    }
    best[threadIdx.x] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st && best[threadIdx.x + st] < best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + st]; __syncthreads(); }
  }
  if (threadIdx.x == 0) {
    const unsigned long long idx = (sync_mode && w != ~0ull) ? best[0] : w;
    out[0] = idx == ~0ull ? -1 : (int)idx;
    out[1] = novf;
    out_fit[0] = idx == ~0ull ? PF_INF : stats[(size_t)idx * 5 + 4];
  }
}

// ===========================================================================
// K4: MAACO ant walk
// ===========================================================================
// Deposit bit matrix, blocked by 64-cell stretch: word (cell, w) -- bit a & 63 set iff successful ant 64 w + (a & 63) visited the
// cell -- sits at [cell >> 6][w][cell & 63], W words per cell allotted.  The W chunks of a stretch (512 B each) are one contiguous
// run, which is what the update pass reads: with [w][cell] every chunk of a stretch lay RC * 8 bytes (2 MB at 512^2) from the next,
// and the pass spent its time in address-translation misses (9 us per batch of eight loads).
PF_DEV size_t bits_idx(int cell, int w, int W) { return ((size_t)(cell >> 6) * W + w) * 64 + (cell & 63); }
struct MaacoArgs {
  Grid G;
  const double* tau;      // tau^alpha when alpha != 1 (host-refreshed), else tau
  const double* eta;      // [RC][2]  eta'^beta for (no turn, turn)
  const double* tep;      // [RC][3]  (eta'^beta no turn, tau, eta'^beta turn): ONE 16-byte load at offset 8*turn gives tau and
                          // eta'[turn] (k_maaco_walk8; refreshed by k_pack_tep before every walk)
  unsigned* visit;        // [nslots][vstride] packed tabu sets (Tabu below)
  unsigned* slot_epoch;   // [nslots]
  int wpr, vstride;       // tabu words per grid row, words per slot (= R * wpr)
  int* work; DevCounters* cnt;
  int start, target, iter, num_iterations; double q0;
  unsigned long long seed; int ant0, n, path_cap;
  int* cells; int* len; double* plen; int* turns; int* status;
  // non-null: a successful ant marks the cells it visited in the deposit bit matrix as it finishes (bit a & 63 of word
  // [a >> 6][cell]; atomicOr is order independent, so the matrix equals k_visit_bits') and leaves its deposit Q / L in dep[a]
  // (MAACO.py:307-308): the separate pass over all paths is gone
  unsigned long long* bits; double* dep; double Q;
  // chunk flags of the bit matrix: byte [cell >> 6][word] != 0 whenever some cell of that 64-cell stretch has a bit in that word
  // (plain stores of the same value; k_tau_update reads only flagged 512-byte chunks and clears the flags it used)
  uint8_t* flag; int fstride;
  int groups;             // k_maaco_walk8: 8-lane groups of a wavefront that walk ants (8; fewer = fewer ants per wave, more waves)
};

__global__ void k_pack_tep(int RC, const double* tau, const double* eta, double* tep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RC) { tep[(size_t)i * 3] = eta[(size_t)i * 2]; tep[(size_t)i * 3 + 1] = tau[i]; tep[(size_t)i * 3 + 2] = eta[(size_t)i * 2 + 1]; }
}
typedef double pf_d2u __attribute__((ext_vector_type(2), aligned(8)));

// The tabu set of one resident ant (MAACO.py:281 `visited`), packed: one 32-bit word covers 16 cells of a grid row, its
// upper half is the epoch (the ant's number in this slot) the 16 bits belong to, so a new ant needs no clearing and a word
// of an older ant reads as empty.  2 bits per cell instead of a 4-byte stamp: a 64-B sector holds 256 cells of a row, so the
// three rows a step probes stay in L2 for hundreds of steps (with stamps nearly every step missed to HBM), and a slot is
// R * C / 4 bytes (64 KB at 512^2).  Epochs wrap at 0xFFF0 -> the slot is wiped.
#define PF_TABU_WRAP 0xFFF0u
PF_DEV bool tabu_test(unsigned w, unsigned epoch, int c) { return (w >> 16) == epoch && ((w >> (c & 15)) & 1u); }
PF_DEV unsigned tabu_set(unsigned w, unsigned epoch, int c) { return ((w >> 16) == epoch ? w : epoch << 16) | (1u << (c & 15)); }
// The last two words an ant stored, kept in registers: a probe of one of them takes the register copy, so a step never
// depends on reading back a store the same wave issued one or two instructions ago.
struct TabuLast {
  int i0, i1; unsigned v0, v1;
  PF_DEV void reset() { i0 = i1 = -1; v0 = v1 = 0; }
  PF_DEV unsigned patch(int idx, unsigned w) const { return idx == i0 ? v0 : idx == i1 ? v1 : w; }
  PF_DEV void stored(int idx, unsigned v) { i1 = i0; v1 = v0; i0 = idx; v0 = v; }
};

// ---- 8-lane group arithmetic without LDS round trips (DPP only) ----
// max over the group, in every lane (quad swaps, then the half-row mirror)
// (every lane of these three patterns has a source lane, so the DPP move needs no previous value -- no copy in front of it -- and
// the maximum is the bare instruction: fmax() puts a canonicalising v_max(x, x) in front of every step.  21 -> 9 instructions.
// The operands are never NaN here: products of finite table entries, or the -1.0 of a lane without a candidate.)
template <int CTRL>
PF_DEV double dpp_full_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
PF_DEV double vmax_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PF_DEV double gmax8(double v) {
  v = vmax_raw(v, dpp_full_d<0xB1>(v));    // quad_perm:[1,0,3,2]
  v = vmax_raw(v, dpp_full_d<0x4E>(v));    // quad_perm:[2,3,0,1]
  v = vmax_raw(v, dpp_full_d<0x141>(v));   // row_half_mirror
  return v;
}
// lane j of each group: ((a0 + a1) + a2 ... ) + aj, added in exactly that order (a sequential Python sum / numpy cumsum)
PF_DEV double gscan8(double a, int k) {
  double x = a;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    double y = dpp_d<0x111, 0xF>(x);   // row_shr:1
    y = k == 0 ? 0.0 : y;              // (lane 8 of a row would read the other group's lane 7)
    x = y + a;
  }
  return x;
}
// the value of the group's lane 7 in every lane
PF_DEV double glast8(double v) {
  const double lo = dpp_d<0x157, 0xF>(v), hi = dpp_d<0x15F, 0xF>(v);   // row_newbcast:7 / :15
  return (lane_id() & 8) ? hi : lo;
}
__global__ __launch_bounds__(64) void k_maaco_walk(MaacoArgs p) {
  const int lane = lane_id();
  const Grid& G = p.G;
  const int R = G.R, C = G.C, RC = R * C;
  unsigned* visit = p.visit + (size_t)blockIdx.x * p.vstride;
  unsigned epoch = p.slot_epoch[blockIdx.x];
  const int WPR = p.wpr;
  TabuLast tl;
  const int k = lane & 7;
  const int mdr = AM_DR[k], mdc = AM_DC[k];
  const unsigned hbit = 1u << AM_TO_HM[k];
  const int sr = row_of(G, p.start), sc = p.start - sr * C;
  const int tr = row_of(G, p.target), tc = p.target - tr * C;
  // MAACO.py:147-150 start->target orientation: static per move
  const int vrS = tr - sr, vcS = tc - sc;
  const bool o1 = !((vcS > 0 && mdc < 0) || (vcS < 0 && mdc > 0) || (vrS > 0 && mdr < 0) || (vrS < 0 && mdr > 0));
  const unsigned O1 = (unsigned)(__ballot(o1 && lane < 8) & 0xFF);
  const double mcost = (mdr != 0 && mdc != 0) ? PF_SQRT2 : 1.0;
  const uint64_t q0_bits = p.q0 >= 1.0 ? ~0ull : (p.q0 < 0.0 ? 0ull : (uint64_t)(p.q0 * 9007199254740992.0));   // q <= q0 as an integer test (k_maaco_walk8)
  unsigned long long steps_tot = 0, cand_tot = 0, cells_tot = 0, ovf_tot = 0;
  for (;;) {
    const int a = next_work(p.work, lane);
    if (a >= p.n) break;
    epoch += 1;
    if (epoch >= PF_TABU_WRAP) {
      for (int i = lane; i < p.vstride; i += 64) visit[i] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      epoch = 1;
    }
    Rng g; g.init(p.seed, DOM_MAACO, (unsigned long long)p.iter, (unsigned long long)(p.ant0 + a));
    int* out = p.cells + (size_t)a * p.path_cap;
    int cr = sr, cc = sc, n = 1, prev_k = -1, nturn = 0, rc = 0;
    double plen = 0.0;
    tl.reset();
    {
      const int wi = sr * WPR + (sc >> 4); const unsigned wv = tabu_set(0u, epoch, sc);
      if (lane == 0) { out[0] = p.start; visit[wi] = wv; }
      tl.stored(wi, wv);
    }
    const int max_steps = RC * 2;                                  // MAACO.py:283 (<= 2^25)
    int steps = 0;
    while (!(cr == tr && cc == tc) && steps < max_steps) {
      const int cur = cr * C + cc;
      const int nr = cr + mdr, nc = cc + mdc;
      const bool inb = lane < 8 && nr >= 0 && nr < R && nc >= 0 && nc < C;
      const int nidx = nr * C + nc;
      const int turn = (prev_k >= 0 && k != prev_k) ? 1 : 0;     // MAACO.py:184-195
      unsigned vw = 0, mmask = 0; double tv = 0.0, ev = 0.0;
      const int widx = nr * WPR + (nc >> 4);
      if (inb) { vw = tl.patch(widx, visit[widx]); tv = p.tau[nidx]; ev = p.eta[(size_t)nidx * 2 + turn]; }
      else if (lane == 9) mmask = G.mm[cur];
      const unsigned M = (unsigned)bcast_i((int)mmask, 9);
      // valid, not tabu, no corner cut (:93-95,:100-120); the low byte of the mask = lanes 0..7
      const unsigned mall = (unsigned)(B((unsigned)nr < (unsigned)R) & B((unsigned)nc < (unsigned)C) & B((M & hbit) != 0u) &
                                       ~(B((vw >> 16) == epoch) & B(((vw >> (nc & 15)) & 1u) != 0u))) & 0xFFu;
      // strategy 2 orientation: current -> target (:152-157)
      // (RowMask[sign vr] & ColMask[sign vc] from two constants: see k_maaco_walk8)
      const unsigned ur = (unsigned)min(max(tr + 1 - cr, 0), 2), uc = (unsigned)min(max(tc + 1 - cc, 0), 2);
      const unsigned O2 = __builtin_amdgcn_ubfe(0xF8FF1Fu, ur << 3, 8u) & __builtin_amdgcn_ubfe(0xD6FF6Bu, uc << 3, 8u);
      unsigned cand = mall & O1;                                  // :165
      if (!cand) cand = mall & O2;                                // :168-169
      if (!cand) cand = mall;                                     // :172-180
      if (!cand) { rc = 1; break; }                               // :287-288
      const int ncand = __builtin_popcount(cand);
      cand_tot += ncand;
      const bool cmine = lane < 8 && ((cand >> k) & 1u);
      // ONE mix serves the step (see k_maaco_walk8): lane j mixes word j + 1 of the ant's stream -- q, the first word of the choice
      // and six more for random.choice's rejection loop, which used to cost a full mix64 per extra draw.
      const uint64_t Wk = g.peek64(1 + (uint64_t)k);
      const bool greedy = (__builtin_amdgcn_ballot_w64((Wk >> 11) <= q0_bits) & 1ull) != 0;   // :232 q = word 1 (lane 0); q <= q0 as integers
      const double attr = cmine ? tv * ev : 0.0;                  // :238 tau^alpha * eta'^beta; the other lanes add exact zeros
      const double Mx = gmax8(cmine ? attr : -1.0);               // (lanes 0..7 are one 8-lane group)
      int pick = 0;
      unsigned msel = cand;                                       // the set random.choice draws from
      bool chosen = false;
      if (greedy) {                                               // :241-250 running max with absolute tolerance, closed form (k_maaco_walk8)
        const unsigned eq = (unsigned)(B(attr == Mx)) & cand;      // (cand = the lanes 0..7 that hold a candidate)
        if (!eq) { rc = 1; break; }
        msel = (unsigned)(B(k >= __builtin_ctz(eq)) & B(fabs(attr - Mx) < 1e-9)) & cand;
      } else if (!(bcast_d(Mx, 0) * 8.0 < 5e-10)) {               // (else the ordered sum is below 1e-9 whatever its rounding: k_maaco_walk8)
        const double sum = bcast_d(gscan8(attr, k), 7);           // :252 sum() in candidate order (ordered 8-lane scan)
        if (!(sum < 1e-9)) {                                      // else :253-254: random.choice over all candidates
          const double p0 = attr / sum;                           // :255
          double pj = p0;
          if (!(sum < 1.0e300)) {                                 // :256-258 cannot renormalise for a finite sum (see k_maaco_walk8)
            const double ps = bcast_d(gscan8(p0, k), 7);
            if (fabs(ps - 1.0) > 1e-6) pj = p0 / ps;
          }
          const double u = Rng::to_unit(((uint64_t)(unsigned)bcast_i((int)(Wk >> 32), 1) << 32) | (unsigned)bcast_i((int)Wk, 1));   // :259 np.random.choice -> one random_sample: word 2
          const double mine = gscan8(pj, k);                      // cdf = cumsum(p); cdf /= cdf[-1]
          const double last = bcast_d(mine, 7);
          const unsigned tm = (unsigned)__ballot(cmine && mine / last <= u) & 0xFFu;   // searchsorted(cdf, u, side='right')
          int idx = tm ? __builtin_popcount(cand & ((2u << (31 - __builtin_clz(tm))) - 1u)) : 0;
          if (idx > ncand - 1) idx = ncand - 1;
          pick = nth_set_bit(cand, idx);
          chosen = true;
          g.advance(2);
        }
      }
      if (!chosen) {
        // random.choice(set) = set[_randbelow(n)]: the first of words 2..8 whose top bit_length(n) bits are below n (a ballot)
        const unsigned nsel = (unsigned)__builtin_popcount(msel);
        const int kb = 32 - __builtin_clz(nsel);
        const unsigned rk = (unsigned)(Wk >> (64 - kb));
        const unsigned acc = (unsigned)B(rk < nsel) & 0xFEu;
        unsigned r;
        if (acc) { const int first = __builtin_ctz(acc); r = (unsigned)bcast_i((int)rk, first); g.advance(1u + (unsigned)first); }
        else { g.advance(8); do { r = (unsigned)(g.next64() >> (64 - kb)); } while (r >= nsel); }
        pick = nth_set_bit(msel, (int)r);
      }
      plen += bcast_d(mcost, pick);                               // :293
      if (prev_k >= 0 && pick != prev_k) nturn += 1;              // :264-276 counted on the fly
      prev_k = pick;
      cr += bcast_i(mdr, pick); cc += bcast_i(mdc, pick);
      if (n >= p.path_cap) { rc = 3; break; }
      {
        const int wi = cr * WPR + (cc >> 4);                        // = lane `pick`'s word, which it holds up to date
        const unsigned wv = tabu_set((unsigned)bcast_i((int)vw, pick), epoch, cc);
        if (lane == 0) { out[n] = cr * C + cc; visit[wi] = wv; }
        tl.stored(wi, wv);
      }
      n += 1; steps += 1;
    }
    if (rc == 0 && !(cr == tr && cc == tc)) rc = 2;               // :301-302 step cap
    steps_tot += (unsigned long long)steps;
    if (lane == 0) {
      p.len[a] = rc == 0 ? n : 0;
      p.plen[a] = rc == 0 ? plen : PF_INF;
      p.turns[a] = rc == 0 ? nturn : -1;
      p.status[a] = rc;
    }
    if (p.bits) {
      const bool good = rc == 0 && n > 0 && plen > 1e-6;           // MAACO.py:307
      if (lane == 0) p.dep[a] = good ? p.Q / plen : 0.0;            // :308
      if (good) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // lane 0 wrote the path
        const unsigned long long bit = 1ull << (a & 63);
        uint8_t* fl = p.flag + (a >> 6);
        for (int i = lane; i < n; i += 64) { const int c = out[i]; atomicOr(&p.bits[bits_idx(c, a >> 6, p.fstride)], bit); fl[(size_t)(c >> 6) * p.fstride] = 1; }
      }
    }
    cells_tot += rc == 0 ? n : 0; ovf_tot += rc == 3;
  }
  if (lane == 0) {
    p.slot_epoch[blockIdx.x] = epoch;
    atomicAdd(&p.cnt->steps, steps_tot); atomicAdd(&p.cnt->candidates, cand_tot); atomicAdd(&p.cnt->path_cells, cells_tot);
    if (ovf_tot) atomicAdd(&p.cnt->overflow, ovf_tot);
  }
}

// ---------------------------------------------------------------------------
// K4, packed form: EIGHT ants per wavefront.  A walk step only ever uses 8 lanes (the 8 moves), so each group
// of 8 lanes walks its own ant; "per-ant uniform" values are replicated in the group's lanes, broadcasts are
// ds_bpermute inside the group, candidate masks are 8-bit slices of the wave ballot.  A group that finishes
// its ant emits the result and immediately fetches the next ant from the queue inside the same loop, so no
// lanes idle until the queue is empty.  Same draws, same arithmetic, same order as k_maaco_walk.
// ---------------------------------------------------------------------------
PF_DEV unsigned gballot8(bool p) { return (unsigned)(__builtin_amdgcn_ballot_w64(p) >> (lane_id() & 56)) & 0xFFu; }
// (g8: my group's byte of a wave mask; compound predicates as B(a) & B(b): pf_device.h)
PF_DEV unsigned g8(pf_u64 m) { return (unsigned)(m >> (lane_id() & 56)) & 0xFFu; }
PF_DEV int gbcast8_i(int v, int k) { return __builtin_amdgcn_ds_bpermute(((lane_id() & 56) + k) << 2, v); }
// index of the idx-th set bit of an 8-bit mask (lane k tests bit k)
PF_DEV int gnth8(unsigned m, int idx, int k) {
  return __builtin_ctz(g8(B(((m >> k) & 1u) != 0u) & B(__builtin_popcount(m & ((1u << k) - 1u)) == idx)) | 0x100u);
}
PF_DEV double gbcast8_d(double v, int k) {
  const int lo = gbcast8_i(__double2loint(v), k), hi = gbcast8_i(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// eight ants per wavefront: the 8 lanes of a group hold the 8 moves of one ant
// AHEAD, the form for batches that leave a SIMD ONE wavefront (8 192 ants: a step then waits ~2 000 clocks for memory, and nothing
// else runs meanwhile): all of a step's loads are issued before anything waits for one of them (the compiler otherwise sinks the
// pheromone load behind the candidate test: a second round trip), and every move lane also asks for the pheromone record and
// the tabu word two steps ahead in its direction -- where the NEXT step's neighbours live -- right behind the step's own loads
// (memory returns in order: they cannot delay them; results unused).  A/B on one box (M evals/s): 8 192 ants @1024^2 2.87 -> 3.04;
// 16 384 ants @512^2 (two wavefronts per SIMD, issue-bound) 8.75 -> 8.50 -- so the host picks the form by occupancy.
template <bool AHEAD>
__global__ __launch_bounds__(64) void k_maaco_walk8(MaacoArgs p) {
  const Grid& G = p.G;
  const int R = G.R, C = G.C, RC = R * C;
  const int lane = lane_id();
  const int k = lane & 7, grp = lane >> 3;
  const int slot = blockIdx.x * 8 + grp;
  unsigned* visit = p.visit + (size_t)slot * p.vstride;
  unsigned epoch = p.slot_epoch[slot];
  const int WPR = p.wpr;
  TabuLast tl; tl.reset();
  const int mdr = AM_DR[k], mdc = AM_DC[k];
  const unsigned hbit = 1u << AM_TO_HM[k];
  const int sr = row_of(G, p.start), sc = p.start - sr * C;
  const int tr = row_of(G, p.target), tc = p.target - tr * C;
  const int vrS = tr - sr, vcS = tc - sc;
  const bool o1 = !((vcS > 0 && mdc < 0) || (vcS < 0 && mdc > 0) || (vrS > 0 && mdr < 0) || (vrS < 0 && mdr > 0));
  const unsigned O1 = gballot8(o1);
  const int max_steps = RC * 2;                                    // MAACO.py:283
  // q0 is in [0.01, 0.99] (MAACO.py:226): q0 2^53 is exact, its floor the largest 53-bit draw that still takes the greedy rule
  // (as `draw < q0_lim`: a `<=` against a run-time bound compiles to two compares, one for the bound's all-ones case)
  const uint64_t q0_lim = p.q0 >= 1.0 ? (1ull << 53) : (p.q0 < 0.0 ? 0ull : (uint64_t)(p.q0 * 9007199254740992.0) + 1ull);
  unsigned long long steps_tot = 0, cand_tot = 0, cells_tot = 0, ovf_tot = 0;
  // per-ant state (replicated in the 8 lanes of the group)
  int a = -1, cr = 0, cc = 0, n = 0, prev_k = -1, nturn = 0, rc = 0;
  int steps = 0;                                                   // (<= 2 R C <= 2^25)
  double plen = 0.0;
  Rng g; g.key = 0; g.ctr = 0; g.kc = 0;
  int* out = p.cells;
  bool alive = grp < p.groups;
#ifdef PF_WALK_PROBE
  unsigned long long pr_wait = 0, pr_rounds = 0, pr_mark = 0, pr_sel0 = 0, pr_sel1 = 0, pr_head = 0, pr_emit = 0, pr_upd = 0, pr_loop = 0, pr_end = 0, pr_act = 0; const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime();
#endif
  // fetch + initialise the next ant of this group (alive = false when the queue is empty)
  auto fetch = [&]() {
    int w = 0;
    if (k == 0) w = atomicAdd(p.work, 1);
    w = gbcast8_i(w, 0);
    if (w >= p.n) alive = false;
    else {
      a = w;
      epoch += 1;
      if (epoch >= PF_TABU_WRAP) {
        for (int i = k; i < p.vstride; i += 8) visit[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        epoch = 1;
      }
      g.init(p.seed, DOM_MAACO, (unsigned long long)p.iter, (unsigned long long)(p.ant0 + a));
      out = p.cells + (size_t)a * p.path_cap;
      cr = sr; cc = sc; n = 1; prev_k = -1; nturn = 0; rc = 0; plen = 0.0; steps = 0;
      tl.reset();
      const int wi = sr * WPR + (sc >> 4); const unsigned wv = tabu_set(0u, epoch, sc);
      if (k == 0) { out[0] = p.start; visit[wi] = wv; }
      tl.stored(wi, wv);
    }
  };
  if (alive) fetch();
  // One wave-uniform test per round: "did an ant finish?" (one round in a hundred).  Emitting, deposit marking, fetching the group's
  // next ant and the end-of-queue test all sit behind it; a round that only steps pays for nothing else.  (A wave none of whose
  // groups got an ant -- the queue was drained by the others' first fetches -- never enters the loop: every exit is behind any_fin.)
  unsigned pft0 = 0, pft1 = 0;
  if (__ballot(alive)) for (;;) {
    bool done = alive & (((cr == tr) & (cc == tc)) | (steps >= max_steps));
    const unsigned pft0_prev = pft0, pft1_prev = pft1;
#ifdef PF_WALK_PROBE
    __builtin_amdgcn_sched_barrier(0); const unsigned long long pr_r0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
    bool pr_stepped = false; unsigned long long pr_u0 = 0;
    if (pr_end) pr_loop += pr_r0 - pr_end;
#endif
    if (alive && !done) {
      // (rows, columns and cells fit 24 bits -- R, C <= 4096 --: v_mad_u32_u24 at full rate and 32-bit byte offsets on a scalar
      // base, where `int` indices cost a quarter-rate 64-bit multiply-add, a sign extension and a 64-bit add per load)
      const unsigned cur = __umul24((unsigned)cr, (unsigned)C) + (unsigned)cc;
      const int nr = cr + mdr, nc = cc + mdc;
      const bool inb = ((unsigned)nr < (unsigned)R) & ((unsigned)nc < (unsigned)C);
      const unsigned nidx = __umul24((unsigned)nr, (unsigned)C) + (unsigned)nc;
      const int turn = ((prev_k >= 0) & (k != prev_k)) ? 1 : 0;    // MAACO.py:184-195
      unsigned vw = 0; double tv = 0.0, ev = 0.0;
      const unsigned M = *((const uint8_t*)G.mm + (size_t)cur);
      const unsigned widx = __umul24((unsigned)nr, (unsigned)WPR) + ((unsigned)nc >> 4);
      // Every step that finds a candidate draws q (:232) and then at least one more 64-bit word (random.choice's first
      // getrandbits at :250 / :254, or numpy's random_sample at :259): both words are mixed here, before the loads below
      // are waited for, and the counter advances only when the step gets that far.
      // ONE mix serves the whole step: lane j of the ant's group mixes word j + 1 of the stream (the same instructions in every
      // lane), i.e. the eight next words at once -- q, the first word of the choice, and six more for random.choice's rejection
      // loop (_randbelow redraws while the k-bit value is >= n: every second draw for n = 1, every fourth for n = 3), which used to
      // cost the wave a full mix64 (~25 instructions, eight quarter-rate multiplies) per extra draw of its unluckiest ant.
      uint64_t Wk = 0;
      if (!AHEAD) Wk = g.peek64(1 + (uint64_t)k);
#ifdef PF_WALK_PROBE
      __builtin_amdgcn_sched_barrier(0); const unsigned long long pr_ta = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
      pr_head += pr_ta - pr_r0; pr_stepped = true; pr_act += 1;   // (per-lane copies: lane 0 reports, so the in-step stamps cover the rounds in which group 0 stepped)
#endif
      {
        // unconditional loads (a move that leaves the map reads cell 0 and is rejected by `inb` below): a branch around them costs
        // a scalar round trip on the mask, and loads under a branch keep the compiler from counting them
        const unsigned widx_c = inb ? widx : 0u, nidx_c = inb ? nidx : 0u;
        const unsigned vw_raw = visit[widx_c];
        // one divergent vector load less per step: tau and eta'[turn] in one
        const unsigned toff = __umul24(nidx_c, 24u) + ((unsigned)turn << 3);
        const pf_d2u te = *(const pf_d2u*)((const char*)p.tep + (size_t)toff);
        if (AHEAD) {
          const int r2 = nr + mdr, c2 = nc + mdc;
          const bool in2 = ((unsigned)r2 < (unsigned)R) & ((unsigned)c2 < (unsigned)C);
          const unsigned t2 = in2 ? __umul24((unsigned)r2, (unsigned)C) + (unsigned)c2 : nidx_c;
          const unsigned w2 = in2 ? __umul24((unsigned)r2, (unsigned)WPR) + ((unsigned)c2 >> 4) : widx_c;
          pft0 = *(const unsigned*)((const char*)p.tep + (size_t)__umul24(t2, 24u));
          pft1 = visit[w2];
          __builtin_amdgcn_sched_barrier(0);                        // (all of the step's loads are issued before anything waits for one of them)
        }
        vw = tl.patch((int)widx_c, vw_raw);
        tv = turn ? te.x : te.y; ev = turn ? te.y : te.x;
        if (AHEAD) asm volatile("" :: "v"(pft0_prev), "v"(pft1_prev));   // (the previous step's touches: older than the loads just waited for)
      }
      if (AHEAD) Wk = g.peek64(1 + (uint64_t)k);                    // (in the shadow of the loads just issued)
#ifdef PF_WALK_PROBE
      { __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        pr_wait += __builtin_amdgcn_s_memtime() - pr_ta; }
#endif
      // (bitwise on purpose: the short-circuit forms compiled to branches, each a scalar round trip on a freshly written lane mask)
      const unsigned mall = g8(B((unsigned)nr < (unsigned)R) & B((unsigned)nc < (unsigned)C) & B((M & hbit) != 0u) &
                               ~(B((vw >> 16) == epoch) & B(((vw >> (nc & 15)) & 1u) != 0u)));
      // strategy 2 orientation, current -> target (:152-157): a move is kept iff its row step does not oppose sign(vr) and its column
      // step does not oppose sign(vc) -- the eight-move mask is RowMask[sign vr] & ColMask[sign vc], two bit-field extracts from
      // constants (moves 0..7 = AM_DR / AM_DC order) instead of eight compares, their scalar mask logic and a ballot
      const unsigned ur = (unsigned)min(max(tr + 1 - cr, 0), 2), uc = (unsigned)min(max(tc + 1 - cc, 0), 2);
      const unsigned O2 = __builtin_amdgcn_ubfe(0xF8FF1Fu, ur << 3, 8u) & __builtin_amdgcn_ubfe(0xD6FF6Bu, uc << 3, 8u);
      unsigned cand = mall & O1;                                    // :165
      if (!cand) cand = mall & O2;                                  // :168-169
      if (!cand) cand = mall;                                       // :172-180
      // (AHEAD: the dead end "uses" the pheromone record too, so that its load stays in front of the candidate test)
      if (!cand) { rc = 1; done = true; if (AHEAD) asm volatile("" :: "v"(tv), "v"(ev)); }   // :287-288
      else {
        const int ncand = __builtin_popcount(cand);
        cand_tot += ncand;
        const bool cmine = (cand >> k) & 1u;
        // :232 q = word 1 of the step (lane 0 of the group holds it): one ballot tells the group which rule applies
        // (q = (w >> 11) 2^-53 exactly, so q <= q0 iff w >> 11 <= floor(q0 2^53): two integer instructions instead of the conversion)
        const bool greedy = (g8(B((Wk >> 11) < q0_lim)) & 1u) != 0;
        const double attr = cmine ? tv * ev : 0.0;                  // :238; the other lanes add an exact zero to the ordered sums below
        int pick = 0;
#ifdef PF_WALK_PROBE
        __builtin_amdgcn_sched_barrier(0); const unsigned long long pr_s0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
        pr_sel0 += pr_s0 - pr_ta;
#endif
        // Both rules end in random.choice over a set of candidates: the greedy rule (:241-250) over its tie set, the other one
        // (:252-254) over all candidates whenever the attractiveness sum is below 1e-9 -- on the 512^2 and 1024^2 maps with beta = 7
        // that is EVERY non-greedy step (eta'^7 ~ 1e-19: SURVEY H8 measured 0 % roulette picks).  So the two rules only differ in
        // the set, and ONE choice serves every ant of the wave; the roulette proper runs behind a wave-uniform test.
        // The largest attractiveness among the candidates is the greedy rule's maximum AND a bound on the other rule's sum: the
        // ordered sum of at most 8 non-negative terms, none above Mx, is at most 8 Mx (1 + 2^-53)^7 -- with 8 Mx < 5e-10 it is
        // below 1e-9 whatever its rounding, and the 7-step ordered scan need not run.
        const double Mx = gmax8(cmine ? attr : -1.0);
        // :241-250, the running maximum with its absolute tolerance, in closed form: the tie set restarts at the FIRST occurrence of
        // the maximum (`attr > max` drops every earlier member there) and from then on collects the candidates within 1e-9 of it
        // (none can exceed it).  NaN neither restarts nor joins, as in the loop.
        const pf_u64 cmm = B(((cand >> k) & 1u) != 0u);
        const unsigned eq = g8(cmm & B(attr == Mx));
        const unsigned bm = g8(cmm & B(k >= __builtin_ctz(eq | 0x100u)) & B(fabs(attr - Mx) < 1e-9));
        const bool tiny = Mx * 8.0 < 5e-10;                         // (a NaN maximum compares false: the sum decides)
        if (greedy && !eq) { rc = 1; done = true; }
        bool chosen = false;                                        // the roulette proper picked (two words of the stream: q and u)
        if (B(true) & ~B(greedy) & ~B(tiny)) {
          // The sums of :252-259 run over the candidates in candidate order.  Candidate j lives in lane j, so each is one
          // ordered 8-lane scan (7 dependent DPP steps, no LDS round trip per candidate); every quotient belongs to one
          // candidate and is computed in its lane.  (The scans run for the whole wave; only the ants that need them use the result.)
          const double sum = glast8(gscan8(attr, k));               // :252
          if (!greedy && !tiny && !(sum < 1e-9)) {                  // else :253-254: random.choice over all candidates, below
            const double p0 = attr / sum;                           // :255 probabilities[j]
            // :256-258 renormalise when |sum(probabilities) - 1| > 1e-6.  For a finite sum of at most 8 non-negative terms
            // that never happens: sum = S(1 + e), |e| <= 7u (u = 2^-53), every quotient is a_j / sum (1 + d_j), |d_j| <= u
            // (an underflowing quotient errs by < 2^-1074), and adding them in order costs another 7u, so
            // |sum(probabilities) - 1| < 16u ~ 2e-15.  Only an overflowed sum takes the general route.
            double pj = p0;
            if (!(sum < 1.0e300)) {
              const double ps = glast8(gscan8(p0, k));              // :256 sum(probabilities)
              if (fabs(ps - 1.0) > 1e-6) pj = p0 / ps;              // :257-258
            }
            const double u = gbcast8_d(Rng::to_unit(Wk), 1);        // :259 numpy.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; word 2 of the step
            const double mine = gscan8(pj, k);                      // cdf[position of my move]
            const double last = glast8(mine);
            const unsigned tm = gballot8(cmine && mine / last <= u);   // searchsorted(cdf, u, side="right")
            int idx = tm ? __builtin_popcount(cand & ((2u << (31 - __builtin_clz(tm))) - 1u)) : 0;
            if (idx > ncand - 1) idx = ncand - 1;
            pick = gnth8(cand, idx, k);
            chosen = true;
            g.advance(2);
          }
        }
        {
          // random.choice(set) = set[_randbelow(n)]: kb = n.bit_length() bits of a word, redrawn while >= n (:250 / :254).  Word j + 1
          // of the step sits in lane j: every lane tests ITS word, the first acceptable one (a ballot) is the draw.
          const unsigned msel = greedy ? bm : cand;
          const unsigned nsel = (unsigned)__builtin_popcount(msel) | (msel ? 0u : 1u);   // (>= 1: a failed ant's value is never used)
          const int kb = 32 - __builtin_clz(nsel);
          unsigned rk = (unsigned)(Wk >> (64 - kb));
          const unsigned acc = g8(B(rk < nsel)) & 0xFEu;             // (word 1 is q: lanes k >= 1)
          unsigned r = (unsigned)gbcast8_i((int)rk, __builtin_ctz(acc | 0x80u));
          if (!chosen) g.advance(acc ? 1u + (unsigned)__builtin_ctz(acc) : 8u);
          if (B(acc == 0u)) {                                        // all seven rejected (n = 1: once in 128 steps): draw on, one word at a time
            if (!chosen && !done && acc == 0u) { do { r = (unsigned)(g.next64() >> (64 - kb)); } while (r >= nsel); }
          }
          if (!chosen) pick = gnth8(msel, (int)r, k);
        }
#ifdef PF_WALK_PROBE
        int t_ = pick; asm volatile("" : "+v"(t_)); __builtin_amdgcn_sched_barrier(0); const unsigned long long pr_s1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
        pr_sel1 += pr_s1 - pr_s0; pr_u0 = pr_s1;
#endif
        if (!done) {
          plen += ((0xA5u >> pick) & 1u) ? PF_SQRT2 : 1.0;         // :293 (moves 0, 2, 5, 7 are the diagonals)
          if (prev_k >= 0 && pick != prev_k) nturn += 1;
          prev_k = pick;
          cr += (int)((0xA940u >> (2 * pick)) & 3u) - 1;            // AM_DR[pick] + 1, two bits a move
          cc += (int)((0x9224u >> (2 * pick)) & 3u) - 1;            // AM_DC[pick] + 1
          {
            // a full path row (rc 3) is the exception: predicated, not a branch of its own
            const bool ovf = n >= p.path_cap;
            rc = ovf ? 3 : rc; done = ovf;
            const int wi = cr * WPR + (cc >> 4);                   // = lane `pick`'s word, which it holds up to date
            const unsigned wv = tabu_set((unsigned)gbcast8_i((int)vw, pick), epoch, cc);
            if (k == 0 && !ovf) { out[n] = cr * C + cc; visit[wi] = wv; }
            if (!ovf) tl.stored(wi, wv);
            n += ovf ? 0 : 1; steps += ovf ? 0 : 1;
          }
        }
      }
    }
#ifdef PF_WALK_PROBE
    __builtin_amdgcn_sched_barrier(0); const unsigned long long pr_e0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
    if (pr_u0) pr_upd += pr_e0 - pr_u0;
#endif
    // an ant finishes in about one round of a hundred: everything that only a finished ant needs sits behind ONE wave-uniform test
    const bool any_fin = __ballot(alive && done) != 0ull;
    if (any_fin && alive && done) {                                 // emit, then fetch a new ant next round
      if (rc == 0 && !(cr == tr && cc == tc)) rc = 2;               // :301-302 step cap
      steps_tot += (unsigned long long)steps;
      if (k == 0) {
        p.len[a] = rc == 0 ? n : 0;
        p.plen[a] = rc == 0 ? plen : PF_INF;
        p.turns[a] = rc == 0 ? nturn : -1;
        p.status[a] = rc;
      }
      cells_tot += rc == 0 ? n : 0; ovf_tot += rc == 3;
    }
#ifdef PF_WALK_PROBE
    pr_rounds += 1; const unsigned long long pr_m0 = __builtin_amdgcn_s_memtime(); pr_emit += pr_m0 - pr_e0;
#endif
    if (any_fin && p.bits) {
      // the ants that finished in this round mark their deposits: the WHOLE wave walks each finished path (64 cells a round;
      // the other groups would only wait for a group that marked alone, 8 cells a round)
      const bool fin = alive && done;
      const bool good = fin && rc == 0 && n > 0 && plen > 1e-6;    // MAACO.py:307
      if (fin && k == 0) p.dep[a] = good ? p.Q / plen : 0.0;        // :308
      unsigned long long gm = __ballot(good && k == 0);
      if (gm) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // lane k == 0 wrote the path
        for (; gm; gm &= gm - 1) {
          const int l = __builtin_ctzll(gm);
          const int aa = bcast_i(a, l), nn = bcast_i(n, l);
          const int* oo = p.cells + (size_t)aa * p.path_cap;
          const unsigned long long bit = 1ull << (aa & 63);
          uint8_t* fl = p.flag + (aa >> 6);
          for (int i = lane; i < nn; i += 256) {                    // four cell loads in flight, then their (unwaited) atomics
            int c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = i + 64 * u < nn ? oo[i + 64 * u] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (c4[u] >= 0) {
              __hip_atomic_fetch_or(&p.bits[bits_idx(c4[u], aa >> 6, p.fstride)], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              fl[(size_t)(c4[u] >> 6) * p.fstride] = 1;
            }
          }
        }
      }
    }
    if (any_fin) {
      if (alive && done) fetch();                                   // the group's next ant starts in the next round
      if (!__ballot(alive)) break;
    }
#ifdef PF_WALK_PROBE
    pr_end = __builtin_amdgcn_s_memtime(); pr_mark += pr_end - pr_m0;
#endif
  }
#ifdef PF_WALK_PROBE
  // (diagnostic build: wave clocks in the counters the MAACO path leaves unused -- scripts/probe_walk_split.py)
  if (lane == 0) { atomicAdd(&p.cnt->pops, pr_wait); atomicAdd(&p.cnt->pushes, __builtin_amdgcn_s_memtime() - pr_t0); atomicAdd(&p.cnt->nbr, pr_rounds);
                   atomicAdd(&p.cnt->deckey, pr_mark); atomicMax(&p.cnt->pruned, __builtin_amdgcn_s_memtime() - pr_t0);
                   atomicAdd(&p.cnt->settled, pr_sel0); atomicAdd(&p.cnt->sequential, pr_sel1);
                   atomicAdd(&p.cnt->candidates, pr_head); atomicAdd(&p.cnt->path_cells, pr_act); (void)pr_emit; atomicAdd(&p.cnt->steps, pr_upd); atomicAdd(&p.cnt->overflow, pr_loop); }
#endif
  if (k == 0) {
    p.slot_epoch[slot] = epoch;
#ifndef PF_WALK_PROBE
    atomicAdd(&p.cnt->candidates, cand_tot); atomicAdd(&p.cnt->path_cells, cells_tot); atomicAdd(&p.cnt->steps, steps_tot);
    if (ovf_tot) atomicAdd(&p.cnt->overflow, ovf_tot);
#endif
  }
}

// Best ant of an iteration, MAACO.py:343-349, without the sequential loop: `L < best` only ever fires up to the first
// occurrence p of the minimum length, which resets (idx, turns) to ant p; afterwards only ants within 1e-9 of that
// minimum with strictly fewer turns take over, so the result is the first ant, among p and the later near-minimum ants,
// that attains their smallest turn count.  out = {best_len, best_turns (inf for none), best_idx as a double (-1 none)}.
PF_DEV void maaco_best_block(int n, const double* plen, const int* turns, double* out) {
  __shared__ unsigned long long red[1024];
  __shared__ double sL; __shared__ int sP;
  const int t = threadIdx.x;
  // (1) minimum length, non-negative doubles compare like their bit patterns (+inf included)
  unsigned long long b = ~0ull;
  for (int i = t; i < n; i += 1024) { const unsigned long long k = (unsigned long long)__double_as_longlong(plen[i]); b = k < b ? k : b; }
  red[t] = b; __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (t < st && red[t + st] < red[t]) red[t] = red[t + st]; __syncthreads(); }
  const unsigned long long lb = red[0];
  __syncthreads();
  // (2) its first occurrence
  unsigned long long pi = ~0ull;
  for (int i = t; i < n; i += 1024) if ((unsigned long long)__double_as_longlong(plen[i]) == lb) { pi = (unsigned long long)i < pi ? (unsigned long long)i : pi; }
  red[t] = pi; __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (t < st && red[t + st] < red[t]) red[t] = red[t + st]; __syncthreads(); }
  if (t == 0) { sL = __longlong_as_double((long long)lb); sP = (int)red[0]; }
  __syncthreads();
  const double Lmin = sL; const int p = sP;
  if (n == 0 || Lmin == PF_INF) { if (t == 0) { out[0] = PF_INF; out[1] = PF_INF; out[2] = -1.0; } return; }
  // (3) smallest turn count among p and the later ants within 1e-9 of Lmin, then its first holder: key = turns << 32 | index
  unsigned long long kb = ~0ull;
  for (int i = p + t; i < n; i += 1024) {
    if (i == p || fabs(plen[i] - Lmin) < 1e-9) {
      const unsigned long long T = turns[i] < 0 ? 0x7FFFFFFFull : (unsigned long long)turns[i];
      const unsigned long long k = (T << 32) | (unsigned long long)i;
      kb = k < kb ? k : kb;
    }
  }
  red[t] = kb; __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (t < st && red[t + st] < red[t]) red[t] = red[t + st]; __syncthreads(); }
  if (t == 0) {
    const unsigned long long k = red[0];
    const unsigned T = (unsigned)(k >> 32);
    out[0] = Lmin; out[1] = T == 0x7FFFFFFFu ? PF_INF : (double)T; out[2] = (double)(unsigned)(k & 0xFFFFFFFFull);
  }
}
__global__ __launch_bounds__(1024) void k_maaco_best(int n, const double* plen, const int* turns, double* out) {
  maaco_best_block(n, plen, turns, out);
}

// ===========================================================================
// K5: pheromone update (evaporate / ordered deposit / clip)
// ===========================================================================
__global__ void k_tau_evaporate(double* tau, int RC, double keep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RC) tau[i] = tau[i] * keep;                             // MAACO.py:305
}
// visit bit matrix [word][cell]: bit (a & 63) of word a >> 6 set iff successful ant a visited cell
__global__ __launch_bounds__(64) void k_visit_bits(int n, int path_cap, const int* cells, const int* len, const double* plen,
                                                   double Q, unsigned long long* bits, int RC, double* dep, uint8_t* flag, int fstride) {
  const int a = blockIdx.x;
  const int L = len[a];
  const bool good = L > 0 && plen[a] != PF_INF && plen[a] > 1e-6;   // MAACO.py:307
  if (threadIdx.x == 0) dep[a] = good ? Q / plen[a] : 0.0;          // :308
  if (!good) return;
  const unsigned long long bit = 1ull << (a & 63);
  uint8_t* fl = flag + (a >> 6);
  for (int i = threadIdx.x; i < L; i += blockDim.x) { const int c = cells[(size_t)a * path_cap + i]; atomicOr(&bits[bits_idx(c, a >> 6, fstride)], bit); fl[(size_t)(c >> 6) * fstride] = 1; }
}
// per cell: add the deposits of the ants that visited it, in ant order (MAACO.py:306-311
// is sequential over ants; a cell is visited at most once per ant because of the tabu set)
// The cells every ant crosses (around the start) receive one deposit per ant: that thread's 16 384-long ordered sum is
// the kernel's critical path, and what it waits for is the deposit value of each ant.  The values of a chunk of
// PF_DEP_CHUNK ants are therefore staged in LDS (128 KB; one 1024-thread block per CU), where a read costs ~64 cycles
// instead of an L2 round trip.
#define PF_DEP_CHUNK 16384
// The ordered deposits of ONE 64-ant word.  Sparse words (the usual case) walk their set bits; when some cell of the wavefront
// has more than PF_DEP_DENSE of its 64 ants set -- the start, the target and their neighbours collect a deposit from (almost)
// every ant, and that thread's ordered sum is the kernel's critical path -- all 64 ants are stepped through instead, without
// count-trailing-zeros or dependent LDS addresses.  Same sums in the same order either way: a clear bit adds exactly nothing.
#ifndef PF_TAU_PROBE
#define PF_TAU_PROBE 0   // timing probes only: 1 = no deposits (loads, zeroing stores and the clip stay); 2 = per-wave clocks left in tau (scripts/probe_tau_waves.py)
#endif
#ifndef PF_TAU_FLY
#define PF_TAU_FLY 4        // chunks of a stretch in flight while the previous PF_TAU_FLY are summed (k_tau_update)
#endif
#ifndef PF_DEP_DENSE
#define PF_DEP_DENSE 8    /* (12 before the two-instruction dense form; 3 / 5 / 8 / 12 / 18 -> maaco512 8.83 / 8.92 / 8.93 / 8.87 / 8.69 M evals/s) */
#endif
PF_DEV double dep_word(double t, unsigned long long x, const double* dw) {
  if (__any((int)__builtin_popcountll(x) > PF_DEP_DENSE)) {
    // per ant three instructions, one of them on the chain: bit -> 0.0 / 1.0 (v_bfe, v_cvt), t = fma(d, bit, t) -- exact: d * 1 = d,
    // d * 0 = 0 and t + 0 = t.  Measured per step on one wavefront (scripts/ubench/exec.hip, s_memtime ticks): this form 13.1, sign
    // test + two v_cndmask + add 26.1, a carry-out lane mask moved into EXEC 35.4 (a scalar register written by the VALU is slow to
    // reach the scalar unit), EXEC <- a mask that is already scalar 9.3 = the bare dependent v_add_f64 (9.4).
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    double d[8], e[8];                                              // values read one batch ahead (wave-uniform addresses: LDS broadcasts)
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = dw[k];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
      if (j0 + 8 < 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = dw[j0 + 8 + k];
      }
      const unsigned y = j0 < 32 ? lo : hi;
#pragma unroll
      for (int k = 0; k < 8; ++k) t = __builtin_fma(d[k], (double)((y >> ((j0 + k) & 31)) & 1u), t);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = e[k];
    }
    return t;
  }
  while (x) {
    // four deposits per trip: the LDS reads go out together, the adds stay in ant order (adding the 0.0 of an
    // absent ant leaves the positive sum unchanged)
    const int j0 = __builtin_ctzll(x); x &= x - 1;
    const bool h1 = x != 0; const int j1 = h1 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const bool h2 = x != 0; const int j2 = h2 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const bool h3 = x != 0; const int j3 = h3 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const double d0 = dw[j0], d1 = dw[j1], d2 = dw[j2], d3 = dw[j3];
    t += d0; t += h1 ? d1 : 0.0; t += h2 ? d2 : 0.0; t += h3 ? d3 : 0.0;
  }
  return t;
}
// The dense path in TWO instructions per ant.  An ant's bit, moved to one of the exponent bits of a double's high word (bits 20..30)
// and masked there, IS a double -- 0.0 or 2^(2^k - 1023) for exponent bit k -- and with the deposit pre-scaled by the inverse power of
// two (dws[j] = d_j 2^(1023 - 2^k), exact: the caller has checked d_j < 4), fma(dws[j], that double, t) adds exactly d_j or exactly
// nothing: one rounding, of t + d_j, as the plain add.  Six shifted views of the word put every ant's bit on an exponent bit
// (ant j on bit 20 + j % 11); per ant a v_and and the fma, where bit -> 0.0 / 1.0 took a v_bfe and a v_cvt_f64_u32.
PF_DEV double dep_word_scaled(double t, unsigned long long x, const double* dw, const double* dws) {
  if (__any((int)__builtin_popcountll(x) > PF_DEP_DENSE)) {
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    unsigned v[6];
    v[0] = lo << 20; v[1] = lo << 9;                                // ants 0..10, 11..21
    v[2] = __builtin_amdgcn_alignbit(hi, lo, 2); v[3] = __builtin_amdgcn_alignbit(hi, lo, 13); v[4] = __builtin_amdgcn_alignbit(hi, lo, 24);   // 22..32, 33..43, 44..54
    v[5] = hi >> 3;                                                 // 55..63
    double d[8], e[8];                                              // values read one batch ahead (wave-uniform addresses: LDS broadcasts)
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = dws[k];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
      if (j0 + 8 < 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = dws[j0 + 8 + k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = j0 + k;
        t = __builtin_fma(d[k], __hiloint2double((int)(v[j / 11] & (1u << (20 + j % 11))), 0), t);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = e[k];
    }
    return t;
  }
  while (x) {
    const int j0 = __builtin_ctzll(x); x &= x - 1;
    const bool h1 = x != 0; const int j1 = h1 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const bool h2 = x != 0; const int j2 = h2 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const bool h3 = x != 0; const int j3 = h3 ? __builtin_ctzll(x) : j0; x &= x - 1;
    const double d0 = dw[j0], d1 = dw[j1], d2 = dw[j2], d3 = dw[j3];
    t += d0; t += h1 ? d1 : 0.0; t += h2 ? d2 : 0.0; t += h3 ? d3 : 0.0;
  }
  return t;
}
#define PF_UPD_CHUNK 8192   /* ants per LDS stage of k_tau_update: their deposits and the pre-scaled ones (2 x 64 KB) */
// Every word of the matrix is read here exactly once, so the kernel also leaves it zeroed for the next iteration (a
// store per non-zero word) instead of the host clearing n/8 bytes per cell -- 512 MB at 16 384 ants on G512 -- every time.
__global__ __launch_bounds__(1024) void k_tau_deposit(double* tau, const uint8_t* occ, int RC, unsigned long long* bits, int nwords,
                                                     const double* dep, int cell0, int cell1, int W) {
  extern __shared__ __attribute__((aligned(16))) double sdep[];    // [PF_DEP_CHUNK]
  const int i = cell0 + blockIdx.x * blockDim.x + threadIdx.x;     // cells [cell0, cell1): the pipelined multi-GPU fold works on row chunks
  const bool live = i < cell1;
  double t = live ? tau[i] : 0.0;
  bool touched = false;
  for (int c0 = 0; c0 < nwords; c0 += PF_DEP_CHUNK / 64) {
    const int cw = nwords - c0 < PF_DEP_CHUNK / 64 ? nwords - c0 : PF_DEP_CHUNK / 64;   // words of this chunk of ants
    __syncthreads();
    for (int k = threadIdx.x; k < cw * 64; k += blockDim.x) sdep[k] = dep[c0 * 64 + k];
    __syncthreads();
    if (!live) continue;
    // the sum must run in ant order, but the (coalesced) word loads need not wait for it: 8 in flight per thread
    for (int w0 = 0; w0 < cw; w0 += 8) {
      unsigned long long b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = (w0 + u < cw) ? bits[bits_idx(i, c0 + w0 + u, W)] : 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (b[u]) bits[bits_idx(i, c0 + w0 + u, W)] = 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long x = b[u];
        touched |= x != 0;
        t = dep_word(t, x, sdep + (w0 + u) * 64);
      }
    }
  }
  if (live && touched && occ[i] != 1) tau[i] = t;
}
__global__ void k_tau_clip(double* tau, const uint8_t* occ, int RC, double tmin, double tmax) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RC) return;
  tau[i] = occ[i] == 1 ? 1e-9 : fmin(fmax(tau[i], tmin), tmax);   // MAACO.py:326-332
}

// MAACO.py:351-358 on the device, so that an iteration needs ONE small copy to the host: does the iteration's best ant take
// over the overall best, and what are the clip bounds of the pheromone update that follows (:312-323, with the overall best
// AFTER this iteration's take).  state = {ib_len, ib_turns, ib_idx, took, best_len, best_turns, tmin, tmax, skip, steps,
// candidates, path_cells, overflow}; skip = 1 when an ant overflowed its path row (the caller redoes the iteration with
// longer rows: the pheromone must not move).
PF_DEV void maaco_take(const double* scan3, double best_len, double best_turns, double rho, int R, int C, const DevCounters* cnt,
                       double* state) {
  const double il = scan3[0], it = scan3[1];
  double took = 0.0;
  if (il < best_len) { best_len = il; best_turns = it; took = 1.0; }                          // :351-354
  else if (fabs(il - best_len) < 1e-9 && it < best_turns) { best_turns = it; took = 1.0; }    // :355-358
  double bl = best_len;                                            // MAACO.py:312-316
  if (bl == PF_INF) bl = (double)(R + C);
  if (bl < 1e-6) bl = 1e-6;
  const double tmax = (1.0 / (1.0 - rho)) * (1.0 / bl);            // :317
  int mx = C > R ? C : R; if (mx < 1) mx = 1;
  state[0] = il; state[1] = it; state[2] = scan3[2]; state[3] = took; state[4] = best_len; state[5] = best_turns;
  state[6] = tmax / (2.0 * mx); state[7] = tmax;                    // :323
  state[8] = cnt->overflow ? 1.0 : 0.0;
  state[9] = (double)cnt->steps; state[10] = (double)cnt->candidates; state[11] = (double)cnt->path_cells; state[12] = (double)cnt->overflow;
}
// The scan and the take-over test of one iteration in ONE launch (a 1024-thread block): thread 0 runs the test once the block's
// scan is in `scan3`, mirrors the 13 doubles into `host_state` (pinned, device-visible host memory: the caller reads them after an
// event, no copy is enqueued) and hands the iteration's private work counter / counters back zeroed for the next walk.
// When the iteration's best ant takes over the overall best (and no ant overflowed), its path row is copied into `best_row`
// ([0] = length, then the cells): the walk buffer is overwritten by the next iteration, the overall best stays in HBM and
// reaches the host only when somebody reads it (pf_maaco_best_path).
__global__ __launch_bounds__(1024) void k_maaco_best_take(int n, const double* plen, const int* turns, double* scan3, double best_len,
                                                         double best_turns, double rho, int R, int C, int* work, DevCounters* cnt,
                                                         double* state, double* host_state, const int* cells, const int* len, int path_cap,
                                                         int* best_row) {
  __shared__ int s_take;
  maaco_best_block(n, plen, turns, scan3);
  if (threadIdx.x == 0) {                                          // (thread 0 wrote scan3 itself: program order)
    maaco_take(scan3, best_len, best_turns, rho, R, C, cnt, state);
    s_take = (state[3] != 0.0 && state[8] == 0.0 && state[2] >= 0.0) ? (int)state[2] : -1;
  }
  __syncthreads();
  const int a = s_take;
  if (a >= 0 && best_row) {
    const int L = len[a];
    for (int i = threadIdx.x; i < L; i += 1024) best_row[1 + i] = cells[(size_t)a * path_cap + i];
    if (threadIdx.x == 0) best_row[0] = L;
  }
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 13; ++k) host_state[k] = state[k];
  *work = 0;
  DevCounters z; memset(&z, 0, sizeof(z)); *cnt = z;
}
// The whole of MAACO.py:304-332 in ONE pass over tau: per cell t = tau * (1 - rho) (:305), then the deposits of the ants that
// visited it in ant order (:306-311; k_tau_deposit's ordered walk of the bit matrix), then the clip (:326-332) -- the same
// fp64 operations in the same order as the three kernels, one read and one write of tau instead of three each.  The clip
// bounds come from `state` (k_maaco_take) or, when state is null, from the arguments.
__global__ __launch_bounds__(1024) void k_tau_update(double* tau, const uint8_t* occ, int RC, unsigned long long* bits, int nwords,
                                                    const double* dep, double keep, const double* state, double tmin_a, double tmax_a,
                                                    uint8_t* flag, int fstride) {
  extern __shared__ __attribute__((aligned(16))) double sdep[];    // [PF_UPD_CHUNK] deposits, then [PF_UPD_CHUNK] pre-scaled (dep_word_scaled)
  double* sdeps = sdep + PF_UPD_CHUNK;
  if (state && state[8] != 0.0) return;                            // an ant overflowed: the iteration is redone, tau stays
  const double tmin = state ? state[6] : tmin_a, tmax = state ? state[7] : tmax_a;
  // a wavefront owns one 64-cell stretch (512-byte word loads); the 16 wavefronts of a block take stretches gridDim.x apart, so the
  // few busy parts of the map -- around the start, the target and the corridors every ant uses -- land on different CUs
  const int lane = threadIdx.x & 63;
  const int seg = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
  const int i = seg * 64 + lane;
  const bool live = i < RC, seg_live = seg * 64 < RC;              // (seg_live is wave-uniform)
  double t = live ? tau[i] * keep : 0.0;                           // :305
#if PF_TAU_PROBE == 2
  const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
  int probe_dense = 0, probe_chunks = 0;
#endif
  uint8_t* frow = flag + (size_t)seg * fstride;
  for (int c0 = 0; c0 < nwords; c0 += PF_UPD_CHUNK / 64) {
    const int cw = nwords - c0 < PF_UPD_CHUNK / 64 ? nwords - c0 : PF_UPD_CHUNK / 64;
    __syncthreads();
    bool big = false;                                              // a deposit of 4 or more (or a NaN) would overflow its scaling by 2^1022
    for (int k = threadIdx.x; k < cw * 64; k += blockDim.x) {
      const double v = dep[c0 * 64 + k];
      sdep[k] = v; sdeps[k] = __builtin_ldexp(v, 1023 - (1 << ((k & 63) % 11))); big |= !(v < 4.0);
    }
    const bool scaled = __syncthreads_or(big) == 0;                 // (block-uniform; Q / L is ~4e-3 with the reference's parameters)
    if (!seg_live) continue;
    for (int k0 = 0; k0 < cw; k0 += 64) {
      // which of the next 64 words have anything in this stretch: one flag byte per lane -> a wave-uniform mask, walked in word
      // (= ant) order; only those chunks are loaded at all (measured: 23 % of them at 512^2 / 16 384 ants, 14 % at 1024^2 / 8 192)
      const int wl = c0 + k0 + lane;
      const bool mine = k0 + lane < cw;
      const uint8_t f = mine ? frow[wl] : (uint8_t)0;
      if (f) frow[wl] = 0;
      unsigned long long m = __ballot(f != 0);
      // PF_TAU_FLY chunks in flight, the next PF_TAU_FLY requested before these are summed.  The loads are unconditional (an empty slot
      // re-reads chunk 0 of the stretch and is masked afterwards): loads under a branch make the compiler wait for ALL of them.
      unsigned long long* cb = bits + (size_t)seg * fstride * 64 + lane;   // bits_idx(i, w, fstride) = cb[w * 64]
      int idx[PF_TAU_FLY], nidx[PF_TAU_FLY];
      unsigned long long b[PF_TAU_FLY], nb[PF_TAU_FLY];
#pragma unroll
      for (int u = 0; u < PF_TAU_FLY; ++u) {
        idx[u] = m ? c0 + k0 + (int)__builtin_ctzll(m) : -1; m &= m - 1;
        b[u] = cb[(size_t)(idx[u] < 0 ? 0 : idx[u]) * 64];
      }
      while (idx[0] >= 0) {
#pragma unroll
        for (int u = 0; u < PF_TAU_FLY; ++u) {
          nidx[u] = m ? c0 + k0 + (int)__builtin_ctzll(m) : -1; m &= m - 1;
          nb[u] = cb[(size_t)(nidx[u] < 0 ? 0 : nidx[u]) * 64];
        }
#pragma unroll
        for (int u = 0; u < PF_TAU_FLY; ++u) {
          if (idx[u] < 0) break;                                    // (wave-uniform)
          const unsigned long long x = live ? b[u] : 0ull;
          // the matrix goes back zeroed; every lane stores (a store under a branch would again cost exact wait counts, and the
          // flagged chunks are 1/4 of the matrix)
          cb[(size_t)idx[u] * 64] = 0ull;
#if PF_TAU_PROBE == 2
          probe_chunks += 1; probe_dense += __any((int)__builtin_popcountll(x) > PF_DEP_DENSE) ? 1 : 0;
#endif
#if PF_TAU_PROBE != 1
          t = scaled ? dep_word_scaled(t, x, sdep + (idx[u] - c0) * 64, sdeps + (idx[u] - c0) * 64) : dep_word(t, x, sdep + (idx[u] - c0) * 64);
#else
          t += x == 12345ull ? 1.0 : 0.0;
#endif
        }
#pragma unroll
        for (int u = 0; u < PF_TAU_FLY; ++u) { idx[u] = nidx[u]; b[u] = nb[u]; }
      }
    }
  }
#if PF_TAU_PROBE == 2
  // (timing probe, wrong pheromone on purpose: lane 0 leaves the wave's shader clocks, lane 1 its dirty chunks, lane 2 the dense ones)
  if (live) tau[i] = lane == 0 ? (double)(__builtin_amdgcn_s_memtime() - probe_t0) : lane == 1 ? (double)probe_chunks : lane == 2 ? (double)probe_dense : t;   // (t stays live: the sums must not be optimised away)
  return;
#endif
  if (live) tau[i] = occ[i] == 1 ? 1e-9 : fmin(fmax(t, tmin), tmax);   // :326-332 (paths never cross obstacles: their words are empty)
}

// ===========================================================================
// K7 + K2b + K1: MPA
// ===========================================================================
#ifdef PF_TRACE
__device__ unsigned long long g_trace[4 * 16384];   // diagnostic build only: per sweep item {t0, t1, pops, wave}
__device__ unsigned long long g_trace3[4 * 16384];  // phase items {1 + idx, inter, cur, -}; fads items {1, node, -, -}
__device__ unsigned long long g_trace2[4 * 16384];  // per sweep item {pops A*#1, result A*#1, pops A*#2, result A*#2}
#endif
struct MpaDev {
  double P, levy_beta, sigma, fads;
  int N, start, target;
  // exact shortest-path length of every cell from the start / to the target on the static grid (no avoid set),
  // or null: admissible lower bounds on the length of any path through a cell (see mpa_phase_item / mpa_fads_item)
  const double* ds; const double* dt;
};
PF_DEV long py_round(double x) { return (long)__builtin_rint(x); }
PF_DEV int clampi(long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }

// ---- target-cell proposals (MPA.py:250-282) -------------------------------------------------------------------
// The proposals are the one place on the path where libm transcendentals (log in normalvariate's accept test, pow /
// sin / cos in the Levy step) feed a DISCRETE decision (an accept, a round()).  The device's ocml functions are not
// glibc's, so a result is trusted only when it is far from every decision boundary: the functions below also report
// `doubt` when an accept test or a rounding lies within a margin that dwarfs any 1-2 ulp disagreement (2^-33
// relative for the accept test, 1e-7 absolute on the fraction for round()).  Doubtful proposals (expected never: ~1e-9
// per call) are recomputed by the host with glibc itself (mpa_resolve_doubts) before the searches start.  Brownian
// arithmetic other than the accept test is IEEE +,*,/,sqrt: bit-identical on both sides.
struct Doubt { double eps_log, eps_round; bool hit; };
PF_DEV double normalvariate_chk(Rng& g, double mu, double sigma, Doubt& d) {   // random.py normalvariate
  double z;
  for (;;) {
    const double u1 = g.random();
    const double u2 = 1.0 - g.random();
    z = 1.7155277699214135 * (u1 - 0.5) / u2;
    const double zz = z * z / 4.0, l = -log(u2);
    if (fabs(zz - l) <= d.eps_log * (1.0 + fabs(l))) d.hit = true;
    if (zz <= l) break;
  }
  return mu + z * sigma;
}
PF_DEV bool near_half(double x, double eps) { return fabs(fabs(x - __builtin_rint(x)) - 0.5) <= eps; }
// MPA._get_levy_target_node, MPA.py:250-264
PF_DEV int levy_target(Rng& g, const Grid& G, int cur, double scale, double beta, double sigma, Doubt& d) {
  const double u = normalvariate_chk(g, 0.0, sigma, d);
  double v = normalvariate_chk(g, 0.0, 1.0, d);
  if (fabs(v) < 1e-9) v = 1e-9;
  double step = 0.05 * u / pow(fabs(v), 1.0 / beta) * scale;
  const double mx = (double)(G.R > G.C ? G.R : G.C) * 0.5;
  step = fmin(fmax(step, -mx), mx);
  const double ang = g.uniform(0.0, 2.0 * 3.141592653589793);
  const double xr = step * sin(ang), xc = step * cos(ang);
  if (near_half(xr, d.eps_round) || near_half(xc, d.eps_round)) d.hit = true;
  const long dr = py_round(xr), dc = py_round(xc);
  const int r = row_of(G, cur), c = cur - r * G.C;
  return clampi(r + dr, 0, G.R - 1) * G.C + clampi(c + dc, 0, G.C - 1);
}
// MPA._get_brownian_target_node, MPA.py:266-282 (elite < 0 == None)
PF_DEV int brownian_target(Rng& g, const Grid& G, int cur, int elite, double scale, Doubt& d) {
  const int cr = row_of(G, cur), cc = cur - cr * G.C;
  long tr_, tc_;
  if (g.random() < 0.7 && elite >= 0) {
    const int er = row_of(G, elite), ec = elite - er * G.C;
    const int dr = er - cr, dc = ec - cc;
    const double dist = __builtin_sqrt((double)((long)dr * dr + (long)dc * dc));
    if (dist > 1e-6) {
      const double fac = fabs(normalvariate_chk(g, 0.0, 1.0, d));
      long kk = py_round(scale * fac * 5.0); if (kk < 1) kk = 1;
      const double ms = dist < (double)kk ? dist : (double)kk;
      tr_ = cr + py_round((double)dr / dist * ms);
      tc_ = cc + py_round((double)dc / dist * ms);
    } else return elite;
  } else {
    long m = py_round((double)(G.R > G.C ? G.R : G.C) * 0.1 * scale * fabs(normalvariate_chk(g, 0.0, 1.0, d)));
    if (m < 1) m = 1;
    const long dr = g.randint(-m, m);
    const long dc = g.randint(-m, m);
    tr_ = cr + dr; tc_ = cc + dc;
  }
  return clampi(tr_, 0, G.R - 1) * G.C + clampi(tc_, 0, G.C - 1);
}

PF_DEV void copy_path(int* dst, const int* src, int n, int lane) {
  for (int i = lane; i < n; i += 64) dst[i] = src[i];
}

struct MpaPhaseArgs {
  Common c; ScoreP sp; MpaDev m;
  int phase, iter; double CF; unsigned long long seed;
  int n, path_cap;
  const int* pop_cells; const int* pop_len; const double* pop_stats;
  const int* gidx;   // [n] index of the predator in the GLOBAL fitness-sorted population (stream key, Levy split)
  const int* slot;   // [n] storage slot of the predator in pop_*
  const int* elite_cells; int elite_len; const double* elite_stats;   // device double[5]
  const int* elite_len_dev;   // non-null: the elite's length lives in HBM (device-resident iteration: the host never learns it)
  int* out_cells; int* out_len; double* out_stats; int* status;
  // explicit mode (pf_mpa_rebuild_batch): no idx/gate draws
  const int* ex_idx; const int* ex_levy; const double* ex_scale; const int* ex_agent;
  // proposals (k_mpa_propose, before the searches): prop[a] = {idx or -1 when the predator does not move, target cell}
  int2* prop; int* doubt_list; int* doubt_n; double eps_log, eps_round;
};

// which path a predator modifies and which one it samples the elite node from (MPA.py:339-377), explicit mode included
struct MpaPlan { bool is_levy; double scale, gate_p; const int* mod; int modL; const double* mod_stats; const int* ref; int refL;
                 const int* prey; int preyL; const double* prey_stats; int gi, slot; };
PF_DEV MpaPlan mpa_plan(const MpaPhaseArgs& p, int a) {
  const int eL = p.elite_len_dev ? *p.elite_len_dev : p.elite_len;
  MpaPlan q;
  q.gi = p.ex_idx ? p.ex_agent[a] : p.gidx[a];                   // index in the fitness-sorted population
  q.slot = p.ex_idx ? a : p.slot[a];
  q.prey = p.pop_cells + (size_t)q.slot * p.path_cap; q.preyL = p.pop_len[q.slot]; q.prey_stats = p.pop_stats + (size_t)q.slot * 5;
  if (p.ex_idx) { q.is_levy = p.ex_levy[a] != 0; q.scale = p.ex_scale[a]; q.mod = q.prey; q.modL = q.preyL; q.mod_stats = q.prey_stats; q.ref = p.elite_cells; q.refL = eL; }
  else if (p.phase == 1) { q.is_levy = false; q.scale = p.m.P; q.mod = q.prey; q.modL = q.preyL; q.mod_stats = q.prey_stats; q.ref = p.elite_cells; q.refL = eL; }
  else if (p.phase == 2) {
    q.is_levy = q.gi < p.m.N / 2;                                 // :351
    q.scale = q.is_levy ? p.m.P : p.m.P * p.CF;                   // :354
    q.mod = q.is_levy ? q.prey : p.elite_cells; q.modL = q.is_levy ? q.preyL : eL;
    q.mod_stats = q.is_levy ? q.prey_stats : p.elite_stats;
    q.ref = q.is_levy ? p.elite_cells : q.prey; q.refL = q.is_levy ? eL : q.preyL;
  } else { q.is_levy = true; q.scale = p.m.P * p.CF; q.mod = p.elite_cells; q.modL = eL; q.mod_stats = p.elite_stats; q.ref = q.prey; q.refL = q.preyL; }
  q.gate_p = p.phase == 1 ? p.m.P : q.scale;                      // :344 / :359 / :372
  return q;
}
// Proposal pass, one thread per predator: the gating draws (:343-344 etc.), the target cell, the work estimate of the
// longest-first queue, and the list of predators whose proposal the host has to confirm (see Doubt).
__global__ void k_mpa_propose(MpaPhaseArgs p, float* est) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= p.n) return;
  const MpaPlan q = mpa_plan(p, a);
  int idx = -1, inter = -1;
  Doubt d; d.eps_log = p.eps_log; d.eps_round = p.eps_round; d.hit = false;
  Rng g; g.init(p.seed, DOM_MPA, (unsigned long long)p.iter, (unsigned long long)q.gi);
  if (p.ex_idx ? (q.modL > 0 && p.ex_idx[a] < q.modL - 1) : (q.modL > 1)) {    // explicit: the :286 early return
    const int i0 = p.ex_idx ? p.ex_idx[a] : (int)g.randint(0, q.modL - 2);      // :343 / :358 / :371
    if (p.ex_idx || g.random() < q.gate_p) {
      idx = i0;
      const int cur = q.mod[idx];
      if (q.is_levy) inter = levy_target(g, p.c.G, cur, q.scale, p.m.levy_beta, p.m.sigma, d);
      else {
        int en = -1;
        if (q.refL > 0) en = q.ref[(int)g.randbelow((unsigned long long)q.refL)];   // random.choice :248
        inter = brownian_target(g, p.c.G, cur, en, q.scale, d);
      }
    }
  }
  p.prop[a] = make_int2(idx, inter);
  if (est) est[a] = idx >= 0 ? (float)(q.modL - idx) : 0.f;
  if (d.hit && idx >= 0) { const int at = atomicAdd(p.doubt_n, 1); p.doubt_list[at] = a; }
}


// One predator of one phase sweep, MPA.py:339-377 + _reconstruct_path_segment :284-318.
__device__ __forceinline__ void mpa_phase_item(const MpaPhaseArgs& p, int a, Slot& s, const Open& O, AStat& tot,
                               unsigned long long& cells, unsigned long long& ovf, int lane) {
  const Grid& G = p.c.G;
  const int RC = G.R * G.C;
  {
    const MpaPlan q = mpa_plan(p, a);
    const int* prey = q.prey; const double* prey_stats = q.prey_stats;
    const int* mod = q.mod; const int modL = q.modL; const double* mod_stats = q.mod_stats;
    int* out = p.out_cells + (size_t)a * p.path_cap;
    int rc = 4, n = modL;                                         // default: the unmodified path + its stats
    bool rebuilt = false;
    {
      const int2 pr = p.prop[a];                                  // k_mpa_propose: {idx or -1, target cell}
      const int idx = first_i(pr.x);
      if (idx >= 0) {
        // ---- _reconstruct_path_segment(mod, ref, idx, is_levy, scale) ----
        // (idx <= modL-2 so the :286 early return cannot trigger)
        slot_begin_eval(s, RC, lane);
        const int cur = mod[idx];
        mark_avoid(s, mod, idx, lane);                            // set(prefix[:-1]) :290
        const int inter = first_i(pr.y);
#ifdef PF_TRACE
        if (lane == 0 && a < 8192) { g_trace3[4 * a] = 1 + idx; g_trace3[4 * a + 1] = inter; g_trace3[4 * a + 2] = cur; }
#endif
        // Exact pruning.  The memory step keeps this candidate only if its fitness is BELOW the prey's (:382), and
        // fitness >= length (the penalty terms are non-negative).  Whatever the two searches find, the rebuilt path
        // runs prefix -> cur -> (inter ->) target, so its length is at least len(prefix) + min(|cur,inter| +
        // dt[inter], dt[cur]) with dt the static shortest distance to the target (avoid sets only lengthen paths).
        // If that bound already reaches the prey's fitness the candidate cannot be accepted; if the rebuild failed
        // instead (:316-317) the candidate would be the unmodified path, which is no better either when it is the
        // prey itself or not fitter than the prey.  Same population, none of the work.
        bool pruned = false;
        if (p.m.dt && !p.ex_idx) {
          double pre = 0.0;
          for (int i = lane; i < idx; i += 64) {
            const int dd = mod[i + 1] - mod[i];
            pre += (dd == 1 || dd == -1 || dd == G.C || dd == -G.C) ? 1.0 : PF_SQRT2;
          }
          pre = wave_sum_d(pre);
          double lb = p.m.dt[cur];
          if (G.occ[inter] != 1 && inter != cur) {
            const int r0 = row_of(G, cur), r1 = row_of(G, inter);
            const long dr_ = r1 - r0, dc_ = (inter - r1 * G.C) - (cur - r0 * G.C);
            lb = fmin(lb, __builtin_sqrt((double)(dr_ * dr_ + dc_ * dc_)) + p.m.dt[inter]);
          }
          lb = (pre + lb) * (1.0 - 1e-9);
          const double prey_fit = prey_stats[4];
          pruned = lb >= prey_fit && (mod == prey || mod_stats[4] >= prey_fit);
        }
        if (pruned) ovf += 1ull << 32;
        else if (idx + 1 > p.path_cap) rc = 3;
        else {
          copy_path(out, mod, idx + 1, lane);                     // :296
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          n = idx + 1;
          int astart = cur; rc = 0;
          for (int seg = 0; seg < 2 && rc != 3; ++seg) {
            int goal;
            if (seg == 0) { if (!(G.occ[inter] != 1 && inter != astart)) continue; goal = inter; }   // :298
            else { if (astart == p.m.target) continue; goal = p.m.target; }                           // :306
            int mlen = 0;
#ifdef PF_TRACE
            const unsigned long long pq0 = tot.pops;
#endif
            const int r2 = astar<1>(G, s, O, astart, goal, out + n - 1, p.path_cap - (n - 1), mlen, tot, lane);
#ifdef PF_TRACE
            if (lane == 0 && a < 8192) { g_trace2[4 * a + 2 * seg] = tot.pops - pq0; g_trace2[4 * a + 2 * seg + 1] = 100 + r2; }
#endif
            if (r2 == 3) { rc = 3; break; }
            if (r2 == 0 && mlen > 1) {                            // :300-305 / :308-309
              if (seg == 0) { mark_avoid(s, out + n, mlen - 1, lane); astart = inter; }
              n += mlen - 1;
            }
          }
          if (rc != 3) {
            // :310-315 dedup is a no-op (see k_decode_batch); :316-317 endpoint check
            const int first = out[0], last = out[n - 1];
            if (first != p.m.start || last != p.m.target) rc = 4; else { rc = 0; rebuilt = true; }
          }
        }
      }
    }
    double sc[5];
    if (rebuilt) score_path(G, p.sp, out, n, lane, sc);
    else if (rc != 3) {
      // no move: phase 1 keeps the prey (:342,:347); phases 2/3 re-score path_to_modify (:356,:363,:369,:376),
      // whose stats are the stored ones (same function, same path)
      n = modL;
      copy_path(out, mod, modL, lane);
      for (int i = 0; i < 5; ++i) sc[i] = mod_stats[i];
      if (modL == 0) { sc[0] = PF_INF; sc[1] = 0; sc[2] = 0; sc[3] = 0; sc[4] = PF_INF; }
    } else n = 0;
    if (lane == 0) { p.out_len[a] = n; p.status[a] = rc; }
    if (rc != 3 && lane < 5) p.out_stats[(size_t)a * 5 + lane] = sc[lane];
    cells += n; ovf += rc == 3;
  }
}
__global__ __launch_bounds__(64) void k_mpa_phase(MpaPhaseArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  Open O = make_open(smem, p.c.S, p.c.tier2);
  Slot s = slot_load(p.c, p.c.G.R * p.c.G.C);
  AStat tot = {0, 0, 0, 0, 0, 0};
  unsigned long long cells = 0, ovf = 0;
  for (;;) {
    const int a = next_agent(p.c, p.n, lane);
    if (a < 0) break;
    mpa_phase_item(p, a, s, O, tot, cells, ovf, lane);
  }
  slot_store(p.c, s, lane);
  flush_counters(p.c.cnt, tot, cells, ovf, lane);
}

struct MpaFadsArgs {
  Common c; ScoreP sp; MpaDev m;
  int iter; double CF; unsigned long long seed;
  int n, path_cap;
  int* pop_cells; int* pop_len; double* pop_stats; const int* gidx; const int* slot;
  int* tmp_cells;   // [nslots][path_cap]
  int* status;
  // MPA._generate_initial_path() (MPA.py:154) is a pure function of the grid: A*(start,target) with no
  // avoid set.  It is computed once (pf_mpa_setup) and reused by the FADs re-init branch (:405).
  const int* init_cells; int init_len; const double* init_stats;
  // candidate mode (fused sweep): the FADs candidate of predator a depends only on its stream and the grid,
  // never on the population, so it is produced alongside the phase sweep into cand_* (cand_len 0 = none);
  // k_mpa_apply does the :402/:408 comparison after the memory step.
  int* cand_cells; int* cand_len; double* cand_stats;
};
// One predator of the FADs sweep MPA.py:387-410 (in place on the post-memory population, or candidate mode).
__device__ __forceinline__ void mpa_fads_item(const MpaFadsArgs& p, int a, Slot& s, const Open& O, AStat& tot,
                              unsigned long long& cells, unsigned long long& ovf, int lane) {
  const Grid& G = p.c.G;
  const int RC = G.R * G.C;
  int* tmp = p.cand_cells ? p.cand_cells + (size_t)a * p.path_cap : p.tmp_cells + (size_t)blockIdx.x * p.path_cap;
  {
    const int gi = p.gidx[a];
    const int slot = p.slot[a];
    int rc = 4, n = 0;
    bool have = false, have_stats = false;
    Rng g; g.init(p.seed, DOM_MPA_FADS, (unsigned long long)p.iter, (unsigned long long)gi);
    if (g.random() < p.m.fads) {                                   // :389
      slot_begin_eval(s, RC, lane);
      if (g.random() < p.CF) {                                     // :390
        const int rr_ = (int)g.randint(0, G.R - 1);                // :391
        const int rc_ = (int)g.randint(0, G.C - 1);
        const int node = rr_ * G.C + rc_;
        // Exact pruning (see mpa_phase_item): any path start -> node -> target is at least ds[node] + dt[node] long,
        // and the candidate is kept only if its fitness is below the predator's (:402), which is at most the fitness
        // read here (the memory step can only lower it).
        bool pruned = false;
        if (p.m.dt && G.occ[node] != 1) {
          pruned = (p.m.ds[node] + p.m.dt[node]) * (1.0 - 1e-9) >= p.pop_stats[(size_t)slot * 5 + 4];
          if (pruned) ovf += 1ull << 32;
        }
        if (G.occ[node] != 1 && !pruned) {                         // :393
          int m1 = 0;
#ifdef PF_TRACE
          const unsigned long long pq0 = tot.pops;
          if (lane == 0 && a < 8192) { g_trace3[4 * (p.n + a)] = 1; g_trace3[4 * (p.n + a) + 1] = node; }
#endif
          int r1 = astar<1>(G, s, O, p.m.start, node, tmp, p.path_cap, m1, tot, lane);   // :394
#ifdef PF_TRACE
          if (lane == 0 && a < 8192) { g_trace2[4 * (p.n + a)] = tot.pops - pq0; g_trace2[4 * (p.n + a) + 1] = 100 + r1; }
          const unsigned long long pq1 = tot.pops;
#endif
          if (r1 == 3) rc = 3;
          else if (r1 == 0 && m1 > 0) {
            mark_avoid(s, tmp, m1 - 1, lane);                      // set(p1[:-1]) :396
            int m2 = 0;
            int r2 = astar<1>(G, s, O, node, p.m.target, tmp + m1 - 1, p.path_cap - (m1 - 1), m2, tot, lane);
#ifdef PF_TRACE
            if (lane == 0 && a < 8192) { g_trace2[4 * (p.n + a) + 2] = tot.pops - pq1; g_trace2[4 * (p.n + a) + 3] = 100 + r2; }
#endif
            if (r2 == 3) rc = 3;
            else if (r2 == 0 && m2 > 0) { n = m1 + m2 - 1; have = true; }   // :398-400 (last is the target by construction)
          }
        }
      } else if (p.init_len > 0) {                                 // :405 re-init path (memoised, see MpaFadsArgs)
        if (p.init_len > p.path_cap) rc = 3;
        else { copy_path(tmp, p.init_cells, p.init_len, lane); n = p.init_len; have = true; have_stats = true; }
      }
    }
    if (have) {
      double sc[5];
      if (have_stats) { for (int i = 0; i < 5; ++i) sc[i] = p.init_stats[i]; }
      else score_path(G, p.sp, tmp, n, lane, sc);
      if (p.cand_cells) {
        if (lane < 5) p.cand_stats[(size_t)a * 5 + lane] = sc[lane];
        rc = 0;
      } else {
        const double curfit = p.pop_stats[(size_t)slot * 5 + 4];
        if (sc[4] < curfit) {                                      // :402 / :408
          copy_path(p.pop_cells + (size_t)slot * p.path_cap, tmp, n, lane);
          if (lane == 0) p.pop_len[slot] = n;
          if (lane < 5) p.pop_stats[(size_t)slot * 5 + lane] = sc[lane];
          rc = 0;
        }
      }
    }
    if (p.cand_cells && lane == 0) p.cand_len[a] = have ? n : 0;
    if (!p.cand_cells && lane == 0) p.status[a] = rc;
    cells += n; ovf += rc == 3;
  }
}
__global__ __launch_bounds__(64) void k_mpa_fads(MpaFadsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  Open O = make_open(smem, p.c.S, p.c.tier2);
  Slot s = slot_load(p.c, p.c.G.R * p.c.G.C);
  AStat tot = {0, 0, 0, 0, 0, 0};
  unsigned long long cells = 0, ovf = 0;
  for (;;) {
    const int a = next_agent(p.c, p.n, lane);
    if (a < 0) break;
    mpa_fads_item(p, a, s, O, tot, cells, ovf, lane);
  }
  slot_store(p.c, s, lane);
  flush_counters(p.c.cnt, tot, cells, ovf, lane);
}

// The sweep of one iteration: phase items [0,n) and FADs candidate items [n,2n) share one longest-first work queue, so an
// iteration pays ONE tail instead of two (the FADs candidates are population independent).  Three passes:
//   k_mpa_plan    one wave per item: which searches, if any (MpaPlan, the gate's verdict from k_mpa_propose, the exact
//                 pruning bound, the FADs gating draws) -> a 32-byte job;
//   k_mpa_search  persistent waves over the queue: nothing but the (up to) two chained MPA._a_star calls of a job, through
//                 ONE astar<1> call site -- the pop loop is the kernel's hot code, and everything else that used to live in
//                 the fused kernel (plans, gates, scoring, the FADs branch) cost it registers: 160 VGPR / 241 SGPR spills
//                 against 118 / 97 for the bare connector;
//   k_mpa_finish  one wave per item: endpoint check, score_path, the fall-back to the unmodified path, FADs candidates.
struct MpaSweepArgs { MpaPhaseArgs ph; MpaFadsArgs fd; };
struct __attribute__((aligned(16))) MpaJob {
  const int* src;     // phase items: the path whose prefix [0, n0) starts the rebuilt path (and whose [0, n0 - 1) is the avoid set)
  int n0;             // cells of src to copy (phase: idx + 1; FADs: 0 -- the first search writes the start itself)
  int astart, g0, g1; // searches astart -> g0 (skipped when g0 < 0) then -> g1
  int kind;           // 0: nothing to search, 1: phase item, 2: FADs detour
  int aux;            // kind 0: phase -> status for k_mpa_finish (4 unmodified, 3 overflow); FADs -> 1 copy the memoised initial path, 3 overflow, 0 none
};
struct MpaRes { int n, rc; };   // cells in the item's buffer after the searches; rc 0 done, 1 FADs detour failed, 3 scratch / path overflow

__global__ __launch_bounds__(64) void k_mpa_plan(MpaSweepArgs q, MpaJob* jobs, MpaRes* res) {
  const MpaPhaseArgs& p = q.ph;
  const MpaFadsArgs& f = q.fd;
  const Grid& G = p.c.G;
  const int lane = lane_id();
  const int item = blockIdx.x;
  if (item >= 2 * p.n) return;
  const bool isph = item < p.n;
  const int a = isph ? item : item - p.n;
  MpaJob j; j.src = nullptr; j.n0 = 0; j.astart = 0; j.g0 = -1; j.g1 = p.m.target; j.kind = 0; j.aux = isph ? 4 : 0;
  unsigned long long pruned_n = 0;
  if (isph) {
    const MpaPlan pl = mpa_plan(p, a);
    const int2 pr = p.prop[a];                                    // k_mpa_propose: {idx or -1, target cell}
    const int idx = first_i(pr.x);
    if (idx >= 0) {
      // ---- _reconstruct_path_segment(mod, ref, idx, is_levy, scale): (idx <= modL-2 so the :286 early return cannot trigger)
      const int* mod = pl.mod;
      const int cur = mod[idx];
      const int inter = first_i(pr.y);
      bool pruned = false;                                        // exact pruning: see mpa_phase_item
      if (p.m.dt) {
        double pre = 0.0;
        for (int i = lane; i < idx; i += 64) {
          const int dd = mod[i + 1] - mod[i];
          pre += (dd == 1 || dd == -1 || dd == G.C || dd == -G.C) ? 1.0 : PF_SQRT2;
        }
        pre = wave_sum_d(pre);
        double lb = p.m.dt[cur];
        if (G.occ[inter] != 1 && inter != cur) {
          const int r0 = row_of(G, cur), r1 = row_of(G, inter);
          const long dr_ = r1 - r0, dc_ = (inter - r1 * G.C) - (cur - r0 * G.C);
          lb = fmin(lb, __builtin_sqrt((double)(dr_ * dr_ + dc_ * dc_)) + p.m.dt[inter]);
        }
        lb = (pre + lb) * (1.0 - 1e-9);
        const double prey_fit = pl.prey_stats[4];
        pruned = lb >= prey_fit && (mod == pl.prey || pl.mod_stats[4] >= prey_fit);
      }
      if (pruned) pruned_n = 1;
      else if (idx + 1 > p.path_cap) j.aux = 3;
      else {
        j.kind = 1; j.src = mod; j.n0 = idx + 1; j.astart = cur;
        j.g0 = (G.occ[inter] != 1 && inter != cur) ? inter : -1;  // :298
      }
    }
  } else {
    Rng g; g.init(f.seed, DOM_MPA_FADS, (unsigned long long)f.iter, (unsigned long long)f.gidx[a]);
    if (g.random() < f.m.fads) {                                   // :389
      if (g.random() < f.CF) {                                     // :390
        const int rr_ = (int)g.randint(0, G.R - 1);                // :391
        const int rc_ = (int)g.randint(0, G.C - 1);
        const int node = rr_ * G.C + rc_;
        bool pruned = false;                                       // exact pruning: see mpa_fads_item
        if (f.m.dt && G.occ[node] != 1) {
          pruned = (f.m.ds[node] + f.m.dt[node]) * (1.0 - 1e-9) >= f.pop_stats[(size_t)f.slot[a] * 5 + 4];
          if (pruned) pruned_n = 1;
        }
        if (G.occ[node] != 1 && !pruned) { j.kind = 2; j.n0 = 0; j.astart = f.m.start; j.g0 = node; }   // :393-394
      } else if (f.init_len > 0) j.aux = f.init_len > f.path_cap ? 3 : 1;   // :405 re-init path (memoised)
    }
  }
  if (lane == 0) {
    jobs[item] = j;
    MpaRes r; r.n = 0; r.rc = 1; res[item] = r;
    if (pruned_n) atomicAdd(&p.c.cnt->pruned, pruned_n);
  }
}

struct MpaSearchArgs { Common c; const MpaJob* jobs; MpaRes* res; int n_items, path_cap; int* ph_cells; int* fd_cells; int n; };

// PR = two wavefronts per search (pf_astar_pr.h): wave 0 pops, wave 1 owns the bucket pool.  128-thread workgroups, one per slot.
template <bool PR>
__global__ __launch_bounds__(PR ? 128 : 64) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_mpa_search(MpaSearchArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  const Grid& G = p.c.G;
  const int RC = G.R * G.C;
#ifdef PF_TWO_WAVE
  if (PR) {
    const int wave = (int)(threadIdx.x >> 6);
    if (wave == 0) pr_init_ctl(smem, lane);
    __syncthreads();
    if (wave == 1) {
      pool_wave<1>(smem, p.c.tier2 + (size_t)blockIdx.x * PF_POOL_STRIDE, p.c.rec + (size_t)blockIdx.x * RC, G.C, lane);
      return;
    }
  }
  PrLink link = {};
  if (PR) link = pr_link(smem);
#else
  static_assert(!PR, "two-wave searches need -DPF_TWO_WAVE");
  PrLink link = {};
#endif
  PrLink* const L = &link;
  Open O = make_open(smem, p.c.S, p.c.tier2);
  Slot s = slot_load(p.c, RC);
  AStat tot = {0, 0, 0, 0, 0, 0};
  unsigned long long ovf = 0;
  for (;;) {
    const int item = next_agent(p.c, p.n_items, lane);
    if (item < 0) break;
    const MpaJob j = p.jobs[item];
    if (first_i(j.kind) == 0) continue;
#ifdef PF_TRACE
    const unsigned long long tr0 = wall_clock64(), pp0 = tot.pops;
#endif
    const bool isph = first_i(j.kind) == 1;
    int* buf = isph ? p.ph_cells + (size_t)item * p.path_cap : p.fd_cells + (size_t)(item - p.n) * p.path_cap;
    slot_begin_eval(s, RC, lane);
    int n = 1, astart = first_i(j.astart), rc = 0;
    const int g0 = first_i(j.g0), g1 = first_i(j.g1);
    if (isph) {
      n = first_i(j.n0);
      mark_avoid(s, j.src, n - 1, lane);                          // set(prefix[:-1]) :290
      copy_path(buf, j.src, n, lane);                             // :296
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    const int cap = p.path_cap;
#pragma unroll 1                                                  // one inlined copy of the pop loop, not two
    for (int seg = 0; seg < 2; ++seg) {
      if (isph && (seg == 0 ? g0 < 0 : astart == g1)) continue;   // :298 / :306
      int mlen = 0;
#ifdef PF_TRACE
      const unsigned long long pq0 = tot.pops;
      if (!isph && seg == 0 && lane == 0 && item < 16384) { g_trace3[4 * item] = 1; g_trace3[4 * item + 1] = g0; }
#endif
      const int r2 = astar<1, false, PR>(G, s, O, astart, seg == 0 ? g0 : g1, buf + n - 1, cap - (n - 1), mlen, tot, lane, nullptr, 0, L);
#ifdef PF_TRACE
      if (lane == 0 && item < 16384) { g_trace2[4 * item + 2 * seg] = tot.pops - pq0; g_trace2[4 * item + 2 * seg + 1] = 100 + r2; }
#endif
      if (r2 == 3) { rc = 3; break; }
      if (isph) {
        if (r2 == 0 && mlen > 1) {                                // :300-305 / :308-309
          if (seg == 0) { mark_avoid(s, buf + n, mlen - 1, lane); astart = g0; }
          n += mlen - 1;
        }
      } else {
        if (!(r2 == 0 && mlen > 0)) { rc = 1; break; }            // :395 / :397
        if (seg == 0) { mark_avoid(s, buf, mlen - 1, lane); astart = g0; n = mlen; }   // set(p1[:-1]) :396
        else n += mlen - 1;                                       // :398-400 (last is the target by construction)
      }
    }
    if (lane == 0) { MpaRes r; r.n = n; r.rc = rc; p.res[item] = r; }
    ovf += rc == 3;
#ifdef PF_TRACE
    if (lane == 0 && item < 16384) {
      g_trace[4 * item] = tr0; g_trace[4 * item + 1] = wall_clock64(); g_trace[4 * item + 2] = tot.pops - pp0; g_trace[4 * item + 3] = blockIdx.x;
    }
#endif
  }
#ifdef PF_TWO_WAVE
  if (PR) pr_exit(smem, lane);
#endif
  slot_store(p.c, s, lane);
  flush_counters(p.c.cnt, tot, 0, ovf, lane);
}

__global__ __launch_bounds__(64) void k_mpa_finish(MpaSweepArgs q, const MpaJob* jobs, const MpaRes* res) {
  const MpaPhaseArgs& p = q.ph;
  const MpaFadsArgs& f = q.fd;
  const Grid& G = p.c.G;
  const int lane = lane_id();
  const int item = blockIdx.x;
  if (item >= 2 * p.n) return;
  const bool isph = item < p.n;
  const int a = isph ? item : item - p.n;
  const MpaJob j = jobs[item];
  const MpaRes r = res[item];
  int n = 0;
  if (isph) {
    const MpaPlan pl = mpa_plan(p, a);
    int* out = p.out_cells + (size_t)a * p.path_cap;
    int rc = j.kind == 1 ? r.rc : j.aux;
    bool rebuilt = false;
    if (j.kind == 1 && rc != 3) {
      // :310-315 dedup is a no-op (see k_decode_batch); :316-317 endpoint check
      n = r.n;
      const int first = out[0], last = out[n - 1];
      if (first != p.m.start || last != p.m.target) rc = 4; else { rc = 0; rebuilt = true; }
    }
    double sc[5];
    if (rebuilt) score_path(G, p.sp, out, n, lane, sc);
    else if (rc != 3) {
      // no move: phase 1 keeps the prey (:342,:347); phases 2/3 re-score path_to_modify (:356,:363,:369,:376),
      // whose stats are the stored ones (same function, same path)
      n = pl.modL;
      copy_path(out, pl.mod, pl.modL, lane);
      for (int i = 0; i < 5; ++i) sc[i] = pl.mod_stats[i];
      if (pl.modL == 0) { sc[0] = PF_INF; sc[1] = 0; sc[2] = 0; sc[3] = 0; sc[4] = PF_INF; }
    } else n = 0;
    if (lane == 0) { p.out_len[a] = n; p.status[a] = rc; }
    if (rc != 3 && lane < 5) p.out_stats[(size_t)a * 5 + lane] = sc[lane];
    if (rc == 3 && j.kind != 1 && lane == 0) atomicAdd(&p.c.cnt->overflow, 1ull);
  } else {
    int* buf = f.cand_cells + (size_t)a * f.path_cap;
    bool have = false;
    double sc[5];
    if (j.kind == 2 && r.rc == 0) { n = r.n; have = true; score_path(G, f.sp, buf, n, lane, sc); }
    else if (j.kind == 0 && j.aux == 1) {                          // :405 re-init path (memoised, see MpaFadsArgs)
      n = f.init_len; have = true;
      copy_path(buf, f.init_cells, n, lane);
      for (int i = 0; i < 5; ++i) sc[i] = f.init_stats[i];
    } else if (j.kind == 0 && j.aux == 3 && lane == 0) atomicAdd(&p.c.cnt->overflow, 1ull);
    if (have && lane < 5) f.cand_stats[(size_t)a * 5 + lane] = sc[lane];
    if (lane == 0) f.cand_len[a] = have ? n : 0;
  }
  if (lane == 0 && n) atomicAdd(&p.c.cnt->path_cells, (unsigned long long)n);
}
// memory step (MPA.py:381-384) then FADs acceptance (:402 / :408) for predator a
__global__ __launch_bounds__(64) void k_mpa_apply(int n, int path_cap, const int* slots, const int* c1_cells, const int* c1_len,
                                                  const double* c1_stats, const int* c2_cells, const int* c2_len,
                                                  const double* c2_stats, int* pop_cells, int* pop_len, double* pop_stats) {
  const int a = blockIdx.x;
  if (a >= n) return;
  const int slot = slots[a];
  double fit = pop_stats[(size_t)slot * 5 + 4];
  int take = 0;
  if (c1_stats[(size_t)a * 5 + 4] < fit) { take = 1; fit = c1_stats[(size_t)a * 5 + 4]; }
  if (c2_len[a] > 0 && c2_stats[(size_t)a * 5 + 4] < fit) take = 2;
  if (!take) return;
  const int* src = take == 1 ? c1_cells : c2_cells;
  const double* st = take == 1 ? c1_stats : c2_stats;
  const int L = take == 1 ? c1_len[a] : c2_len[a];
  for (int i = threadIdx.x; i < L; i += blockDim.x) pop_cells[(size_t)slot * path_cap + i] = src[(size_t)a * path_cap + i];
  if (threadIdx.x < 5) pop_stats[(size_t)slot * 5 + threadIdx.x] = st[(size_t)a * 5 + threadIdx.x];
  if (threadIdx.x == 0) pop_len[slot] = L;
}

// work estimates of the FADs detours (the phase items' come from k_mpa_propose): replay only the gating draws
__global__ void k_plan_mpa_fads(MpaFadsArgs p, float* est) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= p.n) return;
  float e = 0.f;
  Rng g; g.init(p.seed, DOM_MPA_FADS, (unsigned long long)p.iter, (unsigned long long)p.gidx[a]);
  if (g.random() < p.m.fads && g.random() < p.CF) {
    const int rr_ = (int)g.randint(0, p.c.G.R - 1), rc_ = (int)g.randint(0, p.c.G.C - 1);
    const int node = rr_ * p.c.G.C + rc_;
    if (p.c.G.occ[node] != 1) e = cell_dist(p.c.G, p.m.start, node) + cell_dist(p.c.G, node, p.m.target);
  }
  est[a] = e;
}

// memory step MPA.py:381-384
__global__ __launch_bounds__(64) void k_mpa_memory(int n, int path_cap, const int* slots, const int* cand_cells,
                                                   const int* cand_len, const double* cand_stats, int* pop_cells,
                                                   int* pop_len, double* pop_stats) {
  const int a = blockIdx.x;
  if (a >= n) return;
  const int slot = slots[a];
  if (!(cand_stats[(size_t)a * 5 + 4] < pop_stats[(size_t)slot * 5 + 4])) return;
  const int L = cand_len[a];
  for (int i = threadIdx.x; i < L; i += blockDim.x) pop_cells[(size_t)slot * path_cap + i] = cand_cells[(size_t)a * path_cap + i];
  if (threadIdx.x < 5) pop_stats[(size_t)slot * 5 + threadIdx.x] = cand_stats[(size_t)a * 5 + threadIdx.x];
  if (threadIdx.x == 0) pop_len[slot] = L;
}

// ---- device-resident iteration control: stable sort, elite pick, local view ----------------------
// K8: the stable sort behind list.sort(key=fitness) (MPA.py:321,333,412, ga_solver.py:209) and the longest-first work queue --
// hand-written, three launches, no library.  A RANK sort: every element's final position is the number of elements that sort
// before it under the total order (key, current position) -- stable by construction, no passes over digits, no scratch that
// depends on the key width, any n.  The keys are the order-preserving u64 images of the fp64 (or fp32) values, so one 64-bit
// unsigned compare decides a pair.
//   k_sort_prep    key image of every element (through the current list order for list.sort), its payload saved, rank <- 0;
//   k_rank_count   a wavefront owns 64 elements (one per lane) and a slice of the others: the other key is wave-uniform -- a
//                  scalar load -- so a pair costs a compare and an add-with-carry; elements known to lie before / after the
//                  wave's own 64 need no tie-break at all (<= / <); partial counts meet in one atomicAdd per element;
//   k_rank_scatter out[rank[i]] = payload[i].
// Work is n^2 / 64 wave-iterations spread over the whole chip (n = 4 096: 0.26 M, a few microseconds; 16 384: 4.2 M; 65 536:
// 67 M ~ 0.1 ms -- every batch this orders runs for tens of milliseconds).
PF_DEV unsigned long long key_image_f64(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (b == 0x8000000000000000ull) b = 0;                           // -0.0 == 0.0 for list.sort
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);             // negatives reversed below the positives; +inf last
}
// mode 0: key[pos] = image(vals[order[pos] * stride + offset]), payload[pos] = order[pos]   (list.sort of the CURRENT order)
// mode 1: key[i] = ~image(est[i]) (descending), payload[i] = i                              (longest-expected-first queue)
__global__ void k_sort_prep(int n, int mode, const double* vals, int stride, int offset, const int* order, const float* est,
                            unsigned long long* key, int* payload, unsigned* rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) { const int id = order[i]; key[i] = key_image_f64(vals[(size_t)id * stride + offset]); payload[i] = id; }
  else { key[i] = ~key_image_f64((double)est[i]); payload[i] = i; }
  rank[i] = 0u;
}
__global__ __launch_bounds__(64) void k_rank_count(int n, int jsplit, int jchunk, const unsigned long long* __restrict__ key,
                                                   unsigned* __restrict__ rank) {
  const int lane = lane_id();
  const int tile = (int)(blockIdx.x / (unsigned)jsplit), sp = (int)(blockIdx.x - (unsigned)tile * (unsigned)jsplit);
  const int t0 = tile * 64, i = t0 + lane;
  const unsigned long long ki = i < n ? key[i] : ~0ull;
  int j = sp * jchunk;
  int j1 = j + jchunk; if (j1 > n) j1 = n;
  const int a1 = j1 < t0 ? j1 : t0, b1 = j1 < t0 + 64 ? j1 : t0 + 64;
  unsigned cnt = 0;
  // eight keys per scalar load (one s_load_dwordx16), then eight compare + add-with-carry pairs: a load per key would leave the
  // wave waiting ~200 clocks for every single compare
#define PF_RANK_RUN(END, TEST)                                                                  \
  for (; j + 8 <= (END); j += 8) {                                                              \
    unsigned long long kk[8];                                                                   \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) kk[u] = key[j + u];                           \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) { const unsigned long long kj = kk[u]; const int jj = j + u; (void)jj; cnt += (TEST) ? 1u : 0u; } \
  }                                                                                             \
  for (; j < (END); ++j) { const unsigned long long kj = key[j]; const int jj = j; (void)jj; cnt += (TEST) ? 1u : 0u; }
  PF_RANK_RUN(a1, kj <= ki)                                         // every j here is an earlier position: a tie sorts before me
  PF_RANK_RUN(b1, kj < ki || (kj == ki && jj < i))                  // my own 64: the full order (key, position)
  PF_RANK_RUN(j1, kj < ki)                                          // later positions: only strictly smaller keys
#undef PF_RANK_RUN
  if (i < n && cnt) atomicAdd(&rank[i], cnt);
}
__global__ void k_rank_scatter(int n, const unsigned* rank, const int* payload, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[rank[i]] = payload[i];
}
__global__ void k_gather_col(int n, const double* src, int stride, int offset, double* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(size_t)i * stride + offset];
}
// the elite = population[0] after the sort (MPA.py:334): its row, length and stats into the elite buffer
__global__ __launch_bounds__(256) void k_mpa_pick_elite(int path_cap, const int* pop_cells, const int* pop_len, const double* pop_stats,
                                                       const int* order, int lo, int* e_cells, int* e_len, double* e_stats) {
  const int slot = order[0] - lo;                                  // (sharded: order holds global ids, this rank stores [lo, ...))
  const int L = pop_len[slot];
  for (int i = threadIdx.x; i < L; i += 256) e_cells[i] = pop_cells[(size_t)slot * path_cap + i];
  if (threadIdx.x < 5) e_stats[threadIdx.x] = pop_stats[(size_t)slot * 5 + threadIdx.x];
  if (threadIdx.x == 0) *e_len = L;
}
// positions (in the global sorted order) and storage slots of the predators this rank stores, in position order
__global__ __launch_bounds__(1024) void k_mpa_local_view(int N, const int* gorder, int lo, int hi, int* gidx, int* slot) {
  __shared__ int wsum[16];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int p0 = 0; p0 < N; p0 += 1024) {
    const int pos = p0 + threadIdx.x;
    const int gid = pos < N ? gorder[pos] : -1;
    const bool mine = gid >= lo && gid < hi;
    const unsigned long long m = __ballot(mine);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __builtin_popcountll(m);
    __syncthreads();
    int off = base;
    for (int k = 0; k < w; ++k) off += wsum[k];
    if (mine) { const int at = off + __builtin_popcountll(m & ((1ull << lane) - 1ull)); gidx[at] = pos; slot[at] = gid - lo; }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; base += t; }
    __syncthreads();
  }
}

// ---- GA operators on the device (ga_solver.py:136-160, 186-205) ---------------------------------------
// Tournament selection draws from ONE stream per generation (seed, DOM_GA_SELECT, gen, 0), slot after slot, with
// data-dependent draw counts (random.sample's rejection loops): inherently sequential, so one thread replays it --
// N x k draws, microseconds -- against the fitness column in HBM.  psid[s] = storage id of the parent chosen for slot s.
__global__ void k_ga_select(unsigned long long seed, int gen, int n, int k, const double* fit_all, const int* gorder, int* pool, int* psid) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Rng r; r.init(seed, DOM_GA_SELECT, (unsigned long long)gen, 0);
  long long setsize = 21;                                            // random.sample: pool copy for small n, rejection set otherwise
  if (k > 5) { long long p4 = 1; while (p4 < (long long)k * 3) p4 *= 4; setsize += p4; }
  int sel[64];
  for (int s_ = 0; s_ < n; ++s_) {
    if (n <= setsize) {
      for (int i = 0; i < n; ++i) pool[i] = i;
      for (int i = 0; i < k; ++i) { const int j = (int)r.randbelow((unsigned long long)(n - i)); sel[i] = pool[j]; pool[j] = pool[n - i - 1]; }
    } else {
      for (int i = 0; i < k; ++i) {
        int j;
        for (;;) { j = (int)r.randbelow((unsigned long long)n); bool seen = false; for (int q = 0; q < i; ++q) seen |= sel[q] == j; if (!seen) break; }
        sel[i] = j;
      }
    }
    int best = sel[0];                                               // min(tournament, key=fitness): the first minimum
    double bf = fit_all[gorder[best]];
    for (int i = 1; i < k; ++i) { const double f = fit_all[gorder[sel[i]]]; if (f < bf) { bf = f; best = sel[i]; } }
    psid[s_] = gorder[best];
  }
}
// crossover + mutation, one thread per pair of children (stream (seed, DOM_GA, gen, pair), ga_solver.py:144-160,186-194):
// writes the children with index in [child0, child0 + nchild) to out[(index - child0) * W ...]
__global__ void k_ga_breed(unsigned long long seed, int gen, int N, int W, double cx_rate, double mut_rate, const uint8_t* occ, int R, int C,
                           const int* chrom_all, const int* psid, int child0, int nchild, int* out) {
  const int pair = child0 / 2 + blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = 2 * pair;
  if (idx >= child0 + nchild) return;
  const int* p1 = chrom_all + (size_t)psid[idx % N] * W;
  const int* p2 = chrom_all + (size_t)psid[(idx + 1) % N] * W;
  Rng r; r.init(seed, DOM_GA, (unsigned long long)gen, (unsigned long long)pair);
  int point = 0;
  if (r.random() < cx_rate) point = W > 1 ? (int)r.randint(1, W - 1) : 0;   // :145-148
  for (int which = 0; which < 2; ++which) {                          // :154-160, child 1 then child 2 on the same stream
    const int ci = idx + which;
    const bool keep = ci >= child0 && ci < child0 + nchild && ci < N;
    for (int i = 0; i < W; ++i) {
      const bool tail = point > 0 && i >= point;
      int g_ = which == 0 ? (tail ? p2[i] : p1[i]) : (tail ? p1[i] : p2[i]);
      if (r.random() < mut_rate)
        for (;;) {                                                   // :48-53 rejection-sample a free cell
          const int rr = (int)r.randint(0, R - 1), cc = (int)r.randint(0, C - 1);
          if (occ[(size_t)rr * C + cc] != 1) { g_ = rr * C + cc; break; }
        }
      if (keep) out[(size_t)(ci - child0) * W + i] = g_;
    }
  }
}
// new_population[i] = child i if it decoded, else the parent it falls back to (ga_solver.py:198-205: parents[i], i.e.
// p1 for even i, p2 for odd i).  Storage id of the new individual = its child index.  A fallback parent's path is
// copied when this rank stores it, else the row is marked absent (len -1): a path is decode(chromosome), re-derived on demand.
__global__ __launch_bounds__(64) void k_ga_assemble(int n_loc, int W, int cap, int lo, const int* kid_len, const int* kid_chrom,
                                                    const double* kid_stats, const int* kid_cells, const int* psid, const int* chrom_old,
                                                    const double* stats_old, const int* cells_old, const int* len_old, int old_lo, int old_hi,
                                                    int* chrom_new, double* stats_new, int* cells_new, int* len_new) {
  const int i = blockIdx.x;
  if (i >= n_loc) return;
  const int t = threadIdx.x;
  if (kid_len[i] > 0) {
    for (int k = t; k < W; k += 64) chrom_new[(size_t)i * W + k] = kid_chrom[(size_t)i * W + k];
    if (t < 5) stats_new[(size_t)i * 5 + t] = kid_stats[(size_t)i * 5 + t];
    for (int k = t; k < kid_len[i]; k += 64) cells_new[(size_t)i * cap + k] = kid_cells[(size_t)i * cap + k];
    if (t == 0) len_new[i] = kid_len[i];
  } else {
    const int sid = psid[lo + i];
    for (int k = t; k < W; k += 64) chrom_new[(size_t)i * W + k] = chrom_old[(size_t)sid * W + k];
    if (t < 5) stats_new[(size_t)i * 5 + t] = stats_old[(size_t)sid * 5 + t];
    int L = -1;
    if (sid >= old_lo && sid < old_hi) {
      L = len_old[sid - old_lo];
      for (int k = t; k < L; k += 64) cells_new[(size_t)i * cap + k] = cells_old[(size_t)(sid - old_lo) * cap + k];
    }
    if (t == 0) len_new[i] = L;
  }
}

// ===========================================================================
// device self-tests
// ===========================================================================
__global__ void k_selftest_sqrt(int n, const long long* in, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_sqrt((double)in[i]);
}
__global__ void k_selftest_rng(unsigned long long seed, unsigned long long dom, unsigned long long it,
                               unsigned long long agent, unsigned long long* u64, double* f64, long long* i64) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Rng g;
  g.init(seed, dom, it, agent); for (int i = 0; i < 8; ++i) u64[i] = g.next64();
  g.init(seed, dom, it, agent); for (int i = 0; i < 8; ++i) f64[i] = g.random();
  g.init(seed, dom, it, agent);
  for (int i = 0; i < 16; ++i) i64[i] = g.randint(-5, 17);
  for (int i = 0; i < 4; ++i) i64[16 + i] = g.randint(0, 0);
  for (int i = 0; i < 4; ++i) i64[20 + i] = g.randint(0, 1ll << 40);
  i64[34] = (long long)g.ctr;
  g.init(seed, dom, it, agent);
  for (int i = 0; i < 16; ++i) f64[8 + i] = g.normalvariate(0.0, 1.0);
  for (int i = 0; i < 4; ++i) f64[24 + i] = g.normalvariate(0.0, 0.7);
  i64[35] = (long long)g.ctr;
  g.init(seed, dom, it, agent); for (int i = 0; i < 8; ++i) f64[28 + i] = g.uniform(0.0, 2.0 * 3.141592653589793);
  g.init(seed, dom, it, agent);
  const unsigned long long ns[10] = {1, 2, 3, 5, 8, 100, 1000, 7, 1, 1};
  for (int i = 0; i < 10; ++i) i64[24 + i] = (long long)g.randbelow(ns[i]);
  i64[36] = (long long)g.ctr;
}

// proposals of n keyed streams (seed, DOM_MPA, 0, i) from given cells: the device arithmetic + its doubt flags
__global__ void k_selftest_targets(Grid G, unsigned long long seed, int n, int is_levy, double beta, double sigma, double scale,
                                   const int* cur, const int* elite, double eps_log, double eps_round, int* out, unsigned char* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Rng g; g.init(seed, DOM_MPA, 0, (unsigned long long)i);
  Doubt d; d.eps_log = eps_log; d.eps_round = eps_round; d.hit = false;
  out[i] = is_levy ? levy_target(g, G, cur[i], scale, beta, sigma, d) : brownian_target(g, G, cur[i], elite[i], scale, d);
  flag[i] = d.hit ? 1 : 0;
}

// ===========================================================================
// host side
// ===========================================================================
struct pf_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  int R = 0, C = 0, RC = 0;
  uint8_t *d_occ = nullptr, *d_mm_r1 = nullptr, *d_mm_r0 = nullptr, *d_mm_r1_nd = nullptr, *d_mm_r0_nd = nullptr,
          *d_d2near = nullptr;
  std::vector<uint8_t> h_occ;
  int* d_comp[4] = {nullptr, nullptr, nullptr, nullptr};   // component labels per move-mask policy (lazy)
  int nslots = 0;
  int rec_policy = -1;   // which move-mask variant the search records currently carry
  Rec* d_rec = nullptr;
  char* d_tier2 = nullptr;
  uint32_t* d_slot_state = nullptr;
  int* d_work = nullptr;
  DevCounters* d_cnt = nullptr;
  pf_counters last = {};
  double* d_pen = nullptr;
  double pen_min_safe = -1.0;
  int* d_d2wide = nullptr; int wide_W = 0; double* d_penw = nullptr; int penw_n = 0; double penw_min_safe = -1.0;   // min_safe_distance > 15.9
  int edt_radius = 7;      // radius of the obstacle-distance window d2near was built with
  double obst_frac = -1.0; // share of obstacle cells (lazy; picks the plateau kernels on open maps)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  float last_ms = 0.f;
  std::string err;
  // MAACO
  bool maaco_ready = false;
  pf_maaco_params mp = {};
  double *d_tau = nullptr, *d_taua = nullptr, *d_eta = nullptr, *d_dep = nullptr, *d_tep = nullptr;
  bool bits_clean = false;            // the visit-bit matrix is all zero (k_tau_deposit leaves it so)
  int marks_n = 0; const int* marks_cells = nullptr;   // the last walk batch marked its own deposits (for these n ants / this path buffer)
  double* d_mstate = nullptr;         // pf_maaco_iterate's 13 doubles
  double* h_mstate = nullptr; double* h_mstate_dev = nullptr;   // ... mirrored by the device into pinned host memory (no copy is enqueued)
  int* d_best_row = nullptr; int best_row_cap = 0;   // the overall best ant's path row, [0] = length (pf_maaco_iterate / pf_maaco_best_path)
  char* d_mctl = nullptr;             // pf_maaco_iterate's own {work counter @0, DevCounters @16}: zeroed by k_maaco_best_take for the next walk
  bool mctl_clean = false;
  unsigned* d_visit = nullptr; unsigned* d_visit_epoch = nullptr; int maaco_slots = 0; int maaco_cus = 256;
  unsigned long long* d_bits = nullptr; size_t bits_words = 0; int dep_cap = 0;
  uint8_t* d_flag = nullptr;          // chunk flags of d_bits: [(RC + 63) / 64][bits_words] bytes (MaacoArgs::flag)
  // MPA
  bool mpa_ready = false;
  pf_mpa_params mpp = {};
  pf_score_params mps = {};
  int* d_tmp = nullptr; int tmp_cap = 0;
  double* d_elite_stats = nullptr;
  int* d_init_cells = nullptr; int init_len = 0; int init_cap = 0; double* d_init_stats = nullptr;
  double* d_ds = nullptr; double* d_dt = nullptr;   // static shortest distances from the start / to the target (pruning bounds)
  float* d_est = nullptr; int* d_queue = nullptr; int est_cap = 0;   // work estimates and the longest-first queue of a batch
  void* d_jobs = nullptr; void* d_jres = nullptr; int job_cap = 0;   // the sweep's search jobs / results (k_mpa_plan -> k_mpa_search -> k_mpa_finish)
  int2* d_prop = nullptr; int* d_doubt = nullptr; int prop_cap = 0;   // MPA proposals {idx, target cell}; doubt list [0] = count, [1..] = predators
  long long doubts_resolved = 0;
  void* d_scan = nullptr; void* d_scan3 = nullptr;   // results of the small device scans
  unsigned long long* d_okey = nullptr; int* d_opay = nullptr; unsigned* d_orank = nullptr; int okey_cap = 0;   // rank_sort scratch: key images, payloads, ranks
  int* d_elite_cells = nullptr; int* d_elite_len = nullptr;   // the elite of the iteration (MPA.py:334), device resident
  int* d_ga_pool = nullptr; int ga_pool_cap = 0;              // random.sample's pool copy (small populations)
  unsigned long long* d_st_lab = nullptr; int* d_st_touched = nullptr; unsigned char* d_st_par = nullptr; unsigned* d_st_epoch = nullptr;   // pf_settle.h scratch
  int dep_words = 0; long long dep_done = 0;   // MAACO deposit in progress: words of the bit matrix, cells already folded
  ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1;   // RCCL communicator over xGMI (pf_comm_init); collectives run on `stream`
  std::vector<hipEvent_t> span_ev; int span_n = 0; bool span_open = false; double span_ms = 0.0; long long span_cnt = 0;   // pf_span_*: HIP-event timed spans on `stream`
  long long d2h_small = 0, d2h_bulk = 0, d2h_bulk_bytes = 0;   // device-to-host copies the library made (f1/f2 accounting): <= 128 B / larger
};

static long long g_step_cap = 0;   // > 0: lowers the connectors' step cap (pf_set_option "astar_step_cap": tests of the cap path)
static std::string g_create_err;
static int fail(pf_handle* h, const char* what, hipError_t e) {
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  if (h) h->err = m; else g_create_err = m;
  return -1;
}
static int failmsg(pf_handle* h, const std::string& m) { if (h) h->err = m; else g_create_err = m; return -2; }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, #call, e_); } while (0)

// RCCL entry points, resolved from librccl on first use (see pf_comm_* at the end of the file)
namespace {
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
Rccl g_rccl;
const char* rccl_load() {
  if (g_rccl.so) return nullptr;
  // RTLD_LOCAL: a process that also hosts torch has torch's own bundled librccl loaded; the two must not interpose
  void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) return "pf_comm: librccl.so not found (dlopen)";
#define PF_SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(so, name); if (!g_rccl.field) return "pf_comm: librccl misses " name; } while (0)
  PF_SYM(GetUniqueId, "ncclGetUniqueId"); PF_SYM(CommInitRank, "ncclCommInitRank"); PF_SYM(CommDestroy, "ncclCommDestroy");
  PF_SYM(GetErrorString, "ncclGetErrorString"); PF_SYM(AllGather, "ncclAllGather"); PF_SYM(Broadcast, "ncclBroadcast");
  PF_SYM(AllReduce, "ncclAllReduce"); PF_SYM(Send, "ncclSend"); PF_SYM(Recv, "ncclRecv"); PF_SYM(GroupStart, "ncclGroupStart");
  PF_SYM(GroupEnd, "ncclGroupEnd");
#undef PF_SYM
  g_rccl.so = so;
  return nullptr;
}
int nccl_fail(pf_handle* h, const char* what, ncclResult_t r) {
  return failmsg(h, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
}
}  // namespace

// Connected components of the free cells under one move-mask policy: host flood fill over the device-built
// masks (once per policy and grid).  On any failure the labels stay null and the searches simply run in full.
static const int* ensure_comp(pf_handle* h, int policy, const uint8_t* d_mm) {
  if (h->d_comp[policy]) return h->d_comp[policy];
  const int RC = h->RC, C = h->C;
  std::vector<uint8_t> mm(RC);
  if (hipMemcpy(mm.data(), d_mm, RC, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  static const int DR[8] = {0, 0, 1, -1, 1, 1, -1, -1}, DC[8] = {1, -1, 0, 0, 1, -1, 1, -1};   // helper.py:38-52 order
  std::vector<int> comp(RC, -1), stack;
  int next = 0;
  for (int s0 = 0; s0 < RC; ++s0) {
    if (h->h_occ[s0] == 1 || comp[s0] >= 0) continue;
    comp[s0] = next; stack.push_back(s0);
    while (!stack.empty()) {
      const int c = stack.back(); stack.pop_back();
      const unsigned m = mm[c];
      for (int k = 0; k < 8; ++k) if ((m >> k) & 1u) {
        const int n = c + DR[k] * C + DC[k];
        if (comp[n] < 0) { comp[n] = next; stack.push_back(n); }
      }
    }
    next += 1;
  }
  int* d = nullptr;
  if (hipMalloc(&d, (size_t)RC * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, comp.data(), (size_t)RC * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  h->d_comp[policy] = d;
  return d;
}
static Grid make_grid(pf_handle* h, int allow_diag, int restrict_corner) {
  Grid G;
  G.occ = h->d_occ;
  G.mm = allow_diag ? (restrict_corner ? h->d_mm_r1 : h->d_mm_r0) : (restrict_corner ? h->d_mm_r1_nd : h->d_mm_r0_nd);
  G.comp = ensure_comp(h, (allow_diag ? 2 : 0) | (restrict_corner ? 1 : 0), G.mm);
  G.d2near = h->d_d2near; G.R = h->R; G.C = h->C;
  G.magicC = ((1ull << 40) / (uint64_t)h->C) + 1;
  G.step_cap = g_step_cap;
  return G;
}

extern "C" {

const char* pf_version(void) { return "pathfit 0.1 (gfx950)"; }
int pf_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
const char* pf_last_error(pf_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }
void* pf_stream(pf_handle* h) { return h ? (void*)h->stream : nullptr; }
int pf_sync(pf_handle* h) { if (!h) return -2; CK(hipStreamSynchronize(h->stream)); return 0; }

__global__ void k_and_mask(uint8_t* dst, const uint8_t* src, int n, unsigned m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] & m;
}

int pf_create(const uint8_t* grid, int32_t R, int32_t C, int32_t device, pf_handle** out) {
  pf_handle* h = nullptr;
  if (!grid || !out || R < 1 || C < 1 || R > 4096 || C > 4096) return failmsg(nullptr, "pf_create: bad arguments (1 <= R,C <= 4096)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return failmsg(nullptr, "pf_create: no HIP device available (this engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return failmsg(nullptr, "pf_create: device ordinal out of range");
  h = new pf_handle();
  h->device = device; h->R = R; h->C = C; h->RC = R * C;
  #define CKC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(nullptr, #call, e_); pf_destroy(h); return -1; } } while (0)
  CKC(hipSetDevice(device));
  CKC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CKC(hipEventCreate(&h->ev0)); CKC(hipEventCreate(&h->ev1)); CKC(hipEventCreateWithFlags(&h->ev2, hipEventDisableTiming | hipEventReleaseToSystem));
  h->h_occ.resize(h->RC);
  for (int i = 0; i < h->RC; ++i) h->h_occ[i] = grid[i] == 1 ? 1 : 0;
  CKC(hipMalloc(&h->d_occ, h->RC)); CKC(hipMalloc(&h->d_mm_r1, h->RC)); CKC(hipMalloc(&h->d_mm_r0, h->RC));
  CKC(hipMalloc(&h->d_mm_r1_nd, h->RC)); CKC(hipMalloc(&h->d_mm_r0_nd, h->RC)); CKC(hipMalloc(&h->d_d2near, h->RC));
  CKC(hipMemcpyAsync(h->d_occ, h->h_occ.data(), h->RC, hipMemcpyHostToDevice, h->stream));
  const int nb = (h->RC + 255) / 256;
  k_grid_prep<<<nb, 256, 0, h->stream>>>(h->d_occ, R, C, h->d_mm_r1, h->d_mm_r0, h->d_d2near, 7);
  k_and_mask<<<nb, 256, 0, h->stream>>>(h->d_mm_r1_nd, h->d_mm_r1, h->RC, 0x0Fu);
  k_and_mask<<<nb, 256, 0, h->stream>>>(h->d_mm_r0_nd, h->d_mm_r0, h->RC, 0x0Fu);
  CKC(hipMalloc(&h->d_work, sizeof(int)));
  CKC(hipMalloc(&h->d_cnt, sizeof(DevCounters)));
  CKC(hipMalloc(&h->d_pen, 256 * sizeof(double)));
  CKC(hipMalloc(&h->d_elite_stats, 5 * sizeof(double)));
  CKC(hipStreamSynchronize(h->stream));
  #undef CKC
  *out = h;
  return 0;
}

void pf_destroy(pf_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  void* ptrs[] = {h->d_occ, h->d_mm_r1, h->d_mm_r0, h->d_mm_r1_nd, h->d_mm_r0_nd, h->d_d2near, h->d_rec, h->d_slot_state,
                  h->d_work, h->d_cnt, h->d_pen, h->d_tier2, h->d_tau, h->d_taua, h->d_eta, h->d_dep, h->d_tep, h->d_visit, h->d_visit_epoch,
                  h->d_bits, h->d_flag, h->d_mstate, h->d_mctl, h->d_best_row, h->d_d2wide, h->d_penw, h->d_tmp, h->d_elite_stats, h->d_init_cells, h->d_init_stats, h->d_est, h->d_queue, h->d_jobs, h->d_jres, h->d_prop, h->d_doubt, h->d_scan, h->d_scan3, h->d_okey, h->d_opay, h->d_orank, h->d_elite_cells, h->d_elite_len, h->d_ga_pool, h->d_st_lab, h->d_st_touched, h->d_st_par, h->d_st_epoch,
                  h->d_comp[0], h->d_comp[1], h->d_comp[2], h->d_comp[3], h->d_ds, h->d_dt};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (hipEvent_t e : h->span_ev) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->h_mstate) (void)hipHostFree(h->h_mstate);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// Dynamic maps: replace the occupancy of an existing handle (same R x C).  K0 runs again on the
// device (move masks, obstacle distances), everything derived from the old map is dropped (component labels, the
// search records' embedded masks, MPA's initial path and distance bounds are rebuilt by the next pf_mpa_setup).
int pf_update_grid(pf_handle* h, const uint8_t* grid) {
  if (!h) return -2;
  if (!grid) return failmsg(h, "pf_update_grid: bad arguments");
  CK(hipSetDevice(h->device));
  for (int i = 0; i < h->RC; ++i) h->h_occ[i] = grid[i] == 1 ? 1 : 0;
  CK(hipMemcpyAsync(h->d_occ, h->h_occ.data(), h->RC, hipMemcpyHostToDevice, h->stream));
  const int nb = (h->RC + 255) / 256;
  k_grid_prep<<<nb, 256, 0, h->stream>>>(h->d_occ, h->R, h->C, h->d_mm_r1, h->d_mm_r0, h->d_d2near, h->edt_radius);
  k_and_mask<<<nb, 256, 0, h->stream>>>(h->d_mm_r1_nd, h->d_mm_r1, h->RC, 0x0Fu);
  k_and_mask<<<nb, 256, 0, h->stream>>>(h->d_mm_r0_nd, h->d_mm_r0, h->RC, 0x0Fu);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 4; ++k) if (h->d_comp[k]) { (void)hipFree(h->d_comp[k]); h->d_comp[k] = nullptr; }
  h->rec_policy = -1;                                               // the records embed the old move masks
  h->wide_W = 0;                                                    // the wide distance table belongs to the old map
  h->obst_frac = -1.0;
  h->mpa_ready = false; h->maaco_ready = false;                     // their tables (initial path, bounds, tau / eta) belong to the old map
  if (h->d_ds) { (void)hipFree(h->d_ds); h->d_ds = nullptr; }
  if (h->d_dt) { (void)hipFree(h->d_dt); h->d_dt = nullptr; }
  return 0;
}

int pf_dev_alloc(pf_handle* h, int64_t bytes, void** d_out) { if (!h) return -2; if (!d_out || bytes < 0) return failmsg(h, "pf_dev_alloc: bad arguments"); CK(hipSetDevice(h->device)); CK(hipMalloc(d_out, (size_t)(bytes > 0 ? bytes : 1))); return 0; }
int pf_dev_free(pf_handle* h, void* p) { if (!h) return -2; if (!p) return 0; CK(hipSetDevice(h->device)); CK(hipFree(p)); return 0; }
int pf_h2d(pf_handle* h, void* d, const void* s, int64_t n) { if (!h) return -2; if (n == 0) return 0; if (!d || !s || n < 0) return failmsg(h, "pf_h2d: bad arguments"); CK(hipMemcpyAsync(d, s, (size_t)n, hipMemcpyHostToDevice, h->stream)); CK(hipStreamSynchronize(h->stream)); return 0; }
int pf_d2h(pf_handle* h, void* d, const void* s, int64_t n) { if (!h) return -2; if (n == 0) return 0; if (!d || !s || n < 0) return failmsg(h, "pf_d2h: bad arguments"); CK(hipMemcpyAsync(d, s, (size_t)n, hipMemcpyDeviceToHost, h->stream)); CK(hipStreamSynchronize(h->stream)); if (n <= 128) h->d2h_small += 1; else { h->d2h_bulk += 1; h->d2h_bulk_bytes += n; } return 0; }
int pf_d2h_counts(pf_handle* h, int64_t* small_copies, int64_t* bulk_copies, int64_t* bulk_bytes) { if (!h) return -2; if (small_copies) *small_copies = h->d2h_small; if (bulk_copies) *bulk_copies = h->d2h_bulk; if (bulk_bytes) *bulk_bytes = h->d2h_bulk_bytes; return 0; }
int pf_d2d(pf_handle* h, void* d, const void* s, int64_t n) { if (!h) return -2; if (n == 0) return 0; if (!d || !s || n < 0) return failmsg(h, "pf_d2d: bad arguments"); CK(hipMemcpyAsync(d, s, (size_t)n, hipMemcpyDeviceToDevice, h->stream)); CK(hipStreamSynchronize(h->stream)); return 0; }
int pf_memset(pf_handle* h, void* d, int32_t b, int64_t n) { if (!h) return -2; if (n == 0) return 0; if (!d || n < 0) return failmsg(h, "pf_memset: bad arguments"); CK(hipMemsetAsync(d, b, (size_t)n, h->stream)); CK(hipStreamSynchronize(h->stream)); return 0; }
int pf_get_counters(pf_handle* h, pf_counters* out) { if (!h || !out) return -2; *out = h->last; return 0; }
float pf_last_kernel_ms(pf_handle* h) { return h ? h->last_ms : 0.0f; }

}  // extern "C"

// ---- scratch / launch helpers ----------------------------------------------
// resident agent slots per CU and LDS bin capacity; PF_SLOTS_PER_CU / PF_LDS_S override for experiments
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
static const int kSlotsPerCU = env_int("PF_SLOTS_PER_CU", 8);   // search slots (record / pool scratch) per CU = resident one-agent waves per CU at most
#ifdef PF_TWO_WAVE
static int g_two_wave = env_int("PF_TWO_WAVE", 0);   // MPA searches on two-wave workgroups: pop wave + pool wave (pf_astar_pr.h; pf_set_option "two_wave")
#endif                                               // (the two-wave engine is compiled only with -DPF_TWO_WAVE: measured 0.90x, DESIGN.md 4.2 -- and the
                                                     // pop loop's speed depends on what else its kernel carries)
static const int kWavesPerCU = env_int("PF_WAVES_PER_CU", kSlotsPerCU);   // resident one-agent waves per CU (LDS permitting)
static int g_mpa_prune = 1;   // exact bound pruning of MPA rebuilds (pf_set_option "mpa_prune")
static int g_settle_top = env_int("PF_SETTLE_TOP", 0);  // auto mode ("astar_settle" -1): per mille of a DECODE batch, from the head of the longest-first queue, whose A* searches
                                                         // also try the engine (pf_set_option "astar_settle_top")
static int g_settle_tail = env_int("PF_SETTLE_TAIL", 400);   // (r03: 600; re-measured after the r04 trip cuts: 200 / 400 / 600 / 800 -> ga512 23.9 / 23.8 / 23.6 / 23.1 k, pso512 13.2 / 13.4 / 12.9 / 12.9 k) auto mode: per mille of the SEARCH SLOTS; once no more agents of a decode batch than this are unfinished, every search that starts
                                                              // tries the engine (pf_set_option "astar_settle_tail")
static int g_settle = env_int("PF_SETTLE", -1);  // closed-set searches try the 64-nodes-per-trip engine first (pf_settle.h; pf_set_option "astar_settle"):
                                                  // -1 (default) Dijkstra only -- h == 0 makes every node regular, so it is never handed back, and it measures
                                                  // 1.6x (2048 concurrent searches) to 3.5x (one search) faster; 1 also A* (exact too -- certified or handed back --
                                                  // but 5-7 % of the searches ARE handed back and then cost both engines: no gain on the tail-bound batches); 0 never
static double g_doubt_log = 1.0 / 8589934592.0;   // 2^-33 relative margin on normalvariate's accept test (pf_set_option "mpa_doubt_log_e15" overrides, in 1e-15)
static double g_doubt_round = 1e-7;                // absolute margin on the fraction fed to round()   ("mpa_doubt_round_e15")
static int g_maaco_pack8_min = env_int("PF_MAACO_PACK8_MIN", 2048);   // ants per batch from which 8 ants share a wavefront
static int g_maaco_ahead = env_int("PF_MAACO_TOUCH", -1);   // load-ahead form of k_maaco_walk8 (pf_set_option "maaco_load_ahead"): -1 by occupancy (at most one wavefront per SIMD), 0 never, 1 always
static int g_maaco_groups = env_int("PF_MAACO_GROUPS", 8);   // ants per wavefront of k_maaco_walk8 (pf_set_option "maaco_ants_per_wave": 1..8)
static int g_maaco_mark = env_int("PF_MAACO_MARK", 1);   // successful ants mark their deposits in the walk kernel (pf_set_option "maaco_mark_in_walk")
static int g_tabu_epoch = -1;                      // test hook ("maaco_tabu_epoch"): >= 0 -> the next walk batch starts its tabu slots from this epoch (wrap coverage)
static const int kLdsS = 16;
static int ensure_slots(pf_handle* h, int allow_diag = 1, int restrict_corner = 1) {
  CK(hipSetDevice(h->device));
  if (!h->d_rec) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, h->device));
    int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!h->nslots) h->nslots = cus * kSlotsPerCU;
    size_t bytes = (size_t)h->nslots * h->RC * sizeof(Rec);
    // keep the search scratch under ~64 GiB of the 288 GB HBM
    while (bytes > (64ull << 30) && h->nslots > cus) { h->nslots /= 2; bytes = (size_t)h->nslots * h->RC * sizeof(Rec); }
    CK(hipMalloc(&h->d_rec, bytes));
    CK(hipMalloc(&h->d_slot_state, sizeof(uint32_t) * 2 * h->nslots));
    CK(hipMalloc(&h->d_tier2, (size_t)h->nslots * PF_POOL_STRIDE));
    // parallel closed-set engine: 17 bytes per cell and slot (labels start as all ones = "older than any search")
    const size_t rc = (size_t)h->RC, ns = (size_t)h->nslots;
    if (ns * rc * 17 <= (48ull << 30)) {
      CK(hipMalloc(&h->d_st_lab, ns * rc * 8)); CK(hipMalloc(&h->d_st_touched, ns * rc * 8)); CK(hipMalloc(&h->d_st_par, ns * rc));
      CK(hipMalloc(&h->d_st_epoch, ns * sizeof(unsigned)));
      CK(hipMemsetAsync(h->d_st_lab, 0xFF, ns * rc * 8, h->stream));
      CK(hipMemsetAsync(h->d_st_epoch, 0, ns * sizeof(unsigned), h->stream));
    }
  }
  const int policy = (allow_diag ? 2 : 0) | (restrict_corner ? 1 : 0);
  if (h->rec_policy != policy) {
    // records embed the static move mask of this diagonal policy: (re)initialise all slots
    Grid G = make_grid(h, allow_diag, restrict_corner);
    const size_t total = (size_t)h->nslots * h->RC;
    hipLaunchKernelGGL(k_slot_init, dim3(4096), dim3(256), 0, h->stream, h->d_rec, G.mm, h->RC, total);
    CK(hipGetLastError());
    std::vector<uint32_t> init(2 * h->nslots, 1u);
    CK(hipMemcpyAsync(h->d_slot_state, init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice, h->stream));
    CK(hipStreamSynchronize(h->stream));
    h->rec_policy = policy;
  }
  return 0;
}
// Open maps keep thousands of open entries within 1/64 of f (straight runs): they are dispatched to the kernels compiled
// with the plateau refills.  "Open" = fewer than 10 % obstacle cells (pf_set_option "plateau_kernels": -1 auto, 0 never, 1 always).
static int g_plateau_mode = env_int("PF_PLATEAU_KERNELS", -1);
static bool plateau_map(pf_handle* h) {
  if (g_plateau_mode >= 0) return g_plateau_mode != 0;
  if (h->obst_frac < 0.0) { size_t n = 0; for (uint8_t v : h->h_occ) n += v == 1; h->obst_frac = (double)n / (double)h->RC; }
  return h->obst_frac < 0.10;
}
static int ensure_elite_buf(pf_handle* h) {
  if (!h->d_elite_cells) { CK(hipMalloc(&h->d_elite_cells, sizeof(int) * (size_t)h->RC)); CK(hipMalloc(&h->d_elite_len, sizeof(int))); }
  return 0;
}
static Common make_common(pf_handle* h, int allow_diag, int restrict_corner, int S, int retry) {
  Common c;
  c.G = make_grid(h, allow_diag, restrict_corner);
  c.rec = h->d_rec; c.tier2 = h->d_tier2; c.slot_state = h->d_slot_state; c.work = h->d_work; c.queue = nullptr; c.cnt = h->d_cnt; c.S = S; c.retry = retry;
  const bool st_on = g_settle != 0 && h->d_st_lab;
  c.st_lab = st_on ? h->d_st_lab : nullptr; c.st_astar = g_settle > 0; c.st_top = 0; c.st_tail = 0; c.st_touched = h->d_st_touched; c.st_par = h->d_st_par; c.st_epoch = h->d_st_epoch;
  return c;
}
static int begin_batch(pf_handle* h) {
  CK(hipSetDevice(h->device));
  CK(hipMemsetAsync(h->d_cnt, 0, sizeof(DevCounters), h->stream));
  return 0;
}
static int end_batch(pf_handle* h, DevCounters* dc) {
  CK(hipMemcpyAsync(dc, h->d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  h->d2h_small += 1;                                              // the 72-byte counter block of every batch
  h->last.pops = dc->pops; h->last.pushes = dc->pushes; h->last.nbr_examined = dc->nbr; h->last.path_cells = dc->path_cells;
  h->last.steps = dc->steps; h->last.candidates = dc->candidates; h->last.decrease_keys = dc->deckey; h->last.overflow_agents = dc->overflow;
  h->last.pruned_rebuilds = dc->pruned; h->last.settled_searches = dc->settled; h->last.sequential_searches = dc->sequential;
  return 0;
}

// The stable rank sort (K8: k_sort_prep / k_rank_count / k_rank_scatter) of n elements, enqueued on the handle's stream.
// mode 0: d_out[] = the list order d_order[] re-sorted by vals[id * stride + offset] ascending (d_out may be d_order itself: the
// payload is saved before anything is written); mode 1: d_out[] = indices 0..n-1 by est[] descending.
static int rank_sort(pf_handle* h, int n, int mode, const double* d_vals, int stride, int offset, const int* d_order, const float* d_est, int* d_out) {
  if (n > h->okey_cap) {
    for (void* q : {(void*)h->d_okey, (void*)h->d_opay, (void*)h->d_orank}) if (q) CK(hipFree(q));
    CK(hipMalloc(&h->d_okey, sizeof(unsigned long long) * (size_t)n)); CK(hipMalloc(&h->d_opay, sizeof(int) * (size_t)n));
    CK(hipMalloc(&h->d_orank, sizeof(unsigned) * (size_t)n));
    h->okey_cap = n;
  }
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_sort_prep, dim3(nb), dim3(256), 0, h->stream, n, mode, d_vals, stride, offset, d_order, d_est, h->d_okey, h->d_opay, h->d_orank);
  // ~8192 wavefronts per launch when there is that much work; a slice of the other keys is never shorter than 256
  const int tiles = (n + 63) / 64;
  int jsplit = 8192 / tiles; const int jmax = (n + 255) / 256; if (jsplit > jmax) jsplit = jmax; if (jsplit < 1) jsplit = 1;
  const int jchunk = (n + jsplit - 1) / jsplit;
  hipLaunchKernelGGL(k_rank_count, dim3((unsigned)tiles * (unsigned)jsplit), dim3(64), 0, h->stream, n, jsplit, jchunk, (const unsigned long long*)h->d_okey, h->d_orank);
  hipLaunchKernelGGL(k_rank_scatter, dim3(nb), dim3(256), 0, h->stream, n, (const unsigned*)h->d_orank, (const int*)h->d_opay, d_out);
  CK(hipGetLastError());
  return 0;
}

// Sort the batch longest-expected-first from per-agent estimates produced by `plan` (a functor that launches plan
// kernels writing h->d_est[0..n)): the stable rank sort of (estimate, index) pairs, descending.  Nothing
// crosses PCIe and nothing waits: the sweep is queued behind it on the same stream.
template <typename Plan>
static int make_queue(pf_handle* h, int n, Plan plan) {
  if (n > h->est_cap) {
    for (void* q : {(void*)h->d_est, (void*)h->d_queue}) if (q) CK(hipFree(q));
    CK(hipMalloc(&h->d_est, sizeof(float) * (size_t)n)); CK(hipMalloc(&h->d_queue, sizeof(int) * (size_t)n));
    h->est_cap = n;
  }
  plan(h->d_est);
  CK(hipGetLastError());
  return rank_sort(h, n, 1, nullptr, 0, 0, nullptr, h->d_est, h->d_queue);
}

template <typename KArgs, typename Kern>
static int launch_with_retry(pf_handle* h, Kern kern, KArgs& args, int n, bool two_wave = false) {
  if (n <= 0) return 0;
  const int S = kLdsS;
  args.c.S = S; args.c.retry = 0;
#ifdef PF_TWO_WAVE
  const size_t lds = two_wave ? (size_t)PF_PR_LDS_BYTES : open_bytes(S);
#else
  const size_t lds = open_bytes(S);
#endif
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((160 * 1024) / lds); if (per_cu > kWavesPerCU) per_cu = kWavesPerCU; if (per_cu < 1) per_cu = 1;
  int grid = (h->nslots / kSlotsPerCU) * per_cu; if (grid > n) grid = n; if (grid > h->nslots) grid = h->nslots;
  CK(hipMemsetAsync(h->d_work, 0, sizeof(int), h->stream));
  CK(hipMemsetAsync(h->d_cnt, 0, sizeof(DevCounters), h->stream));
  CK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(two_wave ? 128 : 64), lds, h->stream, args);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev1, h->stream));
  DevCounters dc;
  if (end_batch(h, &dc)) return -1;
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return 0;
}

static int ensure_pen(pf_handle* h, double min_safe) {
  if (!(min_safe <= 15.9)) {
    // wide mode: helper.py:67-80 takes any min_safe_distance.  Exact i32 squared distances in a window of radius
    // W = ceil(min_safe) (device EDT), penalty LUT over d2 = 0 .. W^2 + 1 built with the host's libm like the small one.
    if (!(min_safe < 1e9)) return failmsg(h, "min_safe_distance must be finite");
    CK(hipSetDevice(h->device));
    int W = (int)ceil(min_safe); const int mx = h->R > h->C ? h->R : h->C; if (W > mx) W = mx;
    if (h->wide_W != W) {
      if (!h->d_d2wide) CK(hipMalloc(&h->d_d2wide, sizeof(int) * (size_t)h->RC));
      int* g = nullptr; CK(hipMalloc(&g, sizeof(int) * (size_t)h->RC));
      const int nb = (h->RC + 255) / 256;
      k_edt_rows<<<nb, 256, 0, h->stream>>>(h->d_occ, h->R, h->C, W, g);
      k_edt_cols<<<nb, 256, 0, h->stream>>>(g, h->R, h->C, W, W * W + 1, h->d_d2wide);
      CK(hipGetLastError());
      CK(hipStreamSynchronize(h->stream));
      CK(hipFree(g));
      h->wide_W = W; h->penw_min_safe = -1.0;
    }
    if (h->penw_min_safe != min_safe) {
      const int n = W * W + 2;
      std::vector<double> pen((size_t)n);
      for (int d2 = 0; d2 < n; ++d2) {
        const double d = sqrt((double)d2);                         // helper.py:76
        pen[d2] = (d2 <= W * W && d < min_safe) ? pow(min_safe - d, 2.0) : 0.0;   // helper.py:77-78 (float ** 2 -> libm pow)
      }
      if (h->d_penw) CK(hipFree(h->d_penw));
      CK(hipMalloc(&h->d_penw, sizeof(double) * (size_t)n));
      CK(hipMemcpyAsync(h->d_penw, pen.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
      CK(hipStreamSynchronize(h->stream));
      h->penw_n = n; h->penw_min_safe = min_safe;
    }
    return 0;
  }
  if (min_safe > (double)h->edt_radius) {                           // widen the obstacle-distance window once (K0 again, radius 15)
    CK(hipSetDevice(h->device));
    k_grid_prep<<<(h->RC + 255) / 256, 256, 0, h->stream>>>(h->d_occ, h->R, h->C, h->d_mm_r1, h->d_mm_r0, h->d_d2near, 15);
    CK(hipGetLastError());
    h->edt_radius = 15; h->pen_min_safe = -1.0;
  }
  if (h->pen_min_safe == min_safe) return 0;
  double pen[256];
  for (int d2 = 0; d2 < 256; ++d2) {
    double d = sqrt((double)d2);                                   // helper.py:76
    pen[d2] = (d2 < 255 && d < min_safe) ? pow(min_safe - d, 2.0) : 0.0;   // helper.py:77-78 (float ** 2 -> libm pow)
  }
  CK(hipMemcpyAsync(h->d_pen, pen, sizeof(pen), hipMemcpyHostToDevice, h->stream));
  CK(hipStreamSynchronize(h->stream));
  h->pen_min_safe = min_safe;
  return 0;
}
static int make_scorep(pf_handle* h, const pf_score_params* sp, ScoreP* out) {
  if (sp->variant == 0) { if (ensure_pen(h, sp->min_safe)) return -1; }
  else if (h->pen_min_safe < 0) { if (ensure_pen(h, 0.0)) return -1; }
  out->variant = sp->variant; out->restrict_policy = sp->restrict_policy; out->w_turn = sp->w_turn; out->w_safe = sp->w_safe;
  out->diag_pen = sp->diag_pen; out->pen = h->d_pen; out->d2w = nullptr; out->pen_n = 256;
  if (sp->variant == 0 && !(sp->min_safe <= 15.9)) { out->pen = h->d_penw; out->d2w = h->d_d2wide; out->pen_n = h->penw_n; }
  return 0;
}

extern "C" {

int pf_astar_batch(pf_handle* h, int32_t variant, int32_t allow_diag, int32_t restrict_corner, int32_t n,
                   const int32_t* d_start, const int32_t* d_target, const int64_t* d_avoid_off,
                   const int32_t* d_avoid_cells, int32_t path_cap, int32_t* d_cells, int32_t* d_len,
                   int32_t* d_status, int64_t* d_counters) {
  if (!h) return -2;
  if (n < 0 || path_cap < 1 || !d_start || !d_target || !d_cells || !d_len || !d_status) return failmsg(h, "pf_astar_batch: bad arguments");
  if (ensure_slots(h, allow_diag, restrict_corner)) return -1;
  AstarArgs a;
  a.c = make_common(h, allow_diag, restrict_corner, 16, 0);
  a.n = n; a.path_cap = path_cap; a.start = d_start; a.target = d_target;
  a.avoid_off = (const long long*)d_avoid_off; a.avoid_cells = d_avoid_cells;
  a.cells = d_cells; a.len = d_len; a.status = d_status; a.counters = (long long*)d_counters;
  if (n > 64) {
    if (make_queue(h, n, [&](float* est) { hipLaunchKernelGGL(k_plan_astar, dim3((n + 255) / 256), dim3(256), 0, h->stream, a.c.G, n, d_start, d_target, est); })) return -1;
    a.c.queue = h->d_queue;
    // (no head-of-queue settling here: a batch of single searches ends on its longest search, and the longest searches are the
    // ones that meet an anomaly and fall back -- astar1024: 182 -> 237 ms with 3 % of the items settled)
  }
  // sparse maps (plateaus of equal f along open runs) go to the separately compiled kernels with the plateau refills
  const bool plat = plateau_map(h);
  if (variant == PF_ASTAR_REF) return plat ? launch_with_retry(h, k_astar_batch<0, true>, a, n) : launch_with_retry(h, k_astar_batch<0, false>, a, n);
  if (variant == PF_ASTAR_MPA) {
    if (plat) return launch_with_retry(h, k_astar_batch<1, true>, a, n);
#ifdef PF_TWO_WAVE
    if (g_two_wave) return launch_with_retry(h, k_astar_batch<1, false, true>, a, n, true);
#endif
    return launch_with_retry(h, k_astar_batch<1, false>, a, n);
  }
  if (variant == PF_ASTAR_DIJKSTRA) return plat ? launch_with_retry(h, k_astar_batch<2, true>, a, n) : launch_with_retry(h, k_astar_batch<2, false>, a, n);
  return failmsg(h, "pf_astar_batch: unknown variant");
}

int pf_score_batch(pf_handle* h, const pf_score_params* sp, int32_t n, int32_t path_cap, const int32_t* d_cells,
                   const int32_t* d_len, double* d_stats) {
  if (!h) return -2;
  if (!sp || n < 0 || !d_cells || !d_len || !d_stats) return failmsg(h, "pf_score_batch: bad arguments");
  if (n == 0) return 0;
  ScoreArgs a;
  a.G = make_grid(h, 1, 1);
  if (make_scorep(h, sp, &a.sp)) return -1;
  a.n = n; a.path_cap = path_cap; a.cells = d_cells; a.len = d_len; a.stats = d_stats;
  CK(hipSetDevice(h->device));
  int grid = n < 16384 ? n : 16384;
  CK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(k_score_batch, dim3(grid), dim3(64), 0, h->stream, a);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev1, h->stream));
  CK(hipStreamSynchronize(h->stream));
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return 0;
}

int pf_decode_batch(pf_handle* h, int32_t allow_diag, int32_t restrict_corner, int32_t n, int32_t W,
                    const int32_t* d_wp_cells, const double* d_wp_pos, int32_t start, int32_t target,
                    int32_t path_cap, int32_t* d_cells, int32_t* d_len, int32_t* d_status,
                    const pf_score_params* sp, double* d_stats) {
  if (!h) return -2;
  if (n < 0 || W < 0 || W >= PF_MAX_SEARCHES_PER_EVAL || path_cap < 1 || (!d_wp_cells && !d_wp_pos && W > 0) || !d_cells || !d_len || !d_status ||
      start < 0 || start >= h->RC || target < 0 || target >= h->RC || (sp && !d_stats))
    return failmsg(h, "pf_decode_batch: bad arguments");
  if (ensure_slots(h, allow_diag, restrict_corner)) return -1;
  DecodeArgs a;
  a.c = make_common(h, allow_diag, restrict_corner, 16, 0);
  a.do_score = sp != nullptr;
  if (sp) { if (make_scorep(h, sp, &a.sp)) return -1; } else memset(&a.sp, 0, sizeof(a.sp));
  a.n = n; a.W = W; a.path_cap = path_cap; a.start = start; a.target = target;
  a.wp_cells = d_wp_cells; a.wp_pos = d_wp_pos; a.cells = d_cells; a.len = d_len; a.status = d_status; a.stats = d_stats;
  if (n > 64) {
    if (make_queue(h, n, [&](float* est) { hipLaunchKernelGGL(k_plan_decode, dim3((n + 255) / 256), dim3(256), 0, h->stream, a.c.G, n, W, d_wp_cells, d_wp_pos, start, target, est); })) return -1;
    a.c.queue = h->d_queue;
    // A decode is a chain of W + 1 closed-set searches, so a fallback costs one link, not the chain.  r02: the agents at the head
    // of the longest-first queue try the parallel settling engine from the start (ga512 121 -> 108 ms at 6 %).  r03: the TAIL
    // policy below replaces it by default (astar_settle_top 0): under full load the engine is traffic-bound and no faster than
    // the sequential loop, so nobody uses it while the chip is full; once the unfinished agents no longer fill 60 % of the
    // search slots, every search that starts does (one box: sequential only 117.7 ms, head 6 % 111.8, tail 60 % 102.3).
    if (g_settle < 0) a.c.st_top = (int)((long long)n * g_settle_top / 1000);
  }
  // ... and whatever is still running when the batch is nearly done (chains of W + 1 searches: the switch happens between
  // links, each search is certified or handed back on its own)
  // (measured against the chip, not the batch: a batch that never fills the search slots runs on the engine from its start)
  if (g_settle < 0) a.c.st_tail = (int)((long long)h->nslots * g_settle_tail / 1000);
  return plateau_map(h) ? launch_with_retry(h, k_decode_batch<true>, a, n) : launch_with_retry(h, k_decode_batch<false>, a, n);
}

int pf_pso_update(pf_handle* h, int32_t n, int32_t W, double w, double c1, double c2, double max_vel, double* d_pos,
                  double* d_vel, const double* d_pbest, const double* d_gbest, uint64_t seed, uint64_t iter,
                  uint64_t agent0) {
  if (!h) return -2;
  if (n < 0 || W < 1 || !d_pos || !d_vel || !d_pbest || !d_gbest) return failmsg(h, "pf_pso_update: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  PsoArgs a{n, W, h->R, h->C, w, c1, c2, max_vel, d_pos, d_vel, d_pbest, d_gbest, seed, iter, agent0, nullptr, nullptr};
  const int threads = 256, blocks = (n * W + threads - 1) / threads;
  CK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(k_pso_update, dim3(blocks), dim3(threads), 0, h->stream, a);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev1, h->stream));
  CK(hipStreamSynchronize(h->stream));
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return 0;
}

// pf_pso_update that also leaves the pre-update position / velocity in d_pos_keep / d_vel_keep; asynchronous (stream ordered)
int pf_pso_update_keep(pf_handle* h, int32_t n, int32_t W, double w, double c1, double c2, double max_vel, double* d_pos,
                       double* d_vel, const double* d_pbest, const double* d_gbest, uint64_t seed, uint64_t iter,
                       uint64_t agent0, double* d_pos_keep, double* d_vel_keep) {
  if (!h) return -2;
  if (n < 0 || W < 1 || !d_pos || !d_vel || !d_pbest || !d_gbest || !d_pos_keep || !d_vel_keep) return failmsg(h, "pf_pso_update_keep: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  PsoArgs a{n, W, h->R, h->C, w, c1, c2, max_vel, d_pos, d_vel, d_pbest, d_gbest, seed, iter, agent0, d_pos_keep, d_vel_keep};
  const int threads = 256, blocks = (n * W + threads - 1) / threads;
  hipLaunchKernelGGL(k_pso_update, dim3(blocks), dim3(threads), 0, h->stream, a);
  CK(hipGetLastError());
  return 0;
}
// one round of the asynchronous sweep committed in one launch (k_pso_commit); asynchronous (stream ordered)
int pf_pso_commit(pf_handle* h, int32_t m, int32_t W, int32_t path_cap, int32_t n_final, int32_t improver, double* d_pos,
                  double* d_vel, const double* d_pos_keep, const double* d_vel_keep, const double* d_stats, const int32_t* d_len,
                  const int32_t* d_cells, double* d_pbest, double* d_pbest_fit, int32_t* d_pb_cells, int32_t* d_pb_len,
                  double* d_gbest, double* d_gbest_stats, int32_t* d_gbest_path) {
  if (!h) return -2;
  if (m < 0 || W < 1 || path_cap < 1 || n_final < 0 || n_final > m || improver >= n_final || !d_pos || !d_vel || !d_stats || !d_len || !d_cells ||
      !d_pbest || !d_pbest_fit || !d_pb_cells || !d_pb_len || (n_final < m && (!d_pos_keep || !d_vel_keep)) ||
      (improver >= 0 && (!d_gbest || !d_gbest_stats || !d_gbest_path))) return failmsg(h, "pf_pso_commit: bad arguments");
  if (m == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_pso_commit, dim3(m), dim3(64), 0, h->stream, m, W, path_cap, n_final, improver, d_pos, d_vel, d_pos_keep, d_vel_keep,
                     d_stats, d_len, d_cells, d_pbest, d_pbest_fit, d_pb_cells, d_pb_len, d_gbest, d_gbest_stats, d_gbest_path);
  CK(hipGetLastError());
  return 0;
}

int pf_pso_pbest(pf_handle* h, int32_t n, int32_t W, const double* d_pos, const double* d_stats, const int32_t* d_len,
                 double* d_pbest, double* d_pbest_fit, int32_t* d_improved) {
  if (!h) return -2;
  if (n <= 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_pso_pbest, dim3(n), dim3(64), 0, h->stream, n, W, d_pos, d_stats, d_len, d_pbest, d_pbest_fit, d_improved);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(h->stream));
  return 0;
}

int pf_pso_pbest_paths(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                       const int32_t* d_improved, int32_t* d_pb_cells, int32_t* d_pb_len) {
  if (!h) return -2;
  if (n < 0 || path_cap < 1 || !d_cells || !d_len || !d_improved || !d_pb_cells || !d_pb_len) return failmsg(h, "pf_pso_pbest_paths: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_pso_pbest_paths, dim3(n), dim3(64), 0, h->stream, n, path_cap, d_cells, d_len, d_improved, d_pb_cells, d_pb_len);
  CK(hipGetLastError());
  return 0;
}

int pf_pso_scan(pf_handle* h, int32_t n, const double* d_stats, const int32_t* d_len, const int32_t* d_status,
                const double* d_pbest_fit, double gbest_fit, int32_t sync_mode, int32_t* idx_out, double* fit_out,
                int32_t* overflow_out) {
  if (!h) return -2;
  if (n < 0 || !d_stats || !d_len || !d_status || !d_pbest_fit || !idx_out || !fit_out || !overflow_out) return failmsg(h, "pf_pso_scan: bad arguments");
  *idx_out = -1; *fit_out = INFINITY; *overflow_out = 0;
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  if (!h->d_scan) CK(hipMalloc(&h->d_scan, 16));
  hipLaunchKernelGGL(k_pso_scan, dim3(1), dim3(256), 0, h->stream, n, d_stats, d_len, d_status, d_pbest_fit, gbest_fit, sync_mode,
                     (int*)h->d_scan, (double*)((char*)h->d_scan + 8));
  CK(hipGetLastError());
  struct { int idx, ovf; double fit; } r;
  CK(hipMemcpyAsync(&r, h->d_scan, 16, hipMemcpyDeviceToHost, h->stream));   // the sweep's one small D2H
  CK(hipStreamSynchronize(h->stream));
  h->d2h_small += 1;
  *idx_out = r.idx; *fit_out = r.fit; *overflow_out = r.ovf;
  return 0;
}

#ifdef PF_TRACE
extern "C" int pf_debug_trace(pf_handle* h, uint64_t* out) {
  CK(hipSetDevice(h->device));
  CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(uint64_t) * 4 * 16384));
  CK(hipMemcpyFromSymbol(out + 4 * 16384, HIP_SYMBOL(g_trace2), sizeof(uint64_t) * 4 * 16384));
  CK(hipMemcpyFromSymbol(out + 8 * 16384, HIP_SYMBOL(g_trace3), sizeof(uint64_t) * 4 * 16384));
  uint64_t* z = (uint64_t*)calloc(4 * 16384, 8);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace2), z, sizeof(uint64_t) * 4 * 16384));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace3), z, sizeof(uint64_t) * 4 * 16384)); free(z);
  return 0;
}
#endif
#ifdef PF_STAMPS
extern "C" int pf_debug_stamps(pf_handle* h, uint64_t* out, int reset) {
  CK(hipSetDevice(h->device));
  CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_stamps), sizeof(uint64_t) * 24));
  if (reset) { uint64_t z[24] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(pf::g_stamps), z, sizeof(z))); }
  return 0;
}
#endif
int pf_set_option(pf_handle* h, const char* name, int64_t value) {
  if (!name) return failmsg(h, "pf_set_option: bad arguments");
  if (!strcmp(name, "astar_step_cap")) { g_step_cap = value > 0 ? (long long)value : 0; return 0; }
  if (!strcmp(name, "maaco_pack8_min")) { g_maaco_pack8_min = (int)value; return 0; }
  if (!strcmp(name, "maaco_load_ahead")) { g_maaco_ahead = value < 0 ? -1 : (value ? 1 : 0); return 0; }
  if (!strcmp(name, "maaco_ants_per_wave")) { g_maaco_groups = value < 1 ? 1 : (value > 8 ? 8 : (int)value); return 0; }
  if (!strcmp(name, "mpa_prune")) { g_mpa_prune = value != 0; return 0; }
#ifdef PF_TWO_WAVE
  if (!strcmp(name, "two_wave")) { g_two_wave = value != 0; return 0; }
#else
  if (!strcmp(name, "two_wave")) { if (value == 0) return 0; return failmsg(h, "pf_set_option: two_wave is not built in (compile with -DPF_TWO_WAVE)"); }
#endif
  if (!strcmp(name, "maaco_mark_in_walk")) { g_maaco_mark = value != 0; return 0; }
  if (!strcmp(name, "astar_settle")) { g_settle = value < 0 ? -1 : (value != 0); return 0; }
  if (!strcmp(name, "astar_settle_tail")) { g_settle_tail = value < 0 ? 0 : (value > 1000 ? 1000 : (int)value); return 0; }
  if (!strcmp(name, "astar_settle_top")) { g_settle_top = value < 0 ? 0 : (value > 1000 ? 1000 : (int)value); return 0; }
  if (!strcmp(name, "plateau_kernels")) { g_plateau_mode = (int)value; return 0; }
  if (!strcmp(name, "maaco_tabu_epoch")) { g_tabu_epoch = (int)value; return 0; }
  if (!strcmp(name, "mpa_doubt_log_e15")) { g_doubt_log = value < 0 ? 1.0 / 8589934592.0 : (double)value * 1e-15; return 0; }
  if (!strcmp(name, "mpa_doubt_round_e15")) { g_doubt_round = value < 0 ? 1e-7 : (double)value * 1e-15; return 0; }
  return failmsg(h, std::string("pf_set_option: unknown option ") + name);
}
int pf_selftest_sqrt(pf_handle* h, int32_t n, const int64_t* d_in, double* d_out) {
  if (!h || n < 0 || !d_in || !d_out) return failmsg(h, "pf_selftest_sqrt: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_selftest_sqrt, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, (const long long*)d_in, d_out);
  CK(hipGetLastError()); CK(hipStreamSynchronize(h->stream));
  return 0;
}
int pf_selftest_rng(pf_handle* h, uint64_t seed, uint64_t dom, uint64_t it, uint64_t agent, uint64_t* d_u64,
                    double* d_f64, int64_t* d_i64) {
  if (!h || !d_u64 || !d_f64 || !d_i64) return failmsg(h, "pf_selftest_rng: bad arguments");
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_selftest_rng, dim3(1), dim3(64), 0, h->stream, seed, dom, it, agent, (unsigned long long*)d_u64, d_f64, (long long*)d_i64);
  CK(hipGetLastError()); CK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---------------------------------------------------------------------------
// MAACO host side
// ---------------------------------------------------------------------------
static double hdist(int r1, int c1, int r2, int c2) { long dr = r1 - r2, dc = c1 - c2; return sqrt((double)(dr * dr + dc * dc)); }

static int maaco_refresh_taua(pf_handle* h) {
  // alpha != 1: tau^alpha must be libm pow to match the reference (MAACO.py:238); refreshed on the host
  if (h->mp.alpha == 1.0) return 0;
  std::vector<double> t(h->RC);
  CK(hipMemcpyAsync(t.data(), h->d_tau, sizeof(double) * h->RC, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->RC; ++i) t[i] = pow(t[i], h->mp.alpha);
  CK(hipMemcpyAsync(h->d_taua, t.data(), sizeof(double) * h->RC, hipMemcpyHostToDevice, h->stream));
  CK(hipStreamSynchronize(h->stream));
  return 0;
}

int pf_maaco_setup(pf_handle* h, const pf_maaco_params* p) {
  if (!h) return -2;
  if (!p || p->start < 0 || p->start >= h->RC || p->target < 0 || p->target >= h->RC) return failmsg(h, "pf_maaco_setup: bad arguments");
  CK(hipSetDevice(h->device));
  h->mp = *p;
  const int R = h->R, C = h->C, RC = h->RC;
  const int sr = p->start / C, sc = p->start % C, tr = p->target / C, tc = p->target % C;
  double dsT = hdist(sr, sc, tr, tc); if (dsT < 1e-9) dsT = 1e-9;                   // MAACO.py:43-45
  std::vector<double> tau(RC), eta((size_t)RC * 2);
  for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) {
    const int i = r * C + c;
    // pheromone init MAACO.py:58-84
    if (h->h_occ[i] == 1) tau[i] = 1e-9;
    else {
      double dsi = hdist(sr, sc, r, c), diT = hdist(r, c, tr, tc), den = dsi + diT, factor;
      if (den < 1e-9) factor = (hdist(r, c, sr, sc) < 1e-6 || hdist(r, c, tr, tc) < 1e-6) ? 1.0 : 0.1;
      else factor = dsT / den;
      double t = factor * p->C0_initial_pheromone; if (t < 1e-9) t = 1e-9;
      tau[i] = t;
    }
    // eta'^beta for both turn flags, MAACO.py:197-210 + :238 (host libm: identical to math.exp / pow)
    double dsj = hdist(sr, sc, r, c), djT = hdist(r, c, tr, tc), hh;
    if (dsT < 1e-9) hh = p->wh_min; else hh = p->wh_max - (p->wh_max - p->wh_min) * exp(-p->k_h_adaptive * djT / dsT);
    double gg = 1.0 - hh;
    for (int turn = 0; turn < 2; ++turn) {
      double den = gg * dsj + hh * djT + p->a_turn_coef * (double)turn;
      den = den > 1e-9 ? den : 1e-9;
      eta[(size_t)i * 2 + turn] = pow(1.0 / den, p->beta);
    }
  }
  if (!h->d_tau) { CK(hipMalloc(&h->d_tau, sizeof(double) * RC)); CK(hipMalloc(&h->d_taua, sizeof(double) * RC)); CK(hipMalloc(&h->d_eta, sizeof(double) * RC * 2)); }
  CK(hipMemcpyAsync(h->d_tau, tau.data(), sizeof(double) * RC, hipMemcpyHostToDevice, h->stream));
  CK(hipMemcpyAsync(h->d_eta, eta.data(), sizeof(double) * RC * 2, hipMemcpyHostToDevice, h->stream));
  CK(hipStreamSynchronize(h->stream));
  if (!h->d_visit) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, h->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!h->nslots) h->nslots = cus * kSlotsPerCU;
    // one packed tabu set (Tabu, R * ceil(C / 16) words) per resident ant: up to 128 ants per CU (16 waves x 8 ants), within 32 GiB
    const size_t vwords = (size_t)h->R * (size_t)((h->C + 15) >> 4);
    int vs = cus * 128;
    while ((size_t)vs * vwords * sizeof(unsigned) > (32ull << 30) && vs > cus * 32) vs /= 2;
    h->maaco_slots = vs; h->maaco_cus = cus;
    CK(hipMalloc(&h->d_visit, sizeof(unsigned) * (size_t)vs * vwords));
    CK(hipMemsetAsync(h->d_visit, 0, sizeof(unsigned) * (size_t)vs * vwords, h->stream));
    CK(hipMalloc(&h->d_visit_epoch, sizeof(unsigned) * vs));
    CK(hipMemsetAsync(h->d_visit_epoch, 0, sizeof(unsigned) * vs, h->stream));
    CK(hipStreamSynchronize(h->stream));
  }
  CK(hipFuncSetAttribute((const void*)k_tau_update, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PF_UPD_CHUNK * (int)sizeof(double)));
  h->maaco_ready = true;
  if (h->d_best_row) CK(hipMemsetAsync(h->d_best_row, 0, sizeof(int), h->stream));   // a new colony has no best path yet
  return maaco_refresh_taua(h);
}

static double maaco_q0(int it, int K, double q0_initial) {          // MAACO.py:212-226
  double Kt = K, k = it, k0 = 0.7 * Kt, q;
  if (k < k0) q = (fabs(Kt - k0) < 1e-6) ? q0_initial : ((Kt - k) / Kt) * q0_initial;
  else {
    double q_at = ((Kt - k0) / Kt) * q0_initial;
    q = q_at + ((k - k0) / (Kt - k0 + 1e-9)) * (q0_initial * (1 - (Kt - k0) / Kt) / 2.0);
  }
  return fmin(fmax(q, 0.01), 0.99);
}

// room for the deposit bit matrix / the per-ant deposits of a batch of n ants; the matrix is (re)zeroed unless the last
// deposit pass left it so
static int maaco_ensure_bits(pf_handle* h, int n) {
  const size_t words = (size_t)(n + 63) / 64;
  if (words > h->bits_words) {
    if (h->d_bits) CK(hipFree(h->d_bits));
    CK(hipMalloc(&h->d_bits, words * (size_t)((h->RC + 63) / 64) * 64 * sizeof(unsigned long long)));   // (bits_idx: whole stretches)
    if (h->d_flag) CK(hipFree(h->d_flag));
    CK(hipMalloc(&h->d_flag, words * (size_t)((h->RC + 63) / 64)));
    h->bits_words = words;
    h->bits_clean = false;
  }
  if ((int)(words * 64) > h->dep_cap) {
    if (h->d_dep) CK(hipFree(h->d_dep));
    CK(hipMalloc(&h->d_dep, sizeof(double) * (size_t)words * 64)); h->dep_cap = (int)(words * 64);
    CK(hipMemsetAsync(h->d_dep, 0, sizeof(double) * words * 64, h->stream));
  }
  if (!h->bits_clean)      // (the deposit kernels hand the matrix back zeroed; a fresh buffer, a failed or an abandoned batch does not)
  {
    CK(hipMemsetAsync(h->d_bits, 0, h->bits_words * (size_t)((h->RC + 63) / 64) * 64 * sizeof(unsigned long long), h->stream));
    CK(hipMemsetAsync(h->d_flag, 0, h->bits_words * (size_t)((h->RC + 63) / 64), h->stream));
  }
  h->bits_clean = false;
  // (d_dep needs no clearing per batch: the walk kernels / k_visit_bits write the entry of EVERY ant of the batch, 0.0 for a failed
  // one, and an entry beyond the batch is never read -- its bits are zero)
  return 0;
}
// enqueue the walk of ants [ant0, ant0 + n) (nothing waits); the ants mark their own deposits when g_maaco_mark is on
static int maaco_enqueue_walk(pf_handle* h, int32_t iter, uint64_t seed, int32_t ant0, int32_t n, int32_t path_cap,
                              int32_t* d_cells, int32_t* d_len, double* d_plen, int32_t* d_turns, int32_t* d_status, bool mark,
                              bool own_ctl = false) {
  CK(hipSetDevice(h->device));
  MaacoArgs a;
  a.G = make_grid(h, 1, 1);
  a.tau = h->mp.alpha == 1.0 ? h->d_tau : h->d_taua; a.eta = h->d_eta;
  if (!h->d_tep) CK(hipMalloc(&h->d_tep, sizeof(double) * 3 * (size_t)h->RC));
  a.tep = h->d_tep;
  a.visit = h->d_visit; a.slot_epoch = h->d_visit_epoch; a.work = h->d_work; a.cnt = h->d_cnt;
  a.wpr = (h->C + 15) >> 4; a.vstride = h->R * a.wpr;
  a.start = h->mp.start; a.target = h->mp.target; a.iter = iter; a.num_iterations = h->mp.num_iterations;
  a.q0 = maaco_q0(iter, h->mp.num_iterations, h->mp.q0_initial);
  a.seed = seed; a.ant0 = ant0; a.n = n; a.path_cap = path_cap;
  a.cells = d_cells; a.len = d_len; a.plen = d_plen; a.turns = d_turns; a.status = d_status;
  a.bits = nullptr; a.dep = nullptr; a.flag = nullptr; a.fstride = 0; a.Q = h->mp.Q;
  h->marks_n = 0; h->marks_cells = nullptr;
  if (mark) {
    if (maaco_ensure_bits(h, n)) return -1;
    a.bits = h->d_bits; a.dep = h->d_dep; a.flag = h->d_flag; a.fstride = (int)h->bits_words;
    h->marks_n = n; h->marks_cells = d_cells;                       // deposit_begin for exactly this batch finds its marks made
  }
  // eight ants per wavefront (k_maaco_walk8) once the batch can fill the chip that way; else one per wave
  const bool pack8 = n >= g_maaco_pack8_min;
  a.groups = g_maaco_groups;
  int grid = pack8 ? h->maaco_slots / 8 : (h->maaco_slots < 8192 ? h->maaco_slots : 8192);
  const int need = pack8 ? (n + a.groups - 1) / a.groups : n; if (grid > need) grid = need;
  if (own_ctl) {                                                    // pf_maaco_iterate: the previous iteration's last reader left the block zeroed
    if (!h->d_mctl) { CK(hipMalloc(&h->d_mctl, 16 + sizeof(DevCounters))); h->mctl_clean = false; }
    if (!h->mctl_clean) CK(hipMemsetAsync(h->d_mctl, 0, 16 + sizeof(DevCounters), h->stream));
    h->mctl_clean = false;
    a.work = (int*)h->d_mctl; a.cnt = (DevCounters*)(h->d_mctl + 16);
  } else {
    CK(hipMemsetAsync(h->d_work, 0, sizeof(int), h->stream));
    CK(hipMemsetAsync(h->d_cnt, 0, sizeof(DevCounters), h->stream));
  }
  if (g_tabu_epoch >= 0) {                                          // one-shot: later batches carry on from there
    CK(hipMemsetD32Async((hipDeviceptr_t)h->d_visit_epoch, g_tabu_epoch, (size_t)h->maaco_slots, h->stream));
    g_tabu_epoch = -1;
  }
  if (pack8) hipLaunchKernelGGL(k_pack_tep, dim3((h->RC + 255) / 256), dim3(256), 0, h->stream, h->RC, a.tau, a.eta, h->d_tep);
  CK(hipEventRecord(h->ev0, h->stream));
  // (the load-ahead form when the batch leaves every SIMD at most one wavefront: see k_maaco_walk8)
  const bool ahead = g_maaco_ahead < 0 ? grid <= h->maaco_cus * 4 : g_maaco_ahead != 0;
  if (pack8 && ahead) hipLaunchKernelGGL(k_maaco_walk8<true>, dim3(grid), dim3(64), 0, h->stream, a);
  else if (pack8) hipLaunchKernelGGL(k_maaco_walk8<false>, dim3(grid), dim3(64), 0, h->stream, a);
  else hipLaunchKernelGGL(k_maaco_walk, dim3(grid), dim3(64), 0, h->stream, a);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev1, h->stream));
  return 0;
}
int pf_maaco_walk_batch(pf_handle* h, int32_t iter, uint64_t seed, int32_t ant0, int32_t n, int32_t path_cap,
                        int32_t* d_cells, int32_t* d_len, double* d_plen, int32_t* d_turns, int32_t* d_status) {
  if (!h) return -2;
  if (!h->maaco_ready) return failmsg(h, "pf_maaco_walk_batch: call pf_maaco_setup first");
  if (n < 0 || path_cap < 2 || !d_cells || !d_len || !d_plen || !d_turns || !d_status) return failmsg(h, "pf_maaco_walk_batch: bad arguments");
  if (n == 0) { memset(&h->last, 0, sizeof(h->last)); h->marks_n = 0; return 0; }   // an empty batch has empty counters (not the previous batch's overflow)
  if (maaco_enqueue_walk(h, iter, seed, ant0, n, path_cap, d_cells, d_len, d_plen, d_turns, d_status, g_maaco_mark != 0)) return -1;
  DevCounters dc; if (end_batch(h, &dc)) return -1;
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return 0;
}

// One whole iteration of MAACO.solve_path_planning (MAACO.py:340-359) for ants [ant0, ant0 + n) of ONE GPU, enqueued back to
// back: the walks (each successful ant marks its own deposits), the best-of-iteration scan (:343-349), the take-over test
// against the caller's overall best (:351-358), and the pheromone update in one pass (:304-332).  ONE small copy comes back:
// out13 = {ib_len, ib_turns, ib_idx, took, best_len, best_turns, tmin, tmax, skipped, steps, candidates, path_cells,
// overflow_agents}.  If an ant overflowed its path row (overflow_agents > 0) the pheromone is left untouched and the
// caller repeats the call with longer rows (same iteration, same streams).
int pf_maaco_iterate(pf_handle* h, int32_t iter, uint64_t seed, int32_t ant0, int32_t n, int32_t path_cap,
                     int32_t* d_cells, int32_t* d_len, double* d_plen, int32_t* d_turns, int32_t* d_status,
                     double best_len, double best_turns, double* out13) {
  if (!h) return -2;
  if (!h->maaco_ready) return failmsg(h, "pf_maaco_iterate: call pf_maaco_setup first");
  if (n <= 0 || path_cap < 2 || !d_cells || !d_len || !d_plen || !d_turns || !d_status || !out13) return failmsg(h, "pf_maaco_iterate: bad arguments");
  if (!h->d_scan3) CK(hipMalloc(&h->d_scan3, 24));
  if (!h->d_mstate) CK(hipMalloc(&h->d_mstate, 16 * sizeof(double)));
  if (!h->h_mstate) {
    CK(hipHostMalloc((void**)&h->h_mstate, 16 * sizeof(double), hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&h->h_mstate_dev, h->h_mstate, 0));
  }
  double* hs_dev = h->h_mstate_dev;
  if (h->best_row_cap < path_cap + 1) {                             // (a longer row keeps the best so far: same stream, ordered)
    int* nb = nullptr; CK(hipMalloc(&nb, sizeof(int) * ((size_t)path_cap + 1)));
    if (h->d_best_row) { CK(hipMemcpyAsync(nb, h->d_best_row, sizeof(int) * (size_t)h->best_row_cap, hipMemcpyDeviceToDevice, h->stream)); CK(hipStreamSynchronize(h->stream)); CK(hipFree(h->d_best_row)); }
    else CK(hipMemsetAsync(nb, 0, sizeof(int), h->stream));
    h->d_best_row = nb; h->best_row_cap = path_cap + 1;
  }
  if (maaco_enqueue_walk(h, iter, seed, ant0, n, path_cap, d_cells, d_len, d_plen, d_turns, d_status, true, true)) return -1;   // (this path always marks)
  hipLaunchKernelGGL(k_maaco_best_take, dim3(1), dim3(1024), 0, h->stream, n, (const double*)d_plen, (const int*)d_turns, (double*)h->d_scan3,
                     best_len, best_turns, h->mp.rho, h->R, h->C, (int*)h->d_mctl, (DevCounters*)(h->d_mctl + 16), h->d_mstate, hs_dev,
                     (const int*)d_cells, (const int*)d_len, path_cap, h->d_best_row);
  CK(hipEventRecord(h->ev2, h->stream));
  const int words = (n + 63) / 64;
  hipLaunchKernelGGL(k_tau_update, dim3((h->RC + 1023) / 1024), dim3(1024), 2 * PF_UPD_CHUNK * sizeof(double), h->stream, h->d_tau, h->d_occ,
                     h->RC, h->d_bits, words, h->d_dep, 1.0 - h->mp.rho, (const double*)h->d_mstate, 0.0, 0.0, h->d_flag, (int)h->bits_words);
  CK(hipGetLastError());
  // the host needs the 13 doubles, not the pheromone: it waits for the take-over test only, the update pass runs on behind the
  // caller's bookkeeping (everything later on this stream is ordered after it)
  CK(hipEventSynchronize(h->ev2));
  memcpy(out13, h->h_mstate, 13 * sizeof(double));
  h->mctl_clean = true;
  h->d2h_small += 1;
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  memset(&h->last, 0, sizeof(h->last));
  h->last.steps = (unsigned long long)out13[9]; h->last.candidates = (unsigned long long)out13[10];
  h->last.path_cells = (unsigned long long)out13[11]; h->last.overflow_agents = (unsigned long long)out13[12];
  h->marks_n = 0;
  h->bits_clean = out13[8] == 0.0;                                  // the update pass read and zeroed every word (a skipped one did not)
  if (out13[8] == 0.0) return maaco_refresh_taua(h);                // (alpha == 1: nothing to do, nothing waits)
  return 0;
}

// The overall best ant's path as pf_maaco_iterate keeps it in HBM: *len_out cells (0: none yet) into cells_out[cap].
int pf_maaco_best_path(pf_handle* h, int32_t* cells_out, int32_t cap, int32_t* len_out) {
  if (!h) return -2;
  if (!len_out || cap < 0 || (cap > 0 && !cells_out)) return failmsg(h, "pf_maaco_best_path: bad arguments");
  *len_out = 0;
  if (!h->d_best_row) return 0;
  CK(hipSetDevice(h->device));
  int L = 0;
  if (pf_d2h(h, &L, h->d_best_row, sizeof(int))) return -1;
  if (L < 0 || L > h->best_row_cap - 1) return failmsg(h, "pf_maaco_best_path: corrupt row");
  *len_out = L;
  if (L > cap) return failmsg(h, "pf_maaco_best_path: the buffer is too small");
  if (L > 0 && pf_d2h(h, cells_out, h->d_best_row + 1, (int64_t)sizeof(int) * L)) return -1;
  return 0;
}

int pf_maaco_evaporate(pf_handle* h) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_evaporate: setup first");
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_tau_evaporate, dim3((h->RC + 255) / 256), dim3(256), 0, h->stream, h->d_tau, h->RC, 1.0 - h->mp.rho);
  CK(hipGetLastError());
  return 0;                                                        // stream ordered: nothing waits
}

// MAACO.py:306-311 in two steps so that the multi-GPU fold can work on row chunks: _begin marks which successful ant
// visited which cell (bit matrix) and computes Q / L per ant; _cells adds the marked deposits, in ant order, to the
// pheromone of cells [cell0, cell1) (every cell exactly once per update).
int pf_maaco_deposit_begin(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                           const double* d_plen) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_deposit_begin: setup first");
  h->dep_words = 0;
  if (n <= 0) return 0;
  CK(hipSetDevice(h->device));
  const size_t words = (size_t)(n + 63) / 64;
  if (h->marks_n == n && h->marks_cells == d_cells) {
    // the walk batch that produced exactly these paths has marked them already (pf_set_option "maaco_mark_in_walk")
    h->marks_n = 0;
  } else {
    h->marks_n = 0; h->marks_cells = nullptr;                       // whatever an earlier walk marked is wiped / replaced below: never trusted again
    if (maaco_ensure_bits(h, n)) return -1;
    hipLaunchKernelGGL(k_visit_bits, dim3(n), dim3(64), 0, h->stream, n, path_cap, d_cells, d_len, d_plen, h->mp.Q, h->d_bits, h->RC, h->d_dep, h->d_flag, (int)h->bits_words);
    CK(hipGetLastError());
  }
  h->dep_words = (int)words; h->dep_done = 0;
  return 0;
}
// MAACO.py:304-332 in one pass (evaporate, ordered deposits, clip) for the paths of one batch: the single-GPU form of
// pf_maaco_evaporate + pf_maaco_deposit + pf_maaco_clip (the sharded fold keeps those: rank 0 alone evaporates).
int pf_maaco_update(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len, const double* d_plen,
                    double best_len_overall) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_update: setup first");
  if (n < 0 || (n > 0 && (!d_cells || !d_len || !d_plen))) return failmsg(h, "pf_maaco_update: bad arguments");
  if (pf_maaco_deposit_begin(h, n, path_cap, d_cells, d_len, d_plen)) return -1;
  double bl = best_len_overall;                                     // MAACO.py:312-316
  if (bl == INFINITY) bl = (double)(h->R + h->C);
  if (bl < 1e-6) bl = 1e-6;
  const double tmax = (1.0 / (1.0 - h->mp.rho)) * (1.0 / bl);       // :317
  int mx = h->C > h->R ? h->C : h->R; if (mx < 1) mx = 1;
  const double tmin = tmax / (2.0 * mx);                            // :323
  hipLaunchKernelGGL(k_tau_update, dim3((h->RC + 1023) / 1024), dim3(1024), 2 * PF_UPD_CHUNK * sizeof(double), h->stream, h->d_tau, h->d_occ,
                     h->RC, h->d_bits, h->dep_words, h->d_dep, 1.0 - h->mp.rho, (const double*)nullptr, tmin, tmax, h->d_flag, (int)h->bits_words);
  CK(hipGetLastError());
  if (h->dep_words) h->bits_clean = true;
  h->dep_words = 0;
  return maaco_refresh_taua(h);
}
int pf_maaco_deposit_cells(pf_handle* h, int32_t cell0, int32_t cell1) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_deposit_cells: setup first");
  if (cell0 < 0 || cell1 > h->RC || cell0 > cell1) return failmsg(h, "pf_maaco_deposit_cells: bad cell range");
  if (h->dep_words == 0 || cell0 == cell1) return 0;
  CK(hipSetDevice(h->device));
  CK(hipFuncSetAttribute((const void*)k_tau_deposit, hipFuncAttributeMaxDynamicSharedMemorySize, PF_DEP_CHUNK * (int)sizeof(double)));
  hipLaunchKernelGGL(k_tau_deposit, dim3((cell1 - cell0 + 1023) / 1024), dim3(1024), PF_DEP_CHUNK * sizeof(double), h->stream, h->d_tau, h->d_occ,
                     h->RC, h->d_bits, h->dep_words, h->d_dep, cell0, cell1, (int)h->bits_words);
  CK(hipGetLastError());
  h->dep_done += cell1 - cell0;
  if (h->dep_done >= h->RC) h->bits_clean = true;                   // every word has been read and zeroed again
  return 0;
}
int pf_maaco_deposit(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                     const double* d_plen) {
  if (pf_maaco_deposit_begin(h, n, path_cap, d_cells, d_len, d_plen)) return -1;
  return pf_maaco_deposit_cells(h, 0, h->RC);
}
// MAACO.py:343-349 over device columns: -> out3 = {best_len, best_turns, best_idx} (one 24-byte D2H)
int pf_maaco_best_dev(pf_handle* h, int32_t n, const double* d_plen, const int32_t* d_turns, double* out3) {
  if (!h) return -2;
  if (n < 0 || !d_plen || !d_turns || !out3) return failmsg(h, "pf_maaco_best_dev: bad arguments");
  CK(hipSetDevice(h->device));
  if (!h->d_scan3) CK(hipMalloc(&h->d_scan3, 24));
  hipLaunchKernelGGL(k_maaco_best, dim3(1), dim3(1024), 0, h->stream, n, d_plen, d_turns, (double*)h->d_scan3);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(out3, h->d_scan3, 24, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  h->d2h_small += 1;
  return 0;
}

int pf_maaco_clip(pf_handle* h, double best_len_overall) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_clip: setup first");
  CK(hipSetDevice(h->device));
  double bl = best_len_overall;                                     // MAACO.py:312-316
  if (bl == INFINITY) bl = (double)(h->R + h->C);
  if (bl < 1e-6) bl = 1e-6;
  const double tmax = (1.0 / (1.0 - h->mp.rho)) * (1.0 / bl);       // :317
  int mx = h->C > h->R ? h->C : h->R; if (mx < 1) mx = 1;
  const double tmin = tmax / (2.0 * mx);                            // :323
  hipLaunchKernelGGL(k_tau_clip, dim3((h->RC + 255) / 256), dim3(256), 0, h->stream, h->d_tau, h->d_occ, h->RC, tmin, tmax);
  CK(hipGetLastError());
  return maaco_refresh_taua(h);                                    // (alpha == 1: nothing to do, nothing waits)
}

int pf_maaco_get_pheromone(pf_handle* h, double* tau) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_get_pheromone: setup first");
  return pf_d2h(h, tau, h->d_tau, (int64_t)sizeof(double) * h->RC);
}
int pf_maaco_set_pheromone(pf_handle* h, const double* tau) {
  if (!h || !h->maaco_ready) return failmsg(h, "pf_maaco_set_pheromone: setup first");
  if (pf_h2d(h, h->d_tau, tau, (int64_t)sizeof(double) * h->RC)) return -1;
  return maaco_refresh_taua(h);
}
void* pf_maaco_tau_dev(pf_handle* h) { return h ? (void*)h->d_tau : nullptr; }

int pf_maaco_best_scan(int32_t n, const double* plen, const int32_t* turns, int32_t idx0, double* best_len,
                       double* best_turns, int32_t* best_idx) {
  // MAACO.py:343-349, sequential over ants (turns of a failed ant are +inf there)
  for (int i = 0; i < n; ++i) {
    const double L = plen[i];
    const double T = turns[i] < 0 ? INFINITY : (double)turns[i];
    if (L < *best_len) { *best_len = L; *best_idx = idx0 + i; *best_turns = T; }
    else if (fabs(L - *best_len) < 1e-9 && T < *best_turns) { *best_idx = idx0 + i; *best_turns = T; }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// GA host side: the genetic operators (ga_solver.py:48-53, 136-160, 186-194) in native code.  Draw-count dependent and
// tiny, so they stay on the host as in the reference -- but not in Python: the same keyed streams
// (pathfit/rng.py) and the same CPython 3.10 derivations (random.sample / randint / _randbelow) restated in C.
// ---------------------------------------------------------------------------
namespace {
struct HostRng {                               // pathfit/rng.py AgentRandom
  uint64_t key, ctr;
  static uint64_t mix(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }
  HostRng(uint64_t seed, uint64_t dom, uint64_t it, uint64_t agent) : ctr(0) {
    uint64_t k = mix(seed + 0x9E3779B97F4A7C15ull * (dom + 1));
    k = mix(k + 0xD1B54A32D192ED03ull * (it + 1));
    key = mix(k + 0x8CB92BA72F3D8DD7ull * (agent + 1));
  }
  uint64_t next64() { ctr += 1; return mix(key + ctr * 0x9E3779B97F4A7C15ull); }
  double random() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t randbelow(uint64_t n) {             // random.py _randbelow_with_getrandbits
    int k = 0; for (uint64_t t = n; t; t >>= 1) ++k;
    uint64_t r = next64() >> (64 - k);
    while (r >= n) r = next64() >> (64 - k);
    return r;
  }
  int64_t randint(int64_t a, int64_t b) { return a + (int64_t)randbelow((uint64_t)(b - a + 1)); }
  double uniform(double a, double b) { return a + (b - a) * random(); }
  double normalvariate(double mu, double sigma) {   // random.py normalvariate, glibc log as math.log
    double z;
    for (;;) {
      const double u1 = random(), u2 = 1.0 - random();
      z = 1.7155277699214135 * (u1 - 0.5) / u2;
      const double zz = z * z / 4.0;
      if (zz <= -log(u2)) break;
    }
    return mu + z * sigma;
  }
};
}  // namespace

extern "C" {
int pf_ga_select(uint64_t seed, int32_t gen, int32_t n, int32_t tournament_size, const double* fitness, int32_t* parent_idx) {
  if (n <= 0 || tournament_size <= 0 || !fitness || !parent_idx) return -1;
  HostRng r(seed, 7 /* DOM_GA_SELECT */, (uint64_t)gen, 0);
  const int k = tournament_size < n ? tournament_size : n;          // ga_solver.py:139 min(tournament_size, len(population))
  int64_t setsize = 21;                                              // random.sample: pool copy for small n, rejection set otherwise
  if (k > 5) { int64_t p4 = 1; while (p4 < (int64_t)k * 3) p4 *= 4; setsize += p4; }
  std::vector<int32_t> pool, sel((size_t)k);
  for (int s = 0; s < n; ++s) {
    if (n <= setsize) {
      pool.resize((size_t)n);
      for (int i = 0; i < n; ++i) pool[i] = i;
      for (int i = 0; i < k; ++i) { const uint64_t j = r.randbelow((uint64_t)(n - i)); sel[i] = pool[j]; pool[j] = pool[n - i - 1]; }
    } else {
      for (int i = 0; i < k; ++i) {
        int32_t j;
        for (;;) { j = (int32_t)r.randbelow((uint64_t)n); bool seen = false; for (int q = 0; q < i; ++q) seen |= sel[q] == j; if (!seen) break; }
        sel[i] = j;
      }
    }
    int32_t best = sel[0];                                           // min(tournament, key=fitness): the first minimum
    for (int i = 1; i < k; ++i) if (fitness[sel[i]] < fitness[best]) best = sel[i];
    parent_idx[s] = best;
  }
  return 0;
}

int pf_ga_random_chromosomes(uint64_t seed, int32_t attempt0, int32_t n, int32_t W, const uint8_t* occ, int32_t R, int32_t Cc,
                             int32_t* cells) {
  if (n <= 0 || W <= 0 || !occ || R <= 0 || Cc <= 0 || !cells) return -1;
  for (int a = 0; a < n; ++a) {                                      // ga_solver.py:55-56 per attempt, stream (seed, DOM_INIT, 0, attempt)
    HostRng r(seed, 6 /* DOM_INIT */, 0, (uint64_t)(attempt0 + a));
    for (int i = 0; i < W; ++i)
      for (;;) {                                                     // :48-53
        const int rr = (int)r.randint(0, R - 1), cc = (int)r.randint(0, Cc - 1);
        if (occ[(size_t)rr * Cc + cc] != 1) { cells[(size_t)a * W + i] = rr * Cc + cc; break; }
      }
  }
  return 0;
}

int pf_ga_breed(uint64_t seed, int32_t gen, int32_t n, int32_t W, double crossover_rate, double mutation_rate, const uint8_t* occ,
                int32_t R, int32_t Cc, const int32_t* parent_cells, int32_t* child_cells) {
  if (n <= 0 || W <= 0 || !occ || R <= 0 || Cc <= 0 || !parent_cells || !child_cells) return -1;
  std::vector<int32_t> c1((size_t)W), c2((size_t)W);
  int made = 0;
  for (int pair = 0, idx = 0; made < n; ++pair, idx += 2) {          // ga_solver.py:186-194
    const int32_t* p1 = parent_cells + (size_t)(idx % n) * W;
    const int32_t* p2 = parent_cells + (size_t)((idx + 1) % n) * W;
    HostRng r(seed, 4 /* DOM_GA */, (uint64_t)gen, (uint64_t)pair);
    int point = 0;
    if (r.random() < crossover_rate) point = W > 1 ? (int)r.randint(1, W - 1) : 0;   // :145-148
    for (int i = 0; i < W; ++i) { const bool tail = point > 0 && i >= point; c1[i] = tail ? p2[i] : p1[i]; c2[i] = tail ? p1[i] : p2[i]; }
    for (int which = 0; which < 2; ++which) {                        // :154-160, child 1 then child 2 on the same stream
      std::vector<int32_t>& c = which ? c2 : c1;
      for (int i = 0; i < W; ++i)
        if (r.random() < mutation_rate)
          for (;;) {                                                 // :48-53 rejection-sample a free cell
            const int rr = (int)r.randint(0, R - 1), cc = (int)r.randint(0, Cc - 1);
            if (occ[(size_t)rr * Cc + cc] != 1) { c[i] = rr * Cc + cc; break; }
          }
      if (made < n) { for (int i = 0; i < W; ++i) child_cells[(size_t)made * W + i] = c[i]; ++made; }
    }
  }
  return 0;
}
}  // extern "C"

// ---------------------------------------------------------------------------
// MPA host side
// ---------------------------------------------------------------------------
// Exact shortest-path lengths from the start and to the target on the static grid (Dijkstra on the host over the
// device-built move masks; the graph is symmetric).  They are admissible bounds for the MPA pruning tests: every
// avoid set only removes cells.  Skipped (no pruning) for grids above 4 Mi cells or penalty weights below zero.
static int dijkstra_host(const pf_handle* h, const std::vector<uint8_t>& mm, int src, std::vector<double>& dist) {
  static const int DR[8] = {0, 0, 1, -1, 1, 1, -1, -1}, DC[8] = {1, -1, 0, 0, 1, -1, 1, -1};   // helper.py:38-52 order
  const int C = h->C;
  dist.assign(h->RC, __builtin_huge_val());
  if (h->h_occ[src] == 1) return 0;
  typedef std::pair<double, int> QE;
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
  dist[src] = 0.0; pq.push(QE(0.0, src));
  while (!pq.empty()) {
    const QE e = pq.top(); pq.pop();
    if (e.first > dist[e.second]) continue;
    const unsigned m = mm[e.second];
    for (int k = 0; k < 8; ++k) if ((m >> k) & 1u) {
      const int n = e.second + DR[k] * C + DC[k];
      const double t = e.first + (k < 4 ? 1.0 : PF_SQRT2);
      if (t < dist[n]) { dist[n] = t; pq.push(QE(t, n)); }
    }
  }
  return 0;
}
static int mpa_bounds(pf_handle* h) {
  if (h->d_ds) { (void)hipFree(h->d_ds); h->d_ds = nullptr; }
  if (h->d_dt) { (void)hipFree(h->d_dt); h->d_dt = nullptr; }
  if (h->RC > (1 << 22) || h->mps.w_turn < 0.0 || h->mps.w_safe < 0.0 || h->mps.diag_pen < 0.0) return 0;
  const Grid G = make_grid(h, h->mpp.allow_diag, h->mpp.restrict_corner);
  std::vector<uint8_t> mm(h->RC);
  CK(hipMemcpy(mm.data(), G.mm, h->RC, hipMemcpyDeviceToHost));
  std::vector<double> ds, dt;
  dijkstra_host(h, mm, h->mpp.start, ds); dijkstra_host(h, mm, h->mpp.target, dt);
  CK(hipMalloc(&h->d_ds, sizeof(double) * (size_t)h->RC)); CK(hipMalloc(&h->d_dt, sizeof(double) * (size_t)h->RC));
  CK(hipMemcpy(h->d_ds, ds.data(), sizeof(double) * (size_t)h->RC, hipMemcpyHostToDevice));
  CK(hipMemcpy(h->d_dt, dt.data(), sizeof(double) * (size_t)h->RC, hipMemcpyHostToDevice));
  return 0;
}
int pf_mpa_setup(pf_handle* h, const pf_mpa_params* p, const pf_score_params* sp) {
  if (!h) return -2;
  if (!p || !sp || p->start < 0 || p->start >= h->RC || p->target < 0 || p->target >= h->RC) return failmsg(h, "pf_mpa_setup: bad arguments");
  h->mpp = *p; h->mps = *sp; h->mpa_ready = true;
  if (ensure_slots(h, p->allow_diag, p->restrict_corner)) return -1;
  // memoise MPA._generate_initial_path() = _a_star(start, target) (MPA.py:154) and its stats
  const int cap = h->RC;
  if (h->init_cap < cap) {
    if (h->d_init_cells) CK(hipFree(h->d_init_cells));
    CK(hipMalloc(&h->d_init_cells, sizeof(int) * (size_t)cap)); h->init_cap = cap;
  }
  if (!h->d_init_stats) CK(hipMalloc(&h->d_init_stats, sizeof(double) * 5));
  int *d_s = nullptr, *d_t = nullptr, *d_l = nullptr, *d_st = nullptr;
  CK(hipMalloc(&d_s, 4 * sizeof(int))); d_t = d_s + 1; d_l = d_s + 2; d_st = d_s + 3;
  int hv[4] = {p->start, p->target, 0, 0};
  CK(hipMemcpyAsync(d_s, hv, sizeof(hv), hipMemcpyHostToDevice, h->stream));
  int rc = pf_astar_batch(h, PF_ASTAR_MPA, p->allow_diag, p->restrict_corner, 1, d_s, d_t, nullptr, nullptr, cap, h->d_init_cells, d_l, d_st, nullptr);
  if (rc == 0) {
    CK(hipMemcpyAsync(hv, d_s, sizeof(hv), hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    h->init_len = hv[3] == 0 ? hv[2] : 0;
    rc = pf_score_batch(h, sp, 1, cap, h->d_init_cells, d_l, h->d_init_stats);
  }
  (void)hipFree(d_s);
  if (rc == 0) rc = mpa_bounds(h);
  return rc;
}
static MpaDev mpa_dev(const pf_handle* h) {
  MpaDev m; m.P = h->mpp.P_const; m.levy_beta = h->mpp.levy_beta; m.sigma = h->mpp.levy_sigma; m.fads = h->mpp.FADs_rate;
  m.N = h->mpp.num_predators; m.start = h->mpp.start; m.target = h->mpp.target;
  const bool on = g_mpa_prune && h->d_ds && h->d_dt;
  m.ds = on ? h->d_ds : nullptr; m.dt = on ? h->d_dt : nullptr;
  return m;
}

// ---- proposals: k_mpa_propose, then the host confirms the (expected: zero) doubtful ones with glibc ----
static int mpa_launch_propose(pf_handle* h, MpaPhaseArgs& a, float* est) {
  const int n = a.n;
  if (n > h->prop_cap) {
    if (h->d_prop) CK(hipFree(h->d_prop));
    if (h->d_doubt) CK(hipFree(h->d_doubt));
    CK(hipMalloc(&h->d_prop, sizeof(int2) * (size_t)n)); CK(hipMalloc(&h->d_doubt, sizeof(int) * ((size_t)n + 1)));
    h->prop_cap = n;
  }
  a.prop = h->d_prop; a.doubt_n = h->d_doubt; a.doubt_list = h->d_doubt + 1; a.eps_log = g_doubt_log; a.eps_round = g_doubt_round;
  CK(hipMemsetAsync(h->d_doubt, 0, sizeof(int), h->stream));
  if (n > 0) hipLaunchKernelGGL(k_mpa_propose, dim3((n + 255) / 256), dim3(256), 0, h->stream, a, est);
  CK(hipGetLastError());
  return 0;
}
static int d2h_bytes(pf_handle* h, const void* d, void* out, size_t nb) { CK(hipMemcpyAsync(out, d, nb, hipMemcpyDeviceToHost, h->stream)); CK(hipStreamSynchronize(h->stream)); return 0; }
#define d2h_one(h, d, out) d2h_bytes(h, (const void*)(d), (void*)(out), sizeof(*(out)))
static long host_round(double x) { return (long)nearbyint(x); }     // Python round(): half to even
static int host_clamp(long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }
// MPA.py:250-282 on the host (glibc log / pow / sin / cos: what CPython's math module calls)
static int host_levy(HostRng& g, int R, int C, int cur, double scale, double beta, double sigma) {
  const double u = g.normalvariate(0.0, sigma);
  double v = g.normalvariate(0.0, 1.0);
  if (fabs(v) < 1e-9) v = 1e-9;
  double step = 0.05 * u / pow(fabs(v), 1.0 / beta) * scale;
  const double mx = (double)(R > C ? R : C) * 0.5;
  step = fmin(fmax(step, -mx), mx);
  const double ang = g.uniform(0.0, 2.0 * 3.141592653589793);
  const long dr = host_round(step * sin(ang)), dc = host_round(step * cos(ang));
  return host_clamp(cur / C + dr, 0, R - 1) * C + host_clamp(cur % C + dc, 0, C - 1);
}
static int host_brownian(HostRng& g, int R, int C, int cur, int elite, double scale) {
  const int cr = cur / C, cc = cur % C;
  long tr_, tc_;
  if (g.random() < 0.7 && elite >= 0) {
    const int dr = elite / C - cr, dc = elite % C - cc;
    const double dist = sqrt((double)((long)dr * dr + (long)dc * dc));
    if (dist > 1e-6) {
      const double fac = fabs(g.normalvariate(0.0, 1.0));
      long kk = host_round(scale * fac * 5.0); if (kk < 1) kk = 1;
      const double ms = dist < (double)kk ? dist : (double)kk;
      tr_ = cr + host_round((double)dr / dist * ms);
      tc_ = cc + host_round((double)dc / dist * ms);
    } else return elite;
  } else {
    long m = host_round((double)(R > C ? R : C) * 0.1 * scale * fabs(g.normalvariate(0.0, 1.0)));
    if (m < 1) m = 1;
    const long dr = g.randint(-m, m);
    const long dc = g.randint(-m, m);
    tr_ = cr + dr; tc_ = cc + dc;
  }
  return host_clamp(tr_, 0, R - 1) * C + host_clamp(tc_, 0, C - 1);
}
// Recompute the proposals the device flagged (a decision within the margin of a libm disagreement) with the host's
// libm and patch them in.  One small D2H (the count) per sweep; the rest only when the count is not zero.
static int mpa_resolve_doubts(pf_handle* h, const MpaPhaseArgs& a) {
  int nd = 0;
  if (d2h_one(h, a.doubt_n, &nd)) return -1;
  h->d2h_small += 1;
  if (nd <= 0) return 0;
  std::vector<int> list((size_t)nd);
  CK(hipMemcpyAsync(list.data(), a.doubt_list, sizeof(int) * (size_t)nd, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  const int R = h->R, C = h->C;
  int elite_len = a.elite_len;
  if (a.elite_len_dev && d2h_one(h, a.elite_len_dev, &elite_len)) return -1;
  for (int a_ : list) {
    int gi = 0, slot = a_, preyL = 0;
    if (a.ex_idx) { if (d2h_one(h, a.ex_agent + a_, &gi)) return -1; }
    else { if (d2h_one(h, a.gidx + a_, &gi) || d2h_one(h, a.slot + a_, &slot)) return -1; }
    if (d2h_one(h, a.pop_len + slot, &preyL)) return -1;
    const int* prey = a.pop_cells + (size_t)slot * a.path_cap;
    bool is_levy; double scale; const int* mod; int modL; const int* ref; int refL;
    if (a.ex_idx) {
      int lv = 0; double sc = 0.0;
      if (d2h_one(h, a.ex_levy + a_, &lv) || d2h_one(h, a.ex_scale + a_, &sc)) return -1;
      is_levy = lv != 0; scale = sc; mod = prey; modL = preyL; ref = a.elite_cells; refL = elite_len;
    } else if (a.phase == 1) { is_levy = false; scale = a.m.P; mod = prey; modL = preyL; ref = a.elite_cells; refL = elite_len; }
    else if (a.phase == 2) {
      is_levy = gi < a.m.N / 2; scale = is_levy ? a.m.P : a.m.P * a.CF;
      mod = is_levy ? prey : a.elite_cells; modL = is_levy ? preyL : elite_len;
      ref = is_levy ? a.elite_cells : prey; refL = is_levy ? elite_len : preyL;
    } else { is_levy = true; scale = a.m.P * a.CF; mod = a.elite_cells; modL = elite_len; ref = prey; refL = preyL; }
    HostRng g(a.seed, DOM_MPA, (uint64_t)a.iter, (uint64_t)gi);
    int idx;
    if (a.ex_idx) { if (d2h_one(h, a.ex_idx + a_, &idx)) return -1; }
    else { idx = (int)g.randint(0, modL - 2); (void)g.random(); }   // the gate passed (no libm in it): same draws
    int cur = 0, inter;
    if (d2h_one(h, mod + idx, &cur)) return -1;
    if (is_levy) inter = host_levy(g, R, C, cur, scale, a.m.levy_beta, a.m.sigma);
    else {
      int en = -1;
      if (refL > 0) { const int k = (int)g.randbelow((uint64_t)refL); if (d2h_one(h, ref + k, &en)) return -1; }
      inter = host_brownian(g, R, C, cur, en, scale);
    }
    const int2 v = make_int2(idx, inter);
    CK(hipMemcpyAsync(a.prop + a_, &v, sizeof(int2), hipMemcpyHostToDevice, h->stream));
    CK(hipStreamSynchronize(h->stream));
    h->doubts_resolved += 1;
  }
  return 0;
}

int pf_mpa_phase_batch(pf_handle* h, int32_t phase, double CF, int32_t iter, uint64_t seed, int32_t n,
                       int32_t path_cap, const int32_t* d_pop_cells, const int32_t* d_pop_len,
                       const double* d_pop_stats, const int32_t* d_gidx, const int32_t* d_slot, const int32_t* d_elite_cells,
                       int32_t elite_len, const double* d_elite_stats, int32_t* d_out_cells, int32_t* d_out_len,
                       double* d_out_stats, int32_t* d_status) {
  if (!h) return -2;
  if (!h->mpa_ready) return failmsg(h, "pf_mpa_phase_batch: call pf_mpa_setup first");
  if (phase < 1 || phase > 3 || n < 0 || path_cap < 2 || !d_pop_cells || !d_pop_len || !d_pop_stats || !d_gidx || !d_slot ||
      !d_elite_cells || !d_elite_stats || !d_out_cells || !d_out_len || !d_out_stats || !d_status) return failmsg(h, "pf_mpa_phase_batch: bad arguments");
  if (ensure_slots(h, h->mpp.allow_diag, h->mpp.restrict_corner)) return -1;
  MpaPhaseArgs a;
  a.c = make_common(h, h->mpp.allow_diag, h->mpp.restrict_corner, 16, 0);
  if (make_scorep(h, &h->mps, &a.sp)) return -1;
  a.m = mpa_dev(h); a.phase = phase; a.iter = iter; a.CF = CF; a.seed = seed; a.n = n; a.path_cap = path_cap;
  a.pop_cells = d_pop_cells; a.pop_len = d_pop_len; a.pop_stats = d_pop_stats; a.gidx = d_gidx; a.slot = d_slot;
  a.elite_cells = d_elite_cells; a.elite_len = elite_len; a.elite_len_dev = nullptr;
  a.elite_stats = d_elite_stats;
  a.out_cells = d_out_cells; a.out_len = d_out_len; a.out_stats = d_out_stats; a.status = d_status;
  a.ex_idx = nullptr; a.ex_levy = nullptr; a.ex_scale = nullptr; a.ex_agent = nullptr;
  int prc = 0;
  if (make_queue(h, n > 0 ? n : 1, [&](float* est) { prc = mpa_launch_propose(h, a, est); })) return -1;
  if (prc || mpa_resolve_doubts(h, a)) return -1;
  a.c.queue = n > 64 ? h->d_queue : nullptr;
  return launch_with_retry(h, k_mpa_phase, a, n);
}

int pf_mpa_rebuild_batch(pf_handle* h, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                         const int32_t* d_pop_cells, const int32_t* d_pop_len, const double* d_pop_stats,
                         const int32_t* d_elite_cells, int32_t elite_len, const int32_t* d_idx,
                         const int32_t* d_is_levy, const double* d_scale, const int32_t* d_agent,
                         int32_t* d_out_cells, int32_t* d_out_len, double* d_out_stats, int32_t* d_status) {
  if (!h) return -2;
  if (!h->mpa_ready) return failmsg(h, "pf_mpa_rebuild_batch: call pf_mpa_setup first");
  if (n < 0 || path_cap < 2 || !d_pop_cells || !d_pop_len || !d_pop_stats || !d_elite_cells || !d_idx || !d_is_levy ||
      !d_scale || !d_agent || !d_out_cells || !d_out_len || !d_out_stats || !d_status) return failmsg(h, "pf_mpa_rebuild_batch: bad arguments");
  if (ensure_slots(h, h->mpp.allow_diag, h->mpp.restrict_corner)) return -1;
  MpaPhaseArgs a;
  a.c = make_common(h, h->mpp.allow_diag, h->mpp.restrict_corner, 16, 0);
  if (make_scorep(h, &h->mps, &a.sp)) return -1;
  a.m = mpa_dev(h); a.phase = 0; a.iter = iter; a.CF = 0.0; a.seed = seed; a.n = n; a.path_cap = path_cap;
  a.pop_cells = d_pop_cells; a.pop_len = d_pop_len; a.pop_stats = d_pop_stats; a.gidx = nullptr; a.slot = nullptr;
  a.elite_cells = d_elite_cells; a.elite_len = elite_len; a.elite_len_dev = nullptr; a.elite_stats = d_pop_stats;
  a.out_cells = d_out_cells; a.out_len = d_out_len; a.out_stats = d_out_stats; a.status = d_status;
  a.ex_idx = d_idx; a.ex_levy = d_is_levy; a.ex_scale = d_scale; a.ex_agent = d_agent;
  if (mpa_launch_propose(h, a, nullptr) || mpa_resolve_doubts(h, a)) return -1;
  return launch_with_retry(h, k_mpa_phase, a, n);
}

int pf_mpa_fads_batch(pf_handle* h, double CF, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                      const int32_t* d_gidx, const int32_t* d_slot, int32_t* d_pop_cells, int32_t* d_pop_len, double* d_pop_stats, int32_t* d_status) {
  if (!h) return -2;
  if (!h->mpa_ready) return failmsg(h, "pf_mpa_fads_batch: call pf_mpa_setup first");
  if (n < 0 || path_cap < 2 || !d_gidx || !d_slot || !d_pop_cells || !d_pop_len || !d_pop_stats || !d_status) return failmsg(h, "pf_mpa_fads_batch: bad arguments");
  if (ensure_slots(h, h->mpp.allow_diag, h->mpp.restrict_corner)) return -1;
  if (h->tmp_cap < path_cap) {
    if (h->d_tmp) CK(hipFree(h->d_tmp));
    CK(hipMalloc(&h->d_tmp, sizeof(int) * (size_t)h->nslots * path_cap));
    h->tmp_cap = path_cap;
  }
  MpaFadsArgs a;
  a.c = make_common(h, h->mpp.allow_diag, h->mpp.restrict_corner, 16, 0);
  if (make_scorep(h, &h->mps, &a.sp)) return -1;
  a.m = mpa_dev(h); a.iter = iter; a.CF = CF; a.seed = seed; a.n = n; a.path_cap = path_cap;
  a.pop_cells = d_pop_cells; a.pop_len = d_pop_len; a.pop_stats = d_pop_stats; a.gidx = d_gidx; a.slot = d_slot;
  a.tmp_cells = h->d_tmp; a.status = d_status;
  a.init_cells = h->d_init_cells; a.init_len = h->init_len; a.init_stats = h->d_init_stats;
  a.cand_cells = nullptr; a.cand_len = nullptr; a.cand_stats = nullptr;
  if (n > 64) {
    if (make_queue(h, n, [&](float* est) { hipLaunchKernelGGL(k_plan_mpa_fads, dim3((n + 255) / 256), dim3(256), 0, h->stream, a, est); })) return -1;
    a.c.queue = h->d_queue;
  }
  return launch_with_retry(h, k_mpa_fads, a, n);
}

int pf_mpa_iter_batch(pf_handle* h, int32_t phase, double CF, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                      int32_t* d_pop_cells, int32_t* d_pop_len, double* d_pop_stats, const int32_t* d_gidx,
                      const int32_t* d_slot, const int32_t* d_elite_cells, int32_t elite_len, const double* d_elite_stats,
                      int32_t* d_c1_cells, int32_t* d_c1_len, double* d_c1_stats, int32_t* d_c2_cells, int32_t* d_c2_len,
                      double* d_c2_stats, int32_t* d_status) {
  if (!h) return -2;
  if (!h->mpa_ready) return failmsg(h, "pf_mpa_iter_batch: call pf_mpa_setup first");
  if (phase < 1 || phase > 3 || n < 0 || path_cap < 2 || !d_pop_cells || !d_pop_len || !d_pop_stats || !d_gidx || !d_slot ||
      !d_elite_cells || !d_elite_stats || !d_c1_cells || !d_c1_len || !d_c1_stats || !d_c2_cells || !d_c2_len || !d_c2_stats || !d_status)
    return failmsg(h, "pf_mpa_iter_batch: bad arguments");
  if (n == 0) return 0;
  if (ensure_slots(h, h->mpp.allow_diag, h->mpp.restrict_corner)) return -1;
  MpaSweepArgs a;
  a.ph.c = make_common(h, h->mpp.allow_diag, h->mpp.restrict_corner, 16, 0);
  if (make_scorep(h, &h->mps, &a.ph.sp)) return -1;
  a.ph.m = mpa_dev(h); a.ph.phase = phase; a.ph.iter = iter; a.ph.CF = CF; a.ph.seed = seed; a.ph.n = n; a.ph.path_cap = path_cap;
  a.ph.pop_cells = d_pop_cells; a.ph.pop_len = d_pop_len; a.ph.pop_stats = d_pop_stats; a.ph.gidx = d_gidx; a.ph.slot = d_slot;
  a.ph.elite_cells = d_elite_cells; a.ph.elite_len = elite_len; a.ph.elite_stats = d_elite_stats;
  a.ph.elite_len_dev = nullptr;
  if (elite_len < 0) { if (ensure_elite_buf(h)) return -1; a.ph.elite_len_dev = h->d_elite_len; }   // device-resident iteration: the length stays in HBM
  a.ph.out_cells = d_c1_cells; a.ph.out_len = d_c1_len; a.ph.out_stats = d_c1_stats; a.ph.status = d_status;
  a.ph.ex_idx = nullptr; a.ph.ex_levy = nullptr; a.ph.ex_scale = nullptr; a.ph.ex_agent = nullptr;
  a.fd.c = a.ph.c; a.fd.sp = a.ph.sp; a.fd.m = a.ph.m; a.fd.iter = iter; a.fd.CF = CF; a.fd.seed = seed; a.fd.n = n; a.fd.path_cap = path_cap;
  a.fd.pop_cells = d_pop_cells; a.fd.pop_len = d_pop_len; a.fd.pop_stats = d_pop_stats; a.fd.gidx = d_gidx; a.fd.slot = d_slot;
  a.fd.tmp_cells = nullptr; a.fd.status = d_status;
  a.fd.init_cells = h->d_init_cells; a.fd.init_len = h->init_len; a.fd.init_stats = h->d_init_stats;
  a.fd.cand_cells = d_c2_cells; a.fd.cand_len = d_c2_len; a.fd.cand_stats = d_c2_stats;
  int prc = 0;
  if (make_queue(h, 2 * n, [&](float* est) {
        prc = mpa_launch_propose(h, a.ph, est);
        hipLaunchKernelGGL(k_plan_mpa_fads, dim3((n + 255) / 256), dim3(256), 0, h->stream, a.fd, est + n);
      })) return -1;
  if (prc || mpa_resolve_doubts(h, a.ph)) return -1;
  a.ph.c.queue = h->d_queue; a.fd.c.queue = h->d_queue;
  const int S = kLdsS;
  a.ph.c.S = S; a.fd.c.S = S; a.ph.c.retry = 0; a.fd.c.retry = 0;
  if (2 * n > h->job_cap) {
    if (h->d_jobs) CK(hipFree(h->d_jobs));
    if (h->d_jres) CK(hipFree(h->d_jres));
    CK(hipMalloc(&h->d_jobs, sizeof(MpaJob) * 2 * (size_t)n)); CK(hipMalloc(&h->d_jres, sizeof(MpaRes) * 2 * (size_t)n));
    h->job_cap = 2 * n;
  }
  MpaJob* jobs = (MpaJob*)h->d_jobs; MpaRes* jres = (MpaRes*)h->d_jres;
  MpaSearchArgs sa;
  sa.c = a.ph.c; sa.jobs = jobs; sa.res = jres; sa.n_items = 2 * n; sa.path_cap = path_cap; sa.ph_cells = d_c1_cells; sa.fd_cells = d_c2_cells; sa.n = n;
#ifdef PF_TWO_WAVE
  const bool pr = g_two_wave != 0 && !plateau_map(h);               // two wavefronts per search (pf_astar_pr.h)
  const size_t lds = pr ? (size_t)PF_PR_LDS_BYTES : open_bytes(S);
  if (pr) CK(hipFuncSetAttribute((const void*)k_mpa_search<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else
#else
  const size_t lds = open_bytes(S);
#endif
  CK(hipFuncSetAttribute((const void*)k_mpa_search<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((160 * 1024) / lds); if (per_cu > kWavesPerCU) per_cu = kWavesPerCU; if (per_cu < 1) per_cu = 1;
  int grid = (h->nslots / kSlotsPerCU) * per_cu; if (grid > 2 * n) grid = 2 * n; if (grid > h->nslots) grid = h->nslots;
  CK(hipMemsetAsync(h->d_work, 0, sizeof(int), h->stream));
  CK(hipMemsetAsync(h->d_cnt, 0, sizeof(DevCounters), h->stream));
  hipLaunchKernelGGL(k_mpa_plan, dim3(2 * n), dim3(64), 0, h->stream, a, jobs, jres);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev0, h->stream));
#ifdef PF_TWO_WAVE
  if (pr) hipLaunchKernelGGL(k_mpa_search<true>, dim3(grid), dim3(128), lds, h->stream, sa);
  else
#endif
  hipLaunchKernelGGL(k_mpa_search<false>, dim3(grid), dim3(64), lds, h->stream, sa);
  CK(hipGetLastError());
  CK(hipEventRecord(h->ev1, h->stream));
  hipLaunchKernelGGL(k_mpa_finish, dim3(2 * n), dim3(64), 0, h->stream, a, (const MpaJob*)jobs, (const MpaRes*)jres);
  CK(hipGetLastError());
  hipLaunchKernelGGL(k_mpa_apply, dim3(n), dim3(64), 0, h->stream, n, path_cap, d_slot, d_c1_cells, d_c1_len, d_c1_stats,
                     d_c2_cells, d_c2_len, d_c2_stats, d_pop_cells, d_pop_len, d_pop_stats);
  CK(hipGetLastError());
  DevCounters dc;
  if (end_batch(h, &dc)) return -1;
  CK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return 0;
}

// ---- GA generation on the device ----
int pf_ga_select_dev(pf_handle* h, uint64_t seed, int32_t gen, int32_t n, int32_t tournament_size, const double* d_fit_all,
                     const int32_t* d_gorder, int32_t* d_psid) {
  if (!h) return -2;
  if (n <= 0 || tournament_size <= 0 || tournament_size > 64 || !d_fit_all || !d_gorder || !d_psid) return failmsg(h, "pf_ga_select_dev: bad arguments (1 <= tournament_size <= 64)");
  CK(hipSetDevice(h->device));
  const int k = tournament_size < n ? tournament_size : n;
  if (n > h->ga_pool_cap) { if (h->d_ga_pool) CK(hipFree(h->d_ga_pool)); CK(hipMalloc(&h->d_ga_pool, sizeof(int) * (size_t)n)); h->ga_pool_cap = n; }
  hipLaunchKernelGGL(k_ga_select, dim3(1), dim3(1), 0, h->stream, (unsigned long long)seed, gen, n, k, d_fit_all, d_gorder, h->d_ga_pool, d_psid);
  CK(hipGetLastError());
  return 0;
}
int pf_ga_breed_dev(pf_handle* h, uint64_t seed, int32_t gen, int32_t N, int32_t W, double crossover_rate, double mutation_rate,
                    const int32_t* d_chrom_all, const int32_t* d_psid, int32_t child0, int32_t nchild, int32_t* d_out) {
  if (!h) return -2;
  if (N <= 0 || W <= 0 || !d_chrom_all || !d_psid || !d_out || child0 < 0 || nchild < 0 || child0 + nchild > N) return failmsg(h, "pf_ga_breed_dev: bad arguments");
  if (nchild == 0) return 0;
  CK(hipSetDevice(h->device));
  const int pairs = (child0 + nchild - 1) / 2 - child0 / 2 + 1;
  hipLaunchKernelGGL(k_ga_breed, dim3((pairs + 63) / 64), dim3(64), 0, h->stream, (unsigned long long)seed, gen, N, W, crossover_rate, mutation_rate,
                     h->d_occ, h->R, h->C, d_chrom_all, d_psid, child0, nchild, d_out);
  CK(hipGetLastError());
  return 0;
}
int pf_ga_assemble_dev(pf_handle* h, int32_t n_loc, int32_t W, int32_t path_cap, int32_t lo, const int32_t* d_kid_len,
                       const int32_t* d_kid_chrom, const double* d_kid_stats, const int32_t* d_kid_cells, const int32_t* d_psid,
                       const int32_t* d_chrom_old, const double* d_stats_old, const int32_t* d_cells_old, const int32_t* d_len_old,
                       int32_t old_lo, int32_t old_hi, int32_t* d_chrom_new, double* d_stats_new, int32_t* d_cells_new,
                       int32_t* d_len_new) {
  if (!h) return -2;
  if (n_loc < 0 || W <= 0 || path_cap < 1 || !d_kid_len || !d_kid_chrom || !d_kid_stats || !d_kid_cells || !d_psid || !d_chrom_old ||
      !d_stats_old || !d_cells_old || !d_len_old || !d_chrom_new || !d_stats_new || !d_cells_new || !d_len_new) return failmsg(h, "pf_ga_assemble_dev: bad arguments");
  if (n_loc == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_ga_assemble, dim3(n_loc), dim3(64), 0, h->stream, n_loc, W, path_cap, lo, d_kid_len, d_kid_chrom, d_kid_stats, d_kid_cells,
                     d_psid, d_chrom_old, d_stats_old, d_cells_old, d_len_old, old_lo, old_hi, d_chrom_new, d_stats_new, d_cells_new, d_len_new);
  CK(hipGetLastError());
  return 0;
}

// ---- device-resident iteration control -------------------------------------------------------------------------
int pf_sort_order_by_key(pf_handle* h, int32_t n, const double* d_vals, int32_t stride, int32_t offset, int32_t* d_order) {
  if (!h) return -2;
  if (n < 0 || !d_vals || !d_order || stride < 1 || offset < 0 || offset >= stride) return failmsg(h, "pf_sort_order_by_key: bad arguments");
  if (n <= 1) return 0;
  CK(hipSetDevice(h->device));
  // stable, and doubles order as list.sort(key=...) orders them (inf last; the fitness is never NaN)
  return rank_sort(h, n, 0, d_vals, stride, offset, d_order, nullptr, d_order);
}
// the head of a sorted list in ONE small copy: out2 = {d_order[0], d_vals[d_order[0] * stride + offset]}
__global__ void k_sorted_head(const int* order, const double* vals, int stride, int offset, double* out2) {
  const int id = order[0];
  out2[0] = (double)id; out2[1] = vals[(size_t)id * stride + offset];
}
int pf_sorted_head(pf_handle* h, const int32_t* d_order, const double* d_vals, int32_t stride, int32_t offset, double* out2) {
  if (!h) return -2;
  if (!d_order || !d_vals || !out2 || stride < 1 || offset < 0 || offset >= stride) return failmsg(h, "pf_sorted_head: bad arguments");
  CK(hipSetDevice(h->device));
  if (!h->d_scan3) CK(hipMalloc(&h->d_scan3, 24));
  hipLaunchKernelGGL(k_sorted_head, dim3(1), dim3(1), 0, h->stream, d_order, d_vals, stride, offset, (double*)h->d_scan3);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(out2, h->d_scan3, 16, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  h->d2h_small += 1;
  return 0;
}
// out[i] = a[i] + sign * b[i] on device columns (the non-strict MAACO exchange forms its per-rank pheromone delta and applies the
// reduced one with it: no matrix crosses PCIe)
__global__ void k_vec_axpy(int n, const double* a, const double* b, double sign, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + sign * b[i];
}
int pf_vec_add_f64(pf_handle* h, int32_t n, const double* d_a, const double* d_b, double sign, double* d_out) {
  if (!h) return -2;
  if (n < 0 || !d_a || !d_b || !d_out || !(sign == 1.0 || sign == -1.0)) return failmsg(h, "pf_vec_add_f64: bad arguments (sign is +1 or -1)");
  CK(hipSetDevice(h->device));
  if (n > 0) hipLaunchKernelGGL(k_vec_axpy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, d_a, d_b, sign, d_out);
  CK(hipGetLastError());
  return 0;
}
int pf_gather_col(pf_handle* h, int32_t n, const double* d_src, int32_t stride, int32_t offset, double* d_dst) {
  if (!h) return -2;
  if (n < 0 || !d_src || !d_dst || stride < 1 || offset < 0 || offset >= stride) return failmsg(h, "pf_gather_col: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_gather_col, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, d_src, stride, offset, d_dst);
  CK(hipGetLastError());
  return 0;
}
int pf_mpa_elite_buf(pf_handle* h, void** d_cells, void** d_len, void** d_stats) {
  if (!h) return -2;
  CK(hipSetDevice(h->device));
  if (ensure_elite_buf(h)) return -1;
  if (d_cells) *d_cells = h->d_elite_cells;
  if (d_len) *d_len = h->d_elite_len;
  if (d_stats) *d_stats = h->d_elite_stats;
  return 0;
}
int pf_mpa_pick_elite(pf_handle* h, int32_t path_cap, const int32_t* d_pop_cells, const int32_t* d_pop_len,
                      const double* d_pop_stats, const int32_t* d_order, int32_t first_id) {
  if (!h) return -2;
  if (path_cap < 1 || !d_pop_cells || !d_pop_len || !d_pop_stats || !d_order) return failmsg(h, "pf_mpa_pick_elite: bad arguments");
  CK(hipSetDevice(h->device));
  if (ensure_elite_buf(h)) return -1;
  hipLaunchKernelGGL(k_mpa_pick_elite, dim3(1), dim3(256), 0, h->stream, path_cap, d_pop_cells, d_pop_len, d_pop_stats, d_order, first_id,
                     h->d_elite_cells, h->d_elite_len, h->d_elite_stats);
  CK(hipGetLastError());
  return 0;
}
int pf_mpa_local_view(pf_handle* h, int32_t N, const int32_t* d_gorder, int32_t lo, int32_t hi, int32_t* d_gidx, int32_t* d_slot) {
  if (!h) return -2;
  if (N < 0 || !d_gorder || !d_gidx || !d_slot || lo < 0 || hi < lo) return failmsg(h, "pf_mpa_local_view: bad arguments");
  if (N == 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_mpa_local_view, dim3(1), dim3(1024), 0, h->stream, N, d_gorder, lo, hi, d_gidx, d_slot);
  CK(hipGetLastError());
  return 0;
}

int pf_mpa_memory(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_slot,
                  const int32_t* d_cand_cells, const int32_t* d_cand_len, const double* d_cand_stats,
                  int32_t* d_pop_cells, int32_t* d_pop_len, double* d_pop_stats) {
  if (!h) return -2;
  if (n <= 0) return 0;
  CK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_mpa_memory, dim3(n), dim3(64), 0, h->stream, n, path_cap, d_slot, d_cand_cells, d_cand_len,
                     d_cand_stats, d_pop_cells, d_pop_len, d_pop_stats);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(h->stream));
  return 0;
}


long long pf_mpa_doubts_resolved(pf_handle* h) { return h ? h->doubts_resolved : 0; }

int pf_selftest_mpa_targets(pf_handle* h, uint64_t seed, int32_t n, int32_t is_levy, double beta, double sigma, double scale,
                            const int32_t* d_cur, const int32_t* d_elite, int32_t* d_out, int64_t* n_doubt) {
  if (!h || n < 0 || !d_cur || !d_elite || !d_out) return failmsg(h, "pf_selftest_mpa_targets: bad arguments");
  if (n == 0) return 0;
  CK(hipSetDevice(h->device));
  unsigned char* d_flag = nullptr;
  CK(hipMalloc(&d_flag, (size_t)n));
  const Grid G = make_grid(h, 1, 1);
  hipLaunchKernelGGL(k_selftest_targets, dim3((n + 255) / 256), dim3(256), 0, h->stream, G, (unsigned long long)seed, n, is_levy, beta, sigma,
                     scale, d_cur, d_elite, g_doubt_log, g_doubt_round, d_out, d_flag);
  CK(hipGetLastError());
  std::vector<unsigned char> flag((size_t)n);
  CK(hipMemcpyAsync(flag.data(), d_flag, (size_t)n, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  long long nd = 0;
  for (int i = 0; i < n; ++i) if (flag[i]) {                      // the same hand-over as mpa_resolve_doubts
    int cur = 0, el = 0;
    if (d2h_one(h, d_cur + i, &cur) || d2h_one(h, d_elite + i, &el)) { (void)hipFree(d_flag); return -1; }
    HostRng g(seed, DOM_MPA, 0, (uint64_t)i);
    const int t = is_levy ? host_levy(g, h->R, h->C, cur, scale, beta, sigma) : host_brownian(g, h->R, h->C, cur, el, scale);
    CK(hipMemcpyAsync(d_out + i, &t, sizeof(int), hipMemcpyHostToDevice, h->stream));
    CK(hipStreamSynchronize(h->stream));
    nd += 1;
  }
  (void)hipFree(d_flag);
  if (n_doubt) *n_doubt = nd;
  return 0;
}


// ---------------------------------------------------------------------------
// RCCL over xGMI, bound directly (no torch in the data path).  librccl is opened on first use, so single-GPU users never
// load it.  Every collective is enqueued on the handle's stream, in order with the kernels: nothing synchronises the
// host.  The caller ships the 128-byte unique id from rank 0 to the other ranks by any means (a file, an environment
// variable, torch.distributed's store).
// ---------------------------------------------------------------------------
#define NK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_fail(h, #call, r_); } while (0)
#define NEED_COMM(h) do { if (!(h)) return -2; if (!(h)->comm) return failmsg(h, "pf_comm: call pf_comm_init first"); CK(hipSetDevice((h)->device)); } while (0)

int pf_comm_unique_id(void* id128) {
  if (!id128) return -2;
  if (const char* e = rccl_load()) { g_create_err = e; return -2; }
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return -1; }
  memcpy(id128, &id, sizeof(id));
  return 0;
}
int pf_comm_init(pf_handle* h, int32_t rank, int32_t world, const void* id128) {
  if (!h) return -2;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return failmsg(h, "pf_comm_init: bad arguments");
  if (h->comm) return failmsg(h, "pf_comm_init: this handle already has a communicator");
  if (const char* e = rccl_load()) return failmsg(h, e);
  CK(hipSetDevice(h->device));
  ncclUniqueId id; memcpy(&id, id128, sizeof(id));
  NK(g_rccl.CommInitRank(&h->comm, world, id, rank));
  h->comm_rank = rank; h->comm_world = world;
  return 0;
}
int pf_comm_destroy(pf_handle* h) {
  if (!h) return -2;
  if (h->comm) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  return 0;
}
int pf_comm_all_gather(pf_handle* h, const void* d_send, void* d_recv, int64_t bytes_per_rank) {
  NEED_COMM(h);
  if (bytes_per_rank < 0 || !d_send || !d_recv) return failmsg(h, "pf_comm_all_gather: bad arguments");
  if (bytes_per_rank) NK(g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclInt8, h->comm, h->stream));
  return 0;
}
int pf_comm_broadcast(pf_handle* h, void* d_buf, int64_t bytes, int32_t root) {
  NEED_COMM(h);
  if (bytes < 0 || !d_buf || root < 0 || root >= h->comm_world) return failmsg(h, "pf_comm_broadcast: bad arguments");
  if (bytes) NK(g_rccl.Broadcast(d_buf, d_buf, (size_t)bytes, ncclInt8, root, h->comm, h->stream));
  return 0;
}
int pf_comm_all_reduce_f64(pf_handle* h, double* d_buf, int64_t count, int32_t op) {   // op: 0 sum, 1 min, 2 max
  NEED_COMM(h);
  if (count < 0 || !d_buf || op < 0 || op > 2) return failmsg(h, "pf_comm_all_reduce_f64: bad arguments");
  if (count) NK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax), h->comm, h->stream));
  return 0;
}
int pf_comm_send(pf_handle* h, const void* d_buf, int64_t bytes, int32_t peer) {
  NEED_COMM(h);
  if (bytes < 0 || !d_buf || peer < 0 || peer >= h->comm_world || peer == h->comm_rank) return failmsg(h, "pf_comm_send: bad arguments");
  if (bytes) NK(g_rccl.Send(d_buf, (size_t)bytes, ncclInt8, peer, h->comm, h->stream));
  return 0;
}
int pf_comm_recv(pf_handle* h, void* d_buf, int64_t bytes, int32_t peer) {
  NEED_COMM(h);
  if (bytes < 0 || !d_buf || peer < 0 || peer >= h->comm_world || peer == h->comm_rank) return failmsg(h, "pf_comm_recv: bad arguments");
  if (bytes) NK(g_rccl.Recv(d_buf, (size_t)bytes, ncclInt8, peer, h->comm, h->stream));
  return 0;
}
// one step of a ring: send to `to` and receive from `from` as ONE group (both directions progress together)
int pf_comm_sendrecv(pf_handle* h, const void* d_send, int64_t send_bytes, int32_t to, void* d_recv, int64_t recv_bytes, int32_t from) {
  NEED_COMM(h);
  const bool snd = send_bytes > 0 && to >= 0, rcv = recv_bytes > 0 && from >= 0;
  if (send_bytes < 0 || recv_bytes < 0 || (snd && (!d_send || to >= h->comm_world)) || (rcv && (!d_recv || from >= h->comm_world)))   // (a ring step onto itself is legal inside a group)
    return failmsg(h, "pf_comm_sendrecv: bad arguments");
  NK(g_rccl.GroupStart());
  // an error between GroupStart and GroupEnd must still close the group: an open group would silently swallow every later
  // collective of this thread
  ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess;
  if (snd) r1 = g_rccl.Send(d_send, (size_t)send_bytes, ncclInt8, to, h->comm, h->stream);
  if (rcv && r1 == ncclSuccess) r2 = g_rccl.Recv(d_recv, (size_t)recv_bytes, ncclInt8, from, h->comm, h->stream);
  const ncclResult_t r3 = g_rccl.GroupEnd();
  if (r1 != ncclSuccess) return nccl_fail(h, "ncclSend (pf_comm_sendrecv)", r1);
  if (r2 != ncclSuccess) return nccl_fail(h, "ncclRecv (pf_comm_sendrecv)", r2);
  if (r3 != ncclSuccess) return nccl_fail(h, "ncclGroupEnd (pf_comm_sendrecv)", r3);
  return 0;
}
// ---- spans: time stretches of work on the handle's stream with HIP events, without synchronising while they are recorded ----
// (what bench.py brackets the collectives of an iteration with: the exchange runs on the same stream as the kernels, so
// torch.cuda.Event -- torch's current stream -- cannot see it, and a host clock around an asynchronous enqueue sees nothing)
#define PF_SPAN_MAX 512
static int span_fold(pf_handle* h) {                                // synchronise, add up the closed spans, recycle their events
  CK(hipStreamSynchronize(h->stream));
  for (int i = 0; i + 1 < h->span_n; i += 2) {
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, h->span_ev[(size_t)i], h->span_ev[(size_t)i + 1]));
    h->span_ms += ms; h->span_cnt += 1;
  }
  h->span_n = 0;
  return 0;
}
int pf_span_begin(pf_handle* h) {
  if (!h) return -2;
  if (h->span_open) return failmsg(h, "pf_span_begin: the previous span is still open");
  CK(hipSetDevice(h->device));
  if (h->span_n + 2 > 2 * PF_SPAN_MAX && span_fold(h)) return -1;
  while ((int)h->span_ev.size() < h->span_n + 2) { hipEvent_t e; CK(hipEventCreate(&e)); h->span_ev.push_back(e); }
  CK(hipEventRecord(h->span_ev[(size_t)h->span_n], h->stream));
  h->span_open = true;
  return 0;
}
int pf_span_end(pf_handle* h) {
  if (!h) return -2;
  if (!h->span_open) return failmsg(h, "pf_span_end: no span is open");
  CK(hipEventRecord(h->span_ev[(size_t)h->span_n + 1], h->stream));
  h->span_n += 2; h->span_open = false;
  return 0;
}
int pf_span_total(pf_handle* h, double* ms_out, int64_t* count_out, int32_t reset) {
  if (!h) return -2;
  if (h->span_open) return failmsg(h, "pf_span_total: a span is still open");
  CK(hipSetDevice(h->device));
  if (span_fold(h)) return -1;
  if (ms_out) *ms_out = h->span_ms;
  if (count_out) *count_out = h->span_cnt;
  if (reset) { h->span_ms = 0.0; h->span_cnt = 0; }
  return 0;
}

int pf_comm_rank(pf_handle* h) { return h ? h->comm_rank : 0; }
int pf_comm_world(pf_handle* h) { return h && h->comm ? h->comm_world : 1; }

}  // extern "C"
