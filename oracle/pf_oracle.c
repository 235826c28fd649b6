/*
 * pf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the population-fitness hot path of
 * dvnam1605/MAACO-path-planing (reference mounted at /root/reference).  Every
 * function cites the reference file:line it follows.  It exists so that
 *   (1) tests/ can check the HIP kernels against it on the GPU box (where the
 *       Python reference does not exist), and
 *   (2) bench.py's `cpu_baseline` leg can time it (kind = "port").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (maaco-path-planing_amd/) never does.
 *
 * Parity pinning: the reference has no tests or golden vectors of its own
 * (SURVEY.md section 4).  This restatement is pinned by (a) golden vectors
 * captured from the unmodified reference imported in the build container
 * (oracle/capture_golden.py -> tests/golden/), and (b) live comparison against
 * the imported reference in tests/test_oracle_vs_reference.py (skipped where
 * /root/reference is absent).
 *
 * Conventions: a cell is r*C + c (int32).  occ[cell] == 1 is an obstacle
 * (env.py:5); any other value is free (START/TARGET markers 2/3 are free).
 * All reals are IEEE double, evaluated in the reference's order; build with
 * -ffp-contract=off so no FMA contraction changes a rounding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* Keyed counter RNG + CPython 3.10 derived distributions              */
/* ------------------------------------------------------------------ */
/* The reference draws from one global, unseeded MT19937 stream shared by all
 * agents, which no agent-parallel engine can reproduce.  The
 * parity contract is therefore per agent-call: agent `a` of iteration `it` in
 * domain `dom` owns the stream keyed (seed, dom, it, a).  The Python twin
 * (pathfit/rng.py: AgentRandom, a random.Random subclass overriding random()
 * and getrandbits()) is installed into the reference's modules by the capture
 * harness, so CPython itself performs the derivations restated below. */
typedef struct { uint64_t key, ctr; } orc_rng;

static inline uint64_t orc_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31; return z;
}
ORC_API void orc_rng_init(orc_rng* g, uint64_t seed, uint64_t dom, uint64_t it, uint64_t agent) {
  uint64_t k = orc_mix64(seed + 0x9E3779B97F4A7C15ULL * (dom + 1));
  k = orc_mix64(k + 0xD1B54A32D192ED03ULL * (it + 1));
  k = orc_mix64(k + 0x8CB92BA72F3D8DD7ULL * (agent + 1));
  g->key = k; g->ctr = 0;
}
ORC_API uint64_t orc_rng_next64(orc_rng* g) {
  g->ctr += 1;
  return orc_mix64(g->key + g->ctr * 0x9E3779B97F4A7C15ULL);
}
/* random(): 53-bit mantissa, [0,1) */
ORC_API double orc_rng_random(orc_rng* g) { return (double)(orc_rng_next64(g) >> 11) * (1.0 / 9007199254740992.0); }
/* getrandbits(k), 1 <= k <= 64 */
ORC_API uint64_t orc_rng_getrandbits(orc_rng* g, int k) { return orc_rng_next64(g) >> (64 - k); }
/* random.py _randbelow_with_getrandbits: k = n.bit_length(); reject r >= n */
ORC_API uint64_t orc_rng_randbelow(orc_rng* g, uint64_t n) {
  if (!n) return 0;
  int k = 64 - __builtin_clzll(n);
  uint64_t r = orc_rng_getrandbits(g, k);
  while (r >= n) r = orc_rng_getrandbits(g, k);
  return r;
}
/* random.py randint(a,b) = a + _randbelow(b-a+1) */
ORC_API int64_t orc_rng_randint(orc_rng* g, int64_t a, int64_t b) { return a + (int64_t)orc_rng_randbelow(g, (uint64_t)(b - a + 1)); }
/* random.py uniform(a,b) = a + (b-a)*random() */
ORC_API double orc_rng_uniform(orc_rng* g, double a, double b) { return a + (b - a) * orc_rng_random(g); }
/* random.py normalvariate: Kinderman-Monahan ratio of uniforms */
ORC_API double orc_rng_normalvariate(orc_rng* g, double mu, double sigma) {
  const double NV_MAGICCONST = 1.7155277699214135; /* 4*exp(-0.5)/sqrt(2.0) */
  double z;
  for (;;) {
    double u1 = orc_rng_random(g);
    double u2 = 1.0 - orc_rng_random(g);
    z = NV_MAGICCONST * (u1 - 0.5) / u2;
    double zz = z * z / 4.0;
    if (zz <= -log(u2)) break;
  }
  return mu + z * sigma;
}

/* ------------------------------------------------------------------ */
/* a1-a3: grid primitives                                              */
/* ------------------------------------------------------------------ */
/* helper.py:14 / MPA.py:62 */
static inline int orc_free(const uint8_t* occ, int R, int C, int r, int c) {
  return r >= 0 && r < R && c >= 0 && c < C && occ[r * C + c] != 1;
}
/* helper.py:8 / MPA.py:56 / MAACO.py:55: math.sqrt of an exact integer */
static inline double orc_dist(int r1, int c1, int r2, int c2) {
  long dr = r1 - r2, dc = c1 - c2;
  return sqrt((double)(dr * dr + dc * dc));
}
/* helper.py:30-36 / MPA.py:71-77 move order */
static const int HM_DR[8] = {0, 0, 1, -1, 1, 1, -1, -1};
static const int HM_DC[8] = {1, -1, 0, 0, 1, -1, 1, -1};

/* helper.py:18-53 / MPA.py:65-100.  excl: per-cell byte mask or NULL. */
ORC_API int orc_neighbors(const uint8_t* occ, int R, int C, int r, int c, int allow_diag,
                          int restrict_corner, const uint8_t* excl, int32_t* out) {
  int n = 0, nm = allow_diag ? 8 : 4;
  for (int m = 0; m < nm; ++m) {
    int nr = r + HM_DR[m], nc = c + HM_DC[m];
    if (!orc_free(occ, R, C, nr, nc) || (excl && excl[nr * C + nc])) continue;
    if (m >= 4 && restrict_corner) {
      if (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m])) continue;
    }
    out[n++] = nr * C + nc;
  }
  return n;
}

/* ------------------------------------------------------------------ */
/* a4/a5: the two A* connectors                                        */
/* ------------------------------------------------------------------ */
/* Exact priority queue on the total order (f, g, r, c) = (f, g, cell): any
 * exact structure reproduces heapq because the open list never holds two
 * entries for one node (astar.py:92-100, MPA.py:141-150). */
typedef struct { double f, g; int32_t cell; } orc_ent;
typedef struct { orc_ent* e; int32_t* pos; int n, cap; int max_n; } orc_heap;

static inline int ent_lt(const orc_ent* a, const orc_ent* b) {
  if (a->f != b->f) return a->f < b->f;
  if (a->g != b->g) return a->g < b->g;
  return a->cell < b->cell;
}
static void heap_set(orc_heap* h, int i, orc_ent v) { h->e[i] = v; h->pos[v.cell] = i; }
static void heap_up(orc_heap* h, int i) {
  orc_ent v = h->e[i];
  while (i > 0) { int p = (i - 1) >> 1; if (!ent_lt(&v, &h->e[p])) break; heap_set(h, i, h->e[p]); i = p; }
  heap_set(h, i, v);
}
static void heap_down(orc_heap* h, int i) {
  orc_ent v = h->e[i];
  for (;;) {
    int l = 2 * i + 1; if (l >= h->n) break;
    if (l + 1 < h->n && ent_lt(&h->e[l + 1], &h->e[l])) l++;
    if (!ent_lt(&h->e[l], &v)) break;
    heap_set(h, i, h->e[l]); i = l;
  }
  heap_set(h, i, v);
}
static void heap_push(orc_heap* h, orc_ent v) {
  if (h->n == h->cap) { h->cap *= 2; h->e = (orc_ent*)realloc(h->e, sizeof(orc_ent) * h->cap); }
  h->e[h->n] = v; h->pos[v.cell] = h->n; h->n++; heap_up(h, h->n - 1);
  if (h->n > h->max_n) h->max_n = h->n;
}
static orc_ent heap_pop(orc_heap* h) {
  orc_ent top = h->e[0]; h->pos[top.cell] = -1; h->n--;
  if (h->n > 0) { heap_set(h, 0, h->e[h->n]); heap_down(h, 0); }
  return top;
}

/* Workspace reused across solves (the CPU baseline must not pay malloc/clear
 * of R*C arrays per solve any more than a sane port would). */
typedef struct {
  int RC; double* g; int32_t* parent; int32_t* pos; uint8_t* closed; uint32_t* stamp; uint32_t epoch;
  orc_heap h; int32_t* tmp;
} orc_ws;

ORC_API orc_ws* orc_ws_create(int R, int C) {
  orc_ws* w = (orc_ws*)calloc(1, sizeof(orc_ws));
  w->RC = R * C;
  w->g = (double*)malloc(sizeof(double) * w->RC);
  w->parent = (int32_t*)malloc(sizeof(int32_t) * w->RC);
  w->pos = (int32_t*)malloc(sizeof(int32_t) * w->RC);
  w->closed = (uint8_t*)malloc(w->RC);
  w->stamp = (uint32_t*)calloc(w->RC, sizeof(uint32_t));
  w->tmp = (int32_t*)malloc(sizeof(int32_t) * w->RC);
  w->h.cap = 1024; w->h.e = (orc_ent*)malloc(sizeof(orc_ent) * w->h.cap); w->h.pos = w->pos;
  return w;
}
ORC_API void orc_ws_destroy(orc_ws* w) {
  if (!w) return;
  free(w->g); free(w->parent); free(w->pos); free(w->closed); free(w->stamp); free(w->tmp); free(w->h.e); free(w);
}
/* lazily (re)initialise one cell's per-solve state */
static inline void ws_touch(orc_ws* w, int cell) {
  if (w->stamp[cell] != w->epoch) {
    w->stamp[cell] = w->epoch; w->g[cell] = INFINITY; w->parent[cell] = -1; w->pos[cell] = -1; w->closed[cell] = 0;
  }
}

/* stats[0]=pops (loop iterations, the reference's `steps`), [1]=pushes,
 * [2]=max open size, [3]=in-place decrease-keys (variant 0) / suppressed
 * re-pushes (variant 1), [4]=neighbours examined, [5]=status
 * (0 ok, 1 infeasible, 2 step cap). */
static int64_t reconstruct(orc_ws* w, int start, int target, int32_t* out, int64_t cap) {
  /* astar.py:65-69 / MPA.py:124-130 */
  int64_t n = 0; int t = target;
  while (w->stamp[t] == w->epoch && w->parent[t] >= 0) { w->tmp[n++] = t; t = w->parent[t]; }
  w->tmp[n++] = start;
  if (n > cap) return -1;
  for (int64_t i = 0; i < n; ++i) out[i] = w->tmp[n - 1 - i];
  return n;
}

/* Test hook: > 0 lowers the step cap of both connectors below the reference's 3RC / 2RC, so that the cap path can be
 * exercised (the reference's own caps cannot be reached on any grid tried: DESIGN.md 2).  Mirrors the product's
 * pf_set_option("astar_step_cap"). */
static int64_t g_step_cap_override = 0;
ORC_API void orc_set_step_cap(int64_t cap) { g_step_cap_override = cap > 0 ? cap : 0; }

/* variant 0: AStarSolver.solve, astar.py:33-101.  avoid = nodes_to_avoid as a
 * per-cell byte mask (or NULL).  Returns path length in cells, 0 for [],
 * -1 if `cap` is too small. */
static int64_t astar_v0(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner,
                        int start, int target, const uint8_t* avoid, int32_t* out, int64_t cap, int64_t* st, int hzero) {
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  st[5] = 1;
  if (!orc_free(occ, R, C, sr, sc) || !orc_free(occ, R, C, tr, tc)) return 0;   /* :37-39 */
  if (start == target) { if (cap < 1) return -1; out[0] = start; st[5] = 0; return 1; } /* :41 */
  w->epoch++; w->h.n = 0; w->h.max_n = 0;
  ws_touch(w, start);
  w->g[start] = 0.0;
  /* hzero: DijkstraSolver.solve (dijkstra.py:32-97) is the same loop with heap entries (g, node): key (g, g, node) */
  orc_ent e0 = {hzero ? 0.0 : orc_dist(sr, sc, tr, tc), 0.0, start};               /* :45 / dijkstra.py:45 */
  heap_push(&w->h, e0); st[1]++;
  int64_t max_steps = (int64_t)R * C * 3, steps = 0;                               /* :58 */
  if (g_step_cap_override > 0 && g_step_cap_override < max_steps) max_steps = g_step_cap_override;
  int32_t nb[8];
  while (w->h.n > 0 && steps < max_steps) {                                        /* :60 */
    steps++;
    orc_ent cur = heap_pop(&w->h);                                                 /* :62 */
    if (cur.cell == target) {                                                      /* :64 */
      st[0] = steps; st[2] = w->h.max_n; st[5] = 0;
      return reconstruct(w, start, target, out, cap);
    }
    ws_touch(w, cur.cell);
    /* closed_set = nodes_to_avoid - {start,target} plus popped nodes :51-56,:73-74 */
    int in_closed = w->closed[cur.cell] || (avoid && avoid[cur.cell] && cur.cell != start && cur.cell != target);
    if (in_closed) continue;
    w->closed[cur.cell] = 1;
    int r = cur.cell / C, c = cur.cell % C;
    int nm = allow_diag ? 8 : 4;
    for (int m = 0; m < nm; ++m) {                                                 /* helper.py:38-52 */
      int nr = r + HM_DR[m], nc = c + HM_DC[m];
      if (!orc_free(occ, R, C, nr, nc)) continue;
      int n = nr * C + nc;
      ws_touch(w, n);
      if (w->closed[n] || (avoid && avoid[n] && n != start && n != target)) continue; /* exclude_nodes=closed_set :80 */
      if (m >= 4 && restrict_corner &&
          (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
      nb[0] = n; st[4]++;
      double tentative = cur.g + orc_dist(r, c, nr, nc);                           /* :84-85 */
      if (tentative < w->g[n]) {                                                   /* :87 */
        w->parent[n] = cur.cell; w->g[n] = tentative;
        orc_ent e = {hzero ? tentative : tentative + orc_dist(nr, nc, tr, tc), tentative, n};   /* :90 / dijkstra.py:89 */
        if (w->pos[n] < 0) { heap_push(&w->h, e); st[1]++; }                       /* :92-94 */
        else {                                                                     /* :96-100: replace + heapify */
          int i = w->pos[n]; w->h.e[i] = e; heap_up(&w->h, i); st[3]++;
        }
      }
    }
    (void)nb;
  }
  st[0] = steps; st[2] = w->h.max_n; st[5] = (w->h.n > 0) ? 2 : 1;
  return 0;                                                                        /* :101 */
}

/* variant 1: MPA._a_star, MPA.py:106-151.  No closed set; avoid nodes are
 * excluded from neighbour lists outright (no start/target exemption); a node
 * already in the open list keeps its old (f,g) entry when improved (the
 * branch at :144 is dead because :139 already overwrote g_score); popped
 * nodes may be re-pushed; expansion uses g_score[current], not the popped g. */
static int64_t astar_v1(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner,
                        int start, int target, const uint8_t* avoid, int32_t* out, int64_t cap, int64_t* st) {
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  st[5] = 1;
  if (start == target) { if (cap < 1) return -1; out[0] = start; st[5] = 0; return 1; } /* :107 */
  if (!orc_free(occ, R, C, sr, sc) || !orc_free(occ, R, C, tr, tc)) return 0;           /* :109-111 */
  w->epoch++; w->h.n = 0; w->h.max_n = 0;
  ws_touch(w, start);
  w->g[start] = 0.0;
  orc_ent e0 = {0 + orc_dist(sr, sc, tr, tc), 0.0, start};                               /* :113 */
  heap_push(&w->h, e0); st[1]++;
  int64_t max_steps = (int64_t)R * C * 2, steps = 0;                                     /* :118 */
  if (g_step_cap_override > 0 && g_step_cap_override < max_steps) max_steps = g_step_cap_override;
  while (w->h.n > 0 && steps < max_steps) {
    steps++;
    orc_ent cur = heap_pop(&w->h);                                                       /* :122 */
    if (cur.cell == target) {
      st[0] = steps; st[2] = w->h.max_n; st[5] = 0;
      return reconstruct(w, start, target, out, cap);
    }
    int r = cur.cell / C, c = cur.cell % C;
    int nm = allow_diag ? 8 : 4;
    for (int m = 0; m < nm; ++m) {                                                       /* MPA.py:79-99 */
      int nr = r + HM_DR[m], nc = c + HM_DC[m];
      if (!orc_free(occ, R, C, nr, nc)) continue;
      int n = nr * C + nc;
      if (avoid && avoid[n]) continue;
      if (m >= 4 && restrict_corner &&
          (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
      ws_touch(w, n); st[4]++;
      double tentative = w->g[cur.cell] + orc_dist(r, c, nr, nc);                        /* :135 */
      if (tentative < w->g[n]) {                                                         /* :137 (inf == not in g_score) */
        w->parent[n] = cur.cell; w->g[n] = tentative;
        if (w->pos[n] >= 0) { st[3]++; }                                                 /* :142-148: entry left as is */
        else {
          orc_ent e = {tentative + orc_dist(nr, nc, tr, tc), tentative, n};
          heap_push(&w->h, e); st[1]++;                                                  /* :150 */
        }
      }
    }
  }
  st[0] = steps; st[2] = w->h.max_n; st[5] = (w->h.n > 0) ? 2 : 1;
  return 0;
}

ORC_API int64_t orc_astar(orc_ws* w, const uint8_t* occ, int R, int C, int variant, int allow_diag,
                          int restrict_corner, int start, int target, const uint8_t* avoid,
                          int32_t* out, int64_t cap, int64_t* stats) {
  int64_t st[6] = {0, 0, 0, 0, 0, 0};
  /* variant 0 AStarSolver.solve, 1 MPA._a_star, 2 DijkstraSolver.solve */
  int64_t n = variant == 1 ? astar_v1(w, occ, R, C, allow_diag, restrict_corner, start, target, avoid, out, cap, st)
                           : astar_v0(w, occ, R, C, allow_diag, restrict_corner, start, target, avoid, out, cap, st, variant == 2);
  if (stats) memcpy(stats, st, sizeof(st));
  return n;
}

/* ------------------------------------------------------------------ */
/* a8/a9: path scoring                                                 */
/* ------------------------------------------------------------------ */
/* helper.py:67-80: min over ALL obstacles of the squared distance (literal,
 * O(L * n_obst)).  Used to validate the windowed form below. */
static double safety_literal(const uint8_t* occ, int R, int C, const int32_t* path, int64_t L, double min_safe) {
  if (L == 0) return 0.0;
  int any = 0; for (int i = 0; i < R * C && !any; ++i) any = occ[i] == 1;
  if (!any) return 0.0;
  double total = 0.0;
  for (int64_t i = 0; i < L; ++i) {
    int r = path[i] / C, c = path[i] % C; long best = -1;
    for (int o = 0; o < R * C; ++o) if (occ[o] == 1) {
      long dr = o / C - r, dc = o % C - c, d2 = dr * dr + dc * dc;
      if (best < 0 || d2 < best) best = d2;
    }
    double d = sqrt((double)best);
    if (d < min_safe) total += pow(min_safe - d, 2.0);   /* float ** int -> libm pow, helper.py:78 */
  }
  return total / (double)L;
}
/* Same value from a (2w+1)^2 window, w = ceil(min_safe): an obstacle outside
 * the window is farther than min_safe, so it can never be the contributing
 * minimum (SURVEY.md a8 [measured bit-equal]). */
static double safety_window(const uint8_t* occ, int R, int C, const int32_t* path, int64_t L, double min_safe) {
  if (L == 0) return 0.0;
  int w = (int)ceil(min_safe); if (w < 0) w = 0;
  double total = 0.0;
  for (int64_t i = 0; i < L; ++i) {
    int r = path[i] / C, c = path[i] % C; long best = -1;
    for (int dr = -w; dr <= w; ++dr) for (int dc = -w; dc <= w; ++dc) {
      int rr = r + dr, cc = c + dc;
      if (rr < 0 || rr >= R || cc < 0 || cc >= C || occ[rr * C + cc] != 1) continue;
      long d2 = (long)dr * dr + (long)dc * dc;
      if (best < 0 || d2 < best) best = d2;
    }
    if (best < 0) continue;
    double d = sqrt((double)best);
    if (d < min_safe) total += pow(min_safe - d, 2.0);
  }
  return total / (double)L;
}

/* variant 0: helper.calculate_path_stats (helper.py:98-113)
 * variant 1: MPA._calculate_path_stats (MPA.py:215-229; safety == 0.0, :164-173)
 * safety_mode: 0 windowed, 1 literal.  out = {length, turns, safety, diag, fitness} */
ORC_API void orc_score(const uint8_t* occ, int R, int C, const int32_t* path, int64_t L, int variant,
                       double w_turn, double w_safe, double min_safe, int restrict_policy, double diag_pen,
                       int safety_mode, double* out) {
  if (L == 0) { out[0] = INFINITY; out[1] = 0; out[2] = 0.0; out[3] = 0.0; out[4] = INFINITY; return; }
  double length = 0;                                              /* sum() starts at int 0 */
  for (int64_t i = 0; i + 1 < L; ++i)
    length = length + orc_dist(path[i] / C, path[i] % C, path[i + 1] / C, path[i + 1] % C);
  long turns = 0;                                                 /* helper.py:58-65 / MPA.py:202-212 */
  for (int64_t i = 0; i + 2 < L; ++i) {
    int dr1 = path[i + 1] / C - path[i] / C, dc1 = path[i + 1] % C - path[i] % C;
    int dr2 = path[i + 2] / C - path[i + 1] / C, dc2 = path[i + 2] % C - path[i + 1] % C;
    if (dr1 != dr2 || dc1 != dc2) turns++;
  }
  double safety = 0.0;
  if (variant == 0) safety = safety_mode ? safety_literal(occ, R, C, path, L, min_safe)
                                         : safety_window(occ, R, C, path, L, min_safe);
  double diag = 0.0;                                              /* helper.py:82-96 / MPA.py:176-199 */
  if (L >= 2 && restrict_policy) {
    for (int64_t i = 0; i + 1 < L; ++i) {
      int cr = path[i] / C, cc = path[i] % C, nr = path[i + 1] / C, nc = path[i + 1] % C;
      if (abs(nr - cr) == 1 && abs(nc - cc) == 1)
        if (!orc_free(occ, R, C, nr, cc) || !orc_free(occ, R, C, cr, nc)) diag += diag_pen;
    }
  }
  out[0] = length; out[1] = (double)turns; out[2] = safety; out[3] = diag;
  out[4] = length + w_turn * (double)turns + w_safe * safety + diag;   /* helper.py:112 / MPA.py:224-227 */
}

/* ------------------------------------------------------------------ */
/* a6/a7: chained waypoint decode                                      */
/* ------------------------------------------------------------------ */
/* pso.py:61,69-70: int(round(x)) is round-half-even, then clamp to the grid */
ORC_API void orc_pso_round(const double* pos, int W, int R, int C, int32_t* cells) {
  for (int i = 0; i < W; ++i) {
    long r = (long)nearbyint(pos[2 * i]), c = (long)nearbyint(pos[2 * i + 1]);
    if (r > R - 1) r = R - 1; if (r < 0) r = 0;
    if (c > C - 1) c = C - 1; if (c < 0) c = 0;
    cells[i] = (int32_t)(r * C + c);
  }
}
/* GASolver._reconstruct_path_from_chromosome ga_solver.py:58-93 ==
 * PSOSolver._reconstruct_path_from_position pso.py:56-94 after rounding.
 * visited: caller scratch of R*C bytes.  seg_stats accumulates A* counters. */
ORC_API int64_t orc_decode(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner,
                           int start, int target, const int32_t* wps, int W, uint8_t* visited,
                           int32_t* out, int64_t cap, int64_t* stats_sum) {
  int64_t st[6], n = 0;
  if (stats_sum) memset(stats_sum, 0, sizeof(int64_t) * 6);
  if (W == 0) {                                                   /* :59-61 */
    n = orc_astar(w, occ, R, C, 0, allow_diag, restrict_corner, start, target, NULL, out, cap, st);
    if (stats_sum) memcpy(stats_sum, st, sizeof(st));
    return n;
  }
  memset(visited, 0, (size_t)R * C);
  if (cap < 1) return -1;
  out[0] = start; n = 1; visited[start] = 1;                      /* :63-65 */
  int cur = start;
  int32_t* seg = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * C);
  for (int k = 0; k <= W; ++k) {
    int goal = (k < W) ? wps[k] : target;
    /* nodes_to_avoid = nodes_in_path_so_far - {cur, goal}; astar.py:55-56 removes start/target again */
    int64_t m = orc_astar(w, occ, R, C, 0, allow_diag, restrict_corner, cur, goal, visited, seg, (int64_t)R * C, st);
    if (stats_sum) { for (int i = 0; i < 5; ++i) stats_sum[i] += st[i]; if (st[2] > stats_sum[2]) stats_sum[2] = st[2]; }
    if (m == 0 || (m == 1 && cur != goal)) { free(seg); return 0; }     /* :74 / :85 */
    if (n + m - 1 > cap) { free(seg); return -1; }
    for (int64_t i = 1; i < m; ++i) { out[n++] = seg[i]; visited[seg[i]] = 1; }  /* :75-76 */
    cur = goal;
  }
  free(seg);
  /* drop consecutive duplicates :90-93 */
  int64_t u = 1;
  for (int64_t i = 1; i < n; ++i) if (out[i] != out[i - 1]) out[u++] = out[i];
  return u;
}

/* ------------------------------------------------------------------ */
/* a10: PSO velocity / position update                                 */
/* ------------------------------------------------------------------ */
/* pso.py:183-203 for particles [0,n).  Draw order per waypoint:
 * r(c1,row), r(c2,row), r(c1,col), r(c2,col).  Stream (seed, dom=3, it, agent0+p). */
ORC_API void orc_pso_update(int n, int W, double w, double c1, double c2, double max_vel, int R, int C,
                            double* pos, double* vel, const double* pbest, const double* gbest,
                            uint64_t seed, uint64_t it, uint64_t agent0) {
  for (int p = 0; p < n; ++p) {
    orc_rng g; orc_rng_init(&g, seed, 3, it, agent0 + (uint64_t)p);
    for (int d = 0; d < W; ++d) for (int ax = 0; ax < 2; ++ax) {
      int i = (p * W + d) * 2 + ax;
      double hi = (ax == 0) ? (double)(R - 1) : (double)(C - 1);
      double r1 = orc_rng_random(&g);                  /* C leaves call order inside one expression open */
      double r2 = orc_rng_random(&g);
      double v = w * vel[i] + c1 * r1 * (pbest[i] - pos[i]) + c2 * r2 * (gbest[d * 2 + ax] - pos[i]);
      v = fmin(fmax(v, -max_vel), max_vel);            /* np.clip :192-193 */
      double x = pos[i] + v;
      x = fmin(fmax(x, 0.0), hi);                      /* np.clip :201-202 */
      vel[i] = v; pos[i] = x;
    }
  }
}

/* ------------------------------------------------------------------ */
/* a12-a15: MAACO                                                      */
/* ------------------------------------------------------------------ */
typedef struct {
  double alpha, beta, rho, Q, a_turn, wh_max, wh_min, k_h, q0_initial, C0;
  int num_iterations;
} orc_maaco_params;

/* MAACO.py:98 */
static const int AM_DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
static const int AM_DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

/* MAACO.py:58-84 */
ORC_API void orc_maaco_init_pheromone(const uint8_t* occ, int R, int C, int start, int target, double C0, double* tau) {
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  double dsT = orc_dist(sr, sc, tr, tc); if (dsT < 1e-9) dsT = 1e-9;     /* :43-45 */
  for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) {
    double* t = &tau[r * C + c];
    if (occ[r * C + c] == 1) { *t = 1e-9; continue; }
    double dsi = orc_dist(sr, sc, r, c), diT = orc_dist(r, c, tr, tc), den = dsi + diT, factor;
    if (den < 1e-9) factor = (orc_dist(r, c, sr, sc) < 1e-6 || orc_dist(r, c, tr, tc) < 1e-6) ? 1.0 : 0.1;
    else factor = dsT / den;
    *t = factor * C0; if (*t < 1e-9) *t = 1e-9;
  }
}
/* MAACO.py:86-91 */
ORC_API void orc_maaco_dist_to_target(int R, int C, int target, double* dist) {
  for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) dist[r * C + c] = orc_dist(r, c, target / C, target % C);
}
/* MAACO.py:212-226 */
ORC_API double orc_maaco_q0(int it, int K, double q0_initial) {
  double Kt = K, k = it, k0 = 0.7 * Kt, q;
  if (k < k0) q = (fabs(Kt - k0) < 1e-6) ? q0_initial : ((Kt - k) / Kt) * q0_initial;
  else {
    double q_at = ((Kt - k0) / Kt) * q0_initial;
    q = q_at + ((k - k0) / (Kt - k0 + 1e-9)) * (q0_initial * (1 - (Kt - k0) / Kt) / 2.0);
  }
  return fmin(fmax(q, 0.01), 0.99);
}
/* MAACO.py:197-210 */
static double maaco_eta(const orc_maaco_params* P, int start, int C, double dsT, const double* dist, int cand, int turn) {
  double dsj = orc_dist(start / C, start % C, cand / C, cand % C);
  double djT = dist[cand], h;
  if (dsT < 1e-9) h = P->wh_min;
  else h = P->wh_max - (P->wh_max - P->wh_min) * exp(-P->k_h * djT / dsT);
  double g = 1.0 - h;
  double den = g * dsj + h * djT + P->a_turn * (double)turn;
  den = den > 1e-9 ? den : 1e-9;                                   /* max(den, 1e-9) */
  return 1.0 / den;
}

/* MAACO._construct_ant_solution_maaco, MAACO.py:278-302 (+ :122-181 filter,
 * :228-262 selection).  tabu: caller scratch R*C bytes.  Stream
 * (seed, dom=1, it, ant).  np.random.choice(n,p) (:259) is restated as one
 * random() draw u and searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right')
 * (SURVEY.md 5.1 [measured]); the harness patches numpy.random.choice to the
 * same definition on the per-agent stream.  counters: [0]=steps,
 * [1]=candidates summed, [2]=greedy steps, [3]=roulette steps, [4]=uniform-fallback steps.
 * Returns path cells (0 => ant failed: [], inf, inf). */
ORC_API int64_t orc_maaco_walk(const uint8_t* occ, int R, int C, int start, int target,
                               const orc_maaco_params* P, const double* tau, const double* dist,
                               int it, uint64_t seed, uint64_t ant, uint8_t* tabu,
                               int32_t* out, int64_t cap, double* out_len, int64_t* out_turns, int64_t* counters) {
  orc_rng g; orc_rng_init(&g, seed, 1, (uint64_t)it, ant);
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  double dsT = orc_dist(sr, sc, tr, tc); if (dsT < 1e-9) dsT = 1e-9;
  int d_r_ST = tr - sr, d_c_ST = tc - sc;
  memset(tabu, 0, (size_t)R * C);
  int cur = start; int64_t n = 0; double plen = 0.0;
  if (cap < 1) return -1;
  out[n++] = start; tabu[start] = 1;
  int64_t max_steps = (int64_t)R * C * 2, steps = 0;
  double q0 = orc_maaco_q0(it, P->num_iterations, P->q0_initial);
  int64_t cnt[5] = {0, 0, 0, 0, 0};
  while (cur != target && steps < max_steps) {
    int cr = cur / C, cc = cur % C, cand[8], nc_ = 0;
    /* --- _apply_orientation_heuristic_filter :122-181 --- */
    for (int strat = 0; strat < 3 && nc_ == 0; ++strat) {
      for (int m = 0; m < 8; ++m) {
        int dr = AM_DR[m], dc = AM_DC[m], nr = cr + dr, nc = cc + dc;
        if (!(nr >= 0 && nr < R && nc >= 0 && nc < C && occ[nr * C + nc] != 1 && !tabu[nr * C + nc])) continue; /* :93-95 */
        if (dr != 0 && dc != 0 && (occ[nr * C + cc] == 1 || occ[cr * C + nc] == 1)) continue;                   /* :100-120 */
        if (strat < 2) {
          int vr = strat == 0 ? d_r_ST : tr - cr, vc = strat == 0 ? d_c_ST : tc - cc;
          if ((vc > 0 && dc < 0) || (vc < 0 && dc > 0) || (vr > 0 && dr < 0) || (vr < 0 && dr > 0)) continue;
        }
        cand[nc_++] = nr * C + nc;
      }
    }
    if (nc_ == 0) { if (counters) memcpy(counters, cnt, sizeof(cnt)); return 0; }   /* :287-288 */
    cnt[1] += nc_;
    /* --- _select_next_node_with_MAACO_rules :228-262 --- */
    double q = orc_rng_random(&g);                                                    /* :232 */
    double attr[8];
    int pr = -1, pc = -1, have_prev = n >= 2;
    if (have_prev) { pr = cr - out[n - 2] / C; pc = cc - out[n - 2] % C; }
    for (int i = 0; i < nc_; ++i) {
      int turn = 0;                                                                   /* :184-195 */
      if (have_prev) { int dr = cand[i] / C - cr, dc = cand[i] % C - cc; turn = (pr != dr || pc != dc); }
      double eta = maaco_eta(P, start, C, dsT, dist, cand[i], turn);
      attr[i] = pow(tau[cand[i]], P->alpha) * pow(eta, P->beta);                       /* :238 */
    }
    int next;
    if (q <= q0) {                                                                    /* :241-250 */
      double mx = -1; int best[8], nb = 0;
      for (int i = 0; i < nc_; ++i) {
        if (attr[i] > mx) { mx = attr[i]; nb = 0; best[nb++] = cand[i]; }
        else if (fabs(attr[i] - mx) < 1e-9) best[nb++] = cand[i];
      }
      if (nb == 0) { if (counters) memcpy(counters, cnt, sizeof(cnt)); return 0; }     /* None -> :291-292 */
      next = best[orc_rng_randbelow(&g, (uint64_t)nb)];
      cnt[2]++;
    } else {
      double sum = 0;                                                                 /* sum() from int 0 */
      for (int i = 0; i < nc_; ++i) sum = sum + attr[i];
      if (sum < 1e-9) { next = cand[orc_rng_randbelow(&g, (uint64_t)nc_)]; cnt[4]++; }   /* :253-254 */
      else {
        double p[8], ps = 0;
        for (int i = 0; i < nc_; ++i) p[i] = attr[i] / sum;
        for (int i = 0; i < nc_; ++i) ps = ps + p[i];
        if (fabs(ps - 1.0) > 1e-6) for (int i = 0; i < nc_; ++i) p[i] = p[i] / ps;    /* :257-258 */
        double u = orc_rng_random(&g), cdf[8], acc = 0;                               /* :259 */
        for (int i = 0; i < nc_; ++i) { acc = (i == 0) ? p[0] : acc + p[i]; cdf[i] = acc; }
        int idx = 0;
        for (int i = 0; i < nc_; ++i) { double ci = cdf[i] / cdf[nc_ - 1]; if (ci <= u) idx = i + 1; }
        if (idx > nc_ - 1) idx = nc_ - 1;
        next = cand[idx]; cnt[3]++;
      }
    }
    plen += orc_dist(cr, cc, next / C, next % C);                                     /* :293 */
    cur = next;
    if (n >= cap) return -1;
    out[n++] = cur; tabu[cur] = 1; steps++; cnt[0]++;
  }
  if (counters) memcpy(counters, cnt, sizeof(cnt));
  if (cur != target) return 0;                                                        /* :301-302 */
  int64_t turns = 0;                                                                  /* :264-276 */
  for (int64_t i = 0; i + 2 < n; ++i) {
    int dr1 = out[i + 1] / C - out[i] / C, dc1 = out[i + 1] % C - out[i] % C;
    int dr2 = out[i + 2] / C - out[i + 1] / C, dc2 = out[i + 2] % C - out[i + 1] % C;
    if (dr1 != dr2 || dc1 != dc2) turns++;
  }
  *out_len = plen; *out_turns = turns;
  return n;
}

/* MAACO._update_pheromone_trails_maaco, MAACO.py:304-332.  paths in CSR
 * (offsets[n_ants+1]; an empty range is a failed ant), lens[n_ants].
 * best_len_overall = self.best_path_length_overall at call time (inf allowed). */
ORC_API void orc_maaco_update(const uint8_t* occ, int R, int C, double rho, double Q, double* tau,
                              int n_ants, const int64_t* offsets, const int32_t* cells, const double* lens,
                              double best_len_overall) {
  int RC = R * C;
  for (int i = 0; i < RC; ++i) tau[i] = tau[i] * (1.0 - rho);                         /* :305 */
  for (int a = 0; a < n_ants; ++a) {                                                  /* :306-311 */
    int64_t b = offsets[a], e = offsets[a + 1];
    if (!(lens[a] != INFINITY && e > b && lens[a] > 1e-6)) continue;
    double dep = Q / lens[a];
    for (int64_t i = b; i < e; ++i) if (occ[cells[i]] != 1) tau[cells[i]] += dep;
  }
  double bl = best_len_overall;                                                       /* :312-316 */
  if (bl == INFINITY) bl = (double)(R + C);
  if (bl < 1e-6) bl = 1e-6;
  double tmax = (1.0 / (1.0 - rho)) * (1.0 / bl);                                     /* :317 */
  int mx = C > R ? C : R; if (mx < 1) mx = 1;
  double tmin = tmax / (2.0 * mx);                                                    /* :323 */
  for (int i = 0; i < RC; ++i) {
    if (occ[i] == 1) tau[i] = 1e-9;                                                   /* :332 */
    else tau[i] = fmin(fmax(tau[i], tmin), tmax);                                     /* :327-331 */
  }
}

/* ------------------------------------------------------------------ */
/* a16-a17: MPA                                                        */
/* ------------------------------------------------------------------ */
static inline long py_round(double x) { return (long)nearbyint(x); }   /* round(): half-even */
static inline int clampi(long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }

/* MPA._get_levy_target_node MPA.py:250-264.  sigma (the Mantegna constant,
 * :251-253) is computed by the caller with Python's math.gamma. */
ORC_API int orc_mpa_levy_target(orc_rng* g, int R, int C, int cur, double scale, double levy_beta, double sigma) {
  double u = orc_rng_normalvariate(g, 0, sigma);
  double v = orc_rng_normalvariate(g, 0, 1);
  if (fabs(v) < 1e-9) v = 1e-9;
  double step = 0.05 * u / pow(fabs(v), 1 / levy_beta) * scale;
  double mx = (R > C ? R : C) * 0.5;
  step = fmin(fmax(step, -mx), mx);
  double angle = orc_rng_uniform(g, 0, 2 * M_PI);
  long dr = py_round(step * sin(angle)), dc = py_round(step * cos(angle));
  int r = clampi(cur / C + dr, 0, R - 1), c = clampi(cur % C + dc, 0, C - 1);
  return r * C + c;
}
/* MPA._get_brownian_target_node MPA.py:266-282.  elite_node < 0 == None. */
ORC_API int orc_mpa_brownian_target(orc_rng* g, int R, int C, int cur, int elite_node, double scale) {
  int cr = cur / C, cc = cur % C; long tr_, tc_;
  if (orc_rng_random(g) < 0.7 && elite_node >= 0) {
    int dr = elite_node / C - cr, dc = elite_node % C - cc;
    double dist = sqrt((double)((long)dr * dr + (long)dc * dc));
    if (dist > 1e-6) {
      double fac = fabs(orc_rng_normalvariate(g, 0, 1));
      long k = py_round(scale * fac * 5); if (k < 1) k = 1;
      double ms = dist < (double)k ? dist : (double)k;          /* min(dist, max(1,int)) */
      tr_ = cr + py_round((double)dr / dist * ms);
      tc_ = cc + py_round((double)dc / dist * ms);
    } else return elite_node;
  } else {
    long m = py_round((R > C ? R : C) * 0.1 * scale * fabs(orc_rng_normalvariate(g, 0, 1))); if (m < 1) m = 1;
    long dr = orc_rng_randint(g, -m, m), dc = orc_rng_randint(g, -m, m);
    tr_ = cr + dr; tc_ = cc + dc;
  }
  return clampi(tr_, 0, R - 1) * C + clampi(tc_, 0, C - 1);
}

/* n proposals from the keyed streams (seed, DOM_MPA = 2, 0, i): the checker of pf_selftest_mpa_targets */
ORC_API void orc_mpa_targets_batch(uint64_t seed, int64_t n, int is_levy, int R, int C, const int32_t* cur, const int32_t* elite,
                                   double scale, double levy_beta, double sigma, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    orc_rng g; orc_rng_init(&g, seed, 2, 0, (uint64_t)i);
    out[i] = is_levy ? orc_mpa_levy_target(&g, R, C, cur[i], scale, levy_beta, sigma)
                     : orc_mpa_brownian_target(&g, R, C, cur[i], elite[i], scale);
  }
}

/* MPA._reconstruct_path_segment MPA.py:284-318.  The caller has positioned
 * `g` (the per-predator stream).  avoid: scratch R*C bytes.  Returns the new
 * path length in cells, or -2 when the reference falls back to the ORIGINAL
 * path (its :286-287 and :316-317 branches), -1 on cap overflow.
 * target_cell_out: the proposed intermediate cell (for fixtures). */
ORC_API int64_t orc_mpa_rebuild(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner,
                                int start, int target, const int32_t* path, int64_t L,
                                const int32_t* elite, int64_t Le, int64_t idx, int is_levy, double scale,
                                double levy_beta, double sigma, orc_rng* g, uint8_t* avoid,
                                int32_t* out, int64_t cap, int* target_cell_out, int64_t* stats_sum) {
  int64_t st[6];
  if (stats_sum) memset(stats_sum, 0, sizeof(int64_t) * 6);
  if (target_cell_out) *target_cell_out = -1;
  if (L == 0 || idx >= L - 1) return -2;                                        /* :286-287 */
  int cur = path[idx];
  memset(avoid, 0, (size_t)R * C);
  for (int64_t i = 0; i < idx; ++i) avoid[path[i]] = 1;                         /* set(prefix[:-1]) :290 */
  int inter;
  if (is_levy) inter = orc_mpa_levy_target(g, R, C, cur, scale, levy_beta, sigma);
  else {
    int en = -1;
    if (Le > 0) en = elite[orc_rng_randbelow(g, (uint64_t)Le)];                 /* random.choice :248 */
    inter = orc_mpa_brownian_target(g, R, C, cur, en, scale);
  }
  if (target_cell_out) *target_cell_out = inter;
  if (idx + 1 > cap) return -1;
  int64_t n = 0;
  for (int64_t i = 0; i <= idx; ++i) out[n++] = path[i];                        /* :296 */
  int32_t* seg = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * C);
  int astart = cur;
  if (orc_free(occ, R, C, inter / C, inter % C) && inter != astart) {           /* :298 */
    int64_t m = orc_astar(w, occ, R, C, 1, allow_diag, restrict_corner, astart, inter, avoid, seg, (int64_t)R * C, st);
    if (stats_sum) for (int i = 0; i < 5; ++i) stats_sum[i] += st[i];
    if (m > 1) {                                                                /* :300-305 */
      if (n + m - 1 > cap) { free(seg); return -1; }
      for (int64_t i = 1; i < m; ++i) { out[n++] = seg[i]; avoid[seg[i]] = 1; }
      astart = inter;
    }
  }
  if (astart != target) {                                                       /* :306-309 */
    int64_t m = orc_astar(w, occ, R, C, 1, allow_diag, restrict_corner, astart, target, avoid, seg, (int64_t)R * C, st);
    if (stats_sum) for (int i = 0; i < 5; ++i) stats_sum[i] += st[i];
    if (m > 1) {
      if (n + m - 1 > cap) { free(seg); return -1; }
      for (int64_t i = 1; i < m; ++i) out[n++] = seg[i];
    }
  }
  free(seg);
  int64_t u = 1;                                                                /* :310-315 */
  for (int64_t i = 1; i < n; ++i) if (out[i] != out[i - 1]) out[u++] = out[i];
  if (out[0] != start || out[u - 1] != target) return -2;                       /* :316-317 */
  return u;
}
