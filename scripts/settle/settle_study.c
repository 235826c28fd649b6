/*
 * settle_study.c -- STUDY TOOL (CPU), not product code and not the oracle.
 *
 * Question: is the path AStarSolver.solve returns (astar.py:33-101, closed set + working
 * decrease-key) a function of a *settled g-field* that can be computed without the sequential pop order?
 *
 * Model checked here.  Let g be the fixpoint of  g(x) = min over expanded neighbours p of fl(g(p) + c(p,x))  on the
 * region { x : key(x) <= key(goal) }, key(x) = (fl(g(x) + h(x)), g(x), cell(x)), computed by ANY label-correcting
 * schedule (this file uses a lazy heap; a GPU would use buckets).  For a node x let P(x) be the neighbours whose
 * offer equals g(x).  x is REGULAR if some p in P(x) has key(p) < key(x).  Theorem: if every region
 * node is regular the sequential loop pops the region in key order, closes every node with g(x), and
 * came_from[x] = the member of P(x) with the smallest key.  Double rounding (fl(fl(g+c)+h') vs fl(g+h)) makes a node
 * on a ray towards the goal "delayed" now and then: key(x) < key(p) for its only parent.  A delayed node is harmless
 * when it and its parent are adjacent in key order (chain rule below); anything else is reported as "needs the
 * sequential engine".
 *
 * The file includes the oracle's C source to run the reference restatement next to the model on the same inputs.
 */
#include "../../oracle/pf_oracle.c"

typedef struct { double f, g; int32_t cell; } sk;
static inline int sk_lt(double f1, double g1, int c1, double f2, double g2, int c2) {
  if (f1 != f2) return f1 < f2;
  if (g1 != g2) return g1 < g2;
  return c1 < c2;
}
typedef struct { double f; int32_t cell; double g; } lz;   /* lazy heap entry */
static void lz_push(lz** h, int* n, int* cap, lz v) {
  if (*n == *cap) { *cap *= 2; *h = (lz*)realloc(*h, sizeof(lz) * (size_t)*cap); }
  int i = (*n)++;
  while (i > 0) { int p = (i - 1) >> 1; if (!((*h)[p].f > v.f)) break; (*h)[i] = (*h)[p]; i = p; }
  (*h)[i] = v;
}
static lz lz_pop(lz* h, int* n) {
  lz top = h[0]; lz v = h[--(*n)]; int i = 0;
  for (;;) { int l = 2 * i + 1; if (l >= *n) break; if (l + 1 < *n && h[l + 1].f < h[l].f) l++; if (!(h[l].f < v.f)) break; h[i] = h[l]; i = l; }
  if (*n > 0) h[i] = v;
  return top;
}

static int cmp_sk(const void* a, const void* b) {
  const sk* x = (const sk*)a; const sk* y = (const sk*)b;
  return sk_lt(x->f, x->g, x->cell, y->f, y->g, y->cell) ? -1 : (sk_lt(y->f, y->g, y->cell, x->f, x->g, x->cell) ? 1 : 0);
}

/* out stats: [0] region size, [1] delayed nodes, [2] delayed nodes not covered by the chain rule (=> fallback),
 * [3] expansions done by the label-correcting pass, [4] nodes with >= 2 parents in P(x), [5] fixpoint violations (bug),
 * [6] status 0 ok / 1 infeasible, [7] path nodes that are delayed */
ORC_API int64_t settle_v0(const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner, int start, int target,
                          const uint8_t* avoid, int hzero, int32_t* out, int64_t cap, int64_t* stats, double* g_out) {
  const int RC = R * C;
  memset(stats, 0, sizeof(int64_t) * 10);            /* [8] = key rank of the first delayed node (region size if none), [9] = 0 */
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  stats[6] = 1;
  if (!orc_free(occ, R, C, sr, sc) || !orc_free(occ, R, C, tr, tc)) return 0;
  if (start == target) { out[0] = start; stats[6] = 0; return 1; }
  double* g = g_out;
  for (int i = 0; i < RC; ++i) g[i] = INFINITY;
  int hn = 0, hcap = 1024; lz* hp = (lz*)malloc(sizeof(lz) * hcap);
  g[start] = 0.0;
  lz e0 = {hzero ? 0.0 : orc_dist(sr, sc, tr, tc), start, 0.0};
  lz_push(&hp, &hn, &hcap, e0);
  const int nm = allow_diag ? 8 : 4;
#define BLOCKED(n) (avoid && avoid[n] && (n) != start && (n) != target)
  while (hn > 0) {
    lz cur = lz_pop(hp, &hn);
    if (cur.g != g[cur.cell]) continue;                 /* superseded */
    if (cur.cell == target) continue;                   /* the goal is never expanded */
    if (cur.f > g[target]) break;                       /* every remaining f is above the goal's */
    stats[3]++;
    int r = cur.cell / C, c = cur.cell % C;
    for (int m = 0; m < nm; ++m) {
      int nr = r + HM_DR[m], nc = c + HM_DC[m];
      if (!orc_free(occ, R, C, nr, nc)) continue;
      int n = nr * C + nc;
      if (BLOCKED(n)) continue;
      if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
      double t = cur.g + orc_dist(r, c, nr, nc);
      if (t < g[n]) {
        g[n] = t;
        lz e = {hzero ? t : t + orc_dist(nr, nc, tr, tc), n, t};
        lz_push(&hp, &hn, &hcap, e);
      }
    }
  }
  free(hp);
  if (g[target] == INFINITY) return 0;
  const double F = hzero ? g[target] : g[target] + 0.0;
  /* region = labelled nodes with key <= key(goal); the goal's key is (F, g, cell) with h = 0 */
#define KF(n) (hzero ? g[n] : g[n] + orc_dist((n) / C, (n) % C, tr, tc))
#define INREG(n) (g[n] != INFINITY && ((n) == target || sk_lt(KF(n), g[n], (n), F, g[target], target)))
  int nreg = 0;
  sk* srt = (sk*)malloc(sizeof(sk) * (size_t)RC);
  int32_t* pbest = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);   /* chosen parent */
  uint8_t* delayed = (uint8_t*)calloc(RC, 1);
  for (int x = 0; x < RC; ++x) {
    pbest[x] = -1;
    if (occ[x] == 1 || !INREG(x)) continue;
    srt[nreg].f = KF(x); srt[nreg].g = g[x]; srt[nreg].cell = x; nreg++;
    if (x == start) continue;
    int r = x / C, c = x % C;
    double xf = KF(x);
    int best_reg = -1, best_any = -1, np = 0;
    double brf = 0, brg = 0, baf = 0, bag = 0;
    for (int m = 0; m < nm; ++m) {                      /* moves are symmetric: scan the reverse moves */
      int pr = r + HM_DR[m], pc = c + HM_DC[m];
      if (!orc_free(occ, R, C, pr, pc)) continue;
      int p = pr * C + pc;
      if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
      if (BLOCKED(p) || p == target || !INREG(p)) continue;       /* p must be an expanded node */
      double off = g[p] + orc_dist(pr, pc, r, c);
      if (off < g[x]) stats[5]++;
      if (off != g[x]) continue;
      np++;
      double pf_ = KF(p);
      if (best_any < 0 || sk_lt(pf_, g[p], p, baf, bag, best_any)) { best_any = p; baf = pf_; bag = g[p]; }
      if (sk_lt(pf_, g[p], p, xf, g[x], x) && (best_reg < 0 || sk_lt(pf_, g[p], p, brf, brg, best_reg))) { best_reg = p; brf = pf_; brg = g[p]; }
    }
    if (np >= 2) stats[4]++;
    if (best_any < 0) { stats[2]++; stats[1]++; delayed[x] = 2; continue; }   /* its label came from a node outside the final region */
    if (best_reg >= 0) pbest[x] = best_reg; else { pbest[x] = best_any; delayed[x] = 1; stats[1]++; }
  }
  stats[0] = nreg; stats[8] = nreg;
  /* chain rule for delayed nodes: in key order the nodes strictly between x and its parent must all be delayed nodes
   * whose parent chain leads to that same parent (x, x', ... pop right after it, in reverse), and no worse offer may
   * let x be popped before its parent is (S3). */
  if (stats[1] > 0) {
    qsort(srt, nreg, sizeof(sk), cmp_sk);
    int32_t* rank = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);
    for (int i = 0; i < nreg; ++i) rank[srt[i].cell] = i;
    for (int i = 0; i < nreg; ++i) if (delayed[srt[i].cell]) { stats[8] = i; break; }
    for (int i = 0; i < nreg; ++i) {
      int x = srt[i].cell;
      if (delayed[x] != 1) continue;
      int root = pbest[x];
      while (root >= 0 && delayed[root] == 1) root = pbest[root];         /* the first non-delayed ancestor */
      if (root < 0 || delayed[root]) { stats[2]++; continue; }
      int bad = rank[root] < i;                          /* (cannot happen: a delayed node's parents are all later) */
      for (int j = i + 1; j < rank[root] && !bad; ++j) {
        int y = srt[j].cell, a = y;
        if (!delayed[y]) { bad = 1; break; }
        while (a >= 0 && delayed[a] == 1) a = pbest[a];
        if (a != root) bad = 1;
      }
      /* S3: an earlier, worse offer must not give x a key below the root's */
      if (!bad) {
        int r = x / C, c = x % C;
        double rf = KF(root);
        for (int m = 0; m < nm && !bad; ++m) {
          int pr = r + HM_DR[m], pc = c + HM_DC[m];
          if (!orc_free(occ, R, C, pr, pc)) continue;
          int p = pr * C + pc;
          if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
          if (BLOCKED(p) || p == target || !INREG(p) || rank[p] >= rank[root]) continue;
          double off = g[p] + orc_dist(pr, pc, r, c);
          if (off == g[x]) continue;
          double of_ = hzero ? off : off + orc_dist(r, c, tr, tc);
          if (sk_lt(of_, off, x, rf, g[root], root)) bad = 1;
        }
      }
      if (bad) stats[2]++;
    }
    free(rank);
  }
  /* the path: parents from the goal */
  int64_t n = 0; int t = target;
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);
  while (t != start && t >= 0 && n < RC) { if (delayed[t]) stats[7]++; tmp[n++] = t; t = pbest[t]; }
  int64_t ret;
  if (t != start) { stats[2]++; ret = 0; }
  else {
    tmp[n++] = start;
    if (n > cap) ret = -1; else { for (int64_t i = 0; i < n; ++i) out[i] = tmp[n - 1 - i]; ret = n; }
  }
  free(tmp); free(srt); free(pbest); free(delayed);
  stats[6] = 0;
  return ret;
}

/* the oracle's closed-set search, also returning its final g labels of the closed (popped) nodes (inf elsewhere) */
ORC_API int64_t ref_v0_labels(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner, int start, int target,
                              const uint8_t* avoid, int hzero, int32_t* out, int64_t cap, int64_t* st, double* g_out) {
  int64_t s6[6] = {0, 0, 0, 0, 0, 0};
  int64_t n = astar_v0(w, occ, R, C, allow_diag, restrict_corner, start, target, avoid, out, cap, s6, hzero);
  memcpy(st, s6, sizeof(s6));
  for (int i = 0; i < R * C; ++i) g_out[i] = (w->stamp[i] == w->epoch && w->closed[i]) ? w->g[i] : INFINITY;
  return n;
}

/* MPA._a_star next to the same model: how often is its path the model's?  (It need not be: stale keys, re-pops.) */
ORC_API int64_t ref_v1(orc_ws* w, const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner, int start, int target,
                       const uint8_t* avoid, int32_t* out, int64_t cap, int64_t* st) {
  int64_t s6[6] = {0, 0, 0, 0, 0, 0};
  int64_t n = astar_v1(w, occ, R, C, allow_diag, restrict_corner, start, target, avoid, out, cap, s6);
  memcpy(st, s6, sizeof(s6));
  return n;
}

/* ------------------------------------------------------------------------------------------------------------------
 * T-order certificate (second half).  A delayed node x (every argmin parent has a LARGER key) waits in the
 * open list with a worse label until its earliest parent p* pops, then pops right after it (its final key is below
 * everything else in the heap).  So the sequential pop time of a node is
 *     T(x) = (K(x), 0)                       if some argmin parent has T(p) < K(x)          "on time"
 *          = (base(p*), depth(p*) + 1)       otherwise, p* = the argmin parent with the smallest T   "delayed"
 * provided x is not popped EARLIER with a worse label (I1: every offer it received before p* pops keys it above the
 * base) and the delayed nodes hanging off one base form a simple chain (one node per depth).  Under these conditions:
 * labels = the fixpoint, came_from[x] = the argmin parent with the smallest T, expanded nodes = { T(x) < T(goal) }.
 * out stats: [0] region size, [1] delayed nodes, [2] reasons for giving up (0 = certified), [3] expansions,
 * [6] status, [7] delayed nodes on the path */
typedef struct { double bf, bg; int32_t bc; int32_t d; } tkey;   /* base key (f, g, cell) + depth */
static inline int t_lt(const tkey* a, const tkey* b) {
  if (a->bc == b->bc) return a->d < b->d;
  return sk_lt(a->bf, a->bg, a->bc, b->bf, b->bg, b->bc);
}
ORC_API int64_t settle_v0_T(const uint8_t* occ, int R, int C, int allow_diag, int restrict_corner, int start, int target,
                            const uint8_t* avoid, int hzero, int32_t* out, int64_t cap, int64_t* stats, double* g_out) {
  const int RC = R * C;
  memset(stats, 0, sizeof(int64_t) * 8);
  int sr = start / C, sc = start % C, tr = target / C, tc = target % C;
  stats[6] = 1;
  if (!orc_free(occ, R, C, sr, sc) || !orc_free(occ, R, C, tr, tc)) return 0;
  if (start == target) { out[0] = start; stats[6] = 0; return 1; }
  double* g = g_out;
  for (int i = 0; i < RC; ++i) g[i] = INFINITY;
  int hn = 0, hcap = 1024; lz* hp = (lz*)malloc(sizeof(lz) * hcap);
  g[start] = 0.0;
  lz e0 = {hzero ? 0.0 : orc_dist(sr, sc, tr, tc), start, 0.0};
  lz_push(&hp, &hn, &hcap, e0);
  const int nm = allow_diag ? 8 : 4;
  /* label-correcting pass with SLACK: everything that could be popped before a delayed goal is expanded too */
  while (hn > 0) {
    lz cur = lz_pop(hp, &hn);
    if (cur.g != g[cur.cell]) continue;
    if (cur.cell == target) continue;
    if (cur.f > g[target] * (1.0 + 1e-12)) break;
    stats[3]++;
    int r = cur.cell / C, c = cur.cell % C;
    for (int m = 0; m < nm; ++m) {
      int nr = r + HM_DR[m], nc = c + HM_DC[m];
      if (!orc_free(occ, R, C, nr, nc)) continue;
      int n = nr * C + nc;
      if (BLOCKED(n)) continue;
      if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
      double t = cur.g + orc_dist(r, c, nr, nc);
      if (t < g[n]) { g[n] = t; lz e = {hzero ? t : t + orc_dist(nr, nc, tr, tc), n, t}; lz_push(&hp, &hn, &hcap, e); }
    }
  }
  free(hp);
  if (g[target] == INFINITY) return 0;
  const double Fs = g[target] * (1.0 + 1e-12);                    /* nodes with f <= Fs were expanded by the pass */
#define EXPD(n) (g[n] != INFINITY && (n) != target && KF(n) <= Fs)
  tkey* T = (tkey*)malloc(sizeof(tkey) * (size_t)RC);
  int32_t* par = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);
  int32_t* list = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);
  int nl = 0;
  for (int x = 0; x < RC; ++x) {
    par[x] = -1;
    if (occ[x] == 1 || g[x] == INFINITY) continue;
    if (x != target && !(KF(x) <= Fs)) continue;
    T[x].bf = KF(x); T[x].bg = g[x]; T[x].bc = x; T[x].d = 0;
    list[nl++] = x;
  }
  int giveup = 0, rounds = 0, changed = 1;
  while (changed && !giveup) {
    changed = 0;
    if (++rounds > 12) { giveup = 1; break; }
    for (int li = 0; li < nl; ++li) {
      int x = list[li];
      if (x == start) continue;
      int r = x / C, c = x % C;
      int best = -1;
      for (int m = 0; m < nm; ++m) {
        int pr = r + HM_DR[m], pc = c + HM_DC[m];
        if (!orc_free(occ, R, C, pr, pc)) continue;
        int p = pr * C + pc;
        if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
        if (BLOCKED(p) || !EXPD(p)) continue;
        if (g[p] + orc_dist(pr, pc, r, c) != g[x]) continue;
        if (best < 0 || t_lt(&T[p], &T[best])) best = p;
      }
      if (best < 0) { giveup = 2; break; }                       /* its label is not explained by an expanded node */
      tkey kx = {KF(x), g[x], x, 0}, nt;
      if (t_lt(&T[best], &kx)) nt = kx;
      else { nt = T[best]; nt.d += 1; }
      if (nt.bc != T[x].bc || nt.d != T[x].d) { T[x] = nt; changed = 1; }
      par[x] = best;
    }
  }
  /* delayed nodes: simple chains only, and nobody pops early (I1) */
  if (!giveup) {
    for (int li = 0; li < nl && !giveup; ++li) {
      int x = list[li];
      if (T[x].d == 0) continue;
      stats[1]++;
      for (int lj = 0; lj < nl; ++lj) { int y = list[lj]; if (y != x && T[y].d == T[x].d && T[y].bc == T[x].bc) { giveup = 3; break; } }
      if (giveup) break;
      int r = x / C, c = x % C;
      tkey base = {KF(T[x].bc), g[T[x].bc], T[x].bc, 0};
      for (int m = 0; m < nm; ++m) {
        int pr = r + HM_DR[m], pc = c + HM_DC[m];
        if (!orc_free(occ, R, C, pr, pc)) continue;
        int q = pr * C + pc;
        if (m >= 4 && restrict_corner && (!orc_free(occ, R, C, r + HM_DR[m], c) || !orc_free(occ, R, C, r, c + HM_DC[m]))) continue;
        if (BLOCKED(q) || !EXPD(q) || q == par[x]) continue;
        if (!t_lt(&T[q], &T[par[x]])) continue;                  /* arrives after the parent: irrelevant */
        double off = g[q] + orc_dist(pr, pc, r, c);
        double of_ = hzero ? off : off + orc_dist(r, c, tr, tc);
        if (!sk_lt(base.bf, base.bg, base.bc, of_, off, x)) { giveup = 4; break; }   /* x would pop before its parent */
      }
    }
  }
  /* every parent must be expanded BEFORE the goal pops; nodes popped after the goal never offer */
  if (!giveup) {
    for (int li = 0; li < nl && !giveup; ++li) {
      int x = list[li];
      if (x == start) continue;
      if (x != target && !t_lt(&T[x], &T[target])) continue;     /* not part of the run */
      stats[0]++;
      if (!t_lt(&T[par[x]], &T[target]) ) giveup = 5;
      /* its label must be the minimum over the offers of the nodes that really popped before it */
    }
    /* and no node that pops before the goal may have been left unexpanded by the pass (covered by the slack) */
  }
  stats[2] = giveup;
  int64_t ret = 0;
  if (!giveup) {
    int64_t n = 0; int t = target;
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)RC);
    while (t != start && t >= 0 && n < RC) { if (T[t].d) stats[7]++; tmp[n++] = t; t = par[t]; }
    if (t != start) { stats[2] = 6; ret = 0; }
    else { tmp[n++] = start; for (int64_t i = 0; i < n; ++i) out[i] = tmp[n - 1 - i]; ret = n; }
    free(tmp);
  }
  /* labels of the nodes outside the run are not the sequential ones: blank them for the comparison */
  if (!giveup) for (int li = 0; li < nl; ++li) { int x = list[li]; if (x != target && x != start && !t_lt(&T[x], &T[target])) g[x] = INFINITY; }
  free(T); free(par); free(list);
  stats[6] = 0;
  return ret;
}
