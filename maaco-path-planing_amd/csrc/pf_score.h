// pf_score.h -- K1: wave-parallel path scoring with the reference's exact
// floating-point order.
//   variant 0: helper.calculate_path_stats, helper.py:98-113
//   variant 1: MPA._calculate_path_stats,   MPA.py:215-229 (safety == 0.0)
// `length` and `safety` are naive left-to-right fp64 sums in the reference
// (helper.py:106, :79); they are re-created by a uniform serial add chain fed
// through v_readlane from 64 per-lane terms at a time, while turns / corner
// cuts / the obstacle-distance lookup are lane-parallel.  The O(L * n_obst)
// numpy scan of helper.py:67-80 is replaced by one byte lookup per cell into
// d2near (K0) + a penalty LUT built on the host with libm pow (bit-equal, see
// oracle safety_window); for min_safe_distance > 15.9 the byte table is replaced by an exact
// i32 squared-distance transform of the same window (k_edt_rows / k_edt_cols).
#pragma once
#include "pf_device.h"

namespace pf {

struct ScoreP {
  int variant, restrict_policy;
  double w_turn, w_safe, diag_pen;
  const double* pen;  // [256] penalty by clipped squared distance (device, read-only) -- or [pen_n] in wide mode
  const int* d2w;     // wide mode (min_safe_distance > 15.9): exact squared distance to the nearest obstacle per cell, clipped at
  int pen_n;          // pen_n - 1 (device EDT, two separable i32 passes); null: the u8 table G.d2near
};

// out: uniform {length, turns, safety, diag, fitness}
PF_DEV void score_path(const Grid& G, const ScoreP& P, const int* cells, int L, int lane, double* out) {
  if (L <= 0) { out[0] = PF_INF; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0; out[4] = PF_INF; return; }
  const int C = G.C;
  double length = 0.0, safety = 0.0;
  int turns = 0, ncut = 0;
  for (int base = 0; base < L; base += 64) {
    const int i = base + lane;
    int r0 = 0, c0 = 0, r1 = 0, c1 = 0, r2 = 0, c2 = 0;
    const bool v0 = i < L, v1 = i + 1 < L, v2 = i + 2 < L;
    int cell0 = 0;
    if (v0) { cell0 = cells[i]; r0 = row_of(G, cell0); c0 = cell0 - r0 * C; }
    if (v1) { int x = cells[i + 1]; r1 = row_of(G, x); c1 = x - r1 * C; }
    if (v2) { int x = cells[i + 2]; r2 = row_of(G, x); c2 = x - r2 * C; }
    const int dr = r1 - r0, dc = c1 - c0;
    double cost = 0.0;
    bool cut = false;
    if (v1) {
      const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
      if (adr + adc == 1) cost = 1.0;
      else if (adr == 1 && adc == 1) {
        cost = PF_SQRT2;
        // helper.py:91-95 / MPA.py:189-198: corner cells (next_r, cur_c), (cur_r, next_c)
        cut = P.restrict_policy && (G.occ[r1 * C + c0] == 1 || G.occ[r0 * C + c1] == 1);
      } else cost = __builtin_sqrt((double)((long)dr * dr + (long)dc * dc));
    }
    const bool turn = v2 && (dr != r2 - r1 || dc != c2 - c1);        // helper.py:58-65
    double pen = 0.0;
    if (P.variant == 0 && v0) {
      if (P.d2w) { const int d2 = P.d2w[cell0]; pen = P.pen[d2 < P.pen_n - 1 ? d2 : P.pen_n - 1]; }
      else pen = P.pen[G.d2near[cell0]];
    }
    turns += __builtin_popcountll(__ballot(turn));
    ncut += __builtin_popcountll(__ballot(cut));
    // serial fp64 chains, in path order
    int nsteps = L - 1 - base; if (nsteps > 64) nsteps = 64;
    for (int j = 0; j < nsteps; ++j) length = length + bcast_d(cost, j);
    unsigned long long nz = __ballot(pen != 0.0);
    while (nz) { const int j = __builtin_ctzll(nz); nz &= nz - 1; safety = safety + bcast_d(pen, j); }
  }
  if (P.variant == 0) safety = safety / (double)L;                    // helper.py:80
  double diag = 0.0;
  if (L >= 2) for (int k = 0; k < ncut; ++k) diag += P.diag_pen;      // helper.py:95 repeated +=
  out[0] = length; out[1] = (double)turns; out[2] = safety; out[3] = diag;
  out[4] = length + P.w_turn * (double)turns + P.w_safe * safety + diag;   // helper.py:112
}

}  // namespace pf
