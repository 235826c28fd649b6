/*
 * pathfit.h -- C-ABI of the MI355X population-fitness engine (libpathfit.so).
 *
 * The reference (dvnam1605/MAACO-path-planing) has NO plugin / operator / FFI
 * interface: its boundary is the Python method surface of the solver classes
 * (SURVEY.md section 8b).  Each entry point below therefore names the
 * reference *method* whose per-agent inner loop it replaces, batched over the
 * population.  The Python facades in maaco-path-planing_amd/pathfit/ keep the
 * reference's class / constructor / solve() surface and call these through
 * ctypes (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - plain C: pointers, sizes, scalars; no C++/torch types.
 *   - every function returns 0 on success, <0 on error (pf_last_error()).
 *     Per-agent infeasibility is DATA (status arrays), never an error.
 *   - pointers named d_* are DEVICE pointers (pf_dev_alloc or any HIP
 *     allocation on the handle's device, e.g. torch tensor .data_ptr());
 *     all others are host pointers.
 *   - a cell is r*C + c (int32).  Paths are "strided CSR": agent a's cells are
 *     d_cells[a*path_cap .. a*path_cap + d_len[a]); d_len[a]==0 is the
 *     reference's [] (infeasible).
 *   - calls are synchronous at the ABI (they return after the handle's stream
 *     has drained) unless the name ends in _async.
 *   - one host thread per handle.
 *   - status codes: PF_ST_OK 0, PF_ST_INFEASIBLE 1 (reference returned []),
 *     PF_ST_STEP_CAP 2 (reference's 3RC / 2RC pop cap hit, also []),
 *     PF_ST_OVERFLOW 3 (engine scratch or path_cap too small: result not
 *     produced; never silently truncated), PF_ST_KEPT 4 (MPA: reference fell
 *     back to the unmodified path).
 */
#ifndef PATHFIT_H
#define PATHFIT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PF_ST_OK 0
#define PF_ST_INFEASIBLE 1
#define PF_ST_STEP_CAP 2
#define PF_ST_OVERFLOW 3
#define PF_ST_KEPT 4

#define PF_ASTAR_REF 0 /* AStarSolver.solve semantics, astar.py:33-101 */
#define PF_ASTAR_MPA 1 /* MPA._a_star semantics, MPA.py:106-151 */
#define PF_ASTAR_DIJKSTRA 2 /* DijkstraSolver.solve semantics, dijkstra.py:32-97 (variant 0's loop with h == 0) */

typedef struct pf_handle pf_handle;

/* scoring weights: helper.calculate_path_stats (helper.py:98-113) when
 * variant==0, MPA._calculate_path_stats (MPA.py:215-229, safety==0) when 1 */
typedef struct {
  int32_t variant;
  int32_t restrict_policy; /* restrict_diagonal_near_obstacle_policy */
  double w_turn, w_safe, min_safe, diag_pen;
} pf_score_params;

/* MAACO constructor scalars, MAACO.py:11-14 */
typedef struct {
  double alpha, beta, rho, Q, a_turn_coef, wh_max, wh_min, k_h_adaptive, q0_initial, C0_initial_pheromone;
  int32_t num_iterations;
  int32_t start, target; /* cells */
} pf_maaco_params;

/* MPA scalars, MPA.py:10-18 (+ sigma = the Mantegna constant of :251-253,
 * computed by the caller with math.gamma) */
typedef struct {
  double P_const, levy_beta, levy_sigma, FADs_rate;
  int32_t num_predators; /* global population size (for the i < N//2 split, MPA.py:351) */
  int32_t start, target;
  int32_t allow_diag, restrict_corner;
} pf_mpa_params;

/* per-launch counters (roofline numerators) */
typedef struct {
  int64_t pops, pushes, nbr_examined, path_cells, steps, candidates, decrease_keys, overflow_agents;
  /* candidates: MAACO candidate cells examined; for the A*-based calls, open-list entries that took the spill list */
  int64_t pruned_rebuilds; /* MPA rebuilds skipped because a length bound proved the candidate could not be accepted */
  int64_t settled_searches;    /* closed-set searches answered by the parallel label-settling engine (certified equal) */
  int64_t sequential_searches; /* closed-set searches it could not certify: run by the sequential pop loop */
} pf_counters;

/* ---- lifecycle ---------------------------------------------------- */
const char* pf_version(void);
int pf_device_count(void);
/* grid: R*C bytes with the reference's cell values (env.py:4-7: 1 = obstacle,
 * anything else free).  Replaces np.array(grid, dtype=int) + obstacle_nodes
 * (helper.py:121-125).  device: HIP ordinal. */
int pf_create(const uint8_t* grid, int32_t R, int32_t C, int32_t device, pf_handle** out);
void pf_destroy(pf_handle* h);
/* Dynamic maps: replace the occupancy of an existing handle (same R x C, same cell values as pf_create).  The grid
 * preparation runs again on the device; call pf_maaco_setup / pf_mpa_setup again before the next solver batch. */
int pf_update_grid(pf_handle* h, const uint8_t* grid);
const char* pf_last_error(pf_handle* h); /* h may be NULL for create errors */
/* the handle's HIP stream (hipStream_t) so callers can order their own work */
void* pf_stream(pf_handle* h);
int pf_sync(pf_handle* h);

/* ---- device memory plumbing --------------------------------------- */
int pf_dev_alloc(pf_handle* h, int64_t bytes, void** d_out);
int pf_dev_free(pf_handle* h, void* d_ptr);
int pf_h2d(pf_handle* h, void* d_dst, const void* src, int64_t bytes);
int pf_d2h(pf_handle* h, void* dst, const void* d_src, int64_t bytes);
int pf_d2d(pf_handle* h, void* d_dst, const void* d_src, int64_t bytes);
int pf_memset(pf_handle* h, void* d_dst, int32_t byte, int64_t bytes);
/* counters of the most recent batch call */
int pf_get_counters(pf_handle* h, pf_counters* out);
/* timing of the most recent batch call's dominant kernel, measured with HIP
 * events on the handle's stream (ms) */
float pf_last_kernel_ms(pf_handle* h);

/* ---- K2: A* connector batch ---------------------------------------- */
/* Replaces AStarSolver.solve(start, target, nodes_to_avoid) (astar.py:33) or
 * MPA._a_star(start, end, nodes_to_avoid) (MPA.py:106) for n independent
 * queries.  Avoid sets are CSR (d_avoid_off int64[n+1], d_avoid_cells int32)
 * or NULL.  d_counters: int64[n*4] {pops, pushes, max_open, nbr_examined} or NULL. */
int pf_astar_batch(pf_handle* h, int32_t variant, int32_t allow_diag, int32_t restrict_corner, int32_t n,
                   const int32_t* d_start, const int32_t* d_target, const int64_t* d_avoid_off,
                   const int32_t* d_avoid_cells, int32_t path_cap, int32_t* d_cells, int32_t* d_len,
                   int32_t* d_status, int64_t* d_counters);

/* ---- K1: path scoring batch ---------------------------------------- */
/* Replaces BasePathfinder._calculate_stats_for_path (helper.py:138) /
 * MPA._calculate_path_stats (MPA.py:215).  d_stats: double[n*5] =
 * {length, turns, safety, diag, fitness}; empty path -> {inf,0,0,0,inf}. */
int pf_score_batch(pf_handle* h, const pf_score_params* sp, int32_t n, int32_t path_cap,
                   const int32_t* d_cells, const int32_t* d_len, double* d_stats);

/* ---- K3: chained waypoint decode (+ score) ------------------------- */
/* Replaces GASolver._reconstruct_path_from_chromosome (ga_solver.py:58) when
 * d_wp_cells != NULL (int32[n*W]) or PSOSolver._reconstruct_path_from_position
 * (pso.py:56; round-half-even + clamp) when d_wp_pos != NULL (double[n*W*2]),
 * followed by _calculate_stats_for_path when sp != NULL. */
int pf_decode_batch(pf_handle* h, int32_t allow_diag, int32_t restrict_corner, int32_t n, int32_t W,
                    const int32_t* d_wp_cells, const double* d_wp_pos, int32_t start, int32_t target,
                    int32_t path_cap, int32_t* d_cells, int32_t* d_len, int32_t* d_status,
                    const pf_score_params* sp, double* d_stats);

/* ---- K6: PSO velocity/position update ------------------------------ */
/* Replaces the inner loop pso.py:183-203 for particles agent0..agent0+n with
 * the sweep-start gbest (synchronous PSO).  In place on
 * d_pos/d_vel (double[n*W*2]).  Stream (seed, DOM_PSO=3, iter, agent). */
int pf_pso_update(pf_handle* h, int32_t n, int32_t W, double w, double c1, double c2, double max_vel,
                  double* d_pos, double* d_vel, const double* d_pbest, const double* d_gbest, uint64_t seed,
                  uint64_t iter, uint64_t agent0);
/* The same update, asynchronous (stream ordered), leaving the pre-update position / velocity of every particle in
 * d_pos_keep / d_vel_keep: the roll-back copy of the asynchronous sweep (pso.py:222-229: a particle evaluated on a gbest that
 * an earlier particle of the sweep has moved must be evaluated again from its old state). */
int pf_pso_update_keep(pf_handle* h, int32_t n, int32_t W, double w, double c1, double c2, double max_vel,
                       double* d_pos, double* d_vel, const double* d_pbest, const double* d_gbest, uint64_t seed,
                       uint64_t iter, uint64_t agent0, double* d_pos_keep, double* d_vel_keep);
/* One round of the asynchronous sweep committed in one launch, for the evaluated batch [0, m): particles [0, n_final) are
 * final -- pso.py:216-220 (pbest position, fitness AND path row) --, `improver` (< n_final, or -1) is the round's first gbest
 * improver -- pso.py:222-229: its position -> d_gbest[W*2], its five stats -> d_gbest_stats[5], its path -> d_gbest_path
 * ([0] = length, then the cells) --, particles [n_final, m) roll back to d_pos_keep / d_vel_keep.  Asynchronous. */
int pf_pso_commit(pf_handle* h, int32_t m, int32_t W, int32_t path_cap, int32_t n_final, int32_t improver, double* d_pos,
                  double* d_vel, const double* d_pos_keep, const double* d_vel_keep, const double* d_stats, const int32_t* d_len,
                  const int32_t* d_cells, double* d_pbest, double* d_pbest_fit, int32_t* d_pb_cells, int32_t* d_pb_len,
                  double* d_gbest, double* d_gbest_stats, int32_t* d_gbest_path);
/* pbest bookkeeping pso.py:216-220: where stats fitness < pbest_fit (strict),
 * copy pos -> pbest and fitness -> pbest_fit.  d_improved int32[n] out. */
int pf_pso_pbest(pf_handle* h, int32_t n, int32_t W, const double* d_pos, const double* d_stats,
                 const int32_t* d_len, double* d_pbest, double* d_pbest_fit, int32_t* d_improved);

/* ---- K4/K5: MAACO --------------------------------------------------- */
/* Replaces MAACO.__init__ state (pheromone_matrix :58-84, dist table :86-91)
 * plus the per-cell eta'^beta tables derived from :197-210 (host libm, so
 * bit-identical to the reference's math.exp / pow). */
int pf_maaco_setup(pf_handle* h, const pf_maaco_params* p);
/* Replaces MAACO._construct_ant_solution_maaco (MAACO.py:278) for ants
 * ant0..ant0+n of iteration iter.  d_plen double[n] (inf if failed),
 * d_turns int32[n] (-1 if failed).  Stream (seed, DOM_MAACO=1, iter, ant). */
int pf_maaco_walk_batch(pf_handle* h, int32_t iter, uint64_t seed, int32_t ant0, int32_t n, int32_t path_cap,
                        int32_t* d_cells, int32_t* d_len, double* d_plen, int32_t* d_turns, int32_t* d_status);
/* Replaces MAACO._update_pheromone_trails_maaco (MAACO.py:304) in three
 * ordered steps so that a population sharded over GPUs can fold its deposits
 * in global ant order: evaporate (:305), deposit (:306-311, sequential in ant
 * order per cell -- bit-exact, no float atomics), clip (:312-332). */
int pf_maaco_evaporate(pf_handle* h);
int pf_maaco_deposit(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                     const double* d_plen);
int pf_maaco_clip(pf_handle* h, double best_len_overall);
/* The same update (MAACO.py:304-332) in ONE pass over tau for the paths of one batch on one GPU: per cell evaporate (:305),
 * the deposits in ant order (:306-311), clip (:312-332) -- the same fp64 operations in the same order as the three calls
 * above.  Successful ants mark their own deposits at the end of their walk (option "maaco_mark_in_walk", default 1), so
 * the update of the batch that was just walked needs no pass over the paths. */
int pf_maaco_update(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len, const double* d_plen,
                    double best_len_overall);
/* One whole iteration of MAACO.solve_path_planning (MAACO.py:340-359) for the ants of one GPU, enqueued back to back: walks,
 * best-of-iteration scan (:343-349), take-over test against the caller's overall best (:351-358), one-pass pheromone update.
 * The overall best ant's path row stays in HBM (pf_maaco_best_path reads it on demand).
 * ONE 104-byte block comes back: out13 = {ib_len, ib_turns, ib_idx, took, best_len, best_turns, tmin, tmax, skipped, steps,
 * candidates, path_cells, overflow_agents}.  overflow_agents > 0: the pheromone was left untouched (skipped = 1); repeat
 * the call with longer path rows.  The call returns once the take-over test is known (the device mirrors out13 into pinned
 * host memory; no copy is enqueued); the pheromone update may still be running -- every later pf_ call on this handle is
 * ordered after it on the handle's stream, pf_sync waits for it. */
int pf_maaco_iterate(pf_handle* h, int32_t iter, uint64_t seed, int32_t ant0, int32_t n, int32_t path_cap,
                     int32_t* d_cells, int32_t* d_len, double* d_plen, int32_t* d_turns, int32_t* d_status,
                     double best_len, double best_turns, double* out13);
/* MAACO.best_path_overall (MAACO.py:351-358), kept in HBM by pf_maaco_iterate: whenever an iteration's best ant takes over, its
 * path row is copied on the device.  This call materialises it: *len_out cells into cells_out[cap] (0: no ant has arrived yet). */
int pf_maaco_best_path(pf_handle* h, int32_t* cells_out, int32_t cap, int32_t* len_out);
/* pheromone_matrix attribute round trip (double[R*C]) */
int pf_maaco_get_pheromone(pf_handle* h, double* tau);
int pf_maaco_set_pheromone(pf_handle* h, const double* tau);
void* pf_maaco_tau_dev(pf_handle* h); /* device pointer, for collectives */
/* sequential best-of-iteration scan MAACO.py:343-349 over host arrays;
 * inout: best_len/best_turns/best_idx (idx -1 = none yet) */
int pf_maaco_best_scan(int32_t n, const double* plen, const int32_t* turns, int32_t idx0, double* best_len,
                       double* best_turns, int32_t* best_idx);

/* ---- GA host operators (native, no device work) ------------------------ */
/* GASolver._selection, ga_solver.py:136-142: one generation of tournaments on the stream (seed, DOM_GA_SELECT, gen, 0):
 * random.sample(population, min(tournament_size, n)) then the first minimum of fitness.  parent_idx[n] = indices into
 * the (sorted) population. */
int pf_ga_select(uint64_t seed, int32_t gen, int32_t n, int32_t tournament_size, const double* fitness, int32_t* parent_idx);
/* GASolver._create_chromosome, ga_solver.py:55-56 (+48-53), for attempts attempt0 .. attempt0+n-1 of the population
 * initialisation (:95-133): attempt k draws W free cells from the stream (seed, DOM_INIT, 0, k).  cells[n][W]. */
int pf_ga_random_chromosomes(uint64_t seed, int32_t attempt0, int32_t n, int32_t W, const uint8_t* occ, int32_t R, int32_t C,
                             int32_t* cells);
/* GASolver._crossover + _mutate + _generate_random_waypoint for all children of a generation, ga_solver.py:144-160,
 * 186-194, 48-53: pair j (parents 2j, 2j+1 mod n of parent_cells[n][W], cells r*C+c) draws from (seed, DOM_GA, gen, j).
 * occ = host occupancy, 1 = obstacle.  child_cells[n][W]. */
int pf_ga_breed(uint64_t seed, int32_t gen, int32_t n, int32_t W, double crossover_rate, double mutation_rate,
                const uint8_t* occ, int32_t R, int32_t C, const int32_t* parent_cells, int32_t* child_cells);

/* ---- K7 + K2b + K1: MPA --------------------------------------------- */
int pf_mpa_setup(pf_handle* h, const pf_mpa_params* p, const pf_score_params* sp);
/* One phase sweep MPA.py:339-377 over n local predators.  d_gidx[a] = index
 * of predator a in the GLOBAL fitness-sorted population of this iteration (the
 * reference's loop index: it keys the stream (seed, DOM_MPA=2, iter, gidx) and
 * decides the phase-2 Levy/Brownian split, MPA.py:351); d_slot[a] = its
 * storage slot in the strided population d_pop_*.  The kernel draws the start
 * idx + gate, then runs MPA._reconstruct_path_segment (:284-318) or the
 * phase's no-move branch.  d_elite_cells/elite_len/d_elite_stats(double[5]) =
 * the sweep-start elite (may alias the population storage: candidates go to
 * separate buffers).  phase in {1,2,3}; CF per MPA.py:336.
 * Outputs candidate paths (strided, same cap) + stats double[n*5]. */
int pf_mpa_phase_batch(pf_handle* h, int32_t phase, double CF, int32_t iter, uint64_t seed, int32_t n,
                       int32_t path_cap, const int32_t* d_pop_cells, const int32_t* d_pop_len,
                       const double* d_pop_stats, const int32_t* d_gidx, const int32_t* d_slot,
                       const int32_t* d_elite_cells, int32_t elite_len, const double* d_elite_stats,
                       int32_t* d_out_cells, int32_t* d_out_len, double* d_out_stats, int32_t* d_status);
/* FADs sweep MPA.py:387-410 on the post-memory population (in place):
 * stream (seed, DOM_MPA_FADS=5, iter, gidx). */
int pf_mpa_fads_batch(pf_handle* h, double CF, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                      const int32_t* d_gidx, const int32_t* d_slot, int32_t* d_pop_cells, int32_t* d_pop_len,
                      double* d_pop_stats, int32_t* d_status);
/* One whole MPA iteration's device work in a single longest-first work queue: the phase sweep (as
 * pf_mpa_phase_batch, candidates -> d_c1_*) and the FADs candidates (MPA.py:387-410; they depend only on the
 * predator's stream and the grid, so they are produced concurrently -> d_c2_*, d_c2_len 0 = none), then the
 * memory step (:381-384) and the FADs acceptance (:402/:408) in place on the population.  Equivalent to
 * pf_mpa_phase_batch + pf_mpa_memory + pf_mpa_fads_batch, with one tail instead of two.  An agent whose
 * scratch overflowed is reported through pf_get_counters().overflow_agents and d_status (phase items). */
int pf_mpa_iter_batch(pf_handle* h, int32_t phase, double CF, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                      int32_t* d_pop_cells, int32_t* d_pop_len, double* d_pop_stats, const int32_t* d_gidx,
                      const int32_t* d_slot, const int32_t* d_elite_cells, int32_t elite_len, const double* d_elite_stats,
                      int32_t* d_c1_cells, int32_t* d_c1_len, double* d_c1_stats, int32_t* d_c2_cells, int32_t* d_c2_len,
                      double* d_c2_stats, int32_t* d_status);
/* MPA._reconstruct_path_segment (MPA.py:284-318) called directly: predator a
 * modifies population path a against the given elite path with explicit
 * idx / is_levy / scale and stream (seed, DOM_MPA, iter, d_agent[a]). */
int pf_mpa_rebuild_batch(pf_handle* h, int32_t iter, uint64_t seed, int32_t n, int32_t path_cap,
                         const int32_t* d_pop_cells, const int32_t* d_pop_len, const double* d_pop_stats,
                         const int32_t* d_elite_cells, int32_t elite_len, const int32_t* d_idx,
                         const int32_t* d_is_levy, const double* d_scale, const int32_t* d_agent,
                         int32_t* d_out_cells, int32_t* d_out_len, double* d_out_stats, int32_t* d_status);
/* memory step MPA.py:381-384: pop[d_slot[a]] <- cand[a] where cand fitness < pop fitness */
int pf_mpa_memory(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_slot, const int32_t* d_cand_cells,
                  const int32_t* d_cand_len, const double* d_cand_stats, int32_t* d_pop_cells, int32_t* d_pop_len,
                  double* d_pop_stats);

/* Tuning knobs (results never change): "maaco_pack8_min" ants per batch from which eight ants share a wavefront
 * (default 2048); "maaco_load_ahead" the packed walk kernel's load-ahead form (all of a step's loads issued together plus touches of the
 * records two steps ahead): -1 (default) for batches of at most one wavefront per SIMD, 0 never, 1 always;
 * "mpa_prune" 0/1 exact bound pruning of MPA rebuilds (default 1); "two_wave" 0/1 MPA._a_star searches (pf_mpa_iter_batch,
 * pf_astar_batch variant 1) on two-wavefront workgroups -- a pop wave and a pool wave, csrc/pf_astar_pr.h -- default 0:
 * identical pops, measured 0.9x; compiled only with -DPF_TWO_WAVE (PF_EXTRA_FLAGS of build.py), otherwise
 * setting it to 1 is an error; "astar_settle" 0/1 closed-set searches (AStarSolver
 * / Dijkstra / GA / PSO decodes) try the parallel label-settling engine first: -1 (default) the Dijkstra variant
 * always (it is always certified) and the A* searches of the decodes at the head of a batch's longest-first queue
 * ("astar_settle_top", per mille of the batch, default 0 since r03; a decode is a chain of W + 1 searches, so a fallback costs
 * one link) and -- "astar_settle_tail", per mille of the search slots, default 400 -- every decode search that starts once
 * the batch's unfinished agents no longer fill that share of the chip (the long chains the batch ends on, on an idle chip); 1 every A* search too (exact -- certified or handed back to the sequential loop -- but slower
 * on batches of single searches); 0 never (the sequential loop's pop / push counters are the reference's).  Test hook: "astar_step_cap" > 0
 * lowers the connectors' step cap below the reference's 3RC / 2RC (astar.py:58, MPA.py:118) so that the cap path
 * (PF_ST_STEP_CAP) can be exercised; 0 restores the reference's value.  "mpa_doubt_log_e15" / "mpa_doubt_round_e15":
 * margins (in 1e-15; < 0 = default) inside which an MPA proposal is handed to the host's libm (tests widen them to
 * force that route). */
int pf_set_option(pf_handle* h, const char* name, int64_t value);

/* ---- device self-tests (used by tests/ to pin device arithmetic) ----- */
/* out[i] = device sqrt((double)in[i]) -- must equal libm sqrt bit for bit
 * (heuristic astar.py:90 / helper.py:12). */
int pf_selftest_sqrt(pf_handle* h, int32_t n, const int64_t* d_in, double* d_out);
/* device keyed RNG + CPython derivations for key (seed,dom,it,agent):
 * d_u64[8] next64; d_f64[0..8) random(), [8..28) normalvariate (16 x (0,1),
 * 4 x (0,0.7)), [28..36) uniform(0,2pi); d_i64[0..24) randint as in
 * oracle/capture_golden.py cap_rng, [24..34) randbelow(n) for the same n
 * list, [34..37) draw counters after the randint / normal / choice runs. */
int pf_selftest_rng(pf_handle* h, uint64_t seed, uint64_t dom, uint64_t it, uint64_t agent, uint64_t* d_u64,
                    double* d_f64, int64_t* d_i64);

/* ---- iteration control in HBM -------------------------------------------------------------------
 * list.sort(key=fitness) (MPA.py:321,333,412; ga_solver.py:209): d_order holds the list (position -> id); it is
 * re-ordered by a STABLE device sort on key[pos] = d_vals[d_order[pos] * stride + offset].  Stream ordered. */
int pf_sort_order_by_key(pf_handle* h, int32_t n, const double* d_vals, int32_t stride, int32_t offset, int32_t* d_order);
/* population[0] after the sort (MPA.py:334, :413; ga_solver.py:210): out2 = {id at the head of d_order, its key
 * d_vals[id * stride + offset]} in one 16-byte copy */
int pf_sorted_head(pf_handle* h, const int32_t* d_order, const double* d_vals, int32_t stride, int32_t offset, double* out2);
/* d_dst[i] = d_src[i * stride + offset] (a stats column packed for an all_gather) */
int pf_gather_col(pf_handle* h, int32_t n, const double* d_src, int32_t stride, int32_t offset, double* d_dst);
/* d_out[i] = d_a[i] + sign * d_b[i] (sign = +1 / -1; one IEEE operation per element, in place allowed): the non-strict MAACO
 * exchange (all_reduce of per-rank pheromone deltas; no reference counterpart) forms and applies its deltas with it. */
int pf_vec_add_f64(pf_handle* h, int32_t n, const double* d_a, const double* d_b, double sign, double* d_out);
/* The elite of an MPA iteration (MPA.py:334, population[0] after the sort) lives in a library-owned buffer: cells
 * [R*C] int32, length int32, stats double[5].  pf_mpa_pick_elite copies the row of predator d_order[0] - first_id of
 * this rank's store into it; a sharded run broadcasts the three pieces from the owner instead.  Passing
 * elite_len = -1 to pf_mpa_iter_batch makes the sweep read the length from that buffer (the host never learns it). */
int pf_mpa_elite_buf(pf_handle* h, void** d_cells, void** d_len, void** d_stats);
int pf_mpa_pick_elite(pf_handle* h, int32_t path_cap, const int32_t* d_pop_cells, const int32_t* d_pop_len,
                      const double* d_pop_stats, const int32_t* d_order, int32_t first_id);
/* positions (in the global fitness order d_gorder[N] of ids) and storage slots of the predators with ids in [lo, hi),
 * in position order: the d_gidx / d_slot arrays of pf_mpa_iter_batch for a rank that stores ids [lo, hi) */
int pf_mpa_local_view(pf_handle* h, int32_t N, const int32_t* d_gorder, int32_t lo, int32_t hi, int32_t* d_gidx, int32_t* d_slot);

/* ---- one GA generation in HBM (ga_solver.py:178-213) -----------------------------------------------------------------
 * The population is stored by storage id (= child index of the generation that made the individual): chromosomes
 * d_chrom_all [N][W] cells, fitness d_fit_all [N]; d_gorder [N] is the fitness-sorted list (position -> storage id).
 * pf_ga_select_dev: GASolver._selection for all N slots (one sequential stream per generation, replayed by one
 *   device thread) -> d_psid[s] = storage id of the parent chosen for slot s.
 * pf_ga_breed_dev: _crossover + _mutate for the children with index in [child0, child0 + nchild) (thread per pair,
 *   stream (seed, DOM_GA, gen, pair)) -> d_out [nchild][W].
 * pf_ga_assemble_dev: new individual i = child i if it decoded (d_kid_len > 0) else its fallback parent d_psid[lo + i]
 *   (ga_solver.py:204-205); a fallback parent's path row is copied when this rank stores it (ids [old_lo, old_hi)),
 *   else the row is marked absent (length -1).  All stream ordered, no host copies. */
int pf_ga_select_dev(pf_handle* h, uint64_t seed, int32_t gen, int32_t n, int32_t tournament_size, const double* d_fit_all,
                     const int32_t* d_gorder, int32_t* d_psid);
int pf_ga_breed_dev(pf_handle* h, uint64_t seed, int32_t gen, int32_t N, int32_t W, double crossover_rate, double mutation_rate,
                    const int32_t* d_chrom_all, const int32_t* d_psid, int32_t child0, int32_t nchild, int32_t* d_out);
int pf_ga_assemble_dev(pf_handle* h, int32_t n_loc, int32_t W, int32_t path_cap, int32_t lo, const int32_t* d_kid_len,
                       const int32_t* d_kid_chrom, const double* d_kid_stats, const int32_t* d_kid_cells, const int32_t* d_psid,
                       const int32_t* d_chrom_old, const double* d_stats_old, const int32_t* d_cells_old, const int32_t* d_len_old,
                       int32_t old_lo, int32_t old_hi, int32_t* d_chrom_new, double* d_stats_new, int32_t* d_cells_new,
                       int32_t* d_len_new);

/* MAACO.py:306-311 in two steps (the multi-GPU fold works on row chunks): _begin marks the cells of the successful
 * ants and computes Q / L per ant, _cells adds the deposits of cells [cell0, cell1) in ant order.  pf_maaco_deposit =
 * _begin + _cells(0, R*C).  All stream ordered. */
int pf_maaco_deposit_begin(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                           const double* d_plen);
int pf_maaco_deposit_cells(pf_handle* h, int32_t cell0, int32_t cell1);
/* MAACO.py:343-349 (best ant of an iteration: length, then turns within 1e-9) over device columns of n ants;
 * out3 = {best_len, best_turns, best_idx} (-1: no ant arrived).  One 24-byte device-to-host copy. */
int pf_maaco_best_dev(pf_handle* h, int32_t n, const double* d_plen, const int32_t* d_turns, double* out3);

/* ---- multi-GPU exchange: RCCL over xGMI, bound directly (no reference counterpart) -------------
 * One process per GPU.  librccl is opened on first use.  pf_comm_unique_id (rank 0) produces the 128-byte id the
 * launcher ships to the other ranks (a file, an env var, torch.distributed's store); pf_comm_init joins the
 * communicator on the handle's device.  Every collective below is ENQUEUED on the handle's stream, in order with the
 * kernels: no host synchronisation, device pointers only.  Sizes are bytes.  The population solvers need exactly:
 *   all_gather   per-agent (length, turns) / fitness columns and GA chromosomes          (C2, C3)
 *   broadcast    the winner's / elite's path row and stats from its owner                 (C2)
 *   send/recv    the pheromone matrix chunks of the ordered MAACO fold, rank k-1 -> k    (C1, MAACO.py:306-311)
 *   all_reduce   the non-strict MAACO delta sum, timing maxima                           */
int pf_comm_unique_id(void* id128);
int pf_comm_init(pf_handle* h, int32_t rank, int32_t world, const void* id128);
int pf_comm_destroy(pf_handle* h);
int pf_comm_rank(pf_handle* h);
int pf_comm_world(pf_handle* h);
int pf_comm_all_gather(pf_handle* h, const void* d_send, void* d_recv, int64_t bytes_per_rank);
int pf_comm_broadcast(pf_handle* h, void* d_buf, int64_t bytes, int32_t root);
int pf_comm_all_reduce_f64(pf_handle* h, double* d_buf, int64_t count, int32_t op /* 0 sum, 1 min, 2 max */);
int pf_comm_send(pf_handle* h, const void* d_buf, int64_t bytes, int32_t peer);
int pf_comm_recv(pf_handle* h, void* d_buf, int64_t bytes, int32_t peer);
/* one ring step as a single group: send to `to` (skip if < 0) and receive from `from` (skip if < 0) */
int pf_comm_sendrecv(pf_handle* h, const void* d_send, int64_t send_bytes, int32_t to, void* d_recv, int64_t recv_bytes,
                     int32_t from);

/* pso.py:218-219 for the paths: rows of the particles pf_pso_pbest marked improved are copied into the pbest path
 * store (both strided [n][path_cap]); asynchronous (stream ordered). */
int pf_pso_pbest_paths(pf_handle* h, int32_t n, int32_t path_cap, const int32_t* d_cells, const int32_t* d_len,
                       const int32_t* d_improved, int32_t* d_pb_cells, int32_t* d_pb_len);
/* pbest -> gbest scan of one evaluated batch of n particles (pso.py:216-229): *idx_out = the first particle that
 * improves the gbest (feasible, fitness below its pbest AND below gbest_fit), or with sync_mode != 0 the first particle
 * with the smallest such fitness; -1 if none.  *fit_out its fitness, *overflow_out the number of particles whose
 * status is PF_ST_OVERFLOW.  One 16-byte device-to-host copy. */
int pf_pso_scan(pf_handle* h, int32_t n, const double* d_stats, const int32_t* d_len, const int32_t* d_status,
                const double* d_pbest_fit, double gbest_fit, int32_t sync_mode, int32_t* idx_out, double* fit_out,
                int32_t* overflow_out);
/* Device-to-host copies made through this handle since pf_create: copies of at most 128 bytes, larger ones, and the
 * bytes of the larger ones (the solver loops keep populations in HBM: SURVEY.md 8 f1/f2). */
int pf_d2h_counts(pf_handle* h, int64_t* small_copies, int64_t* bulk_copies, int64_t* bulk_bytes);

/* Spans: HIP-event timed stretches of work on the handle's stream (kernels, pf_comm_* collectives), recorded without
 * synchronising.  pf_span_begin / pf_span_end bracket one stretch (not nested); pf_span_total synchronises the stream and
 * returns the summed duration (ms) and the number of spans since the last reset.  Measurement only (bench.py times the
 * per-iteration exchange of the sharded solvers with it: SURVEY.md 8e); no reference counterpart. */
int pf_span_begin(pf_handle* h);
int pf_span_end(pf_handle* h);
int pf_span_total(pf_handle* h, double* ms_out, int64_t* count_out, int32_t reset);

/* Target-cell proposals of MPA._get_levy_target_node / _get_brownian_target_node (MPA.py:250-282) for n keyed streams
 * (seed, DOM_MPA, 0, i) from cells d_cur[i] (and elite cells d_elite[i], < 0 = None): the device arithmetic, with the
 * proposals whose accept test / rounding lies within the libm-disagreement margin recomputed by the host's glibc
 * exactly as the MPA sweeps do.  *n_doubt = how many took that route. */
int pf_selftest_mpa_targets(pf_handle* h, uint64_t seed, int32_t n, int32_t is_levy, double beta, double sigma,
                            double scale, const int32_t* d_cur, const int32_t* d_elite, int32_t* d_out, int64_t* n_doubt);
/* proposals the host had to confirm since pf_create (expected: 0) */
long long pf_mpa_doubts_resolved(pf_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* PATHFIT_H */
