#!/usr/bin/env python3
"""bench.py -- agent-fitness-evals/sec on a 512x512 grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload mpa512|maaco512|pso512|ga512|maaco128|maaco1024|astar1024]

Headline workload (default `mpa512`) = BASELINE.json configs[2]: MPA, 4096 predators on G512 (np.kron 2x of the
reference's 256x256 map), main.py:44-52 parameters.  A "step" is one MPA iteration's population evaluate-and-update hot
path: per predator propose a target cell, stitch with two A* connectors, score, greedy memory, FADs.  One eval = one
predator's pass through it.  The timed region is ONE COMPLETE K-iteration run of the solver (so the three phases
appear in the reference's proportions, MPA.py:339-377); the W warm-up iterations run on a separate instance of the
same configuration.  Inputs (grid, population) are resident in HBM when the timed region starts.  For N>1 every rank
owns 4096 predators (weak scaling); the only exchange is the fitness all_gather + elite broadcast per iteration
(pathfit/dist.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel (HIP-event timed
inside the library on its own stream; `copy_GBs_measured` = a plain 1 GiB device-to-device copy in the same run),
`cpu_baseline` (the CPU oracle port, bounded sample, 1 core; for mpa512 also `all_cores`) and, at N=1 with the default
workload, `extra`: every other BASELINE workload measured in the same process, each with value / ms_per_step / roofline
(pso512 = the parity-exact ASYNCHRONOUS PSO through PSOSolver.solve; `--no-extra` skips them).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "oracle")]

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

MPA_MAIN = dict(FADs_rate=0.2, P_const=0.5, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
                min_safe_distance=1.8, diagonal_obstacle_penalty=100.0)             # main.py:44-52
MAACO_MAIN = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9,
                  q0_initial=0.5, C0_initial_pheromone=0.1)                          # main.py:34-38
W_MAIN = dict(turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
              diagonal_obstacle_penalty_value=100.0)                                 # main.py:21-24
WORKLOADS = ["mpa512", "maaco512", "pso512", "ga512", "maaco128", "maaco1024", "astar1024"]
DOMINANT = {"mpa512": ("mpa_sweep", "k_mpa_search"), "ga512": ("decode", "k_decode_batch"), "pso512": ("decode", "k_decode_batch"),
            "astar1024": ("astar", "k_astar_batch"), "maaco128": ("maaco_walk", "k_maaco_walk"),
            "maaco512": ("maaco_walk", "k_maaco_walk8"), "maaco1024": ("maaco_walk", "k_maaco_walk8")}


def astar_bytes(c):
    """Algorithmic bytes of the A* work of one launch: per pop a 24 B heap entry + the 3x3
    occupancy window (9 B) + 1 B closed mark, 8 B per examined neighbour, 33 B per push; 4 B per emitted cell."""
    return 34 * c["pops"] + 8 * c["nbr_examined"] + 33 * c["pushes"] + 4 * c["path_cells"]


def maaco_bytes(c):
    """9 B window + 8 B tabu probes + 16 B per candidate (tau + eta) + 5 B path/tabu write per step."""
    return 22 * c["steps"] + 16 * c["candidates"]


def gsize_of(w):
    return 128 if w.endswith("128") else (1024 if w.endswith("1024") else 512)


class Run:
    """One workload on one engine: build(), then step() K times; evals per step = per_gpu (x world)."""

    def __init__(self, name, eng, grid, comm, a, rank, world, K, W):
        import pathfit
        from pathfit import env
        from pathfit.dist import ShardedMPA, ShardedMAACO
        self.name, self.eng, self.K, self.W = name, eng, K, W
        self.pre = None            # optional: runs before the timed region (after warm-up)
        gsize = gsize_of(name)
        per_gpu = a.agents
        if name == "mpa512":
            per_gpu = per_gpu or 4096
            total = per_gpu * world

            def make(iters):
                return ShardedMPA(comm, lambda n: pathfit.MPA(grid, total, iters, engine=eng, seed=a.seed, n_local=n, **MPA_MAIN), total)
            self.cfg = {"workload": f"MPA {per_gpu} predators/GPU, 512x512 G512 (BASELINE.json configs[2]), main.py:44-52 params; timed region = "
                                    f"one complete {K}-iteration run (phases 1/2/3 in the reference's proportions)",
                        "agents_per_gpu": per_gpu, "grid": "G512=kron2(G256)", "grid_sha256": env.grid_hash(grid)[:16]}
            st = {"sm": None, "it": 0}
            if a.phase1_only:      # the round-1 protocol, kept for comparison: every step is a phase-1 iteration
                st["sm"] = make(max(K + W, 3 * (K + W)))
                self.cfg["workload"] += " [--phase1-only: phase-1 iterations of a longer run]"

                def step():
                    st["it"] += 1
                    st["sm"].step(st["it"])
                self.warm = step
            else:
                wm = {"sm": make(max(W, 1)), "it": 0}

                def warm():
                    wm["it"] += 1
                    wm["sm"].step(wm["it"])
                self.warm = warm

                def pre():
                    wm["sm"] = None
                    st["sm"] = make(K); st["it"] = 0
                self.pre = pre

                def step():
                    st["it"] += 1
                    st["sm"].step(st["it"])
            self.step = step
            self.bytes_of = astar_bytes
        elif name.startswith("maaco"):
            # maaco512: 16384 ants/GPU; maaco128 = BASELINE.json configs[1] (256 ants, G128); maaco1024 = configs[4]'s per-GPU
            # share (8192 ants, G1024)
            per_gpu = per_gpu or {128: 256, 512: 16384, 1024: 8192}[gsize]
            total = per_gpu * world
            sm = ShardedMAACO(comm, lambda: pathfit.MAACO(grid, total, 100, engine=eng, seed=a.seed, **MAACO_MAIN), total)
            st = {"it": 0}

            def step():
                st["it"] += 1
                sm.step(st["it"])
            self.step = self.warm = step
            self.setup_steps = 1   # set-up, not warm-up: the first iteration allocates the walk / visit-bit buffers
            self.cfg = {"workload": f"MAACO {per_gpu} ants/GPU on {gsize}x{gsize}, main.py:34-38 params (walk + ordered pheromone update)",
                        "agents_per_gpu": per_gpu,
                        "grid": {128: "G128=random_blocks(seed 128)", 512: "G512=kron2(G256)", 1024: "G1024=kron4(G256)"}[gsize],
                        "grid_sha256": env.grid_hash(grid)[:16]}
            self.bytes_of = maaco_bytes
        elif name == "astar1024":
            # BASELINE.json configs[4], second part: a standalone batch of seeded (start, target) pairs on G1024 through the
            # AStarSolver connector (8192 pairs per GPU = 65536 over 8); one eval = one connector solve + path emit
            per_gpu = per_gpu or 8192
            rng = np.random.default_rng(a.seed + rank)
            free = np.flatnonzero(grid.reshape(-1) != 1)
            cap = 16 * 1024 + 64
            d_s, d_t = eng.put(rng.choice(free, per_gpu).astype(np.int32)), eng.put(rng.choice(free, per_gpu).astype(np.int32))
            d_cells, d_len, d_st = eng.buf((per_gpu, cap), np.int32), eng.buf(per_gpu, np.int32), eng.buf(per_gpu, np.int32)

            def step():
                eng.astar_batch(0, d_s, d_t, per_gpu, cap, d_cells, d_len, d_st)
            self.step = self.warm = step
            self.cfg = {"workload": f"A* connector batch (AStarSolver semantics), {per_gpu} uniform free-cell pairs/GPU, G1024 "
                                    "(BASELINE.json configs[4] K2a batch)", "agents_per_gpu": per_gpu, "grid": "G1024=kron4(G256)",
                        "grid_sha256": env.grid_hash(grid)[:16]}
            self.bytes_of = astar_bytes
        elif name == "ga512":
            per_gpu = per_gpu or 2048                     # BASELINE.json configs[3]: 16384 over 8 GPUs
            rng = np.random.default_rng(a.seed + rank)
            free = np.flatnonzero(grid.reshape(-1) != 1)
            sp = pathfit.score_params(0, True, 0.3, 0.8, 1.8, 100.0)
            cap = 16 * 1024 + 64
            d_cells, d_len, d_st, d_stats = (eng.buf((per_gpu, cap), np.int32), eng.buf(per_gpu, np.int32), eng.buf(per_gpu, np.int32),
                                             eng.buf((per_gpu, 5), np.float64))
            d_wp = eng.put(rng.choice(free, (per_gpu, 5)).astype(np.int32).reshape(-1))

            d_fit_loc, d_fit_all = eng.buf(per_gpu, np.float64), eng.buf(per_gpu * world, np.float64)

            def step():
                eng.decode_batch(per_gpu, 5, 0, 512 * 512 - 1, cap, d_cells, d_len, d_st, d_wp, None, sp, d_stats)
                if world > 1:       # C3: the tournament needs the whole fitness column -- device column to device column
                    eng.gather_col(per_gpu, d_stats, 5, 4, d_fit_loc)
                    comm.all_gather(d_fit_loc, 0, d_fit_all, [per_gpu] * world)
            self.step = self.warm = step
            self.cfg = {"workload": f"GA chained-waypoint decode+score of one generation's children, W=5, {per_gpu} agents/GPU, G512 "
                                    "(BASELINE.json configs[3] per-GPU share)", "agents_per_gpu": per_gpu,
                        "grid_sha256": env.grid_hash(grid)[:16]}
            self.bytes_of = astar_bytes
        else:                                              # pso512
            per_gpu = per_gpu or 2048
            self.cfg = {"workload": f"PSO, {per_gpu} particles/GPU, W=5, G512, main.py:109-118 params, ASYNCHRONOUS gbest as pso.py:222-229 "
                                    "(speculate-and-repair through PSOSolver.solve); a step = one sweep of the swarm "
                                    "(BASELINE.json configs[3] per-GPU share)", "agents_per_gpu": per_gpu,
                        "grid_sha256": env.grid_hash(grid)[:16]}

            def make(iters):       # N > 1: one swarm of per_gpu x world particles, sharded in blocks (the asynchronous gbest is repaired across ranks)
                return pathfit.PSOSolver(grid, num_iterations=iters, num_particles=per_gpu * world, num_waypoints_per_particle=5, w=0.7,
                                         c1=1.5, c2=1.5, engine=eng, seed=a.seed, asynchronous=not a.pso_sync,
                                         comm=comm if world > 1 else None, **W_MAIN)
            st = {"ps": None}

            def warm():
                ps = make(1)
                ps.solve()

            def pre():
                st["ps"] = make(K)
                ok = st["ps"].begin()              # initialisation (20 N attempts at most) is set-up, outside the timed region
                assert ok, "PSO initialisation found no feasible particle"
            self.pre = pre

            def step():
                st["ps"].sweep()
            self.warm, self.step = warm, step
            self.bytes_of = astar_bytes
        self.per_gpu = per_gpu


def measure(run, eng, comm, sync_all, world, torch, dist, local_rank, backend):
    K, W = run.K, run.W
    fam, kern = DOMINANT[run.name]
    for _ in range(getattr(run, "setup_steps", 0)):
        run.step()
    for _ in range(W):
        run.warm()
    if run.pre:
        run.pre()
    eng.klog = []
    sync_all()
    if world > 1:                                       # the exchange block counts the timed region only (not warm-up / set-up waits)
        comm.exchange_ms(reset=True)
        run.x0 = (comm.bytes_moved, comm.calls)
    t0 = time.perf_counter()
    for _ in range(K):
        if os.environ.get("PF_BENCH_TRACE"):           # diagnostic: host wall time of every step, to stderr
            ts = time.perf_counter(); run.step(); print(f"step {1e3 * (time.perf_counter() - ts):.2f} ms", file=sys.stderr, flush=True)
        else:
            run.step()
    sync_all()
    dt = time.perf_counter() - t0
    log, eng.klog = eng.klog, None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = run.per_gpu * world * K / dt
    roof = None
    mine = [(ms, c) for (f, ms, c) in log if f == fam]
    if mine:
        kms = sum(ms for ms, _ in mine); kb = sum(run.bytes_of(c) for _, c in mine); n = len(mine)
        achieved = kb / (kms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(kms / n, 3),
                "launches": n, "launches_per_step": round(n / K, 2), "algorithmic_bytes_per_launch": int(kb / n)}
        if run.name == "mpa512":
            run.cfg["rebuilds_proven_rejected_and_skipped"] = int(sum(c["pruned_rebuilds"] for _, c in mine))
            run.cfg["sweep_ms_by_iteration"] = [round(ms, 2) for ms, _ in mine]
        traffic_note(roof, run.name)
    return value, dt, roof


def traffic_note(roof, workload):
    """HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/r04_traffic.json, written by scripts/profile_round.sh: FETCH_SIZE and
    WRITE_SIZE in separate --pmc runs, KB -> B, FETCH x2 per MI355X_MICROARCH.md).  The file records the kernel time it
    was captured at; a figure whose kernel has since changed speed by more than 15 % is reported as stale, not used."""
    for rnd in ("r04", "r03", "r02", "r01"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")))
        except Exception:
            continue
        t = tj.get(workload)
        if not t or t.get("kernel") != roof["kernel"]:
            continue
        ref_ms = t.get("avg_ms_rocprof") or t.get("bench_avg_launch_ms")
        if ref_ms and abs(roof["avg_launch_ms"] - ref_ms) <= 0.15 * ref_ms:
            roof["traffic"] = int(t["traffic_bytes_per_launch"])
            roof["traffic_source"] = f"profiles/{rnd}_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, offline; captured at {ref_ms:.2f} ms/launch)"
        else:
            roof["traffic_source"] = f"profiles/{rnd}_traffic.json is stale for this kernel (captured at {ref_ms} ms/launch): not used"
        return


RC_RCCL_STUCK = 75      # a rank whose direct-RCCL bootstrap never returned: the process is abandoned, a fresh one retries host-staged


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, worker=None, env_extra=None, timeout=None):
    """`python bench.py --gpus N` without torchrun: start N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), relay rank 0's JSON line, and fail loudly if any rank fails or the line does not report
    N ranks.  Runs BEFORE anything in this process has touched the GPU (no torch / pathfit import above this point), and
    never re-execs: the ranks are ordinary child processes.  `worker` replaces [python, bench.py] (the CPU test's stub).
    Returns (exit code, the JSON line or None)."""
    import subprocess
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    port, retry_port = _free_port(), _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PF_BENCH_RETRY_PORT=str(retry_port))
        env.update(env_extra or {})
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line, rcs = None, []
    t_end = None if timeout is None else time.monotonic() + timeout
    # rank 0's stdout is drained by a thread while ALL ranks are polled: a rank that dies at import or in init_process_group ends the
    # run at once (the survivors are killed) instead of leaving rank 0 inside torch's rendezvous until its own 10-30 minute timeout
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            if any(c not in (None, 0) for c in codes) or all(c is not None for c in codes):
                break
            if t_end is not None and time.monotonic() > t_end:
                codes = [124 if c is None else c for c in codes]
                break
            time.sleep(0.05)
        failed = [c for c in codes if c not in (None, 0)]
        # a rank still running when another one failed is killed below and reported as -9; the failing rank's own code comes first
        rcs = failed[:1] + [(-9 if c is None else c) for c in codes] if failed else list(codes)
    finally:
        for pr in procs:                      # exactly the processes started here, by pid
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except Exception:
                pass
    rd.join(timeout=10)
    for ln in out0:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")
    if any(rc != 0 for rc in rcs):
        print(f"[bench launcher] rank exit codes {rcs}", file=sys.stderr)
        return (next(rc for rc in rcs if rc != 0) or 1), line
    if line is None:
        print("[bench launcher] rank 0 printed no JSON line", file=sys.stderr)
        return 1, None
    try:
        got = json.loads(line).get("n_gpus")
    except Exception:
        got = None
    if got != n:
        print(f"[bench launcher] the line reports n_gpus={got!r}, {n} ranks were started", file=sys.stderr)
        return 1, line
    return 0, line


def _retry_host_staged(a):
    """This rank's direct-RCCL bootstrap thread is stuck inside the library: the handle it shares with this thread must not
    be used again.  Every rank reached the same verdict (attach_checked all-reduces it), so every rank starts ONE fresh child
    process of itself with PF_BENCH_TRANSPORT=torch on the next port, relays its output and exits with its code."""
    import subprocess
    # (a fresh child process, never a re-exec: this process has initialised the GPU)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=retry_env(os.environ))
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(rc)              # (the stuck thread would hold a normal interpreter exit up)


def retry_env(environ):
    """Environment of the host-staged retry child.  Every rank derives the SAME rendezvous port from the original one
    (PF_BENCH_RETRY_PORT if the launcher reserved one, else MASTER_PORT + 1), and everything torch.distributed.run's agent
    left behind is dropped: with TORCHELASTIC_USE_AGENT_STORE=True inherited, every rank -- rank 0 included -- would connect
    to the agent's store as a CLIENT on the new port, where nobody listens.  Without it rank 0 hosts the store itself."""
    env = {k: v for k, v in environ.items() if not k.startswith("TORCHELASTIC_") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    port = environ.get("PF_BENCH_RETRY_PORT") or str(int(environ.get("MASTER_PORT", "29500")) + 1)
    env.update(PF_BENCH_TRANSPORT="torch", PF_BENCH_RETRIED="1", MASTER_PORT=str(port), MASTER_ADDR=environ.get("MASTER_ADDR", "127.0.0.1"))
    env.pop("PF_BENCH_RETRY_PORT", None)
    return env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mpa512", choices=WORKLOADS)
    ap.add_argument("--agents", type=int, default=0, help="agents per GPU (default: the config's)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--extra-cpu-seconds", type=float, default=2.0, help="budget of each extra workload's cpu_baseline legs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only (default at N>1 and for non-default workloads)")
    ap.add_argument("--phase1-only", action="store_true", help="mpa512: the round-1 protocol (every step a phase-1 iteration)")
    ap.add_argument("--pso-sync", action="store_true", help="pso512: one batch per sweep (sweep-start gbest) instead of the exact asynchronous mode")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + PF_BENCH_SHARE_GPU=1 rehearses N ranks on one GPU (exchange logic only)")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the 1 GiB device-to-device copies behind roofline.copy_GBs_measured "
                    "(profile runs: they show up as eleven copyBuffer kernels of ~0.4 ms that are not the workload's)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)   # internal: "workload:lo:hi" slice for the all-cores CPU leg
    a = ap.parse_args()
    if a.cpu_worker:                                                       # a child of cpu_baseline(): no GPU, no torch
        wl, lo, hi, k_ = a.cpu_worker.split(":")
        from pathfit import env as env_
        print(json.dumps(_cpu_slice(wl, env_.bench_grid(gsize_of(wl)), a.seed, int(lo), int(hi), a.cpu_seconds, K=int(k_))), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:                      # not under torchrun: start the N ranks ourselves
        rc, line = launch_ranks(a.gpus, sys.argv[1:])
        if line is not None:
            print(line, flush=True)
        sys.exit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    try:
        if os.environ.get("PF_BENCH_NOTORCH"):         # diagnostic (N=1 only): leave torch out of the process
            raise ImportError
        import torch as _t
        torch = _t
    except Exception:
        torch = None
    share_gpu = os.environ.get("PF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_.init_process_group(a.backend, rank=rank, world_size=world)
        dist = dist_

    import pathfit
    from pathfit import env
    from pathfit.dist import Comm

    # N > 1: RCCL over xGMI bound directly (pf_comm_*, device pointers on the engine's stream); torch.distributed only ships the
    # 128-byte unique id and, below, reduces the wall time.  PF_BENCH_TRANSPORT=torch keeps the host-staged torch collectives.
    # ("rccl!" tries the direct binding whatever the torch backend is: the rehearsal of its failure path on one shared GPU)
    tr_env = os.environ.get("PF_BENCH_TRANSPORT", "rccl")       # (a retry after a stuck bootstrap sets "torch")
    use_rccl = dist is not None and ((a.backend == "nccl" and tr_env == "rccl") or tr_env == "rccl!")
    comm = Comm(dist, torch.device("cuda", local_rank) if (dist is not None and a.backend == "nccl") else None,
                transport="rccl" if use_rccl else None)
    K, W = a.steps, a.warmup

    def run_one(name, K_, W_):
        grid = env.bench_grid(gsize_of(name))
        eng = pathfit.Engine(grid, device=local_rank)
        if comm.transport == "rccl" and comm.engine is not None and comm.engine is not eng:
            comm.attach(eng)        # a later leg's engine: a fresh communicator on its handle (the binding itself was checked on the first)
        if comm.transport == "rccl" and comm.engine is None:
            # the direct binding is checked with one collective, under a timeout, and the verdict is agreed by all ranks:
            # if it cannot start everywhere, every rank uses the host-staged torch collectives instead
            if not comm.attach_checked(eng, timeout=float(os.environ.get("PF_COMM_TIMEOUT", "120"))):
                print(f"[bench] pf_comm (RCCL direct) unavailable (rank {rank}: {comm.attach_error!r}); all ranks fall back to "
                      "torch.distributed" + (" in fresh processes (a bootstrap thread is stuck)" if comm.attach_stuck else ""), file=sys.stderr)
                if comm.attach_stuck:
                    if os.environ.get("PF_BENCH_RETRIED"):
                        sys.stdout.flush(); sys.stderr.flush()
                        os._exit(RC_RCCL_STUCK)
                    _retry_host_staged(a)       # does not return

        def sync_all():
            if torch is not None and torch.cuda.is_available():
                torch.cuda.synchronize()
            eng._ck(eng.L.pf_sync(eng.h))
            comm.barrier()
        sync_all()             # torch's lazy device initialisation happens here, before the warm-up
        run = Run(name, eng, grid, comm, a, rank, world, K_, W_)
        if comm.engine is None:
            comm.engine = eng      # (host-staged transports: the engine the exchange buffers live on)
        comm.timed = world > 1
        value, dt, roof = measure(run, eng, comm, sync_all, world, torch, dist, local_rank, a.backend)
        if world > 1:
            # the exchange of the timed region (the spans are folded once, after the region: nothing synchronises inside it)
            b0, c0 = run.x0
            run.cfg["exchange"] = {"transport": comm.transport, "bytes_per_rank_per_step": int((comm.bytes_moved - b0) / max(K_, 1)),
                                   "calls_per_step": round((comm.calls - c0) / max(K_, 1), 2),
                                   "exchange_ms_per_step": round(comm.exchange_ms(reset=True) / max(K_, 1), 4),
                                   "timer": "HIP events on the engine's stream around every pf_comm_* call (rank 0)" if comm.transport == "rccl"
                                            else "host clock around every host-staged torch.distributed call, waits for the slower rank included (rank 0)"}
        return run, eng, grid, value, dt, roof

    run, eng, grid, value, dt, roof = run_one(a.workload, K, W)
    head_cfg = run.cfg

    if roof is not None and rank == 0 and not a.no_copy_probe:
        # the nominal 8 TB/s next to what a plain device-to-device copy reaches on this GPU, same run
        try:
            nb = 1 << 30
            src, dst = eng.buf(nb, np.uint8), eng.buf(nb, np.uint8)
            src.zero(); dst.zero()
            eng.d2d(dst.ptr, src.ptr, nb); eng._ck(eng.L.pf_sync(eng.h))
            t1 = time.perf_counter()
            for _ in range(10):
                eng.d2d(dst.ptr, src.ptr, nb)
            eng._ck(eng.L.pf_sync(eng.h))
            roof["copy_GBs_measured"] = round(10 * 2 * nb / (time.perf_counter() - t1) / 1e9, 1)   # read + write
            src.free(); dst.free()
        except Exception:
            pass

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:      # the CPU baseline is an N=1 figure (the other ranks would only wait for it)
        check = os.environ.get("PF_BENCH_CHECK") == "1"
        kept = [] if check else None
        cpu = cpu_baseline(a.workload, grid, a.seed, a.cpu_seconds, K=max(K, 1), keep=kept)
        if check:       # the oracle's sample is not thrown away: it must BE what the device computes for the same agents
            cpu["checked_against_device"] = check_sample(a.workload, grid, eng, a.seed, max(K, 1), kept)

    extra = None
    out = None
    if rank == 0:
        out = {"metric": "agent-fitness-evals/sec on 512x512 grid", "value": round(value, 2), "unit": "evals/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": head_cfg, "roofline": roof, "cpu_baseline": cpu}
    if world > 1 and a.workload == "mpa512" and not a.no_extra and not a.agents:
        # BASELINE.json defines configs[3] (PSO + GA, 16 384 agents over 8 GPUs = 2 048 per GPU) and configs[4] (MAACO 65 536 ants on
        # G1024 over 8 = 8 192 per GPU) on several GPUs: at N > 1 they run after the headline as short legs on EVERY rank (their
        # exchanges are collectives), each with its own `config.exchange`.  A watchdog bounds the whole section: a leg that hangs
        # (a rank lost inside a collective) costs the extras, never the headline line.
        import threading
        extra = {}
        if out is not None:
            out["extra"] = extra
        budget = float(os.environ.get("PF_BENCH_EXTRA_BUDGET", "240"))
        done = threading.Event()

        def watchdog():
            if not done.wait(budget):
                try:
                    if out is not None:
                        note = f"the multi-GPU extras did not finish within {budget:.0f} s: line printed by the watchdog"
                        try:                       # (the main thread may be filling `extra` this very moment)
                            line = json.dumps(dict(out, extra=dict(dict(extra), _error=note)))
                        except Exception:
                            line = json.dumps(dict({k: v for k, v in out.items() if k != "extra"}, extra={"_error": note}))
                        print(line, flush=True)
                    sys.stderr.flush()
                finally:
                    os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        run = None
        eng.close()
        for name, k_, w_ in (("maaco1024", 4, 1), ("pso512", 2, 1), ("ga512", 2, 1)):
            try:
                r2, e2, _, v2, dt2, roof2 = run_one(name, k_, w_)
                extra[name] = {"value": round(v2, 2), "unit": "evals/s", "n_gpus": world, "steps": k_, "warmup": w_,
                               "ms_per_step": round(dt2 / k_ * 1e3, 3), "roofline": roof2, "config": r2.cfg}
                r2 = None
                e2.close()
            except Exception as ex:
                extra[name] = {"error": repr(ex)[:300]}
                break                  # (the ranks may no longer agree on what comes next: stop the section here)
        done.set()
    if world == 1 and rank == 0 and a.workload == "mpa512" and not a.no_extra and not a.agents:
        # every other BASELINE workload in the same process, short runs (the judge asked for them in the driver-run line)
        extra = {}
        run = None
        eng.close()
        for name, k_, w_ in (("maaco128", 40, 3), ("maaco512", 20, 2), ("maaco1024", 20, 2), ("ga512", 3, 1), ("pso512", 2, 1),
                             ("astar1024", 2, 1)):       # (the MAACO legs are milliseconds per step: enough steps that one slow step does not show)
            try:
                r2, e2, _, v2, dt2, roof2 = run_one(name, k_, w_)
                extra[name] = {"value": round(v2, 2), "unit": "evals/s", "steps": k_, "warmup": w_, "ms_per_step": round(dt2 / k_ * 1e3, 3),
                               "roofline": roof2, "config": r2.cfg}
                r2 = None
                e2.close()
                if not a.no_cpu:                        # the CPU side of every workload, 1 core and all cores, ~2 s each
                    extra[name]["cpu_baseline"] = cpu_baseline(name, env.bench_grid(gsize_of(name)), a.seed, a.extra_cpu_seconds)
            except Exception as ex:                     # an extra must never cost the headline
                extra[name] = {"error": repr(ex)[:300]}

    if rank == 0:
        if extra is not None:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def check_sample(workload, grid, eng, seed, K, kept):
    """PF_BENCH_CHECK=1: the agents the cpu_baseline leg evaluated with the oracle (iteration 1 of the same workload, same seed and
    parameters) against a fresh device instance's iteration 1 -- fitness / length bit for bit, paths by length and crc32.
    Raises on the first mismatch; returns how many agents were compared.  (The checker reads the product's result, never the
    other way round.)"""
    import zlib
    import pathfit
    if not kept:
        return 0
    if workload == "mpa512":
        m = pathfit.MPA(grid, 4096, K, engine=eng, seed=seed, **MPA_MAIN)
        m._sort()
        order = m.order.copy()
        m.step(1)
        cells, lens, stats = m.d_cells.download(), m.d_len.download(), m.d_stats.download()
        for n, fit, ln, crc in kept:
            sl = order[n]
            got = (float(stats[sl, 4]), int(lens[sl]), zlib.crc32(np.ascontiguousarray(cells[sl, :lens[sl]], np.int32).tobytes()))
            if got != (fit, ln, crc):
                raise AssertionError(f"PF_BENCH_CHECK: predator {n} of iteration 1: device {got} != oracle {(fit, ln, crc)}")
    elif workload.startswith("maaco"):
        n_ants = CPU_TOTAL[workload]
        m = pathfit.MAACO(grid, n_ants, 100, engine=eng, seed=seed, **MAACO_MAIN)
        m.iterate_dev(1)
        dc, dl, dp, _, _ = m.walk_bufs()
        cells, lens, plen = dc.download().reshape(n_ants, -1), dl.download(), dp.download()
        for n, L, ln, crc in kept:
            got = (float(plen[n]), int(lens[n]), zlib.crc32(np.ascontiguousarray(cells[n, :lens[n]], np.int32).tobytes()))
            if got != (L, ln, crc):
                raise AssertionError(f"PF_BENCH_CHECK: ant {n} of iteration 1: device {got} != oracle {(L, ln, crc)}")
    else:
        return 0
    return len(kept)


CPU_TOTAL = {"mpa512": 4096, "maaco128": 256, "maaco512": 16384, "maaco1024": 8192, "astar1024": 8192, "ga512": 2048, "pso512": 2048}


def cpu_info():
    """CPU model string (/proc/cpuinfo), hardware concurrency and the cores this process may use."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"model": model, "hardware_concurrency": os.cpu_count(), "usable_cores": usable}


CPU_MIN_SECONDS = 0.5      # a sample shorter than this is below timer / start-up noise: small workloads wrap around until it is reached


def _cpu_slice(workload, grid, seed, lo, hi, budget_s, K=15, keep=None):
    """Agents [lo, hi) of the workload's first step on the CPU oracle (the C port of the reference's algorithm; checker
    code), at most budget_s seconds -> {count, seconds, what}.  A slice that is done in less than CPU_MIN_SECONDS starts over
    (the same agents again) until that much time has passed, so that no rate is quoted from a few milliseconds.  `keep`
    (a list) receives, for the agents of the FIRST pass, what PF_BENCH_CHECK=1 compares with the device: (agent, fitness,
    path length, crc32 of the cell path)."""
    import zlib
    import pf_oracle as po
    import pf_loops
    orc = po.Oracle(grid)
    s, t = 0, grid.size - 1
    span = hi - lo

    def loop(fn):
        k = 0
        t0 = time.perf_counter()
        while span > 0:
            el = time.perf_counter() - t0
            if el >= budget_s or (k >= span and el >= CPU_MIN_SECONDS):
                break
            fn(lo + k % span, k < span)
            k += 1
        return k, time.perf_counter() - t0

    if workload == "mpa512":
        ref = pf_loops.MpaOracle(orc, s, t, 4096, K, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1, w_safe=0.8,
                                 min_safe=1.8, diag_pen=100.0, seed=seed)
        ref._sort()
        elite = ref.pop[0]
        CF = (1.0 - 1 / K) ** (2.0 / K)

        def one(n, first):
            cand = ref.phase_candidate(1, n, elite, CF)
            ind = cand if cand[1][4] < ref.pop[n][1][4] else ref.pop[n]
            ind = ref.fads(1, n, ind, CF)
            if keep is not None and first:
                keep.append((n, float(ind[1][4]), int(len(ind[0])), zlib.crc32(np.ascontiguousarray(ind[0], np.int32).tobytes())))
        what = f"predators of iteration 1 of a {K}-iteration run (phase sweep + memory + FADs)"
    elif workload.startswith("maaco"):
        P = po.MaacoParams(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn=1.0, wh_max=0.9, wh_min=0.2, k_h=0.9, q0_initial=0.5,
                           C0=0.1, num_iterations=100)
        tau, dist = orc.maaco_init(s, t, 0.1)

        def one(n, first):
            p, L, _, _ = orc.maaco_walk(s, t, P, tau, dist, 1, seed, n)
            if keep is not None and first:
                keep.append((n, float(L), int(len(p)), zlib.crc32(np.ascontiguousarray(p, np.int32).tobytes())))
        what = "ant walks of iteration 1"
    elif workload == "astar1024":
        rng = np.random.default_rng(seed)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        ss, tt = rng.choice(free, 8192), rng.choice(free, 8192)

        def one(n, first):
            orc.astar(int(ss[n]), int(tt[n]), None, 0)
        what = "pairs (AStarSolver connector)"
    else:
        rng = np.random.default_rng(seed)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        wp = rng.choice(free, (2048, 5)).astype(np.int32)

        def one(n, first):
            p, _ = orc.decode(s, t, wp[n])
            orc.score(p, 0, 0.3, 0.8, 1.8, True, 100.0)
        what = "chromosomes (W=5 decode + score)"
    count, secs = loop(one)
    return {"count": count, "seconds": secs, "what": what, "wrapped": count > span}


def _cpu_all_cores(workload, seed, budget_s, K=15):
    """The same sample spread over every host core this process may use: one child process per core (fresh
    interpreters, no GPU), each taking a contiguous slice of the workload's agents."""
    import subprocess
    cores = max(1, min(cpu_info()["usable_cores"], 16))   # a one-GPU box's CPU share
    total = CPU_TOTAL[workload]
    per = -(-total // cores)
    procs = []
    for k in range(cores):
        lo, hi = k * per, min(total, (k + 1) * per)
        if lo >= hi:
            break
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{workload}:{lo}:{hi}:{K}", "--seed", str(seed),
                                       "--cpu-seconds", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    tot, secs = 0, 0.0
    for pr in procs:
        out, _ = pr.communicate(timeout=budget_s * 4 + 180)
        r = json.loads(out.strip().splitlines()[-1])
        tot += r["count"]; secs = max(secs, r["seconds"])
    return {"value": round(tot / secs, 3), "cores": len(procs), "seconds": round(secs, 2),
            "sample": f"{tot} evaluations in {len(procs)} processes (each its slice of the agents, repeated while shorter than {CPU_MIN_SECONDS} s)"}


def cpu_baseline(workload, grid, seed, budget_s, all_cores=True, K=15, keep=None):
    """Time the CPU oracle (the C port of the reference's algorithm; checker code, kind 'port') on a bounded sample of the
    SAME workload: single thread and -- SURVEY.md 8d -- on all host cores, with the CPU model and core counts."""
    r = _cpu_slice(workload, grid, seed, 0, CPU_TOTAL[workload], budget_s, K=K, keep=keep)
    out = {"value": round(r["count"] / r["seconds"], 3), "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": (f"all {CPU_TOTAL[workload]} {r['what']}, repeated to {r['count']} evaluations (>= {CPU_MIN_SECONDS} s)" if r["wrapped"] else
                      f"first {r['count']} {r['what']}") + ", same grid/params/seed",
           "seconds": round(r["seconds"], 2), "cpu": cpu_info()}
    if all_cores:
        try:
            out["all_cores"] = _cpu_all_cores(workload, seed, min(budget_s, 8.0), K)
        except Exception as e:                      # the single-core figure stands on its own
            out["all_cores"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
