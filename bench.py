#!/usr/bin/env python3
"""bench.py -- agent-fitness-evals/sec on a 512x512 grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload mpa512|maaco512|pso512|ga512|maaco128|maaco1024|astar1024]

Workload at N=1 (default `mpa512`) = BASELINE.json configs[2]: MPA, 4096
predators on G512 (np.kron 2x of the reference's 256x256 map), main.py:44-52
parameters.  A "step" is one MPA iteration's population evaluate-and-update
hot path: per predator propose a target cell, stitch with two A* connectors,
score, greedy memory, FADs.  One eval = one predator's pass through it.  Inputs
(grid, population) are resident in HBM when the timed region starts.  For N>1
every rank owns 4096 predators (weak scaling); the only exchange is the
fitness all_gather + elite broadcast per iteration (pathfit/dist.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
for the dominant kernel (k_mpa_sweep / k_maaco_walk8 / k_decode_batch; HIP-event
timed inside the library on its own stream; `copy_GBs_measured` = a plain 1 GiB
device-to-device copy in the same run) and `cpu_baseline` (the CPU oracle port,
bounded sample, 1 core; for mpa512 also `all_cores`: the same sample over up to
16 child processes).
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


def astar_bytes(c):
    """Algorithmic bytes of the A* work of one launch: per pop a 24 B heap entry + the 3x3
    occupancy window (9 B) + 1 B closed mark, 8 B per examined neighbour, 33 B per push; 4 B per emitted cell."""
    return 34 * c["pops"] + 8 * c["nbr_examined"] + 33 * c["pushes"] + 4 * c["path_cells"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mpa512", choices=["mpa512", "maaco512", "pso512", "ga512", "maaco128", "maaco1024", "astar1024"])
    ap.add_argument("--agents", type=int, default=0, help="agents per GPU (default: the config's)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + PF_BENCH_SHARE_GPU=1 rehearses N ranks on one GPU (exchange logic only)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)   # internal: "lo:hi" slice for the all-cores CPU leg
    a = ap.parse_args()
    if a.cpu_worker:                                                       # a child of cpu_baseline(): no GPU, no torch
        lo, hi = (int(x) for x in a.cpu_worker.split(":"))
        from pathfit import env as env_
        print(json.dumps(_mpa_cpu_slice(env_.bench_grid(512), a.seed, lo, hi, a.cpu_seconds)), flush=True)
        return

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
    from pathfit.dist import Comm, ShardedMPA, ShardedMAACO

    comm = Comm(dist, torch.device("cuda", local_rank) if (dist is not None and a.backend == "nccl") else None)
    gsize = 128 if a.workload.endswith("128") else (1024 if a.workload.endswith("1024") else 512)
    grid = env.bench_grid(gsize)
    eng = pathfit.Engine(grid, device=local_rank)
    K, W = a.steps, a.warmup

    def sync_all():
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        eng._ck(eng.L.pf_sync(eng.h))
        comm.barrier()

    kern_ms, kern_bytes, launches = 0.0, 0.0, 0
    per_gpu = a.agents
    if a.workload == "mpa512":
        per_gpu = per_gpu or 4096
        total = per_gpu * world
        iters = max(K + W, 3 * (K + W))      # keeps the whole run inside phase 1 (iter <= T/3), like early MPA iterations
        sm = ShardedMPA(comm, lambda n: pathfit.MPA(grid, total, iters, engine=eng, seed=a.seed, n_local=n, **MPA_MAIN), total)
        dominant = "k_mpa_sweep"
        it = 0

        def step():
            nonlocal it, kern_ms, kern_bytes, launches
            it += 1
            e = eng
            # instrument the phase launch (dominant kernel) through the library's HIP-event timer + counters
            orig = e.mpa_iter

            def timed_iter(*args, **kw):
                nonlocal kern_ms, kern_bytes, launches
                orig(*args, **kw)
                c = e.counters()
                kern_ms += e.last_kernel_ms(); kern_bytes += astar_bytes(c); launches += 1
                cfg["rebuilds_proven_rejected_and_skipped"] = cfg.get("rebuilds_proven_rejected_and_skipped", 0) + int(c["pruned_rebuilds"])
            e.mpa_iter = timed_iter
            try:
                sm.step(it)
            finally:
                e.mpa_iter = orig
        cfg = {"workload": "MPA 4096 predators/GPU, 512x512 G512 (BASELINE.json configs[2]), main.py:44-52 params, phase-1 iterations",
               "agents_per_gpu": per_gpu, "grid": "G512=kron2(G256)", "grid_sha256": env.grid_hash(grid)[:16]}
    elif a.workload.startswith("maaco"):
        # maaco512: 16384 ants/GPU; maaco128 = BASELINE.json configs[1] (256 ants, G128); maaco1024 = configs[4]'s per-GPU
        # share (8192 ants, G1024)
        per_gpu = per_gpu or {128: 256, 512: 16384, 1024: 8192}[gsize]
        total = per_gpu * world
        sm = ShardedMAACO(comm, lambda: pathfit.MAACO(grid, total, 100, engine=eng, seed=a.seed, **MAACO_MAIN), total)
        dominant = "k_maaco_walk8" if per_gpu >= 2048 else "k_maaco_walk"   # 8 ants per wavefront from 2048 ants up
        it = 0

        def step():
            nonlocal it, kern_ms, kern_bytes, launches
            it += 1
            orig = eng.maaco_walk

            def timed_walk(*args, **kw):
                nonlocal kern_ms, kern_bytes, launches
                orig(*args, **kw)
                c = eng.counters()
                kern_ms += eng.last_kernel_ms(); launches += 1
                # 9 B window + 8 B tabu probes + 16 B per candidate (tau + eta) + 5 B path/tabu write
                kern_bytes += 22 * c["steps"] + 16 * c["candidates"]
            eng.maaco_walk = timed_walk
            try:
                sm.step(it)
            finally:
                eng.maaco_walk = orig
        cfg = {"workload": f"MAACO ants/GPU on {gsize}x{gsize}, main.py:34-38 params (walk + ordered pheromone update)",
               "agents_per_gpu": per_gpu, "grid": {128: "G128=random_blocks(seed 128)", 512: "G512=kron2(G256)", 1024: "G1024=kron4(G256)"}[gsize],
               "grid_sha256": env.grid_hash(grid)[:16]}
    elif a.workload == "astar1024":
        # BASELINE.json configs[4], second part: a standalone batch of seeded (start, target) pairs on G1024 through the
        # AStarSolver connector (8192 pairs per GPU = 65536 over 8); one eval = one connector solve + path emit
        per_gpu = per_gpu or 8192
        rng = np.random.default_rng(a.seed + rank)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        cap = 16 * 1024 + 64
        d_s, d_t = eng.put(rng.choice(free, per_gpu).astype(np.int32)), eng.put(rng.choice(free, per_gpu).astype(np.int32))
        d_cells, d_len, d_st = eng.buf((per_gpu, cap), np.int32), eng.buf(per_gpu, np.int32), eng.buf(per_gpu, np.int32)
        dominant = "k_astar_batch"

        def step():
            nonlocal kern_ms, kern_bytes, launches
            eng.astar_batch(0, d_s, d_t, per_gpu, cap, d_cells, d_len, d_st)
            kern_ms += eng.last_kernel_ms(); kern_bytes += astar_bytes(eng.counters()); launches += 1
        cfg = {"workload": "A* connector batch (AStarSolver semantics), 8192 uniform free-cell pairs/GPU, G1024 (BASELINE.json configs[4] K2a batch)",
               "agents_per_gpu": per_gpu, "grid": "G1024=kron4(G256)", "grid_sha256": env.grid_hash(grid)[:16]}
    else:
        per_gpu = per_gpu or 2048                     # BASELINE.json configs[3]: 16384 over 8 GPUs
        Wp = 5
        rng = np.random.default_rng(a.seed + rank)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        sp = pathfit.score_params(0, True, 0.3, 0.8, 1.8, 100.0)
        cap = 16 * 1024 + 64
        d_cells, d_len, d_st, d_stats = eng.buf((per_gpu, cap), np.int32), eng.buf(per_gpu, np.int32), eng.buf(per_gpu, np.int32), eng.buf((per_gpu, 5), np.float64)
        if a.workload == "ga512":
            d_wp = eng.put(rng.choice(free, (per_gpu, Wp)).astype(np.int32).reshape(-1)); d_pos = None
        else:
            pos = rng.uniform(0, 511, (per_gpu, Wp, 2)); vel = rng.uniform(-15, 15, (per_gpu, Wp, 2))
            d_pos, d_vel, d_pb, d_gb = eng.put(pos), eng.put(vel), eng.put(pos), eng.put(pos[0]); d_wp = None
        dominant = "k_decode_batch"
        it = 0

        def step():
            nonlocal it, kern_ms, kern_bytes, launches
            it += 1
            if d_pos is not None:
                eng.pso_update(per_gpu, Wp, 0.7, 1.5, 1.5, 76.8, d_pos, d_vel, d_pb, d_gb, a.seed, it, rank * per_gpu)
            eng.decode_batch(per_gpu, Wp, 0, 512 * 512 - 1, cap, d_cells, d_len, d_st, d_wp, d_pos, sp, d_stats)
            kern_ms += eng.last_kernel_ms(); kern_bytes += astar_bytes(eng.counters()); launches += 1
            if world > 1:       # gbest MINLOC exchange (C2)
                comm.all_gather_concat(d_stats.download()[:, 4], [per_gpu] * world)
        cfg = {"workload": f"{'GA' if a.workload == 'ga512' else 'PSO'} chained-waypoint decode+score, W=5, 2048 agents/GPU, G512 "
                           "(BASELINE.json configs[3] per-GPU share)", "agents_per_gpu": per_gpu,
               "grid_sha256": env.grid_hash(grid)[:16]}

    sync_all()             # torch's lazy device initialisation happens here, before the warm-up
    if a.workload.startswith("maaco"):
        step()             # set-up, not warm-up: with torch in the process the SECOND MAACO iteration pays a one-off ~40 ms
                           # (first iteration allocates the walk / visit-bit buffers; not seen without torch, PF_BENCH_NOTORCH=1)
    for _ in range(W):
        step()
    kern_ms, kern_bytes, launches = 0.0, 0.0, 0
    cfg.pop("rebuilds_proven_rejected_and_skipped", None)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(K):
        if os.environ.get("PF_BENCH_TRACE"):           # diagnostic: host wall time of every step, to stderr
            ts = time.perf_counter(); step(); print(f"step {1e3 * (time.perf_counter() - ts):.2f} ms", file=sys.stderr, flush=True)
        else:
            step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank) if a.backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_evals = per_gpu * world * K
    value = total_evals / dt

    roof = None
    if launches:
        avg_ms = kern_ms / launches
        achieved = (kern_bytes / launches) / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                "avg_launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": int(kern_bytes / launches)}

    if roof is not None:
        # HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/r01_traffic.json:
        # FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KB -> B, FETCH x2 per MI355X_MICROARCH.md)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if a.workload in tj and tj[a.workload]["kernel"] == dominant:
                roof["traffic"] = int(tj[a.workload]["traffic_bytes_per_launch"])
                roof["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, offline)"
        except Exception:
            pass

    if roof is not None and rank == 0:
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
        cpu = cpu_baseline(a.workload, grid, a.seed, a.cpu_seconds)

    if rank == 0:
        out = {"metric": "agent-fitness-evals/sec on 512x512 grid", "value": round(value, 2), "unit": "evals/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfg,
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _mpa_cpu_slice(grid, seed, lo, hi, budget_s):
    """Predators [lo, hi) of MPA iteration 1 on the CPU oracle, at most budget_s seconds -> (count, seconds)."""
    import pf_oracle as po
    import pf_loops
    orc = po.Oracle(grid)
    ref = pf_loops.MpaOracle(orc, 0, grid.size - 1, 4096, 15, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1, w_safe=0.8,
                             min_safe=1.8, diag_pen=100.0, seed=seed)
    ref._sort()
    elite = ref.pop[0]
    CF = (1.0 - 1 / 15) ** (2.0 / 15)
    t0 = time.perf_counter()
    n = lo
    while time.perf_counter() - t0 < budget_s and n < hi:
        cand = ref.phase_candidate(1, n, elite, CF)
        ind = cand if cand[1][4] < ref.pop[n][1][4] else ref.pop[n]
        ref.fads(1, n, ind, CF)
        n += 1
    return {"count": n - lo, "seconds": time.perf_counter() - t0}


def _mpa_cpu_all_cores(seed, budget_s):
    """The same sample spread over every host core this process may use: one child process per core (fresh
    interpreters, no GPU), each taking a contiguous slice of the 4096 predators."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                  # a one-GPU box's CPU share
    per = -(-4096 // cores)
    procs = []
    for k in range(cores):
        lo, hi = k * per, min(4096, (k + 1) * per)
        if lo >= hi:
            break
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{lo}:{hi}", "--seed", str(seed),
                                       "--cpu-seconds", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    tot, secs = 0, 0.0
    for pr in procs:
        out, _ = pr.communicate(timeout=budget_s * 4 + 120)
        r = json.loads(out.strip().splitlines()[-1])
        tot += r["count"]; secs = max(secs, r["seconds"])
    return {"value": round(tot / secs, 3), "cores": len(procs), "sample": f"{tot} predators of iteration 1 in {len(procs)} processes"}


def cpu_baseline(workload, grid, seed, budget_s):
    """Time the CPU oracle (the C port of the reference's algorithm; checker code, kind 'port') on a bounded
    sample of the SAME workload, single thread (mpa512: also on all host cores)."""
    import pf_oracle as po
    import pf_loops
    orc = po.Oracle(grid)
    s, t = 0, grid.size - 1
    t0 = time.perf_counter()
    n = 0
    if workload == "mpa512":
        ref = pf_loops.MpaOracle(orc, s, t, 4096, 15, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1, w_safe=0.8,
                                 min_safe=1.8, diag_pen=100.0, seed=seed)
        ref._sort()
        elite = ref.pop[0]
        CF = (1.0 - 1 / 15) ** (2.0 / 15)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 4096:
            cand = ref.phase_candidate(1, n, elite, CF)
            ind = cand if cand[1][4] < ref.pop[n][1][4] else ref.pop[n]
            ref.fads(1, n, ind, CF)
            n += 1
        sample = f"first {n} predators of iteration 1 (phase sweep + memory + FADs), same grid/params/seed"
    elif workload.startswith("maaco"):
        P = po.MaacoParams(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn=1.0, wh_max=0.9, wh_min=0.2, k_h=0.9, q0_initial=0.5,
                           C0=0.1, num_iterations=100)
        tau, dist = orc.maaco_init(s, t, 0.1)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 16384:
            orc.maaco_walk(s, t, P, tau, dist, 1, seed, n)
            n += 1
        sample = f"first {n} ant walks of iteration 1, same grid/params/seed"
    elif workload == "astar1024":
        rng = np.random.default_rng(seed)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        ss, tt = rng.choice(free, 8192), rng.choice(free, 8192)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 8192:
            orc.astar(int(ss[n]), int(tt[n]), None, 0)
            n += 1
        sample = f"first {n} pairs (AStarSolver connector), same grid and seed"
    else:
        rng = np.random.default_rng(seed)
        free = np.flatnonzero(grid.reshape(-1) != 1)
        wp = rng.choice(free, (2048, 5)).astype(np.int32)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 2048:
            p, _ = orc.decode(s, t, wp[n])
            orc.score(p, 0, 0.3, 0.8, 1.8, True, 100.0)
            n += 1
        sample = f"first {n} chromosomes (W=5 decode + score), same grid and seed"
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 3), "unit": "evals/s", "cores": 1, "kind": "port", "sample": sample,
           "seconds": round(dt, 2)}
    if workload == "mpa512":
        try:
            out["all_cores"] = _mpa_cpu_all_cores(seed, min(budget_s, 10.0))
        except Exception as e:                      # the single-core figure stands on its own
            out["all_cores"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
