"""N>1 path on CPU: world_size-2 gloo ranks run the sharded MAACO loop (pathfit/dist.py) over oracle-backed
stand-ins; the result must equal the single-rank loop bit for bit (strict ordered pheromone fold), i.e.
partitioning the population does not change the answer."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(alpha=1.0, beta=2.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)


def _worker(rank, world, port, out_dir, strict, ants=21, chunks=8, cap_factor=None):
    for p in (os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import golden_io as gio
    from dist_fakes import FakeMAACO
    from pathfit.dist import Comm, ShardedMAACO
    g, s, t = gio.grid("fig7")
    comm = Comm(dist, None)           # transport: gloo (host staged) -- the exchange logic is transport independent
    comm.timed = True                 # bench.py's exchange accounting: every collective is timed (host clock for this transport)
    sm = ShardedMAACO(comm, lambda: FakeMAACO(g, s, t, ants, 5, 7, **KW), ants, strict=strict, chunks=chunks)
    if cap_factor:                    # this rank's walk "overflowed and grew its path rows" (MAACO.walk_iteration_dev): path_cap differs per rank
        sm.local.path_cap *= cap_factor[rank]
    path, length, turns = sm.solve_path_planning()
    ms = comm.exchange_ms(reset=True)
    assert ms > 0.0 and comm.calls > 0 and comm.bytes_moved > 0 and comm.exchange_ms() == 0.0, (ms, comm.calls)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), path=np.array(path), length=length, turns=turns,
             tau=sm.local.engine.maaco_get_pheromone(), curve=np.array(sm.local.convergence_curve_data, float))
    dist.barrier()
    dist.destroy_process_group()


def _single(strict=True, ants=21):
    import golden_io as gio
    from dist_fakes import FakeMAACO
    from pathfit.dist import Comm, ShardedMAACO
    g, s, t = gio.grid("fig7")
    sm = ShardedMAACO(Comm(None), lambda: FakeMAACO(g, s, t, ants, 5, 7, **KW), ants, strict=strict)
    path, length, turns = sm.solve_path_planning()
    return np.array(path), length, turns, sm.local.engine.maaco_get_pheromone(), np.array(sm.local.convergence_curve_data, float)


@pytest.mark.parametrize("strict", [True, False])
def test_sharded_maaco_two_ranks_equals_one(tmp_path, strict):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + (1 if strict else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), strict), nprocs=2, join=True)
    ref = _single(strict)
    for r in (0, 1):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["path"], ref[0]) and float(z["length"]) == ref[1] and float(z["turns"]) == ref[2]
        assert np.array_equal(z["curve"], ref[4])
        if strict:
            assert np.array_equal(z["tau"], ref[3])          # ordered fold == sequential deposits, bit for bit
        else:
            assert np.allclose(z["tau"], ref[3], rtol=1e-12)  # all_reduce(SUM): ulp-level deviation only


def test_sharded_maaco_three_ranks_uneven_blocks_pipelined_fold(tmp_path):
    """20 ants over 3 ranks (7 + 7 + 6: the uneven all_gather path) with the ordered fold pipelined over 5 row chunks:
    rank 1 folds chunk j while rank 2 folds chunk j - 1; the matrix every rank ends with is the sequential one."""
    import torch.multiprocessing as mp
    port = 27300 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(3, port, str(tmp_path), True, 20, 5), nprocs=3, join=True)
    ref = _single(True, 20)
    for r in (0, 1, 2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["path"], ref[0]) and float(z["length"]) == ref[1] and float(z["turns"]) == ref[2]
        assert np.array_equal(z["curve"], ref[4]) and np.array_equal(z["tau"], ref[3])


def test_sharded_maaco_ranks_with_different_path_cap(tmp_path):
    """One rank has grown its path rows (an ant overflowed there, only there): the exchange must not depend on path_cap being
    the same everywhere -- the best row travels as (length, exactly that many cells)."""
    import torch.multiprocessing as mp
    port = 25100 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), True, 21, 8, (1, 4)), nprocs=2, join=True)
    ref = _single(True)
    for r in (0, 1):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["path"], ref[0]) and float(z["length"]) == ref[1] and np.array_equal(z["tau"], ref[3])


def test_single_rank_sharded_equals_oracle_loop():
    import pf_loops, pf_oracle as po, golden_io as gio
    g, s, t = gio.grid("fig7")
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, 21, 5, C0=0.1, seed=7, **KW)
    path, length, turns, tau, curve = _single(True)
    assert [r * 20 + c for r, c in path.tolist()] == list(ref["path"]) and length == ref["length"]
    assert np.array_equal(tau, ref["tau"])


def test_shard_helpers():
    from pathfit.dist import shard_range, owner_of, global_stable_order, maaco_best_scan_host
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert owner_of(5, [4, 3, 3]) == (1, 1) and owner_of(9, [4, 3, 3]) == (2, 2)
    assert list(global_stable_order([3.0, 1.0, 3.0, 1.0])) == [1, 3, 0, 2]
    # sequential scan semantics MAACO.py:343-349: a later near-tie with fewer turns takes the path, not the length
    L, T, i = maaco_best_scan_host(np.array([10.0, 10.0 - 5e-10, 10.0 + 4e-10]), np.array([5, 6, 2]))
    assert (L, T, i) == (10.0 - 5e-10, 2.0, 2)


def test_attach_checked_falls_back_when_the_binding_cannot_start():
    """Comm.attach_checked: a direct (RCCL) binding that fails, or never returns, must leave the communicator on the host-staged
    transport instead of raising or hanging (bench.py's N > 1 start-up).  No GPU: the engine is a stub."""
    import time
    from pathfit.dist import Comm

    class _L:
        def __init__(self, mode):
            self.mode = mode

        def pf_comm_unique_id(self, buf):
            if self.mode == "hang":
                time.sleep(30)
            return 1                                   # "failed"

        def pf_last_error(self, h):
            return b"stub: no rccl here"

    class _Eng:
        def __init__(self, mode):
            self.L, self.h = _L(mode), None

    calls = []

    class _Late:                                       # pf_comm_init returns AFTER the deadline: the helper must not touch the engine
        def pf_comm_unique_id(self, buf):
            time.sleep(0.6)
            return 0

        def pf_comm_init(self, h, rank, world, idb):
            calls.append("init")
            return 0

        def pf_comm_destroy(self, h):
            calls.append("destroy")
            return 0

        def pf_comm_all_gather(self, *a):
            calls.append("probe")
            return 0

    class _LateEng:
        L, h = _Late(), None

        def _ck(self, rc):
            assert rc == 0

        def put(self, *a):
            calls.append("put")

        def buf(self, *a):
            calls.append("buf")

    c = Comm(None, None, transport="rccl")
    assert c.attach_checked(_LateEng(), timeout=0.2) is False and c.attach_stuck
    time.sleep(1.0)
    assert calls == ["init", "destroy"], calls

    for mode, timeout in (("fail", 5.0), ("hang", 0.3)):
        c = Comm(None, None, transport="rccl")
        t0 = time.time()
        ok = c.attach_checked(_Eng(mode), timeout=timeout)
        assert ok is False and c.transport == "torch" and time.time() - t0 < 5.0
        assert c.attach_error is not None and c.attach_stuck == (mode == "hang")
