"""bench.py --gpus N without torchrun: the parent starts N rank processes itself (before anything touches a GPU), relays
rank 0's JSON line and fails loudly when a rank fails or the line reports another rank count.  Driven here with a stub
worker (no GPU, no torch): the rendezvous environment the ranks receive is what torch.distributed.run would set."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "stub_worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launcher_starts_n_ranks_and_relays_rank0_line(tmp_path):
    import bench
    w = _stub(tmp_path, """
        import json, os, sys
        r, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        open(os.path.join(os.environ["STUB_DIR"], f"rank{r}"), "w").write(" ".join(sys.argv[1:]))
        print("noise before the line")
        if r == 0:
            print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": n}), flush=True)
    """)
    rc, line = bench.launch_ranks(3, ["--gpus", "3", "--steps", "2"], worker=w, env_extra={"STUB_DIR": str(tmp_path)}, timeout=60)
    assert rc == 0 and json.loads(line)["n_gpus"] == 3
    for r in range(3):
        assert (tmp_path / f"rank{r}").read_text() == "--gpus 3 --steps 2"


def test_launcher_fails_when_a_rank_fails_or_the_count_is_wrong(tmp_path):
    import bench
    w = _stub(tmp_path, """
        import json, os, sys
        r = int(os.environ["RANK"])
        if r == 0:
            print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2}), flush=True)
        sys.exit(7 if (r == 1 and os.environ.get("STUB_FAIL")) else 0)
    """)
    rc, _ = bench.launch_ranks(2, [], worker=w, env_extra={"STUB_FAIL": "1"}, timeout=60)
    assert rc == 7
    rc, line = bench.launch_ranks(2, [], worker=w, timeout=60)
    assert rc == 0 and line
    rc, line = bench.launch_ranks(3, [], worker=w, timeout=60)        # the line says 2 ranks, 3 were started
    assert rc == 1 and line
    w2 = _stub(tmp_path, "print('no json here')")
    rc, line = bench.launch_ranks(2, [], worker=w2, timeout=60)
    assert rc == 1 and line is None


def test_bench_main_takes_the_launcher_branch_before_any_gpu_import(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must not import torch / pathfit in the parent: a child that finds
    WORLD_SIZE set is a rank.  The ranks fail here (no GPU) -- what is checked is that the parent exits non-zero from the
    launcher, having started them."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PF_BENCH_TRANSPORT"] = "torch"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu",
                         "--no-extra", "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode != 0
    assert "[bench launcher]" in pr.stderr


def test_a_rank_that_dies_early_ends_the_run_at_once(tmp_path):
    """Rank 1 dies at start-up while rank 0 would wait (here: sleep) for minutes, as in a rendezvous nobody else joins: the
    launcher polls all ranks, reports rank 1's code and kills rank 0 within seconds."""
    import time
    import bench
    w = _stub(tmp_path, """
        import os, sys, time
        if int(os.environ["RANK"]) == 1:
            sys.exit(9)
        time.sleep(300)
    """)
    t0 = time.monotonic()
    rc, line = bench.launch_ranks(2, [], worker=w, timeout=120)
    assert rc == 9 and line is None and time.monotonic() - t0 < 30


def test_host_staged_retry_env_drops_the_elastic_agent_store(tmp_path):
    """The retry child after a stuck RCCL bootstrap: under torch.distributed.run the parent's environment carries
    TORCHELASTIC_USE_AGENT_STORE=True, with which every rank would be a TCPStore CLIENT on the new port (nobody listens).  The
    child's environment must not inherit it, and every rank must derive the same port."""
    import bench
    base = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511", "RANK": "1", "WORLD_SIZE": "2", "TORCHELASTIC_USE_AGENT_STORE": "True",
            "TORCHELASTIC_RUN_ID": "x", "TORCHELASTIC_RESTART_COUNT": "0", "TORCHELASTIC_MAX_RESTARTS": "0", "PATH": os.environ.get("PATH", "")}
    e0, e1 = bench.retry_env(dict(base, RANK="0")), bench.retry_env(base)
    for e in (e0, e1):
        assert not any(k.startswith("TORCHELASTIC_") for k in e)
        assert e["PF_BENCH_TRANSPORT"] == "torch" and e["PF_BENCH_RETRIED"] == "1" and e["WORLD_SIZE"] == "2"
    assert e0["MASTER_PORT"] == e1["MASTER_PORT"] == "29512" and e0["RANK"] == "0" and e1["RANK"] == "1"
    e2 = bench.retry_env(dict(base, PF_BENCH_RETRY_PORT="40123"))            # the launcher's own ranks: a port it reserved
    assert e2["MASTER_PORT"] == "40123" and "PF_BENCH_RETRY_PORT" not in e2
    # the child really is a store HOST on rank 0: with the agent's variable gone torch's env:// rendezvous starts a TCPStore server
    code = ("import os, torch.distributed as d; d.init_process_group('gloo', rank=0, world_size=1); "
            "print('ok', d.get_world_size()); d.destroy_process_group()")
    env = dict(bench.retry_env(dict(base, RANK="0", WORLD_SIZE="1", MASTER_PORT=str(bench._free_port() - 1))), PYTHONPATH=os.environ.get("PYTHONPATH", ""))
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0 and "ok 1" in pr.stdout, pr.stderr[-500:]
