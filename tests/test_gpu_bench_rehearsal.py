"""`bench.py --gpus 2` rehearsed on ONE GPU (both ranks share it, host-staged gloo transport): the N > 1 line must carry the
headline with its exchange block AND BASELINE.json's multi-GPU configs -- configs[4] (MAACO on G1024, 8 192 ants per GPU) and
configs[3] (PSO and GA, 2 048 agents per GPU) -- as `extra` legs, each with `n_gpus`, a value and its own `config.exchange`
(transport, bytes, calls, time per step).  Scaling itself can only be measured by the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_line_carries_the_multi_gpu_configs():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PF_BENCH_SHARE_GPU="1", PF_BENCH_EXTRA_BUDGET="400")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                         "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    ex = d["config"]["exchange"]
    assert ex["transport"] in ("gloo", "torch", "rccl") and ex["calls_per_step"] > 0 and ex["bytes_per_rank_per_step"] > 0
    assert ex["exchange_ms_per_step"] > 0
    assert "_error" not in d["extra"], d["extra"]
    for name, per_gpu in (("maaco1024", 8192), ("pso512", 2048), ("ga512", 2048)):
        leg = d["extra"][name]
        assert "error" not in leg, leg
        assert leg["n_gpus"] == 2 and leg["value"] > 0 and leg["config"]["agents_per_gpu"] == per_gpu
        e2 = leg["config"]["exchange"]
        assert e2["calls_per_step"] > 0 and e2["bytes_per_rank_per_step"] > 0 and e2["exchange_ms_per_step"] > 0, (name, e2)
