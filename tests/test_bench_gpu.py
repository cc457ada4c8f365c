"""bench.py end to end on the GPU box: the one-GPU line and the N > 1 control flow (two ranks sharing the one GPU of the box:
gloo rendezvous, shard plan, barrier-bracketed timing, the gather) - what the driver runs at round end."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--batch", "4096",
                          "--no-long-runs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64" and d["value"] > 1e6
    assert d["config"]["state_rmse_vs_oracle"] == 0.0 and d["config"]["instances_flagged"] == 0
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 4.0


def test_two_ranks_strong_scaling_on_one_gpu():
    env = dict(os.environ, BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--batch", "8192", "--no-long-runs", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert d["config"]["state_rmse_vs_oracle"] == 0.0 and d["config"]["instances_flagged"] == 0
