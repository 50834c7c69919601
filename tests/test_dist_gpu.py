"""bench.py under torch.distributed.run with ONE rank on the GPU box: exercises the RCCL (nccl backend)
initialisation and the all-gather code path the 8-GPU driver run uses (world_size 1 is the only size a
1-GPU box can host; the world-2 logic is covered on CPU with gloo in test_host_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_bench_under_torchrun_one_rank():
    env = dict(os.environ, YL_BENCH_FORCE_COLLECTIVE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "roofline" in d
