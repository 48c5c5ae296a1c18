"""RCCL itself (backend "nccl") under the collectives bench.py issues at N > 1: one rank per visible GPU (a one-GPU box gives a world
of one: the library, its IPC mode and the tensor types are still the real ones; the multi-rank control flow is covered on gloo,
tests/test_bench_launch.py and tests/test_dist.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_collectives_of_the_bench():
    import torch
    n = max(1, min(torch.cuda.device_count(), 2))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    import socket
    with socket.socket() as s_:       # a free port, as bench.py's launch_ranks picks one
        s_.bind(("127.0.0.1", 0))
        port = str(s_.getsockname()[1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "tools", "rccl_smoke.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["backend"] == "nccl" and out["world_size"] == n
