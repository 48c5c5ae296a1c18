"""bench.py --gpus N: the advertised command must itself start N ranks (`--gpus` was parsed and never used)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vb_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_flag_starts_one_rank_per_gpu(monkeypatch):
    """without a launcher around it, `python bench.py --gpus 4 ...` starts torch.distributed.run with 4 processes as a child
    (never an exec of a process that may have touched the GPU) and passes its own arguments on"""
    b = _load_bench()
    seen = {}

    def fake_call(cmd, **kw):
        seen["cmd"] = cmd
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 7                                  # the children's exit code is the bench's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_world_size_must_match_gpus(monkeypatch):
    b = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        b.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


@pytest.mark.gpu
def test_plain_bench_gpus_2_runs_two_ranks():
    """the whole N > 1 path from the plain command: two ranks (sharing the box's one GPU, collectives on gloo because RCCL refuses two
    ranks on one device), speaker-sharded job, max-over-ranks time, end gather of every utterance's words to rank 0"""
    env = dict(os.environ, VBH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0",
                        "--words", "20000", "--no-cpu-baseline", "--no-determinize"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["gather"]["utterances"] == 128                # 2 x 64 distinct utterances of the job reached rank 0
    assert out["gather"]["counters"]["num_sent"] == 128
    assert out["value"] > 0
