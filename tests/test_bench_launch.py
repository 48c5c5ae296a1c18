"""bench.py --gpus N: the advertised command must itself start N ranks (`--gpus` was parsed and never used)."""
import json
import os
import subprocess
import sys

import numpy as np
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


def test_strong_scaling_shares_of_config5():
    """--scaling strong: the 10 000-utterance corpus (BASELINE config 5) as equal batches per rank, a few speakers rounded up"""
    from voicebridge_amd import dist as vdist
    for world, want in ((1, (10, 1000)), (2, (5, 1000)), (4, (3, 836)), (8, (2, 628))):
        n_spk = 10000 // 4
        share = int(np.bincount(vdist.shard_speakers(np.full(n_spk, 4, np.int64), world), minlength=world).max())
        n_sets = -(-share * 4 // 1024)
        batch = -(-share // n_sets) * 4
        assert (n_sets, batch) == want
        assert 10000 <= n_sets * batch * world <= 10000 * 1.005


@pytest.mark.gpu
def test_strong_scaling_two_ranks_and_a_thread_budget():
    """bench.py --gpus 2 --scaling strong: one fixed corpus over two ranks (gloo rehearsal on the one GPU), two batches per rank and
    step, the lattice stage on two host threads per rank (--host-threads-per-rank)"""
    env = dict(os.environ, VBH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--utterances", "200", "--batch", "64",
                        "--steps", "1", "--warmup", "0", "--det-steps", "1", "--words", "20000", "--no-cpu-baseline", "--no-step-api",
                        "--host-threads-per-rank", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    c = out["config"]
    assert c["workload"].startswith("config5")
    # 50 speakers -> 25 per rank -> 2 batches of 13 speakers = 52 utterances each
    assert (c["batches_per_step_per_gpu"], c["utterances_per_batch"], c["utterances_per_step_job"]) == (2, 52, 208)
    assert c["host_threads_per_rank"] == 2 and out["lattice_stage_in_step"]["threads"] == 2
    assert out["value"] > 0 and "keeps_up" in out["lattice_stage_in_step"]
