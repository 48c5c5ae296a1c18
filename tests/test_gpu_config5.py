"""BASELINE config 5 at its stated size, N = 1 point (VERDICT r4 "next" 1): ONE corpus of 10 000 synthetic 30 s utterances (83.3 h of
audio, 29.98 M frames), DELTA+SAT, speaker-sharded, one pass over it = one step, through `bench.py --scaling strong` exactly as the
driver would start it (a child process; nothing of bench.py is imported).  The N = 2 / 4 / 8 points need a multi-GPU node and are the
driver's to run; tests/test_bench_launch.py rehearses the two-rank share arithmetic on CPU and on one GPU under gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_full_corpus_on_one_gpu():
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--scaling", "strong", "--utterances", "10000", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-step-api"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    j = json.loads(line)
    print("\nconfig 5, N = 1: %.3f M frames/s, %.2f s per pass of the corpus, WER %.2f %%; %s"
          % (j["value"] / 1e6, j["ms_per_step"] / 1e3, j["wer_vs_transcripts"]["wer_percent"], j["config"]["workload"][:160]))
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and j["steps"] == 1
    wl = j["config"]["workload"]
    assert wl.startswith("config5") and "10000 synthetic 30 s utterances" in wl and "10 batch(es) of 1000" in wl
    assert j["search_retries_last_batch"]["failed"] == 0
    # the corpus really was 10 000 utterances of 2998 frames: frames of one pass = value x seconds per pass
    frames = j["value"] * j["ms_per_step"] * 1e-3
    assert abs(frames - 10000 * 2998) < 0.001 * 10000 * 2998
    # WER against the transcripts within 0.3 abs of the 1024-utterance figure of the default bench (5.5 %, BENCH_r04)
    assert abs(j["wer_vs_transcripts"]["wer_percent"] - 5.5) <= 0.3
    assert j["value"] >= 3.2e6
