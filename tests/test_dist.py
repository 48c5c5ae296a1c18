"""The N > 1 path on CPU: world_size 2, gloo.  Speaker sharding + the single end-of-run gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicebridge_amd import dist as vdist


def test_edit_counts():
    assert vdist.edit_counts([1, 2, 3], [1, 2, 3]) == (0, 0, 0)
    assert vdist.edit_counts([1, 2, 3], [1, 3]) == (0, 1, 0)
    assert vdist.edit_counts([1, 3], [1, 2, 3]) == (1, 0, 0)
    assert vdist.edit_counts([1, 2, 3], [1, 9, 3]) == (0, 0, 1)
    assert vdist.edit_counts([], [4, 5]) == (2, 0, 0)
    assert vdist.edit_counts([4, 5], []) == (0, 2, 0)
    i, d, s = vdist.edit_counts([1, 2, 3, 4, 5], [2, 3, 9, 5, 6])
    assert i + d + s == 3


def test_edit_counts_split_is_the_reference_one(tmp_path):
    """the ins/del/sub split of dist.edit_counts = the host ComputeWer's (the reference's tie order,
    K/util/edit-distance-inl.h:104-120): what the multi-GPU gather sums must be what single-GPU scoring prints"""
    import ctypes as C
    from voicebridge_amd import build
    assert vdist.edit_counts([0, 2, 3, 0, 2], [2, 1, 3]) == (1, 3, 0)
    host = C.CDLL(build.HOST_LIB)
    rng = np.random.default_rng(5)
    refs = [rng.integers(1, 5, size=rng.integers(0, 8)).tolist() for _ in range(300)]
    hyps = [rng.integers(1, 5, size=rng.integers(0, 8)).tolist() for _ in range(300)]
    (tmp_path / "ref.txt").write_text("".join("u%03d %s\n" % (i, " ".join(map(str, r))) for i, r in enumerate(refs)))
    (tmp_path / "hyp.txt").write_text("".join("u%03d %s\n" % (i, " ".join(map(str, h))) for i, h in enumerate(hyps)))
    args = ["compute-wer", "--text", "--mode=present", "ark:" + str(tmp_path / "ref.txt"), "ark:" + str(tmp_path / "hyp.txt")]
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    assert host.vb_ComputeWer(len(args), argv, str(tmp_path / "log.txt").encode()) == 0
    c = (C.c_longlong * 8)()
    host.vb_ComputeWerCounts(c)
    assert list(c)[:7] == vdist.score(refs, hyps).tolist()


def test_shard_speakers_balances_frames():
    frames = [1000, 10, 900, 20, 500, 480, 30, 5]
    rank_of = vdist.shard_speakers(frames, 2)
    load = [sum(f for f, r in zip(frames, rank_of) if r == k) for k in range(2)]
    assert abs(load[0] - load[1]) <= 60
    assert sorted(set(rank_of.tolist())) == [0, 1]
    assert vdist.shard_speakers([5, 4, 3], 1).tolist() == [0, 0, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    n_utt = 11
    refs = [rng.integers(1, 50, size=rng.integers(0, 9)).tolist() for _ in range(n_utt)]
    hyps = [[w for w in r if rng.random() > 0.2] + ([7] if rng.random() > 0.7 else []) for r in refs]
    spk = np.arange(n_utt) % 4
    rank_of_spk = vdist.shard_speakers([sum(len(refs[u]) + 1 for u in range(n_utt) if spk[u] == s) for s in range(4)], world)
    mine = [u for u in range(n_utt) if rank_of_spk[spk[u]] == rank]
    counters = vdist.score([refs[u] for u in mine], [hyps[u] for u in mine])
    total, merged = vdist.gather_results(mine, [hyps[u] for u in mine], counters)
    expect = vdist.score(refs, hyps)
    assert total.tolist() == expect.tolist()
    if rank == 0:
        assert sorted(merged) == list(range(n_utt))
        for u in range(n_utt):
            assert merged[u].tolist() == hyps[u]
        open(os.path.join(out_dir, "ok"), "w").write("1")
    else:
        assert merged is None
    dist.destroy_process_group()


def test_gather_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
