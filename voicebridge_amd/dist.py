"""Multi-GPU layer of the hot path: speaker-sharded data parallelism with ONE gather at the end.

The reference parallelises by splitting the data directory per speaker into nj jobs that never
communicate (VB/scr/utils/split_data.cpp:20-100, VB/scr/steps/decode_gmm.cpp:300-329) and merges
the per-job files afterwards.  Here a rank (one process per GPU) owns a set of speakers -- CMVN
statistics and fMLLR transforms are per speaker, so nothing crosses GPUs during decoding -- and
the merge is a single collective over RCCL/xGMI (backend "nccl") at the end of the run:
  * all_reduce(SUM) of the 7 int64 scoring counters
    {word_errs, num_words, ins, del, sub, num_sent, sent_errs}  (VB/src/bin/compute-wer.cpp:112-133)
  * all_gather of per-rank blob sizes, then all_gather of the padded int32 blobs holding
    (utt index, n_words, words...) records; rank 0 unpacks them.
The same code runs on CPU tensors with the "gloo" backend (tests/test_dist.py).
"""
import numpy as np
import torch
import torch.distributed as dist

COUNTER_NAMES = ("word_errs", "num_words", "ins", "del", "sub", "num_sent", "sent_errs")


def shard_speakers(spk_frames, world):
    """Greedy longest-processing-time assignment of speakers to ranks by total frames.
    spk_frames: sequence of frame counts per speaker.  Returns int array rank_of_spk."""
    spk_frames = np.asarray(spk_frames, np.int64)
    order = np.argsort(-spk_frames, kind="stable")
    load = np.zeros(world, np.int64)
    rank_of = np.zeros(len(spk_frames), np.int32)
    for s in order:
        r = int(np.argmin(load))
        rank_of[s] = r
        load[r] += spk_frames[s]
    return rank_of


def edit_counts(ref, hyp):
    """(ins, del, sub) of the minimum edit distance alignment of hyp against ref
    (LevenshteinEditDistance as used by compute-wer, VB/src/bin/compute-wer.cpp:95-110)."""
    ref, hyp = list(ref), list(hyp)
    R, H = len(ref), len(hyp)
    # cost, ins, del, sub per cell; row by row
    prev = [(j, j, 0, 0) for j in range(H + 1)]
    for i in range(1, R + 1):
        cur = [(i, 0, i, 0)]
        for j in range(1, H + 1):
            c_sub = prev[j - 1]
            if ref[i - 1] == hyp[j - 1]:
                best = c_sub
            else:
                best = (c_sub[0] + 1, c_sub[1], c_sub[2], c_sub[3] + 1)
            c_del = prev[j]
            cand = (c_del[0] + 1, c_del[1], c_del[2] + 1, c_del[3])
            if cand[0] < best[0]:
                best = cand
            c_ins = cur[j - 1]
            cand = (c_ins[0] + 1, c_ins[1] + 1, c_ins[2], c_ins[3])
            if cand[0] < best[0]:
                best = cand
            cur.append(best)
        prev = cur
    return prev[H][1], prev[H][2], prev[H][3]


def score(refs, hyps):
    """counters (np.int64[7]) for parallel lists of reference / hypothesis word-id sequences"""
    c = np.zeros(7, np.int64)
    for r, h in zip(refs, hyps):
        i, d, s = edit_counts(r, h)
        c[0] += i + d + s
        c[1] += len(r)
        c[2] += i
        c[3] += d
        c[4] += s
        c[5] += 1
        c[6] += int(i + d + s > 0)
    return c


def pack_words(utt_ids, words):
    """[(utt, words)] -> int32 blob of (utt, n, w...) records"""
    parts = []
    for u, w in zip(utt_ids, words):
        w = np.asarray(w, np.int32)
        parts.append(np.concatenate([np.array([u, len(w)], np.int32), w]))
    return np.concatenate(parts) if parts else np.zeros(0, np.int32)


def unpack_words(blob):
    out, i = {}, 0
    while i < len(blob):
        u, n = int(blob[i]), int(blob[i + 1])
        out[u] = blob[i + 2:i + 2 + n].copy()
        i += 2 + n
    return out


def gather_results(utt_ids, words, counters, device=None, group=None):
    """The end-of-run collective.  Every rank passes its own utterances' word sequences and its
    scoring counters; returns (summed counters on every rank, {utt: words} on rank 0 else None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    c = torch.as_tensor(np.asarray(counters, np.int64)).to(dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    blob = pack_words(utt_ids, words)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=torch.int32, device=dev)
    if len(blob):
        padded[:len(blob)] = torch.from_numpy(blob).to(dev)
    bufs = [torch.zeros(mx, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    merged = None
    if rank == 0:
        merged = {}
        for r in range(world):
            merged.update(unpack_words(bufs[r][:sizes[r]].cpu().numpy()))
    return c.cpu().numpy(), merged
