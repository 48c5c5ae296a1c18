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
    """(ins, del, sub) of the minimum edit distance alignment of hyp against ref, with the tie order of the
    reference's LevenshteinEditDistance (K/util/edit-distance-inl.h:78-122, as compute-wer calls it,
    VB/src/bin/compute-wer.cpp:95-110): a substitution/match only when it is strictly cheaper than both other
    moves, else a deletion when it is strictly cheaper than an insertion, else an insertion.  The total is the
    same under any order; the split into ins/del/sub is not, and this is the one the host ComputeWer prints."""
    ref, hyp = list(ref), list(hyp)
    R = len(ref)
    # column r of a row: (cost, ins, del, sub) of the best alignment of the hypothesis prefix with ref[:r]
    row = [(r, 0, r, 0) for r in range(R + 1)]
    for h in hyp:
        c0 = row[0]
        cur = [(c0[0] + 1, c0[1] + 1, c0[2], c0[3])]
        for r in range(1, R + 1):
            up, left, diag = row[r], cur[r - 1], row[r - 1]
            miss = int(h != ref[r - 1])
            c_ins, c_del, c_sub = up[0] + 1, left[0] + 1, diag[0] + miss
            if c_sub < c_ins and c_sub < c_del:
                cur.append((c_sub, diag[1], diag[2], diag[3] + miss))
            elif c_del < c_ins:
                cur.append((c_del, left[1], left[2] + 1, left[3]))
            else:
                cur.append((c_ins, up[1] + 1, up[2], up[3]))
        row = cur
    return row[R][1], row[R][2], row[R][3]


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
