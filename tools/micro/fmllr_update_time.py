import numpy as np, torch, sys
sys.path.insert(0, '.')
from voicebridge_amd import api, synth
ctx = api.Context(0)
D, n_spk, P = 39, 64, 40
rng = np.random.default_rng(1)
offs, w, means, iv = synth.gmm_model(P, 200, D, seed=D + 1)
gmm = api.DiagGmm(ctx, offs, w, (means * iv).astype(np.float32), iv)
lens = rng.integers(400, 900, n_spk)
foff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64); T = int(foff[-1])
feats = (rng.standard_normal((T, D)) * 1.1 + 0.3).astype(np.float32)
u2s = np.arange(n_spk, dtype=np.int32)
poff = np.arange(T + 1, dtype=np.int64); ppdf = rng.integers(0, P, T).astype(np.int32); pw = np.ones(T, np.float32)
gmm.fmllr_stats(torch.from_numpy(feats).to(ctx.device), foff, u2s, n_spk, poff, ppdf, pw)
for it in (0, 1, 2, 10, 40, 40):
    x, impr, ms = gmm.fmllr_update_full(n_spk, min_count=100.0, num_iters=it)
    print("iters %d: %.2f ms" % (it, ms))
