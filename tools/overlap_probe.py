"""Experiment: do two search kernels on two streams fill each other's tails?
Decodes the same batch with two decoders, first one after the other, then from two threads."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voicebridge_amd import api

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
ctxs = [api.Context(0), api.Context(0)]
ctx = ctxs[0]
waves, utt2spk, n_spk, xforms = bench.build_workload(args, 0)
mf = api.Mfcc(ctx)
post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(39, 40))
d_wave, woff, is_i16 = mf.pack(waves)
ctx.torch_sync()
raw, foff = mf.compute_packed(d_wave, woff, is_i16)
ctx.sync()
stats = post.cmvn_stats(raw, foff, utt2spk, n_spk)
feats = post.apply(raw, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xforms)
ctx.sync()
T = int(foff[-1])
model = bench.feature_space_model(args, feats[: min(T, 20000)].cpu().numpy())
gmm = api.DiagGmm(ctx, *model)
ll = torch.empty((T, gmm.num_pdfs), dtype=torch.float32, device=ctx.device)
ctx.torch_sync()
gmm.loglikes(feats, out=ll)
ctx.sync()
g, t2p = bench.build_graph(args)
graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
Tmax = int(np.max(np.diff(foff)))
decs = [api.LatticeDecoder(c, graph, args.batch, Tmax, beam=args.beam, max_active=args.max_active, min_active=200,
                           lattice_beam=args.lattice_beam, acoustic_scale=0.083333) for c in ctxs]
for c in ctxs:
    c.sync()
for d in decs:
    d.decode(ll, foff, t2p)          # warm (pool allocation)
t0 = time.perf_counter()
for d in decs:
    d.decode(ll, foff, t2p)
t_seq = time.perf_counter() - t0
print("two batches one after the other: %.3f s" % t_seq)
for lag in (0.0, 0.3, 0.6):
    def run(d, wait):
        time.sleep(wait)
        d.decode(ll, foff, t2p)
    th = [threading.Thread(target=run, args=(decs[i], lag * i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("two batches from two threads, second starts %.1f s later: %.3f s" % (lag, time.perf_counter() - t0))
