"""does gmm_rows_kernel compute the right scores while search workgroups share its CUs?  (no search consumes its output here)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from voicebridge_amd import api
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
args.feat_dim = 39
ctx = api.Context(0)
dev = ctx.device
d_wave, woff, utt2spk, n_spk, xforms, gid = bench.build_workload(args, 0, 1, dev)
mf = api.Mfcc(ctx)
post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(39, 40))
ctx.torch_sync()
raw, foff = mf.compute_packed(d_wave, woff, True)
ctx.sync()
stats = post.cmvn_stats(raw, foff, utt2spk, n_spk)
feats = post.apply(raw, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xforms)
ctx.sync()
T = int(foff[-1])
idx = torch.linspace(0, T - 1, min(T, 20000), device=dev).long()
model = bench.feature_space_model(args, feats[idx].cpu().numpy())
os.environ["VBH_GMM_V1"] = "1"
gmm1 = api.DiagGmm(ctx, *model)
del os.environ["VBH_GMM_V1"]
gmm2 = api.DiagGmm(ctx, *model)
ll1 = gmm1.loglikes(feats)
ll2 = torch.empty_like(ll1)
ctx.sync()
g, t2p = bench.build_graph(args)
graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
Tmax = int(np.max(np.diff(foff)))
sctx = api.Context(0, cus=224)
dec = api.LatticeDecoder(sctx, graph, args.batch, Tmax, beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0, acoustic_scale=0.083333)

def compare(tag):
    worst, bad = 0.0, 0
    for a in range(0, T, 1 << 18):
        b = min(T, a + (1 << 18))
        d = (ll2[a:b] - ll1[a:b]).abs()
        bad += int((~torch.isfinite(ll2[a:b])).sum().item())
        d = torch.nan_to_num(d, nan=1e30, posinf=1e30)
        worst = max(worst, float(d.max().item()))
    print("%s: max |v2 - v1| = %.3g, non-finite in v2: %d" % (tag, worst, bad), flush=True)

for nm, t_ in (("feats", feats), ("ll1", ll1), ("ll2", ll2), ("raw", raw), ("d_wave", d_wave)):
    print("%s: [%#x, %#x)" % (nm, t_.data_ptr(), t_.data_ptr() + t_.numel() * t_.element_size()), flush=True)
gmm2.loglikes(feats, out=ll2); ctx.sync()
compare("v2 alone")
which = os.environ.get("BESIDE", "v2")
gmmx = gmm2 if which == "v2" else gmm1
for rep in range(2):
    ll2.fill_(123.0); ctx.torch_sync()
    dec.decode_begin(ll1, foff, t2p)          # the search reads the v1 scores
    t0 = time.time()
    n = 0
    while time.time() - t0 < 1.2:             # v2 runs beside the search grid the whole time
        gmmx.loglikes(feats, out=ll2); ctx.sync(); n += 1
    dec.decode_end()
    compare("%s beside a search grid (rep %d, %d calls)" % (which, rep, n))
print("done", flush=True)
