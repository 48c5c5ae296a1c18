import sys, numpy as np, torch
sys.path.insert(0, '.')
from voicebridge_amd import api, synth
ctx = api.Context(0)
P, G, D = 2500, 15000, 39
offs, w, means, iv = synth.gmm_model(P, G, D, seed=1, mean_scale=1.5, var_spread=0.2)
gmm = api.DiagGmm(ctx, offs, w, (means * iv).astype(np.float32), iv)
for T in (4096, 65536, 1 << 20, 3069952):
    f = torch.randn((T, D), device=ctx.device)
    out = torch.full((T + 8, P), 7.0, device=ctx.device)
    ctx.torch_sync()
    gmm.loglikes(f, out=out[:T])
    ctx.sync()
    tail = out[T:].cpu().numpy()
    print("T", T, "ok; guard rows untouched:", bool((tail == 7.0).all()), "finite:", bool(torch.isfinite(out[:T]).all().item()), flush=True)
