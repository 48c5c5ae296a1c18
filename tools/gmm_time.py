"""Time the all-pdf GMM kernel alone (the bench workload's model and frame count); prints ms and TFLOP/s.  Run under gpurun."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
def note(*a): print(*a, file=sys.stderr, flush=True)
from voicebridge_amd import api, synth
T = int(os.environ.get("T", 3072000)); P, G, D = 2500, int(os.environ.get("G", 15000)), 39
ctx = api.Context(0); note("ctx")
model = synth.gmm_model(P, G, D, seed=1)
gmm = api.DiagGmm(ctx, *model); note("gmm")
feats = torch.randn((T, D), device="cuda:0")
LD = int(os.environ.get("LD", P))
ll = torch.empty((T, LD), device="cuda:0"); torch.cuda.synchronize(); note("buffers")
best = 1e9
for i in range(5):
    torch.cuda.synchronize(); ctx.sync()
    ctx.timer_start(0); gmm.loglikes(feats, out=ll); ctx.timer_stop(0); ctx.sync()
    best = min(best, ctx.timer_ms(0)); note("iter", i, best)
fl = 2.0 * T * G * (2 * D + 2)
print(json.dumps({"ms": best, "TFLOPs": fl / best / 1e9, "ld": LD, "checksum": float(ll[::1000, :P].double().sum())}))
