import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import vbo
from voicebridge_amd import api, synth
from tests.util import load, parse_opts
from tests.test_gpu_decode import arcs_of
ctx = api.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "decode_g2_beam10"
g = load(name); o = parse_opts(g["opts"]); arcs = arcs_of(g)
T = g["loglikes"].shape[0]
ll = torch.from_numpy(g["loglikes"]).to(ctx.device)
graph = api.Graph(ctx, g["fst_final"], g["fst_off"], arcs, 0)
for pi in (7, 25, 3, 1000):
    for Tn in (T, 20, 8, 7, 6):
        kw = dict(beam=o["beam"], max_active=int(o["max_active"]), min_active=int(o["min_active"]), lattice_beam=o["lattice_beam"], prune_interval=pi)
        ref = vbo.decode_lattice_faster(vbo.make_fst(g["fst_final"], g["fst_off"], arcs, 0), g["loglikes"][:Tn], g["tid2pdf"], acwt=o["acwt"], exact_cutoff=3, adm_bucket_bits=6, **kw)
        dec = api.LatticeDecoder(ctx, graph, 1, T, acoustic_scale=o["acwt"], **kw)
        ctx.torch_sync()
        dec.decode(ll[:Tn].contiguous(), [0, Tn], g["tid2pdf"])
        got = dec.result(0)
        sa = set(zip(got["tok_frame"].tolist(), got["tok_state"].tolist())); sb = set(zip(ref["tok_frame"].tolist(), ref["tok_state"].tolist()))
        print("pi", pi, "T", Tn, "gpu", len(sa), "ref", len(sb), "only gpu", sorted(sa - sb)[:6], "only ref", sorted(sb - sa)[:6], "links", got["num_links"], len(ref["link_src"]))
        dec.close()
print("---- detail")
from tests.test_oracle_decode import link_table
def lt(r):
    return link_table(r["link_src"], r["link_dst"], r["link_ilabel"], r["link_olabel"], r["link_graph"], r["link_acoustic"], r["tok_frame"], r["tok_state"])
for Tn in (8, 20):
    kw = dict(beam=o["beam"], max_active=int(o["max_active"]), min_active=int(o["min_active"]), lattice_beam=o["lattice_beam"], prune_interval=1000)
    ref = vbo.decode_lattice_faster(vbo.make_fst(g["fst_final"], g["fst_off"], arcs, 0), g["loglikes"][:Tn], g["tid2pdf"], acwt=o["acwt"], exact_cutoff=3, adm_bucket_bits=6, **kw)
    dec = api.LatticeDecoder(ctx, graph, 1, T, acoustic_scale=o["acwt"], **kw)
    ctx.torch_sync()
    dec.decode(ll[:Tn].contiguous(), [0, Tn], g["tid2pdf"])
    got = dec.result(0)
    A, B = lt(got), lt(ref)
    def show(rows):
        return [(r[0], r[1], r[2], r[3], r[4], r[5], float(np.frombuffer(r[6], np.float32)[0]), float(np.frombuffer(r[7], np.float32)[0])) for r in rows]
    print("T", Tn, "only gpu", show(sorted(set(A) - set(B))), "only ref", show(sorted(set(B) - set(A))))
    for r, nm in ((got, "gpu"), (ref, "ref")):
        m = r["tok_frame"] <= 1
        print(nm, list(zip(r["tok_frame"][m].tolist(), r["tok_state"][m].tolist(), r["tok_cost"][m].tolist(), r["tok_extra"][m].tolist())))
a0 = g["fst_off"]
st = 0
for _ in range(6):
    print("state", st, [(int(a["ilabel"]), int(a["nextstate"]), float(a["weight"])) for a in arcs[a0[st]:a0[st+1]]])
    nxt = [int(a["nextstate"]) for a in arcs[a0[st]:a0[st+1]] if a["ilabel"] == 0]
    if not nxt: break
    st = nxt[0]
