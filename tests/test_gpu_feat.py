"""GPU parity: MFCC and the fused feature post-processing vs the oracle and the golden vectors.
Everything goes through the C ABI (voicebridge_amd.api -> libvoicebridge_hip.so)."""
import numpy as np
import pytest

from tests.util import load, mfcc_kwargs, parse_opts

pytestmark = pytest.mark.gpu

# The contract: 1e-3 absolute on cepstra of magnitude 10..100 (measured on speech-like input: 2.7e-4,
# tests/test_gpu_e2e.py).  The noiseless chirp is the exception that no fixed tolerance fits: most mel bins of a pure tone hold only
# the FFT's rounding residue, and log() turns two FFT algorithms' different residues into absolute cepstral differences of ~0.1.
# tests/golden/mfcc_chirp_exact.npz (oracle/gen_golden_chirp_exact.py) holds the EXACT cepstra (the oracle's formula in double) and
# how far the REFERENCE's own float32 output is from them: 0.087 -- its noise floor on this input (on a noisy input the same
# comparison gives 7.6e-5).  The chirp is therefore checked against the exact values, with the reference's own distance as tolerance.
MFCC_ATOL = 1e-3
CHIRP_VS_REFERENCE_ATOL = 0.2          # two float32 FFTs on a noiseless tone: ~2x the reference's own distance from the exact value
MFCC_CASES = ["mfcc_sine_default", "mfcc_chirp_c0", "mfcc_dc_nosnip", "mfcc_sine_nosnip_c0",
              "mfcc_speechy_htk", "mfcc_speechy_vtln", "mfcc_short_hann", "mfcc_tooshort"]


@pytest.fixture(scope="module")
def ctx():
    from voicebridge_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", MFCC_CASES)
def test_mfcc_golden(ctx, name, generic, monkeypatch):
    """both kernels: the wave-per-frame radix-4 one for 512-point windows and the general radix-2 one"""
    from voicebridge_amd import api
    if generic:
        monkeypatch.setenv("VBH_MFCC_GENERIC", "1")
    g = load(name)
    kw, vtln = mfcc_kwargs(parse_opts(g["opts"]))
    m = api.Mfcc(ctx, vtln_warp=vtln, **kw)
    assert m.num_frames(len(g["wave"])) == int(g["numframes"][0])     # integer: exact
    feats, foff = m.compute([g["wave"].astype(np.float32)])
    assert foff.tolist() == [0, int(g["numframes"][0])]
    if g["feats"].shape[0]:
        if name == "mfcc_chirp_c0":
            ex = load("mfcc_chirp_exact")
            err = float(np.abs(feats.cpu().numpy().astype(np.float64) - ex["exact"]).max())
            print("\nchirp (%s kernel): max |GPU - exact| = %.4f; the reference's own: %.4f" % ("generic" if generic else "512-point", err, float(ex["reference_max_abs_err"])))
            assert err <= float(ex["reference_max_abs_err"])        # at least as close to the exact cepstra as the reference is
            np.testing.assert_allclose(feats.cpu().numpy(), g["feats"], atol=CHIRP_VS_REFERENCE_ATOL, rtol=1e-5)
        else:
            np.testing.assert_allclose(feats.cpu().numpy(), g["feats"], atol=MFCC_ATOL, rtol=1e-5)
    m.close()


def test_mfcc_batch_int16_vs_oracle(ctx):
    """ragged batch incl. an utterance too short for one frame, int16 input, default options"""
    from oracle import vbo
    from voicebridge_amd import api, synth
    waves = [synth.wave(0, 1.3), synth.wave(1, 0.01), synth.wave(2, 2.0), synth.chirp(0.7), synth.wave(3, 0.025)]
    m = api.Mfcc(ctx)
    feats, foff = m.compute(waves)
    feats = feats.cpu().numpy()
    o = vbo.mfcc_opts()
    exp_off = [0]
    for u, w in enumerate(waves):
        ref = vbo.mfcc(w.astype(np.float32), o)
        exp_off.append(exp_off[-1] + len(ref))
        got = feats[foff[u]:foff[u + 1]]
        assert got.shape == ref.shape
        if len(ref):
            np.testing.assert_allclose(got, ref, atol=CHIRP_VS_REFERENCE_ATOL if u == 3 else MFCC_ATOL, rtol=1e-5)
    assert foff.tolist() == exp_off
    m.close()


def test_mfcc_empty_batch(ctx):
    from voicebridge_amd import api
    m = api.Mfcc(ctx)
    feats, foff = m.compute([])
    assert feats.shape[0] == 0 and foff.tolist() == [0]
    m.close()


def test_mfcc_odd_offsets_and_float_input_vs_oracle(ctx):
    """utterances of odd length put their successors at odd sample offsets (the fast kernel's paired 32-bit loads need the
    16-bit path there); float input takes the same code with other loads"""
    from oracle import vbo
    from voicebridge_amd import api, synth
    waves = [synth.wave(0, 0.5)[:-1], synth.wave(1, 0.4)[:-3], synth.wave(2, 0.3)]
    o = vbo.mfcc_opts()
    for cast in (np.int16, np.float32):
        m = api.Mfcc(ctx)
        feats, foff = m.compute([w.astype(cast) for w in waves])
        feats = feats.cpu().numpy()
        for u, w in enumerate(waves):
            np.testing.assert_allclose(feats[foff[u]:foff[u + 1]], vbo.mfcc(w.astype(np.float32), o), atol=MFCC_ATOL, rtol=1e-5)
        m.close()


@pytest.mark.parametrize("generic", [False, True])
def test_mfcc_dither_is_standard_normal_noise(ctx, generic, monkeypatch):
    """Dither (K/feat/feature-window.cc:90-98): dither * N(0,1) on every window sample.  The reference's libc rand() stream
    cannot be replayed, so the check is statistical: on a silent wave the raw log-energy of a frame is log of a
    chi-square(399)-distributed sum (window of 400 samples, mean removed), frames and utterances get different noise, a seed
    reproduces it, and the stream does not depend on how a job is cut into calls."""
    from voicebridge_amd import api
    if generic:
        monkeypatch.setenv("VBH_MFCC_GENERIC", "1")
    silent = [np.zeros(16000 * 4, np.int16), np.zeros(16000 * 4, np.int16)]
    m = api.Mfcc(ctx, dither=1.0)
    a, foff = m.compute(silent)
    a = a.cpu().numpy()
    e = np.exp(a[:, 0].astype(np.float64))                 # column 0 = raw log-energy (use_energy default)
    n = len(e)
    assert n == 2 * 398
    # chi-square with 399 degrees of freedom: mean 399, variance 798
    assert abs(e.mean() - 399.0) < 5 * np.sqrt(798.0 / n)
    assert 0.8 * 798.0 < e.var() < 1.25 * 798.0
    assert not np.allclose(a[foff[0]:foff[1]], a[foff[1]:foff[2]])      # utterances differ
    assert np.abs(np.diff(a[:, 0])).min() > 0                           # frames differ
    b, _ = m.compute(silent)
    np.testing.assert_array_equal(a, b.cpu().numpy())                   # reproducible
    # the second utterance alone, announced as utterance 1 of the job: the same features as inside the batch
    m.set_dither_stream(first_utt_index=1)
    c, _ = m.compute(silent[1:])
    np.testing.assert_array_equal(c.cpu().numpy(), a[foff[1]:foff[2]])
    m.set_dither_stream(seed=1234)
    d, _ = m.compute(silent)
    assert not np.allclose(d.cpu().numpy(), a)
    # scaling: dither = 3 -> variance 9 times larger
    m3 = api.Mfcc(ctx, dither=3.0)
    e3 = np.exp(m3.compute(silent)[0].cpu().numpy()[:, 0].astype(np.float64))
    assert abs(e3.mean() / 9.0 - 399.0) < 5 * np.sqrt(798.0 / n)
    # a loud signal is hardly changed by dither 1.0 (that is its purpose: only digital silence is affected)
    from voicebridge_amd import synth
    w = [synth.wave(5, 1.0)]
    x0 = api.Mfcc(ctx).compute(w)[0].cpu().numpy()
    x1 = api.Mfcc(ctx, dither=1.0).compute(w)[0].cpu().numpy()
    assert 0 < np.abs(x0 - x1).max() < 0.5


@pytest.mark.parametrize("name", ["post_13", "post_9_lda"])
def test_featpost_golden(ctx, name):
    import torch
    from voicebridge_amd import api
    g = load(name)
    o = parse_opts(g["opts"]) if "opts" in g.files else {}
    feats = torch.from_numpy(g["feats"]).to(ctx.device)
    foff = np.array([0, len(g["feats"])], np.int64)
    dw = int(o.get("delta_window", 2))
    # deltas only
    p = api.FeaturePipeline(ctx, cmvn=False, mode="delta", delta_window=dw)
    np.testing.assert_allclose(p.apply(feats, foff).cpu().numpy(), g["ref_deltas"], rtol=1e-5, atol=1e-5)
    # splice only: pure copy, bit-exact
    p = api.FeaturePipeline(ctx, cmvn=False, mode="splice", splice_left=int(o.get("splice_left", 4)), splice_right=int(o.get("splice_right", 4)))
    np.testing.assert_array_equal(p.apply(feats, foff).cpu().numpy(), g["ref_splice"])
    # cmvn (mean, and mean+variance)
    st = g["cmvn_stats"][None]
    p = api.FeaturePipeline(ctx, cmvn=True, mode="none")
    np.testing.assert_allclose(p.apply(feats, foff, utt2spk=[0], cmvn_stats=st).cpu().numpy(), g["ref_cmvn"], rtol=0, atol=1e-6)
    p = api.FeaturePipeline(ctx, cmvn=True, norm_vars=True, mode="none")
    np.testing.assert_allclose(p.apply(feats, foff, utt2spk=[0], cmvn_stats=st).cpu().numpy(), g["ref_cmvn_var"], rtol=1e-6, atol=1e-6)
    # deltas + transform (affine per speaker for post_13, linear global for post_9_lda)
    x = g["xform"]
    if x.shape[1] == g["ref_deltas"].shape[1] + 1:
        p = api.FeaturePipeline(ctx, cmvn=False, mode="delta", delta_window=dw, spk_xform_shape=x.shape)
        got = p.apply(feats, foff, utt2spk=[0], spk_xforms=x[None])
    else:
        p = api.FeaturePipeline(ctx, cmvn=False, mode="delta", delta_window=dw, lda=x)
        got = p.apply(feats, foff)
    np.testing.assert_allclose(got.cpu().numpy(), g["ref_xform"], rtol=1e-5, atol=2e-4)
    # cmvn statistics
    stats = api.FeaturePipeline(ctx).cmvn_stats(feats, foff, [0], 1)
    np.testing.assert_allclose(stats[0], g["ref_stats"], rtol=1e-9)


def test_featpost_full_chain_vs_oracle(ctx):
    """CMVN -> deltas -> per-speaker fMLLR on a ragged multi-speaker batch (DELTA+SAT path) and
    CMVN -> splice -> LDA -> fMLLR (LDA+MLLT+SAT path), incl. utterances shorter than the
    context and crossing the 64-frame chunking."""
    import torch
    from oracle import vbo
    from voicebridge_amd import api, synth
    rng = np.random.default_rng(0)
    lens = [1, 3, 64, 65, 130, 200]
    utt2spk = [0, 1, 0, 2, 1, 2]
    feats = [(rng.standard_normal((n, 13)) * np.linspace(6, 1, 13)).astype(np.float32) for n in lens]
    foff = np.zeros(len(lens) + 1, np.int64)
    foff[1:] = np.cumsum(lens)
    stats = np.zeros((3, 2, 14))
    for f, s in zip(feats, utt2spk):
        stats[s] += vbo.acc_cmvn_stats(f)
    d_in = torch.from_numpy(np.concatenate(feats)).to(ctx.device)
    # DELTA+SAT
    xf = np.stack([synth.affine(39, seed=s) for s in range(3)])
    p = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(39, 40))
    got = p.apply(d_in, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xf).cpu().numpy()
    for u, (f, s) in enumerate(zip(feats, utt2spk)):
        ref = vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(f, stats[s])), xf[s])
        np.testing.assert_allclose(got[foff[u]:foff[u + 1]], ref, rtol=1e-5, atol=2e-4)
    # LDA+MLLT+SAT
    lda = (rng.standard_normal((40, 117)) * 0.1).astype(np.float32)
    xf2 = np.stack([synth.affine(40, seed=10 + s) for s in range(3)])
    p = api.FeaturePipeline(ctx, cmvn=True, norm_vars=True, mode="splice", lda=lda, spk_xform_shape=(40, 41))
    got = p.apply(d_in, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xf2).cpu().numpy()
    for u, (f, s) in enumerate(zip(feats, utt2spk)):
        ref = vbo.transform_feats(vbo.transform_feats(vbo.splice(vbo.apply_cmvn(f, stats[s], True)), lda), xf2[s])
        np.testing.assert_allclose(got[foff[u]:foff[u + 1]], ref, rtol=1e-5, atol=5e-4)
    # statistics of the whole batch per speaker
    st = api.FeaturePipeline(ctx).cmvn_stats(d_in, foff, utt2spk, 3)
    np.testing.assert_allclose(st, stats, rtol=1e-9, atol=1e-9)
