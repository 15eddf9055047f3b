"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(include/lyra_hip.h via lyra_amd.codec), against
  * the committed golden fixtures generated from the reference's own flatbuffers (tests/golden/),
  * the CPU oracle (oracle/lyra_oracle.c) on the same seeded inputs,
  * size-independent properties at the BASELINE.json batch sizes.

Bars: RVQ indices / packets / int8 feature codes bit-exact; PCM bit-exact against the oracle (the kernels
reproduce the oracle's canonical fp32 order exactly) and <= 1 int16 LSB against the
float64-accumulating graph execution that produced the fixtures; log-mel <= 4 ULP (the reference's own
FloatEq bar, log_mel_spectrogram_extractor_impl_test.cc:81-82).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    """The product's default arithmetic: XNNPACK's (what the reference runs)."""
    import lyra_amd
    c = lyra_amd.LyraHip(max_streams=8192)
    assert c.requant == "xnnpack"
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_builtin():
    """The two TFLite-builtin-kernel flavours."""
    import lyra_amd
    cs = {m: lyra_amd.LyraHip(max_streams=64, requant=m) for m in ("exact", "gemmlowp_double", "builtin_mixed")}
    yield cs
    for c in cs.values():
        c.close()


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def synth(B, T, seed=0x4C797261):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(-32768, 32768, size=(T, B, 320)).astype(np.int16)


# ------------------------------------------------------------------------------------------------
# golden fixtures
# ------------------------------------------------------------------------------------------------
def _speech_golden(c, g, suf):
    """Fixtures = graph execution of the reference's flatbuffers (tools/make_golden.py) with the fp32 layers as the
    bias-first fmaf chains real XNNPACK computes: features, indices, lossy features and PCM are all bit-equal."""
    c.reset()
    sid = np.array([5], np.int32)
    for f, hop in enumerate(g["pcm_in"]):
        feat = c.extract(hop[None], sid)
        assert np.array_equal(feat[0], g["feats_" + suf][f]), f"features differ at hop {f}"
        idx = c.rvq_encode(feat, 184)
        assert np.array_equal(idx[0], g["idx_" + suf][f]), f"indices differ at hop {f}"
        lossy = c.rvq_decode(idx)
        if "lossy_" + suf in g.files:
            assert np.array_equal(lossy[0], g["lossy_" + suf][f])
        pcm = c.generate(lossy, sid)
        assert np.array_equal(pcm[0], g["pcm_" + suf][f]), f"PCM differs at hop {f}"


def test_speech_golden_xnnpack(ctx, golden_dir):
    _speech_golden(ctx, _g(golden_dir, "speech_sample1.npz"), "xnnpack")


@pytest.mark.parametrize("mode,suf", [("exact", "exact"), ("gemmlowp_double", "double")])
def test_speech_golden_builtin_modes(ctx_builtin, golden_dir, mode, suf):
    _speech_golden(ctx_builtin[mode], _g(golden_dir, "speech_sample1.npz"), suf)


def test_speech_golden_builtin_mixed(ctx_builtin, golden_dir):
    """Mode "builtin_mixed" (round 6) against the flatbuffers executed in that mode (tools/make_golden.py --mixed)."""
    from test_oracle_golden import _mixed_as_suffixed
    _speech_golden(ctx_builtin["builtin_mixed"], _mixed_as_suffixed(golden_dir), "mixed")


def test_known_answer_packets_fused(ctx, ctx_builtin, golden_dir):
    """Known answers through the fused encode path; embedded bit-stream.  "exact": BASELINE.md section 4 / SURVEY.md A.6;
    "xnnpack": graph execution in round 4 (tests/test_oracle_golden.py holds the same strings for the oracle)."""
    g = _g(golden_dir, "speech_sample1.npz")
    for c, want, full in ((ctx, ["a008090c75b6ce39", "a6890bdc76bc46e8", "a0f9698548b79206"], None),
                          (ctx_builtin["exact"], ["a00809827516b2df", "a6890bde76bcb6e1", "a0f9692544bc120d"],
                           "a00809827516b2df55c14f95327f2cc981d6a9f0d2dbc7")):
        for bits, nbytes in ((64, 8), (120, 15), (184, 23)):
            c.reset()
            pk = [c.encode(g["pcm_in"][f][None], bits)[0] for f in range(3)]
            assert all(p.size == nbytes for p in pk)
            assert [bytes(p[:8]).hex() for p in pk] == want
            if bits == 184 and full:
                assert bytes(pk[0]).hex() == full


def test_noise_golden_batched(ctx, golden_dir):
    g = _g(golden_dir, "noise_4x6.npz")
    ctx.reset()
    ids = np.array([11, 2, 300, 7], np.int32)
    for t in range(6):
        pk = ctx.encode(g["pcm_in"][t], 184, ids)
        idx = np.stack([np.array([[p >> 4, p & 15] for p in row]).reshape(-1) for row in pk])
        assert np.array_equal(idx, g["idx_xnnpack"][t])
        pcm = ctx.decode(pk, 184, ids)
        assert np.array_equal(pcm, g["pcm_xnnpack"][t])


def test_rvq_fixture_and_random(ctx, golden_dir):
    g = _g(golden_dir, "rvq.npz")
    want = [8, 11, 14, 1, 6, 13, 12, 11, 1, 15, 6, 12, 4, 3, 3, 0, 8, 8, 11, 9, 9, 6, 12, 13, 14, 5, 9, 3, 13, 5,
            0, 14, 8, 3, 5, 0, 0, 11, 8, 1, 2, 11, 7, 9, 1, 7]
    for n in (16, 30, 46):
        idx = ctx.rvq_encode(g["fixture"][None], 4 * n)[0]
        assert list(idx[:n]) == want[:n] and np.all(idx[n:] == -1)
        dec = ctx.rvq_decode(idx[None])[0]
        assert np.array_equal(dec, g[f"fixture_dec{n}"])
        rel = np.sqrt(((g["fixture"] - dec) ** 2).sum() / (g["fixture"] ** 2).sum())
        assert rel < 1.11  # residual_vector_quantizer_test.cc:104-111
    idx = ctx.rvq_encode(g["rnd"], 184)
    assert np.array_equal(idx, g["rnd_idx"])
    assert np.array_equal(ctx.rvq_decode(idx), g["rnd_dec"])


# ------------------------------------------------------------------------------------------------
# oracle on the same seeded inputs, many streams, ragged batch sizes, scattered stream ids
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,bits", [(1, 64), (13, 120), (37, 184)])
def test_vs_oracle_bit_exact(ctx, oracle_default, B, bits):
    from oracle import lyra_oracle
    T = 8
    pcm = synth(B, T, seed=1234 + B)
    pcm[2:4] //= 64  # some quiet frames too
    r = lyra_oracle.run_batch(oracle_default, pcm, bits // 4, do_decode=True, threads=4, want_feats=True)
    ctx.reset()
    rng = np.random.default_rng(B)
    ids = rng.permutation(4000)[:B].astype(np.int32)
    for t in range(T):
        feat = ctx.extract(pcm[t], ids)
        assert np.array_equal(feat, r["feats"][t])
        idx = ctx.rvq_encode(feat, bits)
        pk = oracle_default.pack(idx, bits // 4)
        assert np.array_equal(pk, r["packets"][t])
        out = ctx.decode(pk, bits, ids)
        assert np.array_equal(out, r["pcm"][t]), f"PCM not bit-exact at step {t}"


@pytest.mark.parametrize("mode", ["exact", "gemmlowp_double", "builtin_mixed"])
def test_builtin_modes_vs_oracle_bit_exact(ctx_builtin, oracle_exact, oracle_double, oracle_mixed, mode):
    """The two TFLite-builtin flavours stay available on CPU and GPU (what the graphs compute without the delegate)."""
    from oracle import lyra_oracle
    o = {"exact": oracle_exact, "gemmlowp_double": oracle_double, "builtin_mixed": oracle_mixed}[mode]
    c = ctx_builtin[mode]
    B, T, bits = 21, 6, 184
    pcm = synth(B, T, seed=777)
    pcm[3] //= 50
    r = lyra_oracle.run_batch(o, pcm, bits // 4, do_decode=True, threads=4, want_feats=True)
    c.reset()
    ids = np.arange(B, dtype=np.int32)[::-1].copy()
    for t in range(T):
        pk = c.encode(pcm[t], bits, ids)
        assert np.array_equal(pk, r["packets"][t])
        assert np.array_equal(c.decode(pk, bits, ids), r["pcm"][t]), f"PCM not bit-exact at step {t}"


def test_fused_equals_plugin_path(ctx):
    B, T = 19, 4
    pcm = synth(B, T, seed=99)
    ctx.reset()
    a_pk, a_pcm = [], []
    for t in range(T):
        pk = ctx.encode(pcm[t], 120)
        a_pk.append(pk)
        a_pcm.append(ctx.decode(pk, 120))
    ctx.reset()
    for t in range(T):
        feat = ctx.extract(pcm[t])
        idx = ctx.rvq_encode(feat, 120)
        bytes_ = np.array([[(r[2 * j] << 4) | r[2 * j + 1] for j in range(15)] for r in idx], np.uint8)
        assert np.array_equal(bytes_, a_pk[t])
        out = ctx.generate(ctx.rvq_decode(idx))
        assert np.array_equal(out, a_pcm[t])


def test_reset_and_stream_independence(ctx):
    pcm = synth(6, 5, seed=7)
    ctx.reset()
    first = [ctx.encode(pcm[t], 184) for t in range(5)]
    # same audio on other stream ids, interleaved with unrelated traffic on the original ids
    ids = np.array([100, 101, 102, 103, 104, 105], np.int32)
    junk = synth(6, 5, seed=8)
    second = []
    for t in range(5):
        ctx.encode(junk[t], 184)
        second.append(ctx.encode(pcm[t], 184, ids))
    assert all(np.array_equal(a, b) for a, b in zip(first, second))
    # selective reset restores the fresh-state answer for those streams only
    ctx.reset(np.array([0, 1, 2], np.int32))
    again = ctx.encode(pcm[0][:3], 184, np.array([0, 1, 2], np.int32))
    assert np.array_equal(again, first[0][:3])


def test_argument_validation(ctx):
    import lyra_amd
    pcm = np.zeros((2, 320), np.int16)
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.encode(pcm, 185)  # too many bits (residual_vector_quantizer_test.cc QuantizationFailsWithTooManyBits)
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.encode(pcm, 62)   # not divisible by 4
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.encode(pcm, 64, np.array([0, 999999], np.int32))
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.encode(pcm, 64, np.array([3, 3], np.int32))   # one stream twice in a batch would race on its state
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.decode(np.zeros((2, 8), np.uint8), 64, np.array([7, 7], np.int32))
    q = lyra_amd.ResidualVectorQuantizer(ctx)
    feat = np.zeros(64, np.float32)
    assert q.Quantize(feat, 185) is None and q.Quantize(feat, 62) is None
    assert q.DecodeToLossyFeatures("0" * 185) is None and q.DecodeToLossyFeatures("0" * 62) is None
    bits = q.Quantize(feat, 64)
    assert isinstance(bits, str) and len(bits) == 64 and set(bits) <= {"0", "1"}
    assert q.DecodeToLossyFeatures(bits).shape == (64,)


def test_plugin_objects_reference_semantics(ctx, oracle_default, golden_dir):
    """soundstream_encoder_test.cc:51-57, lyra_gan_model_test.cc:60-76 behaviours."""
    import lyra_amd
    from oracle import lyra_oracle
    ctx.reset()
    enc = lyra_amd.SoundStreamEncoder(ctx, 3)
    feats = enc.Extract(np.zeros(320, np.int16))
    assert feats.shape == (64,)
    assert enc.Extract(np.zeros(321, np.int16)) is None
    st = lyra_oracle.Stream(oracle_default)
    assert np.array_equal(feats, st.encode(np.zeros(320, np.int16)))
    gan = lyra_amd.LyraGanModel(ctx, 3)
    assert gan.GenerateSamples(1) is None          # no features yet
    assert gan.AddFeatures(np.zeros(63)) is False
    assert gan.AddFeatures(feats)
    assert gan.GenerateSamples(0).size == 0
    a = gan.GenerateSamples(1)
    b = gan.GenerateSamples(319)
    assert a.size == 1 and b.size == 319
    assert gan.GenerateSamples(1) is None          # 321st sample
    want = st.decode(feats)
    assert np.array_equal(np.concatenate([a, b]), want)


def test_logmel(ctx, oracle_default, golden_dir):
    from oracle import lyra_oracle
    g = _g(golden_dir, "speech_sample1.npz")
    ctx.reset()
    B = 3
    streams = [lyra_oracle.Stream(oracle_default) for _ in range(B)]
    ids = np.array([9, 1, 77], np.int32)
    for t in range(12):
        pcm = np.stack([g["pcm_in"][(t + 5 * b) % 50] for b in range(B)])
        mel = ctx.logmel(pcm, ids)
        ref = np.stack([streams[b].logmel(pcm[b]) for b in range(B)])
        ulp = np.abs(mel.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        assert ulp.max() <= 4
    assert np.allclose(ctx.logmel(np.zeros((1, 320), np.int16), np.array([500], np.int32)),
                       np.log(np.float32(500.0)) / 10)


# ------------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,bits", [(1024, 64), (4096, 184), (8192, 120)])
def test_full_size_properties(ctx, oracle_default, B, bits):
    from oracle import lyra_oracle
    T = 3
    base = synth(64, T, seed=B)
    # every stream b replays base stream b % 64: all replicas must agree bit for bit (stream independence,
    # no cross-stream leakage at full batch), and replica 0..63 must match the oracle.
    pcm = base[:, np.arange(B) % 64]
    r = lyra_oracle.run_batch(oracle_default, base, bits // 4, do_decode=True, threads=8)
    ctx.reset()
    for t in range(T):
        pk = ctx.encode(pcm[t], bits)
        assert pk.shape == (B, (bits + 7) // 8)
        assert np.array_equal(pk[:64], r["packets"][t])
        assert np.array_equal(pk, pk[np.arange(B) % 64])
        out = ctx.decode(pk, bits)
        assert np.array_equal(out[:64], r["pcm"][t])
        assert np.array_equal(out, out[np.arange(B) % 64])
        # embedded bit-stream: fewer bits = prefix (re-encode a copy of the first 64 streams on spare ids)
    # decode-only path (config #4): generate() from features == decode() from packets
    ctx.reset()
    feats = ctx.rvq_decode(oracle_default.unpack(r["packets"][0], bits // 4))
    a = ctx.generate(feats, np.arange(64, dtype=np.int32))
    ctx.reset()
    b = ctx.decode(r["packets"][0], bits, np.arange(64, dtype=np.int32))
    assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# the device-pointer pipeline exactly as bench.py drives it (two alternating buffers, no caller sync between
# encode and decode, encode of step i+1 overlapping decode of step i on the library's streams)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,bits", [(37, 184), (1000, 64), (4096, 184)])
def test_device_pipeline_as_benchmarked(ctx, oracle_default, B, bits):
    import torch
    import lyra_amd
    from oracle import lyra_oracle
    T, R = 6, min(B, 48)
    base = synth(R, T, seed=4242 + B)
    r = lyra_oracle.run_batch(oracle_default, base, bits // 4, do_decode=True, threads=8)
    dev = torch.device("cuda", 0)
    pcm = torch.from_numpy(base[:, np.arange(B) % R].copy()).to(dev)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    nb = lyra_amd.packet_size(bits)
    packets = [torch.empty((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
    out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
    keep_pk, keep_pcm = [], []
    ctx.reset()
    torch.cuda.synchronize()
    for t in range(T):
        ctx.encode_dev(ids, pcm[t], bits, packets[t & 1])
        ctx.decode_dev(ids, packets[t & 1], bits, out[t & 1])
        if t % 2 == 1:   # sync only every other step: consecutive steps overlap in between
            ctx.synchronize()
            keep_pk.append((t, packets[t & 1].cpu().numpy()))
            keep_pcm.append((t, out[t & 1].cpu().numpy()))
    ctx.synchronize()
    for t, pk in keep_pk:
        assert np.array_equal(pk[:R], r["packets"][t]), f"packets differ at step {t}"
        assert np.array_equal(pk, pk[np.arange(B) % R])
    for t, o in keep_pcm:
        assert np.array_equal(o[:R], r["pcm"][t]), f"PCM differs at step {t}"
        assert np.array_equal(o, o[np.arange(B) % R])
