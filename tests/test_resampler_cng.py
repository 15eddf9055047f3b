"""Resampler and ComfortNoiseGenerator: lyra/resampler.cc:30-66, lyra/comfort_noise_generator.cc:74-119.

Both sit on un-vendored audio_dsp classes (QResampler, MelFilterbank::EstimateInverse, InverseSpectrogram), and the
comfort noise takes its phases from a non-deterministic generator: the oracle restates the published constructions
and is pinned by exactly what the reference's own tests hold (CPU tests below = lyra/resampler_test.cc,
lyra/comfort_noise_generator_test.cc, lyra/noise_estimator_test.cc FiveSeconds*).  `gpu` tests compare the device
kernels with the oracle: resampler bit-exact (float FIR, same tap order), comfort noise <= 1 int16 LSB.
"""
import os

import numpy as np
import pytest

from oracle import lyra_oracle
from oracle.logmel_np import LogMelExtractor

RATES = (8000, 16000, 32000, 48000)   # kSupportedSampleRates (lyra_config.h)
SILENCE = np.float32(np.log(np.float32(500.0)) / np.float32(10.0))


def lsd(a, b):
    return 10 * np.sqrt(((np.asarray(a, np.float32) - np.asarray(b, np.float32)) ** 2).sum() / len(a))


# ---- lyra/resampler_test.cc ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("rate", RATES)
def test_oracle_resampling_all_zeros(rate):
    for a, b in ((16000, rate), (rate, 16000)):
        r = lyra_oracle.Resampler(a, b)
        out = r.Resample(np.zeros(a // 50, np.int16))
        assert out.size == b // 50 and not out.any()


def test_oracle_upsample_then_downsample_similar():
    t = np.arange(100)
    samples = (np.sin(2 * np.pi * 1000 * t / 16000) * 100).astype(np.int16)
    up = lyra_oracle.Resampler(16000, 32000).Resample(samples)
    down = lyra_oracle.Resampler(32000, 16000).Resample(up)
    assert up.size == 200 and down.size == 100
    delay = 25   # 17 + floor(17 / 2), resampler_test.cc:79-84
    assert np.abs(samples[:100 - delay].astype(int) - down[delay:].astype(int)).max() <= 25


def test_oracle_resampler_extreme_values_clip():
    x = np.where((np.arange(320) // 2) % 2 == 0, -32768, 32767).astype(np.int16)
    out = lyra_oracle.Resampler(16000, 32000).Resample(x)
    assert out.size == 640      # (and no overflow: ClipToInt16)


def test_oracle_resampler_streaming_equals_one_shot():
    rng = np.random.default_rng(1)
    x = rng.integers(-20000, 20000, 960 * 3).astype(np.int16)
    for a, b in ((48000, 16000), (16000, 48000), (8000, 16000), (16000, 8000)):
        n = x.size // (a // 50) * (a // 50)
        one = lyra_oracle.Resampler(a, b).Resample(x[:n])
        r = lyra_oracle.Resampler(a, b)
        parts = np.concatenate([r.Resample(x[i:i + a // 50]) for i in range(0, n, a // 50)])
        assert np.array_equal(one, parts)


# ---- lyra/comfort_noise_generator_test.cc -----------------------------------------------------------------------------
def test_oracle_cng_no_energy(oracle_exact):
    g = lyra_oracle.ComfortNoiseGenerator(oracle_exact, seed=1)
    assert not g.generate(np.zeros(160, np.float32)).any()      # BasicUseCaseSucceeds: Each(0.0)


def test_oracle_cng_similar_features(oracle_exact):
    """GeneratedNoiseHasSimilarFeatures: LSD(input log-mel, log-mel of the generated noise) < 0.7 after 10 hops."""
    rng = np.random.default_rng(1)
    x = rng.integers(-10000, 10001, 320).astype(np.int16)
    ein, eout = LogMelExtractor(), LogMelExtractor()
    g = lyra_oracle.ComfortNoiseGenerator(oracle_exact, seed=7)
    for _ in range(10):
        fin = ein.extract(x)
        fout = eout.extract(g.generate(fin))
    assert lsd(fin, fout) < 0.7


# ---- lyra/noise_estimator_test.cc: FiveSecondsSilence / FiveSecondsSparseEnergy ------------------------------------------
def test_oracle_five_seconds_silence(oracle_exact):
    g = lyra_oracle.ComfortNoiseGenerator(oracle_exact, seed=3)
    ne = lyra_oracle.NoiseEstimator(oracle_exact)
    silence = np.full(160, SILENCE, np.float32)
    for i in range(250):
        ne.ReceiveSamples(g.generate(silence))
        assert lsd(silence, ne.noise_estimate()) < 0.2, f"noise estimate dissimilar at frame {i}"


def test_oracle_five_seconds_sparse_energy(oracle_exact):
    g = lyra_oracle.ComfortNoiseGenerator(oracle_exact, seed=4)
    ne = lyra_oracle.NoiseEstimator(oracle_exact)
    base = (SILENCE / np.float32(160) * np.arange(160, dtype=np.float32) + SILENCE).astype(np.float32)
    rng = np.random.default_rng(0)
    for _ in range(250):
        f = base.copy()
        f[rng.integers(0, 11, 160) == 0] = 1.0       # kMaxPower with probability 1 / 11
        ne.ReceiveSamples(g.generate(f))
    assert lsd(base, ne.noise_estimate()) < 0.7


# ---- GPU vs oracle ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("rate", [8000, 32000, 48000])
def test_gpu_resampler_bit_exact(golden_dir, rate):
    """Streaming hop by hop, encoder side (external -> 16 kHz) and decoder side (16 kHz -> external), several streams
    with scattered ids, speech + full-scale alternating input (clipping): every sample equal to the oracle's."""
    import lyra_amd
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    ctx = lyra_amd.LyraHip(max_streams=64)
    ids = np.array([5, 0, 33], np.int32)
    T = 12
    for side, a, b in (("encoder", rate, 16000), ("decoder", 16000, rate)):
        hop = a // 50
        rng = np.random.default_rng(rate)
        streams = [pcm[4000:4000 + T * hop], rng.integers(-32768, 32768, T * hop).astype(np.int16),
                   np.where((np.arange(T * hop) // 2) % 2 == 0, -32768, 32767).astype(np.int16)]
        refs = [lyra_oracle.Resampler(a, b) for _ in streams]
        for t in range(T):
            x = np.stack([s[t * hop:(t + 1) * hop] for s in streams])
            got = ctx.resample(x, a, b, ids, side=side)
            assert got.shape == (3, b // 50)
            for i, r in enumerate(refs):
                assert np.array_equal(got[i], r.Resample(x[i])), (side, t, i)
    with pytest.raises(lyra_amd.LyraHipError):
        ctx.resample(np.zeros((1, 441), np.int16), 44100, 16000)
    ctx.close()


@pytest.mark.gpu
def test_gpu_comfort_noise_matches_oracle(golden_dir, oracle_exact):
    """Same features, same seed: device comfort noise within 1 int16 LSB of the oracle over 30 hops (fp64 sin / cos /
    exp differ in the last place between the device and the host libm), silence features give digital silence, and
    the generated noise has the requested log-mel (the reference's LSD < 0.7 criterion)."""
    import lyra_amd
    g = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    ein = LogMelExtractor()
    feats = np.stack([ein.extract(g[(40 + t) * 320:(41 + t) * 320]) for t in range(30)])
    ctx = lyra_amd.LyraHip(max_streams=64)
    ctx.set_cng_seed(1234)
    ids = np.array([7, 2], np.int32)
    refs = [lyra_oracle.ComfortNoiseGenerator(oracle_exact, seed=1234 ^ int(i)) for i in ids]
    eout = LogMelExtractor()
    worst = 0
    for t in range(30):
        f = np.stack([feats[t], feats[29 - t]])
        got = ctx.comfort_noise(f, ids)
        for i, r in enumerate(refs):
            want = r.generate(f[i])
            worst = max(worst, int(np.abs(got[i].astype(int) - want.astype(int)).max()))
    assert worst <= 1, worst
    ctx.reset()
    assert not ctx.comfort_noise(np.zeros((1, 160), np.float32), ids[:1]).any()
    x = np.random.default_rng(1).integers(-10000, 10001, 320).astype(np.int16)
    e1, e2 = LogMelExtractor(), LogMelExtractor()
    for _ in range(10):
        fin = e1.extract(x)
        fout = e2.extract(ctx.comfort_noise(fin[None], ids[1:])[0])
    assert lsd(fin, fout) < 0.7
    ctx.close()


@pytest.mark.gpu
def test_gpu_comfort_noise_from_noise_estimate(golden_dir, oracle_exact):
    """features = NULL: the decoder-side noise estimate of each stream drives the generator (lyra_decoder.cc:328-340)."""
    import lyra_amd
    ctx = lyra_amd.LyraHip(max_streams=16)
    rng = np.random.default_rng(2)
    ids = np.array([3], np.int32)
    for _ in range(60):
        ctx.noise_receive(np.clip(rng.normal(0, 400, (1, 320)), -32768, 32767).astype(np.int16), ids, side="decoder")
    est = ctx.noise_estimate(ids, side="decoder")
    a = ctx.comfort_noise(None, ids, B=1)
    ctx2 = lyra_amd.LyraHip(max_streams=16)
    b = ctx2.comfort_noise(est, ids)
    assert np.array_equal(a, b)
    assert np.abs(a).max() > 50       # audible noise, not silence
    ctx.close(); ctx2.close()


@pytest.mark.gpu
def test_gpu_device_pointer_variants_equal_host_variants(golden_dir):
    """lyra_hip_resample_dev / lyra_hip_noise_receive_dev / lyra_hip_comfort_noise_dev against the host-pointer entry
    points on a second context fed the same hops: resampled audio, is_noise flags and comfort-noise hops identical
    (the host variants are the ones pinned to the oracle above and in test_noise_estimator.py)."""
    import torch
    import lyra_amd
    rng = np.random.Generator(np.random.PCG64(99))
    B, T = 24, 40
    ids = rng.permutation(100)[:B].astype(np.int32)
    speech = np.load(os.path.join(golden_dir, "speech_sample1.npz"))["pcm_in"]          # [50][320]
    hops = np.empty((T, B, 320), np.int16)
    for b in range(B):
        kind = b % 3
        if kind == 0:
            hops[:, b] = speech[(np.arange(T) + b) % 50]
        elif kind == 1:
            hops[:, b] = rng.normal(0, 300, size=(T, 320)).astype(np.int16)
        else:
            hops[:, b] = 0
    audio48 = rng.integers(-20000, 20000, size=(T, B, 960)).astype(np.int16)
    dev = torch.device("cuda", 0)
    a, b_ = lyra_amd.LyraHip(max_streams=128), lyra_amd.LyraHip(max_streams=128)
    a.set_cng_seed(1234); b_.set_cng_seed(1234)
    d_ids = torch.from_numpy(ids).to(dev)
    try:
        for t in range(T):
            # resampler 48 kHz -> 16 kHz (encoder side) and 16 -> 48 (decoder side)
            want = a.resample(audio48[t], 48000, 16000, ids, side="encoder")
            d_out = torch.empty((B, 320), device=dev, dtype=torch.int16)
            b_.resample_dev(d_ids, torch.from_numpy(audio48[t]).to(dev), 48000, 16000, d_out, side="encoder")
            b_.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), want), f"48->16 differs at hop {t}"
            want = a.resample(hops[t], 16000, 48000, ids, side="decoder")
            d_out = torch.empty((B, 960), device=dev, dtype=torch.int16)
            b_.resample_dev(d_ids, torch.from_numpy(hops[t]).to(dev), 16000, 48000, d_out, side="decoder")
            b_.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), want), f"16->48 differs at hop {t}"
            # decoder-side noise estimator, then comfort noise from its estimate
            want = a.noise_receive(hops[t], ids, side="decoder")
            d_flag = torch.empty((B,), device=dev, dtype=torch.int32)
            b_.noise_receive_dev(d_ids, torch.from_numpy(hops[t]).to(dev), d_flag, side="decoder")
            b_.synchronize()
            assert np.array_equal(d_flag.cpu().numpy(), want), f"is_noise differs at hop {t}"
            want = a.comfort_noise(None, ids, B)
            d_pcm = torch.empty((B, 320), device=dev, dtype=torch.int16)
            b_.comfort_noise_dev(d_ids, None, d_pcm)
            b_.synchronize()
            assert np.array_equal(d_pcm.cpu().numpy(), want), f"comfort noise differs at hop {t}"
    finally:
        a.close()
        b_.close()
