"""NoiseEstimator + DTX: lyra/noise_estimator.cc:144-245, lyra/lyra_encoder.cc:131-141.

CPU: the oracle restatement against the reference's own expectations (lyra/noise_estimator_test.cc:175-199
NoiseIdentification; the two "FiveSeconds" tests need the comfort-noise generator and live in test_comfort_noise.py).
GPU: the device recurrence against the oracle -- identical DTX decisions, state within float tolerance -- on
speech, silence and stationary noise, and DTX-enabled encoding (empty packets, encoder state frozen on noise hops).
"""
import os

import numpy as np
import pytest

from oracle import lyra_oracle

SILENCE = np.float32(np.log(np.float32(500.0)) / np.float32(10.0))   # LogMelSpectrogramExtractorImpl::GetSilenceValue


def base_noise():
    rise = SILENCE / np.float32(160)
    return (rise * np.arange(160, dtype=np.float32) + SILENCE).astype(np.float32)


def test_oracle_noise_identification(oracle_default):
    """noise_estimator_test.cc:175-199 with the peer's parameters (10 hops per update, half-lives 20 / 50 hops)."""
    ne = lyra_oracle.NoiseEstimator(oracle_default, 10, float(np.float32(0.5) ** np.float32(1 / 20)),
                                    float(np.float32(0.5) ** np.float32(1 / 50)))
    rng = np.random.default_rng(0)
    base = base_noise()
    for _ in range(250):
        ne.UpdateNoiseEstimate(base + rng.uniform(-0.1, 0.1, 160).astype(np.float32))
    periodic = np.full(160, SILENCE, np.float32)
    periodic[::20] = 1.0
    assert ne.ComputeIsNoise(base)
    assert not ne.ComputeIsNoise(periodic)


def test_oracle_first_hop_and_silence(oracle_default):
    ne = lyra_oracle.NoiseEstimator(oracle_default)
    assert np.all(ne.noise_estimate() == 0) and np.all(ne.noise_bound() == 0)
    is_noise, mel = ne.ReceiveSamples(np.zeros(320, np.int16))
    assert np.allclose(mel, SILENCE) and not is_noise           # |silence - 0| > bound 0: the estimate must update
    # first update: smoothed = tmp_min = current, estimate = min(tmp_min, smoothed) (noise_estimator.cc:52-63,180-186)
    assert np.allclose(ne.noise_estimate(), SILENCE, atol=1e-7) and np.all(ne.noise_bound() == 0)
    flags = [ne.ReceiveSamples(np.zeros(320, np.int16))[0] for _ in range(120)]
    assert all(flags)          # |silence - estimate| = 0 is not > bound 0: every further silent hop is noise
    assert np.allclose(ne.noise_estimate(), SILENCE, atol=1e-7)


def _speech_and_noise(golden_dir, hops=172):
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    speech = pcm[:hops * 320].reshape(hops, 320)
    rng = np.random.default_rng(5)
    noise = np.clip(rng.normal(0, 300, size=(hops, 320)), -32768, 32767).astype(np.int16)
    quiet = (speech // 64).astype(np.int16)
    return speech, noise, quiet


@pytest.mark.gpu
def test_gpu_noise_estimator_matches_oracle(golden_dir, oracle_default):
    """Streams: speech, stationary noise, quiet speech, silence, speech after 60 hops of noise.  Decisions identical,
    noise estimate / bound within 1e-5 of the oracle at every hop, for the encoder-side and the decoder-side slot."""
    import lyra_amd
    speech, noise, quiet = _speech_and_noise(golden_dir)
    silence = np.zeros_like(speech)
    mixed = np.concatenate([noise[:60], speech[:112]])
    streams = np.stack([speech, noise, quiet, silence, mixed], axis=1)     # [T][5][320]
    ctx = lyra_amd.LyraHip(max_streams=64)
    ids = np.array([3, 0, 41, 7, 12], np.int32)
    for side in ("encoder", "decoder"):
        ctx.reset()
        refs = [lyra_oracle.NoiseEstimator(oracle_default) for _ in range(5)]
        for t in range(streams.shape[0]):
            got = ctx.noise_receive(streams[t], ids, side=side)
            want = [r.ReceiveSamples(streams[t, b])[0] for b, r in enumerate(refs)]
            assert list(got.astype(bool)) == want, f"{side}: is_noise differs at hop {t}"
            if t % 10 == 9 or t < 5:
                est = ctx.noise_estimate(ids, side=side)
                for b, r in enumerate(refs):
                    assert np.allclose(est[b], r.noise_estimate(), rtol=1e-5, atol=1e-6), (side, t, b)
    ctx.close()


@pytest.mark.gpu
def test_gpu_dtx_encode(golden_dir, oracle_default):
    """LyraEncoder::Encode with enable_dtx (lyra_encoder.cc:131-156): noise hops give an EMPTY packet and do not run
    the feature extractor (its state stays put); other hops give exactly the packet a DTX-less encoder that only ever
    saw the non-noise hops would give."""
    import lyra_amd
    speech, noise, quiet = _speech_and_noise(golden_dir, 150)
    silence = np.zeros_like(speech)
    streams = np.stack([speech, np.concatenate([silence[:80], speech[:70]]), noise], axis=1)
    B, bits = 3, 120
    ctx = lyra_amd.LyraHip(max_streams=64)
    ids = np.array([9, 2, 30], np.int32)
    nes = [lyra_oracle.NoiseEstimator(oracle_default) for _ in range(B)]
    encs = [lyra_oracle.Stream(oracle_default) for _ in range(B)]
    n_empty = 0
    for t in range(streams.shape[0]):
        pk, nbytes = ctx.encode_dtx(streams[t], bits, ids)
        for b in range(B):
            is_noise, _ = nes[b].ReceiveSamples(streams[t, b])
            if is_noise:
                assert nbytes[b] == 0, (t, b)
                n_empty += 1
            else:
                feat = encs[b].encode(streams[t, b])
                want = oracle_default.pack(oracle_default.rvq_encode(feat, bits // 4), bits // 4)[0]
                assert nbytes[b] == bits // 8 and np.array_equal(pk[b], want), (t, b)
    assert n_empty > 20      # the silence stream does go quiet; otherwise this test checks nothing
    ctx.close()


@pytest.mark.gpu
def test_gpu_dtx_device_pipeline_equals_host_calls(golden_dir):
    """lyra_hip_encode_dtx_dev in a loop with no synchronisation (noise estimator + extractor on the encode-side stream,
    quantizer on its own stream with the live-stream mask of that call, the next call's noise kernel already running)
    against the synchronous host-pointer lyra_hip_encode_dtx on a second context: packets of live hops and the
    packet-length vector of every step identical.  Speech, late-starting speech, stationary noise and digital silence,
    replicated to 64 streams with scattered ids."""
    import torch
    import lyra_amd
    speech, noise, quiet = _speech_and_noise(golden_dir, 90)
    silence = np.zeros_like(speech)
    base = np.stack([speech, np.concatenate([silence[:40], speech[:50]]), noise, silence], axis=1)   # [T][4][320]
    T, B, bits = base.shape[0], 64, 184
    pcm = base[:, np.arange(B) % 4].copy()
    ids = np.random.default_rng(3).permutation(200)[:B].astype(np.int32)
    nb = lyra_amd.packet_size(bits)
    host = lyra_amd.LyraHip(max_streams=256)
    dev_ctx = lyra_amd.LyraHip(max_streams=256)
    dev_ctx.torch_order = False
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    d_ids = torch.from_numpy(ids).to(dev)
    d_pk = torch.zeros((T, B, nb), device=dev, dtype=torch.uint8)
    d_len = torch.full((T, B), -1, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()
    try:
        for t in range(T):
            dev_ctx.encode_dtx_dev(d_ids, d_pcm[t], bits, d_pk[t], d_len[t])
        dev_ctx.synchronize()
        got_pk, got_len = d_pk.cpu().numpy(), d_len.cpu().numpy()
        n_empty = 0
        for t in range(T):
            pk, nbytes = host.encode_dtx(pcm[t], bits, ids)
            assert np.array_equal(got_len[t], nbytes), f"packet lengths differ at step {t}"
            live = nbytes > 0
            assert np.array_equal(got_pk[t][live], pk[live]), f"packets differ at step {t}"
            n_empty += int((~live).sum())
        assert n_empty > 500      # silence and noise streams do go quiet
    finally:
        host.close()
        dev_ctx.close()
