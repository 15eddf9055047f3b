"""Whole-file known answers and the reference's own acceptance criterion.

SURVEY.md A.6 / BASELINE.md section 4 (BASELINE config #1: encoder_main + decoder_main on
testdata/sample1_16kHz.wav): 55,177 samples -> 172 full hops -> 172 packets (1,376 B at 3200 bps); sha256[:16] of the
concatenated packets of the first 150 hops per bitrate and per requantisation mode; per-hop log-spectral distance
< 2.0 between the 64-bin log-mel of input and decoded audio (lyra/lyra_integration_test.cc:49-149, on the first
3 s = 150 hops of sample1 at 64 / 120 / 184 bits).  Inputs: tests/golden/sample_wavs.npz (tools/make_golden.py).

CPU tests pin the oracle; the `gpu` tests run the same checks through the C ABI on the GPU, with B = 1 and with
the file replicated over 4096 streams so that every ring history wraps (172 hops = 9.5 x PHASE_MOD).
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import lyra_oracle
from oracle.logmel_np import LogMelExtractor

SHA = {  # mode -> bits -> sha256[:16] of the first 150 packets of sample1_16kHz.wav
    "exact": {64: "a460adeb4ac4f1c8", 120: "8ddf7e64edc74415", 184: "afe043bd95a159ca"},
    "gemmlowp_double": {64: "ad3f6cb809f3699f", 120: "134112ecd23c4d0f", 184: "9f575b7344f402eb"},
    # round 4, mode "xnnpack" (default): the same 150 hops through oracle/tflite_interp.py executing the flatbuffers with
    # XNNPACK's arithmetic, every op of which equals real XNNPACK (tests/test_xnnpack_witness.py)
    "xnnpack": {64: "6d4759d9ca14c62a", 120: "446e2075921b6970", 184: "6045429471825364"},
    # round 6, mode "builtin_mixed": the same 150 hops through oracle/tflite_interp.py in that mode (per-operator mixture of
    # TFLite's builtin int8 kernels; tools/make_golden.py run_codec(frames, "builtin_mixed"))
    "builtin_mixed": {64: "c1fbbe3ce79e64e2", 120: "c962d6b698b26286", 184: "0ce68bd4e0120eba"},
}
MODES = ["xnnpack", "exact", "gemmlowp_double", "builtin_mixed"]


def hops_of(golden_dir, name):
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))[name]
    n = pcm.size // 320
    return pcm[:n * 320].reshape(n, 320)


def sha16(packets):
    return hashlib.sha256(np.ascontiguousarray(packets).tobytes()).hexdigest()[:16]


def lsd_per_hop(pcm_in, pcm_out):
    """lyra_integration_test.cc:101-142: 64-bin log-mel of both signals, 10 * sqrt(mean((a-b)^2))... per hop."""
    e1, e2 = LogMelExtractor(num_mel=64), LogMelExtractor(num_mel=64)
    out = []
    for a, b in zip(pcm_in, pcm_out):
        x, y = e1.extract(a), e2.extract(b)
        out.append(10 * np.sqrt(((x - y) ** 2).sum() / 64))
    return np.array(out)


# ------------------------------------------------------------------------------------------------------------------
# oracle (CPU)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
def test_oracle_whole_file_packets(golden_dir, oracle_exact, oracle_double, oracle_xnnpack, oracle_mixed, mode):
    o = {"exact": oracle_exact, "gemmlowp_double": oracle_double, "xnnpack": oracle_xnnpack, "builtin_mixed": oracle_mixed}[mode]
    hops = hops_of(golden_dir, "sample1_16kHz")
    assert hops.shape[0] == 172
    for bits in (64, 120, 184):
        r = lyra_oracle.run_batch(o, hops[:, None, :], bits // 4, do_decode=False)
        pk = r["packets"][:, 0]
        assert pk.shape == (172, bits // 8) and pk.size == 172 * (bits // 8)
        if bits == 64:
            assert pk.size == 1376      # the .lyra file of BASELINE config #1
        assert sha16(pk[:150]) == SHA[mode][bits]


@pytest.mark.parametrize("mode", ["xnnpack", "exact"])
def test_oracle_lsd_both_wavs(golden_dir, oracle_exact, oracle_xnnpack, mode):
    o = oracle_exact if mode == "exact" else oracle_xnnpack
    for name in ("sample1_16kHz", "sample2_16kHz"):
        hops = hops_of(golden_dir, name)
        for bits in (64, 120, 184):
            r = lyra_oracle.run_batch(o, hops[:, None, :], bits // 4, do_decode=True)
            lsd = lsd_per_hop(hops[:150], r["pcm"][:150, 0])
            assert lsd.max() < 2.0, (name, bits, lsd.max())


# ------------------------------------------------------------------------------------------------------------------
# GPU, through the C ABI
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx_by_mode():
    import lyra_amd
    made = {}

    def get(mode):
        if mode not in made:
            made[mode] = lyra_amd.LyraHip(max_streams=4096, requant=mode)
        return made[mode]
    yield get
    for c in made.values():
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_whole_file_packets_b1(golden_dir, ctx_by_mode, mode):
    ctx = ctx_by_mode(mode)
    hops = hops_of(golden_dir, "sample1_16kHz")
    sid = np.array([77], np.int32)
    for bits in (64, 120, 184):
        ctx.reset()
        pk = np.stack([ctx.encode(h[None], bits, sid)[0] for h in hops])
        assert pk.shape == (172, bits // 8)
        assert sha16(pk[:150]) == SHA[mode][bits]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_whole_file_b4096_all_hops(golden_dir, ctx_by_mode, oracle_exact, oracle_double, oracle_xnnpack, oracle_mixed, mode):
    """The whole file on 4096 streams at once (device-pointer pipeline as benchmarked), every stream a replica:
    packets hash to the known answer, PCM is bit-exact versus the oracle, all replicas agree, over all 172 hops."""
    import torch
    import lyra_amd
    ctx = ctx_by_mode(mode)
    o = {"exact": oracle_exact, "gemmlowp_double": oracle_double, "xnnpack": oracle_xnnpack, "builtin_mixed": oracle_mixed}[mode]
    hops = hops_of(golden_dir, "sample1_16kHz")
    B, bits = 4096, 184
    ref = lyra_oracle.run_batch(o, hops[:, None, :], bits // 4, do_decode=True)
    dev = torch.device("cuda", 0)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    pcm = torch.from_numpy(hops.copy()).to(dev)
    packets = [torch.empty((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
    out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
    ctx.reset()
    got_pk, got_pcm = [], []
    for t in range(hops.shape[0]):
        x = pcm[t].unsqueeze(0).expand(B, 320).contiguous()
        ctx.encode_dev(ids, x, bits, packets[t & 1])
        ctx.decode_dev(ids, packets[t & 1], bits, out[t & 1])
        ctx.synchronize()
        p, y = packets[t & 1].cpu().numpy(), out[t & 1].cpu().numpy()
        assert (p == p[0]).all() and (y == y[0]).all(), f"replicas disagree at hop {t}"
        got_pk.append(p[0]); got_pcm.append(y[0])
    got_pk, got_pcm = np.stack(got_pk), np.stack(got_pcm)
    assert sha16(got_pk[:150]) == SHA[mode][184]
    assert sha16(got_pk[:150, :8]) == SHA[mode][64]      # embedded bit-stream: the 64-bit packet is a prefix
    assert np.array_equal(got_pk, ref["packets"][:, 0])
    assert np.array_equal(got_pcm, ref["pcm"][:, 0])


@pytest.mark.gpu
def test_gpu_lsd_below_two(golden_dir, ctx_by_mode):
    """The reference's acceptance test on GPU output (lyra_integration_test.cc:131-142), both 16 kHz wavs x 3 bitrates."""
    ctx = ctx_by_mode("xnnpack")
    for name in ("sample1_16kHz", "sample2_16kHz"):
        hops = hops_of(golden_dir, name)
        ids = np.array([5, 6, 7], np.int32)
        ctx.reset()
        outs = {64: [], 120: [], 184: []}
        for h in hops[:150]:
            for k, bits in enumerate((64, 120, 184)):      # three independent streams, one per bitrate
                pk = ctx.encode(h[None], bits, ids[k:k + 1])
                outs[bits].append(ctx.decode(pk, bits, ids[k:k + 1])[0])
        for bits in (64, 120, 184):
            lsd = lsd_per_hop(hops[:150], outs[bits])
            assert lsd.max() < 2.0, (name, bits, lsd.max())
