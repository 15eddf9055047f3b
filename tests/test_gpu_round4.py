"""Round-4 regression tests on the GPU for the round-3 advisor findings:
  * a DTX run_steps at an external rate refuses an estimator set up for another rate (lyra_encoder.cc:82-85);
  * the encoder-side estimator at 48 kHz uses the 48 kHz mel filterbank (decisions differ from the 16 kHz table's and equal
    the oracle's rate-aware estimator);
  * the fourth (noise) stream is exposed;
  * a failed lyra_hip_twin_fetch leaves the context ready for the next request.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_run_steps_dtx_rate_mismatch_is_refused():
    import torch
    import lyra_amd
    B, bits = 8, 64
    dev = torch.device("cuda", 0)
    ctx = lyra_amd.LyraHip(max_streams=B)
    try:
        ids = torch.arange(B, device=dev, dtype=torch.int32)
        ring = torch.zeros((2, B, 960), device=dev, dtype=torch.int16)
        pk = [torch.zeros((B, 8), device=dev, dtype=torch.uint8) for _ in range(2)]
        nb = [torch.zeros((B,), device=dev, dtype=torch.int32) for _ in range(2)]
        out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
        ext = [torch.zeros((B, 960), device=dev, dtype=torch.int16) for _ in range(2)]
        kw = dict(d_pcm_ring=ring, d_packets=pk, d_pcm_out=out, d_packet_bytes=nb, external_rate=48000, d_ext_out=ext,
                  encode=True, decode=True, dtx=True)
        with pytest.raises(lyra_amd.LyraHipError, match="noise estimator is set up for 16000"):
            ctx.run_steps_dev(ids, bits, 2, **kw)
        ctx.set_encoder_sample_rate(48000)
        ctx.run_steps_dev(ids, bits, 2, **kw)
        ctx.synchronize()
    finally:
        ctx.close()


def test_encoder_noise_estimator_uses_the_rate_s_filterbank(golden_dir, oracle_default):
    import os
    import lyra_amd
    from oracle import lyra_oracle
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    hops = pcm[:120 * 320].reshape(120, 320)
    rng = np.random.default_rng(5)
    noise = (rng.normal(0, 300, size=(120, 320))).astype(np.int16)
    sig = np.stack([hops, noise, np.concatenate([noise[:50], hops[:70]])], axis=1)    # [T][3][320]
    ids = np.array([2, 9, 4], np.int32)
    ctx = lyra_amd.LyraHip(max_streams=16)
    try:
        flags = {}
        for rate in (16000, 48000):
            ctx.reset()
            ctx.set_encoder_sample_rate(rate)
            refs = [lyra_oracle.NoiseEstimator(oracle_default, sample_rate_hz=rate) for _ in range(3)]
            got_all = []
            for t in range(sig.shape[0]):
                got = ctx.noise_receive(sig[t], ids, side="encoder")
                want = [int(r.ReceiveSamples(sig[t, b])[0]) for b, r in enumerate(refs)]
                assert list(got.astype(int)) == want, (rate, t)
                got_all.append(got.copy())
            est = ctx.noise_estimate(ids, side="encoder")
            for b, r in enumerate(refs):
                assert np.allclose(est[b], r.noise_estimate(), rtol=0, atol=1e-4), (rate, b)
            flags[rate] = (np.array(got_all), est.copy())
        # the two rates are different estimators: other time constants AND another mel filterbank
        assert np.abs(flags[16000][1] - flags[48000][1]).max() > 1e-2
    finally:
        ctx.close()


def test_noise_stream_accessor_and_twin_fetch_recovery():
    import lyra_amd
    ctx = lyra_amd.LyraHip(max_streams=8)
    L = ctx.L
    try:
        L.lyra_hip_stream_noise.restype = C.c_void_p
        L.lyra_hip_stream_noise.argtypes = [C.c_void_p]
        L.lyra_hip_stream_decode.restype = C.c_void_p
        L.lyra_hip_stream_decode.argtypes = [C.c_void_p]
        sn, sd = L.lyra_hip_stream_noise(ctx.h), L.lyra_hip_stream_decode(ctx.h)
        assert sn and sd and sn != sd

        class Slice(C.Structure):
            _fields_ = [(n, C.c_int32) for n in ("id", "gan_off", "gen_n", "cng_off", "cng_n", "fade", "fade_dir", "out_off",
                                                 "noise_row")]
        L.lyra_hip_twin_conceal.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lyra_hip_twin_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.lyra_hip_twin_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        ids = np.array([0, 1], np.int32)

        def request(n, fetch_n, rate=16000):
            assert L.lyra_hip_twin_conceal(ctx.h, ids.ctypes.data, 2) == 0
            sl = (Slice * 2)(Slice(0, 0, n, 0, 0, 0, 1, 0, -1), Slice(1, 0, n, 0, 0, 0, 1, 0, -1))
            assert L.lyra_hip_twin_assemble(ctx.h, C.addressof(sl), 2, n) == 0, ctx.last_error()
            out = np.zeros((2, fetch_n * rate // 16000), np.int16)
            return L.lyra_hip_twin_fetch(ctx.h, 2, fetch_n, rate, out.ctypes.data), out

        rc, _ = request(320, 160)                  # sample-count mismatch: fails, reports what WAS assembled ...
        assert rc != 0 and "320 assembled" in ctx.last_error()
        rc, _ = request(160, 160)                  # ... and the next request, of another size, starts clean
        assert rc == 0, ctx.last_error()
        rc, _ = request(320, 320, rate=44100)      # unsupported rate: fails ...
        assert rc != 0
        rc, out = request(200, 200)                # ... and again leaves nothing behind
        assert rc == 0, ctx.last_error()
    finally:
        ctx.close()


def test_cu_partitioned_small_context_equals_whole_chip_context(monkeypatch):
    """Contexts of <= 1024 streams create their streams with complementary CU masks (include/lyra_hip.h "Streams"); the
    same work on whole-chip streams (LYRA_HIP_CU_MASKS=0) and on the opposite halves must give the same packets and PCM --
    a CU mask changes where workgroups run, never what they compute -- also through the pipelined `_dev` entry point."""
    import torch
    import lyra_amd
    B, bits, T = 300, 120, 6
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    pcm = torch.randint(-32768, 32768, (T, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    nb = lyra_amd.packet_size(bits)

    def run(masks):
        if masks is None:
            monkeypatch.delenv("LYRA_HIP_CU_MASKS", raising=False)
        else:
            monkeypatch.setenv("LYRA_HIP_CU_MASKS", masks)
        ctx = lyra_amd.LyraHip(max_streams=B)
        try:
            pk = [torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
            out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
            ctx.run_steps_dev(ids, bits, T, d_pcm_ring=pcm, d_packets=pk, d_pcm_out=out, encode=True, decode=True)
            ctx.synchronize()
            return [p.cpu().numpy() for p in pk], [o.cpu().numpy() for o in out]
        finally:
            ctx.close()

    want = run("0")                                            # ordinary priority streams on the whole chip
    for masks in (None, "ff00ff00,00ff00ff,ff00ff00,00ff00ff", "0000ffff,ffff0000,0000ffff,ffff0000"):
        got = run(masks)
        for a, b in zip(want[0] + want[1], got[0] + got[1]):
            assert np.array_equal(a, b), f"masks {masks}"


def test_set_stream_priorities_recreates_streams_and_keeps_results():
    """lyra_hip_set_stream_priorities drains the context and creates its three main streams anew; state, ordering and results
    are unaffected (the decoder twin asks for decode-side-first this way, host/lyra_batch_codec.cc)."""
    import torch
    import lyra_amd
    B, bits, T = 2100, 184, 5     # > 1024 streams: ordinary (priority) streams
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    pcm = torch.randint(-32768, 32768, (2 * T, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    nb = lyra_amd.packet_size(bits)
    L = lyra_amd.codec._load()

    def run(switch):
        ctx = lyra_amd.LyraHip(max_streams=B)
        try:
            pk = [torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
            out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
            kw = dict(d_pcm_ring=pcm, d_packets=pk, d_pcm_out=out, encode=True, decode=True)
            ctx.run_steps_dev(ids, bits, T, first_step=0, **kw)
            if switch:   # in the middle of a session, with work in flight
                ctx.set_stream_priorities(0, 2, 2)
                assert L.lyra_hip_set_stream_priorities(ctx.h, 3, 0, 0) != 0 and b"0 (lowest)" in L.lyra_hip_last_error(ctx.h)
            ctx.run_steps_dev(ids, bits, T, first_step=T, **kw)
            ctx.synchronize()
            return [p.cpu().numpy() for p in pk], [o.cpu().numpy() for o in out]
        finally:
            ctx.close()

    a, b = run(False), run(True)
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert np.array_equal(x, y)
