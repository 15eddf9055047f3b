"""GPU tests added in round 3.

* the two benchmarked shapes no earlier test touched (bench.py --config 5 on one GPU: 32,768 streams at 120 bits;
  --config 4 as benchmarked: decode only, 8,192 streams, two sub-batches), hop by hop against the oracle
  (lyra_benchmark_lib.cc:121-160 is the loop being reproduced);
* lyra_hip_run_steps_dev (many hops from one C call) against the individual `_dev` calls;
* the stream-ordering guarantees of include/lyra_hip.h "Streams" with the caller NOT serialising anything
  (torch_order = False): encode_dev -> another encode-side call -> decode_dev; mixed sub-batch splits;
* the 48 kHz resampler past the point where a wrapping input counter would slip the decimation phase
  (resampler.cc:30-66 keeps no such counter: QResampler's phase is exact for any stream length).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth(B, T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(-32768, 32768, size=(T, B, 320)).astype(np.int16)


def test_config5_32768_streams_one_gpu(oracle_default):
    """bench.py --config 5 at N = 1: one context of 32,768 streams (2.65 GB of state; every (size_t)id * stride),
    120 bits, 20 hops through encode_dev / decode_dev with steps overlapping: packets and PCM of the 64 base streams
    equal to the oracle, and every replica equal to its base stream, at every hop."""
    import torch
    import lyra_amd
    from oracle import lyra_oracle
    B, R, T, bits = 32768, 64, 20, 120
    base = synth(R, T, seed=55555)
    base[7:9] //= 50
    ref = lyra_oracle.run_batch(oracle_default, base, bits // 4, do_decode=True, threads=8)
    dev = torch.device("cuda", 0)
    ctx = lyra_amd.LyraHip(max_streams=B, requant="xnnpack")
    try:
        ctx.torch_order = False
        # replicas scattered over the id space: stream id = a permutation, so base stream r lives at ids r, r+64, ...
        # of the *batch*, which are arbitrary slots of the state arena
        perm = np.random.default_rng(5).permutation(B).astype(np.int32)
        ids = torch.from_numpy(perm).to(dev)
        rep = np.arange(B) % R
        pcm = torch.from_numpy(base[:, rep].copy()).to(dev)
        nb = lyra_amd.packet_size(bits)
        pk = [torch.empty((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
        out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
        torch.cuda.synchronize()
        for t in range(T):
            ctx.encode_dev(ids, pcm[t], bits, pk[t & 1])
            ctx.decode_dev(ids, pk[t & 1], bits, out[t & 1])
            if t & 1:
                ctx.synchronize()
                for tt in (t - 1, t):
                    p, o = pk[tt & 1].cpu().numpy(), out[tt & 1].cpu().numpy()
                    assert np.array_equal(p[:R], ref["packets"][tt]), f"packets differ at hop {tt}"
                    assert np.array_equal(p, p[rep]), f"replica packets differ at hop {tt}"
                    assert np.array_equal(o[:R], ref["pcm"][tt]), f"PCM differs at hop {tt}"
                    assert np.array_equal(o, o[rep]), f"replica PCM differs at hop {tt}"
    finally:
        ctx.close()


@pytest.mark.parametrize("entry", ["generate_dev", "decode_dev"])
def test_config4_decode_only_8192_two_subbatches(oracle_default, entry):
    """bench.py --config 4 as benchmarked: decode only, 8,192 streams, LYRA_HIP_SUBBATCHES = 2, 20 hops, no caller
    synchronisation inside pairs of steps.  Oracle: features -> Stream.decode (lyra_gan_model.cc:53-64) / packets ->
    decode."""
    import torch
    import lyra_amd
    from oracle import lyra_oracle
    B, R, T, bits = 8192, 64, 20, 120
    base = synth(R, T, seed=4444)
    ref = lyra_oracle.run_batch(oracle_default, base, bits // 4, do_decode=True, threads=8)
    feats = np.stack([oracle_default.rvq_decode(oracle_default.unpack(ref["packets"][t], bits // 4)) for t in range(T)])
    dev = torch.device("cuda", 0)
    ctx = lyra_amd.LyraHip(max_streams=B, requant="xnnpack", sub_batches=2)
    try:
        ctx.torch_order = False
        rep = np.arange(B) % R
        ids = torch.arange(B, device=dev, dtype=torch.int32)
        d_feat = torch.from_numpy(feats[:, rep].copy()).to(dev)
        d_pk = torch.from_numpy(ref["packets"][:, rep].copy()).to(dev)
        out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
        torch.cuda.synchronize()
        for t in range(T):
            if entry == "generate_dev":
                ctx.generate_dev(ids, d_feat[t], out[t & 1])
            else:
                ctx.decode_dev(ids, d_pk[t], bits, out[t & 1])
            if t & 1:
                ctx.synchronize()
                for tt in (t - 1, t):
                    o = out[tt & 1].cpu().numpy()
                    assert np.array_equal(o[:R], ref["pcm"][tt]), f"{entry}: PCM differs at hop {tt}"
                    assert np.array_equal(o, o[rep]), f"{entry}: replica PCM differs at hop {tt}"
    finally:
        ctx.close()


def _speech_noise_silence(golden_dir, T, B):
    speech = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    speech = speech[:T * 320].reshape(T, 320)
    rng = np.random.default_rng(17)
    noise = np.clip(rng.normal(0, 300, size=(T, 320)), -32768, 32767).astype(np.int16)
    late = np.concatenate([np.zeros((T // 3, 320), np.int16), speech[:T - T // 3]])
    kinds = [speech, noise, late, np.zeros_like(speech)]
    return np.stack([kinds[b % 4] for b in range(B)], axis=1).copy()     # [T][B][320]


@pytest.mark.parametrize("mode", ["plain", "dtx+noise", "48k", "features", "packet_ring"])
def test_run_steps_equals_individual_calls(golden_dir, mode):
    """lyra_hip_run_steps_dev over [1, 2, 5, 12, 30] cumulative steps on context A against the individual `_dev` calls
    on context B (state is carried, so equal outputs at every checkpoint means equal at every hop that fed them)."""
    import torch
    import lyra_amd
    B, T, bits = 96, 30, 184
    ext = 48000 if mode == "48k" else 16000
    n_ext = 320 * ext // 16000
    pcm16 = _speech_noise_silence(golden_dir, T, B)
    if ext == 16000:
        ring_np = pcm16
    else:
        ring_np = np.repeat(pcm16, 3, axis=2)   # crude 48 kHz rendition; both contexts see the same samples
    dev = torch.device("cuda", 0)
    ids_np = np.random.default_rng(8).permutation(300)[:B].astype(np.int32)
    ids = torch.from_numpy(ids_np).to(dev)
    ring = torch.from_numpy(ring_np).to(dev)
    nb = lyra_amd.packet_size(bits)
    dtx, noise = mode == "dtx+noise", mode == "dtx+noise"
    A, Bc = lyra_amd.LyraHip(max_streams=512), lyra_amd.LyraHip(max_streams=512)
    A.torch_order = Bc.torch_order = False

    def bufs():
        return dict(pk=[torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)],
                    nbytes=[torch.full((B,), -1, device=dev, dtype=torch.int32) for _ in range(2)],
                    out=[torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)],
                    ext=[torch.zeros((B, n_ext), device=dev, dtype=torch.int16) for _ in range(2)],
                    flag=torch.full((B,), -1, device=dev, dtype=torch.int32),
                    in16=torch.zeros((B, 320), device=dev, dtype=torch.int16))
    a, b = bufs(), bufs()
    feats = pkring = None
    if mode == "features":     # a ring of three feature frames: step t generates from frame t % 3
        feats = torch.from_numpy(np.random.default_rng(2).normal(0, 1.5, size=(3, B, 64)).astype(np.float32)).to(dev)
    if mode == "packet_ring":  # decode-only from a ring of four received packets per stream
        pkring = torch.from_numpy(np.random.default_rng(3).integers(0, 256, size=(4, B, nb)).astype(np.uint8)).to(dev)
    torch.cuda.synchronize()
    try:
        done = 0
        for upto in (1, 2, 5, 12, 30):
            decode_only = feats is not None or pkring is not None
            A.run_steps_dev(ids, bits, upto - done, first_step=done, d_pcm_ring=None if decode_only else ring,
                            d_packets=a["pk"], d_pcm_out=a["out"], d_features=feats, d_packet_ring=pkring,
                            d_packet_bytes=a["nbytes"] if dtx else None, d_is_noise=a["flag"] if noise else None,
                            external_rate=ext, d_ext_out=a["ext"] if ext != 16000 else None,
                            encode=not decode_only, decode=True, dtx=dtx, decoder_noise=noise)
            for t in range(done, upto):
                s = t & 1
                if feats is not None:
                    Bc.generate_dev(ids, feats[t % 3], b["out"][s])
                    continue
                if pkring is not None:
                    Bc.decode_dev(ids, pkring[t % 4], bits, b["out"][s])
                    continue
                x = ring[t % T]
                if ext != 16000:
                    Bc.resample_dev(ids, x, ext, 16000, b["in16"], side="encoder")
                    x = b["in16"]
                if dtx:
                    Bc.encode_dtx_dev(ids, x, bits, b["pk"][s], b["nbytes"][s])
                else:
                    Bc.encode_dev(ids, x, bits, b["pk"][s])
                Bc.decode_dev(ids, b["pk"][s], bits, b["out"][s])
                if noise:
                    Bc.noise_receive_dev(ids, b["out"][s], b["flag"], side="decoder")
                if ext != 16000:
                    Bc.resample_dev(ids, b["out"][s], 16000, ext, b["ext"][s], side="decoder")
                Bc.synchronize()     # B is the plain, serialised rendition
            done = upto
            A.synchronize()
            Bc.synchronize()
            for s in range(2 if upto > 1 else 1):
                if dtx:
                    la, lb = a["nbytes"][s].cpu().numpy(), b["nbytes"][s].cpu().numpy()
                    assert np.array_equal(la, lb), f"{mode}: packet lengths differ after {upto} steps"
                    live = la > 0
                    assert np.array_equal(a["pk"][s].cpu().numpy()[live], b["pk"][s].cpu().numpy()[live])
                elif not decode_only:
                    assert torch.equal(a["pk"][s], b["pk"][s]), f"{mode}: packets differ after {upto} steps"
                assert torch.equal(a["out"][s], b["out"][s]), f"{mode}: PCM differs after {upto} steps"
                if ext != 16000:
                    assert torch.equal(a["ext"][s], b["ext"][s]), f"{mode}: external-rate PCM differs after {upto} steps"
            if noise:
                assert torch.equal(a["flag"], b["flag"]), f"{mode}: is_noise differs after {upto} steps"
        if dtx:
            assert int((a["nbytes"][1] == 0).sum()) > 0      # some stream did go quiet
    finally:
        A.close()
        Bc.close()


def test_decode_sees_quantizer_behind_other_encode_side_call(oracle_default):
    """include/lyra_hip.h "Streams" (1): a decode-side call is ordered after EVERY earlier encode-side call.
    encode_dev -> extract_dev (other streams; records a newer encode-side event while the quantizer of encode_dev is
    still running on its own stream) -> decode_dev on the packets.  No caller-side ordering at all."""
    import torch
    import lyra_amd
    from oracle import lyra_oracle
    B, R, T, bits = 4096, 32, 8, 184
    base = synth(R, T, seed=777)
    ref = lyra_oracle.run_batch(oracle_default, base, bits // 4, do_decode=True, threads=8)
    dev = torch.device("cuda", 0)
    ctx = lyra_amd.LyraHip(max_streams=B + 64)
    try:
        ctx.torch_order = False
        rep = np.arange(B) % R
        ids = torch.arange(B, device=dev, dtype=torch.int32)
        other = torch.arange(B, B + 8, device=dev, dtype=torch.int32)
        pcm = torch.from_numpy(base[:, rep].copy()).to(dev)
        junk = torch.zeros((8, 320), device=dev, dtype=torch.int16)
        feat = torch.empty((8, 64), device=dev, dtype=torch.float32)
        nb = lyra_amd.packet_size(bits)
        pk = [torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
        out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
        torch.cuda.synchronize()
        for t in range(T):
            pk[t & 1].zero_()
            torch.cuda.synchronize()
            ctx.encode_dev(ids, pcm[t], bits, pk[t & 1])
            ctx.extract_dev(other, junk, feat)          # tiny: done long before the 4096-frame quantizer
            ctx.decode_dev(ids, pk[t & 1], bits, out[t & 1])
            ctx.synchronize()
            assert np.array_equal(pk[t & 1].cpu().numpy()[:R], ref["packets"][t])
            o = out[t & 1].cpu().numpy()
            assert np.array_equal(o[:R], ref["pcm"][t]), f"decode ran ahead of the quantizer at step {t}"
            assert np.array_equal(o, o[rep])
    finally:
        ctx.close()


def test_mixed_subbatch_splits_keep_feature_buffers_safe():
    """LYRA_HIP_SUBBATCHES = 2: split calls (B >= 128) alternating with unsplit ones (small B, DTX) reuse the same two
    feature buffers; the packets must equal those of an unsplit context fed the same calls."""
    import torch
    import lyra_amd
    bits, T = 184, 12
    big, small = 4096, 40
    pcm = synth(big, T, seed=31)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    nb = lyra_amd.packet_size(bits)
    got = {}
    for sub in (2, 1):
        ctx = lyra_amd.LyraHip(max_streams=big, sub_batches=sub)
        ctx.torch_order = False
        ids_big = torch.arange(big, device=dev, dtype=torch.int32)
        pk = torch.zeros((T, big, nb), device=dev, dtype=torch.uint8)
        ln = torch.zeros((T, small), device=dev, dtype=torch.int32)
        torch.cuda.synchronize()
        try:
            for t in range(T):
                if t % 3 == 0:
                    ctx.encode_dev(ids_big, d_pcm[t], bits, pk[t])
                elif t % 3 == 1:
                    ctx.encode_dev(ids_big[:small], d_pcm[t, :small], bits, pk[t, :small])
                else:
                    ctx.encode_dtx_dev(ids_big[:small], d_pcm[t, :small], bits, pk[t, :small], ln[t])
            ctx.synchronize()
            got[sub] = pk.cpu().numpy()
        finally:
            ctx.close()
    assert np.array_equal(got[2], got[1])


def test_resampler_48k_past_the_old_counter_wrap():
    """1,150 hops of 48 kHz -> 16 kHz (a 2^20-sample input counter wrapped after 1,093 hops and slipped the decimation
    phase by one sample) and 16 -> 48 kHz back, two streams, against the oracle's streaming resampler."""
    import lyra_amd
    from oracle import lyra_oracle
    T = 1150
    rng = np.random.default_rng(48)
    x = (8000 * np.sin(np.arange(T * 960) * 2 * np.pi * 440 / 48000)).astype(np.int16)
    streams = np.stack([x, rng.integers(-20000, 20000, T * 960).astype(np.int16)])
    ctx = lyra_amd.LyraHip(max_streams=16)
    ids = np.array([3, 9], np.int32)
    down = [lyra_oracle.Resampler(48000, 16000) for _ in range(2)]
    up = [lyra_oracle.Resampler(16000, 48000) for _ in range(2)]
    try:
        for t in range(T):
            hop = streams[:, t * 960:(t + 1) * 960]
            got = ctx.resample(hop, 48000, 16000, ids, side="encoder")
            want = np.stack([down[i].Resample(hop[i]) for i in range(2)])
            assert np.array_equal(got, want), f"48->16 differs at hop {t}"
            back = ctx.resample(got, 16000, 48000, ids, side="decoder")
            wantb = np.stack([up[i].Resample(want[i]) for i in range(2)])
            assert np.array_equal(back, wantb), f"16->48 differs at hop {t}"
    finally:
        ctx.close()


def test_run_steps_full_pipeline_at_benchmark_scale(golden_dir):
    """The complete step -- 48 kHz in, encoder resampler, DTX encoder, decoder, decoder-side noise estimator on its own
    stream, output resampler -- at B = 4096, where every kernel is long enough to overlap its neighbours on the four
    library streams: 40 steps from ONE asynchronous lyra_hip_run_steps_dev call against the same calls issued one by one
    with a full synchronise after each, on a second context.  Packets of live hops, packet lengths, PCM at both rates and
    the is_noise flags of the last steps must be identical: any missing ordering edge shows as a difference here."""
    import torch
    import lyra_amd
    B, T, bits, ext = 4096, 40, 120, 48000
    base = _speech_noise_silence(golden_dir, T, 64)
    pcm48 = np.repeat(base[:, np.arange(B) % 64], 3, axis=2).copy()
    dev = torch.device("cuda", 0)
    ids = torch.from_numpy(np.random.default_rng(1).permutation(B).astype(np.int32)).to(dev)
    ring = torch.from_numpy(pcm48).to(dev)
    nb = lyra_amd.packet_size(bits)
    A, Bc = lyra_amd.LyraHip(max_streams=B), lyra_amd.LyraHip(max_streams=B)
    A.torch_order = Bc.torch_order = False
    A.set_encoder_sample_rate(ext); Bc.set_encoder_sample_rate(ext)

    def bufs():
        return dict(pk=[torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)],
                    nbytes=[torch.full((B,), -1, device=dev, dtype=torch.int32) for _ in range(2)],
                    out=[torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)],
                    ext=[torch.zeros((B, 960), device=dev, dtype=torch.int16) for _ in range(2)],
                    flag=torch.full((B,), -1, device=dev, dtype=torch.int32),
                    in16=torch.zeros((B, 320), device=dev, dtype=torch.int16))
    a, b = bufs(), bufs()
    torch.cuda.synchronize()
    try:
        A.run_steps_dev(ids, bits, T, first_step=0, d_pcm_ring=ring, d_packets=a["pk"], d_pcm_out=a["out"],
                        d_packet_bytes=a["nbytes"], d_is_noise=a["flag"], external_rate=ext, d_ext_out=a["ext"],
                        encode=True, decode=True, dtx=True, decoder_noise=True)
        for t in range(T):
            s = t & 1
            Bc.resample_dev(ids, ring[t], ext, 16000, b["in16"], side="encoder"); Bc.synchronize()
            Bc.encode_dtx_dev(ids, b["in16"], bits, b["pk"][s], b["nbytes"][s]); Bc.synchronize()
            Bc.decode_dev(ids, b["pk"][s], bits, b["out"][s]); Bc.synchronize()
            Bc.noise_receive_dev(ids, b["out"][s], b["flag"], side="decoder"); Bc.synchronize()
            Bc.resample_dev(ids, b["out"][s], 16000, ext, b["ext"][s], side="decoder"); Bc.synchronize()
        A.synchronize()
        for s in range(2):
            la, lb = a["nbytes"][s].cpu().numpy(), b["nbytes"][s].cpu().numpy()
            assert np.array_equal(la, lb)
            live = la > 0
            assert np.array_equal(a["pk"][s].cpu().numpy()[live], b["pk"][s].cpu().numpy()[live])
            assert torch.equal(a["out"][s], b["out"][s]) and torch.equal(a["ext"][s], b["ext"][s])
        assert torch.equal(a["flag"], b["flag"])
        assert int((a["nbytes"][1] == 0).sum()) > 100 and int((a["nbytes"][1] > 0).sum()) > 100
    finally:
        A.close()
        Bc.close()


def test_sliced_stage_launches_are_invisible(monkeypatch):
    """LYRA_HIP_TILE_DIV_<KERNEL>=k (api.hip tile_div) launches a stage kernel as k slices with a first-tile offset
    (kernel parameter tile0).  Uneven slice counts on a ragged batch (B = 1003: last tile partial, last slice short) must
    give the same packets and PCM as the plain launch, hop after hop."""
    import lyra_amd
    B, T, bits = 1003, 6, 184
    pcm = synth(B, T, seed=424242)
    plain = lyra_amd.LyraHip(max_streams=B)
    for name, k in (("ENC_S0", 3), ("ENC_S1", 2), ("ENC_S2", 5), ("DEC_S0", 2), ("DEC_S1", 3), ("DEC_S2", 7)):
        monkeypatch.setenv("LYRA_HIP_TILE_DIV_" + name, str(k))
    sliced = lyra_amd.LyraHip(max_streams=B)
    try:
        for t in range(T):
            pa, pb = plain.encode(pcm[t], bits), sliced.encode(pcm[t], bits)
            assert np.array_equal(pa, pb), f"packets differ at hop {t}"
            assert np.array_equal(plain.decode(pa, bits), sliced.decode(pb, bits)), f"PCM differs at hop {t}"
    finally:
        plain.close(); sliced.close()


@pytest.mark.parametrize("sub_batches", [1, 2])
def test_run_steps_48k_interleaved_with_per_call_resamplers(golden_dir, monkeypatch, sub_batches):
    """run_steps keeps the input resampler one step ahead on the quantizer stream and the output resampler on the noise
    stream; the per-call resample_dev entry points use the encode / decode streams and the SAME per-stream slots.  One
    context alternates the two without any synchronisation -- run_steps x4, one step by individual calls, run_steps x4,
    one step by individual calls, ... -- at a batch whose kernels are long enough to overlap; a second context does every
    step by individual calls with a full synchronise after each.  Packets and PCM at both rates must agree at every step
    that is still in a buffer at the end, and the resamplers' state must (two more steps by run_steps on both).
    With LYRA_HIP_SUBBATCHES=2 encode_dev / decode_dev split the batch over two stream pairs while the resamplers do not:
    an unsplit call must then join every stream of the split call before it (api.hip dec_side_begin / enc_cross_begin) --
    this case failed before that edge existed.
    The test is sensitive: a library built with -DLYRA_MUTATE_NO_AHEAD_WAIT (api.hip: the noise-stream -> decode-stream edge
    removed) fails it at the extra decoder-side hop, every time tried."""
    import torch
    import lyra_amd
    B, bits, ext, T = 2048, 184, 48000, 24
    monkeypatch.setenv("LYRA_HIP_SUBBATCHES", str(sub_batches))
    base = _speech_noise_silence(golden_dir, T, 64)
    pcm48 = np.repeat(base[:, np.arange(B) % 64], 3, axis=2).copy()
    dev = torch.device("cuda", 0)
    ids = torch.from_numpy(np.random.default_rng(3).permutation(B).astype(np.int32)).to(dev)
    ring = torch.from_numpy(pcm48).to(dev)
    nb = lyra_amd.packet_size(bits)
    A, Bc = lyra_amd.LyraHip(max_streams=B), lyra_amd.LyraHip(max_streams=B)
    A.torch_order = Bc.torch_order = False

    def bufs():
        return dict(pk=[torch.zeros((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)],
                    out=[torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)],
                    ext=[torch.zeros((B, 960), device=dev, dtype=torch.int16) for _ in range(2)],
                    in16=[torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)])
    a, b = bufs(), bufs()
    torch.cuda.synchronize()

    def per_call(ctx, bf, t, sync):
        s = t & 1
        ctx.resample_dev(ids, ring[t % T], ext, 16000, bf["in16"][s], side="encoder")
        if sync: ctx.synchronize()
        ctx.encode_dev(ids, bf["in16"][s], bits, bf["pk"][s])
        if sync: ctx.synchronize()
        ctx.decode_dev(ids, bf["pk"][s], bits, bf["out"][s])
        if sync: ctx.synchronize()
        ctx.resample_dev(ids, bf["out"][s], 16000, ext, bf["ext"][s], side="decoder")
        if sync: ctx.synchronize()

    def steps(ctx, bf, first, n):
        ctx.run_steps_dev(ids, bits, n, first_step=first, d_pcm_ring=ring, d_packets=bf["pk"], d_pcm_out=bf["out"],
                          external_rate=ext, d_ext_out=bf["ext"], encode=True, decode=True)
    try:
        t = 0
        extra_in = torch.from_numpy(base[0][np.arange(B) % 64].copy()).to(dev)
        xa, xb = torch.zeros((B, 960), device=dev, dtype=torch.int16), torch.zeros((B, 960), device=dev, dtype=torch.int16)
        for _ in range(4):
            steps(A, a, t, 4)
            # straight behind run_steps' last output resampler (noise stream): one more hop through the decoder-side
            # resampler by the per-call entry point (decode stream) -- same slots, nothing in between
            A.resample_dev(ids, extra_in, 16000, ext, xa, side="decoder")
            per_call(A, a, t + 4, sync=False)
            for u in range(t, t + 4):
                per_call(Bc, b, u, sync=True)
            Bc.resample_dev(ids, extra_in, 16000, ext, xb, side="decoder"); Bc.synchronize()
            per_call(Bc, b, t + 4, sync=True)
            A.synchronize()
            assert torch.equal(xa, xb), f"extra decoder-side hop differs after step {t + 4}"
            t += 5
        steps(A, a, t, 2)
        steps(Bc, b, t, 2)
        A.synchronize(); Bc.synchronize()
        for s in range(2):
            assert torch.equal(a["pk"][s], b["pk"][s]), f"packets differ (buffer {s})"
            assert torch.equal(a["out"][s], b["out"][s]), f"16 kHz PCM differs (buffer {s})"
            assert torch.equal(a["ext"][s], b["ext"][s]), f"48 kHz PCM differs (buffer {s})"
    finally:
        A.close(); Bc.close()
