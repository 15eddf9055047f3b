"""Differential test of the library's ordering rules (include/lyra_hip.h "Streams" (1)-(3)): random sequences of hops, each over
a random subset of the streams in a random order, issued through a randomly chosen entry point -- blocking host calls, `_dev`
calls on torch tensors, lyra_hip_run_steps_dev, the two-deep pipelined begin / end pairs -- and resets in between, on
contexts that split batches over 1, 2 or 4 stream sets.  A second context runs the same hops through the plain blocking
calls, unsplit; every packet and every sample must agree.  (Round 6: a split context let chunk 0 of a call overtake chunk 1 of
the previous call on streams that had moved between them; nothing in the suite changed the stream order between calls.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 512


@pytest.mark.parametrize("seed,sub_batches,bits,N", [(0, 1, 120, 512), (1, 2, 184, 512), (2, 2, 64, 512), (3, 4, 120, 512),
                                                      (4, 2, 120, 512), (5, 2, 120, 2048), (6, 4, 184, 2048)])
def test_random_call_sequences_equal_the_blocking_reference(seed, sub_batches, bits, N):
    # (N <= 1024: the contexts partition the chip between the two sides with CU masks; N = 2048: they do not)
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    ref = lyra_amd.LyraHip(max_streams=N)
    x = lyra_amd.LyraHip(max_streams=N, sub_batches=sub_batches)
    dev = torch.device("cuda", 0)
    nb = packet_size(bits)

    def subset():
        B = int(rng.choice([3, 40, 64, 129, 130, 257, 300, N - 1, N]))
        return rng.permutation(N)[:B].astype(np.int32)

    def audio(B, hops=1):
        return rng.integers(-15000, 15000, size=(hops, B, 320)).astype(np.int16)

    def ref_hop(order, pcm):
        pk = ref.encode(pcm, bits, order)
        return pk, ref.decode(pk, bits, order)

    counts = {}
    try:
        for op in range(70):
            kind = rng.choice(["host", "dev", "steps", "pipelined", "reset"], p=[0.2, 0.25, 0.2, 0.25, 0.1])
            counts[kind] = counts.get(kind, 0) + 1
            what = f"op {op} ({kind}, seed {seed})"
            if kind == "reset":
                ids = subset()
                ref.reset(ids); x.reset(ids)
            elif kind == "host":
                order = subset(); pcm = audio(order.size)[0]
                want_pk, want = ref_hop(order, pcm)
                pk = x.encode(pcm, bits, order)
                assert np.array_equal(pk, want_pk), what
                assert np.array_equal(x.decode(pk, bits, order), want), what
            elif kind == "dev":
                order = subset(); pcm = audio(order.size)[0]
                want_pk, want = ref_hop(order, pcm)
                d_ids = torch.from_numpy(order).to(dev)
                d_pcm = torch.from_numpy(pcm).to(dev)
                d_pk = torch.zeros((order.size, nb), dtype=torch.uint8, device=dev)
                d_out = torch.zeros((order.size, 320), dtype=torch.int16, device=dev)
                x.encode_dev(d_ids, d_pcm, bits, d_pk)
                x.decode_dev(d_ids, d_pk, bits, d_out)
                assert np.array_equal(d_pk.cpu().numpy(), want_pk), what
                assert np.array_equal(d_out.cpu().numpy(), want), what
            elif kind == "steps":
                order = subset(); n = int(rng.integers(1, 4)); pcm = audio(order.size, n)
                for t in range(n):
                    want_pk, want = ref_hop(order, pcm[t])
                d_ids = torch.from_numpy(order).to(dev)
                ring = torch.from_numpy(pcm).to(dev)
                d_pk = [torch.zeros((order.size, nb), dtype=torch.uint8, device=dev) for _ in range(2)]
                d_out = [torch.zeros((order.size, 320), dtype=torch.int16, device=dev) for _ in range(2)]
                x.run_steps_dev(d_ids, bits, n, first_step=0, d_pcm_ring=ring, d_packets=d_pk, d_pcm_out=d_out)
                last = (n - 1) & 1
                assert np.array_equal(d_pk[last].cpu().numpy(), want_pk), what
                assert np.array_equal(d_out[last].cpu().numpy(), want), what
            else:   # two hops in flight on each side, the second over another subset in another order
                oa, ob = subset(), subset()
                pa, pb = audio(oa.size)[0], audio(ob.size)[0]
                wa_pk = ref.encode(pa, bits, oa); wb_pk = ref.encode(pb, bits, ob)
                wa = ref.decode(wa_pk, bits, oa); wb = ref.decode(wb_pk, bits, ob)
                x.encode_begin(pa, bits, oa); x.encode_begin(pb, bits, ob)
                ga_pk = x.encode_end()[0]; gb_pk = x.encode_end()[0]
                assert np.array_equal(ga_pk, wa_pk) and np.array_equal(gb_pk, wb_pk), what
                x.decode_begin(ga_pk, bits, oa); x.decode_begin(gb_pk, bits, ob)
                ga = x.decode_end(); gb = x.decode_end()
                assert np.array_equal(ga, wa) and np.array_equal(gb, wb), what
        assert len(counts) == 5, counts
    finally:
        ref.close(); x.close()


@pytest.mark.parametrize("seed,sub_batches", [(0, 1), (1, 2), (2, 2), (3, 4)])
def test_random_sequences_around_the_codec_equal_the_blocking_reference(seed, sub_batches):
    """The same game with the stateful pieces around the two networks -- encoder-side resampler + DTX encoder (its noise
    estimator), decoder, decoder-side noise estimator, decoder-side resampler, comfort noise -- through host calls, `_dev`
    calls and the pipelined 48 kHz DTX encode, 48 kHz outside, quiet stretches so that DTX decisions flip."""
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    bits = 120
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    ref = lyra_amd.LyraHip(max_streams=N)
    x = lyra_amd.LyraHip(max_streams=N, sub_batches=sub_batches)
    for c in (ref, x):
        c.set_encoder_sample_rate(48000)
        c.set_cng_seed(7)
    dev = torch.device("cuda", 0)
    nb = packet_size(bits)
    quiet = rng.permutation(N)[:N // 5]          # a fifth of the streams only ever hear room noise

    def subset():
        B = int(rng.choice([7, 64, 129, 256, 300, 512]))
        return rng.permutation(N)[:B].astype(np.int32)

    def audio48(order):
        a = rng.integers(-15000, 15000, size=(order.size, 960)).astype(np.int16)
        a[np.isin(order, quiet)] //= 1500
        return a

    def ref_hop(order, a48):
        a16 = ref.resample(a48, 48000, 16000, order, side="encoder")
        pk, lens = ref.encode_dtx(a16, bits, order)
        got = lens > 0
        out = np.zeros((order.size, 320), np.int16)
        noise = np.zeros(order.size, np.int32)
        if got.any():      # the application hands over the packets that exist (lyra_decoder.cc: SetEncodedPacket, DecodeSamples)
            out[got] = ref.decode(pk[got], bits, order[got])
            noise[got] = ref.noise_receive(out[got], order[got], side="decoder")
        if (~got).any():   # ... and plays comfort noise from the decoder-side estimate for the others
            out[~got] = ref.comfort_noise(None, order[~got])
        return pk, lens, out, noise, ref.resample(out, 16000, 48000, order, side="decoder")

    try:
        flips = 0
        for op in range(50):
            kind = rng.choice(["host", "dev", "pipelined", "reset"], p=[0.3, 0.35, 0.25, 0.1])
            what = f"op {op} ({kind}, seed {seed})"
            if kind == "reset":
                ids = subset()
                ref.reset(ids); x.reset(ids)
                continue
            order = subset(); a48 = audio48(order)
            w_pk, w_len, w_out, w_noise, w_out48 = ref_hop(order, a48)
            flips += int((w_len == 0).any() and (w_len > 0).any())
            got = w_len > 0
            if kind == "host":
                a16 = x.resample(a48, 48000, 16000, order, side="encoder")
                pk, lens = x.encode_dtx(a16, bits, order)
            elif kind == "pipelined":
                x.encode_begin(a48, bits, order, sample_rate_hz=48000, dtx=True)
                pk, lens = x.encode_end()
            else:
                d_ids = torch.from_numpy(order).to(dev)
                d16 = torch.zeros((order.size, 320), dtype=torch.int16, device=dev)
                d_pk = torch.zeros((order.size, nb), dtype=torch.uint8, device=dev)
                d_len = torch.zeros(order.size, dtype=torch.int32, device=dev)
                x.resample_dev(d_ids, torch.from_numpy(a48).to(dev), 48000, 16000, d16, side="encoder")
                x.encode_dtx_dev(d_ids, d16, bits, d_pk, d_len)
                pk, lens = d_pk.cpu().numpy(), d_len.cpu().numpy()
            assert np.array_equal(lens, w_len), what
            assert np.array_equal(pk[got], w_pk[got]), what
            out = np.zeros((order.size, 320), np.int16)
            noise = np.zeros(order.size, np.int32)
            if kind == "dev" and got.any():
                g_ids = torch.from_numpy(order[got]).to(dev)
                g_out = torch.zeros((int(got.sum()), 320), dtype=torch.int16, device=dev)
                g_noise = torch.zeros(int(got.sum()), dtype=torch.int32, device=dev)
                x.decode_dev(g_ids, torch.from_numpy(pk[got]).to(dev), bits, g_out)
                x.noise_receive_dev(g_ids, g_out, g_noise, side="decoder")
                out[got], noise[got] = g_out.cpu().numpy(), g_noise.cpu().numpy()
            elif got.any():
                out[got] = x.decode(pk[got], bits, order[got])
                noise[got] = x.noise_receive(out[got], order[got], side="decoder")
            if (~got).any():
                if kind == "dev":
                    c_out = torch.zeros((int((~got).sum()), 320), dtype=torch.int16, device=dev)
                    x.comfort_noise_dev(torch.from_numpy(order[~got]).to(dev), None, c_out)
                    out[~got] = c_out.cpu().numpy()
                else:
                    out[~got] = x.comfort_noise(None, order[~got])
            assert np.array_equal(out, w_out), what
            assert np.array_equal(noise, w_noise), what
            if kind == "dev":
                d48 = torch.zeros((order.size, 960), dtype=torch.int16, device=dev)
                x.resample_dev(torch.from_numpy(order).to(dev), torch.from_numpy(out).to(dev), 16000, 48000, d48, side="decoder")
                out48 = d48.cpu().numpy()
            else:
                out48 = x.resample(out, 16000, 48000, order, side="decoder")
            assert np.array_equal(out48, w_out48), what
        assert flips > 5, "the session never mixed empty and full packets"
    finally:
        ref.close(); x.close()


@pytest.mark.parametrize("seed,sub_batches", [(0, 1), (1, 2), (2, 4), (3, 2)])
def test_run_steps_with_random_flags_equals_the_single_calls(seed, sub_batches):
    """lyra_hip_run_steps_dev with random combinations of DTX, decoder-side noise estimator and a 48 kHz outside, over random
    subsets, on contexts that split batches -- against the same hops issued one `_dev` call at a time on an unsplit context
    (lyra_hip.h: what one step is made of)."""
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    N, bits = 512, 120
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    ref = lyra_amd.LyraHip(max_streams=N)
    x = lyra_amd.LyraHip(max_streams=N, sub_batches=sub_batches)
    dev = torch.device("cuda", 0)
    nb = packet_size(bits)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    try:
        for call in range(24):
            B = int(rng.choice([9, 64, 130, 256, 300, 512]))
            order = rng.permutation(N)[:B].astype(np.int32)
            dtx, dnoise, rate = bool(rng.integers(2)), bool(rng.integers(2)), int(rng.choice([16000, 48000]))
            n = int(rng.integers(1, 5))
            what = f"call {call}: B {B}, {n} steps, dtx {dtx}, decoder noise {dnoise}, {rate} Hz (seed {seed})"
            n_ext = 320 * rate // 16000
            pcm = rng.integers(-15000, 15000, size=(n, B, n_ext)).astype(np.int16)
            pcm[:, : B // 4] //= 1500
            for c in (ref, x):
                c.set_encoder_sample_rate(rate)
            d_ids = torch.from_numpy(order).to(dev)
            ring = torch.from_numpy(pcm).to(dev)
            bufs = {}
            for name in ("ref", "x"):
                bufs[name] = dict(pk=[z((B, nb), torch.uint8) for _ in range(2)], out=[z((B, 320), torch.int16) for _ in range(2)],
                                  ln=[z((B,), torch.int32) for _ in range(2)], ext=[z((B, n_ext), torch.int16) for _ in range(2)],
                                  noise=z((B,), torch.int32))
            r = bufs["ref"]
            d16 = z((B, 320), torch.int16)
            for t in range(n):          # one step, call by call (api.hip lyra_hip_run_steps_dev)
                s = t & 1
                src = ring[t]
                if rate != 16000:
                    ref.resample_dev(d_ids, ring[t], rate, 16000, d16, side="encoder")
                    src = d16
                if dtx:
                    ref.encode_dtx_dev(d_ids, src, bits, r["pk"][s], r["ln"][s])
                else:
                    ref.encode_dev(d_ids, src, bits, r["pk"][s])
                ref.decode_dev(d_ids, r["pk"][s], bits, r["out"][s])
                if dnoise:
                    ref.noise_receive_dev(d_ids, r["out"][s], r["noise"], side="decoder")
                if rate != 16000:
                    ref.resample_dev(d_ids, r["out"][s], 16000, rate, r["ext"][s], side="decoder")
                ref.synchronize()       # (one buffer for d16: the next step's resampler must not overtake this step's encoder)
            b = bufs["x"]
            x.run_steps_dev(d_ids, bits, n, first_step=0, d_pcm_ring=ring, d_packets=b["pk"], d_pcm_out=b["out"],
                            d_packet_bytes=b["ln"] if dtx else None, d_is_noise=b["noise"] if dnoise else None,
                            external_rate=rate, d_ext_out=b["ext"] if rate != 16000 else None, dtx=dtx, decoder_noise=dnoise)
            x.synchronize()
            for s in range(min(n, 2)):
                assert torch.equal(b["pk"][s], r["pk"][s]), what + f": packets, set {s}"
                assert torch.equal(b["out"][s], r["out"][s]), what + f": PCM, set {s}"
                if dtx:
                    assert torch.equal(b["ln"][s], r["ln"][s]), what + f": packet lengths, set {s}"
                if rate != 16000:
                    assert torch.equal(b["ext"][s], r["ext"][s]), what + f": external-rate PCM, set {s}"
            if dnoise:
                assert torch.equal(b["noise"], r["noise"]), what + ": is_noise"
    finally:
        ref.close(); x.close()


@pytest.mark.parametrize("rate,sub_batches", [(48000, 1), (8000, 1), (48000, 2), (32000, 1)])
def test_run_steps_all_legs_over_many_hops_and_two_calls(rate, sub_batches):
    """Every optional leg at once (DTX encoder, decoder-side NoiseEstimator, both resamplers) over 11 + 6 hops in two
    lyra_hip_run_steps_dev calls -- long enough for the input resampler's ring of three (it runs two hops ahead) to wrap several
    times and for the noise stream's two-buffer rule to bite (the estimator and the output resampler of a hop are one
    noise-stream call) -- against the same hops issued one `_dev` call at a time with a synchronise after each."""
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    N, B, bits = 320, 300, 184
    rng = np.random.Generator(np.random.PCG64(4100 + rate // 1000 + sub_batches))
    ref = lyra_amd.LyraHip(max_streams=N)
    x = lyra_amd.LyraHip(max_streams=N, sub_batches=sub_batches)
    dev = torch.device("cuda", 0)
    nb = packet_size(bits)
    n_ext = 320 * rate // 16000
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    try:
        for c in (ref, x):
            c.set_encoder_sample_rate(rate)
        order = rng.permutation(N)[:B].astype(np.int32)
        d_ids = torch.from_numpy(order).to(dev)
        total = 17
        pcm = rng.integers(-15000, 15000, size=(total, B, n_ext)).astype(np.int16)
        pcm[:, : B // 3] //= 1500          # a third of the streams stay quiet: DTX and the estimators have something to decide
        ring = torch.from_numpy(pcm).to(dev)
        bufs = {}
        for name in ("ref", "x"):
            bufs[name] = dict(pk=[z((B, nb), torch.uint8) for _ in range(2)], out=[z((B, 320), torch.int16) for _ in range(2)],
                              ln=[z((B,), torch.int32) for _ in range(2)], ext=[z((B, n_ext), torch.int16) for _ in range(2)],
                              noise=z((B,), torch.int32))
        r, b = bufs["ref"], bufs["x"]
        d16 = z((B, 320), torch.int16)
        done = 0
        for n in (11, 6):
            for t in range(done, done + n):
                s = t & 1
                ref.resample_dev(d_ids, ring[t], rate, 16000, d16, side="encoder")
                ref.encode_dtx_dev(d_ids, d16, bits, r["pk"][s], r["ln"][s])
                ref.decode_dev(d_ids, r["pk"][s], bits, r["out"][s])
                ref.noise_receive_dev(d_ids, r["out"][s], r["noise"], side="decoder")
                ref.resample_dev(d_ids, r["out"][s], 16000, rate, r["ext"][s], side="decoder")
                ref.synchronize()
            x.run_steps_dev(d_ids, bits, n, first_step=done, d_pcm_ring=ring, d_packets=b["pk"], d_pcm_out=b["out"],
                            d_packet_bytes=b["ln"], d_is_noise=b["noise"], external_rate=rate, d_ext_out=b["ext"], dtx=True,
                            decoder_noise=True)
            x.synchronize()
            done += n
            what = f"{rate} Hz, split {sub_batches}, after {done} hops"
            for s in range(2):
                assert torch.equal(b["pk"][s], r["pk"][s]), what + f": packets, set {s}"
                assert torch.equal(b["ln"][s], r["ln"][s]), what + f": packet lengths, set {s}"
                assert torch.equal(b["out"][s], r["out"][s]), what + f": PCM, set {s}"
                assert torch.equal(b["ext"][s], r["ext"][s]), what + f": external-rate PCM, set {s}"
            assert torch.equal(b["noise"], r["noise"]), what + ": is_noise"
        lens = b["ln"][0].cpu().numpy()
        assert (lens == 0).any() and (lens > 0).any(), "the session never mixed empty and full packets"
    finally:
        ref.close(); x.close()


@pytest.mark.parametrize("rate,sub_batches,dtx,noise", [(48000, 1, True, True), (8000, 1, False, True), (32000, 2, True, False),
                                                        (48000, 2, False, True), (16000, 1, True, True), (48000, 4, True, True)])
def test_one_call_per_side_at_an_external_rate_equals_the_single_calls(rate, sub_batches, dtx, noise):
    """lyra_hip_encode_ext_dev / lyra_hip_decode_ext_dev (one hop at an external rate, ONE call per side -- resampler + [DTX] +
    encode; decode + [estimator] + resampler) hop after hop WITHOUT any synchronisation between hops, two alternating buffer
    sets, over subsets that change -- against resample / encode[_dtx] / decode / noise_receive / resample issued one by one
    with a synchronise after every hop on an unsplit context."""
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    N, bits, hops = 384, 120, 14
    rng = np.random.Generator(np.random.PCG64(5200 + rate // 1000 + 7 * sub_batches + dtx + 2 * noise))
    ref = lyra_amd.LyraHip(max_streams=N)
    x = lyra_amd.LyraHip(max_streams=N, sub_batches=sub_batches)
    dev = torch.device("cuda", 0)
    nb = packet_size(bits)
    n_ext = 320 * rate // 16000
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    try:
        for c in (ref, x):
            c.set_encoder_sample_rate(rate)
        B = 300
        order = rng.permutation(N)[:B].astype(np.int32)
        d_ids = torch.from_numpy(order).to(dev)
        pcm = rng.integers(-15000, 15000, size=(hops, B, n_ext)).astype(np.int16)
        pcm[:, : B // 3] //= 1500
        ring = torch.from_numpy(pcm).to(dev)
        mk = lambda: dict(pk=[z((B, nb), torch.uint8) for _ in range(2)], out=[z((B, 320), torch.int16) for _ in range(2)],
                          ln=[z((B,), torch.int32) for _ in range(2)], ext=[z((B, n_ext), torch.int16) for _ in range(2)],
                          noise=[z((B,), torch.int32) for _ in range(2)])
        r, b = mk(), mk()
        d16 = z((B, 320), torch.int16)
        keep = {}
        for t in range(hops):
            s = t & 1
            src = ring[t]
            if rate != 16000:
                ref.resample_dev(d_ids, ring[t], rate, 16000, d16, side="encoder")
                src = d16
            if dtx:
                ref.encode_dtx_dev(d_ids, src, bits, r["pk"][s], r["ln"][s])
            else:
                ref.encode_dev(d_ids, src, bits, r["pk"][s])
            ref.decode_dev(d_ids, r["pk"][s], bits, r["out"][s])
            if noise:
                ref.noise_receive_dev(d_ids, r["out"][s], r["noise"][s], side="decoder")
            if rate != 16000:
                ref.resample_dev(d_ids, r["out"][s], 16000, rate, r["ext"][s], side="decoder")
            ref.synchronize()
            keep[t] = {k: r[k][s].clone() for k in r}
            # the path under test: no synchronisation between hops (only every fifth hop, to compare what the two-buffer
            # rule still holds: sets t and t - 1)
            x.encode_ext_dev(d_ids, ring[t], rate, bits, b["pk"][s], b["ln"][s] if dtx else None, dtx=dtx)
            x.decode_ext_dev(d_ids, b["pk"][s], bits, rate, b["out"][s], b["ext"][s] if rate != 16000 else None,
                             b["noise"][s] if noise else None)
            if t % 5 == 4 or t == hops - 1:
                x.synchronize()
                for tt in (t - 1, t):
                    ss, what = tt & 1, f"{rate} Hz, split {sub_batches}, dtx {dtx}, estimator {noise}: hop {tt}"
                    assert torch.equal(b["pk"][ss], keep[tt]["pk"]), what + ": packets"
                    assert torch.equal(b["out"][ss], keep[tt]["out"]), what + ": 16 kHz PCM"
                    if dtx:
                        assert torch.equal(b["ln"][ss], keep[tt]["ln"]), what + ": packet lengths"
                    if noise:
                        assert torch.equal(b["noise"][ss], keep[tt]["noise"]), what + ": is_noise"
                    if rate != 16000:
                        assert torch.equal(b["ext"][ss], keep[tt]["ext"]), what + ": external-rate PCM"
    finally:
        ref.close(); x.close()


def test_one_call_per_side_argument_checks():
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size, LyraHipError
    c = lyra_amd.LyraHip(max_streams=8)
    dev = torch.device("cuda", 0)
    try:
        ids = torch.arange(4, dtype=torch.int32, device=dev)
        pk = torch.zeros((4, packet_size(64)), dtype=torch.uint8, device=dev)
        out16 = torch.zeros((4, 320), dtype=torch.int16, device=dev)
        x48 = torch.zeros((4, 960), dtype=torch.int16, device=dev)
        ln = torch.zeros((4,), dtype=torch.int32, device=dev)
        with pytest.raises(LyraHipError):      # DTX at 48 kHz while the encoder-side estimator is set up for 16 kHz
            c.encode_ext_dev(ids, x48, 48000, 64, pk, ln, dtx=True)
        with pytest.raises(LyraHipError):      # an external rate needs the external-rate output buffer
            c.decode_ext_dev(ids, pk, 64, 48000, out16, None, None)
        x44 = torch.zeros((4, 882), dtype=torch.int16, device=dev)
        with pytest.raises(LyraHipError):      # 44.1 kHz is not a rate of the codec
            c.encode_ext_dev(ids, x44, 44100, 64, pk)
        c.set_encoder_sample_rate(48000)
        c.encode_ext_dev(ids, x48, 48000, 64, pk, ln, dtx=True)
        c.decode_ext_dev(ids, pk, 64, 48000, out16, x48.clone(), ln.clone())
        c.synchronize()
    finally:
        c.close()


def test_one_call_per_side_equals_run_steps_at_benchmark_scale():
    """4096 streams, 48 kHz, DTX encoder and decoder-side estimator, 60 hops: lyra_hip_encode_ext_dev / lyra_hip_decode_ext_dev hop
    after hop without synchronising against ONE lyra_hip_run_steps_dev call with the same flags (itself held against the single
    calls above) -- every buffer of the last two hops and the estimator flags must be equal."""
    import torch
    import lyra_amd
    from lyra_amd.codec import packet_size
    B, bits, hops, rate = 4096, 184, 60, 48000
    rng = np.random.Generator(np.random.PCG64(6001))
    a = lyra_amd.LyraHip(max_streams=B)
    x = lyra_amd.LyraHip(max_streams=B)
    dev = torch.device("cuda", 0)
    nb, n_ext = packet_size(bits), 320 * rate // 16000
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    try:
        for c in (a, x):
            c.set_encoder_sample_rate(rate)
        d_ids = torch.arange(B, dtype=torch.int32, device=dev)
        pcm = rng.integers(-15000, 15000, size=(8, B, n_ext)).astype(np.int16)     # a ring of eight input hops
        pcm[:, : B // 3] //= 1500
        ring = torch.from_numpy(pcm).to(dev)
        mk = lambda: dict(pk=[z((B, nb), torch.uint8) for _ in range(2)], out=[z((B, 320), torch.int16) for _ in range(2)],
                          ln=[z((B,), torch.int32) for _ in range(2)], ext=[z((B, n_ext), torch.int16) for _ in range(2)],
                          noise=z((B,), torch.int32))
        r, b = mk(), mk()
        a.run_steps_dev(d_ids, bits, hops, first_step=0, d_pcm_ring=ring, d_packets=r["pk"], d_pcm_out=r["out"],
                        d_packet_bytes=r["ln"], d_is_noise=r["noise"], external_rate=rate, d_ext_out=r["ext"], dtx=True,
                        decoder_noise=True)
        for t in range(hops):
            s = t & 1
            x.encode_ext_dev(d_ids, ring[t % 8], rate, bits, b["pk"][s], b["ln"][s], dtx=True)
            x.decode_ext_dev(d_ids, b["pk"][s], bits, rate, b["out"][s], b["ext"][s], b["noise"])
        a.synchronize(); x.synchronize()
        for s in range(2):
            for k in ("pk", "ln", "out", "ext"):
                assert torch.equal(b[k][s], r[k][s]), f"{k}, set {s}"
        assert torch.equal(b["noise"], r["noise"]), "is_noise"
        lens = b["ln"][0].cpu().numpy()
        assert (lens == 0).any() and (lens > 0).any(), "the run never mixed empty and full packets"
    finally:
        a.close(); x.close()
