"""The two-deep pipelined host-buffer calls (include/lyra_hip.h "Pipelined host-buffer calls"; BatchLyraEncoder::EncodeAsync /
WaitEncoded, BatchLyraDecoder::DecodeSamplesAsync / WaitDecoded) on the GPU: same kernels in the same order per stream, so
every packet and every decoded sample must equal what the blocking calls give -- over scripted sessions with packet loss,
DTX, odd request sizes and all four sample rates -- and the C ABI's begin / end pair must equal lyra_hip_encode."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("rate,bitrate,dtx", [(16000, 9200, False), (48000, 6000, True), (8000, 3200, False), (32000, 9200, True)])
def test_pipelined_twins_equal_the_blocking_twins(tmp_path, golden_dir, rate, bitrate, dtx):
    from test_batch_codec_semantics import _run_session
    from test_reference_glue import _session
    pcm, script = _session(golden_dir, rate, bitrate)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    pa, la, oa = _run_session(tmp_path / "a", None, rate, bitrate, dtx, pcm, script)
    pb, lb, ob = _run_session(tmp_path / "b", None, rate, bitrate, dtx, pcm, script, pipelined=True)
    assert np.array_equal(la, lb) and np.array_equal(pa, pb), "packets differ between the blocking and the pipelined session"
    assert oa.size == ob.size and np.array_equal(oa, ob), "decoded samples differ between the blocking and the pipelined session"


@pytest.mark.parametrize("sub_batches", [1, 2])
def test_c_abi_encode_begin_end_equals_encode_and_enforces_two_in_flight(sub_batches):
    """sub_batches = 2: the context splits a batch over two extractor / quantizer stream pairs (LYRA_HIP_SUBBATCHES); the
    pipelined call orders its upload in front of, and its download behind, every chunk."""
    import lyra_amd
    from lyra_amd import codec
    L = codec._load()
    vp = C.c_void_p
    B, T, bits = 300, 7, 184
    rng = np.random.Generator(np.random.PCG64(31))
    pcm = rng.integers(-20000, 20000, size=(T, B, 320)).astype(np.int16)
    ids = np.arange(B, dtype=np.int32)[::-1].copy()
    a = lyra_amd.LyraHip(max_streams=512)
    b = lyra_amd.LyraHip(max_streams=512, sub_batches=sub_batches)
    try:
        want = [a.encode(pcm[t], bits, ids) for t in range(T)]
        p = lambda x: x.ctypes.data_as(vp)
        h = b.h if hasattr(b, "h") else b.ctx
        got = []
        out = np.empty((B, 23), np.uint8)
        lens = np.empty(B, np.int32)
        assert L.lyra_hip_encode_end(h, p(out), p(lens)) != 0                     # nothing in flight
        assert L.lyra_hip_encode_begin(h, p(ids), B, p(pcm[0]), 16000, bits, 0) == 0
        for t in range(T):
            if t + 1 < T:
                assert L.lyra_hip_encode_begin(h, p(ids), B, p(pcm[t + 1]), 16000, bits, 0) == 0
                if t == 0:
                    assert L.lyra_hip_encode_begin(h, p(ids), B, p(pcm[t + 1]), 16000, bits, 0) != 0   # a third one is refused
            assert L.lyra_hip_encode_end(h, p(out), p(lens)) == 0
            assert (lens == 23).all()
            got.append(out.copy())
        for t in range(T):
            assert np.array_equal(got[t], want[t]), f"hop {t}"
        # ... and the Python mirror of the pair (lyra_amd/codec.py), at 48 kHz with DTX, against the blocking calls
        a.reset(); b.reset()
        a.set_encoder_sample_rate(48000)
        p48 = rng.integers(-20000, 20000, size=(4, B, 960)).astype(np.int16)
        p48[:, :7] //= 2000          # a few streams are noise: DTX sends empty packets for them
        b.encode_begin(p48[0], bits, ids, sample_rate_hz=48000, dtx=True)
        for t in range(4):
            if t + 1 < 4:
                b.encode_begin(p48[t + 1], bits, ids, sample_rate_hz=48000, dtx=True)
            pk, lens = b.encode_end()
            p16 = a.resample(p48[t], 48000, 16000, ids, side="encoder")
            want_pk, want_len = a.encode_dtx(p16, bits, ids)
            assert np.array_equal(lens, want_len) and np.array_equal(pk, want_pk), f"48 kHz DTX hop {t}"
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("sub_batches", [1, 2])
def test_c_abi_decode_begin_end_equals_decode(sub_batches):
    """lyra_hip_decode_begin / _end (through the Python mirror), two deep, against lyra_hip_decode on a second context: the
    same PCM hop for hop, state carried; a third begin is refused, an end without a begin too."""
    import lyra_amd
    B, T, bits = 700, 8, 120
    rng = np.random.Generator(np.random.PCG64(77))
    pcm = rng.integers(-20000, 20000, size=(T, B, 320)).astype(np.int16)
    ids = np.arange(B, dtype=np.int32)
    enc = lyra_amd.LyraHip(max_streams=1024)
    a = lyra_amd.LyraHip(max_streams=1024)
    b = lyra_amd.LyraHip(max_streams=1024, sub_batches=sub_batches)
    try:
        packets = [enc.encode(pcm[t], bits, ids) for t in range(T)]
        want = [a.decode(packets[t], bits, ids) for t in range(T)]
        with pytest.raises(Exception):
            b._pending_decodes = [B]
            b.decode_end()
        b._pending_decodes = []
        b.decode_begin(packets[0], bits, ids)
        for t in range(T):
            if t + 1 < T:
                b.decode_begin(packets[t + 1], bits, ids)
                if t == 0:
                    with pytest.raises(Exception):
                        b.decode_begin(packets[t + 1], bits, ids)     # a third call in flight is refused
            got = b.decode_end()
            assert np.array_equal(got, want[t]), f"hop {t}"
    finally:
        enc.close(); a.close(); b.close()


@pytest.mark.parametrize("sub_batches", [1, 2])
def test_pipelined_calls_with_stream_ids_that_change_every_hop(sub_batches):
    """Every hop lists the streams in another order (and the small id / packet buffers of a slot are rewritten while the
    previous call's kernels run): a call that read a slot's ids or packets from before its own upload would mix streams up.
    Encode and decode, two deep, against the blocking calls given the same orders."""
    import lyra_amd
    B, T, bits = 700, 8, 120
    rng = np.random.Generator(np.random.PCG64(5))
    pcm = rng.integers(-20000, 20000, size=(T, B, 320)).astype(np.int16)
    orders = [rng.permutation(B).astype(np.int32) for _ in range(T)]
    a = lyra_amd.LyraHip(max_streams=1024)
    b = lyra_amd.LyraHip(max_streams=1024, sub_batches=sub_batches)
    try:
        want_pk = [a.encode(pcm[t][orders[t]], bits, orders[t]) for t in range(T)]
        want_pcm = [a.decode(want_pk[t], bits, orders[t]) for t in range(T)]
        got_pk = []
        b.encode_begin(pcm[0][orders[0]], bits, orders[0])
        for t in range(T):
            if t + 1 < T:
                b.encode_begin(pcm[t + 1][orders[t + 1]], bits, orders[t + 1])
            got_pk.append(b.encode_end()[0])
        for t in range(T):
            assert np.array_equal(got_pk[t], want_pk[t]), f"packets of hop {t}"
        got_pcm = []
        b.decode_begin(want_pk[0], bits, orders[0])
        for t in range(T):
            if t + 1 < T:
                b.decode_begin(want_pk[t + 1], bits, orders[t + 1])
            got_pcm.append(b.decode_end())
        for t in range(T):
            assert np.array_equal(got_pcm[t], want_pcm[t]), f"PCM of hop {t}"
    finally:
        a.close(); b.close()
