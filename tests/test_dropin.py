"""The LITERAL drop-in, built and run: oracle/_ref/liblyra_ref_hip.so holds the reference's own
lyra_encoder.cc / lyra_decoder.cc / lyra_benchmark_lib.cc / cli_example/*_main_lib.cc compiled from the reference tree where
they lie, with CreateFeatureExtractor / CreateQuantizer / CreateGenerativeModel (lyra_components.cc:42-55) supplied by the
product's lyra_amd/host/lyra_hip_components.cc over liblyra_hip.so (oracle/ref_dropin.cc, oracle/Makefile).

  * CPU: the library links, exports its API, and the reference's LyraEncoder::Create -> Encode, LyraDecoder::Create ->
    SetEncodedPacket -> DecodeSamples and lyra_benchmark run through the HIP plugin classes against the fake C ABI of
    tests/host_stub/ (host logic only; no arithmetic is checked there).
  * GPU: the reference's LyraEncoder::Encode -> LyraDecoder::DecodeSamples over sample1_16kHz at 3200 / 6000 / 9200 bps --
    packets and PCM array_equal with the oracle-backed build of the SAME reference sources (oracle/_ref/liblyra_ref.so),
    per-hop LSD < 2 (lyra_integration_test.cc:131-142); the reference's own lyra_benchmark loop (lyra_benchmark_lib.cc:
    199-293) with its timing table; EncodeFile / DecodeFile (what encoder_main / decoder_main call).
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from oracle import lyra_dropin as D

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.skipif(not D.available(), reason="oracle/_ref/liblyra_ref_hip.so not built (needs /root/reference)")

BITRATES = {3200: 64, 6000: 120, 9200: 184}


def test_dropin_exports_and_runs_on_the_fake_abi(tmp_path):
    """No GPU here: liblyra_hip.so is replaced (LD_LIBRARY_PATH; the drop-in carries a RUNPATH) by the CPU stand-in of
    tests/host_stub/fake_lyra_hip.cc.  What runs is the reference's control flow -- the asset probe, the factories, one
    plugin call per hop and kind, the packet round trip, the benchmark loop -- through lyra_hip_components.cc."""
    fake = tmp_path / "fake"
    fake.mkdir()
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", "-pthread", "-o", str(fake / "liblyra_hip.so"),
                           os.path.join(ROOT, "tests", "host_stub", "fake_lyra_hip.cc")])
    script = textwrap.dedent("""
        import ctypes, sys, numpy as np
        sys.path.insert(0, %r)
        from oracle import lyra_oracle, lyra_dropin as D
        o = lyra_oracle.Oracle(mode="xnnpack")
        md = D.make_model_dir(%r)
        L = D.load(o)
        for name in ("dropin_encoder_new", "dropin_encoder_encode", "dropin_decoder_new", "dropin_decoder_set_packet",
                     "dropin_decoder_decode", "dropin_lyra_benchmark", "dropin_encode_file", "dropin_decode_file"):
            assert hasattr(L, name), name
        assert D.LyraEncoder.__init__ and L.dropin_encoder_new(16000, 1, 6000, 0, b"/nonexistent") is None   # asset probe
        enc = D.LyraEncoder(o, 16000, 6000, False, md)
        dec = D.LyraDecoder(o, 16000, md)
        rng = np.random.default_rng(3)
        for t in range(5):
            pk = enc.Encode(rng.integers(-3000, 3000, 320).astype(np.int16))
            assert pk is not None and pk.size == 15
            assert dec.SetEncodedPacket(pk)
            out = dec.DecodeSamples(320)
            assert out is not None and out.size == 320
        assert enc.Encode(np.zeros(100, np.int16)) is None            # lyra_encoder.cc:116-122: wrong hop size
        assert enc.set_bitrate(9200) and enc.Encode(np.zeros(320, np.int16)).size == 23
        calls, device_calls = D.call_stats(o)
        assert calls >= 5 * 4 and device_calls >= 5 * 4, (calls, device_calls)   # Extract, Quantize, Dequantize, Generate
        assert D.lyra_benchmark(o, 10, md) == 0
        assert D.lyra_benchmark(o, 0, md) == -1                        # lyra_benchmark_lib.cc:204-207
        print("ok")
    """) % (ROOT, str(tmp_path / "model"))
    env = dict(os.environ, LD_LIBRARY_PATH=str(fake) + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FAKE_CALL_US="0", LYRA_HIP_NO_TORCH_PRELOAD="1")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    assert "feature_extractor:" in r.stderr and "total:" in r.stderr    # the reference's own table (BENCHMARK build)


# ---- GPU -------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    return D.make_model_dir(tmp_path_factory.mktemp("dropin_model"))


def _hops(golden_dir):
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    n = pcm.size // 320
    return pcm[:n * 320].reshape(n, 320)


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate", sorted(BITRATES))
def test_gpu_reference_classes_on_hip_plugins_vs_oracle_backed_build(oracle_default, golden_dir, model_dir, bitrate):
    """lyra_integration_test.cc:49-149 with the HIP factories: the reference's LyraEncoder / LyraDecoder, created by their
    public Create(), over the first 150 hops of sample1_16kHz.wav."""
    from oracle import lyra_ref as R
    from test_whole_file_known_answers import lsd_per_hop
    assert R.available()
    hops = _hops(golden_dir)[:150]
    enc = D.LyraEncoder(oracle_default, 16000, bitrate, False, model_dir)
    dec = D.LyraDecoder(oracle_default, 16000, model_dir)
    renc = R.LyraEncoder(oracle_default, 16000, BITRATES[bitrate], False)
    rdec = R.LyraDecoder(oracle_default, 16000, 0)
    outs = []
    for t, hop in enumerate(hops):
        pk, want_pk = enc.Encode(hop), renc.Encode(hop)
        assert pk is not None and pk.size == BITRATES[bitrate] // 8 and np.array_equal(pk, want_pk), (bitrate, t)
        assert dec.SetEncodedPacket(pk) and rdec.SetEncodedPacket(want_pk)
        out, want = dec.DecodeSamples(320), rdec.DecodeSamples(320)
        assert out is not None and np.array_equal(out, want), (bitrate, t)
        outs.append(out)
    lsd = lsd_per_hop(hops, np.stack(outs))
    assert lsd.max() < 2.0, (bitrate, lsd.max())
    enc.close(); dec.close()


@pytest.mark.gpu
def test_gpu_reference_decoder_concealment_on_hip_plugins(oracle_default, golden_dir, model_dir):
    """Packets withheld for a stretch and odd request sizes: LyraDecoder::DecodeSamples' concealment, comfort-noise and
    fade branches (lyra_decoder.cc:228-373) drive the HIP generative model exactly as they drive the oracle-backed one."""
    from oracle import lyra_ref as R
    hops = _hops(golden_dir)[20:80]
    enc = D.LyraEncoder(oracle_default, 16000, 6000, False, model_dir)
    dec = D.LyraDecoder(oracle_default, 16000, model_dir, cng_seed=11)
    renc = R.LyraEncoder(oracle_default, 16000, 120, False)
    rdec = R.LyraDecoder(oracle_default, 16000, 11)
    sizes = [320, 100, 220, 320, 7, 313]
    saw_cng = False
    for t, hop in enumerate(hops):
        pk, want_pk = enc.Encode(hop), renc.Encode(hop)
        assert np.array_equal(pk, want_pk)
        if not 15 <= t < 35:
            assert dec.SetEncodedPacket(pk) and rdec.SetEncodedPacket(want_pk)
        for k in ([320] if t % 3 else sizes[(t // 3 % 3) * 2:(t // 3 % 3) * 2 + 2]):
            got, want = dec.DecodeSamples(k), rdec.DecodeSamples(k)
            assert got is not None and want is not None and np.array_equal(got, want), (t, k)
        saw_cng = saw_cng or dec.is_comfort_noise()
        assert dec.is_comfort_noise() == rdec.is_comfort_noise()
    assert saw_cng


@pytest.mark.gpu
@pytest.mark.parametrize("rate,bitrate,dtx", [(8000, 3200, False), (32000, 6000, True), (48000, 9200, False)])
def test_gpu_reference_classes_on_hip_plugins_other_rates_and_dtx(oracle_default, golden_dir, model_dir, rate, bitrate, dtx):
    """The sample rates and the DTX mode of lyra_integration_test.cc:49-149 / lyra_encoder.cc:119-141: the reference's own
    resampling, noise decision and empty-packet logic around the HIP plugins (four scripted streams with loss, a silent
    stretch and odd request sizes -- tests/test_reference_glue.py _session), against the oracle-backed build of the same
    sources, packet for packet and sample for sample."""
    from oracle import lyra_ref as R
    from test_reference_glue import _session
    pcm, script = _session(golden_dir, rate, bitrate)
    n = pcm.shape[1]
    D.load(oracle_default).dropin_set_max_streams(64)
    encs = [D.LyraEncoder(oracle_default, rate, bitrate, dtx, model_dir) for _ in range(n)]
    decs = [D.LyraDecoder(oracle_default, rate, model_dir, cng_seed=0x4C797261 ^ s) for s in range(n)]
    rencs = [R.LyraEncoder(oracle_default, rate, BITRATES[bitrate], dtx) for _ in range(n)]
    rdecs = [R.LyraDecoder(oracle_default, rate, 0x4C797261 ^ s) for s in range(n)]
    saw_empty = False
    for t, (mask, sizes) in enumerate(script):
        for s in range(n):
            pk, want = encs[s].Encode(pcm[t, s]), rencs[s].Encode(pcm[t, s])
            assert pk is not None and want is not None and np.array_equal(pk, want), (t, s)
            saw_empty = saw_empty or pk.size == 0
            if pk.size and mask[s] == "1":
                assert decs[s].SetEncodedPacket(pk) and rdecs[s].SetEncodedPacket(want)
        for k in sizes:
            for s in range(n):
                got, want = decs[s].DecodeSamples(k), rdecs[s].DecodeSamples(k)
                assert got is not None and want is not None and np.array_equal(got, want), (t, s, k)
    assert saw_empty == dtx
    for o in encs + decs:
        o.close()


@pytest.mark.gpu
def test_gpu_reference_lyra_benchmark_on_hip_plugins(oracle_default, model_dir, capfd):
    """The reference's own benchmark loop and report (lyra_benchmark_lib.cc:199-293, compiled with -DBENCHMARK)."""
    assert D.lyra_benchmark(oracle_default, 500, model_dir) == 0
    err = capfd.readouterr().err
    for title in ("feature_extractor", "quantizer_quantize", "quantizer_decode", "model_decode", "total"):
        assert title + ":" in err, err
        assert os.path.exists("/tmp/benchmarks/%s.csv" % title)
    total = np.loadtxt("/tmp/benchmarks/total.csv", skiprows=1)
    assert total.size == 500 and 0 < np.median(total) < 5000          # microseconds per hop, one stream
    sys.stderr.write(err)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "dropin_lyra_benchmark.txt"), "w") as f:
            f.write(err)


@pytest.mark.gpu
def test_gpu_reference_file_codec_on_hip_plugins(oracle_default, golden_dir, model_dir, tmp_path):
    """EncodeFile / DecodeFile (cli_example/encoder_main_lib.cc:99-140, decoder_main_lib.cc:142-222 -- what encoder_main /
    decoder_main call; BASELINE config #1) on the HIP plugins vs the oracle-backed build: .lyra bytes and decoded wav."""
    from oracle import lyra_ref as R
    from test_reference_glue import _read_wav, _write_wav
    pcm = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    _write_wav(tmp_path / "in.wav", pcm)
    ref_dir = R.make_model_dir(tmp_path / "ref_model")
    assert D.encode_file(oracle_default, tmp_path / "in.wav", tmp_path / "hip.lyra", 3200, model_dir)
    assert R.encode_file(oracle_default, tmp_path / "in.wav", tmp_path / "ref.lyra", 3200, ref_dir)
    got = np.fromfile(tmp_path / "hip.lyra", np.uint8)
    assert got.size == 1376 and np.array_equal(got, np.fromfile(tmp_path / "ref.lyra", np.uint8))
    assert D.decode_file(oracle_default, tmp_path / "hip.lyra", tmp_path / "hip.wav", 16000, 3200, model_dir)
    assert R.decode_file(oracle_default, tmp_path / "ref.lyra", tmp_path / "ref.wav", 16000, 3200, ref_dir)
    a, ra = _read_wav(tmp_path / "hip.wav")
    b, rb = _read_wav(tmp_path / "ref.wav")
    assert ra == rb == 16000 and a.size == 172 * 320 and np.array_equal(a, b)
