"""The C++ host side (lyra_amd/host): FeatureExtractorInterface / VectorQuantizerInterface / GenerativeModel
implementations over the C ABI, driven the way LyraEncoder::Encode and LyraDecoder drive the reference's plugins
(lyra/lyra_encoder.cc:143-155, lyra/lyra_decoder.cc:198-207,317-326).  Bit strings and PCM must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [64, 120, 184])
def test_cpp_plugin_surface(oracle_default, golden_dir, tmp_path, bits):
    import lyra_amd
    from oracle import lyra_oracle
    demo = os.path.join(ROOT, "lyra_amd", "plugin_demo")
    assert os.path.exists(demo), "lyra_amd/plugin_demo not built (__graft_entry__.build())"
    g = np.load(os.path.join(golden_dir, "speech_sample1.npz"))
    pcm = g["pcm_in"][:12]
    pin, bits_out, pout = tmp_path / "in.s16", tmp_path / "bits.txt", tmp_path / "out.s16"
    pcm.tofile(pin)
    r = subprocess.run([demo, lyra_amd.default_model_dir(), str(pin), str(bits), str(bits_out), str(pout)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    ref = lyra_oracle.run_batch(oracle_default, pcm[:, None, :], bits // 4, do_decode=True)
    lines = open(bits_out).read().split()
    assert len(lines) == 12 and all(len(l) == bits for l in lines)
    for f, l in enumerate(lines):
        want = "".join(format(int(b), "08b") for b in ref["packets"][f, 0])[:bits]
        assert l == want, f"bit string differs at hop {f}"
    out = np.fromfile(pout, np.int16).reshape(12, 320)
    assert np.array_equal(out, ref["pcm"][:, 0])


@pytest.mark.gpu
def test_cpp_plugins_many_threads_combined_calls(oracle_default, golden_dir, tmp_path):
    """48 codec objects (extractor + quantizer + generative model each) on 48 threads, every one used hop by hop as the
    reference uses its plugins; the plugin layer combines calls that wait at the same time into batched device calls
    (lyra_hip_components.cc "Call combining").  Bit strings and PCM of every stream must equal the oracle's, and
    combining must actually have happened."""
    import lyra_amd
    from oracle import lyra_oracle
    demo = os.path.join(ROOT, "lyra_amd", "plugin_mt_demo")
    assert os.path.exists(demo), "lyra_amd/plugin_mt_demo not built (__graft_entry__.build())"
    n, T, bits = 48, 10, 120
    rng = np.random.Generator(np.random.PCG64(4242))
    speech = np.load(os.path.join(golden_dir, "speech_sample1.npz"))["pcm_in"]      # [50][320]
    pcm = rng.integers(-20000, 20000, size=(T, n, 320)).astype(np.int16)
    pcm[:, 0] = speech[:T]
    pcm[:, 1] = speech[20:20 + T]
    pin, bits_out, pout = tmp_path / "in.s16", tmp_path / "bits.txt", tmp_path / "out.s16"
    pcm.tofile(pin)
    r = subprocess.run([demo, lyra_amd.default_model_dir(), str(pin), str(n), str(bits), str(bits_out), str(pout)],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    ref = lyra_oracle.run_batch(oracle_default, pcm, bits // 4, do_decode=True, threads=8)
    lines = open(bits_out).read().split()
    assert len(lines) == T * n
    for f in range(T):
        for s_ in range(n):
            want = "".join(format(int(b), "08b") for b in ref["packets"][f, s_])[:bits]
            assert lines[f * n + s_] == want, f"bit string differs at hop {f}, stream {s_}"
    out = np.fromfile(pout, np.int16).reshape(T, n, 320)
    assert np.array_equal(out, ref["pcm"])
    first = [l for l in r.stdout.splitlines() if l.startswith("plugin_calls")][0].split()[:6]   # "plugin_calls N device_calls M largest_batch K (...)"
    stats = dict(zip(first[0::2], map(int, first[1::2])))
    assert stats["plugin_calls"] == 4 * T * n
    assert stats["device_calls"] < stats["plugin_calls"] and stats["largest_batch"] >= 2, stats


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate", [3200, 9200])
def test_cpp_batch_codec_twins(oracle_default, golden_dir, tmp_path, bitrate):
    """BatchLyraEncoder / BatchLyraDecoder: the public-API twins for many streams.  Packets and
    PCM of every stream must equal the oracle's; the demo also checks the reference's argument validation."""
    import lyra_amd
    from oracle import lyra_oracle
    demo = os.path.join(ROOT, "lyra_amd", "batch_demo")
    assert os.path.exists(demo), "lyra_amd/batch_demo not built (__graft_entry__.build())"
    g = np.load(os.path.join(golden_dir, "noise_4x6.npz"))
    speech = np.load(os.path.join(golden_dir, "speech_sample1.npz"))["pcm_in"][:6]
    pcm = np.concatenate([g["pcm_in"], speech[:, None, :]], axis=1).astype(np.int16)   # [6 frames][5 streams][320]
    T, n = pcm.shape[0], pcm.shape[1]
    pin, pk, pout = tmp_path / "in.s16", tmp_path / "pk.bin", tmp_path / "out.s16"
    pcm.tofile(pin)
    r = subprocess.run([demo, lyra_amd.default_model_dir(), str(pin), str(n), str(bitrate), str(pk), str(pout)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    bits = {3200: 64, 6000: 120, 9200: 184}[bitrate]
    ref = lyra_oracle.run_batch(oracle_default, pcm, bits // 4, do_decode=True)
    packets = np.fromfile(pk, np.uint8).reshape(T, n, -1)
    assert np.array_equal(packets, ref["packets"])
    out = np.fromfile(pout, np.int16).reshape(T, n, 320)
    assert np.array_equal(out, ref["pcm"])


def _write_wav(path, pcm, rate=16000):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.asarray(pcm, np.int16).tobytes())


def _read_wav(path):
    import wave
    with wave.open(str(path), "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000
        return np.frombuffer(w.readframes(w.getnframes()), np.int16)


@pytest.mark.gpu
def test_cpp_file_transcode_ragged_batch(oracle_default, golden_dir, tmp_path):
    """EncodeFiles / DecodeFiles: several WAV files of different lengths transcoded together
    (streams leave the batch as they run out of full hops; a trailing partial hop is dropped as
    encoder_main_lib.cc:71-73 does).  .lyra bytes and decoded samples must equal the oracle's per file."""
    import lyra_amd
    from oracle import lyra_oracle
    demo = os.path.join(ROOT, "lyra_amd", "file_demo")
    assert os.path.exists(demo), "lyra_amd/file_demo not built (__graft_entry__.build())"
    speech = np.load(os.path.join(golden_dir, "speech_sample1.npz"))["pcm_in"].reshape(-1)   # 50 hops
    noise = np.load(os.path.join(golden_dir, "noise_4x6.npz"))["pcm_in"]                      # [6][4][320]
    files = {
        "speech_long": speech[:320 * 17 + 111],            # 17 full hops + a partial one
        "speech_short": speech[320 * 20:320 * 25],          # 5 hops
        "noise_a": noise[:, 0].reshape(-1),                 # 6 hops
        "noise_b": np.concatenate([noise[:, 1].reshape(-1), noise[:3, 2].reshape(-1)])[:-7],  # 8 hops + partial
        "tiny": speech[:100],                               # no full hop at all -> empty .lyra
    }
    wavs = []
    for name, pcm in files.items():
        _write_wav(tmp_path / f"{name}.wav", pcm)
        wavs.append(str(tmp_path / f"{name}.wav"))
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    r = subprocess.run([demo, lyra_amd.default_model_dir(), "6000", str(out_dir)] + wavs,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    for name, pcm in files.items():
        hops = len(pcm) // 320
        enc = np.fromfile(out_dir / f"{name}.lyra", np.uint8)
        dec = _read_wav(out_dir / f"{name}_decoded.wav")
        assert enc.size == hops * 15 and dec.size == hops * 320
        if hops == 0:
            continue
        ref = lyra_oracle.run_batch(oracle_default, pcm[:hops * 320].reshape(hops, 1, 320), 120 // 4, do_decode=True)
        assert np.array_equal(enc.reshape(hops, 15), ref["packets"][:, 0]), name
        assert np.array_equal(dec.reshape(hops, 320), ref["pcm"][:, 0]), name


def test_cpp_wav_io_roundtrip(tmp_path):
    """CPU side: the RIFF/WAVE PCM16 reader/writer of lyra_file_codec (the reference uses audio_dsp's wav_util):
    reads what Python's wave module writes (also with an odd-sized extra chunk before `data`), writes what it reads."""
    import struct
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "lyra_amd", "host")], stdout=subprocess.DEVNULL)
    demo = os.path.join(ROOT, "lyra_amd", "file_demo")
    pcm = (np.arange(1000) * 37 % 65536 - 32768).astype(np.int16)
    a, b, c = tmp_path / "a.wav", tmp_path / "b.wav", tmp_path / "c.wav"
    _write_wav(a, pcm)
    raw = open(a, "rb").read()
    # splice a LIST chunk of odd length (padded to even) between fmt and data
    extra = b"LIST" + struct.pack("<I", 5) + b"hello" + b"\0"
    spliced = raw[:36] + extra + raw[36:]
    spliced = spliced[:4] + struct.pack("<I", len(spliced) - 8) + spliced[8:]
    open(b, "wb").write(spliced)
    for src in (a, b):
        r = subprocess.run([demo, "--selftest-wav", str(src), str(c)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr)
        assert r.stdout.split() == ["1", "16000", "1000"]
        assert np.array_equal(_read_wav(c), pcm)
    open(b, "wb").write(raw[:20] + struct.pack("<H", 3) + raw[22:])   # format tag 3 (float): must be rejected
    assert subprocess.run([demo, "--selftest-wav", str(b), str(c)], capture_output=True).returncode == 6


def test_cpp_plugins_build_and_link():
    """CPU side: the adapters compile against the C ABI and export the lyra_components factory names."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "lyra_amd", "host")], stdout=subprocess.DEVNULL)
    out = subprocess.run(["nm", "-DC", os.path.join(ROOT, "lyra_amd", "liblyra_hip_plugins.so")],
                         capture_output=True, text=True).stdout
    for name in ("CreateQuantizer", "CreateGenerativeModel", "CreateFeatureExtractor", "CreateLogMelExtractor",
                 "BatchLyraEncoder::Create", "BatchLyraEncoder::Encode", "BatchLyraDecoder::SetEncodedPackets",
                 "BatchLyraDecoder::DecodeSamples", "EncodeFiles", "DecodeFiles", "EncodeWavs",
                 "DecodeFeaturesBatch", "ReadWav16", "WriteWav16"):
        assert f"chromemedia::codec::{name}(" in out
