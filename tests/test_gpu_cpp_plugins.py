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
def test_cpp_plugin_surface(oracle_exact, golden_dir, tmp_path, bits):
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
    ref = lyra_oracle.run_batch(oracle_exact, pcm[:, None, :], bits // 4, do_decode=True)
    lines = open(bits_out).read().split()
    assert len(lines) == 12 and all(len(l) == bits for l in lines)
    for f, l in enumerate(lines):
        want = "".join(format(int(b), "08b") for b in ref["packets"][f, 0])[:bits]
        assert l == want, f"bit string differs at hop {f}"
    out = np.fromfile(pout, np.int16).reshape(12, 320)
    assert np.array_equal(out, ref["pcm"][:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate", [3200, 9200])
def test_cpp_batch_codec_twins(oracle_exact, golden_dir, tmp_path, bitrate):
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
    ref = lyra_oracle.run_batch(oracle_exact, pcm, bits // 4, do_decode=True)
    packets = np.fromfile(pk, np.uint8).reshape(T, n, -1)
    assert np.array_equal(packets, ref["packets"])
    out = np.fromfile(pout, np.int16).reshape(T, n, 320)
    assert np.array_equal(out, ref["pcm"])


def test_cpp_plugins_build_and_link():
    """CPU side: the adapters compile against the C ABI and export the lyra_components factory names."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "lyra_amd", "host")], stdout=subprocess.DEVNULL)
    out = subprocess.run(["nm", "-DC", os.path.join(ROOT, "lyra_amd", "liblyra_hip_plugins.so")],
                         capture_output=True, text=True).stdout
    for name in ("CreateQuantizer", "CreateGenerativeModel", "CreateFeatureExtractor", "CreateLogMelExtractor",
                 "BatchLyraEncoder::Create", "BatchLyraEncoder::Encode", "BatchLyraDecoder::SetEncodedPackets",
                 "BatchLyraDecoder::DecodeSamples"):
        assert f"chromemedia::codec::{name}(" in out
