"""BatchLyraEncoder / BatchLyraDecoder with the reference's full per-stream semantics (SURVEY.md 8f rows 1, 3, 4;
lyra/lyra_encoder.cc:113-156, lyra/lyra_decoder.cc:172-373): resampling, DTX, packet queueing, DecodeSamples(n) that
straddles hops, packet-loss concealment, comfort noise, cross-fades.  The C++ twins run on the GPU through the C ABI
(lyra_amd/decoder_demo); the expectation is oracle/lyra_codec_model.py, the per-stream restatement of the reference's
classes over the CPU oracle.  Packets bit-exact; PCM bit-exact wherever only the generative model speaks, within 2 LSB
where comfort noise is mixed in (device fp64 sin/cos/exp vs host libm, see test_resampler_cng.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_reference_model_decoder_state_machine(oracle_default):
    """CPU: the restated LyraDecoder control flow against the reference's own expectations (lyra_decoder_test.cc):
    right sample counts for arbitrary requests, comfort noise reached after concealment (4 hops) + fade (2 hops) of
    loss, left again 2 hops after packets resume."""
    from oracle import lyra_codec_model as M
    from oracle import lyra_oracle
    rng = np.random.default_rng(0)
    enc = M.RefLyraEncoder(oracle_default, 16000, 64, False)
    dec = M.RefLyraDecoder(oracle_default, 16000, cng_seed=1)
    for t in range(5):
        dec.SetEncodedPacket(enc.Encode(rng.integers(-3000, 3000, 320).astype(np.int16)))
        assert dec.DecodeSamples(320).size == 320 and not dec.is_comfort_noise()
    seen = []
    for t in range(8):                 # eight lost packets
        out = np.concatenate([dec.DecodeSamples(k) for k in (100, 7, 213)])
        assert out.size == 320
        seen.append(dec.is_comfort_noise())
    assert seen == [False] * 5 + [True] * 3        # 4 hops of concealment, 2 hops of fade
    for t in range(3):
        dec.SetEncodedPacket(enc.Encode(rng.integers(-3000, 3000, 320).astype(np.int16)))
        dec.DecodeSamples(320)
    assert not dec.is_comfort_noise()


def _run_session(tmp_path, oracle, rate, bitrate, dtx, pcm, script, demo=None, pipelined=False):
    import lyra_amd
    if demo is None:
        demo = os.path.join(ROOT, "lyra_amd", "decoder_demo")
        assert os.path.exists(demo), "lyra_amd/decoder_demo not built (__graft_entry__.build())"
    T, n, hop = pcm.shape
    pin, sc = tmp_path / "in.s16", tmp_path / "script.txt"
    pk, ln, pout = tmp_path / "pk.bin", tmp_path / "len.i32", tmp_path / "out.s16"
    pcm.tofile(pin)
    sc.write_text("\n".join(f"{mask} " + " ".join(map(str, sizes)) for mask, sizes in script) + "\n")
    # pipelined: the session through EncodeAsync / WaitEncoded and DecodeSamplesAsync / WaitDecoded, two deep (decoder_demo.cc)
    env = dict(os.environ, LYRA_DEMO_PIPELINED="1" if pipelined else "0")
    r = subprocess.run([demo, lyra_amd.default_model_dir(), str(sc), str(pin), str(rate), str(bitrate), str(int(dtx)),
                        str(n), str(pk), str(ln), str(pout)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    ps = {3200: 8, 6000: 15, 9200: 23}[bitrate]
    packets = np.fromfile(pk, np.uint8).reshape(T, n, ps)
    lengths = np.fromfile(ln, np.int32).reshape(T, n)
    out = np.fromfile(pout, np.int16)
    return packets, lengths, out


@pytest.mark.gpu
@pytest.mark.parametrize("rate,bitrate,dtx", [(16000, 6000, False), (48000, 3200, False), (8000, 9200, False),
                                              (16000, 9200, True), (32000, 6000, True)])
def test_batch_codec_session_vs_reference_model(tmp_path, golden_dir, oracle_default, rate, bitrate, dtx):
    from oracle import lyra_codec_model as M
    from oracle import lyra_oracle
    bits = {3200: 64, 6000: 120, 9200: 184}[bitrate]
    speech = np.load(os.path.join(golden_dir, "sample_wavs.npz"))["sample1_16kHz"]
    T, n, hop = 40, 4, rate // 50
    # streams: speech, speech with a silent middle (DTX food), noise, speech delayed
    up = lyra_oracle.Resampler(16000, rate) if rate != 16000 else None
    base = speech[:16000 * 2]
    ext = np.concatenate([up.Resample(base[i:i + 320]) for i in range(0, base.size, 320)]) if up is not None else base
    rng = np.random.default_rng(rate + bitrate)
    s0 = ext[:T * hop]
    s1 = s0.copy(); s1[10 * hop:25 * hop] = 0
    s2 = np.clip(rng.normal(0, 500, T * hop), -32768, 32767).astype(np.int16)
    s3 = np.concatenate([np.zeros(5 * hop, np.int16), ext[:(T - 5) * hop]])
    pcm = np.stack([s.reshape(T, hop) for s in (s0, s1, s2, s3)], axis=1).astype(np.int16)     # [T][n][hop]
    # loss script: burst of 9 lost packets for stream 0, scattered single losses for stream 2, none for 1 and 3;
    # playout in odd chunk sizes that straddle hops
    script = []
    for t in range(T):
        mask = "".join(["0" if 12 <= t < 21 else "1", "1", "0" if t % 7 == 3 else "1", "1"])
        sizes = [hop] if t % 3 == 0 else ([hop // 4 + 3, hop - hop // 4 - 3] if t % 3 == 1 else [1, hop // 2, hop - hop // 2 - 1])
        script.append((mask, sizes))
    packets, lengths, out = _run_session(tmp_path, oracle_default, rate, bitrate, dtx, pcm, script)

    encs = [M.RefLyraEncoder(oracle_default, rate, bits, dtx) for _ in range(n)]
    decs = [M.RefLyraDecoder(oracle_default, rate, cng_seed=0x4C797261 ^ s) for s in range(n)]
    pos = 0
    n_exact = n_total = 0
    worst = 0
    saw_cng = False
    for t, (mask, sizes) in enumerate(script):
        for s in range(n):
            p = encs[s].Encode(pcm[t, s])
            assert lengths[t, s] == p.size, (t, s)
            if p.size:
                assert np.array_equal(packets[t, s], p), (t, s)
                if mask[s] == "1":
                    decs[s].SetEncodedPacket(p)
        for k in sizes:
            got = out[pos:pos + n * k].reshape(n, k)
            pos += n * k
            for s in range(n):
                want = decs[s].DecodeSamples(k)
                d = np.abs(got[s].astype(int) - want.astype(int))
                worst = max(worst, int(d.max()))
                n_exact += int((d == 0).sum()); n_total += k
                saw_cng = saw_cng or decs[s].is_comfort_noise()
    assert pos == out.size
    assert worst <= 2, worst
    assert n_exact / n_total > 0.97
    assert saw_cng                      # the 9-packet burst takes stream 0 all the way into comfort noise
    if dtx:
        assert (lengths == 0).sum() > 5   # the silent stretch is sent as empty packets


def _build_fake_demo(tmp_path):
    """decoder_demo + lyra_batch_codec.cc against tests/host_stub/fake_lyra_hip_codec.cc (no GPU, no product library)."""
    host = os.path.join(ROOT, "lyra_amd", "host")
    exe = str(tmp_path / "decoder_demo_fake")
    subprocess.check_call(["g++"] + _FAKE_DEMO_FLAGS + ["-std=c++17", "-I" + host, "-I" + os.path.join(host, "shims"), "-I" + ROOT,
                           "-o", exe, os.path.join(host, "decoder_demo.cc"), os.path.join(host, "lyra_batch_codec.cc"),
                           os.path.join(ROOT, "tests", "host_stub", "fake_lyra_hip_codec.cc")])
    return exe


_FAKE_DEMO_FLAGS = ["-O2"]


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("rate,bitrate,dtx", [(16000, 6000, False), (48000, 3200, True), (8000, 9200, False),
                                              (32000, 6000, True)])
def test_batch_codec_host_logic_against_fake_abi(tmp_path, rate, bitrate, dtx, pipelined):
    """CPU: the C++ twins' host logic alone -- the C ABI underneath replaced by integer formulas with per-stream call
    counters (tests/host_stub), the reference model assembled from the same formulas (fake_kit.py).  Everything the
    twins do around the device calls is then checked EXACTLY, sample for sample: resampling bookkeeping and leftovers,
    DTX empty packets, which streams are served by which call in which round, queueing, concealment, fades, comfort
    noise, noise-estimator updates on received hops only."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "host_stub"))
    from fake_kit import FakeKit
    from oracle import lyra_codec_model as M
    bits = {3200: 64, 6000: 120, 9200: 184}[bitrate]
    T, n, hop = 60, 6, rate // 50
    rng = np.random.default_rng(rate + bitrate)
    loud = rng.integers(-12000, 12000, size=(T, n, hop)).astype(np.int16)
    pcm = loud.copy()
    pcm[15:30, 1] = rng.integers(-20, 20, size=(15, hop))      # a quiet stretch: DTX food
    pcm[:, 4] = rng.integers(-30, 30, size=(T, hop))           # a stream that is noise throughout
    script = []
    for t in range(T):
        mask = "".join(["0" if 12 <= t < 21 else "1", "1", "0" if t % 7 == 3 else "1", "1", "1",
                        "0" if 40 <= t < 43 else "1"])
        sizes = [hop] if t % 3 == 0 else ([hop // 4 + 3, hop - hop // 4 - 3] if t % 3 == 1 else [1, hop // 2, hop - hop // 2 - 1])
        if t % 11 == 5:
            sizes = [0] + sizes          # DecodeSamples(0) is legal
        script.append((mask, sizes))
    packets, lengths, out = _run_session(tmp_path, None, rate, bitrate, dtx, pcm, script, demo=_build_fake_demo(tmp_path),
                                         pipelined=pipelined)

    encs = [M.RefLyraEncoder(None, rate, bits, dtx, kit=FakeKit()) for _ in range(n)]
    decs = [M.RefLyraDecoder(None, rate, cng_seed=0, kit=FakeKit()) for _ in range(n)]
    pos = 0
    saw_cng = saw_empty = False
    for t, (mask, sizes) in enumerate(script):
        for s_ in range(n):
            p = encs[s_].Encode(pcm[t, s_])
            assert lengths[t, s_] == p.size, (t, s_)
            saw_empty = saw_empty or p.size == 0
            if p.size:
                assert np.array_equal(packets[t, s_], p), (t, s_)
                if mask[s_] == "1":
                    decs[s_].SetEncodedPacket(p)
        for k in sizes:
            got = out[pos:pos + n * k].reshape(n, k)
            pos += n * k
            for s_ in range(n):
                want = decs[s_].DecodeSamples(k)
                assert np.array_equal(got[s_], want), f"tick {t}, stream {s_}, DecodeSamples({k})"
                saw_cng = saw_cng or decs[s_].is_comfort_noise()
    assert pos == out.size and saw_cng and (saw_empty == dtx)


@pytest.mark.parametrize("pipelined", [False, True])
def test_batch_codec_host_logic_under_address_and_ub_sanitizers(tmp_path, monkeypatch, pipelined):
    """The same sessions with the twins' host code built with -fsanitize=address,undefined (CPU only): any report aborts
    the demo, which fails the session."""
    probe = subprocess.run(["g++", "-fsanitize=address,undefined", "-x", "c++", "-", "-o", str(tmp_path / "probe")],
                           input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("no sanitizer runtimes for g++ here")
    monkeypatch.setitem(globals(), "_FAKE_DEMO_FLAGS", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    monkeypatch.setenv("ASAN_OPTIONS", "detect_leaks=1:abort_on_error=1")
    test_batch_codec_host_logic_against_fake_abi(tmp_path, 48000, 3200, True, pipelined)
    test_batch_codec_host_logic_against_fake_abi(tmp_path, 8000, 9200, False, pipelined)


def test_file_transcode_host_logic_against_fake_abi(tmp_path):
    """CPU: EncodeFiles / DecodeFiles (lyra_file_codec.cc: WAV I/O, files of different lengths sharing one batch, files
    leaving the batch as they end, trailing partial hops dropped) over the fake C ABI: every .lyra and every decoded WAV
    must be what the reference model with the same fake components gives for that file alone."""
    import sys
    import wave
    sys.path.insert(0, os.path.join(ROOT, "tests", "host_stub"))
    from fake_kit import FakeKit
    from oracle import lyra_codec_model as M
    host = os.path.join(ROOT, "lyra_amd", "host")
    exe = str(tmp_path / "file_demo_fake")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + host, "-I" + os.path.join(host, "shims"), "-I" + ROOT, "-o", exe,
                           os.path.join(host, "file_demo.cc"), os.path.join(host, "lyra_file_codec.cc"),
                           os.path.join(host, "lyra_batch_codec.cc"),
                           os.path.join(ROOT, "tests", "host_stub", "fake_lyra_hip_codec.cc")])
    rng = np.random.default_rng(5)
    lengths = {"a": 320 * 17 + 111, "b": 320 * 5, "c": 320 * 6, "d": 320 * 11 - 7, "tiny": 100, "e": 320 * 17}
    files = {k: rng.integers(-9000, 9000, n).astype(np.int16) for k, n in lengths.items()}
    wavs = []
    for name, pcm in files.items():
        with wave.open(str(tmp_path / f"{name}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(pcm.tobytes())
        wavs.append(str(tmp_path / f"{name}.wav"))
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    r = subprocess.run([exe, "unused_model_dir", "9200", str(out_dir)] + wavs, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    for name, pcm in files.items():
        hops = len(pcm) // 320
        enc = np.fromfile(out_dir / f"{name}.lyra", np.uint8)
        with wave.open(str(out_dir / f"{name}_decoded.wav"), "rb") as w:
            assert w.getnchannels() == 1 and w.getframerate() == 16000
            dec = np.frombuffer(w.readframes(w.getnframes()), np.int16)
        assert enc.size == hops * 23 and dec.size == hops * 320, name
        e = M.RefLyraEncoder(None, 16000, 184, False, kit=FakeKit())
        d = M.RefLyraDecoder(None, 16000, cng_seed=0, kit=FakeKit())
        for h in range(hops):
            p = e.Encode(pcm[h * 320:(h + 1) * 320])
            assert np.array_equal(enc[h * 23:(h + 1) * 23], p), (name, h)
            d.SetEncodedPacket(p)
            assert np.array_equal(dec[h * 320:(h + 1) * 320], d.DecodeSamples(320)), (name, h)
