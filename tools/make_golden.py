#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the reference's own TFLite
flatbuffers with oracle/tflite_interp.py (needs /root/reference; run in the
build container only).  The committed fixtures are what tests compare the C
oracle and the HIP path against on the GPU box.

    python tools/make_golden.py

Round 4: the interpreter evaluates the fp32 layers as bias-first fmaf chains
(fp32="chain", oracle/chain_f32.c) -- the order real XNNPACK computes
(tests/test_xnnpack_witness.py) and the canonical order of oracle/lyra_oracle.c --
so the C oracle and the GPU reproduce these fixtures bit for bit, floats
included.  (Rounds 1-3 accumulated in float64 and compared PCM to <= 1 LSB.)
Three arithmetic modes of the int8 regions: "xnnpack" (what the reference runs,
default), "exact", "gemmlowp_double" (TFLite builtin kernels).
"""
import hashlib
import os
import re
import sys
import wave

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.tflite_interp import Interpreter  # noqa: E402

REF = "/root/reference/lyra"
MC = REF + "/model_coeffs/"
OUT = os.path.join(ROOT, "tests", "golden")


def synth_pcm(stream_id, steps):
    """Bench input: UnitToInt16Scalar(U(-1,1)) i.i.d. (lyra_benchmark_lib.cc:233-239),
    seeded 0x4C797261 + stream_id."""
    rng = np.random.Generator(np.random.PCG64(0x4C797261 + stream_id))
    u = rng.uniform(-1.0, 1.0, size=steps * 320).astype(np.float32)
    v = np.clip(u * np.float32(32768.0), -32768.0, 32767.0)
    return np.trunc(v).astype(np.int16).reshape(steps, 320)


def run_codec(pcm_frames, mode, acc64=False, fp32="chain"):
    enc = Interpreter(MC + "soundstream_encoder.tflite", requant=mode, acc64=acc64, fp32=fp32)
    gan = Interpreter(MC + "lyragan.tflite", requant=mode, acc64=acc64, fp32=fp32)
    q = Interpreter(MC + "quantizer.tflite")
    feats, idxs, lossy_all, pcm_out, pcm_f = [], [], [], [], []
    for hop in pcm_frames:
        x = (-hop.astype(np.float32)) / np.float32(-32768.0)
        feat = enc.run_signature("serving_default", {"input_audio": x.reshape(1, 320)})["output_0"].reshape(-1)
        idx = q.run_signature("encode", {"input_frames": feat.reshape(1, 1, 64),
                                         "num_quantizers": np.int32(46)})["output_0"].reshape(-1)
        lossy = q.run_signature("decode", {"encoding_indices": idx.reshape(46, 1, 1)})["output_0"].reshape(-1)
        y = gan.run_signature("serving_default", {"input_audio": lossy.reshape(1, 1, 64)})["output_0"].reshape(-1)
        v = np.minimum(np.maximum(y * np.float32(32768.0), np.float32(-32768)), np.float32(32767))
        feats.append(feat); idxs.append(idx); lossy_all.append(lossy)
        pcm_out.append(np.trunc(v).astype(np.int16)); pcm_f.append(y)
    return dict(feats=np.array(feats, np.float32), idx=np.array(idxs, np.int32),
                lossy=np.array(lossy_all, np.float32), pcm=np.array(pcm_out, np.int16),
                pcm_f=np.array(pcm_f, np.float32))


def packets_of(idx, n):
    return np.array([[(int(r[2 * j]) << 4) | int(r[2 * j + 1]) for j in range(n // 2)] for r in idx], np.uint8)


def make_mixed():
    """Round 6: the fourth arithmetic mode, "builtin_mixed" (oracle/tflite_interp.py _requant), on the same 50 hops of
    speech, in a file of its own so that the other fixtures stay byte-identical:  python tools/make_golden.py --mixed"""
    w = wave.open(REF + "/testdata/sample1_16kHz.wav")
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    frames = pcm[:50 * 320].reshape(50, 320)
    g = run_codec(frames, "builtin_mixed")
    np.savez_compressed(os.path.join(OUT, "speech_sample1_mixed.npz"), pcm_in=frames, feats=g["feats"], idx=g["idx"],
                        lossy=g["lossy"], pcm=g["pcm"], pcmf=g["pcm_f"])
    for other in ("exact", "gemmlowp_double", "xnnpack"):
        o = run_codec(frames[:20], other)
        print("builtin_mixed vs", other, ": feature codes differing in 20 hops", int((o["feats"] != g["feats"][:20]).sum()), "of", 20 * 64)
    for n in (16, 30, 46):
        print("speech builtin_mixed", n * 4, "bits sha256[:16] of first 50 packets",
              hashlib.sha256(packets_of(g["idx"], n).tobytes()).hexdigest()[:16])


def main():
    if "--mixed" in sys.argv:
        return make_mixed()
    os.makedirs(OUT, exist_ok=True)
    # --- 1. real speech -----------------------------------------------------
    w = wave.open(REF + "/testdata/sample1_16kHz.wav")
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    NF = 50
    frames = pcm[:NF * 320].reshape(NF, 320)
    g = {}
    for mode in ("xnnpack", "exact", "gemmlowp_double"):
        g[mode] = run_codec(frames, mode)
        r64 = run_codec(frames, mode, True, "numpy")      # information only: sensitivity to the fp32 summation order
        print(mode, "speech: chain vs float64 accumulation: feature codes differing", int((r64["feats"] != g[mode]["feats"]).sum()),
              "indices", int((r64["idx"] != g[mode]["idx"]).sum()), "pcm max diff",
              int(np.abs(r64["pcm"].astype(int) - g[mode]["pcm"].astype(int)).max()))
    # known answers from BASELINE.md section 4 (first 150 hops) are checked in tests via hashes of
    # the first 60; here we just record.
    np.savez_compressed(os.path.join(OUT, "speech_sample1.npz"),
                        pcm_in=frames,
                        feats_xnnpack=g["xnnpack"]["feats"], idx_xnnpack=g["xnnpack"]["idx"],
                        lossy_xnnpack=g["xnnpack"]["lossy"], pcm_xnnpack=g["xnnpack"]["pcm"],
                        pcmf_xnnpack=g["xnnpack"]["pcm_f"],
                        feats_exact=g["exact"]["feats"], idx_exact=g["exact"]["idx"],
                        lossy_exact=g["exact"]["lossy"], pcm_exact=g["exact"]["pcm"],
                        pcmf_exact=g["exact"]["pcm_f"],
                        feats_double=g["gemmlowp_double"]["feats"], idx_double=g["gemmlowp_double"]["idx"],
                        pcm_double=g["gemmlowp_double"]["pcm"], pcmf_double=g["gemmlowp_double"]["pcm_f"])
    for mode in ("xnnpack", "exact"):
        for n in (16, 30, 46):
            h = hashlib.sha256(packets_of(g[mode]["idx"], n).tobytes()).hexdigest()[:16]
            print("speech", mode, n * 4, "bits sha256[:16] of first", NF, "packets", h)
    # --- 2. synthetic white noise (bench input), 4 streams x 6 steps ---------
    S, T = 4, 6
    pin = np.stack([synth_pcm(s, T) for s in range(S)], axis=1)  # [T][S][320]
    outs = [run_codec(pin[:, s], "exact") for s in range(S)]
    outx = [run_codec(pin[:, s], "xnnpack") for s in range(S)]
    np.savez_compressed(os.path.join(OUT, "noise_4x6.npz"), pcm_in=pin,
                        feats=np.stack([o["feats"] for o in outs], 1), idx=np.stack([o["idx"] for o in outs], 1),
                        pcm=np.stack([o["pcm"] for o in outs], 1), pcmf=np.stack([o["pcm_f"] for o in outs], 1),
                        feats_xnnpack=np.stack([o["feats"] for o in outx], 1), idx_xnnpack=np.stack([o["idx"] for o in outx], 1),
                        pcm_xnnpack=np.stack([o["pcm"] for o in outx], 1), pcmf_xnnpack=np.stack([o["pcm_f"] for o in outx], 1))
    # --- 3. RVQ fixture of the reference test --------------------------------
    src = open(REF + "/residual_vector_quantizer_test.cc").read()
    m = re.search(r"features_\{([^}]*)\}", src, re.S)
    feat = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",")], np.float32)
    q = Interpreter(MC + "quantizer.tflite")
    idx = q.run_signature("encode", {"input_frames": feat.reshape(1, 1, 64),
                                     "num_quantizers": np.int32(46)})["output_0"].reshape(-1)
    dec = {}
    for n in (16, 30, 46):
        i2 = idx.copy(); i2[n:] = -1
        dec[n] = q.run_signature("decode", {"encoding_indices": i2.reshape(46, 1, 1)})["output_0"].reshape(-1)
    # random feature vectors on the encoder's output grid + off-grid
    rng = np.random.Generator(np.random.PCG64(7))
    rnd = np.concatenate([(rng.integers(-128, 128, size=(64, 64)) - 20) * np.float32(0.26349151134490967),
                          rng.normal(0, 3, size=(64, 64))]).astype(np.float32)
    ridx = np.stack([q.run_signature("encode", {"input_frames": f.reshape(1, 1, 64), "num_quantizers": np.int32(46)}
                                     )["output_0"].reshape(-1) for f in rnd])
    rdec = np.stack([q.run_signature("decode", {"encoding_indices": i.reshape(46, 1, 1)})["output_0"].reshape(-1)
                     for i in ridx])
    np.savez_compressed(os.path.join(OUT, "rvq.npz"), fixture=feat, fixture_idx=idx.astype(np.int32),
                        fixture_dec16=dec[16], fixture_dec30=dec[30], fixture_dec46=dec[46],
                        rnd=rnd, rnd_idx=ridx.astype(np.int32), rnd_dec=rdec)
    # --- 4. the reference's 16 kHz test wavs as raw PCM (inputs of the whole-file known answers / LSD criterion) ---
    wavs = {}
    for name in ("sample1_16kHz", "sample2_16kHz"):
        w = wave.open(REF + "/testdata/" + name + ".wav")
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        wavs[name] = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    np.savez_compressed(os.path.join(OUT, "sample_wavs.npz"), **wavs)
    print("rvq fixture idx", list(map(int, idx)))
    print("done ->", OUT)


if __name__ == "__main__":
    main()
