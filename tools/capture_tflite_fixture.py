#!/usr/bin/env python3
"""Capture a REAL-TFLite fixture for this path -- the one artefact that would turn parity from "partial" into "pinned".

Not runnable in the build container (no TFLite runtime, no network); meant for any machine that has the reference
checkout and `pip install tflite-runtime` (or ai-edge-litert / tensorflow):

    python tools/capture_tflite_fixture.py --models /path/to/lyra/lyra/model_coeffs \
        --wavs /path/to/lyra/lyra/testdata --out tests/golden/tflite_capture.npz

It runs the reference's three flatbuffers exactly as tflite_model_wrapper.cc:36-103 does -- soundstream_encoder.tflite
(stateful: variables persist across Invoke), quantizer.tflite signatures `encode` / `decode`
(residual_vector_quantizer.cc:77-168), lyragan.tflite -- hop by hop over testdata/sample{1,2}_16kHz.wav, once with the
default delegate (XNNPACK, what lyra_components.cc:42-55 asks for) and once with the reference kernels
(experimental_op_resolver_type=BUILTIN_REF), and stores, in the layout of tests/golden/speech_sample1.npz:

    <wav>/<flavour>/pcm_in [T][320] int16, feats [T][64] f32, idx [T][46] i32, lossy [T][64] f32, pcm [T][320] int16

tests/test_tflite_capture.py lights up when the file exists: GPU == TFLite (indices / packets exact, PCM <= 1 LSB),
and reports which arithmetic mode (`xnnpack` / `exact` / `gemmlowp_double`) matched.
"""
import argparse
import os
import sys
import wave

import numpy as np


def interpreter_module():
    for name in ("tflite_runtime.interpreter", "ai_edge_litert.interpreter", "tensorflow.lite"):
        try:
            mod = __import__(name, fromlist=["Interpreter"])
            return mod
        except Exception:
            continue
    sys.exit("no TFLite runtime importable (tflite_runtime / ai_edge_litert / tensorflow)")


def make(mod, path, reference_kernels):
    kw = {}
    if reference_kernels:
        try:
            kw["experimental_op_resolver_type"] = mod.experimental.OpResolverType.BUILTIN_REF
        except Exception:
            from tensorflow.lite.python.interpreter import OpResolverType   # older layouts
            kw["experimental_op_resolver_type"] = OpResolverType.BUILTIN_REF
    it = mod.Interpreter(model_path=path, **kw)
    return it


def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000, path
        return np.frombuffer(w.readframes(w.getnframes()), np.int16)


def run(mod, models, pcm, reference_kernels, num_bits=184):
    enc = make(mod, os.path.join(models, "soundstream_encoder.tflite"), reference_kernels)
    gan = make(mod, os.path.join(models, "lyragan.tflite"), reference_kernels)
    q = make(mod, os.path.join(models, "quantizer.tflite"), reference_kernels)
    enc.allocate_tensors()
    gan.allocate_tensors()
    q_enc, q_dec = q.get_signature_runner("encode"), q.get_signature_runner("decode")
    ei, eo = enc.get_input_details()[0]["index"], enc.get_output_details()[0]["index"]
    gi, go = gan.get_input_details()[0]["index"], gan.get_output_details()[0]["index"]
    T = pcm.size // 320
    out = dict(pcm_in=pcm[:T * 320].reshape(T, 320).copy(), feats=np.zeros((T, 64), np.float32),
               idx=np.full((T, 46), -1, np.int32), lossy=np.zeros((T, 64), np.float32), pcm=np.zeros((T, 320), np.int16))
    ns = num_bits // 4
    for t in range(T):
        hop = out["pcm_in"][t]
        # Int16ToUnitFloat (dsp_utils.h:106-108)
        enc.set_tensor(ei, (-(hop.astype(np.float32)) / np.float32(-32768.0)).reshape(enc.get_input_details()[0]["shape"]))
        enc.invoke()
        f = enc.get_tensor(eo).reshape(-1).astype(np.float32)
        out["feats"][t] = f
        r = q_enc(input_frames=f.reshape(1, 1, 64), num_quantizers=np.array(ns, np.int32))
        idx = np.asarray(r["output_0"]).reshape(-1)[:ns].astype(np.int32)
        out["idx"][t, :ns] = idx
        full = np.full((46, 1, 1), -1, np.int32)
        full[:ns, 0, 0] = idx
        lossy = np.asarray(q_dec(encoding_indices=full)["output_0"]).reshape(-1).astype(np.float32)
        out["lossy"][t] = lossy
        gan.set_tensor(gi, lossy.reshape(gan.get_input_details()[0]["shape"]))
        gan.invoke()
        y = gan.get_tensor(go).reshape(-1)
        # UnitToInt16 (dsp_utils.h:54-88): scale by 32768, clip, truncate
        out["pcm"][t] = np.clip(y * np.float32(32768.0), -32768.0, 32767.0).astype(np.int16)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", required=True)
    ap.add_argument("--wavs", required=True)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tflite_capture.npz"))
    ap.add_argument("--hops", type=int, default=150)
    a = ap.parse_args()
    mod = interpreter_module()
    blob = {}
    for wav in ("sample1_16kHz", "sample2_16kHz"):
        pcm = read_wav(os.path.join(a.wavs, wav + ".wav"))[:a.hops * 320]
        for flavour, ref in (("xnnpack", False), ("reference_kernels", True)):
            r = run(mod, a.models, pcm, ref)
            for k, v in r.items():
                blob[f"{wav}/{flavour}/{k}"] = v
    np.savez_compressed(a.out, **blob)
    print("wrote", a.out, {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
