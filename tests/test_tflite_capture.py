"""Parity against the REAL TFLite interpreter -- dormant until someone drops tests/golden/tflite_capture.npz
(tools/capture_tflite_fixture.py, needs a TFLite runtime: not available in the build container).  With the file
present: the oracle (CPU, -m "not gpu") and the HIP path (-m gpu) against what tflite_model_wrapper.cc:36-103 computes
for testdata/sample{1,2}_16kHz.wav -- RVQ indices exact, PCM <= 1 LSB, feature codes exact -- for at least one of the three
arithmetic modes ("xnnpack" is expected for the XNNPACK-delegate capture, one of the builtin flavours for the
reference-kernel capture), and it says which one.  Since round 4 the per-op arithmetic of mode "xnnpack" is already held
against a real XNNPACK (tests/test_xnnpack_witness.py); this capture would add the delegate's own op-support decisions of
TensorFlow 2.11."""
MODES = ("xnnpack", "exact", "gemmlowp_double")
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CAPTURE = os.path.join(ROOT, "tests", "golden", "tflite_capture.npz")
needs_capture = pytest.mark.skipif(not os.path.exists(CAPTURE),
                                   reason="no real-TFLite capture (tools/capture_tflite_fixture.py); parity stays partial")


def _cases():
    z = np.load(CAPTURE)
    keys = sorted({k.rsplit("/", 1)[0] for k in z.files})
    return z, keys


def _score(idx, pcm, want):
    T = want["idx"].shape[0]
    ns = int((want["idx"][0] >= 0).sum())
    return (int((idx[:T, :ns] != want["idx"][:, :ns]).sum()), int(np.abs(pcm[:T].astype(int) - want["pcm"].astype(int)).max()))


@needs_capture
def test_oracle_vs_tflite_capture():
    from oracle import lyra_oracle
    lyra_oracle.build()
    z, keys = _cases()
    report = {}
    for key in keys:
        want = {k: z[f"{key}/{k}"] for k in ("pcm_in", "feats", "idx", "lossy", "pcm")}
        for mode in MODES:
            o = lyra_oracle.Oracle(mode=mode)
            st_e, st_d = lyra_oracle.Stream(o), lyra_oracle.Stream(o)
            T = want["pcm_in"].shape[0]
            idx = np.full((T, 46), -1, np.int32)
            pcm = np.zeros((T, 320), np.int16)
            ns = int((want["idx"][0] >= 0).sum())
            for t in range(T):
                f = st_e.encode(want["pcm_in"][t])
                idx[t] = o.rvq_encode(f[None], ns)[0]
                pcm[t] = st_d.decode(o.rvq_decode(idx[t][None])[0])
            report[(key, mode)] = _score(idx, pcm, want)
    print(report)
    for key in keys:
        best = min(report[(key, m)] for m in MODES)
        assert best[0] == 0 and best[1] <= 1, f"{key}: none of the arithmetic modes reproduces TFLite: {report}"


@needs_capture
@pytest.mark.gpu
def test_gpu_vs_tflite_capture():
    import lyra_amd
    z, keys = _cases()
    report = {}
    for key in keys:
        want = {k: z[f"{key}/{k}"] for k in ("pcm_in", "feats", "idx", "lossy", "pcm")}
        ns = int((want["idx"][0] >= 0).sum())
        for mode in MODES:
            ctx = lyra_amd.LyraHip(max_streams=8, requant=mode)
            try:
                T = want["pcm_in"].shape[0]
                idx = np.full((T, 46), -1, np.int32)
                pcm = np.zeros((T, 320), np.int16)
                for t in range(T):
                    f = ctx.extract(want["pcm_in"][t][None])
                    idx[t] = ctx.rvq_encode(f, 4 * ns)[0]
                    pcm[t] = ctx.generate(ctx.rvq_decode(idx[t][None]))[0]
                report[(key, mode)] = _score(idx, pcm, want)
            finally:
                ctx.close()
    print(report)
    for key in keys:
        best = min(report[(key, m)] for m in MODES)
        assert best[0] == 0 and best[1] <= 1, f"{key}: none of the arithmetic modes reproduces TFLite: {report}"
