"""Writes profiles/history/r04_xnnpack_witness.txt: real XNNPACK (torch's libtorch_cpu.so) vs the oracle's arithmetic models, per op.
CPU only, needs /root/reference (this container).  Same comparisons tests/test_xnnpack_witness.py asserts."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import xnn_compare as xc  # noqa: E402


def main():
    import torch
    out = []
    P = out.append
    P("XNNPACK witness -- round 4.  Real XNNPACK = the library torch %s's libtorch_cpu.so exports (332 xnn_* symbols; header" % torch.__version__)
    P("torch/include/xnnpack.h), driven through oracle/xnn_witness.c the way TFLite's delegate drives it (values with the")
    P("flatbuffers' quantisation parameters, one operator per TFLite operator).  Host: %s." % os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[1].strip())
    P("It is newer than the XNNPACK commit TF 2.11 pins and runs x86 micro-kernels: a witness, not the binary of record.")
    P("'differ' = outputs on which real XNNPACK and the named model disagree.\n")
    for name, path, key in (("soundstream_encoder.tflite", xc.ENC, "input_audio"), ("lyragan.tflite", xc.DEC, "input_audio")):
        P("=" * 110)
        P(name)
        P("-- int8 elementwise ops, exhaustive (LEAKY_RELU 256 codes, ADD 65,536 pairs, QUANTIZE +-5 ulp around every rounding")
        P("   boundary + 200,000 random floats, DEQUANTIZE 256 codes): XNNPACK vs oracle mode 'xnnpack' / vs the TFLite builtin form")
        for r in xc.elementwise_exhaustive(path):
            extra = "  (random floats only: %d)" % r["vs_tflite_random"] if "vs_tflite_random" in r else ""
            P("   op %3d %-11s n=%6d   vs xnnpack model: %d   vs TFLite builtin: %d%s" % (r["op"], r["name"], r["n"], r["vs_model"], r["vs_tflite"], extra))
        P("-- int8 convolutions on uniformly random codes: XNNPACK vs fp32 requantisation ('xnnpack') / Q31 single rounding ('exact') / gemmlowp")
        for r in xc.conv_requant_random(path, rows_per_layer=20000):
            P("   op %3d %-18s n=%8d max|acc|=%7d   vs xnnpack: %d   vs exact: %d   vs gemmlowp_double: %d" %
              (r["op"], r["name"], r["n"], r["max_acc"], r["vs_model"], r["vs_exact"], r["vs_gemmlowp"]))
        P("-- fp32 layers on random inputs: XNNPACK vs bias-first fused chain (taps ascending) / vs the round-3 order (chain from 0, bias last; tconv oldest row first)")
        for r in xc.fp32_layers_random(path):
            tail = ""
            if r["last"]:
                tail = "   <- one output channel: x86 nr2 kernel 4x2c4 SSE; vs that model: %d; max rel. diff to the fused chain %.2e" % (r["vs_c4"], r["max_rel_vs_bias_first"])
            P("   op %3d %-18s w%-18s n=%6d   vs bias-first chain: %d   vs round-3 order: %d%s" %
              (r["op"], r["name"], str(r["shape"]), r["n"], r["vs_bias_first"], r["vs_round3_order"], tail))
    P("=" * 110)
    P("every arithmetic op of both graphs, on operands traced from real speech by the graph interpreter in mode 'xnnpack' (state carried):")
    g = np.load(os.path.join(ROOT, "tests", "golden", "speech_sample1.npz"))
    key = "lossy_xnnpack" if "lossy_xnnpack" in g.files else "lossy_exact"
    for name, path, k, frames in (("encoder", xc.ENC, "input_audio", xc.speech_frames(8)),
                                  ("decoder", xc.DEC, "input_audio", [f.reshape(1, 1, 64) for f in g[key][20:28]])):
        res = xc.per_op_on_traced(path, k, frames)
        nq = sum(1 for kk in res if kk[2] == "q")
        nf = sum(1 for kk in res if kk[2] == "f")
        tot = sum(r["n"] for r in res.values())
        bad = sum(r["differ"] for r in res.values())
        ref = sum(r["refused"] for r in res.values())
        P("   %s: %d int8 + %d fp32 ops x %d frames, %d outputs: differ %d, refused by XNNPACK %d" % (name, nq, nf, len(frames), tot, bad, ref))
        for kk, r in res.items():
            if r["last"]:
                P("      (op %d, the one-channel transposed conv, compared with the x86 nr2-kernel model; vs the canonical fused chain: %d of %d "
                  "outputs differ, max abs %.2e)" % (kk[0], r["canon_differ"], r["n"], r["maxabs"]))
    txt = "\n".join(out) + "\n"
    path = os.path.join(ROOT, "profiles", "history", "r04_xnnpack_witness.txt")
    open(path, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
