"""The int8 stages quantise without dividing (lyra_amd/csrc/lyra_dev.h quantize_f: x * RN(1/s) plus one Markstein
correction instead of the IEEE division TFLite's AffineQuantize / the oracle's quantize_f perform).  Bit-exactness is
proved by sweep, on the CPU, with the same IEEE operations the GPU executes (oracle/quantize_proof.c): every QUANTIZE
scale either graph holds, every float of either sign in the 14 binades around the 256 code boundaries; with
LYRA_PROOF_FULL=1 all 2^32 bit patterns per scale (a round-3 run of that is logged in profiles/history/r03_quantize_proof.txt).
"""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def quantize_scales():
    """(name, scale, zero_point) of every `<side>.quant.<n>.q` entry of the weight container (pack_format.h)."""
    raw = open(os.path.join(ROOT, "lyra_amd", "assets", "lyra_v1.lyrapack"), "rb").read()
    assert raw[:8] == b"LYRAPK01"
    n = struct.unpack_from("<I", raw, 8)[0]
    out = []
    for i in range(n):
        off = 16 + 96 * i
        name = raw[off:off + 56].split(b"\0")[0].decode()
        o, nb = struct.unpack_from("<2Q", raw, off + 80)
        if re.search(r"\.quant\.\d+\.q$", name):
            v = np.frombuffer(raw, np.float32, nb // 4, o)
            out.append((name, float(v[0]), int(v[1])))
    return out


def test_division_free_quantize_is_bit_exact():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "oracle", "libquantize_proof.so"))
    L.lo_quantize_proof.restype = C.c_uint64
    L.lo_quantize_proof.argtypes = [C.c_float, C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    scales = quantize_scales()
    assert len(scales) >= 5          # enc: q_r0, q_x1; dec: q0, q1, q3 are the ones the kernels use
    full = int(os.environ.get("LYRA_PROOF_FULL", "0"))
    seen = set()
    for name, s, z in scales:
        if (s, z) in seen:
            continue
        seen.add((s, z))
        first = C.c_uint32(0)
        bad = L.lo_quantize_proof(s, z, min(8, os.cpu_count() or 1), full, C.byref(first))
        assert bad == 0, f"{name}: scale {s!r} zero point {z}: {bad} inputs differ, first bit pattern {first.value:#010x}"
