"""GPU (-m gpu): the one-launch-per-side kernels (csrc/enc_side_kernel.hip, dec_side_kernel.hip; opt-in through
LYRA_HIP_FUSED, bit 0 encoder side, bit 1 decoder side) against the seven-launch path: packets and PCM bit-identical
over several hops (ring phases advance), ragged tile (B not a multiple of 8), scattered stream ids, both requantisation
modes.  The seven-launch path itself is pinned to the oracle in test_gpu_parity.py, which can also be run whole with
LYRA_HIP_FUSED=3 in the environment."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(fused, requant):
    import lyra_amd
    old = os.environ.get("LYRA_HIP_FUSED")
    os.environ["LYRA_HIP_FUSED"] = str(fused)
    try:
        return lyra_amd.LyraHip(max_streams=4200, requant=requant)
    finally:
        if old is None:
            del os.environ["LYRA_HIP_FUSED"]
        else:
            os.environ["LYRA_HIP_FUSED"] = old


@pytest.mark.parametrize("requant", ["exact", "gemmlowp_double"])
@pytest.mark.parametrize("B,bits", [(37, 184), (1, 64), (4096, 120)])
def test_side_kernels_equal_stage_kernels(requant, B, bits):
    T = 12 if B < 100 else 3
    rng = np.random.Generator(np.random.PCG64(B * 7 + bits))
    pcm = rng.integers(-32768, 32768, size=(T, B, 320)).astype(np.int16)
    pcm[1] //= 50
    ids = rng.permutation(4200)[:B].astype(np.int32)
    outs = []
    for fused in (0, 3, 1, 2):
        c = _ctx(fused, requant)
        try:
            run = []
            for t in range(T):
                pk = c.encode(pcm[t], bits, ids)
                run.append((pk, c.decode(pk, bits, ids)))
            outs.append(run)
        finally:
            c.close()
    for other in outs[1:]:
        for t in range(T):
            assert np.array_equal(outs[0][t][0], other[t][0]), f"packets differ at hop {t}"
            assert np.array_equal(outs[0][t][1], other[t][1]), f"PCM differs at hop {t}"
