"""GPU (-m gpu): the one-launch-per-side kernels (csrc/enc_side_kernel.hip, dec_side_kernel.hip; PARKED: only in the
build variant lyra_amd/variants/parked.so = `make -C lyra_amd/csrc parked`, there opt-in through LYRA_HIP_FUSED, bit 0
encoder side, bit 1 decoder side, bits 2 / 3 the pairwise-fused stage kernels; LYRA_HIP_RVQ_WIDE: the all-exact quantizer chain
kernels) against the seven-launch path: packets and PCM bit-identical
over several hops (ring phases advance), ragged tile (B not a multiple of 8), scattered stream ids, both requantisation
modes.  The seven-launch path itself is pinned to the oracle in test_gpu_parity.py, which can also be run whole with
LYRA_HIP_FUSED=3 in the environment."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


PARKED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lyra_amd", "variants", "parked.so")


def _ctx(fused, requant, rvq_chain=0):
    import lyra_amd
    if not os.path.exists(PARKED):
        pytest.skip("lyra_amd/variants/parked.so not built (make -C lyra_amd/csrc parked)")
    saved = {k: os.environ.get(k) for k in ("LYRA_HIP_FUSED", "LYRA_HIP_RVQ_WIDE")}
    os.environ["LYRA_HIP_FUSED"] = str(fused)
    os.environ["LYRA_HIP_RVQ_WIDE"] = str(rvq_chain)
    try:
        return lyra_amd.LyraHip(max_streams=4200, requant=requant, library=PARKED)
    finally:
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("requant", ["xnnpack", "exact", "gemmlowp_double"])
@pytest.mark.parametrize("B,bits", [(37, 184), (1, 64), (4096, 120)])
def test_side_kernels_equal_stage_kernels(requant, B, bits):
    T = 12 if B < 100 else 3
    rng = np.random.Generator(np.random.PCG64(B * 7 + bits))
    pcm = rng.integers(-32768, 32768, size=(T, B, 320)).astype(np.int16)
    pcm[1] //= 50
    ids = rng.permutation(4200)[:B].astype(np.int32)
    outs = []
    # bits 2 / 3 (round 4): encoder stages 1 + 2 / decoder stages 0 + 1 as one launch each (mode xnnpack only).  The last
    # two: the all-exact quantizer chain kernels of rounds 2-3 (LYRA_HIP_RVQ_WIDE = 1, 2) against the screened quantizer
    # the default build ships -- every packet equal means every certified index was the chain's index
    variants = [(0, 0), (3, 0), (1, 0), (2, 0)] + ([(4, 0), (8, 0), (12, 0)] if requant == "xnnpack" else []) + [(0, 1), (0, 2)]
    for fused, rvq_chain in variants:
        c = _ctx(fused, requant, rvq_chain)
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


@pytest.mark.parametrize("sub_batches", [None, 2])
def test_async_pipeline_equals_serial_soak(sub_batches):
    """The three-stream schedule of the `_dev` calls (extractor / quantizer / decoder streams, two alternating feature
    buffers inside the library) with NO synchronisation of any kind for 120 steps at B = 4096, against the same calls
    with the streams forced in call order (lyra_hip_set_serial).  The caller alternates TWO packet buffers (the
    two-buffer rule of include/lyra_hip.h "Streams": an encode that overtook the decode of two steps earlier would tear
    that decode's input) and keeps every step's PCM in a buffer of its own, so every step is checked: a missing
    ordering edge anywhere shows up as a PCM difference at some step."""
    import torch
    import lyra_amd
    B, T, bits = 4096, 120, 184
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    pcm = torch.randint(-32768, 32768, (8, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    nb = lyra_amd.packet_size(bits)

    def run(serial):
        ctx = lyra_amd.LyraHip(max_streams=B, sub_batches=sub_batches)
        ctx.torch_order = False
        ctx.set_serial(serial)
        pk = [torch.empty((B, nb), device=dev, dtype=torch.uint8) for _ in range(2)]
        out = torch.empty((T, B, 320), device=dev, dtype=torch.int16)
        torch.cuda.synchronize()
        try:
            for t in range(T):
                ctx.encode_dev(ids, pcm[t % 8], bits, pk[t & 1])
                ctx.decode_dev(ids, pk[t & 1], bits, out[t])
            ctx.synchronize()
            return [p.cpu().numpy() for p in pk], out.cpu().numpy()
        finally:
            ctx.close()

    a_pk, a_out = run(serial=False)
    s_pk, s_out = run(serial=True)
    for t in range(T):
        assert np.array_equal(a_out[t], s_out[t]), f"PCM differs at step {t}"
    assert np.array_equal(a_pk[0], s_pk[0]) and np.array_equal(a_pk[1], s_pk[1])
