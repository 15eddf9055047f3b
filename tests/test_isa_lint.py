"""The shipped kernels' final ISA must not read an MFMA result from inside an asm block before the MFMA has written it
(tools/isa_lint.py; gfx950 does not interlock that read, tools/hazard_probe.hip -- round 6 found the packed LeakyReLU's
asm multiplies right behind the producing MFMA, correct only by the timing of that particular schedule)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "lyra_amd", "csrc")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not present")
def test_no_asm_block_reads_an_mfma_result_early():
    r = subprocess.run(["make", "-C", CSRC, "-j4", "ODIR=obj_asm", "asm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(os.path.join(CSRC, "obj_asm", f) for f in os.listdir(os.path.join(CSRC, "obj_asm")) if f.endswith(".s"))
    assert len(files) >= 4
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    bad = [b for p in files for b in isa_lint.lint(p)]
    assert not bad, f"asm blocks read MFMA results early in: {bad}"


def test_the_lint_sees_the_pattern():
    """The rule on a hand-written fragment: distance 1 is flagged, a padded read is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    import tempfile
    frag = """_ZN4lyra8x_kernelEv:
\tv_mfma_f32_16x16x4_f32 v[40:43], v63, v75, v[40:43]
\t;;#ASMSTART
\tv_pk_mul_f32 v[52:53], v[42:43], v[56:57]
\t;;#ASMEND
.Lfunc_end0:
_ZN4lyra8y_kernelEv:
\tv_mfma_f32_16x16x4_f32 v[40:43], v63, v75, v[40:43]
\ts_nop 15
\ts_nop 3
\t;;#ASMSTART
\tv_max_f32 v40, v40, v52
\t;;#ASMEND
.Lfunc_end1:
"""
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(frag)
    try:
        assert isa_lint.lint(f.name) == ["x_kernel"]
    finally:
        os.unlink(f.name)
