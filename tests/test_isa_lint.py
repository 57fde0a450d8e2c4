"""Static guard on the gfx950 code objects (no GPU): the dominant kernels must not use scratch memory or index
their registers dynamically -- the Winograd epilogue did (an accumulator array indexed by a runtime plane half:
72 s_set_gpr_idx pairs + 330 v_mov per wave) until round 2 found it in the ISA.  tools/isa_lint.py checks every
kernel (about two minutes); here only the files whose kernels own large accumulator arrays
(conv_f32_first.hip spilled 0.5-1.2 KB per lane in every build until LLVM's code sinking was fenced off)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="no hipcc")
def test_hot_kernels_have_no_scratch_and_no_dynamic_register_indexing():
    csrc = os.path.join(ROOT, "yolo2_light_amd", "csrc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"),
                        os.path.join(csrc, "conv_f32_wino32.hip"), os.path.join(csrc, "conv_f32_smallk.hip"),
                        os.path.join(csrc, "conv_f32_first.hip"), os.path.join(csrc, "conv_f32_firstm.hip"),
                        os.path.join(csrc, "conv_f32_row3.hip")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "conv_f32_wino32_kernel" in r.stdout and "conv_f32_first_kernel" in r.stdout and "conv_f32_row3_kernel" in r.stdout
    assert "conv_f32_firstm_signs_kernel" in r.stdout and "conv_f32_row3v_kernel" in r.stdout
    assert "0 kernel(s) flagged" in r.stdout
