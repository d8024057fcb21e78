"""GPU: the decode GEMV stays bit-identical while a split-bf16 convolution (bf16 MFMAs) runs beside it on another stream.

This is the check that found the packed-FP32 defect of round 2 (profiles/README.md "Round 2: concurrency"): with v_pk_*_f32 in the
GEMV a few elements per thousand launches went wrong under MFMA load.  tools/gemv_stress.hip is built here with the library's own
flags (csrc/Makefile NOPK) from the library's own sources and must report zero mismatching elements, quiet and loaded."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "index-tts_amd", "csrc")
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


@pytest.mark.gpu
def test_gemv_bit_stable_beside_mfma_load(tmp_path):
    exe = str(tmp_path / "gemv_stress")
    srcs = [os.path.join(ROOT, "tools", "gemv_stress.hip")] + [os.path.join(CSRC, f) for f in ("gemv_fx.hip", "prof.hip", "conv1d_bf16x3.hip")]
    b = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", *NOPK, "-I" + CSRC, *srcs, "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    for args in (["1"], ["0", "512", "3520", "7"]):      # LayerNorm-folded GEMV beside the k = 3 conv; plain GEMV beside a k = 7 conv
        r = subprocess.run(["timeout", "-k", "10", "240", exe, *args], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "mismatching elements" in l]
        assert len(lines) == 2, r.stdout
        for l in lines:
            assert int(re.search(r"mismatching elements (\d+)", l).group(1)) == 0, l
