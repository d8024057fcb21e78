"""CPU: the device code of libidxtts_hip.so holds no packed-FP32 VALU instructions.

On gfx950 / ROCm 7.2 a wave's v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 results were measured to go wrong while a wave of ANOTHER
kernel ran bf16 MFMAs on the same SIMD (tools/gemv_stress.hip; profiles/README.md "Round 2: concurrency"), and the stages of
this library are meant to run beside each other (indextts_amd/serving.py), so the build switches the instructions off
(csrc/Makefile NOPK).  This test keeps that switch from getting lost."""
import glob
import os
import re
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_no_packed_fp32_instructions_in_device_code(tmp_path):
    import __graft_entry__ as g
    from indextts_amd import _lib
    so = _lib.library_path() if hasattr(_lib, "library_path") else os.path.join(os.path.dirname(_lib.__file__), "libidxtts_hip.so")
    if not os.path.exists(so):
        g.build()
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    local = tmp_path / "lib.so"
    shutil.copy(so, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True, cwd=tmp_path)
    images = sorted(glob.glob(str(tmp_path / "lib.so.*gfx950*")))
    assert images, "no gfx950 code objects found in the library"
    n_mfma, packed = 0, []
    for img in images:
        dis = subprocess.run([OBJDUMP, "-d", img], check=True, capture_output=True, text=True).stdout
        n_mfma += len(re.findall(r"\bv_mfma_", dis))
        packed += re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)
    assert n_mfma > 1000, "disassembly looks empty"
    assert not packed, f"{len(packed)} packed-FP32 instructions in the device code: the NOPK build flag is missing"
