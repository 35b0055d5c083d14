"""Hardware behaviour the kernels rely on, checked on the device the suite runs on (round 6).  Each test compiles a stand-alone HIP
program from tools/ with hipcc (gfx950) and runs it once:
  * vector loads of BOTH cache policies return in issue order, so a counted `s_waitcnt vmcnt(N)` behind a default-policy load and N
    non-temporal loads covers the first load (every kernel that mixes `load4` and `load4_stream` under compiler-counted waits:
    rows_layernorm, pool_mix_cols*, attn_flash_split*) - tools/r06_load_order_probe.hip;
  * `v_cvt_scalef32_pk_fp8_f32` of a value clamped at 448 / 2^s with the scale operand 2^-s is bit for bit the multiply + clamp +
    `v_cvt_pk_fp8_f32` sequence of the split stores (common.h, pack_fp8x4_shift) - tools/r06_cvt_probe.hip."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


def _build_and_run(src, tmp_path, timeout=240):
    exe = str(tmp_path / os.path.splitext(os.path.basename(src))[0])
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-Wno-unused-result", "-Wno-unused-value", "-o", exe, os.path.join(ROOT, src)],
                   check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], check=True, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    sys.stdout.write(out)
    return out


@pytest.mark.skipif(HIPCC is None, reason="hipcc not found")
def test_loads_of_both_cache_policies_return_in_issue_order(tmp_path):
    out = _build_and_run("tools/r06_load_order_probe.hip", tmp_path)
    assert "RESULT: every first load had landed" in out, out
    assert "out-of-order" not in out


@pytest.mark.skipif(HIPCC is None, reason="hipcc not found")
def test_scaled_fp8_conversion_equals_the_multiply_clamp_convert_sequence(tmp_path):
    out = _build_and_run("tools/r06_cvt_probe.hip", tmp_path)
    assert "differences shift 0: 0, shift 11: 0" in out, out
