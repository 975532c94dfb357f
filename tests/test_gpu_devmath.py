"""Accuracy of the FP64 building blocks of the fast kernels (vk_devmath.h) measured on the hardware itself."""

import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_math_within_two_ulp(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(ROOT, "tools", "devmath_check")
    src = os.path.join(ROOT, "tools", "devmath_check.hip")
    hdr = os.path.join(ROOT, "victor_amd", "csrc", "vk_devmath.h")
    if not os.path.isfile(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        exe = str(tmp_path / "devmath_check")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "victor_amd", "csrc"),
                               src, "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.check_output([exe]).decode().strip().splitlines()[-1]
    res = json.loads(out)
    # the DPP wave reduction: identical in every lane, 6 rounding steps deep
    assert res["wave_sum_uniform"] == 1 and res["wave_sum_rel"] <= 8 * 2.3e-16
    # 4M samples each: sqrt/rsqrt over x in [2e-9, 5e8], recip likewise, exp over [-760, 0]
    assert res["sqrt_ulp"] <= 2.0 and res["rsqrt_ulp"] <= 2.0 and res["recip_ulp"] <= 1.0
    assert res["exp_ulp_normal"] <= 2.5
    # the one-step 1/sqrt of the streaming integrand: 3/8 e^2 with |e| <= 1.05e-7 - 20 ulp (4.4e-15) at worst, one on average
    assert res["rsqrt_nr_ulp_max"] <= 22.0 and res["rsqrt_nr_ulp_mean"] <= 1.5
    # exp_gauss (clamped product, magic-number rounding, integer exponent add), z in [-39, 39]: the plain 256-entry table with a
    # degree-3 polynomial (remainder f^4/24 <= 1.4e-13, one-sided: 2.8e-14 on average) and the replicated 64-entry table with
    # degree 4 (remainder f^5/120 <= 3.9e-14, odd: zero on average); NaN in -> NaN out, saturated arguments -> below 1e-300
    assert res["gauss_rel_max"][0] <= 1.5e-13 and abs(res["gauss_rel_mean"][0]) <= 3.5e-14
    assert res["gauss_rel_max"][1] <= 4.5e-14 and abs(res["gauss_rel_mean"][1]) <= 1e-15
    assert res["gauss_special_ok"] == 1
    # one Newton step from the ~2^-24 hardware seeds: e^2 for 1/x, 3/8 e^2 for 1/sqrt(x)
    assert res["recip_nr_rel"] <= 4e-15 and res["rsqrt_nr_x2_rel"] <= 5e-15
    # the raw hardware seeds are only ~2^-24, which is why each gets a third-order correction step
    assert 1e-9 < res["raw_v_rsq_f64_rel"] < 1e-6 and 1e-9 < res["raw_v_rcp_f64_rel"] < 1e-6
