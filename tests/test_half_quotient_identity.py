"""CPU: the claim behind the one-step division-free encode of 16-bit inputs (protoquant_amd/csrc/quant_device.h: quotient_fast1).

For bf16 / fp16 rows the scale s = amax / 127 comes from a 16-bit amax and every element is a 16-bit magnitude <= amax: the whole
domain is 2.7e8 (bf16, scales on the fast path) / 5.0e8 (fp16) pairs per sign and is enumerated here on the host in C with the
same float operations (fmaf = one rounding).  One correction step must reproduce rintf(x / s) on ALL of it; zero steps must not
(otherwise the enumeration would not be testing anything).  The GPU repeats the enumeration with its own fma in
tests/test_gpu_parity.py::test_half_encode_whole_domain."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def enum_bin(tmp_path_factory):
    out = tmp_path_factory.mktemp("hq") / "half_quotient_enum"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", str(out), os.path.join(HERE, "half_quotient_enum.c"), "-lm"], check=True)
    return str(out)


@pytest.mark.parametrize("fmt,min_pairs", [("bf16", 5.0e8), ("fp16", 1.0e9)])
def test_one_step_quotient_is_exact_on_the_whole_16bit_domain(enum_bin, fmt, min_pairs):
    r = subprocess.run([enum_bin, fmt], capture_output=True, text=True, timeout=600, check=True)
    pairs, bad0, bad1, bad2 = (int(v) for v in r.stdout.split())
    assert pairs > min_pairs, pairs
    assert bad1 == 0 and bad2 == 0, (bad1, bad2)
    assert bad0 > 0            # without a correction step the product x * (1/s) does miss ties: the test discriminates
