"""GPU: a short run of the quantisation / epilogue fuzz harness (tests/fuzz_quant.py) on every -m gpu pass; FUZZ_SECONDS
extends it.  It found the fp16 single-rounding fold that test_rmsnorm_fp16_rounds_to_f32_before_fp16 now pins."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_quant_and_epilogue_fuzz_is_clean():
    from tests import fuzz_quant
    n, bad = fuzz_quant.run(float(os.environ.get("FUZZ_SECONDS", "12")), seed=int(os.environ.get("FUZZ_SEED", "11")))
    assert n > 20 and bad == 0
