"""CPU (-m "not gpu"): the host baselines bench.py reports beside the non-headline workloads (benchlib/cpu_baseline.py) — at toy sizes: they run, they are the oracle's stages
(same bits as oracle.torch_ref composed by hand), and the JSON object carries the contract's keys."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_mlp_block_baseline_is_the_oracle_chain():
    from benchlib import cpu_baseline as CB
    from oracle import torch_ref as R
    build, ops, what = CB.mlp_block(R, M=48, H=128, I=256)
    assert ops == 2.0 * 48 * 512 * 128 + 2.0 * 48 * 128 * 256 and "whole block" in what
    y = build()()
    # by hand, from the same seed
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(48, 128, generator=g).to(torch.bfloat16)
    wgu, sgu = R.quantize_ref((torch.randn(512, 128, generator=g) * 0.02).to(torch.bfloat16), 1)
    wd, sd = R.quantize_ref((torch.randn(128, 256, generator=g) * 0.02).to(torch.bfloat16), 1)
    gu, _, _, _ = R.qlinear_ref(x, wgu, sgu)
    hq, hs, _ = R.silu_mul_quantize_ref(gu[:, :256], gu[:, 256:])
    want = R.epilogue_ref(torch._int_mm(hq, wd.t()), hs, sd, None, torch.bfloat16)
    assert torch.equal(y.view(torch.int16), want.view(torch.int16))


def test_pipeline_baseline_object():
    from benchlib import cpu_baseline as CB
    from oracle import torch_ref as R
    build, ops, what = CB.llama_layer(R, 32, 256, 512, 384, (256, 256), 1024, 256, 512, norms=True, what="toy layer")
    c = CB.cpu_baseline_pipeline(R, build, ops, what, budget_s=2.0, scale=4, scale_note="one layer x 4")
    assert c["kind"] == "port" and c["unit"] == "TOPS" and c["value"] > 0 and c["cores"] >= 1 and "one layer x 4" in c["sample"] and c["samples_per_step"] == 4
    assert abs(c["ms_per_step"] - 4 * c["ms_per_sample"]) <= 0.05 and c["value"] == max(c["thread_sweep_tops_median"].values())
    # the sharded form: down's input is the GATHERED intermediate (k_down != n_gu / 2)
    build, ops, _ = CB.llama_layer(R, 32, 256, 512, 160, (128, 256), 128, 128, 512, norms=True)
    y = build()()
    assert y.shape == (32, 128) and ops == 2.0 * 32 * (160 * 256 + 128 * 256 + 128 * 256 + 128 * 512)
    # the headline form keeps its keys
    h = CB.cpu_baseline(R, 64, 128, 128, budget_s=2.0)
    assert h["kind"] == "port" and set(h["stage_ms_min"]) == {"quantize", "int_mm", "epilogue"} and h["value"] >= (h["value_1_thread"] or 0)
