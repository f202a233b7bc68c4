"""TEST INFRASTRUCTURE — generates tests/golden/fullsize_hashes.json: SHA-256 digests of the WHOLE outputs of
oracle/torch_ref.py (QSPEC around ``torch._int_mm``, CPU) at the full sizes of BASELINE.json's configurations — the 4096^3
qlinear the metric is quoted on, the three GEMMs of the Llama MLP block (configs[2]) and three per-rank shards of the
Llama-3-70B configuration (configs[4]).  Run in the builder container:  ``python oracle/gen_fullsize_hashes.py``.

Inputs come from tests/synth.py (integer-only generator: bit-identical on every host); the digests are a few hundred
bytes and pin every bit of xq, xs, wq, ws, the int32 accumulator and y at sizes no committed .npz could hold.  "Parity
unpinned" by the reference (there is no reference source: /root/reference/CODE_OF_CONDUCT.md:1-80 only): the digests are
outputs of the contract-named primitive plus QSPEC, not of the reference itself."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_ref as R  # noqa: E402
from tests.synth import FULLSIZE_CASES, bell_bf16, fullsize_inputs, sha  # noqa: E402


def bf(a):
    return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16)


def bits(t):
    t = t.contiguous()
    return t.view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.bfloat16 else t.numpy()


def main():
    out = {"_generator": "oracle/gen_fullsize_hashes.py (oracle/torch_ref.py around torch._int_mm, torch %s, CPU)" % torch.__version__,
           "_inputs": "tests/synth.py::fullsize_inputs (bf16)", "cases": {}}
    for name, (M, N, K, seed, bias) in FULLSIZE_CASES.items():
        t0 = time.time()
        x, w, b = fullsize_inputs(name)
        wq, ws = R.quantize_ref(bf(w), 1)
        y, xq, xs, acc = R.qlinear_ref(bf(x), wq, ws, bf(b) if b is not None else None)
        assert torch.equal(acc, torch._int_mm(xq, wq.t()))
        rows = torch.arange(0, M, 97)                                  # exactness cross-check of the primitive on a row sample
        assert torch.equal(acc[rows].to(torch.int64), xq[rows].to(torch.int64) @ wq.to(torch.int64).t())
        out["cases"][name] = dict(M=M, N=N, K=K, seed=seed, bias=bool(bias), x=sha(x), w=sha(w), b=sha(b) if b is not None else None,
                                  xq=sha(bits(xq)), xs=sha(bits(xs)), wq=sha(bits(wq)), ws=sha(bits(ws)), acc=sha(bits(acc)), y=sha(bits(y)),
                                  acc_min=int(acc.min()), acc_max=int(acc.max()))
        print(f"{name}: acc[{int(acc.min())},{int(acc.max())}] {time.time() - t0:.1f}s", flush=True)
    # BASELINE configs[2] as ONE chain, every stage by torch's own eager ops around torch._int_mm: x -> gate, up (two qlinears on the same activation) ->
    # F.silu(gate) * up -> per-token quantisation -> down.  For bf16 the eager silu*mul IS QSPEC S1-S5 (exhaustive test in tests/test_oracle.py), so the
    # digests pin the producer-fused kernel (K1s) and the whole GatedMLP block to torch-generated data at full size.
    t0 = time.time()
    x, wg, _ = fullsize_inputs("cfg3_gate_2048x11008x4096")
    _, wu, _ = fullsize_inputs("cfg3_up_2048x11008x4096")
    _, wd, _ = fullsize_inputs("cfg3_down_2048x4096x11008")
    (gq, gs), (uq, us), (dq, ds) = (R.quantize_ref(bf(w_), 1) for w_ in (wg, wu, wd))
    gate, xq, xs, _ = R.qlinear_ref(bf(x), gq, gs)
    up = R.qlinear_ref(bf(x), uq, us)[0]
    hq, hs, h = R.silu_mul_quantize_ref(gate, up)
    y = R.epilogue_ref(R.int_gemm_ref(hq, dq), hs, ds, None, torch.bfloat16)
    out["chains"] = {"cfg3_mlp_block_2048x4096x11008": dict(
        what="x[2048,4096] -> gate/up (W 11008x4096 each) -> F.silu(gate)*up -> quantize -> down (W 4096x11008); inputs: the x of cfg3_gate and the weights of the three cfg3 cases, no bias",
        x=sha(x), gate=sha(bits(gate)), up=sha(bits(up)), h=sha(bits(h)), hq=sha(bits(hq)), hs=sha(bits(hs)), y=sha(bits(y)))}
    # the producer alone on inputs with a wider spread than GEMM outputs have (|g|, |u| up to 4.0 * 4: both tails of the sigmoid)
    g_in, u_in = bell_bf16(2048, 11008, 41, -13), bell_bf16(2048, 11008, 42, -15)
    hq2, hs2, h2 = R.silu_mul_quantize_ref(bf(g_in), bf(u_in))
    out["chains"]["silu_mul_quant_2048x11008"] = dict(what="quantize(F.silu(g) * u), g = bell_bf16(seed 41, 2^-13), u = bell_bf16(seed 42, 2^-15)",
                                                      g=sha(g_in), u=sha(u_in), h=sha(bits(h2)), hq=sha(bits(hq2)), hs=sha(bits(hs2)))
    print(f"chains: {time.time() - t0:.1f}s", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
