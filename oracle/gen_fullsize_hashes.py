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
from tests.synth import FULLSIZE_CASES, fullsize_inputs, sha  # noqa: E402


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
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
