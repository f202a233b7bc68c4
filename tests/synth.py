"""TEST INFRASTRUCTURE — portable synthetic tensors for the full-size hash fixtures (tests/golden/fullsize_hashes.json).

The fixtures are SHA-256 digests of whole outputs, so the inputs must be bit-identical on the builder container (where
oracle/gen_fullsize_hashes.py runs the torch form) and on the GPU box.  ``torch.randn`` is not promised to be (its CPU
kernels vectorise by instruction set); this generator uses integer arithmetic only: PCG64 raw 64-bit draws (numpy's
``Generator.integers`` over the full uint64 range is the raw stream), four 16-bit fields summed (Irwin-Hall, n = 4: a
bell-shaped value with standard deviation 65536 / sqrt(3)), an exact power-of-two scaling and a hand-written RNE rounding to bf16.
Every fixture also records the SHA-256 of its inputs, so a drift of the generator is reported as such."""
import hashlib

import numpy as np


def bell_bf16(rows: int, cols: int, seed: int, log2_scale: int) -> np.ndarray:
    """[rows, cols] bf16 bit patterns (uint16) of (sum of four uniform 16-bit integers - 131070) * 2**log2_scale:
    mean 0, standard deviation 37837 * 2**log2_scale (1.155 for log2_scale = -15), |value| <= 4 standard deviations * 0.87."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((rows, cols), np.uint16)
    step = max(1, (1 << 24) // max(cols, 1))
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        u = rng.integers(0, 1 << 64, size=(r1 - r0, cols), dtype=np.uint64, endpoint=False)
        s = ((u & np.uint64(0xFFFF)) + ((u >> np.uint64(16)) & np.uint64(0xFFFF)) + ((u >> np.uint64(32)) & np.uint64(0xFFFF)) + (u >> np.uint64(48))).astype(np.int64)
        f = ((s - 131070).astype(np.float32) * np.float32(2.0 ** log2_scale)).astype(np.float32)     # exact: |s - 131070| < 2^24, power-of-two scale
        b = f.view(np.uint32)
        out[r0:r1] = ((b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)   # RNE; no NaN / Inf here
    return out


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# name -> (M, N, K, seed, bias?)   BASELINE.json configs[1], configs[2]'s three GEMMs (Llama MLP 4096 -> 11008 -> 4096 at seq 2048) and the
# per-rank shards of configs[4] (Llama-3-70B over 8 GPUs at M = 4096): q/o, gate/up, down
FULLSIZE_CASES = {
    "cfg2_4096x4096x4096": (4096, 4096, 4096, 1234, False),
    "cfg3_gate_2048x11008x4096": (2048, 11008, 4096, 31, False),
    "cfg3_up_2048x11008x4096": (2048, 11008, 4096, 32, True),
    "cfg3_down_2048x4096x11008": (2048, 4096, 11008, 33, False),
    "l70b_qo_shard_4096x1024x8192": (4096, 1024, 8192, 34, False),
    "l70b_gateup_shard_4096x3584x8192": (4096, 3584, 8192, 35, False),
    "l70b_down_shard_4096x1024x28672": (4096, 1024, 28672, 36, True),
}


def fullsize_inputs(name):
    """(x bf16 bits [M, K], w bf16 bits [N, K], bias bf16 bits [N] or None) of one full-size case"""
    M, N, K, seed, bias = FULLSIZE_CASES[name]
    x = bell_bf16(M, K, seed, -15)                 # std 1.155
    w = bell_bf16(N, K, seed + 1000, -21)          # std 0.018
    b = bell_bf16(1, N, seed + 2000, -22)[0] if bias else None
    return x, w, b
