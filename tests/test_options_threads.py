"""CPU: the library's behaviour switches are an immutable snapshot per C-ABI call (include/pq_hip.h, pq_set_option) — concurrent host threads are defined —
and the planner refuses a fused split-K plan whose grid does not fit the CUs the device reports (PQ_FAKE_CUS)."""
import threading

import pytest

from protoquant_amd import _lib


@pytest.fixture
def L():
    lib = _lib.lib()
    yield lib
    for n in ("PQ_FORCE_VARIANT", "PQ_FAKE_CUS", "PQ_FSK", "PQ_FSK_SYMMETRIC", "PQ_NO_SPLITK", "PQ_NO_MIDM", "PQ_NO_TAILSPLIT"):
        _lib.set_option(n, "")


def test_a_call_never_sees_a_mixture_of_two_option_sets(L):
    """4096 x 6144 x 4096: under auto the dispatcher names the 256 x 256 tile WITH a tail split; under PQ_FORCE_VARIANT=generic "generic64".  The name is
    built from two reads of the switches (variant choice, then the tail plan): unpinned, a flip between the reads yields "sp256_16x16x64" with no tail —
    a result neither option set can produce.  One thread flips the switch 20 000 times while four threads ask."""
    auto, forced = b"sp256_16x16x64 + sp128 tail (N)", b"generic64"
    assert L.pq_gemm_variant_name(4096, 6144, 4096, 4096, 4096) == auto
    stop = threading.Event()
    seen, bad = set(), []

    def ask():
        while not stop.is_set():
            n = L.pq_gemm_variant_name(4096, 6144, 4096, 4096, 4096)
            seen.add(n)
            if n not in (auto, forced):
                bad.append(n)

    ts = [threading.Thread(target=ask) for _ in range(4)]
    for t in ts:
        t.start()
    for i in range(20000):
        assert L.pq_set_option(b"PQ_FORCE_VARIANT", b"generic" if i % 2 == 0 else b"") == 0
    stop.set()
    for t in ts:
        t.join()
    assert not bad, bad[:3]
    assert seen == {auto, forced}


def test_concurrent_setters_do_not_lose_each_others_switches(L):
    """two threads set DIFFERENT switches concurrently (copy - modify - publish under the writers' mutex): both survive"""
    def setter(name, val, n):
        for _ in range(n):
            assert L.pq_set_option(name, val) == 0
    a = threading.Thread(target=setter, args=(b"PQ_NO_TAILSPLIT", b"1", 3000))
    b = threading.Thread(target=setter, args=(b"PQ_NO_MIDM", b"1", 3000))
    a.start(); b.start(); a.join(); b.join()
    assert L.pq_gemm_variant_name(4096, 6144, 4096, 4096, 4096) == b"sp256_16x16x64"          # PQ_NO_TAILSPLIT survived (a tail split otherwise)
    assert L.pq_qlinear_workspace_bytes(1024, 1024, 8192) > 0            # PQ_NO_MIDM survived the other thread's writes (0 = the 64 x 64 ring tiles otherwise)
    assert L.pq_set_option(b"PQ_NOT_A_SWITCH", b"1") != 0 and b"unknown option" in L.pq_last_error()


def _fsk_bytes(tiles, S):
    return ((tiles * 4 * (4 if S == 4 else 2) + 255) // 256) * 256 + tiles * (S - 1) * 256 * 256 * 4


def test_planner_refuses_fused_splitk_beyond_the_devices_cus(L):
    """cfg-3 `down` (128 tiles x 2 slices = 256 workgroups): planned on a 256-CU device; on a device that reports fewer CUs (CU-masked / partitioned:
    PQ_FAKE_CUS) the in-kernel hand-over is refused and the single-pass 128 x 256 tile runs."""
    L.pq_set_option(b"PQ_FAKE_CUS", b"256")
    assert L.pq_qlinear_workspace_bytes(2048, 4096, 11008) == _fsk_bytes(128, 2)
    # the 70B `down` shard: four slices in the ticket form again since round 5 (64 tiles x 4 = 256 workgroups; without a workspace the 128 x 128 ring tile runs)
    assert L.pq_qlinear_workspace_bytes(4096, 1024, 28672) == _fsk_bytes(64, 4) and L.pq_gemm_variant_name(4096, 1024, 28672, 28672, 28672) == b"ring128_16x16x64"
    assert L.pq_qlinear_workspace_bytes(2048, 1024, 28672) == 0 and L.pq_qlinear_workspace_bytes(4096, 1024, 8192) == 0
    L.pq_set_option(b"PQ_FAKE_CUS", b"255")
    assert L.pq_qlinear_workspace_bytes(2048, 4096, 11008) == 0                              # -> 128 x 256 tiles, one pass
    L.pq_set_option(b"PQ_FAKE_CUS", b"304")
    assert L.pq_qlinear_workspace_bytes(2048, 4096, 11008) == _fsk_bytes(128, 2)
    # forced slice counts (experiments): the ticket form runs on any grid, the symmetric forms only when every workgroup can be resident
    L.pq_set_option(b"PQ_FAKE_CUS", b"256")
    L.pq_set_option(b"PQ_FSK", b"2")
    assert L.pq_qlinear_workspace_bytes(4096, 4096, 4096) == _fsk_bytes(256, 2)
    L.pq_set_option(b"PQ_FSK_SYMMETRIC", b"1")
    assert L.pq_qlinear_workspace_bytes(4096, 4096, 4096) == 0
    assert L.pq_qlinear_workspace_bytes(2048, 4096, 4096) == _fsk_bytes(128, 2)
    L.pq_set_option(b"PQ_FSK", b"4")
    assert L.pq_qlinear_workspace_bytes(4096, 1024, 28672) == _fsk_bytes(64, 4)              # forced: 64 tiles x 4 slices = 256 workgroups fit
    L.pq_set_option(b"PQ_FAKE_CUS", b"255")
    assert L.pq_qlinear_workspace_bytes(4096, 1024, 28672) == 0                              # ... and do not on a 255-CU device (symmetric form)


def test_planner_follows_the_devices_cu_count(L):
    """round 5: every "fills the chip" threshold of the planner (tile kind, tail split, split-K forms) is a share of the CUs the device REPORTS, not of a literal 256:
    on a whole MI355X (256) the plans are round 4's; a partitioned or CU-masked device (PQ_FAKE_CUS = 128 / 64 / 32) gets plans made for it."""
    name = lambda M, N, K: L.pq_gemm_variant_name(M, N, K, K, K).decode()      # noqa: E731
    L.pq_set_option(b"PQ_FAKE_CUS", b"256")
    full = {s: name(*s) for s in ((2048, 11008, 4096), (4096, 1024, 8192), (4096, 2048, 4096), (512, 4096, 4096), (4096, 14336, 4096))}
    assert full == {(2048, 11008, 4096): "sp256_16x16x64 + sp128 tail (N)", (4096, 1024, 8192): "ring128_16x16x64", (4096, 2048, 4096): "sp128x256_16x16x64",
                    (512, 4096, 4096): "ring64x128_16x16x64", (4096, 14336, 4096): "sp256_16x16x64 + sp128 tail (N)"}
    L.pq_set_option(b"PQ_FAKE_CUS", b"")
    assert {s: name(*s) for s in full} == full            # (no GPU here: the query falls back to 256; on the GPU box the real count — 256 — gives the same)
    # round 6: the 128 x 160 ring tile where 128 x 256 tiles fill <= 2/3 of the chip and 128 x 160 tiles make one round of it (the 70B fused-qkv shard); PQ_NO_RING160 restores round 5
    assert name(4096, 1280, 8192) == "ring128x160_16x16x64" and name(2048, 2560, 4096) == "ring128x160_16x16x64" and name(4096, 1280, 2048) == "sp128x256_16x16x64"
    L.pq_set_option(b"PQ_NO_RING160", b"1")
    assert name(4096, 1280, 8192) == "sp128x256_16x16x64"
    L.pq_set_option(b"PQ_NO_RING160", b"")
    L.pq_set_option(b"PQ_FAKE_CUS", b"128")
    assert name(4096, 1024, 8192) == "sp128x256_16x16x64" and name(4096, 2048, 4096) == "sp256_16x16x64" and name(512, 4096, 4096) == "ring128_16x16x64"
    assert name(2048, 11008, 4096) == "sp256_16x16x64"    # 344 tiles on 128 CUs: 2.7 rounds, no poorly filled tail worth a second launch
    assert L.pq_qlinear_workspace_bytes(2048, 4096, 11008) == 0          # 128 tiles fill this device: no K split
    L.pq_set_option(b"PQ_FAKE_CUS", b"64")
    assert name(4096, 1024, 8192) == "sp256_16x16x64" and name(1024, 1024, 4096) == "ring128_16x16x64"
    assert L.pq_qlinear_workspace_bytes(1024, 2048, 11264) == _fsk_bytes(32, 2)      # the half-filled grid of THIS device gets the fused split-K
    L.pq_set_option(b"PQ_FAKE_CUS", b"32")
    assert name(1024, 1024, 4096) == "sp128x256_16x16x64" and name(4096, 1280, 8192) == "sp256_16x16x64 + sp128 tail (N)"
    L.pq_set_option(b"PQ_FAKE_CUS", b"")


def test_kslabs_workspace_queries_are_consistent_over_random_shapes():
    """round 6, no GPU (the planner falls back to 256 CUs): for any shape and slab count the short query (always enough) is >= the exact one, the exact one is what the named way
    needs, and a workspace of the exact size never names a way that needs more."""
    import ctypes
    import random
    from protoquant_amd import _lib
    L = _lib.lib()
    rnd = random.Random(6)
    ways = set()
    for _ in range(3000):
        G = rnd.choice([1, 2, 3, 4, 8])
        tps = rnd.choice([1, 2, 3, 4, 7, 8, 14, 28, 43, 56])
        kps = tps * 128 if rnd.random() < 0.9 else tps * 128 + rnd.choice([16, 64, 100])
        K = G * kps
        M = rnd.choice([1, 16, 64, 65, 130, 300, 512, 1024, 2048, 4096, 8192])
        N = rnd.choice([96, 128, 130, 257, 512, 1024, 1280, 2560, 3584, 4096, 7168, 14336])
        lda = kps + rnd.choice([0, 0, 16, 128, 3])
        stride = M * lda + rnd.choice([0, 0, 256, 48, 5])
        a = ctypes.c_void_p(4096 + rnd.choice([0, 0, 0, 8]))
        b = ctypes.c_void_p(8192)
        short = L.pq_qlinear_kslabs_workspace_bytes(M, N, K, kps)
        exact = L.pq_qlinear_kslabs_workspace_bytes_for(a, lda, stride, kps, b, K, M, N, K)
        way = L.pq_kslabs_way_name(a, lda, stride, kps, b, K, M, N, K, exact).decode()
        ways.add(way.split(" x")[0])
        assert short >= exact, (M, N, K, G, lda, stride, short, exact, way)
        if way.startswith("in place: ring"):
            assert exact == 0, (way, exact)
        elif way == "layout pass":
            assert exact >= M * K, (way, exact, M, K)
        elif "fused split-K" in way:
            assert 0 < exact < short and G > 1 and kps % 128 == 0 and kps >= 512 and (lda % 16 == 0) and (stride % 16 == 0) and a.value % 16 == 0, (way, M, N, K, G, lda, stride)
            # without a workspace the same operands take another way — never an error
            assert "fused split-K" not in L.pq_kslabs_way_name(a, lda, stride, kps, b, K, M, N, K, 0).decode()
    assert {"in place: ring128", "layout pass", "in place: fused split-K", "one slab: pq_qlinear_s8"} <= ways, ways
