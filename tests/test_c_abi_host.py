"""GPU: a host program with no Python and no torch (tests/c_abi_smoke.cpp) drives the hot path through include/pq_hip.h alone and
matches the plain-C oracle bit for bit — the drop-in boundary is the C-ABI, not the Python wrapper."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_plain_c_host_through_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    exe = str(tmp_path / "c_abi_smoke")
    pkg, orc = os.path.join(ROOT, "protoquant_amd"), os.path.join(ROOT, "oracle")
    build = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.cpp"),
                            "-o", exe, "-L", pkg, "-lpq_hip", "-L", orc, "-loracle", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}", "-fopenmp"],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.count("bit-identical to the oracle") == 3, run.stdout
