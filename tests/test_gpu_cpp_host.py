"""The C++ host mirror (include/cstark.hpp): a program written against it is compiled with g++, run on the GPU, and its proof is
checked by the restated verifier and against the golden proof (same seed, same options -> same bytes as the Python host)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "certificate-stark_amd")


def build_program(tmp):
    exe = os.path.join(tmp, "host_mirror")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"),
                           "-o", exe, "-pthread", "-L", PKG, "-lcstark_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.gpu
def test_cpp_host_mirror_proves_and_matches_python_host(tmp_path):
    from oracle import verifier as V
    exe = build_program(str(tmp_path))
    prefix = str(tmp_path / "out")
    res = subprocess.run([exe, prefix], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "refused=1 length_checked=1" in res.stdout
    proof = open(prefix + ".proof", "rb").read()
    pub = np.fromfile(prefix + ".pub", np.uint64)
    assert V.verify(proof, pub[:7], pub[7:], options=[42, 8, 0, 0, 0, 4, 256])
    golden = np.load(os.path.join(ROOT, "tests", "golden", "proof_2tx_d3.npz"))
    assert proof == golden["proof"].tobytes()
    # cstark::RescueExample(16, blowup 4): the bytes of the CPU prover, accepted by the restated verifier
    from oracle import oracle as O
    from oracle import prover as OP
    seed = O.to_mont(np.arange(42, 49, dtype=np.uint64))
    rp = open(prefix + ".rescue", "rb").read()
    assert rp == OP.prove_air(O.AIR_RESCUE_CHAIN, (seed, 16), (42, 4, 0, 0, 0, 4, 256))
    assert V.verify_rescue(rp, seed, O.rescue_chain_build_trace(seed, 16)[:7, -1].copy())


def test_cpp_host_mirror_compiles():
    """CPU check: the header is self-contained C++17 and links against the library."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        assert os.path.exists(build_program(tmp))
