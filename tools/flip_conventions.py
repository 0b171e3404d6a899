"""Flip engine conventions (include/cstark_conventions.h) in the product AND the oracle at once, and check that the two still agree.

Every CSTARK_CONV_* value can be overridden with -D.  This tool builds libcstark_hip_flip.so (product) and
oracle/_build/libcs_oracle_flip.so (oracle) with the same overrides; on a GPU box, `--test` runs the GPU-against-oracle parity tests
that do not depend on committed fixtures (those were written under the default conventions) with both libraries selected through
CSTARK_LIB / CS_ORACLE_LIB.  Green = the convention is a single switch that moves both sides together, which is what a maintainer
needs after diffing one real proof of the Rust engine against this library.

  build (CPU container):  python tools/flip_conventions.py --build CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY=0 CSTARK_CONV_COIN_REJECT_ABOVE_P=0
  test  (GPU box):        python tools/flip_conventions.py --test
"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PRODUCT = os.path.join(ROOT, "certificate-stark_amd", "libcstark_hip_flip.so")
ORACLE = os.path.join(ROOT, "oracle", "_build", "libcs_oracle_flip.so")
TESTS = ["tests/test_gpu_prove.py::test_proof_bytes_equal_the_cpu_restatement", "tests/test_gpu_prove.py::test_one_context_across_options_reuses_its_arena_safely",
         "tests/test_gpu_prove.py::test_transaction_basic_proof_verification", "tests/test_gpu_prove.py::test_transaction_basic_proof_verification_fail",
         "tests/test_gpu_commit.py", "tests/test_gpu_composition.py", "tests/test_gpu_prove_small_airs.py", "tests/test_gpu_baseline_configs.py::test_long_range_proof_verifies",
         "tests/test_gpu_sharding.py::test_sharded_proof_equals_single_gpu_proof",
         # round 4: the device-side channel (every TransactionAir proof of these files with the Blake3 coin and no proof of work), the other blowup /
         # folding factors, RescueAir
         "tests/test_gpu_prove.py::test_host_and_device_channel_give_the_same_bytes", "tests/test_gpu_options.py::test_transaction_proof_bytes_over_options",
         "tests/test_gpu_options.py::test_merkle_proof_bytes_over_options", "tests/test_gpu_options.py::test_range_proof_bytes_over_options",
         "tests/test_gpu_options.py::test_schnorr_proof_bytes_over_options", "tests/test_gpu_rescue_chain.py::test_proof_bytes_and_verification"]


def build(defs):
    B = importlib.import_module("certificate_stark_amd.build")
    obj = os.path.join(B.HERE, "_obj_flip")
    os.makedirs(obj, exist_ok=True)
    procs, objs = [], []
    for f in sorted(os.listdir(B.CSRC)):
        if f.endswith(".hip"):
            o = os.path.join(obj, f[:-4] + ".o")
            objs.append(o)
            procs.append(subprocess.Popen([B.hipcc()] + B.FLAGS + ["-D" + d for d in defs] + ["-c", os.path.join(B.CSRC, f), "-o", o]))
    if any(p.wait() for p in procs):
        raise SystemExit("hipcc failed")
    subprocess.check_call([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", PRODUCT] + objs)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "OUT=" + ORACLE, "EXTRA_CFLAGS=" + " ".join("-D" + d for d in defs)])
    print("built", PRODUCT, "and", ORACLE, "with", defs)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build([a for a in sys.argv[1:] if not a.startswith("--")])
    if "--test" in sys.argv:
        env = dict(os.environ, CSTARK_LIB=PRODUCT, CS_ORACLE_LIB=ORACLE)
        sys.exit(subprocess.call([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + TESTS, cwd=ROOT, env=env))
