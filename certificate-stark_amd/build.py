"""Builds the HIP backend: every csrc/*.hip -> one shared library next to this file.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting libcstark_hip.so travels with the tree to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libcstark_hip.so")
# Test-only companion: element-wise entry points for the device field arithmetic (csrc/debug/, declared in
# include/cstark_debug.h).  Not part of the product library; the GPU parity tests and tools/modmul_bench.py load it.
DEBUG_LIB = os.path.join(HERE, "libcstark_debug.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-const-variable", "-Wno-unused-variable",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]  # MFMA results land in VGPRs: the recombination reads them without v_accvgpr_read


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP backend cannot be built (there is no CPU fallback)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Tuning helper: compile the whole library with extra -D flags into libcstark_hip_<name>.so (not used by the product)."""
    out = os.path.join(HERE, "libcstark_hip_%s.so" % name)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    cmd = [hipcc()] + FLAGS + ["-D" + d for d in defines] + ["-shared", "-o", out] + srcs
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cuh"))]
    headers.append(os.path.join(HERE, "..", "include", "cstark.h"))
    headers.append(os.path.abspath(__file__))
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [os.path.join(CSRC, s)] + headers):
            cmd = [hipcc()] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    dbg_srcs = [os.path.join(CSRC, "debug", f) for f in sorted(os.listdir(os.path.join(CSRC, "debug"))) if f.endswith(".hip")]
    dbg = None
    if force or _stale(DEBUG_LIB, dbg_srcs + headers):
        cmd = [hipcc()] + FLAGS + ["-shared", "-o", DEBUG_LIB] + dbg_srcs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        dbg = subprocess.Popen(cmd)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if dbg is not None and dbg.wait() != 0:
        raise RuntimeError("hipcc failed for the debug library")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
