"""Experiment: INV_MDS * vector on the matrix cores (int8 MFMA, byte-diagonal GEMM) vs the VALU dot products: exactness and rate."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd import _lib
from certificate_stark_amd.backend import Backend, to_numpy_u64
P = 2**62 + 2**56 + 2**55 + 1
b = Backend()
npts = 1 << 20
rng = np.random.default_rng(5)
x = rng.integers(0, P, size=(14, npts), dtype=np.uint64)
x[:, 0] = 0; x[:, 1] = P - 1; x[:, 2] = 0x0080808080808080; x[:, 3] = 0x417fffffffffffff
d_in = b.from_numpy_u64(x)
outs = []
for use in (0, 1):
    d_out = b.empty_u64(14, npts)
    ms = C.c_float()
    rc = _lib.load_debug().cstark_debug_mds(C.c_void_p(b.stream.cuda_stream), b._ptr(d_in), b._ptr(d_out), C.c_size_t(npts), use, 1, C.byref(ms))
    assert rc == 0, rc
    outs.append(to_numpy_u64(d_out))
print("exact:", bool((outs[0] == outs[1]).all()), "mismatches:", int((outs[0] != outs[1]).sum()))
if not (outs[0] == outs[1]).all():
    bad = np.argwhere(outs[0] != outs[1])
    print(bad[:10], outs[0][tuple(bad[0])], outs[1][tuple(bad[0])])
for use in (0, 1):
    for iters in (1, 16):
        d_out = b.empty_u64(14, npts)
        ms = C.c_float()
        _lib.load_debug().cstark_debug_mds(C.c_void_p(b.stream.cuda_stream), b._ptr(d_in), b._ptr(d_out), C.c_size_t(npts), use, iters, C.byref(ms))
        print("%s iters=%2d: %.3f ms  -> %.1f ns per matvec-point" % ("MFMA" if use else "VALU", iters, ms.value, ms.value * 1e6 / npts / iters))
