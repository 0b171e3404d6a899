"""Measures the raw Montgomery-product rate of the device (no memory traffic) at several ILP levels."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd import _lib
from certificate_stark_amd.backend import Backend

b = Backend()
blocks, iters = 256 * 8, 4096
out = b.empty_u64(blocks * 256)
for ilp in (1, 2, 4, 8):
    ms = C.c_float()
    rc = _lib.load_debug().cstark_debug_modmul_bench(C.c_void_p(b.stream.cuda_stream), b._ptr(out), blocks, iters, ilp, C.byref(ms))
    assert rc == 0
    n = blocks * 256 * iters * ilp
    print("ilp=%d: %.3f ms, %.3f Tmodmul/s" % (ilp, ms.value, n / ms.value / 1e9))
