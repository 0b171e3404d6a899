"""Raw Montgomery-product rate of the device (no memory traffic) as a function of occupancy and instruction-level parallelism:
blocks of 256 threads (4 waves, one per SIMD) x `ilp` independent dependent-product chains per lane.  waves/SIMD = blocks / 256."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd import _lib
from certificate_stark_amd.backend import Backend

b = Backend()
iters = 4096
out = b.empty_u64(256 * 8 * 256)
print("waves/SIMD  ilp   ms      Tmodmul/s   SIMD-cycles per wave-product at 2.3 GHz")
for wps in (1, 2, 3, 4, 8):
    blocks = 256 * wps
    for ilp in (1, 2, 4, 8):
        ms = C.c_float()
        rc = _lib.load_debug().cstark_debug_modmul_bench(C.c_void_p(b.stream.cuda_stream), b._ptr(out), blocks, iters, ilp, C.byref(ms))
        assert rc == 0
        n = blocks * 256 * iters * ilp
        rate = n / ms.value / 1e9
        print("%9d %4d %8.3f %9.3f %12.1f" % (wps, ilp, ms.value, rate, 1024 * 64 * 2.3e9 / (rate * 1e12)))
