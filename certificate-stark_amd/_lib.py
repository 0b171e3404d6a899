"""ctypes binding of the C ABI declared in include/cstark.h (libcstark_hip.so).

The library is the product; there is no CPU fallback.  Loading fails loudly when the shared object
has not been built, and every compute call fails with CSTARK_ERR_NO_DEVICE when no GPU is visible.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSTARK_LIB") or os.path.join(HERE, "libcstark_hip.so")  # CSTARK_LIB: tuning variants

TX_TRACE_WIDTH, TX_CYCLE_LENGTH, TX_NUM_CONSTRAINTS, TX_NUM_PERIODIC = 94, 1024, 115, 48

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class CstarkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cstark error %d: %s" % (code, msg))
        self.code = code


class TxWitnessStruct(C.Structure):
    _fields_ = [("n_tx", C.c_uint32), ("merkle_depth", C.c_uint32),
                ("initial_roots", u64p), ("final_root", u64p), ("s_old_values", u64p), ("r_old_values", u64p),
                ("s_indices", u64p), ("r_indices", u64p), ("s_paths", u64p), ("r_paths", u64p),
                ("deltas", u64p), ("sig_rx", u64p), ("sig_s", u8p)]


class TxCoeffsStruct(C.Structure):
    _fields_ = [("t_alpha", C.c_uint64 * TX_NUM_CONSTRAINTS), ("t_beta", C.c_uint64 * TX_NUM_CONSTRAINTS),
                ("b_alpha", C.c_uint64 * 4), ("b_beta", C.c_uint64 * 4)]


class OptionsStruct(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_queries", "blowup_factor", "grinding_factor", "hash_fn",
                                           "field_extension", "fri_folding_factor", "fri_max_remainder")]


_lib = None


def load():
    """Returns the loaded library; raises if it has not been built (python -m certificate_stark_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CstarkError(-2, "libcstark_hip.so is missing at %s: build it with __graft_entry__.build() "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch first: its wheel bundles its own HIP runtime, and the process must end up with ONE libamdhip64 -- if this
        # library pulled in the system copy before torch loaded, device discovery fails later ("no HIP device visible")
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.cstark_last_error.restype = C.c_char_p
        lib.cstark_version.restype = C.c_char_p
        _lib = lib
    return _lib


_dbg = None


def load_debug():
    """The test-only companion library (include/cstark_debug.h): element-wise field / tower operations and micro-benchmarks."""
    global _dbg
    if _dbg is None:
        load()
        path = os.path.join(HERE, "libcstark_debug.so")
        if not os.path.exists(path):
            raise CstarkError(-2, "libcstark_debug.so is missing at %s: build it with __graft_entry__.build()" % path)
        _dbg = C.CDLL(path)
    return _dbg


def check(rc):
    if rc != 0:
        raise CstarkError(rc, load().cstark_last_error().decode())
