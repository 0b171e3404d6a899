"""ORACLE (test infrastructure, not product code): ctypes bindings of oracle/_build/libcs_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Field elements are numpy uint64 arrays in f63::BaseElement memory form (Montgomery, R = 2^64).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CS_ORACLE_LIB") or os.path.join(_HERE, "_build", "libcs_oracle.so")  # CS_ORACLE_LIB: tools/flip_conventions.sh

P = 2**62 + 2**56 + 2**55 + 1
R = 2**64
TX_W, TX_CYCLE, TX_NC, TX_NP = 94, 1024, 115, 48

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)


class TxWitnessStruct(C.Structure):
    """struct cstark_tx_witness (include/cstark.h)."""
    _fields_ = [("n_tx", C.c_uint32), ("merkle_depth", C.c_uint32),
                ("initial_roots", u64p), ("final_root", u64p), ("s_old_values", u64p), ("r_old_values", u64p),
                ("s_indices", u64p), ("r_indices", u64p), ("s_paths", u64p), ("r_paths", u64p),
                ("deltas", u64p), ("sig_rx", u64p), ("sig_s", u8p)]


class TxCoeffsStruct(C.Structure):
    """struct cstark_tx_coeffs (include/cstark.h)."""
    _fields_ = [("t_alpha", C.c_uint64 * TX_NC), ("t_beta", C.c_uint64 * TX_NC),
                ("b_alpha", C.c_uint64 * 4), ("b_beta", C.c_uint64 * 4)]


def build(force=False):
    if os.environ.get("CS_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(f) > os.path.getmtime(_SO)
            for f in [os.path.join(_HERE, g) for g in os.listdir(_HERE) if g.endswith((".c", ".h"))] +
            [os.path.join(_HERE, "..", "include", "cstark_conventions.h")]):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        # The test-sized calls are tiny parallel regions: on a many-core host (the GPU boxes expose 128-256 hardware threads, shared)
        # OpenMP's default team turns every one of them into a long spin.  Default to a modest team; OMP_NUM_THREADS or
        # set_num_threads() (bench.py's cpu_baseline leg: all cores for the one big proof) override it.
        if "OMP_NUM_THREADS" not in os.environ:
            _lib.cso_set_num_threads(C.c_int(min(16, os.cpu_count() or 1)))
        _lib.cso_fp_root_of_unity.restype = C.c_uint64
        _lib.cso_fp_generator.restype = C.c_uint64
        _lib.cso_poly_eval.restype = C.c_uint64
        _lib.cso_tx_combined_at.restype = C.c_uint64
        _lib.cso_tx_check_trace.restype = C.c_long
    return _lib


def _p(a, t=u64p):
    return a.ctypes.data_as(t)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- field helpers --------------------------------------------------------------------------------
def to_mont(x):
    x = _u64(x); out = np.empty_like(x); lib().cso_fp_from_u64(_p(x), _p(out), C.c_size_t(x.size)); return out


def from_mont(x):
    x = _u64(x); out = np.empty_like(x); lib().cso_fp_to_u64(_p(x), _p(out), C.c_size_t(x.size)); return out


def _bin(name, a, b):
    a, b = _u64(a), _u64(b); out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(out), C.c_size_t(a.size)); return out


def fp_mul(a, b): return _bin("cso_fp_mul", a, b)
def fp_add(a, b): return _bin("cso_fp_add", a, b)
def fp_sub(a, b): return _bin("cso_fp_sub", a, b)


def fp_inv(a):
    a = _u64(a); out = np.empty_like(a); lib().cso_fp_inv(_p(a), _p(out), C.c_size_t(a.size)); return out


def fp_pow(a, e):
    a = _u64(a); out = np.empty_like(a); lib().cso_fp_pow(_p(a), C.c_uint64(e), _p(out), C.c_size_t(a.size)); return out


def root_of_unity(log_n):
    return int(lib().cso_fp_root_of_unity(C.c_uint(log_n)))


# ---- witness --------------------------------------------------------------------------------------
class TxWitness:
    """Owns the arrays of a cstark_tx_witness (TransactionMetadata, src/lib.rs:183-194)."""
    FIELDS = ("initial_roots", "final_root", "s_old_values", "r_old_values", "s_indices", "r_indices",
              "s_paths", "r_paths", "deltas", "sig_rx", "sig_s")

    def __init__(self, n_tx, depth=15):
        self.n_tx, self.depth = n_tx, depth
        self.initial_roots = np.zeros((n_tx, 7), np.uint64)
        self.final_root = np.zeros(7, np.uint64)
        self.s_old_values = np.zeros((n_tx, 14), np.uint64)
        self.r_old_values = np.zeros((n_tx, 14), np.uint64)
        self.s_indices = np.zeros(n_tx, np.uint64)
        self.r_indices = np.zeros(n_tx, np.uint64)
        self.s_paths = np.zeros((n_tx, depth + 1, 7), np.uint64)
        self.r_paths = np.zeros((n_tx, depth + 1, 7), np.uint64)
        self.deltas = np.zeros(n_tx, np.uint64)
        self.sig_rx = np.zeros((n_tx, 6), np.uint64)
        self.sig_s = np.zeros((n_tx, 32), np.uint8)

    def struct(self):
        s = TxWitnessStruct()
        s.n_tx, s.merkle_depth = self.n_tx, self.depth
        for f in self.FIELDS:
            a = getattr(self, f)
            assert a.flags["C_CONTIGUOUS"]
            setattr(s, f, _p(a, u8p if f == "sig_s" else u64p))
        return s

    @classmethod
    def generate(cls, n_tx, depth=15, seed=0x5EED):
        w = cls(n_tx, depth)
        s = w.struct()
        rc = lib().cso_tx_witness_generate(C.byref(s), C.c_uint64(seed))
        if rc:
            raise RuntimeError("cso_tx_witness_generate failed: %d" % rc)
        return w

    def copy(self):
        w = TxWitness(self.n_tx, self.depth)
        for f in self.FIELDS:
            getattr(w, f)[...] = getattr(self, f)
        return w

    def save(self, path):
        np.savez_compressed(path, n_tx=self.n_tx, depth=self.depth, **{f: getattr(self, f) for f in self.FIELDS})

    @classmethod
    def load(cls, path):
        z = np.load(path)
        w = cls(int(z["n_tx"]), int(z["depth"]))
        for f in cls.FIELDS:
            getattr(w, f)[...] = z[f]
        return w


# ---- state-transition AIR -------------------------------------------------------------------------
def tx_build_trace(w):
    n = w.n_tx * TX_CYCLE
    trace = np.zeros((TX_W, n), np.uint64)
    s = w.struct()
    rc = lib().cso_tx_build_trace(C.byref(s), _p(trace))
    if rc:
        raise RuntimeError("cso_tx_build_trace failed: %d" % rc)
    return trace


def tx_periodic_columns(depth=15):
    out = np.zeros((TX_NP, TX_CYCLE), np.uint64)
    rc = lib().cso_tx_periodic_columns(C.c_uint(depth), _p(out))
    if rc:
        raise RuntimeError("cso_tx_periodic_columns failed: %d" % rc)
    return out


def tx_evaluate_transition(cur, nxt, periodic):
    cur, nxt, periodic = _u64(cur), _u64(nxt), _u64(periodic)
    out = np.zeros(TX_NC, np.uint64)
    lib().cso_tx_evaluate_transition(_p(cur), _p(nxt), _p(periodic), _p(out))
    return out


def tx_check_trace(trace, n_tx, depth=15):
    trace = _u64(trace)
    return int(lib().cso_tx_check_trace(_p(trace), C.c_uint32(n_tx), C.c_uint(depth)))


def tx_constraint_degrees():
    base = np.zeros(TX_NC, np.uint32); cyc = np.zeros(TX_NC, np.uint32)
    lib().cso_tx_constraint_degrees(_p(base, u32p), _p(cyc, u32p))
    return base, cyc


# ---- engine stages ----------------------------------------------------------------------------------
def generator():
    return int(lib().cso_fp_generator())


def ntt(a, inverse=False):
    a = _u64(a).copy()
    log_n = a.size.bit_length() - 1
    assert a.size == 1 << log_n
    (lib().cso_intt if inverse else lib().cso_ntt)(_p(a), C.c_uint(log_n))
    return a


def dft_naive(a):
    a = _u64(a); out = np.empty_like(a)
    lib().cso_dft_naive(_p(a), _p(out), C.c_uint(a.size.bit_length() - 1)); return out


def interpolate_columns(cols):
    cols = _u64(cols).copy()
    width, n = cols.shape
    lib().cso_interpolate_columns(_p(cols), C.c_uint32(width), C.c_uint(n.bit_length() - 1))
    return cols


def lde_columns(coeffs, log_b, offset=None, k0=0, nk=None):
    coeffs = _u64(coeffs)
    width, n = coeffs.shape
    nk = (1 << log_b) - k0 if nk is None else nk
    out = np.zeros((nk, width, n), np.uint64)
    lib().cso_lde_columns(_p(coeffs), _p(out), C.c_uint32(width), C.c_uint(n.bit_length() - 1), C.c_uint(log_b),
                          C.c_uint64(generator() if offset is None else offset), C.c_uint32(k0), C.c_uint32(nk))
    return out


def blake3(data):
    data = bytes(data)
    buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().cso_blake3(_p(np.ascontiguousarray(buf), u8p), C.c_size_t(len(data)), _p(out, u8p))
    return out.tobytes()


def digest(data, hash_fn=0):
    """hash_fn 0 = Blake3_256, 1 = Sha3_256 (winterfell::HashFunction)"""
    data = bytes(data)
    buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().cso_digest(C.c_int(hash_fn), _p(np.ascontiguousarray(buf), u8p), C.c_size_t(len(data)), _p(out, u8p))
    return out.tobytes()


def sha3_256(data):
    return digest(data, 1)


def hash_rows(lde, log_b, k0=0, hash_fn=0):
    lde = _u64(lde)
    nk, width, n = lde.shape
    leaves = np.zeros((n << log_b, 32), np.uint8)
    lib().cso_hash_rows_fn(C.c_int(hash_fn), _p(lde), _p(leaves, u8p), C.c_uint32(width), C.c_uint(n.bit_length() - 1), C.c_uint(log_b),
                           C.c_uint32(k0), C.c_uint32(nk))
    return leaves


def merkle_build(leaves, hash_fn=0):
    leaves = np.ascontiguousarray(leaves, np.uint8)
    L = leaves.shape[0]
    nodes = np.zeros((2 * L, 32), np.uint8)
    nodes[L:] = leaves
    lib().cso_merkle_build_fn(C.c_int(hash_fn), _p(nodes, u8p), C.c_uint(L.bit_length() - 1))
    return nodes


def tx_periodic_table(depth, log_n, log_b):
    out = np.zeros((1 << log_b, TX_NP, 1024), np.uint64)
    lib().cso_tx_periodic_table(C.c_uint(depth), C.c_uint(log_n), C.c_uint(log_b), _p(out))
    return out


def tx_degree_adjustments(log_n, log_b):
    out = np.zeros(TX_NC, np.uint64)
    lib().cso_tx_degree_adjustments(C.c_uint(log_n), C.c_uint(log_b), _p(out)); return out


def make_coeffs(seed=1):
    """Pseudo-random composition coefficients (stand-in for the public-coin draw)."""
    rng = np.random.default_rng(seed)
    draw = lambda k: to_mont(rng.integers(0, P, size=k, dtype=np.uint64))
    cf = TxCoeffsStruct()
    for name, k in (("t_alpha", TX_NC), ("t_beta", TX_NC), ("b_alpha", 4), ("b_beta", 4)):
        v = draw(k)
        for i in range(k):
            getattr(cf, name)[i] = int(v[i])
    return cf


def tx_evaluate_transitions(lde, depth, log_b, k0=0):
    lde = _u64(lde)
    nk, width, n = lde.shape
    assert width == TX_W
    out = np.zeros((nk, TX_NC, n), np.uint64)
    lib().cso_tx_evaluate_transitions(_p(lde), _p(out), C.c_uint(depth), C.c_uint(n.bit_length() - 1), C.c_uint(log_b),
                                      C.c_uint32(k0), C.c_uint32(nk))
    return out


def tx_evaluate_constraints(lde, coeffs, pub_inputs, depth, log_b, k0=0):
    lde = _u64(lde)
    nk, width, n = lde.shape
    pub = _u64(pub_inputs)
    out = np.zeros((nk, n), np.uint64)
    lib().cso_tx_evaluate_constraints(_p(lde), C.byref(coeffs), _p(pub), _p(out), C.c_uint(depth),
                                      C.c_uint(n.bit_length() - 1), C.c_uint(log_b), C.c_uint32(k0), C.c_uint32(nk))
    return out


def poly_eval(coeffs, x):
    coeffs = _u64(coeffs)
    return int(lib().cso_poly_eval(_p(coeffs), C.c_size_t(coeffs.size), C.c_uint64(x)))


def tx_combined_at(trace_coeffs, coeffs, pub_inputs, depth, log_b, z):
    trace_coeffs = _u64(trace_coeffs)
    width, n = trace_coeffs.shape
    pub = _u64(pub_inputs)
    return int(lib().cso_tx_combined_at(_p(trace_coeffs), C.byref(coeffs), _p(pub), C.c_uint(depth),
                                        C.c_uint(n.bit_length() - 1), C.c_uint(log_b), C.c_uint64(z)))


def set_num_threads(n):
    lib().cso_set_num_threads(C.c_int(int(n)))


def num_threads():
    """OpenMP threads the oracle actually uses (cpu_baseline.cores)."""
    return int(lib().cso_num_threads())


# ---- standalone sub-AIRs (air_small.c) ---------------------------------------------------------------
AIR_STATE_TRANSITION, AIR_MERKLE, AIR_SCHNORR, AIR_RANGE, AIR_RESCUE_CHAIN = 0, 1, 2, 3, 4


class AirDescStruct(C.Structure):
    _fields_ = [("width", C.c_uint32), ("n_constraints", C.c_uint32), ("cycle_len", C.c_uint32), ("log_ce_blowup", C.c_uint32),
                ("base", u32p), ("cycles", u32p), ("n_assertions", C.c_uint32), ("a_reg", u32p), ("a_last", u32p), ("a_value", u64p),
                ("a_first", u32p), ("a_stride", u32p), ("a_seq", C.POINTER(C.c_int32))]


class AirDesc:
    """Python-side holder of a cso_air_desc."""

    def __init__(self, width, base, cycles, cycle_len, a_reg, a_last, a_value, a_first=None, a_stride=None, a_seq=None):
        self.width, self.cycle_len = width, cycle_len
        self.a_first = None if a_first is None else np.ascontiguousarray(a_first, np.uint32)
        self.a_stride = None if a_stride is None else np.ascontiguousarray(a_stride, np.uint32)
        self.a_seq = None if a_seq is None else np.ascontiguousarray(a_seq, np.int32)
        self.base = np.ascontiguousarray(base, np.uint32)
        self.cycles = np.ascontiguousarray(cycles, np.uint32)
        self.a_reg = np.ascontiguousarray(a_reg, np.uint32)
        self.a_last = np.ascontiguousarray(a_last, np.uint32)
        self.a_value = np.ascontiguousarray(a_value, np.uint64)
        self.nc, self.na = len(self.base), len(self.a_reg)
        m = max(int(b) + (int(c) if cycle_len else 0) for b, c in zip(self.base, self.cycles))
        self.log_ce = max(1, (m - 1).bit_length())  # next power of two >= max degree, at least 2

    def struct(self):
        s = AirDescStruct()
        s.width, s.n_constraints, s.cycle_len, s.log_ce_blowup = self.width, self.nc, self.cycle_len, self.log_ce
        s.base, s.cycles = _p(self.base, u32p), _p(self.cycles, u32p)
        s.n_assertions, s.a_reg, s.a_last, s.a_value = self.na, _p(self.a_reg, u32p), _p(self.a_last, u32p), _p(self.a_value)
        if self.a_stride is not None:
            s.a_first, s.a_stride, s.a_seq = _p(self.a_first, u32p), _p(self.a_stride, u32p), _p(self.a_seq, C.POINTER(C.c_int32))
        return s


def merkle_build_trace(w):
    trace = np.zeros((65, w.n_tx * 512), np.uint64)
    s = w.struct()
    if lib().cso_merkle_build_trace(C.byref(s), _p(trace)):
        raise RuntimeError("cso_merkle_build_trace failed")
    return trace


def merkle_periodic_columns(depth):
    out = np.zeros((33, 512), np.uint64)
    if lib().cso_merkle_periodic_columns(C.c_uint(depth), _p(out)):
        raise RuntimeError("bad depth")
    return out


def merkle_desc(trace):
    base = np.zeros(106, np.uint32); cyc = np.zeros(106, np.uint32)
    lib().cso_merkle_constraint_degrees(_p(base, u32p), _p(cyc, u32p))
    regs = list(range(58, 65)) * 2
    return AirDesc(65, base, cyc, 512, regs, [0] * 7 + [1] * 7, np.concatenate([trace[58:65, 0], trace[58:65, -1]]))


def range_build_trace(number):
    trace = np.zeros((2, 64), np.uint64)
    lib().cso_range_build_trace(C.c_uint64(number), _p(trace))
    return trace


def range_build_trace_bits(words, log_n):
    """Synthetic long accumulator (BASELINE 'range, 2^16 steps'): words = n/64 little-endian uint64 of an (n-1)-bit integer.
    Returns (trace [2][n], V mod p in memory form)."""
    words = _u64(words)
    n = 1 << log_n
    assert words.size == n // 64 and not (int(words[-1]) >> 63)
    trace = np.zeros((2, n), np.uint64)
    lib().cso_range_build_trace_bits.restype = C.c_uint64
    v = lib().cso_range_build_trace_bits(_p(words), C.c_uint32(log_n), _p(trace))
    return trace, int(v)


def range_desc(number):
    return AirDesc(2, [2, 1], [0, 0], 0, [1, 1], [0, 1], [0, int(to_mont([number])[0])])


def rescue_chain_build_trace(seed, iterations):
    seed = _u64(seed)
    trace = np.zeros((14, 8 * iterations), np.uint64)
    lib().cso_rescue_chain_build_trace(_p(seed), C.c_uint32(iterations), _p(trace))
    return trace


def rescue_chain_periodic_columns():
    out = np.zeros((29, 8), np.uint64)
    lib().cso_rescue_chain_periodic_columns(_p(out))
    return out


def rescue_chain_desc(trace):
    return AirDesc(14, [3] * 14, [1] * 14, 8, list(range(7)) * 2, [0] * 7 + [1] * 7, np.concatenate([trace[:7, 0], trace[:7, -1]]))


def rescue_compute_hash_chain(seed, length):
    seed = _u64(seed); out = np.zeros(7, np.uint64)
    lib().cso_rescue_compute_hash_chain(_p(seed), C.c_uint32(length), _p(out))
    return out


def periodic_table(cols, log_n, log_b):
    cols = _u64(cols)
    np_, C_ = cols.shape
    out = np.zeros((1 << log_b, np_, C_), np.uint64)
    lib().cso_periodic_table(_p(cols), C.c_uint32(np_), C.c_uint(C_.bit_length() - 1), C.c_uint(log_n), C.c_uint(log_b), _p(out))
    return out


def air_evaluate_transitions(air, lde, ptab, nc, k0=0):
    lde = _u64(lde)
    nk, width, n = lde.shape
    if ptab is None:
        np_, cl, pp = 0, 1, None
    else:
        ptab = _u64(ptab); np_, cl, pp = ptab.shape[1], ptab.shape[2], _p(ptab)
    out = np.zeros((nk, nc, n), np.uint64)
    lib().cso_air_evaluate_transitions(C.c_int(air), _p(lde), pp, _p(out), C.c_uint32(width), C.c_uint32(nc), C.c_uint32(np_),
                                       C.c_uint32(cl), C.c_uint(n.bit_length() - 1), C.c_uint32(k0), C.c_uint32(nk))
    return out


def air_combine(desc, lde, evals, t_alpha, t_beta, b_alpha, b_beta, log_b, k0=0, all_cosets=False, avals=None):
    lde, evals = _u64(lde), _u64(evals)
    nk, width, n = lde.shape
    out = np.zeros((nk, n), np.uint64)
    s = desc.struct()
    if avals is not None:
        avals = _u64(avals)
    lib().cso_air_combine(C.byref(s), _p(lde), _p(evals), _p(_u64(t_alpha)), _p(_u64(t_beta)), _p(_u64(b_alpha)), _p(_u64(b_beta)),
                          _p(out), C.c_uint(n.bit_length() - 1), C.c_uint(log_b), C.c_uint32(k0), C.c_uint32(nk), C.c_int(1 if all_cosets else 0),
                          None if avals is None else _p(avals), C.c_uint32(0 if avals is None else avals.shape[1]))
    return out


def sequence_value_polys(values, first_step, log_n):
    """values [n_seq][m] -> coefficient columns [n_seq][n] of c(x) = P(x w^-first_step)"""
    values = _u64(values)
    n_seq, m = values.shape
    out = np.zeros((n_seq, 1 << log_n), np.uint64)
    lib().cso_sequence_value_polys(_p(values), C.c_uint32(n_seq), C.c_uint32(m), C.c_uint32(first_step), C.c_uint(log_n), _p(out))
    return out


def schnorr_desc(w):
    """SchnorrAir: degrees (src/schnorr/air.rs:533-585) and the 61 assertions of get_assertions (:111-226), in order."""
    base, cyc = schnorr_constraint_degrees(w.n_sig)
    one = int(to_mont([1])[0])
    reg, first, stride, seq, val = [], [], [], [], []
    def add(r, f, v=0, q=-1):
        reg.append(r); first.append(f); stride.append(512); val.append(v); seq.append(q)
    for i in range(18): add(i, 0, one if i == 6 else 0)
    add(18, 0)
    for i in range(18): add(19 + i, 0, one if i == 6 else 0)
    for i in range(5): add(37 + i, 0)
    for k in range(6): add(42 + k, 0, 0, k)            # sequence: R.x at the first step of every block
    for i in range(7): add(48 + i, 0)
    for k in range(6): add(k, 511, 0, 6 + k)           # sequence: x(s*G + h*P) == R.x at step 511 of every block
    return AirDesc(56, base, cyc, 512, reg, [0] * len(reg), val, first, stride, seq)


def schnorr_assertion_polys(w, log_n):
    """[12][n] coefficient columns: 6 for the step-0 sequences, 6 for the step-511 sequences (same values)."""
    vals = np.ascontiguousarray(w.sig_rx.T)            # [6][n_sig]
    return np.concatenate([sequence_value_polys(vals, 0, log_n), sequence_value_polys(vals, 511, log_n)])


def random_elements(k, seed):
    rng = np.random.default_rng(seed)
    return to_mont(rng.integers(0, P, size=k, dtype=np.uint64))


# ---- SchnorrAir (standalone) ---------------------------------------------------------------------------
class SchnorrWitness:
    def __init__(self, n_sig):
        self.n_sig = n_sig
        self.messages = np.zeros((n_sig, 28), np.uint64)
        self.sig_rx = np.zeros((n_sig, 6), np.uint64)
        self.sig_s = np.zeros((n_sig, 32), np.uint8)

    @classmethod
    def generate(cls, n_sig, seed=0x5EED):
        w = cls(n_sig)
        lib().cso_schnorr_witness_generate(C.c_uint32(n_sig), C.c_uint64(seed), _p(w.messages), _p(w.sig_rx), _p(w.sig_s, u8p))
        return w


def schnorr_build_trace(w):
    trace = np.zeros((56, 512 * w.n_sig), np.uint64)
    lib().cso_schnorr_build_trace(C.c_uint32(w.n_sig), _p(w.messages), _p(w.sig_rx), _p(w.sig_s, u8p), _p(trace))
    return trace


def schnorr_aux_columns(w):
    out = np.zeros((19, 512 * w.n_sig), np.uint64)
    lib().cso_schnorr_aux_columns(C.c_uint32(w.n_sig), _p(w.messages), _p(out))
    return out


def schnorr_mask_columns():
    out = np.zeros((36, 512), np.uint64)
    lib().cso_schnorr_mask_columns(_p(out))
    return out


def schnorr_evaluate_transitions(lde, aux_lde, ptab, k0=0):
    lde, aux_lde, ptab = _u64(lde), _u64(aux_lde), _u64(ptab)
    nk, width, n = lde.shape
    out = np.zeros((nk, 56, n), np.uint64)
    lib().cso_schnorr_evaluate_transitions(_p(lde), _p(aux_lde), _p(ptab), _p(out), C.c_uint(n.bit_length() - 1), C.c_uint32(k0), C.c_uint32(nk))
    return out


def schnorr_constraint_degrees(n_sig):
    base = np.zeros(56, np.uint32); cyc = np.zeros(56, np.uint32)
    lib().cso_schnorr_constraint_degrees(C.c_uint32(n_sig), _p(base, u32p), _p(cyc, u32p))
    return base, cyc


def composition_columns(combined):
    """combined [b][n] (coset-major) -> [b][n] coefficient columns H_i of H(x) = sum_i x^i H_i(x^b)."""
    combined = _u64(combined)
    b, n = combined.shape
    out = np.zeros((b, n), np.uint64)
    lib().cso_composition_columns(_p(combined), _p(out), C.c_uint(n.bit_length() - 1), C.c_uint(b.bit_length() - 1))
    return out


def evaluate_polys_at(coeffs, points):
    coeffs, points = _u64(coeffs), _u64(points)
    width, n = coeffs.shape
    out = np.zeros((points.size, width), np.uint64)
    lib().cso_evaluate_polys_at(_p(coeffs), C.c_uint32(width), C.c_uint(n.bit_length() - 1), _p(points), C.c_uint32(points.size), _p(out))
    return out


def deep_composition(trace_lde, comp_lde, z, ood_trace, ood_comp, alpha, beta, delta, deg_a, deg_b, log_b, k0=0):
    trace_lde, comp_lde = _u64(trace_lde), _u64(comp_lde)
    nk, width, n = trace_lde.shape
    nb = comp_lde.shape[1]
    out = np.zeros((nk, n), np.uint64)
    lib().cso_deep_composition(_p(trace_lde), _p(comp_lde), C.c_uint32(width), C.c_uint32(nb), C.c_uint64(z), _p(_u64(ood_trace)),
                               _p(_u64(ood_comp)), _p(_u64(alpha)), _p(_u64(beta)), _p(_u64(delta)), C.c_uint64(deg_a), C.c_uint64(deg_b),
                               _p(out), C.c_uint(n.bit_length() - 1), C.c_uint(log_b), C.c_uint32(k0), C.c_uint32(nk))
    return out


def fri_fold(evals, offset, alpha, folding=4):
    """one FRI layer: N evaluations over offset <w_N> (natural order) -> N / folding over offset^folding <w_(N/folding)>"""
    evals = _u64(evals)
    out = np.zeros(evals.size // folding, np.uint64)
    lib().cso_fri_fold(_p(evals), _p(out), C.c_uint(evals.size.bit_length() - 1), C.c_uint(folding.bit_length() - 1), C.c_uint64(offset),
                       C.c_uint64(alpha))
    return out


def fri_fold4(evals, offset, alpha):
    return fri_fold(evals, offset, alpha, 4)


# ---- FieldExtension::Quadratic / Cubic (oracle/ext.c): elements as m-tuples of memory-form base elements ---------------------------
def evaluate_polys_at_ext(coeffs, zp):
    coeffs, zp = _u64(coeffs), _u64(zp)
    width, n = coeffs.shape
    m = zp.size
    out = np.zeros((width, m), np.uint64)
    lib().cso_evaluate_polys_at_ext(_p(coeffs), C.c_uint32(width), C.c_uint(n.bit_length() - 1), _p(zp), _p(out), C.c_int(m))
    return out


def deep_composition_ext(trace_lde, comp_lde, zp, ood_trace, ood_comp, alpha, beta, delta, deg_a, deg_b, log_b):
    trace_lde, comp_lde, zp = _u64(trace_lde), _u64(comp_lde), _u64(zp)
    b, width, n = trace_lde.shape
    m = zp.size
    nb = comp_lde.shape[1] // m
    out = np.zeros((m, b, n), np.uint64)
    lib().cso_deep_composition_ext(_p(trace_lde), _p(comp_lde), C.c_uint32(width), C.c_uint32(nb), _p(zp), _p(_u64(ood_trace)),
                                   _p(_u64(ood_comp)), _p(_u64(alpha)), _p(_u64(beta)), _p(_u64(delta)), _p(_u64(deg_a)), _p(_u64(deg_b)),
                                   _p(out), C.c_uint(n.bit_length() - 1), C.c_uint(log_b), C.c_int(m))
    return out


def fri_fold_ext(evals, offset, alphap, folding=4):
    evals, alphap = _u64(evals), _u64(alphap)
    m, N = evals.shape
    out = np.zeros((m, N // folding), np.uint64)
    lib().cso_fri_fold_ext(_p(evals), _p(out), C.c_uint(N.bit_length() - 1), C.c_uint(folding.bit_length() - 1), C.c_uint64(offset), _p(alphap),
                           C.c_int(m))
    return out


def fri_fold4_ext(evals, offset, alphap):
    return fri_fold_ext(evals, offset, alphap, 4)
