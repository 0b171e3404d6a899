"""CPU verifier for proofs written by cstark_tx_prove -- TEST INFRASTRUCTURE (part of oracle/; never used by the product).

Restates `winterfell::verify::<TransactionAir>(proof, pub_inputs)` as called at /root/reference/src/lib.rs:144-150
(engine code absent from the reference tree: [UPSTREAM-RECALL winterfell v0.3], parity unpinned).  Steps:
  1. rebuild the public coin from context + public inputs, replay the prover's draws (csrc/prove.hip header);
  2. out-of-domain consistency: the 115 transition constraints + 4 assertions evaluated on the OOD frame at z
     (oracle C code, cso_tx_combined_from_frame) must equal sum_i z^i H_i(z^8);
  3. Merkle openings of the queried trace / composition rows against the two roots (BLAKE3);
  4. DEEP composition values at the queried points from the opened rows;
  5. FRI: layer openings, folding consistency layer to layer (factor 4, 8 or 16), remainder commitment, remainder degree.
Field arithmetic here is Python integers on canonical values -- deliberately independent of the Montgomery code paths.
"""
import ctypes as C
import struct

import numpy as np

from . import oracle as O

P = (1 << 62) + (1 << 56) + (1 << 55) + 1
R_INV = pow(1 << 64, -1, P)


def _conventions():
    """the engine conventions of include/cstark_conventions.h, as compiled into the oracle library (shared with the product)"""
    import ctypes as C
    out = (C.c_int64 * 16)()
    O.lib().cso_conventions(out)
    keys = ("generator", "two_adic_root_exp", "lde_offset", "hashed_bytes_montgomery", "transition_exemptions", "coin_first_counter",
            "coin_reject_above_p", "query_dedup", "deep_draws_per_register", "e2_c0", "e2_c1", "e3_c0", "e3_c1", "e3_c2")
    return dict(zip(keys, [int(v) for v in out]))


CONV = _conventions()
GEN = CONV["lde_offset"]                                                     # offset of the evaluation domains
TWO_ADIC_ROOT = pow(CONV["generator"], CONV["two_adic_root_exp"], P)         # order 2^55
assert CONV["transition_exemptions"] == 1


def elem_bytes(a):
    """bytes of field elements (numpy uint64, memory form) as they enter a hash: memory form or canonical, little-endian"""
    a = np.ascontiguousarray(a, np.uint64)
    if CONV["hashed_bytes_montgomery"]:
        return a.tobytes()
    return np.ascontiguousarray(O.from_mont(a.reshape(-1))).tobytes()


def transition_adjustment(ce_size, n, eval_degree):
    O.lib().cso_transition_adjustment.restype = __import__("ctypes").c_uint64
    c = __import__("ctypes").c_uint64
    return int(O.lib().cso_transition_adjustment(c(ce_size), c(n), c(eval_degree)))


def boundary_adjustment(ce_size, n, m):
    O.lib().cso_boundary_adjustment.restype = __import__("ctypes").c_uint64
    c = __import__("ctypes").c_uint64
    return int(O.lib().cso_boundary_adjustment(c(ce_size), c(n), c(m)))


class VerifierError(Exception):
    pass


def from_mont(v):
    return (int(v) * R_INV) % P


def to_mont(v):
    return (int(v) << 64) % P


def root_of_unity(log_n):
    return pow(TWO_ADIC_ROOT, 1 << (55 - log_n), P)


# ---- extension fields (FieldExtension::Quadratic / Cubic; see oracle/ext.c for the assumed polynomials): tuples of canonical
# integers of length m; E2 = F_p[u]/(u^2 - 2u - 2), E3 = F_p[v]/(v^3 + v + 1) -------------------------------------------------------
def e_add(x, y):
    return tuple((a + b) % P for a, b in zip(x, y))


def e_sub(x, y):
    return tuple((a - b) % P for a, b in zip(x, y))


def e_mul(x, y):
    if len(x) == 2:  # u^2 = C1 u + C0
        bd = x[1] * y[1]
        return ((x[0] * y[0] + CONV["e2_c0"] * bd) % P, (x[0] * y[1] + x[1] * y[0] + CONV["e2_c1"] * bd) % P)
    c0, c1, c2 = CONV["e3_c0"], CONV["e3_c1"], CONV["e3_c2"]  # v^3 = C2 v^2 + C1 v + C0
    d0 = x[0] * y[0]
    d1 = x[0] * y[1] + x[1] * y[0]
    d2 = x[0] * y[2] + x[1] * y[1] + x[2] * y[0]
    d3 = x[1] * y[2] + x[2] * y[1]
    d4 = x[2] * y[2]
    return ((d0 + c0 * d3 + c2 * c0 * d4) % P, (d1 + c1 * d3 + (c2 * c1 + c0) * d4) % P, (d2 + c2 * d3 + (c2 * c2 + c1) * d4) % P)


def e_scale(x, s):
    return tuple(a * s % P for a in x)


def e_base(v, m):
    return (v % P,) + (0,) * (m - 1)


def e_gen(m):
    """the adjoined root (u or v) as an element"""
    return (0, 1) + (0,) * (m - 2)


def e_pow(x, e):
    r = e_base(1, len(x))
    while e:
        if e & 1:
            r = e_mul(r, x)
        x = e_mul(x, x)
        e >>= 1
    return r


def e_inv(x):
    # Fermat would need p^m - 2; use the adjugate: x^-1 = adj(x) / norm(x) with the norm in the base field
    m = len(x)
    if m == 2:  # 1/(a + b u) = (a + C1 b - b u) / (a (a + C1 b) - C0 b^2)
        a, b = x
        s_ = (a + CONV["e2_c1"] * b) % P
        t = pow((a * s_ - CONV["e2_c0"] * b * b) % P, -1, P)
        return (s_ * t % P, (-b) * t % P)
    c0, c1, c2 = CONV["e3_c0"], CONV["e3_c1"], CONV["e3_c2"]
    a, b, c = x
    y0, y1, y2 = c0 * c % P, (a + c1 * c) % P, (b + c2 * c) % P            # x v
    w0, w1, w2 = c0 * y2 % P, (y0 + c1 * y2) % P, (y1 + c2 * y2) % P       # x v^2
    r0 = (y1 * w2 - y2 * w1) % P
    r1 = (c * w1 - b * w2) % P
    r2 = (b * y2 - c * y1) % P
    t = pow((a * r0 + y0 * r1 + w0 * r2) % P, -1, P)
    return (r0 * t % P, r1 * t % P, r2 * t % P)


def e_mont(x):
    return [to_mont(v) for v in x]


class Coin:
    def __init__(self, seed_bytes, hash_fn=0):
        self.hash_fn = hash_fn
        self.seed, self.counter = self.h(seed_bytes), CONV["coin_first_counter"] - 1

    def h(self, data):
        return O.digest(data, self.hash_fn)

    def reseed(self, digest):
        self.seed, self.counter = self.h(self.seed + bytes(digest)), CONV["coin_first_counter"] - 1

    def reseed_int(self, v):
        self.seed, self.counter = self.h(self.seed + struct.pack("<Q", v)), CONV["coin_first_counter"] - 1

    def _next(self):
        self.counter += 1
        return struct.unpack("<Q", self.h(self.seed + struct.pack("<Q", self.counter))[:8])[0]

    def draw(self):
        """a field element (canonical integer)"""
        while True:
            v = self._next()
            if not CONV["coin_reject_above_p"]:
                return v % P
            if v < P:
                return v

    def draw_e(self, m=2):
        """an element of the degree-m extension: m base draws"""
        return tuple(self.draw() for _ in range(m))

    def draw_integers(self, count, domain):
        out = []
        while len(out) < count:
            v = self._next() & (domain - 1)
            if not CONV["query_dedup"] or v not in out:
                out.append(v)
        return out


class Reader:
    def __init__(self, b):
        self.b, self.o = bytes(b), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise VerifierError("proof truncated")
        r = self.b[self.o:self.o + n]
        self.o += n
        return r

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def elems(self, n):
        return np.frombuffer(self.take(8 * n), np.uint64)


def fold_positions(pos, rows):
    out = []
    for p in pos:
        r = p & (rows - 1)
        if r not in out:
            out.append(r)
    return out


def merkle_root_from_path(leaf, index, path, hash_fn=0):
    h = leaf
    for sib in path:
        h = O.digest(sib + h, hash_fn) if index & 1 else O.digest(h + sib, hash_fn)
        index >>= 1
    return h


def parse(proof):
    r = Reader(proof)
    if r.take(4) != b"CSTK":
        raise VerifierError("bad magic")
    d = {"version": r.u32(), "air": r.u32(), "width": r.u32(), "log_n": r.u32(), "depth": r.u32()}
    d["options"] = [r.u32() for _ in range(7)]
    shapes = {0: (94, 8), 1: (65, 4), 2: (56, 8), 3: (2, 2), 4: (14, 4)}  # air id -> (trace width, composition columns)
    if d["version"] != 1 or d["air"] not in shapes or d["width"] != shapes[d["air"]][0]:
        raise VerifierError("unsupported proof header")
    W, ce = shapes[d["air"]]
    d["ce"] = ce
    nq, blowup = d["options"][0], d["options"][1]
    folding = d["options"][5]
    if blowup not in (2, 4, 8, 16) or blowup < ce or folding not in (4, 8, 16) or not (6 <= d["log_n"] <= 21) or not (1 <= nq <= 128):
        raise VerifierError("unsupported parameters")  # (blowup below the AIR's constraint-evaluation blowup: the engine refuses it too)
    ext = d["options"][4]
    if ext not in (0, 1, 2):
        raise VerifierError("unsupported field extension")
    em = ext + 1  # words per element of the field the coin draws from
    log_N, log_f = d["log_n"] + blowup.bit_length() - 1, folding.bit_length() - 1
    d["trace_root"], d["cons_root"] = r.take(32), r.take(32)
    nl = r.u32()
    if nl > 16:
        raise VerifierError("bad layer count")
    d["layer_roots"] = [r.take(32) for _ in range(nl)]
    d["rem_commit"] = r.take(32)
    d["ood_cur"], d["ood_next"], d["ood_comp"] = r.elems(W * em), r.elems(W * em), r.elems(ce * em)
    d["nonce"] = r.u64()
    d["trace_rows"] = r.elems(nq * W).reshape(nq, W)
    d["trace_paths"] = [[r.take(32) for _ in range(log_N)] for _ in range(nq)]
    d["cons_rows"] = r.elems(nq * ce * em).reshape(nq, ce * em)
    d["cons_paths"] = [[r.take(32) for _ in range(log_N)] for _ in range(nq)]
    d["layers"] = []
    lg = log_N
    for _ in range(nl):
        if lg < log_f:
            raise VerifierError("too many layers")
        npos = r.u32()
        if npos > nq:
            raise VerifierError("bad layer opening count")
        rows = r.elems(npos * folding * em).reshape(npos, folding * em)
        paths = [[r.take(32) for _ in range(lg - log_f)] for _ in range(npos)]
        d["layers"].append((rows, paths))
        lg -= log_f
    rl = r.u32()
    if rl > 1024:
        raise VerifierError("bad remainder length")
    d["remainder"] = r.elems(rl * em)
    if r.o != len(r.b):
        raise VerifierError("trailing bytes")
    return d


def _poly_at(coeff_cols, x):
    """values of coefficient columns (memory form) at the canonical point x -> canonical integers"""
    cols = np.ascontiguousarray(coeff_cols, np.uint64)
    pts = np.array([to_mont(x)], np.uint64)
    return [from_mont(v) for v in O.evaluate_polys_at(cols, pts)[0]]


class _TxAir:
    """TransactionAir: 115 transition constraints in 5 degree groups, 4 single assertions (src/air.rs:76-108, :175-184)."""
    air, width, ce = 0, 94, 8

    def __init__(self, d, initial_root, final_root):
        self.depth = d["depth"]
        self.pub = [from_mont(v) for v in list(initial_root) + list(final_root)]
        self.pub_bytes = b""
        self.nc, self.na = 115, 4

    def ood_combined(self, d, log_n, z, ta, tb, ba, bb):
        cf = O.TxCoeffsStruct()
        for i in range(115):
            cf.t_alpha[i], cf.t_beta[i] = to_mont(ta[i]), to_mont(tb[i])
        for i in range(4):
            cf.b_alpha[i], cf.b_beta[i] = to_mont(ba[i]), to_mont(bb[i])
        pub4 = np.array([to_mont(self.pub[0]), to_mont(self.pub[1]), to_mont(self.pub[7]), to_mont(self.pub[8])], np.uint64)
        lib = O.lib()
        lib.cso_tx_combined_from_frame.restype = C.c_uint64
        cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
        return from_mont(lib.cso_tx_combined_from_frame(O._p(cur), O._p(nxt), C.byref(cf), O._p(pub4), C.c_uint(self.depth), C.c_uint(log_n),
                                                        C.c_uint(3), C.c_uint64(to_mont(z))))


def _tx_ood_combined_ext(self, d, log_n, z, ta, tb, ba, bb, m):
    n, log_b = 1 << log_n, 3
    B = lambda v: e_base(v, m)
    wn = root_of_unity(log_n)
    cur, nxt = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m)
    pco = O.interpolate_columns(O.tx_periodic_columns(self.depth))
    per = _tuples(O.evaluate_polys_at_ext(pco, e_mont(e_pow(z, n // 1024))).reshape(-1), m)
    cvals = _tx_constraints_over_e(cur, nxt, per, m)
    adj = [int(v) for v in O.tx_degree_adjustments(log_n, log_b)]
    zpow = {}
    acc = B(0)
    for i in range(115):
        if adj[i] not in zpow:
            zpow[adj[i]] = e_pow(z, adj[i])
        acc = e_add(acc, e_mul(cvals[i], e_add(ta[i], e_mul(tb[i], zpow[adj[i]]))))
    w_last = pow(wn, n - 1, P)
    acc = e_mul(acc, e_mul(e_sub(z, B(w_last)), e_inv(e_sub(e_pow(z, n), B(1)))))
    xb = e_pow(z, boundary_adjustment(n << log_b, n, 1))
    first = last = B(0)
    for a in range(2):
        first = e_add(first, e_mul(e_sub(cur[58 + a], B(self.pub[a])), e_add(ba[a], e_mul(bb[a], xb))))
        last = e_add(last, e_mul(e_sub(cur[58 + a], B(self.pub[7 + a])), e_add(ba[2 + a], e_mul(bb[2 + a], xb))))
    return e_add(acc, e_add(e_mul(first, e_inv(e_sub(z, B(1)))), e_mul(last, e_inv(e_sub(z, B(w_last))))))


_TxAir.ood_combined_ext = _tx_ood_combined_ext


class _GenericAir:
    """Standalone AIRs through the generic description: transition values from the oracle's pointwise evaluators, merge restated here."""

    def _merge(self, log_n, z, res, ta, tb, ba, bb, cur, assertions):
        n = 1 << log_n
        ce_size = n * self.ce
        wn = root_of_unity(log_n)
        acc = 0
        for i in range(self.nc):
            ev = int(self.base[i]) * (n - 1) + (int(self.cycles[i]) * (n // self.cycle_len) * (self.cycle_len - 1) if self.cycle_len else 0)
            acc += res[i] * ((ta[i] + tb[i] * pow(z, transition_adjustment(ce_size, n, ev), P)) % P)
        acc = acc % P * ((z - pow(wn, n - 1, P)) % P) % P * pow(pow(z, n, P) - 1, -1, P) % P
        for a, (reg, first, stride, value) in enumerate(assertions):
            m = n // stride if stride else 1
            div = (pow(z, m, P) - pow(wn, (first * m) % n, P)) % P
            term = (cur[reg] - value) % P * ((ba[a] + bb[a] * pow(z, boundary_adjustment(ce_size, n, m), P)) % P) % P
            acc = (acc + term * pow(div, -1, P)) % P
        return acc


def _constraints_over_e(evalfn, frames, n_out, m):
    """Generic form of _tx_constraints_over_e: `frames` is a list of lists of m-tuples (current row, next row, periodic values,
    ...); evalfn(list of uint64 arrays in memory form) -> base-field constraint values.  Total degree <= 8."""
    K = 8 * (m - 1) + 4
    ys = []
    for t in range(K):
        pw = [pow(t, q, P) for q in range(m)]
        arrs = [np.array([to_mont(sum(c * w_ for c, w_ in zip(e, pw)) % P) for e in fr], np.uint64) for fr in frames]
        ys.append([from_mont(v) for v in evalfn(arrs)[:n_out]])
    g = e_gen(m)
    lag = []
    for j in range(K):
        num, den = e_base(1, m), 1
        for q in range(K):
            if q != j:
                num = e_mul(num, e_sub(g, e_base(q, m)))
                den = den * (j - q) % P
        lag.append(e_scale(num, pow(den, -1, P)))
    out = []
    for i in range(n_out):
        acc = e_base(0, m)
        for j in range(K):
            acc = e_add(acc, e_scale(lag[j], ys[j][i]))
        out.append(acc)
    return out


def _merge_e(air, log_n, z, res, ta, tb, ba, bb, cur, assertions, m):
    """_GenericAir._merge over the extension: res / cur / coefficients are m-tuples; assertion values are m-tuples."""
    n = 1 << log_n
    ce_size = n * air.ce
    wn = root_of_unity(log_n)
    B = lambda v: e_base(v, m)
    acc = B(0)
    for i in range(air.nc):
        ev = int(air.base[i]) * (n - 1) + (int(air.cycles[i]) * (n // air.cycle_len) * (air.cycle_len - 1) if air.cycle_len else 0)
        acc = e_add(acc, e_mul(res[i], e_add(ta[i], e_mul(tb[i], e_pow(z, transition_adjustment(ce_size, n, ev))))))
    acc = e_mul(acc, e_mul(e_sub(z, B(pow(wn, n - 1, P))), e_inv(e_sub(e_pow(z, n), B(1)))))
    for a, (reg, first, stride, value) in enumerate(assertions):
        mm = n // stride if stride else 1
        div = e_sub(e_pow(z, mm), B(pow(wn, (first * mm) % n, P)))
        term = e_mul(e_sub(cur[reg], value), e_add(ba[a], e_mul(bb[a], e_pow(z, boundary_adjustment(ce_size, n, mm)))))
        acc = e_add(acc, e_mul(term, e_inv(div)))
    return acc


class _MerkleAir(_GenericAir):
    air, width, ce = 1, 65, 4

    def __init__(self, d, initial_root, final_root):
        self.depth = d["depth"]
        self.pub = [from_mont(v) for v in list(initial_root) + list(final_root)]
        self.pub_bytes = b""
        self.base = np.zeros(106, np.uint32); self.cycles = np.zeros(106, np.uint32)
        O.lib().cso_merkle_constraint_degrees(O._p(self.base, O.u32p), O._p(self.cycles, O.u32p))
        self.cycle_len, self.nc, self.na = 512, 106, 14

    def ood_combined(self, d, log_n, z, ta, tb, ba, bb):
        n = 1 << log_n
        pcols = O.interpolate_columns(O.merkle_periodic_columns(self.depth))
        pv = np.array([to_mont(v) for v in _poly_at(pcols, pow(z, n // 512, P))], np.uint64)
        cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
        res = np.zeros(128, np.uint64)
        O.lib().cso_merkle_evaluate_transition(O._p(cur), O._p(nxt), O._p(pv), O._p(res))
        assertions = [(58 + a % 7, (n - 1) if a >= 7 else 0, 0, self.pub[a]) for a in range(14)]  # src/merkle/update/air.rs:142-170
        return self._merge(log_n, z, [from_mont(v) for v in res[:106]], ta, tb, ba, bb, [from_mont(v) for v in cur], assertions)


    def ood_combined_ext(self, d, log_n, z, ta, tb, ba, bb, m):
        n = 1 << log_n
        pco = O.interpolate_columns(O.merkle_periodic_columns(self.depth))
        per = _tuples(O.evaluate_polys_at_ext(pco, e_mont(e_pow(z, n // 512))).reshape(-1), m)
        cur, nxt = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m)

        def evalfn(arrs):
            res = np.zeros(128, np.uint64)
            O.lib().cso_merkle_evaluate_transition(O._p(arrs[0]), O._p(arrs[1]), O._p(arrs[2]), O._p(res))
            return res
        res = _constraints_over_e(evalfn, [cur, nxt, per], 106, m)
        assertions = [(58 + a % 7, (n - 1) if a >= 7 else 0, 0, e_base(self.pub[a], m)) for a in range(14)]
        return _merge_e(self, log_n, z, res, ta, tb, ba, bb, cur, assertions, m)


class _RangeAir(_GenericAir):
    air, width, ce = 3, 2, 2

    def __init__(self, d, number):
        self.pub = [from_mont(number)]
        self.pub_bytes = b""
        self.base, self.cycles, self.cycle_len, self.nc, self.na = [2, 1], [0, 0], 0, 2, 2

    def ood_combined(self, d, log_n, z, ta, tb, ba, bb):
        n = 1 << log_n
        cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
        res = np.zeros(8, np.uint64)
        O.lib().cso_range_evaluate_transition(O._p(cur), O._p(nxt), None, O._p(res))
        assertions = [(1, 0, 0, 0), (1, n - 1, 0, self.pub[0])]  # src/range/air.rs:79-86
        return self._merge(log_n, z, [from_mont(v) for v in res[:2]], ta, tb, ba, bb, [from_mont(v) for v in cur], assertions)


    def ood_combined_ext(self, d, log_n, z, ta, tb, ba, bb, m):
        n = 1 << log_n
        cur, nxt = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m)

        def evalfn(arrs):
            res = np.zeros(8, np.uint64)
            O.lib().cso_range_evaluate_transition(O._p(arrs[0]), O._p(arrs[1]), None, O._p(res))
            return res
        res = _constraints_over_e(evalfn, [cur, nxt], 2, m)
        assertions = [(1, 0, 0, e_base(0, m)), (1, n - 1, 0, e_base(self.pub[0], m))]
        return _merge_e(self, log_n, z, res, ta, tb, ba, bb, cur, assertions, m)


class _RescueAir(_GenericAir):
    """RescueAir of benches/rescue.rs:145-250: 14 x degree (3; one cycle of 8), seed / result assertions on registers 0..6"""
    air, width, ce = 4, 14, 4

    def __init__(self, d, seed, result):
        self.pub = [from_mont(v) for v in list(seed) + list(result)]
        self.pub_bytes = b""
        self.base, self.cycles, self.cycle_len, self.nc, self.na = [3] * 14, [1] * 14, 8, 14, 14

    def _assertions(self, n, wrap):
        return [(a % 7, (n - 1) if a >= 7 else 0, 0, wrap(self.pub[a])) for a in range(14)]  # get_assertions :224-243

    def ood_combined(self, d, log_n, z, ta, tb, ba, bb):
        n = 1 << log_n
        pcols = O.interpolate_columns(O.rescue_chain_periodic_columns())
        pv = np.array([to_mont(v) for v in _poly_at(pcols, pow(z, n // 8, P))], np.uint64)
        cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
        res = np.zeros(16, np.uint64)
        O.lib().cso_rescue_chain_evaluate_transition(O._p(cur), O._p(nxt), O._p(pv), O._p(res))
        return self._merge(log_n, z, [from_mont(v) for v in res[:14]], ta, tb, ba, bb, [from_mont(v) for v in cur], self._assertions(n, lambda v: v))

    def ood_combined_ext(self, d, log_n, z, ta, tb, ba, bb, m):
        n = 1 << log_n
        pco = O.interpolate_columns(O.rescue_chain_periodic_columns())
        per = _tuples(O.evaluate_polys_at_ext(pco, e_mont(e_pow(z, n // 8))).reshape(-1), m)
        cur, nxt = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m)

        def evalfn(arrs):
            res = np.zeros(16, np.uint64)
            O.lib().cso_rescue_chain_evaluate_transition(O._p(arrs[0]), O._p(arrs[1]), O._p(arrs[2]), O._p(res))
            return res
        res = _constraints_over_e(evalfn, [cur, nxt, per], 14, m)
        return _merge_e(self, log_n, z, res, ta, tb, ba, bb, cur, self._assertions(n, lambda v: e_base(v, m)), m)


class _SchnorrAir(_GenericAir):
    air, width, ce = 2, 56, 8

    def __init__(self, d, w):
        self.w = w
        self.pub = [from_mont(v) for v in list(w.messages.reshape(-1)) + list(w.sig_rx.reshape(-1))]
        self.pub_bytes = w.sig_s.tobytes()
        self.base, self.cycles = O.schnorr_constraint_degrees(w.n_sig)
        self.cycle_len, self.nc, self.na = 512, 56, 61
        if d["depth"] != w.n_sig:
            raise VerifierError("proof is for a different number of signatures")

    def ood_combined(self, d, log_n, z, ta, tb, ba, bb):
        n = 1 << log_n
        w = self.w
        masks = np.array([to_mont(v) for v in _poly_at(O.interpolate_columns(O.schnorr_mask_columns()), pow(z, n // 512, P))], np.uint64)
        aux = np.array([to_mont(v) for v in _poly_at(O.interpolate_columns(O.schnorr_aux_columns(w)), z)], np.uint64)
        cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
        res = np.zeros(56, np.uint64)
        pk, inp = np.ascontiguousarray(aux[:12]), np.ascontiguousarray(aux[12:19])
        O.lib().cso_schnorr_evaluate_transition_at(O._p(cur), O._p(nxt), O._p(masks), O._p(pk), O._p(inp), O._p(res))
        desc = O.schnorr_desc(w)
        seq_at_z = _poly_at(O.schnorr_assertion_polys(w, log_n), z)
        assertions = []
        for a in range(desc.na):
            q = int(desc.a_seq[a])
            assertions.append((int(desc.a_reg[a]), int(desc.a_first[a]), int(desc.a_stride[a]),
                               seq_at_z[q] if q >= 0 else from_mont(desc.a_value[a])))
        return self._merge(log_n, z, [from_mont(v) for v in res], ta, tb, ba, bb, [from_mont(v) for v in cur], assertions)


def _schnorr_ood_combined_ext(self, d, log_n, z, ta, tb, ba, bb, m):
    n = 1 << log_n
    w = self.w
    masks = _tuples(O.evaluate_polys_at_ext(O.interpolate_columns(O.schnorr_mask_columns()), e_mont(e_pow(z, n // 512))).reshape(-1), m)
    aux = _tuples(O.evaluate_polys_at_ext(O.interpolate_columns(O.schnorr_aux_columns(w)), e_mont(z)).reshape(-1), m)
    cur, nxt = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m)

    def evalfn(arrs):
        res = np.zeros(56, np.uint64)
        ax = arrs[3]
        O.lib().cso_schnorr_evaluate_transition_at(O._p(arrs[0]), O._p(arrs[1]), O._p(arrs[2]), O._p(np.ascontiguousarray(ax[:12])),
                                                   O._p(np.ascontiguousarray(ax[12:19])), O._p(res))
        return res
    res = _constraints_over_e(evalfn, [cur, nxt, masks, aux], 56, m)
    desc = O.schnorr_desc(w)
    seq_at_z = _tuples(O.evaluate_polys_at_ext(O.schnorr_assertion_polys(w, log_n), e_mont(z)).reshape(-1), m)
    assertions = []
    for a in range(desc.na):
        q = int(desc.a_seq[a])
        assertions.append((int(desc.a_reg[a]), int(desc.a_first[a]), int(desc.a_stride[a]),
                           seq_at_z[q] if q >= 0 else e_base(from_mont(desc.a_value[a]), m)))
    return _merge_e(self, log_n, z, res, ta, tb, ba, bb, cur, assertions, m)


_SchnorrAir.ood_combined_ext = _schnorr_ood_combined_ext


def verify(proof, initial_root, final_root, options=None):
    """TransactionAir.  Raises VerifierError unless `proof` shows that a valid 94-register trace links initial_root to final_root.
    initial_root / final_root: 7 field elements each, memory form (as TransactionMetadata holds them).
    options: the 7 ProofOptions values the verifier expects (None = accept what the proof states)."""
    d = parse(proof)
    if d["air"] != 0:
        raise VerifierError("not a TransactionAir proof")
    if d["options"][4] in (1, 2):
        return _verify_ext(d, _TxAir(d, initial_root, final_root), options)
    return _verify(d, _TxAir(d, initial_root, final_root), options)


def verify_merkle(proof, initial_root, final_root, options=None):
    """MerkleAir (src/merkle/update/mod.rs:109-127)."""
    d = parse(proof)
    if d["air"] != 1:
        raise VerifierError("not a MerkleAir proof")
    if d["options"][4] in (1, 2):
        return _verify_ext(d, _MerkleAir(d, initial_root, final_root), options)
    return _verify(d, _MerkleAir(d, initial_root, final_root), options)


def verify_range(proof, number, options=None):
    """RangeProofAir (src/range/mod.rs:103-110); number in memory form."""
    d = parse(proof)
    if d["air"] != 3:
        raise VerifierError("not a RangeProofAir proof")
    if d["options"][4] in (1, 2):
        return _verify_ext(d, _RangeAir(d, number), options)
    return _verify(d, _RangeAir(d, number), options)


def verify_rescue(proof, seed, result, options=None):
    """RescueAir (benches/rescue.rs:88-94); seed / result: 7 elements each, memory form."""
    d = parse(proof)
    if d["air"] != 4:
        raise VerifierError("not a RescueAir proof")
    if d["options"][4] in (1, 2):
        return _verify_ext(d, _RescueAir(d, seed, result), options)
    return _verify(d, _RescueAir(d, seed, result), options)


def verify_schnorr(proof, witness, options=None):
    """SchnorrAir (src/schnorr/mod.rs:175-186); witness: messages [n][28], sig_rx [n][6], sig_s [n][32] (all public)."""
    d = parse(proof)
    if d["air"] != 2:
        raise VerifierError("not a SchnorrAir proof")
    if d["options"][4] in (1, 2):
        return _verify_ext(d, _SchnorrAir(d, witness), options)
    return _verify(d, _SchnorrAir(d, witness), options)


def _verify(d, air, options):
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = d["options"]
    if options is not None and list(options) != d["options"]:
        raise VerifierError("proof options differ from the expected ones")
    if hash_fn not in (0, 1) or ext != 0 or max_rem & (max_rem - 1) or not (128 <= max_rem <= 1024):
        raise VerifierError("unsupported options")
    log_n = d["log_n"]
    log_b, log_f = blowup.bit_length() - 1, folding.bit_length() - 1
    log_N = log_n + log_b
    n, N, W, b, ce = 1 << log_n, 1 << log_N, air.width, blowup, air.ce
    log_rem = max_rem.bit_length() - 1
    n_layers, lg = 0, log_N
    while lg > log_rem:
        lg -= log_f
        n_layers += 1
    if n_layers != len(d["layer_roots"]) or len(d["remainder"]) != 1 << lg:
        raise VerifierError("FRI layer structure does not match the options")

    # 1. channel
    seed = bytes([W, log_n]) + struct.pack("<Q", P) + bytes([nq, log_b, grinding, hash_fn, ext, folding, log_rem])
    seed += b"".join(struct.pack("<Q", v) for v in air.pub) + air.pub_bytes
    coin = Coin(seed, hash_fn)
    H = coin.h
    coin.reseed(d["trace_root"])
    ta, tb, ba, bb = [], [], [], []
    for _ in range(air.nc):
        ta.append(coin.draw()); tb.append(coin.draw())
    for _ in range(air.na):
        ba.append(coin.draw()); bb.append(coin.draw())
    coin.reseed(d["cons_root"])
    z = coin.draw()

    # 2. out-of-domain consistency
    lhs = air.ood_combined(d, log_n, z, ta, tb, ba, bb)
    cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
    hz = [from_mont(v) for v in d["ood_comp"]]
    rhs = sum(h * pow(z, i, P) for i, h in enumerate(hz)) % P
    if lhs != rhs:
        raise VerifierError("out-of-domain constraint evaluations are inconsistent")
    coin.reseed(H(elem_bytes(cur) + elem_bytes(nxt)))
    coin.reseed(H(elem_bytes(d["ood_comp"])))
    d_alpha, d_beta = [], []
    for _ in range(W):
        d_alpha.append(coin.draw()); d_beta.append(coin.draw()); [coin.draw() for _ in range(2, CONV["deep_draws_per_register"])]
    d_delta = [coin.draw() for _ in range(ce)]
    deg_a, deg_b = coin.draw(), coin.draw()
    alphas = []
    for root in d["layer_roots"]:
        coin.reseed(root)
        alphas.append(coin.draw())
    if H(elem_bytes(d["remainder"])) != d["rem_commit"]:
        raise VerifierError("remainder does not match its commitment")
    coin.reseed(d["rem_commit"])
    if grinding:
        v = struct.unpack("<Q", H(coin.seed + struct.pack("<Q", d["nonce"]))[:8])[0]
        if v & ((1 << grinding) - 1):
            raise VerifierError("proof of work not satisfied")
    coin.reseed_int(d["nonce"])
    positions = coin.draw_integers(nq, N)

    # 3. trace / composition openings
    for q, pos in enumerate(positions):
        if merkle_root_from_path(H(elem_bytes(d["trace_rows"][q])), pos, d["trace_paths"][q], hash_fn) != d["trace_root"]:
            raise VerifierError("trace opening %d does not match the trace commitment" % q)
        if merkle_root_from_path(H(elem_bytes(d["cons_rows"][q])), pos, d["cons_paths"][q], hash_fn) != d["cons_root"]:
            raise VerifierError("composition opening %d does not match the constraint commitment" % q)

    # 4. DEEP composition at the queried points
    wN, wn = root_of_unity(log_N), root_of_unity(log_n)
    tz = [from_mont(v) for v in d["ood_cur"]]
    tzw = [from_mont(v) for v in d["ood_next"]]
    zw, zb = z * wn % P, pow(z, ce, P)
    deep = []
    for q, pos in enumerate(positions):
        x = GEN * pow(wN, pos, P) % P
        i1, i2, i3 = pow(x - z, -1, P), pow(x - zw, -1, P), pow(x - zb, -1, P)
        row = [from_mont(v) for v in d["trace_rows"][q]]
        crow = [from_mont(v) for v in d["cons_rows"][q]]
        acc = 0
        for c in range(W):
            acc += d_alpha[c] * (row[c] - tz[c]) % P * i1 + d_beta[c] * (row[c] - tzw[c]) % P * i2
        for i in range(ce):
            acc += d_delta[i] * (crow[i] - hz[i]) % P * i3
        deep.append(acc % P * ((deg_a + deg_b * x) % P) % P)

    # 5. FRI
    cur_pos, cur_val = positions, deep
    offset, lgl = GEN, log_N
    invf = pow(folding, -1, P)
    for l in range(n_layers):
        rows_n = 1 << (lgl - log_f)
        rows, paths = d["layers"][l]
        fpos = fold_positions(cur_pos, rows_n)
        if len(fpos) != len(rows):
            raise VerifierError("layer %d: wrong number of openings" % l)
        for t, rp in enumerate(fpos):
            if merkle_root_from_path(H(elem_bytes(rows[t])), rp, paths[t], hash_fn) != d["layer_roots"][l]:
                raise VerifierError("layer %d opening does not match its commitment" % l)
        for p, v in zip(cur_pos, cur_val):
            if from_mont(rows[fpos.index(p & (rows_n - 1))][p >> (lgl - log_f)]) != v:
                raise VerifierError("layer %d: evaluation differs from the previous layer's folding" % l)
        # fold each opened row: `folding` evaluations on the coset x * <zeta>, zeta = w^(N_l / folding)
        wl = root_of_unity(lgl)
        zeta_inv = pow(pow(wl, rows_n, P), -1, P)
        nxt_val = []
        for t, rp in enumerate(fpos):
            v = [from_mont(e) for e in rows[t]]
            x = offset * pow(wl, rp, P) % P
            r = alphas[l] * pow(x, -1, P) % P
            acc, rs = 0, 1
            for s in range(folding):
                cs = sum(v[k] * pow(zeta_inv, s * k, P) for k in range(folding)) % P * invf % P
                acc += cs * rs
                rs = rs * r % P
            nxt_val.append(acc % P)
        cur_pos, cur_val = fpos, nxt_val
        offset = pow(offset, folding, P)
        lgl -= log_f
    rem = [from_mont(v) for v in d["remainder"]]
    for p, v in zip(cur_pos, cur_val):
        if rem[p] != v:
            raise VerifierError("remainder differs from the last layer's folding")
    # remainder degree: evaluations over offset * <w_R> must interpolate to degree < R / blowup
    R = len(rem)
    co = O.ntt(np.array([to_mont(v) for v in rem], np.uint64), inverse=True)
    if any(int(v) != 0 for v in co[R // blowup:]):  # the offset scaling does not change which coefficients vanish
        raise VerifierError("FRI remainder is not a low-degree polynomial")
    return True


def _tuples(words, m):
    w = [from_mont(v) for v in words]
    return [tuple(w[m * i:m * i + m]) for i in range(len(w) // m)]


def _tx_constraints_over_e(cur, nxt, per, m):
    """The 115 transition constraints on a frame with entries in the extension, WITHOUT an extension-valued AIR: each constraint is
    a polynomial with base-field coefficients in the frame and periodic values, of total degree <= 7 (the declared maximum,
    src/air.rs:76-100).  Writing every entry as e(t) = a + b t (+ c t^2) makes t -> C(e(t)) a base-field polynomial of degree
    <= 7 (m - 1); it is sampled with the base-field evaluator at integer points t and read off at t = (the adjoined root) by
    Lagrange interpolation (t -> root is a ring homomorphism F_p[t] -> E)."""
    K = 7 * (m - 1) + 4
    ys = []
    for t in range(K):
        pw = [pow(t, q, P) for q in range(m)]
        ev = lambda e: to_mont(sum(c * w_ for c, w_ in zip(e, pw)) % P)
        c = np.array([ev(e) for e in cur], np.uint64)
        nx = np.array([ev(e) for e in nxt], np.uint64)
        pv = np.array([ev(e) for e in per], np.uint64)
        ys.append([from_mont(v) for v in O.tx_evaluate_transition(c, nx, pv)])
    g = e_gen(m)
    lag = []
    for j in range(K):
        num, den = e_base(1, m), 1
        for q in range(K):
            if q != j:
                num = e_mul(num, e_sub(g, e_base(q, m)))
                den = den * (j - q) % P
        lag.append(e_scale(num, pow(den, -1, P)))
    out = []
    for i in range(115):
        acc = e_base(0, m)
        for j in range(K):
            acc = e_add(acc, e_scale(lag[j], ys[j][i]))
        out.append(acc)
    return out


def _verify_ext(d, air, options):
    """FieldExtension::Quadratic / Cubic proofs of any of the AIRs (layout: oracle/prover.py prove_ext)."""
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = d["options"]
    m = ext + 1
    if options is not None and list(options) != d["options"]:
        raise VerifierError("proof options differ from the expected ones")
    if hash_fn not in (0, 1) or max_rem & (max_rem - 1) or not (128 <= max_rem <= 1024):
        raise VerifierError("unsupported options")
    log_n, log_b, log_f = d["log_n"], blowup.bit_length() - 1, folding.bit_length() - 1
    log_N = log_n + log_b
    n, N, W, b, ce = 1 << log_n, 1 << log_N, air.width, blowup, air.ce
    log_rem = max_rem.bit_length() - 1
    n_layers, lg = 0, log_N
    while lg > log_rem:
        lg -= log_f
        n_layers += 1
    if n_layers != len(d["layer_roots"]) or len(d["remainder"]) != m << lg:
        raise VerifierError("FRI layer structure does not match the options")
    B = lambda v: e_base(v, m)
    seed = bytes([W, log_n]) + struct.pack("<Q", P) + bytes([nq, log_b, grinding, hash_fn, ext, folding, log_rem])
    seed += b"".join(struct.pack("<Q", v) for v in air.pub) + air.pub_bytes
    coin = Coin(seed, hash_fn)
    H = coin.h
    coin.reseed(d["trace_root"])
    ta, tb, ba, bb = [], [], [], []
    for _ in range(air.nc):
        ta.append(coin.draw_e(m)); tb.append(coin.draw_e(m))
    for _ in range(air.na):
        ba.append(coin.draw_e(m)); bb.append(coin.draw_e(m))
    coin.reseed(d["cons_root"])
    z = coin.draw_e(m)

    # out-of-domain consistency over the extension
    wn = root_of_unity(log_n)
    cur, nxt, hz = _tuples(d["ood_cur"], m), _tuples(d["ood_next"], m), _tuples(d["ood_comp"], m)
    lhs = air.ood_combined_ext(d, log_n, z, ta, tb, ba, bb, m)
    rhs, zi = B(0), B(1)
    for h in hz:
        rhs = e_add(rhs, e_mul(h, zi))
        zi = e_mul(zi, z)
    if lhs != rhs:
        raise VerifierError("out-of-domain constraint evaluations are inconsistent")
    coin.reseed(H(elem_bytes(d["ood_cur"]) + elem_bytes(d["ood_next"])))
    coin.reseed(H(elem_bytes(d["ood_comp"])))
    d_alpha, d_beta = [], []
    for _ in range(W):
        d_alpha.append(coin.draw_e(m)); d_beta.append(coin.draw_e(m)); [coin.draw_e(m) for _ in range(2, CONV["deep_draws_per_register"])]
    d_delta = [coin.draw_e(m) for _ in range(ce)]
    deg_a, deg_b = coin.draw_e(m), coin.draw_e(m)
    alphas = []
    for root in d["layer_roots"]:
        coin.reseed(root)
        alphas.append(coin.draw_e(m))
    if H(elem_bytes(d["remainder"])) != d["rem_commit"]:
        raise VerifierError("remainder does not match its commitment")
    coin.reseed(d["rem_commit"])
    if grinding:
        v = struct.unpack("<Q", H(coin.seed + struct.pack("<Q", d["nonce"]))[:8])[0]
        if v & ((1 << grinding) - 1):
            raise VerifierError("proof of work not satisfied")
    coin.reseed_int(d["nonce"])
    positions = coin.draw_integers(nq, N)
    for q, pos in enumerate(positions):
        if merkle_root_from_path(H(elem_bytes(d["trace_rows"][q])), pos, d["trace_paths"][q], hash_fn) != d["trace_root"]:
            raise VerifierError("trace opening %d does not match the trace commitment" % q)
        if merkle_root_from_path(H(elem_bytes(d["cons_rows"][q])), pos, d["cons_paths"][q], hash_fn) != d["cons_root"]:
            raise VerifierError("composition opening %d does not match the constraint commitment" % q)
    # DEEP composition over the extension at the queried points
    wN = root_of_unity(log_N)
    zw, zb = e_scale(z, wn), e_pow(z, ce)
    deep = []
    for q, pos in enumerate(positions):
        x = GEN * pow(wN, pos, P) % P
        i1, i2, i3 = e_inv(e_sub(B(x), z)), e_inv(e_sub(B(x), zw)), e_inv(e_sub(B(x), zb))
        row = [from_mont(v) for v in d["trace_rows"][q]]
        crow = _tuples(d["cons_rows"][q], m)
        s1 = s2 = s3 = B(0)
        for c in range(W):
            s1 = e_add(s1, e_mul(d_alpha[c], e_sub(B(row[c]), cur[c])))
            s2 = e_add(s2, e_mul(d_beta[c], e_sub(B(row[c]), nxt[c])))
        for i in range(ce):
            s3 = e_add(s3, e_mul(d_delta[i], e_sub(crow[i], hz[i])))
        t = e_add(e_add(e_mul(s1, i1), e_mul(s2, i2)), e_mul(s3, i3))
        deep.append(e_mul(t, e_add(deg_a, e_scale(deg_b, x))))
    # FRI over the extension (rows and remainder component-major)
    cur_pos, cur_val = positions, deep
    offset, lgl = GEN, log_N
    invf = pow(folding, -1, P)
    for l in range(n_layers):
        rows_n = 1 << (lgl - log_f)
        rows, paths = d["layers"][l]
        fpos = fold_positions(cur_pos, rows_n)
        if len(fpos) != len(rows):
            raise VerifierError("layer %d: wrong number of openings" % l)
        vals = []
        for t, rp in enumerate(fpos):
            if merkle_root_from_path(H(elem_bytes(rows[t])), rp, paths[t], hash_fn) != d["layer_roots"][l]:
                raise VerifierError("layer %d opening does not match its commitment" % l)
            r = [from_mont(e) for e in rows[t]]
            vals.append([tuple(r[folding * q + k] for q in range(m)) for k in range(folding)])
        for p, v in zip(cur_pos, cur_val):
            if vals[fpos.index(p & (rows_n - 1))][p >> (lgl - log_f)] != v:
                raise VerifierError("layer %d: evaluation differs from the previous layer's folding" % l)
        wl = root_of_unity(lgl)
        zeta_inv = pow(pow(wl, rows_n, P), -1, P)
        nxt_val = []
        for t, rp in enumerate(fpos):
            x = offset * pow(wl, rp, P) % P
            r = e_scale(alphas[l], pow(x, -1, P))
            acc, rs = B(0), B(1)
            for s_ in range(folding):
                cs = B(0)
                for k in range(folding):
                    cs = e_add(cs, e_scale(vals[t][k], pow(zeta_inv, s_ * k, P)))
                acc = e_add(acc, e_mul(e_scale(cs, invf), rs))
                rs = e_mul(rs, r)
            nxt_val.append(acc)
        cur_pos, cur_val = fpos, nxt_val
        offset = pow(offset, folding, P)
        lgl -= log_f
    R = len(d["remainder"]) // m
    comps = [d["remainder"][q * R:(q + 1) * R] for q in range(m)]
    for p, v in zip(cur_pos, cur_val):
        if tuple(from_mont(comp[p]) for comp in comps) != v:
            raise VerifierError("remainder differs from the last layer's folding")
    for comp in comps:
        co = O.ntt(np.ascontiguousarray(comp, np.uint64).copy(), inverse=True)
        if any(int(v) != 0 for v in co[R // blowup:]):
            raise VerifierError("FRI remainder is not a low-degree polynomial")
    return True
