"""CPU verifier for proofs written by cstark_tx_prove -- TEST INFRASTRUCTURE (part of oracle/; never used by the product).

Restates `winterfell::verify::<TransactionAir>(proof, pub_inputs)` as called at /root/reference/src/lib.rs:144-150
(engine code absent from the reference tree: [UPSTREAM-RECALL winterfell v0.3], parity unpinned).  Steps:
  1. rebuild the public coin from context + public inputs, replay the prover's draws (csrc/prove.hip header);
  2. out-of-domain consistency: the 115 transition constraints + 4 assertions evaluated on the OOD frame at z
     (oracle C code, cso_tx_combined_from_frame) must equal sum_i z^i H_i(z^8);
  3. Merkle openings of the queried trace / composition rows against the two roots (BLAKE3);
  4. DEEP composition values at the queried points from the opened rows;
  5. FRI: layer openings, folding consistency layer to layer (factor 4), remainder commitment, remainder degree.
Field arithmetic here is Python integers on canonical values -- deliberately independent of the Montgomery code paths.
"""
import ctypes as C
import struct

import numpy as np

from . import oracle as O

P = (1 << 62) + (1 << 56) + (1 << 55) + 1
R_INV = pow(1 << 64, -1, P)
GEN = 3
TWO_ADIC_ROOT = pow(3, 131, P)  # order 2^55


class VerifierError(Exception):
    pass


def from_mont(v):
    return (int(v) * R_INV) % P


def to_mont(v):
    return (int(v) << 64) % P


def root_of_unity(log_n):
    return pow(TWO_ADIC_ROOT, 1 << (55 - log_n), P)


class Coin:
    def __init__(self, seed_bytes):
        self.seed, self.counter = O.blake3(seed_bytes), 0

    def reseed(self, digest):
        self.seed, self.counter = O.blake3(self.seed + bytes(digest)), 0

    def reseed_int(self, v):
        self.seed, self.counter = O.blake3(self.seed + struct.pack("<Q", v)), 0

    def _next(self):
        self.counter += 1
        return struct.unpack("<Q", O.blake3(self.seed + struct.pack("<Q", self.counter))[:8])[0]

    def draw(self):
        """a field element (canonical integer)"""
        while True:
            v = self._next()
            if v < P:
                return v

    def draw_integers(self, count, domain):
        out = []
        while len(out) < count:
            v = self._next() & (domain - 1)
            if v not in out:
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


def merkle_root_from_path(leaf, index, path):
    h = leaf
    for sib in path:
        h = O.blake3(sib + h) if index & 1 else O.blake3(h + sib)
        index >>= 1
    return h


def parse(proof):
    r = Reader(proof)
    if r.take(4) != b"CSTK":
        raise VerifierError("bad magic")
    d = {"version": r.u32(), "air": r.u32(), "width": r.u32(), "log_n": r.u32(), "depth": r.u32()}
    d["options"] = [r.u32() for _ in range(7)]
    if d["version"] != 1 or d["air"] != 0 or d["width"] != 94:
        raise VerifierError("unsupported proof header")
    nq, blowup = d["options"][0], d["options"][1]
    if blowup != 8 or not (10 <= d["log_n"] <= 21) or not (1 <= nq <= 128):
        raise VerifierError("unsupported parameters")
    log_N = d["log_n"] + 3
    d["trace_root"], d["cons_root"] = r.take(32), r.take(32)
    nl = r.u32()
    if nl > 16:
        raise VerifierError("bad layer count")
    d["layer_roots"] = [r.take(32) for _ in range(nl)]
    d["rem_commit"] = r.take(32)
    d["ood_cur"], d["ood_next"], d["ood_comp"] = r.elems(94), r.elems(94), r.elems(8)
    d["nonce"] = r.u64()
    d["trace_rows"] = r.elems(nq * 94).reshape(nq, 94)
    d["trace_paths"] = [[r.take(32) for _ in range(log_N)] for _ in range(nq)]
    d["cons_rows"] = r.elems(nq * 8).reshape(nq, 8)
    d["cons_paths"] = [[r.take(32) for _ in range(log_N)] for _ in range(nq)]
    d["layers"] = []
    lg = log_N
    for _ in range(nl):
        if lg < 2:
            raise VerifierError("too many layers")
        npos = r.u32()
        if npos > nq:
            raise VerifierError("bad layer opening count")
        rows = r.elems(npos * 4).reshape(npos, 4)
        paths = [[r.take(32) for _ in range(lg - 2)] for _ in range(npos)]
        d["layers"].append((rows, paths))
        lg -= 2
    rl = r.u32()
    if rl > 1024:
        raise VerifierError("bad remainder length")
    d["remainder"] = r.elems(rl)
    if r.o != len(r.b):
        raise VerifierError("trailing bytes")
    return d


def verify(proof, initial_root, final_root, options=None):
    """Raises VerifierError unless `proof` shows that a valid 94-register trace links initial_root to final_root.
    initial_root / final_root: 7 field elements each, memory form (as TransactionMetadata holds them).
    options: the 7 ProofOptions values the verifier expects (None = accept what the proof states)."""
    d = parse(proof)
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = d["options"]
    if options is not None and list(options) != d["options"]:
        raise VerifierError("proof options differ from the expected ones")
    if hash_fn != 0 or ext != 0 or folding != 4 or max_rem & (max_rem - 1) or not (128 <= max_rem <= 1024):
        raise VerifierError("unsupported options")
    log_n, depth = d["log_n"], d["depth"]
    log_b, log_N = 3, log_n + 3
    n, N, W, b = 1 << log_n, 1 << log_N, 94, 8
    log_rem = max_rem.bit_length() - 1
    n_layers, lg = 0, log_N
    while lg > log_rem:
        lg -= 2
        n_layers += 1
    if n_layers != len(d["layer_roots"]) or len(d["remainder"]) != 1 << lg:
        raise VerifierError("FRI layer structure does not match the options")
    pub = [from_mont(v) for v in list(initial_root) + list(final_root)]

    # 1. channel
    seed = bytes([W, log_n]) + struct.pack("<Q", P) + bytes([nq, log_b, grinding, hash_fn, ext, folding, log_rem])
    seed += b"".join(struct.pack("<Q", v) for v in pub)
    coin = Coin(seed)
    coin.reseed(d["trace_root"])
    cf = O.TxCoeffsStruct()
    for i in range(115):
        cf.t_alpha[i], cf.t_beta[i] = to_mont(coin.draw()), to_mont(coin.draw())
    for i in range(4):
        cf.b_alpha[i], cf.b_beta[i] = to_mont(coin.draw()), to_mont(coin.draw())
    coin.reseed(d["cons_root"])
    z = coin.draw()

    # 2. out-of-domain consistency
    pub4 = np.array([to_mont(pub[0]), to_mont(pub[1]), to_mont(pub[7]), to_mont(pub[8])], np.uint64)
    lib = O.lib()
    lib.cso_tx_combined_from_frame.restype = C.c_uint64
    cur, nxt = np.ascontiguousarray(d["ood_cur"]), np.ascontiguousarray(d["ood_next"])
    lhs = from_mont(lib.cso_tx_combined_from_frame(O._p(cur), O._p(nxt), C.byref(cf), O._p(pub4), C.c_uint(depth), C.c_uint(log_n),
                                                   C.c_uint(log_b), C.c_uint64(to_mont(z))))
    hz = [from_mont(v) for v in d["ood_comp"]]
    rhs = sum(h * pow(z, i, P) for i, h in enumerate(hz)) % P
    if lhs != rhs:
        raise VerifierError("out-of-domain constraint evaluations are inconsistent")
    coin.reseed(O.blake3(cur.tobytes() + nxt.tobytes()))
    coin.reseed(O.blake3(np.ascontiguousarray(d["ood_comp"]).tobytes()))
    d_alpha, d_beta = [], []
    for _ in range(W):
        d_alpha.append(coin.draw()); d_beta.append(coin.draw()); coin.draw()
    d_delta = [coin.draw() for _ in range(b)]
    deg_a, deg_b = coin.draw(), coin.draw()
    alphas = []
    for root in d["layer_roots"]:
        coin.reseed(root)
        alphas.append(coin.draw())
    if O.blake3(np.ascontiguousarray(d["remainder"]).tobytes()) != d["rem_commit"]:
        raise VerifierError("remainder does not match its commitment")
    coin.reseed(d["rem_commit"])
    if grinding:
        v = struct.unpack("<Q", O.blake3(coin.seed + struct.pack("<Q", d["nonce"]))[:8])[0]
        if v & ((1 << grinding) - 1):
            raise VerifierError("proof of work not satisfied")
    coin.reseed_int(d["nonce"])
    positions = coin.draw_integers(nq, N)

    # 3. trace / composition openings
    for q, pos in enumerate(positions):
        if merkle_root_from_path(O.blake3(d["trace_rows"][q].tobytes()), pos, d["trace_paths"][q]) != d["trace_root"]:
            raise VerifierError("trace opening %d does not match the trace commitment" % q)
        if merkle_root_from_path(O.blake3(d["cons_rows"][q].tobytes()), pos, d["cons_paths"][q]) != d["cons_root"]:
            raise VerifierError("composition opening %d does not match the constraint commitment" % q)

    # 4. DEEP composition at the queried points
    wN, wn = root_of_unity(log_N), root_of_unity(log_n)
    tz = [from_mont(v) for v in d["ood_cur"]]
    tzw = [from_mont(v) for v in d["ood_next"]]
    zw, zb = z * wn % P, pow(z, b, P)
    deep = []
    for q, pos in enumerate(positions):
        x = GEN * pow(wN, pos, P) % P
        i1, i2, i3 = pow(x - z, -1, P), pow(x - zw, -1, P), pow(x - zb, -1, P)
        row = [from_mont(v) for v in d["trace_rows"][q]]
        crow = [from_mont(v) for v in d["cons_rows"][q]]
        acc = 0
        for c in range(W):
            acc += d_alpha[c] * (row[c] - tz[c]) % P * i1 + d_beta[c] * (row[c] - tzw[c]) % P * i2
        for i in range(b):
            acc += d_delta[i] * (crow[i] - hz[i]) % P * i3
        deep.append(acc % P * ((deg_a + deg_b * x) % P) % P)

    # 5. FRI
    cur_pos, cur_val = positions, deep
    offset, lgl = GEN, log_N
    inv4 = pow(4, -1, P)
    for l in range(n_layers):
        rows_n = 1 << (lgl - 2)
        rows, paths = d["layers"][l]
        fpos = fold_positions(cur_pos, rows_n)
        if len(fpos) != len(rows):
            raise VerifierError("layer %d: wrong number of openings" % l)
        for t, rp in enumerate(fpos):
            if merkle_root_from_path(O.blake3(rows[t].tobytes()), rp, paths[t]) != d["layer_roots"][l]:
                raise VerifierError("layer %d opening does not match its commitment" % l)
        for p, v in zip(cur_pos, cur_val):
            if from_mont(rows[fpos.index(p & (rows_n - 1))][p >> (lgl - 2)]) != v:
                raise VerifierError("layer %d: evaluation differs from the previous layer's folding" % l)
        # fold each opened row: 4 evaluations on the coset x * <zeta>, zeta = w^(N_l/4)
        wl = root_of_unity(lgl)
        zeta_inv = pow(pow(wl, rows_n, P), -1, P)
        nxt_val = []
        for t, rp in enumerate(fpos):
            v = [from_mont(e) for e in rows[t]]
            x = offset * pow(wl, rp, P) % P
            r = alphas[l] * pow(x, -1, P) % P
            acc, rs = 0, 1
            for s in range(4):
                cs = sum(v[k] * pow(zeta_inv, s * k, P) for k in range(4)) % P * inv4 % P
                acc += cs * rs
                rs = rs * r % P
            nxt_val.append(acc % P)
        cur_pos, cur_val = fpos, nxt_val
        offset = pow(offset, 4, P)
        lgl -= 2
    rem = [from_mont(v) for v in d["remainder"]]
    for p, v in zip(cur_pos, cur_val):
        if rem[p] != v:
            raise VerifierError("remainder differs from the last layer's folding")
    # remainder degree: evaluations over offset * <w_R> must interpolate to degree < R / blowup
    R = len(rem)
    co = O.ntt(np.array([to_mont(v) for v in rem], np.uint64), inverse=True)
    off_inv = pow(offset, -1, P)
    max_deg_plus_1 = R // blowup
    if any(int(v) != 0 for v in co[max_deg_plus_1:]):  # the offset scaling does not change which coefficients vanish
        raise VerifierError("FRI remainder is not a low-degree polynomial")
    return True
