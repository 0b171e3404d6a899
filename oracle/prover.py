"""CPU prover -- TEST INFRASTRUCTURE (part of oracle/; never used by the product).

Restates the whole `TransactionExample::prove` pipeline (/root/reference/src/lib.rs:116-141: build_trace + the engine's
Prover::prove [UPSTREAM-RECALL winterfell v0.3, parity unpinned]) from the oracle's stage functions and the channel of
verifier.py, and serialises the proof in the layout of include/cstark.h.  The GPU prover must produce the same bytes."""
import struct

import numpy as np

from . import oracle as O
from . import verifier as V


def _mont(v):
    return V.to_mont(v)


class ShardedProver:
    """The base-field prover as the phases of the product's sharded entry points (cstark_tx_shard_*, include/cstark.h): one proof
    across `world` ranks by LDE coset.  Rank r owns cosets [k0, k0 + nk); prove() below is the one-rank case.  A phase's output that
    other ranks need is returned as a numpy array; the caller exchanges it (all-gather / broadcast / sum)."""

    def __init__(self, w, options, k0=0, nk=8):
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = options
        assert blowup == 8 and hash_fn in (0, 1) and ext == 0 and folding == 4
        self.w, self.options, self.k0, self.nk = w, tuple(options), k0, nk
        self.H = lambda data: O.digest(data, hash_fn)

    def commit(self):
        """trace + interpolation (replicated), extension and row hashes of the owned cosets -> digests [nk][n][32]"""
        hash_fn = self.options[3]
        self.trace = O.tx_build_trace(self.w)
        self.n = self.trace.shape[1]
        self.log_n = self.n.bit_length() - 1
        self.coeffs = O.interpolate_columns(self.trace.copy())
        self.lde = O.lde_columns(self.coeffs, 3, k0=self.k0, nk=self.nk)
        return np.stack([O.hash_rows(self.lde[i:i + 1], 0, hash_fn=hash_fn) for i in range(self.nk)])

    def evaluate(self, leaves_all):
        """leaves_all [8][n][32] coset-major -> tree, channel, coefficients, merged evaluations of the owned cosets [nk][n]"""
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = self.options
        n, log_n, trace = self.n, self.log_n, self.trace
        natural = np.ascontiguousarray(np.transpose(leaves_all, (1, 0, 2))).reshape(8 * n, 32)  # leaf 8 j + k
        self.tnodes = O.merkle_build(natural, hash_fn)
        self.trace_root = self.tnodes[1].tobytes()
        self.log_rem = max_rem.bit_length() - 1
        pub_m = [int(trace[58 + i, 0]) for i in range(7)] + [int(trace[58 + i, n - 1]) for i in range(7)]  # src/prover.rs:106-129
        pub = [V.from_mont(v) for v in pub_m]
        seed = bytes([94, log_n]) + struct.pack("<Q", V.P) + bytes([nq, 3, grinding, hash_fn, ext, folding, self.log_rem])
        seed += b"".join(struct.pack("<Q", v) for v in pub)
        self.coin = coin = V.Coin(seed, hash_fn)
        coin.reseed(self.trace_root)
        cf = O.TxCoeffsStruct()
        for i in range(115):
            cf.t_alpha[i], cf.t_beta[i] = _mont(coin.draw()), _mont(coin.draw())
        for i in range(4):
            cf.b_alpha[i], cf.b_beta[i] = _mont(coin.draw()), _mont(coin.draw())
        pub4 = np.array([pub_m[0], pub_m[1], pub_m[7], pub_m[8]], np.uint64)
        return O.tx_evaluate_constraints(self.lde, cf, pub4, self.w.depth, 3, k0=self.k0)

    def compose(self, combined):
        """combined [8][n] (all cosets); the owner of coset 0: composition .. FRI -> query positions"""
        assert self.k0 == 0
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = self.options
        H, coin, log_n, n = self.H, self.coin, self.log_n, self.n
        log_b, b, W = 3, 8, 94
        log_N, N = log_n + 3, n * 8
        self.ccoef = ccoef = O.composition_columns(combined)
        self.clde = clde = O.lde_columns(ccoef, log_b)
        self.cnodes = O.merkle_build(O.hash_rows(clde, log_b, hash_fn=hash_fn), hash_fn)
        self.cons_root = self.cnodes[1].tobytes()
        coin.reseed(self.cons_root)
        z = coin.draw()
        zm = _mont(z)
        zw = _mont(z * V.root_of_unity(log_n) % V.P)
        zb = _mont(pow(z, b, V.P))
        self.ood_trace = ood_trace = O.evaluate_polys_at(self.coeffs, [zm, zw])          # [2][94]
        self.ood_comp = ood_comp = O.evaluate_polys_at(ccoef, [zb])[0]                   # [8]
        coin.reseed(H(V.elem_bytes(ood_trace)))
        coin.reseed(H(V.elem_bytes(ood_comp)))
        d_alpha, d_beta = [], []
        for _ in range(W):
            d_alpha.append(_mont(coin.draw())); d_beta.append(_mont(coin.draw())); [coin.draw() for _ in range(2, V.CONV["deep_draws_per_register"])]
        d_delta = [_mont(coin.draw()) for _ in range(b)]
        deg_a, deg_b = _mont(coin.draw()), _mont(coin.draw())
        # the DEEP composition polynomial has degree < n: coset 0 of the extended trace determines it (as the product computes it)
        if self.nk == 8:
            deep = O.deep_composition(self.lde, clde, zm, ood_trace.reshape(-1), ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
        else:
            d0 = O.deep_composition(self.lde[:1], clde[:1], zm, ood_trace.reshape(-1), ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
            deep = O.lde_columns(O.interpolate_columns(d0.reshape(1, n).copy()), log_b, offset=int(O.to_mont([1])[0]))[:, 0, :]
        layer = np.ascontiguousarray(deep.T).reshape(-1)            # natural order i = 8 j + k
        self.layers, self.trees, self.roots = [], [], []
        offset, lg = V.GEN, log_N
        while lg > self.log_rem:
            rows = 1 << (lg - 2)
            nodes = O.merkle_build(O.hash_rows(layer.reshape(1, 4, rows), 0, hash_fn=hash_fn), hash_fn)
            self.layers.append(layer); self.trees.append(nodes); self.roots.append(nodes[1].tobytes())
            coin.reseed(self.roots[-1])
            alpha = coin.draw()
            layer = O.fri_fold4(layer, _mont(offset), _mont(alpha))
            offset = pow(offset, 4, V.P)
            lg -= 2
        self.remainder = layer
        self.rem_commit = H(V.elem_bytes(layer))
        coin.reseed(self.rem_commit)
        nonce = 1
        while grinding and struct.unpack("<Q", H(coin.seed + struct.pack("<Q", nonce))[:8])[0] & ((1 << grinding) - 1):
            nonce += 1
        self.nonce = nonce
        coin.reseed_int(nonce)
        self.positions = coin.draw_integers(nq, N)
        return np.array(self.positions, np.uint32)

    def open_rows(self, positions):
        """rows of the extended trace at the positions that lie in the owned cosets, zeros elsewhere: [nq][94]"""
        out = np.zeros((len(positions), 94), np.uint64)
        for q, p in enumerate(positions):
            k, j = int(p) & 7, int(p) >> 3
            if self.k0 <= k < self.k0 + self.nk:
                out[q] = self.lde[k - self.k0, :, j]
        return out

    def finish(self, rows):
        """rows [nq][94] complete -> proof bytes (layout: include/cstark.h)"""
        positions, log_n = self.positions, self.log_n
        log_N = log_n + 3

        def path(nodes, leaves_log, pos):
            L = 1 << leaves_log
            return b"".join(nodes[((L + pos) >> lvl) ^ 1].tobytes() for lvl in range(leaves_log))

        def row(tab, pos):  # tab [b][width][n] coset-major
            return np.ascontiguousarray(tab[pos & 7, :, pos >> 3]).tobytes()

        roots = self.roots
        out = [b"CSTK", struct.pack("<IIIII", 1, 0, 94, log_n, self.w.depth), struct.pack("<7I", *self.options),
               self.trace_root, self.cons_root, struct.pack("<I", len(roots))] + roots + [self.rem_commit, self.ood_trace.tobytes(),
                                                                                         self.ood_comp.tobytes(), struct.pack("<Q", self.nonce)]
        out += [np.ascontiguousarray(rows[q]).tobytes() for q in range(len(positions))] + [path(self.tnodes, log_N, p) for p in positions]
        out += [row(self.clde, p) for p in positions] + [path(self.cnodes, log_N, p) for p in positions]
        cur, lg = positions, log_N
        for l in range(len(self.layers)):
            rows_l = 1 << (lg - 2)
            fpos = V.fold_positions(cur, rows_l)
            out.append(struct.pack("<I", len(fpos)))
            tab = self.layers[l].reshape(4, rows_l)
            out += [np.ascontiguousarray(tab[:, p]).tobytes() for p in fpos]
            out += [path(self.trees[l], lg - 2, p) for p in fpos]
            cur = fpos
            lg -= 2
        out += [struct.pack("<I", self.remainder.size), self.remainder.tobytes()]
        return b"".join(out)


def prove(w, options=(42, 8, 0, 0, 0, 4, 256)):
    if options[4] in (1, 2):
        return prove_ext(w, options)
    p = ShardedProver(w, options)
    leaves = p.commit()
    combined = p.evaluate(leaves)
    positions = p.compose(combined)
    return p.finish(p.open_rows(positions))


def prove_ext(w, options):
    """FieldExtension::Quadratic / Cubic: base-field trace, everything drawn from the coin in the degree-m extension (oracle/ext.c).
    Layout differences: out-of-domain values are m-tuples; composition rows hold 8 m-tuples; FRI rows and the remainder are
    component-major (component 0 of the four points, then component 1, ...)."""
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = options
    assert blowup == 8 and hash_fn in (0, 1) and ext in (1, 2) and folding == 4
    m = ext + 1
    H = lambda data: O.digest(data, hash_fn)
    log_b, b, W = 3, 8, 94
    trace = O.tx_build_trace(w)
    n = trace.shape[1]
    log_n = n.bit_length() - 1
    log_N, N = log_n + 3, n * 8
    log_rem = max_rem.bit_length() - 1
    pub_m = [int(trace[58 + i, 0]) for i in range(7)] + [int(trace[58 + i, n - 1]) for i in range(7)]
    pub = [V.from_mont(v) for v in pub_m]
    coeffs = O.interpolate_columns(trace.copy())
    lde = O.lde_columns(coeffs, log_b)
    tnodes = O.merkle_build(O.hash_rows(lde, log_b, hash_fn=hash_fn), hash_fn)
    trace_root = tnodes[1].tobytes()
    seed = bytes([W, log_n]) + struct.pack("<Q", V.P) + bytes([nq, log_b, grinding, hash_fn, ext, folding, log_rem])
    seed += b"".join(struct.pack("<Q", v) for v in pub)
    coin = V.Coin(seed, hash_fn)
    coin.reseed(trace_root)
    cfs = [O.TxCoeffsStruct() for _ in range(m)]
    for i in range(115):
        a, bt = coin.draw_e(m), coin.draw_e(m)
        for k in range(m):
            cfs[k].t_alpha[i], cfs[k].t_beta[i] = _mont(a[k]), _mont(bt[k])
    for i in range(4):
        a, bt = coin.draw_e(m), coin.draw_e(m)
        for k in range(m):
            cfs[k].b_alpha[i], cfs[k].b_beta[i] = _mont(a[k]), _mont(bt[k])
    pub4 = np.array([pub_m[0], pub_m[1], pub_m[7], pub_m[8]], np.uint64)
    # coefficients multiply base-field values: the components of the merged evaluations are independent base-field merges
    cc = [O.composition_columns(O.tx_evaluate_constraints(lde, cfs[k], pub4, w.depth, log_b)) for k in range(m)]
    ccoef = np.ascontiguousarray(np.stack(cc, axis=1).reshape(m * b, n))  # column m i + k = component k of H_i
    clde = O.lde_columns(ccoef, log_b)
    cnodes = O.merkle_build(O.hash_rows(clde, log_b, hash_fn=hash_fn), hash_fn)
    cons_root = cnodes[1].tobytes()
    coin.reseed(cons_root)

    z = coin.draw_e(m)
    zw = V.e_scale(z, V.root_of_unity(log_n))
    zb = V.e_pow(z, b)
    ood_cur = O.evaluate_polys_at_ext(coeffs, V.e_mont(z))
    ood_next = O.evaluate_polys_at_ext(coeffs, V.e_mont(zw))
    raw = O.evaluate_polys_at_ext(ccoef, V.e_mont(zb))                     # each component polynomial at z^8
    ood_comp = np.zeros((b, m), np.uint64)
    for i in range(b):
        h, gk = V.e_base(0, m), V.e_base(1, m)
        for k in range(m):                                                  # H_i = sum_k root^k H_i,k
            h = V.e_add(h, V.e_mul(gk, tuple(V.from_mont(v) for v in raw[m * i + k])))
            gk = V.e_mul(gk, V.e_gen(m))
        ood_comp[i] = V.e_mont(h)
    ood_trace = np.concatenate([ood_cur, ood_next])
    coin.reseed(H(V.elem_bytes(ood_trace)))
    coin.reseed(H(V.elem_bytes(ood_comp)))
    d_alpha, d_beta = np.zeros((W, m), np.uint64), np.zeros((W, m), np.uint64)
    for c in range(W):
        d_alpha[c], d_beta[c] = V.e_mont(coin.draw_e(m)), V.e_mont(coin.draw_e(m))
        for _ in range(2, V.CONV["deep_draws_per_register"]):
            coin.draw_e(m)
    d_delta = np.array([V.e_mont(coin.draw_e(m)) for _ in range(b)], np.uint64)
    deg_a, deg_b = V.e_mont(coin.draw_e(m)), V.e_mont(coin.draw_e(m))
    deep = O.deep_composition_ext(lde, clde, V.e_mont(z), ood_trace, ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
    layer = np.ascontiguousarray(np.stack([np.ascontiguousarray(deep[k].T).reshape(-1) for k in range(m)]))  # [m][N] natural order

    layers, trees, roots = [], [], []
    offset, lg = V.GEN, log_N
    while lg > log_rem:
        rows = 1 << (lg - 2)
        nodes = O.merkle_build(O.hash_rows(layer.reshape(1, 4 * m, rows), 0, hash_fn=hash_fn), hash_fn)
        layers.append(layer); trees.append(nodes); roots.append(nodes[1].tobytes())
        coin.reseed(roots[-1])
        alpha = coin.draw_e(m)
        layer = O.fri_fold4_ext(layer, _mont(offset), V.e_mont(alpha))
        offset = pow(offset, 4, V.P)
        lg -= 2
    remainder = layer
    rem_commit = H(V.elem_bytes(remainder))
    coin.reseed(rem_commit)
    nonce = 1
    while grinding and struct.unpack("<Q", H(coin.seed + struct.pack("<Q", nonce))[:8])[0] & ((1 << grinding) - 1):
        nonce += 1
    coin.reseed_int(nonce)
    positions = coin.draw_integers(nq, N)

    def path(nodes, leaves_log, pos):
        L = 1 << leaves_log
        return b"".join(nodes[((L + pos) >> lvl) ^ 1].tobytes() for lvl in range(leaves_log))

    def row(tab, pos):
        return np.ascontiguousarray(tab[pos & 7, :, pos >> 3]).tobytes()

    out = [b"CSTK", struct.pack("<IIIII", 1, 0, W, log_n, w.depth), struct.pack("<7I", *options),
           trace_root, cons_root, struct.pack("<I", len(roots))] + roots + [rem_commit, ood_trace.tobytes(), ood_comp.tobytes(),
                                                                           struct.pack("<Q", nonce)]
    out += [row(lde, p) for p in positions] + [path(tnodes, log_N, p) for p in positions]
    out += [row(clde, p) for p in positions] + [path(cnodes, log_N, p) for p in positions]
    cur, lg = positions, log_N
    for l in range(len(layers)):
        rows = 1 << (lg - 2)
        fpos = V.fold_positions(cur, rows)
        out.append(struct.pack("<I", len(fpos)))
        tab = layers[l].reshape(4 * m, rows)
        out += [np.ascontiguousarray(tab[:, p]).tobytes() for p in fpos]
        out += [path(trees[l], lg - 2, p) for p in fpos]
        cur = fpos
        lg -= 2
    out += [struct.pack("<I", remainder.shape[1]), remainder.tobytes()]
    return b"".join(out)
