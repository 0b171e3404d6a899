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


# ---- what differs between the AIRs (the product's AirJob, csrc/prove.hip): how the trace is built, what seeds the channel, how the merged
# ---- constraint evaluations are produced from the extended trace and ONE set of base-field coefficients --------------------------------
class TxJob:
    """TransactionAir (src/air.rs:64-189; TransactionProver src/prover.rs:20-134)"""
    air, width, ce, nc, na = 0, 94, 8, 115, 4

    def __init__(self, w):
        self.w, self.item = w, w.depth

    def build(self):
        return O.tx_build_trace(self.w)

    def public(self, trace):
        n = trace.shape[1]
        self.pub_m = [int(trace[58 + i, 0]) for i in range(7)] + [int(trace[58 + i, n - 1]) for i in range(7)]  # src/prover.rs:106-129
        return self.pub_m, b""

    def combine(self, trace, lde, ta, tb, ba, bb, k0=0):
        cf = O.TxCoeffsStruct()
        for i in range(115):
            cf.t_alpha[i], cf.t_beta[i] = int(ta[i]), int(tb[i])
        for i in range(4):
            cf.b_alpha[i], cf.b_beta[i] = int(ba[i]), int(bb[i])
        pub4 = np.array([self.pub_m[0], self.pub_m[1], self.pub_m[7], self.pub_m[8]], np.uint64)  # get_assertions, src/air.rs:175-184
        return O.tx_evaluate_constraints(lde, cf, pub4, self.w.depth, 3, k0=k0)


# combine(trace, ce_lde, ...): ce_lde [ce][width][n] is the trace on the CONSTRAINT-EVALUATION domain -- the ce = 2^log_ce cosets
# g w_(ce n)^k <w_n> the AIR's degrees need -- which is every (blowup / ce)-th coset of the LDE domain (the same points, whatever the blowup
# factor); the merged evaluations come back as [ce][n].  (TxJob also takes a window [k0, k0 + nk) of its 8 cosets: sharded proofs.)


class MerkleJob:
    """MerkleAir (src/merkle/update/air.rs:36-177; MerkleProver src/merkle/update/prover.rs:19-116; prove: src/merkle/update/mod.rs:81-106)"""
    air, width, ce, nc, na = 1, 65, 4, 106, 14

    def __init__(self, w):
        self.w, self.item = w, w.depth

    def build(self):
        return O.merkle_build_trace(self.w)

    def public(self, trace):
        n = trace.shape[1]
        return [int(trace[58 + i, 0]) for i in range(7)] + [int(trace[58 + i, n - 1]) for i in range(7)], b""

    def combine(self, trace, lde, ta, tb, ba, bb, k0=0):
        assert k0 == 0 and lde.shape[0] == 4
        log_n = trace.shape[1].bit_length() - 1
        if not hasattr(self, "ev"):  # once per proof (extension proofs merge m coefficient sets)
            ptab = O.periodic_table(O.merkle_periodic_columns(self.w.depth), log_n, 2)
            self.ev = O.air_evaluate_transitions(O.AIR_MERKLE, lde, ptab, 106)
        return O.air_combine(O.merkle_desc(trace), lde, self.ev, ta, tb, ba, bb, 2)


class RangeJob:
    """RangeProofAir (src/range/air.rs:23-105; RangeProver src/range/prover.rs:15-59; prove: src/range/mod.rs:75-100).  words / log_n:
    the synthetic long accumulator of cstark_range_prove_bits; number: a field element in memory form (the reference's 64-row proof)."""
    air, width, ce, nc, na, item = 3, 2, 2, 2, 2, 0

    def __init__(self, number=None, words=None, log_n=6):
        self.number, self.words, self.log_n = number, words, log_n

    def build(self):
        if self.words is not None:
            trace, self.number = O.range_build_trace_bits(self.words, self.log_n)
            return trace
        return O.range_build_trace(V.from_mont(self.number))

    def public(self, trace):
        return [int(self.number)], b""

    def combine(self, trace, lde, ta, tb, ba, bb, k0=0):
        assert k0 == 0 and lde.shape[0] == 2
        if not hasattr(self, "ev"):
            self.ev = O.air_evaluate_transitions(O.AIR_RANGE, lde, None, 2)
        return O.air_combine(O.range_desc(V.from_mont(self.number)), lde, self.ev, ta, tb, ba, bb, 1)


class RescueJob:
    """RescueAir, the hash-chain AIR of benches/rescue.rs:145-356 (RescueProver :266-356; prove :66-86; options :370-378: blowup 4)"""
    air, width, ce, nc, na = 4, 14, 4, 14, 14

    def __init__(self, seed, chain_length):
        self.seed, self.item = np.ascontiguousarray(seed, np.uint64), int(chain_length)

    def build(self):
        return O.rescue_chain_build_trace(self.seed, self.item)

    def public(self, trace):  # seed then result: the first / last row of registers 0..6 (get_pub_inputs :331-354)
        n = trace.shape[1]
        return [int(trace[i, 0]) for i in range(7)] + [int(trace[i, n - 1]) for i in range(7)], b""

    def combine(self, trace, lde, ta, tb, ba, bb, k0=0):
        assert k0 == 0 and lde.shape[0] == 4
        log_n = trace.shape[1].bit_length() - 1
        if not hasattr(self, "ev"):
            ptab = O.periodic_table(O.rescue_chain_periodic_columns(), log_n, 2)
            self.ev = O.air_evaluate_transitions(O.AIR_RESCUE_CHAIN, lde, ptab, 14)
        return O.air_combine(O.rescue_chain_desc(trace), lde, self.ev, ta, tb, ba, bb, 2)


class SchnorrJob:
    """SchnorrAir (src/schnorr/air.rs:41-300; SchnorrProver src/schnorr/prover.rs:21-90; prove: src/schnorr/mod.rs:143-172)"""
    air, width, ce, nc, na = 2, 56, 8, 56, 61

    def __init__(self, w):
        self.w, self.item = w, w.n_sig

    def build(self):
        return O.schnorr_build_trace(self.w)

    def public(self, trace):  # messages [n][28] then R.x [n][6] (src/schnorr/air.rs:29-38), then the s halves verbatim
        w = self.w
        return [int(v) for v in w.messages.reshape(-1)] + [int(v) for v in w.sig_rx.reshape(-1)], w.sig_s.tobytes()

    def combine(self, trace, lde, ta, tb, ba, bb, k0=0):
        assert k0 == 0 and lde.shape[0] == 8
        w = self.w
        log_n = trace.shape[1].bit_length() - 1
        if not hasattr(self, "ev"):
            aux_lde = O.lde_columns(O.interpolate_columns(O.schnorr_aux_columns(w)), 3)  # not committed: both sides derive them
            self.ev = O.schnorr_evaluate_transitions(lde, aux_lde, O.periodic_table(O.schnorr_mask_columns(), log_n, 3))
            self.avals = O.lde_columns(O.schnorr_assertion_polys(w, log_n), 3)
        return O.air_combine(O.schnorr_desc(w), lde, self.ev, ta, tb, ba, bb, 3, avals=self.avals)


# the ProofOptions values the reference itself passes: blowup 4 / 8 (src/merkle/update/tests.rs:41-44, src/range/tests.rs:87-90,
# benches/rescue.rs:370-378, src/lib.rs:78-86), any blowup / FRI folding factor from the command line (examples/state-transition.rs:33-34,
# :46-47).  Supported here and by the product: blowup 2 .. 16 (at least the AIR's constraint-evaluation blowup), folding 4 / 8 / 16.
BLOWUPS, FOLDINGS = (2, 4, 8, 16), (4, 8, 16)


def check_options(options, job):
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = options
    assert blowup in BLOWUPS and blowup >= job.ce, "blowup factor below the AIR's constraint-evaluation blowup"
    assert folding in FOLDINGS and hash_fn in (0, 1) and ext in (0, 1, 2)
    assert max_rem in (128, 256, 512, 1024)


def num_fri_layers(log_N, log_rem, log_f):
    """FriOptions::num_fri_layers [UPSTREAM-RECALL]: fold while the domain exceeds fri_max_remainder"""
    layers = 0
    while log_N > log_rem:
        log_N -= log_f
        layers += 1
    return layers


def _path(nodes, leaves_log, pos):
    L = 1 << leaves_log
    return b"".join(nodes[((L + pos) >> lvl) ^ 1].tobytes() for lvl in range(leaves_log))


class ShardedProver:
    """The base-field prover as the phases of the product's sharded entry points (cstark_tx_shard_*, include/cstark.h): one proof
    across `world` ranks by LDE coset.  Rank r owns cosets [k0, k0 + nk); prove() below is the one-rank case (nk = blowup: any supported
    blowup factor; several ranks: blowup 8).  A phase's output that other ranks need is returned as a numpy array; the caller exchanges
    it (all-gather / broadcast / sum)."""

    def __init__(self, w, options, k0=0, nk=None, job=None):
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = options
        self.job = job if job is not None else TxJob(w)
        check_options(options, self.job)
        assert ext == 0
        self.b, self.log_b, self.log_f = blowup, blowup.bit_length() - 1, folding.bit_length() - 1
        nk = blowup if nk is None else nk
        assert nk == blowup or (blowup == 8 and self.job.ce == 8), "sharded proofs: blowup 8, an AIR that is evaluated on all eight cosets"
        self.w, self.options, self.k0, self.nk = w, tuple(options), k0, nk
        self.H = lambda data: O.digest(data, hash_fn)

    def commit(self):
        """trace + interpolation (replicated), extension and row hashes of the owned cosets.  The nk leaves b j + k0 .. b j + k0 + nk - 1 of
        row j are a complete subtree of the trace tree: the rank hashes its bottom log2(nk) levels itself and hands over the n subtree
        roots, [1][n][32] -- the same 32 n bytes at every world size (the subtree's lower nodes stay here for the openings)."""
        hash_fn = self.options[3]
        self.trace = self.job.build()
        self.n = self.trace.shape[1]
        self.log_n = self.n.bit_length() - 1
        self.coeffs = O.interpolate_columns(self.trace.copy())
        self.lde = O.lde_columns(self.coeffs, self.log_b, k0=self.k0, nk=self.nk)
        log_nk = self.nk.bit_length() - 1
        leaves = O.hash_rows(self.lde, log_nk, hash_fn=hash_fn)                      # leaf nk j + (k - k0): a tree of its own
        self.sub = O.merkle_build(leaves, hash_fn)                                    # heap: leaves at [nk n, 2 nk n); the level with n nodes at [n, 2 n)
        return np.ascontiguousarray(self.sub[self.n:2 * self.n]).reshape(1, self.n, 32)

    def evaluate(self, roots_all):
        """roots_all [world][n][32]: every rank's subtree roots -> the upper levels of the tree, channel, coefficients, merged evaluations:
        [ce][n] on the constraint-evaluation domain (one rank), or of the owned cosets (several ranks)"""
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = self.options
        n, log_n, trace, b = self.n, self.log_n, self.trace, self.b
        world = roots_all.shape[0]
        assert world * self.nk == b
        upper = np.ascontiguousarray(np.transpose(roots_all, (1, 0, 2))).reshape(world * n, 32)  # node of the level with world n nodes: world j + r
        self.tnodes = O.merkle_build(upper, hash_fn)  # the tree's levels from there up: the same heap indices as the whole tree's top
        self.trace_root = self.tnodes[1].tobytes()
        self.log_rem = max_rem.bit_length() - 1
        job = self.job
        pub_m, pub_bytes = job.public(trace)
        pub = [V.from_mont(v) for v in pub_m]
        seed = bytes([job.width, log_n]) + struct.pack("<Q", V.P) + bytes([nq, self.log_b, grinding, hash_fn, ext, folding, self.log_rem])
        seed += b"".join(struct.pack("<Q", v) for v in pub) + pub_bytes
        self.coin = coin = V.Coin(seed, hash_fn)
        coin.reseed(self.trace_root)
        ta, tb, ba, bb = (np.zeros(k, np.uint64) for k in (job.nc, job.nc, job.na, job.na))
        for i in range(job.nc):
            ta[i], tb[i] = _mont(coin.draw()), _mont(coin.draw())
        for i in range(job.na):
            ba[i], bb[i] = _mont(coin.draw()), _mont(coin.draw())
        if self.nk == b:  # the constraint-evaluation domain: every (b / ce)-th coset of the LDE domain
            return job.combine(trace, np.ascontiguousarray(self.lde[::b // job.ce]), ta, tb, ba, bb)
        own = job.combine(trace, self.lde, ta, tb, ba, bb, k0=self.k0)
        if self.nk not in (2, 4):
            return own
        # A rank with 2 or 4 cosets hands over ROWS as the product does (cstark_tx_shard_rows): its even cosets, then a share of each
        # of the four odd cosets whose sum over the ranks is that coset's evaluations.  The product's shares come out of the
        # degree-split evaluation (its own even cosets' part of every split polynomial, extended); here the share is simply the
        # direct evaluation on the rank that holds the coset and zero elsewhere -- the same sums, which is all the driver relies on.
        rows = np.zeros((self.nk // 2 + 4, self.n), np.uint64)
        rows[:self.nk // 2] = own[0::2]
        for i in range(1, self.nk, 2):
            rows[self.nk // 2 + (self.k0 + i) // 2] = own[i]
        return rows

    def compose(self, combined):
        """combined: the merged evaluations on the constraint-evaluation domain [ce][n] (several ranks: their rows of evaluate() side by
        side); the owner of coset 0: composition .. FRI -> query positions"""
        assert self.k0 == 0
        if self.nk in (2, 4) and self.nk != self.b and combined.shape[0] != 8:
            world, nkc, rows = 8 // self.nk, self.nk // 2, self.nk // 2 + 4
            parts = combined.reshape(world, rows, self.n)
            full = np.zeros((8, self.n), np.uint64)
            for k in range(0, 8, 2):
                full[k] = parts[(k // 2) // nkc, (k // 2) % nkc]
            for k in range(1, 8, 2):
                acc = np.zeros(self.n, np.uint64)
                for r in range(world):
                    acc = O.fp_add(acc, parts[r, nkc + k // 2])
                full[k] = acc
            combined = full
        nq, blowup, grinding, hash_fn, ext, folding, max_rem = self.options
        H, coin, log_n, n = self.H, self.coin, self.log_n, self.n
        log_b, b, W, ce, log_f = self.log_b, self.b, self.job.width, self.job.ce, self.log_f
        log_N, N = log_n + log_b, n * b
        assert combined.shape == (ce, n)
        self.ccoef = ccoef = O.composition_columns(np.ascontiguousarray(combined))
        self.clde = clde = O.lde_columns(ccoef, log_b)
        self.cnodes = O.merkle_build(O.hash_rows(clde, log_b, hash_fn=hash_fn), hash_fn)
        self.cons_root = self.cnodes[1].tobytes()
        coin.reseed(self.cons_root)
        z = coin.draw()
        zm = _mont(z)
        zw = _mont(z * V.root_of_unity(log_n) % V.P)
        zb = _mont(pow(z, ce, V.P))
        self.ood_trace = ood_trace = O.evaluate_polys_at(self.coeffs, [zm, zw])          # [2][W]
        self.ood_comp = ood_comp = O.evaluate_polys_at(ccoef, [zb])[0]                   # [ce]
        coin.reseed(H(V.elem_bytes(ood_trace)))
        coin.reseed(H(V.elem_bytes(ood_comp)))
        d_alpha, d_beta = [], []
        for _ in range(W):
            d_alpha.append(_mont(coin.draw())); d_beta.append(_mont(coin.draw())); [coin.draw() for _ in range(2, V.CONV["deep_draws_per_register"])]
        d_delta = [_mont(coin.draw()) for _ in range(ce)]
        deg_a, deg_b = _mont(coin.draw()), _mont(coin.draw())
        # the DEEP composition polynomial has degree < n: coset 0 of the extended trace determines it (as the product computes it)
        if self.nk == b:
            deep = O.deep_composition(self.lde, clde, zm, ood_trace.reshape(-1), ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
        else:
            d0 = O.deep_composition(self.lde[:1], clde[:1], zm, ood_trace.reshape(-1), ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
            deep = O.lde_columns(O.interpolate_columns(d0.reshape(1, n).copy()), log_b, offset=int(O.to_mont([1])[0]))[:, 0, :]
        layer = np.ascontiguousarray(deep.T).reshape(-1)            # natural order i = b j + k
        self.layers, self.trees, self.roots = [], [], []
        offset, lg = V.GEN, log_N
        while lg > self.log_rem:
            rows = 1 << (lg - log_f)  # row i of a layer: the `folding` evaluations { e[i + t rows] } that fold into position i
            nodes = O.merkle_build(O.hash_rows(layer.reshape(1, folding, rows), 0, hash_fn=hash_fn), hash_fn)
            self.layers.append(layer); self.trees.append(nodes); self.roots.append(nodes[1].tobytes())
            coin.reseed(self.roots[-1])
            alpha = coin.draw()
            layer = O.fri_fold(layer, _mont(offset), _mont(alpha), folding)
            offset = pow(offset, folding, V.P)
            lg -= log_f
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
        """rows of the extended trace at the positions that lie in the owned cosets, each followed by the bottom log2(nk) siblings of its
        authentication path (the part of the tree only this rank holds); zeros elsewhere: [nq][width + 4 log2(nk)] words"""
        log_nk = self.nk.bit_length() - 1
        out = np.zeros((len(positions), self.job.width + 4 * log_nk), np.uint64)
        for q, p in enumerate(positions):
            k, j = int(p) & (self.b - 1), int(p) >> self.log_b
            if self.k0 <= k < self.k0 + self.nk:
                out[q, :self.job.width] = self.lde[k - self.k0, :, j]
                li = self.nk * self.n + j * self.nk + (k - self.k0)   # the leaf in the rank's own heap
                for lvl in range(log_nk):
                    out[q, self.job.width + 4 * lvl:self.job.width + 4 * lvl + 4] = np.frombuffer(self.sub[(li >> lvl) ^ 1].tobytes(), np.uint64)
        return out

    def finish(self, rows):
        """rows [nq][width + 4 log2(nk)] complete (summed over the ranks) -> proof bytes (layout: include/cstark.h)"""
        positions, log_n, log_f, folding = self.positions, self.log_n, self.log_f, self.options[5]
        log_N = log_n + self.log_b
        log_nk, W = self.nk.bit_length() - 1, self.job.width

        def trace_path(q, pos):  # bottom levels from the owning rank, the rest from the upper tree (its leaves: the subtree roots)
            low = b"".join(np.ascontiguousarray(rows[q, W + 4 * lvl:W + 4 * lvl + 4]).tobytes() for lvl in range(log_nk))
            return low + _path(self.tnodes, log_N - log_nk, pos >> log_nk)

        def row(tab, pos):  # tab [b][width][n] coset-major
            return np.ascontiguousarray(tab[pos & (self.b - 1), :, pos >> self.log_b]).tobytes()

        roots = self.roots
        out = [b"CSTK", struct.pack("<IIIII", 1, self.job.air, self.job.width, log_n, self.job.item), struct.pack("<7I", *self.options),
               self.trace_root, self.cons_root, struct.pack("<I", len(roots))] + roots + [self.rem_commit, self.ood_trace.tobytes(),
                                                                                         self.ood_comp.tobytes(), struct.pack("<Q", self.nonce)]
        out += [np.ascontiguousarray(rows[q, :W]).tobytes() for q in range(len(positions))] + [trace_path(q, p) for q, p in enumerate(positions)]
        out += [row(self.clde, p) for p in positions] + [_path(self.cnodes, log_N, p) for p in positions]
        cur, lg = positions, log_N
        for l in range(len(self.layers)):
            rows_l = 1 << (lg - log_f)
            fpos = V.fold_positions(cur, rows_l)
            out.append(struct.pack("<I", len(fpos)))
            tab = self.layers[l].reshape(folding, rows_l)
            out += [np.ascontiguousarray(tab[:, p]).tobytes() for p in fpos]
            out += [_path(self.trees[l], lg - log_f, p) for p in fpos]
            cur = fpos
            lg -= log_f
        out += [struct.pack("<I", self.remainder.size), self.remainder.tobytes()]
        return b"".join(out)


def _prove_job(job, options):
    if options[4] in (1, 2):
        return prove_ext(job, options)
    p = ShardedProver(None, options, job=job)
    roots = p.commit()
    combined = p.evaluate(roots)
    positions = p.compose(combined)
    return p.finish(p.open_rows(positions))


def prove(w, options=(42, 8, 0, 0, 0, 4, 256)):
    """TransactionExample::prove (src/lib.rs:116-141)"""
    return _prove_job(TxJob(w), options)


def prove_air(air, witness, options=(42, 8, 0, 0, 0, 4, 256), log_n=6):
    """The standalone AIRs' prove() -- the CPU counterpart of cstark_air_prove / cstark_range_prove_bits, byte for byte:
    O.AIR_MERKLE   witness = TxWitness          (src/merkle/update/mod.rs:81-106)
    O.AIR_SCHNORR  witness = SchnorrWitness     (src/schnorr/mod.rs:143-172)
    O.AIR_RANGE    witness = field element in memory form (src/range/mod.rs:75-100), or -- the synthetic long accumulator -- the
                   n / 64 little-endian words of the value together with log_n.
    O.AIR_RESCUE_CHAIN  witness = (seed: 7 elements in memory form, chain_length)   (benches/rescue.rs:66-86)"""
    if air == O.AIR_STATE_TRANSITION:
        return prove(witness, options)
    if air == O.AIR_MERKLE:
        return _prove_job(MerkleJob(witness), options)
    if air == O.AIR_SCHNORR:
        return _prove_job(SchnorrJob(witness), options)
    if air == O.AIR_RESCUE_CHAIN:
        return _prove_job(RescueJob(*witness), options)
    if air == O.AIR_RANGE:
        if isinstance(witness, (int, np.integer)):
            return _prove_job(RangeJob(number=int(witness)), options)
        return _prove_job(RangeJob(words=np.ascontiguousarray(witness, np.uint64), log_n=log_n), options)
    raise ValueError("no prover for this AIR")


def prove_ext(job, options):
    """FieldExtension::Quadratic / Cubic: base-field trace, everything drawn from the coin in the degree-m extension (oracle/ext.c).
    Layout differences: out-of-domain values are m-tuples; composition rows hold ce m-tuples; FRI rows and the remainder are
    component-major (component 0 of the `folding` points, then component 1, ...)."""
    if not hasattr(job, "combine"):
        job = TxJob(job)  # a TxWitness
    nq, blowup, grinding, hash_fn, ext, folding, max_rem = options
    check_options(options, job)
    assert ext in (1, 2)
    m = ext + 1
    H = lambda data: O.digest(data, hash_fn)
    log_b, b, W, ce, log_f = blowup.bit_length() - 1, blowup, job.width, job.ce, folding.bit_length() - 1
    trace = job.build()
    n = trace.shape[1]
    log_n = n.bit_length() - 1
    log_N, N = log_n + log_b, n * b
    log_rem = max_rem.bit_length() - 1
    pub_m, pub_bytes = job.public(trace)
    pub = [V.from_mont(v) for v in pub_m]
    coeffs = O.interpolate_columns(trace.copy())
    lde = O.lde_columns(coeffs, log_b)
    tnodes = O.merkle_build(O.hash_rows(lde, log_b, hash_fn=hash_fn), hash_fn)
    trace_root = tnodes[1].tobytes()
    seed = bytes([W, log_n]) + struct.pack("<Q", V.P) + bytes([nq, log_b, grinding, hash_fn, ext, folding, log_rem])
    seed += b"".join(struct.pack("<Q", v) for v in pub) + pub_bytes
    coin = V.Coin(seed, hash_fn)
    coin.reseed(trace_root)
    ta, tb = np.zeros((m, job.nc), np.uint64), np.zeros((m, job.nc), np.uint64)
    ba, bb = np.zeros((m, job.na), np.uint64), np.zeros((m, job.na), np.uint64)
    for i in range(job.nc):
        a, bt = coin.draw_e(m), coin.draw_e(m)
        for k in range(m):
            ta[k, i], tb[k, i] = _mont(a[k]), _mont(bt[k])
    for i in range(job.na):
        a, bt = coin.draw_e(m), coin.draw_e(m)
        for k in range(m):
            ba[k, i], bb[k, i] = _mont(a[k]), _mont(bt[k])
    # coefficients multiply base-field values: the components of the merged evaluations are independent base-field merges (on the
    # constraint-evaluation domain: every (b / ce)-th LDE coset)
    ce_lde = np.ascontiguousarray(lde[::b // ce])
    cc = [O.composition_columns(np.ascontiguousarray(job.combine(trace, ce_lde, ta[k], tb[k], ba[k], bb[k]))) for k in range(m)]
    ccoef = np.ascontiguousarray(np.stack(cc, axis=1).reshape(m * ce, n))  # column m i + k = component k of H_i
    clde = O.lde_columns(ccoef, log_b)
    cnodes = O.merkle_build(O.hash_rows(clde, log_b, hash_fn=hash_fn), hash_fn)
    cons_root = cnodes[1].tobytes()
    coin.reseed(cons_root)

    z = coin.draw_e(m)
    zw = V.e_scale(z, V.root_of_unity(log_n))
    zb = V.e_pow(z, ce)
    ood_cur = O.evaluate_polys_at_ext(coeffs, V.e_mont(z))
    ood_next = O.evaluate_polys_at_ext(coeffs, V.e_mont(zw))
    raw = O.evaluate_polys_at_ext(ccoef, V.e_mont(zb))                     # each component polynomial at z^ce
    ood_comp = np.zeros((ce, m), np.uint64)
    for i in range(ce):
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
    d_delta = np.array([V.e_mont(coin.draw_e(m)) for _ in range(ce)], np.uint64)
    deg_a, deg_b = V.e_mont(coin.draw_e(m)), V.e_mont(coin.draw_e(m))
    deep = O.deep_composition_ext(lde, clde, V.e_mont(z), ood_trace, ood_comp, d_alpha, d_beta, d_delta, deg_a, deg_b, log_b)
    layer = np.ascontiguousarray(np.stack([np.ascontiguousarray(deep[k].T).reshape(-1) for k in range(m)]))  # [m][N] natural order

    layers, trees, roots = [], [], []
    offset, lg = V.GEN, log_N
    while lg > log_rem:
        rows = 1 << (lg - log_f)
        nodes = O.merkle_build(O.hash_rows(layer.reshape(1, folding * m, rows), 0, hash_fn=hash_fn), hash_fn)
        layers.append(layer); trees.append(nodes); roots.append(nodes[1].tobytes())
        coin.reseed(roots[-1])
        alpha = coin.draw_e(m)
        layer = O.fri_fold_ext(layer, _mont(offset), V.e_mont(alpha), folding)
        offset = pow(offset, folding, V.P)
        lg -= log_f
    remainder = layer
    rem_commit = H(V.elem_bytes(remainder))
    coin.reseed(rem_commit)
    nonce = 1
    while grinding and struct.unpack("<Q", H(coin.seed + struct.pack("<Q", nonce))[:8])[0] & ((1 << grinding) - 1):
        nonce += 1
    coin.reseed_int(nonce)
    positions = coin.draw_integers(nq, N)

    def row(tab, pos):
        return np.ascontiguousarray(tab[pos & (b - 1), :, pos >> log_b]).tobytes()

    out = [b"CSTK", struct.pack("<IIIII", 1, job.air, W, log_n, job.item), struct.pack("<7I", *options),
           trace_root, cons_root, struct.pack("<I", len(roots))] + roots + [rem_commit, ood_trace.tobytes(), ood_comp.tobytes(),
                                                                           struct.pack("<Q", nonce)]
    out += [row(lde, p) for p in positions] + [_path(tnodes, log_N, p) for p in positions]
    out += [row(clde, p) for p in positions] + [_path(cnodes, log_N, p) for p in positions]
    cur, lg = positions, log_N
    for l in range(len(layers)):
        rows = 1 << (lg - log_f)
        fpos = V.fold_positions(cur, rows)
        out.append(struct.pack("<I", len(fpos)))
        tab = layers[l].reshape(folding * m, rows)
        out += [np.ascontiguousarray(tab[:, p]).tobytes() for p in fpos]
        out += [_path(trees[l], lg - log_f, p) for p in fpos]
        cur = fpos
        lg -= log_f
    out += [struct.pack("<I", remainder.shape[1]), remainder.tobytes()]
    return b"".join(out)
