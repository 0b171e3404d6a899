"""Pins the oracle's engine stages: BLAKE3 against the official test vectors, the NTT against a naive DFT,
the LDE against Horner evaluation, the Merkle tree against direct hashing, and the constraint-evaluation
driver through the algebraic fact that the combined evaluations are a polynomial of degree < 8n
(SURVEY.md 8(c) (11),(12)).  CPU only."""
import hashlib
import random

import numpy as np

P = 2**62 + 2**56 + 2**55 + 1
R = 2**64


def unmont(x): return int(x) * pow(R, -1, P) % P


# official BLAKE3 test vectors (test_vectors.json of the BLAKE3 repository): input byte i = i % 251
B3_VECTORS = {
    0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
    1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
    1023: "10108970eeda3eb932baac1428c7a2163b0e924c9a9e25b35bba72b28f70bd11",
    1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
    1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
    2048: "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
}


def test_blake3_official_vectors(oracle):
    for n, hexd in B3_VECTORS.items():
        data = bytes(i % 251 for i in range(n))
        assert oracle.blake3(data).hex() == hexd, n


def test_ntt_vs_naive_and_roundtrip(oracle):
    rng = random.Random(5)
    for log_n in (1, 2, 5, 8):
        a = oracle.to_mont([rng.randrange(P) for _ in range(1 << log_n)])
        f = oracle.ntt(a)
        assert (f == oracle.dft_naive(a)).all()
        assert (oracle.ntt(f, inverse=True) == a).all()
    a = oracle.to_mont([rng.randrange(P) for _ in range(1 << 14)])
    assert (oracle.ntt(oracle.ntt(a), inverse=True) == a).all()


def test_lde_matches_horner(oracle):
    rng = random.Random(6)
    log_n, log_b, width = 6, 3, 3
    n = 1 << log_n
    ev = oracle.to_mont(np.array([[rng.randrange(P) for _ in range(n)] for _ in range(width)], np.uint64).ravel()).reshape(width, n)
    co = oracle.interpolate_columns(ev)
    lde = oracle.lde_columns(co, log_b)
    assert lde.shape == (8, width, n)
    g = 3
    w8n = unmont(oracle.root_of_unity(log_n + log_b))
    wn = unmont(oracle.root_of_unity(log_n))
    coc = [[unmont(v) for v in co[c]] for c in range(width)]
    for (k, c, j) in [(0, 0, 0), (1, 1, 5), (7, 2, 63), (3, 0, 17)]:
        x = g * pow(w8n, k, P) * pow(wn, j, P) % P
        assert x == g * pow(w8n, 8 * j + k, P) % P           # natural LDE index i = 8 j + k
        y = sum(coc[c][m] * pow(x, m, P) for m in range(n)) % P
        assert unmont(lde[k, c, j]) == y
    # the trace domain itself is recovered from the polynomial (offset 1, coset 0)
    back = oracle.lde_columns(co, 0, offset=int(oracle.to_mont([1])[0]))
    assert (back[0] == ev).all()
    # coset subset request equals the slice of the full result
    part = oracle.lde_columns(co, log_b, k0=2, nk=3)
    assert (part == lde[2:5]).all()


def test_row_hash_and_merkle(oracle):
    rng = random.Random(7)
    log_n, log_b, width = 3, 2, 5
    n = 1 << log_n
    lde = oracle.to_mont(np.array([rng.randrange(P) for _ in range(4 * width * n)], np.uint64)).reshape(4, width, n)
    leaves = oracle.hash_rows(lde, log_b)
    for (k, j) in [(0, 0), (3, 7), (2, 4)]:
        row = b"".join(int(lde[k, c, j]).to_bytes(8, "little") for c in range(width))
        assert leaves[4 * j + k].tobytes() == oracle.blake3(row)
    nodes = oracle.merkle_build(leaves)
    L = leaves.shape[0]
    assert (nodes[L:] == leaves).all() and not nodes[0].any()
    for i in (1, 2, 5, L - 1):
        assert nodes[i].tobytes() == oracle.blake3(nodes[2 * i].tobytes() + nodes[2 * i + 1].tobytes())


def test_periodic_table_matches_direct_evaluation(oracle):
    depth, log_n, log_b = 3, 11, 3
    n = 1 << log_n
    tab = oracle.tx_periodic_table(depth, log_n, log_b)
    cols = oracle.tx_periodic_columns(depth)
    co = oracle.interpolate_columns(cols)
    g, w8n, wn = 3, unmont(oracle.root_of_unity(log_n + log_b)), unmont(oracle.root_of_unity(log_n))
    for (k, c, j) in [(0, 0, 0), (5, 19, 700), (7, 33, 1500), (2, 2, 1023)]:
        x = g * pow(w8n, k, P) * pow(wn, j, P) % P
        y = pow(x, n // 1024, P)
        val = sum(unmont(co[c, m]) * pow(y, m, P) for m in range(1024)) % P
        assert unmont(tab[k, c, j % 1024]) == val


def test_combined_evaluations_interpolate_to_the_composition_polynomial(oracle, witness_d3):
    """Out-of-domain consistency (the check the verifier performs): for a valid trace the combined evaluations
    over the 8n-point domain g<w_8n> are those of a polynomial H of degree < 8n, so H(z) -- interpolated from
    the evaluations -- must equal the constraint expression evaluated directly at a random z from the trace
    polynomials.  For an invalid trace or wrong public inputs (the reference's negative test,
    src/lib.rs:153-161) the quotient is not a polynomial and the two values differ."""
    w = witness_d3
    log_n, log_b = 11, 3
    n = 1 << log_n
    cf = oracle.make_coeffs(3)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    z = int(oracle.to_mont([0x1234567890ABCDEF % P])[0])
    ginv = int(oracle.fp_inv(np.array([oracle.generator()], np.uint64))[0])
    zg = int(oracle.fp_mul(np.array([z], np.uint64), np.array([ginv], np.uint64))[0])

    def both(wit, pub_inputs):
        trace = oracle.tx_build_trace(wit)
        co = oracle.interpolate_columns(trace)
        lde = oracle.lde_columns(co, log_b)
        comb = oracle.tx_evaluate_constraints(lde, cf, pub_inputs, wit.depth, log_b)   # [8][n], coset-major
        nat = np.ascontiguousarray(comb.T).ravel()                                     # natural order i = 8 j + k
        h = oracle.ntt(nat, inverse=True)                                              # H(g y) as a polynomial in y
        return oracle.poly_eval(h, zg), oracle.tx_combined_at(co, cf, pub_inputs, wit.depth, log_b, z)
    a, b = both(w, pub)
    assert a == b
    bad_pub = pub.copy(); bad_pub[3] = pub[2]
    a, b = both(w, bad_pub)
    assert a != b
    bad = w.copy(); bad.deltas[1] = oracle.fp_add(bad.deltas[1:2], oracle.to_mont([1]))[0]
    a, b = both(bad, pub)
    assert a != b


def test_degree_adjustments(oracle):
    log_n, log_b = 11, 3
    n = 1 << log_n
    adj = oracle.tx_degree_adjustments(log_n, log_b)
    assert int(adj[0]) == (8 * n - 1 + n - 1) - (5 * (n - 1) + 2 * (n // 1024) * 1023)
    assert int(adj[60]) == (8 * n - 1 + n - 1) - (1 * (n - 1) + (n // 1024) * 1023)


def test_composition_columns_recombine(oracle, witness_d3):
    """H(x) = sum_i x^i H_i(x^8): the column polynomials evaluated at z^8 recombine to the interpolant of the combined
    evaluations at z, which (previous test) equals the constraint expression at z."""
    w = witness_d3
    log_b = 3
    trace = oracle.tx_build_trace(w)
    co = oracle.interpolate_columns(trace)
    lde = oracle.lde_columns(co, log_b)
    cf = oracle.make_coeffs(3)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    comb = oracle.tx_evaluate_constraints(lde, cf, pub, w.depth, log_b)
    cols = oracle.composition_columns(comb)
    z = 0x1234567890ABCDEF % P
    zm = int(oracle.to_mont([z])[0]); z8 = int(oracle.to_mont([pow(z, 8, P)])[0])
    acc = 0
    for i in range(8):
        acc = (acc + pow(z, i, P) * unmont(oracle.poly_eval(cols[i], z8))) % P
    assert acc == unmont(oracle.tx_combined_at(co, cf, pub, w.depth, log_b, zm))


def test_deep_composition_is_low_degree(oracle, witness_d3):
    """The DEEP composition polynomial has degree < n when the out-of-domain frame is consistent, i.e. its 8n evaluations
    interpolate to a polynomial whose coefficients n.. vanish; with a wrong OOD value they do not (this is what FRI tests)."""
    w = witness_d3
    log_b = 3
    trace = oracle.tx_build_trace(w)
    n = trace.shape[1]; log_n = n.bit_length() - 1
    co = oracle.interpolate_columns(trace)
    lde = oracle.lde_columns(co, log_b)
    cf = oracle.make_coeffs(3)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    cols = oracle.composition_columns(oracle.tx_evaluate_constraints(lde, cf, pub, w.depth, log_b))
    comp_lde = oracle.lde_columns(cols, log_b)
    z = int(oracle.to_mont([0x1234567890ABCDEF % P])[0])
    zw = int(oracle.fp_mul(np.array([z], np.uint64), np.array([oracle.root_of_unity(log_n)], np.uint64))[0])
    z8 = int(oracle.fp_pow(np.array([z], np.uint64), 8)[0])
    ood_t = oracle.evaluate_polys_at(co, [z, zw])
    ood_c = oracle.evaluate_polys_at(cols, [z8])[0]
    al, be, de = oracle.random_elements(94, 1), oracle.random_elements(94, 2), oracle.random_elements(8, 3)
    da, db = (int(v) for v in oracle.random_elements(2, 4))

    def high_coeffs(ood_trace):
        deep = oracle.deep_composition(lde, comp_lde, z, ood_trace, ood_c, al, be, de, da, db, log_b)
        nat = np.ascontiguousarray(deep.T).ravel()
        return oracle.ntt(nat, inverse=True)[n:]
    assert not high_coeffs(ood_t).any()
    bad = ood_t.copy(); bad[1, 17] = oracle.fp_add(bad[1, 17:18], oracle.to_mont([1]))[0]
    assert high_coeffs(bad).any()


def test_fri_fold4(oracle):
    """Folding by 4 maps evaluations of f = sum_k x^k f_k(x^4) over g<w_N> to evaluations of sum_k alpha^k f_k(y) over
    g^4<w_{N/4}>; in particular degree < d becomes degree < d/4."""
    rng = np.random.default_rng(8)
    log_n, d = 10, 256
    N = 1 << log_n
    coeffs = np.zeros(N, np.uint64); coeffs[:d] = oracle.to_mont(rng.integers(0, P, size=d, dtype=np.uint64))
    g = oracle.generator()
    ev = oracle.lde_columns(coeffs.reshape(1, N), 0, offset=g)[0, 0]            # f over g<w_N>, natural order
    alpha = int(oracle.to_mont([987654321987654321 % P])[0])
    folded = oracle.fri_fold4(ev, g, alpha)
    # direct: g'(y) = sum_k alpha^k f_k(y), f_k = coefficients k, k+4, ...
    fk = coeffs.reshape(N // 4, 4).T.copy()                                     # [4][N/4]
    ak = [int(oracle.fp_pow(np.array([alpha], np.uint64), k)[0]) for k in range(4)]
    gco = np.zeros(N // 4, np.uint64)
    for k in range(4):
        gco = oracle.fp_add(gco, oracle.fp_mul(fk[k], np.full(N // 4, ak[k], np.uint64)))
    g4 = int(oracle.fp_pow(np.array([g], np.uint64), 4)[0])
    direct = oracle.lde_columns(gco.reshape(1, N // 4), 0, offset=g4)[0, 0]
    assert (folded == direct).all()
    assert not gco[d // 4:].any()


def test_sha3_256_known_answers(oracle):
    """FIPS 202 vectors and Python's own implementation: pins the second hash of ProofOptions (HashFunction::Sha3_256)."""
    import hashlib
    assert oracle.sha3_256(b"").hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    assert oracle.sha3_256(b"abc").hex() == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532"
    for n in (1, 71, 72, 135, 136, 137, 272, 752, 1000):
        m = bytes((7 * i + n) & 0xff for i in range(n))
        assert oracle.sha3_256(m) == hashlib.sha3_256(m).digest()
    rng = np.random.default_rng(1)
    lde = rng.integers(0, 2**62, size=(2, 17, 8), dtype=np.uint64)       # width 17: the padding gets a block of its own
    leaves = oracle.hash_rows(lde, 1, hash_fn=1)
    assert leaves[2 * 3 + 1].tobytes() == hashlib.sha3_256(np.ascontiguousarray(lde[1, :, 3]).tobytes()).digest()
    nodes = oracle.merkle_build(leaves, 1)
    assert nodes[1].tobytes() == hashlib.sha3_256(nodes[2].tobytes() + nodes[3].tobytes()).digest()
