"""GPU parity for the composition-polynomial stage (first of the "next" rows): combined evaluations -> H(x) -> column
split -> LDE -> Blake3 commitment, bit-exact against the oracle; and at full size the recombination identity."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**62 + 2**56 + 2**55 + 1


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


@pytest.mark.parametrize("log_n", [10, 12, 13])
def test_composition_columns_and_commitment(oracle, backend, log_n):
    import torch
    from certificate_stark_amd.backend import to_numpy_u64
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    comb = oracle.to_mont(rng.integers(0, P, size=8 * n, dtype=np.uint64)).reshape(8, n)
    ref_cols = oracle.composition_columns(comb)
    d_cols = backend.composition_columns(backend.from_numpy_u64(comb))
    assert (to_numpy_u64(d_cols) == ref_cols).all()
    # commitment of the composition columns: LDE (width 8) -> row hashes -> tree
    lde = backend.lde_columns(d_cols, 3)
    ref_lde = oracle.lde_columns(ref_cols, 3)
    assert (to_numpy_u64(lde) == ref_lde).all()
    L = n << 3
    nodes = torch.zeros((2 * L, 32), dtype=torch.uint8, device=backend.device)
    backend.hash_rows(lde, 3, leaves=nodes[L:])
    backend.merkle_build(nodes)
    assert (nodes.cpu().numpy() == oracle.merkle_build(oracle.hash_rows(ref_lde, 3))).all()
    # consistency: the column polynomials reproduce the combined evaluations on the LDE domain,
    # H(x_i) = sum_c x_i^c H_c(x_i^8) -- checked on a few points with the oracle
    g, w8n, wn = int(oracle.from_mont([oracle.generator()])[0]), oracle.from_mont([oracle.root_of_unity(log_n + 3)])[0], oracle.from_mont([oracle.root_of_unity(log_n)])[0]
    for (k, j) in [(0, 0), (5, 17), (7, n - 1)]:
        x = g * pow(int(w8n), k, P) * pow(int(wn), j, P) % P
        x8 = int(oracle.to_mont([pow(x, 8, P)])[0])
        acc = sum(pow(x, c, P) * int(oracle.from_mont([oracle.poly_eval(ref_cols[c], x8)])[0]) for c in range(8)) % P
        assert acc == int(oracle.from_mont(comb[k, j:j + 1])[0])


def test_ood_frame_and_deep_composition(oracle, backend):
    """OOD evaluation and DEEP composition against the oracle on a real trace (2 transfers), plus the low-degree property."""
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(2, 3, seed=606)
    trace = oracle.tx_build_trace(w)
    n = trace.shape[1]; log_n, log_b = n.bit_length() - 1, 3
    co = oracle.interpolate_columns(trace)
    lde = oracle.lde_columns(co, log_b)
    cf = oracle.make_coeffs(3)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    cols = oracle.composition_columns(oracle.tx_evaluate_constraints(lde, cf, pub, w.depth, log_b))
    comp_lde = oracle.lde_columns(cols, log_b)
    z = int(oracle.to_mont([0x0FEDCBA987654321 % P])[0])
    zw = int(oracle.fp_mul(np.array([z], np.uint64), np.array([oracle.root_of_unity(log_n)], np.uint64))[0])
    z8 = int(oracle.fp_pow(np.array([z], np.uint64), 8)[0])
    d_co, d_cols = backend.from_numpy_u64(co), backend.from_numpy_u64(cols)
    ood_t = backend.evaluate_polys_at(d_co, [z, zw])
    ood_c = backend.evaluate_polys_at(d_cols, [z8])[0]
    assert (ood_t == oracle.evaluate_polys_at(co, [z, zw])).all()
    assert (ood_c == oracle.evaluate_polys_at(cols, [z8])[0]).all()
    al, be, de = oracle.random_elements(94, 1), oracle.random_elements(94, 2), oracle.random_elements(8, 3)
    da, db = (int(v) for v in oracle.random_elements(2, 4))
    ref = oracle.deep_composition(lde, comp_lde, z, ood_t, ood_c, al, be, de, da, db, log_b)
    d_lde, d_clde = backend.from_numpy_u64(lde), backend.from_numpy_u64(comp_lde)
    got = to_numpy_u64(backend.deep_composition(d_lde, d_clde, z, ood_t, ood_c, al, be, de, da, db, log_b))
    assert (got == ref).all()
    part = to_numpy_u64(backend.deep_composition(d_lde[4:6].contiguous(), d_clde[4:6].contiguous(), z, ood_t, ood_c, al, be, de, da, db, log_b, k0=4))
    assert (part == ref[4:6]).all()
    nat = np.ascontiguousarray(got.T).ravel()
    assert not oracle.ntt(nat, inverse=True)[n:].any()        # degree < n: what FRI will test


def test_fri_layers(oracle, backend):
    """FRI commit phase on the GPU: natural-order view, layer commitments and folding by 4 down to the remainder, against
    the oracle; a low-degree input stays low-degree through every layer."""
    from certificate_stark_amd.backend import to_numpy_u64
    rng = np.random.default_rng(9)
    log_n, log_b = 10, 3
    n = 1 << log_n
    co = np.zeros((1, n), np.uint64); co[0] = oracle.to_mont(rng.integers(0, P, size=n, dtype=np.uint64))
    deep_cm = oracle.lde_columns(co, log_b)[:, 0, :]                       # [8][n] coset-major evaluations of a degree < n polynomial
    d_nat = backend.interleave_cosets(backend.from_numpy_u64(deep_cm))
    nat = np.ascontiguousarray(deep_cm.T).ravel()
    assert (to_numpy_u64(d_nat) == nat).all()
    offset = oracle.generator()
    layer, d_layer, N = nat, d_nat, nat.size
    alphas = oracle.random_elements(8, 77)
    li = 0
    while N > 256:                                                         # fri_max_remainder = 256 (src/lib.rs:85)
        q = N // 4
        ref_nodes = oracle.merkle_build(oracle.hash_rows(layer.reshape(1, 4, q), 0))
        nodes = backend.fri_commit_layer(d_layer)
        assert (nodes.cpu().numpy() == ref_nodes).all()
        alpha = int(alphas[li]); li += 1
        nxt = oracle.fri_fold4(layer, offset, alpha)
        d_nxt = backend.fri_fold4(d_layer, offset, alpha)
        assert (to_numpy_u64(d_nxt) == nxt).all()
        layer, d_layer, N = nxt, d_nxt, q
        offset = int(oracle.fp_pow(np.array([offset], np.uint64), 4)[0])
        # degree bound shrinks with the domain: blowup stays 8
        assert not oracle.ntt(layer, inverse=True)[N // 8:].any()
    assert N == 128 and li == 3
