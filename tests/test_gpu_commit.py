"""GPU parity: K4 Blake3 row hashing and K5 Merkle tree vs the CPU oracle (which is pinned by the official
BLAKE3 vectors), bit-exact, including ragged row widths."""
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


@pytest.mark.parametrize("width,log_n,log_b", [(94, 8, 3), (1, 6, 0), (8, 7, 2), (9, 7, 1), (65, 9, 3), (56, 6, 3), (128, 6, 1), (2, 6, 3), (14, 8, 2)])
def test_row_hash_and_tree(oracle, backend, width, log_n, log_b):
    import torch
    rng = np.random.default_rng(width * 100 + log_n)
    n, b = 1 << log_n, 1 << log_b
    lde = oracle.to_mont(rng.integers(0, P, size=b * width * n, dtype=np.uint64)).reshape(b, width, n)
    ref_leaves = oracle.hash_rows(lde, log_b)
    d_lde = backend.from_numpy_u64(lde)
    L = n * b
    nodes = torch.zeros((2 * L, 32), dtype=torch.uint8, device=backend.device)
    backend.hash_rows(d_lde, log_b, leaves=nodes[L:])
    got = nodes[L:].cpu().numpy()
    assert (got == ref_leaves).all()
    if L >= 2:
        backend.merkle_build(nodes)
        ref_nodes = oracle.merkle_build(ref_leaves)
        assert (nodes.cpu().numpy() == ref_nodes).all()


def test_coset_subset_writes_only_its_leaves(oracle, backend):
    import torch
    rng = np.random.default_rng(5)
    width, log_n, log_b = 7, 6, 3
    n = 1 << log_n
    lde = oracle.to_mont(rng.integers(0, P, size=8 * width * n, dtype=np.uint64)).reshape(8, width, n)
    ref = oracle.hash_rows(lde, log_b)
    leaves = torch.zeros((8 * n, 32), dtype=torch.uint8, device=backend.device)
    backend.hash_rows(backend.from_numpy_u64(lde[2:5]), log_b, k0=2, leaves=leaves)
    got = leaves.cpu().numpy().reshape(n, 8, 32)
    assert (got[:, 2:5] == ref.reshape(n, 8, 32)[:, 2:5]).all()
    assert not got[:, :2].any() and not got[:, 5:].any()


@pytest.mark.parametrize("log_leaves", list(range(1, 20)))
def test_merkle_tree_every_size(oracle, backend, log_leaves):
    """Every tree size from 2 to 2^19 leaves, every node: the paths of merkle_build change with the size (one four-lane workgroup up to
    512 leaves, 256-parent workgroups + one more launch up to 2^18, lane-per-node double levels above)."""
    import torch
    L = 1 << log_leaves
    leaves = np.random.default_rng(100 + log_leaves).integers(0, 256, size=(L, 32), dtype=np.uint8)
    nodes = torch.zeros((2 * L, 32), dtype=torch.uint8, device=backend.device)
    nodes[L:] = torch.from_numpy(leaves).to(backend.device)
    backend.merkle_build(nodes)
    assert (nodes.cpu().numpy()[1:] == oracle.merkle_build(leaves)[1:]).all()


def test_large_tree_2_20_leaves_root(oracle, backend):
    """2^20 leaves on the GPU vs the oracle: exercises the multi-launch level path plus the one-workgroup top."""
    import torch
    rng = np.random.default_rng(6)
    L = 1 << 20
    leaves = rng.integers(0, 256, size=(L, 32), dtype=np.uint8)
    nodes = torch.zeros((2 * L, 32), dtype=torch.uint8, device=backend.device)
    nodes[L:] = torch.from_numpy(leaves).to(backend.device)
    backend.merkle_build(nodes)
    ref = oracle.merkle_build(leaves)
    assert (nodes[:4096].cpu().numpy() == ref[:4096]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("width", [1, 4, 8, 16, 17, 18, 34, 94, 128])
def test_sha3_row_hashes_and_tree(oracle, width):
    """Sha3_256 variants of K4 / K5 against the oracle (itself pinned by FIPS 202 vectors and hashlib): ragged widths around the
    17-word rate boundary, a coset subset, and the tree."""
    import torch
    from certificate_stark_amd.backend import Backend
    b = Backend()
    rng = np.random.default_rng(width)
    n, log_b = 256, 3
    lde = rng.integers(0, 2**62, size=(8, width, n), dtype=np.uint64)
    d_lde = b.from_numpy_u64(lde)
    leaves = b.hash_rows_fn(1, d_lde, log_b)
    ref = oracle.hash_rows(lde, log_b, hash_fn=1)
    assert (leaves.cpu().numpy() == ref).all()
    part = b.hash_rows_fn(1, d_lde[2:5].contiguous(), log_b, k0=2).cpu().numpy().reshape(n, 8, 32)
    assert (part[:, 2:5] == ref.reshape(n, 8, 32)[:, 2:5]).all()
    nodes = torch.zeros((2 * n * 8, 32), dtype=torch.uint8, device=b.device)
    nodes[n * 8:] = leaves
    b.merkle_build_fn(1, nodes)
    assert (nodes.cpu().numpy() == oracle.merkle_build(ref, 1)).all()
    b.close()
