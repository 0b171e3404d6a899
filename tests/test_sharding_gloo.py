"""Multi-process tests (CPU, gloo) of the coset-sharded proof: certificate_stark_amd.sharding.prove_sharded -- the driver the GPU
ranks run over RCCL -- with a stand-in backend whose phases are the CPU oracle's (oracle/prover.py::ShardedProver mirrors the
cstark_tx_shard_* phases).  World sizes 2 and 4: the all-gathers of digests and merged evaluations, the leaf-order reassembly, the
broadcast of the query positions and the reduction of the opened rows must give rank 0 the single-process proof, byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTS = (28, 8, 0, 0, 0, 4, 128)


class OracleShardBackend:
    """The Backend.shard_* surface on CPU tensors, computed by the oracle (test infrastructure)."""

    def __init__(self, w):
        self.w = w

    def shard_commit(self, options, k0, nk):
        from oracle import prover as OP
        self.p = OP.ShardedProver(self.w, (options.num_queries, options.blowup_factor, options.grinding_factor, options.hash_fn,
                                           options.field_extension, options.fri_folding_factor, options.fri_max_remainder), k0, nk)
        return torch.from_numpy(self.p.commit())

    def shard_evaluate(self, leaves_all):
        return torch.from_numpy(self.p.evaluate(leaves_all.numpy()).view(np.int64))

    def shard_compose(self, combined_all):
        return torch.from_numpy(self.p.compose(combined_all.numpy().view(np.uint64)).view(np.int32))

    def shard_open_rows(self, positions):
        return torch.from_numpy(self.p.open_rows(positions.numpy().view(np.uint32)).view(np.int64))

    def shard_finish(self, rows):
        return self.p.finish(rows.numpy().view(np.uint64))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from certificate_stark_amd import sharding
    from certificate_stark_amd.prover import ProofOptions
    w = O.TxWitness.generate(2, 3, seed=77)
    proof = sharding.prove_sharded(OracleShardBackend(w), ProofOptions(*OPTS))
    assert (proof is None) == (rank != 0)
    if rank == 0:
        with open(os.path.join(out_dir, "proof"), "wb") as f:
            f.write(proof)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_proof_equals_single_process_proof(tmp_path, world):
    from oracle import oracle as O
    from oracle import prover as OP
    from oracle import verifier as V
    port = 29500 + (os.getpid() * 7 + world) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    proof = open(os.path.join(str(tmp_path), "proof"), "rb").read()
    w = O.TxWitness.generate(2, 3, seed=77)
    assert proof == OP.prove(w, OPTS)
    assert V.verify(proof, w.initial_roots[0], w.final_root, options=list(OPTS))


def test_coset_range_and_leaf_order():
    from certificate_stark_amd import sharding
    assert [sharding.coset_range(r, 4, 8) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert sharding.coset_range(0, 1, 8) == (0, 8)
    with pytest.raises(ValueError):
        sharding.coset_range(0, 3, 8)
    cm = torch.arange(8 * 4 * 32, dtype=torch.int64).reshape(8, 4, 32).to(torch.uint8)
    nat = sharding.leaves_to_natural_order(cm)
    for k in range(8):
        for j in range(4):
            assert torch.equal(nat[8 * j + k], cm[k, j])
