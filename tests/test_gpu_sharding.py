"""The coset-sharded proof (cstark_tx_shard_*) on ONE GPU: W contexts stand for the W ranks and the test moves the
exchanged buffers between them (concatenation / sum in place of the RCCL all-gather / reduce of sharding.prove_sharded, whose
collective logic is covered by the gloo tests).  The proof bytes must equal the single-GPU proof and the CPU restatement's."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharded_proof(meta, options, world):
    import torch
    from certificate_stark_amd.backend import Backend
    nk = 8 // world
    ranks = [Backend() for _ in range(world)]
    try:
        for b in ranks:
            b.upload_witness(meta)
        leaves = torch.cat([b.shard_commit(options, r * nk, nk) for r, b in enumerate(ranks)])       # all-gather
        combined = torch.cat([b.shard_evaluate(leaves) for b in ranks])                               # all-gather
        positions = ranks[0].shard_compose(combined)                                                  # broadcast
        rows = sum(b.shard_open_rows(positions) for b in ranks)                                       # reduce (sum)
        return ranks[0].shard_finish(rows)
    finally:
        for b in ranks:
            b.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("world,n_tx,depth,opts", [(2, 4, 3, (42, 8, 0, 0, 0, 4, 256)), (4, 8, 15, (28, 8, 0, 0, 0, 4, 128)),
                                                   (8, 16, 15, (96, 8, 4, 1, 0, 4, 256))])
def test_sharded_proof_equals_single_gpu_proof(oracle, world, n_tx, depth, opts):
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata
    w = oracle.TxWitness.generate(n_tx, depth, seed=900 + world)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    proof = _sharded_proof(meta, ProofOptions(*opts), world)
    b = Backend()
    b.upload_witness(meta)
    assert proof == b.prove(ProofOptions(*opts))
    b.close()
    assert proof == OP.prove(w, opts)
    assert V.verify(proof, w.initial_roots[0], w.final_root, options=list(opts))


def test_sharded_proof_at_full_size():
    """BASELINE's headline configuration split over two ranks' worth of cosets: the digests of the CPU prover's proof."""
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata
    from tools.make_proof_digest import section_digests
    meta = TransactionMetadata.load(os.path.join(ROOT, "tests", "golden", "witness_1024_d15.npz"))
    proof = _sharded_proof(meta, ProofOptions(96, 8, 0, 0, 0, 4, 256), 2)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_1024tx_d15_q96.json")))
    assert section_digests(proof, 96) == gold["sections"]
    assert hashlib.sha256(proof).hexdigest() == gold["sha256"]


def test_phases_out_of_order_are_refused():
    import torch
    from certificate_stark_amd import CstarkError
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata
    b = Backend()
    b.upload_witness(TransactionMetadata.build_random(2, 3, seed=5))
    b._shard_options, b._shard_nk = ProofOptions(), 4
    with pytest.raises(CstarkError):
        b.shard_evaluate(torch.zeros((8, 2048, 32), dtype=torch.uint8, device=b.device))
    with pytest.raises(CstarkError):
        b.shard_commit(ProofOptions(), 0, 8)      # a single rank uses cstark_tx_prove
    with pytest.raises(CstarkError):
        b.shard_commit(ProofOptions(), 2, 4)      # misaligned coset window
    b.close()
