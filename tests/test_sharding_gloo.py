"""world_size-2 gloo test (CPU) of the coset-sharded proof path: each rank extends / hashes / evaluates only its
cosets (the CPU oracle stands in for the kernels), the two all-gathers reassemble the commitment and the combined
evaluations, and both must equal the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from certificate_stark_amd import sharding
    log_b, depth = 3, 3
    w = O.TxWitness.generate(1, depth, seed=77)
    trace = O.tx_build_trace(w)                      # replicated (K1)
    co = O.interpolate_columns(trace)                # replicated (K2)
    k0, nk = sharding.coset_range(rank, world, 1 << log_b)
    lde = O.lde_columns(co, log_b, k0=k0, nk=nk)     # K3 on own cosets
    n = trace.shape[1]
    # K4 on own cosets, compact [nk][n][32]: hash each coset as a blowup-1 domain
    leaves_local = np.stack([O.hash_rows(lde[i:i + 1], 0) for i in range(nk)])
    cf = O.make_coeffs(5)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    comb_local = O.tx_evaluate_constraints(lde, cf, pub, depth, log_b, k0=k0)     # K6 on own cosets
    all_leaves = sharding.all_gather_cosets(torch.from_numpy(leaves_local))
    all_comb = sharding.all_gather_cosets(torch.from_numpy(comb_local.view(np.int64)))
    nodes = O.merkle_build(sharding.leaves_to_natural_order(all_leaves).numpy())
    # single-process reference
    lde_full = O.lde_columns(co, log_b)
    ref_nodes = O.merkle_build(O.hash_rows(lde_full, log_b))
    ref_comb = O.tx_evaluate_constraints(lde_full, cf, pub, depth, log_b)
    ok = (nodes == ref_nodes).all() and (all_comb.numpy().view(np.uint64) == ref_comb).all()
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
        f.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def test_coset_sharded_path_matches_single_process(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_coset_range():
    from certificate_stark_amd import sharding
    assert [sharding.coset_range(r, 4, 8) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert sharding.coset_range(0, 1, 8) == (0, 8)
    try:
        sharding.coset_range(0, 3, 8)
        assert False
    except ValueError:
        pass
