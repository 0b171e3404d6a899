"""Per-rank phase times of the coset-sharded proof, measured on ONE GPU: the W ranks of a world are run one after the other on the same
device (each alone on the GPU, as it would be on its own), the exchanged buffers moved by concatenation; per phase the slowest rank
counts.  What this does NOT measure: the collectives themselves (sizes are printed; DESIGN.md 6 prices them) and the ranks' overlap.
    python tools/bench_shard_sim.py [n_tx]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, TransactionMetadata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = TransactionMetadata.load(os.path.join(ROOT, "tests", "golden", "witness_1024_d15.npz"))
n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if n_tx != meta.n_tx:
    full = meta
    meta = TransactionMetadata(*[getattr(full, f) if f == "final_root" else getattr(full, f)[:n_tx] for f in TransactionMetadata.FIELDS])
    meta.final_root = full.initial_roots[n_tx].copy()
opts = ProofOptions(96, 8, 0, 0, 0, 4, 256)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


single = Backend()
single.upload_witness(meta)
single.prove(opts)
ref, t_single = timed(lambda: single.prove(opts))
print("one GPU, cstark_tx_prove: %.2f ms" % t_single)
single.close()
for world in (2, 4, 8):
    for mode in (("split", "1"), ("direct", "0")) if world < 8 else (("direct", "1"),):
        # the switch is read once per process: the direct form of W = 2, 4 is measured by tools/gpu_jobs with CSTARK_SHARD_SPLIT=0
        if mode[1] != os.environ.get("CSTARK_SHARD_SPLIT", "1"):
            continue
        nk = 8 // world
        ranks = [Backend() for _ in range(world)]
        for b in ranks:
            b.upload_witness(meta)
        best = None
        for rep in range(3):
            t = {"commit": 0.0, "evaluate": 0.0, "compose": 0.0, "open": 0.0, "finish": 0.0}
            outs = [timed(lambda b=b, r=r: b.shard_commit(opts, r * nk, nk)) for r, b in enumerate(ranks)]
            t["commit"] = max(o[1] for o in outs)
            leaves = torch.cat([o[0] for o in outs])
            outs = [timed(lambda b=b: b.shard_evaluate(leaves)) for b in ranks]
            t["evaluate"] = max(o[1] for o in outs)
            combined = torch.cat([o[0] for o in outs])
            positions, t["compose"] = timed(lambda: ranks[0].shard_compose(combined))
            outs = [timed(lambda b=b: b.shard_open_rows(positions)) for b in ranks]
            t["open"] = max(o[1] for o in outs)
            rows = sum(o[0] for o in outs)
            proof, t["finish"] = timed(lambda: ranks[0].shard_finish(rows))
            assert proof == ref
            if best is None or sum(t.values()) < sum(best.values()):
                best = t
        ex = (leaves.numel() // world, combined.numel() * 8 // world)
        print("W = %d (%s evaluation at the ranks): per-phase max over ranks %s  sum %.2f ms  | all-gathers per rank: digests %.1f MB, evaluations %.1f MB" % (
            world, mode[0], {k: round(v, 2) for k, v in best.items()}, sum(best.values()), ex[0] / 1e6, ex[1] / 1e6))
        for b in ranks:
            b.close()
        del ranks, leaves, combined
        torch.cuda.empty_cache()
