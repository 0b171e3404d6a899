"""Experiment: proofs/s with several independent proofs in flight on one GPU (one context + stream + host thread each)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, TransactionMetadata, TransactionProver

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "witness_1024_d15.npz")
meta = TransactionMetadata.load(FIXTURE)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for inflight in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]:
    provers = []
    for i in range(inflight):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            p = TransactionProver(ProofOptions(num_queries=96), Backend(0))
            p.load_witness(meta)
            p.prove()
        provers.append((s, p))
    torch.cuda.synchronize()
    out = [None] * inflight

    def work(i):
        s, p = provers[i]
        with torch.cuda.stream(s):
            for _ in range(per):
                out[i] = p.prove()

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(inflight)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(o == out[0] for o in out)
    print("in flight %d: %.2f ms per proof, %.2f proofs/s" % (inflight, dt / (inflight * per) * 1e3, inflight * per / dt), flush=True)
    for s, p in provers:
        p.backend.close()
