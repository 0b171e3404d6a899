export TMPDIR=/tmp; R=$PWD
python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, TransactionMetadata, TransactionProver
b = Backend()
meta = TransactionMetadata.build_random(1024, 15, seed=7)
for o in [(96, 8, 0, 0, 0, 4, 256), (128, 8, 0, 0, 0, 4, 256), (27, 8, 0, 0, 0, 4, 256), (96, 8, 0, 0, 0, 4, 128), (96, 8, 0, 0, 0, 4, 1024), (96, 8, 24, 0, 0, 4, 256),
          (96, 8, 16, 1, 0, 4, 256), (96, 8, 20, 0, 2, 4, 256)]:
    p = TransactionProver(ProofOptions(*o), b)
    p.load_witness(meta)
    p.prove(); p.prove()
    t0 = time.perf_counter()
    for _ in range(3): proof = p.prove()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(o, "%.2f ms  %d bytes  %s" % (ms, len(proof), {k: round(v, 2) for k, v in b.prove_stage_ms().items() if k in ("composition", "ood", "deep", "fri", "queries")}), flush=True)
PY
