"""Times the constraint-evaluation stage alone on random field data (2^log_n rows x 8 cosets), per launch set."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd import _lib
from certificate_stark_amd.backend import Backend

P = 2**62 + 2**56 + 2**55 + 1
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
split = len(sys.argv) > 3 and sys.argv[3] == "split"  # the degree-split evaluator (timing only: the random table is no extension)
b = Backend()
n = 1 << log_n
lde = torch.randint(0, P, (8, 94, n), dtype=torch.int64, device=b.device)
cf = _lib.TxCoeffsStruct()
rng = np.random.default_rng(1)
for name, k in (("t_alpha", 115), ("t_beta", 115), ("b_alpha", 4), ("b_beta", 4)):
    v = rng.integers(1, P, size=k, dtype=np.uint64)
    for i in range(k):
        getattr(cf, name)[i] = int(v[i])
pub = [1, 2, 3, 4]
out = b.empty_u64(8, n)
b.evaluate_constraints(lde, cf, pub, 15, out=out, input_is_lde=split)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    b.evaluate_constraints(lde, cf, pub, 15, out=out, input_is_lde=split)
e1.record()
torch.cuda.synchronize()
print("constraints: %.3f ms per evaluation (2^%d x 8 points)" % (e0.elapsed_time(e1) / reps, log_n))
b.set_part_timing(True)
b.evaluate_constraints(lde, cf, pub, 15, out=out, input_is_lde=split)
print("parts:", {k: round(v, 3) for k, v in b.constraint_part_ms().items()})
