"""Times the Rescue-window part of the split constraint evaluation (launch_rounds_setup + the window kernel) on random field data:
median and minimum over `reps` evaluations with part timing.  usage: bench_rounds.py [log_n] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd import _lib
from certificate_stark_amd.backend import Backend

P = 2**62 + 2**56 + 2**55 + 1
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
b = Backend()
n = 1 << log_n
lde = torch.randint(0, P, (8, 94, n), dtype=torch.int64, device=b.device)
cf = _lib.TxCoeffsStruct()
rng = np.random.default_rng(1)
for name, k in (("t_alpha", 115), ("t_beta", 115), ("b_alpha", 4), ("b_beta", 4)):
    v = rng.integers(1, P, size=k, dtype=np.uint64)
    for i in range(k):
        getattr(cf, name)[i] = int(v[i])
out = b.empty_u64(8, n)
b.set_part_timing(True)
ts = []
for _ in range(reps + 2):
    b.evaluate_constraints(lde, cf, [1, 2, 3, 4], 15, out=out, input_is_lde=True)
    ts.append(b.constraint_part_ms()["rounds"])
ts = sorted(ts[2:])
print("rounds: median %.3f ms, min %.3f ms over %d evaluations (2^%d x 4 even cosets)" % (ts[len(ts) // 2], ts[0], reps, log_n))
