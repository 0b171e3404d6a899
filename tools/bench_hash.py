"""Times the Blake3 row hashing of an 8 x 94 x 2^20 table alone (CSTARK_LIB selects a variant build)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd.backend import Backend
b = Backend(); n = 1 << 20
lde = torch.randint(0, 2**62, (8, 94, n), dtype=torch.int64, device=b.device)
leaves = b.hash_rows(lde, 3)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); leaves = b.hash_rows(lde, 3); e1.record(); torch.cuda.synchronize()
print("hash_rows 8 x 94 x 2^20: %.3f ms" % e0.elapsed_time(e1))
