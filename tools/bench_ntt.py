"""Times interpolation + 8-coset LDE of 94 columns of 2^log_n random field elements."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd.backend import Backend
P = 2**62 + 2**56 + 2**55 + 1
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b = Backend(); n = 1 << log_n
ev = torch.randint(0, P, (94, n), dtype=torch.int64, device=b.device)
co = b.empty_u64(94, n); lde = b.empty_u64(8, 94, n)
for rep in range(3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); b.interpolate_columns(ev, out=co); e[1].record(); b.lde_columns(co, 3, out=lde); e[2].record()
    torch.cuda.synchronize()
print("interpolate %.3f ms, lde %.3f ms" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
