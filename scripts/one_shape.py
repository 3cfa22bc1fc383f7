"""One f-k plan applied a few times to a random block (for rocprofv3 / PMC runs of a single shape):
python scripts/one_shape.py NX NS [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw
nx, ns = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
x = torch.randn((nx, ns), device="cuda")
m = torch.rand((nx, ns), device="cuda")
plan = dw.dsp.get_fk_plan(nx, ns)
plan.set_mask(m)
y = torch.empty_like(x)
for _ in range(reps):
    plan.apply(x, out=y)
torch.cuda.synchronize()
print(plan.info())
