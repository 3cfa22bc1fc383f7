import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw
nx, ns = 13223, 12000
x = torch.randn((nx, ns), device="cuda")
m = torch.rand((nx, ns), device="cuda")
y = torch.empty_like(x)
ref = None
for opts in (None, (7, 1889, 12, 500, 16, 4), (7, 1889, 24, 250, 16, 4), (7, 1889, 30, 200, 16, 4), (7, 1889, 60, 100, 16, 4), (7, 1889, 12, 500, 16, 2), (7, 1889, 12, 500, 16, 1)):
    try:
        plan = dw.dsp.FkPlan(nx, ns, opts=opts)
    except Exception as e:
        print(opts, "ERR", str(e)[:100]); continue
    plan.set_mask(m)
    plan.apply(x, out=y)
    acc = [0.0] * 5
    for _ in range(4):
        _, ms = plan.apply_timed(x, out=y)
        acc = [a + b / 4 for a, b in zip(acc, ms)]
    if ref is None: ref = y.clone()
    err = float((y - ref).abs().max() / ref.abs().max())
    print(json.dumps({"opts": opts, "plan": plan.info(), "passes_ms": [round(a, 3) for a in acc], "total_ms": round(sum(acc), 3), "vs_first": err}), flush=True)
