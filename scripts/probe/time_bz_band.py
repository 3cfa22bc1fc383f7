"""The global-memory Bluestein channel phase (a prime channel count) with masks of different support: which columns it
transforms (d4w_fkd_plan_live_columns' rule) decides its time.  19997 x 120000 by default."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import das4whales_amd as dw
nx, ns = int(os.environ.get("NX", 19997)), int(os.environ.get("NS", 120000))
x = torch.randn((nx, ns), device="cuda")
plan = dw.dsp.FkPlan(nx, ns)
y = torch.empty_like(x)
cases = [("hybrid_filter_design (cosine tapers, 14-30 Hz)", lambda: dw.dsp.hybrid_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0, 1350., 1450., 14., 30.), 0.0),
         ("hybrid_ninf_filter_design", lambda: dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0, 1350., 1450., 3300, 3450, 14., 30.), 0.0),
         ("hybrid_ninf_filter_design, prune_eps 1e-4", lambda: dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0, 1350., 1450., 3300, 3450, 14., 30.), 1e-4),
         ("fk_filter_design (velocity fan, all frequencies)", lambda: dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0), 0.0)]
for name, make, prune in cases:
    plan.set_mask(make(), prune_eps=prune)
    plan.apply(x, out=y)
    _, ms = plan.apply_timed(x, out=y)
    _, ms = plan.apply_timed(x, out=y)
    print(json.dumps({"shape": [nx, ns], "mask": name, "passes_ms": [round(a, 2) for a in ms], "total_ms": round(sum(ms), 2)}), flush=True)
