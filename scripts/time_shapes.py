"""Per-pass f-k times (HIP events), dense random mask (MASK=hybrid_ninf: the scripts' band mask), for any shapes (built-in,
compiled on demand or generic kernels)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(13223, 12000), (8000, 12000), (6000, 12000), (11020, 12000), (300, 12002)]
for nx, ns in shapes:
    x = torch.randn((nx, ns), device="cuda")
    dw.dsp.compile_fk_shape(nx, ns)
    plan = dw.dsp.FkPlan(nx, ns)
    # MASK=hybrid_ninf: the scripts' band mask (scripts/main_mfdetect.py:46-47) instead of a dense random one -- what a
    # Bluestein channel phase in global memory is timed with (it transforms the live columns only)
    prune = float(os.environ.get("PRUNE", 0.0))          # FkPlan.set_mask(..., prune_eps=): opt-in gain level treated as zero
    if os.environ.get("MASK", "dense") == "hybrid_ninf":
        plan.set_mask(dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0, 1350., 1450., 3300, 3450, 14., 30.), prune_eps=prune)
    elif os.environ.get("MASK", "dense") == "classic":
        plan.set_mask(dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0), prune_eps=prune)
    else:
        plan.set_mask(torch.rand((nx, ns), device="cuda"))
    y = torch.empty_like(x)
    plan.apply(x, out=y)
    acc = [0.0] * 5
    for _ in range(5):
        _, ms = plan.apply_timed(x, out=y)
        acc = [a + b / 5 for a, b in zip(acc, ms)]
    print(json.dumps({"shape": [nx, ns], "specialised": bool(dw.fkjit.is_specialised(nx, ns)), "plan": plan.info(), "mask": os.environ.get("MASK", "dense"), "prune_eps": float(os.environ.get("PRUNE", 0.0)), "passes_ms": [round(a, 3) for a in acc], "total_ms": round(sum(acc), 3),
                      "GBps_24B": round(24.0 * nx * ns / (sum(acc) * 1e-3) / 1e9, 1)}), flush=True)
