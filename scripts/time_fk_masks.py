"""f-k filter pass times at 20000 x 120000 for the masks of the bench line (HIP events per pass), without the rest of the bench.
usage: [D4W_FK_PACKB=0] python scripts/time_fk_masks.py [classic ninf dense step4 ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import das4whales_amd as dw
nx, ns, fs, dx = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000)), 200.0, 2.0419046878814697
which = sys.argv[1:] or ["classic", "ninf", "dense"]
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((nx, ns), device="cuda", generator=g)
y = torch.empty_like(x)
plan = dw.dsp.get_fk_plan(nx, ns)
for w in which:
    if w == "classic":
        m = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)
    elif w == "ninf":
        m = dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
    elif w == "step4":
        m = dw.dsp.fk_filter_design((nx, ns), [0, 4 * nx, 4], dx, fs)
    elif w == "hybrid":
        m = dw.dsp.hybrid_filter_design((nx, ns), [0, nx, 1], dx, fs, 1350., 1450., 14., 30.)
    else:
        m = torch.rand((nx, ns), device="cuda", generator=g)
    plan.set_mask(m)
    del m
    plan.apply(x, out=y)
    acc = np.zeros(5)
    n = 8
    for _ in range(n):
        _, ms = plan.apply_timed(x, out=y)
        acc += np.array(ms)
    acc /= n
    od = plan.order()
    print(json.dumps({"mask": w, "order": od["order"], "packB": os.environ.get("D4W_FK_PACKB", "1"), "ms": [round(float(v), 3) for v in acc],
                      "total_ms": round(float(acc.sum()), 3), "frac_24B": round(24.0 * nx * ns / (acc.sum() * 1e-3) / 8e12, 4),
                      "live_rows": od["live_wavenumber_rows"], "band": od["band_columns"], "tail": od["tail_columns"]}), flush=True)
