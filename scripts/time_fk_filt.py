"""dsp.fk_filt (self-designing Gaussian-tapered speed band, reference dsp.py:883-953): design + blur + normalise + fold + apply
per call, on a resident block."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import das4whales_amd as dw
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for nx, ns in ((4000, 12000), (11020, 12000), (20000, 120000)):
    x = torch.randn((nx, ns), device="cuda")
    out = {"shape": [nx, ns]}
    out["fk_filt_ms"] = ev(lambda: dw.dsp.fk_filt(x, 1, 200.0, 1, 2.0419, 1400.0, 3400.0))
    out["hybrid_gs_design_ms"] = ev(lambda: dw.dsp.hybrid_gs_filter_design((nx, ns), [0, nx, 1], 2.0419, 200.0))
    print(json.dumps(out), flush=True)
    del x
