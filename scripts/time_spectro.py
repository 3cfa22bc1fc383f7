"""Spectrogram detector (detect.compute_cross_correlogram_spectrocorr) of an 11 020 x 12 000 block: HIP-event median of the call."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import detect
nx, ns, fs = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000)), 200.0
x = torch.randn((nx, ns), device="cuda")
kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
f = lambda: detect.compute_cross_correlogram_spectrocorr(x, fs, [14., 30.], kernel, 0.8, 0.95)
f(); torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
print(json.dumps({"shape": [nx, ns], "spectrocorr_ms": float(np.median(ts)), "min": float(min(ts)), "dbg": os.environ.get("D4W_SF_DBG", "0"), "note": "D4W_SF_DBG / D4W_SPECTRO_FUSED exist only in a build with scripts/probe/spectro_fused.h pasted in"}))
