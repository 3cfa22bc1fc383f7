"""Two-template matched filter at 20000 x 120000 for the kernel selected by D4W_XF_FUSED (1 = four-stage fused kernel [default],
2 = three-stage fused kernel, 0 = one launch per template): HIP-event median and the difference to the direct form on 64 rows.
(Round 3 also timed a sequential-template variant with it, docs/LAB_NOTEBOOK.md section 9.)"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import detect as ddet
nx, ns, fs = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000)), 200.0
x = torch.randn((nx, ns), device="cuda")
t = np.arange(ns) / fs
tpl = [ddet._normalised_support(ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)),
       ddet._normalised_support(ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78))]
ts = []
for i in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ys = ddet._xcorr_device(x, tpl, normalize=False, method="fft"); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b)); del ys
yf = ddet._xcorr_device(x[:64], tpl, True, "fft"); yd = ddet._xcorr_device(x[:64], tpl, True, "direct")
print(json.dumps({"mode": os.environ.get("D4W_XF_FUSED", "1"), "ms": float(np.median(ts[2:])),
                  "rel_vs_direct": [float((yf[k] - yd[k]).abs().max() / yd[k].abs().max()) for k in range(2)]}))
