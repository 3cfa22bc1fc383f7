"""Timing of the matched filter variants on a resident block: HIP events, median of reps."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import detect as ddet
nx, ns, fs = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000)), 200.0
x = torch.randn((nx, ns), device="cuda")
t = np.arange(ns) / fs
tpl = [ddet._normalised_support(ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)),
       ddet._normalised_support(ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78))]
out = {}
for method in ("fft", "direct"):
    for norm in (True, False):
        ts = []
        for i in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ys = ddet._xcorr_device(x, tpl, normalize=norm, method=method); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b)); del ys
        out["%s%s" % (method, "+stats" if norm else "")] = float(np.median(ts[1:]))
yf = ddet._xcorr_device(x[:64], tpl, True, "fft"); yd = ddet._xcorr_device(x[:64], tpl, True, "direct")
out["fft_vs_direct_rel"] = float((yf[0] - yd[0]).abs().max() / yd[0].abs().max())
print(json.dumps(out))
ts = []
for i in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y0 = ddet._xcorr_device(x, tpl[:1], normalize=False, method="fft"); y1 = ddet._xcorr_device(x, tpl[1:], normalize=False, method="fft"); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b)); del y0, y1
print(json.dumps({"fft_two_single_template_launches": float(np.median(ts[1:]))}))
