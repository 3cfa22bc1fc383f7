"""find_peaks kernel time on different row contents (which part of the kernel costs what)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import dsp
from das4whales_amd._lib import lib, check
nx, ns = int(os.environ.get("NX", 4000)), int(os.environ.get("NS", 12000))
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
g = torch.Generator(device="cuda").manual_seed(1)
noise = torch.randn((nx, ns), device="cuda", generator=g)
inputs = {"zeros": torch.zeros((nx, ns), device="cuda"), "ramp": torch.arange(ns, device="cuda", dtype=torch.float32).repeat(nx, 1),
          "white noise": noise, "envelope of band-passed noise": dsp._analytic(dsp.bp_filt(noise, 200.0, 14, 30), 0)}
cap = ns // 2 + 1
idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda"); cnt = torch.empty(nx, dtype=torch.int32, device="cuda")
out = {}
for name, t in inputs.items():
    ms = ev(lambda: check(lib.d4w_find_peaks_f32(t.data_ptr(), nx, ns, 0.5, idx.data_ptr(), cnt.data_ptr(), cap, None)))
    out[name] = {"ms": round(ms, 3), "peaks_per_row": float(cnt.float().mean())}
print(json.dumps(out))
