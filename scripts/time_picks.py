"""Where the time of pick_times_env goes at 4000 x 12000: envelope kernel, find_peaks kernel, host gather."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import dsp, detect
from das4whales_amd._lib import lib, check
nx, ns = int(os.environ.get("NX", 4000)), int(os.environ.get("NS", 12000))
x = torch.randn((nx, ns), device="cuda")
x = dsp.bp_filt(x, 200.0, 14, 30)
env = dsp._analytic(x, 0)
thr = float(env.max()) * 0.45
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
out = {}
cap = 1024
idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda"); cnt = torch.empty(nx, dtype=torch.int32, device="cuda")
for name, t in (("high thr", thr), ("thr 0 (every local max)", 0.0)):
    c = cap if t > 0 else ns // 2 + 1
    idx = torch.empty((nx, c), dtype=torch.int32, device="cuda")
    out["find_peaks kernel ms, " + name] = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(t), idx.data_ptr(), cnt.data_ptr(), c, None)))
t0 = time.perf_counter(); p = detect._find_peaks_device(env, thr); torch.cuda.synchronize(); out["_find_peaks_device wall ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); p = detect.pick_times_env(x, thr); torch.cuda.synchronize(); out["pick_times_env wall ms"] = (time.perf_counter() - t0) * 1e3
out["n_picks"] = int(sum(len(q) for q in p))
print(json.dumps(out))
