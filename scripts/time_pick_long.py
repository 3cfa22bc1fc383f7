import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import das4whales_amd as dw
nx, ns = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000))
fs = 200.0
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 3)
torch.manual_seed(0)
x = torch.randn((nx, ns), device="cuda")
t = np.arange(ns) / fs
hf = dw.detect.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
c = dw.detect.compute_cross_correlogram(x, hf)
del x
out = {"shape": [nx, ns]}
for frac in (0.45, 0.2, 0.8, 1e9):
    thr = frac * float(c.max())
    out["pick_times thr=%g max" % frac] = ev(lambda: dw.detect.pick_times(c, thr))
    out["picks %g" % frac] = int(dw.detect.pick_times(c, thr).total)
thr = 0.45 * float(c.max())
out["pick_times_env"] = ev(lambda: dw.detect.pick_times_env(c, thr))
print(json.dumps(out))
