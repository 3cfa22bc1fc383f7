"""pick_times_env on an 11020 x 12000 correlogram-like block: envelope kernel, find_peaks kernel, compaction, whole call."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import dsp, detect
from das4whales_amd._lib import lib, check
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
torch.manual_seed(0)
x = torch.randn((nx, ns), device="cuda")
x = dsp.bp_filt(x, 200.0, 14, 30)
t = np.arange(ns) / 200.0
hf = detect.gen_template_fincall(t, 200.0, 17.8, 28.8, 0.68)
c = detect.compute_cross_correlogram(x, hf)
thr = 0.45 * float(c.max())
def ev(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
out = {}
out["envelope kernel ms"] = ev(lambda: dsp._analytic(c, 0))
env = dsp._analytic(c, 0)
cap = 1024
idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda"); cnt = torch.empty(nx, dtype=torch.int32, device="cuda")
out["find_peaks kernel ms"] = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(thr), idx.data_ptr(), cnt.data_ptr(), cap, None)))
out["_find_peaks_device ms"] = ev(lambda: detect._find_peaks_device(env, thr))
out["pick_times_env ms"] = ev(lambda: detect.pick_times_env(c, thr))
out["c.max() sync ms"] = ev(lambda: float(c.max()))
out["n_picks"] = detect.pick_times_env(c, thr).total
print(json.dumps(out))
for name, tt in (("thr=inf (no walks)", 1e30), ("thr=0.45max", thr), ("thr=0.2max", thr * 0.2 / 0.45), ("thr=0 (all maxima)", 0.0)):
    cc = cap if tt > 0 else ns // 2 + 1
    idx = torch.empty((nx, cc), dtype=torch.int32, device="cuda")
    ms = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(tt), idx.data_ptr(), cnt.data_ptr(), cc, None)))
    print("find_peaks %s: %.3f ms, picks %d" % (name, ms, int(cnt.sum())))
emax = float(env.max())
for frac in (1.5, 0.9, 0.7, 0.5, 0.35):
    tt = frac * emax
    idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda")
    ms = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(tt), idx.data_ptr(), cnt.data_ptr(), cap, None)))
    print("find_peaks thr=%.2f max(env): %.3f ms, picks %d" % (frac, ms, int(cnt.sum())))
