import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from das4whales_amd import dsp, detect
from das4whales_amd._lib import lib, check
nx, ns = int(os.environ.get("NX", 4000)), int(os.environ.get("NS", 120000))
torch.manual_seed(0)
x = torch.randn((nx, ns), device="cuda")
x = dsp.bp_filt(x, 200.0, 14, 30)
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
out = {"shape": [nx, ns]}
out["envelope_long_ms"] = ev(lambda: dsp._analytic(x, 0))
env = dsp._analytic(x, 0)
thr = 0.45 * float(env.max())
cap = 4096
idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda"); cnt = torch.empty(nx, dtype=torch.int32, device="cuda")
out["find_peaks_ms"] = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(thr), idx.data_ptr(), cnt.data_ptr(), cap, None)))
out["picks"] = int(cnt.sum())
out["find_peaks_thr_inf_ms"] = ev(lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, 1e30, idx.data_ptr(), cnt.data_ptr(), cap, None)))
print(json.dumps(out))
