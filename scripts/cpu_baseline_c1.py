"""CPU column of BASELINE.md section 3 at the config-1 shape, in full: the NumPy / SciPy float64 restatement of the
reference path (oracle/d4w_oracle.py -- /root/reference is absent on the GPU box) on ONE 4000 x 12000 block,
time.perf_counter, one warm-up + best of 3, design time excluded, plus the best-effort column (float32 half-spectrum
scipy.fft on all cores).  Prints one JSON line; committed under profiles/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import d4w_oracle as orc

nx, ns, fs, dx = 4000, 12000, 200.0, 2.0419046878814697
sel = [9794, 25794, 4]
rng = np.random.default_rng(1234)
x = rng.standard_normal((nx, ns))
mask = np.ascontiguousarray(orc.hybrid_ninf_filter_design((nx, ns), sel, dx, fs, 1350., 1450., 3300, 3450, 14., 30.))
t = np.arange(ns) / fs
hf = orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
lf = orc.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)

def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)

res = {"shape": [nx, ns], "cores_available": len(os.sched_getaffinity(0)), "threads_used": 1,
       "note": "reference path is single-threaded (numpy.fft pocketfft, scipy.signal.filtfilt / correlate do not thread)"}
samples = nx * ns
tf = best_of(lambda: orc.fk_filter_filt(x, mask))
tb = best_of(lambda: orc.bp_filt(x, fs, 14, 30))
tm = best_of(lambda: (orc.compute_cross_correlogram(x, hf), orc.compute_cross_correlogram(x, lf)))
res["fk_filter_filt_s"] = tf
res["bp_filt_s"] = tb
res["compute_cross_correlogram_x2_s"] = tm
res["channel_samples_per_s"] = {"fk": samples / tf, "fk+mf": samples / (tf + tm), "bp+fk+mf": samples / (tb + tf + tm)}
x32 = x.astype(np.float32)
mh = orc.fold_mask_half(mask).astype(np.float32)
tfb = best_of(lambda: orc.fk_filter_filt_best_effort(x32, mh))
tmb = best_of(lambda: orc.compute_cross_correlogram_best_effort(x32, [hf[:136], lf[:156]]))
res["best_effort"] = {"threads_used": len(os.sched_getaffinity(0)), "fk_s": tfb, "mf_x2_s": tmb,
                      "channel_samples_per_s": {"fk": samples / tfb, "fk+mf": samples / (tfb + tmb)}}
print(json.dumps(res))
