"""One timing per public function of the mirror on a resident block (HIP events, median of 3): a sweep to spot paths that are
out of proportion.  NX / NS select the block (default 20000 x 120000)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import das4whales_amd as dw
nx, ns = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000))
fs, dx = 200.0, 2.0419046878814697
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 3)
x = torch.randn((nx, ns), device="cuda")
t = np.arange(ns) / fs
hf = dw.detect.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
out = {"shape": [nx, ns]}
out["taper_data"] = ev(lambda: dw.dsp.taper_data(x))
out["snr_tr_array"] = ev(lambda: dw.dsp.snr_tr_array(x))
out["snr_tr_array_env"] = ev(lambda: dw.dsp.snr_tr_array(x, env=True))
out["bp_filt"] = ev(lambda: dw.dsp.bp_filt(x, fs, 14, 30))
sos = dw.dsp.butterworth_filter([2, 5.0, "hp"], fs)
out["sosfiltfilt_hp2"] = ev(lambda: dw.dsp.sosfiltfilt(sos, x, axis=1))
out["compute_cross_correlogram_1tpl"] = ev(lambda: dw.detect.compute_cross_correlogram(x, hf))      # (row statistics remembered)
def fresh():
    torch.autograd.graph.increment_version(x)          # as if x had just been written: the statistics are formed again
    return dw.detect.compute_cross_correlogram(x, hf)
out["compute_cross_correlogram_1tpl_fresh_block"] = ev(fresh)
lf = dw.detect.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
out["compute_cross_correlograms_2tpl"] = ev(lambda: dw.detect.compute_cross_correlograms(x, [hf, lf]))
c = dw.detect.compute_cross_correlogram(x, hf)
thr = 0.45 * float(c.max())
out["pick_times"] = ev(lambda: dw.detect.pick_times(c, thr))
out["pick_times_env"] = ev(lambda: dw.detect.pick_times_env(c, thr))
pk = dw.detect.pick_times_env(c, thr)
out["convert_pick_times"] = ev(lambda: dw.detect.convert_pick_times(pk))
del c
m1 = dw.dsp.hybrid_filter_design((nx, ns), [0, nx, 1], dx, fs)
out["hybrid_filter_design_dense"] = ev(lambda: dw.dsp.hybrid_filter_design((nx, ns), [0, nx, 1], dx, fs).tensor)
out["fk_filter_sparsefilt_hybrid"] = ev(lambda: dw.dsp.fk_filter_sparsefilt(x, m1))
out["fk_filter_filt_taper"] = ev(lambda: dw.dsp.fk_filter_filt(x, m1, tapering=True))
del m1
dw.dsp.clear_fk_plans()
torch.cuda.empty_cache()
if nx * ns <= 200e6:
    ker = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
    out["spectrocorr"] = ev(lambda: dw.detect.compute_cross_correlogram_spectrocorr(x, fs, [14., 30.], ker, 0.8, 0.95))
out["get_spectrogram_1ch"] = ev(lambda: dw.dsp.get_spectrogram(x[0], fs, nfft=256, overlap_pct=0.95))
out["instant_freq_1ch"] = ev(lambda: dw.dsp.instant_freq(x[0], fs))
out["get_fx_400"] = ev(lambda: dw.dsp.get_fx(x[:, :400].contiguous(), 512))
print(json.dumps(out))
