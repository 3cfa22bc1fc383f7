"""Zero-phase band-pass (dsp.bp_filt, 14-30 Hz, order 8): overlap-save FFT form (interior) + recursion at the row ends
against the recursion alone, several shapes; accuracy of both against the float64 oracle on a few rows."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import das4whales_amd as dw
from oracle import d4w_oracle as orc      # checker only

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

for nx, ns in ((4000, 12000), (11020, 12000), (20000, 120000)):
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((nx, ns), device="cuda", generator=gen) + 0.5
    rows = [0, 1, nx // 2, nx - 1]
    ref = orc.bp_filt(x[rows].cpu().numpy().astype(np.float64), 200.0, 14, 30)
    out = {"shape": [nx, ns]}
    for mode in ("1", "0"):
        os.environ["D4W_BP_FFT"] = mode
        ms = timed(lambda: dw.dsp.bp_filt(x, 200.0, 14, 30))
        y = dw.dsp.bp_filt(x, 200.0, 14, 30)
        err = float(np.max(np.abs(y[rows].cpu().numpy() - ref)) / np.max(np.abs(ref)))
        out["fft+edges" if mode == "1" else "recursion"] = {"ms": round(ms, 3), "rel_err_vs_f64_oracle": err,
                                                             "frac_of_8B_roofline": round(8.0 * nx * ns / (ms * 1e-3) / 8e12, 4)}
    print(json.dumps(out), flush=True)
    del x
