"""Zero-phase band-pass (dsp.bp_filt, 14-30 Hz, order 8) at 20 000 x 120 000: the polyphase cascade on the matrix cores
(D4W_BP_POLY=1, fir_mm.hip) against the overlap-save FFT form; accuracy of both against the float64 oracle on a few rows;
the kernel alone (HIP events around d4w_fir_poly_f32)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import das4whales_amd as dw
from das4whales_amd import dsp, _device as dev
from das4whales_amd._lib import lib, check
from oracle import d4w_oracle as orc      # checker only

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

nx, ns = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000))
gen = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn((nx, ns), device="cuda", generator=gen) + 0.5
rows = [0, 1, nx // 2, nx - 1]
ref = orc.bp_filt(x[rows].cpu().numpy().astype(np.float64), 200.0, 14, 30)
out = {"shape": [nx, ns]}
for mode in ("1", "0"):
    os.environ["D4W_BP_POLY"] = mode
    ms = timed(lambda: dw.dsp.bp_filt(x, 200.0, 14, 30))
    y = dw.dsp.bp_filt(x, 200.0, 14, 30)
    err = float(np.max(np.abs(y[rows].cpu().numpy() - ref)) / np.max(np.abs(ref)))
    out["polyphase" if mode == "1" else "fft"] = {"ms": round(ms, 3), "rel_err_vs_f64_oracle": err,
                                                   "frac_of_8B_roofline": round(8.0 * nx * ns / (ms * 1e-3) / 8e12, 4)}
sos = dsp.butterworth_filter([8, [14, 30], "bp"], 200.0) if hasattr(dsp, "butterworth_filter") else None
pt = dsp._poly_taps(np.asarray(sos), x.device)
y = torch.empty_like(x)
first = x[:, 0].contiguous()
out["kernel_ms"] = round(timed(lambda: check(lib.d4w_fir_poly_f32(dev.ptr(x), nx, ns, dev.ptr(first), 0.0, dev.ptr(pt[0]), dev.ptr(pt[1]),
                                                                      dev.ptr(pt[2]), dev.ptr(y), dev.stream_ptr(x)))), 3)
print(json.dumps(out), flush=True)
