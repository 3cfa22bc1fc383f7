"""Where the FFT-form band-pass spends its time at 20 000 x 120 000: the overlap-save kernel alone, the row-end pieces
(cat, recursion, copies back) -- HIP events around each part."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import das4whales_amd as dw
from das4whales_amd import dsp
from das4whales_amd._lib import lib, check
from das4whales_amd import _device as dev

def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return {"mean": round(float(np.mean(ts)), 3), "min": round(min(ts), 3), "max": round(max(ts), 3)}

nx, ns = 20000, 120000
x = torch.randn((nx, ns), device="cuda") + 0.5
import scipy.signal as sp
b, a = None, None
sos = sp.butter(8, [14 / 100.0, 30 / 100.0], "bp", output="sos")
sos = np.ascontiguousarray(sos, dtype=np.float64)
y = dsp._sosfiltfilt_fft(x, sos, 51)              # fills the tap cache
key = (sos.tobytes(), sos.shape, str(x.device))
t, K, E, dcg = dsp._zp_cache[key]
P = 2 * E
out = {"shape": [nx, ns], "K": K, "E": E}
first = x[:, 0].contiguous()
ws = torch.empty(int(lib.d4w_xcorr_fft_ws_bytes()), dtype=torch.uint8, device=x.device)
out["fir_kernel"] = timed(lambda: check(lib.d4w_fir_fft_f32(dev.ptr(x), nx, ns, dev.ptr(t), int(K), dev.ptr(first), dcg, dev.ptr(y),
                                                              dev.ptr(ws), dev.stream_ptr(x))))
out["first_col"] = timed(lambda: x[:, 0].contiguous())
out["cat"] = timed(lambda: torch.cat((x[:, :P], x[:, ns - P:]), dim=0))
ends = torch.cat((x[:, :P], x[:, ns - P:]), dim=0)
out["recursion_on_ends"] = timed(lambda: dsp._sosfiltfilt_recursive(ends, sos, 51, 0, 0))
ye = dsp._sosfiltfilt_recursive(ends, sos, 51, 0, 0)
def back():
    y[:, :E] = ye[:nx, :E]
    y[:, ns - E:] = ye[nx:, P - E:]
out["copies_back"] = timed(back)
out["whole"] = timed(lambda: dsp._sosfiltfilt_fft(x, sos, 51))
out["empty_like"] = timed(lambda: torch.empty_like(x))
print(json.dumps(out))
