"""Overlap-save FIR kernel on one 11020 x 12000 file: interior-only form (d4w_fir_fft_f32) against the halo form the stream
uses (d4w_fir_fft_halo_f32, neighbours read in place)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.signal as sp
from das4whales_amd import dsp
from das4whales_amd._lib import lib, check
from das4whales_amd import _device as dev
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
x = torch.randn((nx, ns), device="cuda")
left = torch.randn((nx, 1024), device="cuda"); right = torch.randn((nx, 1024), device="cuda")
sos = np.ascontiguousarray(sp.butter(8, [0.14, 0.30], "bp", output="sos"))
def ev(fn, reps=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
t, K, E, dcg = dsp._zero_phase_taps(sos, x.device)
y = torch.empty_like(x); first = x[:, 0].contiguous()
ws = torch.empty(int(lib.d4w_xcorr_fft_ws_bytes()), dtype=torch.uint8, device=x.device)
check(lib.d4w_fir_fft_f32(dev.ptr(x), nx, ns, dev.ptr(t), int(K), dev.ptr(first), dcg, dev.ptr(y), dev.ptr(ws), dev.stream_ptr(x)))
out = {"shape": [nx, ns], "K": K}
out["interior_only_ms"] = ev(lambda: check(lib.d4w_fir_fft_f32(dev.ptr(x), nx, ns, None, int(K), dev.ptr(first), dcg, dev.ptr(y), dev.ptr(ws), dev.stream_ptr(x))))
out["between_neighbours_ms"] = ev(lambda: dsp._sosfiltfilt_between(x, left, right, sos))
print(json.dumps(out))
