import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.signal as sp
import das4whales_amd as dw
from das4whales_amd import dsp, _device as dev
from das4whales_amd._lib import lib, check
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
nx, ns = 20000, 120000
x = torch.randn((nx, ns), device="cuda")
sos = sp.butter(8, [0.14, 0.30], "bp", output="sos")
for tol in (1e-8, 1e-7, 1e-6):
    taps, K, E = dsp._zero_phase_response(sos, tol_taps=tol)
    t = torch.from_numpy(np.ascontiguousarray(taps, dtype=np.float32)).cuda()
    first = x[:, 0].contiguous(); y = torch.empty_like(x)
    ws = torch.empty(int(lib.d4w_xcorr_fft_ws_bytes()), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: check(lib.d4w_fir_fft_f32(dev.ptr(x), nx, ns, dev.ptr(t), int(K), dev.ptr(first), 0.0, dev.ptr(y), dev.ptr(ws), dev.stream_ptr(x))))
    print("tol %g: K %d E %d fir_fft %.3f ms" % (tol, K, E, ms))
P = 2 * E
piece = x[:, :P].contiguous()
print("piece copy %.3f ms" % timed(lambda: x[:, :P].contiguous()))
print("recursion on piece [%d x %d] %.3f ms" % (nx, P, timed(lambda: dsp._sosfiltfilt_recursive(piece, sos, 51, 0, 0))))
print("recursion on piece with default segmentation %.3f ms" % timed(lambda: dsp._sosfiltfilt_recursive(piece, sos, 51)))
yp = dsp._sosfiltfilt_recursive(piece, sos, 51, 0, 0)
print("copy back %.3f ms" % timed(lambda: y[:, :E].copy_(yp[:, :E])))
print("whole bp_filt %.3f ms" % timed(lambda: dsp.bp_filt(x, 200.0, 14, 30)))
