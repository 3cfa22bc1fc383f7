"""Does the matrix-core STFT write outside its output?  The spectrogram goes into the middle of a buffer whose margins hold a
sentinel; so does the input (reads cannot be checked, but a write into the input's margins would show)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from das4whales_amd._lib import lib, check
nx, ns, n_fft, hop, lo, hi = 11020, 12000, 160, 8, 11, 23
nt = int(lib.d4w_stft_frames(ns, hop))
G = 1 << 22
size = nx * (hi - lo + 1) * nt
buf = torch.full((size + 2 * G,), 12345.0, device="cuda")
xin = torch.full((nx * ns + 2 * G,), 777.0, device="cuda")
xin[G:G + nx * ns] = torch.randn(nx * ns, device="cuda")
S = buf[G:G + size]
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    check(lib.d4w_stft_mag_f32(xin[G:].data_ptr(), S.data_ptr(), None, nx, ns, n_fft, hop, lo, hi, st))
torch.cuda.synchronize()
print("output margins intact:", bool((buf[:G] == 12345.0).all().cpu()), bool((buf[G + size:] == 12345.0).all().cpu()),
      "| input margins intact:", bool((xin[:G] == 777.0).all().cpu()), bool((xin[G + nx * ns:] == 777.0).all().cpu()),
      "| output finite:", bool(torch.isfinite(S).all().cpu()), "| untouched outputs:", int((S == 12345.0).sum().cpu()))
