"""Numerical check of the decimate-by-2 polyphase form of the zero-phase band-pass:
   y = 2 b * up2( g2 * down2( a * x ) )  against  y = g * x, g the two-sided response of sosfiltfilt's cascade.
a = b = Kaiser low-pass of N taps, g2 = half-rate taps with spectrum G / (A B) on |f| <= fs/4."""
import sys
import numpy as np
import scipy.signal as sp

fs = 200.0
lo, hi = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (14.0, 30.0)
sos = sp.butter(8, [lo / (fs / 2), hi / (fs / 2)], "bp", output="sos")
n = 1 << 15
imp = np.zeros(n); imp[0] = 1
h = sp.sosfilt(sos, imp)
H = np.fft.rfft(h, 2 * n)
G = (H * np.conj(H)).real                        # G(f) on 2n-point grid, f = k / (2n)
g = np.fft.irfft(G, 2 * n)
g = np.concatenate((g[-8192:], g[:8193]))       # centred, long
def design(N, beta, fc):
    a = sp.firwin(N, fc, window=("kaiser", beta), fs=1.0)
    return a
rng = np.random.default_rng(0)
x = rng.standard_normal(1 << 16)
yref = sp.fftconvolve(x, g, "same")
for N, beta, fc in ((63, 12.0, 0.25), (63, 13.0, 0.25), (47, 12.0, 0.25), (79, 13.0, 0.25), (63, 12.0, 0.24), (63, 14.0, 0.26)):
    a = design(N, beta, fc)
    # half-rate g2: spectrum G(f)/(A(f)^2) for |f| <= 1/4, sampled on the half-rate grid
    M = 1 << 14                                   # half-rate FFT length
    f = np.fft.rfftfreq(M, d=2.0)                 # cycles per full-rate sample, 0 .. 1/4
    A = np.abs(np.fft.rfft(a, 1 << 16))           # zero-phase magnitude on a fine grid (a symmetric)
    fa = np.fft.rfftfreq(1 << 16)
    Af = np.interp(f, fa, A)
    Gf = np.interp(f, np.fft.rfftfreq(2 * n), G)
    G2 = Gf / np.maximum(Af ** 2, 1e-30)
    g2 = np.fft.irfft(G2, M)
    g2 = np.concatenate((g2[-2048:], g2[:2049]))
    tail = np.cumsum(np.abs(g2[2048:])[::-1])[::-1] / np.sum(np.abs(g2[2048:]))
    K2 = int(np.nonzero(tail < 1e-7)[0][0])
    g2t = g2[2048 - K2:2048 + K2 + 1]
    u = sp.fftconvolve(x, a, "same")[::2]
    v = sp.fftconvolve(u, g2t, "same")
    w = np.zeros_like(x); w[::2] = v
    y = 2 * sp.fftconvolve(w, a, "same")
    s = slice(4096, -4096)
    print("N=%d beta=%.0f fc=%.2f: half-rate taps %d, err %.2e" % (N, beta, fc, 2 * K2 + 1, np.max(np.abs(y[s] - yref[s])) / np.max(np.abs(yref[s]))))
