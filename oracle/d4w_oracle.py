"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A float64 NumPy restatement of the DAS4Whales `dsp` / `detect` hot path (reference:
leabouffaut/DAS4Whales @ 2024_08_07, files cited per function as `dsp.py:LINE` /
`detect.py:LINE`, relative to /root/reference/src/das4whales/).  It exists so that the HIP path
can be checked on the GPU box, where /root/reference is absent.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product
package `das4whales_amd` never does (tests/test_boundary.py enforces that).

Parity pinning: every function here is compared against the *real* reference functions, run in
the build container under oracle/ref_harness.py, through the committed fixtures in
tests/golden/*.npz (generator: tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
The reference's own tests pin only `taper_data` and `snr_tr_array` values
(tests/test_dsp.py:85-88,136-141); those vectors are checked too.

Third-party arithmetic the reference delegates to and that is NOT under /root/reference:
  numpy.fft (pocketfft; reference pins numpy<2, pyproject.toml:31; here 2.2.6),
  scipy.signal / scipy.ndimage (unpinned; authors' env 1.12.0, here 1.15.3): butter, freqz,
  lfilter, hilbert, chirp, find_peaks, gaussian_filter -- their published algorithms are
  restated below where the HIP path re-implements them (filtfilt, sosfiltfilt, chirp, hilbert,
  correlate, fftconvolve-same, find_peaks-prominence); `butter`/`freqz`/`lfilter` are used as
  primitives.
  librosa.stft (unpinned; authors' env 0.10.1; ABSENT here) -- restated in `librosa_stft`
  following librosa >= 0.10 defaults; anything downstream of it is a "restated-oracle" result.
"""
import numpy as np
import scipy.signal as sps
from scipy import ndimage


# --------------------------------------------------------------------------------------------
# axes
# --------------------------------------------------------------------------------------------
def _shifted_axes(trace_shape, selected_channels, dx, fs):
    """(k[:,None], f[None,:]) on the fftshift-ed grid -- dsp.py:129-130, 210-211, 344-345."""
    nx, ns = trace_shape
    f = np.fft.fftshift(np.fft.fftfreq(ns, d=1.0 / fs))
    k = np.fft.fftshift(np.fft.fftfreq(nx, d=selected_channels[2] * dx))
    return k, f


# --------------------------------------------------------------------------------------------
# f-k mask designs (closed forms of the reference's row/column loops)
# --------------------------------------------------------------------------------------------
def fk_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400, cp_min=1450,
                     cp_max=3400, cs_max=3500):
    """Classic speed fan with sine tapers -- dsp.py:85-171.

    Row rule (dsp.py:140-161): rows with |k| < 0.005 are zero; otherwise with s = |f/k|:
    ramp up on [cs_min, cp_min], ramp down on [cp_max, cs_max] (assigned in that order, so at an
    overlap the later assignment wins), then s >= cs_max -> 0 and s < cs_min -> 0.
    Returned dense float64, Fortran order like the reference (dsp.py:137).
    """
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    K = k[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.abs(f[None, :] / K)
        m = np.ones(s.shape)
        up = (s >= cs_min) & (s <= cp_min)
        m = np.where(up, np.sin(0.5 * np.pi * (s - cs_min) / (cp_min - cs_min)), m)
        dn = (s >= cp_max) & (s <= cs_max)
        m = np.where(dn, 1.0 - np.sin(0.5 * np.pi * (s - cp_max) / (cs_max - cp_max)), m)
        m = np.where(s >= cs_max, 0.0, m)
        m = np.where(s < cs_min, 0.0, m)
    m = np.where(np.abs(K) < 0.005, 0.0, m)
    return np.asfortranarray(m)


def _first_index_ge(f, val):
    """np.argmax(f >= val) -- dsp.py:221-222 (returns 0 if no element qualifies)."""
    return int(np.argmax(f >= val))


def hybrid_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450.,
                         fmin=15., fmax=25.):
    """Band-pass (4 Hz sine/cos tapers) x speed high-pass -- dsp.py:174-305 (dense result).

    The reference returns sparse.COO.from_numpy(M) (dsp.py:305); the oracle returns M dense.
    """
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    fp_lo, fp_hi = fmin - 4.0, fmax + 4.0                       # dsp.py:216-219
    H = np.zeros_like(f)
    up = (f >= fp_lo) & (f <= fmin)
    H[up] = np.sin(0.5 * np.pi * (f[up] - fp_lo) / (fmin - fp_lo))            # dsp.py:225-226
    H[(f >= fmin) & (f <= fmax)] = 1.0                                          # dsp.py:228
    dn = (f >= fmax) & (f <= fp_hi)
    H[dn] = np.cos(0.5 * np.pi * (f[dn] - fmax) / (fmax - fp_hi))               # dsp.py:230-231
    i0, i1 = _first_index_ge(f, fp_lo), _first_index_ge(f, fp_hi)               # dsp.py:221-222
    M = np.tile(H, (len(k), 1))                                                 # dsp.py:234
    if i1 > i0:
        fc = f[i0:i1][None, :]
        K = k[:, None]
        ks, kp = fc / cs_min, fc / cp_min                                       # dsp.py:243-244
        col = np.zeros((len(k), i1 - i0))
        with np.errstate(divide="ignore", invalid="ignore"):
            neq = (ks != kp)
            a = neq & (K >= -ks) & (K <= -kp)                                   # dsp.py:249-250
            col = np.where(a, -np.sin(0.5 * np.pi * (K + ks) / (kp - ks)), col)
            b = neq & (-K >= -ks) & (-K <= -kp)                                 # dsp.py:253-254
            col = np.where(b, np.sin(0.5 * np.pi * (K - ks) / (kp - ks)), col)
        col = np.where((K < kp) & (K > -kp), 1.0, col)                          # dsp.py:258
        M[:, i0:i1] *= col                                                      # dsp.py:261
    M = M + np.fliplr(M)                                                        # dsp.py:264
    return M


def hybrid_ninf_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400.,
                              cp_min=1450., cp_max=3400, cs_max=3500, fmin=15., fmax=25.):
    """Butterworth-|H|^2 band-pass x speed band-pass -- dsp.py:308-454 (dense result).

    This is the design every reference script uses (scripts/main_mfdetect.py:46-47).
    """
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    ns = len(f)
    b, a = sps.butter(8, [fmin / (fs / 2), fmax / (fs / 2)], "bp")              # dsp.py:348
    H = np.concatenate((np.zeros(ns // 2),
                        np.abs(sps.freqz(b, a, worN=ns // 2)[1]) ** 2))         # dsp.py:349
    fp_lo, fp_hi = fmin - 14.0, fmax + 14.0                                     # dsp.py:354-357
    i0, i1 = _first_index_ge(f, fp_lo), _first_index_ge(f, fp_hi)               # dsp.py:359-360
    M = np.tile(H, (len(k), 1))                                                 # dsp.py:372
    if i1 > i0:
        fc = f[i0:i1][None, :]
        K = k[:, None]
        ks_min, kp_min = fc / cs_max, fc / cp_max                               # dsp.py:381-382
        ks_max, kp_max = fc / cs_min, fc / cp_min                               # dsp.py:384-385
        col = np.zeros((len(k), i1 - i0))
        with np.errstate(divide="ignore", invalid="ignore"):
            a_ = (ks_min != kp_min) & (K >= ks_min) & (K <= kp_min)             # dsp.py:388-391
            col = np.where(a_, np.sin(0.5 * np.pi * (K - ks_min) / (kp_min - ks_min)), col)
            b_ = (ks_max != kp_max) & (K >= kp_max) & (K <= ks_max)             # dsp.py:392-395
            col = np.where(b_, -np.sin(0.5 * np.pi * (K - ks_max) / (ks_max - kp_max)), col)
        col = np.where((K > kp_min) & (K < kp_max), 1.0, col)                   # dsp.py:399
        M[:, i0:i1] *= col                                                      # dsp.py:402
    M = M + np.fliplr(M)                                                        # dsp.py:405
    M = M + np.flipud(M)                                                        # dsp.py:406
    return M


# --------------------------------------------------------------------------------------------
# Pointwise evaluation of two designs at selected (row, column) indices of the SHIFTED grid -- for the
# known-answer tests at shapes where the dense float64 mask (19 GB at 20 000 x 120 000) is not
# affordable.  tests/test_oracle_pointwise.py pins them bit-exactly against the full designs above.
# --------------------------------------------------------------------------------------------
def fk_filter_design_at(trace_shape, selected_channels, dx, fs, ii, jj, cs_min=1400, cp_min=1450,
                        cp_max=3400, cs_max=3500):
    """fk_filter_design(...)[ii, jj] (dsp.py:140-161), ii / jj integer arrays of equal length."""
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    K, F = k[np.asarray(ii)], f[np.asarray(jj)]
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.abs(F / K)
        m = np.ones(s.shape)
        up = (s >= cs_min) & (s <= cp_min)
        m = np.where(up, np.sin(0.5 * np.pi * (s - cs_min) / (cp_min - cs_min)), m)
        dn = (s >= cp_max) & (s <= cs_max)
        m = np.where(dn, 1.0 - np.sin(0.5 * np.pi * (s - cp_max) / (cs_max - cp_max)), m)
        m = np.where(s >= cs_max, 0.0, m)
        m = np.where(s < cs_min, 0.0, m)
    return np.where(np.abs(K) < 0.005, 0.0, m)


def hybrid_ninf_filter_design_at(trace_shape, selected_channels, dx, fs, ii, jj, cs_min=1400.,
                                 cp_min=1450., cp_max=3400, cs_max=3500, fmin=15., fmax=25.):
    """hybrid_ninf_filter_design(...)[ii, jj] (dsp.py:348-406): with C the matrix before the flips,
    M[i, j] = C[i, j] + C[i, ns-1-j] + C[nx-1-i, j] + C[nx-1-i, ns-1-j] (dsp.py:405-406)."""
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    nx, ns = len(k), len(f)
    b, a = sps.butter(8, [fmin / (fs / 2), fmax / (fs / 2)], "bp")              # dsp.py:348
    H = np.concatenate((np.zeros(ns // 2),
                        np.abs(sps.freqz(b, a, worN=ns // 2)[1]) ** 2))         # dsp.py:349
    i0, i1 = _first_index_ge(f, fmin - 14.0), _first_index_ge(f, fmax + 14.0)   # dsp.py:354-360

    def core(i, j):
        K, fc = k[i], f[j]
        ks_min, kp_min = fc / cs_max, fc / cp_max                               # dsp.py:381-382
        ks_max, kp_max = fc / cs_min, fc / cp_min                               # dsp.py:384-385
        col = np.zeros(K.shape)
        with np.errstate(divide="ignore", invalid="ignore"):
            a_ = (ks_min != kp_min) & (K >= ks_min) & (K <= kp_min)             # dsp.py:388-391
            col = np.where(a_, np.sin(0.5 * np.pi * (K - ks_min) / (kp_min - ks_min)), col)
            b_ = (ks_max != kp_max) & (K >= kp_max) & (K <= ks_max)             # dsp.py:392-395
            col = np.where(b_, -np.sin(0.5 * np.pi * (K - ks_max) / (ks_max - kp_max)), col)
        col = np.where((K > kp_min) & (K < kp_max), 1.0, col)                   # dsp.py:399
        inside = (j >= i0) & (j < i1)
        return np.where(inside, H[j] * col, H[j])                               # dsp.py:372,402

    i, j = np.asarray(ii), np.asarray(jj)
    return (core(i, j) + core(i, ns - 1 - j)) + (core(nx - 1 - i, j) + core(nx - 1 - i, ns - 1 - j))


def folded_gain_at(design_at, trace_shape, kx, kt):
    """M_h(kx, kt) = (M'(kx, kt) + M'(-kx, -kt)) / 2 at UNSHIFTED integer bins: the gain
    the reference's real(ifft2(fft2(x) M')) applies to cos(2 pi (kx c / nx + kt n / ns) + phi).
    design_at(ii, jj) evaluates the mask at shifted-grid indices."""
    nx, ns = trace_shape
    kx, kt = np.asarray(kx), np.asarray(kt)
    sh = lambda a, n: (a + n // 2) % n                                          # unshifted bin -> fftshift-ed index
    return 0.5 * (design_at(sh(kx % nx, nx), sh(kt % ns, ns)) + design_at(sh((-kx) % nx, nx), sh((-kt) % ns, ns)))


def hybrid_gs_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450.,
                            fmin=15., fmax=25.):
    """Box band x box |k| < f/cp_min, Gaussian-blurred (sigma 20) -- dsp.py:457-579."""
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    H = np.zeros_like(f)
    H[(f >= fmin) & (f <= fmax)] = 1.0                                          # dsp.py:508
    i0, i1 = _first_index_ge(f, fmin - 4.0), _first_index_ge(f, fmax + 4.0)     # dsp.py:503-505
    M = np.tile(H, (len(k), 1))                                                 # dsp.py:511
    if i1 > i0:
        kp = f[i0:i1][None, :] / cp_min
        K = k[:, None]
        M[:, i0:i1] *= np.where((K < kp) & (K > -kp), 1.0, 0.0)                 # dsp.py:533-536
    M = M + np.fliplr(M)                                                        # dsp.py:539
    return ndimage.gaussian_filter(M, 20)                                       # dsp.py:540


def hybrid_ninf_gs_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400.,
                                 cp_min=1450., cp_max=3400, cs_max=3500, fmin=15., fmax=25.):
    """Box band x box -f/cp_min < k < -f/cp_max, blur THEN flips -- dsp.py:582-702."""
    k, f = _shifted_axes(trace_shape, selected_channels, dx, fs)
    H = np.zeros_like(f)
    H[(f >= fmin) & (f <= fmax)] = 1.0                                          # dsp.py:633
    i0, i1 = _first_index_ge(f, fmin - 4.0), _first_index_ge(f, fmax + 4.0)     # dsp.py:628-630
    M = np.tile(H, (len(k), 1))
    if i1 > i0:
        fc = f[i0:i1][None, :]
        K = k[:, None]
        M[:, i0:i1] *= np.where((K > -fc / cp_min) & (K < -fc / cp_max), 1.0, 0.0)  # dsp.py:653
    M = ndimage.gaussian_filter(M, 20)                                          # dsp.py:659
    M = M + np.fliplr(M)                                                        # dsp.py:660
    M = M + np.flipud(M)                                                        # dsp.py:661
    return M


# --------------------------------------------------------------------------------------------
# f-k mask application
# --------------------------------------------------------------------------------------------
def tukey_window(n, alpha=0.03):
    """scipy.signal.windows.tukey(n, alpha) restated (symmetric form) -- used at dsp.py:721."""
    if n == 1:
        return np.ones(1)
    if alpha <= 0:
        return np.ones(n)
    if alpha >= 1:
        return np.hanning(n)
    i = np.arange(n, dtype=float)
    width = int(np.floor(alpha * (n - 1) / 2.0))
    w = np.ones(n)
    n1 = i[: width + 1]
    w[: width + 1] = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (n - 1))))
    n3 = i[n - width - 1:]
    w[n - width - 1:] = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (n - 1))))
    return w


def taper_data(trace):
    """trace * tukey(ns, 0.03) along time -- dsp.py:705-722 (returns a new array)."""
    return np.asarray(trace, dtype=float) * tukey_window(trace.shape[1], 0.03)[None, :]


def fk_filter_filt(trace, fk_filter_matrix, tapering=False):
    """real(ifft2(ifftshift(fftshift(fft2(x)) * M))) -- dsp.py:725-756 and :759-786.

    `fk_filter_matrix` is the dense mask on the shifted grid (the COO variant, dsp.py:759-786,
    is arithmetically identical).
    """
    x = np.asarray(trace, dtype=float)
    if tapering:
        x = taper_data(x)                                                       # dsp.py:744-745
    F = np.fft.fftshift(np.fft.fft2(x))                                         # dsp.py:748
    return np.fft.ifft2(np.fft.ifftshift(F * np.asarray(fk_filter_matrix))).real  # dsp.py:751-756


def fold_mask_half(fk_filter_matrix):
    """Hermitian fold M_h = (M'(k,f) + M'(-k,-f))/2 on the UNSHIFTED grid.

    Because x is real, Re(ifft2(F*M')) == ifft2(F*M_h); so the HIP path may always work on the
    half spectrum.  Returns M_h[:, :ns//2+1] (float64).
    """
    Mu = np.fft.ifftshift(np.asarray(fk_filter_matrix, dtype=float))
    nx, ns = Mu.shape
    refl = Mu[(-np.arange(nx)) % nx][:, (-np.arange(ns)) % ns]
    return (0.5 * (Mu + refl))[:, : ns // 2 + 1]


def fk_filter_filt_half(trace, fk_filter_matrix):
    """Same result as fk_filter_filt via rfft2 / folded half mask (even ns) -- checks A.3."""
    x = np.asarray(trace, dtype=float)
    Mh = fold_mask_half(fk_filter_matrix)
    return np.fft.irfft2(np.fft.rfft2(x) * Mh, s=x.shape)


def fk_filter_filt_best_effort(trace32, mask_half32, workers=-1):
    """Best-effort CPU form of the same filter (NOT what the reference runs; bench.py's second CPU
    column): float32, half spectrum with the pre-folded mask, scipy.fft with all cores."""
    import scipy.fft as sfft
    F = sfft.rfft2(trace32, workers=workers)
    F *= mask_half32
    return sfft.irfft2(F, s=trace32.shape, workers=workers)


def compute_cross_correlogram_best_effort(x32, templates, workers=-1):
    """Best-effort CPU matched filter: float32, one batched rfft of the block shared by all templates,
    scipy.fft with all cores (the reference re-transforms the row per template, detect.py:163-164)."""
    import scipy.fft as sfft
    x = (x32 - x32.mean(axis=1, keepdims=True)) / np.max(np.abs(x32), axis=1, keepdims=True)
    n = sfft.next_fast_len(x.shape[1] + max(len(t) for t in templates) - 1, real=True)
    X = sfft.rfft(x, n, axis=1, workers=workers)
    outs = []
    for t in templates:
        T = np.conj(sfft.rfft(np.asarray(t, dtype=np.float32), n))
        outs.append(sfft.irfft(X * T[None, :], n, axis=1, workers=workers)[:, : x.shape[1]])
    return outs


def fk_filt(data, tint, fs, xint, dx, c_min, c_max):
    """Self-designing Gaussian-tapered speed band -- dsp.py:883-953."""
    x = np.asarray(data, dtype=float)
    nx, ns = x.shape
    f = np.fft.fftshift(np.fft.fftfreq(ns, d=tint / fs))                        # dsp.py:923
    k = np.fft.fftshift(np.fft.fftfreq(nx, d=xint * dx))                        # dsp.py:924
    ff, kk = np.meshgrid(f, k)
    g = 1.0 * ((ff < kk * c_min) & (ff < -kk * c_min))                          # dsp.py:930
    g2 = 1.0 * ((ff < kk * c_max) & (ff < -kk * c_max))                         # dsp.py:931
    g = g + np.fliplr(g)                                                        # dsp.py:934
    g = g - (g2 + np.fliplr(g2))                                                # dsp.py:936
    g = ndimage.gaussian_filter(g, 20)                                          # dsp.py:940
    g = (g - g.min()) / (g.max() - g.min())                                     # dsp.py:945
    F = np.fft.fftshift(np.fft.fft2(x)) * g                                     # dsp.py:919,948
    return np.fft.ifft2(np.fft.ifftshift(F)).real                               # dsp.py:950,953


# --------------------------------------------------------------------------------------------
# 1-D zero-phase IIR filters
# --------------------------------------------------------------------------------------------
def odd_ext(x, n):
    """scipy.signal._arraytools.odd_ext along the last axis."""
    left = 2 * x[..., :1] - x[..., n:0:-1]
    right = 2 * x[..., -1:] - x[..., -2:-(n + 2):-1]
    return np.concatenate((left, x, right), axis=-1)


def filtfilt_ba(b, a, x):
    """scipy.signal.filtfilt(b, a, x, axis=-1) with its defaults, restated.

    padtype='odd', padlen = 3*max(len(a), len(b)); forward lfilter started at zi*ext[0],
    backward lfilter on the reversed output started at zi*y[-1]; crop.
    """
    x = np.asarray(x, dtype=float)
    padlen = 3 * max(len(a), len(b))
    if x.shape[-1] <= padlen:
        raise ValueError("The length of the input vector x must be greater than padlen, "
                         "which is %d." % padlen)
    ext = odd_ext(x, padlen)
    zi = sps.lfilter_zi(b, a)
    y, _ = sps.lfilter(b, a, ext, axis=-1, zi=zi * ext[..., :1])
    y, _ = sps.lfilter(b, a, y[..., ::-1], axis=-1, zi=zi * y[..., -1:])
    return y[..., ::-1][..., padlen:-padlen]


def bp_filt(data, fs, fmin, fmax):
    """Butterworth-8 band-pass, ba form, filtfilt along time -- dsp.py:859-880."""
    b, a = sps.butter(8, [fmin / (fs / 2), fmax / (fs / 2)], "bp")              # dsp.py:878
    return filtfilt_ba(b, a, data)                                              # dsp.py:879


def butterworth_filter(filterspec, fs):
    """SOS Butterworth design -- dsp.py:789-827."""
    order, crit, kind = filterspec
    return sps.butter(order, np.array(crit) / (fs / 2), btype=kind, output="sos")  # dsp.py:821-825


def sosfiltfilt(sos, x):
    """scipy.signal.sosfiltfilt(sos, x, axis=-1) restated (user-side call, Example.py:55).

    padlen = 3*(2*n_sections + 1 - min(#(b2==0), #(a2==0))), odd extension, per-section
    steady-state initial conditions (sosfilt_zi) scaled by the first sample, forward/backward.
    """
    sos = np.atleast_2d(np.asarray(sos, dtype=float))
    x = np.asarray(x, dtype=float)
    nsec = sos.shape[0]
    ntaps = 2 * nsec + 1
    ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())
    padlen = 3 * ntaps
    if x.shape[-1] <= padlen:
        raise ValueError("The length of the input vector x must be greater than padlen, "
                         "which is %d." % padlen)
    ext = odd_ext(x, padlen)
    zi = sps.sosfilt_zi(sos)                                   # (nsec, 2)
    zshape = (nsec,) + (1,) * (x.ndim - 1) + (2,)
    zi = zi.reshape(zshape)
    y, _ = sps.sosfilt(sos, ext, axis=-1, zi=zi * ext[..., :1][None])
    y, _ = sps.sosfilt(sos, y[..., ::-1], axis=-1, zi=zi * y[..., -1:][None])
    return y[..., ::-1][..., padlen:-padlen]


# --------------------------------------------------------------------------------------------
# spectral views / metrics
# --------------------------------------------------------------------------------------------
def hilbert(x):
    """scipy.signal.hilbert along the last axis: ifft(fft(x) * h), h = [1,2,..,2,(1),0,..]."""
    x = np.asarray(x, dtype=float)
    n = x.shape[-1]
    h = np.zeros(n)
    if n % 2 == 0:
        h[0] = h[n // 2] = 1
        h[1:n // 2] = 2
    else:
        h[0] = 1
        h[1:(n + 1) // 2] = 2
    return np.fft.ifft(np.fft.fft(x, axis=-1) * h, axis=-1)


def envelope(x):
    """|hilbert(x)| along time (scripts/main_mfdetect.py:58; detect.py:192)."""
    return np.abs(hilbert(x))


def snr_tr_array(trace, env=False):
    """10 log10(x^2 / std(x)^2) (population std), optionally on the envelope -- dsp.py:956-976."""
    x = np.asarray(trace, dtype=float)
    sd = np.std(x, axis=1, keepdims=True)
    with np.errstate(divide="ignore"):
        if env:
            return 10 * np.log10(np.abs(hilbert(x)) ** 2 / sd ** 2)             # dsp.py:975
        return 10 * np.log10(x ** 2 / sd ** 2)                                  # dsp.py:976


def get_fx(trace, nfft):
    """2|fftshift(fft(x, nfft))|/nfft * 1e9 -- dsp.py:18-38."""
    return 2 * np.abs(np.fft.fftshift(np.fft.fft(trace, nfft), axes=1)) / nfft * 1e9


def instant_freq(channel, fs):
    """diff(unwrap(angle(hilbert(x))))/(2 pi) fs -- dsp.py:830-856."""
    return np.diff(np.unwrap(np.angle(hilbert(channel)))) / (2.0 * np.pi) * fs


def librosa_stft(y, n_fft=2048, hop_length=None, **_ignored):
    """Restatement of librosa.stft as the reference calls it (dsp.py:66-68, detect.py:382).

    librosa >= 0.10 defaults: win_length = n_fft, periodic Hann window, center=True with
    pad_mode='constant' (zeros), frame t starts at t*hop in the padded signal,
    n_frames = 1 + len(y)//hop, rows = rfft bins 0..n_fft/2.  
    """
    y = np.asarray(y, dtype=float)
    if hop_length is None:
        hop_length = n_fft // 4
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)     # get_window('hann', fftbins=True)
    yp = np.concatenate((np.zeros(n_fft // 2), y, np.zeros(n_fft // 2)))
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.fft.rfft(yp[idx] * win[:, None], axis=0)


def get_spectrogram(waveform, fs, nfft=128, overlap_pct=0.8):
    """dB spectrogram normalised by its max -- dsp.py:41-78."""
    hop = int(np.floor(nfft * (1 - overlap_pct)))                               # dsp.py:68
    S = np.abs(librosa_stft(waveform, n_fft=nfft, hop_length=hop))              # dsp.py:66
    tt = np.linspace(0, len(waveform) / fs, num=S.shape[1])                     # dsp.py:74
    ff = np.linspace(0, fs / 2, num=S.shape[0])                                 # dsp.py:75
    with np.errstate(divide="ignore"):
        p = 20 * np.log10(S / np.max(S))                                        # dsp.py:76
    return p, tt, ff


# --------------------------------------------------------------------------------------------
# matched filter
# --------------------------------------------------------------------------------------------
def chirp_linear(t, f0, f1, t1):
    """scipy.signal.chirp(..., method='linear'): cos(2 pi (f0 t + (f1-f0)/(2 t1) t^2))."""
    return np.cos(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / t1 * t * t))


def chirp_hyperbolic(t, f0, f1, t1):
    """scipy.signal.chirp(..., method='hyperbolic') restated (f0 != f1):

    sing = -f1 t1/(f0 - f1); phase = 2 pi (-sing f0) log|1 - t/sing|.
    """
    if f0 == f1:
        return np.cos(2 * np.pi * f0 * t)
    sing = -f1 * t1 / (f0 - f1)
    return np.cos(2 * np.pi * (-sing * f0) * np.log(np.abs(1 - t / sing)))


def gen_linear_chirp(fmin, fmax, duration, sampling_rate):
    """Down-sweep fmax -> fmin, linear -- detect.py:20-41."""
    t = np.arange(0, duration, 1 / sampling_rate)
    return chirp_linear(t, fmax, fmin, duration)


def gen_hyperbolic_chirp(fmin, fmax, duration, sampling_rate):
    """Down-sweep fmax -> fmin, hyperbolic -- detect.py:44-65."""
    t = np.arange(0, duration, 1 / sampling_rate)
    return chirp_hyperbolic(t, fmax, fmin, duration)


def gen_template_fincall(time, fs, fmin=15., fmax=25., duration=1., window=True):
    """Hann-windowed hyperbolic chirp zero-padded to len(time) -- detect.py:68-93."""
    c = gen_hyperbolic_chirp(fmin, fmax, duration, fs)
    tpl = np.zeros(np.shape(time))
    tpl[:len(c)] = c * np.hanning(len(c)) if window else c                      # detect.py:87-90
    return tpl


def shift_xcorr(x, y):
    """Positive-lag cross-correlation c[k] = sum_n x[n+k] y[n], k = 0..N-1 -- detect.py:96-112.

    (scipy.signal.correlate(x, y, 'full', 'fft')[len(x)-1:]; computed here with zero-padded FFTs.)
    """
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    n = len(x) + len(y) - 1
    nfft = 1 << int(np.ceil(np.log2(n)))
    c = np.fft.irfft(np.fft.rfft(x, nfft) * np.conj(np.fft.rfft(y, nfft)), nfft)
    return c[: len(x)]


def shift_nxcorr(x, y):
    """shift_xcorr / (std(x) std(y) len(x)) -- detect.py:115-137."""
    return shift_xcorr(x, y) / (np.std(x) * np.std(y) * len(x))


def compute_cross_correlogram(data, template):
    """Peak-normalised matched filter, row by row -- detect.py:140-166.

    rows: (x - mean)/max|x| (max of the UN-de-meaned row, detect.py:157); template:
    (y - mean)/max|y| over its zero-padded length (detect.py:158).  Output floating point
    (the reference's np.empty_like(data) would truncate for integer input -- not replicated).
    """
    x = np.asarray(data, dtype=float)
    xn = (x - x.mean(axis=1, keepdims=True)) / np.max(np.abs(x), axis=1, keepdims=True)
    t = np.asarray(template, dtype=float)
    t = (t - t.mean()) / np.max(np.abs(t))
    n = x.shape[1] + len(t) - 1
    nfft = 1 << int(np.ceil(np.log2(n)))
    T = np.conj(np.fft.rfft(t, nfft))
    c = np.fft.irfft(np.fft.rfft(xn, nfft, axis=1) * T[None, :], nfft, axis=1)
    return c[:, : x.shape[1]]


def compute_cross_correlogram_reference_form(data, template):
    """detect.compute_cross_correlogram in the reference's OWN form, statement by statement (detect.py:156-166 and
    shift_xcorr, detect.py:111-112): a Python loop over the rows, scipy.signal.correlate(row, template, 'full', 'fft') with
    the full-length zero-padded template, lags len(x) - 1 onwards.  compute_cross_correlogram above is the batched
    restatement the parity tests use (same numbers to 1e-12, tests/test_oracle_golden.py); this one exists so that the
    CPU baseline can time what a user of the reference actually runs (bench.py cpu_baseline.reference_form)."""
    import scipy.signal as sp
    data = np.asarray(data, dtype=float)
    norm_data = (data - np.mean(data, axis=1, keepdims=True)) / np.max(np.abs(data), axis=1, keepdims=True)   # detect.py:157
    template = np.asarray(template, dtype=float)
    template = (template - np.mean(template)) / np.max(np.abs(template))                                      # detect.py:158
    cross_correlogram = np.empty_like(data)                                                                   # detect.py:161
    for i in range(data.shape[0]):                                                                            # detect.py:163
        corr = sp.correlate(norm_data[i, :], template, mode="full", method="fft")                             # detect.py:111
        cross_correlogram[i, :] = corr[len(norm_data[i, :]) - 1:]                                             # detect.py:112,164
    return cross_correlogram


# --------------------------------------------------------------------------------------------
# peak picking
# --------------------------------------------------------------------------------------------
def find_peaks_prominence(x, threshold):
    """scipy.signal.find_peaks(x, prominence=threshold)[0] restated.

    Strict local maxima (plateaus -> middle sample, floor); prominence with wlen=None: walk left
    and right until a sample higher than the peak (or the array end), take the minimum on each
    side, prominence = peak - max(left_min, right_min); keep prominence >= threshold.
    Pure-Python loops: small cases only.
    """
    x = np.asarray(x, dtype=float)
    n = len(x)
    peaks = []
    i = 1
    while i < n - 1:
        if x[i - 1] < x[i]:
            j = i + 1
            while j < n - 1 and x[j] == x[i]:
                j += 1
            if x[j] < x[i]:
                peaks.append((i + j - 1) // 2)
                i = j
                continue
        i += 1
    out = []
    for p in peaks:
        lo = x[p]
        q = p
        while q >= 0 and x[q] <= x[p]:
            lo = min(lo, x[q])
            q -= 1
        ro = x[p]
        q = p
        while q < n and x[q] <= x[p]:
            ro = min(ro, x[q])
            q += 1
        if x[p] - max(lo, ro) >= threshold:
            out.append(p)
    return np.asarray(out, dtype=np.int64)


def pick_times_env(corr_m, threshold):
    """Per row find_peaks(|hilbert(c)|, prominence=thr) -- detect.py:169-195 (row order)."""
    return [sps.find_peaks(np.abs(hilbert(c)), prominence=threshold)[0] for c in np.asarray(corr_m, dtype=float)]


def pick_times(corr_m, threshold):
    """Per row find_peaks(c, prominence=thr) -- detect.py:249-274."""
    return [sps.find_peaks(c, prominence=threshold)[0] for c in np.asarray(corr_m, dtype=float)]


def convert_pick_times(peaks_indexes_m):
    """Ragged list -> 2 x K array, row 0 = channel index, row 1 = time index -- detect.py:277-303."""
    ch = [np.full(len(p), i, dtype=np.int64) for i, p in enumerate(peaks_indexes_m)]
    if not ch:
        return np.zeros((2, 0))
    return np.asarray((np.concatenate(ch), np.concatenate([np.asarray(p) for p in peaks_indexes_m])))


def select_picked_times(idx_tp, tstart, tend, fs):
    """Keep picks with tstart*fs <= time index <= tend*fs -- detect.py:306-330."""
    keep = (idx_tp[1] >= tstart * fs) & (idx_tp[1] <= tend * fs)
    return (idx_tp[0][keep], idx_tp[1][keep])


# --------------------------------------------------------------------------------------------
# spectrogram correlation
# --------------------------------------------------------------------------------------------
def get_sliced_nspectrogram(trace, fs, fmin, fmax, nperseg, nhop):
    """|STFT| / max, rows with fmin <= f <= fmax -- detect.py:334-408."""
    S = np.abs(librosa_stft(trace, n_fft=nperseg, hop_length=nhop))             # detect.py:382
    nf, nt = S.shape
    tt = np.linspace(0, len(trace) / fs, num=nt)                                # detect.py:385
    ff = np.linspace(0, fs / 2, num=nf)                                         # detect.py:386
    p = S / np.max(S)                                                           # detect.py:387
    keep = np.where((ff >= fmin) & (ff <= fmax))                                # detect.py:390
    return p[keep], ff[keep], tt


def buildkernel(f0, f1, bdwdth, dur, f, t, samp, fmin, fmax):
    """Hat-function kernel along a hyperbolic sweep, Hann-weighted in time -- detect.py:411-492."""
    nt = np.size(np.nonzero((t < dur * 8) & (t > dur * 7)))                     # detect.py:456
    tvec = np.linspace(0, dur, nt)
    fcurve = f0 * f1 * dur / ((f0 - f1) * tvec + f1 * dur)                      # detect.py:470
    x = np.asarray(f)[:, None] - fcurve[None, :]
    K = (1 - x ** 2 / bdwdth ** 2) * np.exp(-x ** 2 / (2 * bdwdth ** 2))        # detect.py:471
    return tvec, np.asarray(f), K * np.hanning(nt)[None, :]                     # detect.py:474


def xcorr2d(spectro, kernel):
    """Sum over f of per-row 'same' correlation with the kernel, clipped, / (median * n_t) -- detect.py:579-602."""
    S = np.asarray(spectro, dtype=float)
    K = np.asarray(kernel, dtype=float)
    nt, nk = S.shape[1], K.shape[1]
    full = np.zeros(nt + nk - 1)
    for r in range(S.shape[0]):
        full += np.convolve(S[r], K[r, ::-1])                                   # detect.py:597-598
    start = (nk - 1) // 2                                                       # centred 'same' crop
    c = full[start:start + nt]
    c[c < 0] = 0                                                                # detect.py:599
    return c / (np.median(S) * nk)                                              # detect.py:600


def nxcorr2d(spectro, kernel):
    """max over the frequency lag of the 2-D 'same' correlation / (std(S) std(K) n_t) -- detect.py:544-576.

    scipy.signal.correlate(S, K, 'same')[a, t] = sum_{fk, j} S[a + fk - nfk//2, t + j - nk//2] K[fk, j].
    """
    S = np.asarray(spectro, dtype=float)
    K = np.asarray(kernel, dtype=float)
    nf, nt = S.shape
    nfk, nk = K.shape
    Sp = np.zeros((nf + nfk, nt + nk))
    Sp[nfk // 2:nfk // 2 + nf, nk // 2:nk // 2 + nt] = S
    corr = np.zeros((nf, nt))
    for fk in range(nfk):
        for j in range(nk):
            corr += Sp[fk:fk + nf, j:j + nt] * K[fk, j]
    corr /= np.std(S) * np.std(K) * nt                                          # detect.py:573
    return np.max(corr, axis=0)


def xcorr(t, f, Sxx, tvec, fvec, BlueKernel):
    """Valid-lag correlation of the kernel with the spectrogram -- detect.py:605-647."""
    nk, nfk = np.size(tvec), np.size(fvec)
    S = np.asarray(Sxx, dtype=float)
    K = np.asarray(BlueKernel, dtype=float)
    n = np.size(t) - (nk - 1)
    c = np.array([np.sum(K * S[:nfk, i:i + nk]) for i in range(n)])             # detect.py:634-638
    c /= np.median(S) * nk
    c[0] = 0
    c[-1] = 0
    c[c < 0] = 0
    return [np.asarray(t)[int(nk / 2) - 1:-int(np.ceil(nk / 2))], c]


def buildkernel_from_template(fmin, fmax, dur, fs, nperseg, nhop):
    """Sliced normalised spectrogram of the windowed hyperbolic chirp -- detect.py:495-541."""
    tpl = gen_hyperbolic_chirp(fmin, fmax, dur, fs)
    tpl = tpl * np.hanning(len(tpl))
    return get_sliced_nspectrogram(tpl, fs, fmin, fmax, nperseg, nhop)[0]


def spectrocorr_params(fs, flims, kernel, win_size, overlap_pct):
    """Parameter derivation of compute_cross_correlogram_spectrocorr -- detect.py:680-696."""
    nperseg = int(win_size * fs)
    nhop = int(np.floor(nperseg * (1 - overlap_pct)))
    fmin, fmax = flims
    f1, f0, dur, bw = kernel["f1"], kernel["f0"], kernel["dur"], kernel["bdwidth"]
    if fmax - f1 < 2 * bw:
        fmax = f1 + 3 * bw
    if f0 - fmin < 2 * bw:
        fmin = f0 - 3 * bw
    return nperseg, nhop, fmin, fmax, f0, f1, dur, bw


def compute_cross_correlogram_spectrocorr(data, fs, flims, kernel, win_size, overlap_pct):
    """Per-channel spectrogram x hat-kernel correlation -- detect.py:650-709."""
    data = np.asarray(data, dtype=float)
    nperseg, nhop, fmin, fmax, f0, f1, dur, bw = spectrocorr_params(fs, flims, kernel, win_size, overlap_pct)
    _, ff, tt = get_sliced_nspectrogram(data[0], fs, fmin, fmax, nperseg, nhop)   # detect.py:699
    _, _, ker = buildkernel(f0, f1, bw, dur, ff, tt, fs, fmin, fmax)              # detect.py:702
    out = np.empty((data.shape[0], len(tt)))
    for i in range(data.shape[0]):                                                # detect.py:705-707
        S, _, _ = get_sliced_nspectrogram(data[i], fs, fmin, fmax, nperseg, nhop)
        out[i] = xcorr2d(S, ker)
    return out


# --------------------------------------------------------------------------------------------
# deterministic synthetic strain block (SURVEY.md 8d "S-small" recipe, scaled by arguments)
# --------------------------------------------------------------------------------------------
# =============================================================================================
# Image pipeline of the Gabor detector: improcess.py + scripts/main_gabordetect.py
#
# Third-party arithmetic that is ABSENT here (authors' env: opencv-python 4.9.0.80, torchvision 0.17.2,
# DAS4Whales_ExampleNotebook.md:67-69) and therefore restated from the published algorithms --
# "parity unpinned" for these two:
#   cv2.getGaborKernel (OpenCV 4.9 modules/imgproc/src/gabor.cpp), cv2.filter2D (correlation, anchor
#   at the kernel centre, BORDER_REFLECT_101);
#   torchvision.transforms.Resize on a tensor = torch.nn.functional.interpolate(mode="bilinear",
#   align_corners=False, antialias=True) (torchvision 0.17 default antialias=True; non-float input is
#   cast to float32, interpolated and cast back).  torch IS present here, so `resize_bilinear_aa`
#   below (the NumPy restatement of aten's _upsample_bilinear2d_aa weights) is pinned against
#   torch's own CPU kernel in tests/test_oracle_image.py.
# =============================================================================================
def scale_pixels(img):
    """improcess.py:23-40."""
    return (img - img.min()) / (img.max() - img.min())


def trace2image(trace):
    """improcess.py:43-62: |hilbert| / std per row, min-max scaled to [0, 255]."""
    trace = np.asarray(trace, dtype=np.float64)
    image = np.abs(hilbert(trace)) / np.std(trace, axis=1, keepdims=True)
    return scale_pixels(image) * 255


def angle_fromspeed(c0, fs, dx, selected_channels):
    """improcess.py:65-96 (degrees)."""
    return np.arctan(c0 / (fs * dx * selected_channels[2])) * 180 / np.pi


def get_gabor_kernel(ksize, sigma, theta, lambd, gamma, psi=np.pi * 0.5):
    """cv2.getGaborKernel(ksize=(w, h), ..., ktype=CV_64F), OpenCV 4.9 gabor.cpp: the kernel has
    2*(w//2)+1 columns and 2*(h//2)+1 rows and is written mirrored (kernel[ymax-y][xmax-x])."""
    sigma_x, sigma_y = sigma, sigma / gamma
    xmax, ymax = ksize[0] // 2, ksize[1] // 2
    c, s = np.cos(theta), np.sin(theta)
    y, x = np.mgrid[-ymax:ymax + 1, -xmax:xmax + 1].astype(np.float64)
    xr = x * c + y * s
    yr = -x * s + y * c
    v = np.exp(-0.5 / sigma_x ** 2 * xr * xr - 0.5 / sigma_y ** 2 * yr * yr) * np.cos(2 * np.pi / lambd * xr + psi)
    return v[::-1, ::-1].copy()


def gabor_filt_design(theta_c0):
    """improcess.py:99-140: (up, down) 101 x 101 kernels."""
    up = get_gabor_kernel((100, 100), 4, np.pi / 2 + np.deg2rad(theta_c0), 20, 0.15, 0)
    return up, np.flipud(up)


def filter2d(img, kernel):
    """cv2.filter2D(img, cv2.CV_64F, kernel) (scripts/main_gabordetect.py:109,132): correlation,
    anchor = kernel centre, border reflect-101 (= scipy 'mirror')."""
    return ndimage.correlate(np.asarray(img, dtype=np.float64), np.asarray(kernel, dtype=np.float64), mode="mirror")


def _aa_weights(in_size, out_size):
    """Rows of (first input index, weights) of aten's antialiased bilinear resize along one axis
    (ATen/native/cpu/UpSampleKernel.cpp, _compute_indices_weights_aa with the triangle filter)."""
    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    rows = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((j + xmin - center + 0.5) * invscale)) for j in range(xsize)])
        rows.append((xmin, w / w.sum()))
    return rows


def resize_bilinear_aa(img, out_h, out_w):
    """torch.nn.functional.interpolate(img[None, None], (out_h, out_w), mode='bilinear',
    align_corners=False, antialias=True)[0, 0]: horizontal pass, then vertical pass."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    tmp = np.empty((h, out_w))
    for j, (x0, wt) in enumerate(_aa_weights(w, out_w)):
        tmp[:, j] = img[:, x0:x0 + len(wt)] @ wt
    out = np.empty((out_h, out_w))
    for i, (y0, wt) in enumerate(_aa_weights(h, out_h)):
        out[i] = wt @ tmp[y0:y0 + len(wt)]
    return out


def binning(image, ft, fx):
    """improcess.py:395-420: transforms.Resize((int(H*fx), int(W*ft))) of ToTensor()(image).  A bool
    image (the mask, scripts/main_gabordetect.py:163) is interpolated as float32 0/1 and cast back
    to bool, i.e. True wherever any True pixel has non-zero weight."""
    image = np.asarray(image)
    oh, ow = int(image.shape[0] * fx), int(image.shape[1] * ft)
    if image.dtype == bool:
        return resize_bilinear_aa(image.astype(np.float64), oh, ow) != 0
    return resize_bilinear_aa(image, oh, ow)


def apply_smooth_mask(array, mask, sigma=1.5):
    """improcess.py:423-454: the Gaussian-smoothed mask is computed but NOT used; the product is with
    the raw mask (:452)."""
    return array * mask


def gabor_mask_pipeline(trf_fk, fs, dx, selected_channels, c0=1500., threshold=9100., threshold2=150.):
    """scripts/main_gabordetect.py:78-166 without the plots."""
    image = trace2image(trf_fk)
    theta_c0 = angle_fromspeed(c0, fs, dx, selected_channels)
    imagebin = binning(image, 1 / 10, 1 / 10)
    up, down = gabor_filt_design(theta_c0)
    fimage = filter2d(imagebin, up) + filter2d(imagebin, down)
    binary = fimage > threshold
    score = filter2d(binary.astype(float), up) + filter2d(binary.astype(float), down)
    mask = score > threshold2
    mask_sparse = binning(mask, 10, 10)
    masked_tr = apply_smooth_mask(np.asarray(trf_fk, dtype=np.float64), mask_sparse)
    return {"image": image, "imagebin": imagebin, "fimage": fimage, "binary": binary, "score": score,
            "mask": mask, "mask_sparse": mask_sparse, "masked_tr": masked_tr}


def synth_block(nx, ns, fs=200.0, dx=2.0419046878814697, step=4, seed=1234, n_calls=6,
                n_waves=40):
    """White noise + slow 'ocean-wave' plane waves + fin-whale notes on hyperbolic moveouts."""
    rng = np.random.default_rng(seed)
    t = np.arange(ns) / fs
    xpos = np.arange(nx) * step * dx
    d = 1e-9 * rng.standard_normal((nx, ns))
    for _ in range(n_waves):
        f = rng.uniform(0.5, 8.0)
        c = rng.uniform(5.0, 300.0) * rng.choice([-1.0, 1.0])
        ph = rng.uniform(0, 2 * np.pi)
        d += 1e-8 / np.sqrt(n_waves) * np.cos(2 * np.pi * f * (t[None, :] - xpos[:, None] / c) + ph)
    hf = gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    lf = gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
    for i in range(n_calls):
        tpl = hf if i % 2 == 0 else lf
        L = int(np.max(np.nonzero(tpl)[0])) + 1
        x0 = rng.uniform(xpos[0], xpos[-1])
        r = rng.uniform(1000.0, 5000.0)
        t0 = rng.uniform(0.05, 0.7) * ns / fs
        arr = t0 + np.sqrt(r * r + (xpos - x0) ** 2) / 1500.0
        idx = np.round(arr * fs).astype(int)
        for c_i in range(nx):
            a = idx[c_i]
            if 0 <= a < ns - L:
                d[c_i, a:a + L] += 5e-9 * tpl[:L]
    return d
