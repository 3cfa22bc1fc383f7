"""Closed-form answers of the f-k filter that are cheap at any size (test infrastructure).

On-grid plane waves: for integer bins (kx, kt), 0 <= kt <= ns/2,
    x[c, n] = sum_i a_i cos(2 pi (kx_i c / nx + kt_i n / ns) + phi_i)
the reference's real(ifft2(fft2(x) * M')) (dsp.py:748-756) returns
    y[c, n] = sum_i a_i M_h(kx_i, kt_i) cos(...),   M_h = (M'(k, f) + M'(-k, -f)) / 2   
Unit impulse at (c0, n0): y[c, n] = Re ifft2(M')[c - c0, n - n0]."""
import numpy as np


def pick_plane_waves(nx, ns, sel, dx, fs, rng, n_waves=36):
    """Bins spread over the pass band, the tapers, the stop band and the special rows / columns of the
    speed-fan designs: returns integer arrays (kx, kt), amplitudes and phases."""
    dk = 1.0 / (nx * sel[2] * dx)
    df = fs / ns
    kx, kt = [], []

    def add_speed(speed, f_hz, sign):
        j = int(round(f_hz / df))
        i = int(round(f_hz / speed / dk))
        if 0 < i < nx // 2 and 0 < j < ns // 2:
            kx.append(i if sign > 0 else nx - i)
            kt.append(j)
    for speed in (1500.0, 2000.0, 2600.0, 3200.0):                 # inside the fan, inside 14-30 Hz
        for f_hz in (16.0, 21.5, 27.0):
            add_speed(speed, f_hz, +1 if len(kx) % 2 else -1)
    for speed, f_hz in ((1425.0, 20.0), (3450.0, 24.0), (1380.0, 18.0), (3350.0, 22.0), (1440.0, 40.0),
                        (2500.0, 8.0), (2500.0, 60.0), (900.0, 20.0), (6000.0, 20.0), (300.0, 3.0)):
        add_speed(speed, f_hz, +1 if len(kx) % 2 else -1)         # tapers, stop band, out-of-band columns
    kx += [0, nx // 2, 0, 1, nx - 1, nx // 3]                      # DC / Nyquist rows and columns
    kt += [0, ns // 2, int(round(20.0 / df)), 0, ns // 2, 1]
    while len(kx) < n_waves:                                       # anywhere
        kx.append(int(rng.integers(0, nx)))
        kt.append(int(rng.integers(0, ns // 2 + 1)))
    kx, kt = np.array(kx), np.array(kt)
    return kx, kt, rng.uniform(0.5, 2.0, len(kx)), rng.uniform(0, 2 * np.pi, len(kx))


def wave_factors(nx, ns, kx, kt, ph):
    """[nx, 2K] and [2K, ns] float64 factors with x = A @ diag(a, a) @ B, from
    cos(alpha_c + beta_n) = cos alpha cos beta - sin alpha sin beta (phases reduced in integers)."""
    c, n = np.arange(nx)[:, None], np.arange(ns)[None, :]
    alpha = 2 * np.pi * ((kx[None, :] * c) % nx) / nx + ph[None, :]
    beta = 2 * np.pi * ((kt[:, None] * n) % ns) / ns
    A = np.concatenate((np.cos(alpha), -np.sin(alpha)), axis=1)
    B = np.concatenate((np.cos(beta), np.sin(beta)), axis=0)
    return A, B


def impulse_response_rows(mask_rows_fn, nx, ns, rows):
    """Rows `rows` of Re ifft2(M') for a real mask M' on the UNSHIFTED grid, in float64:
    h[c, :] = Re ifft_n( (1/nx) sum_k M'[k, :] exp(2 pi i k c / nx) ).  mask_rows_fn(k0, k1) returns the
    float64 rows k0:k1 of M' (so that the caller can stream a mask that does not fit in float64)."""
    rows = np.asarray(rows)
    acc = np.zeros((len(rows), ns), dtype=np.complex128)
    step = 512
    for k0 in range(0, nx, step):
        k1 = min(nx, k0 + step)
        ph = 2 * np.pi * ((np.arange(k0, k1)[None, :] * rows[:, None]) % nx) / nx
        acc += np.exp(1j * ph) @ mask_rows_fn(k0, k1)
    return np.fft.ifft(acc / nx, axis=1).real


def assert_picks_match(got, ref_signal, thr, what=""):
    """the pick index set must equal scipy.signal.find_peaks(ref_signal, prominence=thr)[0]
    except for peaks whose float64 prominence lies within 1e-4 * thr of the threshold.  Every differing index is
    inspected: it must be a local maximum of the reference signal and marginal in that sense."""
    import scipy.signal as sps
    ref_signal = np.asarray(ref_signal, dtype=np.float64)
    ref = sps.find_peaks(ref_signal, prominence=thr)[0]
    gs, rs = set(int(i) for i in got), set(int(i) for i in ref)
    diff = sorted(gs ^ rs)
    top = float(np.max(np.abs(ref_signal)))
    for i in diff:
        assert 0 < i < len(ref_signal) - 1, (what, i, "differing pick at the row edge")
        # a flat top: two neighbouring samples of the reference within float32 resolution of each other, the float32 signal
        # has its maximum on the other one -- the pick moves by one sample (one index only in `got`, its neighbour only in the
        # reference picks); both are inspected when their turn comes and both are accepted here
        swapped = any(0 <= j < len(ref_signal) and ((i in gs) != (j in gs)) and ((i in rs) != (j in rs)) and (j in gs or j in rs)
                      and abs(ref_signal[i] - ref_signal[j]) <= 2e-6 * top for j in (i - 1, i + 1))
        if swapped:
            continue
        assert ref_signal[i] >= ref_signal[i - 1] and ref_signal[i] >= ref_signal[i + 1], \
            (what, i, "differing pick is not a local maximum of the reference", ref_signal[i - 2:i + 3].tolist(), top,
             [j in rs for j in range(i - 2, i + 3)], [j in gs for j in range(i - 2, i + 3)])
        prom = float(sps.peak_prominences(ref_signal, [i])[0][0])
        assert abs(prom - thr) <= 1e-4 * thr, (what, i, "differing pick is not marginal: prominence %.9g vs threshold %.9g" % (prom, thr))
    return len(diff), len(ref)


def synth_block_device(nx, ns, device, fs=200.0, dx=2.0419046878814697, step=1, seed=1234, n_calls=6, n_waves=40,
                       noise=1e-9, ocean_amp=1e-8, call_amp=5e-9):
    """SURVEY 8(d) S-large recipe on the device in float32: white noise `noise` + `n_waves` slow "ocean-wave" plane waves
    (f ~ U(0.5, 8) Hz, apparent speed ~ U(5, 300) m/s, amplitude `ocean_amp` EACH) + `n_calls` fin-whale notes (HF / LF
    alternated, hyperbolic moveout at 1500 m/s, amplitude `call_amp`).  Same recipe as oracle.synth_block (which divides the
    ocean amplitude by sqrt(n_waves)); returns the block and one dict per note: template index, source position / range /
    emission time, and `flank` = [(channel, arrival sample)] on the flanks of its moveout."""
    import torch
    from oracle import d4w_oracle as orc
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((nx, ns), dtype=torch.float32, device=device, generator=g) * noise
    t = np.arange(ns) / fs
    xpos = np.arange(nx) * step * dx
    f = rng.uniform(0.5, 8.0, n_waves)
    c = rng.uniform(5.0, 300.0, n_waves) * rng.choice([-1.0, 1.0], n_waves)
    ph = rng.uniform(0, 2 * np.pi, n_waves)
    alpha = -2 * np.pi * (f / c)[None, :] * xpos[:, None] + ph[None, :]            # cos(alpha_c + beta_n)
    beta = 2 * np.pi * f[:, None] * t[None, :]
    A = torch.from_numpy(np.concatenate((np.cos(alpha), -np.sin(alpha)), axis=1).astype(np.float32)).to(device)
    B = torch.from_numpy(np.concatenate((np.cos(beta), np.sin(beta)), axis=0).astype(np.float32)).to(device)
    rows = 2000
    for r0 in range(0, nx, rows):
        x[r0:r0 + rows] += ocean_amp * (A[r0:r0 + rows] @ B)
    del A, B
    tpls = [orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68), orc.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)]
    calls = []
    flat = x.view(-1)
    for i in range(n_calls):
        tpl = tpls[i % 2]
        L = int(np.max(np.nonzero(tpl)[0])) + 1
        x0 = rng.uniform(xpos[0], xpos[-1])
        r = rng.uniform(1000.0, 5000.0)
        t0 = rng.uniform(0.05, 0.7) * ns / fs
        idx = np.round((t0 + np.sqrt(r * r + (xpos - x0) ** 2) / 1500.0) * fs).astype(np.int64)
        ok = np.nonzero((idx >= 0) & (idx < ns - L))[0]
        base = torch.from_numpy(ok * ns + idx[ok]).to(device)
        tp = torch.from_numpy((call_amp * tpl[:L]).astype(np.float32)).to(device)
        pos = (base[:, None] + torch.arange(L, device=device)[None, :]).reshape(-1)
        flat.index_add_(0, pos, tp.repeat(len(ok)))
        # where to look for the note: a channel on the FLANK of the moveout (offset ~ range: apparent speed ~ 2100 m/s).  At the
        # apex the apparent speed is infinite and hybrid_ninf_filter_design removes the arrival by design (dsp.py:372-395).
        cands = [int(np.argmin(np.abs(xpos - (x0 + sg * r)))) for sg in (-1.0, 1.0)]
        cands = [c for c in cands if abs(abs(xpos[c] - x0) - r) < 0.25 * r and 0 <= idx[c] < ns - 2 * L]
        calls.append({"template": i % 2, "x0": float(x0), "range": float(r), "t0": float(t0),
                      "flank": [(c, int(idx[c])) for c in cands]})
    return x, calls


# ------------------------------------------------------------------------------------------
# OpenCV pieces of the Gabor image pipeline (f3), pinned WITHOUT the restatement on both sides.
# cv2 is absent here, so the pins are the library's documented definitions written out as plain
# loops (reference call sites: improcess.py:116-123 cv2.getGaborKernel, scripts/main_gabordetect.py:109,132
# cv2.filter2D), independent of oracle/d4w_oracle.py (NumPy closed form / scipy.ndimage) and of the HIP kernels:
#   * cv::borderInterpolate, BORDER_REFLECT_101 ("gfedcb|abcdefgh|gfedcba"):  i < 0 -> -i,  i >= n -> 2 n - 2 - i
#   * cv::filter2D: dst(x, y) = sum_{x', y'} kernel(x', y') src(x + x' - anchor.x, y + y' - anchor.y), a CORRELATION,
#     anchor = kernel centre (ksize / 2), borderType = BORDER_DEFAULT = BORDER_REFLECT_101
#   * cv::getGaborKernel (imgproc/src/gabor.cpp): xmax = ksize.width / 2, ymax = ksize.height / 2; for y in -ymax..ymax,
#     x in -xmax..xmax: xr = x cos t + y sin t, yr = -x sin t + y cos t,
#     v = exp(-xr^2 / (2 sigma^2) - yr^2 / (2 (sigma / gamma)^2)) cos(2 pi xr / lambd + psi), stored at
#     kernel[ymax - y][xmax - x]
# ------------------------------------------------------------------------------------------
def reflect101(i, n):
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * n - 2 - i
    return i


def filter2d_loops(img, ker):
    """cv::filter2D by definition (quadruple loop; small images only)."""
    img = np.asarray(img, dtype=np.float64)
    ker = np.asarray(ker, dtype=np.float64)
    h, w = img.shape
    kh, kw = ker.shape
    ay, ax = kh // 2, kw // 2
    out = np.zeros((h, w))
    for y in range(h):
        for x in range(w):
            acc = 0.0
            for j in range(kh):
                yy = reflect101(y + j - ay, h)
                for i in range(kw):
                    acc += ker[j, i] * img[yy, reflect101(x + i - ax, w)]
            out[y, x] = acc
    return out


def gabor_kernel_loops(ksize, sigma, theta, lambd, gamma, psi):
    """cv::getGaborKernel by definition (double loop, math module only)."""
    import math
    xmax, ymax = ksize[0] // 2, ksize[1] // 2
    sx, sy = sigma, sigma / gamma
    c, s = math.cos(theta), math.sin(theta)
    k = np.zeros((2 * ymax + 1, 2 * xmax + 1))
    for y in range(-ymax, ymax + 1):
        for x in range(-xmax, xmax + 1):
            xr = x * c + y * s
            yr = -x * s + y * c
            k[ymax - y, xmax - x] = math.exp(-0.5 * xr * xr / (sx * sx) - 0.5 * yr * yr / (sy * sy)) * math.cos(2 * math.pi / lambd * xr + psi)
    return k


def check_filter2d(filter2d, tol):
    """Known answers any cv2.filter2D stand-in must give; filter2d(img, kernel) -> ndarray."""
    rng = np.random.default_rng(42)
    # 1. impulse image -> the kernel flipped about its anchor (a correlation, not a convolution), also for an EVEN size
    for kh, kw in ((5, 3), (4, 6)):
        ker = rng.standard_normal((kh, kw))
        img = np.zeros((21, 19))
        img[10, 9] = 1.0
        out = np.asarray(filter2d(img, ker), dtype=np.float64)
        ay, ax = kh // 2, kw // 2
        want = np.zeros_like(img)
        for j in range(kh):
            for i in range(kw):
                want[10 - (j - ay), 9 - (i - ax)] = ker[j, i]
        assert np.max(np.abs(out - want)) <= tol * np.max(np.abs(ker)), ("impulse", kh, kw)
    # 2. constant image -> sum of the kernel everywhere, borders included (reflect-101 of a constant is the constant)
    ker = rng.standard_normal((7, 9))
    out = np.asarray(filter2d(np.full((12, 15), 2.5), ker), dtype=np.float64)
    assert np.max(np.abs(out - 2.5 * ker.sum())) <= tol * 2.5 * np.abs(ker).sum()
    # 3. a border worked by hand: row a b c d e with kernel [1 2 4] -> out[0] = 1 b + 2 a + 4 b (index -1 reflects to 1)
    row = np.array([[1.0, 10.0, 100.0, 1000.0, 10000.0]])
    out = np.asarray(filter2d(row, np.array([[1.0, 2.0, 4.0]])), dtype=np.float64)
    assert np.allclose(out[0], [1 * 10 + 2 * 1 + 4 * 10, 1 * 1 + 2 * 10 + 4 * 100, 1 * 10 + 2 * 100 + 4 * 1000,
                                1 * 100 + 2 * 1000 + 4 * 10000, 1 * 1000 + 2 * 10000 + 4 * 1000], rtol=tol)
    # 4. the definition as loops, kernels LARGER than the image (the reflections wrap more than once), ragged sizes
    for (h, w, kh, kw) in ((7, 9, 5, 3), (4, 5, 9, 11), (6, 3, 3, 7), (1, 8, 3, 5)):
        img, ker = rng.standard_normal((h, w)), rng.standard_normal((kh, kw))
        out = np.asarray(filter2d(img, ker), dtype=np.float64)
        want = filter2d_loops(img, ker)
        assert np.max(np.abs(out - want)) <= tol * np.max(np.abs(want)), (h, w, kh, kw)


def check_gabor_kernel(get_gabor_kernel, tol=1e-12):
    """Known answers any cv2.getGaborKernel stand-in must give; get_gabor_kernel(ksize, sigma, theta, lambd, gamma, psi)."""
    import math
    # centre sample = cos(psi); shape 2 (k // 2) + 1 per axis (100 -> 101 as improcess.py:116,123 relies on)
    for psi in (0.0, 0.3, math.pi / 2):
        k = np.asarray(get_gabor_kernel((100, 100), 4, 1.2, 20, 0.15, psi))
        assert k.shape == (101, 101) and abs(k[50, 50] - math.cos(psi)) <= tol
    # theta = 0: xr = x, yr = y -> closed form per sample, written MIRRORED: kernel[ymax - y][xmax - x]
    sigma, lambd, gamma, psi = 2.0, 5.0, 0.5, 0.3
    k = np.asarray(get_gabor_kernel((9, 7), sigma, 0.0, lambd, gamma, psi))
    assert k.shape == (7, 9)
    for (x, y) in ((1, 2), (-3, 1), (4, -3), (0, 0), (-4, 3)):
        want = math.exp(-x * x / (2 * sigma ** 2) - (y * gamma) ** 2 / (2 * sigma ** 2)) * math.cos(2 * math.pi * x / lambd + psi)
        assert abs(k[3 - y, 4 - x] - want) <= tol, (x, y)
    # theta = pi / 2: xr = y, yr = -x
    k = np.asarray(get_gabor_kernel((9, 7), sigma, math.pi / 2, lambd, gamma, psi))
    for (x, y) in ((1, 2), (-3, 1), (4, -3)):
        c, s = math.cos(math.pi / 2), math.sin(math.pi / 2)
        xr, yr = x * c + y * s, -x * s + y * c
        want = math.exp(-xr * xr / (2 * sigma ** 2) - (yr * gamma) ** 2 / (2 * sigma ** 2)) * math.cos(2 * math.pi * xr / lambd + psi)
        assert abs(k[3 - y, 4 - x] - want) <= tol, (x, y)
    # psi = 0: an even function of (x, y) -> the mirrored write is invisible; psi = pi / 2 at theta = 0: odd in x
    k = np.asarray(get_gabor_kernel((11, 11), 3.0, 0.7, 6.0, 0.4, 0.0))
    assert np.max(np.abs(k - k[::-1, ::-1])) <= tol
    k = np.asarray(get_gabor_kernel((11, 11), 3.0, 0.0, 6.0, 0.4, math.pi / 2))
    assert np.max(np.abs(k + k[:, ::-1])) <= 1e-12 and np.max(np.abs(k - k[::-1, :])) <= tol
    # the scripts' own kernel (improcess.py:116-123) sample by sample against the loops
    th = math.pi / 2 + math.radians(37.0)
    assert np.max(np.abs(np.asarray(get_gabor_kernel((100, 100), 4, th, 20, 0.15, 0)) - gabor_kernel_loops((100, 100), 4, th, 20, 0.15, 0))) <= tol
