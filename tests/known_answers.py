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
    diff = sorted(set(int(i) for i in got) ^ set(int(i) for i in ref))
    for i in diff:
        assert 0 < i < len(ref_signal) - 1, (what, i, "differing pick at the row edge")
        assert ref_signal[i] >= ref_signal[i - 1] and ref_signal[i] >= ref_signal[i + 1], (what, i, "differing pick is not a local maximum of the reference")
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
