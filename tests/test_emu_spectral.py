"""Row spectral operator LOGIC (analytic signal, STFT magnitude, median select, spectrogram
correlation, peak picking, get_fx) on the CPU emulator build: same HIP sources, same C ABI, host
pointers, checked against the reference's golden outputs and the CPU oracle."""
import ctypes

import numpy as np
import pytest
import scipy.signal as sps

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5


@pytest.fixture(scope="module")
def emu():
    return load_emu()


def rel(y, ref):
    return np.max(np.abs(y - ref)) / np.max(np.abs(ref))


def ok(lib, rc):
    assert rc == 0, lib.d4w_last_error()


def analytic(lib, x, mode, fs=0.0, var=None):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    y = np.empty((nx, ns - 1 if mode == 3 else ns), dtype=np.float32)
    ok(lib, lib.d4w_analytic_f32(vp(xf), vp(y), nx, ns, mode, vp(var) if var is not None else None,
                                 ctypes.c_double(fs), None))
    return y


@pytest.mark.parametrize("ns", [480, 360, 250, 77, 2 * 19 * 7])      # packed (even) and complex (odd) paths
def test_envelope_and_hilbert(emu, ns):
    rng = np.random.default_rng(ns)
    x = rng.standard_normal((5, ns)) + 0.3
    z = orc.hilbert(x)
    assert rel(analytic(emu, x, 0), np.abs(z)) < TOL
    assert rel(analytic(emu, x, 1), z.imag) < TOL
    assert np.allclose(np.abs(z), np.abs(sps.hilbert(x, axis=1)))


def test_snr_golden(emu, golden):
    g = golden("fk_40x480.npz")
    x = np.ascontiguousarray(g["x"], dtype=np.float32)
    nx, ns = x.shape
    var = np.empty(nx, dtype=np.float32)
    for env, key in ((0, "snr"), (1, "snr_env")):
        y = np.empty_like(x)
        ok(emu, emu.d4w_snr_f32(vp(x), vp(y), nx, ns, env, vp(var), None))
        ref = g[key]
        fin = np.isfinite(ref) & (ref > -60)
        assert np.max(np.abs(y[fin] - ref[fin])) < 2e-3            # dB; 1e-5 relative on the power ratio is 4e-5 dB
    assert rel(var, np.var(g["x"], axis=1)) < TOL
    # the reference's own value-pinning test (tests/test_dsp.py:136-141)
    r = golden("ref_test_vectors.npz")
    xi = np.ascontiguousarray(r["snr_in"], dtype=np.float32)
    y = np.empty_like(xi)
    v2 = np.empty(2, dtype=np.float32)
    ok(emu, emu.d4w_snr_f32(vp(xi), vp(y), 2, 5, 0, vp(v2), None))
    assert np.allclose(y, r["snr_expected"], atol=1e-4)


@pytest.mark.parametrize("ns", [1, 2, 3, 7, 64, 255, 1000, 1003, 12000])
def test_row_variance_one_sweep(emu, ns):
    """d4w_row_var_f32 (np.std(x, axis=1)**2 of dsp.py:975) in one sweep: lane-wise shifted float64 sums merged by the parallel
    update -- against float64 np.var on rows that break the textbook one-pass formula: an offset a million standard deviations
    away, a first sample that is an outlier, a constant row, a silent row; aligned and unaligned (odd) row lengths."""
    rng = np.random.default_rng(ns)
    x = rng.standard_normal((6, ns))
    x[1] += 1e6
    x[2, 0] = 1e7
    x[3] = 3.25
    x[4] = 0.0
    x[5] *= 1e-10
    xf = np.ascontiguousarray(x, dtype=np.float32)
    var = np.full(6, np.nan, dtype=np.float32)
    ok(emu, emu.d4w_row_var_f32(vp(xf), 6, ns, vp(var), None))
    ref = np.var(xf.astype(np.float64), axis=1)
    assert var[3] == 0 and var[4] == 0
    live = ref > 0
    assert np.all(np.abs(var[live] - ref[live]) <= 3e-7 * ref[live]), (var, ref)
    assert np.all(var[~live] == 0)


def test_instant_freq_golden(emu, golden):
    g = golden("fk_40x480.npz")
    fs = float(g["fs"])
    y = analytic(emu, g["x"][3:4], 3, fs=fs)[0]
    ref = g["ifreq"]
    assert y.shape == ref.shape
    # wrapped phase increments: compare modulo fs (a +-pi increment may resolve to either sign)
    d = np.abs(y - ref)
    d = np.minimum(d, np.abs(d - fs))
    assert np.max(d) < 2e-3 * fs / 2


def test_get_fx_golden(emu, golden):
    g = golden("fk_40x480.npz")
    x = np.ascontiguousarray(g["x"][:, :400], dtype=np.float32)
    nx, ns = x.shape
    for nfft in (512, 300):
        y = np.empty((nx, nfft), dtype=np.float32)
        ok(emu, emu.d4w_fx_f32(vp(x), vp(y), nx, ns, nfft, None))
        ref = g["fx"] if nfft == 512 else orc.get_fx(g["x"][:, :400], nfft)
        assert rel(y, ref) < TOL


def stft(lib, x, n_fft, hop, lo, hi):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    nt = lib.d4w_stft_frames(ns, hop)
    S = np.full((nx, hi - lo + 1, nt), np.nan, dtype=np.float32)
    mx = np.empty(nx, dtype=np.float32)
    ok(lib, lib.d4w_stft_mag_f32(vp(xf), vp(S), vp(mx), nx, ns, n_fft, hop, lo, hi, None))
    return S, mx


@pytest.mark.parametrize("n_fft,hop,ns", [(160, 8, 2000), (256, 12, 2000), (128, 25, 999), (64, 3, 130), (512, 100, 3000), (100, 7, 700),
                                          (148, 7, 1500), (2 * 101, 10, 1200), (74, 40, 900)])   # 37, 101: Bluestein frame transform
def test_stft_magnitude(emu, n_fft, hop, ns):
    rng = np.random.default_rng(n_fft + hop)
    x = rng.standard_normal((3, ns))
    S, mx = stft(emu, x, n_fft, hop, 0, n_fft // 2)
    for c in range(3):
        ref = np.abs(orc.librosa_stft(x[c], n_fft=n_fft, hop_length=hop))
        assert S[c].shape == ref.shape
        assert rel(S[c], ref) < TOL
        assert abs(mx[c] - ref.max()) < TOL * ref.max()
    S2, mx2 = stft(emu, x, n_fft, hop, 5, 17)                       # sliced bins, max still over all bins
    assert np.array_equal(S2, S[:, 5:18]) and np.array_equal(mx2, mx)
    # kept bins only (rowmax = NULL): the two-factor frame lengths; the others need the row maximum
    xf = np.ascontiguousarray(x, dtype=np.float32)
    S3 = np.full_like(S2, np.nan)
    rc = emu.d4w_stft_mag_f32(vp(xf), vp(S3), None, 3, ns, n_fft, hop, 5, 17, None)
    if emu.d4w_stft_mm_eligible(n_fft, hop, 5, 17):                # the matrix-core form (stft_mm.hip): its own rounding
        assert rc == 0 and rel(S3, S2.astype(np.float64)) < 2e-6
    elif n_fft in (128, 160, 256, 512):
        assert rc == 0 and np.array_equal(S3, S2)
    else:
        assert rc != 0


@pytest.mark.parametrize("n_fft,hop,ns,lo,hi", [(160, 8, 12000, 12, 24), (160, 8, 4999, 0, 15), (128, 16, 3001, 60, 64), (96, 24, 2500, 3, 3),
                                                (32, 8, 700, 1, 16), (160, 32, 9000, 70, 80), (64, 8, 100, 1, 9)])
def test_stft_on_the_matrix_cores(emu, n_fft, hop, ns, lo, hi):
    """d4w_stft_mag_f32 without a row maximum for the detector's call shapes runs as frames x DFT rows on the matrix cores:
    several chunks of 256 frames per row, rows of odd length (unaligned rows: sample-by-sample loads), the first / last frames
    hanging over the row ends (zero padding), 1-16 kept bins incl. DC and Nyquist, a large offset (power-of-two chunk scale)."""
    if hi > n_fft // 2:
        hi = n_fft // 2
    assert emu.d4w_stft_mm_eligible(n_fft, hop, lo, hi) == 1
    rng = np.random.default_rng(ns)
    x = rng.standard_normal((3, ns)) * np.array([[1.0], [300.0], [1e-3]]) + np.array([[0.0], [50.0], [0.0]])
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nt = emu.d4w_stft_frames(ns, hop)
    S = np.full((3, hi - lo + 1, nt), np.nan, dtype=np.float32)
    ok(emu, emu.d4w_stft_mag_f32(vp(xf), vp(S), None, 3, ns, n_fft, hop, lo, hi, None))
    for c in range(3):
        ref = np.abs(orc.librosa_stft(xf[c].astype(np.float64), n_fft=n_fft, hop_length=hop))
        assert rel(S[c], ref[lo:hi + 1]) < 2e-6 * (np.abs(ref).max() / np.abs(ref[lo:hi + 1]).max())
    assert emu.d4w_stft_mm_eligible(256, 8, 0, 5) == 0 and emu.d4w_stft_mm_eligible(160, 7, 0, 5) == 0
    assert emu.d4w_stft_mm_eligible(160, 8, 0, 16) == 0 and emu.d4w_stft_mm_eligible(160, 40, 0, 5) == 0
    assert emu.d4w_stft_mag_mm_f32(vp(xf), vp(S), 3, ns, 256, 8, 0, 5, None) != 0


def test_spectrogram_and_nspectrogram_golden(emu, golden):
    g = golden("detect_12x2000.npz")
    x, fs = g["x"], float(g["fs"])
    # dsp.get_spectrogram(x[0], fs, nfft=256, overlap_pct=0.95) -> hop 12
    S, mx = stft(emu, x[5:6], 256, 12, 0, 128)
    per = S[0].size
    ok(emu, emu.d4w_scale_rows_f32(vp(S), 1, ctypes.c_size_t(per), vp(mx), 1, None))
    ref = g["spec_p"]
    assert S[0].shape == ref.shape
    # dB values: 1e-5 of the maximum in the linear domain (the tolerance of the magnitudes themselves)
    assert np.max(np.abs(10.0 ** (S[0].astype(np.float64) / 20) - 10.0 ** (ref / 20))) < TOL
    top = ref > -40
    assert np.max(np.abs(S[0][top] - ref[top])) < 1e-3
    # detect.get_sliced_nspectrogram(x[0], fs, 14, 30, 160, 8)
    ff = np.linspace(0, fs / 2, 81)
    keep = np.where((ff >= g["nspec_ff"][0] - 1e-9) & (ff <= g["nspec_ff"][-1] + 1e-9))[0]
    S, mx = stft(emu, x[5:6], 160, 8, int(keep[0]), int(keep[-1]))
    ok(emu, emu.d4w_scale_rows_f32(vp(S), 1, ctypes.c_size_t(S[0].size), vp(mx), 0, None))
    assert rel(S[0], g["nspec"]) < TOL


@pytest.mark.parametrize("n", [1, 2, 7, 4096, 19513, 1000])
def test_row_median(emu, n):
    rng = np.random.default_rng(n)
    v = rng.standard_normal((3, n)).astype(np.float32)
    v[1] = np.abs(v[1])
    v[2] = np.round(v[2] * 2) / 2                                    # many duplicates
    med = np.empty(3, dtype=np.float32)
    ok(emu, emu.d4w_row_median_f32(vp(v), 3, ctypes.c_size_t(n), vp(med), None))
    assert np.allclose(med, np.median(v.astype(np.float64), axis=1), rtol=1e-6, atol=0)


def test_row_median_even_counts_and_narrow_rows(emu):
    """Rows of the detector's size with an even count: magnitudes spread over octaves, everything inside one quarter-octave
    (one top-digit bin holds the whole row), the upper middle value in the NEXT bin / many bins above the lower one, all-equal
    rows, negative rows, a lower middle value alone in its bin.  (Written for a variant that compacted the selected bin into
    LDS after the first sweep -- two sweeps instead of three, no faster: 437 -> 439 us per file, the kernel is bound by its
    histogram atomics, not by the sweeps -- and kept for the cases.)"""
    rng = np.random.default_rng(99)
    n = 19514                                                                # even
    rows = [np.abs(rng.standard_normal(n)) * rng.uniform(0.5, 2.0, n),      # spread over octaves: compacted
            1.0 + 0.05 * rng.random(n),                                      # one quarter-octave bin: 19 514 keys > the buffer
            np.concatenate((np.full(n // 2, 1.0), np.full(n // 2, 1.5))),    # lower middle the last 1.0, upper middle 1.5
            np.concatenate((np.full(n // 2, 1.0), 1e6 + rng.random(n // 2))),   # upper middle many bins above
            np.full(n, -3.25),
            -np.abs(rng.standard_normal(n)),
            np.concatenate((rng.random(n // 2 - 1) * 0.5, [0.75], 4.0 + rng.random(n // 2)))]   # lower middle alone in its bin
    v = np.stack([rng.permutation(r) for r in rows]).astype(np.float32)
    med = np.empty(len(v), dtype=np.float32)
    ok(emu, emu.d4w_row_median_f32(vp(v), len(v), ctypes.c_size_t(n), vp(med), None))
    ref = np.median(v.astype(np.float64), axis=1)
    assert np.array_equal(med, ref.astype(np.float32)), (med, ref)
    vo = np.ascontiguousarray(v[:, :n - 1])                                  # odd count
    ok(emu, emu.d4w_row_median_f32(vp(vo), len(vo), ctypes.c_size_t(n - 1), vp(med), None))
    assert np.array_equal(med, np.median(vo.astype(np.float64), axis=1).astype(np.float32))


def spectrocorr(lib, S, K, off, nout, zero_ends=0):
    Sf = np.ascontiguousarray(S, dtype=np.float32)
    Kf = np.ascontiguousarray(K, dtype=np.float32)
    nx, nf, nt = Sf.shape
    med = np.empty(nx, dtype=np.float32)
    ok(lib, lib.d4w_row_median_f32(vp(Sf), nx, ctypes.c_size_t(nf * nt), vp(med), None))
    out = np.empty((nx, nout), dtype=np.float32)
    ok(lib, lib.d4w_spectrocorr_f32(vp(Sf), nx, nf, nt, vp(Kf), Kf.shape[1], off, nout, vp(med), zero_ends,
                                    vp(out), None))
    return out


def test_xcorr2d_and_spectrocorr_golden(emu, golden):
    g = golden("detect_12x2000.npz")
    K = g["ker"]
    nk = K.shape[1]
    out = spectrocorr(emu, g["nspec"][None], K, nk // 2, g["nspec"].shape[1])
    assert rel(out[0], g["xcorr2d"]) < TOL
    # whole pipeline per channel: STFT (raw, sliced) -> median -> correlation; the max cancels
    x, fs = g["x"], float(g["fs"])
    ff = np.linspace(0, fs / 2, 81)
    keep = np.where((ff >= g["nspec_ff"][0] - 1e-9) & (ff <= g["nspec_ff"][-1] + 1e-9))[0]
    S, _ = stft(emu, x, 160, 8, int(keep[0]), int(keep[-1]))
    out = spectrocorr(emu, S, K, nk // 2, S.shape[2])
    assert rel(out, g["spectrocorr"]) < TOL
    # odd kernel length and a wide (chunked over f) spectrogram vs the oracle
    rng = np.random.default_rng(0)
    S3 = np.abs(rng.standard_normal((2, 40, 700)))
    K3 = rng.standard_normal((40, 24))
    out = spectrocorr(emu, S3, K3, 24 // 2, 700)
    for c in range(2):
        assert rel(out[c], orc.xcorr2d(S3[c], K3)) < TOL


def test_xcorr_valid_mode_golden(emu, golden):
    """detect.xcorr (detect.py:605-647): valid lags, first / last value forced to zero."""
    g = golden("detect_12x2000.npz")
    K = g["ker"]
    nk = K.shape[1]
    nt = g["nspec"].shape[1]
    out = spectrocorr(emu, g["nspec"][None], K, 0, nt - nk + 1, zero_ends=1)
    assert out.shape[1] == g["xcorr_v"].shape[0]
    assert rel(out[0], g["xcorr_v"]) < TOL and out[0, 0] == 0 and out[0, -1] == 0


@pytest.mark.parametrize("ns", [40, 1500, 12000, 20000])
def test_find_peaks_rows_that_cannot_reach_the_threshold(emu, ns):
    """Rows whose (maximum - minimum) is below the prominence threshold leave the kernel after the summaries (no peak of such a row
    can have that prominence); rows exactly AT the threshold and above it do not: every row against scipy, staged rows (<= 16 384
    samples) and rows read in place, a quiet file with one loud channel as the detector sees it."""
    rng = np.random.default_rng(ns)
    nx = 7
    x = np.abs(rng.standard_normal((nx, ns))).astype(np.float32)
    x[2, ns // 3] = 30.0                                             # the loud channel
    x[3] = 0.25
    x[3, 5] = 0.25 + 8.0                                             # prominence exactly the threshold (8.0 is exact in float32)
    x[4] = 0.25
    x[4, 5] = np.nextafter(np.float32(8.25), np.float32(0))          # ... one ulp below it
    x[5, 7] = np.nan
    x[6, ns // 2] = 9.0
    thr = 8.0
    cap = 64
    idx = np.full((nx, cap), -1, dtype=np.int32)
    cnt = np.full(nx, -1, dtype=np.int32)
    ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
    for c in range(nx):
        if c == 5:
            continue                                                 # (a NaN row: scipy's answer depends on comparison order; only "no crash")
        ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
        assert cnt[c] == len(ref), c
        assert np.array_equal(idx[c, :cnt[c]], ref)
    assert cnt[0] == 0 and cnt[1] == 0 and cnt[2] == 1 and cnt[3] == 1 and cnt[4] == 0 and cnt[6] == 1


def test_find_peaks_matches_scipy(emu):
    rng = np.random.default_rng(5)
    nx, ns = 6, 1500
    x = rng.standard_normal((nx, ns)).astype(np.float32)
    x[1] = np.round(x[1] * 3) / 3                                    # plateaus
    x[2, :] = 0.0                                                    # flat row: no peaks
    x[3] = np.sin(np.arange(ns) * 0.05).astype(np.float32) + 0.01 * x[3]
    x[4, -1] = 10.0                                                  # maximum on the edge
    x[5, 0] = 10.0
    for thr in (0.0, 0.8, 2.5, 4.0 / 3.0, 2.0):                     # the last two sit exactly on plateau prominences
        cap = ns // 2 + 1
        idx = np.full((nx, cap), -1, dtype=np.int32)
        cnt = np.empty(nx, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        for c in range(nx):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert cnt[c] == len(ref), (thr, c)
            assert np.array_equal(idx[c, :cnt[c]], ref)
            if thr == 0.8 and c < 2:
                assert np.array_equal(orc.find_peaks_prominence(x[c], thr), ref)
    # capacity overflow reports the true count and writes only `cap` entries
    idx = np.full((nx, 4), -1, dtype=np.int32)
    cnt = np.empty(nx, dtype=np.int32)
    ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(0.0), vp(idx), vp(cnt), 4, None))
    ref0 = sps.find_peaks(x[0].astype(np.float64), prominence=0.0)[0]
    assert cnt[0] == len(ref0) and np.array_equal(idx[0], ref0[:4])


@pytest.mark.parametrize("ns", [2900, 12000, 17000])
def test_find_peaks_few_candidates_take_the_wave_walk(emu, ns):
    """Rows whose candidates are few (an envelope with the threshold a fraction of its strongest peak, ~10 maxima worth a
    walk per row) walk with a whole wave per side (fp_walk_wave: ballots over samples / block summaries / super-block
    summaries); every decision equals scipy's: smooth envelopes with bursts (walks of thousands of samples that end at a
    higher burst, at a deep dip or at the row end), plateaus on the burst tops, a burst cut by either row end, staged rows and
    rows read in place (17 000 samples), thresholds from 'only the strongest' to 'every burst'."""
    rng = np.random.default_rng(77 + ns)
    nx = 7
    t = np.arange(ns)
    x = np.empty((nx, ns), dtype=np.float32)
    for c in range(nx):
        env = 0.05 + 0.02 * np.abs(rng.standard_normal(ns))
        env = np.convolve(env, np.ones(9) / 9, "same")
        for k in range(int(rng.integers(3, 14))):
            p, w, a = rng.integers(0, ns), rng.uniform(15, 300), rng.uniform(0.2, 1.0)
            env += a * np.exp(-0.5 * ((t - p) / w) ** 2)
        x[c] = env
    x[1] = np.round(x[1] * 40) / 40                                   # plateaus on tops and bases
    x[2, :40] += np.linspace(2.0, 0.0, 40)                            # the largest sample on the left edge
    x[3, -25:] += np.linspace(0.0, 2.0, 25)                           # ... and on the right edge
    x[4] = x[4].max() - x[4]                                          # inverted: wide tops, narrow deep dips
    for frac in (0.9, 0.45, 0.2, 0.08):
        thr = frac * float(x.max() - x.min())
        cap = 1024
        idx = np.full((nx, cap), -1, dtype=np.int32)
        cnt = np.empty(nx, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        for c in range(nx):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert cnt[c] == len(ref), (ns, frac, c, cnt[c], len(ref))
            assert np.array_equal(idx[c, :cnt[c]], ref), (ns, frac, c)


def test_pick_times_golden(emu, golden):
    g = golden("detect_12x2000.npz")
    thr = float(g["thr"])
    for key, env in (("picks", False), ("picks_env", True)):
        c = np.ascontiguousarray(g["corr_hf"], dtype=np.float32)
        if env:
            c = analytic(emu, c, 0)
        nx, ns = c.shape
        cap = ns // 2 + 1
        idx = np.empty((nx, cap), dtype=np.int32)
        cnt = np.empty(nx, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(c), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        got = np.asarray([(r, t) for r in range(nx) for t in idx[r, :cnt[r]]], dtype=np.int64).T.reshape(2, -1)
        ref = g[key]
        a = set(map(tuple, got.T.tolist()))
        b = set(map(tuple, ref.T.tolist()))
        # parity rule: identical except peaks whose prominence is within 1e-4*thr of thr
        assert len(a ^ b) <= max(1, len(b) // 100), (key, sorted(a ^ b))


def test_argument_errors(emu):
    x = np.zeros((2, 64), dtype=np.float32)
    y = np.zeros((2, 64), dtype=np.float32)
    assert emu.d4w_analytic_f32(vp(x), vp(y), 2, 64, 7, None, ctypes.c_double(0), None) == -1
    big = np.zeros((1, 2 * 20011), dtype=np.float32)                                                # prime 20011: no room
    assert emu.d4w_analytic_f32(vp(big), vp(np.empty_like(big)), 1, 2 * 20011, 0, None, ctypes.c_double(0), None) == -1
    assert b"prime" in emu.d4w_last_error() and emu.d4w_analytic_row_fits_lds(2 * 20011) == 0
    xl = np.zeros((1, 90000), dtype=np.float32)                                                     # a window with a prime too long for
    assert emu.d4w_stft_mag_f32(vp(xl), vp(np.zeros((1, 8, 10), dtype=np.float32)), vp(y), 1, 90000, 2 * 20011, 10000, 0, 7, None) == -1
    assert b"prime" in emu.d4w_last_error()                                                         # the Bluestein tile as well
    assert emu.d4w_stft_mag_f32(vp(x), vp(y), vp(y), 2, 64, 15, 4, 0, 7, None) == -1                # odd n_fft
    assert emu.d4w_stft_mag_f32(vp(x), vp(y), vp(y), 2, 64, 16, 4, 0, 9, None) == -1                # bin range


@pytest.mark.parametrize("nx,ns", [(3, 480), (2, 2 * 3 * 5 * 7 * 11), (5, 96), (37, 2 * 41), (7, 2 * 3 * 67), (3, 481), (5, 405), (2, 83),
                                   (4, 1091), (18, 48), (100, 600), (8, 480), (154, 48), (1102, 48)])
def test_analytic_long_row_path(emu, nx, ns):
    """The HBM four-step path (used for rows beyond one workgroup's LDS) on small rows: all four
    modes agree with the single-workgroup kernel and the oracle.  Row lengths with a prime factor > 31 (2 x 41, 2 x 3 x 67)
    run the global-memory Bluestein form of the time transform (fkd_bt_*); the row count is free (37).  Odd lengths (481 = 13 x 37,
    405 = 3^4 x 5, the primes 83 and 1091) go through as complex rows of their own length with scipy's one-sided multiplier.  The last three
    shapes have shape-specialised f-k kernels: their time phase + pass B with the Hilbert pair operation (every row its own
    Hermitian partner) runs instead of the generic kernels."""
    rng = np.random.default_rng(ns)
    x = (rng.standard_normal((nx, ns)) + 0.2).astype(np.float32)
    emu.d4w_analytic_long_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(emu.d4w_analytic_long_ws_bytes(nx, ns), dtype=np.uint8)
    var = np.var(x.astype(np.float64), axis=1).astype(np.float32)
    z = orc.hilbert(x)
    for mode in range(4):
        y = np.empty((nx, ns - 1 if mode == 3 else ns), dtype=np.float32)
        ok(emu, emu.d4w_analytic_long_f32(vp(x), vp(y), nx, ns, mode, vp(var), ctypes.c_double(200.0), vp(ws), None))
        y_short = analytic(emu, x, mode, fs=200.0, var=var)
        if mode == 0:
            assert rel(y, np.abs(z)) < TOL
        elif mode == 1:
            assert rel(y, z.imag) < TOL
        elif mode == 2:
            lin, ref = 10.0 ** (y.astype(np.float64) / 10), np.abs(z) ** 2 / var[:, None]
            assert np.max(np.abs(lin - ref)) / np.max(ref) < TOL
        else:
            d = np.abs(y - y_short)
            assert np.max(np.minimum(d, np.abs(d - 200.0))) < 0.5      # noise rows: |z| ~ 0 samples are ill-conditioned
            assert np.median(d) < 1e-3
    assert emu.d4w_analytic_row_fits_lds(12000) == 1 and emu.d4w_analytic_row_fits_lds(120000) == 0


@pytest.mark.parametrize("ns", [20000, 270000])
def test_find_peaks_long_rows_block_walks(emu, ns):
    """Rows beyond the LDS staging limit (global-memory walks) and beyond 4096 blocks of 64 (larger
    blocks): smooth rows whose prominence walks span many blocks, plus noise, vs SciPy."""
    rng = np.random.default_rng(ns)
    t = np.arange(ns)
    rows = [np.sin(t * 0.0007) + 0.3 * np.sin(t * 0.013) + 0.001 * rng.standard_normal(ns),
            np.abs(sps.hilbert(rng.standard_normal(ns))),
            np.linspace(-1, 1, ns) + 0.05 * np.sin(t * 0.05)]
    x = np.ascontiguousarray(np.stack(rows), dtype=np.float32)
    for thr in (0.0, 0.3, 1.2):
        cap = ns // 2 + 1
        idx = np.empty((3, cap), dtype=np.int32)
        cnt = np.empty(3, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(x), 3, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        for c in range(3):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert cnt[c] == len(ref), (thr, c, cnt[c], len(ref))
            assert np.array_equal(idx[c, :cnt[c]], ref)


@pytest.mark.parametrize("ns", [2 * 37, 2 * 1009, 2 * 3 * 43, 1091, 41])
def test_analytic_and_fx_lengths_with_large_prime_factors(emu, ns):
    """Row lengths whose transform (ns / 2 for even ns, ns for odd) has a prime factor > 31 run the Bluestein
    form of the single-row transform."""
    rng = np.random.default_rng(ns)
    x = np.ascontiguousarray(rng.standard_normal((3, ns)), dtype=np.float32)
    assert emu.d4w_analytic_row_fits_lds(ns) == 1
    z = orc.hilbert(x.astype(np.float64))
    y = np.empty_like(x)
    assert emu.d4w_analytic_f32(vp(x), vp(y), 3, ns, 0, None, ctypes.c_double(0.0), None) == 0, emu.d4w_last_error()
    assert rel(y, np.abs(z)) < TOL
    assert emu.d4w_analytic_f32(vp(x), vp(y), 3, ns, 1, None, ctypes.c_double(0.0), None) == 0
    assert rel(y, z.imag) < TOL
    nfft = ns if ns % 2 else ns // 2 + 1 if (ns // 2 + 1) % 2 else ns // 2        # an awkward nfft as well
    fx = np.empty((3, nfft), dtype=np.float32)
    assert emu.d4w_fx_f32(vp(x), vp(fx), 3, ns, nfft, None) == 0, emu.d4w_last_error()
    assert rel(fx, orc.get_fx(x.astype(np.float64), nfft)) < TOL


def test_pack_picks_table(emu):
    """d4w_pack_picks_i64: the ragged per-row index lists of d4w_find_peaks_f32 as the packed 2 x K (channel, time)
    table of detect.convert_pick_times (reference detect.py:277-303)."""
    rng = np.random.default_rng(12)
    nx, ns, cap, thr = 9, 700, 256, 1.5
    x = rng.standard_normal((nx, ns)).astype(np.float32)
    x[4] = 0.0                                          # a row without picks
    idx = np.zeros((nx, cap), dtype=np.int32)
    cnt = np.zeros(nx, dtype=np.int32)
    ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
    # the offsets, the total and the largest count of a picker call come from one launch (d4w_pick_offsets_i64)
    off = np.full(nx, -1, dtype=np.int64)
    summ = np.full(2, -1, dtype=np.int64)
    ok(emu, emu.d4w_pick_offsets_i64(vp(cnt), nx, vp(off), vp(summ), None))
    assert np.array_equal(off, np.cumsum(cnt)) and summ[0] == cnt.max() and summ[1] == cnt.sum()
    for n in (1, 63, 1024, 1025, 20000, 131070):      # one row per thread, several, ragged last threads
        c2 = rng.integers(0, 70000, n).astype(np.int32)
        o2, s2 = np.empty(n, dtype=np.int64), np.empty(2, dtype=np.int64)
        ok(emu, emu.d4w_pick_offsets_i64(vp(c2), n, vp(o2), vp(s2), None))
        assert np.array_equal(o2, np.cumsum(c2.astype(np.int64))) and s2[0] == c2.max() and s2[1] == c2.astype(np.int64).sum()
    total = int(off[-1])
    out = np.full((2, total), -1, dtype=np.int64)
    emu.d4w_pack_picks_i64.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    ok(emu, emu.d4w_pack_picks_i64(vp(idx), vp(cnt), vp(off), nx, cap, total, vp(out), None))
    ref = [sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0] for c in range(nx)]
    want = np.asarray((np.concatenate([np.full(len(p), i) for i, p in enumerate(ref)]), np.concatenate(ref)))
    assert total == want.shape[1] and np.array_equal(out, want)


@pytest.mark.parametrize("ns", [16388, 20000, 1920 * 9, 1920 * 9 + 4, 40004])
def test_find_peaks_long_rows_window_sweep(emu, ns):
    """Rows beyond the LDS staging limit pass through LDS in windows of 1920 + 2 x 64 samples and settle their maxima
    there (fp_sweep_segments): oscillating rows (a maximum every ~9 samples, the raw correlograms of detect.pick_times,
    detect.py:249-274), plateaus lying across window edges and at the row ends, maxima whose walks leave the window,
    a row of white noise, lengths that end on / just after a window edge -- vs SciPy."""
    rng = np.random.default_rng(ns)
    t = np.arange(ns)
    osc = np.sin(2 * np.pi * t * 23.0 / 200.0) * np.abs(sps.hilbert(rng.standard_normal(ns))) * 0.3
    plat = (np.sin(t * 0.11) * (1.0 + 0.5 * np.sin(t * 0.0031))).astype(np.float32)
    plat = np.round(plat * 4.0) / 4.0                                    # quantised: plateaus everywhere
    for e in (1920, 2 * 1920, 3 * 1920 - 64, 4 * 1920 + 64, 7 * 1920 + 1):             # plateaus across the core / halo edges
        plat[e - 3:e + 3] = 3.0
        plat[e - 70:e - 60] = 2.5
    plat[:5] = 9.0                                                       # plateaus touching the row ends are no peaks
    plat[-5:] = 9.0
    slow = np.sin(t * 0.0009) + 0.2 * np.sin(t * 0.7)                    # walks of thousands of samples
    env = np.abs(sps.hilbert(np.convolve(rng.standard_normal(ns), np.hanning(24), "same")))    # a handful of maxima per window
    half = ns // 2
    x = np.ascontiguousarray(np.stack([osc, plat, slow, rng.standard_normal(ns), env,
                                       np.concatenate([env[:half] * 0.2, osc[half:] + 0.5]),  # quiet first: left to the marking sweep
                                       np.concatenate([osc[:half] + 0.5, env[half:] * 0.2])]), dtype=np.float32)
    nx = x.shape[0]
    for thr in (0.0, 0.2, 0.45 * float(x[0].max()), 0.45 * float(env.max()), 1.5, 1e30):
        cap = ns // 2 + 1
        idx = np.empty((nx, cap), dtype=np.int32)
        cnt = np.empty(nx, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        for c in range(nx):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert cnt[c] == len(ref), (thr, c, cnt[c], len(ref))
            assert np.array_equal(idx[c, :cnt[c]], ref)


@pytest.mark.parametrize("ns", [12000, 16384, 10001])
def test_find_peaks_more_candidates_than_one_list(emu, ns):
    """Rows with more maxima than the 4096-entry candidate list (white noise has ~ns/3): thr = 0 sends the scan through
    the word-round fallback, a threshold through the single merged list; staged rows with 16-byte and scalar loads."""
    rng = np.random.default_rng(ns)
    x = rng.standard_normal((2, ns)).astype(np.float32)
    x[1, ::2] = np.abs(x[1, ::2]) + 4.0                                 # every second sample a maximum: ns / 2 candidates
    x[1, 1::2] = -np.abs(x[1, 1::2])
    for thr in (0.0, 1.0, 5.0):
        cap = ns // 2 + 1
        idx = np.empty((2, cap), dtype=np.int32)
        cnt = np.empty(2, dtype=np.int32)
        ok(emu, emu.d4w_find_peaks_f32(vp(x), 2, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None))
        for c in range(2):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert cnt[c] == len(ref), (thr, c, cnt[c], len(ref))
            assert np.array_equal(idx[c, :cnt[c]], ref)


def test_fuzz_analytic_any_row_length(emu):
    """Random row lengths (even and odd, prime factors up to 43) through the single-workgroup kernel and through the long-row
    path: envelope and Hilbert transform against the oracle."""
    emu.d4w_analytic_long_ws_bytes.restype = ctypes.c_size_t
    rng = np.random.default_rng(9)
    small = [2, 3, 4, 5, 6, 7, 8, 10, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43]
    done = 0
    while done < 14:
        L = int(np.prod(rng.choice(small, size=int(rng.integers(1, 4)))))
        if L > 3000:
            continue
        ns = 2 * L if rng.random() < 0.5 else (L if L % 2 else L + 1)
        nx = int(rng.integers(1, 4))
        x = (rng.standard_normal((nx, ns)) + 0.1).astype(np.float32)
        z = orc.hilbert(x)
        for mode in (0, 1):
            ref = np.abs(z) if mode == 0 else z.imag
            assert rel(analytic(emu, x, mode), ref) < TOL, ("single workgroup", ns, mode)
            ws = np.empty(emu.d4w_analytic_long_ws_bytes(nx, ns), dtype=np.uint8)
            y = np.empty((nx, ns), dtype=np.float32)
            ok(emu, emu.d4w_analytic_long_f32(vp(x), vp(y), nx, ns, mode, None, ctypes.c_double(200.0), vp(ws), None))
            assert rel(y, ref) < TOL, ("long-row path", ns, mode)
        done += 1
