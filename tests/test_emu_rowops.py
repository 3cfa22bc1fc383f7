"""Row-operator kernel LOGIC (zero-phase SOS filter, matched filter) on the CPU emulator build:
same HIP sources, same C ABI, host pointers.  Complements the -m gpu parity tests."""
import ctypes

import numpy as np
import pytest
import scipy.signal as sps

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5


@pytest.fixture(scope="module")
def emu():
    lib = load_emu()
    lib.d4w_sosfiltfilt_ws_bytes.restype = ctypes.c_size_t
    return lib


def rel(y, ref):
    return np.max(np.abs(y - ref)) / np.max(np.abs(ref))


def sosfiltfilt_emu(lib, x, sos, padlen, seg_len=0, warm=0):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    zi = np.ascontiguousarray(sps.sosfilt_zi(sos), dtype=np.float64)
    ws = np.empty(lib.d4w_sosfiltfilt_ws_bytes(nx, ns, padlen), dtype=np.uint8)
    y = np.empty_like(xf)
    rc = lib.d4w_sosfiltfilt_f32(vp(xf), vp(y), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, seg_len, warm,
                                 vp(ws), None)
    assert rc == 0, lib.d4w_last_error()
    return y


def test_bp_filt_golden_single_segment(emu, golden):
    g = golden("fk_40x480.npz")
    fs = float(g["fs"])
    sos = sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos")
    y = sosfiltfilt_emu(emu, g["x"], sos, padlen=51)
    assert rel(y, g["y_bp"]) < TOL                      # reference dsp.bp_filt (ba form, float64)


def test_sosfiltfilt_golden(emu, golden):
    g = golden("fk_40x480.npz")
    assert rel(sosfiltfilt_emu(emu, g["x"], g["sos_hp"], padlen=9), g["y_sos_hp"]) < TOL
    assert rel(sosfiltfilt_emu(emu, g["x"], g["sos_bp"], padlen=33), g["y_sos_bp"]) < TOL


def test_segmented_rows_match_exact(emu):
    """Segments with warm-up reproduce the whole-row recursion (ragged: 70 rows, ns not a
    multiple of the chunk or the segment)."""
    rng = np.random.default_rng(3)
    nx, ns, fs = 70, 3001, 200.0
    x = rng.standard_normal((nx, ns)) + 3.0
    sos = sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos")
    ref = orc.sosfiltfilt(sos, x)
    y1 = sosfiltfilt_emu(emu, x, sos, padlen=51)
    y2 = sosfiltfilt_emu(emu, x, sos, padlen=51, seg_len=1024, warm=768)
    assert rel(y1, ref) < TOL
    assert rel(y2, ref) < TOL
    assert rel(y2, y1) < 2e-6


def test_sosfiltfilt_errors(emu):
    x = np.zeros((2, 40), dtype=np.float32)
    sos = sps.butter(8, [0.14, 0.3], "bp", output="sos")
    zi = sps.sosfilt_zi(sos)
    ws = np.empty(1 << 16, dtype=np.uint8)
    rc = emu.d4w_sosfiltfilt_f32(vp(x), vp(x), 2, 40, vp(sos), vp(zi), 8, 51, 0, 0, vp(ws), None)
    assert rc == -1 and b"padlen, which is 51" in emu.d4w_last_error()


def xcorr_emu(lib, x, taps_list, normalize=True, lens=True):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    lt = max(4, -(-max(len(t) for t in taps_list) // 4) * 4)
    taps = np.zeros((len(taps_list), lt), dtype=np.float32)
    for i, t in enumerate(taps_list):
        taps[i, :len(t)] = t
    mean = np.empty(nx, dtype=np.float64)          # float64 row means (include/d4w.h)
    mx = np.empty(nx, dtype=np.float32)
    if normalize:
        assert lib.d4w_row_stats_f32(vp(xf), nx, ns, vp(mean), vp(mx), None) == 0
    ys = [np.empty_like(xf) for _ in taps_list]
    if lens:
        rc = lib.d4w_xcorr_lens_f32(vp(xf), nx, ns, vp(mean) if normalize else None, vp(mx) if normalize else None,
                                    vp(taps), len(taps_list), lt, len(taps_list[0]), len(taps_list[-1]),
                                    vp(ys[0]), vp(ys[1]) if len(ys) > 1 else None, None)
    else:
        rc = lib.d4w_xcorr_f32(vp(xf), nx, ns, vp(mean) if normalize else None, vp(mx) if normalize else None,
                               vp(taps), len(taps_list), lt, vp(ys[0]), vp(ys[1]) if len(ys) > 1 else None, None)
    assert rc == 0, lib.d4w_last_error()
    return ys, mean, mx


def norm_taps(tpl):
    """detect.py:158 on the host: (y - mean(y)) / max|y| over the zero-padded template, support only."""
    tpl = np.asarray(tpl, dtype=np.float64)
    L = int(np.max(np.nonzero(tpl)[0])) + 1
    return ((tpl - tpl.mean()) / np.max(np.abs(tpl)))[:L]


def test_cross_correlogram_golden(emu, golden):
    d = golden("detect_12x2000.npz")
    (yh, yl), mean, mx = xcorr_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])])
    assert np.allclose(mean, d["x"].mean(axis=1), atol=1e-6 * np.abs(d["x"]).max())
    assert np.allclose(mx, np.abs(d["x"]).max(axis=1), rtol=1e-6)
    assert rel(yh, d["corr_hf"]) < TOL                  # reference detect.compute_cross_correlogram
    assert rel(yl, d["corr_lf"]) < TOL
    (y1,), _, _ = xcorr_emu(emu, d["x"], [norm_taps(d["hf"])])
    assert np.array_equal(y1, yh)                       # fused and single-template paths agree bit for bit
    # longer template first (the kernel orders them itself), and the padded-length entry point
    (zl, zh), _, _ = xcorr_emu(emu, d["x"], [norm_taps(d["lf"]), norm_taps(d["hf"])])
    assert np.array_equal(zl, yl) and np.array_equal(zh, yh)
    (ph, pl), _, _ = xcorr_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])], lens=False)
    assert np.array_equal(ph, yh) and np.array_equal(pl, yl)


@pytest.mark.parametrize("nx,ns,l0,l1", [(3, 2100, 7, 300), (2, 4097, 136, 156), (5, 999, 33, 1), (1, 64, 64, 10)])
def test_xcorr_ragged_shapes(emu, nx, ns, l0, l1):
    """Odd row counts (row pairs), rows that are not a multiple of the lag tile or of 4, supports that
    are odd / longer than one tap round / longer than the tile remainder, both environment tilings."""
    rng = np.random.default_rng(ns + l0)
    x = rng.standard_normal((nx, ns))
    t0, t1 = rng.standard_normal(l0), rng.standard_normal(l1)
    (y0, y1), _, _ = xcorr_emu(emu, x, [t0, t1], normalize=False)
    for c in range(nx):
        assert rel(y0[c], orc.shift_xcorr(x[c], np.pad(t0, (0, ns - l0)))) < TOL
        assert rel(y1[c], orc.shift_xcorr(x[c], np.pad(t1, (0, ns - l1)))) < TOL


def test_shift_xcorr_long_template(emu, golden):
    """detect.shift_xcorr with a dense full-length second operand (several tap rounds, ragged row)."""
    rng = np.random.default_rng(5)
    n = 1237
    x = rng.standard_normal(n)
    y = rng.standard_normal(n)
    (c,), _, _ = xcorr_emu(emu, x[None, :], [y], normalize=False)
    assert rel(c[0], orc.shift_xcorr(x, y)) < TOL
    d = golden("detect_12x2000.npz")
    (c2,), _, _ = xcorr_emu(emu, d["x"][2:3], [d["hf"][:137]], normalize=False)
    assert rel(c2[0], d["xc"]) < TOL


# ------------------------------------------------------------------------------------------
# overlap-save FFT matched filter (csrc/xcorr_fft.hip)
# ------------------------------------------------------------------------------------------
def xcorr_fft_emu(lib, x, taps_list, normalize=True):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    lt = max(4, -(-max(len(t) for t in taps_list) // 4) * 4)
    taps = np.zeros((len(taps_list), lt), dtype=np.float32)
    for i, t in enumerate(taps_list):
        taps[i, :len(t)] = t
    mean = np.empty(nx, dtype=np.float64)          # float64 row means (include/d4w.h)
    mx = np.empty(nx, dtype=np.float32)
    if normalize:
        assert lib.d4w_row_stats_f32(vp(xf), nx, ns, vp(mean), vp(mx), None) == 0
    lib.d4w_xcorr_fft_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(lib.d4w_xcorr_fft_ws_bytes(), dtype=np.uint8)
    ys = [np.full_like(xf, np.nan) for _ in taps_list]
    rc = lib.d4w_xcorr_fft_f32(vp(xf), nx, ns, vp(mean) if normalize else None, vp(mx) if normalize else None,
                               vp(taps), len(taps_list), lt, len(taps_list[0]), len(taps_list[-1]),
                               vp(ys[0]), vp(ys[1]) if len(ys) > 1 else None, vp(ws), None)
    assert rc == 0, lib.d4w_last_error()
    return ys


@pytest.mark.parametrize("switch", [{"D4W_XF_TPAIR": "1"}, {"D4W_XF_FUSED": "2"}, {"D4W_XF_FUSED": "0"}])
def test_xcorr_fft_template_pair_kernel(golden, switch):
    """The alternative two-template kernels give the same correlograms as the default four-stage fused
    one: templates packed through one inverse transform (D4W_XF_TPAIR=1), the three-stage fused kernel
    (D4W_XF_FUSED=2), one launch per template (D4W_XF_FUSED=0).  The library reads the switches once
    per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); import tests.test_emu_rowops as t; "
            "from tests.emu_util import load_emu; emu = load_emu(); "
            "d = np.load(%r); yh, yl = t.xcorr_fft_emu(emu, d['x'], [t.norm_taps(d['hf']), t.norm_taps(d['lf'])]); "
            "assert t.rel(yh, d['corr_hf']) < 1e-5 and t.rel(yl, d['corr_lf']) < 1e-5; "
            "rng = np.random.default_rng(1); x = rng.standard_normal((3, 9001)); a, b = rng.standard_normal(161), rng.standard_normal(7); "
            "y0, y1 = t.xcorr_fft_emu(emu, x, [a, b], normalize=False); "
            "assert t.rel(y0[2], t.orc.shift_xcorr(x[2], np.pad(a, (0, 9001 - 161)))) < 1e-5; "
            "assert t.rel(y1[1], t.orc.shift_xcorr(x[1], np.pad(b, (0, 9001 - 7)))) < 1e-5; print('ok')"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detect_12x2000.npz")))
    env = dict(os.environ, **switch)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_xcorr_fft_golden(emu, golden):
    d = golden("detect_12x2000.npz")
    yh, yl = xcorr_fft_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])])
    assert rel(yh, d["corr_hf"]) < TOL                  # reference detect.compute_cross_correlogram
    assert rel(yl, d["corr_lf"]) < TOL
    (y1,) = xcorr_fft_emu(emu, d["x"], [norm_taps(d["lf"])])
    assert rel(y1, d["corr_lf"]) < TOL
    (yd, _), _, _ = xcorr_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])])
    assert rel(yh, yd.astype(np.float64)) < 3e-6        # FFT and direct forms agree to rounding


@pytest.mark.parametrize("nx,ns,l0,l1", [(3, 9001, 161, 7), (2, 4096, 136, 156), (1, 3937, 1, 160), (4, 1300, 50, 50)])
def test_xcorr_fft_ragged(emu, nx, ns, l0, l1):
    """Odd row counts, odd row lengths (unaligned row pairs), several blocks with a ragged tail, a
    single block shorter than the transform, the maximum support."""
    rng = np.random.default_rng(ns + l0)
    x = rng.standard_normal((nx, ns)) + 0.5
    t0, t1 = rng.standard_normal(l0), rng.standard_normal(l1)
    y0, y1 = xcorr_fft_emu(emu, x, [t0, t1], normalize=False)
    for c in range(nx):
        assert rel(y0[c], orc.shift_xcorr(x[c], np.pad(t0, (0, ns - l0)))) < TOL
        assert rel(y1[c], orc.shift_xcorr(x[c], np.pad(t1, (0, ns - l1)))) < TOL
    assert emu.d4w_xcorr_fft_max_support() == 161


@pytest.mark.parametrize("ns", [1001, 1000, 2056, 12304])
def test_raw2strain_ingest(emu, ns):
    """data_handle.load_das_data channel selection + raw2strain (data_handle.py:157-176,213-214): scalar loads (rows that
    do not start on 16 bytes), 16-byte loads with the row held in registers, and rows longer than the registers hold."""
    rng = np.random.default_rng(9)
    nch = 23
    raw = (rng.standard_normal((nch, ns)) * 3e4 + 1.5e5).astype(np.int32)
    scale = 2.3e-11
    c0, c1, step = 3, 21, 4
    ref = raw[c0:c1:step].astype(np.float64)
    ref -= ref.mean(axis=1, keepdims=True)                      # raw2strain
    ref *= scale
    nx = ref.shape[0]
    for dt, code in ((np.int32, 0), (np.int16, 1), (np.float32, 2), (np.float64, 3)):
        r = (raw // 8).astype(dt) if dt == np.int16 else raw.astype(dt)
        rr = r[c0:c1:step].astype(np.float64)
        rr = (rr - rr.mean(axis=1, keepdims=True)) * scale
        y = np.empty((nx, ns), dtype=np.float32)
        rc = emu.d4w_raw2strain_f32(vp(np.ascontiguousarray(r)), code, ns, c0, step, nx, ctypes.c_double(scale), vp(y), None)
        assert rc == 0, emu.d4w_last_error()
        assert rel(y, rr) < 1e-6
    assert emu.d4w_raw2strain_f32(vp(raw), 7, ns, 0, 1, 2, ctypes.c_double(1.0), vp(y), None) == -1


def test_xcorr_dc_tail_exact(emu):
    """With the DC-tail term the correlogram equals detect.compute_cross_correlogram for templates of
    non-zero mean (rows of non-zero mean, ragged length, support longer / shorter than a scan chunk)."""
    rng = np.random.default_rng(12)
    for nx, ns, L in ((3, 2600, 40), (2, 1031, 300), (1, 5000, 1)):
        x = rng.standard_normal((nx, ns)) + 0.4
        tpl = np.zeros(ns)
        tpl[:L] = np.abs(rng.standard_normal(L)) + 0.2            # clearly non-zero mean
        ref = orc.compute_cross_correlogram(x, tpl)
        taps = norm_taps(tpl)
        coef = tpl.mean() / np.max(np.abs(tpl))
        (y,), mean, mx = xcorr_emu(emu, x, [taps])
        assert rel(y, ref) > 1e-3                                  # without the tail: far off
        xf = np.ascontiguousarray(x, dtype=np.float32)
        rc = emu.d4w_xcorr_dc_tail_f32(vp(xf), nx, ns, vp(mean), vp(mx), ctypes.c_double(coef), L, vp(y), None)
        assert rc == 0, emu.d4w_last_error()
        assert rel(y, ref) < TOL, (nx, ns, L, rel(y, ref))


def test_xcorr_dc_tail_is_decided_per_row_on_the_data(emu):
    """The DC-tail term's weight is a property of the row (|coef| g x a prefix sum of the de-meaned row): d4w_row_prefix_max_f32
    bounds it, d4w_xcorr_dc_tail_rows_f32 leaves a row alone only when the term cannot exceed eps of the row's own largest
    correlation, and forms the row maximum of every row it changes again.  Rows: white, band-limited (no low frequencies:
    tiny prefix sums), a slow drift and a step (prefix sums of ns/4 samples' size: 1e-3 of the correlogram with the
    fin-whale template, which a rule that predicts the term from the template alone lets pass)."""
    rng = np.random.default_rng(120)
    fs, ns = 200.0, 6000
    t = np.arange(ns) / fs
    tpl = orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    L = int(np.nonzero(tpl)[0][-1]) + 1
    coef = tpl.mean() / np.max(np.abs(tpl))
    assert coef != 0.0
    w = rng.standard_normal((4, ns))
    band = sps.sosfiltfilt(sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos"), rng.standard_normal(ns))
    x = np.stack([w[0], band, 0.05 * w[1] + np.sin(2 * np.pi * t / 30.0), np.where(t < 15, 1.0, -1.0) + 0.01 * w[2]]) + 0.3
    nx = len(x)
    ref = orc.compute_cross_correlogram(x, tpl)
    (y,), mean, mx = xcorr_emu(emu, x, [norm_taps(tpl)])
    row_err = lambda k: np.max(np.abs(y[k] - ref[k])) / np.max(np.abs(ref[k]))
    before = [row_err(k) for k in range(nx)]
    assert before[1] < 1e-6 and before[2] > 1e-4 and before[3] > 1e-4, before      # the hole: drift and step rows without the term
    xf = np.ascontiguousarray(x, dtype=np.float32)
    pm = np.empty(nx, dtype=np.float32)
    assert emu.d4w_row_prefix_max_f32(vp(xf), nx, ns, vp(mean), vp(pm), None) == 0, emu.d4w_last_error()
    d = xf.astype(np.float64) - mean[:, None]
    P = np.abs(np.cumsum(d, axis=1)).max(axis=1)
    assert np.allclose(pm, P, rtol=1e-4, atol=1e-3 * np.abs(d).max()), (pm, P)
    # the same maxima together with the statistics in one launch
    mean2, mx2, pm2 = np.empty(nx), np.empty(nx, dtype=np.float32), np.empty(nx, dtype=np.float32)
    assert emu.d4w_row_stats_prefix_f32(vp(xf), nx, ns, vp(mean2), vp(mx2), vp(pm2), None) == 0, emu.d4w_last_error()
    assert np.array_equal(mean2, mean) and np.array_equal(mx2, mx) and np.array_equal(pm2, pm)
    rmax = y.max(axis=1).astype(np.float32)
    rmax0 = rmax.copy()
    y0 = y.copy()
    eps = 1e-6
    rc = emu.d4w_xcorr_dc_tail_rows_f32(vp(xf), nx, ns, vp(mean), vp(mx), ctypes.c_double(coef), L, vp(y), vp(pm), vp(rmax),
                                         ctypes.c_double(eps), None)
    assert rc == 0, emu.d4w_last_error()
    changed = [not np.array_equal(y[k], y0[k]) for k in range(nx)]
    assert changed == [True, False, True, True], changed
    for k in range(nx):
        assert row_err(k) < (eps * 1.5 if not changed[k] else 2e-6), (k, row_err(k))
        assert rmax[k] == y[k].max()
    assert rmax[1] == rmax0[1]
    # pmax without rowmax is refused; without either, every row gets the term (d4w_xcorr_dc_tail_f32)
    assert emu.d4w_xcorr_dc_tail_rows_f32(vp(xf), nx, ns, vp(mean), vp(mx), ctypes.c_double(coef), L, vp(y), vp(pm), None,
                                          ctypes.c_double(eps), None) == -1
    y1 = y0.copy()
    assert emu.d4w_xcorr_dc_tail_rows_f32(vp(xf), nx, ns, vp(mean), vp(mx), ctypes.c_double(coef), L, vp(y1), None, None,
                                          ctypes.c_double(0.0), None) == 0
    assert not np.array_equal(y1[1], y0[1]) and np.array_equal(y1[0], y[0])


def test_bandpass_low_edge_uses_float64_states(emu):
    """A 5 Hz band edge at 200 Hz puts poles at radius > 0.98: the float32 recursion's rounding noise
    (2e-5) exceeds the budget, the library switches to float64 states (float32 I/O)."""
    rng = np.random.default_rng(40)
    x = rng.standard_normal((8, 3250)) + rng.standard_normal((8, 1)) * 3
    fs, lo, hi = 200.0, 5.074, 38.506
    sos = sps.butter(8, [lo / (fs / 2), hi / (fs / 2)], "bp", output="sos")
    ref = orc.bp_filt(x, fs, lo, hi)
    y = sosfiltfilt_emu(emu, x, sos, padlen=51)
    assert rel(y, ref) < TOL


def test_sosfiltfilt_dc_offset_and_lowpass(emu):
    """Rows with a large offset in front of a narrow low band (the offset is taken out before the float32
    recursion and c |H(1)|^2 put back after), and a low-pass whose DC gain is 1 (the put-back term)."""
    rng = np.random.default_rng(41)
    fs = 200.0
    x = rng.standard_normal((70, 3654)) + rng.standard_normal((70, 1)) * 30
    sos = sps.butter(8, [8.258 / (fs / 2), 19.632 / (fs / 2)], "bp", output="sos")
    # ground truth = the float64 second-order-section filter: for this narrow low band the reference's
    # 17-coefficient `ba` form (dsp.py:878-879) is itself 5.5e-5 away from it in float64
    ref = sps.sosfiltfilt(sos, x, axis=1, padlen=51)
    assert rel(orc.bp_filt(x, fs, 8.258, 19.632), ref) > 3e-5
    assert rel(sosfiltfilt_emu(emu, x, sos, padlen=51), ref) < TOL
    assert rel(sosfiltfilt_emu(emu, x, sos, padlen=51, seg_len=1024, warm=2048), ref) < TOL
    lp = sps.butter(4, 0.2, "lp", output="sos")
    ref_lp = sps.sosfiltfilt(lp, x, axis=1)
    assert rel(sosfiltfilt_emu(emu, x, lp, padlen=15), ref_lp) < TOL
    buf = np.ascontiguousarray(x, dtype=np.float32)                     # in place (x aliases y)
    zi = np.ascontiguousarray(sps.sosfilt_zi(lp), dtype=np.float64)
    ws = np.empty(emu.d4w_sosfiltfilt_ws_bytes(70, 3654, 15), dtype=np.uint8)
    assert emu.d4w_sosfiltfilt_f32(vp(buf), vp(buf), 70, 3654, vp(np.ascontiguousarray(lp)), vp(zi), lp.shape[0], 15, 0, 0,
                                   vp(ws), None) == 0
    assert rel(buf, ref_lp) < TOL


@pytest.mark.parametrize("ns,K", [(9000, 460), (5000, 0), (12000, 1024), (4500, 30)])
def test_fir_fft_interior(emu, ns, K):
    """d4w_fir_fft_f32 (overlap-save FFT blocks, the interior of the zero-phase band-pass): y[n] = sum_j taps[j] x[n - K + j]
    on K <= n < ns - K, nothing written outside, the subtracted row constant restored through dc_gain."""
    rng = np.random.default_rng(ns + K)
    nx = 5
    x = (rng.standard_normal((nx, ns)) + np.arange(nx)[:, None] * 30.0).astype(np.float32)      # large per-row offsets
    taps = rng.standard_normal(2 * K + 1) * np.hanning(2 * K + 3)[1:-1]
    taps = ((taps + taps[::-1]) / 2).astype(np.float32)
    first = np.ascontiguousarray(x[:, 0])
    y = np.full((nx, ns), np.nan, dtype=np.float32)
    emu.d4w_xcorr_fft_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(emu.d4w_xcorr_fft_ws_bytes(), dtype=np.uint8)
    emu.d4w_fir_fft_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    dcg = float(taps.astype(np.float64).sum())
    rc = emu.d4w_fir_fft_f32(vp(x), nx, ns, vp(taps), K, vp(first), dcg, vp(y), vp(ws), None)
    assert rc == 0, emu.d4w_last_error()
    assert np.isnan(y[:, :K]).all() and np.isnan(y[:, ns - K:]).all()
    ref = np.stack([np.convolve(r.astype(np.float64), taps.astype(np.float64)[::-1], "valid") for r in x])
    got = y[:, K:ns - K]
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - ref)) < 1e-5 * np.max(np.abs(ref))
    assert emu.d4w_fir_fft_f32(vp(x), nx, ns, vp(taps), 1026, vp(first), dcg, vp(y), vp(ws), None) != 0


@pytest.mark.parametrize("nx,ns", [(3, 4100), (4, 9000), (2, 3936 * 2)])
def test_xcorr_fft_continuation(emu, nx, ns):
    """d4w_xcorr_fft_cont_f32: the last lags read the head of the next file's rows in place -- the same numbers as
    correlating the concatenation [x | head] with x's own row statistics and keeping the first ns lags; without a
    continuation it is d4w_xcorr_fft_f32."""
    rng = np.random.default_rng(nx * ns)
    x = (rng.standard_normal((nx, ns)) + 0.3).astype(np.float32)
    nxt = (rng.standard_normal((nx, 500)) - 0.2).astype(np.float32)             # pitch 500, only the first L - 1 are read
    t0, t1 = rng.standard_normal(136) * np.hanning(136), rng.standard_normal(156) * np.hanning(156)
    L = 156
    lt = 156
    taps = np.zeros((2, lt), dtype=np.float32)
    taps[0, :136], taps[1, :156] = t0, t1
    mean, mx = np.empty(nx, dtype=np.float64), np.empty(nx, dtype=np.float32)
    assert emu.d4w_row_stats_f32(vp(x), nx, ns, vp(mean), vp(mx), None) == 0
    emu.d4w_xcorr_fft_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(emu.d4w_xcorr_fft_ws_bytes(), dtype=np.uint8)

    def run(xin, n, xn=None, ld=0, nn=0):
        ys = [np.full((nx, n), np.nan, dtype=np.float32) for _ in range(2)]
        rc = emu.d4w_xcorr_fft_cont_f32(vp(xin), nx, n, vp(xn) if xn is not None else None, ld, nn, vp(mean), vp(mx), vp(taps), 2,
                                        lt, 136, 156, vp(ys[0]), vp(ys[1]), vp(ws), None)
        assert rc == 0, emu.d4w_last_error()
        return ys
    cont = run(x, ns, nxt, 500, L - 1)
    ext = np.ascontiguousarray(np.concatenate((x, nxt[:, :L - 1]), axis=1))
    ref = run(ext, ns + L - 1)
    for c, r in zip(cont, ref):
        assert np.max(np.abs(c - r[:, :ns])) <= 2e-6 * np.max(np.abs(r))
    plain = run(x, ns)
    assert not np.array_equal(plain[0][:, -L:], cont[0][:, -L:])                 # the continuation did change the last lags
    assert np.max(np.abs(plain[0][:, :ns - L] - cont[0][:, :ns - L])) <= 2e-6 * np.max(np.abs(plain[0]))   # same block transform, other tail
    # float64 ground truth of the last lags of row 0, template 1
    xa = (np.concatenate((x[0], nxt[0, :L - 1])).astype(np.float64) - float(mean[0])) / float(mx[0])
    for k in (ns - 1, ns - 77, ns - L + 1):
        want = float(np.dot(xa[k:k + 156], t1))
        assert abs(cont[1][0, k] - want) < 1e-5 * np.max(np.abs(ref[1]))
    assert emu.d4w_xcorr_fft_cont_f32(vp(x), nx, ns, vp(nxt), 500, L - 1, vp(mean), vp(mx), vp(taps), 1, lt, 136, 136,
                                      vp(cont[0]), None, vp(ws), None) != 0      # one template: no continuation form


@pytest.mark.parametrize("nx,ns,K,nl,nr", [(3, 5000, 100, 100, 130), (4, 4096 - 200, 100, 256, 100), (5, 9001, 462, 1024, 1024),
                                            (2, 300, 60, 64, 61)])
def test_fir_fft_between_neighbours(emu, nx, ns, K, nl, nr):
    """d4w_fir_fft_halo_f32: a row with neighbours on both sides -- every column of y, the halos read in place from
    pitched buffers (the left halo's LAST K columns, the right halo's first), equal to the FIR over the concatenation."""
    rng = np.random.default_rng(ns + K)
    x = (rng.standard_normal((nx, ns)) + np.arange(nx)[:, None] * 20.0).astype(np.float32)
    left = (rng.standard_normal((nx, nl + 7)) + np.arange(nx)[:, None] * 20.0).astype(np.float32)     # pitch nl + 7
    right = (rng.standard_normal((nx, nr + 3)) + np.arange(nx)[:, None] * 20.0).astype(np.float32)
    taps = rng.standard_normal(2 * K + 1) * np.hanning(2 * K + 3)[1:-1]
    taps = ((taps + taps[::-1]) / 2).astype(np.float32)
    first = np.ascontiguousarray(x[:, 0])
    y = np.full((nx, ns), np.nan, dtype=np.float32)
    emu.d4w_xcorr_fft_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(emu.d4w_xcorr_fft_ws_bytes(), dtype=np.uint8)
    cv, ci = ctypes.c_void_p, ctypes.c_int
    emu.d4w_fir_fft_halo_f32.argtypes = [cv, ci, ci, cv, ci, ci, cv, ci, ci, cv, ci, cv, ctypes.c_double, cv, cv, cv]
    dcg = float(taps.astype(np.float64).sum())
    lview = left[:, 7:]                                     # the nl samples adjacent to the row
    rc = emu.d4w_fir_fft_halo_f32(vp(x), nx, ns, lview.ctypes.data, left.shape[1], nl, vp(right), right.shape[1], nr, vp(taps), K,
                                  vp(first), dcg, vp(y), vp(ws), None)
    assert rc == 0, emu.d4w_last_error()
    assert not np.isnan(y).any()
    xv = np.concatenate((lview, x, right[:, :nr]), axis=1).astype(np.float64)
    full = np.stack([np.convolve(r, taps.astype(np.float64)[::-1], "valid") for r in xv])       # index m <-> centre m + K of xv
    ref = full[:, nl - K: nl - K + ns]
    assert ref.shape == y.shape
    assert np.max(np.abs(y - ref)) < 1e-5 * np.max(np.abs(ref))
    # halos shorter than the half width, or a missing side: refused
    assert emu.d4w_fir_fft_halo_f32(vp(x), nx, ns, lview.ctypes.data, left.shape[1], K - 1, vp(right), right.shape[1], nr, vp(taps),
                                    K, vp(first), dcg, vp(y), vp(ws), None) != 0
    assert emu.d4w_fir_fft_halo_f32(vp(x), nx, ns, None, 0, 0, vp(right), right.shape[1], nr, vp(taps), K, vp(first), dcg, vp(y),
                                    vp(ws), None) != 0


# ------------------------------------------------------------------------------------------
# matched filter as a Toeplitz product on the matrix cores (csrc/xcorr_mm.hip)
# ------------------------------------------------------------------------------------------
def xcorr_mm_emu(lib, x, taps_list, normalize=True, nxt=None, n_next=0, stats=None):
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    lt = max(4, -(-max(len(t) for t in taps_list) // 4) * 4)
    taps = np.zeros((len(taps_list), lt), dtype=np.float32)
    for i, t in enumerate(taps_list):
        taps[i, :len(t)] = t
    mean = np.empty(nx, dtype=np.float64)          # float64 row means (include/d4w.h)
    mx = np.empty(nx, dtype=np.float32)
    if stats is not None:
        mean, mx = stats
    elif normalize:
        assert lib.d4w_row_stats_f32(vp(xf), nx, ns, vp(mean), vp(mx), None) == 0
    ys = [np.full_like(xf, np.nan) for _ in taps_list]
    rc = lib.d4w_xcorr_mm_f32(vp(xf), nx, ns, vp(nxt) if nxt is not None else None, nxt.shape[1] if nxt is not None else 0, n_next,
                              vp(mean) if normalize else None, vp(mx) if normalize else None,
                              vp(taps), len(taps_list), lt, len(taps_list[0]), len(taps_list[-1]),
                              vp(ys[0]), vp(ys[1]) if len(ys) > 1 else None, None)
    assert rc == 0, lib.d4w_last_error()
    return ys


def test_f16_conversions_of_the_emulator():
    """The emulator's binary16 helpers against NumPy's float16 (round to nearest even, subnormals)."""
    import subprocess, sys, os, textwrap
    src = textwrap.dedent("""
        #include "hip_emu.h"
        extern "C" void conv(const float* x, unsigned short* h, float* back, int n) {
            for (int i = 0; i < n; ++i) { h[i] = hipemu::f32_to_f16(x[i]); back[i] = hipemu::f16_to_f32(h[i]); }
        }""")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    out = os.path.join(here, "_build", "f16conv.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-DD4W_EMU", "-I", here, "-x", "c++", "-", "-o", out],
                   input=src, text=True, check=True)
    lib = ctypes.CDLL(out)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 5, 20000),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e6, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 6.1e-5, 6.0e-5,
                         0.333251953125 + 2.0 ** -13]]).astype(np.float32)
    h = np.empty(x.size, dtype=np.uint16)
    back = np.empty(x.size, dtype=np.float32)
    lib.conv(vp(x), vp(h), vp(back), x.size)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16)
    assert np.array_equal(h, want.view(np.uint16))
    assert np.array_equal(back, want.astype(np.float32))


def test_xcorr_mm_golden(emu, golden):
    d = golden("detect_12x2000.npz")
    yh, yl = xcorr_mm_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])])
    assert rel(yh, d["corr_hf"]) < 2e-6                 # reference detect.compute_cross_correlogram
    assert rel(yl, d["corr_lf"]) < 2e-6
    (y1,) = xcorr_mm_emu(emu, d["x"], [norm_taps(d["lf"])])
    assert rel(y1, d["corr_lf"]) < 2e-6
    zl, zh = xcorr_mm_emu(emu, d["x"], [norm_taps(d["lf"]), norm_taps(d["hf"])])     # longer template first: the 6 + 6 kernel
    assert rel(zh, d["corr_hf"]) < 2e-6 and rel(zl, d["corr_lf"]) < 2e-6
    fh, fl = xcorr_fft_emu(emu, d["x"], [norm_taps(d["hf"]), norm_taps(d["lf"])])
    assert rel(yh, fh.astype(np.float64)) < 3e-6        # matrix-core and FFT forms agree to rounding


@pytest.mark.parametrize("nx,ns,l0,l1", [(3, 9001, 161, 7), (2, 4096, 136, 156), (1, 4289, 1, 177), (4, 1300, 50, 50), (2, 100, 100, 3),
                                          (1, 8192 + 4288, 136, 156), (2, 4500, 241, 20), (1, 5000, 60, 200),
                                          (2, 5100, 450, 300), (1, 4097, 497, 498), (1, 9000, 1024, 243), (2, 700, 700, 5)])
def test_xcorr_mm_ragged(emu, nx, ns, l0, l1):
    """Odd row lengths (unaligned rows: the sample-by-sample loads and stores), chunks with a ragged tail, a row shorter
    than a chunk, the maximum support of the fused kernel (177) and of the 8-step one-template kernel (241: two templates then
    run one after the other), the deeper one-template kernels (12 and 16 k-steps: up to 497 taps in one launch -- the reference
    script's own 450-sample template, scripts/main_mfdetect.py:70 --, longer templates in sections of 496 taps that accumulate),
    a template longer than the row, rows without normalisation (per-chunk power-of-two scale) and with a large offset."""
    rng = np.random.default_rng(ns + l0)
    x = rng.standard_normal((nx, ns)) * 37.0 + 0.5
    t0, t1 = rng.standard_normal(l0) * 5.0, rng.standard_normal(l1) * 0.01
    y0, y1 = xcorr_mm_emu(emu, x, [t0, t1], normalize=False)
    for c in range(nx):
        assert rel(y0[c], orc.shift_xcorr(x[c], np.pad(t0, (0, ns - l0)))) < 2e-6
        assert rel(y1[c], orc.shift_xcorr(x[c], np.pad(t1, (0, ns - l1)))) < 2e-6
    assert emu.d4w_xcorr_mm_max_support() == 16 * 496
    xs = (x + 1000.0).astype(np.float32)
    z0, z1 = xcorr_mm_emu(emu, xs, [t0, t1])
    xn = (xs.astype(np.float64) - xs.astype(np.float64).mean(axis=1, keepdims=True)) / np.abs(xs).max(axis=1, keepdims=True)
    for c in range(nx):
        assert rel(z0[c], orc.shift_xcorr(xn[c], np.pad(t0, (0, ns - l0)))) < 1e-5    # float32 row mean of a 1000x offset
    big = 16 * 496 + 1
    assert emu.d4w_xcorr_mm_f32(vp(xs), nx, ns, None, 0, 0, None, None, vp(np.zeros((1, big + 3), np.float32)), 1, big + 3, big, big,
                                vp(z0), None, None) != 0


def test_xcorr_mm_longest_template_in_sections(emu):
    """The longest template the matrix-core form takes, 16 sections of 496 taps = 7936 (what 'auto' routes there instead of the
    direct form since round 5; measured 3 x faster per tap than the direct FIR at 700 and 1024 taps, profiles/r05g): sixteen
    launches that accumulate into y in float32 -- against float64 at the kernel's usual 2e-6."""
    rng = np.random.default_rng(7936)
    nx, ns, L = 2, 12000, 16 * 496
    x = rng.standard_normal((nx, ns)) + 0.25
    t0 = rng.standard_normal(L)
    (y0,) = xcorr_mm_emu(emu, x, [t0], normalize=False)
    for c in range(nx):
        e = rel(y0[c], orc.shift_xcorr(x[c], np.pad(t0, (0, ns - L))))
        assert e < 2e-6, (c, e)


@pytest.mark.parametrize("nx,ns", [(3, 4100), (2, 9000), (2, 4096 * 2)])
def test_xcorr_mm_continuation(emu, nx, ns):
    """The record continues in xnext: same numbers as correlating [x | head] with x's own statistics."""
    rng = np.random.default_rng(nx * ns)
    x = (rng.standard_normal((nx, ns)) + 0.3).astype(np.float32)
    nxt = (rng.standard_normal((nx, 500)) - 0.2).astype(np.float32)
    t0, t1 = rng.standard_normal(136) * np.hanning(136), rng.standard_normal(156) * np.hanning(156)
    L = 156
    mean, mx = np.empty(nx, dtype=np.float64), np.empty(nx, dtype=np.float32)
    assert emu.d4w_row_stats_f32(vp(x), nx, ns, vp(mean), vp(mx), None) == 0
    cont = xcorr_mm_emu(emu, x, [t0, t1], nxt=nxt, n_next=L - 1, stats=(mean, mx))
    ext = np.ascontiguousarray(np.concatenate((x, nxt[:, :L - 1]), axis=1))
    ref = xcorr_mm_emu(emu, ext, [t0, t1], stats=(mean, mx))
    for c, r in zip(cont, ref):
        assert np.max(np.abs(c - r[:, :ns])) <= 2e-6 * np.max(np.abs(r))
    xa = (np.concatenate((x[0], nxt[0, :L - 1])).astype(np.float64) - float(mean[0])) / float(mx[0])
    for k in (ns - 1, ns - 77, ns - L + 1):
        assert abs(cont[1][0, k] - float(np.dot(xa[k:k + 156], t1))) < 2e-6 * np.max(np.abs(ref[1]))


@pytest.mark.parametrize("offset", [1e3, 1e4, 1e5, 1e6, 3e6])
def test_offset_heavy_rows_are_demeaned_in_two_floats(emu, offset):
    """detect.py:157 de-means in float64.  Rows whose OFFSET is 10^3..10^5 x their signal's deviation, a template with a
    non-zero sum (the mean's error would enter every lag times that sum): the float64 row mean of d4w_row_stats_f32 enters
    every correlator as a two-float value, (x - hi) - lo, and the three forms hold 1e-5 against a float64 reference where a
    float32 mean left 1.2e-5 .. 2.8e-5 at 1000 x (round 4's known limit).  Input = the float32 rows both sides see."""
    rng = np.random.default_rng(int(offset) + 11)
    nx, ns, L = 3, 9000, 137
    sigma = 0.37
    x = (rng.standard_normal((nx, ns)) * sigma + offset * sigma * np.array([1.0, -1.0, 0.731])[:, None]).astype(np.float32)
    tpl = np.abs(rng.standard_normal(L)) + 0.3                       # sum(tpl) ~ 150: nothing cancels a mean error
    x64 = x.astype(np.float64)
    mean = np.empty(nx, dtype=np.float64)
    mx = np.empty(nx, dtype=np.float32)
    assert emu.d4w_row_stats_f32(vp(x), nx, ns, vp(mean), vp(mx), None) == 0
    # the mean itself: float64-grade (1e-9 of the deviation), not float32 of the offset
    assert np.max(np.abs(mean - x64.mean(axis=1))) < 1e-7 * sigma
    assert np.array_equal(mx, np.abs(x).max(axis=1))
    xn = (x64 - x64.mean(axis=1, keepdims=True)) / np.abs(x64).max(axis=1, keepdims=True)
    ref = np.stack([orc.shift_xcorr(r, np.pad(tpl, (0, ns - L))) for r in xn])
    (ym,) = xcorr_mm_emu(emu, x, [tpl])
    (yf,) = xcorr_fft_emu(emu, x, [tpl])
    (yd,), _, _ = xcorr_emu(emu, x, [tpl])
    for name, y in (("mm", ym), ("fft", yf), ("direct", yd)):
        e = max(rel(y[c], ref[c]) for c in range(nx))
        # measured 1.6e-7 .. 4.6e-7 at every offset: rows that are all offset scale their chunks by their own power of two
        # (1 / max|x| alone left the matrix-core form's samples in binary16's subnormal range: 8e-7 at 10^5, 2.3e-5 at 10^6)
        assert e < 2e-6, (name, offset, e)


@pytest.mark.parametrize("l0,l1", [(136, 156), (450, 20), (700, 3), (1, 1)])
def test_xcorr_mm_row_maxima_from_the_epilogue(emu, l0, l1):
    """d4w_xcorr_mm_rowmax_f32: max over the lags of every correlogram row out of the kernel's epilogue (two integer atomics
    standing in for a float max) = the maximum of what was stored -- fused two-template launch, one-template launches, a
    template in sections (only the last section's sums count), rows whose correlogram is negative throughout, ragged row ends."""
    rng = np.random.default_rng(l0 * 7 + l1)
    nx, ns = 5, 9001
    x = np.ascontiguousarray(rng.standard_normal((nx, ns)) + 0.3, dtype=np.float32)
    t0, t1 = rng.standard_normal(l0), rng.standard_normal(l1)
    if l0 == 1:                                   # x > 0 against a negative tap: every lag is negative
        x = np.abs(x) + 0.1
        t0, t1 = np.array([-0.7]), np.array([0.4])
    lt = max(4, -(-max(l0, l1) // 4) * 4)
    taps = np.zeros((2, lt), dtype=np.float32)
    taps[0, :l0], taps[1, :l1] = t0, t1
    y0, y1 = np.full_like(x, np.nan), np.full_like(x, np.nan)
    m0, m1 = np.full(nx, np.nan, np.float32), np.full(nx, np.nan, np.float32)
    rc = emu.d4w_xcorr_mm_rowmax_f32(vp(x), nx, ns, None, 0, 0, None, None, vp(taps), 2, lt, l0, l1, vp(y0), vp(y1), vp(m0), vp(m1), None)
    assert rc == 0, emu.d4w_last_error()
    assert np.array_equal(m0, y0.max(axis=1)) and np.array_equal(m1, y1.max(axis=1))
    if l0 == 1:
        assert np.all(m0 < 0)
    z0, z1 = xcorr_mm_emu(emu, x, [t0, t1], normalize=False)
    assert np.array_equal(z0, y0) and np.array_equal(z1, y1)             # the plain entry point: same values
    assert emu.d4w_xcorr_mm_rowmax_f32(vp(x), nx, ns, None, 0, 0, None, None, vp(taps), 2, lt, l0, l1, vp(y0), vp(y1), vp(m0), None, None) != 0


def xcorr_mm_tail_emu(lib, x, templates, want_max=False):
    """d4w_xcorr_mm_tail_f32 the way detect.compute_cross_correlograms drives it: the templates' supports, tail coefficients
    mean(t) / max|t| over the zero-padded length."""
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nx, ns = xf.shape
    taps_list, coefs = [], []
    for tpl in templates:
        tp, c = norm_taps(tpl), float(tpl.mean() / np.max(np.abs(tpl)))
        taps_list.append(tp)
        coefs.append(c)
    lt = max(4, -(-max(len(t) for t in taps_list) // 4) * 4)
    taps = np.zeros((len(taps_list), lt), dtype=np.float32)
    for i, t in enumerate(taps_list):
        taps[i, :len(t)] = t
    mean, mx = np.empty(nx, dtype=np.float64), np.empty(nx, dtype=np.float32)
    assert lib.d4w_row_stats_f32(vp(xf), nx, ns, vp(mean), vp(mx), None) == 0
    ys = [np.full_like(xf, np.nan) for _ in taps_list]
    rm = [np.full(nx, np.nan, np.float32) for _ in taps_list] if want_max else [None] * 2
    rc = lib.d4w_xcorr_mm_tail_f32(vp(xf), nx, ns, None, 0, 0, vp(mean), vp(mx), vp(taps), len(taps_list), lt, len(taps_list[0]),
                                   len(taps_list[-1]), ctypes.c_double(coefs[0]), ctypes.c_double(coefs[-1] if len(coefs) > 1 else 0.0),
                                   vp(ys[0]), vp(ys[1]) if len(ys) > 1 else None,
                                   vp(rm[0]) if want_max else None, vp(rm[1]) if want_max and len(ys) > 1 else None, None)
    assert rc == 0, lib.d4w_last_error()
    return (ys, rm) if want_max else ys


def test_zero_padded_template_tail_inside_the_matrix_core_correlator(emu):
    """Round 6: the constant tail of the de-meaned zero-padded template (detect.py:158) is added in the correlator's own epilogue
    (inside a block of 16 lags as part of the Toeplitz product itself, one prefix per block from a scan in the conversion phase,
    the prefix at a chunk's start carried along the row): the whole of detect.compute_cross_correlogram
    in one pass, on EVERY row -- white, band-limited, a slow drift, a step (the rows a white-noise prediction let through at
    3e-4 .. 2e-3 in rounds 1-4) -- for rows of several chunks, ragged row ends, both fin-call templates in one launch, one
    template alone, a template with a clearly non-zero mean, supports that are not multiples of 4."""
    rng = np.random.default_rng(606)
    fs, ns = 200.0, 13001                       # 4 chunks of 4096 lags, the last one ragged
    t = np.arange(ns) / fs
    hf = orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    lf = orc.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
    w = rng.standard_normal((4, ns))
    band = sps.sosfiltfilt(sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos"), rng.standard_normal(ns))
    x = np.stack([w[0], band, 0.05 * w[1] + np.sin(2 * np.pi * t / 50.0), np.where(t < 40, 1.0, -1.0) + 0.01 * w[2],
                  w[3] + 3.0]) + 0.3
    xf = np.ascontiguousarray(x, dtype=np.float32).astype(np.float64)          # the float32 rows the kernel sees
    refs = [orc.compute_cross_correlogram(xf, tp) for tp in (hf, lf)]
    (yh, yl), (mh, ml) = xcorr_mm_tail_emu(emu, x, [hf, lf], want_max=True)
    for y, ref, name in ((yh, refs[0], "hf"), (yl, refs[1], "lf")):
        for k in range(len(x)):
            e = np.max(np.abs(y[k] - ref[k])) / np.max(np.abs(ref[k]))
            assert e < 2e-6, (name, k, e)
    assert np.array_equal(mh, yh.max(axis=1)) and np.array_equal(ml, yl.max(axis=1))     # maxima of the FINAL values
    # without the term the drifting rows are far off (what the term is worth)
    plain = xcorr_mm_emu(emu, x, [norm_taps(hf)])[0]
    assert np.max(np.abs(plain[2] - refs[0][2])) / np.max(np.abs(refs[0][2])) > 1e-4
    # one template alone (the one-template kernels), a support that is not a multiple of 4, a clearly non-zero mean
    tpl = np.zeros(ns)
    tpl[:203] = np.abs(rng.standard_normal(203)) + 0.2
    (y1,) = xcorr_mm_tail_emu(emu, x, [tpl])
    ref1 = orc.compute_cross_correlogram(xf, tpl)
    for k in range(len(x)):
        e = np.max(np.abs(y1[k] - ref1[k])) / np.max(np.abs(ref1[k]))
        assert e < 2e-6, ("one template", k, e)
    (y2,) = xcorr_mm_tail_emu(emu, x, [lf])
    assert np.max(np.abs(y2 - refs[1])) / np.max(np.abs(refs[1])) < 2e-6
    # the contract: the rows' statistics
    taps = np.zeros((1, 8), np.float32)
    y = np.empty((5, ns), np.float32)
    xs = np.ascontiguousarray(x, dtype=np.float32)
    assert emu.d4w_xcorr_mm_tail_f32(vp(xs), 5, ns, None, 0, 0, None, None, vp(taps), 1, 8, 8, 8, ctypes.c_double(1e-3),
                                     ctypes.c_double(0.0), vp(y), None, None, None, None) == -1


def test_row_maximum_of_a_row_with_nan_is_nan(emu):
    """np.max propagates NaN (scripts/main_mfdetect.py:82 takes the threshold from it); the epilogue's float max used to drop it
    (ADVICE r05): a row that holds a NaN sample now leaves NaN as its row maximum, the other rows theirs."""
    rng = np.random.default_rng(9)
    nx, ns = 3, 9000
    x = np.ascontiguousarray(rng.standard_normal((nx, ns)), dtype=np.float32)
    x[1, 4500] = np.nan
    t0, t1 = rng.standard_normal(136), rng.standard_normal(156)
    taps = np.zeros((2, 156), dtype=np.float32)
    taps[0, :136], taps[1, :156] = t0, t1
    y0, y1 = np.empty_like(x), np.empty_like(x)
    m0, m1 = np.empty(nx, np.float32), np.empty(nx, np.float32)
    assert emu.d4w_xcorr_mm_rowmax_f32(vp(x), nx, ns, None, 0, 0, None, None, vp(taps), 2, 156, 136, 156, vp(y0), vp(y1), vp(m0), vp(m1), None) == 0
    assert np.isnan(m0[1]) and np.isnan(m1[1]) and np.isnan(y0[1]).any()
    for k in (0, 2):
        assert m0[k] == y0[k].max() and m1[k] == y1[k].max()


@pytest.mark.parametrize("order,nx,ns", [(10, 13, 700), (9, 5, 130), (3, 67, 1205), (8, 1, 64)])
def test_sos_sections_on_adjacent_lanes(emu, order, nx, ns):
    """sos_pass_lanes (one exact segment per row: the sections of a row on adjacent lanes, outputs handed on by a DPP move):
    8 lanes per row up to 8 sections, 16 for 9-10; row counts that leave lanes, waves and workgroups ragged; rows shorter
    than one staged chunk and rows of many chunks -- against scipy.signal.sosfiltfilt (odd extension, steady-state zi)."""
    rng = np.random.default_rng(order * 100 + nx)
    fs = 200.0
    x = rng.standard_normal((nx, ns)) + rng.standard_normal((nx, 1)) * 5
    sos = sps.butter(order, [20 / (fs / 2), 45 / (fs / 2)], "bp", output="sos")
    assert sos.shape[0] == order
    padlen = min(3 * (2 * order + 1), ns - 1)
    ref = sps.sosfiltfilt(sos, x, axis=1, padlen=padlen)
    assert rel(sosfiltfilt_emu(emu, x, sos, padlen=padlen), ref) < TOL


@pytest.mark.parametrize("phases", [(0,), (1, 2)])
def test_row_end_pieces_in_place(emu, phases):
    """d4w_sosfiltfilt_ends_f32: the left and right `piece` samples of every row filtered as rows of their own (filtfilt's edge
    rule at both ends of a piece) read in place from x, their `keep` outer outputs written in place into y -- equal to
    scipy.signal.sosfiltfilt of the gathered pieces; everything else of y untouched; as one call or as forward / backward phases."""
    rng = np.random.default_rng(77)
    nx, ns, piece, keep, padlen, fs = 11, 1000, 300, 120, 51, 200.0
    x = np.ascontiguousarray(rng.standard_normal((nx, ns)) + rng.standard_normal((nx, 1)) * 4, dtype=np.float32)
    sos = np.ascontiguousarray(sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos"))
    zi = np.ascontiguousarray(sps.sosfilt_zi(sos))
    y = np.full_like(x, 7.5)
    emu.d4w_sosfiltfilt_ends_ws_bytes.restype = ctypes.c_size_t
    ws = np.empty(emu.d4w_sosfiltfilt_ends_ws_bytes(nx, piece, padlen), dtype=np.uint8)
    for ph in phases:
        rc = emu.d4w_sosfiltfilt_ends_f32(vp(x), vp(y), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, piece, keep, ph, vp(ws), None)
        assert rc == 0, emu.d4w_last_error()
    left = sps.sosfiltfilt(sos, x[:, :piece].astype(np.float64), axis=1, padlen=padlen)
    right = sps.sosfiltfilt(sos, x[:, ns - piece:].astype(np.float64), axis=1, padlen=padlen)
    scale = np.max(np.abs(left))
    assert np.max(np.abs(y[:, :keep] - left[:, :keep])) < TOL * scale
    assert np.max(np.abs(y[:, ns - keep:] - right[:, piece - keep:])) < TOL * scale
    assert np.all(y[:, keep:ns - keep] == 7.5)
    assert emu.d4w_sosfiltfilt_ends_f32(vp(x), vp(x), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, piece, keep, 0, vp(ws), None) != 0
    # one end only (the first / last file of a stream): exactly that side's columns of the two-sided result, the rest untouched
    for sides, sl in ((1, slice(0, keep)), (2, slice(ns - keep, ns))):
        y1 = np.full_like(x, 7.5)
        for ph in phases:
            rc = emu.d4w_sosfiltfilt_ends_sides_f32(vp(x), vp(y1), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, piece, keep, ph, sides,
                                                    vp(ws), None)
            assert rc == 0, emu.d4w_last_error()
        assert np.array_equal(y1[:, sl], y[:, sl])
        mask = np.ones(ns, dtype=bool)
        mask[sl] = False
        assert np.all(y1[:, mask] == 7.5)
    assert emu.d4w_sosfiltfilt_ends_sides_f32(vp(x), vp(y), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, piece, keep, 0, 4, vp(ws), None) != 0
