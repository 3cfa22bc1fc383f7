"""Time-first order of the shape-specialised f-k passes (das4whales_amd/csrc/fk_tf.h) on the CPU emulator: the same
filter as the channel-first order and as the oracle, with the half spectrum compacted to the frequency columns the mask
needs (band columns through the channel transform, wavenumber-independent "tail" columns scaled in place, zero columns
dropped).  Reference: dsp.fk_filter_filt dsp.py:725-756; hybrid_ninf_filter_design dsp.py:308-454."""
import ctypes
import os

import numpy as np
import pytest

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5


@pytest.fixture(scope="module")
def emu():
    lib = load_emu()
    lib.d4w_fk_plan_order.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    return lib


def run(lib, x, mask, order=None, taper=0, stats=False):
    nx, ns = x.shape
    old = os.environ.get("D4W_FK_ORDER")
    if order:
        os.environ["D4W_FK_ORDER"] = order
    try:
        plan = ctypes.c_void_p()
        assert lib.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0, lib.d4w_last_error()
        m = np.ascontiguousarray(mask, dtype=np.float32)
        xf = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(xf)
        assert lib.d4w_fk_set_mask_dense_f32(plan, vp(m), None) == 0, lib.d4w_last_error()
        info, by = (ctypes.c_int * 6)(), (ctypes.c_double * 2)()
        assert lib.d4w_fk_plan_order(plan, info, by) == 0
        if stats:
            mean = np.empty(nx, dtype=np.float64)
            mx = np.empty(nx, dtype=np.float32)
            lib.d4w_fk_apply_stats_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
            assert lib.d4w_fk_apply_stats_f32(plan, vp(xf), vp(y), taper, vp(mean), vp(mx), None) == 0, lib.d4w_last_error()
        else:
            assert lib.d4w_fk_apply_f32(plan, vp(xf), vp(y), taper, None) == 0, lib.d4w_last_error()
        # a second call on the same plan (workspace reuse)
        y2 = np.empty_like(xf)
        assert lib.d4w_fk_apply_f32(plan, vp(xf), vp(y2), taper, None) == 0
        assert np.array_equal(y, y2, equal_nan=True)
        lib.d4w_fk_plan_destroy(plan)
    finally:
        if order:
            if old is None:
                del os.environ["D4W_FK_ORDER"]
            else:
                os.environ["D4W_FK_ORDER"] = old
    res = (y, list(info), list(by))
    return res + ((mean, mx),) if stats else res


def rel(y, ref):
    return np.max(np.abs(y - ref)) / np.max(np.abs(ref))


def band_mask(rng, nx, ns, lo, hi, tail_to, tail_gain=1e-3):
    """Shifted-grid mask: random gains for lo <= |f bin| < hi, a wavenumber-independent skirt up to tail_to, zero beyond."""
    m = np.zeros((nx, ns))
    fb = np.abs(np.fft.fftshift(np.arange(ns) - ns * (np.arange(ns) >= (ns + 1) // 2)))       # |f| bin of every shifted column
    band = (fb >= lo) & (fb < hi)
    m[:, band] = rng.random((nx, int(band.sum())))
    tail = (fb >= hi) & (fb < tail_to)
    skirt = tail_gain * np.exp(-(fb - hi) / 7.0)
    m[:, tail] = skirt[tail][None, :]
    return m


@pytest.mark.parametrize("nx,ns", [(18, 48), (8, 480), (100, 600), (154, 48), (1102, 48)])
def test_time_first_equals_oracle_dense_mask(emu, nx, ns):
    """Every column alive and wavenumber-dependent: time-first is the same five transforms in another order."""
    rng = np.random.default_rng(nx * 7 + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))                          # arbitrary non-Hermitian mask
    ref = orc.fk_filter_filt(x, m)
    y, info, _ = run(emu, x, m, order="tf")
    assert info[0] == 1 and info[1] == ns // 2 + 1 and info[2] == 0
    assert rel(y, ref) < TOL
    yt, _, _ = run(emu, x, m, order="tf", taper=1)
    assert rel(yt, orc.fk_filter_filt(x, m, tapering=True)) < TOL


@pytest.mark.parametrize("nx,ns,lo,hi,tail_to", [(100, 600, 20, 70, 130), (100, 600, 0, 40, 301), (8, 480, 30, 90, 150),
                                                 (100, 600, 100, 140, 140), (18, 48, 0, 3, 9)])
def test_band_tail_dead_columns(emu, nx, ns, lo, hi, tail_to):
    """Band + wavenumber-independent skirt + zero columns: chosen automatically, fewer columns kept than the half
    spectrum, same output as the oracle and as the channel-first order."""
    rng = np.random.default_rng(ns + lo)
    x = rng.standard_normal((nx, ns))
    m = band_mask(rng, nx, ns, lo, hi, tail_to)
    ref = orc.fk_filter_filt(x, m)
    y_tf, info, by = run(emu, x, m, order="tf")
    assert info[0] == 1
    assert rel(y_tf, ref) < TOL
    y_cf, info_cf, _ = run(emu, x, m, order="cf")
    assert info_cf[0] == 0 and rel(y_cf, ref) < TOL
    y_auto, info_auto, by = run(emu, x, m)
    assert rel(y_auto, ref) < TOL
    assert info_auto[0] == (1 if by[1] < 0.97 * by[0] else 0)
    if tail_to < ns // 2 and ns >= 480:
        assert info[1] + info[2] < ns // 2 and (info[2] > 0) == (tail_to > hi)      # zero columns dropped, skirt kept as tail columns


def test_pure_time_filter_and_nyquist(emu):
    """A mask that does not depend on the wavenumber at all (a 1-D filter along time): no band columns, the channel
    transform is skipped altogether; with and without the Nyquist column."""
    rng = np.random.default_rng(5)
    nx, ns = 100, 600
    x = rng.standard_normal((nx, ns))
    g = rng.random(ns // 2 + 1)
    for nyq in (0.0, 0.7):
        g[-1] = nyq
        full = np.concatenate((g[:-1], g[:0:-1]))                  # unshifted, even in f; index ns/2 = Nyquist
        m = np.tile(np.fft.fftshift(full)[None, :], (nx, 1))
        ref = orc.fk_filter_filt(x, m)
        y, info, _ = run(emu, x, m, order="tf")
        assert info[0] == 1 and info[1] == (1 if nyq else 0)
        assert rel(y, ref) < TOL


def test_row_stats_epilogue_in_time_first_order(emu):
    rng = np.random.default_rng(9)
    nx, ns = 100, 600
    x = rng.standard_normal((nx, ns))
    m = band_mask(rng, nx, ns, 10, 60, 100)
    y, info, _, (mean, mx) = run(emu, x, m, order="tf", stats=True)
    assert info[0] == 1
    ref = orc.fk_filter_filt(x, m)
    assert rel(y, ref) < TOL
    assert np.max(np.abs(mean - y.mean(axis=1))) < 1e-5 * np.max(np.abs(y))
    assert np.allclose(mx, np.abs(y).max(axis=1), rtol=1e-6)


def test_hybrid_ninf_design_runs_time_first(emu):
    """The scripts' design on a small block: the looped columns are band columns, the Butterworth skirts tail columns."""
    nx, ns, fs, dx = 100, 600, 200.0, 2.0419046878814697
    rng = np.random.default_rng(3)
    x = rng.standard_normal((nx, ns))
    m = orc.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
    m = np.asarray(m.todense() if hasattr(m, "todense") else m)
    ref = orc.fk_filter_filt(x, m)
    y, info, by = run(emu, x, m)
    assert rel(y, ref) < TOL
    y_tf, info_tf, _ = run(emu, x, m, order="tf")
    assert info_tf[0] == 1 and info_tf[2] > 0 and rel(y_tf, ref) < TOL


def test_degenerate_masks(emu):
    """All-zero mask (nothing kept in either order: the output is exactly zero), a single live frequency column, and a
    mask whose only non-zero entry is the Nyquist column."""
    rng = np.random.default_rng(17)
    nx, ns = 100, 600
    x = rng.standard_normal((nx, ns))
    m = np.zeros((nx, ns))
    for order in ("tf", "cf", None):
        y, info, _ = run(emu, x, m, order=order)
        assert np.all(y == 0.0)
    m1 = np.zeros((nx, ns))
    m1[:, ns // 2 + 37] = rng.random(nx) + 0.5                     # f = +37 bins only (the fold adds the mirrored half)
    ref = orc.fk_filter_filt(x, m1)
    y, info, _ = run(emu, x, m1, order="tf")
    assert info[0] == 1 and info[1] <= 20 and rel(y, ref) < TOL     # one cell of N1 NA = 20 columns kept
    m2 = np.zeros((nx, ns))
    m2[:, 0] = rng.random(nx) + 0.5                                # shifted column 0 = the Nyquist frequency
    ref = orc.fk_filter_filt(x, m2)
    y, info, _ = run(emu, x, m2, order="tf")
    assert info[0] == 1 and info[1] == 1 and rel(y, ref) < TOL
    y, _, _ = run(emu, x, m2)
    assert rel(y, ref) < TOL


def test_nan_gain_spreads_like_numpy(emu):
    """A NaN in the mask makes the reference's product NaN and the inverse transform spreads it over the block; the
    column holding it stays a band column here (never dropped as 'zero'), so the output is NaN as well."""
    rng = np.random.default_rng(19)
    nx, ns = 18, 48
    x = rng.standard_normal((nx, ns))
    m = band_mask(rng, nx, ns, 2, 9, 15)
    m[3, ns // 2 + 4] = np.nan
    for order in ("tf", "cf"):
        y, _, _ = run(emu, x, m, order=order)
        assert np.isnan(y).any()
