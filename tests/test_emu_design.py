"""Mask-design kernels (design.hip) on the CPU emulator build vs the golden masks produced by the
real reference and vs the oracle's closed forms."""
import ctypes

import numpy as np
import pytest
import scipy.signal as sps

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

ARGS = dict(cs_min=1350., cp_min=1450., cp_max=3300., cs_max=3450., fmin=14., fmax=30.)


@pytest.fixture(scope="module")
def emu():
    lib = load_emu()
    lib.d4w_design_mask_f32.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]
    lib.d4w_gaussian_filter_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    lib.d4w_minmax_normalise_f32.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return lib


def design(lib, mode, shape, step_dx, fs, params, i0=0, i1=0, hrow=None):
    nx, ns = shape
    out = np.empty((nx, ns), dtype=np.float32)
    p8 = np.zeros(8)
    p8[:len(params)] = params
    h = np.ascontiguousarray(hrow, dtype=np.float64) if hrow is not None else None
    rc = lib.d4w_design_mask_f32(mode, nx, ns, step_dx, 1.0 / fs, vp(p8), i0, i1, vp(h) if h is not None else None,
                                 vp(out), None)
    assert rc == 0, lib.d4w_last_error()
    return out


def gauss(lib, m, sigma=20.0):
    out, tmp = np.empty_like(m), np.empty_like(m)
    assert lib.d4w_gaussian_filter_f32(vp(m), vp(out), vp(tmp), m.shape[0], m.shape[1], sigma, None) == 0
    return out


def first_ge(f, v):
    return int(np.argmax(f >= v))


def test_designs_vs_reference_golden(emu, golden):
    g = golden("fk_40x480.npz")
    shape, sel, dx, fs = g["x"].shape, list(g["sel"]), float(g["dx"]), float(g["fs"])
    step = sel[2] * dx
    f = np.fft.fftshift(np.fft.fftfreq(shape[1], d=1 / fs))
    A = ARGS
    m0 = design(emu, 0, shape, step, fs, [1400, 1450, 3400, 3500])
    assert np.max(np.abs(m0 - g["m_classic"])) < 2e-7
    m1 = design(emu, 1, shape, step, fs, [A["cs_min"], A["cp_min"], A["fmin"], A["fmax"]],
                first_ge(f, A["fmin"] - 4), first_ge(f, A["fmax"] + 4))
    assert np.max(np.abs(m1 - g["m_hybrid"])) < 2e-7
    b, a = sps.butter(8, [A["fmin"] / (fs / 2), A["fmax"] / (fs / 2)], "bp")
    H = np.concatenate((np.zeros(shape[1] // 2), np.abs(sps.freqz(b, a, worN=shape[1] // 2)[1]) ** 2))
    m2 = design(emu, 2, shape, step, fs, [A["cs_min"], A["cp_min"], A["cp_max"], A["cs_max"]],
                first_ge(f, A["fmin"] - 14), first_ge(f, A["fmax"] + 14), H)
    assert np.max(np.abs(m2 - g["m_ninf"])) < 5e-7
    m3 = gauss(emu, design(emu, 3, shape, step, fs, [A["cs_min"], A["cp_min"], A["fmin"], A["fmax"]],
                           first_ge(f, A["fmin"] - 4), first_ge(f, A["fmax"] + 4)))
    assert np.max(np.abs(m3 - g["m_gs"])) < 2e-6


def test_gaussian_and_normalise(emu):
    rng = np.random.default_rng(1)
    from scipy import ndimage
    # smaller than the radius: multiple reflections; >= 2048 along an axis: the overlap-save FFT form with reflected halos
    for shape in [(37, 201), (100, 64), (9, 1100), (30, 4200), (2100, 24)]:
        m = rng.random(shape).astype(np.float32)
        assert np.max(np.abs(gauss(emu, m, 20.0) - ndimage.gaussian_filter(m.astype(np.float64), 20))) < 2e-6
        assert np.max(np.abs(gauss(emu, m, 3.0) - ndimage.gaussian_filter(m.astype(np.float64), 3))) < 2e-6
    x = (rng.random((13, 77)) * 5 - 2).astype(np.float32)
    ref = (x - x.min()) / (x.max() - x.min())
    assert emu.d4w_minmax_normalise_f32(vp(x), x.size, None) == 0
    assert np.max(np.abs(x - ref)) < 1e-6


@pytest.mark.parametrize("nx,ns,opts", [(40, 480, None), (18, 48, None), (38, 406, [19, 2, 7, 29, 4, 4]), (100, 600, None),
                                        (7, 14, None)])
def test_design_folded_straight_into_the_plan(emu, nx, ns, opts):
    """d4w_fk_set_mask_design_f32 (closed form evaluated in pass-B order, no dense mask) leaves exactly what
    d4w_design_mask_f32 + d4w_fk_set_mask_dense_f32 leave: the filtered blocks agree bit for bit, the liveness too."""
    emu.d4w_fk_set_mask_design_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    emu.d4w_fk_set_mask_dense_pruned_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    fs, step = 200.0, 2.0419046878814697
    f = np.fft.fftshift(np.fft.fftfreq(ns, d=1 / fs))
    A = ARGS
    b, a = sps.butter(8, [A["fmin"] / (fs / 2), A["fmax"] / (fs / 2)], "bp")
    H = np.concatenate((np.zeros(ns // 2), np.abs(sps.freqz(b, a, worN=ns // 2)[1]) ** 2))
    cases = [(0, [1400, 1450, 3400, 3500], 0, 0, None, 0.0),
             (1, [A["cs_min"], A["cp_min"], A["fmin"], A["fmax"]], first_ge(f, A["fmin"] - 4), first_ge(f, A["fmax"] + 4), None, 0.0),
             (2, [A["cs_min"], A["cp_min"], A["cp_max"], A["cs_max"]], first_ge(f, A["fmin"] - 14), first_ge(f, A["fmax"] + 14), H, 0.0),
             (2, [A["cs_min"], A["cp_min"], A["cp_max"], A["cs_max"]], first_ge(f, A["fmin"] - 14), first_ge(f, A["fmax"] + 14), H, 4e-6)]
    x = np.random.default_rng(nx * ns).standard_normal((nx, ns)).astype(np.float32)
    o = (ctypes.c_int * 6)(*opts) if opts else None
    for mode, params, i0, i1, hrow, eps in cases:
        dense = design(emu, mode, (nx, ns), step, fs, params, i0, i1, hrow)
        p8 = np.zeros(8)
        p8[:len(params)] = params
        h = np.ascontiguousarray(hrow, dtype=np.float64) if hrow is not None else None
        ys, live = [], []
        for analytic in (False, True):
            plan = ctypes.c_void_p()
            assert emu.d4w_fk_plan_create_ex(nx, ns, o, ctypes.byref(plan)) == 0, emu.d4w_last_error()
            if analytic:
                rc = emu.d4w_fk_set_mask_design_f32(plan, mode, step, 1.0 / fs, vp(p8), i0, i1, vp(h) if h is not None else None,
                                                    eps, None)
            else:
                rc = emu.d4w_fk_set_mask_dense_pruned_f32(plan, vp(dense), eps, None)
            assert rc == 0, emu.d4w_last_error()
            y = np.empty_like(x)
            assert emu.d4w_fk_apply_f32(plan, vp(x), vp(y), 0, None) == 0, emu.d4w_last_error()
            live.append(emu.d4w_fk_plan_live_rows(plan))
            emu.d4w_fk_plan_destroy(plan)
            ys.append(y)
        assert np.array_equal(ys[0], ys[1]), (mode, eps)
        assert live[0] == live[1]
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    p8 = np.zeros(8)
    assert emu.d4w_fk_set_mask_design_f32(plan, 3, step, 1.0 / fs, vp(p8), 0, 0, None, 0.0, None) != 0    # blurred designs: refused
    assert b"closed form" in emu.d4w_last_error()
    emu.d4w_fk_plan_destroy(plan)


@pytest.mark.parametrize("nx,ns,opts", [(40, 480, None), (18, 48, None), (38, 406, [19, 2, 7, 29, 4, 4]), (100, 600, None)])
def test_normalisation_folded_into_the_mask_upload(emu, nx, ns, opts):
    """d4w_minmax_f32 (image.hip) + d4w_fk_set_mask_dense_affine_f32 == d4w_minmax_normalise_f32 + d4w_fk_set_mask_dense_f32, bit for
    bit (dsp.fk_filt's (g - min) / (max - min), dsp.py:945, without the separate pass over the mask)."""
    cv, ci = ctypes.c_void_p, ctypes.c_int
    emu.d4w_minmax_f32.argtypes = [cv, ctypes.c_size_t, cv, cv]
    emu.d4w_fk_set_mask_dense_affine_f32.argtypes = [cv, cv, ctypes.c_float, ctypes.c_float, cv]
    rng = np.random.default_rng(nx + ns)
    g = (rng.random((nx, ns)) * 3.0 - 0.7).astype(np.float32)
    x = rng.standard_normal((nx, ns)).astype(np.float32)
    o = (ci * 6)(*opts) if opts else None
    lohi = np.zeros(2, dtype=np.float32)
    assert emu.d4w_minmax_f32(vp(g), g.size, vp(lohi), None) == 0
    assert lohi[0] == g.min() and lohi[1] == g.max()
    a = np.float32(1.0) / (lohi[1] - lohi[0])
    b = -lohi[0] * a
    ys = []
    for fused in (False, True):
        plan = cv()
        assert emu.d4w_fk_plan_create_ex(nx, ns, o, ctypes.byref(plan)) == 0, emu.d4w_last_error()
        if fused:
            assert emu.d4w_fk_set_mask_dense_affine_f32(plan, vp(g), ctypes.c_float(a), ctypes.c_float(b), None) == 0
        else:
            gn = g.copy()
            assert emu.d4w_minmax_normalise_f32(vp(gn), gn.size, None) == 0
            assert emu.d4w_fk_set_mask_dense_f32(plan, vp(gn), None) == 0
        y = np.empty_like(x)
        assert emu.d4w_fk_apply_f32(plan, vp(x), vp(y), 0, None) == 0, emu.d4w_last_error()
        emu.d4w_fk_plan_destroy(plan)
        ys.append(y)
    assert np.array_equal(ys[0], ys[1])
