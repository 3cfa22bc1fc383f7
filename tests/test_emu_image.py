"""Image operators (image.hip) in the CPU emulator build against the oracle -- kernel logic only;
the GPU parity tests are tests/test_image_gpu.py."""
import ctypes

import numpy as np
import pytest

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5
sz = ctypes.c_size_t


@pytest.fixture(scope="module")
def emu():
    return load_emu()


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / max(np.max(np.abs(ref)), 1e-300))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def test_minmax_scale_threshold_mask(emu):
    rng = np.random.default_rng(0)
    x = f32(rng.standard_normal(5003) * 7 - 2)
    mm = np.zeros(2, dtype=np.float32)
    assert emu.d4w_minmax_f32(vp(x), sz(x.size), vp(mm), None) == 0
    assert mm[0] == x.min() and mm[1] == x.max()
    y = np.empty_like(x)
    assert emu.d4w_scale_pixels_f32(vp(x), vp(y), sz(x.size), vp(mm), ctypes.c_double(255.0), None) == 0
    assert rel(y, orc.scale_pixels(x.astype(np.float64)) * 255) < TOL
    assert emu.d4w_threshold_f32(vp(x), vp(y), sz(x.size), ctypes.c_double(0.25), None) == 0
    assert np.array_equal(y != 0, x.astype(np.float64) > 0.25)
    m = f32(rng.random(x.size) > 0.5)
    assert emu.d4w_mask_mul_f32(vp(x), vp(m), vp(y), sz(x.size), None) == 0
    assert np.array_equal(y, x * m)
    neg = f32([-3.0, -1.0, -2.0])
    assert emu.d4w_minmax_f32(vp(neg), sz(3), vp(mm), None) == 0 and mm[0] == -3 and mm[1] == -1
    # np.min / np.max propagate NaN (the threshold 0.45 * np.max(corr) of scripts/main_mfdetect.py:82 is NaN then, and picks nothing):
    # both forms of the reduction (one workgroup for small inputs, the three-launch form) do too
    for n in (700, 300 * 1200):
        bad = rng.random(n).astype(np.float32) - 0.5
        bad[n // 3] = np.nan
        assert emu.d4w_minmax_f32(vp(bad), sz(n), vp(mm), None) == 0 and np.isnan(mm[0]) and np.isnan(mm[1]), (n, mm)
        bad[n // 3] = 2.0
        assert emu.d4w_minmax_f32(vp(bad), sz(n), vp(mm), None) == 0 and mm[0] == bad.min() and mm[1] == 2.0


@pytest.mark.parametrize("h,w,oh,ow", [(40, 120, 4, 12), (37, 101, 3, 10), (4, 12, 40, 120), (50, 33, 17, 80), (9, 300, 9, 30)])
def test_resize_bilinear_aa(emu, h, w, oh, ow):
    rng = np.random.default_rng(h * w)
    x = f32(rng.standard_normal((h, w)))
    y = np.empty((oh, ow), dtype=np.float32)
    emu.d4w_resize_ws_bytes.restype = sz
    ws = np.empty(emu.d4w_resize_ws_bytes(h, w, oh, ow), dtype=np.uint8)
    assert emu.d4w_resize_bilinear_aa_f32(vp(x), h, w, vp(y), oh, ow, vp(ws), None) == 0, emu.d4w_last_error()
    assert rel(y, orc.resize_bilinear_aa(x.astype(np.float64), oh, ow)) < TOL


@pytest.mark.parametrize("h,w,kh,kw", [(24, 160, 101, 101), (70, 130, 5, 9), (3, 4, 7, 7), (33, 65, 1, 1), (40, 70, 4, 6), (9, 600, 3, 113),
                                       (21, 300, 5, 115), (6, 2100, 2, 33)])
def test_filter2d(emu, h, w, kh, kw):
    """Kernels of <= 113 columns run on the matrix cores (filter2d_mm.hip: one wave per 256 columns, 4 output rows per
    workgroup, ragged tiles, more than 8 tiles per row -> several workgroups per row block), wider ones the direct form."""
    rng = np.random.default_rng(kh * 100 + kw)
    img = f32(rng.standard_normal((h, w)))
    ker = f32(rng.standard_normal((kh, kw)))
    if (kh, kw) == (101, 101):
        up, down = orc.gabor_filt_design(42.56)
        ker = f32(up)
    out = np.empty_like(img)
    emu.d4w_filter2d_ws_bytes.restype = sz
    ws = np.empty(emu.d4w_filter2d_ws_bytes(kh, kw), dtype=np.uint8)
    assert emu.d4w_filter2d_f32(vp(img), h, w, vp(ker), kh, kw, vp(out), 0, vp(ws), None) == 0, emu.d4w_last_error()
    ref = orc.filter2d(img.astype(np.float64), ker.astype(np.float64))
    assert rel(out, ref) < TOL
    assert emu.d4w_filter2d_f32(vp(img), h, w, vp(ker), kh, kw, vp(out), 1, vp(ws), None) == 0      # accumulate
    assert rel(out, 2 * ref) < TOL


def test_filter2d_rejects_oversized_kernel_and_inplace(emu):
    img = np.zeros((8, 8), dtype=np.float32)
    ker = np.zeros((301, 301), dtype=np.float32)
    ws = np.empty(4 * 301 * 320, dtype=np.uint8)
    assert emu.d4w_filter2d_f32(vp(img), 8, 8, vp(ker), 301, 301, vp(np.empty_like(img)), 0, vp(ws), None) != 0
    assert emu.d4w_filter2d_f32(vp(img), 8, 8, vp(ker), 3, 3, vp(img), 0, vp(ws), None) != 0


def test_analytic_env_over_std(emu):
    rng = np.random.default_rng(5)
    x = f32(rng.standard_normal((5, 400)) * rng.uniform(0.5, 3, (5, 1)))
    var = np.empty(5, dtype=np.float32)
    y = np.empty_like(x)
    assert emu.d4w_row_var_f32(vp(x), 5, 400, vp(var), None) == 0
    assert emu.d4w_analytic_f32(vp(x), vp(y), 5, 400, 4, vp(var), ctypes.c_double(0.0), None) == 0, emu.d4w_last_error()
    x64 = x.astype(np.float64)
    assert rel(y, np.abs(orc.hilbert(x64)) / np.std(x64, axis=1, keepdims=True)) < TOL
