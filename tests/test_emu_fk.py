"""f-k filter kernel LOGIC on the CPU emulator build (tests/emu/hip_emu.h): same HIP sources,
same C ABI, host pointers.  Complements -- never replaces -- the -m gpu parity tests."""
import ctypes

import numpy as np
import pytest

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5   # north-star tolerance: max|y - ref| <= 1e-5 * max|ref| in float32


@pytest.fixture(scope="module")
def emu():
    return load_emu()


def fk_emu(lib, x, mask, opts=None, taper=0):
    nx, ns = x.shape
    plan = ctypes.c_void_p()
    o = (ctypes.c_int * 6)(*opts) if opts else None
    rc = lib.d4w_fk_plan_create_ex(nx, ns, o, ctypes.byref(plan))
    assert rc == 0, lib.d4w_last_error()
    m = np.ascontiguousarray(mask, dtype=np.float32)
    xf = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(xf)
    assert lib.d4w_fk_set_mask_dense_f32(plan, vp(m), None) == 0
    assert lib.d4w_fk_apply_f32(plan, vp(xf), vp(y), taper, None) == 0, lib.d4w_last_error()
    lib.d4w_fk_plan_destroy(plan)
    return y


def rel(y, ref):
    return np.max(np.abs(y - ref)) / np.max(np.abs(ref))


def test_golden_masks(emu, golden):
    g = golden("fk_40x480.npz")
    x = g["x"]
    assert rel(fk_emu(emu, x, g["m_classic"]), g["y_classic"]) < TOL
    assert rel(fk_emu(emu, x, g["m_ninf"], opts=[8, 5, 6, 40, 8, 7]), g["y_ninf"]) < TOL     # non-Hermitian mask
    assert rel(fk_emu(emu, x, g["m_classic"], opts=[4, 10, 8, 30, 7, 16], taper=1), g["y_classic_taper"]) < TOL


def test_odd_channels_and_prime_radices(emu):
    rng = np.random.default_rng(0)
    # every prime radix 7..31 of the generic kernels (fft_lds.h lds_stage_prime_t), forward and inverse, on both axes
    for nx, ns, opts in [(38, 406, [19, 2, 7, 29, 4, 4]), (7, 14, None), (1, 64, None), (64, 2, None), (26, 102, None),
                         (46, 62, None), (62, 52, None), (34, 46, None), (11 * 23, 2 * 13 * 17, None), (31 * 3, 2 * 29 * 2, None)]:
        x = rng.standard_normal((nx, ns))
        m = rng.random((nx, ns))
        assert rel(fk_emu(emu, x, m, opts), orc.fk_filter_filt(x, m)) < TOL


def test_in_place(emu, golden):
    g = golden("fk_30x360.npz")
    nx, ns = g["x"].shape
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    m = np.ascontiguousarray(g["m_ninf"], dtype=np.float32)
    buf = np.ascontiguousarray(g["x"], dtype=np.float32)
    assert emu.d4w_fk_set_mask_dense_f32(plan, vp(m), None) == 0
    assert emu.d4w_fk_apply_f32(plan, vp(buf), vp(buf), 0, None) == 0
    emu.d4w_fk_plan_destroy(plan)
    assert rel(buf, g["y_ninf"]) < TOL


@pytest.mark.parametrize("nx,ns", [(18, 48), (8, 480), (100, 600), (154, 48), (1102, 48)])
def test_shape_specialised_kernels(emu, nx, ns):
    """Shapes in fk_filter.hip's kFastShapes run the fat-stage register-FFT kernels (fk_fast.h);
    opts[0] = -1 forces the generic passes on the same shape: both must match the oracle."""
    rng = np.random.default_rng(nx + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))                 # arbitrary non-Hermitian mask
    ref = orc.fk_filter_filt(x, m)
    y_fast = fk_emu(emu, x, m)
    assert rel(y_fast, ref) < TOL
    y_gen = fk_emu(emu, x, m, opts=[-1, 0, 0, 0, 0, 0])
    assert rel(y_gen, ref) < TOL
    assert rel(fk_emu(emu, x, m, taper=1), orc.fk_filter_filt(x, m, tapering=True)) < TOL
    info = (ctypes.c_int * 8)()
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    emu.d4w_fk_plan_info(plan, info)
    emu.d4w_fk_plan_destroy(plan)
    assert list(info)[:2] == [nx, ns]


@pytest.mark.parametrize("nx,ns", [(18, 48), (100, 600)])
def test_dead_row_pruning(emu, nx, ns):
    """Masks with all-zero wavenumber rows: the specialised path skips those rows in passes C, B
    and C' (exact: they would be multiplied by zero).  Same result as the unpruned run."""
    rng = np.random.default_rng(5 * nx + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    ks = np.fft.fftshift(np.arange(nx))               # shifted-grid row -> wavenumber index
    kabs = np.minimum(ks, nx - ks)
    m[kabs > nx // 5, :] = 0.0                        # keep low |k| only (what a speed fan does)
    m[nx // 2 + 2, ns // 3] = 0.7                     # one-sided live entry: its partner row must stay live too
    ref = orc.fk_filter_filt(x, m)
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    mf = np.ascontiguousarray(m, dtype=np.float32)
    xf = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(xf)
    assert emu.d4w_fk_set_mask_dense_f32(plan, vp(mf), None) == 0
    live = emu.d4w_fk_plan_live_rows(plan)
    assert 0 < live < nx
    assert emu.d4w_fk_apply_f32(plan, vp(xf), vp(y), 0, None) == 0
    assert rel(y, ref) < TOL
    # all-zero mask: nothing is live, output is exactly zero
    z = np.zeros_like(mf)
    assert emu.d4w_fk_set_mask_dense_f32(plan, vp(z), None) == 0
    assert emu.d4w_fk_plan_live_rows(plan) == 0
    assert emu.d4w_fk_apply_f32(plan, vp(xf), vp(y), 0, None) == 0
    assert np.all(y == 0)
    # back to a dense mask on the same plan: pruning switches off again
    md = np.ascontiguousarray(rng.random((nx, ns)), dtype=np.float32)
    assert emu.d4w_fk_set_mask_dense_f32(plan, vp(md), None) == 0
    assert emu.d4w_fk_plan_live_rows(plan) == nx
    assert emu.d4w_fk_apply_f32(plan, vp(xf), vp(y), 0, None) == 0
    assert rel(y, orc.fk_filter_filt(x, md)) < TOL
    emu.d4w_fk_plan_destroy(plan)


@pytest.mark.parametrize("nx,ns,opts", [(18, 48, None), (8, 480, None), (100, 600, None), (40, 480, None), (100, 600, [-1, 0, 0, 0, 0, 0])])
def test_apply_with_row_statistics(emu, nx, ns, opts):
    """d4w_fk_apply_stats_f32: mean / max|.| of every filtered row from the last pass's epilogue
    (shape-specialised kernels, several tiles per run) or from the separate row pass (generic)."""
    rng = np.random.default_rng(nx * ns)
    x = (rng.standard_normal((nx, ns)) + 0.7).astype(np.float32)
    m = np.ascontiguousarray(rng.random((nx, ns)), dtype=np.float32)
    plan = ctypes.c_void_p()
    o = (ctypes.c_int * 6)(*opts) if opts else None
    assert emu.d4w_fk_plan_create_ex(nx, ns, o, ctypes.byref(plan)) == 0, emu.d4w_last_error()
    assert emu.d4w_fk_set_mask_dense_f32(plan, vp(m), None) == 0
    y0, y1 = np.empty_like(x), np.empty_like(x)
    mean = np.full(nx, np.nan, dtype=np.float64)          # float64 row means (include/d4w.h)
    mx = np.full(nx, np.nan, dtype=np.float32)
    assert emu.d4w_fk_apply_f32(plan, vp(x), vp(y0), 0, None) == 0
    assert emu.d4w_fk_apply_stats_f32(plan, vp(x), vp(y1), 0, vp(mean), vp(mx), None) == 0, emu.d4w_last_error()
    emu.d4w_fk_plan_destroy(plan)
    assert np.array_equal(y0, y1)                       # same filter output, bit for bit
    y64 = y1.astype(np.float64)
    assert np.allclose(mx, np.abs(y64).max(axis=1), rtol=1e-6)
    assert np.max(np.abs(mean - y64.mean(axis=1))) < 1e-6 * np.abs(y64).max()


@pytest.mark.parametrize("nx,ns", [(37, 48), (74, 40), (41 * 6, 24), (127, 16), (2 * 3 * 43, 64), (211, 6)])
def test_channel_count_with_large_prime_factor(emu, nx, ns):
    """nx with a prime factor > 31: the c2 sub-transform of pass C runs as a Bluestein convolution
    (fk_passC_bluestein); everything else is the ordinary five-pass scheme."""
    rng = np.random.default_rng(nx)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    assert rel(fk_emu(emu, x, m), orc.fk_filter_filt(x, m)) < TOL
    assert rel(fk_emu(emu, x, np.ones((nx, ns))), x) < TOL
    assert rel(fk_emu(emu, x, m, taper=1), orc.fk_filter_filt(x, m, tapering=True)) < TOL


def test_channel_count_too_long_for_bluestein_tile(emu, monkeypatch):
    """A prime channel count > 4096 does not fit pass C's Bluestein tile: the plan runs the global-memory form (the generic
    distributed plan at world 1 behind the same entry points; here: affine mask fold, taper, row statistics in ONE apply --
    the emulator walks 4099 workgroups per pass; both axes beyond their tiles at once:
    tests/test_emu_fk_dist.py::test_dist_record_length_with_large_prime_factor, tests/test_fuzz_gpu.py)."""
    monkeypatch.setenv("D4W_FKD_BZ_CHUNK", "3")
    nx, ns = 4099, 8
    rng = np.random.default_rng(5)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    info = (ctypes.c_int * 8)()
    assert emu.d4w_fk_plan_info(plan, info) == 0 and list(info)[:2] == [nx, ns]
    assert emu.d4w_fk_plan_live_rows(plan) == nx
    xf = x.astype(np.float32)
    y = np.empty_like(xf)
    assert emu.d4w_fk_apply_f32(plan, vp(xf), vp(y), 0, None) != 0 and b"no mask" in emu.d4w_last_error()
    mf = m.astype(np.float32)
    a, b = np.float32(0.75), np.float32(0.125)
    assert emu.d4w_fk_set_mask_dense_affine_f32(plan, vp(mf), ctypes.c_float(a), ctypes.c_float(b), None) == 0
    mean = np.empty(nx, dtype=np.float64)
    mx = np.empty(nx, dtype=np.float32)
    assert emu.d4w_fk_apply_stats_f32(plan, vp(xf), vp(y), 1, vp(mean), vp(mx), None) == 0, emu.d4w_last_error()
    emu.d4w_fk_plan_destroy(plan)
    ref = orc.fk_filter_filt(x, (mf * a + b).astype(np.float64), tapering=True)
    assert rel(y, ref) < TOL
    assert np.max(np.abs(mean - y.astype(np.float64).mean(axis=1))) < 1e-6 * np.abs(y).max()
    assert np.allclose(mx, np.abs(y.astype(np.float64)).max(axis=1), rtol=1e-6)


@pytest.mark.parametrize("nx,ns", [(2, 5), (6, 15), (9, 27)])
def test_odd_record_length_by_zero_interleaving(emu, nx, ns):
    """What dsp._fk_apply_odd does for odd ns: z[2n] = x[n], z[2n+1] = 0 filtered with the mask repeated twice
    along f (time axis back on the unshifted grid) returns y interleaved with zeros."""
    rng = np.random.default_rng(ns)
    x, m = rng.standard_normal((nx, ns)), rng.random((nx, ns))
    mu = np.roll(m, -(ns // 2), axis=1)
    x2 = np.zeros((nx, 2 * ns))
    x2[:, ::2] = x
    y2 = fk_emu(emu, x2, np.concatenate((mu, mu), axis=1))
    ref = orc.fk_filter_filt(x, m)
    assert rel(y2[:, ::2], ref) < TOL
    assert np.max(np.abs(y2[:, 1::2])) < TOL * np.max(np.abs(ref))


@pytest.mark.parametrize("nx,ns,sw", [(18, 48, 1), (18, 48, 3), (100, 600, 1), (100, 600, 5), (8, 480, 2)])
def test_slab_ordered_passes(emu, nx, ns, sw, monkeypatch):
    """D4W_FK_SLAB: passes A/C and C'/A' run slab by slab (sw column blocks of every n1 sub-row per slab) so that
    the intermediate stays in the Infinity Cache; same tiles, same arithmetic -> bit-identical output, incl. the row
    statistics epilogue and dead-row skipping."""
    rng = np.random.default_rng(nx + ns + sw)
    x = np.ascontiguousarray(rng.standard_normal((nx, ns)) + 0.3, dtype=np.float32)
    m = rng.random((nx, ns))
    ks = np.fft.fftshift(np.arange(nx))
    m[np.minimum(ks, nx - ks) > nx // 4, :] = 0.0
    m = np.ascontiguousarray(m, dtype=np.float32)
    outs = []
    for env in (None, str(sw)):
        if env is None:
            monkeypatch.delenv("D4W_FK_SLAB", raising=False)
        else:
            monkeypatch.setenv("D4W_FK_SLAB", env)
        plan = ctypes.c_void_p()
        assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
        assert emu.d4w_fk_set_mask_dense_f32(plan, vp(m), None) == 0
        y, ys = np.empty_like(x), np.empty_like(x)
        mean, mx = np.empty(nx, np.float64), np.empty(nx, np.float32)
        assert emu.d4w_fk_apply_f32(plan, vp(x), vp(y), 1, None) == 0, emu.d4w_last_error()
        assert emu.d4w_fk_apply_stats_f32(plan, vp(x), vp(ys), 1, vp(mean), vp(mx), None) == 0, emu.d4w_last_error()
        emu.d4w_fk_plan_destroy(plan)
        assert np.array_equal(y, ys)
        outs.append((y, mean, mx))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][2], outs[1][2])
    assert np.allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-6 * np.abs(outs[0][0]).max())
    assert rel(outs[1][0], orc.fk_filter_filt(x.astype(np.float64), m.astype(np.float64), tapering=True)) < TOL


@pytest.mark.parametrize("nx,ns,opts", [(18, 48, None), (100, 600, None), (40, 480, None), (38, 406, [19, 2, 7, 29, 4, 4]),
                                        (7, 14, None), (1, 64, None), (64, 2, None), (37, 48, None)])
def test_tiled_mask_fold_equals_gather(emu, nx, ns, opts, monkeypatch):
    """fk_fold_mask_tiled (coalesced runs + LDS transpose) against the plain gather kernel: the folded masks, hence
    the filter outputs, are bit-identical."""
    rng = np.random.default_rng(nx * 3 + ns)
    x, m = rng.standard_normal((nx, ns)), rng.random((nx, ns))
    monkeypatch.setenv("D4W_FK_FOLD_GATHER", "1")
    y0 = fk_emu(emu, x, m, opts)
    monkeypatch.delenv("D4W_FK_FOLD_GATHER")
    y1 = fk_emu(emu, x, m, opts)
    assert np.array_equal(y0, y1)
    assert rel(y1, orc.fk_filter_filt(x, m)) < TOL


def test_opt_in_tail_pruning(emu):
    """d4w_fk_set_mask_dense_pruned_f32: rows whose folded gains stay below eps * max count as dead; eps = 0 is the
    exact filter; the pruned output equals the filter with those rows zeroed."""
    nx, ns = 100, 600
    rng = np.random.default_rng(9)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    ks = np.fft.fftshift(np.arange(nx))
    tail = np.minimum(ks, nx - ks) > nx // 5
    m[tail, :] *= 1e-6                                  # "Butterworth tails": tiny but non-zero everywhere
    emu.d4w_fk_set_mask_dense_pruned_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    plan = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(plan)) == 0
    mf, xf = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(x, np.float32)
    y = np.empty_like(xf)
    assert emu.d4w_fk_set_mask_dense_pruned_f32(plan, vp(mf), 0.0, None) == 0
    assert emu.d4w_fk_plan_live_rows(plan) == nx
    assert emu.d4w_fk_set_mask_dense_pruned_f32(plan, vp(mf), 4e-6, None) == 0
    live = emu.d4w_fk_plan_live_rows(plan)
    assert live == nx - int(tail.sum())
    assert emu.d4w_fk_apply_f32(plan, vp(xf), vp(y), 0, None) == 0
    mz = m.copy()
    mz[tail, :] = 0.0
    assert rel(y, orc.fk_filter_filt(x, mz)) < TOL
    assert rel(y, orc.fk_filter_filt(x, m)) < TOL      # and within tolerance of the exact filter on this input
    assert emu.d4w_fk_set_mask_dense_pruned_f32(plan, vp(mf), -1.0, None) != 0
    emu.d4w_fk_plan_destroy(plan)


@pytest.mark.parametrize("nx,ns", [(12, 74), (20, 246), (74, 86), (6, 12002), (9, 2 * 5 * 67)])
def test_record_length_with_large_prime_factor(emu, nx, ns):
    """ns / 2 with a prime factor > 31 (numpy.fft.fft2 at dsp.py:748 takes any length): the n2 sub-transforms of pass B
    run as Bluestein convolutions (fk_passB_bluestein); 12002 = 2 x 17 x 353 is a 60-s file cut two samples long."""
    rng = np.random.default_rng(nx + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    assert rel(fk_emu(emu, x, m), orc.fk_filter_filt(x, m)) < TOL
    assert rel(fk_emu(emu, x, np.ones((nx, ns))), x) < TOL
    assert rel(fk_emu(emu, x, m, taper=1), orc.fk_filter_filt(x, m, tapering=True)) < TOL


@pytest.mark.parametrize("nx,ns,chunk", [(8, 2 * 4099, 0), (5, 2 * 2 * 2053, 3), (6, 2 * 6007, 4)])
def test_record_length_too_long_for_bluestein_tile(emu, nx, ns, chunk, monkeypatch):
    """ns / 2 whose part with prime factors > 31 exceeds 2048 (pass B's Bluestein tile): the time transform of the packed rows
    runs as a Bluestein convolution in global memory (fkd_bt_*), a chunk of rows at a time (D4W_FKD_BT_CHUNK pins a chunk
    smaller than the block, with a ragged last one); 12014 = 2 x 6007 is a 60-s file cut 14 samples long."""
    if chunk:
        monkeypatch.setenv("D4W_FKD_BT_CHUNK", str(chunk))
    rng = np.random.default_rng(nx + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    assert rel(fk_emu(emu, x, m), orc.fk_filter_filt(x, m)) < TOL
    assert rel(fk_emu(emu, x, m, taper=1), orc.fk_filter_filt(x, m, tapering=True)) < TOL
    if chunk == 3:
        assert rel(fk_emu(emu, x, np.ones((nx, ns))), x) < TOL


def test_fuzz_generic_kernels_with_prime_radices(emu):
    """Random shapes whose axes are products of the radices 2..31 through the generic kernels (opts[0] = -1), with and without
    the taper: every lds_stage_prime_t<R> in forward and inverse direction, batch-fast and contiguous order."""
    rng = np.random.default_rng(7)
    small = [2, 3, 4, 5, 6, 7, 8, 10, 11, 13, 17, 19, 23, 29, 31]
    done = 0
    while done < 16:
        nx = int(np.prod(rng.choice(small, size=int(rng.integers(1, 4)))))
        M = int(np.prod(rng.choice(small, size=int(rng.integers(1, 4)))))
        if nx > 1200 or M > 1200:
            continue
        x = rng.standard_normal((nx, 2 * M))
        m = rng.uniform(0, 1, (nx, 2 * M))
        taper = int(rng.integers(0, 2))
        y = fk_emu(emu, x, m, opts=[-1, 0, 0, 0, 0, 0], taper=taper)
        assert rel(y, orc.fk_filter_filt(x, m, tapering=bool(taper))) < TOL, (nx, 2 * M)
        done += 1


@pytest.mark.parametrize("ns", [1, 2, 3, 5, 64, 67, 480, 12000])
def test_taper_touches_the_ramps_only_and_equals_the_full_product(emu, ns):
    """d4w_taper_f32 multiplies the two cosine ramps of tukey(ns, 0.03) and leaves the flat part alone (x * 1.0f is x): bit for bit
    the float32 product with the whole window, for any length incl. the degenerate ones, NaN / inf / -0 in the flat part untouched,
    and the reference's literal vector (tests/test_dsp.py:85-88)."""
    import scipy.signal.windows as sw
    rng = np.random.default_rng(ns)
    x = (rng.standard_normal((3, ns)) * 1e3).astype(np.float32)
    if ns >= 64:
        x[1, ns // 2] = np.nan
        x[2, ns // 2] = -0.0
        x[0, ns // 2 + 1] = np.inf
    want = x * sw.tukey(ns, 0.03).astype(np.float32)[None, :]
    y = x.copy()
    assert emu.d4w_taper_f32(vp(y), 3, ns, None) == 0, emu.d4w_last_error()
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
    if ns == 5:
        v = np.array([[1.0, 2.0, 3.0, 4.0, 5.0]], dtype=np.float32)
        assert emu.d4w_taper_f32(vp(v), 1, 5, None) == 0
        assert np.array_equal(v, [[0.0, 2.0, 3.0, 4.0, 0.0]])
