"""GPU parity tests of the f-k mask designers (dsp.fk_filter_design, hybrid_*_filter_design) and of
the self-designing dsp.fk_filt, vs the masks / outputs of the real reference (golden fixtures)."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
ARGS = dict(cs_min=1350., cp_min=1450., cp_max=3300, cs_max=3450, fmin=14., fmax=30.)


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available()
    import das4whales_amd as dw_
    return dw_


def test_designs_golden(dw, golden):
    g = golden("fk_40x480.npz")
    shape, sel, dx, fs = g["x"].shape, list(g["sel"]), float(g["dx"]), float(g["fs"])
    m = dw.dsp.fk_filter_design(shape, sel, dx, fs)
    assert m.shape == shape and np.max(np.abs(np.asarray(m) - g["m_classic"])) < 5e-7
    mh = dw.dsp.hybrid_filter_design(shape, sel, dx, fs, cs_min=1350., cp_min=1450., fmin=14., fmax=30.)
    assert np.max(np.abs(mh.todense() - g["m_hybrid"])) < 5e-7
    mn = dw.dsp.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS)
    assert np.max(np.abs(mn.todense() - g["m_ninf"])) < 1e-6
    mg = dw.dsp.hybrid_gs_filter_design(shape, sel, dx, fs, cs_min=1350., cp_min=1450., fmin=14., fmax=30.)
    assert np.max(np.abs(mg.todense() - g["m_gs"])) < 3e-6
    mng = dw.dsp.hybrid_ninf_gs_filter_design(shape, sel, dx, fs, **ARGS)
    assert np.max(np.abs(mng.todense() - g["m_ninf_gs"])) < 3e-6
    # sparse.COO duck typing used by tools.disp_comprate (tools.py:248) and dsp.py:784
    assert mn.data.ndim == 1 and len(mn.data) == np.count_nonzero(mn.todense()) == mn.nnz
    # design -> apply without leaving the device, as the scripts chain them (main_mfdetect.py:46-55)
    x = g["x"]
    assert rel(dw.dsp.fk_filter_sparsefilt(x, mn), g["y_ninf"]) < 1e-5
    assert rel(dw.dsp.fk_filter_filt(x, m), g["y_classic"]) < 1e-5
    assert rel(dw.dsp.fk_filter_sparsefilt(x, mng), g["y_ninf_gs"]) < 1e-5
    with pytest.raises(ValueError):
        dw.dsp.hybrid_ninf_filter_design((40, 481), sel, dx, fs)


def test_fk_filt_golden(dw, golden):
    g = golden("fk_40x480.npz")
    y = dw.dsp.fk_filt(g["x"], 1, float(g["fs"]), 4, float(g["dx"]), 1400., 3400.)
    assert rel(y, g["y_fkfilt"]) < 1e-5


def test_designs_config1_shape_vs_oracle(dw):
    shape, sel, dx, fs = (4000, 12000), [9794, 25794, 4], 2.0419046878814697, 200.0
    m = dw.dsp.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS).todense()
    ref = orc.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS)
    assert np.max(np.abs(m - ref)) < 1e-6
    del m, ref
    mc = dw.dsp.fk_filter_design(shape, sel, dx, fs).todense()
    assert np.max(np.abs(mc - orc.fk_filter_design(shape, sel, dx, fs))) < 1e-6


def test_gaussian_designs_fft_blur_vs_oracle(dw):
    """Shapes whose axes are longer than an FFT block: the sigma-20 blur runs as overlap-save passes with reflected halos
    (+ transposes for the channel axis); *_gs designs and fk_filt against the oracle (scipy.ndimage.gaussian_filter)."""
    shape, sel, dx, fs = (2500, 6000), [0, 10000, 4], 2.0419046878814697, 200.0
    mg = dw.dsp.hybrid_gs_filter_design(shape, sel, dx, fs, cs_min=1350., cp_min=1450., fmin=14., fmax=30.).todense()
    ref = orc.hybrid_gs_filter_design(shape, sel, dx, fs, cs_min=1350., cp_min=1450., fmin=14., fmax=30.)
    e = float(np.max(np.abs(mg - ref)))
    print("hybrid_gs_filter_design 2500 x 6000 (FFT blur): max abs err %.3e" % e)
    assert e < 3e-6
    del mg, ref
    mng = dw.dsp.hybrid_ninf_gs_filter_design(shape, sel, dx, fs, **ARGS).todense()
    assert np.max(np.abs(mng - orc.hybrid_ninf_gs_filter_design(shape, sel, dx, fs, **ARGS))) < 3e-6
    del mng
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2100, 4200))                     # odd sizes: partial transpose tiles, partial FFT blocks
    y = dw.dsp.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)
    assert rel(y, orc.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)) < 1e-5


def test_design_goes_into_the_plan_without_a_dense_mask(dw):
    """BASELINE configs[3] block: fk_filter_design / hybrid_ninf_filter_design return the closed form (DesignedMask);
    the filter folds it straight into the plan.  No 9.6 GB mask is allocated, and the result is bit-identical to
    designing the dense mask and folding that."""
    import time
    nx, ns, dx, fs = 20000, 120000, 2.0419046878814697, 200.0
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((nx, ns), device="cuda", generator=gen)
    plan = dw.dsp.get_fk_plan(nx, ns)
    for name, mk, eps in (("classic", lambda: dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs), 0.0),
                          ("hybrid_ninf", lambda: dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, **ARGS), 0.0),
                          ("hybrid_ninf pruned", lambda: dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, **ARGS), 4e-6)):
        m = mk()
        assert isinstance(m, dw.dsp.DesignedMask) and m._tensor is None
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        plan.set_mask(m, prune_eps=eps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        assert m._tensor is None                                           # still no dense mask
        assert torch.cuda.max_memory_allocated() - base < 64 << 20
        live = plan.live_rows()
        y = plan.apply(x)
        dense = m.tensor                                                   # now materialise it (9.6 GB) and fold that
        t0 = time.perf_counter()
        plan.set_mask(dense, prune_eps=eps)
        torch.cuda.synchronize()
        ms_dense = (time.perf_counter() - t0) * 1e3
        assert plan.live_rows() == live
        y2 = plan.apply(x)
        assert torch.equal(y, y2), name
        print("%s: design -> plan %.1f ms (dense mask -> plan %.1f ms), %d live rows" % (name, ms, ms_dense, live))
        del y, y2, dense, m


def test_fk_filt_reuses_its_mask_and_notices_other_masks(dw, golden):
    """dsp.fk_filt designs its mask from (shape, tint, fs, xint, dx, c_min, c_max) only: a second call with the same arguments
    reuses the plan's folded mask (bit-identical output), other arguments or another mask on the same plan in between
    rebuild it."""
    g = golden("fk_40x480.npz")
    x = np.asarray(g["x"])
    fs, dx = float(g["fs"]), float(g["dx"])
    y1 = dw.dsp.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)
    y2 = dw.dsp.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)
    assert np.array_equal(y1, y2)
    y3 = dw.dsp.fk_filt(x, 1, fs, 4, dx, 1500., 3000.)
    assert not np.array_equal(y1, y3)
    dw.dsp.fk_filter_filt(x, np.ones_like(x))                  # another mask through the same cached plan
    y4 = dw.dsp.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)
    assert np.array_equal(y1, y4)
    ref = orc.fk_filt(x, 1, fs, 4, dx, 1400., 3400.)
    assert rel(y4, ref) < 1e-5
