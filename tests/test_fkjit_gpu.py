"""Shape-specialised f-k kernels compiled on demand (das4whales_amd/fkjit.py): the same templates as the built-in
shapes, instantiated for a configuration the chooser picks; parity against the oracle, the generic kernels and the
closed-form answers of tests/known_answers.py."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc
from tests import known_answers as ka

pytestmark = pytest.mark.gpu
TOL = 1e-5
DX, FS = 2.0419046878814697, 200.0


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import das4whales_amd as dw_
    return dw_


def test_small_shape_against_oracle(dw):
    nx, ns = 96, 480
    assert dw.dsp.compile_fk_shape(nx, ns)
    from das4whales_amd import fkjit
    assert fkjit.is_specialised(nx, ns) and not fkjit.is_specialised(nx + 2, ns)
    rng = np.random.default_rng(1)
    x, m = rng.standard_normal((nx, ns)), rng.uniform(size=(nx, ns))
    ks = np.fft.fftshift(np.arange(nx))
    m[np.minimum(ks, nx - ks) > 30, :] = 0.0                    # dead rows: pruned by the specialised path
    dw.dsp.clear_fk_plans()
    plan = dw.dsp.get_fk_plan(nx, ns)
    info = plan.info()
    assert info["C1"] * info["C2"] == nx and info["N1"] * info["N2"] == ns // 2
    y = dw.dsp.fk_filter_filt(x, m, tapering=True)
    ref = orc.fk_filter_filt(x, m, tapering=True)
    assert np.max(np.abs(y - ref)) < TOL * np.max(np.abs(ref))
    assert 0 < plan.live_rows() < nx
    gen = dw.dsp.FkPlan(nx, ns, opts=(-1, 0, 0, 0, 0, 0))
    gen.set_mask(m)
    xt = torch.from_numpy(x).float().cuda()
    yg = gen.apply(xt, taper=True).cpu().numpy()
    assert np.max(np.abs(y - yg)) < 3e-6 * np.max(np.abs(ref))


def test_file_shape_8000_channels(dw):
    """8000 x 12000 (a channel selection without built-in kernels): compiled configuration vs the plane-wave answers,
    and faster than the generic passes."""
    nx, ns = 8000, 12000
    assert dw.dsp.compile_fk_shape(nx, ns)
    shape, sel = (nx, ns), [0, nx * 4, 4]
    mask = dw.dsp.hybrid_ninf_filter_design(shape, sel, DX, FS, 1350., 1450., 3300, 3450, 14., 30.)
    at = lambda i, j: orc.hybrid_ninf_filter_design_at(shape, sel, DX, FS, i, j, cs_min=1350., cp_min=1450., cp_max=3300,
                                                       cs_max=3450, fmin=14., fmax=30.)
    dw.dsp.clear_fk_plans()
    plan = dw.dsp.get_fk_plan(nx, ns)
    plan.set_mask(mask)
    rng = np.random.default_rng(8)
    kx, kt, amp, ph = ka.pick_plane_waves(nx, ns, sel, DX, FS, rng)
    g = orc.folded_gain_at(at, shape, kx, kt)
    A, B = ka.wave_factors(nx, ns, kx, kt, ph)
    torch.set_float32_matmul_precision("highest")
    Ad, Bd = torch.from_numpy(A.astype(np.float32)).cuda(), torch.from_numpy(B.astype(np.float32)).cuda()
    x = (Ad * torch.from_numpy(np.tile(amp, 2).astype(np.float32)).cuda()) @ Bd
    ref = (Ad * torch.from_numpy(np.tile(amp * g, 2).astype(np.float32)).cuda()) @ Bd
    y = plan.apply(x)
    assert float((y - ref).abs().max()) < TOL * float(ref.abs().max())
    gen = dw.dsp.FkPlan(nx, ns, opts=(-1, 0, 0, 0, 0, 0))
    gen.set_mask(mask)

    def timed(p):
        p.apply(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            p.apply(x)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 5
    t_jit, t_gen = timed(plan), timed(gen)
    print("8000 x 12000: compiled configuration %.3f ms, generic passes %.3f ms" % (t_jit, t_gen))
    assert t_jit < t_gen


def test_channel_count_with_large_prime_factor(dw):
    """74 = 2 x 37 channels and the 13223 = 7 x 1889 channels of scripts/main_mfdetect.py's selection: compiled passes A and
    B around the generic Bluestein pass C (configuration with C2X > 1), against the oracle / the generic plan."""
    nx, ns = 74, 480
    assert dw.dsp.compile_fk_shape(nx, ns)
    rng = np.random.default_rng(74)
    x, m = rng.standard_normal((nx, ns)), rng.uniform(size=(nx, ns))
    dw.dsp.clear_fk_plans()
    y = dw.dsp.fk_filter_filt(x, m, tapering=True)
    ref = orc.fk_filter_filt(x, m, tapering=True)
    assert np.max(np.abs(y - ref)) < TOL * np.max(np.abs(ref))
    m[::3] = 0.0                                          # all-zero rows: this path prunes nothing, still exact
    y = dw.dsp.fk_filter_filt(x, m)
    ref = orc.fk_filter_filt(x, m)
    assert np.max(np.abs(y - ref)) < TOL * np.max(np.abs(ref))
    nx, ns = 13223, 12000
    assert dw.dsp.compile_fk_shape(nx, ns)
    dw.dsp.clear_fk_plans()
    gen_ = torch.Generator(device="cuda").manual_seed(5)
    xt = torch.randn((nx, ns), device="cuda", generator=gen_)
    mt = torch.rand((nx, ns), device="cuda", generator=gen_)
    plan = dw.dsp.get_fk_plan(nx, ns)
    plan.set_mask(mt)
    gen = dw.dsp.FkPlan(nx, ns, opts=(-1, 0, 0, 0, 0, 0))
    gen.set_mask(mt)
    y1, y2 = plan.apply(xt), gen.apply(xt)
    assert float((y1 - y2).abs().max()) < 3e-6 * float(y2.abs().max())

    def timed(p):
        p.apply(xt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            p.apply(xt)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 5
    t_jit, t_gen = timed(plan), timed(gen)
    print("13223 x 12000: compiled A / B + Bluestein C %.3f ms, generic passes %.3f ms" % (t_jit, t_gen))
    assert t_jit < t_gen
