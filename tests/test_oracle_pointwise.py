"""Pins the oracle's pointwise mask evaluators (used by the full-size known-answer GPU tests, where the
dense float64 mask does not fit) against the golden-pinned full designs, and the plane-wave identity the
GPU tests rely on against the oracle's f-k filter.  CPU only."""
import functools

import numpy as np
import pytest

from oracle import d4w_oracle as orc

ARGS_SCRIPTS = dict(cs_min=1350., cp_min=1450., cp_max=3300, cs_max=3450, fmin=14., fmax=30.)


@pytest.mark.parametrize("shape,sel", [((40, 480), [0, 160, 4]), ((30, 360), [0, 30, 1]), ((64, 1000), [5, 133, 2])])
def test_pointwise_designs_bit_exact(shape, sel):
    dx, fs = 2.0419046878814697, 200.0
    ii, jj = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    full = orc.fk_filter_design(shape, sel, dx, fs)
    assert np.array_equal(orc.fk_filter_design_at(shape, sel, dx, fs, ii.ravel(), jj.ravel()).reshape(shape), full)
    full = orc.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS_SCRIPTS)
    got = orc.hybrid_ninf_filter_design_at(shape, sel, dx, fs, ii.ravel(), jj.ravel(), **ARGS_SCRIPTS).reshape(shape)
    assert np.array_equal(got, full)


def test_plane_wave_identity_against_oracle_filter():
    """y = sum_i a_i M_h(kx_i, kt_i) cos(theta_i) for on-grid plane waves, M_h from folded_gain_at; and the
    impulse-response rows of tests/known_answers.py against the oracle's filter of a unit impulse."""
    from tests import known_answers as ka
    nx, ns, sel, dx, fs = 96, 1200, [0, 384, 4], 2.0419046878814697, 200.0
    rng = np.random.default_rng(3)
    kx, kt, amp, ph = ka.pick_plane_waves(nx, ns, sel, dx, fs, rng)
    A, B = ka.wave_factors(nx, ns, kx, kt, ph)
    x = (A * np.tile(amp, 2)) @ B
    imp = np.zeros((nx, ns))
    imp[17, 333] = 1.0
    rows = np.array([17, 18, 0, 95, 60])
    for full, at in [(orc.fk_filter_design((nx, ns), sel, dx, fs),
                      functools.partial(orc.fk_filter_design_at, (nx, ns), sel, dx, fs)),
                     (orc.hybrid_ninf_filter_design((nx, ns), sel, dx, fs, **ARGS_SCRIPTS),
                      lambda i, j: orc.hybrid_ninf_filter_design_at((nx, ns), sel, dx, fs, i, j, **ARGS_SCRIPTS))]:
        g = orc.folded_gain_at(at, (nx, ns), kx, kt)
        assert np.count_nonzero(g) >= 6 and np.count_nonzero(g == 0) >= 6 and np.count_nonzero((g > 0) & (g < 0.99)) >= 1
        y = orc.fk_filter_filt(x, full)
        ref = (A * np.tile(amp * g, 2)) @ B
        assert np.max(np.abs(y - ref)) < 1e-12 * max(np.max(np.abs(y)), 1.0)
        mu = np.fft.ifftshift(np.asarray(full))
        h = ka.impulse_response_rows(lambda a, b: mu[a:b], nx, ns, (rows - 17) % nx)
        yi = orc.fk_filter_filt(imp, full)
        assert np.max(np.abs(np.roll(yi[rows], -333, axis=1) - h)) < 1e-13
