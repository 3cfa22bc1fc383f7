"""Host logic of the on-demand shape compiler (das4whales_amd/fkjit.py): the configuration chooser respects every
constraint the kernel templates assert.  CPU only (nothing is compiled here)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fkjit():
    sys.path.insert(0, ROOT)
    if not os.path.exists(os.path.join(ROOT, "das4whales_amd", "lib", "libd4w.so")):
        import __graft_entry__ as ge
        ge.build()
    from das4whales_amd import fkjit as m
    return m


@pytest.mark.parametrize("nx,ns", [(20000, 120000), (4000, 12000), (11020, 12000), (8000, 12000), (1000, 12000), (2000, 24000),
                                   (12000, 60000), (16384, 16384), (50, 12000), (3, 48), (13223, 12000), (4001, 1200), (29 * 31 * 16, 2 * 25 * 16 * 10 * 10)])
def test_chooser_respects_template_constraints(fkjit, nx, ns):
    cfg = fkjit.choose_config(nx, ns)
    assert cfg is not None
    C1, C2A, C2B, N1, NA, NB, NC, TA, TC, thrA, thrC, thrB, wgA, wgC, wgB, C2X = cfg
    assert C1 * C2A * C2B * C2X == nx and 2 * N1 * NA * NB * NC == ns and (C2X == 1 or C2A * C2B == 1)
    assert max(C1, C2A, C2B, N1, NA, NB, NC) <= 32 and C2B <= 32
    N2, M = NA * NB * NC, ns // 2
    assert TA == TC and N2 % TA == 0 and M % TC == 0
    assert N1 * TA <= thrA and C1 * TA <= thrA and C2B * TC <= thrC and C2A * TC <= thrC
    assert 2 * NB * NC <= thrB and NA * NB <= thrB and max(thrA, thrC, thrB) <= 1024
    lds_a = (C1 * N1 * TA + 2 * N1 * TA) * 8 + 2 * N1 * 4
    lds_c = (C2A * (C2B + 1) * TC + C2A * C2B) * 8 + C1 * C2A * 4
    lds_b = (2 * (N2 + NA * NB) + 2 * NB * NC) * 8
    assert max(lds_a, lds_c, lds_b) <= 160 * 1024
    assert min(wgA, wgC, wgB) >= 1


def test_chooser_declines_shapes_without_a_configuration(fkjit):
    assert fkjit.choose_config(2 * 4099, 12000) is None       # rough part beyond the Bluestein tile
    assert fkjit.choose_config(37 * 64, 12000) is None        # smooth part 64 > 32 rows per pass-A tile
    assert fkjit.choose_config(300, 12002) is None            # 6001 = 17 x 353
    assert fkjit.choose_config(40, 481) is None
    assert fkjit.is_specialised(20000, 120000) and not fkjit.is_specialised(20001, 120000)


def test_failed_configuration_is_remembered_and_warns(fkjit):
    """A shape with no admissible configuration (both axes carry huge primes): compile_fk_shape returns False, says why,
    remembers it (no second attempt in this process) and -- with warn=True, as the automatic path of dsp.get_fk_plan /
    dsp._analytic asks -- raises a RuntimeWarning instead of degrading silently."""
    import time
    import warnings
    nx, ns = 2 * 10007, 2 * 2 * 60013
    assert fkjit.choose_config(nx, ns) is None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert fkjit.compile_fk_shape(nx, ns, warn=True) is False
    assert any(issubclass(x.category, RuntimeWarning) and "generic kernels" in str(x.message) for x in w)
    assert "configuration" in fkjit.failure_reason(nx, ns)
    t0 = time.perf_counter()
    assert fkjit.compile_fk_shape(nx, ns) is False                # remembered: no second search, no compiler run
    assert time.perf_counter() - t0 < 0.5


def test_failed_build_leaves_no_files_and_a_marker(fkjit, tmp_path, monkeypatch):
    """A compiler that fails: the generated source and the partial object are removed, a .failed marker keyed by the kernel
    headers' hash stops further attempts across processes, and the reason is reported."""
    fake = tmp_path / "hipcc"
    fake.write_text("#!/bin/sh\necho 'error: simulated' >&2\nexit 1\n")
    fake.chmod(0o755)
    jitdir = tmp_path / "jit"
    monkeypatch.setenv("HIPCC", str(fake))
    monkeypatch.setattr(fkjit, "_JITDIR", str(jitdir))
    nx, ns = 640, 9600                                            # has a configuration, is not built in
    assert fkjit.choose_config(nx, ns) is not None and not fkjit.is_specialised(nx, ns)
    fkjit._failed.clear()
    assert fkjit.compile_fk_shape(nx, ns) is False
    left = sorted(os.listdir(jitdir))
    assert len(left) == 1 and left[0].endswith(".failed"), left
    assert "simulated" in fkjit.failure_reason(nx, ns)
    fkjit._failed.clear()                                         # "another process": the marker alone stops the retry
    fake.write_text("#!/bin/sh\ntouch %s\nexit 1\n" % (tmp_path / "ran_again"))
    assert fkjit.compile_fk_shape(nx, ns) is False
    assert not (tmp_path / "ran_again").exists()
    fkjit._failed.clear()
