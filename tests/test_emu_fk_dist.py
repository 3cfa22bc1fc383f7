"""Distributed f-k filter LOGIC on the CPU emulator build: the per-rank C-ABI phases
(d4w_fkd_time_fwd / chan_apply / time_inv) of every rank run in ONE process and the two all-to-all
exchanges are played by NumPy, exactly as include/d4w.h describes them.  Parity vs the reference
goldens and the oracle.  (tests/test_shard_gloo.py runs the real exchange with gloo.)"""
import ctypes

import numpy as np
import pytest

from oracle import d4w_oracle as orc
from tests.emu_util import load_emu, vp

TOL = 1e-5


@pytest.fixture(scope="module")
def emu():
    return load_emu()


def rel(y, ref):
    return np.max(np.abs(y - ref)) / np.max(np.abs(ref))


class Rank:
    def __init__(self, lib, nx, ns, world, rank):
        self.lib = lib
        self.h = ctypes.c_void_p()
        assert lib.d4w_fkd_plan_create(nx, ns, world, rank, ctypes.byref(self.h)) == 0, lib.d4w_last_error()
        info = (ctypes.c_int * 12)()
        assert lib.d4w_fkd_plan_info(self.h, info) == 0
        (_, _, _, _, self.a, self.b, self.N1, self.N2, self.nq, self.C1, self.C2, _) = list(info)
        own = (ctypes.c_int * self.N1)()
        assert lib.d4w_fkd_plan_q1_owner(self.h, own) == 0
        self.owner = np.array(list(own))

    def close(self):
        self.lib.d4w_fkd_plan_destroy(self.h)


def fk_sharded_emu(lib, x, mask, world, taper=False, setter=None):
    nx, ns = x.shape
    M = ns // 2
    ranks = [Rank(lib, nx, ns, world, r) for r in range(world)]
    N1, N2 = ranks[0].N1, ranks[0].N2
    owner = ranks[0].owner
    assert all(np.array_equal(rk.owner, owner) for rk in ranks)          # every rank derives the same map
    assert ranks[0].a == 0 and ranks[-1].b == nx and all(ranks[i].b == ranks[i + 1].a for i in range(world - 1))
    assert sorted(owner.tolist()) == sorted(sum(([r] * ranks[r].nq for r in range(world)), []))
    mf = np.ascontiguousarray(mask, dtype=np.float32)
    xf = np.ascontiguousarray(x, dtype=np.float32)
    z = []
    for rk in ranks:
        if setter is not None:
            setter(rk)
        else:
            assert lib.d4w_fkd_set_mask_dense_f32(rk.h, vp(mf), None) == 0
        xl = np.ascontiguousarray(xf[rk.a:rk.b])
        zl = np.empty((rk.b - rk.a, N1, N2, 2), dtype=np.float32)
        assert lib.d4w_fkd_time_fwd_f32(rk.h, vp(xl), vp(zl), int(taper), None) == 0, lib.d4w_last_error()
        z.append(zl)
    # all-to-all: sub-row q1 of every channel -> rank owner[q1]; receiver concatenates in rank order
    qidx = [np.nonzero(owner == s)[0] for s in range(world)]
    slabs = [np.ascontiguousarray(np.concatenate([z[r][:, qidx[s]] for r in range(world)], axis=0))
             for s in range(world)]
    for s, rk in enumerate(ranks):
        assert slabs[s].shape == (nx, rk.nq, N2, 2)
        assert lib.d4w_fkd_chan_apply_f32(rk.h, vp(slabs[s]) if rk.nq else None, None) == 0, lib.d4w_last_error()
    # all-to-all back
    y = np.empty((nx, ns), dtype=np.float32)
    for r, rk in enumerate(ranks):
        zl = z[r]
        for s in range(world):
            zl[:, qidx[s]] = slabs[s][rk.a:rk.b]
        assert lib.d4w_fkd_time_inv_f32(rk.h, vp(zl), None) == 0, lib.d4w_last_error()
        y[rk.a:rk.b] = zl.reshape(rk.b - rk.a, ns)
    for rk in ranks:
        rk.close()
    assert M == N1 * N2
    return y


@pytest.mark.parametrize("world", [1, 2, 3])
def test_dist_golden_40x480(emu, golden, world):
    g = golden("fk_40x480.npz")
    assert rel(fk_sharded_emu(emu, g["x"], g["m_ninf"], world), g["y_ninf"]) < TOL       # non-Hermitian design
    assert rel(fk_sharded_emu(emu, g["x"], g["m_classic"], world), g["y_classic"]) < TOL
    if world == 2:
        assert rel(fk_sharded_emu(emu, g["x"], g["m_classic"], world, taper=True), g["y_classic_taper"]) < TOL


def test_dist_uneven_blocks_and_primes(emu, golden):
    g = golden("fk_30x360.npz")                         # 30 channels over 4 ranks: blocks of 8, 8, 7, 7
    assert rel(fk_sharded_emu(emu, g["x"], g["m_ninf"], 4), g["y_ninf"]) < TOL
    rng = np.random.default_rng(2)
    nx, ns = 2 * 19, 2 * 7 * 11 * 3                     # loop-based prime radices on both axes
    x = rng.standard_normal((nx, ns))
    m = rng.uniform(0, 1, (nx, ns))
    assert rel(fk_sharded_emu(emu, x, m, 3), orc.fk_filter_filt(x, m)) < TOL


def test_dist_matches_single_gpu_path(emu):
    """Same answer as the single-device plan (d4w_fk_apply_f32) to rounding, more ranks than classes."""
    rng = np.random.default_rng(4)
    nx, ns = 24, 96
    x = rng.standard_normal((nx, ns)).astype(np.float32)
    m = rng.uniform(0, 1, (nx, ns)).astype(np.float32)
    h = ctypes.c_void_p()
    assert emu.d4w_fk_plan_create(nx, ns, ctypes.byref(h)) == 0
    assert emu.d4w_fk_set_mask_dense_f32(h, vp(m), None) == 0
    y1 = np.empty_like(x)
    assert emu.d4w_fk_apply_f32(h, vp(x), vp(y1), 0, None) == 0
    emu.d4w_fk_plan_destroy(h)
    for world in (2, 8):
        y = fk_sharded_emu(emu, x, m, world)
        assert rel(y, y1.astype(np.float64)) < 2e-6
        assert rel(y, orc.fk_filter_filt(x, m)) < TOL


@pytest.mark.parametrize("nx,ns,world,chunk", [(37, 48, 1, 0), (37, 48, 2, 0), (74, 120, 3, 0), (2 * 43, 96, 2, 16),
                                                (131, 40, 1, 5), (211, 24, 4, 0)])
def test_dist_channel_count_with_large_prime_factor(emu, nx, ns, world, chunk, monkeypatch):
    """A channel count with a prime factor > 31 on the generic distributed plan: the channel transform of the slab runs as a
    Bluestein convolution in global memory (fkd_bz_in / passes A, C of the padded length / fkd_bz_filter / ... / fkd_bz_out),
    a chunk of slab columns at a time (D4W_FKD_BZ_CHUNK pins a chunk narrower than the slab, with a ragged last one)."""
    if chunk:
        monkeypatch.setenv("D4W_FKD_BZ_CHUNK", str(chunk))
    rng = np.random.default_rng(nx + world)
    x = rng.standard_normal((nx, ns))
    m = rng.uniform(0, 1, (nx, ns))
    assert rel(fk_sharded_emu(emu, x, m, world), orc.fk_filter_filt(x, m)) < TOL
    if world == 2:
        assert rel(fk_sharded_emu(emu, x, m, world, taper=True), orc.fk_filter_filt(x, m, tapering=True)) < TOL
        assert rel(fk_sharded_emu(emu, x, np.ones((nx, ns)), world), x) < TOL


@pytest.mark.parametrize("nx,ns,world,chunk", [(37, 96, 1, 7), (74, 120, 2, 0), (131, 80, 1, 5)])
def test_dist_bluestein_channel_phase_transforms_the_live_columns_only(emu, nx, ns, world, chunk, monkeypatch):
    """A band mask (zero outside a frequency band, as every mask of the reference's designs is) on the global-memory Bluestein
    channel phase: only the slab columns whose folded gains -- or whose Hermitian partner column's -- are not all zero go
    through the scratch (d4w_fkd_plan::bz_cols), the others leave the pair op as zeros.  Symmetric bands, a band on the
    positive frequencies only (the partner closure), a band touching DC and Nyquist, a chunk narrower than the live set; the
    same result with the list switched off."""
    if chunk:
        monkeypatch.setenv("D4W_FKD_BZ_CHUNK", str(chunk))
    rng = np.random.default_rng(nx * 7 + world)
    x = rng.standard_normal((nx, ns))
    f = np.arange(-(ns // 2), ns - ns // 2) / (ns / 2.0)                  # the mask's fftshift-ed frequency grid
    for keep in ((np.abs(f) >= 0.15) & (np.abs(f) <= 0.35), (f >= 0.2) & (f <= 0.5), (np.abs(f) <= 0.1) | (np.abs(f) >= 0.9)):
        m = rng.uniform(0.1, 1, (nx, ns)) * keep[None, :]
        ref = orc.fk_filter_filt(x, m)
        seen = []

        def setter(rk, m=m):
            assert emu.d4w_fkd_set_mask_dense_f32(rk.h, vp(np.ascontiguousarray(m, dtype=np.float32)), None) == 0
            info = (ctypes.c_int * 2)()
            assert emu.d4w_fkd_plan_live_columns(rk.h, info) == 0
            seen.append(tuple(info))
        y = fk_sharded_emu(emu, x, m, world, setter=setter)
        assert rel(y, ref) < TOL
        live, total = sum(a for a, _ in seen), sum(b for _, b in seen)
        assert total == ns // 2 and 0 < live < 0.8 * total, seen            # (the band and its mirror image)
    assert rel(fk_sharded_emu(emu, x, np.zeros((nx, ns)), world), np.ones((nx, ns))) == 1.0      # all-zero mask: zeros


@pytest.mark.parametrize("nx,ns,world", [(12, 74, 1), (10, 2 * 41 * 3, 2), (37, 2 * 43, 3)])
def test_dist_record_length_with_large_prime_factor(emu, nx, ns, world):
    """ns / 2 with a prime factor > 31 on the generic distributed plan: the local rows' time transform is the global-memory
    Bluestein form (fkd_bt_*), the half spectrum a single natural-order class (one rank runs the channel phase)."""
    rng = np.random.default_rng(ns + world)
    x = rng.standard_normal((nx, ns))
    m = rng.uniform(0, 1, (nx, ns))
    assert rel(fk_sharded_emu(emu, x, m, world), orc.fk_filter_filt(x, m)) < TOL
    assert rel(fk_sharded_emu(emu, x, m, world, taper=True), orc.fk_filter_filt(x, m, tapering=True)) < TOL


def test_dist_plan_errors(emu):
    h = ctypes.c_void_p()
    assert emu.d4w_fkd_plan_create(40, 481, 2, 0, ctypes.byref(h)) == -1
    assert emu.d4w_fkd_plan_create(40, 480, 2, 2, ctypes.byref(h)) == -1
    assert emu.d4w_fkd_plan_create(4, 480, 8, 0, ctypes.byref(h)) == -1       # more ranks than channels


def fk_sharded_packed_emu(lib, x, mask, world, taper=False):
    """The PACKED protocol of include/d4w.h (shapes with specialised kernels): the time phase writes the send buffer,
    destination rank major; the exchanges are plain concatenations -- no index packing anywhere."""
    nx, ns = x.shape
    ranks = [Rank(lib, nx, ns, world, r) for r in range(world)]
    assert all(lib.d4w_fkd_plan_is_packed(rk.h) == 1 for rk in ranks)
    N1, N2 = ranks[0].N1, ranks[0].N2
    owner = ranks[0].owner
    assert all(np.array_equal(rk.owner, owner) for rk in ranks)
    nq = [int(np.sum(owner == s)) for s in range(world)]
    assert [rk.nq for rk in ranks] == nq
    mf = np.ascontiguousarray(mask, dtype=np.float32)
    xf = np.ascontiguousarray(x, dtype=np.float32)
    send = []
    for rk in ranks:
        assert lib.d4w_fkd_set_mask_dense_f32(rk.h, vp(mf), None) == 0, lib.d4w_last_error()
        nxl = rk.b - rk.a
        buf = np.full(nxl * N1 * N2 * 2, np.nan, dtype=np.float32)
        xl = np.ascontiguousarray(xf[rk.a:rk.b])
        assert lib.d4w_fkd_time_fwd_packed_f32(rk.h, vp(xl), vp(buf), int(taper), None) == 0, lib.d4w_last_error()
        assert not np.isnan(buf).any()
        send.append(buf)
    # all_to_all_single: rank r's block for s = elements [off_r[s], off_r[s+1])
    def splits(r):
        nxl = ranks[r].b - ranks[r].a
        return np.concatenate(([0], np.cumsum([nxl * nq[s] * N2 * 2 for s in range(world)])))
    slabs = []
    for s, rk in enumerate(ranks):
        parts = [send[r][splits(r)[s]:splits(r)[s + 1]] for r in range(world)]
        slab = np.ascontiguousarray(np.concatenate(parts))
        assert slab.size == nx * nq[s] * N2 * 2
        assert lib.d4w_fkd_chan_apply_f32(rk.h, vp(slab) if nq[s] else None, None) == 0, lib.d4w_last_error()
        slabs.append(slab)
    y = np.empty((nx, ns), dtype=np.float32)
    for r, rk in enumerate(ranks):
        nxl = rk.b - rk.a
        parts = [slabs[s][rk.a * nq[s] * N2 * 2:rk.b * nq[s] * N2 * 2] for s in range(world)]
        back = np.ascontiguousarray(np.concatenate(parts))
        yl = np.full((nxl, ns), np.nan, dtype=np.float32)
        assert lib.d4w_fkd_time_inv_packed_f32(rk.h, vp(back), vp(yl), None) == 0, lib.d4w_last_error()
        y[rk.a:rk.b] = yl
        # the same pass with the row statistics in its epilogue, in two row chunks
        ys = np.full((nxl, ns), np.nan, dtype=np.float32)
        mean, mx = np.zeros(nxl, np.float64), np.zeros(nxl, np.float32)          # float64 row means (include/d4w.h)
        cut = min(nxl, rk.C1 * max(1, (nxl // rk.C1) // 2))
        for l0, l1 in ((0, cut), (cut, nxl)):
            if l1 > l0:
                assert lib.d4w_fkd_time_inv_packed_rows_stats_f32(rk.h, vp(back), vp(ys), l0, l1, vp(mean), vp(mx), None) == 0, lib.d4w_last_error()
        assert np.array_equal(ys, yl)
        assert np.allclose(mx, np.abs(yl).max(axis=1), rtol=1e-6)
        assert np.max(np.abs(mean - yl.astype(np.float64).mean(axis=1))) < 1e-6 * max(np.abs(yl).max(), 1e-30)
    for rk in ranks:
        rk.close()
    return y


@pytest.mark.parametrize("nx,ns", [(18, 48), (100, 600), (8, 480)])
@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_dist_packed_specialised_shapes(emu, nx, ns, world):
    """Shapes with specialised kernels run the packed distributed plan (pass kernels of fk_fast.h in their slab modes):
    same filter as the oracle for dense masks, masks with dead wavenumber rows, with the taper, uneven blocks."""
    if world > nx:
        pytest.skip("more ranks than channels")
    rng = np.random.default_rng(nx * 7 + ns + world)
    x = rng.standard_normal((nx, ns))
    m = rng.uniform(0, 1, (nx, ns))
    assert rel(fk_sharded_packed_emu(emu, x, m, world), orc.fk_filter_filt(x, m)) < TOL
    ks = np.fft.fftshift(np.arange(nx))
    m[np.minimum(ks, nx - ks) > nx // 4, :] = 0.0            # dead rows: pruned per rank
    assert rel(fk_sharded_packed_emu(emu, x, m, world, taper=True), orc.fk_filter_filt(x, m, tapering=True)) < TOL


@pytest.mark.parametrize("world", [1, 3])
def test_dist_design_straight_into_the_plans(emu, world):
    """Generic (unpacked) distributed plan: d4w_fkd_set_mask_design_f32 == design the dense mask + fold it, bit for bit."""
    cd, ci, cv = ctypes.c_double, ctypes.c_int, ctypes.c_void_p
    emu.d4w_design_mask_f32.argtypes = [ci, ci, ci, cd, cd, cv, ci, ci, cv, cv, cv]
    emu.d4w_fkd_set_mask_design_f32.argtypes = [cv, ci, cd, cd, cv, ci, ci, cv, cv]
    nx, ns, fs, step = 40, 480, 200.0, 2.0419046878814697
    x = np.random.default_rng(3).standard_normal((nx, ns))
    p8 = np.array([1400., 1450., 3400., 3500., 0., 0., 0., 0.])
    dense = np.empty((nx, ns), dtype=np.float32)
    assert emu.d4w_design_mask_f32(0, nx, ns, step, 1.0 / fs, vp(p8), 0, 0, None, vp(dense), None) == 0

    def setter(rk):
        assert emu.d4w_fkd_set_mask_design_f32(rk.h, 0, step, 1.0 / fs, vp(p8), 0, 0, None, None) == 0, emu.d4w_last_error()
    y_design = fk_sharded_emu(emu, x, None, world, setter=setter)
    y_dense = fk_sharded_emu(emu, x, dense, world)
    assert np.array_equal(y_design, y_dense)
    assert rel(y_design, orc.fk_filter_filt(x, dense.astype(np.float64))) < TOL


def test_dist_plan_geometry_of_the_bench_block_over_8_ranks(emu):
    """BASELINE configs[3] on 8 ranks (20 000 x 120 000, N1 = 25 sub-rows of 2400): every rank derives the same owner map,
    a sub-row and its Hermitian partner (q1, N1 - q1) live on one rank, every sub-row has exactly one owner, the channel
    blocks tile the channels, and the busiest rank holds 4 of the 25 sub-rows -- the 0.78 balance of the channel phase
    DESIGN.md section 6 quotes (13 Hermitian classes cannot be dealt more evenly over 8 ranks)."""
    nx, ns, world = 20000, 120000, 8
    ranks = [Rank(emu, nx, ns, world, r) for r in range(world)]
    try:
        N1, owner = ranks[0].N1, ranks[0].owner
        assert N1 * ranks[0].N2 == ns // 2
        assert all(rk.N1 == N1 and np.array_equal(rk.owner, owner) for rk in ranks)
        assert ranks[0].a == 0 and ranks[-1].b == nx and all(ranks[i].b == ranks[i + 1].a for i in range(world - 1))
        assert all(rk.b - rk.a == nx // world for rk in ranks)
        for q in range(1, N1):
            assert owner[q] == owner[N1 - q], q
        per = [int(np.sum(owner == r)) for r in range(world)]
        assert per == [rk.nq for rk in ranks] and sum(per) == N1
        classes = N1 // 2 + 1
        assert max(per) == 2 * -(-classes // world) or max(per) == 2 * -(-classes // world) - 1
        assert max(per) == 4 and min(per) >= 2                       # 25 / (8 x 4) = 0.78
    finally:
        for rk in ranks:
            rk.close()
