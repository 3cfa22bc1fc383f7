"""world_size-2 gloo tests (CPU) of the channel-block sharding + single all-gather reassembly that
the multi-GPU path uses with RCCL on the GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_shard():
    """Import das4whales_amd/shard.py without triggering the package's GPU-library import."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("d4w_shard", os.path.join(ROOT, "das4whales_amd", "shard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, nx, ns, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = _load_shard()
        g = torch.Generator().manual_seed(7)
        x = torch.randn((nx, ns), generator=g)                     # same on every rank
        fn = lambda blk: torch.cumsum(blk, dim=1) * 0.5 + blk.flip(1)   # row-independent stand-in
        a, b = shard.channel_block(nx, world, rank)
        y_local = fn(x[a:b])
        y = shard.all_gather_rows(y_local, nx)
        y2 = shard.map_channel_blocks(fn, x)
        # the direct form (N - 1 grouped isend / irecv pairs straight into the result's row ranges: every point-to-point link
        # busy), blocking and behind other work
        y3 = shard.all_gather_rows(y_local, nx, how="direct")
        y4, work = shard.all_gather_rows(y_local, nx, how="direct", async_op=True)
        work.wait()
        ok = bool(torch.equal(y, fn(x)) and torch.equal(y2, y) and torch.equal(y3, y) and torch.equal(y4, y))
        q.put((rank, ok, (a, b)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nx,ns", [(2, 10, 33), (2, 7, 16), (3, 8, 5), (3, 2, 9)])    # even / uneven blocks, a rank without rows
def test_all_gather_rows(world, nx, ns):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ns, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    blocks = sorted(b for _, _, b in res)
    assert blocks[0][0] == 0 and blocks[-1][1] == nx and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


def _picks_worker(rank, world, port, nx, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        shard = _load_shard()
        rng = np.random.default_rng(11)                             # same picks on every rank
        rows = [np.sort(rng.choice(5000, size=int(rng.integers(0, 9)), replace=False)) for _ in range(nx)]
        if rank == 1:
            pass
        a, b = shard.channel_block(nx, world, rank)
        loc = rows[a:b]
        ch = np.concatenate([np.full(len(p), i, dtype=np.int64) for i, p in enumerate(loc)]) if loc else np.zeros(0, dtype=np.int64)
        tt = np.concatenate(loc) if loc else np.zeros(0, dtype=np.int64)
        got = shard.all_gather_picks(torch.from_numpy(np.stack((ch, tt)).astype(np.int64)), a)
        ref_ch = np.concatenate([np.full(len(p), i, dtype=np.int64) for i, p in enumerate(rows)])
        ref = np.stack((ref_ch, np.concatenate(rows)))
        q.put((rank, bool(np.array_equal(got.numpy(), ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nx", [(2, 11), (3, 7)])
def test_all_gather_picks(world, nx):
    """The picks of a channel-sharded block reassembled with global channel indices, in channel order (uneven blocks,
    ranks without picks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_picks_worker, args=(r, world, port, nx, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_channel_block_partition():
    shard = _load_shard()
    for nx in (1, 7, 8, 20000, 11020):
        for world in (1, 2, 3, 8):
            blocks = [shard.channel_block(nx, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nx
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------------------------
# exact distributed f-k filter: the REAL exchange code of das4whales_amd/shard.py
# (all_to_all_single x2 + all-gather) over gloo, with the CPU emulator build of the HIP sources
# standing in for libd4w.so (same C ABI, host pointers) -- world sizes 2 and 3
# ------------------------------------------------------------------------------------------
def _fk_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        import numpy as np
        from tests.emu_util import load_emu
        emu = load_emu()
        for name in ("d4w_fkd_plan_create", "d4w_fkd_plan_info", "d4w_fkd_plan_q1_owner", "d4w_fkd_set_mask_dense_f32",
                     "d4w_fkd_time_fwd_f32", "d4w_fkd_chan_apply_f32", "d4w_fkd_time_inv_f32"):
            getattr(emu, name).restype = ctypes.c_int
        emu.d4w_fkd_plan_create.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_void_p)]
        emu.d4w_fkd_plan_destroy.argtypes = [ctypes.c_void_p]
        emu.d4w_fkd_plan_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        emu.d4w_fkd_plan_q1_owner.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        emu.d4w_fkd_set_mask_dense_f32.argtypes = [ctypes.c_void_p] * 3
        emu.d4w_fkd_time_fwd_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
        emu.d4w_fkd_chan_apply_f32.argtypes = [ctypes.c_void_p] * 3
        emu.d4w_fkd_time_inv_f32.argtypes = [ctypes.c_void_p] * 3

        def check(rc):
            assert rc == 0, emu.d4w_last_error()
        shard = _load_shard()
        g = np.load(os.path.join(ROOT, "tests", "golden", "fk_40x480.npz"))
        x = torch.from_numpy(g["x"]).float()
        nx = x.shape[0]
        a, b = shard.channel_block(nx, world, rank)
        res = {}
        for key in ("ninf", "classic"):
            plan = shard.ShardedFkPlan(nx, x.shape[1], native=(emu, check))
            plan.set_mask(torch.from_numpy(g["m_" + key]).float())
            y_loc = shard.fk_filter_sharded(x[a:b], None, nx, plan=plan)
            ref = g["y_" + key]
            res[key + "_local"] = float(np.max(np.abs(y_loc.numpy() - ref[a:b])) / np.max(np.abs(ref)))
            y_all = shard.fk_filter_sharded(x[a:b], None, nx, gather=True, plan=plan)    # + single all-gather
            plan.MAX_CALL_ELEMS = 5000                    # force the row-chunked exchange (3+ calls each way)
            y_chunked = shard.fk_filter_sharded(x[a:b], None, nx, plan=plan)
            res[key + "_chunked"] = 0.0 if torch.equal(y_chunked, y_loc) else 1.0
            res[key + "_gathered"] = float(np.max(np.abs(y_all.numpy() - ref)) / np.max(np.abs(ref)))
        # a channel count with a prime factor > 31 (2 x 37): the slab's channel transform is the global-memory Bluestein form;
        # ns / 2 = 43 as well: the rows' time transform too, and the half spectrum is one class owned by rank 0
        from oracle import d4w_oracle as orc
        rng = np.random.default_rng(3)
        for tag, nx2, ns2 in (("rough_nx", 74, 48), ("rough_both", 74, 86)):
            x2 = rng.standard_normal((nx2, ns2))
            m2 = rng.uniform(0, 1, (nx2, ns2))
            a2, b2 = shard.channel_block(nx2, world, rank)
            plan = shard.ShardedFkPlan(nx2, ns2, native=(emu, check))
            plan.set_mask(torch.from_numpy(m2).float())
            y_all = shard.fk_filter_sharded(torch.from_numpy(x2[a2:b2]).float(), None, nx2, gather=True, plan=plan)
            ref = orc.fk_filter_filt(x2, m2)
            res[tag + "_gathered"] = float(np.max(np.abs(y_all.numpy() - ref)) / np.max(np.abs(ref)))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fk_filter_sharded_gloo(world):
    from tests.emu_util import build_emu
    build_emu()                                           # build once, before the ranks race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fk_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, r in res:
        for k, v in r.items():
            assert v < 1e-5, (rank, k, v)


# ------------------------------------------------------------------------------------------
# PACKED distributed plan (shapes with specialised kernels): grouped send / recv in row chunks,
# no index packing -- the real exchange code of shard.ShardedFkPlan._apply_packed over gloo
# ------------------------------------------------------------------------------------------
def _fk_packed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        import numpy as np
        from oracle import d4w_oracle as orc
        from tests.emu_util import load_emu
        emu = load_emu()
        vp_, ci = ctypes.c_void_p, ctypes.c_int
        emu.d4w_fkd_plan_create.argtypes = [ci] * 4 + [ctypes.POINTER(vp_)]
        emu.d4w_fkd_plan_destroy.argtypes = [vp_]
        emu.d4w_fkd_plan_info.argtypes = [vp_, ctypes.POINTER(ci)]
        emu.d4w_fkd_plan_q1_owner.argtypes = [vp_, ctypes.POINTER(ci)]
        emu.d4w_fkd_set_mask_dense_f32.argtypes = [vp_] * 3
        emu.d4w_fkd_chan_apply_f32.argtypes = [vp_] * 3
        emu.d4w_fkd_time_fwd_packed_f32.argtypes = [vp_] * 3 + [ci, vp_]
        emu.d4w_fkd_time_inv_packed_f32.argtypes = [vp_] * 4
        emu.d4w_fkd_time_fwd_packed_rows_f32.argtypes = [vp_] * 3 + [ci, ci, ci, vp_]
        emu.d4w_fkd_time_inv_packed_rows_f32.argtypes = [vp_] * 3 + [ci, ci, vp_]

        def check(rc):
            assert rc == 0, emu.d4w_last_error()
        shard = _load_shard()
        rng = np.random.default_rng(21)
        nx, ns = 100, 600                                  # FkShapeT3: C1 = 5 rows per time-phase tile
        x = rng.standard_normal((nx, ns))
        m = rng.uniform(0, 1, (nx, ns))
        ks = np.fft.fftshift(np.arange(nx))
        m[np.minimum(ks, nx - ks) > 30, :] = 0.0           # dead wavenumber rows
        ref = orc.fk_filter_filt(x, m, tapering=True)
        a, b = shard.channel_block(nx, world, rank)
        plan = shard.ShardedFkPlan(nx, ns, native=(emu, check))
        assert plan.packed
        plan.set_mask(torch.from_numpy(m).float())
        xl = torch.from_numpy(x[a:b]).float()
        res = {}
        for chunks in (1, 3):
            plan.CHUNKS = chunks
            y = plan.apply(xl, taper=True)
            res["chunks%d" % chunks] = float(np.max(np.abs(y.numpy() - ref[a:b])) / np.max(np.abs(ref)))
        y_all = shard.fk_filter_sharded(xl, None, nx, tapering=True, gather=True, plan=plan)
        res["gathered"] = float(np.max(np.abs(y_all.numpy() - ref)) / np.max(np.abs(ref)))
        # a closed-form design evaluated per rank for the sub-rows it owns (no dense mask on any rank) is bit-identical
        # to designing the dense mask and folding it
        import scipy.signal as sps
        cd = ctypes.c_double
        emu.d4w_design_mask_f32.argtypes = [ci, ci, ci, cd, cd, vp_, ci, ci, vp_, vp_, vp_]
        emu.d4w_fkd_set_mask_design_f32.argtypes = [vp_, ci, cd, cd, vp_, ci, ci, vp_, vp_]
        fs, step = 200.0, 2.0419046878814697
        f = np.fft.fftshift(np.fft.fftfreq(ns, d=1 / fs))
        bb, aa = sps.butter(8, [14. / (fs / 2), 30. / (fs / 2)], "bp")

        class Design:                                      # duck type of das4whales_amd.dsp.DesignedMask
            shape, mode, k_spacing, t_spacing = (nx, ns), 2, step, 1.0 / fs
            params = [1350., 1450., 3300., 3450., 0., 0., 0., 0.]
            i0, i1 = int(np.argmax(f >= 0.)), int(np.argmax(f >= 44.))
            hrow = torch.from_numpy(np.concatenate((np.zeros(ns // 2), np.abs(sps.freqz(bb, aa, worN=ns // 2)[1]) ** 2)))

            def hrow_on(self, device):
                return self.hrow
        dmask = Design()
        dense = np.empty((nx, ns), dtype=np.float32)
        p8 = np.array(dmask.params)
        check(emu.d4w_design_mask_f32(2, nx, ns, step, 1.0 / fs, p8.ctypes.data, dmask.i0, dmask.i1, dmask.hrow.data_ptr(),
                                      dense.ctypes.data, None))
        plan.set_mask(torch.from_numpy(dense))
        y_dense = plan.apply(xl)
        plan.set_mask(dmask)
        y_design = plan.apply(xl)
        res["design_vs_dense"] = 0.0 if torch.equal(y_dense, y_design) else 1.0
        refd = orc.fk_filter_filt(x, dense.astype(np.float64))
        res["design"] = float(np.max(np.abs(y_design.numpy() - refd[a:b])) / np.max(np.abs(refd)))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fk_filter_sharded_packed_gloo(world):
    from tests.emu_util import build_emu
    build_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fk_packed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, r in res:
        for k, v in r.items():
            assert v < 1e-5, (rank, k, v)
