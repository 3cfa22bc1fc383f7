"""GPU test of the distributed f-k filter code path (das4whales_amd.shard.fk_filter_sharded) on the
one GPU a test box has: a world-size-1 RCCL process group runs the real all_to_all_single /
all-gather calls and the real HIP kernels of the time / channel phases (multi-rank exchange logic
is covered by tests/test_shard_gloo.py and tests/test_emu_fk_dist.py)."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def pg():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_sharded_golden(pg, golden):
    import das4whales_amd as dw
    from das4whales_amd import shard
    g = golden("fk_40x480.npz")
    x = torch.from_numpy(g["x"]).float().cuda()
    for key in ("ninf", "classic", "hybrid"):
        y = shard.fk_filter_sharded(x, g["m_" + key], x.shape[0])
        assert rel(y.cpu().numpy(), g["y_" + key]) < TOL
    y = shard.fk_filter_sharded(x, g["m_classic"], x.shape[0], tapering=True, gather=True)
    assert rel(y.cpu().numpy(), g["y_classic_taper"]) < TOL
    assert dw.dsp.fk_filter_filt is not None


def test_sharded_config1_block_matches_single_device(pg):
    """4000 x 12000 block, scripts' hybrid_ninf design made on the device: the pencil pipeline and
    the single-device five-pass plan agree to rounding, and both match the float64 oracle rows."""
    import das4whales_amd as dw
    from das4whales_amd import shard
    nx, ns, fs, dx = 4000, 12000, 200.0, 2.0419046878814697
    gen = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((nx, ns), device="cuda", generator=gen)
    mask = dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, cs_min=1350., cp_min=1450.,
                                            cp_max=3300, cs_max=3450, fmin=14., fmax=30.)
    y1 = dw.dsp.fk_filter_filt(x, mask)
    y2 = shard.fk_filter_sharded(x, mask, nx)
    scale = float(y1.abs().max())
    err = float((y1 - y2).abs().max()) / scale
    print("sharded vs single-device 4000x12000: %.3e" % err)
    assert err < 3e-6
    ident = torch.ones((nx, ns), device="cuda")
    y3 = shard.fk_filter_sharded(x, ident, nx)
    assert float((y3 - x).abs().max()) / float(x.abs().max()) < TOL


def test_sharded_plan_of_a_shape_without_built_in_kernels(pg):
    """ShardedFkPlan asks das4whales_amd/fkjit.py for kernels of a new large shape like dsp.get_fk_plan does (8000 x 12000 is
    compiled by __graft_entry__.build(), so this finds the cached object) and runs the packed plan; prime channel counts and
    record lengths run the generic plan's global-memory Bluestein forms.  Both equal the single-device filter."""
    import das4whales_amd as dw
    from das4whales_amd import shard
    gen = torch.Generator(device="cuda").manual_seed(7)
    for nx, ns, packed in ((8000, 12000, True), (2 * 4099, 2 * 2053 * 2, False)):
        x = torch.randn((nx, ns), device="cuda", generator=gen)
        m = torch.rand((nx, ns), device="cuda", generator=gen)
        plan = shard.ShardedFkPlan(nx, ns)
        assert plan.packed == packed
        plan.set_mask(m)
        y2 = shard.fk_filter_sharded(x, None, nx, plan=plan)
        y1 = dw.dsp.fk_filter_filt(x, m)
        assert float((y1 - y2).abs().max()) / float(y1.abs().max()) < 3e-6, (nx, ns)
        del plan


def test_sharded_bench_shape_identity(pg):
    """20 000 x 120 000 (BASELINE configs[3] block on one rank): an all-ones mask returns the input."""
    from das4whales_amd import shard
    nx, ns = 20000, 120000
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((nx, ns), device="cuda", generator=gen)
    plan = shard.ShardedFkPlan(nx, ns)
    ident = torch.ones((nx, ns), device="cuda")
    plan.set_mask(ident)
    del ident
    y = plan.apply(x)
    err = float((y - x).abs().max()) / float(x.abs().max())
    print("sharded 20000x120000 identity-mask error %.3e (N1=%d N2=%d, packed plan: %s)" % (err, plan.N1, plan.N2, plan.packed))
    assert err < TOL
    assert plan.packed                      # this shape has specialised kernels: packed exchange layout, 7 passes
    del y
    # the classic fan (dead wavenumber rows pruned per rank) against the single-device five-pass plan
    import das4whales_amd as dw
    fan = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419046878814697, 200.0)
    plan.set_mask(fan)                        # the closed form straight into the plan (no dense mask)
    assert fan._tensor is None
    y = plan.apply(x, taper=True)
    plan.set_mask(fan.tensor)                 # ... and folded from the dense mask: bit-identical
    assert torch.equal(plan.apply(x, taper=True), y)
    y1 = dw.dsp.fk_filter_filt(x, fan, tapering=True)
    err = float((y - y1).abs().max()) / float(y1.abs().max())
    print("sharded (packed) vs single-device 20000x120000, classic fan + taper: %.3e" % err)
    assert err < 3e-6
