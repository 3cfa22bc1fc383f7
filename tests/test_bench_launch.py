"""bench.py --gpus N must start its own N ranks: run without a launcher it re-executes itself
under torch.distributed.run; here on CPU ranks (--backend gloo, emulator kernels) -- the JSON line must say n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus2_self_launch_gloo():
    from tests.emu_util import build_emu
    build_emu()
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["rel_err_vs_oracle"] < 1e-5
    assert "emulated" in out["data"]
    assert out["gather"]["in_timed_step"].startswith("none") and out["gather"]["all_gather_tx_ms"] > 0
    assert out["gather"]["picks_gathered"] > 0 and sum(out["config"]["plan"]["sub_rows_per_rank"]) == out["config"]["plan"]["N1"]


def test_bench_gpus8_self_launch_gloo():
    """Eight CPU ranks through the launch path, the pencil f-k exchange and both ways of collecting the result (the t-x
    all-gather and the gather of picks); the line reports how the channel phase is balanced over the ranks."""
    from tests.emu_util import build_emu
    build_emu()
    r = _run(["--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--nx", "96", "--ns", "960"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["rel_err_vs_oracle"] < 1e-5
    plan = out["config"]["plan"]
    per = plan["sub_rows_per_rank"]
    assert len(per) == 8 and sum(per) == plan["N1"]
    assert abs(plan["channel_phase_balance"] - plan["N1"] / (8.0 * max(per))) < 1e-3
    assert out["gather"]["all_gather_tx_ms"] > 0 and out["gather"]["all_gather_picks_ms"] > 0
    # both ways of reassembling the t-x matrix are printed (one collective, or N - 1 point-to-point pairs per rank
    # with every link busy) and agree
    assert out["gather"]["all_gather_tx_direct_ms"] > 0 and out["gather"]["direct_equals_collective"] is True


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in this container: `bench.py --gpus 8` must exit non-zero, not print an n_gpus = 1 line."""
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("8 GPUs visible")
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
