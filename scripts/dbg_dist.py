"""Timing / debugging aid for the sharded f-k filter on one GPU (world-size-1 RCCL group)."""
import sys, os, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from das4whales_amd import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
nx, ns = int(sys.argv[1]), int(sys.argv[2])
def T(msg, t0):
    torch.cuda.synchronize(); print("%-28s %.3f s" % (msg, time.time() - t0), flush=True); return time.time()
t = time.time()
x = torch.randn((nx, ns), device="cuda"); t = T("randn", t)
plan = shard.ShardedFkPlan(nx, ns); t = T("plan", t)
m = torch.ones((nx, ns), device="cuda"); plan.set_mask(m); del m; t = T("mask", t)
for it in range(2):
    y = plan.apply(x); t = T("apply %d" % it, t)
print("err", float((y - x).abs().max()) / float(x.abs().max()))
dist.destroy_process_group()
