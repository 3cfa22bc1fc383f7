import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, das4whales_amd as dw
x = torch.randn((20000, 120000), device="cuda") + 0.5
for _ in range(4):
    y = dw.dsp.bp_filt(x, 200.0, 14, 30)
torch.cuda.synchronize()
