"""Target for rocprofv3 --kernel-trace --stats: dsp.bp_filt NREP times on an NX x NS block (default the 60-s file shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw
nx, ns, nrep = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000)), int(os.environ.get("NREP", 20))
x = torch.randn((nx, ns), device="cuda") + 0.5
for _ in range(nrep):
    y = dw.dsp.bp_filt(x, 200.0, 14, 30)
torch.cuda.synchronize()
