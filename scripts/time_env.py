"""Envelope of an 11 020 x 12 000 block (the stream's correlogram shape), a few launches: the workload of the PMC passes that compare
analytic_rows with analytic_rows_pair (scripts/r06_session.sh envelope_pmc)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import dsp
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
x = torch.randn((nx, ns), device="cuda")
for _ in range(int(os.environ.get("REPS", 6))):
    y = dsp._analytic(x, 0)
torch.cuda.synchronize()
