"""Does the Infinity Cache keep a pass's output for the next pass?  Times pass C (forward then
inverse, alternating) over (a) the same tile range repeatedly, (b) ranges that walk the block."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw
from das4whales_amd._lib import lib, check
lib.d4w_fk_debug_run_pass.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nx, ns = 20000, 120000
x = torch.randn((nx, ns), dtype=torch.float32, device="cuda") * 1e-3
plan = dw.dsp.FkPlan(nx, ns)
st = torch.cuda.current_stream().cuda_stream
ntC = (ns // 2 // 16) * 25
ntB = 250000
def run(passes, ranges, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 0
    for r in range(reps):
        for (a, b) in ranges(r):
            for p in passes:
                check(lib.d4w_fk_debug_run_pass(plan._h, x.data_ptr(), p, a, b, st)); n += b - a
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e6 / n      # ns per tile-pass
for ntile in (500, 1000, 2000, 4000, 8000):
    mb = ntile * 102.4e3 / 1e6
    same = run((1, 3), lambda r: [(0, ntile)], 20)
    walk = run((1, 3), lambda r: [((r * ntile) % (ntC - ntile), (r * ntile) % (ntC - ntile) + ntile)], 20)
    print("pass C  %5d tiles (%6.1f MB): same range %.1f ns/tile, walking %.1f ns/tile  (full pass: %.1f)" % (ntile, mb, same, walk, 3.8e6 / ntC))
for npair in (1000, 2600, 5200, 10400):
    mb = npair * 38.4e3 / 1e6
    same = run((2,), lambda r: [(0, npair)], 20)
    walk = run((2,), lambda r: [((r * npair) % (ntB - npair), (r * npair) % (ntB - npair) + npair)], 20)
    print("pass B  %5d pairs (%6.1f MB): same range %.1f ns/pair, walking %.1f ns/pair (full pass: %.1f)" % (npair, mb, same, walk, 5.85e6 / ntB))
