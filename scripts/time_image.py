"""Stage times of the Gabor image pipeline (das4whales_amd.improcess.gabor_mask) on resident data.
    python scripts/time_image.py [--nx 4000] [--ns 12000]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import das4whales_amd as dw  # noqa: E402
from das4whales_amd import improcess as ip  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4000)
    ap.add_argument("--ns", type=int, default=12000)
    args = ap.parse_args()
    nx, ns = args.nx, args.ns
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((nx, ns), device="cuda", generator=g)
    up, down = ip.gabor_filt_design(42.56)
    out = {"shape": [nx, ns]}
    out["trace2image_ms"], image = timed(lambda: ip._trace2image_device(x))
    bh, bw = int(nx * 0.1), int(ns * 0.1)
    out["binning_down_ms"], imagebin = timed(lambda: ip._resize_device(image, bh, bw))
    out["filter2d_pair_ms"], fimage = timed(lambda: ip._filter2d_device(imagebin, [up, down]))
    thr = float(fimage.float().mean() + fimage.float().std())
    binary = (fimage > thr).float()
    out["filter2d_pair_binary_ms"], score = timed(lambda: ip._filter2d_device(binary, [up, down]))
    mask = (score > float(score.max()) * 0.3).float()
    if (bh * 10, bw * 10) == (nx, ns):
        out["binning_up_ms"], ms = timed(lambda: ip._resize_device(mask, nx, ns))
        out["mask_mul_ms"], _ = timed(lambda: ip._mask_mul_device(x, ms))
        out["gabor_mask_total_ms"], _ = timed(lambda: ip.gabor_mask(x, 200., 2.0419, [0, nx * 4, 4], 1500., thr, float(score.max()) * 0.3), reps=5)
    flop = 2.0 * 2 * bh * bw * 101 * 101
    out["filter2d_TFLOPs"] = flop / out["filter2d_pair_ms"] * 1e-9
    out["trace2image_GBps"] = (nx * ns * 4 * (2 + 2 + 1 + 2)) / out["trace2image_ms"] * 1e-6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
