import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import das4whales_amd as dw
rng = np.random.default_rng(0)
nx, ns, L = 4, 20000, 120
x = rng.standard_normal((nx, ns))
xd = torch.from_numpy(x.astype(np.float32)).cuda()
for eps in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
    tp = np.concatenate(([1.0], eps * rng.standard_normal(L - 1)))       # one big tap, the rest at eps: hi halves of the small taps are binary16 subnormals below 6e-5
    mean32, mx32 = xd.mean(dim=1).contiguous(), xd.abs().amax(dim=1).contiguous()
    y = dw.detect._xcorr_device(xd, [tp], normalize=True, method="mm", stats=(mean32, mx32))[0].cpu().numpy()
    x64 = xd.double().cpu().numpy()
    xn = (x64 - mean32.double().cpu().numpy()[:, None]) / np.abs(x64).max(axis=1, keepdims=True)
    ref = np.stack([np.correlate(np.concatenate((r, np.zeros(L - 1))), tp, "valid") for r in xn])
    small = np.stack([np.correlate(np.concatenate((r, np.zeros(L - 1))), np.concatenate(([0.0], tp[1:])), "valid") for r in xn])
    e = np.abs(y - ref).max() / np.abs(ref).max()
    print("eps %.0e: err %.2e   (the small taps contribute %.2e of the output: lost if subnormal operands are flushed)" % (eps, e, np.abs(small).max() / np.abs(ref).max()))
