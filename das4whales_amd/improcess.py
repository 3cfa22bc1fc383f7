"""improcess -- the image pipeline of the Gabor detector: the part of the
reference's `das4whales.improcess` that scripts/main_gabordetect.py:78-166 runs, on the GPU.

    image    = trace2image(trf_fk)                         |hilbert| / std, min-max scaled to [0, 255]
    imagebin = binning(image, 1/10, 1/10)                  torchvision Resize (antialiased bilinear)
    up, down = gabor_filt_design(angle_fromspeed(c0, fs, dx, selected_channels))
    fimage   = filter2d(imagebin, up) + filter2d(imagebin, down)          (cv2.filter2D in the script)
    mask     = (filter2d(fimage > thr, up) + filter2d(fimage > thr, down)) > thr2
    masked   = apply_smooth_mask(trf_fk, binning(mask, 10, 10))

`gabor_mask` runs those steps in one call with everything resident on the device.  cv2 and torchvision
are not used (and not installed): `gabor_filt_design` evaluates OpenCV's getGaborKernel formula on the
host in float64, `filter2d` and `binning` are HIP kernels (csrc/image.hip) that follow cv2.filter2D
(correlation, BORDER_REFLECT_101) and aten's antialiased bilinear interpolation.  The functions of
improcess.py that the detector does not use (Canny / Hough / Radon / bilateral experiments) are out
of scope.  Arrays: NumPy in -> NumPy out (float dtype kept, compute in float32), CUDA tensor in ->
CUDA tensor out; masks are bool.
"""
import numpy as np
import torch

from . import _device as dev
from . import dsp
from ._lib import lib, check


def _n(t):
    return int(t.numel())


def _minmax(x):
    mm = torch.empty(2, dtype=torch.float32, device=x.device)
    check(lib.d4w_minmax_f32(dev.ptr(x), _n(x), dev.ptr(mm), dev.stream_ptr(x)))
    return mm


def _scale_pixels_device(x, gain, out=None):
    y = torch.empty_like(x) if out is None else out
    with torch.cuda.device(x.device):
        mm = _minmax(x)
        check(lib.d4w_scale_pixels_f32(dev.ptr(x), dev.out_ptr(y), _n(x), dev.ptr(mm), float(gain), dev.stream_ptr(x)))
    return y


def scale_pixels(img):
    """(img - img.min()) / (img.max() - img.min()) -- reference improcess.py:23-40."""
    x = dev.to_device_f32(img)
    return dev.like_input(_scale_pixels_device(x, 1.0), img)


def _trace2image_device(x):
    nx, ns = x.shape
    var = torch.empty(nx, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.d4w_row_var_f32(dev.ptr(x), nx, ns, dev.ptr(var), dev.stream_ptr(x)))
    env = dsp._analytic(x, 4, var=var)                      # |hilbert(x)| / std(x) per row
    return _scale_pixels_device(env, 255.0, out=env)


def trace2image(trace):
    """|hilbert(trace, axis=1)| / std(trace, axis=1), min-max scaled to [0, 255] -- reference
    improcess.py:43-62."""
    if getattr(trace, "ndim", 0) != 2:
        raise ValueError("trace must be a 2-D [channel x time] array")
    return dev.like_input(_trace2image_device(dev.to_device_f32(trace)), trace)


def angle_fromspeed(c0, fs, dx, selected_channels):
    """Angle (degrees) of a c0 m/s arrival in the [channel x time] pixel grid -- reference
    improcess.py:65-96 (prints the ratio and the angle like the reference)."""
    ratio = c0 / (fs * dx * selected_channels[2])
    print('Detection speed ratio: ', ratio)
    theta_c0 = np.arctan(ratio) * 180 / np.pi
    print('Angle: ', theta_c0)
    return theta_c0


def get_gabor_kernel(ksize, sigma, theta, lambd, gamma, psi=np.pi * 0.5):
    """cv2.getGaborKernel(ksize, sigma, theta, lambd, gamma, psi, ktype=CV_64F) evaluated on the host:
    exp(-(x'^2 / sigma^2 + gamma^2 y'^2 / sigma^2) / 2) cos(2 pi x' / lambd + psi) on the grid
    [-h//2, h//2] x [-w//2, w//2], stored mirrored in both axes as OpenCV does."""
    xmax, ymax = int(ksize[0]) // 2, int(ksize[1]) // 2
    c, s = np.cos(theta), np.sin(theta)
    yy, xx = np.meshgrid(np.arange(-ymax, ymax + 1, dtype=np.float64), np.arange(-xmax, xmax + 1, dtype=np.float64),
                         indexing="ij")
    xr, yr = xx * c + yy * s, yy * c - xx * s
    sx, sy = float(sigma), float(sigma) / float(gamma)
    g = np.exp(-0.5 * (xr * xr / (sx * sx) + yr * yr / (sy * sy))) * np.cos(2.0 * np.pi / lambd * xr + psi)
    return np.ascontiguousarray(g[::-1, ::-1])


def gabor_filt_design(theta_c0, plot=False):
    """The two 101 x 101 Gabor kernels oriented along +-theta_c0 -- reference improcess.py:99-140
    (ksize 100, sigma 4, lambda 20, gamma 0.15, psi 0).  `plot` is accepted and ignored."""
    up = get_gabor_kernel((100, 100), 4, np.pi / 2 + np.deg2rad(theta_c0), 20, 0.15, 0)
    return up, np.flipud(up)


def _filter2d_device(img, kernels):
    """sum_k filter2D(img, kernels[k]) on the device; img float32 CUDA [h, w]."""
    h, w = img.shape
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        for i, k in enumerate(kernels):
            kd = dev.to_device_f32(k, img.device)
            kh, kw = kd.shape
            ws = torch.empty(int(lib.d4w_filter2d_ws_bytes(kh, kw)), dtype=torch.uint8, device=img.device)
            check(lib.d4w_filter2d_f32(dev.ptr(img), h, w, dev.ptr(kd), kh, kw, dev.ptr(out), int(i > 0), dev.ptr(ws),
                                       dev.stream_ptr(img)))
    return out


def filter2d(img, kernel):
    """cv2.filter2D(img, cv2.CV_64F, kernel) (scripts/main_gabordetect.py:109,132): correlation with the
    anchor at the kernel centre and BORDER_REFLECT_101 borders.  A bool image is taken as 0 / 1."""
    if getattr(img, "ndim", 0) != 2 or getattr(kernel, "ndim", 0) != 2:
        raise ValueError("img and kernel must be 2-D")
    x = dev.to_device_f32(img)
    y = _filter2d_device(x, [kernel])
    if dev.is_tensor(img):
        return y
    return y.cpu().numpy().astype(np.float64)


def _resize_device(x, oh, ow):
    h, w = x.shape
    if oh < 1 or ow < 1:
        raise ValueError(f"binning: output size ({oh}, {ow}) is empty")
    y = torch.empty((oh, ow), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws = torch.empty(int(lib.d4w_resize_ws_bytes(h, w, oh, ow)), dtype=torch.uint8, device=x.device)
        check(lib.d4w_resize_bilinear_aa_f32(dev.ptr(x), h, w, dev.ptr(y), oh, ow, dev.ptr(ws), dev.stream_ptr(x)))
    return y


def _is_bool(a):
    return (dev.is_tensor(a) and a.dtype == torch.bool) or (not dev.is_tensor(a) and np.asarray(a).dtype == bool)


def binning(image, ft, fx):
    """transforms.Resize((int(H * fx), int(W * ft)))(ToTensor()(image)) -- reference
    improcess.py:395-420: antialiased bilinear interpolation (torchvision >= 0.17 default).  A bool
    image comes back as bool: True wherever a True pixel has non-zero weight (torchvision casts to
    float32, interpolates and casts back)."""
    if getattr(image, "ndim", 0) != 2:
        raise ValueError("image must be 2-D")
    oh, ow = int(image.shape[0] * fx), int(image.shape[1] * ft)
    if _is_bool(image):
        y = _resize_device(dev.to_device_f32(image), oh, ow) != 0
        return y if dev.is_tensor(image) else y.cpu().numpy()
    return dev.like_input(_resize_device(dev.to_device_f32(image), oh, ow), image)


def _mask_mul_device(x, m):
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.d4w_mask_mul_f32(dev.ptr(x), dev.ptr(m), dev.ptr(y), _n(x), dev.stream_ptr(x)))
    return y


def apply_smooth_mask(array, mask, sigma=1.5):
    """array * mask -- reference improcess.py:423-454.  The reference also blurs the mask with a
    Gaussian (sigma) and normalises it, but multiplies by the RAW mask (:452); the unused blur is not
    computed here."""
    if tuple(array.shape) != tuple(mask.shape):
        raise ValueError(f"operands could not be broadcast together with shapes {tuple(array.shape)} {tuple(mask.shape)}")
    x = dev.to_device_f32(array)
    m = dev.to_device_f32(mask, x.device)
    return dev.like_input(_mask_mul_device(x, m), array)


def gabor_mask(trf_fk, fs, dx, selected_channels, c0=1500., threshold=9100., threshold2=150., bin_factor=10):
    """scripts/main_gabordetect.py:78-166 in one call, all intermediates on the device.

    Returns a dict: "image" (trace2image), "imagebin", "fimage" (Gabor line score), "mask" (bool, binned
    grid), "mask_sparse" (bool, full grid) and "masked_tr" (trf_fk * mask_sparse).  NumPy input gives
    NumPy outputs (float arrays in the input dtype), CUDA tensors give CUDA tensors."""
    if getattr(trf_fk, "ndim", 0) != 2:
        raise ValueError("trf_fk must be a 2-D [channel x time] array")
    x = dev.to_device_f32(trf_fk)
    nx, ns = x.shape
    image = _trace2image_device(x)
    ratio = c0 / (fs * dx * selected_channels[2])
    up, down = gabor_filt_design(np.arctan(ratio) * 180 / np.pi)
    bh, bw = int(nx * (1 / bin_factor)), int(ns * (1 / bin_factor))
    imagebin = _resize_device(image, bh, bw)
    both = up + down                                        # filter2D is linear in the kernel: one pass for the pair
    fimage = _filter2d_device(imagebin, [both])
    binary = torch.empty_like(fimage)
    with torch.cuda.device(x.device):
        check(lib.d4w_threshold_f32(dev.ptr(fimage), dev.ptr(binary), _n(fimage), float(threshold), dev.stream_ptr(x)))
        score = _filter2d_device(binary, [both])
        mask = torch.empty_like(score)
        check(lib.d4w_threshold_f32(dev.ptr(score), dev.ptr(mask), _n(score), float(threshold2), dev.stream_ptr(x)))
    uh, uw = int(bh * bin_factor), int(bw * bin_factor)
    if (uh, uw) != (nx, ns):
        raise ValueError(f"operands could not be broadcast together with shapes ({nx},{ns}) ({uh},{uw})")
    # binning() of a bool mask is bool (True wherever a True pixel has non-zero weight, improcess.py:416-420): the
    # interpolated weights are binarised before the multiplication (array * bool mask, improcess.py:452)
    mask_sparse = (_resize_device(mask, uh, uw) != 0)
    masked = _mask_mul_device(x, mask_sparse.to(torch.float32))
    res = {"image": image, "imagebin": imagebin, "fimage": fimage, "score": score, "mask": mask != 0,
           "mask_sparse": mask_sparse, "masked_tr": masked}
    if dev.is_tensor(trf_fk):
        return res
    out = {}
    for k, v in res.items():
        out[k] = v.cpu().numpy() if v.dtype == torch.bool else dev.like_input(v, trf_fk)
    return out
