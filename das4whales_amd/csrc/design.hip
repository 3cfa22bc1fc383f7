// f-k mask design on the device (one-off per shape): closed-form, loop-free restatements of
// dsp.fk_filter_design / hybrid_filter_design / hybrid_ninf_filter_design (reference dsp.py:85-454,
// SURVEY.md A.8), the box indicators of the *_gs designs and of dsp.fk_filt (dsp.py:457-702,
// 883-953), the separable sigma-20 Gaussian (scipy.ndimage.gaussian_filter, reflect boundary) and
// the min/max used by fk_filt's normalisation.  Axis values and tapers are evaluated in float64
// exactly as NumPy forms them (integer index x 1/(n d)); masks are stored as float32 on the
// fftshift-ed (k, f) grid, row-major [nx][ns] -- the layout d4w_fk_set_mask_dense_f32 takes.
#include <algorithm>
#include <limits>
#include <map>
#include <mutex>

#include "d4w_internal.h"
#include "design_eval.h"

namespace d4w {

__global__ __launch_bounds__(256) void design_kernel(DesignArgs A, int mode, float* __restrict__ out) {
    const int i = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < A.ns; j += gridDim.x * blockDim.x)
        out[(size_t)i * A.ns + j] = (float)design_value(A, mode, i, j);
}

// out = in + fliplr(in) (+ flipud of that)  -- the post-blur flips of hybrid_ninf_gs, dsp.py:660-661
__global__ __launch_bounds__(256) void flip_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int nx, int ns) {
    const int i = blockIdx.y, ri = nx - 1 - i;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ns; j += gridDim.x * blockDim.x) {
        const int rj = ns - 1 - j;
        out[(size_t)i * ns + j] = (in[(size_t)i * ns + j] + in[(size_t)i * ns + rj]) +
                                  (in[(size_t)ri * ns + j] + in[(size_t)ri * ns + rj]);
    }
}

// scipy.ndimage reflect boundary: (d c b a | a b c d | d c b a)
__device__ __forceinline__ int reflect_idx(int i, int n) {
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return (i < n) ? i : period - 1 - i;
}

constexpr int kBlurMaxR = 128;
struct BlurTaps {
    int radius;
    float w[2 * kBlurMaxR + 1];
};

// Gaussian along the contiguous axis: one workgroup per (row, 1024-column chunk), chunk + halo staged in LDS (reflected
// indices, four loads in flight per lane); a lane forms FOUR consecutive outputs from 16-byte LDS pieces -- every piece
// feeds 16 FMAs against the 7 taps it meets (wave-uniform scalars) -- a quarter of the LDS reads of one output per lane,
// which bound the first version (20.9 ms for a 20 000 x 120 000 mask: 161 LDS reads per output).  Taps in ascending
// order per output, as scipy.ndimage.correlate1d sums them.
constexpr int kBlurChunk = 1024;
__global__ __launch_bounds__(256) void blur_rows_kernel(BlurTaps T, const float* __restrict__ in,
                                                        float* __restrict__ out, int nx, int ns) {
    __shared__ __attribute__((aligned(16))) float buf[kBlurChunk + 2 * kBlurMaxR + 8];
    const int i = blockIdx.y, j0 = blockIdx.x * kBlurChunk, R = T.radius;
    const float* row = in + (size_t)i * ns;
    const int span = kBlurChunk + 2 * R;
    constexpr int kAhead = 4;
    for (int t0 = threadIdx.x; t0 < span; t0 += kAhead * 256) {
        float q[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int t = t0 + k * 256;
            q[k] = (t < span) ? row[reflect_idx(j0 - R + t, ns)] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int t = t0 + k * 256;
            if (t < span) buf[t] = q[k];
        }
    }
    for (int t = span + threadIdx.x; t < span + 8; t += 256) buf[t] = 0.f;       // the last pieces read a few floats past the span
    __syncthreads();
    const int nk = 2 * R + 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float4* sp4 = reinterpret_cast<const float4*>(buf) + threadIdx.x;       // outputs 4 tid .. 4 tid + 3 of the chunk
    for (int c = 0; 4 * c < nk + 3; ++c) {
        const float4 v4 = sp4[c];
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            const int j = 4 * c - 3 + d;                              // tap index: sample e of the piece meets output q at j = 4c + e - q
            if (j >= 0 && j < nk) {
                const float tap = T.w[j];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = d - 3 + q;
                    if (e >= 0 && e < 4) acc[q] = fmaf(tap, v[e], acc[q]);
                }
            }
        }
    }
    const int j = j0 + 4 * (int)threadIdx.x;
    float* o = out + (size_t)i * ns + j;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (j + q < ns) o[q] = acc[q];
}

// the K samples before and after every row under scipy's "reflect" rule, as the halos d4w_fir_fft_halo_f32 reads
__global__ __launch_bounds__(256) void reflect_halo_kernel(const float* __restrict__ in, int rows, int cols, int K,
                                                           float* __restrict__ left, float* __restrict__ right) {
    const size_t n = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / K), j = (int)(i - (size_t)r * K);
        const float* row = in + (size_t)r * cols;
        left[i] = row[reflect_idx(j - K, cols)];
        right[i] = row[reflect_idx(cols + j, cols)];
    }
}

// [rows][cols] -> [cols][rows] through a 64 x 64 LDS tile (the Gaussian along the strided axis runs as two transposes
// around the row kernel: the column kernel it replaces read every input row 21 times -- 123 of the 147 ms of a
// 20 000 x 120 000 *_gs design)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * k][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][ty + 4 * k];
    }
}

// per-block min / max partials (finished on the host: a few hundred floats)
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    __shared__ float smin[4], smax[4];
    float lo = 3.4e38f, hi = -3.4e38f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x / 64] = lo;
        smax[threadIdx.x / 64] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ __launch_bounds__(256) void affine_kernel(float* __restrict__ x, size_t n, float a, float b) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = fmaf(x[i], a, b);
}

}  // namespace d4w

using namespace d4w;

// per-device scratch of the FFT form of the blur: tap vector, overlap-save workspace, the two halo strips (grow-only)
struct BlurScratch {
    float* taps = nullptr; std::vector<float> host_taps;
    void* ws = nullptr;
    float* halo = nullptr; size_t halo_floats = 0;
};
static std::mutex g_blur_mu;
static std::map<int, BlurScratch> g_blur;

// Gaussian along the contiguous axis of [rows][cols] as ONE overlap-save FFT pass (d4w_fir_fft_halo_f32: 8 B per value
// instead of 161 FMAs per value -- the direct kernels are bound by VALU / LDS issue: >= 11 ms per axis at
// 20 000 x 120 000); the reflect rule enters as halo strips.  Rows shorter than an FFT block keep the direct kernel.
static int blur_axis(const BlurTaps& T, const float* in, float* out, int rows, int cols, void* stream) {
    const int R = T.radius, K = R + (R & 1);
    if (cols < 2048 || K > d4w_fir_fft_max_halfwidth()) {
        for (int r0 = 0; r0 < rows; r0 += 65535) {                    // grid.y limit: slabs of rows
            const int nr = std::min(65535, rows - r0);
            D4W_LAUNCH(blur_rows_kernel, dim3(ceil_div(cols, kBlurChunk), nr), dim3(256), 0, stream, T, in + (size_t)r0 * cols,
                       out + (size_t)r0 * cols, nr, cols);
        }
        return D4W_OK;
    }
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    std::lock_guard<std::mutex> lk(g_blur_mu);
    BlurScratch& S = g_blur[devid];
    std::vector<float> taps(2 * K + 1, 0.f);
    for (int j = 0; j <= 2 * R; ++j) taps[j + (K - R)] = T.w[j];
    if (!S.taps) D4W_HIP(hipMalloc((void**)&S.taps, (2 * 1024 + 1) * sizeof(float)));
    if (!S.ws) D4W_HIP(hipMalloc(&S.ws, d4w_xcorr_fft_ws_bytes()));
    if (S.host_taps != taps) {
        D4W_HIP(hipStreamSynchronize((hipStream_t)stream));           // an earlier blur on this stream may still read the old taps
        D4W_HIP(hipMemcpy(S.taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
        S.host_taps = taps;
    }
    for (int r0 = 0; r0 < rows; r0 += 2 * 65535) {                    // the FIR kernel's grid limit: slabs of rows
        const int nr = std::min(2 * 65535, rows - r0);
        const size_t need = 2 * (size_t)nr * K;
        if (S.halo_floats < need) {
            D4W_HIP(hipStreamSynchronize((hipStream_t)stream));
            if (S.halo) (void)hipFree(S.halo);
            S.halo = nullptr; S.halo_floats = 0;
            D4W_HIP(hipMalloc((void**)&S.halo, need * sizeof(float)));
            S.halo_floats = need;
        }
        float* left = S.halo;
        float* right = S.halo + (size_t)nr * K;
        const float* src = in + (size_t)r0 * cols;
        D4W_LAUNCH(reflect_halo_kernel, dim3((unsigned)std::min<size_t>(((size_t)nr * K + 255) / 256, 4096)), dim3(256), 0, stream, src, nr,
                   cols, K, left, right);
        int rc = d4w_fir_fft_halo_f32(src, nr, cols, left, K, K, right, K, K, S.taps, K, nullptr, 0.0, out + (size_t)r0 * cols, S.ws, stream);
        if (rc) return rc;
    }
    return D4W_OK;
}

extern "C" {

int d4w_design_mask_f32(int mode, int nx, int ns, double k_spacing, double t_spacing, const double* params8_host,
                        int i0, int i1, const double* hrow_dev, float* mask, void* stream) {
    if (!mask || !params8_host || nx < 1 || ns < 1 || mode < 0 || mode > 5) return fail(D4W_EINVAL, "bad argument");
    if (mode == 2 && !hrow_dev) return fail(D4W_EINVAL, "hybrid_ninf needs the |H|^2 row");
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    const DesignArgs A = make_design_args(nx, ns, k_spacing, t_spacing, params8_host, i0, i1, hrow_dev);
    D4W_LAUNCH(design_kernel, dim3(std::min(ceil_div(ns, 256), 64), nx), dim3(256), 0, stream, A, mode, mask);
    return D4W_OK;
}

int d4w_flip_sum_f32(const float* in, float* out, int nx, int ns, void* stream) {
    if (!in || !out || in == out || nx < 1 || ns < 1 || nx > 65535) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(flip_sum_kernel, dim3(std::min(ceil_div(ns, 256), 64), nx), dim3(256), 0, stream, in, out, nx, ns);
    return D4W_OK;
}

int d4w_gaussian_filter_f32(const float* in, float* out, float* tmp, int nx, int ns, double sigma, void* stream) {
    if (!in || !out || !tmp || nx < 1 || ns < 1 || sigma <= 0) return fail(D4W_EINVAL, "bad argument");
    const int R = (int)(4.0 * sigma + 0.5);        // scipy: int(truncate * sd + 0.5), truncate = 4
    if (R > kBlurMaxR) return fail(D4W_EINVAL, "sigma = %g needs radius %d > %d", sigma, R, kBlurMaxR);
    if (ceil_div(nx, 8) > 65535 || nx > 65535) return fail(D4W_EINVAL, "nx too large");
    BlurTaps T;
    T.radius = R;
    double sum = 0.0;
    std::vector<double> w(2 * R + 1);
    for (int i = -R; i <= R; ++i) { w[i + R] = exp(-0.5 * (double)i * i / (sigma * sigma)); sum += w[i + R]; }
    for (int i = 0; i <= 2 * R; ++i) T.w[i] = (float)(w[i] / sum);
    // scipy filters axis 0 first, then axis 1: transpose, rows of the transposed grid, transpose back, rows
    if (ceil_div(nx, 64) > 65535 || ceil_div(ns, 64) > 65535 || ns > 65535 * 64)
        return fail(D4W_EINVAL, "grid %d x %d too large for the blur's launch grids", nx, ns);
    int rc;
    D4W_LAUNCH(transpose_kernel, dim3(ceil_div(ns, 64), ceil_div(nx, 64)), dim3(256), 0, stream, in, tmp, nx, ns);       // tmp = [ns][nx]
    if ((rc = blur_axis(T, (const float*)tmp, out, ns, nx, stream))) return rc;
    D4W_LAUNCH(transpose_kernel, dim3(ceil_div(nx, 64), ceil_div(ns, 64)), dim3(256), 0, stream, (const float*)out, tmp, ns, nx);   // tmp = [nx][ns]
    if ((rc = blur_axis(T, (const float*)tmp, out, nx, ns, stream))) return rc;
    return D4W_OK;
}

/* (x - min) / (max - min) in place (dsp.fk_filt, dsp.py:945); synchronises the stream */
int d4w_minmax_normalise_f32(float* x, size_t n, void* stream) {
    if (!x || n < 1) return fail(D4W_EINVAL, "bad argument");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    float* part = nullptr;
    D4W_HIP(hipMalloc((void**)&part, sizeof(float) * 2 * blocks));
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, part);
    std::vector<float> h(2 * blocks);
    hipError_t e = hipMemcpyAsync(h.data(), part, sizeof(float) * 2 * blocks, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(part);
    if (e != hipSuccess) return fail(D4W_EHIP, "min/max reduction failed: %s", hipGetErrorString(e));
    float lo = h[0], hi = h[1];
    for (int b = 1; b < blocks; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
    // a constant g: NumPy's (g - min) / (max - min) is 0 / 0 = NaN everywhere (the reference's own test_fk_filt
    // runs into this on its 2 x 5 block and only checks the shape) -- same here, no error
    const float a = (hi > lo) ? 1.0f / (hi - lo) : std::numeric_limits<float>::quiet_NaN();
    D4W_LAUNCH(affine_kernel, dim3(blocks), dim3(256), 0, stream, x, n, a, -lo * a);
    return D4W_OK;
}

}  // extern "C"
