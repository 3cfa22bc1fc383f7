// Image operators of the Gabor detector on MI355X (gfx950) -- SURVEY.md 8(f) row f3:
//   * improcess.scale_pixels / trace2image (improcess.py:23-62): global min / max + affine map
//     (the envelope / std image itself is d4w_analytic_f32 mode 4);
//   * improcess.binning (improcess.py:395-420) = torchvision Resize = antialiased bilinear
//     interpolation (aten _upsample_bilinear2d_aa), separable: horizontal pass, then vertical pass;
//   * cv2.filter2D (scripts/main_gabordetect.py:109,132): 2-D correlation, anchor at the kernel
//     centre, BORDER_REFLECT_101, LDS-tiled with an 8-row register window per thread;
//   * the threshold steps (:122,133) and improcess.apply_smooth_mask (improcess.py:423-454).
// HBM-streaming work except filter2D, which is a dense FP32 contraction on a 100x smaller image
// (VALU-bound).  No MFMA.
#include <map>
#include <mutex>

#include "d4w_internal.h"

namespace d4w {

constexpr int kImThreads = 256;

// order-preserving float <-> unsigned key (atomicMin / atomicMax work on the keys)
__device__ __forceinline__ unsigned im_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float im_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// quiet NaN with the sign bit clear (its key is the largest there is; the negated one's the smallest)
__device__ __forceinline__ float kImNaN() { return __uint_as_float(0x7FC00000u); }
__device__ __forceinline__ float im_nanmin(float a, float b) { return (a != a) ? a : (b != b) ? b : fminf(a, b); }
__device__ __forceinline__ float im_nanmax(float a, float b) { return (a != a) ? a : (b != b) ? b : fmaxf(a, b); }

__global__ void minmax_init(unsigned* keys) {
    keys[0] = 0xffffffffu;
    keys[1] = 0u;
}

__global__ __launch_bounds__(kImThreads) void minmax_reduce(const float* __restrict__ x, size_t n, unsigned* keys) {
    __shared__ float red_lo[kImThreads / 64], red_hi[kImThreads / 64];
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * kImThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kImThreads) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        bad |= (v != v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    // np.min / np.max propagate NaN (fminf / fmaxf drop it): a NaN anywhere makes both results NaN
    bad = __any(bad) != 0;
    if ((threadIdx.x & 63) == 0) { red_lo[threadIdx.x >> 6] = bad ? -kImNaN() : lo; red_hi[threadIdx.x >> 6] = bad ? kImNaN() : hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = red_lo[0]; hi = red_hi[0];
        for (int w = 1; w < kImThreads / 64; ++w) { lo = im_nanmin(lo, red_lo[w]); hi = im_nanmax(hi, red_hi[w]); }
        atomicMin(&keys[0], im_key(lo));                 // (-NaN has the smallest key, +NaN the largest: either one sticks)
        atomicMax(&keys[1], im_key(hi));
    }
}

// small inputs (the per-row maxima a detection threshold is taken from: nx values): one workgroup, one launch
__global__ __launch_bounds__(kImThreads) void minmax_small(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float red_lo[kImThreads / 64], red_hi[kImThreads / 64];
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (size_t i = threadIdx.x; i < n; i += kImThreads) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        bad |= (v != v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    // np.min / np.max propagate NaN (fminf / fmaxf drop it): a NaN anywhere makes both results NaN
    bad = __any(bad) != 0;
    if ((threadIdx.x & 63) == 0) { red_lo[threadIdx.x >> 6] = bad ? -kImNaN() : lo; red_hi[threadIdx.x >> 6] = bad ? kImNaN() : hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = red_lo[0]; hi = red_hi[0];
        for (int w = 1; w < kImThreads / 64; ++w) { lo = im_nanmin(lo, red_lo[w]); hi = im_nanmax(hi, red_hi[w]); }
        // through the same order-preserving keys as the three-launch form (what a NaN or a -0 turns into is the same)
        out[0] = im_unkey(im_key(lo));
        out[1] = im_unkey(im_key(hi));
    }
}

__global__ void minmax_decode(unsigned* keys) {
    float* f = reinterpret_cast<float*>(keys);
    const float lo = im_unkey(keys[0]), hi = im_unkey(keys[1]);
    f[0] = lo;
    f[1] = hi;
}

// y = (x - lo) / (hi - lo) * gain      (improcess.py:39 then :61)
__global__ __launch_bounds__(kImThreads) void scale_pixels(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                           const float* __restrict__ minmax, float gain) {
    const float lo = minmax[0], inv = 1.0f / (minmax[1] - minmax[0]);
    for (size_t i = (size_t)blockIdx.x * kImThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kImThreads)
        y[i] = (x[i] - lo) * inv * gain;
}

// y = x > thr ? 1 : 0 (compared in float64 like the reference's float64 image against a Python float)
__global__ __launch_bounds__(kImThreads) void threshold_gt(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                           double thr) {
    for (size_t i = (size_t)blockIdx.x * kImThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kImThreads)
        y[i] = ((double)x[i] > thr) ? 1.0f : 0.0f;
}

// y = x * m    (array * mask, improcess.py:452; bool masks arrive as 0 / 1 floats)
__global__ __launch_bounds__(kImThreads) void mask_mul(const float* __restrict__ x, const float* __restrict__ m,
                                                       float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kImThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kImThreads)
        y[i] = x[i] * m[i];
}

// ---------------------------------------------------------------------------------------------
// antialiased bilinear resize: per-axis tables (first input index, tap count, normalised weights)
// built on the host in float64 exactly as aten does, cached per (device, in, out)
// ---------------------------------------------------------------------------------------------
struct ResizeAxisDev {
    const int* first;     // [out]
    const int* count;     // [out]
    const float* w;       // [out][kmax]
    int kmax;
};

struct ResizeAxisHost {
    ResizeAxisDev dev;
    std::vector<void*> allocs;
};

static std::mutex g_resize_mu;
static std::map<std::tuple<int, int, int>, ResizeAxisHost*> g_resize;

static int resize_axis_get(int in_size, int out_size, ResizeAxisDev* out) {
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    std::lock_guard<std::mutex> lk(g_resize_mu);
    auto key = std::make_tuple(devid, in_size, out_size);
    auto it = g_resize.find(key);
    if (it != g_resize.end()) { *out = it->second->dev; return D4W_OK; }
    // aten/native/cpu/UpSampleKernel.cpp (_compute_indices_weights_aa, triangle filter, align_corners = false)
    const double scale = (double)in_size / (double)out_size;
    const double support = scale >= 1.0 ? scale : 1.0;
    const double invscale = scale >= 1.0 ? 1.0 / scale : 1.0;
    const int kmax = (int)std::ceil(support) * 2 + 1;
    std::vector<int> first(out_size), count(out_size);
    std::vector<float> w((size_t)out_size * kmax, 0.f);
    std::vector<double> tmp(kmax);
    for (int i = 0; i < out_size; ++i) {
        const double center = scale * (i + 0.5);
        const int xmin = std::max((int)(long long)(center - support + 0.5), 0);
        const int xsize = std::min((int)(long long)(center + support + 0.5), in_size) - xmin;
        double total = 0.0;
        for (int j = 0; j < xsize; ++j) {
            const double t = std::fabs((j + xmin - center + 0.5) * invscale);
            tmp[j] = t < 1.0 ? 1.0 - t : 0.0;
            total += tmp[j];
        }
        first[i] = xmin;
        count[i] = xsize;
        for (int j = 0; j < xsize; ++j) w[(size_t)i * kmax + j] = (float)(tmp[j] / total);
    }
    ResizeAxisHost* h = new ResizeAxisHost();
    auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
        void* p = nullptr;
        D4W_HIP(hipMalloc(&p, std::max<size_t>(bytes, 4)));
        h->allocs.push_back(p);
        D4W_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        *dst = p;
        return D4W_OK;
    };
    int rc;
    if ((rc = up(first.data(), first.size() * sizeof(int), (const void**)&h->dev.first)) ||
        (rc = up(count.data(), count.size() * sizeof(int), (const void**)&h->dev.count)) ||
        (rc = up(w.data(), w.size() * sizeof(float), (const void**)&h->dev.w))) {
        for (void* p : h->allocs) (void)hipFree(p);
        delete h;
        return rc;
    }
    h->dev.kmax = kmax;
    g_resize[key] = h;
    *out = h->dev;
    return D4W_OK;
}

// horizontal pass: t[r][ox] = sum_j w[ox][j] x[r][first[ox] + j]
// When the axis shrinks, neighbouring outputs read windows `scale` samples apart: a lane-per-output gather touches scale x 4
// bytes per lane and every line ~kmax times (1.0 ms for an 11 020 x 12 000 -> 1200 binning).  The span of a workgroup's
// outputs is therefore staged in LDS with coalesced loads first (four in flight per lane) whenever it fits.
constexpr int kRsSpan = 4096;
__global__ __launch_bounds__(kImThreads) void resize_rows(const float* __restrict__ x, int w, float* __restrict__ t, int ow,
                                                          ResizeAxisDev A) {
    __shared__ float seg[kRsSpan];
    const int ox0 = blockIdx.x * kImThreads, ox = ox0 + threadIdx.x;
    const int oxl = min(ox0 + kImThreads, ow) - 1;
    const int s0 = A.first[ox0], span = A.first[oxl] + A.count[oxl] - s0;
    const float* row = x + (size_t)blockIdx.y * w;
    const bool staged = span <= kRsSpan;                             // block-uniform
    if (staged) {
        constexpr int kAhead = 4;
        for (int i0 = threadIdx.x; i0 < span; i0 += kAhead * kImThreads) {
            float q[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int i = i0 + k * kImThreads;
                q[k] = (i < span) ? row[s0 + i] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int i = i0 + k * kImThreads;
                if (i < span) seg[i] = q[k];
            }
        }
        __syncthreads();
    }
    if (ox >= ow) return;
    const float* wt = A.w + (size_t)ox * A.kmax;
    const int n = A.count[ox], f = A.first[ox];
    float acc = 0.f;
    if (staged) {
        const float* sp = seg + (f - s0);
        for (int j = 0; j < n; ++j) acc = fmaf(wt[j], sp[j], acc);
    } else {
        for (int j = 0; j < n; ++j) acc = fmaf(wt[j], row[f + j], acc);
    }
    t[(size_t)blockIdx.y * ow + ox] = acc;
}

// vertical pass: y[oy][ox] = sum_j w[oy][j] t[first[oy] + j][ox]
__global__ __launch_bounds__(kImThreads) void resize_cols(const float* __restrict__ t, int ow, float* __restrict__ y,
                                                          ResizeAxisDev A) {
    const int ox = blockIdx.x * kImThreads + threadIdx.x;
    if (ox >= ow) return;
    const int oy = blockIdx.y;
    const float* col = t + (size_t)A.first[oy] * ow + ox;
    const float* wt = A.w + (size_t)oy * A.kmax;
    const int n = A.count[oy];
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(wt[j], col[(size_t)j * ow], acc);
    y[(size_t)oy * ow + ox] = acc;
}

// ---------------------------------------------------------------------------------------------
// filter2D: out[y][x] (+)= sum_{ky,kx} K[ky][kx] img[y + ky - kh/2][x + kx - kw/2], reflect-101
//   workgroup = 32 x 64 outputs, 16 waves; the (32 + kh - 1) x (64 + kw - 1) input patch sits in
//   LDS.  Wave (g, s): output rows 8g .. 8g+7 and every 4th kernel column starting at s (the four
//   partial sums meet in LDS at the end: 16 waves per CU hide the LDS latency one patch per CU
//   would otherwise expose).  A thread owns 8 consecutive output rows of one column and walks down
//   its column once per kernel column: one LDS read feeds 8 FMAs whose kernel values K[i - r][kx]
//   are a sliding window of the zero-padded, transposed kernel Kp[kx][7 + ky] held in SGPRs; the
//   walk is unrolled by 8 rows (8 LDS reads + one 16-dword scalar load in flight per 64 FMAs).
// ---------------------------------------------------------------------------------------------
constexpr int kF2Rows = 8;                      // output rows per thread
constexpr int kF2TileH = 4 * kF2Rows;           // 4 row groups
constexpr int kF2TileW = 64;
constexpr int kF2Split = 4;                     // kernel-column splits
constexpr int kF2Threads = 64 * 4 * kF2Split;

__host__ __device__ inline int f2_walk(int kh) { return (kh + kF2Rows - 1 + 7) / 8 * 8; }     // rows a thread walks
__host__ __device__ inline int f2_kp(int kh) { return f2_walk(kh) + 8; }                      // padded kernel column

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

// Kp[kx][j], j in [0, f2_kp): K[j - (R - 1)][kx] inside, 0 outside
__global__ void f2_pad_kernel(const float* __restrict__ K, int kh, int kw, float* __restrict__ Kp) {
    const int kp = f2_kp(kh);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kw * kp; i += gridDim.x * blockDim.x) {
        const int kx = i / kp, ky = i % kp - (kF2Rows - 1);
        Kp[i] = (ky >= 0 && ky < kh) ? K[ky * kw + kx] : 0.f;
    }
}

template <bool ACC>
__global__ __launch_bounds__(kF2Threads) void filter2d_tile(const float* __restrict__ img, int h, int w,
                                                            const float* __restrict__ Kp, int kh, int kw,
                                                            float* __restrict__ out) {
    D4W_DYN_LDS(smem_raw);
    float* tile = reinterpret_cast<float*>(smem_raw);
    const int pw = kF2TileW + kw - 1, ph = kF2TileH + kh - 1;
    const int walk = f2_walk(kh), kp = f2_kp(kh);
    const int prows = kF2TileH - kF2Rows + walk;       // patch rows incl. the zero rows the unrolled walk touches
    const int x0 = blockIdx.x * kF2TileW - kw / 2, y0 = blockIdx.y * kF2TileH - kh / 2;
    for (int e = threadIdx.x; e < pw * prows; e += kF2Threads) {
        const int ty = e / pw, tx = e % pw;
        tile[e] = (ty < ph) ? img[(size_t)reflect101(y0 + ty, h) * w + reflect101(x0 + tx, w)] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: taps via SMEM
    const int ty = (wid & 3) * kF2Rows, split = wid >> 2;
    float acc[kF2Rows];
#pragma unroll
    for (int r = 0; r < kF2Rows; ++r) acc[r] = 0.f;
    for (int kx = split; kx < kw; kx += kF2Split) {
        const float* kcol = Kp + kx * kp;
        const float* tcol = tile + ty * pw + tx + kx;
        for (int i0 = 0; i0 < walk; i0 += 8) {
            float v[8], k[15];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = tcol[(i0 + u) * pw];
#pragma unroll
            for (int j = 0; j < 15; ++j) k[j] = kcol[i0 + j];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < kF2Rows; ++r) acc[r] = fmaf(k[u + (kF2Rows - 1) - r], v[u], acc[r]);
        }
    }
    __syncthreads();                                    // the patch is dead: its LDS holds the partial sums
    float* red = tile;                                  // [split][32 rows][64]
#pragma unroll
    for (int r = 0; r < kF2Rows; ++r) red[(split * kF2TileH + ty + r) * kF2TileW + tx] = acc[r];
    __syncthreads();
    if (split != 0) return;
    const int ox = blockIdx.x * kF2TileW + tx;
    if (ox >= w) return;
#pragma unroll
    for (int r = 0; r < kF2Rows; ++r) {
        const int oy = blockIdx.y * kF2TileH + ty + r;
        if (oy < h) {
            float sum = acc[r];
#pragma unroll
            for (int q = 1; q < kF2Split; ++q) sum += red[(q * kF2TileH + ty + r) * kF2TileW + tx];
            float* o = out + (size_t)oy * w + ox;
            *o = ACC ? *o + sum : sum;
        }
    }
}

static inline int im_grid(size_t n) { return (int)std::min<size_t>((n + kImThreads - 1) / kImThreads, 4096); }

}  // namespace d4w

using namespace d4w;

extern "C" {

int d4w_minmax_f32(const float* x, size_t n, float* minmax, void* stream) {
    if (!x || !minmax || n < 1) return fail(D4W_EINVAL, "bad argument");
    if (n <= (size_t)64 * 1024) {                                 // (round 5: one launch instead of three)
        D4W_LAUNCH(minmax_small, dim3(1), dim3(kImThreads), 0, stream, x, n, minmax);
        return D4W_OK;
    }
    unsigned* keys = reinterpret_cast<unsigned*>(minmax);
    D4W_LAUNCH(minmax_init, dim3(1), dim3(1), 0, stream, keys);
    // >= 4096 values per workgroup: every workgroup ends with two atomics on the same pair of words, and 4096 workgroups of
    // them took 0.1 ms on a 5-MB image (12 ns per contended atomic) where the read itself takes a few microseconds
    D4W_LAUNCH(minmax_reduce, dim3((int)std::min<size_t>(std::max<size_t>(n / 4096, 1), 2048)), dim3(kImThreads), 0, stream, x, n, keys);
    D4W_LAUNCH(minmax_decode, dim3(1), dim3(1), 0, stream, keys);
    return D4W_OK;
}

int d4w_scale_pixels_f32(const float* x, float* y, size_t n, const float* minmax, double gain, void* stream) {
    if (!x || !y || !minmax || n < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(scale_pixels, dim3(im_grid(n)), dim3(kImThreads), 0, stream, x, y, n, minmax, (float)gain);
    return D4W_OK;
}

int d4w_threshold_f32(const float* x, float* y, size_t n, double thr, void* stream) {
    if (!x || !y || n < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(threshold_gt, dim3(im_grid(n)), dim3(kImThreads), 0, stream, x, y, n, thr);
    return D4W_OK;
}

int d4w_mask_mul_f32(const float* x, const float* mask, float* y, size_t n, void* stream) {
    if (!x || !mask || !y || n < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(mask_mul, dim3(im_grid(n)), dim3(kImThreads), 0, stream, x, mask, y, n);
    return D4W_OK;
}

size_t d4w_resize_ws_bytes(int h, int w, int oh, int ow) {
    (void)w; (void)oh;
    return (h > 0 && ow > 0) ? (size_t)h * ow * sizeof(float) : 0;
}

int d4w_resize_bilinear_aa_f32(const float* x, int h, int w, float* y, int oh, int ow, void* ws, void* stream) {
    if (!x || !y || !ws || h < 1 || w < 1 || oh < 1 || ow < 1) return fail(D4W_EINVAL, "bad argument");
    if (h > 65535 || oh > 65535) return fail(D4W_EINVAL, "image height %d / %d exceeds the grid limit 65535", h, oh);
    ResizeAxisDev ax, ay;
    int rc;
    if ((rc = resize_axis_get(w, ow, &ax)) || (rc = resize_axis_get(h, oh, &ay))) return rc;
    float* t = (float*)ws;
    D4W_LAUNCH(resize_rows, dim3((ow + kImThreads - 1) / kImThreads, h), dim3(kImThreads), 0, stream, x, w, t, ow, ax);
    D4W_LAUNCH(resize_cols, dim3((ow + kImThreads - 1) / kImThreads, oh), dim3(kImThreads), 0, stream, (const float*)t, ow, y, ay);
    return D4W_OK;
}

static size_t f2_direct_ws_bytes(int kh, int kw) { return ((size_t)kw * f2_kp(kh) * sizeof(float) + 255) & ~(size_t)255; }

size_t d4w_filter2d_ws_bytes(int kh, int kw) {
    if (kh <= 0 || kw <= 0) return 0;
    // the padded kernel of the direct form, then (kernels of <= 113 columns) the Toeplitz fragment table of the matrix-core form
    return f2_direct_ws_bytes(kh, kw) + (d4w_filter2d_mm_eligible(kh, kw) ? d4w_filter2d_mm_ws_bytes(kh, kw) : 0);
}

int d4w_filter2d_f32(const float* img, int h, int w, const float* kernel, int kh, int kw, float* out, int accumulate,
                     void* ws, void* stream) {
    if (!img || !kernel || !out || !ws || h < 1 || w < 1 || kh < 1 || kw < 1) return fail(D4W_EINVAL, "bad argument");
    if (img == out) return fail(D4W_EINVAL, "filter2d cannot run in place");
    // kernels of <= 113 columns (the detector's 101 x 101 Gabor pair): row-by-row Toeplitz products on the matrix cores
    if (d4w_filter2d_mm_eligible(kh, kw))
        return d4w_filter2d_mm_f32(img, h, w, kernel, kh, kw, out, accumulate, (char*)ws + f2_direct_ws_bytes(kh, kw), stream);
    const size_t patch = (size_t)(kF2TileW + kw - 1) * (kF2TileH - kF2Rows + f2_walk(kh)) * sizeof(float);
    const size_t lds = std::max(patch, (size_t)kF2Split * kF2TileH * kF2TileW * sizeof(float));
    if (lds > 160 * 1024) return fail(D4W_EINVAL, "kernel %d x %d needs %zu bytes of LDS (limit 163840)", kh, kw, lds);
    float* Kp = (float*)ws;
    D4W_LAUNCH(f2_pad_kernel, dim3(64), dim3(kImThreads), 0, stream, kernel, kh, kw, Kp);
    const dim3 grid((w + kF2TileW - 1) / kF2TileW, (h + kF2TileH - 1) / kF2TileH);
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)filter2d_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)filter2d_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (accumulate) D4W_LAUNCH(filter2d_tile<true>, grid, dim3(kF2Threads), lds, stream, img, h, w, (const float*)Kp, kh, kw, out);
    else D4W_LAUNCH(filter2d_tile<false>, grid, dim3(kF2Threads), lds, stream, img, h, w, (const float*)Kp, kh, kw, out);
    return D4W_OK;
}

}  // extern "C"
