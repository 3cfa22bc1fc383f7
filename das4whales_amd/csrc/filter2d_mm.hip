// cv2.filter2D of the Gabor detector on the matrix cores (gfx950): replaces the two cv2.filter2D calls per pass of
// scripts/main_gabordetect.py:109,132 (correlation with a 101 x 101 kernel, anchor at the kernel centre, BORDER_REFLECT_101).
//
// 10 201 multiply-adds per pixel: the direct kernel (image.hip: filter2d_tile) runs at the vector ALUs' FMA rate (39 Tflop/s,
// 0.7 ms per 1102 x 1200 image), which made the two filter passes the longest step of the Gabor mask.  Row by row of the
// kernel the correlation is the banded-Toeplitz product of xcorr_mm.hip,
//
//      out[y][16 a + i] += sum_u K[j][u - i] * P[y + j][16 a + u],      j < kh,  i < 16,  u < 16 + kw - 1,
//
// (P = the image with its reflected border): for every kernel row j a 16 x K Toeplitz operand (4 k-steps of 32 for kw <= 113)
// against 16 windows of an image row.  A workgroup owns TH = 4 output rows over the whole width (one wave per 256 columns);
// it walks the kernel rows j = 0 .. kh - 1, keeps the TH + 1 image rows it needs in an LDS ring as binary16 hi / lo pairs
// (row y0 + j + TH is converted while step j multiplies: one barrier per step), fetches row j's Toeplitz fragments from a
// table built once per call (kh x 4 k-steps x hi / lo x 1 KiB, L2-resident) and issues 4 rows x 4 k-steps x 3 split products.
// 6 M matrix instructions per image = 0.04 ms of the matrix pipe; the pass is bound by the step chain (101 barriers, ~1.3
// waves per SIMD for a 1102-row image): 0.30 ms against 0.71 ms for the direct kernel (profiles/r04g/).
// Operand split and accumulation as in mm_common.h; the image is scaled by a power of two taken from its min / max (a pre-pass
// over the 5-MB image), the kernel by a power of two of its largest tap.
#include <cstdlib>

#include "mm_common.h"

namespace d4w {

constexpr int kFmTH = 4;                         // output rows per workgroup
constexpr int kFmKS = 4;                         // k-steps per kernel row: kw + 15 <= 128
constexpr int kFmMaxW = 8;                       // waves (256-column tiles) per workgroup
constexpr int kFmHalo = 32 * kFmKS;              // columns staged beyond the tiles

struct FmArgs {
    const float* img;
    float* out;
    const mm_h8* frag;       // [kh][kFmKS][2 (hi, lo)][64 lanes]
    const float* scales;     // [0] = power of two >= max|K|, [1] = image min, [2] = image max (d4w_minmax_f32 layout at [1])
    int h, w, kh, kw, accumulate, ntile;     // ntile = 256-column tiles per workgroup (= waves)
};

// Toeplitz fragments of every kernel row: frag[j][kk][hl][lane (i = lane & 15, g = lane >> 4)][jj] = K[j][32 kk + 8 g + jj - i] / ks
__global__ __launch_bounds__(256) void fm_build_frags(const float* __restrict__ K, int kh, int kw, mm_h8* __restrict__ frag,
                                                      float* __restrict__ scales) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    float m = 0.f;
    for (int i = tid; i < kh * kw; i += 256) m = fmaxf(m, fabsf(K[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (m > 0.f) (void)frexpf(m, &e);
    e = min(max(e, -100), 100);
    const float up = ldexpf(1.0f, e), down = ldexpf(1.0f, -e);
    if (blockIdx.x == 0 && tid == 0) scales[0] = up;
    const int total = kh * kFmKS * 64;
    for (int t = blockIdx.x * 256 + tid; t < total; t += gridDim.x * 256) {
        const int l = t & 63, kk = (t >> 6) % kFmKS, j = t / (64 * kFmKS);
        const int i = l & 15, g = l >> 4;
        mm_h8 fh, fl;
        static_for<8>([&](auto jq) {
            constexpr int jj = decltype(jq)::value;
            const int u = 32 * kk + 8 * g + jj - i;
            const float v = (u >= 0 && u < kw) ? K[j * kw + u] * down : 0.f;
            mm_half hi, lo;
            mm_split(v, hi, lo);
            mm_set(fh, jj, hi);
            mm_set(fl, jj, lo);
        });
        frag[((size_t)(j * kFmKS + kk) * 2 + 0) * 64 + l] = fh;
        frag[((size_t)(j * kFmKS + kk) * 2 + 1) * 64 + l] = fl;
    }
}

__device__ __forceinline__ int fm_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(64 * kFmMaxW) void filter2d_mm_rows(FmArgs P) {
    D4W_DYN_LDS(smem_raw);
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int lane = tid & 63, wv = mm_uniform(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int h = P.h, w = P.w, kh = P.kh, ay = kh / 2, ax = P.kw / 2;
    const int cols = 256 * P.ntile + kFmHalo;                      // staged columns of a row (multiple of 8)
    mm_half* ring = reinterpret_cast<mm_half*>(smem_raw);          // [kFmTH + 1 slots][hi | lo][cols]
    const int x0 = (int)blockIdx.x * 256 * P.ntile, y0 = (int)blockIdx.y * kFmTH;
    // image scale: a power of two >= max(|min|, |max|)
    float iup, idown;
    {
        const float m = fmaxf(fabsf(P.scales[1]), fabsf(P.scales[2]));
        int e = 0;
        if (m > 0.f) (void)frexpf(m, &e);
        e = min(max(e, -100), 100);
        iup = ldexpf(1.0f, e);
        idown = ldexpf(1.0f, -e);
    }
    // image row y0 + r - ay (reflected) -> ring slot r mod (TH + 1): a thread owns the staged columns tid, tid + nthr, ...
    // (at most kFmNPT of them).  Loading (global, latency) and converting (LDS) are separate steps so that the row a step
    // adds is requested one step ahead, under the previous step's products.
    constexpr int kFmNPT = 6;                                      // (256 n + 128) / (64 n) <= 6
    constexpr int kFmPF = 4;                                       // ring of register sets: what step j uses was requested at step j - 3
    float raw[kFmPF][kFmNPT];
    int cidx[kFmNPT];                                              // reflected image column of every staged column: once
    static_for<kFmNPT>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        const int c = tid + q * nthr;
        cidx[q] = (c < cols) ? fm_reflect101(x0 + c - ax, w) : 0;
    });
    auto load_row = [&](int r, float (&dst)[kFmNPT]) {             // global -> registers
        const float* src = P.img + (size_t)fm_reflect101(y0 + r - ay, h) * w;
        static_for<kFmNPT>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            dst[q] = src[cidx[q]];
        });
    };
    auto store_row = [&](int r, const float (&srcv)[kFmNPT]) {     // registers -> hi / lo halves in ring slot r mod (TH + 1)
        mm_half* sh = ring + (size_t)(r % (kFmTH + 1)) * 2 * cols;
        mm_half* sl = sh + cols;
        static_for<kFmNPT>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const int c = tid + q * nthr;
            if (c < cols) {
                mm_half hi, lo;
                mm_split(srcv[q] * idown, hi, lo);
                sh[c] = hi;
                sl[c] = lo;
            }
        });
    };
    mm_h8 fh[kFmPF][kFmKS], fl[kFmPF][kFmKS];
    auto load_frags = [&](int j, mm_h8 (&dh)[kFmKS], mm_h8 (&dl)[kFmKS]) {   // kernel row j's Toeplitz fragments (L2)
        static_for<kFmKS>([&](auto kq) {
            constexpr int kk = decltype(kq)::value;
            dh[kk] = P.frag[((size_t)(j * kFmKS + kk) * 2 + 0) * 64 + lane];
            dl[kk] = P.frag[((size_t)(j * kFmKS + kk) * 2 + 1) * 64 + lane];
        });
    };
    for (int r = 0; r < kFmTH; ++r) {
        load_row(r, raw[0]);
        store_row(r, raw[0]);
    }
    // A fragment set and an image row are requested THREE steps before they are used: a step is ~400 cycles of matrix
    // instructions, an L2 / Infinity-Cache hit 1500-5000 (one step ahead the kernel ran at the latency: 3.4 us per step)
    static_for<kFmPF - 1>([&](auto ss) {
        constexpr int s_ = decltype(ss)::value;
        if (s_ < kh) load_frags(s_, fh[s_], fl[s_]);
        if (s_ + 1 < kh) load_row(s_ + kFmTH, raw[s_]);
    });
    __syncthreads();
    mm_f4 ch[kFmTH], cl[kFmTH];
    static_for<kFmTH>([&](auto tt) { ch[decltype(tt)::value] = mm_zero(); cl[decltype(tt)::value] = mm_zero(); });
    for (int j0 = 0; j0 < kh; j0 += kFmPF) {
        static_for<kFmPF>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value, s3 = (s_ + kFmPF - 1) % kFmPF;
            const int j = j0 + s_;
            if (j < kh) {
                if (j + 1 < kh) store_row(j + kFmTH, raw[s_]);     // requested at step j - 3; into the slot step j - 1 freed
                if (j + kFmPF - 1 < kh) load_frags(j + kFmPF - 1, fh[s3], fl[s3]);
                if (j + kFmPF < kh) load_row(j + kFmPF - 1 + kFmTH, raw[s3]);
                static_for<kFmTH>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    const mm_half* sh = ring + (size_t)((j + t) % (kFmTH + 1)) * 2 * cols;
                    const mm_half* sl = sh + cols;
                    const int base = 256 * wv + 16 * n16 + 8 * g;   // this lane's window piece (multiple of 8 halves)
                    mm_h8 xh[kFmKS], xl[kFmKS];                    // the row's eight window pieces first, then its twelve products
                    static_for<kFmKS>([&](auto kq) {
                        constexpr int kk = decltype(kq)::value;
                        xh[kk] = *reinterpret_cast<const mm_h8*>(sh + base + 32 * kk);
                        xl[kk] = *reinterpret_cast<const mm_h8*>(sl + base + 32 * kk);
                    });
                    mm_sched_fence();
                    static_for<kFmKS>([&](auto kq) {
                        constexpr int kk = decltype(kq)::value;
                        ch[t] = mm_mfma(fh[s_][kk], xh[kk], ch[t]);
                        cl[t] = mm_mfma(fh[s_][kk], xl[kk], cl[t]);
                        cl[t] = mm_mfma(fl[s_][kk], xh[kk], cl[t]);
                    });
                    mm_sched_fence();
                });
                lds_barrier();                                     // orders the LDS traffic only: the requests stay in flight
            }
        });
    }
    const float osc = P.scales[0] * iup;
    static_for<kFmTH>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        const int y = y0 + t, x = x0 + 256 * wv + 16 * n16 + 4 * g;
        if (y < h) {
            float* o = P.out + (size_t)y * w + x;
            static_for<4>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                if (x + r < w) {
                    const float v = fmaf(mm_get(cl[t], r), kMmLoInv, mm_get(ch[t], r)) * osc;
                    o[r] = P.accumulate ? o[r] + v : v;
                }
            });
        }
    });
}

}  // namespace d4w

using namespace d4w;

extern "C" {

// 1 when d4w_filter2d_f32 runs this kernel size on the matrix cores (D4W_F2D_MM=0 switches it off)
int d4w_filter2d_mm_eligible(int kh, int kw) {
    static const int on = [] { const char* v = getenv("D4W_F2D_MM"); return v ? atoi(v) : 1; }();
    return on && kh >= 1 && kh <= 1024 && kw >= 1 && kw + 15 <= 32 * kFmKS;
}

size_t d4w_filter2d_mm_ws_bytes(int kh, int kw) {
    (void)kw;
    return (kh > 0) ? 64 + (size_t)kh * kFmKS * 2 * 64 * sizeof(mm_h8) : 0;
}

// ws: DEVICE scratch of d4w_filter2d_mm_ws_bytes(kh, kw) bytes (16-byte aligned)
int d4w_filter2d_mm_f32(const float* img, int h, int w, const float* kernel, int kh, int kw, float* out, int accumulate,
                        void* ws, void* stream) {
    if (!img || !kernel || !out || !ws || h < 1 || w < 1 || kh < 1 || kw < 1) return fail(D4W_EINVAL, "bad argument");
    if (img == out) return fail(D4W_EINVAL, "filter2d cannot run in place");
    if (!d4w_filter2d_mm_eligible(kh, kw)) return fail(D4W_EINVAL, "kernel %d x %d has no matrix-core form (kw <= %d)", kh, kw, 32 * kFmKS - 15);
    if ((h + kFmTH - 1) / kFmTH > 65535) return fail(D4W_EINVAL, "image height %d exceeds the grid limit", h);
    float* scales = (float*)ws;
    mm_h8* frag = reinterpret_cast<mm_h8*>((char*)ws + 64);
    int rc = d4w_minmax_f32(img, (size_t)h * w, scales + 1, stream);
    if (rc) return rc;
    D4W_LAUNCH(fm_build_frags, dim3(std::min(ceil_div(kh * kFmKS * 64, 256), 256)), dim3(256), 0, stream, kernel, kh, kw, frag, scales);
    FmArgs P;
    P.img = img; P.out = out; P.frag = frag; P.scales = scales; P.h = h; P.w = w; P.kh = kh; P.kw = kw; P.accumulate = accumulate;
    P.ntile = std::min(kFmMaxW, ceil_div(w, 256));
    const int cols = 256 * P.ntile + kFmHalo;
    const size_t lds = (size_t)(kFmTH + 1) * 2 * cols * sizeof(mm_half);
    const dim3 grid(ceil_div(w, 256 * P.ntile), ceil_div(h, kFmTH));
    D4W_LAUNCH(filter2d_mm_rows, grid, dim3(64 * P.ntile), lds, stream, P);
    return D4W_OK;
}

}  // extern "C"
