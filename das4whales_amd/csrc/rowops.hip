// Row-independent time-axis operators on MI355X (gfx950): the zero-phase IIR filter
// (dsp.bp_filt / scipy.signal.sosfiltfilt on dsp.butterworth_filter designs, reference
// dsp.py:789-827,859-880) and the matched filter (detect.compute_cross_correlogram /
// detect.shift_xcorr, reference detect.py:96-166).  See DESIGN.md "band-pass" and
// "matched filter" for the data layout and the roofline that bounds each kernel.
#include <algorithm>

#include "d4w_internal.h"

namespace d4w {

// =============================================================================================
// zero-phase SOS cascade
//
// One wave = 64 rows x one time segment.  Lanes own rows; the recursion walks along time, so
// the [64 rows][32 samples] chunk is staged through LDS: coalesced 128-byte row pieces on the
// global side, conflict-free ds_read_b128 / ds_write_b128 per lane on the LDS side (row pitch
// 36 floats: 9*l mod 16 is a bijection on every 16-lane b128 group).  Segments of one row run
// concurrently; a segment that does not start at the row edge is warmed up over W samples.
// =============================================================================================
constexpr int kSosRows = 64;     // rows per wave (one per lane)
constexpr int kSosChunk = 32;    // time samples staged per LDS round trip
constexpr int kSosPitch = kSosChunk + 4;
constexpr int kSosMaxSec = 10;

struct SosCoef {
    float b0, b1, b2, a1, a2, z1, z2, pad;
};
struct SosArgs {
    SosCoef s[kSosMaxSec];
};

// forward pass input: scipy odd extension of the row, virtual index i in [-padlen, ns + padlen)
__device__ __forceinline__ float sos_fetch_fwd(const float* __restrict__ row, int ns, int i) {
    if (i < 0) return 2.0f * row[0] - row[-i];
    if (i >= ns) return 2.0f * row[ns - 1] - row[2 * (ns - 1) - i];
    return row[i];
}
// backward pass input: forward output, continued over the right extension by the edge buffer
__device__ __forceinline__ float sos_fetch_bwd(const float* __restrict__ row, const float* __restrict__ edge,
                                               int ns, int i) {
    return (i >= ns) ? edge[i - ns] : row[i];
}

template <int NSEC, bool REV>
__global__ __launch_bounds__(kSosRows) void sos_pass(SosArgs A, const float* __restrict__ src,
                                                     const float* __restrict__ edge_in,
                                                     float* __restrict__ dst, float* __restrict__ edge_out,
                                                     int nx, int ns, int padlen, int S, int W) {
    __shared__ __attribute__((aligned(16))) float tile[kSosRows * kSosPitch];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * kSosRows;
    const int a = blockIdx.y * S;
    const int b = min(a + S, ns);
    int i_start, count;
    if (!REV) {
        const bool exact = (a - W <= 0);
        i_start = exact ? -padlen : a - W;
        const int i_end = (b == ns) ? ns + padlen : b;
        count = i_end - i_start;
    } else {
        const bool exact = (b + W >= ns);
        i_start = exact ? ns + padlen - 1 : b + W - 1;
        count = i_start - a + 1;
    }
    const int my_row = min(row0 + lane, nx - 1);
    const float* my_src = src + (size_t)my_row * ns;
    const float x0 = REV ? sos_fetch_bwd(my_src, edge_in + (size_t)my_row * padlen, ns, i_start)
                         : sos_fetch_fwd(my_src, ns, i_start);
    float s1[NSEC], s2[NSEC];
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        s1[s] = A.s[s].z1 * x0;
        s2[s] = A.s[s].z2 * x0;
    }
    for (int m0 = 0; m0 < count; m0 += kSosChunk) {
        // ---- stage in: lanes walk time within a row (coalesced), two rows per wave instruction
#pragma unroll 4
        for (int e = lane; e < kSosRows * kSosChunk; e += kSosRows) {
            const int rl = e / kSosChunk, ml = e % kSosChunk;
            const int m = m0 + ml;
            const int row = min(row0 + rl, nx - 1);
            float v = 0.f;
            if (m < count) {
                const int i = REV ? i_start - m : i_start + m;
                const float* r = src + (size_t)row * ns;
                v = REV ? sos_fetch_bwd(r, edge_in + (size_t)row * padlen, ns, i) : sos_fetch_fwd(r, ns, i);
            }
            tile[rl * kSosPitch + ml] = v;
        }
        __syncthreads();
        // ---- recursion: lane = row, 4 samples per LDS access
        float4* mine = reinterpret_cast<float4*>(tile + lane * kSosPitch);
#pragma unroll 2
        for (int g = 0; g < kSosChunk / 4; ++g) {
            float4 v4 = mine[g];
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = v[j];
#pragma unroll
                for (int s = 0; s < NSEC; ++s) {
                    const SosCoef& c = A.s[s];
                    const float y = fmaf(c.b0, x, s1[s]);
                    s1[s] = fmaf(c.b1, x, fmaf(-c.a1, y, s2[s]));
                    s2[s] = fmaf(c.b2, x, -c.a2 * y);
                    x = y;
                }
                v[j] = x;
            }
            mine[g] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        // ---- stage out (only this segment's own samples; the last forward segment also feeds
        //      the right-extension outputs to the edge buffer for the backward pass)
#pragma unroll 4
        for (int e = lane; e < kSosRows * kSosChunk; e += kSosRows) {
            const int rl = e / kSosChunk, ml = e % kSosChunk;
            const int m = m0 + ml;
            const int row = row0 + rl;
            if (m < count && row < nx) {
                const int i = REV ? i_start - m : i_start + m;
                const float v = tile[rl * kSosPitch + ml];
                if (i >= a && i < b) dst[(size_t)row * ns + i] = v;
                else if (!REV && i >= ns) edge_out[(size_t)row * padlen + (i - ns)] = v;
            }
        }
        __syncthreads();
    }
}

// =============================================================================================
// per-row mean and max|x|   (detect.py:157: (x - mean) / max|x|, max of the un-de-meaned row)
// one workgroup per row, wavefront shuffle reduction, then one LDS hop across the four waves
// =============================================================================================
constexpr int kStatThreads = 256;

__global__ __launch_bounds__(kStatThreads) void row_stats(const float* __restrict__ x, int ns,
                                                          float* __restrict__ mean, float* __restrict__ maxabs) {
    __shared__ float red_s[kStatThreads / 64], red_m[kStatThreads / 64];
    const float* row = x + (size_t)blockIdx.x * ns;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, mx = 0.f;
    int i = threadIdx.x;
    for (; i + 3 * kStatThreads < ns; i += 4 * kStatThreads) {
        const float v0 = row[i], v1 = row[i + kStatThreads], v2 = row[i + 2 * kStatThreads],
                    v3 = row[i + 3 * kStatThreads];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
    }
    for (; i < ns; i += kStatThreads) {
        const float v = row[i];
        s0 += v;
        mx = fmaxf(mx, fabsf(v));
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) {
        red_s[wave] = s;
        red_m[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tm = 0.f;
        for (int w = 0; w < kStatThreads / 64; ++w) {
            ts += red_s[w];
            tm = fmaxf(tm, red_m[w]);
        }
        mean[blockIdx.x] = ts / (float)ns;
        maxabs[blockIdx.x] = tm;
    }
}

// =============================================================================================
// matched filter: direct-form correlation, NT templates fused over one read of x
//   y_t[c][k] = g[c] * sum_n (x[c][n+k] - m[c]) * taps[t][n]
// A workgroup owns kXcTile consecutive lags of one row.  The de-meaned row piece sits in LDS;
// every thread keeps 4 consecutive lags in registers and slides an 8-sample window over the
// taps (one conflict-free ds_read_b128 per 16*NT FMAs).  Taps are wave-uniform -> scalar loads.
// VALU-bound: 2*(L_0 + L_1) flop per 4 + 4*NT bytes.
// =============================================================================================
constexpr int kXcThreads = 256;
constexpr int kXcTile = kXcThreads * 4;   // lags per workgroup
constexpr int kXcTapBlock = 256;          // taps per LDS staging round

template <int NT>
__global__ __launch_bounds__(kXcThreads) void xcorr_fir(const float* __restrict__ x, int ns,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ maxabs,
                                                        const float* __restrict__ taps, int ltaps,
                                                        float* __restrict__ y0, float* __restrict__ y1) {
    __shared__ __attribute__((aligned(16))) float xs[kXcTile + kXcTapBlock + 4];
    const int tid = threadIdx.x;
    const int rowi = blockIdx.y;
    const int k0 = blockIdx.x * kXcTile;
    const float* row = x + (size_t)rowi * ns;
    const float m = mean ? mean[rowi] : 0.f;
    float g = 1.f;
    if (maxabs) {
        const float a = maxabs[rowi];
        g = (a > 0.f) ? 1.0f / a : 0.f;
    }
    float acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;

    for (int n0 = 0; n0 < ltaps; n0 += kXcTapBlock) {
        const int nb = min(kXcTapBlock, ltaps - n0);          // taps in this round (multiple of 4)
        const int need = kXcTile + nb;                        // samples k0+n0 .. k0+n0+need-1
        for (int j = tid; j < need; j += kXcThreads) {
            const int i = k0 + n0 + j;
            xs[j] = (i < ns) ? row[i] - m : 0.f;              // beyond the row: zero padding
        }
        __syncthreads();
        const float4* win = reinterpret_cast<const float4*>(xs) + tid;
        float4 lo = win[0];
        for (int tb = 0; tb < nb / 4; ++tb) {
            const float4 hi = win[tb + 1];
            const float w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float* tp = taps + (size_t)t * ltaps + n0 + 4 * tb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = tp[q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(w[r + q], c, acc[t][r]);
                }
            }
            lo = hi;
        }
        __syncthreads();
    }
    const int k = k0 + 4 * tid;
    if (k < ns) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float* out = (t == 0 ? y0 : y1) + (size_t)rowi * ns + k;
            if (k + 3 < ns && (((size_t)rowi * ns + k) & 3) == 0) {
                *reinterpret_cast<float4*>(out) =
                    make_float4(acc[t][0] * g, acc[t][1] * g, acc[t][2] * g, acc[t][3] * g);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k + r < ns) out[r] = acc[t][r] * g;
            }
        }
    }
}

}  // namespace d4w

using namespace d4w;

template <bool REV>
static int sos_launch(int nsec, dim3 grid, void* stream, const SosArgs& A, const float* src,
                      const float* edge_in, float* dst, float* edge_out, int nx, int ns, int padlen,
                      int S, int W) {
    switch (nsec) {
#define D4W_SOS_CASE(N)                                                                               \
    case N:                                                                                           \
        hipLaunchKernelGGL((sos_pass<N, REV>), grid, dim3(kSosRows), 0, (hipStream_t)stream, A, src,  \
                           edge_in, dst, edge_out, nx, ns, padlen, S, W);                             \
        break;
        D4W_SOS_CASE(1) D4W_SOS_CASE(2) D4W_SOS_CASE(3) D4W_SOS_CASE(4) D4W_SOS_CASE(5)
        D4W_SOS_CASE(6) D4W_SOS_CASE(7) D4W_SOS_CASE(8) D4W_SOS_CASE(9) D4W_SOS_CASE(10)
#undef D4W_SOS_CASE
        default:
            return fail(D4W_EINVAL, "nsec = %d not in 1..%d", nsec, kSosMaxSec);
    }
    D4W_HIP(hipGetLastError());
    return D4W_OK;
}

extern "C" {

size_t d4w_sosfiltfilt_ws_bytes(int nx, int ns, int padlen) {
    if (nx < 1 || ns < 1 || padlen < 0) return 0;
    return ((size_t)nx * ns + (size_t)nx * std::max(padlen, 1)) * sizeof(float);
}

int d4w_sosfiltfilt_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi,
                        int nsec, int padlen, int seg_len, int warm, void* ws, void* stream) {
    if (!x || !y || !sos || !zi || !ws) return fail(D4W_EINVAL, "NULL argument");
    if (nx < 1 || ns < 1 || padlen < 0) return fail(D4W_EINVAL, "bad shape %d x %d (padlen %d)", nx, ns, padlen);
    if (nsec < 1 || nsec > kSosMaxSec) return fail(D4W_EINVAL, "nsec = %d not in 1..%d", nsec, kSosMaxSec);
    if (ns <= padlen)
        return fail(D4W_EINVAL, "The length of the input vector x must be greater than padlen, which is %d.", padlen);
    SosArgs A;
    memset(&A, 0, sizeof(A));
    for (int s = 0; s < nsec; ++s) {
        const double* c = sos + 6 * s;
        if (c[3] == 0.0) return fail(D4W_EINVAL, "section %d has a0 = 0", s);
        const double a0 = c[3];
        A.s[s] = SosCoef{(float)(c[0] / a0), (float)(c[1] / a0), (float)(c[2] / a0), (float)(c[4] / a0),
                         (float)(c[5] / a0), (float)zi[2 * s], (float)zi[2 * s + 1], 0.f};
    }
    int S = seg_len, W = warm;
    if (S <= 0 || W <= 0 || S >= ns) { S = ns; W = ns; }           // one exact segment per row
    S = ((S + kSosChunk - 1) / kSosChunk) * kSosChunk;
    const int nseg = ceil_div(ns, S);
    float* t = (float*)ws;
    float* edge = t + (size_t)nx * ns;
    const dim3 grid(ceil_div(nx, kSosRows), nseg);
    int rc = sos_launch<false>(nsec, grid, stream, A, x, nullptr, t, edge, nx, ns, padlen, S, W);
    if (rc) return rc;
    return sos_launch<true>(nsec, grid, stream, A, t, edge, y, nullptr, nx, ns, padlen, S, W);
}

int d4w_row_stats_f32(const float* x, int nx, int ns, float* mean, float* maxabs, void* stream) {
    if (!x || !mean || !maxabs || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_stats, dim3(nx), dim3(kStatThreads), 0, stream, x, ns, mean, maxabs);
    return D4W_OK;
}

int d4w_xcorr_f32(const float* x, int nx, int ns, const float* mean, const float* maxabs, const float* taps,
                  int ntpl, int ltaps, float* y0, float* y1, void* stream) {
    if (!x || !taps || !y0 || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (ntpl < 1 || ntpl > 2 || (ntpl == 2 && !y1)) return fail(D4W_EINVAL, "ntpl = %d (1 or 2 templates per call)", ntpl);
    if (ltaps < 4 || (ltaps & 3)) return fail(D4W_EINVAL, "ltaps = %d must be a positive multiple of 4", ltaps);
    const dim3 grid(ceil_div(ns, kXcTile), nx);
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    if (ntpl == 1)
        D4W_LAUNCH(xcorr_fir<1>, grid, dim3(kXcThreads), 0, stream, x, ns, mean, maxabs, taps, ltaps, y0, y1);
    else
        D4W_LAUNCH(xcorr_fir<2>, grid, dim3(kXcThreads), 0, stream, x, ns, mean, maxabs, taps, ltaps, y0, y1);
    return D4W_OK;
}

}  // extern "C"
