// In-place mixed-radix FFT stages on an LDS-resident tile.
//
// Forward = decimation in frequency (natural order in, digit-reversed order out), inverse =
// decimation in time (digit-reversed in, natural out), so a forward/inverse pair never needs a
// reordering pass: every butterfly reads R elements and writes the same R LDS locations, one
// __syncthreads() per stage.  The digit-reversed position <-> frequency map is materialised on
// the host (FkPlan tables) for the one place that needs actual frequencies (the f-k pair op).
//
// Position p of frequency k for radices (R0, R1, ..., Rn-1), L = prod R:
//     k = d0 + R0*(d1 + R1*(d2 + ...)),   p = d0*L/R0 + d1*L/(R0 R1) + ... + d_{n-1}
// hence frequency L-1-k sits at position L-1-p (digit-wise complement).
#pragma once
#include "fft_radix.h"

namespace d4w {

constexpr int kMaxStages = 10;

struct AxisDesc {
    int L;                    // transform length
    int nstage;               // number of radix stages (0 when L == 1)
    int radix[kMaxStages];    // DIF order
    int nhi;                  // entries of the coarse twiddle table = ceil(L / 64)
    const float2* tw2;        // device table, two-level: [0,64) -> W_L^a ; [64, 64+nhi) -> W_L^(64 b)
};

// Two-level twiddle table staged in LDS: W_L^t = lo[t & 63] * hi[t >> 6]  (one extra rounding).
// Keeping twiddles in LDS means the FFT stages issue no vector-memory loads, so global prefetch
// loads of the next tile stay in flight across the whole compute phase (vmcnt is in-order).
constexpr int kTwLo = 64;
struct TwLds {
    const float2* lo;
    const float2* hi;
    __device__ __forceinline__ float2 get(int t) const { return c_mul(lo[t & (kTwLo - 1)], hi[t >> 6]); }
};
__device__ __forceinline__ int tw_lds_elems(const AxisDesc& ax) { return kTwLo + ax.nhi; }
// cooperative copy of an axis table into LDS (caller synchronises)
__device__ __forceinline__ TwLds tw_stage(const AxisDesc& ax, float2* dst, int tid, int nthr) {
    const int n = kTwLo + ax.nhi;
    for (int i = tid; i < n; i += nthr) dst[i] = ax.tw2[i];
    TwLds t;
    t.lo = dst;
    t.hi = dst + kTwLo;
    return t;
}

// a / d for 0 <= a < 2^24 via a float reciprocal and a +-1 fix-up (~6 VALU ops instead of the
// ~35 of a 32-bit integer division)
struct FDiv {
    int d;
    float rd;
    __device__ __forceinline__ explicit FDiv(int d_) : d(d_), rd(d4w_rcp((float)d_)) {}
    __device__ __forceinline__ int div(int a) const {
        int q = (int)((float)a * rd);
        const int r = a - d4w_mul24(q, d);
        q += (r >= d) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    }
};

// One radix-R stage over `nb0*nb1` independent transforms living in the same LDS tile.
//   element (transform b = (b0, b1), index n) is at  buf[b0*bs0 + b1*bs1 + n*es]
//   BATCH_FAST: consecutive threads walk the batch index (strided-axis tiles: conflict-free,
//   twiddle loads are wave-uniform); otherwise consecutive threads walk the butterfly index
//   (contiguous rows).
// Twiddles: ONE table load w = W_Ls^j per butterfly; the powers w^2..w^(R-1) are formed by a
// log-depth product tree in registers (<= 4 roundings deep for R = 10), instead of R-1 gathers.
template <int R, bool INV, bool BATCH_FAST>
__device__ __forceinline__ void lds_stage(float2* buf, int L, int Ls, int es, int nb0, int bs0,
                                          int nb1, int bs1, const TwLds tw,
                                          int tid, int nthr) {
    const int m = Ls / R;          // butterflies per sub-transform group
    const int nbf = L / R;         // butterflies per transform
    const int nb = nb0 * nb1;
    const int total = nbf * nb;
    const int twstep = L / Ls;
    const int qs = m * es;
    const FDiv d_outer(BATCH_FAST ? nb : nbf), d_m(m), d_nb0(nb0);
    for (int w = tid; w < total; w += nthr) {
        int bf, b;
        if (BATCH_FAST) {
            bf = d_outer.div(w);
            b = w - d4w_mul24(bf, nb);
        } else {
            b = d_outer.div(w);
            bf = w - d4w_mul24(b, nbf);
        }
        const int g = d_m.div(bf);
        const int j = bf - d4w_mul24(g, m);
        const int b1 = (nb1 == 1) ? 0 : d_nb0.div(b);
        const int b0 = b - d4w_mul24(b1, nb0);
        float2* p = buf + d4w_mul24(b0, bs0) + d4w_mul24(b1, bs1) + d4w_mul24(d4w_mul24(g, Ls) + j, es);
        float2 x[R];
        static_for<R>([&](auto qq) { constexpr int q = decltype(qq)::value; x[q] = p[q * qs]; });
        if (m > 1) {
            float2 pw[R];
            pw[1] = tw.get(d4w_mul24(j, twstep));
            static_for<R - 2>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 2;
                pw[q] = c_mul(pw[q / 2], pw[q - q / 2]);
            });
            if (!INV) {
                dft<R>(x);
                static_for<R - 1>([&](auto qq) {
                    constexpr int q = decltype(qq)::value + 1;
                    x[q] = c_mul(x[q], pw[q]);
                });
            } else {
                static_for<R - 1>([&](auto qq) {
                    constexpr int q = decltype(qq)::value + 1;
                    x[q] = c_mulc(x[q], pw[q]);
                });
                idft<R>(x);
            }
        } else {
            if (!INV) dft<R>(x); else idft<R>(x);
        }
        static_for<R>([&](auto qq) { constexpr int q = decltype(qq)::value; p[q * qs] = x[q]; });
    }
}

// Loop-based stage for larger prime radices (7 <= R <= 31 in the generic kernels): the naive
// O(R^2) DFT with LDS re-reads and a private result array.  Compact and register-light, slow;
// it exists so that the real OOI channel counts (5510 = 2*5*19*29, 11020) are supported.
template <bool INV, bool BATCH_FAST>
__device__ __attribute__((noinline)) void lds_stage_prime(int R, float2* buf, int L, int Ls, int es,
                                                          int nb0, int bs0, int nb1, int bs1,
                                                          const TwLds tw, int tid, int nthr) {
    const int m = Ls / R;
    const int nbf = L / R;
    const int nb = nb0 * nb1;
    const int total = nbf * nb;
    const int twstep = L / Ls;
    const int wr = L / R;          // W_R^a = tw[a * wr]
    const int qs = m * es;
    for (int w = tid; w < total; w += nthr) {
        int bf, b;
        if (BATCH_FAST) {
            bf = w / nb;
            b = w - bf * nb;
        } else {
            b = w / nbf;
            bf = w - b * nbf;
        }
        const int g = bf / m;
        const int j = bf - g * m;
        const int b1 = b / nb0;
        const int b0 = b - b1 * nb0;
        float2* p = buf + b0 * bs0 + b1 * bs1 + (g * Ls + j) * es;
        float2 y[32];
#pragma unroll 1
        for (int k = 0; k < R; ++k) {
            float2 acc = make_float2(0.f, 0.f);
            int a = 0;                                   // (q*k) mod R
#pragma unroll 1
            for (int q = 0; q < R; ++q) {
                float2 x = p[q * qs];
                if (INV) {
                    x = c_mulc(x, tw.get(j * q * twstep));
                    acc = c_add(acc, c_mulc(x, tw.get(a * wr)));
                } else {
                    acc = c_add(acc, c_mul(x, tw.get(a * wr)));
                }
                a += k;
                if (a >= R) a -= R;
            }
            if (!INV) acc = c_mul(acc, tw.get(j * k * twstep));
            y[k] = acc;
        }
#pragma unroll 1
        for (int k = 0; k < R; ++k) p[k * qs] = y[k];
    }
}

// The same stage for one prime R, unrolled in registers: every butterfly loads its R inputs once, pairs them into
// s_q = x_q + x_(R-q), d_q = x_q - x_(R-q), and forms the outputs k and R - k together from the real sums
//   A_k = x_0 + sum_q cos(2 pi q k / R) s_q,  B_k = sum_q sin(2 pi q k / R) d_q:   y_k = A_k -/+ i B_k,  y_(R-k) = A_k +/- i B_k
// -- (R - 1)^2 real FMAs instead of R^2 complex ones with an LDS twiddle lookup each (the OOI channel counts 5510 = 2 5 19 29
// and 11020 on blocks too small for compiled kernels run through here).  noinline: one copy per (R, direction, order).
template <int R, bool INV, bool BATCH_FAST>
__device__ __attribute__((noinline)) void lds_stage_prime_t(float2* buf, int L, int Ls, int es,
                                                            int nb0, int bs0, int nb1, int bs1,
                                                            const TwLds tw, int tid, int nthr) {
    constexpr int H = (R - 1) / 2;
    const int m = Ls / R;
    const int nbf = L / R;
    const int nb = nb0 * nb1;
    const int total = nbf * nb;
    const int twstep = L / Ls;
    const int wr = L / R;          // W_R^a = tw[a * wr]
    const int qs = m * es;
    float cs[H], sn[H];            // cos / sin (2 pi a / R), a = 1..H
    static_for<H>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        const float2 t = tw.get((a + 1) * wr);
        cs[a] = t.x;
        sn[a] = -t.y;
    });
    for (int w = tid; w < total; w += nthr) {
        int bf, b;
        if (BATCH_FAST) {
            bf = w / nb;
            b = w - bf * nb;
        } else {
            b = w / nbf;
            bf = w - b * nbf;
        }
        const int g = bf / m;
        const int j = bf - g * m;
        const int b1 = b / nb0;
        const int b0 = b - b1 * nb0;
        float2* p = buf + b0 * bs0 + b1 * bs1 + (g * Ls + j) * es;
        float2 x[R];
        static_for<R>([&](auto qq) { constexpr int q = decltype(qq)::value; x[q] = p[q * qs]; });
        if (INV && m > 1)
            static_for<R - 1>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 1;
                x[q] = c_mulc(x[q], tw.get(j * q * twstep));
            });
        float2 y0 = x[0];
        static_for<H>([&](auto qq) {
            constexpr int q = decltype(qq)::value + 1;
            const float2 a = x[q], c = x[R - q];
            x[q] = c_add(a, c);
            x[R - q] = c_sub(a, c);
            y0 = c_add(y0, x[q]);
        });
        p[0] = y0;
        static_for<H>([&](auto kk) {
            constexpr int k = decltype(kk)::value + 1;
            float2 A = x[0], B = make_float2(0.f, 0.f);
            static_for<H>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 1;
                constexpr int a = (q * k) % R;
                constexpr int ai = (a <= H) ? a : R - a;
                const float c = cs[ai - 1], sg = (a <= H) ? sn[ai - 1] : -sn[ai - 1];
                A.x = fmaf(c, x[q].x, A.x);
                A.y = fmaf(c, x[q].y, A.y);
                B.x = fmaf(sg, x[R - q].x, B.x);
                B.y = fmaf(sg, x[R - q].y, B.y);
            });
            float2 yk = make_float2(A.x + B.y, A.y - B.x);      // A - i B
            float2 yr = make_float2(A.x - B.y, A.y + B.x);      // A + i B
            if (INV) {
                const float2 t = yk;
                yk = yr;
                yr = t;
            } else if (m > 1) {
                yk = c_mul(yk, tw.get(j * k * twstep));
                yr = c_mul(yr, tw.get(j * (R - k) * twstep));
            }
            p[k * qs] = yk;
            p[(R - k) * qs] = yr;
        });
    }
}

#define D4W_FOR_EACH_PRIME_RADIX(X) X(7) X(11) X(13) X(17) X(19) X(23) X(29) X(31)

// Radices the planner may emit.  FAST kernels carry only the fully unrolled small set (what the
// benchmark shapes 4000/20000 x 12000/120000 need); GENERIC kernels add the loop-based primes.
#define D4W_FOR_EACH_FAST_RADIX(X) X(2) X(3) X(4) X(5) X(6) X(8) X(10)

template <bool INV, bool BATCH_FAST, bool GENERIC>
__device__ __forceinline__ void lds_stage_dispatch(int R, float2* buf, int L, int Ls, int es,
                                                   int nb0, int bs0, int nb1, int bs1,
                                                   const TwLds tw, int tid, int nthr) {
    switch (R) {
#define D4W_CASE(RR) \
    case RR: lds_stage<RR, INV, BATCH_FAST>(buf, L, Ls, es, nb0, bs0, nb1, bs1, tw, tid, nthr); break;
        D4W_FOR_EACH_FAST_RADIX(D4W_CASE)
#undef D4W_CASE
#define D4W_CASE(RR) \
    case RR: if (GENERIC) lds_stage_prime_t<RR, INV, BATCH_FAST>(buf, L, Ls, es, nb0, bs0, nb1, bs1, tw, tid, nthr); break;
        D4W_FOR_EACH_PRIME_RADIX(D4W_CASE)
#undef D4W_CASE
        default:
            if (GENERIC) lds_stage_prime<INV, BATCH_FAST>(R, buf, L, Ls, es, nb0, bs0, nb1, bs1, tw, tid, nthr);
            break;
    }
}

// Full transform along one axis of an LDS tile.  The caller must have synchronised the tile
// before the call; on return the tile is synchronised again.
template <bool INV, bool BATCH_FAST, bool GENERIC>
__device__ __forceinline__ void lds_fft(float2* buf, const AxisDesc& ax, const TwLds tw, int es, int nb0,
                                        int bs0, int nb1, int bs1, int tid, int nthr) {
    if (!INV) {
        int Ls = ax.L;
        for (int s = 0; s < ax.nstage; ++s) {
            lds_stage_dispatch<false, BATCH_FAST, GENERIC>(ax.radix[s], buf, ax.L, Ls, es, nb0, bs0, nb1, bs1, tw, tid, nthr);
            Ls /= ax.radix[s];
            lds_barrier();
        }
    } else {
        int Ls = 1;
        for (int s = ax.nstage - 1; s >= 0; --s) {
            Ls *= ax.radix[s];
            lds_stage_dispatch<true, BATCH_FAST, GENERIC>(ax.radix[s], buf, ax.L, Ls, es, nb0, bs0, nb1, bs1, tw, tid, nthr);
            lds_barrier();
        }
    }
}

}  // namespace d4w
