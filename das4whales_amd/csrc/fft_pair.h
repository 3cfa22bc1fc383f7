// Register butterflies on PAIRS of complex numbers in structure-of-arrays form: one value of each of
// two independent transforms (two rows) rides the two halves of a 64-bit VGPR pair, so every real
// addition / multiplication of the butterfly is ONE v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 for
// both transforms -- half the VALU issue slots of the scalar forms and, unlike packing (re, im) of
// one number, no swizzles: re and im stay in separate registers, multiplication by +-i is a
// renaming plus one negation, twiddles are per-lane scalars broadcast to both halves.
#pragma once
#include "fft_radix.h"

namespace d4w {

struct c2 {
    v2f re, im;      // re = (re of transform A, re of transform B), im likewise
};

__device__ __forceinline__ c2 c2_make(float2 a, float2 b) { return c2{v2_make(a.x, b.x), v2_make(a.y, b.y)}; }
__device__ __forceinline__ float2 c2_a(c2 v) { return make_float2(v2_x(v.re), v2_x(v.im)); }
__device__ __forceinline__ float2 c2_b(c2 v) { return make_float2(v2_y(v.re), v2_y(v.im)); }
__device__ __forceinline__ c2 c2_add(c2 a, c2 b) { return c2{v2_add(a.re, b.re), v2_add(a.im, b.im)}; }
__device__ __forceinline__ c2 c2_sub(c2 a, c2 b) { return c2{v2_sub(a.re, b.re), v2_sub(a.im, b.im)}; }
__device__ __forceinline__ c2 c2_conj(c2 a) { return c2{a.re, v2_neg(a.im)}; }
__device__ __forceinline__ c2 c2_scale(c2 a, float s) { return c2{v2_muls(a.re, s), v2_muls(a.im, s)}; }
__device__ __forceinline__ c2 c2_scale2(c2 a, v2f s) { return c2{v2_mul(a.re, s), v2_mul(a.im, s)}; }
// a + (-i) b and a + (+i) b without materialising the rotation
__device__ __forceinline__ c2 c2_add_mi(c2 a, c2 b) { return c2{v2_add(a.re, b.im), v2_sub(a.im, b.re)}; }
__device__ __forceinline__ c2 c2_add_pi(c2 a, c2 b) { return c2{v2_sub(a.re, b.im), v2_add(a.im, b.re)}; }
__device__ __forceinline__ c2 c2_mul_mi(c2 a) { return c2{a.im, v2_neg(a.re)}; }
__device__ __forceinline__ c2 c2_mul_pi(c2 a) { return c2{v2_neg(a.im), a.re}; }
// a * w and a * conj(w), w one complex scalar for both halves
__device__ __forceinline__ c2 c2_mulw(c2 a, float2 w) {
    return c2{v2_fnma(a.im, w.y, v2_muls(a.re, w.x)), v2_fma(a.im, w.x, v2_muls(a.re, w.y))};
}
__device__ __forceinline__ c2 c2_mulwc(c2 a, float2 w) {
    return c2{v2_fma(a.im, w.y, v2_muls(a.re, w.x)), v2_fnma(a.re, w.y, v2_muls(a.im, w.x))};
}

// x * exp(-2 pi i m/n) with literal constants
template <int M_, int N_>
__device__ __forceinline__ c2 c2_rot(c2 x) {
    constexpr int m = ((M_ % N_) + N_) % N_;
    if constexpr (m == 0) {
        return x;
    } else if constexpr (4 * m == N_) {
        return c2_mul_mi(x);
    } else if constexpr (2 * m == N_) {
        return c2{v2_neg(x.re), v2_neg(x.im)};
    } else if constexpr (4 * m == 3 * N_) {
        return c2_mul_pi(x);
    } else {
        constexpr float c = (float)ct_cos2pi(m, N_);
        constexpr float s = (float)(-ct_sin2pi(m, N_));
        return c2_mulw(x, make_float2(c, s));
    }
}

template <int R>
__device__ __forceinline__ void dftp(c2 (&x)[R]);

template <>
__device__ __forceinline__ void dftp<2>(c2 (&x)[2]) {
    const c2 a = x[0], b = x[1];
    x[0] = c2_add(a, b);
    x[1] = c2_sub(a, b);
}

template <>
__device__ __forceinline__ void dftp<4>(c2 (&x)[4]) {
    const c2 a = c2_add(x[0], x[2]), b = c2_sub(x[0], x[2]);
    const c2 c = c2_add(x[1], x[3]), d = c2_sub(x[1], x[3]);
    x[0] = c2_add(a, c);
    x[2] = c2_sub(a, c);
    x[1] = c2_add_mi(b, d);
    x[3] = c2_add_pi(b, d);
}

// a * s + c with a real literal s (both halves)
__device__ __forceinline__ c2 c2_fmas(c2 a, float s, c2 c) { return c2{v2_fma(a.re, s, c.re), v2_fma(a.im, s, c.im)}; }

template <>
__device__ __forceinline__ void dftp<1>(c2 (&)[1]) {}

template <>
__device__ __forceinline__ void dftp<3>(c2 (&x)[3]) {
    constexpr float s60 = (float)ct_sin2pi(1, 3);
    const c2 t1 = c2_add(x[1], x[2]);
    const c2 t2 = c2_fmas(t1, -0.5f, x[0]);
    const c2 t3 = c2_scale(c2_sub(x[1], x[2]), s60);
    x[0] = c2_add(x[0], t1);
    x[1] = c2_add_mi(t2, t3);
    x[2] = c2_add_pi(t2, t3);
}

template <>
__device__ __forceinline__ void dftp<5>(c2 (&x)[5]) {
    constexpr float c1 = (float)ct_cos2pi(1, 5), cc2 = (float)ct_cos2pi(2, 5);
    constexpr float s1 = (float)ct_sin2pi(1, 5), s2 = (float)ct_sin2pi(2, 5);
    const c2 t1 = c2_add(x[1], x[4]), t2 = c2_add(x[2], x[3]);
    const c2 t3 = c2_sub(x[1], x[4]), t4 = c2_sub(x[2], x[3]);
    const c2 a1 = c2_fmas(t1, c1, c2_fmas(t2, cc2, x[0]));
    const c2 a2 = c2_fmas(t1, cc2, c2_fmas(t2, c1, x[0]));
    const c2 b1 = c2_fmas(t3, s1, c2_scale(t4, s2));
    const c2 b2 = c2_fmas(t3, s2, c2_scale(t4, -s1));
    x[0] = c2_add(x[0], c2_add(t1, t2));
    x[1] = c2_add_mi(a1, b1);
    x[4] = c2_add_pi(a1, b1);
    x[2] = c2_add_mi(a2, b2);
    x[3] = c2_add_pi(a2, b2);
}

// every other radix, mirroring dft<R> of fft_radix.h: composite via one Cooley-Tukey split in registers
// (n = R2 n1 + n2, k = k1 + R1 k2), odd primes by the paired cosine / sine sums
template <int R>
__device__ __forceinline__ void dftp(c2 (&x)[R]) {
    constexpr int R1 = split_r1(R);
    if constexpr (R1 == R) {
        static_assert(R % 2 == 1, "prime radix 2 has its own butterfly");
        constexpr int H = (R - 1) / 2;
        c2 sq[H], dq[H];
        static_for<H>([&](auto qq) {
            constexpr int q = decltype(qq)::value + 1;
            sq[q - 1] = c2_add(x[q], x[R - q]);
            dq[q - 1] = c2_sub(x[q], x[R - q]);
        });
        const c2 x0 = x[0];
        c2 sum = x0;
        static_for<H>([&](auto qq) { sum = c2_add(sum, sq[decltype(qq)::value]); });
        x[0] = sum;
        static_for<H>([&](auto kk) {
            constexpr int k = decltype(kk)::value + 1;
            c2 A = x0, B = c2{v2_make(0.f, 0.f), v2_make(0.f, 0.f)};
            static_for<H>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 1;
                constexpr float c = (float)ct_cos2pi((q * k) % R, R);
                constexpr float sn = (float)ct_sin2pi((q * k) % R, R);
                A = c2_fmas(sq[q - 1], c, A);
                B = c2_fmas(dq[q - 1], sn, B);
            });
            x[k] = c2_add_mi(A, B);
            x[R - k] = c2_add_pi(A, B);
        });
    } else {
        constexpr int R2 = R / R1;
        c2 u[R2][R1];
        static_for<R2>([&](auto nn2) {
            constexpr int n2 = decltype(nn2)::value;
            c2 col[R1];
            static_for<R1>([&](auto nn1) { col[decltype(nn1)::value] = x[R2 * decltype(nn1)::value + n2]; });
            dftp<R1>(col);
            static_for<R1>([&](auto kk1) {
                constexpr int k1 = decltype(kk1)::value;
                u[n2][k1] = c2_rot<n2 * k1, R>(col[k1]);
            });
        });
        static_for<R1>([&](auto kk1) {
            constexpr int k1 = decltype(kk1)::value;
            c2 row[R2];
            static_for<R2>([&](auto nn2) { row[decltype(nn2)::value] = u[decltype(nn2)::value][k1]; });
            dftp<R2>(row);
            static_for<R2>([&](auto kk2) { x[k1 + R1 * decltype(kk2)::value] = row[decltype(kk2)::value]; });
        });
    }
}

// unnormalised inverse by the swap trick: exchanging re and im is a renaming here
template <int R>
__device__ __forceinline__ void idftp(c2 (&x)[R]) {
    static_for<R>([&](auto i) {
        constexpr int k = decltype(i)::value;
        x[k] = c2{x[k].im, x[k].re};
    });
    dftp<R>(x);
    static_for<R>([&](auto i) {
        constexpr int k = decltype(i)::value;
        x[k] = c2{x[k].im, x[k].re};
    });
}

}  // namespace d4w
