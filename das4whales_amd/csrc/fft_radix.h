// Register-resident radix-R DFT butterflies (forward sign exp(-2 pi i/R)); the inverse is obtained
// by the swap trick IDFT(x) = swap(DFT(swap(x))).  All twiddle constants are evaluated at compile
// time in double precision, so the unrolled butterflies contain only literal operands.
#pragma once
#include <type_traits>
#include <utility>

#include "d4w_internal.h"

namespace d4w {

// ---------------------------------------------------------------------------------------------
// compile-time cos/sin(2 pi m / n), exact octant reduction on the rational m/n
// ---------------------------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double ct_sin_small(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int k = 1; k <= 12; ++k) {
        term *= -x2 / ((2.0 * k) * (2.0 * k + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double ct_cos_small(double x) {  // |x| <= pi/4
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k <= 12; ++k) {
        term *= -x2 / ((2.0 * k - 1.0) * (2.0 * k));
        sum += term;
    }
    return sum;
}
// cos(2 pi num/den), 0 <= num/den <= 1/4
constexpr double ct_cos_q(long long num, long long den) {
    if (8 * num <= den) return ct_cos_small(2.0 * kPi * (double)num / (double)den);
    // cos(2 pi t) = sin(2 pi (1/4 - t))
    return ct_sin_small(2.0 * kPi * (double)(den - 4 * num) / (double)(4 * den));
}
constexpr double ct_cos2pi(long long m, long long n) {
    m %= n;
    if (m < 0) m += n;
    if (2 * m > n) m = n - m;                                 // cos even about t = 1/2
    if (4 * m > n) return -ct_cos_q(n - 2 * m, 2 * n);        // cos(2 pi t) = -cos(2 pi (1/2 - t))
    return ct_cos_q(m, n);
}
constexpr double ct_sin2pi(long long m, long long n) {        // sin(2 pi m/n) = cos(2 pi (m/n - 1/4))
    return ct_cos2pi(4 * m - n, 4 * n);
}

// ---------------------------------------------------------------------------------------------
// compile-time loop
// ---------------------------------------------------------------------------------------------
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// x * exp(-2 pi i m/n) with literal constants (trivial rotations folded)
template <int M_, int N_>
__device__ __forceinline__ float2 rot_const(float2 x) {
    constexpr int m = ((M_ % N_) + N_) % N_;
    if constexpr (m == 0) {
        return x;
    } else if constexpr (4 * m == N_) {
        return c_mul_mi(x);
    } else if constexpr (2 * m == N_) {
        return make_float2(-x.x, -x.y);
    } else if constexpr (4 * m == 3 * N_) {
        return c_mul_pi(x);
    } else {
        constexpr float c = (float)ct_cos2pi(m, N_);
        constexpr float s = (float)(-ct_sin2pi(m, N_));
        return make_float2(fmaf(x.x, c, -x.y * s), fmaf(x.x, s, x.y * c));
    }
}

// ---------------------------------------------------------------------------------------------
// radix kernels
// ---------------------------------------------------------------------------------------------
constexpr int smallest_factor(int r) {
    for (int f = 2; f * f <= r; ++f)
        if (r % f == 0) return f;
    return r;
}
// preferred split R = R1 * R2 for composite radices (R1 from the hand-written set)
constexpr int split_r1(int r) {
    if (r % 4 == 0 && r > 4) return 4;
    if (r % 5 == 0 && r > 5) return 5;
    if (r % 3 == 0 && r > 3) return 3;
    if (r % 2 == 0 && r > 2) return 2;
    return smallest_factor(r);
}

template <int R>
__device__ __forceinline__ void dft(float2 (&x)[R]);

template <>
__device__ __forceinline__ void dft<1>(float2 (&)[1]) {}

template <>
__device__ __forceinline__ void dft<2>(float2 (&x)[2]) {
    float2 a = x[0], b = x[1];
    x[0] = c_add(a, b);
    x[1] = c_sub(a, b);
}

template <>
__device__ __forceinline__ void dft<3>(float2 (&x)[3]) {
    constexpr float s60 = (float)ct_sin2pi(1, 3);
    float2 t1 = c_add(x[1], x[2]);
    float2 t2 = make_float2(fmaf(-0.5f, t1.x, x[0].x), fmaf(-0.5f, t1.y, x[0].y));
    float2 t3 = c_scale(c_sub(x[1], x[2]), s60);
    x[0] = c_add(x[0], t1);
    x[1] = c_add(t2, c_mul_mi(t3));
    x[2] = c_add(t2, c_mul_pi(t3));
}

template <>
__device__ __forceinline__ void dft<4>(float2 (&x)[4]) {
    float2 a = c_add(x[0], x[2]), b = c_sub(x[0], x[2]);
    float2 c = c_add(x[1], x[3]), d = c_sub(x[1], x[3]);
    x[0] = c_add(a, c);
    x[2] = c_sub(a, c);
    x[1] = c_add(b, c_mul_mi(d));
    x[3] = c_add(b, c_mul_pi(d));
}

template <>
__device__ __forceinline__ void dft<5>(float2 (&x)[5]) {
    constexpr float c1 = (float)ct_cos2pi(1, 5), c2 = (float)ct_cos2pi(2, 5);
    constexpr float s1 = (float)ct_sin2pi(1, 5), s2 = (float)ct_sin2pi(2, 5);
    float2 t1 = c_add(x[1], x[4]), t2 = c_add(x[2], x[3]);
    float2 t3 = c_sub(x[1], x[4]), t4 = c_sub(x[2], x[3]);
    float2 a1 = make_float2(fmaf(c1, t1.x, fmaf(c2, t2.x, x[0].x)), fmaf(c1, t1.y, fmaf(c2, t2.y, x[0].y)));
    float2 a2 = make_float2(fmaf(c2, t1.x, fmaf(c1, t2.x, x[0].x)), fmaf(c2, t1.y, fmaf(c1, t2.y, x[0].y)));
    float2 b1 = make_float2(fmaf(s1, t3.x, s2 * t4.x), fmaf(s1, t3.y, s2 * t4.y));
    float2 b2 = make_float2(fmaf(s2, t3.x, -s1 * t4.x), fmaf(s2, t3.y, -s1 * t4.y));
    x[0] = c_add(x[0], c_add(t1, t2));
    x[1] = c_add(a1, c_mul_mi(b1));
    x[4] = c_add(a1, c_mul_pi(b1));
    x[2] = c_add(a2, c_mul_mi(b2));
    x[3] = c_add(a2, c_mul_pi(b2));
}

// generic: composite via one Cooley-Tukey split in registers, primes via the naive O(R^2) DFT
template <int R>
__device__ __forceinline__ void dft(float2 (&x)[R]) {
    constexpr int R1 = split_r1(R);
    if constexpr (R1 == R) {
        // odd prime: the conjugate symmetry of the twiddles halves the work of the naive O(R^2) sum --
        //   s_q = x_q + x_(R-q), d_q = x_q - x_(R-q)  (q = 1 .. H = (R-1)/2)
        //   A_k = x_0 + sum_q cos(2 pi q k / R) s_q,   B_k = sum_q sin(2 pi q k / R) d_q
        //   X_k = A_k - i B_k,   X_(R-k) = A_k + i B_k
        // real coefficients only (2 FMAs per term instead of a complex multiply), outputs in pairs
        static_assert(R % 2 == 1, "prime radix 2 has its own butterfly");
        constexpr int H = (R - 1) / 2;
        float2 sq[H], dq[H];
        static_for<H>([&](auto qq) {
            constexpr int q = decltype(qq)::value + 1;
            sq[q - 1] = c_add(x[q], x[R - q]);
            dq[q - 1] = c_sub(x[q], x[R - q]);
        });
        const float2 x0 = x[0];
        float2 sum = x0;
        static_for<H>([&](auto qq) { sum = c_add(sum, sq[decltype(qq)::value]); });
        x[0] = sum;
        static_for<H>([&](auto kk) {
            constexpr int k = decltype(kk)::value + 1;
            float2 A = x0, B = make_float2(0.f, 0.f);
            static_for<H>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 1;
                constexpr float c = (float)ct_cos2pi((q * k) % R, R);
                constexpr float sn = (float)ct_sin2pi((q * k) % R, R);
                A = make_float2(fmaf(c, sq[q - 1].x, A.x), fmaf(c, sq[q - 1].y, A.y));
                B = make_float2(fmaf(sn, dq[q - 1].x, B.x), fmaf(sn, dq[q - 1].y, B.y));
            });
            x[k] = c_add(A, c_mul_mi(B));
            x[R - k] = c_add(A, c_mul_pi(B));
        });
    } else {
        // n = R2*n1 + n2 ; k = k1 + R1*k2
        constexpr int R2 = R / R1;
        float2 u[R2][R1];
        static_for<R2>([&](auto nn2) {
            constexpr int n2 = decltype(nn2)::value;
            float2 col[R1];
            static_for<R1>([&](auto nn1) { col[decltype(nn1)::value] = x[R2 * decltype(nn1)::value + n2]; });
            dft<R1>(col);
            static_for<R1>([&](auto kk1) {
                constexpr int k1 = decltype(kk1)::value;
                u[n2][k1] = rot_const<n2 * k1, R>(col[k1]);
            });
        });
        static_for<R1>([&](auto kk1) {
            constexpr int k1 = decltype(kk1)::value;
            float2 row[R2];
            static_for<R2>([&](auto nn2) { row[decltype(nn2)::value] = u[decltype(nn2)::value][k1]; });
            dft<R2>(row);
            static_for<R2>([&](auto kk2) { x[k1 + R1 * decltype(kk2)::value] = row[decltype(kk2)::value]; });
        });
    }
}

template <int R>
__device__ __forceinline__ void idft(float2 (&x)[R]) {  // unnormalised inverse
    static_for<R>([&](auto i) {
        constexpr int k = decltype(i)::value;
        x[k] = make_float2(x[k].y, x[k].x);
    });
    dft<R>(x);
    static_for<R>([&](auto i) {
        constexpr int k = decltype(i)::value;
        x[k] = make_float2(x[k].y, x[k].x);
    });
}

}  // namespace d4w
