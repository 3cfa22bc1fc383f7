// Host-side FFT planning shared by the translation units: radix factorisation, the DIF
// digit-order position <-> frequency maps and the two-level twiddle tables of fft_lds.h.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <vector>

#include "fft_lds.h"

namespace d4w {

static const int kSupportedRadix[] = {2, 3, 4, 5, 6, 7, 8, 10, 11, 13, 17, 19, 23, 29, 31};
static const int kFastRadix[] = {2, 3, 4, 5, 6, 8, 10};   // keep in sync with D4W_FOR_EACH_FAST_RADIX

static inline bool factor_radices(int L, std::vector<int>& out) {
    out.clear();
    if (L <= 0) return false;
    int e2 = 0, e3 = 0, e5 = 0;
    while (L % 2 == 0) { L /= 2; ++e2; }
    while (L % 3 == 0) { L /= 3; ++e3; }
    while (L % 5 == 0) { L /= 5; ++e5; }
    std::vector<int> other;
    for (int p = 7; p <= 31 && L > 1; ++p)
        while (L % p == 0) { L /= p; other.push_back(p); }
    if (L != 1) return false;   // prime factor > 31: needs Bluestein (not implemented)
    // fewest stages with radices <= 10: pair 2s with 5s into 10s, then 8/4, then 6 = 2*3
    std::vector<int> even, odd;
    int n10 = std::min(e2, e5);
    // keep 2^3 groups for radix 8 when that saves a stage: e.g. 2^5 5^3 -> 4,10,10,10 (4 stages)
    e2 -= n10; e5 -= n10;
    if (e2 % 3 == 1 && e2 >= 4) { even.push_back(4); even.push_back(4); e2 -= 4; }
    while (e2 >= 3) { even.push_back(8); e2 -= 3; }
    if (e2 == 2) { even.push_back(4); e2 = 0; }
    if (e2 == 1) {
        if (e3 > 0) { even.push_back(6); --e3; }
        else even.push_back(2);
    }
    std::sort(even.begin(), even.end(), [](int x, int y) { return x > y; });
    out = even;                                   // power-of-two-ish radices first (large strides)
    for (int i = 0; i < n10; ++i) out.push_back(10);
    for (int i = 0; i < e5; ++i) out.push_back(5);
    for (int i = 0; i < e3; ++i) out.push_back(3);
    for (int p : other) out.push_back(p);
    if ((int)out.size() > kMaxStages) return false;
    for (int r : out) {
        bool ok = false;
        for (int sr : kSupportedRadix) ok |= (sr == r);
        if (!ok) return false;
    }
    return true;
}

// position -> frequency for the DIF digit order (see fft_lds.h)
static inline std::vector<int> pos_to_freq(int L, const std::vector<int>& rad) {
    std::vector<int> f(L, 0);
    for (int p = 0; p < L; ++p) {
        int rem = p, weight = L, k = 0, mult = 1;
        for (int r : rad) {
            weight /= r;
            const int dgt = rem / weight;
            rem -= dgt * weight;
            k += dgt * mult;
            mult *= r;
        }
        f[p] = k;
    }
    return f;
}

// two-level table (fft_lds.h TwLds): [0,64) -> W_L^a, [64, 64+nhi) -> W_L^(64 b)
static inline std::vector<float2> twiddle_table2(int L, int* nhi_out) {
    const int nhi = (L + kTwLo - 1) / kTwLo;
    std::vector<float2> t(kTwLo + nhi);
    for (int i = 0; i < kTwLo; ++i) {
        const double a = -2.0 * M_PI * (double)(i % L) / (double)L;
        t[i] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int b = 0; b < nhi; ++b) {
        const double a = -2.0 * M_PI * (double)((kTwLo * (long long)b) % L) / (double)L;
        t[kTwLo + b] = make_float2((float)cos(a), (float)sin(a));
    }
    *nhi_out = nhi;
    return t;
}

static inline float2 wexp(long long num, long long den) {   // exp(-2 pi i num/den)
    num %= den;
    const double a = -2.0 * M_PI * (double)num / (double)den;
    return make_float2((float)cos(a), (float)sin(a));
}


// true when every stage of the axis is one of the fully unrolled radices (FAST kernels)
static inline bool axis_needs_generic(const AxisDesc& ax) {
    for (int s = 0; s < ax.nstage; ++s) {
        bool fast = false;
        for (int r : kFastRadix) fast |= (r == ax.radix[s]);
        if (!fast) return true;
    }
    return false;
}

// in-place radix-2 FFT (forward sign) of a power-of-two length, double precision: plan-time tables only
static inline void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const int n = (int)re.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

// Length of a Bluestein convolution that must hold n values: of the form 2^a 3^b 5^c (every factor an unrolled radix) between
// n and the next power of two, the one with the least length x stages (the next power of two can be almost twice as long;
// a length made of many small factors pays in LDS round trips).  D4W_BLUESTEIN_POW2=1 keeps powers of two.
static inline int smooth_len_235(long n) {
    const char* p2 = getenv("D4W_BLUESTEIN_POW2");
    long pow2 = 1;
    while (pow2 < n) pow2 *= 2;
    if (p2 && atoi(p2) > 0) return (int)pow2;
    std::vector<int> rad;
    long best = pow2;
    double best_cost = factor_radices((int)pow2, rad) ? (double)pow2 * (double)std::max<size_t>(rad.size(), 1) : 1e300;
    for (long a = 1; a < pow2; a *= 2)
        for (long b = a; b < pow2; b *= 3)
            for (long c = b; c < pow2; c *= 5) {
                if (c < n || !factor_radices((int)c, rad)) continue;
                const double cost = (double)c * (double)std::max<size_t>(rad.size(), 1);
                if (cost < best_cost) { best_cost = cost; best = c; }
            }
    return (int)best;
}

// in-place forward DFT of ANY length (double precision, plan-time tables only): a power of two directly, anything else as
// a chirp convolution through power-of-two transforms
static inline void host_dft_any(std::vector<double>& re, std::vector<double>& im) {
    const int n = (int)re.size();
    if ((n & (n - 1)) == 0) { host_fft_pow2(re, im); return; }
    int P = 1;
    while (P < 2 * n - 1) P *= 2;
    std::vector<double> wr(n), wi(n), ar(P, 0.0), ai(P, 0.0), br(P, 0.0), bi(P, 0.0);
    for (int k = 0; k < n; ++k) {
        const double ph = M_PI * (double)(((long long)k * k) % (2LL * n)) / (double)n;
        wr[k] = cos(ph); wi[k] = -sin(ph);                      // w[k] = exp(-i pi k^2 / n)
        ar[k] = re[k] * wr[k] - im[k] * wi[k];
        ai[k] = re[k] * wi[k] + im[k] * wr[k];
        br[k] = wr[k]; bi[k] = -wi[k];                          // conj(w), wrapped
        if (k) { br[P - k] = wr[k]; bi[P - k] = -wi[k]; }
    }
    host_fft_pow2(ar, ai);
    host_fft_pow2(br, bi);
    for (int p = 0; p < P; ++p) {                               // conj(A B): the inverse transform as conj(FFT(conj(.))) / P
        const double cr = ar[p] * br[p] - ai[p] * bi[p], ci = ar[p] * bi[p] + ai[p] * br[p];
        ar[p] = cr; ai[p] = -ci;
    }
    host_fft_pow2(ar, ai);
    for (int k = 0; k < n; ++k) {
        const double cr = ar[k] / P, ci = -ai[k] / P;
        re[k] = cr * wr[k] - ci * wi[k];
        im[k] = cr * wi[k] + ci * wr[k];
    }
}

}  // namespace d4w
