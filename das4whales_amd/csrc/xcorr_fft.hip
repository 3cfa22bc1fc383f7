// Matched filter by overlap-save FFT correlation on MI355X (gfx950): replaces the per-row
// scipy.signal.correlate(x, template, 'full', 'fft') of detect.compute_cross_correlogram /
// detect.shift_xcorr (reference detect.py:96-166) for templates of short support (the fin-whale
// call templates have 136 / 156 non-zero samples).
//
// The direct form (rowops.hip: xcorr_fir) is VALU-bound at 2 (L0 + L1) = 584 flop per sample; this
// kernel spends ~110 flop per sample and streams: one read of x, one write per template.
//   * a row is cut into blocks of B = 4096 samples that advance by S = B - 160 lags; the circular
//     correlation of a block is exact for its first S lags (support <= 161);
//   * a block is read as MB = 2048 packed complex samples; ONE complex FFT gives the block's real
//     spectrum after the untangle, which is multiplied by conj(T_t(f)) for each template t and
//     re-tangled, and one inverse FFT per template returns 2 lags per complex output;
//   * MB = 16 x 16 x 8 "fat" register stages exactly as in fk_fast.h (pass B): the first radix-16 is
//     applied to the registers the global loads landed in, the last one feeds the global stores, the
//     middle item does [radix 8 | pair op x 2 templates | inverse radix 8] for a group of 8
//     positions and its Hermitian partner group;
//   * a workgroup transforms the same block of TWO adjacent rows (every middle item then has work
//     for all 256 threads, and the template spectra are read once per row pair).
// Spectra tables are built per call by xcf_spectra (a few hundred microseconds of a 4096 x 161 DFT).
#include <cstdlib>

// complex arithmetic as v_pk_* on 64-bit register pairs in this translation unit: the kernel is
// VALU-bound and add-dominated (measured 13.2 -> 11.1 ms at 20000 x 120000); the f-k pass kernels
// lose a few percent with it (register pairs / swizzle moves) and keep the scalar forms
#define D4W_PKMATH 1
#include "fft_radix.h"

namespace d4w {

constexpr int kXfB = 4096, kXfMB = 2048;           // block length (real), packed length (complex)
constexpr int kXfNA = 16, kXfNB = 16, kXfNC = 8;   // DIF radices, MB = NA * NB * NC
constexpr int kXfM1 = kXfNB * kXfNC;               // 128
constexpr int kXfNG = kXfNA * kXfNB;               // 256 groups of NC positions
constexpr int kXfPad = 160;                        // lags lost per block = max support - 1
constexpr int kXfStep = kXfB - kXfPad;             // 3936 lags per block
constexpr int kXfThreads = 256;
constexpr int kXfRowP = kXfMB + kXfNG;             // LDS row pitch: one pad element per group

__host__ __device__ constexpr int xf_ad(int e) { return e + e / kXfNC; }
// frequency held by position e after the three DIF stages (digits a', b', d)
__host__ __device__ constexpr int xf_freq(int e) {
    return (e / kXfM1) + kXfNA * ((e / kXfNC) % kXfNB) + kXfNA * kXfNB * (e % kXfNC);
}

struct XfTables {
    const float2* gp;     // [ntpl][MB] conj(T_t(f)) at position e (f = xf_freq(e))
    const float* gn;      // [ntpl]     conj(T_t(MB)) (real)
    const float2* tw1;    // [M1]       W_MB^j
    const float2* tw2;    // [NB][NC]   W_M1^(j2 b)
    const float2* wg;     // [NG]       W_B^(a' + NA b') : untangle twiddle of a group's first frequency
};

template <int R>
__device__ __forceinline__ void xf_pw_tree(float2 w1, float2 (&pw)[R]) {
    pw[0] = make_float2(1.f, 0.f);
    pw[1] = w1;
    static_for<R - 2>([&](auto qq) {
        constexpr int q = decltype(qq)::value + 2;
        pw[q] = c_mul(pw[q / 2], pw[q - q / 2]);
    });
}

// spectra + twiddle tables (one launch per call; 2 x 2048 x L MACs)
__global__ __launch_bounds__(256) void xcf_spectra(const float* __restrict__ taps, int ntpl, int ltaps,
                                                   int len0, int len1, float2* __restrict__ gp,
                                                   float* __restrict__ gn, float2* __restrict__ tw1,
                                                   float2* __restrict__ tw2, float2* __restrict__ wg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntpl * kXfMB) {
        const int t = i / kXfMB, e = i - t * kXfMB;
        const int f = xf_freq(e), L = t ? len1 : len0;
        const float* tp = taps + (size_t)t * ltaps;
        double re = 0.0, im = 0.0;
        for (int n = 0; n < L; ++n) {
            float s, c;
            sincospif(2.0f * (float)((f * n) & (kXfB - 1)) / (float)kXfB, &s, &c);
            re += (double)tp[n] * c;
            im += (double)tp[n] * s;                   // conj(T(f)) = sum t[n] exp(+2 pi i f n / B)
        }
        gp[i] = make_float2((float)re, (float)im);
    }
    if (i < ntpl) {
        const int L = i ? len1 : len0;
        double s = 0.0;
        for (int n = 0; n < L; ++n) s += (n & 1) ? -(double)taps[(size_t)i * ltaps + n] : (double)taps[(size_t)i * ltaps + n];
        gn[i] = (float)s;
    }
    if (i < kXfM1) {
        float s, c;
        sincospif(-2.0f * (float)i / (float)kXfMB, &s, &c);
        tw1[i] = make_float2(c, s);
        const int b = i / kXfNC, j2 = i % kXfNC;
        sincospif(-2.0f * (float)((j2 * b) % kXfM1) / (float)kXfM1, &s, &c);
        tw2[i] = make_float2(c, s);
    }
    if (i < kXfNG) {
        const int f0 = (i / kXfNB) + kXfNA * (i % kXfNB);
        float s, c;
        sincospif(-2.0f * (float)f0 / (float)kXfB, &s, &c);
        wg[i] = make_float2(c, s);
    }
}

// pair op for one frequency pair: A = Z[f], Bc = conj(Z[MB - f]), w = W_B^f, gf = conj(T(f)),
// gmc = conj(T(f + MB)) = conj(conj(T(MB - f)))^* ... passed already conjugated (see call sites)
__device__ __forceinline__ void xf_pair(float2 A, float2 Bs, float2 w, float2 gf, float2 gm, float2& na, float2& nb) {
    const float2 Bc = c_conj(Bs);
    const float2 E = c_scale(c_add(A, Bc), 0.5f);
    const float2 O = c_mul_mi(c_scale(c_sub(A, Bc), 0.5f));
    const float2 tO = c_mul(w, O);
    const float2 Yp = c_mul(c_add(E, tO), gf);            // X(f)      conj(T(f))
    const float2 Ym = c_mulc(c_sub(E, tO), gm);           // X(f + MB) conj(T(f + MB)),  conj(T(f+MB)) = conj(gm)
    const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
    const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
    na = c_add(S, D);
    nb = c_conj(c_sub(S, D));
}

template <int NT, int WAVES>
__global__ __launch_bounds__(kXfThreads, WAVES) void xcorr_fft_blocks(XfTables T, const float* __restrict__ x, int nx,
                                                                  int ns, const float* __restrict__ mean,
                                                                  const float* __restrict__ maxabs,
                                                                  float* __restrict__ y0, float* __restrict__ y1) {
    constexpr int NA = kXfNA, NB = kXfNB, NC = kXfNC, M1 = kXfM1, MB = kXfMB, ROWP = kXfRowP;
    D4W_DYN_LDS(smem_raw);
    float2* bufA = reinterpret_cast<float2*>(smem_raw);        // [2 rows][ROWP]: block spectra, then each template's correlation
    float2* tw1 = bufA + 2 * ROWP;                              // [M1]
    float2* tw2 = tw1 + M1;                                     // [NB][NC]
    const int tid = threadIdx.x;
    if (tid < M1) {
        tw1[tid] = T.tw1[tid];
        tw2[tid] = T.tw2[tid];
    }
    const int r = tid >> 7, rem = tid & 127;                    // row of the pair, item within the row
    const int row = 2 * blockIdx.y + r;
    const bool live = row < nx;
    const int k0 = blockIdx.x * kXfStep;                        // first lag / first sample of the block
    const float* xr = x + (size_t)(live ? row : 0) * ns;
    const float mu = (mean && live) ? mean[row] : 0.f;
    float gain = 1.f;
    if (maxabs && live) {
        const float a = maxabs[row];
        gain = (a > 0.f) ? 1.0f / a : 0.f;
    }
    // ---------------- S1: radix NA on the packed samples z[m] = x[k0 + 2m] + i x[k0 + 2m + 1], m = j1 + a M1
    float2 pf[NA];
    {
        const int j1 = rem;
        const bool vec = ((((size_t)row * ns + k0) & 1) == 0);  // 8-byte aligned pairs
        static_for<NA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            const int i = k0 + 2 * (j1 + a * M1);
            float2 v = make_float2(0.f, 0.f);
            if (live) {
                if (vec && i + 1 < ns) {
                    v = *reinterpret_cast<const float2*>(xr + i);
                    v.x -= mu;
                    v.y -= mu;
                } else {
                    if (i < ns) v.x = xr[i] - mu;
                    if (i + 1 < ns) v.y = xr[i + 1] - mu;
                }
            }
            pf[a] = v;
        });
    }
    __syncthreads();                                            // twiddle tables visible
    {
        const int j1 = rem;
        dft<NA>(pf);
        float2 pw[NA];
        xf_pw_tree<NA>(tw1[j1], pw);
        float2* rowp = bufA + r * ROWP;
        static_for<NA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            rowp[xf_ad(j1 + a * M1)] = (a == 0) ? pf[0] : c_mul(pf[a], pw[a]);
        });
    }
    lds_barrier();
    // ---------------- S2: radix NB in place, x W_M1^(j2 b')
    {
        const int g = rem >> 3, j2 = rem & 7;
        float2* rowp = bufA + r * ROWP;
        float2 v[NB];
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            v[b] = rowp[xf_ad(g * M1 + j2 + b * NC)];
        });
        dft<NB>(v);
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            rowp[xf_ad(g * M1 + j2 + b * NC)] = (b == 0) ? v[0] : c_mul(v[b], tw2[b * NC + j2]);
        });
    }
    lds_barrier();
    // ---------------- per template: MID (radix NC on a group and its Hermitian partner group, pair op,
    //                  inverse radix NC), S2' and S1'.  Item p < 127: a proper pair (Gi < PG); p = 127: the
    //                  two self-paired groups 0 (digit partner (NC - d) % NC, f = 0 pairs with the Nyquist
    //                  bin) and NB / 2.
    int Gi, PG;
    {
        const int p = rem;
        if (p < 112) { const int g = 1 + (p >> 4), b = p & 15; Gi = g * NB + b; PG = (NA - g) * NB + (NB - 1 - b); }
        else if (p < 120) { const int b = p - 112; Gi = (NA / 2) * NB + b; PG = (NA / 2) * NB + (NB - 1 - b); }
        else if (p < 127) { const int b = p - 119; Gi = b; PG = NB - b; }
        else { Gi = 0; PG = NB / 2; }
    }
    const bool selfitem = (rem == 127);
    const float sc = gain / (float)MB;
    // the item's two groups of the block spectrum stay in registers for every template, so the
    // correlation of each template can overwrite the row buffer in place (37 KiB of LDS per
    // workgroup = four workgroups per CU)
    float2 a[NC], b[NC];
    {
        const float2* rowp = bufA + r * ROWP;
        const float2* ga = rowp + xf_ad(Gi * NC);
        const float2* gb = rowp + xf_ad(PG * NC);
        static_for<NC>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            a[d] = ga[d];
            b[d] = gb[d];
        });
        dft<NC>(a);
        dft<NC>(b);
    }
    const float2 wa0 = T.wg[Gi], wb0 = T.wg[PG];       // W_B^f, f = f0(G) + 256 d: W_B^(256 d) are literals
    static_for<NT>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        {
            const float2* gpa = T.gp + (size_t)t * MB + Gi * NC;
            const float2* gpb = T.gp + (size_t)t * MB + PG * NC;
            float2 GA[NC], GB[NC];
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                GA[d] = gpa[d];
                GB[d] = gpb[d];
            });
            const float gny = T.gn[t];
            float2 ra[NC], rb[NC];
            if (!selfitem) {
                float2 nb[NC];
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pn = NC - 1 - d;
                    xf_pair(a[d], b[pn], rot_const<d, 16>(wa0), GA[d], GB[pn], ra[d], nb[d]);
                });
                static_for<NC>([&](auto dd) {
                    constexpr int e = decltype(dd)::value;
                    rb[e] = nb[NC - 1 - e];
                });
            } else {
                // group 0 (array a): partner digit (NC - d) % NC; d = 0 pairs X(0) with the Nyquist bin
                float2 na[NC], nb[NC];
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pz = (NC - d) % NC;
                    const float2 gm = (d == 0) ? make_float2(gny, 0.f) : GA[pz];
                    xf_pair(a[d], a[pz], rot_const<d, 16>(wa0), GA[d], gm, na[d], nb[d]);
                });
                static_for<NC>([&](auto dd) {
                    constexpr int e = decltype(dd)::value;
                    constexpr int pz = (NC - e) % NC;
                    ra[e] = (e < pz) ? na[e] : nb[pz];
                });
                // group NB / 2 (array b): partner digit NC - 1 - d, no self-paired position
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pn = NC - 1 - d;
                    xf_pair(b[d], b[pn], rot_const<d, 16>(wb0), GB[d], GB[pn], na[d], nb[d]);
                });
                static_for<NC>([&](auto dd) {
                    constexpr int e = decltype(dd)::value;
                    constexpr int pn = NC - 1 - e;
                    rb[e] = (e < pn) ? na[e] : nb[pn];
                });
            }
            idft<NC>(ra);
            idft<NC>(rb);
            float2* outp = bufA + r * ROWP;
            float2* oa = outp + xf_ad(Gi * NC);
            float2* ob = outp + xf_ad(PG * NC);
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                oa[d] = ra[d];
                ob[d] = rb[d];
            });
        }
        lds_barrier();
        // ---------------- S2': inverse radix NB
        {
            const int g = rem >> 3, j2 = rem & 7;
            float2* rowp = bufA + r * ROWP;
            float2 v[NB];
            static_for<NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                const float2 xv = rowp[xf_ad(g * M1 + j2 + b * NC)];
                v[b] = (b == 0) ? xv : c_mulc(xv, tw2[b * NC + j2]);
            });
            idft<NB>(v);
            static_for<NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                rowp[xf_ad(g * M1 + j2 + b * NC)] = v[b];
            });
        }
        lds_barrier();
        // ---------------- S1': inverse radix NA -> lags k0 + 2m, k0 + 2m + 1 (m = j1 + a M1), the first S of them
        {
            const int j1 = rem;
            float2 pw[NA];
            xf_pw_tree<NA>(tw1[j1], pw);
            const float2* rowp = bufA + r * ROWP;
            float2 v[NA];
            static_for<NA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                const float2 xv = rowp[xf_ad(j1 + a * M1)];
                v[a] = (a == 0) ? xv : c_mulc(xv, pw[a]);
            });
            idft<NA>(v);
            if (live) {
                float* yr = (t == 0 ? y0 : y1) + (size_t)row * ns;
                const bool vec = ((((size_t)row * ns + k0) & 1) == 0);
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const int m = j1 + a * M1;
                    const int k = k0 + 2 * m;
                    if (2 * m < kXfStep && k < ns) {
                        const float2 o = c_scale(v[a], sc);
                        if (vec && k + 1 < ns) *reinterpret_cast<float2*>(yr + k) = o;
                        else {
                            yr[k] = o.x;
                            if (k + 1 < ns) yr[k + 1] = o.y;
                        }
                    }
                });
            }
        }
        if (t + 1 < NT) lds_barrier();                          // the row buffer is rewritten by the next template
    });
}

constexpr size_t kXfWsFloats = 2 * 2 * kXfMB + 8 + 2 * kXfM1 * 2 + 2 * kXfNG;

}  // namespace d4w

using namespace d4w;

extern "C" {

int d4w_xcorr_fft_max_support(void) { return kXfPad + 1; }

size_t d4w_xcorr_fft_ws_bytes(void) { return kXfWsFloats * sizeof(float); }

int d4w_xcorr_fft_f32(const float* x, int nx, int ns, const float* mean, const float* maxabs, const float* taps,
                      int ntpl, int ltaps, int len0, int len1, float* y0, float* y1, void* ws, void* stream) {
    if (!x || !taps || !y0 || !ws || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (ntpl < 1 || ntpl > 2 || (ntpl == 2 && !y1)) return fail(D4W_EINVAL, "ntpl = %d (1 or 2 templates per call)", ntpl);
    if (ntpl == 1) len1 = len0;
    if (len0 < 1 || len1 < 1 || len0 > ltaps || len1 > ltaps || std::max(len0, len1) > kXfPad + 1)
        return fail(D4W_EINVAL, "template supports (%d, %d) must lie in 1..min(ltaps = %d, %d)", len0, len1, ltaps, kXfPad + 1);
    if (nx > 2 * 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 131070", nx);
    float* w = (float*)ws;
    XfTables T;
    float2* gp = (float2*)w;
    float* gn = w + 2 * 2 * kXfMB;
    float2* tw1 = (float2*)(gn + 8);
    float2* tw2 = tw1 + kXfM1;
    float2* wg = tw2 + kXfM1;
    T.gp = gp; T.gn = gn; T.tw1 = tw1; T.tw2 = tw2; T.wg = wg;
    D4W_LAUNCH(xcf_spectra, dim3(ceil_div(ntpl * kXfMB, 256)), dim3(256), 0, stream, taps, ntpl, ltaps, len0, len1, gp, gn,
               tw1, tw2, wg);
    const dim3 grid(ceil_div(ns, kXfStep), ceil_div(nx, 2));
    const size_t lds = ((size_t)2 * kXfRowP + 2 * kXfM1) * sizeof(float2);
    // register budget: 4 waves per SIMD (128 VGPRs) spills the two-template kernel; 3 (168 VGPRs) does not
    static const int waves = [] { const char* v = getenv("D4W_XF_WAVES"); return v ? atoi(v) : 3; }();
#define D4W_XF_LAUNCH(NT, WV)                                                                                   \
    do {                                                                                                        \
        D4W_LAUNCH((xcorr_fft_blocks<NT, WV>), grid, dim3(kXfThreads), lds, stream, T, x, nx, ns, mean, maxabs, \
                   y0, y1);                                                                                     \
    } while (0)
    if (ntpl == 1) { if (waves == 4) D4W_XF_LAUNCH(1, 4); else D4W_XF_LAUNCH(1, 3); }
    else { if (waves == 4) D4W_XF_LAUNCH(2, 4); else if (waves == 2) D4W_XF_LAUNCH(2, 2); else D4W_XF_LAUNCH(2, 3); }
#undef D4W_XF_LAUNCH
    return D4W_OK;
}

}  // extern "C"
