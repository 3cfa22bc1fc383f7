// Matched filter by overlap-save FFT correlation on MI355X (gfx950): replaces the per-row
// scipy.signal.correlate(x, template, 'full', 'fft') of detect.compute_cross_correlogram /
// detect.shift_xcorr (reference detect.py:96-166) for templates of short support (the fin-whale
// call templates have 136 / 156 non-zero samples).
//
// The direct form (rowops.hip: xcorr_fir) spends 2 (L0 + L1) = 584 flop per sample; this kernel ~90.
//   * a row is cut into blocks of B = 4096 samples that advance by S = B - 160 lags; the circular
//     correlation of a block is exact for its first S lags (support <= 161);
//   * a block is read as MB = 2048 packed complex samples; ONE complex FFT gives the block's real
//     spectrum after the untangle, which is multiplied by conj(T(f)) and re-tangled, and one inverse
//     FFT returns 2 lags per complex output;
//   * MB = 16 x 16 x 8 "fat" register stages exactly as in fk_fast.h (pass B): the first radix-16 is
//     applied to the registers the global loads landed in, the last one feeds the global stores, the
//     middle item does [radix 8 | pair op | inverse radix 8] for a group of 8 positions and its
//     Hermitian partner group;
//   * a workgroup (128 threads) transforms the same block of TWO adjacent rows at once, the two rows
//     riding the two halves of packed registers (fft_pair.h): every butterfly add / multiply is one
//     v_pk_* instruction for both rows;
//   * one launch per template (xcorr_fft_blocks); xcorr_fft_tpair is the one-read alternative that
//     packs the two TEMPLATES of one row through a single inverse transform (D4W_XF_TPAIR=1).
// Spectra and twiddle tables are built per call by xcf_spectra (~30 us).  docs/LAB_NOTEBOOK.md 3.3 has the
// measurements and the variants that were tried.
#include <cstdlib>

#include "fft_pair.h"

namespace d4w {

constexpr int kXfB = 4096, kXfMB = 2048;           // block length (real), packed length (complex)
constexpr int kXfNA = 16, kXfNB = 16, kXfNC = 8;   // DIF radices, MB = NA * NB * NC
constexpr int kXfM1 = kXfNB * kXfNC;               // 128
constexpr int kXfNG = kXfNA * kXfNB;               // 256 groups of NC positions
constexpr int kXfPad = 160;                        // lags lost per block = max support - 1
constexpr int kXfStep = kXfB - kXfPad;             // 3936 lags per block
constexpr int kXfThreads = 128;               // one item per thread in every stage
constexpr int kXfRowP = kXfMB + kXfNG;             // LDS row pitch: one pad element per group

__host__ __device__ constexpr int xf_ad(int e) { return e + e / kXfNC; }
// frequency held by position e after the three DIF stages (digits a', b', d)
__host__ __device__ constexpr int xf_freq(int e) {
    return (e / kXfM1) + kXfNA * ((e / kXfNC) % kXfNB) + kXfNA * kXfNB * (e % kXfNC);
}

struct XfTables {
    const float2* gp;     // [ntpl][MB] conj(T_t(f)) / 4 at position e (f = xf_freq(e))
    const float* gn;      // [ntpl]     conj(T_t(MB)) / 4 (real)
    const float2* tw1;    // [M1]       W_MB^j
    const float2* tw2;    // [NB][NC]   W_M1^(j2 b)
    const float2* wg;     // [NG]       W_B^(a' + NA b') : untangle twiddle of a group's first frequency
    const float2* twa;    // [NA][M1]   W_MB^(j a): stage-1 twiddles (a table read costs no VALU slot; the
                          //            power tree it replaces was 56 scalar multiplies per stage)
};

// stage-1 twiddles of item j1: loads issued early, consumed after the butterfly
template <int R>
__device__ __forceinline__ void xf_pw_load(const float2* __restrict__ twa, int j1, float2 (&pw)[R]) {
    pw[0] = make_float2(1.f, 0.f);
    static_for<R - 1>([&](auto qq) {
        constexpr int q = decltype(qq)::value + 1;
        pw[q] = twa[q * kXfM1 + j1];
    });
}

// spectra + twiddle tables (one launch per call; 2 x 2048 x L MACs)
__global__ __launch_bounds__(256) void xcf_spectra(const float* __restrict__ taps, int ntpl, int ltaps,
                                                   int len0, int len1, float2* __restrict__ gp,
                                                   float* __restrict__ gn, float2* __restrict__ tw1,
                                                   float2* __restrict__ tw2, float2* __restrict__ wg,
                                                   float2* __restrict__ twa) {
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
        gp[i] = make_float2((float)(0.25 * re), (float)(0.25 * im));   // 1/4: the pair op's factors 1/2, folded
    }
    if (i < ntpl) {
        const int L = i ? len1 : len0;
        double s = 0.0;
        for (int n = 0; n < L; ++n) s += (n & 1) ? -(double)taps[(size_t)i * ltaps + n] : (double)taps[(size_t)i * ltaps + n];
        gn[i] = (float)(0.25 * s);
    }
    if (i < kXfM1) {
        float s, c;
        sincospif(-2.0f * (float)i / (float)kXfMB, &s, &c);
        tw1[i] = make_float2(c, s);
        const int b = i / kXfNC, j2 = i % kXfNC;
        sincospif(-2.0f * (float)((j2 * b) % kXfM1) / (float)kXfM1, &s, &c);
        tw2[i] = make_float2(c, s);
    }
    if (i < kXfNA * kXfM1) {
        const int a = i / kXfM1, j = i % kXfM1;
        float sn, cs;
        sincospif(-2.0f * (float)((a * j) % kXfMB) / (float)kXfMB, &sn, &cs);
        twa[i] = make_float2(cs, sn);
    }
    if (i < kXfNG) {
        const int f0 = (i / kXfNB) + kXfNA * (i % kXfNB);
        float s, c;
        sincospif(-2.0f * (float)f0 / (float)kXfB, &s, &c);
        wg[i] = make_float2(c, s);
    }
}

// pair op for one frequency pair of both rows: A = Z[f], Bs = Z[MB - f], w = W_B^f, gf = conj(T(f)),
// gm = conj(T(MB - f)) (so that conj(T(f + MB)) = conj(gm)); na -> position of f, nb -> position of MB - f
__device__ __forceinline__ void xf_pair(c2 A, c2 Bs, float2 w, float2 gf, float2 gm, c2& na, c2& nb) {
    // the four factors 1/2 of the untangle / re-tangle algebra are folded into the tables (gf, gm = G / 4)
    const c2 Bc = c2_conj(Bs);
    const c2 E = c2_add(A, Bc);                            // 2 E
    const c2 O = c2_mul_mi(c2_sub(A, Bc));                 // 2 O
    const c2 tO = c2_mulw(O, w);
    const c2 Yp = c2_mulw(c2_add(E, tO), gf);             // X(f)      conj(T(f))      / 2
    const c2 Ym = c2_mulwc(c2_sub(E, tO), gm);            // X(f + MB) conj(T(f + MB)) / 2
    const c2 S = c2_add(Yp, Ym);
    const c2 D = c2_mul_pi(c2_mulwc(c2_sub(Yp, Ym), w));
    na = c2_add(S, D);
    nb = c2_conj(c2_sub(S, D));
}

// LDS element: (re_A, re_B, im_A, im_B)
// LDS tile STORES of these kernels are 8 bytes wide, not 16 (round 5).  With the matrix-core STFT (stft_mm_rows) resident on
// the same CU from another HIP stream, whole blocks of the first row of a row pair -- dwords 0 and 2 of the tile's 16-byte
// words -- came out 1-10 % off in a few workgroups per launch (scripts/probe/stream_race2.py, fence off: 6 of 6 trials with
// ds_read_b128 + ds_write_b128, 1 of 24 / 2 of 45 with 8-byte stores).  Round 5 fenced the two families against each other across
// streams (hazard_enter / hazard_leave below).  Round 6 ran these kernels beside kernels of OTHER libraries (hipBLASLt / rocBLAS
// GEMMs in binary16 and bfloat16, MIOpen's convolution; tests/test_concurrent_gpu.py): clean beside every one of them except
// a rocBLAS product 256 x 32768 x 256 (Tensile MT64x64x128), beside which 39-40 of 40 trials came out up to 37 % off -- a
// fence cannot know such a neighbour.  So the kernels now exclude co-residency by construction: every workgroup claims the
// compute unit's LDS (xcorr_fft_blocks: SUBS, xf_lds_claim): 0 of 20 trials beside that product, and beside the library's own
// STFT without the fence.  The 8-byte stores stay (free); the probe switches of round 5 (16-byte stores, 8-byte reads, LDS
// padding, the in-kernel read-back check) are gone from this file -- D4W_XF_LDS_CLAIM=0 is the A/B that shows the hazard;
// scripts/probe/ and DESIGN.md section 1 keep what the probes found.  The mechanism on the hardware's side is still not known.
__device__ __forceinline__ c2 xf_ld(const float4* p) {
    const float4 v = lds_read4(p);
    return c2{v2_make(v.x, v.y), v2_make(v.z, v.w)};
}
__device__ __forceinline__ void xf_st(float4* p, c2 v) {
#if !defined(D4W_EMU)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef volatile f2_t __attribute__((address_space(3))) * lds_f2_ptr;
    f2_t lo, hi;
    lo.x = v2_x(v.re); lo.y = v2_y(v.re); hi.x = v2_x(v.im); hi.y = v2_y(v.im);
    ((lds_f2_ptr)p)[0] = lo;
    ((lds_f2_ptr)p)[1] = hi;
#else
    *p = make_float4(v2_x(v.re), v2_y(v.re), v2_x(v.im), v2_y(v.im));
#endif
}

// FUSED (two templates off ONE read and ONE forward transform of the block): 256 threads; waves 0-1 run
// the forward stages into `buf`, then all four waves read their middle-stage groups of the block spectrum,
// and waves 0-1 / 2-3 carry template 0 / 1 through pair op and inverse stages -- template 0 in a second
// row buffer, template 1 in place in `buf` (every item owns its two groups, and a barrier separates the
// last read of the spectrum from the first write).  Same registers and code size as the one-template
// kernel, same 8 waves per CU (two 76 KiB workgroups), 3 instead of 4 transforms per row block and
// 12 instead of 16 bytes of HBM traffic per sample.
// A row seen through its neighbours (d4w_fir_fft_halo_f32): virtual samples [left (n_left) | the row (ns) | right (n_right)];
// lag / sample index i of the kernel is virtual sample v0 + i.  All zero = the plain row.
struct XfHalo {
    const float* left;      // [nx][ld_left], the n_left samples before the row (NULL: zeros)
    const float* right;     // [nx][ld_right], the n_right samples after it (NULL: zeros)
    int ld_left, n_left, ld_right, n_right, v0;
};

// SUBS (round 6): blocks of a row pair that ONE workgroup transforms side by side, each on its own kXfThreads threads and its own
// LDS -- the one-template kernel runs SUBS = 4: 512 threads and 152 KiB, i.e. the whole LDS of a compute unit (the launch asks for
// 159 KiB), with the occupancy the four 128-thread workgroups per CU had before.  The point is what CANNOT be resident beside it:
// these kernels returned wrong blocks with certain LDS-fed matrix kernels on the same CU (the library's own STFT in round 5; a
// rocBLAS GEMM, 256 x 32768 x 256 in binary16 / bfloat16, in 39-40 of 40 trials in round 6), and with the CU's LDS claimed no
// kernel that needs LDS can be (0 of 20 trials each; profiles/r06g/splitk_neighbour.txt).  Blocks beyond the row's last one do
// nothing (their lags lie outside the output) but keep the barriers' company.
#ifndef D4W_XF_LOOP
#define D4W_XF_LOOP 1
#endif
#ifndef D4W_XF_SPIN
#define D4W_XF_SPIN 0
#endif
constexpr int kXfSubs = 4;
constexpr size_t kXfItemLds = (size_t)kXfRowP * sizeof(float4) + 2 * kXfM1 * sizeof(float2);

template <int NT, bool FUSED, bool HALO = false, int SUBS = 1>      // HALO: a separate instantiation, the plain kernels keep their code
__global__ __launch_bounds__((FUSED ? 2 : SUBS) * kXfThreads, SUBS > 1 ? 1 : 2) void xcorr_fft_blocks(XfTables T, const float* __restrict__ x, int nx,
                                                                  int ns, const double* __restrict__ mean,
                                                                  const float* __restrict__ maxabs,
                                                                  float* __restrict__ y0, float* __restrict__ y1,
                                                                  int step, int yshift, int ns_out, float dcg, XfHalo H,
                                                                  const float* __restrict__ pivot, int ld, int ngrp) {
    // mean: the rows' float64 means (matched filter); pivot: instead, a float32 value per row that is taken off the samples
    // and given back through dcg (the zero-phase FIR's dynamic-range pivot, d4w_fir_fft_f32); ld: row pitch of x and y when
    // the rows handed over are a column window of longer rows (d4w_fir_fft_cols_f32), 0 = ns
    // step = lags kept per block (B - (support - 1)); lags k < ns_out are stored, lag k at column k + yshift of its row
    // (the zero-phase FIR use, d4w_fir_fft_f32: taps centred at yshift); dcg: mean[row] * dcg is added to every output
    // (the gain the subtracted constant would have had)
    constexpr int NA = kXfNA, NB = kXfNB, NC = kXfNC, M1 = kXfM1, MB = kXfMB, ROWP = kXfRowP;
    static_assert(!(FUSED && SUBS > 1), "sub-blocks are for the one-template kernel");
    D4W_DYN_LDS(smem_raw);
#ifdef D4W_EMU
    const int sub = SUBS > 1 ? (int)(threadIdx.x >> 7) : 0;
#else
    const int sub = SUBS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) : 0;          // 128 threads = two whole waves
#endif
    float4* buf = reinterpret_cast<float4*>(smem_raw + (size_t)sub * kXfItemLds);   // [ROWP] block spectra of both rows, then each template's correlation
    float2* tw1 = reinterpret_cast<float2*>(buf + (FUSED ? 2 : 1) * ROWP);        // [M1]
    float2* tw2 = tw1 + M1;                                     // [NB][NC]
#ifdef D4W_EMU
    const int tsel = FUSED ? (int)(threadIdx.x >> 7) : 0;
#else
    const int tsel = FUSED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) : 0;
#endif
    const int tid = (FUSED || SUBS > 1) ? (int)(threadIdx.x & (kXfThreads - 1)) : (int)threadIdx.x;
    const bool fwd = !FUSED || tsel == 0;                         // this wave runs the forward stages
    float4* mine = FUSED ? (tsel ? buf : buf + ROWP) : buf;       // row buffer of this wave's template
    if (fwd) {
        tw1[tid] = T.tw1[tid];
        tw2[tid] = T.tw2[tid];
    }
    // SUBS > 1: the two waves of a sub-block meet at THEIR OWN barrier -- an LDS counter both add to -- so that the four blocks of
    // a workgroup run out of step with each other like the four workgroups per CU they replace (with one shared s_barrier all
    // eight waves load, wait and store together and nothing hides the memory latency: 8.1 against 5.7 ms at 20 000 x 120 000)
    unsigned* cnt = reinterpret_cast<unsigned*>(smem_raw + (size_t)SUBS * kXfItemLds) + sub;
    if constexpr (SUBS > 1) {
        if (tid == 0) *cnt = 0u;
        __syncthreads();
    }
    unsigned bar_epoch = 0;                                         // barriers this wave has been through x 2: what the counter reads when both waves are there
    auto bar = [&]() {
#ifndef D4W_EMU
        if constexpr (SUBS > 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // this wave's LDS traffic is done
            bar_epoch += 2u;
            if ((tid & 63) == 0) (void)atomicAdd(cnt, 1u);                      // ds_add_u32, nothing returned: the wave knows what to wait for
            while ((int)(*reinterpret_cast<volatile unsigned*>(cnt) - bar_epoch) < 0) __builtin_amdgcn_s_sleep(D4W_XF_SPIN);
            asm volatile("" ::: "memory");
            return;
        }
#endif
        lds_barrier();
    };
    const int rowA = 2 * blockIdx.y;
    const bool hasB = rowA + 1 < nx;
    const int rowB = hasB ? rowA + 1 : rowA;
    const size_t pitch = ld ? (size_t)ld : (size_t)ns;
    const float* xa = x + (size_t)rowA * pitch;
    const float* xb = x + (size_t)rowB * pitch;
    Mean2 mua = mean2_load(mean, rowA), mub = mean2_load(mean, rowB);
    if (pivot) {
        mua.hi = pivot[rowA];
        mub.hi = pivot[rowB];
    }
    float ga_ = 1.f, gb_ = 1.f;
    if (maxabs) {
        const float a = maxabs[rowA], b = maxabs[rowB];
        ga_ = (a > 0.f) ? 1.0f / a : 0.f;
        gb_ = (b > 0.f) ? 1.0f / b : 0.f;
    }
    const v2f sc = v2_make(ga_ / (float)MB, gb_ / (float)MB);
    // a workgroup walks the groups of SUBS blocks grp, grp + gridDim.x, ... of its row pair (SUBS = 1: its one block)
    int grp = (int)blockIdx.x;
    do {
    // (the table operands below depend on the lane only: left alone the compiler hoists ~70 registers' worth of them out of
    // this loop and spills; an opaque zero in their addresses keeps them where they are used -- they are L1 / L2 hits)
    int lz = 0;
#ifndef D4W_EMU
    if constexpr (SUBS > 1) asm volatile("" : "+v"(lz));          // (per lane: the lanes' table addresses are formed in here too)
#endif
    const int tl = tid + lz;                                    // the lane's item number, opaque per group for the same reason
    const int k0 = (grp * SUBS + sub) * step;                   // first lag / first sample of the block
    const int c0 = HALO ? H.v0 + k0 - H.n_left : k0;             // the block's first sample in the row's own coordinates
    // 8-byte aligned sample pairs (the base may be a column window of longer rows: the address decides, not the index)
    const bool veca = ((reinterpret_cast<uintptr_t>(xa + c0) & 7) == 0), vecb = ((reinterpret_cast<uintptr_t>(xb + c0) & 7) == 0);
    const bool interior = (!HALO || c0 >= 0) && (c0 + kXfB <= ns) && veca && vecb;   // whole block inside both rows, 8-byte aligned pairs
    // NT templates per launch: the host launches NT = 1 once per template.  Measured at 20000 x 120000
    // (HF + LF): two NT = 1 launches 9.1 ms; one NT = 2 launch 11.2 ms (57 KiB of straight-line code
    // against a 64 KiB instruction cache shared by two CUs), 11.6 ms when the block spectrum is kept
    // in registers across the templates instead (VGPR spills).
    // middle-stage item of this thread: p < 127 a proper pair of groups (Gi < PG), p = 127 the two
    // self-paired groups 0 and NB / 2 (see the MID stage)
    int Gi, PG;
    {
        const int p = tl;
        if (p < 112) { const int g = 1 + (p >> 4), b = p & 15; Gi = g * NB + b; PG = (NA - g) * NB + (NB - 1 - b); }
        else if (p < 120) { const int b = p - 112; Gi = (NA / 2) * NB + b; PG = (NA / 2) * NB + (NB - 1 - b); }
        else if (p < 127) { const int b = p - 119; Gi = b; PG = NB - b; }
        else { Gi = 0; PG = NB / 2; }
    }
    const bool selfitem = (tl == 127);
    static_for<NT>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        // ---------------- S1: radix NA on the packed samples z[m] = x[k0 + 2m] + i x[k0 + 2m + 1], m = j1 + a M1
        c2 pf[NA];
        {
            const int j1 = tl;
            // sample c of the row in its own coordinates; outside [0, ns) the neighbours' halos (de-meaned alike), then zeros
            auto sample = [&](const float* xr, int row, Mean2 mu, int c) -> float {
                if (HALO && c < 0) {
                    const int l = c + H.n_left;
                    return (H.left && l >= 0) ? demean(H.left[(size_t)row * H.ld_left + l], mu) : 0.f;
                }
                if (c < ns) return demean(xr[c], mu);
                const int rr = c - ns;
                return (HALO && H.right && rr < H.n_right) ? demean(H.right[(size_t)row * H.ld_right + rr], mu) : 0.f;
            };
            auto fetch = [&](const float* xr, int row, Mean2 mu, bool vec, int i) -> float2 {
                const int c = HALO ? i + H.v0 - H.n_left : i;
                if (vec && (!HALO || c >= 0) && c + 1 < ns) {
                    float2 v = *reinterpret_cast<const float2*>(xr + c);
                    v.x = demean(v.x, mu);
                    v.y = demean(v.y, mu);
                    return v;
                }
                return make_float2(sample(xr, row, mu, c), sample(xr, row, mu, c + 1));
            };
            if (!fwd) {
            } else if (interior) {
                const float2* pa = reinterpret_cast<const float2*>(xa + c0) + j1;
                const float2* pb = reinterpret_cast<const float2*>(xb + c0) + j1;
                const v2f mu2 = v2_make(mua.hi, mub.hi), mu2l = v2_make(mua.lo, mub.lo);
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const float2 va = pa[a * M1], vb = pb[a * M1];
                    pf[a] = c2{v2_sub(v2_sub(v2_make(va.x, vb.x), mu2), mu2l), v2_sub(v2_sub(v2_make(va.y, vb.y), mu2), mu2l)};
                });
            } else if (veca && vecb && ns >= 4) {        // (without neighbours too: the last block of a row, a quarter of the blocks of a 60-s file)
                // a block that straddles a file boundary: most of its samples are still in the row -- their 8-byte loads
                // go out together (clamped addresses), the few samples beyond the row are fetched one by one afterwards
                const v2f mu2 = v2_make(mua.hi, mub.hi), mu2l = v2_make(mua.lo, mub.lo);
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    // clamped to the row without changing the parity of the index (the pairs stay 8-byte aligned)
                    const int par = c0 & 1, c = c0 + 2 * (j1 + a * M1), cc = par + (min(max(c - par, 0), ns - 2 - par) & ~1);
                    const float2 va = *reinterpret_cast<const float2*>(xa + cc);
                    const float2 vb = *reinterpret_cast<const float2*>(xb + cc);
                    pf[a] = c2{v2_sub(v2_sub(v2_make(va.x, vb.x), mu2), mu2l), v2_sub(v2_sub(v2_make(va.y, vb.y), mu2), mu2l)};
                });
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const int i = k0 + 2 * (j1 + a * M1), c = c0 + 2 * (j1 + a * M1);
                    if (c < 0 || c + 1 >= ns) pf[a] = c2_make(fetch(xa, rowA, mua, veca, i), fetch(xb, rowB, mub, vecb, i));
                });
            } else {
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const int i = k0 + 2 * (j1 + a * M1);
                    pf[a] = c2_make(fetch(xa, rowA, mua, veca, i), fetch(xb, rowB, mub, vecb, i));
                });
            }
        }
        float2 pw[NA];
        xf_pw_load<NA>(T.twa, tl, pw);                        // in flight together with the samples
        if (t == 0) bar();                                          // twiddle tables visible (and, from the second group on, the last reads of the row buffer done)
        if (fwd) {
            const int j1 = tl;
            dftp<NA>(pf);
            static_for<NA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                xf_st(buf + xf_ad(j1 + a * M1), (a == 0) ? pf[0] : c2_mulw(pf[a], pw[a]));
            });
        }
        // the middle stage's table operands: issued here, in flight across S2
        float2 GA[NC], GB[NC];
        {
            const float2* gpa = T.gp + (size_t)(FUSED ? tsel : t) * MB + Gi * NC;
            const float2* gpb = T.gp + (size_t)(FUSED ? tsel : t) * MB + PG * NC;
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                GA[d] = gpa[d];
                GB[d] = gpb[d];
            });
        }
        const float gny = T.gn[FUSED ? tsel : t];
        const float2 wa0 = T.wg[Gi], wb0 = T.wg[PG];       // W_B^f, f = f0(G) + 256 d: W_B^(256 d) are literals
        bar();
        // ---------------- S2: radix NB in place, x W_M1^(j2 b')
        if (fwd) {
            const int g = tl >> 3, j2 = tl & 7;
            c2 v[NB];
            static_for<NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                v[b] = xf_ld(buf + xf_ad(g * M1 + j2 + b * NC));
            });
            dftp<NB>(v);
            static_for<NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                xf_st(buf + xf_ad(g * M1 + j2 + b * NC), (b == 0) ? v[0] : c2_mulw(v[b], tw2[b * NC + j2]));
            });
        }
        bar();
        // ---------------- per template: MID (radix NC on a group and its Hermitian partner group, pair op,
        //                  inverse radix NC), S2' and S1'.  Item p < 127: a proper pair (Gi < PG); p = 127: the
        //                  two self-paired groups 0 (digit partner (NC - d) % NC, f = 0 pairs with the Nyquist
        //                  bin) and NB / 2.
        // the item's two groups of the block spectrum
        c2 a[NC], b[NC];
        {
            const float4* ga = buf + xf_ad(Gi * NC);
            const float4* gb = buf + xf_ad(PG * NC);
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                a[d] = xf_ld(ga + d);
                b[d] = xf_ld(gb + d);
            });
            dftp<NC>(a);
            dftp<NC>(b);
        }
        if (FUSED) bar();                                   // every read of the spectrum precedes the in-place writes
        {
            c2 ra[NC], rb[NC];
            if (!selfitem) {
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pn = NC - 1 - d;
                    xf_pair(a[d], b[pn], rot_const<d, 16>(wa0), GA[d], GB[pn], ra[d], rb[pn]);
                });
            } else {
                // group 0 (array a): partner digit (NC - d) % NC; d = 0 pairs X(0) with the Nyquist bin,
                // d = 0 and d = NC / 2 are their own partners (the partner-side value is kept)
                static_for<NC / 2 + 1>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pz = (NC - d) % NC;
                    const float2 gm = (d == 0) ? make_float2(gny, 0.f) : GA[pz];
                    c2 na;
                    xf_pair(a[d], a[pz], rot_const<d, 16>(wa0), GA[d], gm, na, ra[pz]);
                    if constexpr (pz != d) ra[d] = na;
                });
                // group NB / 2 (array b): partner digit NC - 1 - d, no self-paired position
                static_for<NC / 2>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pn = NC - 1 - d;
                    xf_pair(b[d], b[pn], rot_const<d, 16>(wb0), GB[d], GB[pn], rb[d], rb[pn]);
                });
            }
            idftp<NC>(ra);
            idftp<NC>(rb);
            float4* oa = mine + xf_ad(Gi * NC);
            float4* ob = mine + xf_ad(PG * NC);
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                xf_st(oa + d, ra[d]);
                xf_st(ob + d, rb[d]);
            });
        }
        bar();
        float2 pwi[NA];
        xf_pw_load<NA>(T.twa, tl, pwi);                       // for S1', in flight across S2'
        // ---------------- S2': inverse radix NB
        {
            const int g = tl >> 3, j2 = tl & 7;
            c2 v[NB];
            static_for<NB>([&](auto bb) {
                constexpr int bq = decltype(bb)::value;
                const c2 xv = xf_ld(mine + xf_ad(g * M1 + j2 + bq * NC));
                v[bq] = (bq == 0) ? xv : c2_mulwc(xv, tw2[bq * NC + j2]);
            });
            idftp<NB>(v);
            static_for<NB>([&](auto bb) {
                constexpr int bq = decltype(bb)::value;
                xf_st(mine + xf_ad(g * M1 + j2 + bq * NC), v[bq]);
            });
        }
        bar();
        // ---------------- S1': inverse radix NA -> lags k0 + 2m, k0 + 2m + 1 (m = j1 + a M1), the first S of them
        {
            const int j1 = tl;
            c2 v[NA];
            static_for<NA>([&](auto aa) {
                constexpr int aq = decltype(aa)::value;
                const c2 xv = xf_ld(mine + xf_ad(j1 + aq * M1));
                v[aq] = (aq == 0) ? xv : c2_mulwc(xv, pwi[aq]);
            });
            idftp<NA>(v);
            float* ya = ((FUSED ? tsel : t) == 0 ? y0 : y1) + (size_t)rowA * pitch + yshift;
            float* yb = ((FUSED ? tsel : t) == 0 ? y0 : y1) + (size_t)rowB * pitch + yshift;
            const v2f dc = v2_make((mua.hi + mua.lo) * dcg, (mub.hi + mub.lo) * dcg);
            // stores: with neighbours (HALO) the block's lags can all lie inside the output although its samples straddle a
            // file boundary -- the wide store path is chosen from the output's own geometry then
            const bool oveca = ((reinterpret_cast<uintptr_t>(ya + k0) & 7) == 0);
            const bool ovecb = ((reinterpret_cast<uintptr_t>(yb + k0) & 7) == 0);
            const bool ointerior = HALO ? (k0 + step <= ns_out && oveca && ovecb) : (interior && oveca && ovecb);
            if (ointerior) {
                float2* oa = reinterpret_cast<float2*>(ya + k0) + j1;
                float2* ob = reinterpret_cast<float2*>(yb + k0) + j1;
                static_for<NA>([&](auto aa) {
                    constexpr int aq = decltype(aa)::value;
                    if (2 * (j1 + aq * M1) < step) {
                        c2 o = c2_scale2(v[aq], sc);
                        o = c2{v2_add(o.re, dc), v2_add(o.im, dc)};
                        st_stream(oa + aq * M1, c2_a(o));
                        if (hasB) st_stream(ob + aq * M1, c2_b(o));
                    }
                });
            } else {
                auto put = [&](float* yr, bool vec, int k, float2 o) {
                    if (k >= ns_out) return;
                    if (vec && k + 1 < ns_out) *reinterpret_cast<float2*>(yr + k) = o;
                    else {
                        yr[k] = o.x;
                        if (k + 1 < ns_out) yr[k + 1] = o.y;
                    }
                };
                static_for<NA>([&](auto aa) {
                    constexpr int aq = decltype(aa)::value;
                    const int m = j1 + aq * M1;
                    if (2 * m < step) {
                        c2 o = c2_scale2(v[aq], sc);
                        o = c2{v2_add(o.re, dc), v2_add(o.im, dc)};
                        put(ya, oveca, k0 + 2 * m, c2_a(o));
                        if (hasB) put(yb, ovecb, k0 + 2 * m, c2_b(o));
                    }
                });
            }
        }
        if (t + 1 < NT) bar();                          // the row buffer is rewritten by the next template
    });
    if constexpr (SUBS == 1 || !D4W_XF_LOOP) break;             // no loop in the generated code (D4W_XF_LOOP=0: one group per workgroup)
    grp += (int)gridDim.x;
    } while (grp < ngrp);
}

// ---------------------------------------------------------------------------------------------
// Two templates, one row per workgroup: the block is read ONCE, its forward transform runs in scalar
// complex arithmetic, and the two correlations ride the halves of packed registers through the pair
// op and ONE inverse transform (fft_pair.h with the pair = the two templates).  Same VALU work as
// two row-pair launches, 12 instead of 16 bytes of HBM traffic per sample.
//   forward S1, S2 (float2 LDS elements) -> every item reads its two groups into registers, radix 8
//   -> barrier -> pair op for both templates -> inverse radix 8 -> (float4 LDS elements, in place)
//   -> S2', S1' packed -> y0 and y1.
// ---------------------------------------------------------------------------------------------
// conj(a) * w for packed a and one complex scalar w
__device__ __forceinline__ c2 c2_cmulw(c2 a, float2 w) {
    return c2{v2_fma(a.im, w.y, v2_muls(a.re, w.x)), v2_fnma(a.im, w.x, v2_muls(a.re, w.y))};
}

// A = Z[f], Bs = Z[MB - f] (scalars); gf = (conj T0(f), conj T1(f)), gm = the same at MB - f
__device__ __forceinline__ void xf_tpair(float2 A, float2 Bs, float2 w, c2 gf, c2 gm, c2& na, c2& nb) {
    const float2 Bc = c_conj(Bs);
    const float2 E = c_add(A, Bc);                          // 2 E (the factors 1/2 live in the tables)
    const float2 O = c_mul_mi(c_sub(A, Bc));
    const float2 tO = c_mul(w, O);
    const c2 Yp = c2_mulw(gf, c_add(E, tO));              // X(f)      conj(T_t(f))      / 2
    const c2 Ym = c2_cmulw(gm, c_sub(E, tO));             // X(f + MB) conj(T_t(f + MB)) / 2
    const c2 S = c2_add(Yp, Ym);
    const c2 D = c2_mul_pi(c2_mulwc(c2_sub(Yp, Ym), w));
    na = c2_add(S, D);
    nb = c2_conj(c2_sub(S, D));
}

__global__ __launch_bounds__(kXfThreads, 2) void xcorr_fft_tpair(XfTables T, const float* __restrict__ x, int ns,
                                                                 const double* __restrict__ mean,
                                                                 const float* __restrict__ maxabs,
                                                                 float* __restrict__ y0, float* __restrict__ y1) {
    constexpr int NA = kXfNA, NB = kXfNB, NC = kXfNC, M1 = kXfM1, MB = kXfMB, ROWP = kXfRowP;
    D4W_DYN_LDS(smem_raw);
    float4* buf = reinterpret_cast<float4*>(smem_raw);          // [ROWP] packed correlations of the two templates
    float2* bufs = reinterpret_cast<float2*>(smem_raw);         // the same memory as float2: the block spectrum
    float2* tw1 = reinterpret_cast<float2*>(buf + ROWP);        // [M1]
    float2* tw2 = tw1 + M1;                                     // [NB][NC]
    const int tid = threadIdx.x;
    tw1[tid] = T.tw1[tid];
    tw2[tid] = T.tw2[tid];
    const int row = blockIdx.y;
    const int k0 = blockIdx.x * kXfStep;
    const float* xr = x + (size_t)row * ns;
    const Mean2 mu = mean2_load(mean, row);
    float gain = 1.f;
    if (maxabs) {
        const float a = maxabs[row];
        gain = (a > 0.f) ? 1.0f / a : 0.f;
    }
    const float sc = gain / (float)MB;
    const bool vec = ((((size_t)row * ns + k0) & 1) == 0);
    const bool interior = (k0 + kXfB <= ns) && vec;
    // ---------------- S1 (scalar): radix NA on z[m] = x[k0 + 2m] + i x[k0 + 2m + 1], m = j1 + a M1
    {
        const int j1 = tid;
        float2 pf[NA];
        if (interior) {
            const float2* pa = reinterpret_cast<const float2*>(xr + k0) + j1;
            static_for<NA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                const float2 v = pa[a * M1];
                pf[a] = make_float2(demean(v.x, mu), demean(v.y, mu));
            });
        } else {
            static_for<NA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                const int i = k0 + 2 * (j1 + a * M1);
                float2 v = make_float2(0.f, 0.f);
                if (i < ns) v.x = demean(xr[i], mu);
                if (i + 1 < ns) v.y = demean(xr[i + 1], mu);
                pf[a] = v;
            });
        }
        __syncthreads();                                        // twiddle tables visible
        dft<NA>(pf);
        float2 pw[NA];
        xf_pw_load<NA>(T.twa, j1, pw);
        static_for<NA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            bufs[xf_ad(j1 + a * M1)] = (a == 0) ? pf[0] : c_mul(pf[a], pw[a]);
        });
    }
    lds_barrier();
    // ---------------- S2 (scalar): radix NB in place, x W_M1^(j2 b')
    {
        const int g = tid >> 3, j2 = tid & 7;
        float2 v[NB];
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            v[b] = bufs[xf_ad(g * M1 + j2 + b * NC)];
        });
        dft<NB>(v);
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            bufs[xf_ad(g * M1 + j2 + b * NC)] = (b == 0) ? v[0] : c_mul(v[b], tw2[b * NC + j2]);
        });
    }
    lds_barrier();
    // ---------------- MID: the item's two groups (see xcorr_fft_blocks for the item list)
    int Gi, PG;
    {
        const int p = tid;
        if (p < 112) { const int g = 1 + (p >> 4), b = p & 15; Gi = g * NB + b; PG = (NA - g) * NB + (NB - 1 - b); }
        else if (p < 120) { const int b = p - 112; Gi = (NA / 2) * NB + b; PG = (NA / 2) * NB + (NB - 1 - b); }
        else if (p < 127) { const int b = p - 119; Gi = b; PG = NB - b; }
        else { Gi = 0; PG = NB / 2; }
    }
    const bool selfitem = (tid == 127);
    {
        float2 a[NC], b[NC];
        {
            const float2* ga = bufs + xf_ad(Gi * NC);
            const float2* gb = bufs + xf_ad(PG * NC);
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                a[d] = ga[d];
                b[d] = gb[d];
            });
        }
        // template spectra of both groups, packed over the two templates
        c2 GA[NC], GB[NC];
        {
            const float2* g0a = T.gp + Gi * NC;
            const float2* g0b = T.gp + PG * NC;
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                GA[d] = c2_make(g0a[d], g0a[MB + d]);
                GB[d] = c2_make(g0b[d], g0b[MB + d]);
            });
        }
        const float2 wa0 = T.wg[Gi], wb0 = T.wg[PG];
        const c2 gny = c2{v2_make(T.gn[0], T.gn[1]), v2_make(0.f, 0.f)};
        lds_barrier();                                          // every item has its groups: the buffer may be overwritten
        dft<NC>(a);
        dft<NC>(b);
        c2 ra[NC], rb[NC];
        if (!selfitem) {
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d;
                xf_tpair(a[d], b[pn], rot_const<d, 16>(wa0), GA[d], GB[pn], ra[d], rb[pn]);
            });
        } else {
            static_for<NC / 2 + 1>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pz = (NC - d) % NC;
                c2 na;
                xf_tpair(a[d], a[pz], rot_const<d, 16>(wa0), GA[d], (d == 0) ? gny : GA[pz], na, ra[pz]);
                if constexpr (pz != d) ra[d] = na;
            });
            static_for<NC / 2>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d;
                xf_tpair(b[d], b[pn], rot_const<d, 16>(wb0), GB[d], GB[pn], rb[d], rb[pn]);
            });
        }
        idftp<NC>(ra);
        idftp<NC>(rb);
        float4* oa = buf + xf_ad(Gi * NC);
        float4* ob = buf + xf_ad(PG * NC);
        static_for<NC>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            xf_st(oa + d, ra[d]);
            xf_st(ob + d, rb[d]);
        });
    }
    lds_barrier();
    // ---------------- S2' (packed over the templates)
    {
        const int g = tid >> 3, j2 = tid & 7;
        c2 v[NB];
        static_for<NB>([&](auto bb) {
            constexpr int bq = decltype(bb)::value;
            const c2 xv = xf_ld(buf + xf_ad(g * M1 + j2 + bq * NC));
            v[bq] = (bq == 0) ? xv : c2_mulwc(xv, tw2[bq * NC + j2]);
        });
        idftp<NB>(v);
        static_for<NB>([&](auto bb) {
            constexpr int bq = decltype(bb)::value;
            xf_st(buf + xf_ad(g * M1 + j2 + bq * NC), v[bq]);
        });
    }
    lds_barrier();
    // ---------------- S1' -> y0, y1
    {
        const int j1 = tid;
        float2 pw[NA];
        xf_pw_load<NA>(T.twa, j1, pw);
        c2 v[NA];
        static_for<NA>([&](auto aa) {
            constexpr int aq = decltype(aa)::value;
            const c2 xv = xf_ld(buf + xf_ad(j1 + aq * M1));
            v[aq] = (aq == 0) ? xv : c2_mulwc(xv, pw[aq]);
        });
        idftp<NA>(v);
        float* ya = y0 + (size_t)row * ns;
        float* yb = y1 + (size_t)row * ns;
        if (interior) {
            float2* oa = reinterpret_cast<float2*>(ya + k0) + j1;
            float2* ob = reinterpret_cast<float2*>(yb + k0) + j1;
            static_for<NA>([&](auto aa) {
                constexpr int aq = decltype(aa)::value;
                if (2 * (j1 + aq * M1) < kXfStep) {
                    const c2 o = c2_scale(v[aq], sc);
                    oa[aq * M1] = c2_a(o);
                    ob[aq * M1] = c2_b(o);
                }
            });
        } else {
            auto put = [&](float* yr, int k, float2 o) {
                if (k >= ns) return;
                yr[k] = o.x;
                if (k + 1 < ns) yr[k + 1] = o.y;
            };
            static_for<NA>([&](auto aa) {
                constexpr int aq = decltype(aa)::value;
                const int m = j1 + aq * M1;
                if (2 * m < kXfStep) {
                    const c2 o = c2_scale(v[aq], sc);
                    put(ya, k0 + 2 * m, c2_a(o));
                    put(yb, k0 + 2 * m, c2_b(o));
                }
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two templates, FOUR radix stages (MB = 8 x 8 x 8 x 4), 512 threads: the kernel the two-template
// matched filter runs.  Same algebra as xcorr_fft_blocks<1, true> (one read and ONE forward transform of
// a block of two packed rows, waves 0-3 / 4-7 carry template 0 / 1 through pair op and inverse stages,
// template 1 in place in the spectrum buffer, template 0 in a second row buffer), but every stage has
// 256 items of radix <= 8: ~100 VGPRs instead of 212-233, so that two 76-KiB workgroups put 16 waves on
// a CU instead of 8 -- the three-stage kernel is bound by what its 8 waves can overlap, not by VALU
// issue or HBM bandwidth.  One more LDS round trip per direction.
//   position e = d0 256 + d1 32 + d2 4 + d3 holds frequency d0 + 8 d1 + 64 d2 + 512 d3 after the forward
//   stages (DIF, digit-reversed in place); groups of 4 consecutive positions; Hermitian partner of group
//   (d0, d1, d2): (8-d0, 7-d1, 7-d2) | d0 = 0: (0, 8-d1, 7-d2) | d0 = d1 = 0: (0, 0, 8-d2); groups 0 and 4
//   are their own partners (item 255).
// ---------------------------------------------------------------------------------------------
constexpr int kX4Items = 256;                      // items per stage = threads per template half
constexpr int kX4NG = kXfMB / 4;                   // 512 groups of 4 positions
constexpr int kX4RowP = kXfMB + kXfMB / 8;         // one pad element per 8 positions

__host__ __device__ constexpr int x4_ad(int e) { return e + (e >> 3); }
__host__ __device__ constexpr int x4_freq(int e) { return (e >> 8) + 8 * ((e >> 5) & 7) + 64 * ((e >> 2) & 7) + 512 * (e & 3); }
__host__ __device__ constexpr int x4_partner(int G) {
    const int d0 = G >> 6, d1 = (G >> 3) & 7, d2 = G & 7;
    if (d0) return ((8 - d0) << 6) | ((7 - d1) << 3) | (7 - d2);
    if (d1) return ((8 - d1) << 3) | (7 - d2);
    if (d2) return 8 - d2;
    return 0;
}

struct X4Tables {
    const float2* gp;     // [2][MB]   conj(T_t(f)) / 4 at position e (f = x4_freq(e))
    const float* gn;      // [2]       conj(T_t(MB)) / 4 (real)
    const float2* tw1;    // [8][256]  W_MB^(j q)
    const float2* tw2;    // [8][32]   W_256^(j q)
    const float2* tw3;    // [8][4]    W_32^(j q)
    const float2* wg;     // [512]     W_B^(frequency of the group's first position)
    const int2* pairs;    // [256]     (Gi, PG) of every middle-stage item; item 255 = the self-paired groups (0, 4)
};

__global__ __launch_bounds__(256) void xcf_tables4(const float* __restrict__ taps, int ltaps, int len0, int len1,
                                                   float2* __restrict__ gp, float* __restrict__ gn,
                                                   float2* __restrict__ tw1, float2* __restrict__ tw2,
                                                   float2* __restrict__ tw3, float2* __restrict__ wg,
                                                   int2* __restrict__ pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * kXfMB) {
        const int t = i / kXfMB, e = i - t * kXfMB;
        const int f = x4_freq(e), L = t ? len1 : len0;
        const float* tp = taps + (size_t)t * ltaps;
        double re = 0.0, im = 0.0;
        for (int n = 0; n < L; ++n) {
            float sn, cs;
            sincospif(2.0f * (float)((f * n) & (kXfB - 1)) / (float)kXfB, &sn, &cs);
            re += (double)tp[n] * cs;
            im += (double)tp[n] * sn;
        }
        gp[i] = make_float2((float)(0.25 * re), (float)(0.25 * im));
        const int q = e >> 8, j = e & 255;                       // stage-1 twiddles W_MB^(j q)
        float sn, cs;
        if (t == 0) {
            sincospif(-2.0f * (float)((q * j) % kXfMB) / (float)kXfMB, &sn, &cs);
            tw1[e] = make_float2(cs, sn);
        }
    }
    if (i < 2) {
        const int L = i ? len1 : len0;
        double sacc = 0.0;
        for (int n = 0; n < L; ++n) sacc += (n & 1) ? -(double)taps[(size_t)i * ltaps + n] : (double)taps[(size_t)i * ltaps + n];
        gn[i] = (float)(0.25 * sacc);
    }
    if (i < 256) {
        const int q = i >> 5, j = i & 31;
        float sn, cs;
        sincospif(-2.0f * (float)((q * j) % 256) / 256.0f, &sn, &cs);
        tw2[i] = make_float2(cs, sn);
    }
    if (i < 32) {
        const int q = i >> 2, j = i & 3;
        float sn, cs;
        sincospif(-2.0f * (float)((q * j) % 32) / 32.0f, &sn, &cs);
        tw3[i] = make_float2(cs, sn);
    }
    if (i < kX4NG) {
        float sn, cs;
        sincospif(-2.0f * (float)x4_freq(4 * i) / (float)kXfB, &sn, &cs);
        wg[i] = make_float2(cs, sn);
    }
    if (i == 0) {
        int n = 0;
        for (int G = 0; G < kX4NG; ++G) {
            const int PG = x4_partner(G);
            if (G < PG) pairs[n++] = make_int2(G, PG);
        }
        pairs[kX4Items - 1] = make_int2(0, 4);
    }
}

template <bool CONT>      // CONT: the rows continue in xnext (a separate instantiation: the plain kernel keeps its registers)
__global__ __launch_bounds__(2 * kX4Items, 4) void xcorr_fft_fused4(X4Tables T, const float* __restrict__ x, int nx, int ns,
                                                                    const double* __restrict__ mean,
                                                                    const float* __restrict__ maxabs,
                                                                    float* __restrict__ y0, float* __restrict__ y1,
                                                                    const float* __restrict__ xnext, int ld_next, int n_next) {
    constexpr int MB = kXfMB, ROWP = kX4RowP;
    D4W_DYN_LDS(smem_raw);
    float4* buf = reinterpret_cast<float4*>(smem_raw);            // [ROWP] block spectrum, then template 1's correlation
    float2* tw2 = reinterpret_cast<float2*>(buf + 2 * ROWP);      // [8][32]
    float2* tw3 = tw2 + 256;                                      // [8][4]
    const int tsel = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int tid = (int)(threadIdx.x & (kX4Items - 1));
    const bool fwd = tsel == 0;
    float4* mine = tsel ? buf : buf + ROWP;
    if (fwd) {
        tw2[tid] = T.tw2[tid];
        if (tid < 32) tw3[tid] = T.tw3[tid];
    }
    const int rowA = 2 * blockIdx.y;
    const bool hasB = rowA + 1 < nx;
    const int rowB = hasB ? rowA + 1 : rowA;
    const int k0 = blockIdx.x * kXfStep;
    const float* xa = x + (size_t)rowA * ns;
    const float* xb = x + (size_t)rowB * ns;
    const Mean2 mua = mean2_load(mean, rowA), mub = mean2_load(mean, rowB);
    float ga_ = 1.f, gb_ = 1.f;
    if (maxabs) {
        const float a = maxabs[rowA], b = maxabs[rowB];
        ga_ = (a > 0.f) ? 1.0f / a : 0.f;
        gb_ = (b > 0.f) ? 1.0f / b : 0.f;
    }
    const v2f sc = v2_make(ga_ / (float)MB, gb_ / (float)MB);
    const bool veca = ((((size_t)rowA * ns + k0) & 1) == 0), vecb = ((((size_t)rowB * ns + k0) & 1) == 0);
    const bool interior = (k0 + kXfB <= ns) && veca && vecb;
    // ---------------- S1: radix 8 on the packed samples z[m] = x[k0 + 2m] + i x[k0 + 2m + 1], m = tid + 256 q
    c2 pf[8];
    float2 pw[8];
    if (fwd) {
        // beyond the row: the first n_next samples of the record's continuation (the next file's rows, pitch ld_next),
        // de-meaned like the row's own samples, then zeros
        auto sample = [&](const float* xr, const float* xn, Mean2 mu, int i) -> float {
            if (i < ns) return demean(xr[i], mu);
            if (CONT && i - ns < n_next) return demean(xn[i - ns], mu);
            return 0.f;
        };
        auto fetch = [&](const float* xr, const float* xn, Mean2 mu, bool vec, int i) -> float2 {
            if (vec && i + 1 < ns) {
                float2 v = *reinterpret_cast<const float2*>(xr + i);
                v.x = demean(v.x, mu);
                v.y = demean(v.y, mu);
                return v;
            }
            return make_float2(sample(xr, xn, mu, i), sample(xr, xn, mu, i + 1));
        };
        const float* xna = CONT ? xnext + (size_t)rowA * ld_next : nullptr;
        const float* xnb = CONT ? xnext + (size_t)rowB * ld_next : nullptr;
        if (interior) {
            const float2* pa = reinterpret_cast<const float2*>(xa + k0) + tid;
            const float2* pb = reinterpret_cast<const float2*>(xb + k0) + tid;
            const v2f mu2 = v2_make(mua.hi, mub.hi), mu2l = v2_make(mua.lo, mub.lo);
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const float2 va = pa[q * 256], vb = pb[q * 256];
                pf[q] = c2{v2_sub(v2_sub(v2_make(va.x, vb.x), mu2), mu2l), v2_sub(v2_sub(v2_make(va.y, vb.y), mu2), mu2l)};
            });
        } else if (veca && vecb && ns >= 4) {
            // the last block of a row (a quarter of all blocks for 12 000-sample rows): the samples still inside the row go out
            // as 8-byte loads together (clamped addresses), the ones beyond it are fetched one by one afterwards
            const v2f mu2 = v2_make(mua.hi, mub.hi), mu2l = v2_make(mua.lo, mub.lo);
            const int par = k0 & 1;
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int i = k0 + 2 * (tid + q * 256), ic = par + (min(max(i - par, 0), ns - 2 - par) & ~1);
                const float2 va = *reinterpret_cast<const float2*>(xa + ic);
                const float2 vb = *reinterpret_cast<const float2*>(xb + ic);
                pf[q] = c2{v2_sub(v2_sub(v2_make(va.x, vb.x), mu2), mu2l), v2_sub(v2_sub(v2_make(va.y, vb.y), mu2), mu2l)};
            });
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int i = k0 + 2 * (tid + q * 256);
                if (i + 1 >= ns) pf[q] = c2_make(fetch(xa, xna, mua, veca, i), fetch(xb, xnb, mub, vecb, i));
            });
        } else {
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int i = k0 + 2 * (tid + q * 256);
                pf[q] = c2_make(fetch(xa, xna, mua, veca, i), fetch(xb, xnb, mub, vecb, i));
            });
        }
        static_for<7>([&](auto qq) {
            constexpr int q = decltype(qq)::value + 1;
            pw[q] = T.tw1[q * 256 + tid];
        });
    }
    __syncthreads();                                              // LDS twiddle tables visible
    if (fwd) {
        dftp<8>(pf);
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            xf_st(buf + x4_ad(tid + q * 256), (q == 0) ? pf[0] : c2_mulw(pf[q], pw[q]));
        });
    }
    // the middle stage's table operands: issued here, in flight across S2 / S3
    const int2 grp = T.pairs[tid];
    const int Gi = grp.x, PG = grp.y;
    const bool selfitem = (tid == kX4Items - 1);
    float2 GA[4], GB[4];
    {
        const float2* gpa = T.gp + (size_t)tsel * MB + Gi * 4;
        const float2* gpb = T.gp + (size_t)tsel * MB + PG * 4;
        static_for<4>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            GA[d] = gpa[d];
            GB[d] = gpb[d];
        });
    }
    const float gny = T.gn[tsel];
    const float2 wa0 = T.wg[Gi], wb0 = T.wg[PG];
    lds_barrier();
    // ---------------- S2: radix 8 inside every block of 256, x W_256^(j q)
    if (fwd) {
        const int e0 = (tid >> 5) * 256 + (tid & 31), j = tid & 31;
        c2 v[8];
        static_for<8>([&](auto qq) { constexpr int q = decltype(qq)::value; v[q] = xf_ld(buf + x4_ad(e0 + q * 32)); });
        dftp<8>(v);
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            xf_st(buf + x4_ad(e0 + q * 32), (q == 0) ? v[0] : c2_mulw(v[q], tw2[q * 32 + j]));
        });
    }
    lds_barrier();
    // ---------------- S3: radix 8 inside every block of 32, x W_32^(j q)
    if (fwd) {
        const int e0 = (tid >> 2) * 32 + (tid & 3), j = tid & 3;
        c2 v[8];
        static_for<8>([&](auto qq) { constexpr int q = decltype(qq)::value; v[q] = xf_ld(buf + x4_ad(e0 + q * 4)); });
        dftp<8>(v);
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            xf_st(buf + x4_ad(e0 + q * 4), (q == 0) ? v[0] : c2_mulw(v[q], tw3[q * 4 + j]));
        });
    }
    lds_barrier();
    // ---------------- MID: radix 4 on a group and its Hermitian partner group, pair op, inverse radix 4
    c2 a[4], b[4];
    static_for<4>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        a[d] = xf_ld(buf + x4_ad(Gi * 4 + d));
        b[d] = xf_ld(buf + x4_ad(PG * 4 + d));
    });
    dftp<4>(a);
    dftp<4>(b);
    lds_barrier();                                                // every read of the spectrum precedes the in-place writes
    {
        c2 ra[4], rb[4];
        if (!selfitem) {
            static_for<4>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = 3 - d;
                xf_pair(a[d], b[pn], rot_const<d, 8>(wa0), GA[d], GB[pn], ra[d], rb[pn]);
            });
        } else {
            // group 0 (array a): partner digit (4 - d) % 4, f = 0 pairs with the Nyquist bin; group 4 (array b): 3 - d
            static_for<3>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pz = (4 - d) % 4;
                const float2 gm = (d == 0) ? make_float2(gny, 0.f) : GA[pz];
                c2 na;
                xf_pair(a[d], a[pz], rot_const<d, 8>(wa0), GA[d], gm, na, ra[pz]);
                if constexpr (pz != d) ra[d] = na;
            });
            static_for<2>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = 3 - d;
                xf_pair(b[d], b[pn], rot_const<d, 8>(wb0), GB[d], GB[pn], rb[d], rb[pn]);
            });
        }
        idftp<4>(ra);
        idftp<4>(rb);
        static_for<4>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            xf_st(mine + x4_ad(Gi * 4 + d), ra[d]);
            xf_st(mine + x4_ad(PG * 4 + d), rb[d]);
        });
    }
    lds_barrier();
    float2 pwi[8];
    static_for<7>([&](auto qq) {
        constexpr int q = decltype(qq)::value + 1;
        pwi[q] = T.tw1[q * 256 + tid];                           // for S1', in flight across S3' / S2'
    });
    // ---------------- S3', S2': inverse radix 8
    {
        const int e0 = (tid >> 2) * 32 + (tid & 3), j = tid & 3;
        c2 v[8];
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const c2 xv = xf_ld(mine + x4_ad(e0 + q * 4));
            v[q] = (q == 0) ? xv : c2_mulwc(xv, tw3[q * 4 + j]);
        });
        idftp<8>(v);
        static_for<8>([&](auto qq) { constexpr int q = decltype(qq)::value; xf_st(mine + x4_ad(e0 + q * 4), v[q]); });
    }
    lds_barrier();
    {
        const int e0 = (tid >> 5) * 256 + (tid & 31), j = tid & 31;
        c2 v[8];
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const c2 xv = xf_ld(mine + x4_ad(e0 + q * 32));
            v[q] = (q == 0) ? xv : c2_mulwc(xv, tw2[q * 32 + j]);
        });
        idftp<8>(v);
        static_for<8>([&](auto qq) { constexpr int q = decltype(qq)::value; xf_st(mine + x4_ad(e0 + q * 32), v[q]); });
    }
    lds_barrier();
    // ---------------- S1': inverse radix 8 -> lags k0 + 2m, k0 + 2m + 1 (m = tid + 256 q), the first kXfStep of them
    {
        c2 v[8];
        static_for<8>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const c2 xv = xf_ld(mine + x4_ad(tid + q * 256));
            v[q] = (q == 0) ? xv : c2_mulwc(xv, pwi[q]);
        });
        idftp<8>(v);
        float* ya = (tsel == 0 ? y0 : y1) + (size_t)rowA * ns;
        float* yb = (tsel == 0 ? y0 : y1) + (size_t)rowB * ns;
        if (interior) {
            float2* oa = reinterpret_cast<float2*>(ya + k0) + tid;
            float2* ob = reinterpret_cast<float2*>(yb + k0) + tid;
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                if (2 * (tid + q * 256) < kXfStep) {
                    const c2 o = c2_scale2(v[q], sc);
                    st_stream(oa + q * 256, c2_a(o));
                    if (hasB) st_stream(ob + q * 256, c2_b(o));
                }
            });
        } else {
            auto put = [&](float* yr, bool vec, int k, float2 o) {
                if (k >= ns) return;
                if (vec && k + 1 < ns) *reinterpret_cast<float2*>(yr + k) = o;
                else {
                    yr[k] = o.x;
                    if (k + 1 < ns) yr[k + 1] = o.y;
                }
            };
            static_for<8>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int m = tid + q * 256;
                if (2 * m < kXfStep) {
                    const c2 o = c2_scale2(v[q], sc);
                    put(ya, veca, k0 + 2 * m, c2_a(o));
                    if (hasB) put(yb, vecb, k0 + 2 * m, c2_b(o));
                }
            });
        }
    }
}

// extra tables of the four-stage kernel: gp [2][MB] + tw1 [8][256] + tw2 [8][32] + tw3 [8][4] + wg [512] (float2), pairs [256] (int2)
constexpr size_t kX4WsFloats = 2 * (2 * kXfMB + 8 * 256 + 8 * 32 + 8 * 4 + kX4NG + kX4Items);

constexpr size_t kXfWsFloats = 2 * 2 * kXfMB + 8 + 2 * kXfM1 * 2 + 2 * kXfNG + 2 * kXfNA * kXfM1;

}  // namespace d4w

using namespace d4w;

// ---- the cross-stream fence between this file's kernels and stft_mm_rows (d4w_internal.h: hazard_enter / hazard_leave)
#include <mutex>
namespace d4w {
namespace {
constexpr int kHzMaxDev = 64;
std::mutex g_hz_mu;
hipEvent_t g_hz_ev[kHzMaxDev][2];   // [device][family]: recorded behind the family's last launch
bool g_hz_have[kHzMaxDev][2];
// Off by default since round 6: the overlap-save kernels claim their compute unit's LDS (xf_lds_claim), so the matrix-core STFT
// cannot be resident beside them whatever stream it comes from (40 trials per pair without the fence: profiles/r06h).
// D4W_HAZARD_FENCE=1 puts the cross-stream serialisation of round 5 back.
bool hazard_on() {
    static const int on = [] { const char* v = getenv("D4W_HAZARD_FENCE"); return v ? atoi(v) : 0; }();
    return on != 0;
}
}
// ONE critical section from the wait to the record: hazard_enter returns with the mutex held (unless it fails), hazard_leave
// records the family's event behind the launches and releases it.  Two host threads can therefore not both pass the wait
// before either has recorded -- the gap the round-5 form had: thread A waits on B's (not yet recorded) event,
// thread B waits on A's, both launch, the two families run side by side.
int hazard_enter(int self, void* stream) {
    if (!hazard_on()) return D4W_OK;
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    if (devid < 0 || devid >= kHzMaxDev)
        return fail(D4W_EINVAL, "device %d is beyond the %d devices the cross-stream fence keeps events for", devid, kHzMaxDev);
    g_hz_mu.lock();
    if (g_hz_have[devid][1 - self]) {
        const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, g_hz_ev[devid][1 - self], 0);
        if (e != hipSuccess) {
            g_hz_mu.unlock();
            return fail(D4W_EHIP, "hipStreamWaitEvent failed: %s (cross-stream fence)", hipGetErrorString(e));
        }
    }
    return D4W_OK;
}
int hazard_leave(int self, void* stream) {
    if (!hazard_on()) return D4W_OK;
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e == hipSuccess && !g_hz_have[devid][self]) {
        e = hipEventCreateWithFlags(&g_hz_ev[devid][self], hipEventDisableTiming);
        g_hz_have[devid][self] = (e == hipSuccess);
    }
    if (e == hipSuccess) e = hipEventRecord(g_hz_ev[devid][self], (hipStream_t)stream);
    g_hz_mu.unlock();
    if (e != hipSuccess) return fail(D4W_EHIP, "cross-stream fence: %s", hipGetErrorString(e));
    return D4W_OK;
}
}  // namespace d4w

// Every workgroup of the overlap-save kernels asks for 159 KiB of LDS whatever it uses: no other workgroup that needs LDS fits on
// its compute unit (see xcorr_fft_blocks: SUBS).  The one-template kernel fills the claim with four blocks' worth of threads;
// the fused two-template forms (method = "fft", an A/B reference since round 4) run one workgroup per CU instead of two for it.
// D4W_XF_LDS_CLAIM=n: n KiB instead (0: only what the kernel uses -- the A/B that shows the hazard).
static int mm_num_cus_xf() {
#ifdef D4W_EMU
    return 2;
#else
    static int cached[64] = {0};
    int devid = 0, v = 0;
    if (hipGetDevice(&devid) != hipSuccess) return 256;
    if (devid >= 0 && devid < 64 && cached[devid] > 0) return cached[devid];
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, devid) == hipSuccess && v > 0) {
        if (devid >= 0 && devid < 64) cached[devid] = v;
        return v;
    }
    return 256;
#endif
}

// grid of the SUBS-block kernel for `g` = (blocks per row, row pairs): a workgroup walks the groups of its row pair -- all of
// them when the row pairs alone fill the chip twice over, so that a workgroup lives for a whole row (one workgroup per CU: nothing
// covers the next one's start-up), else as many walkers per row pair as it takes
static dim3 xf_sub_grid(dim3 g) {
    const int ngrp = ceil_div((int)g.x, kXfSubs), ncu = mm_num_cus_xf();
#if defined(D4W_XF_LOOP) && D4W_XF_LOOP
    const int per_row = std::max(1, std::min(ngrp, ceil_div(2 * ncu, (int)g.y)));
#else
    const int per_row = ngrp;
    (void)ncu;
#endif
    return dim3(per_row, g.y);
}

static size_t xf_lds_claim(size_t need) {
    static const int kib = [] { const char* v = getenv("D4W_XF_LDS_CLAIM"); return v ? atoi(v) : 159; }();
    return kib > 0 ? std::max(need, (size_t)kib * 1024) : need;
}

extern "C" {

int d4w_xcorr_fft_max_support(void) { return kXfPad + 1; }

size_t d4w_xcorr_fft_ws_bytes(void) { return (kXfWsFloats + kX4WsFloats) * sizeof(float); }

int d4w_xcorr_fft_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs, const float* taps,
                      int ntpl, int ltaps, int len0, int len1, float* y0, float* y1, void* ws, void* stream) {
    return d4w_xcorr_fft_cont_f32(x, nx, ns, nullptr, 0, 0, mean, maxabs, taps, ntpl, ltaps, len0, len1, y0, y1, ws, stream);
}

static int d4w_xcorr_fft_cont_f32_run(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next, const double* mean,
                           const float* maxabs, const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0,
                           float* y1, void* ws, void* stream) {
    if (!x || !y0 || !ws || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (xnext && (ntpl != 2 || n_next < 0 || ld_next < n_next))
        return fail(D4W_EINVAL, "a continuation needs two templates (the fused kernel) and 0 <= n_next <= ld_next");
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
    float2* twa = wg + kXfNG;
    T.gp = gp; T.gn = gn; T.tw1 = tw1; T.tw2 = tw2; T.wg = wg; T.twa = twa;
    // taps == NULL: ws still holds the tables a previous call built for the same templates (same ntpl, supports and kernel
    // selection) -- a stream of files pays the template spectra once
    if (taps)
        D4W_LAUNCH(xcf_spectra, dim3(ceil_div(std::max(ntpl * kXfMB, kXfNA * kXfM1), 256)), dim3(256), 0, stream, taps, ntpl,
                   ltaps, len0, len1, gp, gn, tw1, tw2, wg, twa);
    const dim3 grid(ceil_div(ns, kXfStep), ceil_div(nx, 2));
    const size_t lds = xf_lds_claim((size_t)kXfRowP * sizeof(float4) + 2 * kXfM1 * sizeof(float2));
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)xcorr_fft_blocks<1, false, false, kXfSubs>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)xcorr_fft_blocks<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    // D4W_XF_TPAIR=1: both templates off ONE read of x (one row per workgroup, the two correlations
    // packed through one inverse transform).  Measured at 20000 x 120000: 8.96 ms against 8.93 ms for the
    // two row-pair launches -- both forms are bound by VALU issue (~2000 packed instructions per row
    // block), not by the 12 vs 16 bytes per sample they move -- so the default stays with the kernel
    // that has the simpler forward stage.
    static const int pairmode = [] { const char* v = getenv("D4W_XF_TPAIR"); return v ? atoi(v) : 0; }();
    if (ntpl == 2 && pairmode && nx <= 65535) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)xcorr_fft_tpair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr2 = true;
        }
        D4W_LAUNCH(xcorr_fft_tpair, dim3(grid.x, nx), dim3(kXfThreads), lds, stream, T, x, ns, mean, maxabs, y0, y1);
        return D4W_OK;
    }
    // both templates off one read and one forward transform of every block
    // (D4W_XF_FUSED: 1 = four-stage 512-thread kernel [default], 2 = three-stage 256-thread kernel, 0 = one launch per template)
    static const int fusedmode = [] { const char* v = getenv("D4W_XF_FUSED"); return v ? atoi(v) : 1; }();
    if (ntpl == 2 && fusedmode == 1) {                            // four-stage, 512-thread kernel (default)
        float2* f4 = reinterpret_cast<float2*>(w + kXfWsFloats);
        X4Tables Q;
        float2* gp4 = f4;
        float2* q1 = gp4 + 2 * kXfMB;
        float2* q2 = q1 + 8 * 256;
        float2* q3 = q2 + 8 * 32;
        float2* qg = q3 + 8 * 4;
        int2* qp = reinterpret_cast<int2*>(qg + kX4NG);
        Q.gp = gp4; Q.gn = gn; Q.tw1 = q1; Q.tw2 = q2; Q.tw3 = q3; Q.wg = qg; Q.pairs = qp;
        if (taps)
            D4W_LAUNCH(xcf_tables4, dim3(ceil_div(2 * kXfMB, 256)), dim3(256), 0, stream, taps, ltaps, len0, len1, gp4, gn, q1, q2,
                       q3, qg, qp);
        const size_t lds4 = xf_lds_claim(2 * (size_t)kX4RowP * sizeof(float4) + (256 + 32) * sizeof(float2));
        static bool attr4 = false;
        if (!attr4) {
            (void)hipFuncSetAttribute((const void*)xcorr_fft_fused4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)xcorr_fft_fused4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr4 = true;
        }
        if (xnext)
            D4W_LAUNCH(xcorr_fft_fused4<true>, grid, dim3(2 * kX4Items), lds4, stream, Q, x, nx, ns, mean, maxabs, y0, y1, xnext,
                       ld_next, n_next);
        else
            D4W_LAUNCH(xcorr_fft_fused4<false>, grid, dim3(2 * kX4Items), lds4, stream, Q, x, nx, ns, mean, maxabs, y0, y1,
                       (const float*)nullptr, 0, 0);
        return D4W_OK;
    }
    if (xnext) return fail(D4W_EINVAL, "a continuation runs the four-stage fused kernel only (D4W_XF_FUSED / D4W_XF_TPAIR are set)");
    if (ntpl == 2 && fusedmode) {
        const size_t lds2 = xf_lds_claim(2 * (size_t)kXfRowP * sizeof(float4) + 2 * kXfM1 * sizeof(float2));
        D4W_LAUNCH((xcorr_fft_blocks<1, true>), grid, dim3(2 * kXfThreads), lds2, stream, T, x, nx, ns, mean, maxabs, y0, y1,
                   kXfStep, 0, ns, 0.f, XfHalo{}, (const float*)nullptr, 0, (int)grid.x);
        return D4W_OK;
    }
    for (int t = 0; t < ntpl; ++t) {
        XfTables Tt = T;
        Tt.gp = gp + (size_t)t * kXfMB;
        Tt.gn = gn + t;
        D4W_LAUNCH((xcorr_fft_blocks<1, false, false, kXfSubs>), xf_sub_grid(grid), dim3(kXfSubs * kXfThreads),
                   xf_lds_claim(kXfSubs * kXfItemLds + 16), stream, Tt, x, nx, ns, mean, maxabs,
                   t == 0 ? y0 : y1, (float*)nullptr, kXfStep, 0, ns, 0.f, XfHalo{}, (const float*)nullptr, 0, ceil_div((int)grid.x, kXfSubs));
    }
    return D4W_OK;
}

int d4w_xcorr_fft_cont_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next, const double* mean,
                           const float* maxabs, const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0,
                           float* y1, void* ws, void* stream) {
    int rc = hazard_enter(0, stream);      // (never beside stft_mm_rows: d4w_internal.h)
    if (rc) return rc;
    rc = d4w_xcorr_fft_cont_f32_run(x, nx, ns, xnext, ld_next, n_next, mean, maxabs, taps, ntpl, ltaps, len0, len1, y0, y1, ws, stream);
    const int rl = hazard_leave(0, stream);
    return rc ? rc : rl;
}

/* Zero-phase FIR along time by overlap-save FFT blocks (the interior of a zero-phase IIR filter: its two-sided
 * response truncated where it has decayed): y[r][n] = sum_j taps[j] x[r][n - K + j], j < 2K + 1, for K <= n < ns - K.
 * The columns within K of either row end are NOT written.  first[r] (a constant per row, e.g. the row's first sample)
 * is subtracted from the samples before the transform and first[r] * dc_gain added back (dc_gain = sum of the taps
 * as the exact filter has it), which keeps a large offset out of the float32 transform. */
int d4w_fir_fft_max_halfwidth(void) { return (kXfB - 2048) / 2; }

int d4w_fir_fft_f32(const float* x, int nx, int ns, const float* taps, int K, const float* first, double dc_gain,
                    float* y, void* ws, void* stream) {
    return d4w_fir_fft_cols_f32(x, nx, ns, taps, K, first, dc_gain, y, K, ns - K, ws, stream);
}

static int d4w_fir_fft_cols_f32_run(const float* x0, int nx, int ns0, const float* taps, int K, const float* first, double dc_gain,
                         float* y0, int col0, int col1, void* ws, void* stream) {
    if (!x0 || !y0 || !ws || nx < 1 || ns0 < 1) return fail(D4W_EINVAL, "bad argument");
    if (K < 0 || (K & 1) || K > d4w_fir_fft_max_halfwidth()) return fail(D4W_EINVAL, "half width %d must be even and <= %d", K, d4w_fir_fft_max_halfwidth());
    if (ns0 <= 2 * K) return fail(D4W_EINVAL, "rows of %d samples have no interior for a half width of %d", ns0, K);
    if (col0 < K || col1 > ns0 - K || col0 >= col1) return fail(D4W_EINVAL, "output columns [%d, %d) must lie inside the interior [%d, %d)", col0, col1, K, ns0 - K);
    if (nx > 2 * 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 131070", nx);
    // the window [col0, col1) of the output = the plain interior filter of the rows x[r][col0 - K .. ns0) (pitch ns0)
    const int sft = col0 - K, ld = ns0, ns = ns0 - sft, ns_out = col1 - col0;
    const float* x = x0 + sft;
    float* y = y0 + sft;
    float* w = (float*)ws;
    XfTables T;
    float2* gp = (float2*)w;
    float* gn = w + 2 * 2 * kXfMB;
    float2* tw1 = (float2*)(gn + 8);
    float2* tw2 = tw1 + kXfM1;
    float2* wg = tw2 + kXfM1;
    float2* twa = wg + kXfNG;
    T.gp = gp; T.gn = gn; T.tw1 = tw1; T.tw2 = tw2; T.wg = wg; T.twa = twa;
    const int L = 2 * K + 1;
    if (taps)                                                    // NULL: ws holds the tables of a previous call with these taps
        D4W_LAUNCH(xcf_spectra, dim3(ceil_div(std::max(kXfMB, kXfNA * kXfM1), 256)), dim3(256), 0, stream, taps, 1, L, L, L, gp, gn,
                   tw1, tw2, wg, twa);
    const int step = kXfB - 2 * K;
    const dim3 grid(ceil_div(ns_out, step), ceil_div(nx, 2));
    const size_t lds = xf_lds_claim((size_t)kXfRowP * sizeof(float4) + 2 * kXfM1 * sizeof(float2));
    (void)hipFuncSetAttribute((const void*)xcorr_fft_blocks<1, false, false, kXfSubs>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    D4W_LAUNCH((xcorr_fft_blocks<1, false, false, kXfSubs>), xf_sub_grid(grid), dim3(kXfSubs * kXfThreads),
               xf_lds_claim(kXfSubs * kXfItemLds + 16), stream, T, x, nx, ns, (const double*)nullptr, (const float*)nullptr, y,
               (float*)nullptr, step, K, ns_out, (float)dc_gain, XfHalo{}, first, ld, ceil_div((int)grid.x, kXfSubs));
    return D4W_OK;
}

int d4w_fir_fft_cols_f32(const float* x0, int nx, int ns0, const float* taps, int K, const float* first, double dc_gain,
                         float* y0, int col0, int col1, void* ws, void* stream) {
    int rc = hazard_enter(0, stream);      // (never beside stft_mm_rows: d4w_internal.h)
    if (rc) return rc;
    rc = d4w_fir_fft_cols_f32_run(x0, nx, ns0, taps, K, first, dc_gain, y0, col0, col1, ws, stream);
    const int rl = hazard_leave(0, stream);
    return rc ? rc : rl;
}

static int d4w_fir_fft_halo_f32_run(const float* x, int nx, int ns, const float* left, int ld_left, int n_left, const float* right,
                         int ld_right, int n_right, const float* taps, int K, const float* first, double dc_gain, float* y,
                         void* ws, void* stream) {
    if (!x || !y || !ws || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (K < 0 || (K & 1) || K > d4w_fir_fft_max_halfwidth()) return fail(D4W_EINVAL, "half width %d must be even and <= %d", K, d4w_fir_fft_max_halfwidth());
    if ((left && (n_left < K || ld_left < n_left)) || (right && (n_right < K || ld_right < n_right)) || (!left && n_left) || (!right && n_right))
        return fail(D4W_EINVAL, "a halo holds at least the half width (%d) samples per row (n_left = %d, n_right = %d)", K, n_left, n_right);
    if (!left || !right) return fail(D4W_EINVAL, "both halos are needed: a row end without a neighbour is d4w_sosfiltfilt_f32's edge rule");
    if (nx > 2 * 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 131070", nx);
    float* w = (float*)ws;
    XfTables T;
    float2* gp = (float2*)w;
    float* gn = w + 2 * 2 * kXfMB;
    float2* tw1 = (float2*)(gn + 8);
    float2* tw2 = tw1 + kXfM1;
    float2* wg = tw2 + kXfM1;
    float2* twa = wg + kXfNG;
    T.gp = gp; T.gn = gn; T.tw1 = tw1; T.tw2 = tw2; T.wg = wg; T.twa = twa;
    const int L = 2 * K + 1;
    if (taps)                                                    // NULL: ws holds the tables of a previous call with these taps
        D4W_LAUNCH(xcf_spectra, dim3(ceil_div(std::max(kXfMB, kXfNA * kXfM1), 256)), dim3(256), 0, stream, taps, 1, L, L, L, gp, gn,
                   tw1, tw2, wg, twa);
    const int step = kXfB - 2 * K;
    const dim3 grid(ceil_div(ns, step), ceil_div(nx, 2));
    const size_t lds = xf_lds_claim((size_t)kXfRowP * sizeof(float4) + 2 * kXfM1 * sizeof(float2));
    (void)hipFuncSetAttribute((const void*)xcorr_fft_blocks<1, false, true, kXfSubs>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // lag k of the virtual row starting K samples before the row = output sample k of the row
    XfHalo H{left, right, ld_left, n_left, ld_right, n_right, n_left - K};
    D4W_LAUNCH((xcorr_fft_blocks<1, false, true, kXfSubs>), xf_sub_grid(grid), dim3(kXfSubs * kXfThreads),
               xf_lds_claim(kXfSubs * kXfItemLds + 16), stream, T, x, nx, ns, (const double*)nullptr, (const float*)nullptr, y,
               (float*)nullptr, step, 0, ns, (float)dc_gain, H, first, 0, ceil_div((int)grid.x, kXfSubs));
    return D4W_OK;
}

int d4w_fir_fft_halo_f32(const float* x, int nx, int ns, const float* left, int ld_left, int n_left, const float* right,
                         int ld_right, int n_right, const float* taps, int K, const float* first, double dc_gain, float* y,
                         void* ws, void* stream) {
    int rc = hazard_enter(0, stream);      // (never beside stft_mm_rows: d4w_internal.h)
    if (rc) return rc;
    rc = d4w_fir_fft_halo_f32_run(x, nx, ns, left, ld_left, n_left, right, ld_right, n_right, taps, K, first, dc_gain, y, ws, stream);
    const int rl = hazard_leave(0, stream);
    return rc ? rc : rl;
}

}  // extern "C"
