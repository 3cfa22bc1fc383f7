// Matched filter as a banded-Toeplitz product on the MI355X matrix cores (gfx950): replaces the per-row
// scipy.signal.correlate(x, template, 'full', 'fft') of detect.compute_cross_correlogram (reference
// detect.py:140-166; positive lags, detect.py:111-112) for templates of short support (the fin-whale call
// templates have 136 / 156 non-zero samples).
//
// Why matrix cores.  The correlation costs 2 (L0 + L1) = 584 flop per sample in its direct form -- 8.9 ms of the
// packed-FMA rate for a 20 000 x 120 000 block, which is why rounds 1-3 ran it as an overlap-save FFT
// (xcorr_fft.hip, ~90 flop per sample).  That kernel is bound by neither HBM nor VALU issue but by the chain of
// LDS round trips and barriers of three 2048-point transforms per row block (46 % of the HBM roofline for three
// rounds).  The direct form IS a matrix product with a banded Toeplitz factor,
//
//      y[16 a + i] = sum_u  t[u - i] * x[16 a + u],      i < 16,  u < 16 + L - 1,
//
// i.e. C[i][a] = sum_u A[i][u] B[u][a] with A[i][u] = t[u - i] (16 x K, K = 32 ceil((L + 15) / 32), the template's
// Toeplitz matrix, zero outside the band, resident in registers for the whole launch) and B[u][a] = x[16 a + u]
// (overlapping windows of the row, read straight out of an LDS copy of the row chunk: lane (a, g) takes the 8
// consecutive samples 16 a + 32 kk + 8 g .. + 7 as ONE 16-byte LDS read).  v_mfma_f32_16x16x32_f16 does 16 384
// flop per instruction; 11 k-steps x 3 products per 256 lags leave the matrix pipe ~35 % busy at the HBM rate,
// the vector ALUs only convert and store, and the kernel is a plain stream: 4 B read + 8 B written per sample.
//
// float32 through binary16 factors.  Every operand is split into two binary16 values, v = hi + lo 2^-11
// (hi = rn16(v), lo = rn16((v - hi) 2^11): 22-23 significant bits), and a product keeps three of the four partial
// products, hi hi + (hi lo + lo hi) 2^-11 -- each exact in the float32 accumulator of the matrix instruction; the
// dropped lo lo 2^-22 term is below float32 rounding.  Rows are scaled to |x| <= 2 before the split (1 / max|x| of
// detect.py:157 when the caller normalises, otherwise a power of two per chunk), templates by a power of two, so
// nothing overflows binary16; small values use its subnormals, which the conversions and the matrix instruction keep.  Measured against a float64 correlation the result is as close as the float32
// FFT kernel's (tests/test_rowops_gpu.py).
//
// Launch shape.  Persistent workgroups (256 threads, 4 waves) walk chunks of 4096 lags of one row; the chunk's
// 4096 + 192 samples are loaded one chunk AHEAD into registers (17 floats per lane), converted and written to
// one of two LDS buffers (hi / lo arrays in sample order: the four lane groups of a fragment read hit 16 different
// 16-byte slots each), ONE barrier per chunk, then every wave runs four 16 x 16 tiles (256 lags each,
// both templates) and streams the results out with 16-byte non-temporal stores (lane (a, g) holds lags
// 16 a + 4 g .. + 3: 1 KiB contiguous per wave and template).  Chunks are dealt to the XCDs in contiguous
// ranges, so the 192-sample halo of a chunk is an L2 hit.
#include <cstdlib>

#include "mm_common.h"

namespace d4w {

#ifndef D4W_MM_CH
#define D4W_MM_CH 4096
#endif
constexpr int kMmCH = D4W_MM_CH;                 // lags per chunk (8192 measured: see DESIGN 3.3)
constexpr int kMmKS = 6;                         // k-steps of 32 of the two-template kernels -> Toeplitz depth 192, supports <= 177
constexpr int kMmKSLong = 8;                     // ... of the one-template kernel for longer supports: depth 256, supports <= 241
constexpr int kMmKSMax = 16;                     // deepest one-template kernel: depth 512, a SECTION of <= 497 taps (128 VGPRs of fragments)
constexpr int kMmSection = 32 * kMmKSMax - 16;   // taps per section of a longer template (a multiple of 16: shifted loads stay 16-byte aligned)
constexpr int kMmMaxSupport = 32 * kMmKSLong - 15;
constexpr int kMmMaxSections = 16;               // templates of up to 16 x 496 taps run section by section (one accumulate launch each)
constexpr int kMmThreads = 256;
// geometry of a chunk for a Toeplitz depth of KSM k-steps
template <int KSM>
struct MmGeom {
    static constexpr int Halo = 32 * KSM;                        // samples staged beyond the chunk
    static constexpr int Stage = kMmCH + Halo;                   // samples per chunk (4288 at KSM = 6)
    static constexpr int Q = (Stage + 4 * kMmThreads - 1) / (4 * kMmThreads);      // 16-byte loads per lane: 5
    static constexpr int LastQ = (Stage - (Q - 1) * 4 * kMmThreads) / 4;         // lanes that take the last load (48 at KSM = 6)
    static constexpr int Arr = Stage + 8;                        // halves per LDS array
};
// LDS index of sample h of the chunk: the plain order.  ds_read_b128 is served in four NON-contiguous 16-lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS), and with lane (a, g) reading the 16-byte
// slot 2 a + g (+ const) every group touches 16 different slots: no bank conflict.  (Round 4's first build padded 16 B
// per 256 B for contiguous groups: 61 % of its LDS cycles were conflicts, profiles/r04a/pmc_sq_matched_filter.txt.)
__host__ __device__ constexpr int mm_pidx(int h) { return h; }

struct MmArgs {
    const float* x;         // [nx][ns]
    const float* xnext;     // [nx][ld_next] or NULL: the record's continuation (first n_next samples of every row)
    const double* mean;     // [nx] float64 row means or NULL (consumed as a two-float value, d4w_internal.h Mean2)
    const float* maxabs;    // [nx] or NULL: the 1 / max|x| of the normalisation (output scale only)
    const float* taps;      // [ntpl][ltaps]
    float* y0;
    float* y1;
    int nx, ns, ld_next, n_next, ltaps, len0, len1;
    float* rowmax0;         // [nx] or NULL: max over the lags of every row of y0 (y1), formed in the epilogue (float bits,
    float* rowmax1;         //   initialised to -inf by the host; combined with integer atomics, mm_atomic_fmax)
    int clamp;              // scaled samples are clamped into binary16's finite range before the split (continuations, D4W_MM_CLAMP=1)
    int shift;              // the taps given are taps [shift, shift + len0) of a longer template: lag k reads x[k + shift + n]
    int accumulate;         // add to y0 instead of overwriting it (the later sections of a long template)
    // TAIL kernels (the zero-padded template's constant tail, detect.py:158, added in the epilogue -- see xcorr_mm_rows):
    float tail0, tail1;             // mean(t) / max|t| of template 0 / 1 over its zero-padded length (0: nothing to add)
};

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside the 16-lane DPP rows, then row_bcast:15 /
// row_bcast:31 carry the row totals on (six v_add_f32 with DPP operands, no LDS traffic)
__device__ __forceinline__ float mm_wave_scan(float v) {
#ifdef D4W_EMU
    const int lane = (int)(threadIdx.x & 63);
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(v, (unsigned)off);
        if (lane >= off) v += n;
    }
    return v;
#else
    auto dpp = [](float x, auto ctrl, auto rmask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});      // row_shr:1
    v += dpp(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});      // row_shr:2
    v += dpp(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});      // row_shr:4
    v += dpp(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});      // row_shr:8
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});      // row_bcast:15 -> rows 1, 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});      // row_bcast:31 -> rows 2, 3
    return v;
#endif
}

// KS0 / KS1: k-steps of template 0 / 1 (KS1 = 0: one template); WPS: workgroups per compute unit the registers are budgeted for.
//
// TAIL: the kernel also adds the constant tail of the de-meaned ZERO-PADDED template (detect.py:158 normalises the template over
// its padded length, which leaves -mean(t) / max|t| on the padding): lag k receives  tail_t * P[k + L_t],  P[j] = sum_{i < j} xh[i]
// the prefix sum of the normalised row (P[j >= ns] = the row's sum = 0).  Rounds 1-5 added the term in a second pass over x and y,
// decided per row on prefix maxima from a third sweep (d4w_row_stats_prefix_f32, d4w_xcorr_dc_tail_rows_f32): the public call
// cost 1.5 x the kernel.  Here it is exact for every row at no extra pass, in two pieces:
//   * INSIDE a block of 16 lags the term is itself a Toeplitz product: P[16 a + i + L] - P[16 a] = sum_{u < L + i} xh[16 a + u],
//     i.e. the taps t[d] + tail for -15 <= d < L in place of t[d] (d = u - i; zero beyond) -- the matrix instructions that form
//     the correlation form it with it, for nothing (the staged taps get the constant added, that is all);
//   * what remains is ONE number per block of 16 lags, tail * P[16 a]: the conversion phase sums each lane's four samples, a
//     wave prefix scan (DPP) gives the prefix at every fourth lane = every block of 16 samples, one float per block goes to
//     LDS (1 KiB per chunk), and a tile's epilogue adds tail * (row prefix at the chunk + segment offset + block prefix) to its
//     lags.  A wave's tiles are the segments it converted itself: only the 16 segment totals cross waves;
//   * the prefix at the chunk's START is carried in a register (float64): a workgroup of a TAIL kernel walks the chunks of ONE
//     ROW in order and takes whole rows (row = workgroup + k x grid) instead of chunks dealt over the grid.  (The first two
//     builds kept the chunk dealing and handed every chunk's sum to the workgroups holding the row's later chunks -- 8-byte
//     {tag, value} granules, ticketed chunks: correct, and 3.4 x slower than the kernel without the term, 21.9 against 6.4 ms:
//     a granule takes several microseconds from one compute unit to another while the memory system is saturated, and every
//     chunk waited for one; the third build kept a full prefix array in LDS, 17 KiB per buffer: 8.1 against 6.1 ms;
//     profiles/r06b, r06c, r06d.)  The price of whole rows is the tail of the launch (20 000 rows over 512 workgroups: 39.06
//     rows each) and 512 streams a row apart instead of 8 compact windows.
// WMAX: the epilogue also leaves the rows' maxima (P.rowmax0 / rowmax1) -- a separate instantiation, so that the kernels without
// them carry neither the branches nor the registers (round 5 had it as a kernel-argument branch in every tile's epilogue).
// WSPLIT (round 6; measured slower, kept behind D4W_MM_FUSED=3): two templates with the WAVES split between them -- waves 0, 1 run template 0 and waves 2, 3 template 1, eight
// tiles each, instead of every wave running both templates on four tiles.  A wave then keeps ONE template's Toeplitz fragments
// (44 registers instead of 88): the kernel fits three workgroups per compute unit like the one-template kernel (163 against 201
// registers), and the third workgroup is worth 8.5 % to that kernel (profiles/r06l/mm_wgs.txt).  The matrix work per wave is the
// same (144 instructions per chunk); the sample fragments are read from LDS by two waves instead of one -- and that is what it
// loses on: 6.32 against 6.11 ms for the kernel where every wave runs both templates (profiles/r06m/mm_wave_split.txt).
// Instantiated as <KS, 0, 3, ., ., true>: the one-template code with the template, output, tail coefficient and row maxima
// chosen by the wave.
template <int KS0, int KS1, int WPS, bool TAIL = false, bool WMAX = false, bool WSPLIT = false>
__global__ __launch_bounds__(kMmThreads, WPS) void xcorr_mm_rows(MmArgs P) {
    static_assert(!(WSPLIT && KS1 > 0), "the wave-split kernel is the one-template code run by two wave pairs");
    constexpr int KSM = KS0 > KS1 ? KS0 : KS1;
    using GEO = MmGeom<KSM>;
    constexpr int kMmHalo = GEO::Halo, kMmStage = GEO::Stage, kMmQ = GEO::Q, kMmLastQ = GEO::LastQ, kMmArr = GEO::Arr;
    constexpr int kSeg = 4 * kMmQ;                                  // 256-thread segments of a stage (1024 samples each), <= 20
    D4W_DYN_LDS(smem_raw);
    mm_half* lds = reinterpret_cast<mm_half*>(smem_raw);           // [2 buffers][hi | lo][kMmArr]
    float* red = reinterpret_cast<float*>(lds + 4 * kMmArr);       // [2][4] chunk maxima of the waves
    constexpr int kBlk = kMmCH / 4;                                 // lanes x loads of a chunk's own samples (1024): one prefix each, every fourth one is a block of 16's
    float* pb = red + 8;                                            // TAIL: [2][kBlk] prefix before each lane's four samples, inside its wave's segment
    float* wt = pb + 2 * kBlk;                                      // TAIL: [2][16] the segments' totals (a segment = 256 samples = one wave's share of 1024)
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = mm_uniform(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int ns = P.ns;
    const int tsel = WSPLIT ? (wv >> 1) : 0;                        // WSPLIT: the template this wave runs (wave-uniform)

    // ---- the chunks of this workgroup: XCD j (workgroup id mod 8) owns the contiguous range [j T / 8, (j + 1) T / 8)
    const int nchunk = (ns + kMmCH - 1) / kMmCH;
    const long long total = (long long)P.nx * nchunk;
    const int nparts = min(8, (int)gridDim.x);
    const int xcd = (int)blockIdx.x % nparts, wq = (int)blockIdx.x / nparts, nq = ((int)gridDim.x - xcd + nparts - 1) / nparts;
    const long long lo_c = total * xcd / nparts, hi_c = total * (xcd + 1) / nparts;
    // ---- the templates' Toeplitz fragments: A_t[kk][i = n16][u = 32 kk + 8 g + j] = t[u - i] / ts_t, split hi / lo
    mm_h8 a0h[KS0], a0l[KS0];
    mm_h8 a1h[KS1 ? KS1 : 1], a1l[KS1 ? KS1 : 1];
    float osc0 = 1.f, osc1 = 1.f;                                   // output scales: the power of two taken out of the taps
    // TAIL: whole rows (row = workgroup + k x grid), their chunks in order; else chunk lo_c + wq + k nq of the XCD's range
#ifdef D4W_MM_V_TAIL_DEALT          // (probe builds: the tail kernels with the chunks dealt over the grid -- WRONG prefixes, timing only)
    constexpr bool kRows = false;
#else
    constexpr bool kRows = TAIL;
#endif
    auto next_chunk = [&](long long c) -> long long {
        if constexpr (kRows) {
            const long long r = c / nchunk;
            return (c - r * nchunk + 1 < nchunk) ? c + 1 : (r + (long long)gridDim.x) * nchunk;
        } else {
            return c + nq;
        }
    };
    const long long end_c = kRows ? total : hi_c;
    long long c_n = kRows ? (long long)blockIdx.x * nchunk : lo_c + wq;
    {
        // taps -> LDS first (zero outside the support), so that the 8 x KS fragment values of a lane are LDS reads
        float* tl = reinterpret_cast<float*>(smem_raw);              // [2][16 + kMmHalo] before the row buffers are in use
        constexpr int TLP = 16 + kMmHalo;
        for (int i = tid; i < 2 * TLP; i += kMmThreads) {
            const int t = i / TLP, u = i - t * TLP - 15;
            const int L = t ? P.len1 : P.len0;
            float tv = (u >= 0 && u < L && (t == 0 || KS1 > 0 || WSPLIT)) ? P.taps[(size_t)t * P.ltaps + u] : 0.f;
            if constexpr (TAIL) {                                   // the term inside a block of 16 lags: t[d] + tail for every d < L
                if (u < L && (t == 0 || KS1 > 0 || WSPLIT)) tv += t ? P.tail1 : P.tail0;
            }
            tl[i] = tv;
        }
        __syncthreads();
        auto build = [&](const float* tp, int L, auto& ah, auto& al, auto ks, float& osc) {
            constexpr int KS = decltype(ks)::value;
            float m = 0.f;
            for (int i = lane; i < TLP; i += 64) m = fmaxf(m, fabsf(tp[i]));
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float up, down;
            mm_pow2_scale(m, up, down);
            osc = up;
            (void)L;
            static_for<KS>([&](auto kq) {
                constexpr int kk = decltype(kq)::value;
                static_for<8>([&](auto jq) {
                    constexpr int j = decltype(jq)::value;
                    const float v = tp[15 + 32 * kk + 8 * g + j - n16] * down;
                    mm_half hi, lo;
                    mm_split(v, hi, lo);
                    mm_set(ah[kk], j, hi);
                    mm_set(al[kk], j, lo);
                });
            });
        };
        build(tl + (WSPLIT ? tsel * TLP : 0), WSPLIT && tsel ? P.len1 : P.len0, a0h, a0l, std::integral_constant<int, KS0>{}, osc0);
        if constexpr (KS1 > 0) build(tl + TLP, P.len1, a1h, a1l, std::integral_constant<int, KS1>{}, osc1);
        __syncthreads();                                            // the row buffers take this space over
    }

    float4 pre[kMmQ];                                               // the chunk being loaded (raw samples)
    Mean2 mu_n{0.f, 0.f};                                           // its row's mean (hi + lo) ...
    float g_n = 1.f;                                                // ... and 1 / maxabs
    bool heavy_n = false;                                           // the row is (nearly) all offset: scale it chunk by chunk
    bool tail_n = false;                                            // the chunk reaches beyond the row (wave-uniform)
    int row_n = 0, c0_n = 0;

    auto issue = [&](long long c) {                                 // global loads of chunk c into pre[]
        row_n = (int)(c / nchunk);
        c0_n = (int)(c - (long long)row_n * nchunk) * kMmCH;
        mu_n = mean2_load(P.mean, row_n);
        g_n = 1.f;
        const int s0 = c0_n + P.shift;                              // first sample of the chunk's stage
        tail_n = s0 + kMmStage > ns;
        if (P.maxabs) {
            const float a = P.maxabs[row_n];
            g_n = (a > 0.f) ? 1.0f / a : 0.f;
            // |mean| within 1 / 128 of max|x|: the deviations are at most a hundredth of what 1 / max|x| normalises by and
            // would sit in (or below) binary16's subnormal range -- such rows take the per-chunk power of two as well
            heavy_n = fabsf(mu_n.hi) > 0.9921875f * a;
        }
        const float* xr = P.x + (size_t)row_n * ns;
        const bool al = (reinterpret_cast<uintptr_t>(xr + s0) & 15) == 0;
        if (al && s0 + kMmStage <= ns) {
            const float4* p = reinterpret_cast<const float4*>(xr + s0) + tid;
            static_for<kMmQ>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                if (q < kMmQ - 1 || tid < kMmLastQ) pre[q] = mm_load4_stream(p + q * kMmThreads);
            });
        } else {
            // a row end, an unaligned row, or the record's continuation.  Clamped addresses and selects instead of branches, so
            // that a lane's loads go out together; what lies beyond the data (filled with the mean's high part here) is set to
            // the exact zero of the zero-padded correlation after the de-meaning, by index (tail chunks only)
            if (al && (ns & 3) == 0) {                              // every 16-byte group lies inside the row or beyond it
                static_for<kMmQ>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    if (q < kMmQ - 1 || tid < kMmLastQ) {
                        const int i = s0 + 4 * (tid + q * kMmThreads);
                        const float4 v = *reinterpret_cast<const float4*>(xr + min(i, ns - 4));
                        pre[q] = (i < ns) ? v : make_float4(mu_n.hi, mu_n.hi, mu_n.hi, mu_n.hi);
                    }
                });
            } else {
                static_for<kMmQ>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    if (q < kMmQ - 1 || tid < kMmLastQ) {
                        const int i = s0 + 4 * (tid + q * kMmThreads);
                        const float a0 = xr[min(i, ns - 1)], a1 = xr[min(i + 1, ns - 1)], a2 = xr[min(i + 2, ns - 1)], a3 = xr[min(i + 3, ns - 1)];
                        pre[q] = make_float4(i < ns ? a0 : mu_n.hi, i + 1 < ns ? a1 : mu_n.hi, i + 2 < ns ? a2 : mu_n.hi, i + 3 < ns ? a3 : mu_n.hi);
                    }
                });
            }
            if (P.xnext && P.n_next > 0 && s0 + kMmStage > ns) {    // the head of the next file behind the row
                const float* xn = P.xnext + (size_t)row_n * P.ld_next;
                const int n_next = P.n_next;
                static_for<kMmQ>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    if (q < kMmQ - 1 || tid < kMmLastQ) {
                        const int d = s0 + 4 * (tid + q * kMmThreads) - ns;
                        const float b0 = xn[min(max(d, 0), n_next - 1)], b1 = xn[min(max(d + 1, 0), n_next - 1)];
                        const float b2 = xn[min(max(d + 2, 0), n_next - 1)], b3 = xn[min(max(d + 3, 0), n_next - 1)];
                        if (d >= 0 && d < n_next) pre[q].x = b0;
                        if (d + 1 >= 0 && d + 1 < n_next) pre[q].y = b1;
                        if (d + 2 >= 0 && d + 2 < n_next) pre[q].z = b2;
                        if (d + 3 >= 0 && d + 3 < n_next) pre[q].w = b3;
                    }
                });
            }
        }
    };

    if (c_n < end_c) issue(c_n);
    int buf = 0;
    constexpr bool want_max = WMAX;
    // TAIL: prefix of the normalised row at the chunk's first sample, a float64 kept as two wave-uniform floats (scalar registers)
    float pst_hi = 0.f, pst_lo = 0.f;
    for (long long c = c_n; c < end_c;) {
        const int row = row_n, c0 = c0_n;
        const Mean2 mu = mu_n;
        const bool tail = tail_n;
        const int n_valid = ns + ((P.xnext && P.n_next > 0) ? P.n_next : 0) - c0 - P.shift;     // samples of the stage that exist
        float gsc = g_n, osx = 1.f;                                 // x scale applied before the split, and what undoes it
        const float gout = g_n;                                     // the normalisation's factor, applied to the outputs when a chunk scales itself
        const bool own_scale = !P.maxabs || heavy_n;                // wave- and workgroup-uniform (one row per chunk)
        mm_half* bh = lds + (size_t)buf * 2 * kMmArr;
        mm_half* bl = bh + kMmArr;
        const long long c_next = next_chunk(c);
        const int cin = c0 / kMmCH;                                 // the chunk's number inside its row
        if (TAIL && cin == 0) { pst_hi = 0.f; pst_lo = 0.f; }
        const float pst = pst_hi;
        // ---- convert the loaded chunk: (x - mu) * scale -> hi / lo halves in LDS
        if (own_scale) {
            // no row maximum from the caller, or a row that is all offset: this chunk's own power of two.  (With a row maximum the rows are scaled by
            // 1 / max|x| alone: a row whose signal is small against its offset then sits low in the binary16 range, which is
            // harmless -- v_cvt_f16_f32 and the matrix instruction keep binary16 subnormals, scripts/probe/denorm_probe.py:
            // taps at 1e-7 of the largest one still come out at 3e-7 -- and the per-chunk reduction costs a barrier, 3 %.)
            float m = 0.f;
            static_for<kMmQ>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                if (q < kMmQ - 1 || tid < kMmLastQ) {
                    const float4 v = pre[q];
                    const int at = 4 * (tid + q * kMmThreads);
                    float d0 = demean(v.x, mu), d1 = demean(v.y, mu), d2 = demean(v.z, mu), d3 = demean(v.w, mu);
                    if (tail) {
                        if (at >= n_valid) d0 = 0.f;
                        if (at + 1 >= n_valid) d1 = 0.f;
                        if (at + 2 >= n_valid) d2 = 0.f;
                        if (at + 3 >= n_valid) d3 = 0.f;
                    }
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
                }
            });
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) red[wv + 4 * buf] = m;
            lds_barrier();
            m = fmaxf(fmaxf(red[4 * buf], red[4 * buf + 1]), fmaxf(red[4 * buf + 2], red[4 * buf + 3]));
            mm_pow2_scale(m, osx, gsc);
        }
        const float mlg = -mu.lo * gsc;
        static_for<kMmQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            const bool mine = q < kMmQ - 1 || tid < kMmLastQ;
            const int at = mm_pidx(4 * (tid + q * kMmThreads));
            if (mine) {
                const float4 v = pre[q];
                // ((x - hi) - lo) g as (x - hi) g - lo g: the two-float mean at the instruction count of a float32 one (x - hi is
                // exact where the offset dominates, the product is rounded once)
#ifdef D4W_MM_V_FLOATMEAN          // (probe builds: what the two-float mean costs)
                s[0] = (v.x - mu.hi) * gsc; s[1] = (v.y - mu.hi) * gsc; s[2] = (v.z - mu.hi) * gsc; s[3] = (v.w - mu.hi) * gsc;
#else
                s[0] = fmaf(v.x - mu.hi, gsc, mlg); s[1] = fmaf(v.y - mu.hi, gsc, mlg); s[2] = fmaf(v.z - mu.hi, gsc, mlg); s[3] = fmaf(v.w - mu.hi, gsc, mlg);
#endif
                // the next file's head (or statistics that are not the rows' own, D4W_MM_CLAMP=1) may leave |v| beyond binary16's
                // range: inf - inf would turn a whole tile into NaN where the float32 forms stay finite; one v_med3_f32 per
                // sample, only where asked for (a kernel argument: a scalar branch)
                if (P.clamp && !own_scale) static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; s[e] = mm_clamp_half(s[e]); });
                if (tail) {                                          // beyond the data: the zero padding of the correlation
                    const int a0 = 4 * (tid + q * kMmThreads);
                    static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; if (a0 + e >= n_valid) s[e] = 0.f; });
                }
#ifdef D4W_MM_V_OLDSPLIT
                mm_half h[4], l[4];
                static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; mm_split(s[e], h[e], l[e]); });
                mm_put4(bh + at, h);
                mm_put4(bl + at, l);
#else
                mm_split_put4(s, bh + at, bl + at);
#endif
            }
#ifdef D4W_MM_V_TAIL_NOSCAN         // (probe builds: no prefix scan, no tail term -- timing only)
            if constexpr (false) {
#else
            if constexpr (TAIL && q < kMmCH / (4 * kMmThreads)) {   // the chunk's own 4096 samples (the halo adds no block of lags)
#endif
                // prefix of the scaled samples at every fourth lane = every block of 16 samples, inside this wave's segment
                const float t4 = (s[0] + s[1]) + (s[2] + s[3]);
                const float inc = mm_wave_scan(t4);
                pb[buf * kBlk + kMmThreads * q + tid] = inc - t4;        // (every lane stores: no exec-mask juggling; the epilogue reads every fourth)
                if (lane == 63) wt[buf * 16 + 4 * q + wv] = inc;
            }
        });
        // ---- next chunk's loads fly across the barrier and the matrix phase
        if (c_next < end_c) issue(c_next);
        lds_barrier();
        // ---- 16 tiles of 256 lags, 4 per wave: C[i][a] (+)= A[i][u] B[u][a]
        float* ya = ((WSPLIT && tsel) ? P.y1 : P.y0) + (size_t)row * ns;
        float* yb = KS1 ? P.y1 + (size_t)row * ns : nullptr;
        const bool valign = ((reinterpret_cast<uintptr_t>(ya + c0) & 15) == 0) && (!KS1 || (reinterpret_cast<uintptr_t>(yb + c0) & 15) == 0);
        const float oxs = (own_scale && P.maxabs) ? osx * gout : osx;
        const float o0 = osc0 * oxs, o1 = osc1 * oxs;
        float segoff = 0.f;                                         // TAIL: the prefix at each segment's start (lane l: segment l), formed at the first tile's end
        // the wave's four tiles as ONE software pipeline over (tile, k-step): the fragment pair of step s + PF is requested
        // before the six products of step s are issued (mm_sched_fence keeps hipcc from sinking the reads back to their use),
        // so an LDS round trip hides under 12 matrix instructions instead of stalling the wave at every k-step
        constexpr int NTW = kMmCH / 256 / (WSPLIT ? 2 : 4), NST = NTW * KSM, PF = 2;
        auto tile_of = [&](int ti) { return WSPLIT ? (wv & 1) + 2 * ti : wv + 4 * ti; };     // the wave's ti-th tile
        auto frag = [&](const mm_half* arr, int T, int kk) -> mm_h8 {
            const int gr = 32 * T + 2 * n16 + g + 4 * kk;           // 16-byte granule: sample 256 T + 16 n16 + 32 kk + 8 g
            return *reinterpret_cast<const mm_h8*>(arr + mm_pidx(8 * gr));
        };
        mm_h8 fh[PF + 1], fl[PF + 1];
        static_for<PF>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value;
            fh[s_] = frag(bh, tile_of(s_ / KSM), s_ % KSM);
            fl[s_] = frag(bl, tile_of(s_ / KSM), s_ % KSM);
        });
        mm_f4 c0h = mm_zero(), c0l = mm_zero(), c1h = mm_zero(), c1l = mm_zero();
        float vmax0 = -INFINITY, vmax1 = -INFINITY;                 // this lane's largest stored value of the chunk
        float vsum = 0.f;                                           // ... and a sum that is NaN when any stored value was (np.max propagates NaN)
        static_for<NST>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value, ti = s_ / KSM, kk = s_ % KSM;
            if constexpr (s_ + PF < NST) {
                fh[(s_ + PF) % (PF + 1)] = frag(bh, tile_of((s_ + PF) / KSM), (s_ + PF) % KSM);
                fl[(s_ + PF) % (PF + 1)] = frag(bl, tile_of((s_ + PF) / KSM), (s_ + PF) % KSM);
            }
            mm_sched_fence();
            const mm_h8 xh = fh[s_ % (PF + 1)], xl = fl[s_ % (PF + 1)];
            if constexpr (kk < KS0) {
                c0h = mm_mfma(a0h[kk], xh, c0h);
                c0l = mm_mfma(a0h[kk], xl, c0l);
            }
            if constexpr (kk < KS1) {
                c1h = mm_mfma(a1h[kk], xh, c1h);
                c1l = mm_mfma(a1h[kk], xl, c1l);
            }
            if constexpr (kk < KS0) c0l = mm_mfma(a0l[kk], xh, c0l);
            if constexpr (kk < KS1) c1l = mm_mfma(a1l[kk], xh, c1l);
            mm_sched_fence();
            if constexpr (kk == KSM - 1) {                          // the tile is complete: scale, combine, stream out
                const int T = tile_of(ti);
                const int kl = 256 * T + 16 * n16 + 4 * g;          // this lane's four lags inside the chunk ...
                const int k = c0 + kl;                              // ... and inside the row
                if constexpr (ti == 0) {
                    // (here, not ahead of the matrix instructions: nothing before the first tile's end needs it)
                // TAIL: the prefix at each segment's start (lane l: segment l) and the chunk's sum, which moves the row's prefix on (every
                // wave forms the same values from the same LDS words)
                if constexpr (TAIL) {
                    const float w = lane < 16 ? wt[buf * 16 + lane] : 0.f;
                    const float inc = mm_wave_scan(w);
                    segoff = inc - w;
        #ifdef D4W_EMU
                    const float own = __shfl(inc, 15);
        #else
                    const float own = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 15));
        #endif
                    const double pd = ((double)pst_hi + (double)pst_lo) + (double)(own * oxs);
                    const float ph = (float)pd, pl2 = (float)(pd - (double)ph);
        #ifdef D4W_EMU
                    pst_hi = ph; pst_lo = pl2;
        #else
                    pst_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ph)));
                    pst_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pl2)));
        #endif
                }
                }
                float r0[4], r1[4];
                float a0 = 0.f, a1 = 0.f;                           // TAIL: tail_t x (prefix at the block's first sample), the same for the lane's four lags
#ifdef D4W_MM_V_TAIL_NOSCAN
                if constexpr (false) {
#else
                if constexpr (TAIL) {
#endif
                    // tile T = the segment this wave converted: row prefix at the chunk + segment offset + block prefix
#ifdef D4W_EMU
                    const float so = __shfl(segoff, T);
#else
                    const float so = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, segoff), T));
#endif
                    // (tile T = segment T of the stage = load T / 4 of wave T % 4)
                    const float pbk = fmaf(so + pb[buf * kBlk + kMmThreads * (T >> 2) + 64 * (T & 3) + 4 * n16], oxs, pst);
                    a0 = ((WSPLIT && tsel) ? P.tail1 : P.tail0) * pbk;
                    a1 = P.tail1 * pbk;
                }
                static_for<4>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    // (the tail's addend rides the scaling multiply: an FMA instead of a multiply)
                    r0[r] = fmaf(fmaf(mm_get(c0l, r), kMmLoInv, mm_get(c0h, r)), o0, a0);
                    if constexpr (KS1 > 0) r1[r] = fmaf(fmaf(mm_get(c1l, r), kMmLoInv, mm_get(c1h, r)), o1, a1);
                });
                c0h = mm_zero(); c0l = mm_zero(); c1h = mm_zero(); c1l = mm_zero();
                if (valign && k + 3 < ns) {
                    if (KS1 == 0 && P.accumulate) {                 // a later section of a long template
                        const float4 o = mm_load4_stream(reinterpret_cast<const float4*>(ya + k));
                        r0[0] += o.x; r0[1] += o.y; r0[2] += o.z; r0[3] += o.w;
                    }
                    mm_store4(ya + k, r0[0], r0[1], r0[2], r0[3]);
                    if constexpr (KS1 > 0) mm_store4(yb + k, r1[0], r1[1], r1[2], r1[3]);
                    if (want_max) {
                        vmax0 = fmaxf(vmax0, fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])));
#ifndef D4W_MM_V_NONAN
                        vsum += (r0[0] + r0[1]) + (r0[2] + r0[3]);
#endif
                        if constexpr (KS1 > 0) {
                            vmax1 = fmaxf(vmax1, fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
#ifndef D4W_MM_V_NONAN
                            vsum += (r1[0] + r1[1]) + (r1[2] + r1[3]);
#endif
                        }
                    }
                } else {                                            // a row end or an unaligned row (tiles beyond the row: nothing)
                    for (int r = 0; r < 4; ++r)
                        if (k + r < ns) {
                            const float v0 = (KS1 == 0 && P.accumulate) ? ya[k + r] + r0[r] : r0[r];
                            ya[k + r] = v0;
                            if constexpr (KS1 > 0) yb[k + r] = r1[r];
                            if (want_max) {
                                vmax0 = fmaxf(vmax0, v0);
                                vsum += v0;
                                if constexpr (KS1 > 0) { vmax1 = fmaxf(vmax1, r1[r]); vsum += r1[r]; }
                            }
                        }
                }
            }
        });
        if (want_max) {                                             // one atomic per wave, chunk and template
            for (int o = 32; o > 0; o >>= 1) {
                vmax0 = fmaxf(vmax0, __shfl_xor(vmax0, o));
                if constexpr (KS1 > 0) vmax1 = fmaxf(vmax1, __shfl_xor(vmax1, o));
            }
            // a NaN among the stored values: the row's maximum is NaN, as np.max has it (fmaxf drops NaN; the sum does not -- it
            // is also NaN for a chunk that holds +inf and -inf, where a NaN maximum is no loss).  Both templates read the same
            // samples, so a NaN in one is a NaN in the other.
            const bool bad = __any(vsum != vsum);
            if (lane == 0) {
                mm_atomic_fmax(((WSPLIT && tsel) ? P.rowmax1 : P.rowmax0) + row, bad ? __uint_as_float(0x7FC00000u) : vmax0);
                if constexpr (KS1 > 0) mm_atomic_fmax(P.rowmax1 + row, bad ? __uint_as_float(0x7FC00000u) : vmax1);
            }
        }
        buf ^= 1;
        c = c_next;
    }
}

}  // namespace d4w

using namespace d4w;

extern "C" {

int d4w_xcorr_mm_max_support(void) { return kMmSection * kMmMaxSections; }

// one template of any support <= d4w_xcorr_mm_max_support(): sections of kMmSection taps, the first one overwriting y, the
// later ones (x shifted by the section's first tap) accumulating into it
// launch of xcorr_mm_rows<KS0, KS1, WPS, TAIL> with or without the row maxima
#define D4W_MM_LAUNCH(KS0, KS1, WPS, TAIL, grid, lds, stream, Q)                                                              \
    do {                                                                                                                      \
        if ((Q).rowmax0) D4W_LAUNCH((xcorr_mm_rows<KS0, KS1, WPS, TAIL, true>), dim3(grid), dim3(kMmThreads), lds, stream, Q);  \
        else D4W_LAUNCH((xcorr_mm_rows<KS0, KS1, WPS, TAIL, false>), dim3(grid), dim3(kMmThreads), lds, stream, Q);             \
    } while (0)

static int mm_one_template(MmArgs P, const float* taps, int len, float* y, float* rowmax, int grid, void* stream) {
    auto lds_of = [](int arr) { return (size_t)4 * arr * sizeof(mm_half) + 8 * sizeof(float); };
    const int nsec = (len <= 32 * kMmKSMax - 15) ? 1 : ceil_div(len, kMmSection);
    // sections of equal length (a multiple of 16 taps, so that the shifted 16-byte loads stay aligned): 700 taps run as
    // 352 + 348 through the 12-step kernel twice instead of 496 + 204 through the 16- and the 8-step kernels
    const int per = (nsec == 1) ? len : 16 * ceil_div(ceil_div(len, nsec), 16);
    for (int j = 0; j < nsec; ++j) {
        const int first = j * per;
        MmArgs Q = P;
        Q.taps = taps + first;
        Q.len0 = Q.len1 = std::min(per, len - first);
        Q.y0 = y;
        Q.y1 = nullptr;
        Q.shift = first;
        Q.accumulate = j > 0;
        Q.rowmax0 = (j == nsec - 1) ? rowmax : nullptr;             // the maxima of the finished sums
        Q.rowmax1 = nullptr;
        const int ks = ceil_div(Q.len0 + 15, 32);
        if (ks <= kMmKS)
            D4W_MM_LAUNCH(kMmKS, 0, 3, false, grid, lds_of(MmGeom<kMmKS>::Arr), stream, Q);
        else if (ks <= kMmKSLong)
            D4W_MM_LAUNCH(kMmKSLong, 0, 2, false, grid, lds_of(MmGeom<kMmKSLong>::Arr), stream, Q);
        else if (ks <= 12)
            D4W_MM_LAUNCH(12, 0, 2, false, grid, lds_of(MmGeom<12>::Arr), stream, Q);
        else
            D4W_MM_LAUNCH(kMmKSMax, 0, 2, false, grid, lds_of(MmGeom<kMmKSMax>::Arr), stream, Q);
    }
    return D4W_OK;
}

int d4w_xcorr_mm_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next, const double* mean,
                     const float* maxabs, const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0, float* y1,
                     void* stream) {
    return d4w_xcorr_mm_rowmax_f32(x, nx, ns, xnext, ld_next, n_next, mean, maxabs, taps, ntpl, ltaps, len0, len1, y0, y1, nullptr,
                                   nullptr, stream);
}

int d4w_xcorr_mm_rowmax_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next, const double* mean,
                            const float* maxabs, const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0, float* y1,
                            float* rowmax0, float* rowmax1, void* stream) {
    return d4w_xcorr_mm_tail_f32(x, nx, ns, xnext, ld_next, n_next, mean, maxabs, taps, ntpl, ltaps, len0, len1, 0.0, 0.0, y0, y1,
                                 rowmax0, rowmax1, stream);
}

// longest support the kernels take WITH the zero-padded template's tail added in the epilogue: one launch per template
int d4w_xcorr_mm_tail_max_support(void) { return 32 * kMmKSMax - 15; }

int d4w_xcorr_mm_tail_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next, const double* mean,
                          const float* maxabs, const float* taps, int ntpl, int ltaps, int len0, int len1, double tail0, double tail1,
                          float* y0, float* y1, float* rowmax0, float* rowmax1, void* stream) {
    if (!x || !y0 || !taps || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (ntpl == 2 && ((rowmax0 == nullptr) != (rowmax1 == nullptr))) return fail(D4W_EINVAL, "rowmax0 and rowmax1 go together");
    if (ntpl < 1 || ntpl > 2 || (ntpl == 2 && !y1)) return fail(D4W_EINVAL, "ntpl = %d (1 or 2 templates per call)", ntpl);
    if (xnext && (n_next < 0 || ld_next < n_next)) return fail(D4W_EINVAL, "a continuation needs 0 <= n_next <= ld_next");
    if (ntpl == 1) { len1 = len0; tail1 = 0.0; }
    if (len0 < 1 || len1 < 1 || len0 > ltaps || len1 > ltaps || std::max(len0, len1) > d4w_xcorr_mm_max_support())
        return fail(D4W_EINVAL, "template supports (%d, %d) must lie in 1..min(ltaps = %d, %d)", len0, len1, ltaps, d4w_xcorr_mm_max_support());
    const bool tails = tail0 != 0.0 || tail1 != 0.0;
    if (tails) {
        // the tail is a prefix sum of the NORMALISED row: the rows' statistics, and one launch per template
        if (!mean || !maxabs) return fail(D4W_EINVAL, "the zero-padded template's tail needs the rows' statistics (mean, maxabs)");
        if (std::max(len0, len1) > d4w_xcorr_mm_tail_max_support())
            return fail(D4W_EINVAL, "with a tail the supports (%d, %d) must be <= %d", len0, len1, d4w_xcorr_mm_tail_max_support());
    }
    MmArgs P;
    P.x = x; P.xnext = xnext; P.mean = mean; P.maxabs = maxabs; P.taps = taps; P.y0 = y0; P.y1 = y1;
    P.nx = nx; P.ns = ns; P.ld_next = ld_next; P.n_next = xnext ? n_next : 0; P.ltaps = ltaps; P.len0 = len0; P.len1 = len1;
    P.shift = 0; P.accumulate = 0;
    static const int env_clamp = [] { const char* v = getenv("D4W_MM_CLAMP"); return v ? atoi(v) : 0; }();
    P.clamp = (xnext != nullptr && n_next > 0) || env_clamp;
    P.rowmax0 = rowmax0; P.rowmax1 = (ntpl == 2) ? rowmax1 : nullptr;
    P.tail0 = (float)tail0; P.tail1 = (float)tail1;
    if (rowmax0) {                                                  // -inf: the identity of the epilogue's integer-atomic float max
        D4W_HIP(hipMemsetD32Async((hipDeviceptr_t)rowmax0, (int)0xFF800000u, (size_t)nx, (hipStream_t)stream));
        if (ntpl == 2) D4W_HIP(hipMemsetD32Async((hipDeviceptr_t)rowmax1, (int)0xFF800000u, (size_t)nx, (hipStream_t)stream));
    }
    const int nchunk = ceil_div(ns, kMmCH);
    const long long total = (long long)nx * nchunk;
    // persistent workgroups per compute unit: 2 for two templates (207 VGPRs: the Toeplitz fragments of both templates stay
    // in registers; a 168-register build for three workgroups spills and ran 8.5 ms against 6.6), 3 for one template
    // (149 VGPRs), 2 for the deeper one-template kernels.  D4W_MM_WGS overrides the count (measurements).
    static const int env_wgs = [] { const char* v = getenv("D4W_MM_WGS"); const int n = v ? atoi(v) : 0; return n < 0 ? 0 : (n > 8 ? 8 : n); }();
    const int ks0 = ceil_div(len0 + 15, 32), ks1 = ceil_div(len1 + 15, 32);
    const bool fused = ntpl == 2 && std::max(ks0, ks1) <= kMmKS;
    const int per_cu = env_wgs ? env_wgs : ((ntpl == 1 && ks0 <= kMmKS) ? 3 : 2);
    const int ncu = mm_num_cus();
    const int grid = (int)std::min<long long>(total, (long long)ncu * per_cu);
    auto lds_of = [](int arr, bool tl) {
        return (size_t)4 * arr * sizeof(mm_half) + 8 * sizeof(float) + (tl ? ((size_t)2 * (kMmCH / 4) + 2 * 16) * sizeof(float) : 0);
    };
    // two templates of <= 177 samples: every wave both templates, two workgroups per CU (the kernel of rounds 4-6).  D4W_MM_FUSED=3:
    // the wave-split kernel at three workgroups per CU -- built and measured in round 6, 6.32 against 6.11 ms (with the tail 6.84
    // against 6.66): the second read of every sample fragment costs more than the third workgroup brings (profiles/r06m)
    static const int fused_form = [] { const char* v = getenv("D4W_MM_FUSED"); return v ? atoi(v) : 2; }();
    const int per_cu_ws = env_wgs ? env_wgs : 3;
#define D4W_MM_LAUNCH_WS(TAIL, grid, lds, Q)                                                                                            \
    do {                                                                                                                                \
        if ((Q).rowmax0) D4W_LAUNCH((xcorr_mm_rows<kMmKS, 0, 3, TAIL, true, true>), dim3(grid), dim3(kMmThreads), lds, stream, Q);       \
        else D4W_LAUNCH((xcorr_mm_rows<kMmKS, 0, 3, TAIL, false, true>), dim3(grid), dim3(kMmThreads), lds, stream, Q);                  \
    } while (0)
    if (tails) {
        // whole rows per workgroup (the prefix is carried along a row): at most one workgroup per row
        const int grid_t = (int)std::min<long long>((long long)nx, (long long)ncu * per_cu);
        if (fused && fused_form != 2) {
            D4W_MM_LAUNCH_WS(true, (int)std::min<long long>((long long)nx, (long long)ncu * per_cu_ws), lds_of(MmGeom<kMmKS>::Arr, true), P);
            return D4W_OK;
        }
        if (fused) {
            const size_t lds = lds_of(MmGeom<kMmKS>::Arr, true);
            if (ks0 <= 5)
                D4W_MM_LAUNCH(5, kMmKS, 2, true, grid_t, lds, stream, P);
            else
                D4W_MM_LAUNCH(kMmKS, kMmKS, 2, true, grid_t, lds, stream, P);
            return D4W_OK;
        }
        for (int t = 0; t < ntpl; ++t) {
            MmArgs Q = P;
            Q.taps = taps + (size_t)t * ltaps;
            Q.len0 = Q.len1 = t ? len1 : len0;
            Q.tail0 = t ? (float)tail1 : (float)tail0;
            Q.tail1 = 0.f;
            Q.y0 = t ? y1 : y0;
            Q.y1 = nullptr;
            Q.rowmax0 = t ? rowmax1 : rowmax0;
            Q.rowmax1 = nullptr;
            const int ks = ceil_div(Q.len0 + 15, 32);
            if (ks <= kMmKS) {
                // (three workgroups per compute unit fit the registers only without the row maxima)
                if (Q.rowmax0)
                    D4W_LAUNCH((xcorr_mm_rows<kMmKS, 0, 2, true, true>), dim3(grid_t), dim3(kMmThreads), lds_of(MmGeom<kMmKS>::Arr, true), stream, Q);
                else
                    D4W_LAUNCH((xcorr_mm_rows<kMmKS, 0, 3, true, false>), dim3(grid_t), dim3(kMmThreads), lds_of(MmGeom<kMmKS>::Arr, true), stream, Q);
            }
            else if (ks <= kMmKSLong)
                D4W_MM_LAUNCH(kMmKSLong, 0, 2, true, grid_t, lds_of(MmGeom<kMmKSLong>::Arr, true), stream, Q);
            else if (ks <= 12)
                D4W_MM_LAUNCH(12, 0, 2, true, grid_t, lds_of(MmGeom<12>::Arr, true), stream, Q);
            else
                D4W_MM_LAUNCH(kMmKSMax, 0, 2, true, grid_t, lds_of(MmGeom<kMmKSMax>::Arr, true), stream, Q);
        }
        return D4W_OK;
    }
    if (!fused) {
        // one template, or a support beyond 177 samples: the templates one after the other through the one-template kernels
        // (the Toeplitz fragments of one template alone fill the registers the fused kernel splits between two)
        int rc = mm_one_template(P, taps, len0, y0, rowmax0, grid, stream);
        if (rc == D4W_OK && ntpl == 2) rc = mm_one_template(P, taps + ltaps, len1, y1, rowmax1, grid, stream);
        return rc;
    }
    const size_t lds = lds_of(MmGeom<kMmKS>::Arr, false);
    if (fused_form != 2) {
        D4W_MM_LAUNCH_WS(false, (int)std::min<long long>(total, (long long)ncu * per_cu_ws), lds, P);
        return D4W_OK;
    }
    if (ks0 <= 5)
        D4W_MM_LAUNCH(5, kMmKS, 2, false, grid, lds, stream, P);
    else
        D4W_MM_LAUNCH(kMmKS, kMmKS, 2, false, grid, lds, stream, P);
    return D4W_OK;
}

}  // extern "C"
