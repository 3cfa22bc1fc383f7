// EXPERIMENT, NOT BUILT (round 4): the zero-phase band-pass as a decimate-by-2 polyphase cascade on the matrix cores.
// Correct (CPU test build: 3e-6 of the row maximum against the float64 response on offset / spike / zero rows; GPU: same
// error as the FFT form against the float64 filtfilt) but SLOWER than the overlap-save FFT kernel it was meant to replace:
//   20 000 x 120 000:  four barriers per frame 6.40 ms;  two-barrier software pipeline 6.76;  + the wave's two tiles side by
//   side 6.72;  one workgroup per CU 9.9  --  d4w_fir_fft_f32: 5.47 ms.
// Counters (profiles/r04g/pmc_sq_fir_poly.txt, 4000 rows): 503 VALU (132 of them matrix), 127 LDS and 172 scalar
// instructions per wave and 4096-sample frame = ~900 issue quad-cycles; with the 120 VGPRs of the half-rate Toeplitz
// fragments only two waves fit a SIMD, they are issuing 2 x 33 % of the time and waiting 44 % (barriers, LDS, memory), 24 % of
// the LDS cycles are bank conflicts (stage 1 reads columns 64 bytes apart).  The three split stages per sample cost what
// the shorter filter saves.  To build it: copy to das4whales_amd/csrc/, declare d4w_fir_poly_shape / d4w_fir_poly_f32 in
// include/d4w.h and _lib.py (signatures below), design the taps with scripts/probe/polyphase_bp.py's formulas.
// Zero-phase band-pass (the INTERIOR of dsp.bp_filt / scipy.signal.sosfiltfilt, reference dsp.py:859-880) as a
// decimate-by-2 polyphase cascade on the MI355X matrix cores (gfx950).
//
// Away from the row ends the zero-phase filter is the convolution with its two-sided response g (925 taps for the
// 14-30 Hz Butterworth-8): as a banded-Toeplitz product in xcorr_mm.hip's form that is 90 matrix instructions per 256
// samples, more than the matrix pipe does in the HBM time of the block.  g is band-limited (below 1e-7 of its peak
// beyond fs / 4), so the host factors it (dsp._polyphase_taps, checked numerically against g before it is used):
//
//      u[m]      = sum_j a[j]  x[2 m + j]          63-tap anti-alias low-pass, every second output     (stage 1)
//      v[m]      = sum_k g2[k] u[m + k]            the response at half rate, <= 465 taps               (stage 2)
//      y[2m + p] = sum_k bp[k] v[m + k]            the two phases of the 63-tap interpolator            (stage 3)
//
// 9 + 45 + 12 matrix instructions per 512 samples = 33 per 256, the matched filter's count.  Every stage is the product
// C[i][c] = sum_u A[i][u] B[u][c] of xcorr_mm.hip with a Toeplitz A (stage 1: A[i][u] = a[u - 2 i], columns 32 samples
// apart) and B read straight out of an LDS copy of the previous stage's output, operands split into binary16 hi / lo
// pairs (mm_common.h).  Outputs of a stage index the FIRST sample their taps meet, so nothing has to be centred: the
// result for time t appears 524 samples "late", and frame j of 4096 input samples writes the outputs
// [4096 j - 628, 4096 (j + 1) - 628) of its row (16-byte aligned stores).
//
// A workgroup (256 threads) walks whole rows frame by frame and keeps what the next frame needs in LDS: the last 64
// samples, 480 values of u and 64 of v.  The data of a frame are scaled by the power of two that bounds the last three
// frames (what the histories were computed from), the histories are re-scaled exactly when that power changes -- nothing
// leaves the binary16 range whatever the record does.  The frames run as a software pipeline with TWO barriers each:
// stage 1 of frame f next to stage 3 of frame f - 1, then stage 2 of frame f next to the conversion of frame f + 1 (u and v
// alternate between two arrays each, so no phase writes what its other half reads); the loads of frame f + 2 are in
// flight meanwhile.  8 B per sample: 4 read, 4 written.
#include <cstdlib>

#include "mm_common.h"

namespace d4w {

constexpr int kPfCH = 4096, kPfH = kPfCH / 2, kPfThreads = 256;
constexpr int kPfKS1 = 3, kPfKS2 = 15, kPfKS3 = 2;       // k-steps of 32: Toeplitz depths 96 / 480 / 64
constexpr int kPfNa = 63, kPfNg = 461, kPfNb = 32;      // taps: a, g2 (<= 32 KS2 - 15), each interpolator phase
constexpr int kPfS1 = 2;                                 // leading zeros of stage 1's taps: makes the output lag a multiple of 4
constexpr int kPfXh = 64, kPfUh = 32 * kPfKS2, kPfVh = 64;           // histories kept in front of a frame's new values
constexpr int kPfXn = kPfXh + kPfCH, kPfUn = kPfUh + kPfH, kPfVn = kPfVh + kPfH;
constexpr int kPfGc = 230;                               // centre tap of stage 2 (461 taps)
// frame j's stage-3 outputs start at y' index 2 (2048 j - 32 - Uh - Vh); y'[n] is the response at time n + 31 + 2 Gc + 31 + S1
constexpr int kPfLag = 2 * (32 + kPfUh + kPfVh) - (kPfNa / 2 + 2 * kPfGc + kPfNa / 2 + kPfS1);
static_assert(kPfLag == 628 && kPfLag % 4 == 0, "frame j writes the outputs [4096 j - 628, 4096 (j + 1) - 628)");
constexpr int kPfPad = 8;
// LDS (halves): X hi / lo, two frames of U hi / lo and of V hi / lo, stage-1 and stage-3 fragment tables; then 8 floats for the frame maxima
constexpr int kPfT1 = kPfKS1 * 2 * 64 * 8, kPfT3 = 2 * kPfKS3 * 2 * 64 * 8;
constexpr int kPfHalves = 2 * (kPfXn + kPfPad) + 4 * (kPfUn + kPfPad) + 4 * (kPfVn + kPfPad) + kPfT1 + kPfT3;

struct PfArgs {
    const float* x;         // [nx][ns]
    const float* first;     // [nx] or NULL: subtracted before, first * dcg added after
    const float* ta;        // [na]  stage 1
    const float* tg;        // [ng]  stage 2; its centre tap is tg[gc]
    const float* tb;        // [2][nb] stage 3 phases
    float* y;               // [nx][ns]
    float dcg;
    int nx, ns, na, ng, nb;
};

__global__ __launch_bounds__(kPfThreads, 2) void fir_poly_rows(PfArgs P) {
    D4W_DYN_LDS(smem_raw);
    mm_half* Xh = reinterpret_cast<mm_half*>(smem_raw);
    mm_half* Xl = Xh + kPfXn + kPfPad;
    mm_half* Uh = Xl + kPfXn + kPfPad;                              // [2 frames][hi | lo][kPfUn + pad]
    mm_half* Vh = Uh + 4 * (kPfUn + kPfPad);                        // [2 frames][hi | lo][kPfVn + pad]
    mm_half* T1 = Vh + 4 * (kPfVn + kPfPad);                        // [kk][hi | lo][lane][8]
    mm_half* T3 = T1 + kPfT1;                                       // [phase][kk][hi | lo][lane][8]
    float* red = reinterpret_cast<float*>(T3 + kPfT3);             // [2][4]
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = mm_uniform(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int ns = P.ns;

    // ---- Toeplitz fragments: stage 2 in registers for the whole launch, stages 1 and 3 as LDS tables (16 bytes per lane)
    mm_h8 a2h[kPfKS2], a2l[kPfKS2];
    float osc1 = 1.f, osc2 = 1.f, osc3 = 1.f;                       // the powers of two taken out of the taps
    {
        float* tl = reinterpret_cast<float*>(smem_raw);            // staging (the row buffers are not in use yet)
        constexpr int L2 = 15 + 32 * kPfKS2, L1 = 30 + 32 * kPfKS1, L3 = 15 + 32 * kPfKS3;
        float* t2 = tl;                                             // t2[15 + d] = g2[d]
        float* t1 = t2 + L2;                                        // t1[30 + d] = a[d - kPfS1]
        float* t3 = t1 + L1;                                        // t3[p][15 + d] = b_p[d]
        for (int i = tid; i < L2; i += kPfThreads) { const int d = i - 15; t2[i] = (d >= 0 && d < P.ng) ? P.tg[d] : 0.f; }
        for (int i = tid; i < L1; i += kPfThreads) { const int d = i - 30 - kPfS1; t1[i] = (d >= 0 && d < P.na) ? P.ta[d] : 0.f; }
        for (int i = tid; i < 2 * L3; i += kPfThreads) {
            const int p = i / L3, d = i - p * L3 - 15;
            t3[i] = (d >= 0 && d < P.nb) ? P.tb[p * P.nb + d] : 0.f;
        }
        __syncthreads();
        auto scale_of = [&](const float* tp, int n, float& up, float& down) {
            float m = 0.f;
            for (int i = lane; i < n; i += 64) m = fmaxf(m, fabsf(tp[i]));
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            mm_pow2_scale(m, up, down);
        };
        float d1, d2, d3;
        scale_of(t1, L1, osc1, d1);
        scale_of(t2, L2, osc2, d2);
        scale_of(t3, 2 * L3, osc3, d3);
        static_for<kPfKS2>([&](auto kq) {
            constexpr int kk = decltype(kq)::value;
            static_for<8>([&](auto jq) {
                constexpr int j = decltype(jq)::value;
                mm_half hi, lo;
                mm_split(t2[15 + 32 * kk + 8 * g + j - n16] * d2, hi, lo);
                mm_set(a2h[kk], j, hi);
                mm_set(a2l[kk], j, lo);
            });
        });
        __syncthreads();                                            // every lane has read t2 / the scales; the tables overwrite nothing
        // the tables live behind the row buffers: written now, read in every frame
        if (wv == 0) {
            for (int kk = 0; kk < kPfKS1; ++kk)
                for (int j = 0; j < 8; ++j) {
                    mm_half hi, lo;
                    mm_split(t1[30 + 32 * kk + 8 * g + j - 2 * n16] * d1, hi, lo);       // A1[i][u] = a[u - 2 i - S1]
                    T1[((kk * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    T1[((kk * 2 + 1) * 64 + lane) * 8 + j] = lo;
                }
        } else if (wv == 1) {
            for (int p = 0; p < 2; ++p)
                for (int kk = 0; kk < kPfKS3; ++kk)
                    for (int j = 0; j < 8; ++j) {
                        mm_half hi, lo;
                        mm_split(t3[p * L3 + 15 + 32 * kk + 8 * g + j - n16] * d3, hi, lo);
                        T3[(((p * kPfKS3 + kk) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                        T3[(((p * kPfKS3 + kk) * 2 + 1) * 64 + lane) * 8 + j] = lo;
                    }
        }
        __syncthreads();
    }

    const int nfr = (ns + kPfLag + kPfCH - 1) / kPfCH;              // frames of a row (the last ones only flush the cascade)
    const int myrows = ((int)blockIdx.x < P.nx) ? (P.nx - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int F = myrows * nfr;                                     // this workgroup's frames, row after row
    float4 pre[4];                                                  // the frame being loaded: samples 4 (tid + 256 q) .. + 3
    float fr_n = 0.f;                                               // its row's offset
    auto issue = [&](int f) {
        const int row = (int)blockIdx.x + (f / nfr) * (int)gridDim.x, j = f % nfr;
        fr_n = P.first ? P.first[row] : 0.f;
        const float* xr = P.x + (size_t)row * ns;
        const int c0 = j * kPfCH;
        const bool al = (reinterpret_cast<uintptr_t>(xr) & 15) == 0 && (ns & 3) == 0;
        if (al && c0 + kPfCH <= ns) {
            const float4* p = reinterpret_cast<const float4*>(xr + c0) + tid;
            static_for<4>([&](auto qq) { constexpr int q = decltype(qq)::value; pre[q] = mm_load4_stream(p + q * kPfThreads); });
        } else {
            // a row end or an unaligned row: clamped addresses and selects; beyond the row the offset itself (-> 0 after it is
            // subtracted: the zero-padded convolution, whose outputs near the row ends the caller replaces anyway)
            static_for<4>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int i = c0 + 4 * (tid + q * kPfThreads);
                float v[4];
                for (int e = 0; e < 4; ++e) v[e] = (i + e < ns) ? xr[min(i + e, ns - 1)] : fr_n;
                pre[q] = make_float4(v[0], v[1], v[2], v[3]);
            });
        }
    };
    // wave maxima of |pre - offset| -> red[buffer][wave]
    auto frame_max = [&](int rb) {
        const float fr = fr_n;
        float m = 0.f;
        static_for<4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 v = pre[q];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x - fr), fabsf(v.y - fr))), fmaxf(fabsf(v.z - fr), fabsf(v.w - fr)));
        });
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[4 * rb + wv] = m;
    };
    // state of the frame whose samples are in X (frame f) and of the one before it
    float m1 = 0.f, m2 = 0.f;                                       // maxima of the two frames before the one being converted
    float up_x = 1.f, fr_x = 0.f;                                   // scale and offset of the frame in X
    float4 tail = make_float4(0.f, 0.f, 0.f, 0.f);                  // lanes >= 240: the last 64 samples of the frame in X, raw
    // (x - first) / up -> hi / lo halves of frame f (in pre[]); red[rb] holds its wave maxima.  Returns with up_x / fr_x of frame f.
    auto convert = [&](int f, int rb) {
        const int j = f % nfr;
        const float fr = fr_n;
        const float mj = fmaxf(fmaxf(red[4 * rb], red[4 * rb + 1]), fmaxf(red[4 * rb + 2], red[4 * rb + 3]));
        if (j == 0) { m1 = 0.f; m2 = 0.f; }
        float up, down;
        mm_pow2_scale(fmaxf(mj, fmaxf(m1, m2)), up, down);          // bounds this frame and the two before it (the histories' inputs)
        m2 = m1;
        m1 = mj;
        if (tid >= kPfThreads - kPfXh / 4) {                        // the 64 samples before the frame, from the registers that kept them
            const float s[4] = {(tail.x - fr) * down, (tail.y - fr) * down, (tail.z - fr) * down, (tail.w - fr) * down};
            mm_half h[4], l[4];
            static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; mm_split((j == 0) ? 0.f : s[e], h[e], l[e]); });
            const int at = 4 * (tid - (kPfThreads - kPfXh / 4));
            mm_put4(Xh + at, h);
            mm_put4(Xl + at, l);
            tail = pre[3];
        }
        static_for<4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 v = pre[q];
            const float s[4] = {(v.x - fr) * down, (v.y - fr) * down, (v.z - fr) * down, (v.w - fr) * down};
            mm_half h[4], l[4];
            static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; mm_split(s[e], h[e], l[e]); });
            const int at = kPfXh + 4 * (tid + q * kPfThreads);
            mm_put4(Xh + at, h);
            mm_put4(Xl + at, l);
        });
        up_x = up;
        fr_x = fr;
    };
    // history of a stage: the last n values of the previous frame's array to the front of this frame's, re-scaled (exactly:
    // a power of two) when the frames' scales differ; zeros at the start of a row.  4 halves per lane.
    auto move = [&](mm_half* dst, const mm_half* src, int n, bool zero, float ratio) {
        for (int i = 4 * tid; i < n; i += 4 * kPfThreads) {
            mm_half h[4];
            static_for<4>([&](auto ee) {
                constexpr int e = decltype(ee)::value;
                h[e] = mm_to_half(zero ? 0.f : mm_to_float(src[i + e]) * ratio);
            });
            mm_put4(dst + i, h);
        }
    };

    // one matrix phase: the wave's two tiles of 256 outputs as ONE software pipeline over (tile, k-step), the fragment
    // pair of step s + 2 requested before the products of step s (xcorr_mm.hip)
    auto phase = [&](const mm_half* bh, const mm_half* bl, auto ksq, auto strq, auto nsetq, auto&& afr, auto&& emit) {
        constexpr int KS = decltype(ksq)::value, STR = decltype(strq)::value, NSET = decltype(nsetq)::value;
        constexpr int NTW = kPfH / 256 / 4, NST = NTW * KS, PF = 2;
        auto frag = [&](const mm_half* arr, int T, int kk) -> mm_h8 {
            return *reinterpret_cast<const mm_h8*>(arr + 16 * STR * T + STR * n16 + 32 * kk + 8 * g);
        };
        mm_h8 fh[PF + 1], fl[PF + 1];
        static_for<PF>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value;
            fh[s_] = frag(bh, wv + 4 * (s_ / KS), s_ % KS);
            fl[s_] = frag(bl, wv + 4 * (s_ / KS), s_ % KS);
        });
        mm_f4 c0h = mm_zero(), c0l = mm_zero(), c1h = mm_zero(), c1l = mm_zero();
        static_for<NST>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value, ti = s_ / KS, kk = s_ % KS;
            if constexpr (s_ + PF < NST) {
                fh[(s_ + PF) % (PF + 1)] = frag(bh, wv + 4 * ((s_ + PF) / KS), (s_ + PF) % KS);
                fl[(s_ + PF) % (PF + 1)] = frag(bl, wv + 4 * ((s_ + PF) / KS), (s_ + PF) % KS);
            }
            mm_sched_fence();
            const mm_h8 xh = fh[s_ % (PF + 1)], xl = fl[s_ % (PF + 1)];
            const mm_h8 ah0 = afr(std::integral_constant<int, 0>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 0>{});
            const mm_h8 al0 = afr(std::integral_constant<int, 0>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 1>{});
            c0h = mm_mfma(ah0, xh, c0h);
            c0l = mm_mfma(ah0, xl, c0l);
            if constexpr (NSET > 1) {
                const mm_h8 ah1 = afr(std::integral_constant<int, 1>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 0>{});
                const mm_h8 al1 = afr(std::integral_constant<int, 1>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 1>{});
                c1h = mm_mfma(ah1, xh, c1h);
                c1l = mm_mfma(ah1, xl, c1l);
                c0l = mm_mfma(al0, xh, c0l);
                c1l = mm_mfma(al1, xh, c1l);
            } else {
                c0l = mm_mfma(al0, xh, c0l);
            }
            mm_sched_fence();
            if constexpr (kk == KS - 1) {
                float r0[4], r1[4];
                static_for<4>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    r0[r] = fmaf(mm_get(c0l, r), kMmLoInv, mm_get(c0h, r));
                    r1[r] = (NSET > 1) ? fmaf(mm_get(c1l, r), kMmLoInv, mm_get(c1h, r)) : 0.f;
                });
                c0h = mm_zero(); c0l = mm_zero(); c1h = mm_zero(); c1l = mm_zero();
                emit(wv + 4 * ti, r0, r1);
            }
        });
    };
    // the same for a stage with one set of fragments: the wave's two tiles side by side, so that a k-step issues six
    // products on four independent accumulators (three products on two, the third waiting for the second, left the matrix
    // pipe idle a quarter of the time and gave the LDS reads only 48 cycles per step to hide under)
    auto phase_pair = [&](const mm_half* bh, const mm_half* bl, auto ksq, auto strq, auto&& afr, auto&& emit) {
        constexpr int KS = decltype(ksq)::value, STR = decltype(strq)::value, PF = 2;
        auto frag = [&](const mm_half* arr, int T, int kk) -> mm_h8 {
            return *reinterpret_cast<const mm_h8*>(arr + 16 * STR * T + STR * n16 + 32 * kk + 8 * g);
        };
        mm_h8 fh[2][PF + 1], fl[2][PF + 1];
        static_for<PF>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value;
            fh[0][s_] = frag(bh, wv, s_);
            fl[0][s_] = frag(bl, wv, s_);
            fh[1][s_] = frag(bh, wv + 4, s_);
            fl[1][s_] = frag(bl, wv + 4, s_);
        });
        mm_f4 c0h = mm_zero(), c0l = mm_zero(), c1h = mm_zero(), c1l = mm_zero();
        static_for<KS>([&](auto ss) {
            constexpr int kk = decltype(ss)::value;
            if constexpr (kk + PF < KS) {
                fh[0][(kk + PF) % (PF + 1)] = frag(bh, wv, kk + PF);
                fl[0][(kk + PF) % (PF + 1)] = frag(bl, wv, kk + PF);
                fh[1][(kk + PF) % (PF + 1)] = frag(bh, wv + 4, kk + PF);
                fl[1][(kk + PF) % (PF + 1)] = frag(bl, wv + 4, kk + PF);
            }
            mm_sched_fence();
            const mm_h8 ah = afr(std::integral_constant<int, 0>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 0>{});
            const mm_h8 al = afr(std::integral_constant<int, 0>{}, std::integral_constant<int, kk>{}, std::integral_constant<int, 1>{});
            const mm_h8 x0h = fh[0][kk % (PF + 1)], x0l = fl[0][kk % (PF + 1)], x1h = fh[1][kk % (PF + 1)], x1l = fl[1][kk % (PF + 1)];
            c0h = mm_mfma(ah, x0h, c0h);
            c1h = mm_mfma(ah, x1h, c1h);
            c0l = mm_mfma(ah, x0l, c0l);
            c1l = mm_mfma(ah, x1l, c1l);
            c0l = mm_mfma(al, x0h, c0l);
            c1l = mm_mfma(al, x1h, c1l);
            mm_sched_fence();
        });
        float r0[4], r1[4];
        static_for<4>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            r0[r] = fmaf(mm_get(c0l, r), kMmLoInv, mm_get(c0h, r));
            r1[r] = fmaf(mm_get(c1l, r), kMmLoInv, mm_get(c1h, r));
        });
        emit(wv, r0);
        emit(wv + 4, r1);
    };
    auto put = [&](mm_half* dh, mm_half* dl, int at, const float (&r)[4], float sc) {
        mm_half h[4], l[4];
        static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; mm_split(r[e] * sc, h[e], l[e]); });
        mm_put4(dh + at, h);
        mm_put4(dl + at, l);
    };

    if (F > 0) {
        issue(0);
        frame_max(0);
        lds_barrier();
        convert(0, 0);
        if (F > 1) issue(1);
        lds_barrier();
    }
    // Software pipeline over the frames, two barriers per frame:
    //   P(f): stage 1 of frame f (X -> U[f & 1]) and its history | stage 3 of frame f - 1 (V[(f - 1) & 1] -> y) | maxima of frame f + 1
    //   Q(f): stage 2 of frame f (U[f & 1] -> V[f & 1]) and its history | frame f + 1 -> X | loads of frame f + 2
    // U and V alternate between two arrays, so a phase never writes what it (or the other half of the phase) reads.
    float up_f = 1.f, up_p = 1.f, fr_p = 0.f;                       // scale of frame f; scale and offset of frame f - 1
    for (int f = 0; f <= F; ++f) {
        const int b = f & 1;
        mm_half* Uhb = Uh + b * 2 * (kPfUn + kPfPad);
        mm_half* Ulb = Uhb + kPfUn + kPfPad;
        mm_half* Vhb = Vh + b * 2 * (kPfVn + kPfPad);
        mm_half* Vlb = Vhb + kPfVn + kPfPad;
        const mm_half* Uhp = Uh + (b ^ 1) * 2 * (kPfUn + kPfPad);
        const mm_half* Ulp = Uhp + kPfUn + kPfPad;
        const mm_half* Vhp = Vh + (b ^ 1) * 2 * (kPfVn + kPfPad);
        const mm_half* Vlp = Vhp + kPfVn + kPfPad;
        const bool first_of_row = (f % nfr) == 0;
        float ratio = 1.f;
        if (f < F) {
            up_f = up_x;
            ratio = up_p / up_f;                                    // the histories were scaled by 1 / up_p
            // ---- P: stage 1 of frame f
            move(Uhb, Uhp + kPfH, kPfUh, first_of_row, ratio);
            move(Ulb, Ulp + kPfH, kPfUh, first_of_row, ratio);
            mm_h8 a1[kPfKS1][2];
            static_for<kPfKS1>([&](auto kq) {
                constexpr int kk = decltype(kq)::value;
                a1[kk][0] = *reinterpret_cast<const mm_h8*>(T1 + ((kk * 2 + 0) * 64 + lane) * 8);
                a1[kk][1] = *reinterpret_cast<const mm_h8*>(T1 + ((kk * 2 + 1) * 64 + lane) * 8);
            });
            phase_pair(Xh, Xl, std::integral_constant<int, kPfKS1>{}, std::integral_constant<int, 32>{},
                       [&](auto, auto kq, auto hl) { return a1[decltype(kq)::value][decltype(hl)::value]; },
                       [&](int T, const float (&r0)[4]) { put(Uhb, Ulb, kPfUh + 256 * T + 16 * n16 + 4 * g, r0, osc1); });
        }
        if (f >= 1) {
            // ---- P: stage 3 of frame f - 1: both phases of the interpolator, 8 consecutive outputs per lane and tile
            const int fp = f - 1, rowp = (int)blockIdx.x + (fp / nfr) * (int)gridDim.x, jp = fp % nfr;
            float* yr = P.y + (size_t)rowp * ns;
            const bool yal = (reinterpret_cast<uintptr_t>(yr) & 15) == 0 && (ns & 3) == 0;
            mm_h8 a3[2][kPfKS3][2];
            static_for<2 * kPfKS3>([&](auto pq) {
                constexpr int p = decltype(pq)::value / kPfKS3, kk = decltype(pq)::value % kPfKS3;
                a3[p][kk][0] = *reinterpret_cast<const mm_h8*>(T3 + (((p * kPfKS3 + kk) * 2 + 0) * 64 + lane) * 8);
                a3[p][kk][1] = *reinterpret_cast<const mm_h8*>(T3 + (((p * kPfKS3 + kk) * 2 + 1) * 64 + lane) * 8);
            });
            const float osy = osc3 * up_p, add = P.dcg * fr_p;
            const int tbase = jp * kPfCH - kPfLag + 32 * n16 + 8 * g;
            phase(Vhp, Vlp, std::integral_constant<int, kPfKS3>{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{},
                  [&](auto pq, auto kq, auto hl) { return a3[decltype(pq)::value][decltype(kq)::value][decltype(hl)::value]; },
                  [&](int T, const float (&r0)[4], const float (&r1)[4]) {
                      const int t0 = tbase + 512 * T;
                      float o[8];
                      static_for<4>([&](auto rr) {
                          constexpr int r = decltype(rr)::value;
                          o[2 * r] = fmaf(r0[r], osy, add);
                          o[2 * r + 1] = fmaf(r1[r], osy, add);
                      });
                      if (yal && t0 >= 0 && t0 + 8 <= ns) {
                          mm_store4(yr + t0, o[0], o[1], o[2], o[3]);
                          mm_store4(yr + t0 + 4, o[4], o[5], o[6], o[7]);
                      } else {
                          for (int e = 0; e < 8; ++e)
                              if (t0 + e >= 0 && t0 + e < ns) yr[t0 + e] = o[e];
                      }
                  });
        }
        if (f + 1 < F) frame_max(f & 1);                            // of frame f + 1 (its loads were issued a phase ago)
        lds_barrier();
        if (f < F) {
            // ---- Q: stage 2 of frame f
            move(Vhb, Vhp + kPfH, kPfVh, first_of_row, ratio);
            move(Vlb, Vlp + kPfH, kPfVh, first_of_row, ratio);
            phase_pair(Uhb, Ulb, std::integral_constant<int, kPfKS2>{}, std::integral_constant<int, 16>{},
                       [&](auto, auto kq, auto hl) { return decltype(hl)::value ? a2l[decltype(kq)::value] : a2h[decltype(kq)::value]; },
                       [&](int T, const float (&r0)[4]) { put(Vhb, Vlb, kPfVh + 256 * T + 16 * n16 + 4 * g, r0, osc2); });
            up_p = up_f;
            fr_p = fr_x;
        }
        if (f + 1 < F) {
            // ---- Q: frame f + 1 -> X (stage 1 of frame f is done with it), then the loads of frame f + 2
            convert(f + 1, f & 1);
            if (f + 2 < F) issue(f + 2);
        }
        lds_barrier();
    }
}

}  // namespace d4w

using namespace d4w;

extern "C" {

void d4w_fir_poly_shape(int* na, int* ng, int* nb, int* edge) {
    if (na) *na = kPfNa;
    if (ng) *ng = kPfNg;
    if (nb) *nb = kPfNb;
    if (edge) *edge = kPfNa / 2 + 2 * kPfGc + kPfNa / 2 + 6;       // reach of the cascade either side (+ the tap shifts)
}

int d4w_fir_poly_f32(const float* x, int nx, int ns, const float* first, double dc_gain, const float* ta, const float* tg,
                     const float* tb, float* y, void* stream) {
    if (!x || !y || !ta || !tg || !tb || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    const int na = kPfNa, ng = kPfNg, nb = kPfNb;
    PfArgs P;
    P.x = x; P.first = first; P.ta = ta; P.tg = tg; P.tb = tb; P.y = y; P.dcg = (float)dc_gain;
    P.nx = nx; P.ns = ns; P.na = na; P.ng = ng; P.nb = nb;
    const size_t lds = (size_t)kPfHalves * sizeof(mm_half) + 8 * sizeof(float);
    static const int env_wgs = [] { const char* v = getenv("D4W_PF_WGS"); const int n = v ? atoi(v) : 0; return n < 1 ? 2 : (n > 4 ? 4 : n); }();
    const int grid = (int)std::min<long long>(nx, (long long)mm_num_cus() * env_wgs);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)fir_poly_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    D4W_LAUNCH(fir_poly_rows, dim3(grid), dim3(kPfThreads), lds, stream, P);
    return D4W_OK;
}

}  // extern "C"
