// Short-time Fourier transform magnitudes on the matrix cores (gfx950): the STFT of detect.compute_cross_correlogram_spectrocorr
// (reference detect.py:650-709 -> get_sliced_nspectrogram, detect.py:382-392: |librosa.stft(x, n_fft = 160, hop_length = 8)|,
// periodic Hann, center = True with zero padding, bins 14-30 Hz kept).
//
// With 95 % overlap a frame-by-frame FFT transforms every sample twenty times: the register-FFT kernel of rounds 2-3
// (spectral.hip: stft_fat) spends ~35 Gflop per 11 020 x 12 000 file and is bound by VALU issue (1.5 ms).  Only the 13 bins
// between 14 and 30 Hz are kept, and then the transform is a small dense product with a Hankel factor,
//
//      X[b][t] = sum_u  w[u] e^{-2 pi i b u / N} * x[hop t - N/2 + u],      b in the kept bins,  u < N,
//
// i.e. C[b][t] = sum_u A[b][u] B[u][t] with A = the windowed DFT rows (real and imaginary parts: two 16 x K operands that stay
// in registers for the whole launch) and B[u][t] = x[hop t + u - N/2], overlapping windows of the row read straight out of an
// LDS copy -- with hop = 8 lane (t, g) of a fragment takes the 8 consecutive samples 8 t + 32 kk + 8 g .. + 7, ONE 16-byte LDS
// read, and lanes that share a slot share an address.  30 matrix instructions per 16 frames (5 k-steps x re / im x 3 split
// products, mm_common.h) = 0.2 ms of the matrix pipe per file; the kernel reads the row once and writes the kept bins once.
// Operands are split into binary16 hi / lo pairs as in xcorr_mm.hip; the samples of a chunk are scaled by a power of two first.
//
// Eligible calls (d4w_stft_mag_f32 dispatches here): no row maximum asked for (the detector's normalisation cancels),
// <= 16 kept bins, n_fft a multiple of 32 up to 160, hop a multiple of 8 up to 32.  Everything else runs the FFT kernels.
#include <cstdlib>

#include "mm_common.h"

namespace d4w {

constexpr int kSmThreads = 256;
constexpr int kSmTileP = 68;                     // row pitch of a wave's [16 bins][64 frames] output tile: 4 rows apart = 16 banks apart
constexpr int kSmMaxHop = 32, kSmMaxFft = 160;
constexpr int kSmStageMax = 255 * kSmMaxHop + kSmMaxFft + 8;   // samples staged per chunk at most (8328: 33 KiB of hi / lo halves)
// frames per chunk: as many as the staging buffer holds, in units of 256 (a wave works on runs of 64 frames = 4 tiles, 4
// waves) -- 768 at hop 8 (two chunks per 60-s row), 256 at hop 32
__host__ __device__ constexpr int sm_frames_per_chunk(int n_fft, int hop) {
    const int f = ((kSmStageMax - 8 - n_fft) / hop + 1) / 256 * 256;
    return f < 256 ? 256 : f;
}

struct SmArgs {
    const float* x;     // [nx][ns]
    float* S;           // [nx][nbins][nt]
    int nx, ns, n_fft, hop, bin_lo, nbins, nt;
};

// power of two >= a (a >= 0, finite)
__device__ __forceinline__ void sm_pow2_scale(float a, float& up, float& down) {
    int e = 0;
    if (a > 0.f) (void)frexpf(a, &e);
    e = min(max(e, -100), 100);
    up = ldexpf(1.0f, e);
    down = ldexpf(1.0f, -e);
}

template <int KS>                                 // n_fft = 32 KS
__global__ __launch_bounds__(kSmThreads, 3) void stft_mm_rows(SmArgs P) {
    D4W_DYN_LDS(smem_raw);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = mm_uniform(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int N = 32 * KS, hop = P.hop, ns = P.ns, nt = P.nt;
    const int FC = sm_frames_per_chunk(N, hop);
    const int staged = (FC - 1) * hop + N;                         // samples a chunk of frames covers
    const int arr = (staged + 7) & ~7;                             // halves per LDS array
    mm_half* bh = reinterpret_cast<mm_half*>(smem_raw);
    mm_half* bl = bh + arr;
    float* red = reinterpret_cast<float*>(bl + arr);               // [4] chunk maxima of the waves
    float* tiles = red + 4;                                        // [4 waves][16 bins][kSmTileP] magnitudes on their way out

    // ---- the windowed DFT rows of the kept bins: A_re[i][u] = w[u] cos(2 pi (bin_lo + i) u / N), A_im = -w sin, i = n16,
    //      u = 32 kk + 8 g + j; rows beyond the kept bins are zero.  Periodic Hann: w[u] = 0.5 - 0.5 cos(2 pi u / N).
    mm_h8 arh[KS], arl[KS], aih[KS], ail[KS];
    {
        const int b = P.bin_lo + n16;
        const bool live = n16 < P.nbins;
        static_for<KS>([&](auto kq) {
            constexpr int kk = decltype(kq)::value;
            static_for<8>([&](auto jq) {
                constexpr int j = decltype(jq)::value;
                const int u = 32 * kk + 8 * g + j;
                float sw, cw, sn, cs;
                sincospif(2.0f * (float)u / (float)N, &sw, &cw);
                sincospif(2.0f * (float)((b * u) % N) / (float)N, &sn, &cs);
                const float w = live ? 0.5f - 0.5f * cw : 0.f;
                mm_half h, l;
                mm_split(w * cs, h, l);
                mm_set(arh[kk], j, h);
                mm_set(arl[kk], j, l);
                mm_split(-w * sn, h, l);
                mm_set(aih[kk], j, h);
                mm_set(ail[kk], j, l);
            });
        });
    }

    const int nchunk = (nt + FC - 1) / FC;
    const long long total = (long long)P.nx * nchunk;
    constexpr int PER = (kSmStageMax + 4 * kSmThreads - 1) / (4 * kSmThreads);      // groups of 4 samples per lane
    float4 v[PER];                                                 // the chunk's samples on their way into LDS
    auto issue = [&](long long c) {
        const int row = (int)(c / nchunk), f0 = (int)(c - (long long)row * nchunk) * FC;
        const float* xr = P.x + (size_t)row * ns;
        const int s0 = f0 * hop - N / 2;                           // first sample of the chunk (negative: zero padding)
        const bool al = ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) && ((s0 & 3) == 0);
        static_for<PER>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const int i = 4 * (tid + q * kSmThreads);              // offset inside the chunk
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < staged) {
                const int s = s0 + i;
                if (al && s >= 0 && s + 3 < ns) t = *reinterpret_cast<const float4*>(xr + s);
                else {
                    // clamped addresses and selects: the loads of a lane go out together
                    const float a0 = xr[min(max(s, 0), ns - 1)], a1 = xr[min(max(s + 1, 0), ns - 1)];
                    const float a2 = xr[min(max(s + 2, 0), ns - 1)], a3 = xr[min(max(s + 3, 0), ns - 1)];
                    t.x = (s >= 0 && s < ns) ? a0 : 0.f;
                    t.y = (s + 1 >= 0 && s + 1 < ns) ? a1 : 0.f;
                    t.z = (s + 2 >= 0 && s + 2 < ns) ? a2 : 0.f;
                    t.w = (s + 3 >= 0 && s + 3 < ns) ? a3 : 0.f;
                }
            }
            v[q] = t;
        });
    };
    // (Loading chunk c + 1 under the products of chunk c was measured and dropped: the 36 registers it keeps alive across the
    // tile loop cost the third resident workgroup, 0.74 ms against 0.65 -- three workgroups per CU overlap each other's phases.)
    for (long long c = blockIdx.x; c < total; c += gridDim.x) {
        const int row = (int)(c / nchunk), f0 = (int)(c - (long long)row * nchunk) * FC;
        issue(c);
        // ---- stage the loaded chunk: chunk maximum, power-of-two scale, hi / lo halves
        float m = 0.f;
        static_for<PER>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 t = v[q];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
        });
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        lds_barrier();                                             // the previous chunk's fragment reads are done
        if (lane == 0) red[wv] = m;
        lds_barrier();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float up, down;
        sm_pow2_scale(m, up, down);
        static_for<PER>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const int i = 4 * (tid + q * kSmThreads);
            if (i < arr) {
                const float s[4] = {v[q].x * down, v[q].y * down, v[q].z * down, v[q].w * down};
                mm_half h[4], l[4];
                static_for<4>([&](auto ee) { constexpr int e = decltype(ee)::value; mm_split(s[e], h[e], l[e]); });
#ifndef D4W_SM_NO_STAGE
                mm_put4(bh + i, h);
                mm_put4(bl + i, l);
#else
                if (h[0] == (mm_half)12345.f) { mm_put4(bh + i, h); mm_put4(bl + i, l); }
#endif
            }
        });
        lds_barrier();
        // ---- runs of 64 frames (4 tiles of 16) dealt to the 4 waves: C[bin][frame] (+)= A[bin][u] B[u][frame].  The magnitudes
        //      of a run meet in a wave-private LDS tile [16 bins][64 frames], so that a bin leaves as ONE 256-byte run per
        //      wave instead of four 64-byte pieces: the spectrogram's rows are nt = 1 + ns / hop floats long (odd), every
        //      64-byte piece straddled two sectors and the kernel wrote 1.5 x its output (PMC WRITE_SIZE) at 1.2 TB/s
        float* tile = tiles + wv * (16 * kSmTileP);
        for (int ri = 0; ri < FC / 64 / 4; ++ri) {
            const int R = wv + 4 * ri;                              // run index inside the chunk
            if (f0 + 64 * R >= nt) break;
            static_for<4>([&](auto tq) {
                constexpr int q = decltype(tq)::value;
                const int T = 4 * R + q;
                // six accumulators (hi hi, hi lo, lo hi for the real and the imaginary rows): no product waits for the one before it
                mm_f4 crh = mm_zero(), cra = mm_zero(), crb = mm_zero(), cih = mm_zero(), cia = mm_zero(), cib = mm_zero();
                const int base = (16 * T + n16) * hop + 8 * g;      // first sample of this lane's window piece (multiple of 8)
                static_for<KS>([&](auto kq) {
                    constexpr int kk = decltype(kq)::value;
#ifndef D4W_SM_CONST_B
                    const mm_h8 xh = *reinterpret_cast<const mm_h8*>(bh + base + 32 * kk);
                    const mm_h8 xl = *reinterpret_cast<const mm_h8*>(bl + base + 32 * kk);
#else
                    const mm_h8 xh = aih[kk], xl = ail[kk];            // (probe: operands that do not come out of LDS)
#endif
#ifndef D4W_SM_NO_MFMA          // (probe builds, scripts/probe/stream_race2.py: which part of this kernel disturbs the overlap-save FFT kernels?)
                    crh = mm_mfma(arh[kk], xh, crh);
                    cih = mm_mfma(aih[kk], xh, cih);
                    cra = mm_mfma(arh[kk], xl, cra);
                    cia = mm_mfma(aih[kk], xl, cia);
                    crb = mm_mfma(arl[kk], xh, crb);
                    cib = mm_mfma(ail[kk], xh, cib);
#else
                    crh[0] += (float)xh[0] + (float)xl[1];
#endif
                });
                static_for<4>([&](auto rr) {                        // this lane: frame 16 q + n16 of the run, bins 4 g .. 4 g + 3
                    constexpr int r = decltype(rr)::value;
                    const float re = fmaf(mm_get(cra, r) + mm_get(crb, r), kMmLoInv, mm_get(crh, r));
                    const float im = fmaf(mm_get(cia, r) + mm_get(cib, r), kMmLoInv, mm_get(cih, r));
#ifndef D4W_SM_NO_TILE
                    tile[(4 * g + r) * kSmTileP + 16 * q + n16] = mm_sqrt(fmaf(re, re, im * im)) * up;
#else
                    if (re == 12345.f) tile[(4 * g + r) * kSmTileP + 16 * q + n16] = mm_sqrt(fmaf(re, re, im * im)) * up;
#endif
                });
            });
            mm_wave_sync();                                         // the run's tile is complete (one wave wrote it, the same wave reads it)
            const int f = f0 + 64 * R + lane;
            float* srow = P.S + (size_t)row * P.nbins * nt + f;
            if (f < nt) {
                float o[16];
                static_for<16>([&](auto bb) { constexpr int b_ = decltype(bb)::value; o[b_] = tile[b_ * kSmTileP + lane]; });   // all reads, then all stores
                static_for<16>([&](auto bb) {
                    constexpr int b_ = decltype(bb)::value;
                    if (b_ < P.nbins) mm_store1(srow + (size_t)b_ * nt, o[b_]);
                });
            }
            mm_wave_sync();                                         // before the next run overwrites the tile
        }
    }
}

}  // namespace d4w

using namespace d4w;

// 1 when d4w_stft_mag_f32(x, S, rowmax = NULL, ...) runs on the matrix cores for these parameters
extern "C" int d4w_stft_mm_eligible(int n_fft, int hop, int bin_lo, int bin_hi) {
    static const int on = [] { const char* v = getenv("D4W_STFT_MM"); return v ? atoi(v) : 1; }();
    return on && n_fft >= 32 && n_fft <= kSmMaxFft && n_fft % 32 == 0 && hop >= 8 && hop <= kSmMaxHop && hop % 8 == 0 &&
           bin_lo >= 0 && bin_hi >= bin_lo && bin_hi - bin_lo + 1 <= 16 && bin_hi <= n_fft / 2;
}

static int stft_mag_mm_run(const float* x, float* S, int nx, int ns, int n_fft, int hop, int bin_lo, int bin_hi, void* stream);
extern "C" int d4w_stft_mag_mm_f32(const float* x, float* S, int nx, int ns, int n_fft, int hop, int bin_lo, int bin_hi, void* stream) {
    int rc = hazard_enter(1, stream);      // (never beside the overlap-save FFT kernels: d4w_internal.h)
    if (rc) return rc;
    rc = stft_mag_mm_run(x, S, nx, ns, n_fft, hop, bin_lo, bin_hi, stream);
    const int rl = hazard_leave(1, stream);
    return rc ? rc : rl;
}
static int stft_mag_mm_run(const float* x, float* S, int nx, int ns, int n_fft, int hop, int bin_lo, int bin_hi, void* stream) {
    if (!x || !S || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (!d4w_stft_mm_eligible(n_fft, hop, bin_lo, bin_hi))
        return fail(D4W_EINVAL, "n_fft = %d / hop = %d / bins [%d, %d] have no matrix-core STFT (n_fft % 32 == 0 <= 160, hop % 8 == 0 <= 32, <= 16 bins)",
                    n_fft, hop, bin_lo, bin_hi);
    SmArgs P;
    P.x = x; P.S = S; P.nx = nx; P.ns = ns; P.n_fft = n_fft; P.hop = hop; P.bin_lo = bin_lo; P.nbins = bin_hi - bin_lo + 1;
    P.nt = 1 + ns / hop;
    const int FC = sm_frames_per_chunk(n_fft, hop);
    const long long total = (long long)nx * ceil_div(P.nt, FC);
    const int ncu = mm_num_cus();
    const int grid = (int)std::min<long long>(total, (long long)ncu * 3);
    const int staged = (FC - 1) * hop + n_fft, arr = (staged + 7) & ~7;
    const size_t lds = (size_t)2 * arr * sizeof(mm_half) + 4 * sizeof(float) + (size_t)4 * 16 * kSmTileP * sizeof(float);
    switch (n_fft / 32) {
    case 1: D4W_LAUNCH(stft_mm_rows<1>, dim3(grid), dim3(kSmThreads), lds, stream, P); break;
    case 2: D4W_LAUNCH(stft_mm_rows<2>, dim3(grid), dim3(kSmThreads), lds, stream, P); break;
    case 3: D4W_LAUNCH(stft_mm_rows<3>, dim3(grid), dim3(kSmThreads), lds, stream, P); break;
    case 4: D4W_LAUNCH(stft_mm_rows<4>, dim3(grid), dim3(kSmThreads), lds, stream, P); break;
    default: D4W_LAUNCH(stft_mm_rows<5>, dim3(grid), dim3(kSmThreads), lds, stream, P); break;
    }
    return D4W_OK;
}
