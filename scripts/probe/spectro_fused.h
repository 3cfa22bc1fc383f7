// Experiment, not built (round 6): detect.compute_cross_correlogram_spectrocorr (reference detect.py:650-709) of a channel in ONE
// launch -- STFT magnitudes on the matrix cores into an LDS tile [13 bins][1501 frames] (78 KB), np.median by a radix select on the
// tile, the kernel correlation out of the tile: HBM sees the samples in and the correlogram out (54 KB per channel instead of
// 48 + 78 written + 78 x 2..4 read).  Correct: the emulator test below-mentioned and 58 GPU tests green with it as the default,
// 5e-6 row-relative against the three launches on every channel of 3001 x 12 000 and 700 x 6000 blocks.  And SLOWER: 2.2-2.3 ms per
// 11 020 x 12 000 file against 1.24-1.34 ms for stft_mm_rows + row_median + spectro_corr (profiles/r06r/spectro_phases_v*.txt:
// D4W_SF_DBG skips phases -- STFT 0.8 ms, select 0.8-0.9, correlation 0.55 after the rewrite that reads a bin row's window and
// taps in one go (1.2 before), against 0.47 / 0.43 / 0.34 for the separate kernels).  The reason is the tile itself: 78 KB of
// magnitudes + 48 KB of staged samples + tables leave ONE 512-thread workgroup per compute unit, whose phases (stage, matrix
// products, three select sweeps with five barriers each, correlation) run one after the other on eight waves with nothing else
// resident to fill the waits; the separate kernels keep 12-30 waves per compute unit busy and stream at 2-4 TB/s, which costs
// less than the serialisation.  (Two findings on the way: LDS atomics are NOT what bounds the select -- without them, or with
// plain stores, the time is the same; and the key-range-linear histogram bins of this select, instead of the keys' top bits,
// change nothing either.)  Same lesson as the f-k chain through the Infinity Cache and the fat analytic kernel
// (analytic_fat.h): on this part a second pass over HBM by a well-occupied kernel beats a fused kernel that owns a compute unit.
// To try again: paste the first part before the closing brace of namespace d4w in csrc/stft_mm.hip and the second part at the
// file's end; declare d4w_spectro_rows_eligible / d4w_spectro_rows_f32 in include/d4w.h and _lib.py; detect.py calls it when
// eligible (git history of this file's commit has the test: tests/test_emu_spectral.py::test_spectrogram_detector_in_one_launch).
// ---------------------------------------------------------------------------------------------
// The spectrogram detector of one channel in ONE launch (round 6): detect.compute_cross_correlogram_spectrocorr
// (reference detect.py:650-709) is, per channel, STFT magnitudes -> np.median of the kept bins -> kernel correlation along
// time, clipped at zero, over (median x kernel length).  As three launches (stft_mm_rows, spectral.hip row_median,
// spectro_corr) the 13 x 1501 magnitudes of a 60-s channel were written once and read two to four times: 0.86 GB per
// 11 020-channel file each way, 1.24 ms of the stream's 5 ms.  They are 78 KB per channel -- they fit the LDS of a compute
// unit.  One 512-thread workgroup owns a compute unit and walks channels: the row's samples are staged as binary16 hi / lo
// halves (the whole 60-s row at once: 48 KB), the eight waves run the matrix products of stft_mm_rows over three runs of 64
// frames each and leave the magnitudes in an LDS tile [bins][off + frames] (zero columns on both sides are the correlation's
// zero padding), a three-sweep radix select over the tile gives the median (the algorithm of row_median, on LDS), and every
// lane forms four correlation lags from 16-byte LDS reads of the tile (the inner loop of spectro_corr: sums in the same
// order).  HBM traffic per channel: the row in (48 KB), the correlogram out (6 KB).  The next channel's samples are requested
// as soon as the staging registers are free and fly under the matrix phase, the select and the correlation.
// ---------------------------------------------------------------------------------------------
constexpr int kSfThreads = 512, kSfWaves = kSfThreads / 64;
constexpr int kSfPerMax = 6;                                       // 16-byte loads per lane and chunk: <= 12 288 staged samples
constexpr int kSfStageMax = 4 * kSfThreads * kSfPerMax - 8;
constexpr int kSfBins = 2048, kSfBinsPer = kSfBins / kSfThreads;   // histogram bins of the select, per thread
constexpr int kSfTapPitch = 64, kSfPieces = 8;                    // kernel taps in LDS: <= 64 frames per bin row; window pieces kept in registers
constexpr size_t kSfLdsMax = 160 * 1024 - 512;

struct SfArgs {
    const float* x;     // [nx][ns]
    const float* K;     // [nbins][nk]
    float* out;         // [nx][nout]
    int nx, ns, n_fft, hop, bin_lo, nbins, nt, nk, off, nout, zero_ends;
    int FC;             // frames per staged chunk (a multiple of 64)
    int pitch;          // floats per bin row of the LDS tile: off + nt + room for the last window, = 4 (mod 8)
    int dbg;
};

__device__ __forceinline__ unsigned sf_key(float v) {              // order-preserving keys (spectral.hip med_key)
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sf_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

template <int KS>                                 // n_fft = 32 KS
__global__ __launch_bounds__(kSfThreads) void spectro_rows(SfArgs P) {
    D4W_DYN_LDS(smem_raw);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = mm_uniform(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int N = 32 * KS, hop = P.hop, ns = P.ns, nt = P.nt, FC = P.FC, pitch = P.pitch, nbins = P.nbins, off = P.off;
    const int staged = (FC - 1) * hop + N;
    const int arr = (staged + 7) & ~7;
    float* Sl = reinterpret_cast<float*>(smem_raw);                // [nbins][pitch] magnitudes, frame t at column off + t
    mm_half* bh = reinterpret_cast<mm_half*>(Sl + nbins * pitch);
    mm_half* bl = bh + arr;
    unsigned* hist = reinterpret_cast<unsigned*>(bl + arr);        // [kSfBins]
    unsigned* wtot = hist + kSfBins;                               // [kSfWaves]
    unsigned* sc = wtot + kSfWaves;                                // bin, k, next, above
    float* red = reinterpret_cast<float*>(sc + 8);                 // [kSfWaves] chunk maxima of the waves
    float* rhi = red + kSfWaves;                                   // [kSfWaves] (red / rhi: the waves' smallest / largest magnitude of a row)
    float* Kl = rhi + kSfWaves;                                    // [nbins][kSfTapPitch] the kernel's taps (16-byte aligned rows)
    mm_h8* Al = reinterpret_cast<mm_h8*>(Kl + nbins * kSfTapPitch); // [4 KS][64 lanes] the DFT-row fragments
    for (int i = tid; i < nbins * kSfTapPitch; i += kSfThreads) {
        const int f = i / kSfTapPitch, j = i - f * kSfTapPitch;
        Kl[i] = (j < P.nk) ? P.K[(size_t)f * P.nk + j] : 0.f;
    }

    // ---- the windowed DFT rows of the kept bins (as in stft_mm_rows)
    mm_h8 arh[KS], arl[KS], aih[KS], ail[KS];
    {
        const int b = P.bin_lo + n16;
        const bool live = n16 < nbins;
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
    // ... parked in LDS (one copy: they depend on the lane only) and fetched at the top of every chunk's matrix phase, so that they
    // do not occupy 80 registers through the select and the correlation (the first build spilled them)
    if (wv == 0)
        static_for<KS>([&](auto kq) {
            constexpr int kk = decltype(kq)::value;
            Al[(4 * kk + 0) * 64 + lane] = arh[kk];
            Al[(4 * kk + 1) * 64 + lane] = arl[kk];
            Al[(4 * kk + 2) * 64 + lane] = aih[kk];
            Al[(4 * kk + 3) * 64 + lane] = ail[kk];
        });
    // the tile's columns outside [off, off + nt) are the zero padding of the correlation: written once, never touched again
    for (int i = tid; i < nbins * pitch; i += kSfThreads) Sl[i] = 0.f;

    float4 v[kSfPerMax];                                           // a chunk's samples on their way into LDS
    auto issue = [&](int row, int f0) {
        const float* xr = P.x + (size_t)row * ns;
        const int s0 = f0 * hop - N / 2;                           // first sample of the chunk (negative: zero padding)
        const bool al = ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) && ((s0 & 3) == 0);
        static_for<kSfPerMax>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const int i = 4 * (tid + q * kSfThreads);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < staged) {
                const int s = s0 + i;
                if (al && s >= 0 && s + 3 < ns) t = *reinterpret_cast<const float4*>(xr + s);
                else {
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

    const int nchunk = (nt + FC - 1) / FC;
    const int n = nbins * nt;                                      // values the median is taken over
    int row = (int)blockIdx.x;
    if (row < P.nx) issue(row, 0);
    for (; row < P.nx; row += (int)gridDim.x) {
        // ================= STFT magnitudes of the row into the tile =================
        float vmin = INFINITY, vmax = 0.f;                         // of the magnitudes this lane stores
        bool vnan = false;
        for (int ci = 0; ci < nchunk; ++ci) {
            const int f0 = ci * FC;
            float m = 0.f;
            static_for<kSfPerMax>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const float4 t = v[q];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
            });
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            lds_barrier();                                         // the previous chunk's fragment reads / the previous row's correlation are done
            if (lane == 0) red[wv] = m;
            lds_barrier();
            m = 0.f;
            for (int w = 0; w < kSfWaves; ++w) m = fmaxf(m, red[w]);
            float up, down;
            sm_pow2_scale(m, up, down);
            static_for<kSfPerMax>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                const int i = 4 * (tid + q * kSfThreads);
                if (i < arr) {
                    const float s[4] = {v[q].x * down, v[q].y * down, v[q].z * down, v[q].w * down};
                    mm_split_put4(s, bh + i, bl + i);
                }
            });
            // the staging registers are free: the next chunk (or the next channel's first one) flies under everything below
            if (ci + 1 < nchunk) issue(row, f0 + FC);
            else if (row + (int)gridDim.x < P.nx) issue(row + (int)gridDim.x, 0);
            lds_barrier();
            static_for<KS>([&](auto kq) {
                constexpr int kk = decltype(kq)::value;
                arh[kk] = Al[(4 * kk + 0) * 64 + lane];
                arl[kk] = Al[(4 * kk + 1) * 64 + lane];
                aih[kk] = Al[(4 * kk + 2) * 64 + lane];
                ail[kk] = Al[(4 * kk + 3) * 64 + lane];
            });
            if (!(P.dbg & 4))
            for (int R = wv; R < FC / 64; R += kSfWaves) {          // runs of 64 frames dealt to the waves
                if (f0 + 64 * R >= nt) break;
                static_for<4>([&](auto tq) {
                    constexpr int q = decltype(tq)::value;
                    const int T = 4 * R + q;
                    mm_f4 crh = mm_zero(), cra = mm_zero(), crb = mm_zero(), cih = mm_zero(), cia = mm_zero(), cib = mm_zero();
                    const int base = (16 * T + n16) * hop + 8 * g;
                    static_for<KS>([&](auto kq) {
                        constexpr int kk = decltype(kq)::value;
                        const mm_h8 xh = *reinterpret_cast<const mm_h8*>(bh + base + 32 * kk);
                        const mm_h8 xl = *reinterpret_cast<const mm_h8*>(bl + base + 32 * kk);
                        crh = mm_mfma(arh[kk], xh, crh);
                        cih = mm_mfma(aih[kk], xh, cih);
                        cra = mm_mfma(arh[kk], xl, cra);
                        cia = mm_mfma(aih[kk], xl, cia);
                        crb = mm_mfma(arl[kk], xh, crb);
                        cib = mm_mfma(ail[kk], xh, cib);
                    });
                    const int frame = f0 + 64 * R + 16 * q + n16;  // this lane: bins 4 g .. 4 g + 3 of this frame
                    static_for<4>([&](auto rr) {
                        constexpr int r = decltype(rr)::value;
                        const float re = fmaf(mm_get(cra, r) + mm_get(crb, r), kMmLoInv, mm_get(crh, r));
                        const float im = fmaf(mm_get(cia, r) + mm_get(cib, r), kMmLoInv, mm_get(cih, r));
                        if (4 * g + r < nbins && frame < nt) {
                            const float mag = mm_sqrt(fmaf(re, re, im * im)) * up;
                            Sl[(4 * g + r) * pitch + off + frame] = mag;
                            vmin = fminf(vmin, mag);
                            vmax = fmaxf(vmax, mag);
                            vnan |= (mag != mag);
                        }
                    });
                });
            }
        }
        // the row's smallest / largest magnitude (every lane kept those of the values it stored) and whether any was NaN
        {
            float lo = vmin, hi = vmax;
            for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
            const unsigned long long nb = __ballot(vnan);
            if (lane == 0) { red[wv] = lo; rhi[wv] = hi; wtot[wv] = nb ? 1u : 0u; }
        }
        lds_barrier();                                             // the row's magnitudes are complete
        // ================= np.median of the tile's n values =================
        // Radix select on order-preserving keys, with the digits taken from key - (smallest key) instead of from the key's top
        // bits: a row of magnitudes shares sign and exponent, so the top 11 bits of its keys fall into a handful of histogram
        // bins -- dozens of lanes of a wave adding to one LDS word, serialised (row_median counts four hot bins in registers and
        // is still bound by that: 24 k cycles per row).  Spread linearly over [smallest, largest] in 2048 bins, an octave of
        // values covers > 100 bins and the LDS atomics run nearly conflict-free.  <= 3 sweeps (11 + 11 + 10 bits).
        float medv;
        {
            float lo = red[0], hi = rhi[0];
            unsigned anynan = wtot[0];
            for (int w = 1; w < kSfWaves; ++w) { lo = fminf(lo, red[w]); hi = fmaxf(hi, rhi[w]); anynan |= wtot[w]; }
            unsigned base = sf_key(lo), range = sf_key(hi) - base;     // keys in [base, base + range]
            unsigned kk = (unsigned)((n - 1) / 2);
            medv = lo;                                                 // all values equal (a silent channel: all zero)
            if (anynan) medv = __uint_as_float(0x7FC00000u);           // np.median of a row with a NaN is NaN
            else if (range != 0u && !(P.dbg & 1)) {
                int shift = max(0, (32 - __builtin_clz(range)) - 11);    // (range >> shift) < 2048
                unsigned above = 0xFFFFFFFFu, bin = 0u, h_sel = 0u;
                for (int pass = 0; pass < 3; ++pass) {
                    const bool last = (shift == 0);
                    for (int i = tid; i < kSfBins; i += kSfThreads) hist[i] = 0u;
                    if (tid == 0) { sc[2] = 0xFFFFFFFFu; sc[3] = 0xFFFFFFFFu; }
                    lds_barrier();
                    for (int t = tid; t < nt; t += kSfThreads) {
                        // one frame's bins at once: all LDS reads, then all bin indices (vector ALU only), then all atomics -- an
                        // atomic between two reads makes the wave wait for it before it may look at the second read (LDS operations
                        // retire in order and the counter cannot tell them apart across a branch): 0.8 ms of waiting per file
                        float col[16];
                        unsigned idx[16];
                        static_for<16>([&](auto bb) { constexpr int b = decltype(bb)::value; col[b] = (b < nbins) ? Sl[b * pitch + off + t] : 0.f; });
                        static_for<16>([&](auto bb) {
                            constexpr int b = decltype(bb)::value;
                            const unsigned k = sf_key(col[b]), d = k - base;
                            const bool in = (b < nbins) && k >= base && d <= range;
                            idx[b] = in ? (d >> shift) : 0xFFFFFFFFu;
                            above = (last && (b < nbins) && k >= base && d > range) ? min(above, k) : above;
                        });
                        static_for<16>([&](auto bb) {
                            constexpr int b = decltype(bb)::value;
                            if (idx[b] != 0xFFFFFFFFu) { if (P.dbg & 16) hist[idx[b]] = 1u; else if (!(P.dbg & 8)) atomicAdd(&hist[idx[b]], 1u); }
                        });
                    }
                    lds_barrier();
                    // the bin holding rank kk: kSfBinsPer bins per thread, inclusive scan of the threads' sums
                    unsigned h[kSfBinsPer], sum = 0u;
#pragma unroll
                    for (int t4 = 0; t4 < kSfBinsPer; ++t4) { h[t4] = hist[kSfBinsPer * tid + t4]; sum += h[t4]; }
                    unsigned incl = sum;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned upv = __shfl_up(incl, o);
                        if (lane >= o) incl += upv;
                    }
                    if (lane == 63) wtot[wv] = incl;
                    lds_barrier();
                    unsigned before = 0u;
                    for (int w = 0; w < wv; ++w) before += wtot[w];
                    unsigned run = incl + before - sum;            // keys in the bins before this thread's
#pragma unroll
                    for (int t4 = 0; t4 < kSfBinsPer; ++t4) {
                        if (kk >= run && kk < run + h[t4]) {       // exactly one (thread, bin)
                            sc[0] = (unsigned)(kSfBinsPer * tid + t4);
                            sc[1] = kk - run;
                        }
                        run += h[t4];
                    }
                    lds_barrier();
                    bin = sc[0];
                    kk = sc[1];
                    if (last) {
                        h_sel = hist[bin];
                        // the next key above the selected one: the next non-empty bin of this sweep, else the smallest key beyond its window
#pragma unroll
                        for (int t4 = 0; t4 < kSfBinsPer; ++t4) {
                            const unsigned bq = (unsigned)(kSfBinsPer * tid + t4);
                            if (bq > bin && h[t4]) atomicMin(&sc[2], bq);
                        }
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) above = min(above, (unsigned)__shfl_xor((int)above, o));
                        if (lane == 0) atomicMin(&sc[3], above);
                        lds_barrier();
                        const unsigned a = base + bin;
                        unsigned b = a;
                        if ((n & 1) == 0 && kk + 1 >= h_sel) b = (sc[2] != 0xFFFFFFFFu) ? base + sc[2] : sc[3];
                        medv = (n & 1) ? sf_unkey(a) : 0.5f * (sf_unkey(a) + sf_unkey(b));
                        break;
                    }
                    base += bin << shift;
                    range = min(range - (bin << shift), (1u << shift) - 1u);
                    shift = max(0, shift - 11);
                    lds_barrier();                                 // sc / hist / wtot are rewritten by the next sweep
                }
            }
        }

        const float den = medv * (float)P.nk;
        // ================= kernel correlation along time, four lags per lane (spectro_corr's inner loop on the tile) =================
        float* orow = P.out + (size_t)row * P.nout;
        const int np = (P.nk + 3 + 3) / 4;                          // 16-byte pieces of a lane's window
        for (int t0 = (P.dbg & 2) ? P.nout : 0; t0 < P.nout; t0 += 4 * kSfThreads) {
            const int t = t0 + 4 * tid;
            if (t < P.nout) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if (np <= kSfPieces) {
                    for (int f = 0; f < nbins; ++f) {
                        // the row's window pieces and its taps (LDS, one address per wave) in registers first, then the products in
                        // spectro_corr's order: tap j = 4 c + e - q of lag q meets sample e of piece c
                        const float4* sr4 = reinterpret_cast<const float4*>(Sl + f * pitch + t);
                        const float4* kr4 = reinterpret_cast<const float4*>(Kl + f * kSfTapPitch);
                        float4 kprev = make_float4(0.f, 0.f, 0.f, 0.f);
                        static_for<kSfPieces>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            if (c < np) {
                                const float4 w = sr4[c], kc = kr4[c];
                                const float vv[4] = {w.x, w.y, w.z, w.w};
                                // taps 4 c - 3 .. 4 c - 1 are values 1 .. 3 of the previous tap piece, 4 c .. 4 c + 3 this one
                                const float tp[7] = {kprev.y, kprev.z, kprev.w, kc.x, kc.y, kc.z, kc.w};
                                static_for<7>([&](auto dd) {
                                    constexpr int d = decltype(dd)::value;
                                    constexpr int j = 4 * c - 3 + d;
                                    if (j >= 0 && j < P.nk) {      // taps outside [0, nk) are skipped, not multiplied by zero (wave-uniform)
#pragma unroll
                                        for (int qq = 0; qq < 4; ++qq) {
                                            const int e = d - 3 + qq;
                                            if (e >= 0 && e < 4) acc[qq] = fmaf(vv[e], tp[d], acc[qq]);
                                        }
                                    }
                                });
                                kprev = kc;
                            }
                        });
                    }
                } else {
                    for (int f = 0; f < nbins; ++f) {
                        const float* kr = Kl + f * kSfTapPitch;
                        const float4* sr4 = reinterpret_cast<const float4*>(Sl + f * pitch + t);
                        for (int c = 0; 4 * c < P.nk + 3; ++c) {
                            const float4 v4 = sr4[c];
                            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                            for (int d = 0; d < 7; ++d) {
                                const int j = 4 * c - 3 + d;
                                if (j >= 0 && j < P.nk) {
                                    const float tap = kr[j];
#pragma unroll
                                    for (int qq = 0; qq < 4; ++qq) {
                                        const int e = d - 3 + qq;
                                        if (e >= 0 && e < 4) acc[qq] = fmaf(vv[e], tap, acc[qq]);
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int tq = t + qq;
                    if (tq < P.nout) {
                        float o = acc[qq] / den;
                        if (P.zero_ends && (tq == 0 || tq == P.nout - 1)) o = 0.f;
                        if (o < 0.f) o = 0.f;                        // NaN (0/0 on an all-zero row) passes through
                        mm_store1(orow + tq, o);
                    }
                }
            }
        }
    }
}


// ---- second part: geometry and C entry points
// geometry of the fused spectrogram detector for these parameters: frames per staged chunk and the tile's row pitch; 0 when
// the call has no fused form
static int spectro_rows_geometry(int ns, int n_fft, int hop, int bin_lo, int bin_hi, int nk, int off, int nout, int* FC_out, int* pitch_out, size_t* lds_out) {
    static const int on = [] { const char* v = getenv("D4W_SPECTRO_FUSED"); return v ? atoi(v) : 1; }();
    if (!on || !d4w_stft_mm_eligible(n_fft, hop, bin_lo, bin_hi) || ns < 1 || nk < 1 || off < 0 || nout < 1) return 0;
    const int nbins = bin_hi - bin_lo + 1, nt = 1 + ns / hop;
    if (off > nk || nout > nt + nk || nk > kSfTapPitch) return 0;
    // columns: frames at [off, off + nt); a lane's last window piece ends at round4(nout) + round4(nk + 3) + 3
    int pitch = std::max(off + nt, ((nout + 3) & ~3) + ((nk + 6) & ~3)) + 4;
    pitch = (pitch + 3) & ~3;
    if (pitch % 8 != 4) pitch += 4;                                // rows 4 apart = 16 banks apart (the magnitude stores)
    int FC = ((kSfStageMax - n_fft) / hop + 1) / 64 * 64;
    FC = std::min(FC, (nt + 63) / 64 * 64);
    for (; FC >= 64; FC -= 64) {
        const int staged = (FC - 1) * hop + n_fft, arr = (staged + 7) & ~7;
        const size_t lds = (size_t)nbins * pitch * sizeof(float) + (size_t)2 * arr * sizeof(mm_half) +
                           (size_t)(kSfBins + kSfWaves + 8) * sizeof(unsigned) + 2 * kSfWaves * sizeof(float) +
                           (size_t)nbins * kSfTapPitch * sizeof(float) + (size_t)4 * (n_fft / 32) * 64 * sizeof(mm_h8);
        if (staged <= kSfStageMax && lds <= kSfLdsMax) {
            if (FC_out) *FC_out = FC;
            if (pitch_out) *pitch_out = pitch;
            if (lds_out) *lds_out = lds;
            return 1;
        }
    }
    return 0;
}

extern "C" int d4w_spectro_rows_eligible(int ns, int n_fft, int hop, int bin_lo, int bin_hi, int nk, int off, int nout) {
    return spectro_rows_geometry(ns, n_fft, hop, bin_lo, bin_hi, nk, off, nout, nullptr, nullptr, nullptr);
}

extern "C" int d4w_spectro_rows_f32(const float* x, int nx, int ns, int n_fft, int hop, int bin_lo, int bin_hi, const float* K,
                                    int nk, int off, int nout, int zero_ends, float* out, void* stream) {
    if (!x || !K || !out || nx < 1) return fail(D4W_EINVAL, "bad argument");
    SfArgs P;
    size_t lds = 0;
    if (!spectro_rows_geometry(ns, n_fft, hop, bin_lo, bin_hi, nk, off, nout, &P.FC, &P.pitch, &lds))
        return fail(D4W_EINVAL, "no fused spectrogram detector for ns = %d, n_fft = %d, hop = %d, bins [%d, %d], kernel of %d frames "
                    "(d4w_spectro_rows_eligible)", ns, n_fft, hop, bin_lo, bin_hi, nk);
    P.x = x; P.K = K; P.out = out; P.nx = nx; P.ns = ns; P.n_fft = n_fft; P.hop = hop; P.bin_lo = bin_lo;
    P.nbins = bin_hi - bin_lo + 1; P.nt = 1 + ns / hop; P.nk = nk; P.off = off; P.nout = nout; P.zero_ends = zero_ends;
    { const char* v = getenv("D4W_SF_DBG"); P.dbg = v ? atoi(v) : 0; }
    int rc = hazard_enter(1, stream);      // (the STFT's family: d4w_internal.h)
    if (rc) return rc;
    const int grid = std::min(nx, mm_num_cus());
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        D4W_LAUNCH(kern, dim3(grid), dim3(kSfThreads), lds, stream, P);
        return D4W_OK;
    };
    switch (n_fft / 32) {
    case 1: rc = launch(spectro_rows<1>); break;
    case 2: rc = launch(spectro_rows<2>); break;
    case 3: rc = launch(spectro_rows<3>); break;
    case 4: rc = launch(spectro_rows<4>); break;
    default: rc = launch(spectro_rows<5>); break;
    }
    const int rl = hazard_leave(1, stream);
    return rc ? rc : rl;
}

/* ---- the emulator test that pinned it (tests/test_emu_spectral.py), as a diff:
diff --git a/tests/test_emu_spectral.py b/tests/test_emu_spectral.py
index 6aa0d27..ea4bbf4 100644
--- a/tests/test_emu_spectral.py
+++ b/tests/test_emu_spectral.py
@@ -238,6 +238,59 @@ def test_xcorr2d_and_spectrocorr_golden(emu, golden):
         assert rel(out[c], orc.xcorr2d(S3[c], K3)) < TOL
 
 
+@pytest.mark.parametrize("n_fft,hop,ns,lo,hi,nk", [(160, 8, 12000, 12, 24, 21), (160, 8, 4999, 0, 15, 20), (128, 16, 3001, 60, 64, 7),
+                                                   (96, 24, 2500, 3, 3, 1), (32, 8, 700, 1, 16, 30), (160, 8, 16000, 11, 23, 20),
+                                                   (64, 8, 100, 1, 9, 5)])
+def test_spectrogram_detector_in_one_launch(emu, golden, n_fft, hop, ns, lo, hi, nk):
+    """d4w_spectro_rows_f32 (STFT magnitudes -> median -> kernel correlation with a channel's bins held in LDS) against the three
+    launches it replaces and against the oracle's chain: one chunk per row and several (16 000 samples: 2001 frames in four chunks), odd row
+    lengths, 1-16 bins, kernels of 1-30 frames, both correlation modes, more channels than workgroups, a silent channel
+    (0 / 0 = NaN as in the reference) and a channel with an offset."""
+    if hi > n_fft // 2:
+        hi = n_fft // 2
+    nt = emu.d4w_stft_frames(ns, hop)
+    rng = np.random.default_rng(ns + nk)
+    nx = 5
+    x = rng.standard_normal((nx, ns)) * np.array([[1.0], [300.0], [1e-3], [0.0], [2.0]]) + np.array([[0.0], [50.0], [0.0], [0.0], [0.1]])
+    xf = np.ascontiguousarray(x, dtype=np.float32)
+    K = np.ascontiguousarray(rng.standard_normal((hi - lo + 1, nk)), dtype=np.float32)
+    for off, nout, ze in ((nk // 2, nt, 0), (0, nt - nk + 1, 1)):
+        assert emu.d4w_spectro_rows_eligible(ns, n_fft, hop, lo, hi, nk, off, nout) == 1
+        out = np.full((nx, nout), -7.0, dtype=np.float32)
+        ok(emu, emu.d4w_spectro_rows_f32(vp(xf), nx, ns, n_fft, hop, lo, hi, vp(K), nk, off, nout, ze, vp(out), None))
+        S = np.empty((nx, hi - lo + 1, nt), dtype=np.float32)
+        ok(emu, emu.d4w_stft_mag_f32(vp(xf), vp(S), None, nx, ns, n_fft, hop, lo, hi, None))
+        ref3 = spectrocorr(emu, S, K, off, nout, zero_ends=ze)
+        inner = slice(1, -1) if ze else slice(None)
+        assert np.all(np.isnan(out[3, inner])) and np.all(np.isnan(ref3[3, inner])) and np.array_equal(out[3], ref3[3], equal_nan=True)
+        live = [0, 1, 2, 4]
+        assert rel(out[live], ref3[live]) < 2e-6                      # (the two forms scale a row's samples by different powers of two)
+        for c in live:
+            Sc = np.abs(orc.librosa_stft(xf[c].astype(np.float64), n_fft=n_fft, hop_length=hop))[lo:hi + 1]
+            raw = np.array([sum(np.dot(Sc[f, max(t - off, 0):min(t - off + nk, nt)],
+                                       K[f, max(off - t, 0):max(off - t, 0) + (min(t - off + nk, nt) - max(t - off, 0))].astype(np.float64))
+                                for f in range(hi - lo + 1)) for t in range(nout)])
+            want = np.maximum(raw, 0) / (np.median(Sc) * nk)
+            if ze:
+                want[0] = want[-1] = 0
+            assert np.max(np.abs(out[c] - want)) < 2e-5 * max(np.max(np.abs(want)), 1e-30), (c, off)
+    # the golden detector block through the fused call
+    g = golden("detect_12x2000.npz")
+    xg, fs = np.ascontiguousarray(g["x"], dtype=np.float32), float(g["fs"])
+    ff = np.linspace(0, fs / 2, 81)
+    keep = np.where((ff >= g["nspec_ff"][0] - 1e-9) & (ff <= g["nspec_ff"][-1] + 1e-9))[0]
+    Kg = np.ascontiguousarray(g["ker"], dtype=np.float32)
+    ntg = emu.d4w_stft_frames(xg.shape[1], 8)
+    outg = np.empty((xg.shape[0], ntg), dtype=np.float32)
+    ok(emu, emu.d4w_spectro_rows_f32(vp(xg), xg.shape[0], xg.shape[1], 160, 8, int(keep[0]), int(keep[-1]), vp(Kg), Kg.shape[1],
+                                     Kg.shape[1] // 2, ntg, 0, vp(outg), None))
+    assert rel(outg, g["spectrocorr"]) < TOL
+    # calls without a fused form
+    assert emu.d4w_spectro_rows_eligible(12000, 256, 8, 0, 5, 20, 10, 1501) == 0        # no matrix-core STFT
+    assert emu.d4w_spectro_rows_eligible(120000, 160, 8, 11, 23, 20, 10, 15001) == 0    # 13 x 15 001 magnitudes do not fit
+    assert emu.d4w_spectro_rows_f32(vp(xg), 1, 2000, 256, 8, 0, 5, vp(Kg), 3, 1, 251, 0, vp(outg), None) != 0
+
+
 def test_xcorr_valid_mode_golden(emu, golden):
     """detect.xcorr (detect.py:605-647): valid lags, first / last value forced to zero."""
     g = golden("detect_12x2000.npz")

*/
