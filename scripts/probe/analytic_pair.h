// Experiment, not built (round 6): the analytic signal of a row (scipy.signal.hilbert; spectral.hip analytic_rows) with the
// butterflies in PACKED registers -- the row's two half-length sub-transforms on the halves of 64-bit register pairs
// (fft_pair.h), one 16-byte LDS word per index.  Correct (the emulator suite and 57 GPU tests green with it as the default,
// 2e-7 .. 1e-6 against SciPy) and SLOWER: 0.65-0.66 ms per 11 020 x 12 000 block against 0.57-0.59 ms for analytic_rows on the
// same lease (profiles/r06q/envelope_ab.txt; 256 / 320 / 384 / 512 threads per row: 0.63 / 0.77 / 0.74 / 0.64).  Why
// (profiles/r06q/pmc_envelope.txt, valu_rate.txt): the vector instructions per block fall from 269 M to 189 M, not by half
// (index arithmetic, twiddle powers, the untangle and the sweeps are scalar code either way), and on this part a v_pk_*_f32
// costs 1.57 x the issue time of a scalar v_fma / v_add (5.2 against 3.3 cycles per wave instruction at four waves per SIMD;
// scripts/ubench/valu_rate.hip) -- two values per instruction buy 27 % at most, which 124 registers per lane (two 512-thread
// workgroups per compute unit instead of three) and five busy waves per stage more than give back.  The matrix cores do not
// help either: a radix-r stage as a split-binary16 product is 3 x 8 r flop per point plus the same conversions and twiddle
// multiplications on the vector unit that the butterfly costs today.
// To try again: paste this before row_var in csrc/spectral.hip, #include "fft_pair.h", and launch it from d4w_analytic_f32
// for ns % 4 == 0, ns / 4 = 2^a 3^b 5^c, modes 0, 1, 2, 4:
//     if (mode != kAnIfreq && ns % 4 == 0 && ns >= 64 && h->dev.bs_L == 0 && !h->generic) {      // h = row_fft_get(ns / 2)
//         const RowFftHost* hh = nullptr;  rc = row_fft_get(ns / 4, &hh);  if (rc) return rc;
//         if (hh->dev.bs_L == 0 && !hh->generic) {
//             PairFftDev P{hh->dev.ax, hh->dev.pos, h->dev.wpack, h->dev.wfull};
//             const size_t plds = (size_t)(ns / 4) * sizeof(float4) + (size_t)(kTwLo + hh->dev.ax.nhi) * sizeof(float2);
//             sp_allow_lds(analytic_rows_pair, plds);
//             D4W_LAUNCH(analytic_rows_pair, dim3(nx), dim3(an_threads), plds, stream, P, x, ns, y, mode, var);
//             return D4W_OK;
//         }
//     }
// ---------------------------------------------------------------------------------------------
// The same operator with the butterflies in PACKED registers (round 6).  analytic_rows is bound by vector
// issue (24 k wave instructions per 12 000-sample row); the matrix cores do not help a transform whose
// stages are 10 points wide (a radix-r stage as a split-binary16 product costs 3 * 8 r flops per point
// plus the same conversions and twiddles on the vector unit that the butterfly costs today).  What
// halves the vector work is v_pk_*_f32 on two INDEPENDENT transforms riding the halves of 64-bit
// register pairs (fft_pair.h) -- and one row holds two: after the first radix-2 step of the M-point
// transform (M = ns / 2 packed complex samples),
//     a[n] = z[n] + z[n + M/2],   b[n] = (z[n] - z[n + M/2]) W_M^n,      Z[2k] = DFT(a)[k],  Z[2k + 1] = DFT(b)[k],
// a and b are two M/2-point transforms with the same twiddles at every stage.  The tile keeps them as
// one 16-byte word per index, (re a, re b, im a, im b): one ds_read_b128 per butterfly input lands in
// two aligned register pairs, every butterfly add / multiply is one packed instruction for both, and
// the tile is as large as before (three workgroups per CU).  The radix-2 step rides the global loads,
// its inverse the output sweep; the real-spectrum untangle in between works on single values (its
// partner of bin f is M - f: the same half, another position).
// Needs ns % 4 == 0 and ns / 4 = 2^a 3^b 5^c; modes 0, 1, 2, 4 (mode 3 looks at neighbouring samples and
// stays on analytic_rows).
// ---------------------------------------------------------------------------------------------
struct PairFftDev {
    AxisDesc ax;          // the M/2-point axis
    const int* pos;       // [M/2] frequency -> position of that axis
    const float2* wpack;  // [M] exp(-2 pi i f / (2 M))
    const float2* wfull;  // [M] exp(-2 pi i n / M)
};

template <int R, bool INV>
__device__ __forceinline__ void lds_stage_pair(float4* buf, int L, int Ls, const TwLds tw, int tid, int nthr) {
    const int m = Ls / R;          // butterflies per sub-transform group
    const int nbf = L / R;
    const int twstep = L / Ls;
    const FDiv d_m(m);
    for (int w = tid; w < nbf; w += nthr) {
        const int g = d_m.div(w);
        const int j = w - d4w_mul24(g, m);
        float4* p = buf + d4w_mul24(g, Ls) + j;
        c2 x[R];
        static_for<R>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 v = p[q * m];
            x[q] = c2{v2_make(v.x, v.y), v2_make(v.z, v.w)};
        });
        if (m > 1) {
            float2 pw[R];
            pw[1] = tw.get(d4w_mul24(j, twstep));
            static_for<R - 2>([&](auto qq) {
                constexpr int q = decltype(qq)::value + 2;
                pw[q] = c_mul(pw[q / 2], pw[q - q / 2]);
            });
            if (!INV) {
                dftp<R>(x);
                static_for<R - 1>([&](auto qq) {
                    constexpr int q = decltype(qq)::value + 1;
                    x[q] = c2_mulw(x[q], pw[q]);
                });
            } else {
                static_for<R - 1>([&](auto qq) {
                    constexpr int q = decltype(qq)::value + 1;
                    x[q] = c2_mulwc(x[q], pw[q]);
                });
                idftp<R>(x);
            }
        } else {
            if (!INV) dftp<R>(x); else idftp<R>(x);
        }
        static_for<R>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            p[q * m] = make_float4(v2_x(x[q].re), v2_y(x[q].re), v2_x(x[q].im), v2_y(x[q].im));
        });
    }
}

template <bool INV>
__device__ __forceinline__ void lds_fft_pair(float4* buf, const AxisDesc& ax, const TwLds tw, int tid, int nthr) {
    auto stage = [&](int R, int Ls) {
        switch (R) {
#define D4W_CASE(RR) case RR: lds_stage_pair<RR, INV>(buf, ax.L, Ls, tw, tid, nthr); break;
            D4W_FOR_EACH_FAST_RADIX(D4W_CASE)
#undef D4W_CASE
            default: break;
        }
        lds_barrier();
    };
    if (!INV) {
        int Ls = ax.L;
        for (int s = 0; s < ax.nstage; ++s) { stage(ax.radix[s], Ls); Ls /= ax.radix[s]; }
    } else {
        int Ls = 1;
        for (int s = ax.nstage - 1; s >= 0; --s) { Ls *= ax.radix[s]; stage(ax.radix[s], Ls); }
    }
}

__global__ __launch_bounds__(kAnMaxThreads) void analytic_rows_pair(PairFftDev F, const float* __restrict__ x, int ns,
                                                                 float* __restrict__ y, int mode,
                                                                 const float* __restrict__ var) {
    D4W_DYN_LDS(smem_raw);
    float4* tile = reinterpret_cast<float4*>(smem_raw);
    float* tf = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int H = F.ax.L, M = 2 * H;
    const TwLds tw = tw_stage(F.ax, reinterpret_cast<float2*>(tile + H), tid, nthr);
    const float2* x2 = reinterpret_cast<const float2*>(x + (size_t)blockIdx.x * ns);     // ns % 4 == 0: 16-byte aligned rows
    constexpr int kAhead = 4;                                         // index pairs in flight per lane
    // ---- load + first radix-2 step (decimation in frequency)
    for (int n0 = tid; n0 < H; n0 += kAhead * nthr) {
        float2 q0[kAhead], q1[kAhead], wn[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int n = n0 + k * nthr;
            const bool in = n < H;
            q0[k] = in ? x2[n] : make_float2(0.f, 0.f);
            q1[k] = in ? x2[n + H] : make_float2(0.f, 0.f);
            wn[k] = in ? F.wfull[n] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int n = n0 + k * nthr;
            if (n < H) {
                const float2 a = c_add(q0[k], q1[k]);
                const float2 b = c_mul(c_sub(q0[k], q1[k]), wn[k]);
                tile[n] = make_float4(a.x, b.x, a.y, b.y);
            }
        }
    }
    lds_barrier();
    lds_fft_pair<false>(tile, F.ax, tw, tid, nthr);
    // ---- real-spectrum untangle, x (-i sgn f), re-tangle: bin f of the M-point spectrum is value (f & 1) of the word at
    //      pos[f >> 1]
    for (int f = tid; f <= H; f += nthr) {
        const int g = (f == 0) ? 0 : M - f;
        const int ia = 4 * F.pos[f >> 1] + (f & 1), ib = 4 * F.pos[g >> 1] + (g & 1);
        const float2 a = make_float2(tf[ia], tf[ia + 2]), bc = make_float2(tf[ib], -tf[ib + 2]);
        float2 out_a = make_float2(0.f, 0.f), out_b = make_float2(0.f, 0.f);
        if (f != 0) {
            const float2 w = F.wpack[f];
            const float2 E = c_scale(c_add(a, bc), 0.5f);
            const float2 O = c_mul_mi(c_scale(c_sub(a, bc), 0.5f));
            const float2 tO = c_mul(w, O);
            const float2 Yp = c_mul_mi(c_add(E, tO));             // -i X(f)
            const float2 Ym = c_mul_pi(c_sub(E, tO));             // +i X(f + M)  (negative frequency)
            const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
            const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
            out_a = c_add(S, D);
            out_b = c_conj(c_sub(S, D));
        }
        tf[ia] = out_a.x;
        tf[ia + 2] = out_a.y;
        if (ib != ia) {
            tf[ib] = out_b.x;
            tf[ib + 2] = out_b.y;
        }
    }
    lds_barrier();
    lds_fft_pair<true>(tile, F.ax, tw, tid, nthr);
    // ---- last radix-2 step of the inverse (decimation in time) + output: h[n] = a[n] + conj(W_M^n) b[n],
    //      h[n + M/2] = a[n] - conj(W_M^n) b[n]; h[m] = M (H[x][2m], H[x][2m + 1])
    const float scale = 1.0f / (float)M;
    const float inv_var = (mode == kAnSnr || mode == kAnEnvStd) ? 1.0f / var[blockIdx.x] : 0.f;
    float2* y2 = reinterpret_cast<float2*>(y + (size_t)blockIdx.x * ns);
    auto val = [&](float re, float im) -> float {
        const float p = fmaf(re, re, im * im);
        if (mode == kAnEnvelope) return sqrtf(p);
        if (mode == kAnHilbert) return im;
        if (mode == kAnEnvStd) return sqrtf(p * inv_var);
        return 10.0f * log10f(p * inv_var);
    };
    for (int n0 = tid; n0 < H; n0 += kAhead * nthr) {
        float2 q0[kAhead], q1[kAhead], wn[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int n = n0 + k * nthr;
            const bool in = n < H;
            q0[k] = in ? x2[n] : make_float2(0.f, 0.f);
            q1[k] = in ? x2[n + H] : make_float2(0.f, 0.f);
            wn[k] = in ? F.wfull[n] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int n = n0 + k * nthr;
            if (n < H) {
                const float4 v = tile[n];
                const float2 a = make_float2(v.x * scale, v.z * scale);
                const float2 t = c_mulc(make_float2(v.y * scale, v.w * scale), wn[k]);
                const float2 h0 = c_add(a, t), h1 = c_sub(a, t);
                y2[n] = make_float2(val(q0[k].x, h0.x), val(q0[k].y, h0.y));
                y2[n + H] = make_float2(val(q1[k].x, h1.x), val(q1[k].y, h1.y));
            }
        }
    }
}
