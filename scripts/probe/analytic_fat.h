// Experiment, not built (round 5): the analytic signal of 12 000-sample rows as THREE register-resident radix stages
// (20 x 20 x 15, compile-time indices, the structure of the f-k filter's pass B with the Hilbert pair operation; one row per
// workgroup, also tried as persistent workgroups with the next row prefetched).  Correct (2e-6 against scipy.signal.hilbert on
// the CPU test build, 38 GPU tests green with it as the default) and SLOWER: 782-793 us per 11 020 x 12 000 block against 492 us
// for analytic_rows (profiles/r05j/stream_kernels_analytic_fat.txt).  analytic_rows is bound by vector issue (269 M
// wave-instructions per block, profiles/r05i/pmc_sq_stream.txt), but the radix-20 / radix-15 register butterflies with their
// twiddle powers cost about as many instructions per row (~28 K wave-instructions against 24 K) -- 6000 = 2^4 3 5^3 has no cheap
// large radix -- and the 56-KB tile + 161 VGPRs leave two workgroups of 7 waves per compute unit instead of three of 8.
// The host side (tables: W_N2^j, W_M1^(j2 b), the untangle twiddle of a group's first frequency, the mirror group) is in git
// history of round 5 (commit "analytic_fat"); to try again paste this before row_var in csrc/spectral.hip.
// ---------------------------------------------------------------------------------------------
// The same operator for the row lengths that matter (the 60-s file at 200 Hz: 12 000 samples) as THREE register-resident
// radix stages with compile-time indices -- the structure of the f-k filter's pass B (fk_fast.h: fkf_passB with the Hilbert
// pair operation), one row per workgroup: the packed row of N2 = NA NB NC complex values is transformed by a radix-NA
// butterfly on the registers the global loads landed in, a radix-NB stage in LDS, and a middle item that runs
// [radix NC | real-spectrum untangle x (-i sgn f) re-tangle | inverse radix NC] for a group of NC positions and its mirror
// group; the inverse stages mirror the forward ones and the last butterfly feeds the epilogue (|x + i H|, ...) and the stores.
// analytic_rows runs the same transform as four or five generic radix <= 10 stages whose index arithmetic, twiddle
// products and LDS round trips made it bound by vector issue (269 M wave-instructions per 11 020 x 12 000 block = 0.44 of
// its 0.49 ms, profiles/r05i/pmc_sq_stream.txt).  Positions: e = d0 (NB NC) + d1 NC + d2 holds frequency
// d0 + NA (d1 + NB d2); LDS position e lives at e + e / NC (one pad per group).
// ---------------------------------------------------------------------------------------------
template <int NA_, int NB_, int NC_, int THR_>
struct AnFatCfg {
    static constexpr int NA = NA_, NB = NB_, NC = NC_, N2 = NA_ * NB_ * NC_, M1 = NB_ * NC_, NG = NA_ * NB_, THR = THR_;
    static constexpr int ROWP = N2 + NG;
    static constexpr size_t lds = (size_t)(ROWP + 2 * M1) * sizeof(float2);
    static_assert(M1 <= THR_ && NG <= THR_, "one S1 / middle item per thread");
};
struct AnFatDev {
    const float2* tw1;     // [M1]      W_N2^j
    const float2* tw2;     // [NB][NC]  W_M1^(j2 b)
    const float2* wc0;     // [NG]      W_(2 N2)^f of the group's first position (the untangle twiddle)
    const int* pgz;        // [NG]      mirror group of group G (frequency N2 - f)
};

template <class G>
__global__ __launch_bounds__(G::THR) void analytic_fat(AnFatDev T, const float* __restrict__ x, float* __restrict__ y, int nx, int mode,
                                                       const float* __restrict__ var) {
    D4W_DYN_LDS(smem_raw);
    constexpr int N2 = G::N2, NA = G::NA, NB = G::NB, NC = G::NC, M1 = G::M1, NG = G::NG, THR = G::THR;
    float2* row = reinterpret_cast<float2*>(smem_raw);
    float2* tw1 = row + G::ROWP;
    float2* tw2 = tw1 + M1;
    const int tid = threadIdx.x;
    for (int i = tid; i < M1; i += THR) {
        tw1[i] = T.tw1[i];
        tw2[i] = T.tw2[i];
    }
    const int j1 = tid;
    const bool it1 = tid < M1;
    const int aj1 = j1 + j1 / NC;                                // padded position of j1; of j1 + a M1: aj1 + a (M1 + NB)
    // persistent workgroups: the next row's samples are loaded into the butterfly registers as soon as S1 has consumed the
    // current row's (a row's load latency would otherwise stand in front of its five LDS phases)
    float2 pf[NA];
    auto issue = [&](int r) {
        const float2* xr2 = reinterpret_cast<const float2*>(x + (size_t)r * (2 * N2));
        if (it1) static_for<NA>([&](auto aa) { constexpr int a = decltype(aa)::value; pf[a] = xr2[j1 + a * M1]; });
    };
    int r = blockIdx.x;
    if (r < nx) issue(r);
    __syncthreads();
    for (; r < nx; r += gridDim.x) {
    const float2* x2 = reinterpret_cast<const float2*>(x + (size_t)r * (2 * N2));
    float2* y2 = reinterpret_cast<float2*>(y + (size_t)r * (2 * N2));
    // ---------------- S1
    if (it1) {
        dft<NA>(pf);
        row[aj1] = pf[0];
        pw_each<NA>(tw1[j1], [&](auto aa, float2 w) {
            constexpr int a = decltype(aa)::value + 1;
            row[aj1 + a * (M1 + NB)] = c_mul(pf[a], w);
        });
    }
    lds_barrier();
    if (r + (int)gridDim.x < nx) issue(r + gridDim.x);           // pf is free: the next row's loads fly under the LDS phases
    // ---------------- S2 (in place)
    for (int it = tid; it < NA * NC; it += THR) {
        const int g = it / NC, j2 = it - g * NC;
        const int gj2 = g * (M1 + NB) + j2;
        float2 v[NB];
        static_for<NB>([&](auto bb) { constexpr int b = decltype(bb)::value; v[b] = row[gj2 + b * (NC + 1)]; });
        dft<NB>(v);
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            row[gj2 + b * (NC + 1)] = (b == 0) ? v[0] : c_mul(v[b], tw2[b * NC + j2]);
        });
    }
    lds_barrier();
    // ---------------- MID: group Gi and its mirror group (the row is its own Hermitian partner)
    if (tid < NG) {
        const int Gi = tid, PG = T.pgz[Gi];
        if (PG >= Gi) {
            const bool selfg = (PG == Gi), rev0 = (Gi == 0);     // group 0 mirrors its digits as (NC - d) % NC
            float2* ga = row + Gi * (NC + 1);
            float2* gb = row + PG * (NC + 1);
            float2 a[NC], b[NC];
            static_for<NC>([&](auto dd) { constexpr int d = decltype(dd)::value; a[d] = ga[d]; b[d] = gb[d]; });
            dft<NC>(a);
            dft<NC>(b);
            const float2 w0 = T.wc0[Gi];
            float2 na[NC], nb[NC];
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d, pz = (NC - d) % NC;
                const float2 bs = rev0 ? b[pz] : b[pn];
                const float2 Bc = c_conj(bs);
                const float2 w = rot_const<d, 2 * NC>(w0);
                const float2 E = c_scale(c_add(a[d], Bc), 0.5f);
                const float2 O = c_mul_mi(c_scale(c_sub(a[d], Bc), 0.5f));
                const float2 tO = c_mul(w, O);
                const bool dcn = (d == 0 && rev0);               // the (f = 0, Nyquist) pair: no Hilbert transform
                const float2 Yp = dcn ? make_float2(0.f, 0.f) : c_mul_mi(c_add(E, tO));
                const float2 Ym = dcn ? make_float2(0.f, 0.f) : c_mul_pi(c_sub(E, tO));
                const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
                const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
                na[d] = c_add(S, D);
                nb[d] = c_conj(c_sub(S, D));
            });
            if (!selfg) {
                static_for<NC>([&](auto dd) {
                    constexpr int e = decltype(dd)::value;
                    constexpr int pn = NC - 1 - e, pz = (NC - e) % NC;
                    a[e] = na[e];
                    b[e] = rev0 ? nb[pz] : nb[pn];
                });
                idft<NC>(a);
                idft<NC>(b);
                static_for<NC>([&](auto dd) { constexpr int d = decltype(dd)::value; ga[d] = a[d]; gb[d] = b[d]; });
            } else {
                static_for<NC>([&](auto dd) {
                    constexpr int e = decltype(dd)::value;
                    constexpr int pn = NC - 1 - e, pz = (NC - e) % NC;
                    const float2 vn = (e < pn) ? na[e] : nb[pn];
                    const float2 vz = (e < pz) ? na[e] : nb[pz];
                    a[e] = rev0 ? vz : vn;
                });
                idft<NC>(a);
                static_for<NC>([&](auto dd) { constexpr int d = decltype(dd)::value; ga[d] = a[d]; });
            }
        }
    }
    lds_barrier();
    // ---------------- S2'
    for (int it = tid; it < NA * NC; it += THR) {
        const int g = it / NC, j2 = it - g * NC;
        const int gj2 = g * (M1 + NB) + j2;
        float2 v[NB];
        static_for<NB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            const float2 xv = row[gj2 + b * (NC + 1)];
            v[b] = (b == 0) ? xv : c_mulc(xv, tw2[b * NC + j2]);
        });
        idft<NB>(v);
        static_for<NB>([&](auto bb) { constexpr int b = decltype(bb)::value; row[gj2 + b * (NC + 1)] = v[b]; });
    }
    lds_barrier();
    // ---------------- S1' -> epilogue -> global
    if (it1) {
        float2 q[NA];                                            // the row's own samples again (L2), ahead of the butterfly
        static_for<NA>([&](auto aa) { constexpr int a = decltype(aa)::value; q[a] = x2[j1 + a * M1]; });
        float2 v[NA];
        v[0] = row[aj1];
        pw_each<NA>(tw1[j1], [&](auto aa, float2 w) {
            constexpr int a = decltype(aa)::value + 1;
            v[a] = c_mulc(row[aj1 + a * (M1 + NB)], w);
        });
        idft<NA>(v);
        const float scale = 1.0f / (float)N2;
        const float inv_var = (mode == kAnSnr || mode == kAnEnvStd) ? 1.0f / var[r] : 0.f;
        auto val = [&](float re, float im) -> float {
            const float p = fmaf(re, re, im * im);
            if (mode == kAnEnvelope) return sqrtf(p);
            if (mode == kAnHilbert) return im;
            if (mode == kAnEnvStd) return sqrtf(p * inv_var);
            return 10.0f * log10f(p * inv_var);
        };
        static_for<NA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            y2[j1 + a * M1] = make_float2(val(q[a].x, v[a].x * scale), val(q[a].y, v[a].y * scale));
        });
    }
    lds_barrier();                                               // the row buffer is rewritten by the next row's S1
    }
}

