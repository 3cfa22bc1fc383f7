// EXPERIMENT, measured and NOT adopted (round 3) -- kept for the record, not part of the build.
// Row-pair packed pass B (v_pk_* butterflies for a sub-row and its Hermitian partner at once): correct on the emulator
// and on the GPU (all f-k tests), but SLOWER than the scalar kernels on MI355X at 20 000 x 120 000:
//     dense fused pass B 8.0 ms (scalar 5.9), Bf 9.4 / 7.3 ms (3.8), Bi 6.1 / 4.9 ms (3.5)   [3 / 2 workgroups per CU asked]
// Packing halves the VALU instructions but not the state: a 128-thread workgroup must hold the same 38 KB pair in prefetch
// registers (75 per thread) and a packed radix-20 butterfly needs 160 transient registers -- 256 VGPRs + up to 128 AGPRs,
// one wave per SIMD, four waves per CU, and the stage chain (LDS round trips, barriers) is then latency-bound.  The scalar
// kernels run eight waves per CU at ~87 % VALU issue.  What would work is a FOUR-radix split (10 x 8 x 6 x 5: 40 prefetch
// registers, every stage ~240 items of 256 threads), which changes the position order of the n2 axis and with it every
// pass-B table of the plan -- not done.  To try it again: wire FkPbCfg / fkf_passBp into fk_entry.h and launch the regular
// pairs (two distinct sub-rows, k1 != 0, sorted first in the work lists) through it, the rest through fkf_passB / fkf_passBt.
//
// Row-pair PACKED pass B of the shape-specialised f-k filter (all three forms: the fused middle pass of the channel-first
// order and the two halves Bf / Bi of the time-first order, fk_fast.h / fk_tf.h).
//
// Pass B always works on a sub-row and its Hermitian partner sub-row -- two transforms of the same length with the same
// twiddles.  Here the two ride the halves of 64-bit register pairs (fft_pair.h: c2 = one complex value of each row,
// every butterfly add / multiply ONE v_pk_* instruction for both rows), so the n2 transforms cost half the VALU issue
// slots of fkf_passB / fkf_passBt, which are bound by exactly that (175 Gflop of scalar butterflies per
// dense pass).  LDS holds ONE row of c2 = {reA, reB, imA, imB} (16-byte elements, same bytes as two float2 rows).
//   S1  item (j < NB NC)          : radix NA on the prefetch registers of both rows, x W_N2^(j a')        -> LDS
//   S2  item (g < NA, j2 < NC)    : radix NB in place, x W_(NB NC)^(j2 b')
//   MID item (Gi < NG / 2)        : radix NC on group Gi and on its mirror group PG = NG-1-Gi (both rows each), the TWO
//                                   Hermitian pair operations (A_Gi with B_PG, A_PG with B_Gi), inverse radix NC -- or
//                                   the stores to / loads from the compact spectrum W (time-first halves)
//   S2', S1' inverse; S1' feeds the global stores of both rows.
// 128 threads (every stage has ~NG / 2 items), three workgroups per CU.
// Only REGULAR pairs run here: two distinct sub-rows whose n1 frequency is not 0 (mirror group NG-1-Gi, mirror digit
// NC-1-d, no Nyquist column, no self-paired row).  The others -- 1 / N1 of the work list -- go through fkf_passB /
// fkf_passBt in a second launch on their own list.
#pragma once
#include "fft_pair.h"
#include "fk_tf.h"

namespace d4w {

constexpr int kPbThreads = 128;

template <class G>
struct FkPbCfg {
    static constexpr int NA = G::NA, NB = G::NB, NC = G::NC, N2 = G::N2, M1 = NB * NC, NG = NA * NB;
    static constexpr int NI = (NG + 1) / 2;                       // MID items (the middle group of an odd NG pairs with itself)
    static constexpr int ROWP = N2 + NG;                           // one pad element per group of NC
    static constexpr bool ok = (M1 <= kPbThreads) && (NI <= kPbThreads) && (G::C2X == 1);
    static constexpr size_t lds = (size_t)ROWP * sizeof(float4) + (size_t)2 * M1 * sizeof(float2) + (size_t)NC * NB * sizeof(int2);
};

__device__ __forceinline__ c2 pb_ld(const float4* p) {
    const float4 v = lds_read4(p);
    return c2{v2_make(v.x, v.y), v2_make(v.z, v.w)};
}
__device__ __forceinline__ void pb_st(float4* p, c2 v) { *p = make_float4(v2_x(v.re), v2_y(v.re), v2_x(v.im), v2_y(v.im)); }

// PHASE 0: fused middle pass (channel-first order; mask from P.mask, work list F.pairs).
// PHASE 1: Bf (time-first): forward + untangle -> T.W.      PHASE 2: Bi: T.W -> re-tangle + inverse.
template <class G, int PHASE>
__global__ __launch_bounds__(kPbThreads) void fkf_passBp(FkDev P, FkFastDev F, FkTfDev T, float2* __restrict__ data, int tbase, int npairs) {
    typedef FkPbCfg<G> Cf;
    constexpr int N2 = Cf::N2, NA = Cf::NA, NB = Cf::NB, NC = Cf::NC, M1 = Cf::M1, NG = Cf::NG, NI = Cf::NI;
    constexpr int THR = kPbThreads;
    D4W_DYN_LDS(smem_raw);
    float4* rows = reinterpret_cast<float4*>(smem_raw);             // [ROWP] c2
    float2* tw1 = reinterpret_cast<float2*>(rows + Cf::ROWP);       // [M1]      W_N2^j
    float2* tw2 = tw1 + M1;                                         // [NB][NC]  W_M1^(j2 b)
    int2* ctab = reinterpret_cast<int2*>(tw2 + M1);                 // [NC][NB]  (time-first)
    const int tid = threadIdx.x;
    for (int i = tid; i < M1; i += THR) {
        tw1[i] = F.twB1[i];
        tw2[i] = F.twB2[i];
    }
    if constexpr (PHASE != 0)
        for (int i = tid; i < NC * NB; i += THR) ctab[i] = T.ctab[i];
    __syncthreads();
    auto ad = [](int e) { return e + e / NC; };
    const int2* __restrict__ plist = (PHASE == 0) ? F.pairs : T.pairs;

    // S1 / S1' item
    const int j1 = tid;
    const bool it1 = tid < M1;
    // MID item: groups Gi and PG = NG - 1 - Gi (the same group for the middle item of an odd NG)
    const int Gi = tid, PG = NG - 1 - tid;
    const bool mid = tid < NI;
    const bool selfg = (NG % 2 == 1) && (Gi == PG);
    float2 wc[NC];                 // W_ns^(N1 k2) of group Gi's positions; group PG's are c0 conj(wc[NC-1-d]), c0 = W_ns^(M - N1)
    int colG[NC], colP[NC];        // time-first: columns of the two groups' positions inside a sub-row block, -1 = not kept
    if (mid) {
        const float2* wcp = P.wcol + Gi * NC;
        static_for<NC>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            wc[d] = wcp[d];
            if constexpr (PHASE != 0) {
                const int2 cg = ctab[d * NB + Gi % NB], cp = ctab[d * NB + PG % NB];
                colG[d] = (cg.y >> 28) ? cg.x + (Gi / NB) * (cg.y & 0x0FFFFFFF) : -1;
                colP[d] = (cp.y >> 28) ? cp.x + (PG / NB) * (cp.y & 0x0FFFFFFF) : -1;
            }
        });
    }
    // W_ns^(N1 (N2 - 1)): the column twiddle of the last position, which holds k2 = N2 - 1 (every digit of position 0 mirrored)
    const float2 cm = P.wcol[N2 - 1];

    int t = tbase + blockIdx.x;
    const int gstep = gridDim.x;
    int2 pr_cur = make_int2(0, 0), pr_nxt = make_int2(0, 0);
    if (t < npairs) pr_cur = plist[t];
    if (t + gstep < npairs) pr_nxt = plist[t + gstep];

    // ---- prefetch registers
    c2 pf[NA];                                                      // PHASE 0 / 1: the pair's samples, one pair ahead
    auto issue = [&](int2 pr) {
        if (it1) {
            const float2* pa = data + (size_t)pr.x * N2 + j1;
            const float2* pb = data + (size_t)pr.y * N2 + j1;
            static_for<NA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                pf[a] = c2_make(pa[a * M1], pb[a * M1]);
            });
        }
    };
    // gains at the four position sets of a MID item: row A group Gi, row A group PG, row B group PG, row B group Gi
    // (PHASE 0: the folded mask, one pair ahead; PHASE 1: the tail gains, reloaded when the sub-row changes)
    struct Gains {
        float aG[NC], aP[NC], bP[NC], bG[NC];
    };
    Gains gcur, gnxt;
    int q1_loaded = -1;
    auto issue_gains = [&](Gains& O, int2 pr) {
        if (!mid) return;
        if constexpr (PHASE == 0) {
            const float* mA = P.mask + (size_t)pr.x * N2;
            const float* mB = P.mask + (size_t)pr.y * N2;
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                O.aG[d] = mA[(unsigned)(Gi * NC + d)];
                O.aP[d] = mA[(unsigned)(PG * NC + d)];
                O.bP[d] = mB[(unsigned)(PG * NC + d)];
                O.bG[d] = mB[(unsigned)(Gi * NC + d)];
            });
        } else {
            const int q1A = pr.x % G::N1, q1B = pr.y % G::N1;
            const float* mA = T.tgain + q1A * N2;
            const float* mB = T.tgain + q1B * N2;
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                O.aG[d] = mA[(unsigned)(Gi * NC + d)];
                O.aP[d] = mA[(unsigned)(PG * NC + d)];
                O.bP[d] = mB[(unsigned)(PG * NC + d)];
                O.bG[d] = mB[(unsigned)(Gi * NC + d)];
            });
        }
    };
    // PHASE 2: the pair's compact spectrum, loaded right after the previous MID
    float2 WaG[NC], WaP[NC], WbP[NC], WbG[NC];
    auto issue_w = [&](int2 pr) {
        if (!mid) return;
        const int rA = pr.x / G::N1, q1A = pr.x - rA * G::N1;
        const int rB = pr.y / G::N1, q1B = pr.y - rB * G::N1;
        const float2* WA = T.W + (size_t)rA * T.Lc + q1A * T.RW;
        const float2* WB = T.W + (size_t)rB * T.Lc + q1B * T.RW;
        const float2 z = make_float2(0.f, 0.f);
        static_for<NC>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            WaG[d] = (colG[d] >= 0) ? WA[(unsigned)colG[d]] : z;
            WaP[d] = (colP[d] >= 0) ? WA[(unsigned)colP[d]] : z;
            WbP[d] = (colP[d] >= 0) ? WB[(unsigned)colP[d]] : z;
            WbG[d] = (colG[d] >= 0) ? WB[(unsigned)colG[d]] : z;
        });
    };

    if (t < npairs) {
        if constexpr (PHASE == 0) issue_gains(gcur, pr_cur);
        if constexpr (PHASE != 2) issue(pr_cur);
        if constexpr (PHASE == 2) issue_w(pr_cur);
    }
    for (; t < npairs; t += gstep) {
        const int2 pr = pr_cur;
        int2 pr_nn = pr_cur;
        if (t + 2 * gstep < npairs) pr_nn = plist[t + 2 * gstep];
        const int rA = pr.x / G::N1, q1A = pr.x - rA * G::N1;
        const int rB = pr.y / G::N1, q1B = pr.y - rB * G::N1;
        (void)q1B;
        if constexpr (PHASE != 2) {
            // ---------------- S1 (both rows)
            if (it1) {
                dftp<NA>(pf);
                float2 pw[NA];
                pw_tree<NA>(tw1[j1], pw);
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    pb_st(rows + ad(j1 + a * M1), (a == 0) ? pf[0] : c2_mulw(pf[a], pw[a]));
                });
            }
            lds_barrier();
            if (t + gstep < npairs) issue(pr_nxt);
            // ---------------- S2 (in place)
            for (int it = tid; it < NA * NC; it += THR) {
                const int g = it / NC, j2 = it - g * NC;
                c2 v[NB];
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    v[b] = pb_ld(rows + ad(g * M1 + j2 + b * NC));
                });
                dftp<NB>(v);
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    pb_st(rows + ad(g * M1 + j2 + b * NC), (b == 0) ? v[0] : c2_mulw(v[b], tw2[b * NC + j2]));
                });
            }
            lds_barrier();
        }
        // ---------------- MID
        // per pair: row twiddle W_ns^k1 (group Gi's positions: wr wc[d]; group PG's: wr cm conj(wc[NC-1-d'])), and for the
        // time-first halves omega = twc[rA] twc[rB] (fk_tf.h)
        const float2 wr = P.wrow[q1A];
        const float2 wr2 = c_mul(wr, cm);
        float2 om = make_float2(1.f, 0.f);
        if constexpr (PHASE != 0) om = c_mul(P.twc[rA], P.twc[rB]);
        if constexpr (PHASE == 1) {
            if (q1A != q1_loaded) {                                  // the work list is sorted by sub-row: rare
                issue_gains(gcur, pr);
                q1_loaded = q1A;
            }
        }
        if (mid) {
            c2 Pg[NC], Qg[NC];
            if constexpr (PHASE != 2) {
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    Pg[d] = pb_ld(rows + ad(Gi * NC + d));
                    Qg[d] = pb_ld(rows + ad(PG * NC + d));
                });
                dftp<NC>(Pg);
                if (!selfg) dftp<NC>(Qg);
            }
            float2* WA = nullptr;
            float2* WB = nullptr;
            if constexpr (PHASE == 1) {
                WA = T.W + (size_t)rA * T.Lc + q1A * T.RW;
                WB = T.W + (size_t)rB * T.Lc + q1B * T.RW;
            }
            // one Hermitian pair operation: a = Z_A at a position, b = Z_B at the mirror position, w = W_ns^f of a's frequency.
            //   forward : Yp = X_A[f], Ym = X_A[f - M]      inverse : a <- S + D, b <- omega conj(S - D)
            auto untangle = [&](float2 a, float2 b, float2 w, float2& Yp, float2& Ym) {
                const float2 Bc = (PHASE == 0) ? c_conj(b) : c_mul(om, c_conj(b));
                const float2 E = c_scale(c_add(a, Bc), 0.5f);
                const float2 O = c_mul_mi(c_scale(c_sub(a, Bc), 0.5f));
                const float2 tO = c_mul(w, O);
                Yp = c_add(E, tO);
                Ym = c_sub(E, tO);
            };
            auto retangle = [&](float2 Yp, float2 Ym, float2 w, float2& na, float2& nb) {
                const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
                const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
                na = c_add(S, D);
                nb = (PHASE == 0) ? c_conj(c_sub(S, D)) : c_mul(om, c_conj(c_sub(S, D)));
            };
            // new values: row A at (Gi, d) / (PG, e), row B at (Gi, d) / (PG, e)
            float2 naG[NC], naP[NC], nbG[NC], nbP[NC];
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d;
                const float2 w1 = c_mul(wr, wc[d]);                    // frequency of (Gi, d) in row A's sub-row
                const float2 w2 = c_mul(wr2, c_conj(wc[d]));           // frequency of (PG, pn) in row A's sub-row
                float2 Yp1, Ym1, Yp2 = make_float2(0.f, 0.f), Ym2 = make_float2(0.f, 0.f);
                if constexpr (PHASE != 2) {
                    // op 1: A at (Gi, d) with B at (PG, pn);  op 2: A at (PG, pn) with B at (Gi, d)
                    const c2 qsrc = selfg ? Pg[pn] : Qg[pn];
                    untangle(c2_a(Pg[d]), c2_b(qsrc), w1, Yp1, Ym1);
                    if (!selfg) untangle(c2_a(qsrc), c2_b(Pg[d]), w2, Yp2, Ym2);
                }
                naP[pn] = make_float2(0.f, 0.f);
                nbG[d] = make_float2(0.f, 0.f);
                if constexpr (PHASE == 0) {
                    retangle(c_scale(Yp1, gcur.aG[d]), c_scale(Ym1, gcur.bP[pn]), w1, naG[d], nbP[pn]);
                    if (!selfg) retangle(c_scale(Yp2, gcur.aP[pn]), c_scale(Ym2, gcur.bG[d]), w2, naP[pn], nbG[d]);
                } else if constexpr (PHASE == 1) {
                    // X_A[f] = Yp, X_B[M - f] = omega conj(Ym)
                    if (colG[d] >= 0) WA[(unsigned)colG[d]] = c_scale(Yp1, gcur.aG[d]);
                    if (colP[pn] >= 0) WB[(unsigned)colP[pn]] = c_scale(c_mul(om, c_conj(Ym1)), gcur.bP[pn]);
                    if (!selfg) {
                        if (colP[pn] >= 0) WA[(unsigned)colP[pn]] = c_scale(Yp2, gcur.aP[pn]);
                        if (colG[d] >= 0) WB[(unsigned)colG[d]] = c_scale(c_mul(om, c_conj(Ym2)), gcur.bG[d]);
                    }
                } else {
                    retangle(WaG[d], c_mul(om, c_conj(WbP[pn])), w1, naG[d], nbP[pn]);
                    if (!selfg) retangle(WaP[pn], c_mul(om, c_conj(WbG[d])), w2, naP[pn], nbG[d]);
                }
            });
            c2 Pn[NC], Qn[NC];
            if constexpr (PHASE != 1) {
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    Pn[d] = c2_make(naG[d], selfg ? nbP[d] : nbG[d]);      // the middle group of an odd NG mirrors onto itself
                    Qn[d] = c2_make(naP[d], nbP[d]);
                });
            }
            if constexpr (PHASE != 1) {
                idftp<NC>(Pn);
                if (!selfg) idftp<NC>(Qn);
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    pb_st(rows + ad(Gi * NC + d), Pn[d]);
                    if (!selfg) pb_st(rows + ad(PG * NC + d), Qn[d]);
                });
            }
        }
        lds_barrier();
        if constexpr (PHASE == 2) {
            if (t + gstep < npairs) issue_w(pr_nxt);                 // next pair's spectrum, ahead of this pair's stores
        }
        if constexpr (PHASE != 1) {
            // ---------------- S2'
            for (int it = tid; it < NA * NC; it += THR) {
                const int g = it / NC, j2 = it - g * NC;
                c2 v[NB];
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    const c2 x = pb_ld(rows + ad(g * M1 + j2 + b * NC));
                    v[b] = (b == 0) ? x : c2_mulwc(x, tw2[b * NC + j2]);
                });
                idftp<NB>(v);
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    pb_st(rows + ad(g * M1 + j2 + b * NC), v[b]);
                });
            }
            lds_barrier();
            if constexpr (PHASE == 0) {
                if (t + gstep < npairs) issue_gains(gnxt, pr_nxt);   // next pair's mask values, ahead of this pair's stores
            }
            // ---------------- S1' -> global (both rows)
            if (it1) {
                c2 v[NA];
                float2 pw[NA];
                pw_tree<NA>(tw1[j1], pw);
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const c2 x = pb_ld(rows + ad(j1 + a * M1));
                    v[a] = (a == 0) ? x : c2_mulwc(x, pw[a]);
                });
                idftp<NA>(v);
                float2* oa = data + (size_t)pr.x * N2 + j1;
                float2* ob = data + (size_t)pr.y * N2 + j1;
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    oa[a * M1] = c2_a(v[a]);
                    ob[a * M1] = c2_b(v[a]);
                });
            }
            lds_barrier();
            if constexpr (PHASE == 0) gcur = gnxt;
        }
        pr_cur = pr_nxt;
        pr_nxt = pr_nn;
    }
}

}  // namespace d4w
