// TIME-FIRST order of the shape-specialised f-k passes.
//
// The c2 (channel) and n2 (time) sub-transforms commute, so the five passes of fk_fast.h can also run as
//     A (c1, n1)  ->  Bf (n2 forward + real-spectrum untangle)  ->  Cm (c2 forward x mask x c2 inverse)  ->  Bi  ->  A'
// with the HALF SPECTRUM COMPACTED between Bf and Bi: Bf writes only the frequency columns the mask needs into the
// workspace W [nx][Lc]; Cm runs on the columns whose gain depends on the wavenumber ("band" columns); columns whose gain
// is the same for every wavenumber ("tail" columns -- the Butterworth skirts hybrid_ninf_filter_design leaves outside its
// looped columns, dsp.py:348-360) are scaled by Bf and skip the channel transform altogether; columns whose gain is zero
// are never written.  Bytes per channel-sample: 24 + 18 f_band + 8 f_tail, against 24 + 18 f_live_rows in channel-first
// order (42 when nothing is dead) -- the planner picks the order per mask (fk_filter.hip, fk_mask_finish).
//
// Frequencies come in CELLS of NA consecutive n2 frequencies: k2 = d0 + NA (d1 + NB d2), d0 < NA, cell (d2, d1) -- exactly
// the digits the MID item of pass B holds (thread Gi = d0 NB + d1 owns digit d2 = 0 .. NC-1 of group Gi).  A cell is band,
// tail or dead for all N1 sub-rows alike.  A row of W is N1 sub-row blocks of RW columns, [band cells | tail cells] each,
// + one block for the Nyquist column f = M:   column of (q1, d0, d1, d2) = q1 RW + ctab[d2][d1].x + d0 cnt,
// cnt = cells of that class in digit d2 -- the lanes of a wave (consecutive Gi) write consecutive columns, and a thread's
// NC column offsets inside a sub-row block are constants it computes once (the sub-row block's base is wave-uniform).
//
// Before the c2 transform the Hermitian partner of row (q, c2) is the row (q', c2) with kc1(q') = -kc1(q); as sequences
// in time  row_B = omega conj(row_A),  omega = twc[rA] twc[rB]  (the four-step twiddles W_nx^(c2 kc1) the two rows carry):
// one wave-uniform factor per pair in the untangle / re-tangle algebra of fkf_passB.
#pragma once
#include "fk_fast.h"

namespace d4w {

struct FkTfDev {
    const int2* pairs;     // work list of passes Bf / Bi: (keyA, keyB), key = row * N1 + q1
    const float* tgain;    // [N1 pos q1][N2 pos e]: 1 band, C2 x the wavenumber-independent gain of a tail column, 0 dead
    const int2* ctab;      // [NC][NB] {x = first column of the cell inside a sub-row block, y = cnt | class << 28}
    const float* cmask;    // [nx pos r][Lc] folded mask at the band columns (same indexing as W)
    float2* W;             // [nx][Lc] compact half spectrum
    const unsigned* cmlive; // [Cm tiles][C2A] bit b = the band mask of row q C2 + g C2B + b is non-zero somewhere in the tile's strip
    int RW;                // columns of a sub-row block (a multiple of TC)
    int bw;                // band columns of a sub-row block, rounded up to a multiple of TC (they come first)
    int col_nyq;           // column of f = M (= N1 RW), -1: dead
    int Lc;                // row pitch of W and cmask
};

// PHASE 1 = Bf: n2 forward of a sub-row pair, untangle, x tail gain -> W.
// PHASE 2 = Bi: W -> re-tangle, n2 inverse of the pair -> data.
// Stages, LDS layout and special cases (a row that is its own partner, the k1 = 0 sub-row, the (0, Nyquist) pair) are
// those of fkf_passB, whose MID step this kernel cuts in two.
template <class G, int PHASE>
__global__ __launch_bounds__(G::THRB, G::WAVES_B) void fkf_passBt(FkDev P, FkFastDev F, FkTfDev T, float2* __restrict__ data,
                                                                  int tbase, int npairs) {
    D4W_DYN_LDS(smem_raw);
    constexpr int N2 = G::N2, NA = G::NA, NB = G::NB, NC = G::NC, M1 = NB * NC, NG = NA * NB, ROWP = G::ROWP;
    constexpr int THR = G::THRB;
    float2* rows = reinterpret_cast<float2*>(smem_raw);
    float2* tw1 = rows + 2 * ROWP;          // [M1]      W_N2^j
    float2* tw2 = tw1 + M1;                 // [NB][NC]  W_M1^(j2 b)
    int2* ctab = reinterpret_cast<int2*>(tw2 + M1);     // [NC][NB]
    const int tid = threadIdx.x;
    for (int i = tid; i < M1; i += THR) {
        tw1[i] = F.twB1[i];
        tw2[i] = F.twB2[i];
    }
    for (int i = tid; i < NC * NB; i += THR) ctab[i] = T.ctab[i];
    __syncthreads();
    // column of frequency digits (d0, d1, d2) inside a sub-row block, -1: not kept
    auto cellcol = [&](int dd0, int dd1, int dd2) -> int {
        const int2 c = ctab[dd2 * NB + dd1];
        return (c.y >> 28) ? c.x + dd0 * (c.y & 0x0FFFFFFF) : -1;
    };

    const int r1 = tid / M1, j1 = tid % M1;              // S1 / S1' item
    const int aj1 = j1 + j1 / NC;                        // padded position of j1; of j1 + a M1: aj1 + a (M1 + NB)
    const bool it1 = tid < 2 * M1;
    const int Gi = tid;                                  // MID item
    const bool midrange = Gi < NG;
    const int d0 = Gi / NB, d1 = Gi % NB;
    int PGz = 0;
    // column twiddle W_ns^(N1 k2) of the group's first position; digit d adds the literal rotation exp(-2 pi i d / (2 NC))
    float2 wc0 = make_float2(1.f, 0.f);
    int cxA[NC], cxB[NC];          // columns of the thread's own digits / of its partner group's (PG = NG-1-Gi) digits
    if (midrange) {
        PGz = P.mirror0[Gi * NC] / NC;
        wc0 = P.wcol[Gi * NC];
        const int PGn = NG - 1 - Gi;
        static_for<NC>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            cxA[d] = cellcol(d0, d1, d);
            cxB[d] = cellcol(PGn / NB, PGn % NB, d);
        });
    }
    int t = tbase + blockIdx.x;
    const int gstep = gridDim.x;
    int2 pr_cur = make_int2(0, 0), pr_nxt = make_int2(0, 0);
    if (t < npairs) pr_cur = T.pairs[t];
    if (t + gstep < npairs) pr_nxt = T.pairs[t + gstep];

    if constexpr (PHASE == 1) {
        float2 pf[NA];
        auto issue = [&](int2 pr) {
            if (it1 && (r1 == 0 || pr.x != pr.y)) {
                const float2* p = data + (size_t)(r1 ? pr.y : pr.x) * N2 + j1;
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    pf[a] = p[a * M1];
                });
            }
        };
        // tail gains of the pair's two sub-rows at the thread's positions: the work list is sorted by sub-row, so a workgroup's
        // consecutive pairs share them -- loaded when the sub-row changes (once in ~ nx / (2 grid) pairs), not per pair
        float ma[NC], mbr[NC];
        int q1_loaded = -1;
        auto load_gains = [&](int q1A, int q1B) {
            if (!midrange) return;
            const int PG = (q1A == 0) ? PGz : (NG - 1 - Gi);
            const float* mA = T.tgain + q1A * N2;
            const float* mB = T.tgain + q1B * N2;
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                ma[d] = mA[(unsigned)(Gi * NC + d)];
                mbr[d] = mB[(unsigned)(PG * NC + d)];
            });
        };
        if (t < npairs) issue(pr_cur);
        for (; t < npairs; t += gstep) {
            const int2 pr = pr_cur;
            int2 pr_nn = pr_cur;
            if (t + 2 * gstep < npairs) pr_nn = T.pairs[t + 2 * gstep];
            const bool same = (pr.x == pr.y);
            const int nrows = same ? 1 : 2;
            const int rA = pr.x / G::N1, q1A = pr.x - rA * G::N1;
            const int rB = pr.y / G::N1, q1B = pr.y - rB * G::N1;
            const bool k1zero = (q1A == 0);
            // ---------------- S1
            if (it1 && r1 < nrows) {
                dft<NA>(pf);
                float2* row = rows + r1 * ROWP;
                row[aj1] = pf[0];
                pw_each<NA>(tw1[j1], [&](auto aa, float2 w) {
                    constexpr int a = decltype(aa)::value + 1;
                    row[aj1 + a * (M1 + NB)] = c_mul(pf[a], w);
                });
            }
            lds_barrier();
            if (t + gstep < npairs) issue(pr_nxt);
            // ---------------- S2 (in place)
            for (int it = tid; it < nrows * NA * NC; it += THR) {
                const int r = it / (NA * NC), rem = it - r * (NA * NC);
                const int g = rem / NC, j2 = rem - g * NC;
                const int gj2 = g * (M1 + NB) + j2;              // padded position of (g, j2, b = 0): e + e / NC, j2 < NC
                float2* row = rows + r * ROWP;
                float2 v[NB];
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    v[b] = row[gj2 + b * (NC + 1)];
                });
                dft<NB>(v);
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    row[gj2 + b * (NC + 1)] = (b == 0) ? v[0] : c_mul(v[b], tw2[b * NC + j2]);
                });
            }
            lds_barrier();
            // ---------------- MID, first half: radix NC, untangle, x gain -> W
            const int PG = k1zero ? PGz : (NG - 1 - Gi);
            if (q1A != q1_loaded) {
                load_gains(q1A, q1B);
                q1_loaded = q1A;
            }
            if (midrange && (!same || PG >= Gi)) {
                const bool selfg = same && (PG == Gi);
                const bool rev0 = k1zero && (Gi == 0);
                const float2* ga = rows + Gi * (NC + 1);
                const float2* gb = rows + (same ? 0 : ROWP) + PG * (NC + 1);
                float2 a[NC], b[NC];
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    a[d] = ga[d];
                    b[d] = gb[d];
                });
                dft<NC>(a);
                dft<NC>(b);
                const float2 om = c_mul(P.twc[rA], P.twc[rB]), wr = P.wrow[q1A];
                const float2 w0 = c_mul(wr, wc0);
                float2* WA = T.W + (size_t)rA * T.Lc + q1A * T.RW;          // the pair's sub-row blocks: wave-uniform bases
                float2* WB = T.W + (size_t)rB * T.Lc + q1B * T.RW;
                const int pd0 = PGz / NB, pd1 = PGz - pd0 * NB;            // the k1 = 0 sub-row mirrors its groups differently
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    constexpr int pn = NC - 1 - d, pz = (NC - d) % NC;
                    const float2 bs = rev0 ? b[pz] : b[pn];
                    const float gB = rev0 ? mbr[pz] : mbr[pn];
                    const float2 Bc = c_mul(om, c_conj(bs));
                    const float2 w = rot_const<d, 2 * NC>(w0);
                    const float2 E = c_scale(c_add(a[d], Bc), 0.5f);
                    const float2 O = c_mul_mi(c_scale(c_sub(a[d], Bc), 0.5f));
                    const float2 tO = c_mul(w, O);
                    const float2 Yp = c_add(E, tO);              // X_A[f],      f = k1 + N1 k2(Gi, d)
                    const float2 Ym = c_sub(E, tO);              // X_A[f - M];  X_B[M - f] = omega conj(Ym)
                    if (cxA[d] >= 0) WA[(unsigned)cxA[d]] = c_scale(Yp, ma[d]);
                    if (d == 0 && rev0) {                        // f = 0 and the Nyquist column of row A (and of row B)
                        if (T.col_nyq >= 0) (T.W + (size_t)rA * T.Lc)[T.col_nyq] = Ym;
                        if (!same) {
                            if (cxA[d] >= 0) WB[(unsigned)cxA[d]] = c_scale(c_mul(om, c_conj(Yp)), ma[d]);
                            if (T.col_nyq >= 0) (T.W + (size_t)rB * T.Lc)[T.col_nyq] = c_mul(om, c_conj(Ym));
                        }
                    } else if (!selfg) {
                        int cB;
                        if (k1zero) cB = cellcol(pd0, pd1, rev0 ? pz : pn);    // wave-uniform branch, 1 / N1 of the pairs
                        else cB = cxB[pn];
                        if (cB >= 0) WB[(unsigned)cB] = c_scale(c_mul(om, c_conj(Ym)), gB);
                    }
                });
            }
            lds_barrier();
            pr_cur = pr_nxt;
            pr_nxt = pr_nn;
        }
    } else {
        // ---------------------------------------------------------------- PHASE 2
        float2 Wp[NC], Wm[NC];
        float2 wr_n = make_float2(1.f, 0.f), om_n = make_float2(1.f, 0.f);
        auto issue_w = [&](int2 pr) {
            if (!midrange) return;
            const int rA = pr.x / G::N1, q1A = pr.x - rA * G::N1;
            const int rB = pr.y / G::N1, q1B = pr.y - rB * G::N1;
            const bool k1z = (q1A == 0);
            const int PG = k1z ? PGz : (NG - 1 - Gi);
            const bool rev0 = k1z && (Gi == 0);
            const float2* WA = T.W + (size_t)rA * T.Lc + q1A * T.RW;
            const float2* WB = T.W + (size_t)rB * T.Lc + q1B * T.RW;
            int cb[NC];
            if (k1z) {
                const int pd0 = PG / NB, pd1 = PG - pd0 * NB;
                static_for<NC>([&](auto dd) { constexpr int d = decltype(dd)::value; cb[d] = cellcol(pd0, pd1, d); });
            } else
                static_for<NC>([&](auto dd) { constexpr int d = decltype(dd)::value; cb[d] = cxB[d]; });
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d, pz = (NC - d) % NC;
                Wp[d] = (cxA[d] >= 0) ? WA[(unsigned)cxA[d]] : make_float2(0.f, 0.f);
                if (d == 0 && rev0) {
                    Wm[d] = (T.col_nyq >= 0) ? (T.W + (size_t)rA * T.Lc)[T.col_nyq] : make_float2(0.f, 0.f);      // Y_A[M] itself
                } else {
                    const int cB = rev0 ? cb[pz] : cb[pn];
                    Wm[d] = (cB >= 0) ? WB[(unsigned)cB] : make_float2(0.f, 0.f);
                }
            });
            wr_n = P.wrow[q1A];
            om_n = c_mul(P.twc[rA], P.twc[rB]);
        };
        if (t < npairs) issue_w(pr_cur);
        for (; t < npairs; t += gstep) {
            const int2 pr = pr_cur;
            int2 pr_nn = pr_cur;
            if (t + 2 * gstep < npairs) pr_nn = T.pairs[t + 2 * gstep];
            const bool same = (pr.x == pr.y);
            const int nrows = same ? 1 : 2;
            const bool k1zero = (pr.x % G::N1 == 0);
            const int PG = k1zero ? PGz : (NG - 1 - Gi);
            // ---------------- MID, second half: re-tangle, inverse radix NC -> LDS
            if (midrange && (!same || PG >= Gi)) {
                const bool selfg = same && (PG == Gi);
                const bool rev0 = k1zero && (Gi == 0);
                float2* ga = rows + Gi * (NC + 1);
                float2* gb = rows + (same ? 0 : ROWP) + PG * (NC + 1);
                const float2 om = om_n, wr = wr_n;
                const float2 w0 = c_mul(wr, wc0);
                float2 a[NC], b[NC], na[NC], nb[NC];
                static_for<NC>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    const float2 Yp = Wp[d];
                    const float2 Ym = (d == 0 && rev0) ? Wm[d] : c_mul(om, c_conj(Wm[d]));
                    const float2 w = rot_const<d, 2 * NC>(w0);
                    const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
                    const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
                    na[d] = c_add(S, D);
                    nb[d] = c_mul(om, c_conj(c_sub(S, D)));
                });
                if (!selfg) {
                    static_for<NC>([&](auto dd) {                 // un-permute the partner results
                        constexpr int e = decltype(dd)::value;
                        constexpr int pn = NC - 1 - e, pz = (NC - e) % NC;
                        a[e] = na[e];
                        b[e] = rev0 ? nb[pz] : nb[pn];
                    });
                    idft<NC>(a);
                    idft<NC>(b);
                    static_for<NC>([&](auto dd) {
                        constexpr int d = decltype(dd)::value;
                        ga[d] = a[d];
                        gb[d] = b[d];
                    });
                } else {
                    static_for<NC>([&](auto dd) {
                        constexpr int e = decltype(dd)::value;
                        constexpr int pn = NC - 1 - e, pz = (NC - e) % NC;
                        const float2 vn = (e < pn) ? na[e] : nb[pn];
                        const float2 vz = (e < pz) ? na[e] : nb[pz];
                        a[e] = rev0 ? vz : vn;
                    });
                    idft<NC>(a);
                    static_for<NC>([&](auto dd) {
                        constexpr int d = decltype(dd)::value;
                        ga[d] = a[d];
                    });
                }
            }
            lds_barrier();
            if (t + gstep < npairs) issue_w(pr_nxt);          // next pair's spectrum, ahead of this pair's stores
            // ---------------- S2'
            for (int it = tid; it < nrows * NA * NC; it += THR) {
                const int r = it / (NA * NC), rem = it - r * (NA * NC);
                const int g = rem / NC, j2 = rem - g * NC;
                const int gj2 = g * (M1 + NB) + j2;              // padded position of (g, j2, b = 0): e + e / NC, j2 < NC
                float2* row = rows + r * ROWP;
                float2 v[NB];
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    const float2 x = row[gj2 + b * (NC + 1)];
                    v[b] = (b == 0) ? x : c_mulc(x, tw2[b * NC + j2]);
                });
                idft<NB>(v);
                static_for<NB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    row[gj2 + b * (NC + 1)] = v[b];
                });
            }
            lds_barrier();
            // ---------------- S1' -> global
            if (it1 && r1 < nrows) {
                float2 v[NA];
                const float2* row = rows + r1 * ROWP;
                v[0] = row[aj1];
                pw_each<NA>(tw1[j1], [&](auto aa, float2 w) {
                    constexpr int a = decltype(aa)::value + 1;
                    v[a] = c_mulc(row[aj1 + a * (M1 + NB)], w);
                });
                idft<NA>(v);
                float2* o = data + (size_t)(r1 ? pr.y : pr.x) * N2 + j1;
                static_for<NA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    o[a * M1] = v[a];
                });
            }
            lds_barrier();
            pr_cur = pr_nxt;
            pr_nxt = pr_nn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass Cm: c2 forward, x folded mask, c2 inverse of TC contiguous BAND columns of W for one c1-position q, in one visit
// (passes C and C' of the channel-first order fused; the radix-RB item multiplies its own outputs and transforms them
// back, so the tile needs two barriers instead of four).  Tile layout and twiddles as fkf_passC.
// ---------------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(G::THRC) void fkf_passCm(FkDev P, FkFastDev F, FkTfDev T, int tbase, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C2A * (G::C2B + 1) * G::TC;
    constexpr int RA = G::C2A, RB = G::C2B, TC = G::TC;
    const int tid = threadIdx.x;
    const int hi = tid / TC, tt = tid % TC;            // hi = j (radix-RA items) or g (radix-RB items)
    const bool actA = hi < RB;                         // items of the radix-RA steps
    const bool actB = hi < RA;                         // items of the radix-RB step
    for (int i = tid; i < (G::CTREE ? RB : RA * RB); i += G::THRC) twl[i] = F.twC[(G::CTREE ? RB : 0) + i];
    __syncthreads();
    // tile t -> (q, column block): per q the N1 sub-row blocks' band columns (nb1 strips each), then the Nyquist strip
    const int nb1 = T.bw / TC, tq = G::N1 * nb1 + (T.col_nyq >= 0 ? 1 : 0);
    const size_t LC = (size_t)T.Lc;
    auto tile_qp = [&](int t, int& q, int& p0) {
        q = t / tq;
        const int u = t - q * tq;
        if (u < G::N1 * nb1) {
            const int q1 = u / nb1;
            p0 = q1 * T.RW + (u - q1 * nb1) * TC;
        } else
            p0 = T.col_nyq;
    };
    float2 pfA[RA], pfB[RA];
    float mk[RB];
    auto issue = [&](float2 (&pf)[RA], int t) {
        int q, p0;
        tile_qp(t, q, p0);
        const float2* bh = T.W + ((size_t)q * G::C2 + hi) * LC + p0 + tt;
        static_for<RA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            pf[a] = bh[(size_t)(a * RB) * LC];
        });
    };
    // The band mask of a tile is mostly zeros for the speed-fan designs (inside the band only the wavenumbers of the fan pass:
    // ~6 of a thread's 32 rows): one word per (tile, g) says which rows to read, loaded a tile before the mask values
    unsigned bits_nxt = 0xFFFFFFFFu;
    auto issue_bits = [&](int t) { bits_nxt = T.cmlive[(size_t)t * RA + hi]; };
    auto issue_mask = [&](int t, unsigned bits) {
        int q, p0;
        tile_qp(t, q, p0);
        const float* mp = T.cmask + ((size_t)q * G::C2 + (size_t)hi * RB) * LC + p0 + tt;
        static_for<RB>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            mk[b] = ((bits >> b) & 1u) ? mp[(size_t)b * LC] : 0.f;
        });
    };
    const int gstep = gridDim.x;
    auto body = [&](float2 (&pf)[RA], int t) {
        int q, p0;
        tile_qp(t, q, p0);
        float2* bh = T.W + ((size_t)q * G::C2 + hi) * LC + p0 + tt;
        if (actA) {
            dft<RA>(pf);
            float2 pw[RA];
            if constexpr (G::CTREE) pw_tree<RA>(twl[hi], pw);
            static_for<RA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                const float2 w = G::CTREE ? pw[a] : twl[a * RB + hi];
                tile[(hi + a * (RB + 1)) * TC + tt] = (a == 0) ? pf[0] : c_mul(pf[a], w);
            });
        }
        lds_barrier();
        if (t + 2 * gstep < ntiles && actA) issue(pf, t + 2 * gstep);     // pf is free: S1 consumed it
        if (actB) {
            float2 v[RB];
            static_for<RB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                v[b] = tile[(hi * (RB + 1) + b) * TC + tt];
            });
            dft<RB>(v);
            static_for<RB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                v[b] = c_scale(v[b], mk[b]);
            });
            idft<RB>(v);
            static_for<RB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                tile[(hi * (RB + 1) + b) * TC + tt] = v[b];
            });
            if (t + gstep < ntiles) issue_mask(t + gstep, bits_nxt);      // mk is free; ahead of this tile's stores
            if (t + 2 * gstep < ntiles) issue_bits(t + 2 * gstep);
        }
        lds_barrier();
        if (actA) {
            float2 v[RA];
            static_for<RA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                v[a] = tile[(hi + a * (RB + 1)) * TC + tt];
            });
            float2 pw[RA];
            if constexpr (G::CTREE) pw_tree<RA>(twl[hi], pw);
            static_for<RA - 1>([&](auto aa) {
                constexpr int a = decltype(aa)::value + 1;
                v[a] = c_mulc(v[a], G::CTREE ? pw[a] : twl[a * RB + hi]);
            });
            idft<RA>(v);
            static_for<RA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                bh[(size_t)(a * RB) * LC] = v[a];
            });
        }
    };
    int t = tbase + blockIdx.x;
    if (t < ntiles && actA) issue(pfA, t);
    if (t + gstep < ntiles && actA) issue(pfB, t + gstep);
    if (t < ntiles && actB) {
        issue_bits(t);
        issue_mask(t, bits_nxt);
        if (t + gstep < ntiles) issue_bits(t + gstep);
    }
    for (; t < ntiles; t += 2 * gstep) {
        body(pfA, t);
        if (t + gstep < ntiles) body(pfB, t + gstep);
    }
}

}  // namespace d4w
