// Shape-specialised f-k pass kernels ("fat-stage" register FFTs).
//
// Same five-pass factorisation, tables and digit-reversed in-place conventions as the generic
// kernels in fk_filter.hip, but every axis is cut into at most three radices (<= 32) that are
// compile-time constants:
//   * a whole radix-R butterfly lives in one thread's registers; the FIRST butterfly of a pass is
//     applied directly to the registers the global prefetch landed in and the LAST one feeds the
//     global stores, so a two-radix axis costs ONE LDS exchange (write, barrier, read) instead of
//     tile-in + one round trip per radix + tile-out;
//   * pass B fuses [last forward radix | real-spectrum pair op x mask | first inverse radix] on
//     one thread (a group of NC consecutive positions and its Hermitian partner group);
//   * all index arithmetic is constant-folded, twiddles of the exchange stages sit in LDS, the
//     four-step twiddles W_nx^(c2 kc1) are wave-uniform scalar loads;
//   * persistent workgroups prefetch the next tile's butterfly inputs straight into registers
//     while the current tile is in its LDS phase.
// Instantiated for the shapes listed in fk_filter.hip (kFastShapes); any other shape runs the
// generic kernels.
#pragma once
#include "fft_lds.h"

#ifndef D4W_ABL
#define D4W_ABL 0      // timing-ablation switch for scripts/ab.sh experiments (0 = product code)
#endif

namespace d4w {

template <int C1_, int C2A_, int C2B_, int N1_, int NA_, int NB_, int NC_, int TA_, int TC_, int THRA_,
          int THRC_, int THRB_, bool CTREE_ = false, int WAVES_B_ = 1, int C2X_ = 1>
struct FkFastCfg {
    // C2X > 1: the c2 axis has a factor C2X with prime factors > 31 and its whole sub-transform runs as the generic Bluestein
    // pass C (fk_filter.hip); the configuration then sets C2A = C2B = 1 and only passes A and B are the kernels below
    static constexpr int C2X = C2X_;
    static constexpr int WAVES_B = WAVES_B_;   // min waves per SIMD pass B is compiled for (register cap)
    // CTREE: pass C forms W_C2^(j a) as powers of W_C2^j in registers instead of reading a
    // [C2A][C2B] table from LDS (frees C2A*C2B*8 bytes of LDS per workgroup)
    static constexpr bool CTREE = CTREE_;
    static constexpr int C1 = C1_, C2A = C2A_, C2B = C2B_, C2 = C2A_ * C2B_ * C2X_, NX = C1_ * C2A_ * C2B_ * C2X_;
    static constexpr int N1 = N1_, NA = NA_, NB = NB_, NC = NC_, N2 = NA_ * NB_ * NC_, M = N1_ * NA_ * NB_ * NC_;
    static constexpr int TA = TA_, TC = TC_, THRA = THRA_, THRC = THRC_, THRB = THRB_;
    // pass A: tile [C1][N1][TA] + double-buffered four-step twiddle strip [2][N1][TA]
    static constexpr size_t ldsA = (size_t)(C1 * N1 * TA + 2 * N1 * TA) * sizeof(float2) + 2 * N1 * sizeof(int);
    // pass C: tile [(C2A)(C2B + 1)][TC] (one pad row per C2B rows) + twiddles [C2A][C2B]
    static constexpr size_t ldsC = (size_t)(C2A * (C2B + 1) * TC + (CTREE_ ? C2B : C2A * C2B)) * sizeof(float2) +
                                   (size_t)C1 * C2A * sizeof(unsigned);
    static_assert(C2B_ <= 32, "pass C keeps one live bit per radix-C2B output in a 32-bit word");
    // pass B: two rows of N2 (+ one pad element per NC) + tw1 [NB*NC] + tw2 [NB][NC]
    static constexpr int ROWP = N2 + NA * NB;
    static constexpr size_t ldsB = (size_t)(2 * ROWP + 2 * NB * NC) * sizeof(float2);
    static_assert(N1 * TA <= THRA && C1 * TA <= THRA, "pass A: one butterfly per thread");
    static_assert(C2B * TC <= THRC && C2A * TC <= THRC, "pass C: one butterfly per thread");
    static_assert(2 * NB * NC <= THRB && NA * NB <= THRB, "pass B: one S1 / mid item per thread");
    static_assert(N2 % TA == 0 && M % TC == 0, "tiles must divide the axes");
};

// extra device tables of a fast plan (FkDev carries the generic ones)
struct FkFastDev {
    const float2* twC;    // [C2A][C2B]  W_C2^(j a)
    const float2* twB1;   // [NB*NC]     W_N2^j
    const float2* twB2;   // [NB][NC]    W_(NB*NC)^(j2 b)
    // Dead-row pruning (exact): a wavenumber row whose folded mask is all zero, together with
    // its Hermitian partner row, comes out of pass B as zeros whatever it held, so pass C does
    // not write it, pass B skips its pairs and pass C' reads zeros instead of memory.
    const unsigned* live; // [C1][C2A] bit b = row position q*C2 + g*C2B + b is live; NULL = all live
    const int2* pairs;    // pass-B work list actually run (all pairs, or the live ones)
};

// Runtime geometry of the distributed (channel-sharded) layouts, DESIGN.md 6.  The single-device passes (MODE 0)
// ignore it: their pitches and strides are compile-time constants.
//   MODE 1 of pass A  = TIME phase on the local rows [nrows][M]: the (c1, n1) tile keeps its shape but the c1
//                       index walks C1 CONSECUTIVE rows and no channel transform is applied; sub-row q1 of every row
//                       is written to / read from the PACKED exchange buffer, destination rank major:
//                       element q1_off[q1] + row * q1_pitch[q1] (what the all-to-all sends without any repacking);
//   MODE 2 of pass A  = c1 transform on the slab [nx][ncols] (all channels x the columns of the owned sub-rows): the
//                       n1 index walks N1 ADJACENT column strips and no time transform is applied;
//   MODE 1 of pass C / pass B = the same kernels on the slab (row pitch ncols, sub-rows keyed r * nq + jq).
struct FkGeo {
    int nrows, ncols;            // valid rows of the local block / columns of the slab
    int nq;                      // owned sub-rows per channel (pass B), blocks per row (pass C)
    const int* q1_off;           // [N1] MODE 1 of pass A
    const int* q1_pitch;         // [N1]
    const int* q1_of;            // [nq] pass B: n1 position of local sub-row jq
};

template <int R>
__device__ __forceinline__ void pw_tree(float2 w1, float2 (&pw)[R]) {
    pw[0] = make_float2(1.f, 0.f);
    if constexpr (R > 1) pw[1] = w1;
    static_for<(R > 2 ? R - 2 : 0)>([&](auto qq) {
        constexpr int q = decltype(qq)::value + 2;
        pw[q] = c_mul(pw[q / 2], pw[q - q / 2]);
    });
}

// The same powers handed to `f(a, w^a)`, a = 1 .. R-1, one at a time: w^a = (w^4)^(a / 4) w^(a % 4) from seven kept values
// instead of an array of R (the tree above keeps all R powers live across the butterfly stores: 2 R registers; this form
// 14, products at most three deep like the tree's).
template <int R, class F>
__device__ __forceinline__ void pw_each(float2 w1, F&& f) {
    constexpr int NH = (R + 3) / 4;
    float2 lo[4], hi[NH > 0 ? NH : 1];
    lo[0] = make_float2(1.f, 0.f);
    lo[1] = w1;
    lo[2] = c_mul(w1, w1);
    lo[3] = c_mul(lo[2], w1);
    hi[0] = make_float2(1.f, 0.f);
    if constexpr (NH > 1) hi[1] = c_mul(lo[2], lo[2]);
    static_for<(NH > 2 ? NH - 2 : 0)>([&](auto kk) {
        constexpr int k = decltype(kk)::value + 2;
        hi[k] = c_mul(hi[k / 2], hi[k - k / 2]);
    });
    static_for<(R > 1 ? R - 1 : 0)>([&](auto aa) {
        constexpr int a = decltype(aa)::value + 1;
        if constexpr (a < 4) f(aa, lo[a]);
        else if constexpr (a % 4 == 0) f(aa, hi[a / 4]);
        else f(aa, c_mul(hi[a / 4], lo[a % 4]));
    });
}

// ---------------------------------------------------------------------------------------------
// pass A forward: tile = (all c1) x (all n1) x TA columns of one c2.
//   S1 item (n1, tt): DFT over c1 in the prefetch registers                       -> LDS
//   S2 item (q,  tt): DFT over n1, x W_M^(n2 k1) (strip staged in LDS by the S1 items)
//                     x W_nx^(c2 kc1(q)) (one prefetched value per thread)        -> global
// (The four-step twiddles must not be wave-uniform scalar loads inside the tile loop: 25
//  s_loads per tile on the critical path cost 2 ms of a 5.6 ms pass.)
// ---------------------------------------------------------------------------------------------
template <int N>
struct FkPrefetchA {          // what one thread prefetches for one pass-A tile
    float2 pf[N];
    float2 tw;                // strip element W_M^(n2 k1) this thread stages (forward: its own S1 item)
    float2 win;               // packed Tukey window of the item's column (forward, TAPER)
    float2 tc;                // W_nx^(c2 kc1(q)) of the thread's q
};

template <class G, bool TAPER, int MODE = 0>
__global__ __launch_bounds__(G::THRA) void fkf_passA_fwd(FkDev P, const float2* __restrict__ src,
                                                         float2* __restrict__ dst, int tbase, int ntiles, int sw, int sbase,
                                                         FkGeo geo = FkGeo()) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C1 * G::N1 * G::TA;
    // tile u -> (c2 = u / sw, column block sbase + u % sw): sw = N2 / TA, sbase = 0 is the plain order; a slab
    // (sbase, sw) restricts the pass to sw column blocks of every n1 sub-row (see fkf_passC)
    const int tid = threadIdx.x;
    const int hi = tid / G::TA, tt = tid % G::TA;      // hi = n1 (S1) or q (S2)
    const bool act1 = hi < G::N1, act2 = hi < G::C1;
    const int gstep = gridDim.x;
    typedef FkPrefetchA<G::C1> Pre;
    Pre A, B;       // two register sets, see fkf_passC
    int* qtab = reinterpret_cast<int*>(twl + 2 * G::N1 * G::TA);     // MODE 1: [N1] offsets, [N1] pitches of the packed buffer
    if constexpr (MODE == 1) {
        for (int i = tid; i < G::N1; i += G::THRA) {
            qtab[i] = geo.q1_off[i];
            qtab[G::N1 + i] = geo.q1_pitch[i];
        }
        __syncthreads();
    }
    // element offset of (c1 index, tile) for the loads; MODE 0: row c1 C2 + c2 of [nx][M], column n1 N2 + b0 + tt
    auto issue = [&](Pre& R, int t) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        if (act1) {
            if constexpr (MODE == 0) {
                const int col = hi * G::N2 + b0 + tt;
                const float2* p = src + (size_t)c2 * G::M + col;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    R.pf[c1] = fk_ldg<0>(p + (size_t)c1 * G::C2 * G::M);
                });
                R.tw = P.twt[col];
                if (TAPER) R.win = P.win[col];
            } else if constexpr (MODE == 1) {            // C1 consecutive local rows, group c2
                const int col = hi * G::N2 + b0 + tt;
                const float2* p = src + (size_t)c2 * G::C1 * G::M + col;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    R.pf[c1] = (c2 * G::C1 + c1 < geo.nrows) ? p[(size_t)c1 * G::M] : make_float2(0.f, 0.f);
                });
                R.tw = P.twt[col];
                if (TAPER) R.win = P.win[col];
            } else {                                     // slab: N1 adjacent strips of column block b0 / TA
                const int col = (b0 / G::TA) * (G::N1 * G::TA) + hi * G::TA + tt;
                const float2* p = src + (size_t)c2 * geo.ncols + col;
                const bool ok = col < geo.ncols;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    R.pf[c1] = ok ? p[(size_t)c1 * G::C2 * geo.ncols] : make_float2(0.f, 0.f);
                });
            }
        }
        if (MODE != 1 && act2) R.tc = P.twc[hi * G::C2 + c2];
    };
    int par = 0;
    // Two register sets, each prefetching TWO tiles ahead: tile t + 2 gstep is loaded into the set of tile t as
    // soon as S1 has consumed it, i.e. BEFORE tile t's stores are issued.  A wave's vector-memory operations
    // retire in order, so a prefetch issued after a tile's stores (one tile ahead, at the top of the next
    // iteration) could not be consumed before those stores were acknowledged -- the store latency of a
    // write-heavy pass then sat on the loop's critical path.
    auto body = [&](Pre& R, int t) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        float2* tw_cur = twl + par * (G::N1 * G::TA);
        if (act1) {
            if (TAPER && MODE != 2) {
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    R.pf[c1].x *= R.win.x;
                    R.pf[c1].y *= R.win.y;
                });
            }
            if constexpr (MODE != 1) dft<G::C1>(R.pf);
            static_for<G::C1>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                tile[(q * G::N1 + hi) * G::TA + tt] = R.pf[q];
            });
            if constexpr (MODE != 2) tw_cur[hi * G::TA + tt] = R.tw;
        }
        const float2 tc_cur = R.tc;
        lds_barrier();
        if (t + 2 * gstep < ntiles) issue(R, t + 2 * gstep);       // R is free: S1 consumed it
        float2 v[G::N1];
        if (act2) {
            static_for<G::N1>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                v[k] = tile[(hi * G::N1 + k) * G::TA + tt];
            });
        }
        lds_barrier();
        if (act2) {
            if constexpr (MODE == 0) {
                dft<G::N1>(v);
                float2* o = dst + ((size_t)hi * G::C2 + c2) * G::M + b0 + tt;
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    fk_stg<0>(o + q1 * G::N2, c_mul(v[q1], c_mul(tw_cur[q1 * G::TA + tt], tc_cur)));
                });
            } else if constexpr (MODE == 1) {
                dft<G::N1>(v);
                const int row = c2 * G::C1 + hi;
                if (row < geo.nrows) {
                    static_for<G::N1>([&](auto kk) {
                        constexpr int q1 = decltype(kk)::value;
                        dst[(size_t)qtab[q1] + (size_t)row * qtab[G::N1 + q1] + b0 + tt] = c_mul(v[q1], tw_cur[q1 * G::TA + tt]);
                    });
                }
            } else {
                const int colb = (b0 / G::TA) * (G::N1 * G::TA) + tt;
                float2* o = dst + ((size_t)hi * G::C2 + c2) * geo.ncols + colb;
                static_for<G::N1>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    if (colb + k * G::TA < geo.ncols) o[k * G::TA] = c_mul(v[k], tc_cur);
                });
            }
        }
        par ^= 1;
    };
    int t = tbase + blockIdx.x;     // tiles [tbase, ntiles): a sub-range when passes are chunked
    if (t < ntiles) issue(A, t);
    if (t + gstep < ntiles) issue(B, t + gstep);
    for (; t < ntiles; t += 2 * gstep) {
        body(A, t);
        if (t + gstep < ntiles) body(B, t + gstep);
    }
}

// pass A inverse (axes in the opposite order, so that each thread again needs ONE W_nx value):
//   S1' item (q,  tt): x conj(W_M^(n2 k1) W_nx^(c2 kc1(q))), inverse DFT over q1 (n1 axis) -> LDS
//   S2' item (n1, tt): inverse DFT over q (c1 axis), x 1/(nx M)                            -> global
// The W_M strip element of tile i+1 travels with the data prefetch of tile i+1 (issued at the top
// of iteration i) and is written to the other half of the strip double buffer before the first
// barrier of iteration i+... see the body: it is in LDS one full iteration before it is read.
template <class G, int MODE = 0>
__global__ __launch_bounds__(G::THRA) void fkf_passA_inv(FkDev P, float2* __restrict__ data, int tbase, int ntiles, int sw, int sbase,
                                                         FkGeo geo = FkGeo(), const float2* __restrict__ packed = nullptr) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C1 * G::N1 * G::TA;
    constexpr int STRIP = G::N1 * G::TA;
    const int tid = threadIdx.x;
    const int hi = tid / G::TA, tt = tid % G::TA;      // hi = q (S1') or n1 (S2'); also q1 for the strip
    const bool act1 = hi < G::C1, act2 = hi < G::N1;
    const int gstep = gridDim.x;
    typedef FkPrefetchA<G::N1> Pre;
    Pre A, B;
    int* qtab = reinterpret_cast<int*>(twl + 2 * STRIP);
    if constexpr (MODE == 1) {
        for (int i = tid; i < G::N1; i += G::THRA) {
            qtab[i] = geo.q1_off[i];
            qtab[G::N1 + i] = geo.q1_pitch[i];
        }
        __syncthreads();
    }
    auto issue = [&](Pre& R, int t) {                   // data + W_nx value + strip element of tile t
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        // strip element FIRST: it is consumed one iteration before the data, and vmcnt retires
        // loads in order -- waiting for the oldest load of a set does not wait for the rest
        if (MODE != 2 && act2) R.tw = P.twt[hi * G::N2 + b0 + tt];
        if (act1) {
            if constexpr (MODE == 0) {
                const float2* p = data + ((size_t)hi * G::C2 + c2) * G::M + b0 + tt;
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = fk_ldg<D4W_FK_NT_AINV>(p + q1 * G::N2);
                });
                R.tc = P.twc[hi * G::C2 + c2];
            } else if constexpr (MODE == 1) {            // sub-rows of local row c2 C1 + hi from the packed buffer
                const int row = c2 * G::C1 + hi;
                const bool ok = row < geo.nrows;
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = ok ? packed[(size_t)qtab[q1] + (size_t)row * qtab[G::N1 + q1] + b0 + tt] : make_float2(0.f, 0.f);
                });
            } else {                                     // slab row q C2 + c2, N1 adjacent strips
                const int colb = (b0 / G::TA) * STRIP + tt;
                const float2* p = data + ((size_t)hi * G::C2 + c2) * geo.ncols + colb;
                static_for<G::N1>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    R.pf[k] = (colb + k * G::TA < geo.ncols) ? p[k * G::TA] : make_float2(0.f, 0.f);
                });
                R.tc = P.twc[hi * G::C2 + c2];
            }
        }
    };
    int par = 0;
    // strip of tile i lives in twl[par(i)]: written during iteration i-1 (before its first barrier)
    auto body = [&](Pre& R, Pre& Rn, int t, bool first) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        const float2* tw_cur = twl + par * STRIP;
        const bool more = (t + gstep < ntiles);
        if constexpr (MODE != 2) {
            if (first) {                                    // very first tile of this workgroup
                if (act2) twl[par * STRIP + hi * G::TA + tt] = R.tw;
                __syncthreads();
            }
            // the NEXT tile's registers were issued one iteration ago (or in the prologue): its strip
            // element is already here, publish it; then start the loads of the tile after that
            if (more && act2) twl[(par ^ 1) * STRIP + hi * G::TA + tt] = Rn.tw;
        }
        if (act1) {
            if constexpr (MODE == 0) {
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = c_mulc(R.pf[q1], c_mul(tw_cur[q1 * G::TA + tt], R.tc));
                });
                idft<G::N1>(R.pf);
            } else if constexpr (MODE == 1) {
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = c_mulc(R.pf[q1], tw_cur[q1 * G::TA + tt]);
                });
                idft<G::N1>(R.pf);
            } else {
                static_for<G::N1>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    R.pf[k] = c_mulc(R.pf[k], R.tc);
                });
            }
            static_for<G::N1>([&](auto kk) {
                constexpr int n1 = decltype(kk)::value;
                tile[(hi * G::N1 + n1) * G::TA + tt] = R.pf[n1];
            });
        }
        lds_barrier();
        // R's registers are free now: prefetch tile t + 2 gstep into them
        if (t + 2 * gstep < ntiles) issue(R, t + 2 * gstep);
        float2 v[G::C1];
        if (act2) {
            static_for<G::C1>([&](auto cc) {
                constexpr int q = decltype(cc)::value;
                v[q] = tile[(q * G::N1 + hi) * G::TA + tt];
            });
        }
        lds_barrier();
        if (act2) {
            if constexpr (MODE == 0) {
                idft<G::C1>(v);
                float2* o = data + (size_t)c2 * G::M + hi * G::N2 + b0 + tt;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    fk_stg<D4W_FK_NT_AINV>(o + (size_t)c1 * G::C2 * G::M, c_scale(v[c1], P.scale));
                });
            } else if constexpr (MODE == 1) {            // C1 consecutive local rows, no channel transform
                float2* o = data + (size_t)c2 * G::C1 * G::M + hi * G::N2 + b0 + tt;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    if (c2 * G::C1 + c1 < geo.nrows) o[(size_t)c1 * G::M] = c_scale(v[c1], P.scale);
                });
            } else {
                idft<G::C1>(v);
                const int col = (b0 / G::TA) * STRIP + hi * G::TA + tt;
                if (col < geo.ncols) {
                    float2* o = data + (size_t)c2 * geo.ncols + col;
                    static_for<G::C1>([&](auto cc) {
                        constexpr int c1 = decltype(cc)::value;
                        o[(size_t)c1 * G::C2 * geo.ncols] = c_scale(v[c1], P.scale);
                    });
                }
            }
        }
        par ^= 1;
    };
    int t = tbase + blockIdx.x;     // tiles [tbase, ntiles): a sub-range when passes are chunked
    if (t < ntiles) issue(A, t);
    if (t + gstep < ntiles) issue(B, t + gstep);
    bool first = true;
    for (; t < ntiles; t += 2 * gstep) {
        body(A, B, t, first);
        first = false;
        if (t + gstep < ntiles) body(B, A, t + gstep, false);
    }
}

// Inverse time phase of REAL rows (fkf_passA_inv MODE 1: C1 consecutive local rows per tile, input from the packed buffer, no
// channel transform) whose epilogue forms the analytic-signal quantity from H[x] and x instead of storing H[x] for a separate
// combine pass (d4w_analytic_long_f32: 36 -> 28 bytes per sample).  mode 0: |x + i H|; 2: 10 log10(|z|^2 / var[row]);
// 4: |z| / sqrt(var[row]).  The tile's x samples are loaded ONE TILE AHEAD, after the current tile's values are formed and
// before its stores go out (a load issued after the stores would wait for their acknowledgement).
template <class G>
__global__ __launch_bounds__(G::THRA) void fkf_passA_inv_env(FkDev P, float2* __restrict__ out, int tbase, int ntiles, int sw, int sbase,
                                                             FkGeo geo, const float2* __restrict__ packed, const float2* __restrict__ xsrc,
                                                             int mode, const float* __restrict__ var) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C1 * G::N1 * G::TA;
    constexpr int STRIP = G::N1 * G::TA;
    const int tid = threadIdx.x;
    const int hi = tid / G::TA, tt = tid % G::TA;      // hi = row within the tile (S1') or n1 (S2'); also q1 for the strip
    const bool act1 = hi < G::C1, act2 = hi < G::N1;
    const int gstep = gridDim.x;
    typedef FkPrefetchA<G::N1> Pre;
    Pre A, B;
    int* qtab = reinterpret_cast<int*>(twl + 2 * STRIP);
    for (int i = tid; i < G::N1; i += G::THRA) {
        qtab[i] = geo.q1_off[i];
        qtab[G::N1 + i] = geo.q1_pitch[i];
    }
    __syncthreads();
    auto issue = [&](Pre& R, int t) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        if (act2) R.tw = P.twt[hi * G::N2 + b0 + tt];
        if (act1) {
            const int row = c2 * G::C1 + hi;
            const bool ok = row < geo.nrows;
            static_for<G::N1>([&](auto kk) {
                constexpr int q1 = decltype(kk)::value;
                R.pf[q1] = ok ? packed[(size_t)qtab[q1] + (size_t)row * qtab[G::N1 + q1] + b0 + tt] : make_float2(0.f, 0.f);
            });
        }
    };
    float2 xv[G::C1];                                   // x of the tile's output positions (S2' item: n1 = hi, column tt)
    auto issue_x = [&](int t) {
        if (!act2) return;
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        const float2* xp = xsrc + (size_t)c2 * G::C1 * G::M + hi * G::N2 + b0 + tt;
        static_for<G::C1>([&](auto cc) {
            constexpr int c1 = decltype(cc)::value;
            xv[c1] = (c2 * G::C1 + c1 < geo.nrows) ? xp[(size_t)c1 * G::M] : make_float2(0.f, 0.f);
        });
    };
    int par = 0;
    auto body = [&](Pre& R, Pre& Rn, int t, bool first) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        const float2* tw_cur = twl + par * STRIP;
        const bool more = (t + gstep < ntiles);
        if (first) {
            if (act2) twl[par * STRIP + hi * G::TA + tt] = R.tw;
            __syncthreads();
        }
        if (more && act2) twl[(par ^ 1) * STRIP + hi * G::TA + tt] = Rn.tw;
        if (act1) {
            static_for<G::N1>([&](auto kk) {
                constexpr int q1 = decltype(kk)::value;
                R.pf[q1] = c_mulc(R.pf[q1], tw_cur[q1 * G::TA + tt]);
            });
            idft<G::N1>(R.pf);
            static_for<G::N1>([&](auto kk) {
                constexpr int n1 = decltype(kk)::value;
                tile[(hi * G::N1 + n1) * G::TA + tt] = R.pf[n1];
            });
        }
        lds_barrier();
        if (t + 2 * gstep < ntiles) issue(R, t + 2 * gstep);
        float2 v[G::C1];
        if (act2) {
            static_for<G::C1>([&](auto cc) {
                constexpr int q = decltype(cc)::value;
                v[q] = tile[(q * G::N1 + hi) * G::TA + tt];
            });
        }
        lds_barrier();
        if (act2) {
            static_for<G::C1>([&](auto cc) {
                constexpr int c1 = decltype(cc)::value;
                const float2 h = c_scale(v[c1], P.scale);
                const float2 x2 = xv[c1];
                float ex = fmaf(x2.x, x2.x, h.x * h.x), ey = fmaf(x2.y, x2.y, h.y * h.y);
                if (mode == 0) {
                    ex = sqrtf(ex);
                    ey = sqrtf(ey);
                } else {
                    const int row = min(c2 * G::C1 + c1, geo.nrows - 1);
                    const float iv = 1.0f / var[row];
                    if (mode == 2) {
                        ex = 10.0f * log10f(ex * iv);
                        ey = 10.0f * log10f(ey * iv);
                    } else {
                        ex = sqrtf(ex * iv);
                        ey = sqrtf(ey * iv);
                    }
                }
                v[c1] = make_float2(ex, ey);
            });
            if (more) issue_x(t + gstep);                 // xv is free: next tile's x, ahead of this tile's stores
            float2* o = out + (size_t)c2 * G::C1 * G::M + hi * G::N2 + b0 + tt;
            static_for<G::C1>([&](auto cc) {
                constexpr int c1 = decltype(cc)::value;
                if (c2 * G::C1 + c1 < geo.nrows) o[(size_t)c1 * G::M] = v[c1];
            });
        }
        par ^= 1;
    };
    int t = tbase + blockIdx.x;
    if (t < ntiles) {
        issue(A, t);
        issue_x(t);
    }
    if (t + gstep < ntiles) issue(B, t + gstep);
    bool first = true;
    for (; t < ntiles; t += 2 * gstep) {
        body(A, B, t, first);
        first = false;
        if (t + gstep < ntiles) body(B, A, t + gstep, false);
    }
}

// pass A inverse with the row statistics of the filtered block in its epilogue (what the matched filter
// normalises by, detect.py:157: mean and max|.| of every output row), so the consumer does not re-read
// the block for them.  A thread's S2' item holds one packed sample of every c1 row; the tile order is
// cut into RUNS of `run` consecutive tiles (same c2, i.e. the same C1 rows) that one workgroup walks in
// order, accumulating per-thread partial sums / maxima in registers and reducing them once per run
// (wave shuffle + one float atomicAdd / integer atomicMax per row and wave).
// MODE 1: the distributed plan's inverse time phase (fkf_passA_inv MODE 1: C1 consecutive local rows per tile, input from
// the packed exchange buffer, no channel transform) with the same epilogue; tile0 = first tile of the row chunk.
template <class G, int MODE = 0>
__global__ __launch_bounds__(G::THRA) void fkf_passA_inv_stats(FkDev P, float2* __restrict__ data, int run, int nruns,
                                                               double* __restrict__ rowmean,
                                                               unsigned* __restrict__ rowmaxbits, int sw, int sbase,
                                                               FkGeo geo = FkGeo(), const float2* __restrict__ packed = nullptr,
                                                               int tile0 = 0) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C1 * G::N1 * G::TA;
    constexpr int STRIP = G::N1 * G::TA;
    const int tid = threadIdx.x;
    const int hi = tid / G::TA, tt = tid % G::TA;
    const bool act1 = hi < G::C1, act2 = hi < G::N1;
    const int myruns = (nruns - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nseq = max(myruns, 0) * run;              // tiles this workgroup walks, in order
    // Run r of a c2 row group takes the column blocks r, r + R, r + 2 R, ... (R = sw / run runs per group): the R workgroups
    // that walk one group side by side then touch ADJACENT 128-byte strips of the same rows at the same time (DRAM page
    // locality), instead of strips a whole run apart
    const int R = max(sw / max(run, 1), 1);
    auto tile_of = [&](int sq) {
        const int k = sq / run, j = sq - k * run;
        const int rid = (int)blockIdx.x + k * (int)gridDim.x;
        const int grp = rid / R, r = rid - grp * R;
        return tile0 + grp * (R * run) + r + j * R;
    };
    int* qtab = reinterpret_cast<int*>(twl + 2 * STRIP);
    if constexpr (MODE == 1) {
        for (int i = tid; i < G::N1; i += G::THRA) {
            qtab[i] = geo.q1_off[i];
            qtab[G::N1 + i] = geo.q1_pitch[i];
        }
        __syncthreads();
    }
    typedef FkPrefetchA<G::N1> Pre;
    Pre A, B;
    auto issue = [&](Pre& R, int t) {
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        if (act2) R.tw = P.twt[hi * G::N2 + b0 + tt];
        if (act1) {
            if constexpr (MODE == 0) {
                const float2* p = data + ((size_t)hi * G::C2 + c2) * G::M + b0 + tt;
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = fk_ldg<D4W_FK_NT_AINV>(p + q1 * G::N2);
                });
                R.tc = P.twc[hi * G::C2 + c2];
            } else {
                const int row = c2 * G::C1 + hi;
                const bool ok = row < geo.nrows;
                static_for<G::N1>([&](auto kk) {
                    constexpr int q1 = decltype(kk)::value;
                    R.pf[q1] = ok ? packed[(size_t)qtab[q1] + (size_t)row * qtab[G::N1 + q1] + b0 + tt] : make_float2(0.f, 0.f);
                });
            }
        }
    };
    float asum[G::C1], amax[G::C1];
    static_for<G::C1>([&](auto cc) { asum[decltype(cc)::value] = 0.f; amax[decltype(cc)::value] = 0.f; });
    const double inv_ns = 1.0 / (double)P.d.ns;
    int par = 0;
    auto body = [&](Pre& R, Pre& Rn, int sq, bool first) {
        const int t = tile_of(sq);
        const int c2 = t / sw, b0 = (sbase + t % sw) * G::TA;
        const float2* tw_cur = twl + par * STRIP;
        const bool more = (sq + 1 < nseq);
        if (first) {
            if (act2) twl[par * STRIP + hi * G::TA + tt] = R.tw;
            __syncthreads();
        }
        if (more && act2) twl[(par ^ 1) * STRIP + hi * G::TA + tt] = Rn.tw;
        if (act1) {
            static_for<G::N1>([&](auto kk) {
                constexpr int q1 = decltype(kk)::value;
                if constexpr (MODE == 0) R.pf[q1] = c_mulc(R.pf[q1], c_mul(tw_cur[q1 * G::TA + tt], R.tc));
                else R.pf[q1] = c_mulc(R.pf[q1], tw_cur[q1 * G::TA + tt]);
            });
            idft<G::N1>(R.pf);
            static_for<G::N1>([&](auto kk) {
                constexpr int n1 = decltype(kk)::value;
                tile[(hi * G::N1 + n1) * G::TA + tt] = R.pf[n1];
            });
        }
        lds_barrier();
        if (sq + 2 < nseq) issue(R, tile_of(sq + 2));
        float2 v[G::C1];
        if (act2) {
            static_for<G::C1>([&](auto cc) {
                constexpr int q = decltype(cc)::value;
                v[q] = tile[(q * G::N1 + hi) * G::TA + tt];
            });
        }
        lds_barrier();
        if (act2) {
            if constexpr (MODE == 0) {
                idft<G::C1>(v);
                float2* o = data + (size_t)c2 * G::M + hi * G::N2 + b0 + tt;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    const float2 y2 = c_scale(v[c1], P.scale);
                    fk_stg<D4W_FK_NT_AINV>(o + (size_t)c1 * G::C2 * G::M, y2);
                    asum[c1] += y2.x + y2.y;
                    amax[c1] = fmaxf(amax[c1], fmaxf(fabsf(y2.x), fabsf(y2.y)));
                });
            } else {
                float2* o = data + (size_t)c2 * G::C1 * G::M + hi * G::N2 + b0 + tt;
                static_for<G::C1>([&](auto cc) {
                    constexpr int c1 = decltype(cc)::value;
                    const float2 y2 = c_scale(v[c1], P.scale);
                    if (c2 * G::C1 + c1 < geo.nrows) o[(size_t)c1 * G::M] = y2;
                    asum[c1] += y2.x + y2.y;
                    amax[c1] = fmaxf(amax[c1], fmaxf(fabsf(y2.x), fabsf(y2.y)));
                });
            }
        }
        if ((sq % run) == run - 1) {                     // end of a run: its C1 rows are complete for these columns
            // wave partials -> LDS (the tile is free between the second barrier and the next S1' writes) -> ONE atomic pair per
            // row and workgroup (per row and wave before: 7 x the atomics, which kept shorter runs from paying)
            constexpr int NW = (G::THRA + 63) / 64;
            float* part = reinterpret_cast<float*>(tile);        // [NW][C1][2]
            static_for<G::C1>([&](auto cc) {
                constexpr int c1 = decltype(cc)::value;
                float sv = asum[c1], mv = amax[c1];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    sv += __shfl_xor(sv, off);
                    mv = fmaxf(mv, __shfl_xor(mv, off));
                }
                if ((tid & 63) == 0) {
                    part[((tid >> 6) * G::C1 + c1) * 2] = sv;
                    part[((tid >> 6) * G::C1 + c1) * 2 + 1] = mv;
                }
                asum[c1] = 0.f;
                amax[c1] = 0.f;
            });
            __syncthreads();
            if (tid < G::C1) {
                float sv = 0.f, mv = 0.f;
                for (int w = 0; w < NW; ++w) {
                    sv += part[(w * G::C1 + tid) * 2];
                    mv = fmaxf(mv, part[(w * G::C1 + tid) * 2 + 1]);
                }
                const size_t row = (MODE == 0) ? (size_t)tid * G::C2 + c2 : (size_t)c2 * G::C1 + tid;
                if (MODE == 0 || (int)row < geo.nrows) {
                    atomicAdd(rowmean + row, (double)sv * inv_ns);      // float64 mean: d4w_internal.h, Mean2
                    atomicMax(rowmaxbits + row, __float_as_uint(mv));      // mv >= 0: bit order = value order
                }
            }
            __syncthreads();
        }
        par ^= 1;
    };
    if (nseq > 0) issue(A, tile_of(0));
    if (nseq > 1) issue(B, tile_of(1));
    bool first = true;
    for (int sq = 0; sq < nseq; sq += 2) {
        body(A, B, sq, first);
        first = false;
        if (sq + 1 < nseq) body(B, A, sq + 1, false);
    }
}

// ---------------------------------------------------------------------------------------------
// pass C: FFT over c2 = C2A x C2B for TC contiguous columns of one c1-position q.
//   forward : S1 item (j < C2B, tt): radix C2A over rows j + a C2B, x W_C2^(j a') -> LDS
//             S2 item (g < C2A, tt): radix C2B over rows g C2B + b               -> global
//   inverse : the same two steps backwards.
// LDS row index c2 + c2 / C2B (one pad row per C2B rows) keeps the S2 reads conflict-free.
// ---------------------------------------------------------------------------------------------
template <class G, bool INV, int MODE = 0>
__global__ __launch_bounds__(G::THRC) void fkf_passC(FkDev P, FkFastDev F, float2* __restrict__ data, int tbase, int ntiles,
                                                     int sw, int sbase, FkGeo geo = FkGeo()) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    float2* twl = tile + G::C2A * (G::C2B + 1) * G::TC;
    // tile u -> (q, n1, column block): u = (q N1 + n1) sw + j covers the column blocks sbase + j, j < sw, of every
    // n1 sub-row; sw = N2 / TC, sbase = 0 is the plain order q (M / TC) + block.  A slab (sbase, sw) holds the
    // same columns as the pass-A tiles with the same (sbase, sw): the two passes can then run slab by slab.
    constexpr int NBS = G::N2 / G::TC;
    constexpr int RA = G::C2A, RB = G::C2B, TC = G::TC;
    const int tid = threadIdx.x;
    const int hi = tid / TC, tt = tid % TC;            // hi = j (radix-RA items) or g (radix-RB items)
    const bool actA = hi < RB;                         // items of the radix-RA step
    const bool actB = hi < RA;                         // items of the radix-RB step
    // table layout [a][j]; row a = 1 is W_C2^j, all the tree variant needs
    for (int i = tid; i < (G::CTREE ? RB : RA * RB); i += G::THRC) twl[i] = F.twC[(G::CTREE ? RB : 0) + i];
    // live-row bits of every (q, g) in LDS: no vector-memory load on the tile loop's critical path
    unsigned* livel = reinterpret_cast<unsigned*>(twl + (G::CTREE ? RB : RA * RB));
    for (int i = tid; i < G::C1 * RA; i += G::THRC) livel[i] = F.live ? F.live[i] : 0xFFFFFFFFu;
    __syncthreads();
    // MODE 1 (slab [nx][geo.ncols]): tile u -> (q = u / geo.nq, column block u % geo.nq), row pitch geo.ncols
    auto PITCH = [&]() -> size_t { return MODE == 0 ? (size_t)G::M : (size_t)geo.ncols; };
    auto tile_qp = [&](int t, int& q, int& p0) {
        if constexpr (MODE == 0) {
            const int hq = t / sw;
            q = hq / G::N1;
            p0 = ((hq - q * G::N1) * NBS + sbase + t % sw) * TC;
        } else {
            q = t / geo.nq;
            p0 = (t - q * geo.nq) * TC;
        }
    };
    constexpr int NPF = INV ? RB : RA;
    // Two register sets, each prefetching TWO tiles ahead (see fkf_passA_fwd): tile t + 2 gstep is loaded into
    // the set of tile t right after S1 has consumed it, before tile t's stores are issued; the loop is unrolled
    // by two so that the sets swap roles without copies.
    float2 pfA[NPF], pfB[NPF];
    auto issue = [&](float2 (&pf)[NPF], int t) {
        int q, p0;
        tile_qp(t, q, p0);
        // row offsets = a per-thread part (hi) + a wave-uniform multiple of the pitch: with a runtime pitch (MODE 1) the
        // uniform part stays in scalar registers instead of one 64-bit vector address per load
        const float2* base = data + ((size_t)q * G::C2) * PITCH() + p0 + tt;
        if constexpr (!INV) {
            const float2* bh = base + (size_t)hi * PITCH();
            static_for<RA>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                pf[a] = fk_ldg<D4W_FK_NT_CFWD>(bh + (size_t)(a * RB) * PITCH());
            });
        } else {
            const unsigned bits = livel[q * RA + hi];       // dead rows are zeros by construction
            const float2* bh = base + (size_t)(hi * RB) * PITCH();
            static_for<RB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                pf[b] = ((bits >> b) & 1u) ? fk_ldg<0>(bh + (size_t)b * PITCH()) : make_float2(0.f, 0.f);
            });
        }
    };
    const bool act_first = INV ? actB : actA;
    const bool act_second = INV ? actA : actB;
    const int gstep = gridDim.x;
    auto body = [&](float2 (&pf)[NPF], int t) {
        int q, p0;
        tile_qp(t, q, p0);
        float2* base = data + ((size_t)q * G::C2) * PITCH() + p0 + tt;
        if (D4W_ABL == 4) {          // timing ablation: stream the tile through registers only
            if (act_first) {
                static_for<NPF>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    if constexpr (!INV) (base + (size_t)hi * PITCH())[(size_t)(a * RB) * PITCH()] = pf[a];
                    else (base + (size_t)(hi * RB) * PITCH())[(size_t)a * PITCH()] = pf[a];
                });
            }
            if (t + 2 * gstep < ntiles && act_first) issue(pf, t + 2 * gstep);
            return;
        }
        if (act_first) {
            if constexpr (!INV) {
                dft<RA>(pf);
                float2 pw[RA];
                if constexpr (G::CTREE) pw_tree<RA>(twl[hi], pw);
                static_for<RA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const float2 w = G::CTREE ? pw[a] : twl[a * RB + hi];
                    const float2 v = (a == 0) ? pf[0] : c_mul(pf[a], w);
                    tile[(hi + a * (RB + 1)) * TC + tt] = v;
                });
            } else {
                idft<RB>(pf);
                static_for<RB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    tile[(hi * (RB + 1) + b) * TC + tt] = pf[b];
                });
            }
        }
        lds_barrier();
        if (t + 2 * gstep < ntiles && act_first) issue(pf, t + 2 * gstep);     // pf is free: S1 consumed it
        constexpr int NV = INV ? RA : RB;
        float2 v[NV];
        if (act_second) {
            if constexpr (!INV) {
                static_for<RB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    v[b] = tile[(hi * (RB + 1) + b) * TC + tt];
                });
            } else {
                static_for<RA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    v[a] = tile[(hi + a * (RB + 1)) * TC + tt];
                });
            }
        }
        lds_barrier();
        if (act_second) {
            if constexpr (!INV) {
                dft<RB>(v);
                const unsigned bits = livel[q * RA + hi];   // dead rows are never read again
                static_for<RB>([&](auto bb) {
                    constexpr int b = decltype(bb)::value;
                    if ((bits >> b) & 1u) fk_stg<D4W_FK_NT_CFWD>(base + (size_t)(hi * RB) * PITCH() + (size_t)b * PITCH(), v[b]);
                });
            } else {
                float2 pw[RA];
                if constexpr (G::CTREE) pw_tree<RA>(twl[hi], pw);
                static_for<RA - 1>([&](auto aa) {
                    constexpr int a = decltype(aa)::value + 1;
                    v[a] = c_mulc(v[a], G::CTREE ? pw[a] : twl[a * RB + hi]);
                });
                idft<RA>(v);
                static_for<RA>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    fk_stg<0>(base + (size_t)hi * PITCH() + (size_t)(a * RB) * PITCH(), v[a]);
                });
            }
        }
    };
    int t = tbase + blockIdx.x;     // tiles [tbase, ntiles): a sub-range when passes are chunked
    if (t < ntiles && act_first) issue(pfA, t);
    if (t + gstep < ntiles && act_first) issue(pfB, t + gstep);
    for (; t < ntiles; t += 2 * gstep) {
        body(pfA, t);
        if (t + gstep < ntiles) body(pfB, t + gstep);
    }
}

// ---------------------------------------------------------------------------------------------
// pass B: contiguous n2 transform (N2 = NA NB NC) of a sub-row and its Hermitian partner,
// real-spectrum pair op with the folded mask, inverse.  See fk_filter.hip (fk_passB) for the
// pair-op algebra; positions here are e = d0 (NB NC) + d1 NC + d2 (digit-reversed, in place).
//   S1  item (row, j < NB NC)       : radix NA on the prefetch registers, x W_N2^(j a')  -> LDS
//   S2  item (row, g < NA, j2 < NC) : radix NB in place, x W_(NB NC)^(j2 b')
//   MID item (G < NA NB)            : radix NC on group G of row A and on its partner group of
//                                     row B, pair op x mask, inverse radix NC, in place
//   S2', S1' : inverse of S2, S1; S1' feeds the global stores.
// LDS position e lives at e + e / NC (one pad per group: the MID reads are conflict-free).
// ---------------------------------------------------------------------------------------------
// MODE 1 (slab): a pair key is r * geo.nq + jq (row position, local sub-row) instead of r * N1 + q1, and the n1 position
// of local sub-row jq is geo.q1_of[jq]; data and mask are both laid out [nx][nq][N2].
// HILB: the pair operation of the Hilbert transform along time instead of the folded mask -- multiplier -i for 0 < f < M,
// +i for the mirrored half, 0 at f = 0 and at the Nyquist frequency (scipy.signal.hilbert: the analytic signal's imaginary
// part); no mask is read.  Used on real rows (each row its own Hermitian partner: d4w_analytic_long_f32).
template <class G, int MODE = 0, bool HILB = false>
__global__ __launch_bounds__(G::THRB, G::WAVES_B) void fkf_passB(FkDev P, FkFastDev F, float2* __restrict__ data, int tbase, int npairs,
                                                                 FkGeo geo = FkGeo()) {
    D4W_DYN_LDS(smem_raw);
    constexpr int N2 = G::N2, NA = G::NA, NB = G::NB, NC = G::NC, M1 = NB * NC, NG = NA * NB, ROWP = G::ROWP;
    constexpr int THR = G::THRB;
    float2* rows = reinterpret_cast<float2*>(smem_raw);
    float2* tw1 = rows + 2 * ROWP;          // [M1]      W_N2^j
    float2* tw2 = tw1 + M1;                 // [NB][NC]  W_M1^(j2 b)
    const int tid = threadIdx.x;
    for (int i = tid; i < M1; i += THR) {
        tw1[i] = F.twB1[i];
        tw2[i] = F.twB2[i];
    }
    __syncthreads();

    // S1 / S1' item
    const int r1 = tid / M1, j1 = tid % M1;
    const int aj1 = j1 + j1 / NC;                                // padded position of j1; of j1 + a M1: aj1 + a (M1 + NB)
    const bool it1 = tid < 2 * M1;
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
    // MID operands of a pair (mask values of the item's two groups, Nyquist-column mask value, row twiddle):
    // loaded ONE PAIR AHEAD and before the current pair's stores -- a wave's memory operations retire in order,
    // so operands requested after the previous pair's stores could not be used before those stores were
    // acknowledged.  The per-thread constants (column twiddles, mirror group) are loaded once.
    const int Gi = tid;
    const bool midrange = Gi < NG;
    int PGz = 0;
    // column twiddle W_ns^(N1 k2) of the group's first position; digit d of the group adds NA NB d to k2, i.e. the literal
    // rotation exp(-2 pi i d / (2 NC)) -- no per-thread table of NC twiddles
    float2 wc0 = make_float2(1.f, 0.f);
    if (midrange) {
        PGz = P.mirror0[Gi * NC] / NC;
        wc0 = P.wcol[Gi * NC];
    }
    struct MidOps {
        float ma[NC], mbr[NC];
        float nyq;
        float2 wr;
    };
    MidOps cur, nxt;
    auto split = [&](int key, int& rpos, int& q1) {      // pair key -> (row position, n1 position)
        if constexpr (MODE == 0) {
            rpos = key / G::N1;
            q1 = key - rpos * G::N1;
        } else {
            rpos = key / geo.nq;
            q1 = geo.q1_of[key - rpos * geo.nq];
        }
    };
    auto issue_mid = [&](MidOps& O, int2 pr) {
        if (!midrange) return;
        int rpos, q1;
        split(pr.x, rpos, q1);
        if constexpr (HILB) {
            O.nyq = 0.f;
            O.wr = P.wrow[q1];
            return;
        }
        const int PG = (q1 == 0) ? PGz : (NG - 1 - Gi);
        const float* mA = P.mask + (size_t)pr.x * N2 + Gi * NC;
        const float* mB = P.mask + (size_t)pr.y * N2 + PG * NC;
        if constexpr (NC % 2 == 0 && N2 % 2 == 0) {              // 8-byte lanes: half the load instructions
            static_for<NC / 2>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                const float2 a2 = reinterpret_cast<const float2*>(mA)[d], b2 = reinterpret_cast<const float2*>(mB)[d];
                O.ma[2 * d] = a2.x; O.ma[2 * d + 1] = a2.y;
                O.mbr[2 * d] = b2.x; O.mbr[2 * d + 1] = b2.y;
            });
        } else {
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                O.ma[d] = mA[d];
                O.mbr[d] = mB[d];
            });
        }
        O.nyq = P.nyq[rpos];
        O.wr = P.wrow[q1];
    };
    // the work list is read two tiles ahead (wave-uniform scalar loads whose latency would
    // otherwise sit in front of every prefetch)
    int t = tbase + blockIdx.x;     // tiles [tbase, ntiles): a sub-range when passes are chunked
    const int gstep = gridDim.x;
    int2 pr_cur = make_int2(0, 0), pr_nxt = make_int2(0, 0);
    if (t < npairs) pr_cur = F.pairs[t];
    if (t + gstep < npairs) pr_nxt = F.pairs[t + gstep];
    if (t < npairs) {
        issue_mid(cur, pr_cur);
        issue(pr_cur);
    }
    for (; t < npairs; t += gstep) {
        const int2 pr = pr_cur;
        int2 pr_nn = pr_cur;
        if (t + 2 * gstep < npairs) pr_nn = F.pairs[t + 2 * gstep];
        const bool same = (pr.x == pr.y);
        const int nrows = same ? 1 : 2;
        int rpos, q1;
        split(pr.x, rpos, q1);
        const bool k1zero = (q1 == 0);
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
        const int PG = k1zero ? PGz : (NG - 1 - Gi);
        const float (&ma)[NC] = cur.ma;
        const float (&mbr)[NC] = cur.mbr;
        const float nyq = cur.nyq;
        const float2 wr = cur.wr;
        if (t + gstep < npairs) issue(pr_nxt);
        // ---------------- S2 (in place)
        for (int it = tid; it < nrows * NA * NC; it += THR) {
            const int r = it / (NA * NC), rem = it - r * (NA * NC);
            const int g = rem / NC, j2 = rem - g * NC;
            const int gj2 = g * (M1 + NB) + j2;                  // padded position of (g, j2, b = 0): e + e / NC, j2 < NC
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
        // ---------------- MID
        if (midrange && (!same || PG >= Gi)) {
            const bool selfg = same && (PG == Gi);
            const bool rev0 = k1zero && (Gi == 0);           // partner digit (NC - d) % NC instead of NC-1-d
            float2* ga = rows + Gi * (NC + 1);
            float2* gb = rows + (same ? 0 : ROWP) + PG * (NC + 1);
            float2 a[NC], b[NC];
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                a[d] = ga[d];
                b[d] = gb[d];
            });
            dft<NC>(a);
            dft<NC>(b);
            const float2 w0 = c_mul(wr, wc0);
            float2 na[NC], nb[NC];                           // new A[d], new B[partner(d)]
            static_for<NC>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                constexpr int pn = NC - 1 - d, pz = (NC - d) % NC;
                const float2 bs = rev0 ? b[pz] : b[pn];
                float mb = rev0 ? mbr[pz] : mbr[pn];
                if (d == 0 && rev0) mb = nyq;
                const float2 Bc = c_conj(bs);
                const float2 w = rot_const<d, 2 * NC>(w0);
                const float2 E = c_scale(c_add(a[d], Bc), 0.5f);
                const float2 O = c_mul_mi(c_scale(c_sub(a[d], Bc), 0.5f));
                const float2 tO = c_mul(w, O);
                float2 Yp, Ym;
                if constexpr (HILB) {
                    const bool dcn = (d == 0 && rev0);          // the (f = 0, Nyquist) pair: no Hilbert transform
                    Yp = dcn ? make_float2(0.f, 0.f) : c_mul_mi(c_add(E, tO));
                    Ym = dcn ? make_float2(0.f, 0.f) : c_mul_pi(c_sub(E, tO));
                    (void)mb;
                } else {
                    Yp = c_scale(c_add(E, tO), ma[d]);
                    Ym = c_scale(c_sub(E, tO), mb);
                }
                const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
                const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
                na[d] = c_add(S, D);
                nb[d] = c_conj(c_sub(S, D));
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
                // group paired with itself: position e takes A's value when it is the first member
                // of its pair, B's value otherwise (self-paired positions: B's, as the generic kernel)
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
        // ---------------- S2'
        for (int it = tid; it < nrows * NA * NC; it += THR) {
            const int r = it / (NA * NC), rem = it - r * (NA * NC);
            const int g = rem / NC, j2 = rem - g * NC;
            const int gj2 = g * (M1 + NB) + j2;                  // padded position of (g, j2, b = 0): e + e / NC, j2 < NC
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
        if (t + gstep < npairs) issue_mid(nxt, pr_nxt);       // next pair's MID operands, ahead of this pair's stores
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
        cur = nxt;
    }
}

}  // namespace d4w
