// f-k filter application on MI355X (gfx950): y = real(ifft2(fft2(x) * M')), M' = unshifted mask.
// Replaces dsp.fk_filter_filt / dsp.fk_filter_sparsefilt (reference dsp.py:725-786).
//
// Formulation (see DESIGN.md "f-k filter"):
//   * the real [nx][ns] block is reinterpreted as a complex [nx][M] block, M = ns/2
//     (z[c][m] = x[c][2m] + i x[c][2m+1]) -- no conversion pass, the bytes are the same;
//   * a 2-D complex FFT of z is computed with both axes split four-step style,
//     nx = C1*C2 (c = c1*C2 + c2) and M = N1*N2 (m = n1*N2 + n2), every 1-D sub-transform running
//     in LDS as an in-place DIF (forward) / DIT (inverse) pair, so spectra stay in digit-reversed
//     order and no transposition is ever written to HBM;
//   * because x is real, the spectrum X(k,f) of x follows from Z(k,f) and conj(Z(-k,M-f)); the
//     "pair op" in the middle pass untangles the two, applies the Hermitian-folded mask to
//     X(k,f) and X(k,f+M), and re-tangles -- so no R2C/C2R pass exists either.
//
// Five HBM passes, each reading and writing the block once, all in place on y:
//   A  fwd : tile (all c1) x (all n1) x TA contiguous n2 : 2-D FFT over (c1,n1), 4-step twiddles
//   C  fwd : tile (all c2) x TC contiguous columns       : FFT over c2
//   B  mid : two contiguous sub-rows of N2 (a row and its Hermitian partner):
//            FFT over n2 -> pair op with mask -> inverse FFT over n2
//   C' inv, A' inv (+ 1/(nx*M) scale).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>

#include "fft_lds.h"
#include "fft_host.h"

#include "fk_entry.h"
#include "design_eval.h"

namespace d4w {

constexpr int kThreads = 256;      // threads of the small helper kernels
#ifndef D4W_FK_THREADS
#define D4W_FK_THREADS 512
#endif
constexpr int kMaxThreads = D4W_FK_THREADS;   // block size of the pass kernels
constexpr int kMaxTile = 8192;     // complex elements per LDS tile (64 KiB)
constexpr int kMaxTileBs = 8192;   // ... of the Bluestein pass C (16384 = 128 KiB, one workgroup per CU, measured slower: 2.36 vs 2.06 ms at 13223 x 12000)
constexpr int kPF = kMaxTile / kMaxThreads;   // prefetch registers (float2) per thread

// Every pass kernel is persistent (grid = 2 workgroups per CU) and software-pipelined:
//     prefetch tile i+1 (global -> registers, no wait) | FFT stages of tile i in LDS | store tile i
// The FFT stages touch only LDS (twiddles included), so the prefetch loads stay in flight across
// the whole compute phase and the stores of tile i drain under the next iteration.

// ---------------------------------------------------------------------------------------------
// pass A : (c1, n1) strided 2-D sub-transform.  tile id = c2 * ntx + bx  (bx fastest: adjacent
// workgroups touch adjacent 64/128-byte segments of the same rows)
// ---------------------------------------------------------------------------------------------
template <bool TAPER, bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fk_passA_fwd(FkDev P, const float2* __restrict__ src,
                                                            float2* __restrict__ dst, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.C1 * d.N1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dN1(d.N1), dntx(ntx);
    const TwLds tw_c1 = tw_stage(P.ax_c1, tile + nelem, tid, nthr);
    const TwLds tw_n1 = tw_stage(P.ax_n1, tile + nelem + tw_lds_elems(P.ax_c1), tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int seg = dTA.div(w), tt = w - seg * TA;
                const int c1 = dN1.div(seg), n1 = seg - c1 * d.N1;
                if (tt < ncol) {
                    const size_t row = (size_t)c1 * d.C2 + c2;
                    const int col = n1 * d.N2 + b0 + tt;
                    v = src[row * d.M + col];
                    if (TAPER) {
                        const float2 wv = P.win[col];
                        v.x *= wv.x;
                        v.y *= wv.y;
                    }
                }
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<false, true, GENERIC>(tile, P.ax_c1, tw_c1, d.N1 * TA, d.N1 * TA, 1, 1, 0, tid, nthr);
        lds_fft<false, true, GENERIC>(tile, P.ax_n1, tw_n1, TA, TA, 1, d.C1, d.N1 * TA, tid, nthr);
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
        for (int w = tid; w < nelem; w += nthr) {
            const int seg = dTA.div(w), tt = w - seg * TA;
            const int q = dN1.div(seg), q1 = seg - q * d.N1;
            if (tt < ncol) {
                const float2 t1 = P.twc[q * d.C2 + c2];
                const float2 t2 = P.twt[q1 * d.N2 + b0 + tt];
                const size_t row = (size_t)q * d.C2 + c2;
                dst[row * d.M + q1 * d.N2 + b0 + tt] = c_mul(tile[w], c_mul(t1, t2));
            }
        }
        lds_barrier();
        t = next;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fk_passA_inv(FkDev P, float2* __restrict__ data, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.C1 * d.N1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dN1(d.N1), dntx(ntx);
    const TwLds tw_c1 = tw_stage(P.ax_c1, tile + nelem, tid, nthr);
    const TwLds tw_n1 = tw_stage(P.ax_n1, tile + nelem + tw_lds_elems(P.ax_c1), tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int seg = dTA.div(w), tt = w - seg * TA;
                const int q = dN1.div(seg), q1 = seg - q * d.N1;
                if (tt < ncol) {
                    const size_t row = (size_t)q * d.C2 + c2;
                    v = data[row * d.M + q1 * d.N2 + b0 + tt];
                }
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) {
                const int seg = dTA.div(w), tt = w - seg * TA;
                const int q = dN1.div(seg), q1 = seg - q * d.N1;
                float2 v = pf[it];
                if (tt < ncol) v = c_mulc(v, c_mul(P.twc[q * d.C2 + c2], P.twt[q1 * d.N2 + b0 + tt]));
                tile[w] = v;
            }
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<true, true, GENERIC>(tile, P.ax_n1, tw_n1, TA, TA, 1, d.C1, d.N1 * TA, tid, nthr);
        lds_fft<true, true, GENERIC>(tile, P.ax_c1, tw_c1, d.N1 * TA, d.N1 * TA, 1, 1, 0, tid, nthr);
        for (int w = tid; w < nelem; w += nthr) {
            const int seg = dTA.div(w), tt = w - seg * TA;
            const int c1 = dN1.div(seg), n1 = seg - c1 * d.N1;
            if (tt < ncol) {
                const size_t row = (size_t)c1 * d.C2 + c2;
                data[row * d.M + n1 * d.N2 + b0 + tt] = c_scale(tile[w], P.scale);
            }
        }
        lds_barrier();
        t = next;
    }
}

// ---------------------------------------------------------------------------------------------
// pass C : c2 sub-transform over C2 consecutive rows, TC contiguous columns; tile id = q*ntx + bx
// ---------------------------------------------------------------------------------------------
template <bool INV, bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fk_passC(FkDev P, float2* __restrict__ data, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TC = d.TC;
    const int nelem = d.C2 * TC;
    const int ntx = (d.M + TC - 1) / TC;
    const FDiv dTC(TC), dntx(ntx);
    const TwLds tw = tw_stage(P.ax_c2, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int q = dntx.div(t), p0 = (t - q * ntx) * TC;
        const int ncol = min(TC, d.M - p0);
        const float2* base = data + ((size_t)q * d.C2) * d.M + p0;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int c2 = dTC.div(w), tt = w - c2 * TC;
                if (tt < ncol) v = base[(size_t)c2 * d.M + tt];
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<INV, true, GENERIC>(tile, P.ax_c2, tw, TC, TC, 1, 1, 0, tid, nthr);
        const int q = dntx.div(t), p0 = (t - q * ntx) * TC;
        const int ncol = min(TC, d.M - p0);
        float2* base = data + ((size_t)q * d.C2) * d.M + p0;
        for (int w = tid; w < nelem; w += nthr) {
            const int c2 = dTC.div(w), tt = w - c2 * TC;
            if (tt < ncol) base[(size_t)c2 * d.M + tt] = tile[w];
        }
        lds_barrier();
        t = next;
    }
}

// pass C when C2 has a prime factor > 31 (Bluestein): the tile holds bs_L >= 2 C2 - 1 rows per column,
//   a[n] = x[n] w[n] (n < C2, zero above), A = FFT_L(a), A *= FFT_L(conj w wrapped) / L, c = IFFT_L(A),
//   X[k] = w[k] c[k] (k < C2), w[n] = exp(-i pi n^2 / C2); the inverse transform conjugates w and the filter.
// Output in NATURAL order (the plan's c2 position -> wavenumber map is the identity).  Exact in the same
// sense as the other passes (float32 rounding), slower: twice the LDS tile and three transforms per column.
template <bool INV>
__global__ __launch_bounds__(kMaxThreads) void fk_passC_bluestein(FkDev P, float2* __restrict__ data, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TC = d.TC, L = P.bs_L;
    const int nelem = d.C2 * TC, ntile = L * TC;
    const int ntx = (d.M + TC - 1) / TC;
    const TwLds tw = tw_stage(P.ax_bs, tile + ntile, tid, nthr);
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (8 of them, each with its own L2), and a
    // tile reads only TC * 8 bytes of every row -- a quarter of a 64-byte sector at TC = 2.  XCD x therefore walks its own
    // contiguous eighth of the tiles, so that the neighbours sharing a tile's sectors run on the same L2.
    const int nxcd = (gridDim.x % 8 == 0 && ntiles >= 64) ? 8 : 1;
    const int xcd = blockIdx.x % nxcd, jx = blockIdx.x / nxcd, gx = gridDim.x / nxcd;
    const int t8 = (ntiles + nxcd - 1) / nxcd, tend = min(ntiles, (xcd + 1) * t8);
    for (int t = xcd * t8 + jx; t < tend; t += gx) {
        const int q = t / ntx, p0 = (t - q * ntx) * TC;
        const int ncol = min(TC, d.M - p0);
        float2* base = data + ((size_t)q * d.C2) * d.M + p0;
        for (int w = tid; w < ntile; w += nthr) {
            const int c2 = w / TC, tt = w - c2 * TC;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem && tt < ncol) {
                const float2 ch = P.bs_chirp[c2];
                v = INV ? c_mulc(base[(size_t)c2 * d.M + tt], ch) : c_mul(base[(size_t)c2 * d.M + tt], ch);
            }
            tile[w] = v;
        }
        lds_barrier();
        lds_fft<false, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        for (int w = tid; w < ntile; w += nthr) {
            const float2 f = P.bs_filt[w / TC];
            tile[w] = INV ? c_mulc(tile[w], f) : c_mul(tile[w], f);
        }
        lds_barrier();
        lds_fft<true, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        for (int w = tid; w < nelem; w += nthr) {
            const int c2 = w / TC, tt = w - c2 * TC;
            if (tt < ncol) {
                const float2 ch = P.bs_chirp[c2];
                base[(size_t)c2 * d.M + tt] = INV ? c_mulc(tile[w], ch) : c_mul(tile[w], ch);
            }
        }
        lds_barrier();
    }
}

// Time-first order (fk_tf.h) when the c2 axis is a Bluestein convolution: passes C and C' fused on the BAND strips of the
// compact spectrum W -- chirp, FFT_L, x filter, IFFT_L, chirp (= the c2 transform, natural order), x band mask, and the same
// backwards with conjugated tables -- one read and one write of the strip for four length-L transforms.  Tile order as
// fkf_passCm: per c1 position q the band strips of the N1 sub-row blocks, then the Nyquist strip.
__global__ __launch_bounds__(kMaxThreads) void fk_passCm_bluestein(FkDev P, FkTfDev T, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TC = d.TC, L = P.bs_L;
    const int nelem = d.C2 * TC, ntile = L * TC;
    const TwLds tw = tw_stage(P.ax_bs, tile + ntile, tid, nthr);
    const int nb1 = T.bw / TC, tq = d.N1 * nb1 + (T.col_nyq >= 0 ? 1 : 0);
    const size_t LC = (size_t)T.Lc;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int q = t / tq, u = t - q * tq;
        int p0;
        if (u < d.N1 * nb1) {
            const int q1 = u / nb1;
            p0 = q1 * T.RW + (u - q1 * nb1) * TC;
        } else
            p0 = T.col_nyq;
        float2* base = T.W + ((size_t)q * d.C2) * LC + p0;
        const float* mbase = T.cmask + ((size_t)q * d.C2) * LC + p0;
        for (int w = tid; w < ntile; w += nthr) {
            const int c2 = w / TC, tt = w - c2 * TC;
            tile[w] = (w < nelem) ? c_mul(base[(size_t)c2 * LC + tt], P.bs_chirp[c2]) : make_float2(0.f, 0.f);
        }
        lds_barrier();
        lds_fft<false, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        for (int w = tid; w < ntile; w += nthr) tile[w] = c_mul(tile[w], P.bs_filt[w / TC]);
        lds_barrier();
        lds_fft<true, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        // spectrum at wavenumber position c2 (natural order): x chirp x mask, then straight into the inverse convolution
        // (x conj chirp); everything beyond the C2 rows is the zero padding again
        for (int w = tid; w < ntile; w += nthr) {
            const int c2 = w / TC, tt = w - c2 * TC;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const float2 ch = P.bs_chirp[c2];
                v = c_mulc(c_scale(c_mul(tile[w], ch), mbase[(size_t)c2 * LC + tt]), ch);
            }
            tile[w] = v;
        }
        lds_barrier();
        lds_fft<false, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        for (int w = tid; w < ntile; w += nthr) tile[w] = c_mulc(tile[w], P.bs_filt[w / TC]);
        lds_barrier();
        lds_fft<true, true, false>(tile, P.ax_bs, tw, TC, TC, 1, 1, 0, tid, nthr);
        for (int w = tid; w < nelem; w += nthr) {
            const int c2 = w / TC, tt = w - c2 * TC;
            base[(size_t)c2 * LC + tt] = c_mulc(tile[w], P.bs_chirp[c2]);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// pass B : contiguous n2 transform + real-spectrum pair op + mask + inverse
//
// For packed element a = (k, f), f = k1 + N1*k2 in [0, M), with partner b = (-k, (M - f) mod M):
//   A = Z[a], B = conj(Z[b]);  E = (A+B)/2 (even-sample spectrum), O = -i (A-B)/2 (odd-sample)
//   X(k, f)   = E + W O,  X(k, f+M) = E - W O,  W = exp(-2 pi i f / ns)
//   Y+ = M_h(k, f) X(k, f),  Y- = M_h(k, f+M) X(k, f+M),  M_h(k, f+M) = M_h(-k, M-f) (Nyquist
//   column M_h(k, M) when f = 0)
//   Zy[a] = S + D,  Zy[b] = conj(S - D),  S = (Y+ + Y-)/2,  D = i conj(W) (Y+ - Y-)/2.
// Work list: P.pairs[t] = (keyA, keyB), key = row position * N1 + q1, keyA <= keyB, each
// Hermitian pair of sub-rows exactly once (keyA == keyB for the self-paired ones).
// ---------------------------------------------------------------------------------------------
template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fk_passB(FkDev P, float2* __restrict__ data, int npairs) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N2 = d.N2;
    const TwLds tw = tw_stage(P.ax_n2, tile + 2 * N2, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int2 pr = P.pairs[t];
        const float2* rowA = data + (size_t)pr.x * N2;
        const float2* rowB = data + (size_t)pr.y * N2;
        const int nload = (pr.x == pr.y) ? N2 : 2 * N2;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nload) v = (w < N2) ? rowA[w] : rowB[w - N2];
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < npairs) issue(t);
    while (t < npairs) {
        const int2 pr = P.pairs[t];
        const bool same = (pr.x == pr.y);
        const int nload = same ? N2 : 2 * N2;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nload) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < npairs) issue(next);
        float2* A = tile;
        float2* B = same ? tile : tile + N2;
        const int nrows = same ? 1 : 2;
        lds_fft<false, false, GENERIC>(tile, P.ax_n2, tw, 1, nrows, N2, 1, 0, tid, nthr);

        const int r = pr.x / d.N1, q1 = pr.x - r * d.N1;
        const bool k1zero = (q1 == 0);
        const float* mA = P.mask + (size_t)pr.x * N2;
        const float* mB = P.mask + (size_t)pr.y * N2;
        const float2 wr = P.wrow[q1];
        const float nyq = P.nyq[r];
        for (int i = tid; i < N2; i += nthr) {
            const int j = k1zero ? P.mirror0[i] : (N2 - 1 - i);
            if (same && j < i) continue;
            const float2 a = A[i];
            const float2 Bc = c_conj(B[j]);
            const float ma = mA[i];
            const float mb = (k1zero && i == 0) ? nyq : mB[j];
            const float2 w = c_mul(wr, P.wcol[i]);
            const float2 E = c_scale(c_add(a, Bc), 0.5f);
            const float2 O = c_mul_mi(c_scale(c_sub(a, Bc), 0.5f));
            const float2 tO = c_mul(w, O);
            const float2 Yp = c_scale(c_add(E, tO), ma);
            const float2 Ym = c_scale(c_sub(E, tO), mb);
            const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
            const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
            A[i] = c_add(S, D);
            B[j] = c_conj(c_sub(S, D));
        }
        lds_barrier();
        lds_fft<true, false, GENERIC>(tile, P.ax_n2, tw, 1, nrows, N2, 1, 0, tid, nthr);
        float2* rowA = data + (size_t)pr.x * N2;
        float2* rowB = data + (size_t)pr.y * N2;
        for (int w = tid; w < nload; w += nthr) {
            if (w < N2) rowA[w] = tile[w];
            else rowB[w - N2] = tile[w];
        }
        lds_barrier();
        t = next;
    }
}

// pass B when N2 has a prime factor > 31: the n2 sub-transforms of the row pair run as Bluestein convolutions of
// length bn_L inside the tile (see fk_passC_bluestein), spectra in NATURAL order (the plan's position -> frequency
// tables of the n2 axis are the identity, so the pair op below is fk_passB's, table for table).
__global__ __launch_bounds__(kMaxThreads) void fk_passB_bluestein(FkDev P, float2* __restrict__ data, int npairs) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N2 = d.N2, L = P.bn_L;
    const TwLds tw = tw_stage(P.ax_bn, tile + 2 * L, tid, nthr);
    for (int t = blockIdx.x; t < npairs; t += gridDim.x) {
        const int2 pr = P.pairs[t];
        const bool same = (pr.x == pr.y);
        const int nrows = same ? 1 : 2;
        float2* rowA = data + (size_t)pr.x * N2;
        float2* rowB = data + (size_t)pr.y * N2;
        for (int w = tid; w < nrows * L; w += nthr) {
            const int r = w / L, n = w - r * L;
            tile[w] = (n < N2) ? c_mul((r ? rowB : rowA)[n], P.bn_chirp[n]) : make_float2(0.f, 0.f);
        }
        lds_barrier();
        lds_fft<false, false, false>(tile, P.ax_bn, tw, 1, nrows, L, 1, 0, tid, nthr);
        for (int w = tid; w < nrows * L; w += nthr) tile[w] = c_mul(tile[w], P.bn_filt[w % L]);
        lds_barrier();
        lds_fft<true, false, false>(tile, P.ax_bn, tw, 1, nrows, L, 1, 0, tid, nthr);
        for (int w = tid; w < nrows * N2; w += nthr) {
            const int r = w / N2, k = w - r * N2;
            tile[r * L + k] = c_mul(tile[r * L + k], P.bn_chirp[k]);
        }
        lds_barrier();
        // ---- pair op x mask (fk_passB)
        float2* A = tile;
        float2* B = same ? tile : tile + L;
        const int r = pr.x / d.N1, q1 = pr.x - r * d.N1;
        const bool k1zero = (q1 == 0);
        const float* mA = P.mask + (size_t)pr.x * N2;
        const float* mB = P.mask + (size_t)pr.y * N2;
        const float2 wr = P.wrow[q1];
        const float nyq = P.nyq[r];
        for (int i = tid; i < N2; i += nthr) {
            const int j = k1zero ? P.mirror0[i] : (N2 - 1 - i);
            if (same && j < i) continue;
            const float2 a = A[i];
            const float2 Bc = c_conj(B[j]);
            const float ma = mA[i];
            const float mb = (k1zero && i == 0) ? nyq : mB[j];
            const float2 w = c_mul(wr, P.wcol[i]);
            const float2 E = c_scale(c_add(a, Bc), 0.5f);
            const float2 O = c_mul_mi(c_scale(c_sub(a, Bc), 0.5f));
            const float2 tO = c_mul(w, O);
            const float2 Yp = c_scale(c_add(E, tO), ma);
            const float2 Ym = c_scale(c_sub(E, tO), mb);
            const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
            const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
            A[i] = c_add(S, D);
            B[j] = c_conj(c_sub(S, D));
        }
        lds_barrier();
        // ---- inverse n2 transforms (conjugate chirp and filter), zero padding restored first
        for (int w = tid; w < nrows * L; w += nthr) {
            const int rr = w / L, n = w - rr * L;
            tile[w] = (n < N2) ? c_mulc(tile[w], P.bn_chirp[n]) : make_float2(0.f, 0.f);
        }
        lds_barrier();
        lds_fft<false, false, false>(tile, P.ax_bn, tw, 1, nrows, L, 1, 0, tid, nthr);
        for (int w = tid; w < nrows * L; w += nthr) tile[w] = c_mulc(tile[w], P.bn_filt[w % L]);
        lds_barrier();
        lds_fft<true, false, false>(tile, P.ax_bn, tw, 1, nrows, L, 1, 0, tid, nthr);
        for (int w = tid; w < nrows * N2; w += nthr) {
            const int rr = w / N2, k = w - rr * N2;
            (rr ? rowB : rowA)[k] = c_mulc(tile[rr * L + k], P.bn_chirp[k]);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// mask fold + permutation into pass-B order (one-off per mask)
// ---------------------------------------------------------------------------------------------
// mask values as they are read by the fold: m, or m * a + b (dsp.fk_filt's min-max normalisation, dsp.py:945, applied to
// every value exactly as the separate pass did -- one fewer read and write of the dense mask)
struct FkAffine {
    float a, b;
    int on;
    __device__ __forceinline__ float operator()(float m) const { return on ? fmaf(m, a, b) : m; }
};

__global__ __launch_bounds__(kThreads) void fk_fold_mask(FkDims d, const float* __restrict__ ms,
                                                          const int* __restrict__ rowk,
                                                          const int* __restrict__ k1_of_q1,
                                                          const int* __restrict__ k2_of_i,
                                                          float* __restrict__ mask,
                                                          float* __restrict__ nyq, unsigned* __restrict__ rowmaxbits, FkAffine aff) {
    const int r = blockIdx.y;
    const int k = rowk[r];
    const int km = (d.nx - k) % d.nx;
    const int sx = d.nx / 2, st = d.ns / 2;
    const size_t rowp = (size_t)((k + sx) % d.nx) * d.ns;    // shifted-grid row of +k
    const size_t rowm = (size_t)((km + sx) % d.nx) * d.ns;   // shifted-grid row of -k
    unsigned vbits = 0u;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.M; p += gridDim.x * blockDim.x) {
        const int q1 = p / d.N2, i = p - q1 * d.N2;
        const int f = k1_of_q1[q1] + d.N1 * k2_of_i[i];
        const int fm = (d.ns - f) % d.ns;
        const float v = 0.5f * (aff(ms[rowp + (f + st) % d.ns]) + aff(ms[rowm + (fm + st) % d.ns]));
        mask[(size_t)r * d.M + p] = v;
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int f = d.M, fm = d.ns - d.M;
        const float v = 0.5f * (aff(ms[rowp + (f + st) % d.ns]) + aff(ms[rowm + (fm + st) % d.ns]));
        nyq[r] = v;
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
    }
    if (vbits) atomicMax(rowmaxbits + r, vbits);
}

// The same fold with the mask evaluated in closed form (design_eval.h) instead of read: a reference design goes
// straight into pass-B order and no dense [nx][ns] mask exists (9.6 GB at 20 000 x 120 000).  Each of the two
// values is rounded to float32 before the average exactly as the dense design kernel stores it, so the folded
// mask is bit-identical to design + fold.
__global__ __launch_bounds__(kThreads) void fk_fold_design(FkDims d, DesignArgs A, int mode,
                                                            const int* __restrict__ rowk,
                                                            const int* __restrict__ k1_of_q1,
                                                            const int* __restrict__ k2_of_i,
                                                            float* __restrict__ mask,
                                                            float* __restrict__ nyq, unsigned* __restrict__ rowmaxbits) {
    const int r = blockIdx.y;
    const int k = rowk[r];
    const int km = (d.nx - k) % d.nx;
    const int sx = d.nx / 2, st = d.ns / 2;
    const int ip = (k + sx) % d.nx, im = (km + sx) % d.nx;
    unsigned vbits = 0u;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.M; p += gridDim.x * blockDim.x) {
        const int q1 = p / d.N2, i = p - q1 * d.N2;
        const int f = k1_of_q1[q1] + d.N1 * k2_of_i[i];
        const int fm = (d.ns - f) % d.ns;
        const float v = design_folded(A, mode, ip, (f + st) % d.ns, im, (fm + st) % d.ns);
        mask[(size_t)r * d.M + p] = v;
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int f = d.M, fm = d.ns - d.M;
        const float v = design_folded(A, mode, ip, (f + st) % d.ns, im, (fm + st) % d.ns);
        nyq[r] = v;
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
    }
    if (vbits) atomicMax(rowmaxbits + r, vbits);
}

constexpr int kFoldThreads = 1024;   // 16 waves per workgroup: the tile's loads are latency-bound with fewer in flight
// The same fold with both sides of the permutation coalesced (the gather above reads 4 useful bytes per 64-byte
// sector: 178 GB of traffic and 23 ms at 20000 x 120000).  The first radix R0 of the n2 transform carries the LOWEST
// digit d0 of k2 and the HIGHEST of the position: k2 = d0 + R0 k2'(i'), i = d0 (N2 / R0) + i'.  So for one i' the
// frequencies f = k1 + N1 k2 over all (k1, d0) form ONE contiguous run of RL = N1 R0 samples of the mask row
// (and a reversed run of the partner row), and for one (k1, d0) the positions of IB consecutive i' are contiguous.
// Tile = (row r, all k1, all d0, IB consecutive i'): runs in, LDS transpose [RL][IB + 1], runs out.
// Also leaves max |M_h| of every row (Nyquist column included) in rowmaxbits -- liveness and the opt-in tail pruning.
__global__ __launch_bounds__(kFoldThreads) void fk_fold_mask_tiled(FkDims d, int R0, int IB, const float* __restrict__ ms,
                                                                const int* __restrict__ rowk,
                                                                const int* __restrict__ q1_of_k1,
                                                                const int* __restrict__ k2_of_i,
                                                                float* __restrict__ mask, float* __restrict__ nyq,
                                                                unsigned* __restrict__ rowmaxbits, FkAffine aff) {
    D4W_DYN_LDS(smem_raw);
    float* buf = reinterpret_cast<float*>(smem_raw);
    const int r = blockIdx.y;
    const int k = rowk[r];
    const int km = (d.nx - k) % d.nx;
    const int sx = d.nx / 2, st = d.ns / 2;
    const float* rowp = ms + (size_t)((k + sx) % d.nx) * d.ns + st;    // + f : M(+k, f),  0 <= f < M
    const float* rowm = ms + (size_t)((km + sx) % d.nx) * d.ns + st;   // - f : M(-k, -f)
    const int RL = d.N1 * R0, N2r = d.N2 / R0, pitch = IB + 1;
    const int ib0 = blockIdx.x * IB, nb = min(IB, N2r - ib0);
    // max |M_h| as the bit pattern of |v|: unsigned order = value order, and a NaN gain (above +inf) keeps its
    // row alive like NumPy would
    unsigned vbits = 0u;
    const FDiv dRL(RL);
#pragma unroll 4
    for (int w = threadIdx.x; w < RL * nb; w += blockDim.x) {
        const int j = dRL.div(w), u = w - j * RL;
        const int f = u + RL * (k2_of_i[ib0 + j] / R0);
        const float v = 0.5f * (aff(rowp[f]) + aff(rowm[-f]));
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
        buf[u * pitch + j] = v;
    }
    __syncthreads();
    const FDiv dnb(nb), dN1(d.N1);
#pragma unroll 4
    for (int w = threadIdx.x; w < RL * nb; w += blockDim.x) {
        const int u = dnb.div(w), j = w - u * nb;
        const int d0 = dN1.div(u), k1 = u - d0 * d.N1;
        mask[(size_t)r * d.M + (size_t)q1_of_k1[k1] * d.N2 + d0 * N2r + ib0 + j] = buf[u * pitch + j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float v = 0.5f * (aff(rowp[d.M - d.ns]) + aff(rowm[d.M - d.ns]));   // f = M (Nyquist): shifted column 0 of both rows
        nyq[r] = v;
        vbits = max(vbits, __float_as_uint(v) & 0x7fffffffu);
    }
    for (int off = 32; off >= 1; off >>= 1) vbits = max(vbits, __shfl_xor(vbits, off));
    if ((threadIdx.x & 63) == 0 && vbits) atomicMax(rowmaxbits + r, vbits);
}

// Column statistics of the folded mask (time-first order, fk_tf.h): min and max over all wavenumber rows of every
// half-spectrum column, as order-preserving unsigned keys (a NaN gain lands above +inf in the max key).  One thread per
// column position (coalesced along the row), row chunks combined by atomics.
__device__ __forceinline__ unsigned fk_ord_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(kThreads) void fk_col_minmax(FkDims d, const float* __restrict__ mask, int rows_per_block,
                                                           unsigned* __restrict__ kmin, unsigned* __restrict__ kmax) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.M) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(d.nx, r0 + rows_per_block);
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int r = r0; r < r1; ++r) {
        const unsigned k = fk_ord_key(mask[(size_t)r * d.M + p]);
        lo = min(lo, k);
        hi = max(hi, k);
    }
    if (r1 > r0) {
        atomicMin(kmin + p, lo);
        atomicMax(kmax + p, hi);
    }
}

// Band columns of the folded mask in the compact column order of the time-first passes: cmask[r][col] = mask[r][src[col]]
// (src = -2: the Nyquist column nyq[r]; -1: padding).
__global__ __launch_bounds__(kThreads) void fk_gather_cmask(FkDims d, const float* __restrict__ mask, const float* __restrict__ nyq,
                                                             const int* __restrict__ src, int Lband, float* __restrict__ cmask) {
    const int r = blockIdx.y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < Lband; c += gridDim.x * blockDim.x) {
        const int sp = src[c];
        cmask[(size_t)r * Lband + c] = (sp >= 0) ? mask[(size_t)r * d.M + sp] : (sp == -2 ? nyq[r] : 0.f);
    }
}

// Liveness words of pass Cm (fk_tf.h): for tile (q, strip p0 .. p0 + TC) and radix-RB item g, bit b = the band mask of row
// q C2 + g RB + b has a non-zero (or NaN) value in the strip.  One thread per (tile, g).
__global__ __launch_bounds__(kThreads) void fk_cm_livebits(const float* __restrict__ cmask, int Lc, int C2, int RA, int RB, int TC, int N1,
                                                            int nb1, int RW, int col_nyq, int ntiles, unsigned* __restrict__ live) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntiles * RA) return;
    const int t = i / RA, g = i - t * RA;
    const int tq = N1 * nb1 + (col_nyq >= 0 ? 1 : 0);
    const int q = t / tq, u = t - q * tq;
    int p0;
    if (u < N1 * nb1) {
        const int q1 = u / nb1;
        p0 = q1 * RW + (u - q1 * nb1) * TC;
    } else
        p0 = col_nyq;
    unsigned bits = 0u;
    for (int b = 0; b < RB; ++b) {
        const float* mp = cmask + ((size_t)q * C2 + (size_t)g * RB + b) * Lc + p0;
        bool any = false;
        for (int c = 0; c < TC; ++c) any |= !(mp[c] == 0.f);
        if (any) bits |= 1u << b;
    }
    live[i] = bits;
}

// x[c][n] *= tukey(ns, 0.03)[n] in place (dsp.taper_data, dsp.py:705-722).  The window is exactly 1 between its two cosine
// ramps -- 1.5 % of the row at each end -- and x * 1.0f is x: only the ramps are read and written (W columns at each end,
// ramp[0 .. 2 W) = the window's first and last W values).  Rounds 1-5 multiplied the whole block: 4.5 ms per 20 000 x 120 000
// block for 0.15 ms of work.
__global__ __launch_bounds__(kThreads) void taper_ramps(float* __restrict__ x, const float* __restrict__ ramp, int nx, int ns, int W) {
    const size_t per_row = (size_t)2 * W, total = (size_t)nx * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i / per_row;
        const int j = (int)(i - c * per_row);
        const int n = (j < W) ? j : ns - 2 * W + j;
        x[c * (size_t)ns + n] *= ramp[j];
    }
}

// ---------------------------------------------------------------------------------------------
// host side: planner
// ---------------------------------------------------------------------------------------------
// scipy.signal.windows.tukey(n, alpha), symmetric (used by dsp.taper_data, dsp.py:721)
static std::vector<float> tukey_window(int n, double alpha) {
    std::vector<float> w(n, 1.0f);
    if (n <= 1 || alpha <= 0) return w;
    if (alpha >= 1.0) {
        for (int i = 0; i < n; ++i) w[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / (n - 1)));
        return w;
    }
    const int width = (int)floor(alpha * (n - 1) / 2.0);
    for (int i = 0; i <= width; ++i)
        w[i] = (float)(0.5 * (1.0 + cos(M_PI * (-1.0 + 2.0 * i / alpha / (n - 1)))));
    for (int i = n - width - 1; i < n; ++i)
        w[i] = (float)(0.5 * (1.0 + cos(M_PI * (-2.0 / alpha + 1.0 + 2.0 * i / alpha / (n - 1)))));
    return w;
}


// shape-specialised kernels: fk_entry.h (FkFastEntry, fast_entry<G>) -- one table entry per instantiated shape
//                 C1  C2A C2B  N1  NA  NB  NC  TA  TC  thrA thrC thrB
// 20000 x 120000: 128-byte column strips in pass C (TC = 16) are worth 3.8 vs 6.9 ms per pass over 64-byte ones
using FkShapeBench = FkFastCfg<25, 25, 32, 25, 20, 12, 10, 16, 16, 448, 512, 256>;
// BASELINE configs[0..1] geometry and the real OOI RAPID channel count (11020 = 20 x 19 x 29: the two
// primes are register butterflies of the c2 axis)
using FkShapeC1 = FkFastCfg<20, 10, 20, 5, 10, 12, 10, 16, 16, 320, 320, 256>;      // 4000 x 12000
using FkShapeOOI = FkFastCfg<20, 19, 29, 5, 10, 12, 10, 16, 16, 320, 464, 256>;     // 11020 x 12000
// the reference notebook's channel selection on the OOI RAPID cable ([20000, 65000, 10] m -> 5510 channels;
// scripts/main_mfdetect.py's [12000, 66000, 5] m gives 13223 = 7 x 1889 channels: Bluestein pass C)
using FkShapeNB = FkFastCfg<10, 19, 29, 5, 10, 12, 10, 16, 16, 192, 464, 256>;      // 5510 x 12000
using FkShapeT1 = FkFastCfg<3, 2, 3, 2, 2, 3, 2, 2, 2, 64, 64, 64>;                 // 18 x 48    (tests)
using FkShapeT2 = FkFastCfg<2, 2, 2, 2, 8, 3, 5, 2, 2, 64, 64, 64>;                 // 8 x 480    (tests)
using FkShapeT3 = FkFastCfg<5, 4, 5, 5, 4, 3, 5, 4, 4, 64, 64, 64>;                 // 100 x 600  (tests)
using FkShapeT4 = FkFastCfg<2, 7, 11, 2, 2, 3, 2, 2, 2, 64, 64, 64>;                // 154 x 48   (tests: prime butterflies 7, 11)
using FkShapeT5 = FkFastCfg<2, 19, 29, 2, 2, 3, 2, 2, 2, 64, 64, 64>;               // 1102 x 48  (tests: the OOI shapes' radices 19, 29)

// shape configurations registered at run time (compiled on demand, das4whales_amd/fkjit.py); entries are never removed
static std::mutex g_dyn_mu;
static std::vector<FkFastEntry*>& dyn_shapes() {
    static std::vector<FkFastEntry*> v;
    return v;
}

static const std::vector<FkFastEntry>& fast_shapes() {
    static const std::vector<FkFastEntry> v = {
        fast_entry<FkShapeBench>(1, 1, 2),
        fast_entry<FkShapeC1>(2, 2, 2),
        fast_entry<FkShapeOOI>(2, 2, 2),
        fast_entry<FkShapeNB>(2, 2, 2),
        fast_entry<FkShapeT1>(2, 2, 2),
        fast_entry<FkShapeT2>(2, 2, 2),
        fast_entry<FkShapeT3>(2, 2, 2),
        fast_entry<FkShapeT4>(2, 2, 2),
        fast_entry<FkShapeT5>(2, 2, 2),
    };
    return v;
}

}  // namespace d4w

using namespace d4w;

struct d4w_fkd_plan;
// the distributed plan's generic form at world 1 carries the channel counts whose part with prime factors > 31 does not
// fit the LDS Bluestein tile (its channel phase is a Bluestein convolution in global memory, fkd_bz_*)
extern "C" {
static int fkd_plan_build(int nx, int ns, int world, int rank, bool want_mask, d4w_fkd_plan** out);
}
static int fkd_set_mask_dense_affine(d4w_fkd_plan* pl, const float* mask_shifted, float a, float b, int on, void* stream);
static void fkd_set_prune(d4w_fkd_plan* pl, double prune_eps);      // the gain level the next mask's dead columns are decided by

struct d4w_fk_plan {
    FkDev dev;
    d4w_fkd_plan* big = nullptr;           // set: every call goes through this world-1 distributed plan
    std::vector<void*> allocs;
    int* d_rowk = nullptr;
    int* d_k1 = nullptr;
    int* d_k2 = nullptr;
    float* d_mask = nullptr;
    float* d_nyq = nullptr;
    float* d_win_flat = nullptr;
    bool has_mask = false;
    bool genericA = false, genericB = false, genericC = false;
    size_t ldsA = 0, ldsB = 0, ldsC = 0;
    int threads = kMaxThreads;
    int npairs = 0;
    int num_cu = 256;
    int wg_per_cu = 2;
    const FkFastEntry* fast = nullptr;     // shape-specialised kernels, or nullptr = generic passes
    FkFastDev fdev;
    // dead-row pruning of the specialised path (set per mask)
    std::vector<int2> h_pairs;             // full pass-B work list (host copy)
    unsigned* d_rowmax = nullptr;          // [nx] bit pattern of max |M_h| per row (fk_fold_mask_tiled)
    int* d_q1_of_k1 = nullptr;             // [N1] position of time-axis digit k1
    int fold_R0 = 0, fold_IB = 0;          // tiling of fk_fold_mask_tiled (0: the gather kernel)
    double prune_eps = 0.0;                // tail pruning of the current mask (0 = exact)
    unsigned* d_livebits = nullptr;        // [C1][C2A]
    int2* d_pairs_live = nullptr;          // [npairs] compacted list
    int npairs_run = 0;                    // pairs the specialised pass B runs
    int live_rows = 0;
    int wgA = 1, wgC = 1, wgB = 1, wgBi = 1, wgBf = 1;
    int slab_sw = 0;                       // > 0: passes A/C and C'/A' run slab by slab, sw column blocks per slab
    // time-first order (fk_tf.h): chosen per mask by fk_mask_finish when it moves fewer bytes than dead-row skipping
    bool tf = false;
    FkTfDev tfdev;
    int npairsT = 0;
    unsigned* d_colkeys = nullptr;         // [2][M] min / max keys of the mask columns
    float* d_tgain = nullptr;              // [M]
    int2* d_ctab = nullptr;                // [NC][NB]
    int* d_colsrc = nullptr;               // [Lc] gather table: mask position of every column of W (-1 none, -2 Nyquist)
    float* d_cmask = nullptr;              // [nx][Lc] folded mask at the band columns
    float2* d_W = nullptr;                 // [nx][Lc] compact half spectrum (capacity cap_W elements, shared by the three)
    unsigned* d_cmlive = nullptr;          // [Cm tiles][C2A] liveness words of the band mask
    size_t cap_W = 0;
    int tf_band_cols = 0, tf_tail_cols = 0;   // per row: band (incl. Nyquist) and tail columns kept
    double bytes_cf = 42.0, bytes_tf = 42.0;  // modelled bytes per channel-sample of the two orders for the current mask
    // One plan, several host threads / streams (dsp.get_fk_plan hands the same plan to every thread filtering a shape): the
    // folded mask and its tables are rewritten by every set_mask, W of the time-first order and the exchange buffers of
    // `big` by every apply.  Every operation on the plan (set_mask*, apply*) takes the host mutex for its launch sequence,
    // makes its stream wait for the event the previous operation recorded if that ran on ANOTHER stream, and records the
    // event again at its end: a total order on the device, free when everything runs on one stream.
    std::mutex apply_mu;
    hipEvent_t apply_done = nullptr;
    hipStream_t apply_stream = nullptr;
    bool apply_used = false;
};

template <typename T>
static int upload(d4w_fk_plan* pl, const std::vector<T>& h, const T** out) {
    void* p = nullptr;
    D4W_HIP(hipMalloc(&p, std::max<size_t>(h.size(), 1) * sizeof(T)));
    pl->allocs.push_back(p);
    D4W_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return D4W_OK;
}

static int make_axis(d4w_fk_plan* pl, int L, AxisDesc* ax, std::vector<int>* p2f,
                     const std::vector<int>* forced = nullptr) {
    std::vector<int> rad;
    if (forced) rad = *forced;
    else if (!factor_radices(L, rad))
        return fail(D4W_EINVAL, "length %d has a prime factor > 31 and this axis has no Bluestein form (only the channel axis of the f-k filter has); dsp.supported_length(n) gives the nearest shorter supported length", L);
    ax->L = L;
    ax->nstage = (L == 1) ? 0 : (int)rad.size();
    for (int i = 0; i < kMaxStages; ++i) ax->radix[i] = (i < (int)rad.size() && L > 1) ? rad[i] : 1;
    if (L == 1) rad.clear();
    *p2f = pos_to_freq(L, rad);
    return upload(pl, twiddle_table2(L, &ax->nhi), &ax->tw2);
}

template <typename K, typename... Args>
static int launch_k(K kern, dim3 grid, dim3 blk, size_t lds, void* stream, Args... args) {
    hipLaunchKernelGGL(kern, grid, blk, lds, (hipStream_t)stream, args...);
    D4W_HIP(hipGetLastError());
    return D4W_OK;
}

// n with every prime factor <= 31 divided out
static int rough_part(int n) {
    for (int p = 2; p <= 31; ++p)
        while (n % p == 0) n /= p;
    return n;
}

static int largest_divisor_le(int n, int lim) {
    int best = 1;
    for (int dd = 1; dd <= lim && dd <= n; ++dd)
        if (n % dd == 0) best = dd;
    return best;
}

extern "C" {

const char* d4w_last_error(void) { return d4w::g_err; }

const char* d4w_version(void) {
    return D4W_BUILD_TAG;
}

/* 1 when the shape runs shape-specialised kernels (built in or registered), 0 when it runs the generic passes */
int d4w_fk_shape_is_specialised(int nx, int ns) {
    for (const FkFastEntry& e : fast_shapes())
        if (e.nx == nx && e.ns == ns) return 1;
    std::lock_guard<std::mutex> lk(g_dyn_mu);
    for (const FkFastEntry* e : dyn_shapes())
        if (e->nx == nx && e->ns == ns) return 1;
    return 0;
}

/* Registers a shape configuration compiled on demand (the caller passes the FkFastEntry its own translation unit
 * built from csrc/fk_entry.h; entry_size guards against a header mismatch). */
int d4w_fk_register_shape(const void* entry, size_t entry_size) {
    if (!entry || entry_size != sizeof(FkFastEntry)) return fail(D4W_EINVAL, "shape entry of %zu bytes, expected %zu (stale build?)", entry_size, sizeof(FkFastEntry));
    FkFastEntry* e = new FkFastEntry(*static_cast<const FkFastEntry*>(entry));
    if (e->nx < 1 || e->ns < 2 || e->C2X < 1 || e->C1 * e->C2A * e->C2B * e->C2X != e->nx || (e->C2X > 1 && e->C2A * e->C2B != 1) ||
        2 * e->N1 * e->NA * e->NB * e->NC != e->ns || e->TA != e->TC) {
        delete e;
        return fail(D4W_EINVAL, "inconsistent shape entry");
    }
    std::lock_guard<std::mutex> lk(g_dyn_mu);
    dyn_shapes().push_back(e);
    return D4W_OK;
}

int d4w_fk_plan_destroy(d4w_fk_plan* pl) {
    if (pl && pl->big) { d4w_fkd_plan_destroy(pl->big); pl->big = nullptr; }
    if (!pl) return D4W_OK;
    for (void* p : pl->allocs) (void)hipFree(p);
    if (pl->d_cmask) (void)hipFree(pl->d_cmask);
    if (pl->d_W) (void)hipFree(pl->d_W);
    if (pl->apply_done) (void)hipEventDestroy(pl->apply_done);
    if (pl->d_colsrc) (void)hipFree(pl->d_colsrc);
    if (pl->d_cmlive) (void)hipFree(pl->d_cmlive);
    delete pl;
    return D4W_OK;
}

static int fk_plan_build(int nx, int ns, const int* opts, bool alloc_mask, bool fast_only, d4w_fk_plan** out);

int d4w_fk_plan_create_ex(int nx, int ns, const int* opts, d4w_fk_plan** out) {
    return fk_plan_build(nx, ns, opts, true, false, out);
}

// alloc_mask = false: tables only (the distributed plan keeps its own slab mask); fast_only: D4W_EINVAL unless the
// shape has specialised kernels
static int fk_plan_build(int nx, int ns, const int* opts, bool alloc_mask, bool fast_only, d4w_fk_plan** out) {
    if (!out) return fail(D4W_EINVAL, "plan pointer is NULL");
    *out = nullptr;
    if (nx < 1 || ns < 2) return fail(D4W_EINVAL, "bad shape %d x %d", nx, ns);
    if (ns % 2) return fail(D4W_EINVAL, "ns = %d must be even (packed real transform)", ns);
    const int M = ns / 2;
    int o[6] = {0, 0, 0, 0, 0, 0};
    if (opts) memcpy(o, opts, sizeof(o));

    // --- shape-specialised kernels (unless the caller pins a tiling or asks for the generic path
    //     with opts[0] = -1 / D4W_FK_GENERIC=1)
    const FkFastEntry* fast = nullptr;
    {
        bool pinned = false;
        for (int i = 0; i < 6; ++i) pinned |= (o[i] != 0);
        if (o[0] < 0) o[0] = 0;                      // -1 = "generic kernels, planner's tiling"
        const char* g = getenv("D4W_FK_GENERIC");
        const char* vs = getenv("D4W_FK_VARIANT");
        const int want = vs ? atoi(vs) : 0;
        if (!pinned && !(g && atoi(g) > 0)) {
            for (const FkFastEntry& e : fast_shapes())
                if (e.nx == nx && e.ns == ns && (e.variant == want || (!fast && e.variant == 0))) fast = &e;
            if (!fast) {
                std::lock_guard<std::mutex> lk(g_dyn_mu);
                for (const FkFastEntry* e : dyn_shapes())
                    if (e->nx == nx && e->ns == ns) fast = e;
            }
        }
    }
    if (fast_only && !fast) return fail(D4W_EINVAL, "no shape-specialised kernels for %d x %d", nx, ns);
    // An axis whose part with prime factors > 31 does not fit the Bluestein tiles of passes C / B: the global-memory forms
    // (fkd_bz_* / fkd_bt_*) of the generic distributed plan at world 1, behind this plan's entry points
    auto make_big = [&]() -> int {
        d4w_fkd_plan* big = nullptr;
        int rcb = fkd_plan_build(nx, ns, 1, 0, true, &big);
        if (rcb) return rcb;
        int info[12];
        (void)d4w_fkd_plan_info(big, info);
        d4w_fk_plan* bp = new d4w_fk_plan();
        memset(&bp->dev, 0, sizeof(bp->dev));
        bp->dev.d = FkDims{nx, ns, M, info[9], info[10], info[6], info[7], 1, 1};
        bp->big = big;
        bp->live_rows = nx;
        *out = bp;
        return D4W_OK;
    };
    // A prime factor > 31 of ns / 2 goes into N2, whose sub-transforms (pass B) then run as Bluestein convolutions of
    // length bn_L = 2^a 3^b 5^c >= 2 N2 - 1 inside the tile (two rows of bn_L); N1 keeps the smooth part.
    int bn_L = 0, bn_N2 = 0;
    if (!fast && rough_part(M) > 1) {
        bn_N2 = rough_part(M);
        bn_L = smooth_len_235(2L * bn_N2 - 1);
        if (2L * bn_L > kMaxTile) return make_big();
    }
    // admissible time splits: N1 | M with N2 = M / N1 fitting one LDS row pair
    int n1_min = 0;
    for (int cand = 1; cand <= M && !n1_min; ++cand)
        if (M % cand == 0 && M / cand <= kMaxTile / 2) n1_min = cand;
    if (bn_L) n1_min = M / bn_N2;
    if (!n1_min) return fail(D4W_EINVAL, "ns/2 = %d has no factorisation with N2 <= %d", M, kMaxTile / 2);
    // --- split the channel axis
    int C1 = o[0], C2 = o[1];
    if (C1 <= 0 || C2 <= 0 || C1 * C2 != nx) {
        // measured on MI355X (20000 x 120000): a long c2 axis with 64-byte column segments in
        // pass C leaves room for 128-byte segments in the 2-D pass A and wins overall
        C2 = largest_divisor_le(nx, kMaxTile / 8);
        // ... but not at the price of fewer than 8 c1 rows: pass A's tile is C1 x N1 x TA, and a tile of a few
        // dozen elements costs more than a shorter c2 axis saves (1000 x 12000: C1 = 1 1.9 ms, C1 = 10 0.47 ms)
        if (nx / C2 < 8)
            for (int cand = C2; cand >= 1; --cand)
                if (nx % cand == 0 && nx / cand >= 8) { C2 = cand; break; }
        C1 = nx / C2;
    }
    // a prime factor > 31 of nx goes into C2, whose sub-transform then runs as a Bluestein convolution of
    // length bs_L (pass C); C1 keeps the smooth part
    int bs_L = 0;
    if (fast && fast->C2X > 1) {                       // specialised passes A and B around the generic Bluestein pass C
        C2 = fast->C2X;
        C1 = nx / C2;
        bs_L = smooth_len_235(2L * C2 - 1);
        if (bs_L > kMaxTile) return fail(D4W_EINVAL, "registered configuration: Bluestein factor %d too long", C2);
    }
    if (!fast && rough_part(nx) > 1) {
        C2 = rough_part(nx);
        C1 = nx / C2;
        for (int f = 2; f <= 31 && (long)C1 * n1_min > kMaxTile; ++f)      // pass A's tile must hold C1 x N1 columns
            while (C1 % f == 0 && (long)C1 * n1_min > kMaxTile && 2L * C2 * f - 1 <= kMaxTile) { C2 *= f; C1 /= f; }
        bs_L = smooth_len_235(2L * C2 - 1);
        if (bs_L > kMaxTile) {
            // too long for the Bluestein tile of pass C: the global-memory form
            return make_big();
        }
    }
    // --- split the time axis.  Pass A's cost falls like 1 / (C1 N1) until its tile holds ~100 columns, pass B's
    //     grows with N1 (shorter rows): N1 ~ sqrt(1000 / C1), at least the smallest admissible one
    //     (measured at 1000 x 12000 and 500 x 120000 over C1 = 1..10, N1 = 2..75)
    int N1 = o[2], N2 = o[3];
    if (bn_L) {
        // smooth factors move from N1 into N2 while the Bluestein tile allows it and pass A's tile needs it
        N1 = M / bn_N2; N2 = bn_N2;
        for (int f = 2; f <= 31 && (long)C1 * N1 > kMaxTile; ++f)
            while (N1 % f == 0 && (long)C1 * N1 > kMaxTile && 2L * (2L * N2 * f - 1) <= kMaxTile) { N2 *= f; N1 /= f; }
        bn_L = smooth_len_235(2L * N2 - 1);
        if (2L * bn_L > kMaxTile || (long)C1 * N1 > kMaxTile) return make_big();
    } else if (N1 <= 0 || N2 <= 0 || N1 * N2 != M) {
        N1 = n1_min;
        const double target = sqrt(1000.0 / (double)C1);
        double best = fabs(log((double)N1 / target));
        if (target > n1_min)
            for (int cand = n1_min + 1; cand <= M && cand <= 4 * target; ++cand) {
                if (M % cand || M / cand < 64 || (long)C1 * cand * 16 > kMaxTile) continue;
                const double sc = fabs(log((double)cand / target));
                if (sc < best) { best = sc; N1 = cand; }
            }
        N2 = M / N1;
    }
    int TC = o[5] > 0 ? o[5] : 16;
    while (TC > 1 && (long)(bs_L ? bs_L : C2) * TC > (bs_L ? kMaxTileBs : kMaxTile)) TC /= 2;
    int TA = o[4] > 0 ? o[4] : 16;
    while (TA > 1 && (long)C1 * N1 * TA > kMaxTile) TA /= 2;
    std::vector<int> r_c1, r_c2, r_n1, r_n2;
    if (fast) {
        C1 = fast->C1; C2 = fast->C2A * fast->C2B * fast->C2X; N1 = fast->N1; N2 = fast->NA * fast->NB * fast->NC;
        TA = fast->TA;
        if (!bs_L) TC = fast->TC;                       // Bluestein pass C keeps its own strip width
        r_c1 = {C1}; r_c2 = {fast->C2A, fast->C2B}; r_n1 = {N1}; r_n2 = {fast->NA, fast->NB, fast->NC};
    }
    if (!fast && ((long)(bs_L ? bs_L : C2) * TC > (bs_L ? kMaxTileBs : kMaxTile) || (long)C1 * N1 * TA > kMaxTile || 2L * (bn_L ? bn_L : N2) > kMaxTile))
        return fail(D4W_EINVAL, "shape %d x %d does not fit the LDS tiling (C1=%d C2=%d N1=%d N2=%d)",
                    nx, ns, C1, C2, N1, N2);

    d4w_fk_plan* pl = new d4w_fk_plan();
    memset(&pl->dev, 0, sizeof(pl->dev));
    FkDims& d = pl->dev.d;
    d = FkDims{nx, ns, M, C1, C2, N1, N2, TA, TC};
    pl->dev.scale = (float)(1.0 / ((double)nx * (double)M));
    int rc;
    std::vector<int> f_c1, f_c2, f_n1, f_n2;
#define D4W_TRY(x) do { rc = (x); if (rc != D4W_OK) { d4w_fk_plan_destroy(pl); return rc; } } while (0)
    pl->fast = fast;
    D4W_TRY(make_axis(pl, C1, &pl->dev.ax_c1, &f_c1, fast ? &r_c1 : nullptr));
    if (bs_L) {
        std::vector<int> one, f_L;
        D4W_TRY(make_axis(pl, 1, &pl->dev.ax_c2, &one));                      // unused: pass C runs fk_passC_bluestein
        D4W_TRY(make_axis(pl, bs_L, &pl->dev.ax_bs, &f_L));
        f_c2.resize(C2);
        for (int i = 0; i < C2; ++i) f_c2[i] = i;                             // natural output order
        std::vector<float2> chirp(C2), filt(bs_L);
        std::vector<double> bre(bs_L, 0.0), bim(bs_L, 0.0);
        for (int n = 0; n < C2; ++n) {
            const double ph = M_PI * (double)(((long long)n * n) % (2LL * C2)) / (double)C2;
            chirp[n] = make_float2((float)cos(ph), (float)-sin(ph));
            bre[n] = cos(ph); bim[n] = sin(ph);                               // conj(chirp)
            if (n) { bre[bs_L - n] = cos(ph); bim[bs_L - n] = sin(ph); }
        }
        host_dft_any(bre, bim);
        for (int p = 0; p < bs_L; ++p)
            filt[p] = make_float2((float)(bre[f_L[p]] / bs_L), (float)(bim[f_L[p]] / bs_L));
        D4W_TRY(upload(pl, chirp, &pl->dev.bs_chirp));
        D4W_TRY(upload(pl, filt, &pl->dev.bs_filt));
        pl->dev.bs_L = bs_L;
    } else
        D4W_TRY(make_axis(pl, C2, &pl->dev.ax_c2, &f_c2, fast ? &r_c2 : nullptr));
    D4W_TRY(make_axis(pl, N1, &pl->dev.ax_n1, &f_n1, fast ? &r_n1 : nullptr));
    if (bn_L) {
        std::vector<int> one, f_L;
        D4W_TRY(make_axis(pl, 1, &pl->dev.ax_n2, &one));                      // unused: pass B runs fk_passB_bluestein
        D4W_TRY(make_axis(pl, bn_L, &pl->dev.ax_bn, &f_L));
        f_n2.resize(N2);
        for (int i = 0; i < N2; ++i) f_n2[i] = i;                             // natural output order
        std::vector<float2> chirp(N2), filt(bn_L);
        std::vector<double> bre(bn_L, 0.0), bim(bn_L, 0.0);
        for (int n = 0; n < N2; ++n) {
            const double ph = M_PI * (double)(((long long)n * n) % (2LL * N2)) / (double)N2;
            chirp[n] = make_float2((float)cos(ph), (float)-sin(ph));
            bre[n] = cos(ph); bim[n] = sin(ph);                               // conj(chirp)
            if (n) { bre[bn_L - n] = cos(ph); bim[bn_L - n] = sin(ph); }
        }
        host_dft_any(bre, bim);
        for (int p = 0; p < bn_L; ++p)
            filt[p] = make_float2((float)(bre[f_L[p]] / bn_L), (float)(bim[f_L[p]] / bn_L));
        D4W_TRY(upload(pl, chirp, &pl->dev.bn_chirp));
        D4W_TRY(upload(pl, filt, &pl->dev.bn_filt));
        pl->dev.bn_L = bn_L;
    } else
        D4W_TRY(make_axis(pl, N2, &pl->dev.ax_n2, &f_n2, fast ? &r_n2 : nullptr));
    if (fast) {     // exchange-stage twiddles of the specialised kernels
        const int RA = fast->C2A, RB = fast->C2B, M1 = fast->NB * fast->NC;
        std::vector<float2> twC((size_t)RA * RB), twB1(M1), twB2(M1);
        for (int a = 0; a < RA; ++a)
            for (int j = 0; j < RB; ++j) twC[(size_t)a * RB + j] = wexp((long long)j * a, C2);
        for (int j = 0; j < M1; ++j) twB1[j] = wexp(j, N2);
        for (int b = 0; b < fast->NB; ++b)
            for (int j2 = 0; j2 < fast->NC; ++j2) twB2[(size_t)b * fast->NC + j2] = wexp((long long)j2 * b, M1);
        D4W_TRY(upload(pl, twC, &pl->fdev.twC));
        D4W_TRY(upload(pl, twB1, &pl->fdev.twB1));
        D4W_TRY(upload(pl, twB2, &pl->fdev.twB2));
    }

    // inverse maps frequency -> position
    std::vector<int> p_c1(C1), p_c2(C2), p_n1(N1), p_n2(N2);
    for (int p = 0; p < C1; ++p) p_c1[f_c1[p]] = p;
    for (int p = 0; p < C2; ++p) p_c2[f_c2[p]] = p;
    for (int p = 0; p < N1; ++p) p_n1[f_n1[p]] = p;
    for (int p = 0; p < N2; ++p) p_n2[f_n2[p]] = p;

    // four-step twiddles
    std::vector<float2> twc((size_t)C1 * C2), twt((size_t)N1 * N2);
    for (int q = 0; q < C1; ++q)
        for (int c2 = 0; c2 < C2; ++c2) twc[(size_t)q * C2 + c2] = wexp((long long)c2 * f_c1[q], nx);
    for (int q1 = 0; q1 < N1; ++q1)
        for (int b = 0; b < N2; ++b) twt[(size_t)q1 * N2 + b] = wexp((long long)b * f_n1[q1], M);
    D4W_TRY(upload(pl, twc, &pl->dev.twc));
    D4W_TRY(upload(pl, twt, &pl->dev.twt));

    // Hermitian partner maps
    std::vector<int> rowk(nx), rowpart(nx), q1part(N1), mirror0(N2);
    for (int q = 0; q < C1; ++q)
        for (int p2 = 0; p2 < C2; ++p2) rowk[q * C2 + p2] = f_c1[q] + C1 * f_c2[p2];
    for (int r = 0; r < nx; ++r) {
        const int km = (nx - rowk[r]) % nx;
        rowpart[r] = p_c1[km % C1] * C2 + p_c2[km / C1];
    }
    for (int q1 = 0; q1 < N1; ++q1) q1part[q1] = p_n1[(N1 - f_n1[q1]) % N1];
    for (int i = 0; i < N2; ++i) mirror0[i] = p_n2[(N2 - f_n2[i]) % N2];
    {   // pass-B work list: every Hermitian pair of sub-rows once
        std::vector<int2> pairs;
        pairs.reserve((size_t)nx * N1 / 2 + 4);
        for (int r = 0; r < nx; ++r)
            for (int q1 = 0; q1 < N1; ++q1) {
                const long keyA = (long)r * N1 + q1, keyB = (long)rowpart[r] * N1 + q1part[q1];
                if (keyB >= keyA) pairs.push_back(make_int2((int)keyA, (int)keyB));
            }
        pl->npairs = (int)pairs.size();
        D4W_TRY(upload(pl, pairs, &pl->dev.pairs));
        pl->h_pairs = pairs;
        pl->npairs_run = pl->npairs;
        pl->live_rows = nx;
        pl->fdev.pairs = pl->dev.pairs;
        pl->fdev.live = nullptr;
        {
            void* q = nullptr;
            if (hipMalloc(&q, (size_t)nx * sizeof(unsigned)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
            pl->allocs.push_back(q); pl->d_rowmax = (unsigned*)q;
        }
        if (fast) {
            void* q = nullptr;
            if (hipMalloc(&q, (size_t)C1 * fast->C2A * sizeof(unsigned)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
            pl->allocs.push_back(q); pl->d_livebits = (unsigned*)q;
            if (hipMalloc(&q, pairs.size() * sizeof(int2)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
            pl->allocs.push_back(q); pl->d_pairs_live = (int2*)q;
        }
    }
    if (fast && fast->Bt_fwd) {
        // time-first order: work list of passes Bf / Bi.  BEFORE the c2 transform the Hermitian partner of row
        // (q, c2) is (q', c2), kc1(q') = -kc1(q) (fk_tf.h); the partner sub-row is the same q1' as in pass B.
        std::vector<int2> pairsT;
        pairsT.reserve((size_t)nx * N1 / 2 + 4);
        for (int r = 0; r < nx; ++r) {
            const int q = r / C2, c2 = r - q * C2;
            const int rb = p_c1[(C1 - f_c1[q]) % C1] * C2 + c2;
            for (int q1 = 0; q1 < N1; ++q1) {
                const long keyA = (long)r * N1 + q1, keyB = (long)rb * N1 + q1part[q1];
                if (keyB >= keyA) pairsT.push_back(make_int2((int)keyA, (int)keyB));
            }
        }
        // sorted by sub-row: a workgroup's consecutive pairs then share the tail gains of their sub-row (fkf_passBt PHASE 1)
        std::stable_sort(pairsT.begin(), pairsT.end(), [N1](const int2& a, const int2& b) { return a.x % N1 < b.x % N1; });
        pl->npairsT = (int)pairsT.size();
        D4W_TRY(upload(pl, pairsT, &pl->tfdev.pairs));
        void* q = nullptr;
        const size_t nctab = (size_t)fast->NC * fast->NB;
        if (hipMalloc(&q, 2 * (size_t)M * sizeof(unsigned)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        pl->allocs.push_back(q); pl->d_colkeys = (unsigned*)q;
        if (hipMalloc(&q, (size_t)M * sizeof(float)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        pl->allocs.push_back(q); pl->d_tgain = (float*)q;
        if (hipMalloc(&q, nctab * sizeof(int2)) != hipSuccess) { d4w_fk_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        pl->allocs.push_back(q); pl->d_ctab = (int2*)q;
    }
    D4W_TRY(upload(pl, rowpart, &pl->dev.row_partner));
    D4W_TRY(upload(pl, q1part, &pl->dev.q1_partner));
    D4W_TRY(upload(pl, mirror0, &pl->dev.mirror0));
    const int *c_rowk, *c_k1, *c_k2;
    D4W_TRY(upload(pl, rowk, &c_rowk));
    D4W_TRY(upload(pl, f_n1, &c_k1));
    D4W_TRY(upload(pl, f_n2, &c_k2));
    {
        const int* c_q1k;
        D4W_TRY(upload(pl, p_n1, &c_q1k));
        pl->d_q1_of_k1 = const_cast<int*>(c_q1k);
        // tiling of the mask fold: runs of N1 * R0 samples in, IB positions out, transposed in <= 144 KiB of LDS
        const int R0 = pl->dev.ax_n2.nstage > 0 ? pl->dev.ax_n2.radix[0] : 1;
        const long RL = (long)N1 * R0;
        int IB = (int)std::min<long>(32, 16384 / RL - 1);          // <= 64 KiB of LDS: two 1024-thread workgroups per CU
        IB = std::min(IB, N2 / R0);
        if (IB >= 1) { pl->fold_R0 = R0; pl->fold_IB = IB; }
    }
    pl->d_rowk = const_cast<int*>(c_rowk);
    pl->d_k1 = const_cast<int*>(c_k1);
    pl->d_k2 = const_cast<int*>(c_k2);

    std::vector<float2> wrow(N1), wcol(N2);
    for (int q1 = 0; q1 < N1; ++q1) wrow[q1] = wexp(f_n1[q1], ns);
    for (int i = 0; i < N2; ++i) wcol[i] = wexp((long long)N1 * f_n2[i], ns);
    D4W_TRY(upload(pl, wrow, &pl->dev.wrow));
    D4W_TRY(upload(pl, wcol, &pl->dev.wcol));

    // tukey(ns, 0.03) both flat (taper_data) and packed (pass A)
    std::vector<float> win = tukey_window(ns, 0.03);
    std::vector<float2> winp(M);
    for (int m = 0; m < M; ++m) winp[m] = make_float2(win[2 * m], win[2 * m + 1]);
    D4W_TRY(upload(pl, winp, &pl->dev.win));

    void* p = nullptr;
    if (alloc_mask) {
        if (hipMalloc(&p, (size_t)nx * M * sizeof(float)) != hipSuccess) {
            d4w_fk_plan_destroy(pl);
            return fail(D4W_ENOMEM, "hipMalloc of the %zu-byte mask failed", (size_t)nx * M * sizeof(float));
        }
        pl->allocs.push_back(p);
        pl->d_mask = (float*)p;
        pl->dev.mask = pl->d_mask;
    }
    if (hipMalloc(&p, (size_t)nx * sizeof(float)) != hipSuccess) {
        d4w_fk_plan_destroy(pl);
        return fail(D4W_ENOMEM, "hipMalloc failed");
    }
    pl->allocs.push_back(p);
    pl->d_nyq = (float*)p;
    pl->dev.nyq = pl->d_nyq;
#undef D4W_TRY

    auto needs_generic = [](const AxisDesc& ax) {
        for (int s = 0; s < ax.nstage; ++s) {
            bool fast = false;
            for (int r : kFastRadix) fast |= (r == ax.radix[s]);
            if (!fast) return true;
        }
        return false;
    };
    pl->genericA = needs_generic(pl->dev.ax_c1) || needs_generic(pl->dev.ax_n1);
    pl->genericB = needs_generic(pl->dev.ax_n2);
    pl->genericC = needs_generic(pl->dev.ax_c2);
    pl->ldsA = ((size_t)C1 * N1 * TA + 2 * kTwLo + pl->dev.ax_c1.nhi + pl->dev.ax_n1.nhi) * sizeof(float2);
    pl->ldsC = ((size_t)C2 * TC + kTwLo + pl->dev.ax_c2.nhi) * sizeof(float2);
    if (bs_L) pl->ldsC = ((size_t)bs_L * TC + kTwLo + pl->dev.ax_bs.nhi) * sizeof(float2);
    pl->ldsB = ((size_t)2 * N2 + kTwLo + pl->dev.ax_n2.nhi) * sizeof(float2);
    if (bn_L) pl->ldsB = ((size_t)2 * bn_L + kTwLo + pl->dev.ax_bn.nhi) * sizeof(float2);
    auto env_int = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return (v && atoi(v) > 0) ? atoi(v) : dflt;
    };
    pl->wg_per_cu = env_int("D4W_FK_WG_PER_CU", 2048 / kMaxThreads);     // tuning knob
    if (fast) {
        pl->wgA = env_int("D4W_FK_WG_A", fast->wgA);
        pl->wgC = env_int("D4W_FK_WG_C", fast->wgC);
        pl->wgB = env_int("D4W_FK_WG_B", fast->wgB);
        pl->wgBi = pl->wgBf = pl->wgB;
        pl->slab_sw = env_int("D4W_FK_SLAB", 0);
    }
    {
        int devid = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
            pl->num_cu = prop.multiProcessorCount;
        if (fast) {
            (void)hipFuncSetAttribute((const void*)fast->A_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsA);
            (void)hipFuncSetAttribute((const void*)fast->A_fwd_taper, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsA);
            (void)hipFuncSetAttribute((const void*)fast->A_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsA);
            (void)hipFuncSetAttribute((const void*)fast->A_inv_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsA);
            (void)hipFuncSetAttribute((const void*)fast->C_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsC);
            (void)hipFuncSetAttribute((const void*)fast->C_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsC);
            (void)hipFuncSetAttribute((const void*)fast->B_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsB);
            const void* fa[] = {(const void*)fast->T_fwd, (const void*)fast->T_fwd_taper, (const void*)fast->T_inv, (const void*)fast->T_inv_stats,
                                (const void*)fast->Ac_fwd, (const void*)fast->Ac_inv, (const void*)fast->T_inv_env};
            for (const void* f : fa) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsA);
            (void)hipFuncSetAttribute((const void*)fast->Cs_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsC);
            (void)hipFuncSetAttribute((const void*)fast->Cs_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsC);
            (void)hipFuncSetAttribute((const void*)fast->Bs_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsB);
            (void)hipFuncSetAttribute((const void*)fast->Bs_hilb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsB);
            if (fast->Bt_fwd) {
                (void)hipFuncSetAttribute((const void*)fast->Bt_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsBt);
                (void)hipFuncSetAttribute((const void*)fast->Bt_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsBt);
                if (fast->C_mid) (void)hipFuncSetAttribute((const void*)fast->C_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast->ldsC);
                // (below) resident workgroups of the pass-B forms from what their registers and LDS allow
            }
            // Pass B in its three forms is bound by VALU issue AND by what its few waves can overlap: a CU takes as many
            // workgroups of each form as that form's registers (512 per SIMD lane, granules of 8) and LDS (160 KiB) allow,
            // at most 3 (4 measured slower) -- at 20 000 x 120 000: fused B 161 VGPRs and Bi 150 -> three 4-wave workgroups
            // (Bi 3.54 -> 3.2 ms, the part's copy rate), Bf 210 -> two.  The table's wgB is the floor.
            auto resident = [&](const void* fn, size_t lds) {
                hipFuncAttributes fa;
                if (!fn || hipFuncGetAttributes(&fa, fn) != hipSuccess || fa.numRegs <= 0) return pl->wgB;
                const int waves_simd = 512 / (((int)fa.numRegs + 7) / 8 * 8);
                const int by_regs = waves_simd * 4 / std::max(1, (fast->thrB + 63) / 64);
                const int by_lds = (int)((160 * 1024) / std::max<size_t>(lds, 1));
                return std::max(pl->wgB, std::min(3, std::min(by_regs, by_lds)));
            };
            pl->wgBi = env_int("D4W_FK_WG_BI", resident((const void*)fast->Bt_inv, fast->ldsBt));
            pl->wgBf = env_int("D4W_FK_WG_BF", resident((const void*)fast->Bt_fwd, fast->ldsBt));
            if (!getenv("D4W_FK_WG_B")) pl->wgB = resident((const void*)fast->B_mid, fast->ldsB);

        }
        const size_t lds_max = std::max(pl->ldsA, std::max(pl->ldsB, pl->ldsC));
        if (!fast && lds_max > 64 * 1024) {
            // tiles are capped at 64 KiB; the twiddle tables push the request slightly above the
            // default dynamic-LDS limit
            const void* fns[] = {
                (const void*)fk_passA_fwd<true, true>, (const void*)fk_passA_fwd<true, false>,
                (const void*)fk_passA_fwd<false, true>, (const void*)fk_passA_fwd<false, false>,
                (const void*)fk_passA_inv<true>, (const void*)fk_passA_inv<false>,
                (const void*)fk_passC<false, true>, (const void*)fk_passC<false, false>,
                (const void*)fk_passC<true, true>, (const void*)fk_passC<true, false>,
                (const void*)fk_passC_bluestein<false>, (const void*)fk_passC_bluestein<true>,
                (const void*)fk_passB<true>, (const void*)fk_passB<false>, (const void*)fk_passB_bluestein};
            for (const void* f : fns)
                (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        }
        if (bs_L) {
            (void)hipFuncSetAttribute((const void*)fk_passC_bluestein<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            (void)hipFuncSetAttribute((const void*)fk_passC_bluestein<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            (void)hipFuncSetAttribute((const void*)fk_passCm_bluestein, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        }
    }
    *out = pl;
    return D4W_OK;
}

int d4w_fk_plan_create(int nx, int ns, d4w_fk_plan** out) { return d4w_fk_plan_create_ex(nx, ns, nullptr, out); }

int d4w_fk_plan_info(const d4w_fk_plan* pl, int* info) {
    if (!pl || !info) return fail(D4W_EINVAL, "NULL argument");
    const FkDims& d = pl->dev.d;
    const int v[8] = {d.nx, d.ns, d.C1, d.C2, d.N1, d.N2, d.TA, d.TC};
    memcpy(info, v, sizeof(v));
    return D4W_OK;
}

static int fk_mask_finish(d4w_fk_plan* pl, double prune_eps, void* stream);

// device-side order of the operations on one plan (see d4w_fk_plan::apply_mu); the caller holds pl->apply_mu
static int fk_plan_enter(d4w_fk_plan* pl, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (pl->apply_used && pl->apply_stream != st) D4W_HIP(hipStreamWaitEvent(st, pl->apply_done, 0));
    return D4W_OK;
}
static int fk_plan_leave(d4w_fk_plan* pl, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!pl->apply_done) D4W_HIP(hipEventCreateWithFlags(&pl->apply_done, hipEventDisableTiming));
    D4W_HIP(hipEventRecord(pl->apply_done, st));
    pl->apply_stream = st;
    pl->apply_used = true;
    return D4W_OK;
}

static int fk_set_mask_run(d4w_fk_plan* pl, const float* mask_shifted, double prune_eps, void* stream, FkAffine aff);

static int fk_set_mask_impl(d4w_fk_plan* pl, const float* mask_shifted, double prune_eps, void* stream, FkAffine aff = FkAffine{1.f, 0.f, 0}) {
    if (!pl) return fail(D4W_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(pl->apply_mu);
    int rc = fk_plan_enter(pl, stream);
    if (rc != D4W_OK) return rc;
    rc = fk_set_mask_run(pl, mask_shifted, prune_eps, stream, aff);
    const int rl = fk_plan_leave(pl, stream);
    return rc != D4W_OK ? rc : rl;
}

static int fk_set_mask_run(d4w_fk_plan* pl, const float* mask_shifted, double prune_eps, void* stream, FkAffine aff) {
    if (!(prune_eps >= 0.0 && prune_eps < 1.0)) return fail(D4W_EINVAL, "prune_eps = %g not in [0, 1)", prune_eps);
    if (pl && pl->big) {                            // (no dead-row pruning on this path: prune_eps acts on the Bluestein channel phase's columns)
        fkd_set_prune(pl->big, prune_eps);
        int rc = fkd_set_mask_dense_affine(pl->big, mask_shifted, aff.a, aff.b, aff.on, stream);
        if (rc == D4W_OK) pl->has_mask = true;
        return rc;
    }
    if (!pl || !mask_shifted || !pl->d_mask) return fail(D4W_EINVAL, "NULL argument");
    const FkDims& d = pl->dev.d;
    hipStream_t st = (hipStream_t)stream;
    D4W_HIP(hipMemsetAsync(pl->d_rowmax, 0, (size_t)d.nx * sizeof(unsigned), st));
    const char* gather = getenv("D4W_FK_FOLD_GATHER");
    if (pl->fold_IB > 0 && !(gather && atoi(gather) > 0)) {
        const int N2r = d.N2 / pl->fold_R0;
        const size_t lds = (size_t)d.N1 * pl->fold_R0 * (pl->fold_IB + 1) * sizeof(float);
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)fk_fold_mask_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024); });
        if (d.nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", d.nx);
        D4W_LAUNCH(fk_fold_mask_tiled, dim3(ceil_div(N2r, pl->fold_IB), d.nx), dim3(kFoldThreads), lds, stream, d, pl->fold_R0,
                   pl->fold_IB, mask_shifted, (const int*)pl->d_rowk, (const int*)pl->d_q1_of_k1, (const int*)pl->d_k2,
                   pl->d_mask, pl->d_nyq, pl->d_rowmax, aff);
    } else {
        dim3 grid(std::min(ceil_div(d.M, kThreads), 64), d.nx);
        D4W_LAUNCH(fk_fold_mask, grid, dim3(kThreads), 0, stream, d, mask_shifted, (const int*)pl->d_rowk,
                   (const int*)pl->d_k1, (const int*)pl->d_k2, pl->d_mask, pl->d_nyq, pl->d_rowmax, aff);
    }
    return fk_mask_finish(pl, prune_eps, stream);
}

// Gains the filter treats as zero.  Opt-in: prune_eps * max|M_h| (approximate, FkPlan.set_mask(m, prune_eps=...)).
// Always: D4W_FK_ROUND_EPS (default 2^-24) / sqrt(nx ns) * max|M_h| -- what such gains add to ANY output sample is at
// most  g ||x||_F  =  2^-24 max|M_h| rms(x)  (Cauchy-Schwarz over the spectrum): half a float32 ulp of the input's RMS,
// below the rounding noise of the float32 transforms themselves, whatever the input.
static double fk_zero_gain(const FkDims& d, double prune_eps, double gmax) {
    static const double round_eps = [] {
        const char* v = getenv("D4W_FK_ROUND_EPS");
        return v ? atof(v) : 5.9604644775390625e-8;
    }();
    return std::max(prune_eps, round_eps / sqrt((double)d.nx * (double)d.ns)) * gmax;
}

static inline float fk_key_value(unsigned k) {
    const unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float v;
    memcpy(&v, &b, sizeof(v));
    return v;
}

// the time-first workspace (12 B x nx x Lc, up to 1.3x the data block) goes back to the allocator
static void fk_tf_release(d4w_fk_plan* pl) {
    if (pl->d_W) (void)hipFree(pl->d_W);
    if (pl->d_cmask) (void)hipFree(pl->d_cmask);
    if (pl->d_colsrc) (void)hipFree(pl->d_colsrc);
    if (pl->d_cmlive) (void)hipFree(pl->d_cmlive);
    pl->d_W = nullptr; pl->d_cmask = nullptr; pl->d_colsrc = nullptr; pl->d_cmlive = nullptr; pl->cap_W = 0;
}

// Time-first order for the mask just folded (fk_tf.h): classify the half-spectrum columns, lay the kept ones out,
// model the bytes of both orders; when time-first wins, build the pass tables, the compact band mask and the workspace.
static int fk_tf_setup(d4w_fk_plan* pl, double zero_gain, void* stream) {
    const FkDims& d = pl->dev.d;
    const FkFastEntry& F = *pl->fast;
    hipStream_t st = (hipStream_t)stream;
    const int NA = F.NA, NB = F.NB, NC = F.NC, N1 = d.N1, N2 = d.N2, M = d.M, TC = d.TC;
    pl->tf = false;
    pl->bytes_cf = 24.0 + 18.0 * (double)pl->live_rows / d.nx;
    pl->bytes_tf = 42.0;
    const char* ord = getenv("D4W_FK_ORDER");
    if (ord && !strcmp(ord, "cf")) return D4W_OK;
    D4W_HIP(hipMemsetAsync(pl->d_colkeys, 0xFF, (size_t)M * sizeof(unsigned), st));
    D4W_HIP(hipMemsetAsync(pl->d_colkeys + M, 0, (size_t)M * sizeof(unsigned), st));
    const int rpb = 512;
    D4W_LAUNCH(fk_col_minmax, dim3(ceil_div(M, kThreads), ceil_div(d.nx, rpb)), dim3(kThreads), 0, stream, d,
               (const float*)pl->d_mask, rpb, pl->d_colkeys, pl->d_colkeys + M);
    std::vector<unsigned> keys(2 * (size_t)M);
    std::vector<float> nyq(d.nx);
    D4W_HIP(hipMemcpyAsync(keys.data(), pl->d_colkeys, keys.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    D4W_HIP(hipMemcpyAsync(nyq.data(), pl->d_nyq, (size_t)d.nx * sizeof(float), hipMemcpyDeviceToHost, st));
    D4W_HIP(hipStreamSynchronize(st));
    // column classes at the positions p = q1 N2 + e: 0 dead, 1 band, 2 tail (gain the same for every wavenumber)
    std::vector<unsigned char> ccls(M);
    std::vector<float> cgain(M);
    for (int p = 0; p < M; ++p) {
        const float lo = fk_key_value(keys[p]), hi = fk_key_value(keys[M + p]);
        const double amax = std::max(fabs((double)lo), fabs((double)hi));
        cgain[p] = 0.5f * (lo + hi);
        if (!(lo == lo) || !(hi == hi) || std::isinf(lo) || std::isinf(hi)) ccls[p] = 1;        // NaN / inf gains: as NumPy, through the product
        else if (amax <= zero_gain) ccls[p] = 0;
        else if ((double)hi - (double)lo <= zero_gain) ccls[p] = 2;
        else ccls[p] = 1;
    }
    bool nyq_live = false;
    for (int r = 0; r < d.nx; ++r) nyq_live |= !(fabs((double)nyq[r]) <= zero_gain);
    // cells (d2, d1): the strongest class of their N1 NA columns; position e = d0 NB NC + d1 NC + d2
    std::vector<int> cell(NC * NB, 0);
    for (int q1 = 0; q1 < N1; ++q1)
        for (int e = 0; e < N2; ++e) {
            const int c = ccls[(size_t)q1 * N2 + e];
            if (!c) continue;
            const int d1 = (e / NC) % NB, d2 = e % NC;
            int& cc = cell[d2 * NB + d1];
            if (c == 1) cc = 1;
            else if (cc == 0) cc = 2;
        }
    // layout of a row of W: N1 sub-row blocks of RW columns, [band cells (bw, a multiple of TC) | tail cells] each, then one
    // strip for the Nyquist column
    std::vector<int2> ctab(NC * NB);
    int rw[3] = {0, 0, 0}, ncell[3] = {0, 0, 0};
    for (int c : cell) ncell[c]++;
    const int bw = ceil_div(NA * ncell[1], TC) * TC;
    for (int d2 = 0; d2 < NC; ++d2) {
        int cnt[3] = {0, 0, 0};
        for (int d1 = 0; d1 < NB; ++d1) cnt[cell[d2 * NB + d1]]++;
        int rank[3] = {0, 0, 0};
        for (int d1 = 0; d1 < NB; ++d1) {
            const int c = cell[d2 * NB + d1];
            ctab[d2 * NB + d1] = c ? make_int2((c == 2 ? bw : 0) + rw[c] + rank[c], cnt[c] | (c << 28)) : make_int2(0, 0);
            rank[c]++;
        }
        rw[1] += NA * cnt[1];
        rw[2] += NA * cnt[2];
    }
    const int RW = std::max(TC, ceil_div(bw + rw[2], TC) * TC);
    const int Lc = ceil_div(N1 * RW + (nyq_live ? TC : 0), 16) * 16;
    pl->tf_band_cols = N1 * rw[1] + (nyq_live ? 1 : 0);
    pl->tf_tail_cols = N1 * rw[2];
    pl->bytes_tf = 24.0 + 18.0 * (double)(N1 * bw + (nyq_live ? TC : 0)) / M + 8.0 * (double)(N1 * rw[2]) / M;
    const bool force = ord && !strcmp(ord, "tf");
    // channel-first stays (it needs no workspace): what an earlier mask of this plan allocated for the other order is freed
    if ((!force && !(pl->bytes_tf < 0.97 * pl->bytes_cf)) || (size_t)Lc * sizeof(float2) >= ((size_t)1 << 31)) {   // 32-bit column offsets inside a row
        fk_tf_release(pl);
        return D4W_OK;
    }
    // tables
    std::vector<float> tgain(M, 0.f);
    std::vector<int> colsrc((size_t)Lc, -1);
    for (int q1 = 0; q1 < N1; ++q1)
        for (int e = 0; e < N2; ++e) {
            const int d0 = e / (NB * NC), d1 = (e / NC) % NB, d2 = e % NC;
            const int c = cell[d2 * NB + d1];
            const size_t p = (size_t)q1 * N2 + e;
            if (c == 1) {
                tgain[p] = 1.f;
                const int2 ct = ctab[d2 * NB + d1];
                colsrc[(size_t)q1 * RW + ct.x + d0 * (ct.y & 0x0FFFFFFF)] = (int)p;
            } else if (c == 2)
                tgain[p] = cgain[p] * (float)d.C2;      // the band columns pick up C2 from the unnormalised c2 forward + inverse of Cm
        }
    if (nyq_live) colsrc[(size_t)N1 * RW] = -2;
    const size_t need = (size_t)d.nx * Lc;
    if (need > pl->cap_W) {
        fk_tf_release(pl);
        // Not enough memory for the workspace is not an error: the channel-first order works in place in the output block
        // and gives the same result (more bytes moved).  Everything allocated so far goes back; pl->tf stays false.
        const bool ok = hipMalloc((void**)&pl->d_W, need * sizeof(float2)) == hipSuccess &&
                        hipMalloc((void**)&pl->d_cmask, need * sizeof(float)) == hipSuccess &&
                        hipMalloc((void**)&pl->d_colsrc, ((size_t)M + 64 * (size_t)N1 + 64) * sizeof(int)) == hipSuccess &&
                        // Cm tiles: at most C1 (N1 (N2 / TC + 2) + 1) whatever the mask
                        hipMalloc((void**)&pl->d_cmlive, (size_t)d.C1 * ((size_t)N1 * (N2 / TC + 2) + 1) * F.C2A * sizeof(unsigned)) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            fk_tf_release(pl);
            static bool told = false;
            if (!told && getenv("D4W_VERBOSE")) fprintf(stderr, "das4whales_amd: %zu bytes for the time-first f-k workspace not available, channel-first order kept\n", need * 12);
            told = true;
            return D4W_OK;
        }
        pl->cap_W = need;
    }
    D4W_HIP(hipMemcpyAsync(pl->d_tgain, tgain.data(), (size_t)M * sizeof(float), hipMemcpyHostToDevice, st));
    D4W_HIP(hipMemcpyAsync(pl->d_ctab, ctab.data(), ctab.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    D4W_HIP(hipMemcpyAsync(pl->d_colsrc, colsrc.data(), (size_t)Lc * sizeof(int), hipMemcpyHostToDevice, st));
    D4W_HIP(hipMemsetAsync(pl->d_W, 0, need * sizeof(float2), st));        // padding columns stay finite
    if (d.nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", d.nx);
    D4W_LAUNCH(fk_gather_cmask, dim3(std::min(ceil_div(Lc, kThreads), 64), d.nx), dim3(kThreads), 0, stream, d,
               (const float*)pl->d_mask, (const float*)pl->d_nyq, (const int*)pl->d_colsrc, Lc, pl->d_cmask);
    {
        const int nb1 = bw / TC, ntCm = (N1 * nb1 + (nyq_live ? 1 : 0)) * d.C1;
        if (ntCm > 0 && !pl->dev.bs_L)
            D4W_LAUNCH(fk_cm_livebits, dim3(ceil_div(ntCm * F.C2A, kThreads)), dim3(kThreads), 0, stream, (const float*)pl->d_cmask, Lc,
                       d.C2, F.C2A, F.C2B, TC, N1, nb1, RW, nyq_live ? N1 * RW : -1, ntCm, pl->d_cmlive);
    }
    D4W_HIP(hipStreamSynchronize(st));          // the host tables above are temporaries
    FkTfDev& T = pl->tfdev;
    T.tgain = pl->d_tgain; T.ctab = pl->d_ctab; T.cmask = pl->d_cmask; T.W = pl->d_W; T.cmlive = pl->d_cmlive;
    T.RW = RW; T.bw = bw;
    T.col_nyq = nyq_live ? N1 * RW : -1;
    T.Lc = Lc;
    pl->tf = true;
    return D4W_OK;
}

// after a fold (d_mask, d_nyq, d_rowmax written on `stream`): liveness of the wavenumber rows, order of the passes
static int fk_mask_finish(d4w_fk_plan* pl, double prune_eps, void* stream) {
    const FkDims& d = pl->dev.d;
    hipStream_t st = (hipStream_t)stream;
    pl->has_mask = true;
    pl->prune_eps = prune_eps;
    pl->npairs_run = pl->npairs;
    pl->live_rows = d.nx;
    pl->fdev.pairs = pl->dev.pairs;
    pl->fdev.live = nullptr;
    pl->tf = false;
    const char* np = getenv("D4W_FK_NOPRUNE");
    if (pl->fast && !(np && atoi(np) > 0)) {
        // Dead rows (not with a Bluestein pass C, which has no row skipping: only the pass order is chosen there): a wavenumber row whose folded gains (and whose Hermitian partner's) are all zero -- or below the
        // zero-gain threshold above: rounding level always, prune_eps * max |M_h| opt-in (the Butterworth tails of
        // hybrid_ninf_filter_design, dsp.py:348-349, never reach zero: 7.7e-7 at fmax + 14 Hz; treating them as zero
        // changes the output by at most that gain times the spectral content the input holds there)
        std::vector<unsigned> rmax(d.nx);
        D4W_HIP(hipMemcpyAsync(rmax.data(), pl->d_rowmax, (size_t)d.nx * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        D4W_HIP(hipStreamSynchronize(st));
        float gmax = 0.f;
        std::vector<float> rm(d.nx);
        for (int r = 0; r < d.nx; ++r) {
            memcpy(&rm[r], &rmax[r], sizeof(float));
            if (rm[r] == rm[r] && !std::isinf(rm[r]) && rm[r] > gmax) gmax = rm[r];
        }
        const double zero_gain = fk_zero_gain(d, prune_eps, gmax);
        const float thr = (float)zero_gain;
        std::vector<int> live(d.nx);
        for (int r = 0; r < d.nx; ++r) live[r] = !(rm[r] <= thr);             // NaN rows stay alive
        const int N1 = d.N1;
        std::vector<char> lv(d.nx, 0);
        std::vector<int2> run;
        run.reserve(pl->h_pairs.size());
        for (const int2& pr : pl->h_pairs) {
            const int ra = pr.x / N1, rb = pr.y / N1;
            if (live[ra] || live[rb]) {
                lv[ra] = lv[rb] = 1;
                run.push_back(pr);
            }
        }
        int nlive = 0;
        for (int r = 0; r < d.nx; ++r) nlive += lv[r];
        if (nlive < d.nx && !pl->dev.bs_L) {
            const int RA = pl->fast->C2A, RB = pl->fast->C2B;
            std::vector<unsigned> bits((size_t)d.C1 * RA, 0u);
            for (int r = 0; r < d.nx; ++r)
                if (lv[r]) {
                    const int q = r / d.C2, p2 = r % d.C2;
                    bits[(size_t)q * RA + p2 / RB] |= 1u << (p2 % RB);
                }
            D4W_HIP(hipMemcpyAsync(pl->d_livebits, bits.data(), bits.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
            if (!run.empty())
                D4W_HIP(hipMemcpyAsync(pl->d_pairs_live, run.data(), run.size() * sizeof(int2), hipMemcpyHostToDevice, st));
            D4W_HIP(hipStreamSynchronize(st));      // bits / run are host temporaries
            pl->fdev.live = pl->d_livebits;
            pl->fdev.pairs = pl->d_pairs_live;
            pl->npairs_run = (int)run.size();
            pl->live_rows = nlive;
        }
        if (pl->fast->Bt_fwd && pl->d_colkeys) {
            int rc = fk_tf_setup(pl, zero_gain, stream);
            if (rc != D4W_OK) return rc;
        }
    }
    return D4W_OK;
}

/* Order and modelled traffic of the passes for the current mask: info[0] = 1 time-first / 0 channel-first,
 * info[1] = band columns kept per row (incl. the Nyquist column), info[2] = tail columns, info[3] = ns / 2,
 * info[4] = live wavenumber rows, info[5] = nx; bytes[0], bytes[1] = modelled bytes per channel-sample of the
 * channel-first and of the time-first order. */
int d4w_fk_plan_order(const d4w_fk_plan* pl, int* info6, double* bytes2) {
    if (!pl || !info6 || !bytes2) return fail(D4W_EINVAL, "NULL argument");
    info6[0] = pl->tf ? 1 : 0; info6[1] = pl->tf_band_cols; info6[2] = pl->tf_tail_cols; info6[3] = pl->dev.d.M;
    info6[4] = pl->live_rows; info6[5] = pl->dev.d.nx;
    bytes2[0] = pl->bytes_cf; bytes2[1] = pl->bytes_tf;
    return D4W_OK;
}

int d4w_fk_set_mask_dense_f32(d4w_fk_plan* pl, const float* mask_shifted, void* stream) {
    return fk_set_mask_impl(pl, mask_shifted, 0.0, stream);
}

int d4w_fk_set_mask_dense_pruned_f32(d4w_fk_plan* pl, const float* mask_shifted, double prune_eps, void* stream) {
    return fk_set_mask_impl(pl, mask_shifted, prune_eps, stream);
}

int d4w_fk_set_mask_dense_affine_f32(d4w_fk_plan* pl, const float* mask_shifted, float scale, float offset, void* stream) {
    return fk_set_mask_impl(pl, mask_shifted, 0.0, stream, FkAffine{scale, offset, 1});
}

static int fk_set_mask_design_run(d4w_fk_plan* pl, int mode, double k_spacing, double t_spacing, const double* params8_host,
                                  int i0, int i1, const double* hrow_dev, double prune_eps, void* stream);

int d4w_fk_set_mask_design_f32(d4w_fk_plan* pl, int mode, double k_spacing, double t_spacing, const double* params8_host,
                               int i0, int i1, const double* hrow_dev, double prune_eps, void* stream) {
    if (!pl) return fail(D4W_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(pl->apply_mu);
    int rc = fk_plan_enter(pl, stream);
    if (rc != D4W_OK) return rc;
    rc = fk_set_mask_design_run(pl, mode, k_spacing, t_spacing, params8_host, i0, i1, hrow_dev, prune_eps, stream);
    const int rl = fk_plan_leave(pl, stream);
    return rc != D4W_OK ? rc : rl;
}

static int fk_set_mask_design_run(d4w_fk_plan* pl, int mode, double k_spacing, double t_spacing, const double* params8_host,
                                  int i0, int i1, const double* hrow_dev, double prune_eps, void* stream) {
    if (pl && pl->big) {
        if (!(prune_eps >= 0.0 && prune_eps < 1.0)) return fail(D4W_EINVAL, "prune_eps = %g not in [0, 1)", prune_eps);
        fkd_set_prune(pl->big, prune_eps);
        int rc = d4w_fkd_set_mask_design_f32(pl->big, mode, k_spacing, t_spacing, params8_host, i0, i1, hrow_dev, stream);
        if (rc == D4W_OK) pl->has_mask = true;
        return rc;
    }
    if (!pl || !params8_host || !pl->d_mask) return fail(D4W_EINVAL, "NULL argument");
    if (mode < 0 || mode > 2)
        return fail(D4W_EINVAL, "mode %d has no closed form (the Gaussian-blurred designs go through d4w_design_mask_f32)", mode);
    if (mode == 2 && !hrow_dev) return fail(D4W_EINVAL, "hybrid_ninf needs the |H|^2 row");
    if (!(prune_eps >= 0.0 && prune_eps < 1.0)) return fail(D4W_EINVAL, "prune_eps = %g not in [0, 1)", prune_eps);
    const FkDims& d = pl->dev.d;
    if (d.nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", d.nx);
    D4W_HIP(hipMemsetAsync(pl->d_rowmax, 0, (size_t)d.nx * sizeof(unsigned), (hipStream_t)stream));
    const DesignArgs A = make_design_args(d.nx, d.ns, k_spacing, t_spacing, params8_host, i0, i1, hrow_dev);
    dim3 grid(std::min(ceil_div(d.M, kThreads), 64), d.nx);
    D4W_LAUNCH(fk_fold_design, grid, dim3(kThreads), 0, stream, d, A, mode, (const int*)pl->d_rowk, (const int*)pl->d_k1,
               (const int*)pl->d_k2, pl->d_mask, pl->d_nyq, pl->d_rowmax);
    return fk_mask_finish(pl, prune_eps, stream);
}

int d4w_fk_plan_live_rows(const d4w_fk_plan* pl) { return pl ? pl->live_rows : -1; }

// Where the row statistics of a filtered block come from.  The last pass's epilogue pays where a pass-A tile is large: at
// 20 000 x 120 000 (C1 N1 = 625 points per strip column) it adds 0.2 ms to a 3.7-ms pass against 1.7 ms for a separate sweep
// of y.  At the 60-s file shapes (C1 N1 = 100) the same epilogue turned a 215-us pass into 390 us (500 us with shorter runs:
// the per-run reduction is what costs, profiles/r05d/stream_kernels.txt) against 137 us for d4w_row_stats_f32: small tiles
// on large blocks take the sweep.  Plans without the specialised kernels always sweep.
static bool fk_stats_in_epilogue(const d4w_fk_plan* pl) {
    if (!pl->fast) return false;
    const FkDims& d = pl->dev.d;
    static const int epi_env = [] { const char* v = getenv("D4W_FK_STATS_EPILOGUE"); return v ? atoi(v) : -1; }();
    return epi_env >= 0 ? epi_env > 0 : (d.C1 * d.N1 >= 256 || (long long)d.nx * d.ns < (1ll << 22));
}

static int fk_apply_run(d4w_fk_plan* pl, const float* x, float* y, int taper, void* stream, hipEvent_t* ev, double* row_mean,
                        float* row_maxabs);

static int fk_apply_impl(d4w_fk_plan* pl, const float* x, float* y, int taper, void* stream, hipEvent_t* ev,
                         double* row_mean = nullptr, float* row_maxabs = nullptr) {
    if (!pl || !x || !y) return fail(D4W_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(pl->apply_mu);
    int rc = fk_plan_enter(pl, stream);
    if (rc != D4W_OK) return rc;
    rc = fk_apply_run(pl, x, y, taper, stream, ev, row_mean, row_maxabs);
    const int rl = fk_plan_leave(pl, stream);
    return rc != D4W_OK ? rc : rl;
}

static int fk_apply_run(d4w_fk_plan* pl, const float* x, float* y, int taper, void* stream, hipEvent_t* ev, double* row_mean,
                        float* row_maxabs) {
    if (!pl || !x || !y) return fail(D4W_EINVAL, "NULL argument");
    if ((row_mean == nullptr) != (row_maxabs == nullptr)) return fail(D4W_EINVAL, "row_mean and row_maxabs go together");
    if (!pl->has_mask) return fail(D4W_EINVAL, "no mask set on this plan");
    if (pl->big) {
        // time phase of all rows into y (as the packed half spectrum), channel phase on it in place, time phase back
        hipStream_t sb = (hipStream_t)stream;
        int rc;
        if (ev) D4W_HIP(hipEventRecord(ev[0], sb));
        if ((rc = d4w_fkd_time_fwd_f32(pl->big, x, y, taper, stream))) return rc;
        if (ev) { D4W_HIP(hipEventRecord(ev[1], sb)); D4W_HIP(hipEventRecord(ev[2], sb)); }
        if ((rc = d4w_fkd_chan_apply_f32(pl->big, y, stream))) return rc;
        if (ev) { D4W_HIP(hipEventRecord(ev[3], sb)); D4W_HIP(hipEventRecord(ev[4], sb)); }
        if ((rc = d4w_fkd_time_inv_f32(pl->big, y, stream))) return rc;
        if (row_mean && (rc = d4w_row_stats_f32(y, pl->dev.d.nx, pl->dev.d.ns, row_mean, row_maxabs, stream))) return rc;
        if (ev) D4W_HIP(hipEventRecord(ev[5], sb));
        return D4W_OK;
    }
    const FkDev& P = pl->dev;
    const FkDims& d = P.d;
    const float2* src = reinterpret_cast<const float2*>(x);
    float2* dst = reinterpret_cast<float2*>(y);
    const dim3 blk(pl->threads);
    const int ntA = ceil_div(d.N2, d.TA) * d.C2;
    const int ntC = ceil_div(d.M, d.TC) * d.C1;
    const int ntB = pl->npairs;
    const int persist = pl->num_cu * pl->wg_per_cu;
    const dim3 gridA(std::min(ntA, persist)), gridC(std::min(ntC, persist)), gridB(std::min(ntB, persist));
    hipStream_t st = (hipStream_t)stream;
    if (pl->fast && row_mean && !fk_stats_in_epilogue(pl)) {
        int rc = fk_apply_run(pl, x, y, taper, stream, ev, nullptr, nullptr);
        if (rc) return rc;
        return d4w_row_stats_f32(y, d.nx, d.ns, row_mean, row_maxabs, stream);
    }
    if (pl->fast) {
        const FkFastEntry& F = *pl->fast;
        const int NBX = d.N2 / d.TA, NBC = d.N2 / d.TC;
        const int fA = NBX * d.C2, fC = (d.M / d.TC) * d.C1;
        const dim3 gA(std::min(fA, pl->num_cu * pl->wgA)), gC(std::min(fC, pl->num_cu * pl->wgC)),
            gB(std::max(1, std::min(pl->npairs_run, pl->num_cu * pl->wgB)));
        int rc;
#define D4W_MARK(i) do { if (ev) D4W_HIP(hipEventRecord(ev[i], st)); } while (0)
        // row statistics in the epilogue of the last pass: tiles walked in runs of `run` (a divisor of the tiles
        // per c2 of the pass / slab, so that a run stays on the same C1 rows)
        auto stats_run = [](int per_c2, long long) {
            static const int run_env = [] { const char* v = getenv("D4W_FK_RUN_A"); return v ? atoi(v) : 30; }();
            int run = 1;
            for (int r = 1; r <= per_c2 && r <= std::max(run_env, 1); ++r) if (per_c2 % r == 0) run = r;
            return run;
        };
        if (row_mean) {
            D4W_HIP(hipMemsetAsync(row_mean, 0, (size_t)d.nx * sizeof(double), st));
            D4W_HIP(hipMemsetAsync(row_maxabs, 0, (size_t)d.nx * sizeof(float), st));
        }
        const int sw = pl->slab_sw;
        if (sw > 0 && d.TA == d.TC && NBX % sw == 0 && sw < NBX) {
            // Slab order: passes A and C run back to back on one slab of sw column blocks of every n1
            // sub-row (all channels x N1 sw TA columns), slab after slab, so that what pass A wrote is still in the
            // 256 MiB Infinity Cache when pass C reads it -- and the same for C' -> A'.  Same kernels, same tiles,
            // same arithmetic as the plain order; only the order of the tiles differs.
            const int nslab = NBX / sw, nA = d.C2 * sw, nC = d.C1 * d.N1 * sw;
            const dim3 gsA(std::min(nA, pl->num_cu * pl->wgA)), gsC(std::min(nC, pl->num_cu * pl->wgC));
            D4W_MARK(0);
            for (int sl = 0; sl < nslab; ++sl) {
                if ((rc = launch_k(taper ? F.A_fwd_taper : F.A_fwd, gsA, dim3(F.thrA), F.ldsA, stream, P, src, dst, 0, nA, sw, sl * sw, FkGeo()))) return rc;
                if ((rc = launch_k(F.C_fwd, gsC, dim3(F.thrC), F.ldsC, stream, P, pl->fdev, dst, 0, nC, sw, sl * sw, FkGeo()))) return rc;
            }
            D4W_MARK(1);
            D4W_MARK(2);
            if ((rc = launch_k(F.B_mid, gB, dim3(F.thrB), F.ldsB, stream, P, pl->fdev, dst, 0, pl->npairs_run, FkGeo()))) return rc;
            D4W_MARK(3);
            const int run = stats_run(sw, nA), nruns = nA / run;
            for (int sl = 0; sl < nslab; ++sl) {
                if ((rc = launch_k(F.C_inv, gsC, dim3(F.thrC), F.ldsC, stream, P, pl->fdev, dst, 0, nC, sw, sl * sw, FkGeo()))) return rc;
                if (row_mean) {
                    if ((rc = launch_k(F.A_inv_stats, dim3(std::min(nruns, pl->num_cu * pl->wgA)), dim3(F.thrA), F.ldsA, stream, P,
                                       dst, run, nruns, row_mean, (unsigned*)row_maxabs, sw, sl * sw, FkGeo(), (const float2*)nullptr, 0))) return rc;
                } else if ((rc = launch_k(F.A_inv, gsA, dim3(F.thrA), F.ldsA, stream, P, dst, 0, nA, sw, sl * sw, FkGeo(), (const float2*)nullptr))) return rc;
            }
            D4W_MARK(4);
            D4W_MARK(5);
            return D4W_OK;
        }
        if (pl->tf) {
            // time-first order (fk_tf.h): A, Bf (-> W), Cm (on W), Bi (W ->), A'
            const FkTfDev& T = pl->tfdev;
            const int ntCm = (d.N1 * (T.bw / d.TC) + (T.col_nyq >= 0 ? 1 : 0)) * d.C1;
            const dim3 gCm(std::max(1, std::min(ntCm, pl->num_cu * pl->wgC)));
            D4W_MARK(0);
            if ((rc = launch_k(taper ? F.A_fwd_taper : F.A_fwd, gA, dim3(F.thrA), F.ldsA, stream, P, src, dst, 0, fA, NBX, 0, FkGeo()))) return rc;
            D4W_MARK(1);
            const dim3 gBf(std::max(1, std::min(pl->npairsT, pl->num_cu * pl->wgBf)));
            if ((rc = launch_k(F.Bt_fwd, gBf, dim3(F.thrB), F.ldsBt, stream, P, pl->fdev, T, dst, 0, pl->npairsT))) return rc;
            D4W_MARK(2);
            if (ntCm > 0) {
                if (P.bs_L) rc = launch_k(fk_passCm_bluestein, dim3(std::max(1, std::min(ntCm, persist))), blk, pl->ldsC, stream, P, T, ntCm);
                else rc = launch_k(F.C_mid, gCm, dim3(F.thrC), F.ldsC, stream, P, pl->fdev, T, 0, ntCm);
                if (rc) return rc;
            }
            D4W_MARK(3);
            const dim3 gBi(std::max(1, std::min(pl->npairsT, pl->num_cu * pl->wgBi)));
            if ((rc = launch_k(F.Bt_inv, gBi, dim3(F.thrB), F.ldsBt, stream, P, pl->fdev, T, dst, 0, pl->npairsT))) return rc;
            D4W_MARK(4);
            if (row_mean) {
                const int run = stats_run(NBX, fA), nruns = fA / run;
                if ((rc = launch_k(F.A_inv_stats, dim3(std::min(nruns, pl->num_cu * pl->wgA)), dim3(F.thrA), F.ldsA, stream, P,
                                   dst, run, nruns, row_mean, (unsigned*)row_maxabs, NBX, 0, FkGeo(), (const float2*)nullptr, 0))) return rc;
            } else if ((rc = launch_k(F.A_inv, gA, dim3(F.thrA), F.ldsA, stream, P, dst, 0, fA, NBX, 0, FkGeo(), (const float2*)nullptr))) return rc;
            D4W_MARK(5);
            return D4W_OK;
        }
        D4W_MARK(0);
        if ((rc = launch_k(taper ? F.A_fwd_taper : F.A_fwd, gA, dim3(F.thrA), F.ldsA, stream, P, src, dst, 0, fA, NBX, 0, FkGeo()))) return rc;
        D4W_MARK(1);
        if (P.bs_L) rc = launch_k(fk_passC_bluestein<false>, gridC, blk, pl->ldsC, stream, P, dst, ntC);      // C2X > 1
        else rc = launch_k(F.C_fwd, gC, dim3(F.thrC), F.ldsC, stream, P, pl->fdev, dst, 0, fC, NBC, 0, FkGeo());
        if (rc) return rc;
        D4W_MARK(2);
        if ((rc = launch_k(F.B_mid, gB, dim3(F.thrB), F.ldsB, stream, P, pl->fdev, dst, 0, pl->npairs_run, FkGeo()))) return rc;
        D4W_MARK(3);
        if (P.bs_L) rc = launch_k(fk_passC_bluestein<true>, gridC, blk, pl->ldsC, stream, P, dst, ntC);
        else rc = launch_k(F.C_inv, gC, dim3(F.thrC), F.ldsC, stream, P, pl->fdev, dst, 0, fC, NBC, 0, FkGeo());
        if (rc) return rc;
        D4W_MARK(4);
        if (row_mean) {
            const int run = stats_run(NBX, fA), nruns = fA / run;
            if ((rc = launch_k(F.A_inv_stats, dim3(std::min(nruns, pl->num_cu * pl->wgA)), dim3(F.thrA), F.ldsA, stream, P,
                               dst, run, nruns, row_mean, (unsigned*)row_maxabs, NBX, 0, FkGeo(), (const float2*)nullptr, 0))) return rc;
        } else if ((rc = launch_k(F.A_inv, gA, dim3(F.thrA), F.ldsA, stream, P, dst, 0, fA, NBX, 0, FkGeo(), (const float2*)nullptr))) return rc;
        D4W_MARK(5);
#undef D4W_MARK
        return D4W_OK;
    }
    // FAST kernels carry only the unrolled radices; GENERIC ones add the loop-based primes
    const bool gA = pl->genericA, gB = pl->genericB, gC = pl->genericC;
    int rc;
#define D4W_MARK(i) do { if (ev) D4W_HIP(hipEventRecord(ev[i], st)); } while (0)
    D4W_MARK(0);
    if (taper)
        rc = launch_k(gA ? fk_passA_fwd<true, true> : fk_passA_fwd<true, false>, gridA, blk, pl->ldsA, stream, P, src, dst, ntA);
    else
        rc = launch_k(gA ? fk_passA_fwd<false, true> : fk_passA_fwd<false, false>, gridA, blk, pl->ldsA, stream, P, src, dst, ntA);
    if (rc) return rc;
    D4W_MARK(1);
    if (P.bs_L) rc = launch_k(fk_passC_bluestein<false>, gridC, blk, pl->ldsC, stream, P, dst, ntC);
    else rc = launch_k(gC ? fk_passC<false, true> : fk_passC<false, false>, gridC, blk, pl->ldsC, stream, P, dst, ntC);
    if (rc) return rc;
    D4W_MARK(2);
    if (P.bn_L) rc = launch_k(fk_passB_bluestein, gridB, blk, pl->ldsB, stream, P, dst, ntB);
    else rc = launch_k(gB ? fk_passB<true> : fk_passB<false>, gridB, blk, pl->ldsB, stream, P, dst, ntB);
    if (rc) return rc;
    D4W_MARK(3);
    if (P.bs_L) rc = launch_k(fk_passC_bluestein<true>, gridC, blk, pl->ldsC, stream, P, dst, ntC);
    else rc = launch_k(gC ? fk_passC<true, true> : fk_passC<true, false>, gridC, blk, pl->ldsC, stream, P, dst, ntC);
    if (rc) return rc;
    D4W_MARK(4);
    if ((rc = launch_k(gA ? fk_passA_inv<true> : fk_passA_inv<false>, gridA, blk, pl->ldsA, stream, P, dst, ntA))) return rc;
    if (row_mean && (rc = d4w_row_stats_f32(y, d.nx, d.ns, row_mean, row_maxabs, stream))) return rc;   // generic kernels: separate pass
    D4W_MARK(5);
#undef D4W_MARK
    return D4W_OK;
}

int d4w_fk_apply_f32(d4w_fk_plan* pl, const float* x, float* y, int taper, void* stream) {
    return fk_apply_impl(pl, x, y, taper, stream, nullptr);
}

int d4w_fk_stats_in_epilogue(const d4w_fk_plan* pl) { return pl && fk_stats_in_epilogue(pl) ? 1 : 0; }

int d4w_fk_apply_stats_f32(d4w_fk_plan* pl, const float* x, float* y, int taper, double* row_mean, float* row_maxabs,
                           void* stream) {
    if (!row_mean || !row_maxabs) return fail(D4W_EINVAL, "NULL argument");
    return fk_apply_impl(pl, x, y, taper, stream, nullptr, row_mean, row_maxabs);
}

int d4w_fk_apply_timed_stats_f32(d4w_fk_plan* pl, const float* x, float* y, int taper, double* row_mean,
                                 float* row_maxabs, void* stream, float* ms5);

int d4w_fk_apply_timed_f32(d4w_fk_plan* pl, const float* x, float* y, int taper, void* stream, float* ms5) {
    return d4w_fk_apply_timed_stats_f32(pl, x, y, taper, nullptr, nullptr, stream, ms5);
}

int d4w_fk_apply_timed_stats_f32(d4w_fk_plan* pl, const float* x, float* y, int taper, double* row_mean,
                                 float* row_maxabs, void* stream, float* ms5) {
    if (!ms5) return fail(D4W_EINVAL, "NULL argument");
    hipEvent_t ev[6];
    for (int i = 0; i < 6; ++i) D4W_HIP(hipEventCreate(&ev[i]));
    int rc = fk_apply_impl(pl, x, y, taper, stream, ev, row_mean, row_maxabs);
    if (rc == D4W_OK) {
        hipError_t e = hipEventSynchronize(ev[5]);
        if (e != hipSuccess) rc = fail(D4W_EHIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        for (int i = 0; i < 5 && rc == D4W_OK; ++i) {
            e = hipEventElapsedTime(&ms5[i], ev[i], ev[i + 1]);
            if (e != hipSuccess) rc = fail(D4W_EHIP, "hipEventElapsedTime: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

/* measurement aid (not in d4w.h): run ONE pass of a shape-specialised plan over the tile range
 * [t_begin, t_end) in place on `data`.  pass: 0 A, 1 C, 2 B, 3 C', 4 A'. */
int d4w_fk_debug_run_pass(d4w_fk_plan* pl, float* data, int pass, int t_begin, int t_end, void* stream) {
    if (!pl || !data || !pl->fast) return fail(D4W_EINVAL, "needs a shape-specialised plan");
    const FkFastEntry& F = *pl->fast;
    const FkDev& P = pl->dev;
    float2* d2 = reinterpret_cast<float2*>(data);
    if (pass == 2) t_end = std::min(t_end, pl->npairs_run);
    const int n = t_end - t_begin;
    if (n <= 0) return D4W_OK;
    switch (pass) {
        case 0: return launch_k(F.A_fwd, dim3(std::min(n, pl->num_cu * pl->wgA)), dim3(F.thrA), F.ldsA, stream, P, (const float2*)d2, d2, t_begin, t_end, P.d.N2 / P.d.TA, 0, FkGeo());
        case 1: return launch_k(F.C_fwd, dim3(std::min(n, pl->num_cu * pl->wgC)), dim3(F.thrC), F.ldsC, stream, P, pl->fdev, d2, t_begin, t_end, P.d.N2 / P.d.TC, 0, FkGeo());
        case 2: return launch_k(F.B_mid, dim3(std::min(n, pl->num_cu * pl->wgB)), dim3(F.thrB), F.ldsB, stream, P, pl->fdev, d2, t_begin, t_end, FkGeo());
        case 3: return launch_k(F.C_inv, dim3(std::min(n, pl->num_cu * pl->wgC)), dim3(F.thrC), F.ldsC, stream, P, pl->fdev, d2, t_begin, t_end, P.d.N2 / P.d.TC, 0, FkGeo());
        case 4: return launch_k(F.A_inv, dim3(std::min(n, pl->num_cu * pl->wgA)), dim3(F.thrA), F.ldsA, stream, P, d2, t_begin, t_end, P.d.N2 / P.d.TA, 0, FkGeo(), (const float2*)nullptr);
    }
    return fail(D4W_EINVAL, "pass %d", pass);
}

int d4w_taper_f32(float* x, int nx, int ns, void* stream) {
    if (!x || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    // the window's ramps, per (device, ns), uploaded once and kept (a few KB): no allocation, no host synchronisation per call
    struct Ramps { float* dev; int W; };
    static std::mutex mu;
    static std::map<std::pair<int, int>, Ramps> cache;
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    Ramps r{nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find({devid, ns});
        if (it == cache.end()) {
            const std::vector<float> win = tukey_window(ns, 0.03);
            int W = 0;                                             // columns at each end where the window is not 1 (it is symmetric)
            for (int i = 0; i < ns; ++i)
                if (win[i] != 1.0f || win[ns - 1 - i] != 1.0f) W = std::max(W, std::min(i, ns - 1 - i) + 1);
            W = std::min(W, ns / 2);
            std::vector<float> ramp((size_t)2 * std::max(W, 1), 1.0f);
            for (int j = 0; j < W; ++j) { ramp[j] = win[j]; ramp[W + j] = win[ns - W + j]; }
            if (ns % 2 && W == ns / 2 && win[ns / 2] != 1.0f)
                return fail(D4W_EINVAL, "taper of %d samples has no flat part", ns);       // (alpha = 0.03: never)
            float* d = nullptr;
            D4W_HIP(hipMalloc((void**)&d, ramp.size() * sizeof(float)));
            hipError_t e = hipMemcpy(d, ramp.data(), ramp.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(d); return fail(D4W_EHIP, "tukey upload failed: %s", hipGetErrorString(e)); }
            it = cache.emplace(std::make_pair(devid, ns), Ramps{d, W}).first;
        }
        r = it->second;
    }
    if (r.W == 0) return D4W_OK;                                   // the window is 1 everywhere (ns <= 1)
    const size_t total = (size_t)nx * 2 * r.W;
    const int blocks = (int)std::min<size_t>((total + kThreads - 1) / kThreads, 4096);
    D4W_LAUNCH(taper_ramps, dim3(blocks), dim3(kThreads), 0, stream, x, (const float*)r.dev, nx, ns, r.W);
    return D4W_OK;
}

}  // extern "C"

// =============================================================================================
// Distributed f-k filter (one channel block per GPU): pencil decomposition with ONE exchange each
// way.  The packed 2-D transform factorises per axis, so the five passes regroup as
//   time phase   (local rows)        : n1 sub-FFT + four-step twiddle (pass A with C1 = 1), then the
//                                      contiguous n2 sub-FFT of every sub-row            [fkd_rows_n2]
//   all-to-all                       : re-shard by n1-position q1 (sub-rows of N2 bins); a q1 and its
//                                      Hermitian partner q1' always travel to the same rank
//   channel phase (all nx channels   : c1 sub-FFT + twiddle (pass A with N1 = 1), c2 sub-FFT (pass C),
//                  of the owned q1s)   real-spectrum pair op x folded mask                [fkd_pair_slab],
//                                      inverse c2, inverse c1 (x 1/(nx M))
//   all-to-all back, inverse time phase.
// The generic pass kernels are reused unchanged through two FkDev descriptors with a degenerate
// axis each.  Host plumbing (pack, RCCL all_to_all_single, unpack) lives in das4whales_amd/shard.py.
// =============================================================================================
namespace d4w {

// contiguous n2 sub-FFT of every sub-row (forward: natural -> digit-reversed; inverse: back)
template <bool INV, bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_rows_n2(FkDev P, float2* __restrict__ data, int nsub) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N2 = P.d.N2;
    const TwLds tw = tw_stage(P.ax_n2, tile + N2, tid, nthr);
    for (int t = blockIdx.x; t < nsub; t += gridDim.x) {
        float2* row = data + (size_t)t * N2;
        for (int w = tid; w < N2; w += nthr) tile[w] = row[w];
        lds_barrier();
        lds_fft<INV, false, GENERIC>(tile, P.ax_n2, tw, 1, 1, 0, 1, 0, tid, nthr);
        for (int w = tid; w < N2; w += nthr) row[w] = tile[w];
        lds_barrier();
    }
}

struct FkdSlab {
    int nx, nq, N2;           // slab [nx row positions][nq owned q1][N2 positions]
    const int* q1_of;         // [nq] n1-position of local column block jq
    const int* jq_partner;    // [nq] local index of the Hermitian partner q1
    const int* row_partner;   // [nx]
    const int* mirror0;       // [N2]
    const float2* wrow;       // [N1] indexed by q1
    const float2* wcol;       // [N2]
    const float* mask;        // [nx][nq][N2] folded mask of the owned sub-rows
    const float* nyq;         // [nx]
    int hilbert;              // 0: x folded mask (f-k filter); 1: x (-i sign(f)), rows self-paired (analytic signal)
};

// pair op of fk_passB on the slab layout (see the algebra there); one thread per Hermitian pair
__global__ __launch_bounds__(kThreads) void fkd_pair_slab(FkdSlab S, float2* __restrict__ slab) {
    const int jq = blockIdx.z, r = blockIdx.y;
    const int rp = S.hilbert ? r : S.row_partner[r], jp = S.jq_partner[jq];
    const long keyA = (long)r * S.nq + jq, keyB = (long)rp * S.nq + jp;
    if (keyB < keyA) return;
    const bool same = (keyA == keyB);
    const bool k1zero = (S.q1_of[jq] == 0);
    float2* A = slab + (size_t)keyA * S.N2;
    float2* B = slab + (size_t)keyB * S.N2;
    const float* mA = S.hilbert ? nullptr : S.mask + (size_t)keyA * S.N2;
    const float* mB = S.hilbert ? nullptr : S.mask + (size_t)keyB * S.N2;
    const float2 wr = S.wrow[S.q1_of[jq]];
    const float nyq = S.hilbert ? 0.f : S.nyq[r];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.N2; i += gridDim.x * blockDim.x) {
        const int j = k1zero ? S.mirror0[i] : (S.N2 - 1 - i);
        if (same && j < i) continue;
        const float2 a = A[i];
        const float2 Bc = c_conj(B[j]);
        const float2 w = c_mul(wr, S.wcol[i]);
        const float2 E = c_scale(c_add(a, Bc), 0.5f);
        const float2 O = c_mul_mi(c_scale(c_sub(a, Bc), 0.5f));
        const float2 tO = c_mul(w, O);
        float2 Yp, Ym;
        if (S.hilbert) {
            // spectrum of the Hilbert transform: -i X(f) for 0 < f < M, +i X(f) for the negative
            // frequencies f + M, zero at DC and Nyquist (scipy's h = 1 there belongs to the real part)
            const bool dc = k1zero && i == 0;
            Yp = dc ? make_float2(0.f, 0.f) : c_mul_mi(c_add(E, tO));
            Ym = dc ? make_float2(0.f, 0.f) : c_mul_pi(c_sub(E, tO));
        } else {
            const float ma = mA[i];
            const float mb = (k1zero && i == 0) ? nyq : mB[j];
            Yp = c_scale(c_add(E, tO), ma);
            Ym = c_scale(c_sub(E, tO), mb);
        }
        const float2 Sm = c_scale(c_add(Yp, Ym), 0.5f);
        const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
        A[i] = c_add(Sm, D);
        B[j] = c_conj(c_sub(Sm, D));
    }
}

// ---------------------------------------------------------------------------------------------
// Channel transform of ANY length (a channel count with a prime factor > 31) as a Bluestein convolution in global
// memory: column by column,
//   a[n] = z[n] w[n] (n < nx, zero up to L = 2^a 3^b 5^c >= 2 nx - 1),  A = FFT_L(a),  A *= FFT_L(conj w, wrapped) / L,
//   c = IFFT_L(A),  Z[k] = w[k] c[k] (k < nx),  w[n] = exp(-i pi n^2 / nx);  the inverse conjugates w and the filter.
// The two length-L transforms are passes A and C of the generic kernels on a scratch [L][Wc] (a chunk of Wc slab columns
// at a time) with the elementwise steps folded into them: three launches per direction, 5 sweeps of the scratch.  Rows come
// out in natural wavenumber order.  Tiles and pipelining as fk_passA_fwd / fk_passC / fk_passA_inv with N1 = 1.
// ---------------------------------------------------------------------------------------------
//   fkd_bz_passA_fwd : slab rows x chirp (zero beyond nx / beyond the chunk's columns) -> c1 transform, x W_L -> scratch
//   fkd_bz_passC     : c2 transform, x filter at the row position, inverse c2 transform, in place on the scratch
//   fkd_bz_passA_inv : x conj W_L, inverse c1 transform, x chirp x scale -> slab rows < nx
struct FkdBz {
    const float2* chirp;      // [nx]
    const float2* filt;       // [L] at row positions
    size_t pitch;             // slab row pitch (complex)
    int nx, ncol, inv;        // rows of the slab, valid columns of this chunk, 1: conjugated tables
    float scale;
    const int* cols;          // NULL: scratch column t = slab column t; else slab column cols[t] (the live columns, below)
};

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bz_passA_fwd(FkDev P, FkdBz Z, const float2* __restrict__ slab,
                                                                float2* __restrict__ S, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.C1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dntx(ntx);
    const TwLds tw_c1 = tw_stage(P.ax_c1, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int c1 = dTA.div(w), tt = w - c1 * TA;
                const int row = c1 * d.C2 + c2, col = b0 + tt;
                if (row < Z.nx && col < Z.ncol) {
                    float2 ch = Z.chirp[row];
                    if (Z.inv) ch.y = -ch.y;
                    v = c_mul(slab[(size_t)row * Z.pitch + (Z.cols ? Z.cols[col] : col)], ch);
                }
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<false, true, GENERIC>(tile, P.ax_c1, tw_c1, TA, TA, 1, 1, 0, tid, nthr);
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
        for (int w = tid; w < nelem; w += nthr) {
            const int q = dTA.div(w), tt = w - q * TA;
            if (tt < ncol) {
                const size_t row = (size_t)q * d.C2 + c2;
                S[row * d.M + b0 + tt] = c_mul(tile[w], P.twc[q * d.C2 + c2]);
            }
        }
        lds_barrier();
        t = next;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bz_passA_inv(FkDev P, FkdBz Z, const float2* __restrict__ S,
                                                                float2* __restrict__ slab, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.C1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dntx(ntx);
    const TwLds tw_c1 = tw_stage(P.ax_c1, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int q = dTA.div(w), tt = w - q * TA;
                if (tt < ncol) v = S[((size_t)q * d.C2 + c2) * d.M + b0 + tt];
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
        const int c2 = dntx.div(t), b0 = (t - c2 * ntx) * TA;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) {
                const int q = dTA.div(w);
                tile[w] = c_mulc(pf[it], P.twc[q * d.C2 + c2]);
            }
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<true, true, GENERIC>(tile, P.ax_c1, tw_c1, TA, TA, 1, 1, 0, tid, nthr);
        for (int w = tid; w < nelem; w += nthr) {
            const int c1 = dTA.div(w), tt = w - c1 * TA;
            const int row = c1 * d.C2 + c2, col = b0 + tt;
            if (row < Z.nx && col < Z.ncol) {
                float2 ch = Z.chirp[row];
                if (Z.inv) ch.y = -ch.y;
                slab[(size_t)row * Z.pitch + (Z.cols ? Z.cols[col] : col)] = c_mul(tile[w], c_scale(ch, Z.scale));
            }
        }
        lds_barrier();
        t = next;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bz_passC(FkDev P, FkdBz Z, float2* __restrict__ S, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TC = d.TC;
    const int nelem = d.C2 * TC;
    const int ntx = (d.M + TC - 1) / TC;
    const FDiv dTC(TC), dntx(ntx);
    const TwLds tw = tw_stage(P.ax_c2, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int q = dntx.div(t), p0 = (t - q * ntx) * TC;
        const int ncol = min(TC, d.M - p0);
        const float2* base = S + ((size_t)q * d.C2) * d.M + p0;
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int c2 = dTC.div(w), tt = w - c2 * TC;
                if (tt < ncol) v = base[(size_t)c2 * d.M + tt];
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<false, true, GENERIC>(tile, P.ax_c2, tw, TC, TC, 1, 1, 0, tid, nthr);
        const int q = dntx.div(t), p0 = (t - q * ntx) * TC;
        for (int w = tid; w < nelem; w += nthr) {
            float2 f = Z.filt[q * d.C2 + dTC.div(w)];
            if (Z.inv) f.y = -f.y;
            tile[w] = c_mul(tile[w], f);
        }
        lds_barrier();
        lds_fft<true, true, GENERIC>(tile, P.ax_c2, tw, TC, TC, 1, 1, 0, tid, nthr);
        const int ncol = min(TC, d.M - p0);
        float2* base = S + ((size_t)q * d.C2) * d.M + p0;
        for (int w = tid; w < nelem; w += nthr) {
            const int c2 = dTC.div(w), tt = w - c2 * TC;
            if (tt < ncol) base[(size_t)c2 * d.M + tt] = tile[w];
        }
        lds_barrier();
        t = next;
    }
}

// The time transform of the packed rows (length M = ns / 2 with a prime factor > 31) the same way, row by row: a chunk
// of rows in a scratch [R][L], L = 2^a 3^b 5^c >= 2 M - 1, and per direction
//   fkd_bt_passA_fwd : row x (window) x chirp, zero beyond M -> n1 transform, x W_L -> scratch   (fk_passA_fwd, C1 = 1)
//   fkd_bt_rows      : n2 transform, x filter at the position, inverse n2 transform             (fkd_rows_n2 twice)
//   fkd_bt_passA_inv : x conj W_L, inverse n1 transform, x chirp x scale -> the row's first M elements
struct FkdBt {
    const float2* chirp;      // [M]
    const float2* filt;       // [L] at positions [q1][i]
    const float2* win;        // [M] packed window, or NULL
    size_t pitch;             // row pitch of the rows outside the scratch (complex) = M
    int m, inv;               // M; 1: conjugated tables
    float scale;
};

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bt_passA_fwd(FkDev P, FkdBt Z, const float2* __restrict__ rows,
                                                                float2* __restrict__ S, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.N1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dntx(ntx);
    const TwLds tw_n1 = tw_stage(P.ax_n1, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int row = dntx.div(t), b0 = (t - row * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int n1 = dTA.div(w), tt = w - n1 * TA;
                const int col = n1 * d.N2 + b0 + tt;
                if (tt < ncol && col < Z.m) {
                    v = rows[(size_t)row * Z.pitch + col];
                    if (Z.win) {
                        const float2 wv = Z.win[col];
                        v.x *= wv.x;
                        v.y *= wv.y;
                    }
                    float2 ch = Z.chirp[col];
                    if (Z.inv) ch.y = -ch.y;
                    v = c_mul(v, ch);
                }
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) tile[w] = pf[it];
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<false, true, GENERIC>(tile, P.ax_n1, tw_n1, TA, TA, 1, 1, d.N1 * TA, tid, nthr);
        const int row = dntx.div(t), b0 = (t - row * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
        for (int w = tid; w < nelem; w += nthr) {
            const int q1 = dTA.div(w), tt = w - q1 * TA;
            if (tt < ncol) S[(size_t)row * d.M + q1 * d.N2 + b0 + tt] = c_mul(tile[w], P.twt[q1 * d.N2 + b0 + tt]);
        }
        lds_barrier();
        t = next;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bt_passA_inv(FkDev P, FkdBt Z, const float2* __restrict__ S,
                                                                float2* __restrict__ rows, int ntiles) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const FkDims& d = P.d;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TA = d.TA;
    const int nelem = d.N1 * TA;
    const int ntx = (d.N2 + TA - 1) / TA;
    const FDiv dTA(TA), dntx(ntx);
    const TwLds tw_n1 = tw_stage(P.ax_n1, tile + nelem, tid, nthr);
    float2 pf[kPF];
    auto issue = [&](int t) {
        const int row = dntx.div(t), b0 = (t - row * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            float2 v = make_float2(0.f, 0.f);
            if (w < nelem) {
                const int q1 = dTA.div(w), tt = w - q1 * TA;
                if (tt < ncol) v = S[(size_t)row * d.M + q1 * d.N2 + b0 + tt];
            }
            pf[it] = v;
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    while (t < ntiles) {
        const int row = dntx.div(t), b0 = (t - row * ntx) * TA;
        const int ncol = min(TA, d.N2 - b0);
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            const int w = tid + it * nthr;
            if (w < nelem) {
                const int q1 = dTA.div(w), tt = w - q1 * TA;
                float2 v = pf[it];
                if (tt < ncol) v = c_mulc(v, P.twt[q1 * d.N2 + b0 + tt]);
                tile[w] = v;
            }
        }
        lds_barrier();
        const int next = t + gridDim.x;
        if (next < ntiles) issue(next);
        lds_fft<true, true, GENERIC>(tile, P.ax_n1, tw_n1, TA, TA, 1, 1, d.N1 * TA, tid, nthr);
        for (int w = tid; w < nelem; w += nthr) {
            const int n1 = dTA.div(w), tt = w - n1 * TA;
            const int col = n1 * d.N2 + b0 + tt;
            if (tt < ncol && col < Z.m) {
                float2 ch = Z.chirp[col];
                if (Z.inv) ch.y = -ch.y;
                rows[(size_t)row * Z.pitch + col] = c_mul(tile[w], c_scale(ch, Z.scale));
            }
        }
        lds_barrier();
        t = next;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(kMaxThreads) void fkd_bt_rows(FkDev P, FkdBt Z, float2* __restrict__ S, int nsub) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N2 = P.d.N2, N1 = P.d.N1;
    const TwLds tw = tw_stage(P.ax_n2, tile + N2, tid, nthr);
    for (int t = blockIdx.x; t < nsub; t += gridDim.x) {
        float2* row = S + (size_t)t * N2;
        const float2* f = Z.filt + (size_t)(t % N1) * N2;
        for (int w = tid; w < N2; w += nthr) tile[w] = row[w];
        lds_barrier();
        lds_fft<false, false, GENERIC>(tile, P.ax_n2, tw, 1, 1, 0, 1, 0, tid, nthr);
        for (int w = tid; w < N2; w += nthr) {
            float2 fv = f[w];
            if (Z.inv) fv.y = -fv.y;
            tile[w] = c_mul(tile[w], fv);
        }
        lds_barrier();
        lds_fft<true, false, GENERIC>(tile, P.ax_n2, tw, 1, 1, 0, 1, 0, tid, nthr);
        for (int w = tid; w < N2; w += nthr) row[w] = tile[w];
        lds_barrier();
    }
}

// folded mask of the owned sub-rows (fk_fold_mask restricted to a q1 subset)
__global__ __launch_bounds__(kThreads) void fkd_fold_mask(int nx, int ns, int N1, int N2, int nq,
                                                           const float* __restrict__ ms,
                                                           const int* __restrict__ rowk,
                                                           const int* __restrict__ q1_of,
                                                           const int* __restrict__ k1_of_q1,
                                                           const int* __restrict__ k2_of_i,
                                                           float* __restrict__ mask, float* __restrict__ nyq, FkAffine aff) {
    const int r = blockIdx.y;
    const int k = rowk[r];
    const int km = (nx - k) % nx;
    const int sx = nx / 2, st = ns / 2, M = ns / 2;
    const size_t rowp = (size_t)((k + sx) % nx) * ns;
    const size_t rowm = (size_t)((km + sx) % nx) * ns;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nq * N2; p += gridDim.x * blockDim.x) {
        const int jq = p / N2, i = p - jq * N2;
        const int f = k1_of_q1[q1_of[jq]] + N1 * k2_of_i[i];
        const int fm = (ns - f) % ns;
        mask[(size_t)r * nq * N2 + p] = 0.5f * (aff(ms[rowp + (f + st) % ns]) + aff(ms[rowm + (fm + st) % ns]));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        nyq[r] = 0.5f * (aff(ms[rowp + (M + st) % ns]) + aff(ms[rowm + ((ns - M) + st) % ns]));
}

// ... evaluated in closed form instead of read (fk_fold_design restricted to a q1 subset): a rank never holds the
// dense mask, only the gains of the sub-rows it owns
__global__ __launch_bounds__(kThreads) void fkd_fold_design(int nx, int ns, int N1, int N2, int nq, DesignArgs A, int mode,
                                                             const int* __restrict__ rowk,
                                                             const int* __restrict__ q1_of,
                                                             const int* __restrict__ k1_of_q1,
                                                             const int* __restrict__ k2_of_i,
                                                             float* __restrict__ mask, float* __restrict__ nyq) {
    const int r = blockIdx.y;
    const int k = rowk[r];
    const int km = (nx - k) % nx;
    const int sx = nx / 2, st = ns / 2, M = ns / 2;
    const int ip = (k + sx) % nx, im = (km + sx) % nx;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nq * N2; p += gridDim.x * blockDim.x) {
        const int jq = p / N2, i = p - jq * N2;
        const int f = k1_of_q1[q1_of[jq]] + N1 * k2_of_i[i];
        const int fm = (ns - f) % ns;
        mask[(size_t)r * nq * N2 + p] = design_folded(A, mode, ip, (f + st) % ns, im, (fm + st) % ns);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        nyq[r] = design_folded(A, mode, ip, (M + st) % ns, im, ((ns - M) + st) % ns);
}

// where a distributed plan's mask comes from: a dense shifted-grid mask, or a closed-form design
struct FkdMaskSrc {
    const float* dense = nullptr;
    DesignArgs A{};
    int mode = -1;
    FkAffine aff{1.f, 0.f, 0};           // dense only, generic plan only: a * m + b applied while folding
};

}  // namespace d4w

struct d4w_fkd_plan {
    int nx = 0, ns = 0, M = 0, world = 1, rank = 0;
    int row_begin = 0, row_end = 0;      // this rank's channel block
    int N1 = 1, N2 = 1, C1 = 1, C2 = 1, TA_t = 1, TA_c = 1, TC = 1;
    std::vector<int> owner;              // [N1] rank owning each n1-position
    std::vector<int> myq;                // owned n1-positions, ascending
    d4w_fk_plan tp, cp;                  // time-phase / channel-phase descriptors (tables owned here)
    FkdSlab slab;
    size_t lds_t = 0, lds_c1 = 0, lds_c2 = 0, lds_n2 = 0;
    bool gen_t = false, gen_n2 = false, gen_c1 = false, gen_c2 = false;
    int* d_rowk = nullptr; int* d_k1 = nullptr; int* d_k2 = nullptr; int* d_q1of = nullptr;
    float* d_mask = nullptr; float* d_nyq = nullptr;
    bool has_mask = false;
    int num_cu = 256;
    // ---- channel count with a prime factor > 31: Bluestein convolution of length bz_L in global memory (fkd_bz_*)
    int bz_L = 0, bz_W = 0;                 // transform length (0: off), columns per scratch chunk
    float2* bz_S = nullptr;                 // [bz_L][bz_W]
    const float2* bz_chirp = nullptr;       // [nx]    exp(-i pi n^2 / nx)
    const float2* bz_filt = nullptr;        // [bz_L]  FFT of the wrapped conjugate chirp / bz_L at the row positions of cp
    // the slab columns whose folded gains (or whose Hermitian partner column's) are not all zero: the only ones whose channel
    // transform matters -- a dead column leaves the pair op as zeros whatever it held, and the inverse transform of zeros is
    // zeros.  A fin-whale band of 14-30 Hz keeps a third of the columns (the band and its mirror image in the packed
    // spectrum): 19 997 x 120 000 runs 6 scratch chunks per direction instead of 18.
    int* bz_cols = nullptr;                 // [W] DEVICE, the first bz_nlive entries in use
    unsigned* bz_flags = nullptr;           // [W] DEVICE scratch of the scan (column maxima, then the live flags)
    int bz_nlive = -1;                      // -1: every column (no list)
    double prune_eps = 0.0;                 // opt-in gain level (x the largest gain) below which a column counts as dead
    std::vector<int> h_jqpart, h_mirror0;   // host copies of the pair op's column maps
    // ---- ns / 2 with a prime factor > 31: the same for the time transform of the packed rows (fkd_bt_*); the half
    //      spectrum then comes out in natural order, N1 = 1, N2 = ns / 2, while tp describes the length-bt_L transform
    int bt_L = 0, bt_R = 0;                 // transform length (0: off), rows per scratch chunk
    float2* bt_S = nullptr;                 // [bt_R][bt_L]
    const float2* bt_chirp = nullptr;       // [M]     exp(-i pi m^2 / M)
    const float2* bt_filt = nullptr;        // [bt_L]  at the positions [q1][i] of tp
    const float2* bt_win = nullptr;         // [M]     packed tukey(ns, 0.03)
    // ---- packed path: the shape has specialised kernels (fk_fast.h, FkGeo).  The exchange buffers need no repacking:
    //      the time phase writes sub-row q1 of local row l at  blk_off[owner[q1]] + l * nq[owner] * N2 + jq(q1) * N2
    //      (destination rank major), so that the all-to-all delivers the slab [nx][nq][N2] as it stands.
    d4w_fk_plan* sp = nullptr;              // tables of the single-device plan of the whole shape (no mask of its own)
    FkDev dev_t, dev_c;                     // time-phase / channel-phase argument blocks
    FkFastDev fdev_c;
    FkGeo geo_t, geo_c;
    std::vector<int2> h_pairs_slab;         // pass-B work list in slab keys r * nq + jq
    int2* d_pairs_slab = nullptr; int2* d_pairs_live = nullptr;
    int2* d_pairs_self = nullptr; int npairs_self = 0;   // pass-B work list "every row is its own Hermitian partner" (analytic signal)
    unsigned* d_livebits = nullptr; unsigned* d_rowmax = nullptr;
    int npairs_run = 0, live_rows = 0;
};

static void fkd_block(int nx, int world, int r, int* a, int* b) {
    const int base = nx / world, extra = nx % world;
    *a = r * base + std::min(r, extra);
    *b = *a + base + (r < extra ? 1 : 0);
}

extern "C" {

int d4w_fkd_plan_destroy(d4w_fkd_plan* pl) {
    if (!pl) return D4W_OK;
    for (void* p : pl->tp.allocs) (void)hipFree(p);
    for (void* p : pl->cp.allocs) (void)hipFree(p);
    if (pl->sp) d4w_fk_plan_destroy(pl->sp);
    delete pl;
    return D4W_OK;
}

namespace d4w {
// bit pattern of max |mask| over the owned sub-rows of every row position (slab mask [nx][W]) and its Nyquist gain
__global__ __launch_bounds__(kThreads) void fkd_row_max(const float* __restrict__ mask, const float* __restrict__ nyq, int W,
                                                         unsigned* __restrict__ rowmaxbits) {
    const int r = blockIdx.x;
    unsigned vb = (threadIdx.x == 0) ? (__float_as_uint(nyq[r]) & 0x7fffffffu) : 0u;
    for (int p = threadIdx.x; p < W; p += blockDim.x) vb = max(vb, __float_as_uint(mask[(size_t)r * W + p]) & 0x7fffffffu);
    for (int off = 32; off >= 1; off >>= 1) vb = max(vb, __shfl_xor(vb, off));
    if ((threadIdx.x & 63) == 0 && vb) atomicMax(rowmaxbits + r, vb);
}
}  // namespace d4w

// Packed path of the distributed plan: shapes with specialised kernels (fk_fast.h).  Returns D4W_EINVAL (and leaves
// *out NULL) when the shape has none -- the caller then builds the generic plan.
static int fkd_plan_build_packed(int nx, int ns, int world, int rank, bool want_mask, d4w_fkd_plan** out, bool time_only = false) {
    *out = nullptr;
    const char* g = getenv("D4W_FKD_GENERIC");
    if (g && atoi(g) > 0) return fail(D4W_EINVAL, "generic distributed plan requested");
    d4w_fk_plan* sp = nullptr;
    int rc = fk_plan_build(nx, ns, nullptr, false, true, &sp);
    if (rc) return rc;
    const FkFastEntry& F = *sp->fast;
    // (time_only: only the time phase and pass B are used -- the analytic signal of long rows -- and the channel-phase
    // constraints do not apply)
    if (!time_only && (F.C2X > 1 || F.TA != F.TC || (F.NA * F.NB * F.NC) % (F.N1 * F.TA) != 0)) {     // pass A MODE 2 walks N1 adjacent strips inside a sub-row
        d4w_fk_plan_destroy(sp);
        return fail(D4W_EINVAL, "shape config not usable for the packed distributed plan");
    }
    d4w_fkd_plan* pl = new d4w_fkd_plan();
    pl->sp = sp;
    const FkDims& d = sp->dev.d;
    pl->nx = nx; pl->ns = ns; pl->M = d.M; pl->world = world; pl->rank = rank;
    fkd_block(nx, world, rank, &pl->row_begin, &pl->row_end);
    pl->N1 = d.N1; pl->N2 = d.N2; pl->C1 = d.C1; pl->C2 = d.C2; pl->TA_t = d.TA; pl->TA_c = d.TA; pl->TC = d.TC;
    pl->num_cu = sp->num_cu;
    const int N1 = d.N1, N2 = d.N2, nxl = pl->row_end - pl->row_begin;
    // ownership of the n1-positions (single radix: position = frequency digit): Hermitian classes {q1, N1 - q1} dealt
    // to the least loaded rank
    pl->owner.assign(N1, -1);
    {
        std::vector<int> load(world, 0);
        for (int q1 = 0; q1 < N1; ++q1) {
            if (pl->owner[q1] >= 0) continue;
            int best = 0;
            for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
            const int qp = (N1 - q1) % N1;
            pl->owner[q1] = best; ++load[best];
            if (qp != q1) { pl->owner[qp] = best; ++load[best]; }
        }
    }
    std::vector<int> nq_of(world, 0), jq_of(N1, 0);
    for (int q1 = 0; q1 < N1; ++q1) jq_of[q1] = nq_of[pl->owner[q1]]++;
    for (int q1 = 0; q1 < N1; ++q1) if (pl->owner[q1] == rank) pl->myq.push_back(q1);
    const int nq = (int)pl->myq.size();
    const long W = (long)std::max(nq, 1) * N2;
    if ((long)nx * W >= (1L << 31) || (long)nxl * d.M >= (1L << 31)) {
        d4w_fkd_plan_destroy(pl);
        return fail(D4W_EINVAL, "block too large for 32-bit element offsets");
    }
    // packed buffer layout of THIS rank's local rows
    std::vector<int> qoff(N1), qpitch(N1);
    {
        std::vector<long> blk(world + 1, 0);
        for (int sr = 0; sr < world; ++sr) blk[sr + 1] = blk[sr] + (long)nxl * nq_of[sr] * N2;
        for (int q1 = 0; q1 < N1; ++q1) {
            const int o = pl->owner[q1];
            qoff[q1] = (int)(blk[o] + (long)jq_of[q1] * N2);
            qpitch[q1] = nq_of[o] * N2;
        }
    }
    std::vector<int> myq = pl->myq;
    if (myq.empty()) myq.push_back(0);
    const int *c_qoff, *c_qpitch, *c_q1of;
#define D4W_TRY(x) do { rc = (x); if (rc != D4W_OK) { d4w_fkd_plan_destroy(pl); return rc; } } while (0)
    D4W_TRY(upload(sp, qoff, &c_qoff));
    D4W_TRY(upload(sp, qpitch, &c_qpitch));
    D4W_TRY(upload(sp, myq, &c_q1of));
    pl->d_q1of = const_cast<int*>(c_q1of);
    pl->geo_t = FkGeo{nxl, d.M, nq, c_qoff, c_qpitch, c_q1of};
    pl->geo_c = FkGeo{nx, (int)W, (int)(W / d.TC), c_qoff, c_qpitch, c_q1of};     // pass C: nq = column blocks per row
    pl->dev_t = sp->dev;
    pl->dev_t.scale = 1.0f;
    pl->dev_c = sp->dev;
    // pass-B work list in slab keys
    std::vector<int> jq_mine(N1, -1);
    for (int j = 0; j < nq; ++j) jq_mine[pl->myq[j]] = j;
    for (const int2& pr : sp->h_pairs) {
        const int ra = pr.x / N1, qa = pr.x % N1, rb = pr.y / N1, qb = pr.y % N1;
        if (jq_mine[qa] < 0) continue;
        pl->h_pairs_slab.push_back(make_int2(ra * nq + jq_mine[qa], rb * nq + jq_mine[qb]));
    }
    {
        const int2* c_pairs;
        D4W_TRY(upload(sp, pl->h_pairs_slab, &c_pairs));
        pl->d_pairs_slab = const_cast<int2*>(c_pairs);
        void* q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(pl->h_pairs_slab.size(), 1) * sizeof(int2)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        sp->allocs.push_back(q); pl->d_pairs_live = (int2*)q;
        if (hipMalloc(&q, (size_t)d.C1 * F.C2A * sizeof(unsigned)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        sp->allocs.push_back(q); pl->d_livebits = (unsigned*)q;
        if (hipMalloc(&q, (size_t)nx * sizeof(unsigned)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        sp->allocs.push_back(q); pl->d_rowmax = (unsigned*)q;
    }
    pl->fdev_c = sp->fdev;
    pl->fdev_c.pairs = pl->d_pairs_slab;
    pl->fdev_c.live = nullptr;
    pl->npairs_run = (int)pl->h_pairs_slab.size();
    pl->live_rows = nx;
    if (want_mask) {
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)nx * W * sizeof(float)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc of the slab mask failed"); }
        sp->allocs.push_back(q); pl->d_mask = (float*)q;
        pl->d_nyq = sp->d_nyq;
    }
    pl->dev_c.mask = pl->d_mask;
    pl->dev_c.nyq = pl->d_nyq;
#undef D4W_TRY
    *out = pl;
    return D4W_OK;
}

static int fkd_plan_build(int nx, int ns, int world, int rank, bool want_mask, d4w_fkd_plan** out) {
    if (!out) return fail(D4W_EINVAL, "plan pointer is NULL");
    *out = nullptr;
    if (nx < 1 || ns < 2 || (ns & 1)) return fail(D4W_EINVAL, "bad shape %d x %d (ns must be even)", nx, ns);
    if (world < 1 || rank < 0 || rank >= world || world > nx) return fail(D4W_EINVAL, "bad world %d / rank %d", world, rank);
    if (want_mask && fkd_plan_build_packed(nx, ns, world, rank, want_mask, out) == D4W_OK) return D4W_OK;
    const int M = ns / 2;
    // ns / 2 with a prime factor > 31: the time transform of the packed rows is a Bluestein convolution of length bt_L in
    // global memory (fkd_bt_*); tp describes that length and the half spectrum comes out in natural order (N1 = 1)
    const int bt_L = (rough_part(M) > 1) ? smooth_len_235(2L * M - 1) : 0;
    const int Mt = bt_L ? bt_L : M;
    // time axis Mt = tN1 * tN2: tN2 in one LDS row, and enough n1-positions (Hermitian classes) to balance the ranks
    int tN1 = 0;
    for (int cand = 1; cand <= Mt; ++cand) {
        if (Mt % cand) continue;
        if (Mt / cand > kMaxTile / 2) continue;
        if (tN1 == 0) tN1 = cand;                            // smallest admissible
        if (cand >= 4 * (bt_L ? 1 : world) && cand <= 512) { tN1 = cand; break; }
        if (cand > 512) break;
    }
    if (tN1 == 0) return fail(D4W_EINVAL, "ns/2 = %d has no factorisation with N2 <= %d", M, kMaxTile / 2);
    const int tN2 = Mt / tN1;
    const int N1 = bt_L ? 1 : tN1, N2 = bt_L ? M : tN2;      // layout of the half spectrum: [row][N1][N2]
    // a channel count with a prime factor > 31: the channel transform is a Bluestein convolution of length bz_L = 2^a 3^b 5^c
    // in global memory (fkd_bz_*), and the channel-phase descriptor below is the one of that length
    // (D4W_FKD_BZ_MINPRIME = p: also channel counts with a prime factor >= p, whose loop-based radix stages are slow)
    auto largest_prime = [](int n) {
        int lp = 1;
        for (int p = 2; (long)p * p <= n; ++p)
            while (n % p == 0) { lp = p; n /= p; }
        return std::max(lp, n);
    };
    static const int bz_minprime = [] { const char* v = getenv("D4W_FKD_BZ_MINPRIME"); return (v && atoi(v) > 1) ? atoi(v) : 32; }();
    const int bz_L = (want_mask && largest_prime(nx) >= bz_minprime) ? smooth_len_235(2L * nx - 1) : 0;
    // (no mask = the time phase only, d4w_analytic_long_f32: any row count, the channel descriptor stays a dummy)
    const int Lc = !want_mask ? 1 : (bz_L ? bz_L : nx);
    int C2 = largest_divisor_le(Lc, kMaxTile / 8);
    int C1 = Lc / C2;
    if (C1 > kMaxTile) return fail(D4W_EINVAL, "nx = %d does not factor into C1 <= %d, C2 <= %d", nx, kMaxTile, kMaxTile / 8);
    d4w_fkd_plan* pl = new d4w_fkd_plan();
    pl->nx = nx; pl->ns = ns; pl->M = M; pl->world = world; pl->rank = rank;
    fkd_block(nx, world, rank, &pl->row_begin, &pl->row_end);
    pl->N1 = N1; pl->N2 = N2; pl->C1 = C1; pl->C2 = C2;
    const int nxl = pl->row_end - pl->row_begin;
    int rc;
#define D4W_TRY(x) do { rc = (x); if (rc != D4W_OK) { d4w_fkd_plan_destroy(pl); return rc; } } while (0)
    // ---------------- time-phase descriptor: C1 = 1, C2 = local rows
    d4w_fk_plan& tp = pl->tp;
    memset(&tp.dev, 0, sizeof(tp.dev));
    // one degenerate axis leaves the whole LDS tile to the other: long contiguous strips
    int TA = 512;
    while (TA > 1 && ((long)tN1 * TA > kMaxTile || TA > tN2)) TA /= 2;
    if ((long)tN1 * TA > kMaxTile) { d4w_fkd_plan_destroy(pl); return fail(D4W_EINVAL, "N1 = %d exceeds the LDS tile", tN1); }
    pl->TA_t = TA;
    tp.dev.d = FkDims{nxl, 2 * Mt, Mt, 1, nxl, tN1, tN2, TA, 1};
    tp.dev.scale = 1.0f;
    std::vector<int> f_one, tf_n1, tf_n2, f_n1, f_n2, f_c1, f_c2;
    D4W_TRY(make_axis(&tp, 1, &tp.dev.ax_c1, &f_one));
    D4W_TRY(make_axis(&tp, 1, &tp.dev.ax_c2, &f_one));
    D4W_TRY(make_axis(&tp, tN1, &tp.dev.ax_n1, &tf_n1));
    D4W_TRY(make_axis(&tp, tN2, &tp.dev.ax_n2, &tf_n2));
    {
        std::vector<float2> ones((size_t)nxl, make_float2(1.f, 0.f)), twt((size_t)tN1 * tN2);
        for (int q1 = 0; q1 < tN1; ++q1)
            for (int b = 0; b < tN2; ++b) twt[(size_t)q1 * tN2 + b] = wexp((long long)b * tf_n1[q1], Mt);
        D4W_TRY(upload(&tp, ones, &tp.dev.twc));
        D4W_TRY(upload(&tp, twt, &tp.dev.twt));
        std::vector<float> win = tukey_window(ns, 0.03);
        std::vector<float2> winp(M);
        for (int m = 0; m < M; ++m) winp[m] = make_float2(win[2 * m], win[2 * m + 1]);
        D4W_TRY(upload(&tp, winp, &tp.dev.win));
        pl->bt_win = tp.dev.win;
    }
    pl->gen_t = axis_needs_generic(tp.dev.ax_n1);
    pl->gen_n2 = axis_needs_generic(tp.dev.ax_n2);
    pl->lds_t = ((size_t)tN1 * TA + 2 * kTwLo + tp.dev.ax_c1.nhi + tp.dev.ax_n1.nhi) * sizeof(float2);
    pl->lds_n2 = ((size_t)tN2 + kTwLo + tp.dev.ax_n2.nhi) * sizeof(float2);
    if (bt_L) {
        f_n1.assign(1, 0);
        f_n2.resize(M);
        for (int i = 0; i < M; ++i) f_n2[i] = i;
        std::vector<float2> chirp(M), filt(bt_L);
        std::vector<double> bre(bt_L, 0.0), bim(bt_L, 0.0);
        for (int n = 0; n < M; ++n) {
            const double ph = M_PI * (double)(((long long)n * n) % (2LL * M)) / (double)M;
            chirp[n] = make_float2((float)cos(ph), (float)-sin(ph));
            bre[n] = cos(ph); bim[n] = sin(ph);                               // conj(chirp)
            if (n) { bre[bt_L - n] = cos(ph); bim[bt_L - n] = sin(ph); }
        }
        host_dft_any(bre, bim);
        for (int q1 = 0; q1 < tN1; ++q1)
            for (int i = 0; i < tN2; ++i) {
                const int f = tf_n1[q1] + tN1 * tf_n2[i];                     // frequency at position [q1][i]
                filt[(size_t)q1 * tN2 + i] = make_float2((float)(bre[f] / bt_L), (float)(bim[f] / bt_L));
            }
        D4W_TRY(upload(&tp, chirp, &pl->bt_chirp));
        D4W_TRY(upload(&tp, filt, &pl->bt_filt));
        const char* ch = getenv("D4W_FKD_BT_CHUNK");
        const int lim = (ch && atoi(ch) > 0) ? atoi(ch) : std::max(1, (int)(((size_t)1 << 27) / (size_t)bt_L));
        const int R = std::min(nxl, lim);
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)R * bt_L * sizeof(float2)) != hipSuccess) {
            d4w_fkd_plan_destroy(pl);
            return fail(D4W_ENOMEM, "hipMalloc of the Bluestein scratch (%d x %d complex) failed", R, bt_L);
        }
        tp.allocs.push_back(p);
        pl->bt_S = (float2*)p;
        pl->bt_L = bt_L; pl->bt_R = R;
    } else {
        f_n1 = tf_n1;
        f_n2 = tf_n2;
    }

    // ---------------- ownership of the n1-positions: Hermitian classes {q1, q1'} dealt round-robin
    std::vector<int> p_n1(N1), q1part(N1);
    for (int p = 0; p < N1; ++p) p_n1[f_n1[p]] = p;
    for (int q1 = 0; q1 < N1; ++q1) q1part[q1] = p_n1[(N1 - f_n1[q1]) % N1];
    pl->owner.assign(N1, -1);
    {
        std::vector<int> load(world, 0);
        for (int q1 = 0; q1 < N1; ++q1) {
            if (pl->owner[q1] >= 0) continue;
            int best = 0;
            for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
            pl->owner[q1] = best; ++load[best];
            if (q1part[q1] != q1) { pl->owner[q1part[q1]] = best; ++load[best]; }
        }
    }
    for (int q1 = 0; q1 < N1; ++q1) if (pl->owner[q1] == rank) pl->myq.push_back(q1);
    const int nq = (int)pl->myq.size();
    const int W = std::max(nq, 1) * N2;                       // slab width (complex columns)

    // ---------------- channel-phase descriptor: N1 = 1, "N2" = slab width
    d4w_fk_plan& cp = pl->cp;
    memset(&cp.dev, 0, sizeof(cp.dev));
    int TC = 16, TAc = 512;
    // Bluestein: the scratch [bz_L][Wc] holds a chunk of Wc slab columns at a time (<= 1 GiB; D4W_FKD_BZ_CHUNK pins Wc)
    int Wc = W;
    if (bz_L) {
        const char* ch = getenv("D4W_FKD_BZ_CHUNK");
        int lim = (ch && atoi(ch) > 0) ? atoi(ch) : std::max(16, (int)(((size_t)1 << 27) / (size_t)bz_L) & ~15);
        Wc = std::min(W, lim);
        if (Wc < W && !(ch && atoi(ch) > 0)) {
            // equal chunks: the passes sweep the scratch's full width whatever the last chunk holds
            const int nchunk = ceil_div(W, Wc);
            Wc = std::min(Wc, (ceil_div(W, nchunk) + 15) & ~15);
        }
    }
    while (TC > 1 && (long)C2 * TC > kMaxTile) TC /= 2;
    while (TAc > 1 && ((long)C1 * TAc > kMaxTile || TAc > Wc)) TAc /= 2;
    pl->TC = TC; pl->TA_c = TAc;
    cp.dev.d = FkDims{Lc, 2 * Wc, Wc, C1, C2, 1, Wc, TAc, TC};
    // Bluestein: 1 / bz_L sits in the filter table and 1 / (nx M) in fkd_bz_out of the inverse transform
    cp.dev.scale = bz_L ? 1.0f : (float)(1.0 / ((double)nx * (double)M));
    D4W_TRY(make_axis(&cp, C1, &cp.dev.ax_c1, &f_c1));
    D4W_TRY(make_axis(&cp, C2, &cp.dev.ax_c2, &f_c2));
    D4W_TRY(make_axis(&cp, 1, &cp.dev.ax_n1, &f_one));
    D4W_TRY(make_axis(&cp, 1, &cp.dev.ax_n2, &f_one));
    {
        std::vector<float2> twc((size_t)C1 * C2), ones((size_t)Wc, make_float2(1.f, 0.f));
        for (int q = 0; q < C1; ++q)
            for (int c2 = 0; c2 < C2; ++c2) twc[(size_t)q * C2 + c2] = wexp((long long)c2 * f_c1[q], Lc);
        D4W_TRY(upload(&cp, twc, &cp.dev.twc));
        D4W_TRY(upload(&cp, ones, &cp.dev.twt));
    }
    if (bz_L) {
        std::vector<float2> chirp(nx), filt(bz_L);
        std::vector<double> bre(bz_L, 0.0), bim(bz_L, 0.0);
        for (int n = 0; n < nx; ++n) {
            const double ph = M_PI * (double)(((long long)n * n) % (2LL * nx)) / (double)nx;
            chirp[n] = make_float2((float)cos(ph), (float)-sin(ph));
            bre[n] = cos(ph); bim[n] = sin(ph);                               // conj(chirp)
            if (n) { bre[bz_L - n] = cos(ph); bim[bz_L - n] = sin(ph); }
        }
        host_dft_any(bre, bim);
        for (int q = 0; q < C1; ++q)
            for (int p2 = 0; p2 < C2; ++p2) {
                const int f = f_c1[q] + C1 * f_c2[p2];                        // frequency at row position q C2 + p2
                filt[(size_t)q * C2 + p2] = make_float2((float)(bre[f] / bz_L), (float)(bim[f] / bz_L));
            }
        D4W_TRY(upload(&cp, chirp, &pl->bz_chirp));
        D4W_TRY(upload(&cp, filt, &pl->bz_filt));
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)bz_L * Wc * sizeof(float2)) != hipSuccess) {
            d4w_fkd_plan_destroy(pl);
            return fail(D4W_ENOMEM, "hipMalloc of the Bluestein scratch (%d x %d complex) failed", bz_L, Wc);
        }
        cp.allocs.push_back(p);
        pl->bz_S = (float2*)p;
        pl->bz_L = bz_L; pl->bz_W = Wc;
    }
    pl->gen_c1 = axis_needs_generic(cp.dev.ax_c1);
    pl->gen_c2 = axis_needs_generic(cp.dev.ax_c2);
    pl->lds_c1 = ((size_t)C1 * TAc + 2 * kTwLo + cp.dev.ax_c1.nhi + cp.dev.ax_n1.nhi) * sizeof(float2);
    pl->lds_c2 = ((size_t)C2 * TC + kTwLo + cp.dev.ax_c2.nhi) * sizeof(float2);

    // ---------------- pair-op tables
    std::vector<int> p_c1(C1), p_c2(C2), p_n2(N2), rowk(nx), rowpart(nx), mirror0(N2), jqpart(std::max(nq, 1), 0);
    for (int p = 0; p < C1; ++p) p_c1[f_c1[p]] = p;
    for (int p = 0; p < C2; ++p) p_c2[f_c2[p]] = p;
    for (int p = 0; p < N2; ++p) p_n2[f_n2[p]] = p;
    if (bz_L || !want_mask) {                      // Bluestein: natural wavenumber order
        for (int r = 0; r < nx; ++r) { rowk[r] = r; rowpart[r] = (nx - r) % nx; }
    } else {
        for (int q = 0; q < C1; ++q)
            for (int p2 = 0; p2 < C2; ++p2) rowk[q * C2 + p2] = f_c1[q] + C1 * f_c2[p2];
        for (int r = 0; r < nx; ++r) {
            const int km = (nx - rowk[r]) % nx;
            rowpart[r] = p_c1[km % C1] * C2 + p_c2[km / C1];
        }
    }
    for (int i = 0; i < N2; ++i) mirror0[i] = p_n2[(N2 - f_n2[i]) % N2];
    for (int j = 0; j < nq; ++j)
        for (int j2 = 0; j2 < nq; ++j2)
            if (pl->myq[j2] == q1part[pl->myq[j]]) jqpart[j] = j2;
    std::vector<float2> wrow(N1), wcol(N2);
    for (int q1 = 0; q1 < N1; ++q1) wrow[q1] = wexp(f_n1[q1], ns);
    for (int i = 0; i < N2; ++i) wcol[i] = wexp((long long)N1 * f_n2[i], ns);
    pl->h_jqpart = jqpart; pl->h_mirror0 = mirror0;
    std::vector<int> myq = pl->myq;
    if (myq.empty()) myq.push_back(0);
    const int *c_rowk, *c_k1, *c_k2, *c_q1of;
    FkdSlab& S = pl->slab;
    S.nx = nx; S.nq = nq; S.N2 = N2;
    D4W_TRY(upload(&cp, myq, &c_q1of));
    D4W_TRY(upload(&cp, jqpart, &S.jq_partner));
    D4W_TRY(upload(&cp, rowpart, &S.row_partner));
    D4W_TRY(upload(&cp, mirror0, &S.mirror0));
    D4W_TRY(upload(&cp, wrow, &S.wrow));
    D4W_TRY(upload(&cp, wcol, &S.wcol));
    D4W_TRY(upload(&cp, rowk, &c_rowk));
    D4W_TRY(upload(&cp, f_n1, &c_k1));
    D4W_TRY(upload(&cp, f_n2, &c_k2));
    S.q1_of = c_q1of;
    pl->d_rowk = const_cast<int*>(c_rowk); pl->d_k1 = const_cast<int*>(c_k1); pl->d_k2 = const_cast<int*>(c_k2);
    pl->d_q1of = const_cast<int*>(c_q1of);
    S.hilbert = 0;
    if (want_mask) {
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)nx * W * sizeof(float)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc of the slab mask failed"); }
        cp.allocs.push_back(p); pl->d_mask = (float*)p;
        if (hipMalloc(&p, (size_t)nx * sizeof(float)) != hipSuccess) { d4w_fkd_plan_destroy(pl); return fail(D4W_ENOMEM, "hipMalloc failed"); }
        cp.allocs.push_back(p); pl->d_nyq = (float*)p;
    }
    S.mask = pl->d_mask; S.nyq = pl->d_nyq;
#undef D4W_TRY
    {
        int devid = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
            pl->num_cu = prop.multiProcessorCount;
        const void* fns[] = {
            (const void*)fk_passA_fwd<true, true>, (const void*)fk_passA_fwd<true, false>,
            (const void*)fk_passA_fwd<false, true>, (const void*)fk_passA_fwd<false, false>,
            (const void*)fk_passA_inv<true>, (const void*)fk_passA_inv<false>,
            (const void*)fk_passC<false, true>, (const void*)fk_passC<false, false>,
            (const void*)fk_passC<true, true>, (const void*)fk_passC<true, false>,
            (const void*)fkd_rows_n2<false, true>, (const void*)fkd_rows_n2<false, false>,
            (const void*)fkd_rows_n2<true, true>, (const void*)fkd_rows_n2<true, false>,
            (const void*)fkd_bz_passA_fwd<true>, (const void*)fkd_bz_passA_fwd<false>,
            (const void*)fkd_bz_passA_inv<true>, (const void*)fkd_bz_passA_inv<false>,
            (const void*)fkd_bz_passC<true>, (const void*)fkd_bz_passC<false>,
            (const void*)fkd_bt_passA_fwd<true>, (const void*)fkd_bt_passA_fwd<false>,
            (const void*)fkd_bt_passA_inv<true>, (const void*)fkd_bt_passA_inv<false>,
            (const void*)fkd_bt_rows<true>, (const void*)fkd_bt_rows<false>};
        for (const void* f : fns)
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    *out = pl;
    return D4W_OK;
}

int d4w_fkd_plan_create(int nx, int ns, int world, int rank, d4w_fkd_plan** out) {
    return fkd_plan_build(nx, ns, world, rank, true, out);
}

/* info12 = {nx, ns, world, rank, row_begin, row_end, N1, N2, nq_local, C1, C2, packed (0 / 1)} */
int d4w_fkd_plan_info(const d4w_fkd_plan* pl, int* info) {
    if (!pl || !info) return fail(D4W_EINVAL, "NULL argument");
    const int v[12] = {pl->nx, pl->ns, pl->world, pl->rank, pl->row_begin, pl->row_end, pl->N1, pl->N2,
                       (int)pl->myq.size(), pl->C1, pl->C2, pl->sp ? 1 : 0};
    memcpy(info, v, sizeof(v));
    return D4W_OK;
}

/* info2 = {slab columns the Bluestein channel phase transforms with the mask set last, slab columns} ({0, 0}: the plan's
 * channel phase is not the global-memory Bluestein form) */
int d4w_fkd_plan_live_columns(const d4w_fkd_plan* pl, int* info2) {
    if (!pl || !info2) return fail(D4W_EINVAL, "NULL argument");
    const int W = (int)pl->myq.size() * pl->N2;
    info2[0] = pl->bz_L ? (pl->bz_nlive >= 0 ? pl->bz_nlive : W) : 0;
    info2[1] = pl->bz_L ? W : 0;
    return D4W_OK;
}

int d4w_fkd_plan_q1_owner(const d4w_fkd_plan* pl, int* owner) {
    if (!pl || !owner) return fail(D4W_EINVAL, "NULL argument");
    memcpy(owner, pl->owner.data(), pl->owner.size() * sizeof(int));
    return D4W_OK;
}

}  // extern "C"
namespace d4w {
// colmax[c] = max over the rows of |folded gain| at slab column c, as the bits of a non-negative float (NaN: +inf)
__global__ __launch_bounds__(kThreads) void fkd_col_max(const float* __restrict__ mask, int nx, int W,
                                                        unsigned* __restrict__ colmax) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    float m = 0.f;
    for (int r = blockIdx.y; r < nx; r += gridDim.y) {
        const float g = fabsf(mask[(size_t)r * W + c]);
        m = (g != g) ? INFINITY : fmaxf(m, g);
    }
    atomicMax(&colmax[c], __float_as_uint(m));
}
// the gains of the columns that are not transformed become exact zeros (they are below fk_zero_gain, the level the
// specialised plans drop as well)
__global__ __launch_bounds__(kThreads) void fkd_zero_dead_cols(float* __restrict__ mask, int nx, int W,
                                                               const unsigned* __restrict__ live) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W || live[c]) return;
    for (int r = blockIdx.y; r < nx; r += gridDim.y) mask[(size_t)r * W + c] = 0.f;
}
}  // namespace d4w
extern "C" {

// The live columns of a Bluestein channel phase (see d4w_fkd_plan::bz_cols): column maxima of the folded mask, the zero-gain
// level of fk_zero_gain (2^-24 / sqrt(nx ns) of the largest gain: the Gaussian tails of hybrid_ninf never reach an exact
// zero), closure under the pair op's column map, one small copy each way.  D4W_FKD_BZ_BAND=0 keeps every column (A/B).
static int fkd_bz_live_columns(d4w_fkd_plan* pl, void* stream) {
    pl->bz_nlive = -1;
    static const int band_env = [] { const char* v = getenv("D4W_FKD_BZ_BAND"); return v ? atoi(v) : 1; }();
    const int nq = (int)pl->myq.size(), N2 = pl->N2, W = nq * N2;
    if (!pl->bz_L || !band_env || nq == 0 || !pl->d_mask) return D4W_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!pl->bz_cols) {
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)W * sizeof(int)) != hipSuccess) return fail(D4W_ENOMEM, "hipMalloc of the live-column list failed");
        pl->cp.allocs.push_back(p); pl->bz_cols = (int*)p;
        if (hipMalloc(&p, (size_t)W * sizeof(unsigned)) != hipSuccess) return fail(D4W_ENOMEM, "hipMalloc of the live-column flags failed");
        pl->cp.allocs.push_back(p); pl->bz_flags = (unsigned*)p;
    }
    const dim3 grid(ceil_div(W, kThreads), std::min(pl->nx, 64));
    D4W_HIP(hipMemsetAsync(pl->bz_flags, 0, (size_t)W * sizeof(unsigned), st));
    D4W_LAUNCH(fkd_col_max, grid, dim3(kThreads), 0, stream, (const float*)pl->d_mask, pl->nx, W, pl->bz_flags);
    std::vector<unsigned> cm((size_t)W);
    D4W_HIP(hipMemcpyAsync(cm.data(), pl->bz_flags, (size_t)W * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    D4W_HIP(hipStreamSynchronize(st));
    float gmax = 0.f;
    for (int c = 0; c < W; ++c) { float g; memcpy(&g, &cm[c], 4); if (g < INFINITY) gmax = std::max(gmax, g); }
    FkDims dd{};
    dd.nx = pl->nx; dd.ns = pl->ns;
    const float zero_gain = (float)fk_zero_gain(dd, pl->prune_eps, (double)gmax);
    std::vector<unsigned> live((size_t)W, 0u);
    auto is_live = [&](size_t c) { float g; memcpy(&g, &cm[c], 4); return g > zero_gain; };
    for (int jq = 0; jq < nq; ++jq) {
        const bool k1zero = (pl->myq[jq] == 0);
        const int jp = pl->h_jqpart[jq];
        for (int i = 0; i < N2; ++i) {
            const int j = k1zero ? pl->h_mirror0[i] : (N2 - 1 - i);
            const size_t a = (size_t)jq * N2 + i, b = (size_t)jp * N2 + j;
            if (is_live(a) || is_live(b) || (k1zero && i == 0)) live[a] = live[b] = 1u;     // (the DC / Nyquist column: gains in d_nyq)
        }
    }
    std::vector<int> cols;
    cols.reserve((size_t)W);
    for (int c = 0; c < W; ++c) if (live[c]) cols.push_back(c);
    if ((int)cols.size() >= W) return D4W_OK;                                     // nothing to skip
    D4W_HIP(hipMemcpyAsync(pl->bz_cols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice, st));
    D4W_HIP(hipMemcpyAsync(pl->bz_flags, live.data(), (size_t)W * sizeof(unsigned), hipMemcpyHostToDevice, st));
    D4W_LAUNCH(fkd_zero_dead_cols, grid, dim3(kThreads), 0, stream, pl->d_mask, pl->nx, W, (const unsigned*)pl->bz_flags);
    D4W_HIP(hipStreamSynchronize(st));                                            // (the host vectors go out of scope)
    pl->bz_nlive = (int)cols.size();
    return D4W_OK;
}

static int fkd_launch_fold(const FkdMaskSrc& src, int nx, int ns, int N1, int N2, int nq, const int* rowk, const int* q1of,
                           const int* k1, const int* k2, float* mask, float* nyq, void* stream) {
    dim3 grid(std::min(ceil_div(nq * N2, kThreads), 64), nx);
    if (src.dense)
        D4W_LAUNCH(fkd_fold_mask, grid, dim3(kThreads), 0, stream, nx, ns, N1, N2, nq, src.dense, rowk, q1of, k1, k2, mask, nyq, src.aff);
    else
        D4W_LAUNCH(fkd_fold_design, grid, dim3(kThreads), 0, stream, nx, ns, N1, N2, nq, src.A, src.mode, rowk, q1of, k1, k2,
                   mask, nyq);
    return D4W_OK;
}

static int fkd_set_mask_packed(d4w_fkd_plan* pl, const FkdMaskSrc& src, void* stream) {
    d4w_fk_plan* sp = pl->sp;
    const FkDims& d = sp->dev.d;
    const int nq = (int)pl->myq.size();
    hipStream_t st = (hipStream_t)stream;
    pl->has_mask = true;
    pl->fdev_c.pairs = pl->d_pairs_slab;
    pl->fdev_c.live = nullptr;
    pl->npairs_run = (int)pl->h_pairs_slab.size();
    pl->live_rows = d.nx;
    if (nq == 0) return D4W_OK;
    const int W = nq * d.N2;
    if (int rc = fkd_launch_fold(src, d.nx, d.ns, d.N1, d.N2, nq, (const int*)sp->d_rowk, (const int*)pl->d_q1of,
                                 (const int*)sp->d_k1, (const int*)sp->d_k2, pl->d_mask, pl->d_nyq, stream))
        return rc;
    const char* np = getenv("D4W_FK_NOPRUNE");
    if (np && atoi(np) > 0) return D4W_OK;
    // rows whose gains are all zero in the owned sub-rows (and whose Hermitian partner's are): skipped by passes
    // C, B, C' of this rank's slab, exactly as in the single-device plan
    D4W_HIP(hipMemsetAsync(pl->d_rowmax, 0, (size_t)d.nx * sizeof(unsigned), st));
    D4W_LAUNCH(fkd_row_max, dim3(d.nx), dim3(kThreads), 0, stream, (const float*)pl->d_mask, (const float*)pl->d_nyq, W,
               pl->d_rowmax);
    std::vector<unsigned> rmax(d.nx);
    D4W_HIP(hipMemcpyAsync(rmax.data(), pl->d_rowmax, (size_t)d.nx * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    D4W_HIP(hipStreamSynchronize(st));
    std::vector<char> lv(d.nx, 0);
    std::vector<int2> run;
    run.reserve(pl->h_pairs_slab.size());
    for (const int2& pr : pl->h_pairs_slab) {
        const int ra = pr.x / nq, rb = pr.y / nq;
        if (rmax[ra] || rmax[rb]) {
            lv[ra] = lv[rb] = 1;
            run.push_back(pr);
        }
    }
    int nlive = 0;
    for (int r = 0; r < d.nx; ++r) nlive += lv[r];
    if (nlive < d.nx) {
        const int RA = sp->fast->C2A, RB = sp->fast->C2B;
        std::vector<unsigned> bits((size_t)d.C1 * RA, 0u);
        for (int r = 0; r < d.nx; ++r)
            if (lv[r]) {
                const int q = r / d.C2, p2 = r % d.C2;
                bits[(size_t)q * RA + p2 / RB] |= 1u << (p2 % RB);
            }
        D4W_HIP(hipMemcpyAsync(pl->d_livebits, bits.data(), bits.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
        if (!run.empty())
            D4W_HIP(hipMemcpyAsync(pl->d_pairs_live, run.data(), run.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        D4W_HIP(hipStreamSynchronize(st));
        pl->fdev_c.live = pl->d_livebits;
        pl->fdev_c.pairs = pl->d_pairs_live;
        pl->npairs_run = (int)run.size();
        pl->live_rows = nlive;
    }
    return D4W_OK;
}

int d4w_fkd_plan_is_packed(const d4w_fkd_plan* pl) { return (pl && pl->sp) ? 1 : 0; }

static int fkd_set_mask_src(d4w_fkd_plan* pl, const FkdMaskSrc& src, void* stream) {
    if (pl->nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", pl->nx);
    if (pl->sp) return fkd_set_mask_packed(pl, src, stream);
    const int nq = (int)pl->myq.size();
    if (nq > 0)
        if (int rc = fkd_launch_fold(src, pl->nx, pl->ns, pl->N1, pl->N2, nq, (const int*)pl->d_rowk, (const int*)pl->d_q1of,
                                     (const int*)pl->d_k1, (const int*)pl->d_k2, pl->d_mask, pl->d_nyq, stream))
            return rc;
    if (int rc = fkd_bz_live_columns(pl, stream)) return rc;
    pl->has_mask = true;
    return D4W_OK;
}

int d4w_fkd_set_mask_dense_f32(d4w_fkd_plan* pl, const float* mask_shifted, void* stream) {
    if (!pl || !mask_shifted || !pl->d_mask) return fail(D4W_EINVAL, "NULL argument");
    FkdMaskSrc src;
    src.dense = mask_shifted;
    return fkd_set_mask_src(pl, src, stream);
}

}  // extern "C"
static void fkd_set_prune(d4w_fkd_plan* pl, double prune_eps) { pl->prune_eps = prune_eps; }
static int fkd_set_mask_dense_affine(d4w_fkd_plan* pl, const float* mask_shifted, float a, float b, int on, void* stream) {
    if (!pl || !mask_shifted || !pl->d_mask) return fail(D4W_EINVAL, "NULL argument");
    if (on && pl->sp) return fail(D4W_EINVAL, "the packed distributed plan folds no affine map");
    FkdMaskSrc src;
    src.dense = mask_shifted;
    src.aff = FkAffine{a, b, on};
    return fkd_set_mask_src(pl, src, stream);
}
extern "C" {

int d4w_fkd_set_mask_design_f32(d4w_fkd_plan* pl, int mode, double k_spacing, double t_spacing, const double* params8_host,
                                int i0, int i1, const double* hrow_dev, void* stream) {
    if (!pl || !params8_host || !pl->d_mask) return fail(D4W_EINVAL, "NULL argument");
    if (mode < 0 || mode > 2)
        return fail(D4W_EINVAL, "mode %d has no closed form (the Gaussian-blurred designs go through d4w_design_mask_f32)", mode);
    if (mode == 2 && !hrow_dev) return fail(D4W_EINVAL, "hybrid_ninf needs the |H|^2 row");
    FkdMaskSrc src;
    src.A = make_design_args(pl->nx, pl->ns, k_spacing, t_spacing, params8_host, i0, i1, hrow_dev);
    src.mode = mode;
    return fkd_set_mask_src(pl, src, stream);
}

/* packed path: x_loc [nxl][ns] -> packed [dest rank][nxl][nq(dest)][N2] complex (what the all-to-all sends) */
int d4w_fkd_time_fwd_packed_rows_f32(d4w_fkd_plan* pl, const float* x_loc, float* packed, int taper, int l0, int l1,
                                     void* stream) {
    if (!pl || !x_loc || !packed) return fail(D4W_EINVAL, "NULL argument");
    if (!pl->sp) return fail(D4W_EINVAL, "this shape runs the generic distributed plan: d4w_fkd_time_fwd_f32");
    const FkFastEntry& F = *pl->sp->fast;
    const FkDims& d = pl->sp->dev.d;
    const int nxl = pl->row_end - pl->row_begin;
    if (l0 < 0 || l1 > nxl || l0 > l1 || l0 % d.C1) return fail(D4W_EINVAL, "row range [%d, %d) must start at a multiple of %d inside the local block", l0, l1, d.C1);
    const int NBX = d.N2 / d.TA, t0 = (l0 / d.C1) * NBX, t1 = ceil_div(l1, d.C1) * NBX;
    if (t1 <= t0) return D4W_OK;
    FkGeo geo = pl->geo_t;
    geo.nrows = l1;                                   // rows >= l1 of the last group belong to the next chunk
    return launch_k(taper ? F.T_fwd_taper : F.T_fwd, dim3(std::min(t1 - t0, pl->num_cu * pl->sp->wgA)), dim3(F.thrA), F.ldsA, stream,
                    pl->dev_t, reinterpret_cast<const float2*>(x_loc), reinterpret_cast<float2*>(packed), t0, t1, NBX, 0, geo);
}

int d4w_fkd_time_fwd_packed_f32(d4w_fkd_plan* pl, const float* x_loc, float* packed, int taper, void* stream) {
    if (!pl) return fail(D4W_EINVAL, "NULL argument");
    return d4w_fkd_time_fwd_packed_rows_f32(pl, x_loc, packed, taper, 0, pl->row_end - pl->row_begin, stream);
}

/* packed path: packed [src rank][nxl][nq(src)][N2] (what the second all-to-all delivers) -> y_loc [nxl][ns] */
int d4w_fkd_time_inv_packed_rows_f32(d4w_fkd_plan* pl, const float* packed, float* y_loc, int l0, int l1, void* stream) {
    if (!pl || !packed || !y_loc) return fail(D4W_EINVAL, "NULL argument");
    if (!pl->sp) return fail(D4W_EINVAL, "this shape runs the generic distributed plan: d4w_fkd_time_inv_f32");
    const FkFastEntry& F = *pl->sp->fast;
    const FkDims& d = pl->sp->dev.d;
    const int nxl = pl->row_end - pl->row_begin;
    if (l0 < 0 || l1 > nxl || l0 > l1 || l0 % d.C1) return fail(D4W_EINVAL, "row range [%d, %d) must start at a multiple of %d inside the local block", l0, l1, d.C1);
    const int NBX = d.N2 / d.TA, t0 = (l0 / d.C1) * NBX, t1 = ceil_div(l1, d.C1) * NBX;
    if (t1 <= t0) return D4W_OK;
    FkGeo geo = pl->geo_t;
    geo.nrows = l1;
    return launch_k(F.T_inv, dim3(std::min(t1 - t0, pl->num_cu * pl->sp->wgA)), dim3(F.thrA), F.ldsA, stream, pl->dev_t,
                    reinterpret_cast<float2*>(y_loc), t0, t1, NBX, 0, geo, reinterpret_cast<const float2*>(packed));
}

/* ... with the row statistics of the filtered rows (mean, max|.|: what the matched filter normalises by, detect.py:157)
 * from the pass's epilogue; row_mean / row_maxabs [nxl] must be zeroed before the first chunk */
int d4w_fkd_time_inv_packed_rows_stats_f32(d4w_fkd_plan* pl, const float* packed, float* y_loc, int l0, int l1,
                                           double* row_mean, float* row_maxabs, void* stream) {
    if (!pl || !packed || !y_loc || !row_mean || !row_maxabs) return fail(D4W_EINVAL, "NULL argument");
    if (!pl->sp) return fail(D4W_EINVAL, "this shape runs the generic distributed plan");
    const FkFastEntry& F = *pl->sp->fast;
    const FkDims& d = pl->sp->dev.d;
    const int nxl = pl->row_end - pl->row_begin;
    if (l0 < 0 || l1 > nxl || l0 > l1 || l0 % d.C1) return fail(D4W_EINVAL, "row range [%d, %d) must start at a multiple of %d inside the local block", l0, l1, d.C1);
    const int NBX = d.N2 / d.TA, t0 = (l0 / d.C1) * NBX, t1 = ceil_div(l1, d.C1) * NBX;
    if (t1 <= t0) return D4W_OK;
    int run = 1;
    for (int r = 1; r <= NBX && r <= 30; ++r) if (NBX % r == 0) run = r;
    const int nruns = (t1 - t0) / run;
    FkGeo geo = pl->geo_t;
    geo.nrows = l1;
    return launch_k(F.T_inv_stats, dim3(std::min(nruns, pl->num_cu * pl->sp->wgA)), dim3(F.thrA), F.ldsA, stream, pl->dev_t,
                    reinterpret_cast<float2*>(y_loc), run, nruns, row_mean, (unsigned*)row_maxabs, NBX, 0, geo,
                    reinterpret_cast<const float2*>(packed), t0);
}

int d4w_fkd_time_inv_packed_f32(d4w_fkd_plan* pl, const float* packed, float* y_loc, void* stream) {
    if (!pl) return fail(D4W_EINVAL, "NULL argument");
    return d4w_fkd_time_inv_packed_rows_f32(pl, packed, y_loc, 0, pl->row_end - pl->row_begin, stream);
}

static int fkd_chan_apply_packed(d4w_fkd_plan* pl, float* slab, void* stream) {
    const FkFastEntry& F = *pl->sp->fast;
    const FkDims& d = pl->sp->dev.d;
    const int nq = (int)pl->myq.size();
    if (nq == 0) return D4W_OK;
    const int W = nq * d.N2;
    const int nblkA = ceil_div(W, d.N1 * d.TA), nblkC = W / d.TC;
    const int ntA = d.C2 * nblkA, ntC = d.C1 * nblkC;
    float2* d2 = reinterpret_cast<float2*>(slab);
    const int cu = pl->num_cu;
    const dim3 gA(std::min(ntA, cu * pl->sp->wgA)), gC(std::min(ntC, cu * pl->sp->wgC)),
        gB(std::max(1, std::min(pl->npairs_run, cu * pl->sp->wgB)));
    int rc;
    // pass A MODE 2: tile u -> (c2 = u / nblkA, column block u % nblkA)
    if ((rc = launch_k(F.Ac_fwd, gA, dim3(F.thrA), F.ldsA, stream, pl->dev_c, (const float2*)d2, d2, 0, ntA, nblkA, 0, pl->geo_c))) return rc;
    if ((rc = launch_k(F.Cs_fwd, gC, dim3(F.thrC), F.ldsC, stream, pl->dev_c, pl->fdev_c, d2, 0, ntC, nblkC, 0, pl->geo_c))) return rc;
    FkGeo gb = pl->geo_c;
    gb.nq = nq;
    if (pl->npairs_run > 0 &&
        (rc = launch_k(F.Bs_mid, gB, dim3(F.thrB), F.ldsB, stream, pl->dev_c, pl->fdev_c, d2, 0, pl->npairs_run, gb))) return rc;
    if ((rc = launch_k(F.Cs_inv, gC, dim3(F.thrC), F.ldsC, stream, pl->dev_c, pl->fdev_c, d2, 0, ntC, nblkC, 0, pl->geo_c))) return rc;
    return launch_k(F.Ac_inv, gA, dim3(F.thrA), F.ldsA, stream, pl->dev_c, d2, 0, ntA, nblkA, 0, pl->geo_c, (const float2*)nullptr);
}

}  // extern "C"
// time transform of the local rows as a Bluestein convolution through the scratch, a chunk of rows at a time
// (inv = 0: packed real rows -> half spectrum in natural order; inv = 1: back, x tp.dev.scale)
static int fkd_time_bluestein(d4w_fkd_plan* pl, const float2* src, float2* dst, int inv, int taper, void* stream) {
    FkDev P = pl->tp.dev;
    P.scale = 1.0f;
    const int nxl = P.d.nx, L = pl->bt_L, R = pl->bt_R;
    const int persist = pl->num_cu * 4;
    const dim3 blk(kMaxThreads);
    FkdBt Z;
    Z.chirp = pl->bt_chirp; Z.filt = pl->bt_filt; Z.win = (taper && !inv) ? pl->bt_win : nullptr;
    Z.pitch = (size_t)pl->M; Z.m = pl->M; Z.inv = inv;
    Z.scale = inv ? pl->tp.dev.scale : 1.0f;
    (void)L;
    for (int r0 = 0; r0 < nxl; r0 += R) {
        const int nr = std::min(R, nxl - r0);
        const int ntA = ceil_div(P.d.N2, P.d.TA) * nr, nsub = nr * P.d.N1;
        int rc;
        if ((rc = launch_k(pl->gen_t ? fkd_bt_passA_fwd<true> : fkd_bt_passA_fwd<false>, dim3(std::min(ntA, persist)), blk, pl->lds_t, stream, P, Z, src + (size_t)r0 * pl->M, pl->bt_S, ntA))) return rc;
        if ((rc = launch_k(pl->gen_n2 ? fkd_bt_rows<true> : fkd_bt_rows<false>, dim3(std::min(nsub, persist)), blk, pl->lds_n2, stream, P, Z, pl->bt_S, nsub))) return rc;
        if ((rc = launch_k(pl->gen_t ? fkd_bt_passA_inv<true> : fkd_bt_passA_inv<false>, dim3(std::min(ntA, persist)), blk, pl->lds_t, stream, P, Z, (const float2*)pl->bt_S, dst + (size_t)r0 * pl->M, ntA))) return rc;
    }
    return D4W_OK;
}
extern "C" {

int d4w_fkd_time_fwd_f32(d4w_fkd_plan* pl, const float* x_loc, float* z_loc, int taper, void* stream) {
    if (!pl || !x_loc || !z_loc) return fail(D4W_EINVAL, "NULL argument");
    if (pl->sp) return fail(D4W_EINVAL, "this shape runs the packed distributed plan: d4w_fkd_time_fwd_packed_f32");
    if (pl->bt_L) return fkd_time_bluestein(pl, reinterpret_cast<const float2*>(x_loc), reinterpret_cast<float2*>(z_loc), 0, taper, stream);
    const FkDev& P = pl->tp.dev;
    const int nxl = P.d.nx;
    const int ntA = ceil_div(P.d.N2, P.d.TA) * nxl, nsub = nxl * pl->N1;
    const int persist = pl->num_cu * 4;
    const float2* src = reinterpret_cast<const float2*>(x_loc);
    float2* dst = reinterpret_cast<float2*>(z_loc);
    const dim3 blk(kMaxThreads);
    int rc;
    if (taper) rc = launch_k(pl->gen_t ? fk_passA_fwd<true, true> : fk_passA_fwd<true, false>, dim3(std::min(ntA, persist)), blk, pl->lds_t, stream, P, src, dst, ntA);
    else rc = launch_k(pl->gen_t ? fk_passA_fwd<false, true> : fk_passA_fwd<false, false>, dim3(std::min(ntA, persist)), blk, pl->lds_t, stream, P, src, dst, ntA);
    if (rc) return rc;
    return launch_k(pl->gen_n2 ? fkd_rows_n2<false, true> : fkd_rows_n2<false, false>, dim3(std::min(nsub, persist)), blk, pl->lds_n2, stream, P, dst, nsub);
}

int d4w_fkd_time_inv_f32(d4w_fkd_plan* pl, float* z_loc, void* stream) {
    if (!pl || !z_loc) return fail(D4W_EINVAL, "NULL argument");
    if (pl->sp) return fail(D4W_EINVAL, "this shape runs the packed distributed plan: d4w_fkd_time_inv_packed_f32");
    if (pl->bt_L) return fkd_time_bluestein(pl, reinterpret_cast<const float2*>(z_loc), reinterpret_cast<float2*>(z_loc), 1, 0, stream);
    const FkDev& P = pl->tp.dev;
    const int nxl = P.d.nx;
    const int ntA = ceil_div(P.d.N2, P.d.TA) * nxl, nsub = nxl * pl->N1;
    const int persist = pl->num_cu * 4;
    float2* d2 = reinterpret_cast<float2*>(z_loc);
    const dim3 blk(kMaxThreads);
    int rc = launch_k(pl->gen_n2 ? fkd_rows_n2<true, true> : fkd_rows_n2<true, false>, dim3(std::min(nsub, persist)), blk, pl->lds_n2, stream, P, d2, nsub);
    if (rc) return rc;
    return launch_k(pl->gen_t ? fk_passA_inv<true> : fk_passA_inv<false>, dim3(std::min(ntA, persist)), blk, pl->lds_t, stream, P, d2, ntA);
}

int d4w_fkd_chan_apply_f32(d4w_fkd_plan* pl, float* slab, void* stream) {
    if (!pl || (!slab && !pl->myq.empty())) return fail(D4W_EINVAL, "NULL argument");
    if (!pl->has_mask) return fail(D4W_EINVAL, "no mask set on this plan");
    if (pl->sp) return fkd_chan_apply_packed(pl, slab, stream);
    const int nq = (int)pl->myq.size();
    if (nq == 0) return D4W_OK;
    const FkDev& P = pl->cp.dev;
    const FkDims& d = P.d;
    const int ntA = ceil_div(d.N2, d.TA) * d.C2, ntC = ceil_div(d.M, d.TC) * d.C1;
    const int persist = pl->num_cu * 4;
    float2* d2 = reinterpret_cast<float2*>(slab);
    const dim3 blk(kMaxThreads);
    int rc;
    if (pl->bz_L) {
        // channel DFT, pair operation with the folded mask, inverse channel DFT -- each DFT a Bluestein convolution
        // through the scratch, a chunk of columns at a time (the columns are independent)
        if (pl->nx > 65535 || nq > 65535) return fail(D4W_EINVAL, "slab %d x %d exceeds the pair-op grid", pl->nx, nq);
        const int W = nq * pl->N2, Wc = pl->bz_W, L = pl->bz_L;
        float2* S = pl->bz_S;
        (void)L;
        // the live columns only, gathered into the scratch through the list (dead columns: zeros out of the pair op)
        const bool listed = pl->bz_nlive >= 0;
        const int Wl = listed ? pl->bz_nlive : W;
        for (int inv = 0; inv < 2; ++inv) {
            for (int c0 = 0; c0 < Wl; c0 += Wc) {
                FkdBz Z;
                Z.chirp = pl->bz_chirp; Z.filt = pl->bz_filt; Z.pitch = (size_t)W; Z.nx = pl->nx;
                Z.ncol = std::min(Wc, Wl - c0); Z.inv = inv;
                Z.scale = inv ? (float)(1.0 / ((double)pl->nx * (double)pl->M)) : 1.0f;
                Z.cols = listed ? pl->bz_cols + c0 : nullptr;
                float2* base = listed ? d2 : d2 + c0;
                if ((rc = launch_k(pl->gen_c1 ? fkd_bz_passA_fwd<true> : fkd_bz_passA_fwd<false>, dim3(std::min(ntA, persist)), blk, pl->lds_c1, stream, P, Z, (const float2*)base, S, ntA))) return rc;
                if ((rc = launch_k(pl->gen_c2 ? fkd_bz_passC<true> : fkd_bz_passC<false>, dim3(std::min(ntC, persist)), blk, pl->lds_c2, stream, P, Z, S, ntC))) return rc;
                if ((rc = launch_k(pl->gen_c1 ? fkd_bz_passA_inv<true> : fkd_bz_passA_inv<false>, dim3(std::min(ntA, persist)), blk, pl->lds_c1, stream, P, Z, (const float2*)S, base, ntA))) return rc;
            }
            if (!inv && (rc = launch_k(fkd_pair_slab, dim3(std::min(ceil_div(pl->N2, kThreads), 8), pl->nx, nq), dim3(kThreads), 0,
                                        stream, pl->slab, d2))) return rc;
        }
        return D4W_OK;
    }
    if ((rc = launch_k(pl->gen_c1 ? fk_passA_fwd<false, true> : fk_passA_fwd<false, false>, dim3(std::min(ntA, persist)), blk, pl->lds_c1, stream, P, (const float2*)d2, d2, ntA))) return rc;
    if ((rc = launch_k(pl->gen_c2 ? fk_passC<false, true> : fk_passC<false, false>, dim3(std::min(ntC, persist)), blk, pl->lds_c2, stream, P, d2, ntC))) return rc;
    if (pl->nx > 65535 || nq > 65535) return fail(D4W_EINVAL, "slab %d x %d exceeds the pair-op grid", pl->nx, nq);
    if ((rc = launch_k(fkd_pair_slab, dim3(std::min(ceil_div(pl->N2, kThreads), 8), pl->nx, nq), dim3(kThreads), 0,
                       stream, pl->slab, d2))) return rc;
    if ((rc = launch_k(pl->gen_c2 ? fk_passC<true, true> : fk_passC<true, false>, dim3(std::min(ntC, persist)), blk, pl->lds_c2, stream, P, d2, ntC))) return rc;
    return launch_k(pl->gen_c1 ? fk_passA_inv<true> : fk_passA_inv<false>, dim3(std::min(ntA, persist)), blk, pl->lds_c1, stream, P, d2, ntA);
}

/* ------------------------------------------------------------------------------------------
 * Analytic signal of rows too long for one workgroup's LDS (e.g. 120 000 samples): the four-step
 * time phase of the distributed plan over ALL rows (world = 1), the Hilbert pair op on the packed
 * spectrum, the inverse time phase, and one elementwise combine with x.  Same modes as
 * d4w_analytic_f32 (spectral.hip).  ws: [nx][ns] float32 scratch.
 * ------------------------------------------------------------------------------------------ */
}  // extern "C"

namespace d4w {
__global__ __launch_bounds__(kThreads) void analytic_combine(const float* __restrict__ x, const float* __restrict__ h,
                                                              float* __restrict__ y, int ns, int mode,
                                                              const float* __restrict__ var, float fscale) {
    const size_t base = (size_t)blockIdx.y * ns;
    const int nout = (mode == 3) ? ns - 1 : ns;
    float* yr = y + (size_t)blockIdx.y * nout;
    const float inv_var = (mode == 2 || mode == 4) ? 1.0f / var[blockIdx.y] : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += gridDim.x * blockDim.x) {
        const float re = x[base + i], im = h[base + i];
        float v;
        if (mode == 0) v = sqrtf(fmaf(re, re, im * im));
        else if (mode == 1) v = im;
        else if (mode == 2) v = 10.0f * log10f(fmaf(re, re, im * im) * inv_var);
        else if (mode == 4) v = sqrtf(fmaf(re, re, im * im) * inv_var);
        else {
            const float2 p = c_mulc(make_float2(x[base + i + 1], h[base + i + 1]), make_float2(re, im));
            v = atan2f(p.y, p.x) * fscale;
        }
        yr[i] = v;
    }
}
// Odd row lengths on the long-row path: the row as a COMPLEX sequence of its own length (no packing), the one-sided
// multiplier of scipy.signal.hilbert on its spectrum, the inverse transform = the analytic signal itself.
__global__ __launch_bounds__(kThreads) void analytic_widen(const float* __restrict__ x, float2* __restrict__ z, int ns) {
    const size_t base = (size_t)blockIdx.y * ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x)
        z[base + i] = make_float2(x[base + i], 0.f);
}

// Z[row][q1][i] holds frequency k1[q1] + N1 k2[i]: x 1 at f = 0, x 2 for 0 < f <= (ns - 1) / 2, x 0 above (odd ns)
__global__ __launch_bounds__(kThreads) void analytic_onesided(float2* __restrict__ z, int ns, int N1, int N2,
                                                               const int* __restrict__ k1, const int* __restrict__ k2) {
    const size_t base = (size_t)blockIdx.y * ns;
    const int half = (ns - 1) / 2;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < ns; p += gridDim.x * blockDim.x) {
        const int q1 = p / N2, i = p - q1 * N2;
        const int f = k1[q1] + N1 * k2[i];
        const float g = (f == 0) ? 1.f : (f <= half ? 2.f : 0.f);
        z[base + p] = c_scale(z[base + p], g);
    }
}

__global__ __launch_bounds__(kThreads) void analytic_combine_z(const float2* __restrict__ z, float* __restrict__ y, int ns,
                                                                int mode, const float* __restrict__ var, float fscale) {
    const size_t base = (size_t)blockIdx.y * ns;
    const int nout = (mode == 3) ? ns - 1 : ns;
    float* yr = y + (size_t)blockIdx.y * nout;
    const float inv_var = (mode == 2 || mode == 4) ? 1.0f / var[blockIdx.y] : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += gridDim.x * blockDim.x) {
        const float2 a = z[base + i];
        float v;
        if (mode == 0) v = sqrtf(fmaf(a.x, a.x, a.y * a.y));
        else if (mode == 1) v = a.y;
        else if (mode == 2) v = 10.0f * log10f(fmaf(a.x, a.x, a.y * a.y) * inv_var);
        else if (mode == 4) v = sqrtf(fmaf(a.x, a.x, a.y * a.y) * inv_var);
        else {
            const float2 p = c_mulc(z[base + i + 1], a);
            v = atan2f(p.y, p.x) * fscale;
        }
        yr[i] = v;
    }
}
}  // namespace d4w

static std::mutex g_long_mu;
static std::map<std::tuple<int, int, int>, d4w_fkd_plan*> g_long_plans;
static std::map<std::tuple<int, int, int>, d4w_fkd_plan*> g_long_fast;     // packed plans of specialised shapes (successes only)
constexpr size_t kLongPlansMax = 16;   // a stream of varying channel selections must not keep adding device tables

// drop every cached long-row plan (g_long_mu held): kernels still in flight may read their tables, so the device drains first
static void long_plans_drop_locked() {
    (void)hipDeviceSynchronize();
    for (auto& kv : g_long_plans) if (kv.second) d4w_fkd_plan_destroy(kv.second);
    for (auto& kv : g_long_fast) if (kv.second) d4w_fkd_plan_destroy(kv.second);
    g_long_plans.clear();
    g_long_fast.clear();
}

extern "C" {

/* Frees the plans d4w_analytic_long_f32 caches per (device, nx, ns) (synchronises the device).  They are also dropped
 * automatically once more than 16 shapes have been seen. */
int d4w_analytic_long_clear(void) {
    std::lock_guard<std::mutex> lk(g_long_mu);
    long_plans_drop_locked();
    return D4W_OK;
}

/* even ns: [nx][ns] float32 (the Hilbert transform); odd ns: [nx][ns] complex (the row as a complex sequence) */
size_t d4w_analytic_long_ws_bytes(int nx, int ns) {
    return (nx > 0 && ns > 0) ? (size_t)nx * ns * sizeof(float) * ((ns & 1) ? 2 : 1) : 0;
}

int d4w_analytic_long_f32(const float* x, float* y, int nx, int ns, int mode, const float* var, double fs,
                          void* ws, void* stream) {
    if (!x || !y || !ws || nx < 1 || ns < 2) return fail(D4W_EINVAL, "bad argument");
    if (mode < 0 || mode > 4) return fail(D4W_EINVAL, "mode = %d not in 0..4", mode);
    if ((mode == 2 || mode == 4) && !var) return fail(D4W_EINVAL, "modes 2 and 4 need the row variances");
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    float* h = (float*)ws;
    if (ns & 1) {
        // odd rows: "packed rows" of 2 ns floats = complex rows of length ns through the time phase of the generic plan
        // (any length: a prime factor > 31 of ns runs the global-memory Bluestein form)
        d4w_fkd_plan* pl = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_long_mu);
            auto key = std::make_tuple(devid, nx, -ns);
            auto it = g_long_plans.find(key);
            if (it == g_long_plans.end()) {
                if (g_long_plans.size() + g_long_fast.size() >= kLongPlansMax) long_plans_drop_locked();
                int rc = fkd_plan_build(nx, 2 * ns, 1, 0, false, &pl);
                if (rc) return rc;
                pl->tp.dev.scale = (float)(1.0 / (double)ns);
                g_long_plans[key] = pl;
            } else {
                pl = it->second;
            }
        }
        float2* z = reinterpret_cast<float2*>(ws);
        const dim3 grid(std::min(ceil_div(ns, kThreads), 128), nx);
        D4W_LAUNCH(analytic_widen, grid, dim3(kThreads), 0, stream, x, z, ns);
        int rc = d4w_fkd_time_fwd_f32(pl, h, h, 0, stream);
        if (rc) return rc;
        D4W_LAUNCH(analytic_onesided, grid, dim3(kThreads), 0, stream, z, ns, pl->N1, pl->N2, (const int*)pl->d_k1, (const int*)pl->d_k2);
        if ((rc = d4w_fkd_time_inv_f32(pl, h, stream))) return rc;
        D4W_LAUNCH(analytic_combine_z, grid, dim3(kThreads), 0, stream, (const float2*)z, y, ns, mode, var, (float)(fs / (2.0 * M_PI)));
        return D4W_OK;
    }
    // Shapes with specialised f-k kernels: the time phase of the packed plan (pass A MODE 1: n1 transform of the real rows),
    // pass B with the Hilbert pair operation on the work list "sub-row q1 of a row pairs with sub-row N1 - q1 of the SAME
    // row", the inverse time phase in place, the combine pass -- 36 B per sample on the fast kernels instead of 52 through
    // the generic ones.  D4W_LONG_FAST=0 keeps the generic path.
    {
        static const int fast_env = [] { const char* v = getenv("D4W_LONG_FAST"); return v ? atoi(v) : 1; }();
        d4w_fkd_plan* fp = nullptr;
        if (fast_env) {
            std::lock_guard<std::mutex> lk(g_long_mu);
            auto key = std::make_tuple(devid, nx, ns);
            auto it = g_long_fast.find(key);
            if (it != g_long_fast.end()) {
                fp = it->second;
            } else {
                if (fkd_plan_build_packed(nx, ns, 1, 0, false, &fp, true) == D4W_OK) {
                    const int N1 = fp->N1;
                    std::vector<int2> ps;
                    ps.reserve((size_t)nx * (N1 / 2 + 1));
                    for (int r = 0; r < nx; ++r)
                        for (int q1 = 0; q1 <= N1 / 2; ++q1)       // single-radix n1: position = frequency digit
                            ps.push_back(make_int2(r * N1 + q1, r * N1 + (N1 - q1) % N1));
                    void* q = nullptr;
                    if (hipMalloc(&q, ps.size() * sizeof(int2)) != hipSuccess ||
                        hipMemcpy(q, ps.data(), ps.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) {
                        if (q) (void)hipFree(q);
                        d4w_fkd_plan_destroy(fp);
                        fp = nullptr;
                    } else {
                        fp->sp->allocs.push_back(q);
                        fp->d_pairs_self = (int2*)q;
                        fp->npairs_self = (int)ps.size();
                    }
                } else {
                    fp = nullptr;
                }
                // only successes are remembered: a shape registered later (d4w_fk_register_shape, fkjit) is picked up by
                // the next call; the failed attempt costs a table lookup
                if (fp) {
                    if (g_long_plans.size() + g_long_fast.size() >= kLongPlansMax) long_plans_drop_locked();
                    g_long_fast[key] = fp;
                }
            }
        }
        if (fp) {
            const FkFastEntry& F = *fp->sp->fast;
            const FkDims& d = fp->sp->dev.d;
            const int NBX = d.N2 / d.TA, nt = ceil_div(nx, d.C1) * NBX;
            FkGeo geo = fp->geo_t;
            geo.nrows = nx;
            FkDev dv = fp->dev_t;
            dv.scale = (float)(1.0 / (double)d.M);               // forward / inverse pair of the packed rows
            dv.mask = nullptr;
            dv.nyq = nullptr;
            float2* h2 = reinterpret_cast<float2*>(h);
            const dim3 gA(std::min(nt, fp->num_cu * fp->sp->wgA));
            int rc = launch_k(F.T_fwd, gA, dim3(F.thrA), F.ldsA, stream, fp->dev_t, reinterpret_cast<const float2*>(x), h2, 0, nt, NBX, 0, geo);
            if (rc) return rc;
            FkFastDev fd = fp->fdev_c;
            fd.pairs = fp->d_pairs_self;
            fd.live = nullptr;
            FkGeo gb = fp->geo_c;
            gb.nq = d.N1;
            const dim3 gB(std::max(1, std::min(fp->npairs_self, fp->num_cu * fp->sp->wgB)));
            if ((rc = launch_k(F.Bs_hilb, gB, dim3(F.thrB), F.ldsB, stream, dv, fd, h2, 0, fp->npairs_self, gb))) return rc;
            static const int fuse_env = [] { const char* v = getenv("D4W_LONG_FUSE"); return v ? atoi(v) : 1; }();
            if (mode == 1)                                         // H[x] itself: the inverse pass writes it where it belongs
                return launch_k(F.T_inv, gA, dim3(F.thrA), F.ldsA, stream, dv, reinterpret_cast<float2*>(y), 0, nt, NBX, 0, geo, (const float2*)h2);
            if (mode != 3 && fuse_env)                             // |z|, SNR, |z| / std: formed in the inverse pass's epilogue (28 B / sample)
                return launch_k(F.T_inv_env, gA, dim3(F.thrA), F.ldsA, stream, dv, reinterpret_cast<float2*>(y), 0, nt, NBX, 0, geo,
                                (const float2*)h2, reinterpret_cast<const float2*>(x), mode, var);
            if ((rc = launch_k(F.T_inv, gA, dim3(F.thrA), F.ldsA, stream, dv, h2, 0, nt, NBX, 0, geo, (const float2*)h2))) return rc;
            const float fscale = (float)(fs / (2.0 * M_PI));
            D4W_LAUNCH(analytic_combine, dim3(std::min(ceil_div(ns, kThreads), 128), nx), dim3(kThreads), 0, stream, x,
                       (const float*)h, y, ns, mode, var, fscale);
            return D4W_OK;
        }
    }
    d4w_fkd_plan* pl = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_long_mu);
        auto key = std::make_tuple(devid, nx, ns);
        auto it = g_long_plans.find(key);
        if (it == g_long_plans.end()) {
            if (g_long_plans.size() + g_long_fast.size() >= kLongPlansMax) long_plans_drop_locked();
            int rc = fkd_plan_build(nx, ns, 1, 0, false, &pl);
            if (rc) return rc;
            pl->slab.hilbert = 1;
            pl->tp.dev.scale = (float)(1.0 / (double)pl->M);     // forward / inverse pair of the packed rows
            g_long_plans[key] = pl;
        } else {
            pl = it->second;
        }
    }
    int rc = d4w_fkd_time_fwd_f32(pl, x, h, 0, stream);
    if (rc) return rc;
    if ((rc = launch_k(fkd_pair_slab, dim3(std::min(ceil_div(pl->N2, kThreads), 8), nx, pl->N1), dim3(kThreads), 0, stream,
                       pl->slab, reinterpret_cast<float2*>(h)))) return rc;
    if ((rc = d4w_fkd_time_inv_f32(pl, h, stream))) return rc;
    const float fscale = (float)(fs / (2.0 * M_PI));
    D4W_LAUNCH(analytic_combine, dim3(std::min(ceil_div(ns, kThreads), 128), nx), dim3(kThreads), 0, stream, x,
               (const float*)h, y, ns, mode, var, fscale);
    return D4W_OK;
}

}  // extern "C"
