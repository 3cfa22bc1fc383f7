// Row-independent time-axis operators on MI355X (gfx950): the zero-phase IIR filter
// (dsp.bp_filt / scipy.signal.sosfiltfilt on dsp.butterworth_filter designs, reference
// dsp.py:789-827,859-880) and the matched filter (detect.compute_cross_correlogram /
// detect.shift_xcorr, reference detect.py:96-166).  See DESIGN.md "band-pass" and
// "matched filter" for the data layout and the roofline that bounds each kernel.
#include <algorithm>
#include <cstdlib>

#include "fft_radix.h"     // static_for (and d4w_internal.h)

namespace d4w {

// =============================================================================================
// zero-phase SOS cascade
//
// One wave = 64 rows x one time segment.  Lanes own rows; the recursion walks along time, so
// the [64 rows][32 samples] chunk is staged through LDS: coalesced 128-byte row pieces on the
// global side, conflict-free ds_read_b128 / ds_write_b128 per lane on the LDS side (row pitch
// 36 floats: 9*l mod 16 is a bijection on every 16-lane b128 group).  Segments of one row run
// concurrently; a segment that does not start at the row edge is warmed up over W samples.
// =============================================================================================
constexpr int kSosRows = 64;     // rows per wave (one per lane)
constexpr int kSosChunk = 32;    // time samples staged per LDS round trip
constexpr int kSosPitch = kSosChunk + 4;
constexpr int kSosMaxSec = 10;

template <typename T>
struct SosCoefT {
    T b0, b1, b2, a1, a2, z1, z2, pad;
};
template <typename T>
struct SosArgsT {
    SosCoefT<T> s[kSosMaxSec];
};
typedef SosCoefT<float> SosCoef;
typedef SosArgsT<float> SosArgs;

// fused multiply-add in the recursion's precision
__device__ __forceinline__ float sos_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double sos_fma(double a, double b, double c) { return fma(a, b, c); }

// forward pass input: scipy odd extension of the row, virtual index i in [-padlen, ns + padlen)
__device__ __forceinline__ float sos_fetch_fwd(const float* __restrict__ row, int ns, int i) {
    if (i < 0) return 2.0f * row[0] - row[-i];
    if (i >= ns) return 2.0f * row[ns - 1] - row[2 * (ns - 1) - i];
    return row[i];
}
// backward pass input: forward output, continued over the right extension by the edge buffer
__device__ __forceinline__ float sos_fetch_bwd(const float* __restrict__ row, const float* __restrict__ edge,
                                               int ns, int i) {
    return (i >= ns) ? edge[i - ns] : row[i];
}

// dst[r][0..ncols) = src[r][0..ncols) for nrows rows with their own pitches: the row-end pieces of the band-pass move in one
// launch each (torch splits a slice copy of a tensor with more than 2^31 elements into ~10 launches of a few microseconds)
__global__ __launch_bounds__(256) void copy_cols(const float* __restrict__ src, size_t ld_src, float* __restrict__ dst,
                                                 size_t ld_dst, int ncols) {
    const float* s = src + (size_t)blockIdx.y * ld_src;
    float* d = dst + (size_t)blockIdx.y * ld_dst;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) d[c] = s[c];
}

__global__ __launch_bounds__(256) void sos_first_samples(const float* __restrict__ x, int nx, int ns,
                                                         float* __restrict__ first) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nx) first[r] = x[(size_t)r * ns];
}

// T = float for well-conditioned cascades; T = double (states and arithmetic, float32 I/O) when poles
// sit close to the unit circle: the float32 recursion's rounding noise grows like 1 / (1 - |pole|)^2
// (5 Hz band edge at 200 Hz: 2e-5 of the output, above the 1e-5 budget; 14-30 Hz: 1.5e-6)
template <int NSEC, bool REV, typename T>
__global__ __launch_bounds__(kSosRows) void sos_pass(SosArgsT<T> A, const float* __restrict__ src,
                                                     const float* __restrict__ edge_in,
                                                     float* __restrict__ dst, float* __restrict__ edge_out,
                                                     int nx, int ns, int padlen, int S, int W,
                                                     const float* __restrict__ xorig /* [nx] first samples */,
                                                     float dc_gain2) {
    __shared__ __attribute__((aligned(16))) float tile[kSosRows * kSosPitch];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * kSosRows;
    const int a = blockIdx.y * S;
    const int b = min(a + S, ns);
    int i_start, count;
    if (!REV) {
        const bool exact = (a - W <= 0);
        i_start = exact ? -padlen : a - W;
        const int i_end = (b == ns) ? ns + padlen : b;
        count = i_end - i_start;
    } else {
        const bool exact = (b + W >= ns);
        i_start = exact ? ns + padlen - 1 : b + W - 1;
        count = i_start - a + 1;
    }
    const int my_row = min(row0 + lane, nx - 1);
    const float* my_src = src + (size_t)my_row * ns;
    // The row's first sample c is taken out before the forward recursion and c |H(1)|^2 put back after
    // the backward one: filtfilt of a constant is exactly that constant times the squared DC gain (odd
    // extension and steady-state initial conditions keep a constant in steady state), and a large offset
    // in front of a band-pass would otherwise sit in the float32 states and drown the output's low bits.
    const float c_row = xorig[my_row];
    const float c_sub = REV ? 0.f : c_row, c_add = REV ? c_row * dc_gain2 : 0.f;
    const float x0 = (REV ? sos_fetch_bwd(my_src, edge_in + (size_t)my_row * padlen, ns, i_start)
                          : sos_fetch_fwd(my_src, ns, i_start)) - c_sub;
    T s1[NSEC], s2[NSEC];
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        s1[s] = A.s[s].z1 * (T)x0;
        s2[s] = A.s[s].z2 * (T)x0;
    }
    // A chunk is INTERIOR when its 32 samples lie inside the row proper (no extension, no edge buffer)
    // and inside this segment's walk: its loads are then branch-free, all 32 of a lane in flight at
    // once, and issued one chunk ahead (the branchy edge fetch serialises its loads and is kept for
    // the few chunks that touch a row end).
    const bool full_rows = (row0 + kSosRows <= nx);
    auto interior = [&](int m0) {
        if (m0 + kSosChunk > count) return false;
        const int lo = REV ? i_start - m0 - (kSosChunk - 1) : i_start + m0;
        return lo >= 0 && lo + kSosChunk <= ns;
    };
    // Interior chunks address memory as (wave-uniform 64-bit base of the row group) + (32-bit lane offset): element
    // e = lane + 64 k sits in row (lane >> 5) + 2 k at sample lane & 31, so nothing but the row term changes with k
    // and no table of 64-bit addresses has to live in registers.
    static_assert(kSosChunk == 32 && kSosRows == 64, "lane -> (row, sample) map of the interior chunks");
    const float* src_grp = src + (size_t)row0 * ns;
    float* dst_grp = dst + (size_t)row0 * ns;
    const unsigned rl0 = (unsigned)(lane >> 5), ml0 = (unsigned)(lane & 31), rmax = (unsigned)(nx - 1 - row0);
    float pre[kSosChunk];
    auto issue = [&](int m0) {
        const unsigned i = (unsigned)(REV ? i_start - (m0 + (int)ml0) : i_start + m0 + (int)ml0);
#pragma unroll
        for (int k = 0; k < kSosChunk; ++k) {
            const unsigned rl = min(rl0 + 2u * k, rmax);
            pre[k] = src_grp[rl * (unsigned)ns + i];
        }
    };
    int pre_for = -1;
    if (count > 0 && interior(0)) { issue(0); pre_for = 0; }
    for (int m0 = 0; m0 < count; m0 += kSosChunk) {
        // ---- stage in: lanes walk time within a row (coalesced), two rows per wave instruction
        if (pre_for == m0) {
#pragma unroll
            for (int k = 0; k < kSosChunk; ++k) {
                const int e = lane + k * kSosRows;
                tile[(e / kSosChunk) * kSosPitch + (e % kSosChunk)] = pre[k];
            }
        } else {
#pragma unroll 4
            for (int e = lane; e < kSosRows * kSosChunk; e += kSosRows) {
                const int rl = e / kSosChunk, ml = e % kSosChunk;
                const int m = m0 + ml;
                const int row = min(row0 + rl, nx - 1);
                float v = 0.f;
                if (m < count) {
                    const int i = REV ? i_start - m : i_start + m;
                    const float* r = src + (size_t)row * ns;
                    v = REV ? sos_fetch_bwd(r, edge_in + (size_t)row * padlen, ns, i) : sos_fetch_fwd(r, ns, i);
                }
                tile[rl * kSosPitch + ml] = v;
            }
        }
        lds_barrier();          // LDS only: __syncthreads() would also drain the prefetch loads and the stores
        const int nextm = m0 + kSosChunk;
        if (nextm < count && interior(nextm)) { issue(nextm); pre_for = nextm; }
        // ---- recursion: lane = row, 4 samples per LDS access
        float4* mine = reinterpret_cast<float4*>(tile + lane * kSosPitch);
#pragma unroll 2
        for (int g = 0; g < kSosChunk / 4; ++g) {
            float4 v4 = mine[g];
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T x = (T)(v[j] - c_sub);
#pragma unroll
                for (int s = 0; s < NSEC; ++s) {
                    const SosCoefT<T>& c = A.s[s];
                    const T y = sos_fma(c.b0, x, s1[s]);
                    s1[s] = sos_fma(c.b1, x, sos_fma(-c.a1, y, s2[s]));
                    s2[s] = sos_fma(c.b2, x, -c.a2 * y);
                    x = y;
                }
                v[j] = (float)x + c_add;
            }
            mine[g] = make_float4(v[0], v[1], v[2], v[3]);
        }
        lds_barrier();
        // ---- stage out (only this segment's own samples; the last forward segment also feeds
        //      the right-extension outputs to the edge buffer for the backward pass)
        const int lo = REV ? i_start - m0 - (kSosChunk - 1) : i_start + m0;
        if (full_rows && m0 + kSosChunk <= count && lo >= a && lo + kSosChunk <= b) {
            const unsigned i = (unsigned)(REV ? i_start - (m0 + (int)ml0) : i_start + m0 + (int)ml0);
#pragma unroll
            for (int k = 0; k < kSosChunk; ++k) {
                const unsigned rl = rl0 + 2u * k;
                dst_grp[rl * (unsigned)ns + i] = tile[rl * kSosPitch + ml0];
            }
        } else {
#pragma unroll 4
            for (int e = lane; e < kSosRows * kSosChunk; e += kSosRows) {
                const int rl = e / kSosChunk, ml = e % kSosChunk;
                const int m = m0 + ml;
                const int row = row0 + rl;
                if (m < count && row < nx) {
                    const int i = REV ? i_start - m : i_start + m;
                    const float v = tile[rl * kSosPitch + ml];
                    if (i >= a && i < b) dst[(size_t)row * ns + i] = v;
                    else if (!REV && i >= ns) edge_out[(size_t)row * padlen + (i - ns)] = v;
                }
            }
        }
        lds_barrier();          // the chunk's stores stay in flight under the next chunk
    }
}

// =============================================================================================
// The same cascade with the SECTIONS of a row spread over adjacent lanes (a systolic line): lane (row rr, section s) of a wave
// runs section s on sample t - s at step t and hands its output to lane s + 1 through one DPP move, so a wave advances
// 64 / G rows (G = 8 lanes per row; 16 for 9-10 sections) by one sample per step with ONE biquad per lane instead of the
// whole cascade per lane.  sos_pass gives a lane a row: rows of a few hundred samples -- the two row-end pieces the
// overlap-save band-pass leaves to the exact recursion (2 nx pieces of 2 E = 1204 samples: 345 one-wave workgroups on 256
// compute units, 0.53 ms per direction at 11 020 rows) -- then run ~1.3 waves per compute unit through a chain of 24 dependent
// operations per sample.  Here the same pieces make 8 x the waves and a step is 4 dependent operations.  Whole rows only (one
// exact segment: odd extension at both ends, steady-state initial conditions, the right-extension outputs handed to the
// backward pass), rows x sections = the lanes of a wave.  Input / output pass through LDS in chunks of 64 samples per row
// (coalesced 256-byte row pieces on the global side); the four waves of a workgroup are independent.
// =============================================================================================
constexpr int kSlThreads = 256;
constexpr int kSlCh = 64;

__device__ __forceinline__ void sl_wave_sync() {
#ifdef D4W_EMU
    (void)__shfl_xor(0, 1);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// The input of a lane's section at one step: the output `v` of the section below -- S lanes down inside the 16-lane DPP row
// -- or, for the section-0 lanes (the first S lanes of a DPP row, which have no lane S below), the row's next sample `fresh`:
// ONE v_mov_b32_dpp whose unwritten lanes keep the old value of the destination.
template <int S>
__device__ __forceinline__ float sl_input(float fresh, float v, bool sec0) {
#ifdef D4W_EMU
    const float below = __shfl_up(v, (unsigned)S);
    return sec0 ? fresh : below;
#else
    (void)sec0;
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fresh), __float_as_int(v), 0x110 + S /* row_shr:S */, 0xf, 0xf, false));
#endif
}
template <int S>
__device__ __forceinline__ double sl_input(double fresh, double v, bool sec0) {
#ifdef D4W_EMU
    const double below = __shfl_up(v, (unsigned)S);
    return sec0 ? fresh : below;
#else
    (void)sec0;
    const long long f = __double_as_longlong(fresh), b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(f & 0xFFFFFFFFll), (int)(b & 0xFFFFFFFFll), 0x110 + S, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), 0x110 + S, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
#endif
}

// Where the rows of a pass live: row r < nsplit starts at base + r ld, row r >= nsplit at base + (r - nsplit) ld + off2 --
// whole rows of a block (nsplit = all rows), or the LEFT pieces of all rows followed by their RIGHT pieces read / written in
// place in the block (d4w_sosfiltfilt_ends_f32: no gathered copy of the pieces, no scattered copy of the results).
struct SlRows {
    const float* base;
    size_t ld;
    long long off2;
    int nsplit;
    __device__ __forceinline__ const float* row(int r) const {
        return (r < nsplit) ? base + (size_t)r * ld : base + (size_t)(r - nsplit) * ld + off2;
    }
};

template <int G, bool REV, typename T>
__global__ __launch_bounds__(kSlThreads) void sos_pass_lanes(SosArgsT<T> A, int nsec, SlRows src, const float* __restrict__ edge_in,
                                                             SlRows dstr, int keep_a0, int keep_a1, int keep_b0, int keep_b1,
                                                             float* __restrict__ edge_out, int nx, int ns, int padlen,
                                                             SlRows piv, float dc_gain2) {
    // src: the pass's input rows of ns samples (forward: the data; backward: the forward outputs); dstr: where outputs
    // go, of which rows < nsplit keep the samples [keep_a0, keep_a1) and the others [keep_b0, keep_b1); piv: the ORIGINAL rows
    // (their first sample is taken out before the forward recursion and put back, times the squared DC gain, after the
    // backward one, as in sos_pass)
    constexpr int R = 64 / G;                                   // rows per wave
    constexpr int SH = 16 / G;                                  // rows interleaved inside a 16-lane DPP row = the shift to the section below
    // row pitches that spread the rows of a wave over the LDS banks: the last sections of the R rows write the same ring slot
    // of their rows in one instruction (a pitch of 256 floats put all of them on one bank: 75 % of the LDS cycles of the first
    // build were bank conflicts, profiles/r05d/pmc_bp), and the R rows' 16-byte input reads start 68 floats apart
    constexpr int kInP = kSlCh + 4, kRingP = 4 * kSlCh + 1;
    __shared__ __attribute__((aligned(16))) float lin[kSlThreads / 64][R][kInP];
    __shared__ float lout[kSlThreads / 64][R][kRingP];          // ring of outputs per row (four chunks), indexed by STEP
    __shared__ float sink[kSlThreads / 64][5 * kSlCh + 64];     // where the lanes of the other sections write (a slot per lane)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // lane -> (row, section): the sections of a row sit SH lanes apart inside one DPP row (G = 8: two rows interleaved), so
    // that "the section below" is row_shr:SH and the lanes without one are exactly the section-0 lanes
    const int rr = (lane >> 4) * SH + (lane & (SH - 1)), sec = (lane & 15) / SH;
    const int row0 = (blockIdx.x * (kSlThreads / 64) + wv) * R;
    if (row0 >= nx) return;                                     // whole wave (no workgroup barrier below)
    const int my_row = min(row0 + rr, nx - 1);
    // this lane's section (static indices into the kernel argument: a per-lane index would go through scratch memory)
    T b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0, z1 = 0, z2 = 0;
#pragma unroll
    for (int q = 0; q < kSosMaxSec; ++q)
        if (q == sec) { b0 = A.s[q].b0; b1 = A.s[q].b1; b2 = A.s[q].b2; a1 = A.s[q].a1; a2 = A.s[q].a2; z1 = A.s[q].z1; z2 = A.s[q].z2; }
    const int i_start = REV ? ns + padlen - 1 : -padlen;
    const int count = REV ? ns + padlen : ns + 2 * padlen;
    auto fetch = [&](int row, int m) -> float {                 // input sample m of `row` (virtual index i_start +- m)
        const int i = REV ? i_start - m : i_start + m;
        const float* r = src.row(row);
        return REV ? sos_fetch_bwd(r, edge_in + (size_t)row * padlen, ns, i) : sos_fetch_fwd(r, ns, i);
    };
    const float c_mine = piv.row(my_row)[0];
    const float x0 = fetch(my_row, 0) - (REV ? 0.f : c_mine);
    T s1 = z1 * (T)x0, s2 = z2 * (T)x0;
    T y = 0;
    const bool last_sec = (sec == nsec - 1);
    const int nchunks = (count + nsec - 1 + kSlCh - 1) / kSlCh;
    float pv[R];                                                // the pivots of the wave's rows
#pragma unroll
    for (int k = 0; k < R; ++k) pv[k] = piv.row(min(row0 + k, nx - 1))[0];
    auto flush = [&](int c) {                                   // outputs m in [64 c, 64 c + 64) of the wave's rows
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int m = c * kSlCh + lane, row = row0 + k;
            if (m < count && row < nx) {
                const int i = REV ? i_start - m : i_start + m;
                // the last section produced sample m at step m + nsec - 1; the pivot goes back in here (backward pass)
                const float v = lout[wv][k][(m + nsec - 1) & (4 * kSlCh - 1)] + (REV ? pv[k] * dc_gain2 : 0.f);
                const bool ta = row < dstr.nsplit;
                if (i >= (ta ? keep_a0 : keep_b0) && i < (ta ? keep_a1 : keep_b1)) const_cast<float*>(dstr.row(row))[i] = v;
                else if (!REV && i >= ns) edge_out[(size_t)row * padlen + (i - ns)] = v;
            }
        }
    };
    // chunk c + 1 is loaded while chunk c runs its 64 steps (the loads' latency, ~2 us under load, would otherwise stand in
    // front of every 64 steps of ~1 us)
    float pre[R];
    auto load_chunk = [&](int c) {                              // lane = sample of the chunk, one coalesced 256-byte piece per row
        const int m = c * kSlCh + lane;
        const int lo = REV ? i_start - (c * kSlCh + kSlCh - 1) : i_start + c * kSlCh;       // the chunk's index range
        if (lo >= 0 && lo + kSlCh <= ns && c * kSlCh + kSlCh <= count) {                    // inside the rows: plain loads
            const int i = REV ? i_start - m : i_start + m;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int row = min(row0 + k, nx - 1);
                pre[k] = src.row(row)[i] - (REV ? 0.f : pv[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int row = min(row0 + k, nx - 1);
                pre[k] = (m < count) ? fetch(row, m) - (REV ? 0.f : pv[k]) : 0.f;
            }
        }
    };
    if (nchunks > 0) load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        const int m0 = c * kSlCh;
        sl_wave_sync();                                         // the previous chunk's reads of lin are done
#pragma unroll
        for (int k = 0; k < R; ++k) lin[wv][k][lane] = pre[k];
        sl_wave_sync();
        // the row's 64 inputs of this chunk in registers (every lane of a row reads the same addresses: broadcasts), so that
        // no LDS round trip sits between two steps of the recursion
        float in[kSlCh];
#pragma unroll
        for (int q = 0; q < kSlCh / 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(&lin[wv][rr][0])[q];
            in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
        }
        // A wave's vector-memory operations retire in order: stores issued AFTER the next chunk's loads would have to be
        // acknowledged before those loads can be waited for with a count the compiler can know (the stores are conditional:
        // it waits for everything).  So the outputs go out two chunks late, BEFORE the next loads are issued -- by the time
        // those loads are waited for, a chunk of steps later, the stores in front of them are long done.
        if (c > 1) flush(c - 2);
        if (c + 1 < nchunks) load_chunk(c + 1);                 // in flight under the steps below
        // ---- 64 steps: section `sec` works on sample t - sec.  Only the first G steps of a row need a predicate (a section
        // must keep its initial state until its first sample arrives); past the row's end the sections run on, their
        // outputs land in slots of the ring that are never flushed
        // the last section's lanes write the row's ring, every other lane a sink slot of its own: no exec-mask switch per step,
        // and the slot of step t = m0 + j is a constant offset from a per-chunk base
        float* ring = (last_sec ? &lout[wv][rr][0] : &sink[wv][lane]) + (m0 & (3 * kSlCh));
        auto step = [&](int j, bool guarded) {
            const T xin = sl_input<SH>((T)in[j], y, sec == 0);
            if (!guarded || m0 + j - sec >= 0) {
                y = sos_fma(b0, xin, s1);
                s1 = sos_fma(b1, xin, sos_fma(-a1, y, s2));
                s2 = sos_fma(b2, xin, -a2 * y);
            }
            ring[j] = (float)y;
        };
        if (c == 0) {
            static_for<kSlCh>([&](auto jj) { constexpr int j = decltype(jj)::value; step(j, j < G); });
        } else {
            static_for<kSlCh>([&](auto jj) { constexpr int j = decltype(jj)::value; step(j, false); });
        }
    }
    sl_wave_sync();
    if (nchunks > 1) flush(nchunks - 2);
    if (nchunks > 0 && (nchunks - 1) * kSlCh < count) flush(nchunks - 1);
}

// =============================================================================================
// per-row mean and max|x|   (detect.py:157: (x - mean) / max|x|, max of the un-de-meaned row)
// one workgroup per row, wavefront shuffle reduction, then one LDS hop across the four waves.
// The mean is returned as a FLOAT64 (the reference de-means in float64): the sums run over x - pivot
// (pivot = the row's first sample, so a row that is all offset sums its small deviations, each exact or
// rounded relative to the DEVIATION) in float32 lanes and are put together in float64; consumers split
// the value into a two-float (hi, lo) pair (d4w_internal.h, Mean2).
// =============================================================================================
constexpr int kStatThreads = 256;

// Four consecutive de-meaned samples of a row piece (zeros past the end); one 16-byte load where the address allows.
__device__ __forceinline__ void tail_load4(const float* __restrict__ row, int j, int ns, bool vec, Mean2 m, float (&v)[4]) {
    if (vec && j + 3 < ns) {
        const float4 q = *reinterpret_cast<const float4*>(row + j);
        v[0] = demean(q.x, m); v[1] = demean(q.y, m); v[2] = demean(q.z, m); v[3] = demean(q.w, m);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (j + k < ns) ? demean(row[j + k], m) : 0.f;
    }
}

// Workgroup-wide exclusive prefix of the threads' sums `loc` (1024 samples per step): returns the sum of all earlier
// threads' values, `total` = the whole step's sum.  Two barriers per call (wsum is reused by the next step).
__device__ __forceinline__ float tail_scan(float loc, float* wsum, float& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float incl = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(incl, off);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    float before = 0.f;
    total = 0.f;
    for (int w = 0; w < kStatThreads / 64; ++w) {
        if (w < wave) before += wsum[w];
        total += wsum[w];
    }
    __syncthreads();
    return before + incl - loc;
}

// max_j |P[j]|, P[j] = sum_{i < j} (x[i] - m), of one row by one workgroup (every thread returns the workgroup's value in
// lane-reduced form: the caller finishes across the waves).  wsum: kStatThreads / 64 floats of LDS.
__device__ __forceinline__ float prefix_max_sweep(const float* __restrict__ row, int ns, Mean2 m, float* wsum) {
    const bool vec = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    const int tid = threadIdx.x;
    double carry = 0.0;
    float best = 0.f;
    for (int j0 = 0; j0 < ns; j0 += 4 * kStatThreads) {
        float v[4], total;
        tail_load4(row, j0 + 4 * tid, ns, vec, m, v);
        float pre = (float)carry + tail_scan((v[0] + v[1]) + (v[2] + v[3]), wsum, total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pre += v[k];                                     // P[j0 + 4 tid + k + 1] (samples past the end add zero)
            best = fmaxf(best, fabsf(pre));
        }
        carry += (double)total;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    return best;
}


__global__ __launch_bounds__(kStatThreads) void row_stats(const float* __restrict__ x, int ns,
                                                          double* __restrict__ mean, float* __restrict__ maxabs,
                                                          float* __restrict__ pmax) {
    __shared__ double red_s[kStatThreads / 64];
    __shared__ float red_m[kStatThreads / 64];
    __shared__ double s_mean;
    const float* row = x + (size_t)blockIdx.x * ns;
    const float pv = row[0];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, mx = 0.f;
    double sd = 0.0;                                        // the lane's float32 partial sums, folded every 64 terms
    if ((reinterpret_cast<uintptr_t>(row) & 15) == 0 && (ns & 3) == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        const int n4 = ns >> 2;
        int i = threadIdx.x, it = 0;
        for (; i + 3 * kStatThreads < n4; i += 4 * kStatThreads) {
            const float4 a = r4[i], b = r4[i + kStatThreads], c = r4[i + 2 * kStatThreads], d = r4[i + 3 * kStatThreads];
            s0 += ((a.x - pv) + (a.y - pv)) + ((a.z - pv) + (a.w - pv));
            s1 += ((b.x - pv) + (b.y - pv)) + ((b.z - pv) + (b.w - pv));
            s2 += ((c.x - pv) + (c.y - pv)) + ((c.z - pv) + (c.w - pv));
            s3 += ((d.x - pv) + (d.y - pv)) + ((d.z - pv) + (d.w - pv));
            mx = fmaxf(fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)))),
                       fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
            mx = fmaxf(fmaxf(mx, fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)))),
                       fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
            if ((++it & 15) == 0) {
                sd += (double)((s0 + s1) + (s2 + s3));
                s0 = s1 = s2 = s3 = 0.f;
            }
        }
        for (; i < n4; i += kStatThreads) {
            const float4 a = r4[i];
            s0 += ((a.x - pv) + (a.y - pv)) + ((a.z - pv) + (a.w - pv));
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
    } else {
        int i = threadIdx.x, it = 0;
        for (; i + 3 * kStatThreads < ns; i += 4 * kStatThreads) {
            const float v0 = row[i], v1 = row[i + kStatThreads], v2 = row[i + 2 * kStatThreads],
                        v3 = row[i + 3 * kStatThreads];
            s0 += v0 - pv; s1 += v1 - pv; s2 += v2 - pv; s3 += v3 - pv;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
            if ((++it & 63) == 0) {
                sd += (double)((s0 + s1) + (s2 + s3));
                s0 = s1 = s2 = s3 = 0.f;
            }
        }
        for (; i < ns; i += kStatThreads) {
            const float v = row[i];
            s0 += v - pv;
            mx = fmaxf(mx, fabsf(v));
        }
    }
    double s = sd + (double)((s0 + s1) + (s2 + s3));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) {
        red_s[wave] = s;
        red_m[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0;
        float tm = 0.f;
        for (int w = 0; w < kStatThreads / 64; ++w) {
            ts += red_s[w];
            tm = fmaxf(tm, red_m[w]);
        }
        s_mean = (double)pv + ts / (double)ns;
        mean[blockIdx.x] = s_mean;
        maxabs[blockIdx.x] = tm;
    }
    if (!pmax) return;
    // the prefix maxima of the de-meaned row (row_prefix_max below) in the same launch: the second sweep of a 60-s row finds
    // it in L2
    __syncthreads();
    const double mu = s_mean;
    Mean2 m;
    m.hi = (float)mu;
    m.lo = (float)(mu - (double)m.hi);
    float best = prefix_max_sweep(row, ns, m, red_m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red_m[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = red_m[0];
        for (int w = 1; w < kStatThreads / 64; ++w) b = fmaxf(b, red_m[w]);
        pmax[blockIdx.x] = b;
    }
}

// =============================================================================================
// fused ingest: data_handle.load_das_data + raw2strain (data_handle.py:157-176,
// 213-214): select channels c0 + r * cstep of the raw [nch][ns] matrix, convert to floating point,
// remove each channel's mean, multiply by scale_factor -- one kernel, the raw row is read from HBM
// once (the second sweep hits L2), float64 accumulation and subtraction, float32 result.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(kStatThreads) void raw2strain_rows(const T* __restrict__ raw, int ns, int c0, int cstep,
                                                                double scale, float* __restrict__ y) {
    __shared__ double red[kStatThreads / 64];
    __shared__ double s_mean;
    const T* row = raw + ((size_t)c0 + (size_t)blockIdx.x * cstep) * ns;
    double s = 0.0;
    for (int i = threadIdx.x; i < ns; i += kStatThreads) s += (double)row[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x / 64] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kStatThreads / 64; ++w) t += red[w];
        s_mean = t / (double)ns;
    }
    __syncthreads();
    const double mu = s_mean;
    float* out = y + (size_t)blockIdx.x * ns;
    for (int i = threadIdx.x; i < ns; i += kStatThreads) out[i] = (float)(((double)row[i] - mu) * scale);
}

// The same with 16-byte loads, a dozen of them in flight per lane: rows of up to kR2sHold x 256 vectors (12 288 four-byte
// samples: the 60-s files) stay in registers between the sum and the subtraction -- one read of the raw row --, longer rows
// are swept twice, a dozen vectors per lane at a time.  Rows must start on 16 bytes (the host checks).
constexpr int kR2sHold = 12;
template <typename T>
__global__ __launch_bounds__(kStatThreads) void raw2strain_rows_vec(const T* __restrict__ raw, int ns, int c0, int cstep,
                                                                    double scale, float* __restrict__ y) {
    constexpr int VEC = 16 / (int)sizeof(T);
    struct alignas(16) Vec { T v[VEC]; };
    __shared__ double red[kStatThreads / 64];
    __shared__ double s_mean;
    const Vec* row = reinterpret_cast<const Vec*>(raw + ((size_t)c0 + (size_t)blockIdx.x * cstep) * ns);
    float* out = y + (size_t)blockIdx.x * ns;
    const int nvec = ns / VEC, tid = threadIdx.x;
    const bool held = nvec <= kR2sHold * kStatThreads;
    Vec q[kR2sHold];
    double s = 0.0;
    for (int v0 = 0; v0 < nvec; v0 += kR2sHold * kStatThreads) {
#pragma unroll
        for (int k = 0; k < kR2sHold; ++k) {
            const int i = v0 + k * kStatThreads + tid;
            if (i < nvec) q[k] = row[i];
        }
#pragma unroll
        for (int k = 0; k < kR2sHold; ++k) {
            const int i = v0 + k * kStatThreads + tid;
            if (i < nvec) {
                double t = 0.0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) t += (double)q[k].v[e];
                s += t;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) red[tid / 64] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kStatThreads / 64; ++w) t += red[w];
        s_mean = t / (double)ns;
    }
    __syncthreads();
    const double mu = s_mean;
    for (int v0 = 0; v0 < nvec; v0 += kR2sHold * kStatThreads) {
        if (!held) {
#pragma unroll
            for (int k = 0; k < kR2sHold; ++k) {
                const int i = v0 + k * kStatThreads + tid;
                if (i < nvec) q[k] = row[i];
            }
        }
#pragma unroll
        for (int k = 0; k < kR2sHold; ++k) {
            const int i = v0 + k * kStatThreads + tid;
            if (i < nvec) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = (float)(((double)q[k].v[e] - mu) * scale);
                if constexpr (VEC >= 4) {
#pragma unroll
                    for (int e = 0; e < VEC; e += 4)
                        *reinterpret_cast<float4*>(out + (size_t)i * VEC + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
                } else {
                    *reinterpret_cast<float2*>(out + (size_t)i * VEC) = make_float2(o[0], o[1]);
                }
            }
        }
    }
}

// =============================================================================================
// DC tail of the de-meaned zero-padded template (detect.py:158): the reference normalises the template
// over its zero-padded length, which leaves the constant -mean(t)/max|t| on the padded part.  Its
// contribution to lag k is  coef * sum_{n >= L, n + k < ns} xh[n + k]  with xh the normalised row; as the
// de-meaned row sums to zero this equals  -coef * P[k + L],  P[j] = sum_{i < j} xh[i]  (zero for
// k + L >= ns), and with coef = -mean(t)/max|t|:   y[k] += (mean(t)/max|t|) * g * sum_{i < k+L} (x[i] - m).
// One workgroup walks a row in order (chunks of 1024 samples, workgroup prefix sum, float64 carry).
// Negligible for the fin-whale templates (|coef| ~ 5e-7) and applied by the host only when it matters.
// =============================================================================================
// max_j |P[j]|, P[j] = sum_{i < j} (x[i] - m): what the DC-tail term of a row can reach, |coef| g max|P| -- the number the
// per-row decision of xcorr_dc_tail is taken on (a band-passed row has prefix sums of a few samples' size, a drifting one of
// ns/4 samples' size: the term's weight is a property of the DATA, not of the template alone).
__global__ __launch_bounds__(kStatThreads) void row_prefix_max(const float* __restrict__ x, int ns,
                                                               const double* __restrict__ mean, float* __restrict__ pmax) {
    __shared__ float wsum[kStatThreads / 64];
    __shared__ float red[kStatThreads / 64];
    const float* row = x + (size_t)blockIdx.x * ns;
    const float best = prefix_max_sweep(row, ns, mean2_load(mean, blockIdx.x), wsum);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        float b = red[0];
        for (int w = 1; w < kStatThreads / 64; ++w) b = fmaxf(b, red[w]);
        pmax[blockIdx.x] = b;
    }
}

// pmax / rowmax / eps (all or none): the per-row decision -- a row whose term cannot exceed eps x its own largest
// correlation (rowmax: the correlator's epilogue, d4w_xcorr_mm_rowmax_f32) is left as it is, every other row gets the term
// and its rowmax entry is formed again over the updated lags.
__global__ __launch_bounds__(kStatThreads) void xcorr_dc_tail(const float* __restrict__ x, int ns,
                                                              const double* __restrict__ mean,
                                                              const float* __restrict__ maxabs, float coef, int L,
                                                              float* __restrict__ y, const float* __restrict__ pmax,
                                                              float* __restrict__ rowmax, float eps) {
    __shared__ float wsum[kStatThreads / 64];
    __shared__ float red[kStatThreads / 64];
    const float* row = x + (size_t)blockIdx.x * ns;
    float* out = y + (size_t)blockIdx.x * ns;
    const Mean2 m = mean2_load(mean, blockIdx.x);
    float g = 1.f;
    if (maxabs) {
        const float a = maxabs[blockIdx.x];
        g = (a > 0.f) ? 1.0f / a : 0.f;
    }
    const float cg = coef * g;
    if (pmax && fabsf(cg) * pmax[blockIdx.x] <= eps * fmaxf(rowmax[blockIdx.x], 0.f)) return;
    const bool vec = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    const int tid = threadIdx.x;
    double carry = 0.0;                                      // sum of (x - m) over all earlier chunks
    float best = -INFINITY;
    for (int j0 = 0; j0 < ns; j0 += 4 * kStatThreads) {
        float v[4], total;
        tail_load4(row, j0 + 4 * tid, ns, vec, m, v);
        float pre = (float)carry + tail_scan((v[0] + v[1]) + (v[2] + v[3]), wsum, total);      // P[j0 + 4 tid]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 4 * tid + k;                  // P[j] belongs to lag j - L
            if (j >= L && j < ns) {
                const float o = out[j - L] + cg * pre;
                out[j - L] = o;
                best = fmaxf(best, o);
            }
            pre += v[k];
        }
        carry += (double)total;
    }
    if (!rowmax) return;
    for (int k = max(ns - L, 0) + tid; k < ns; k += kStatThreads) best = fmaxf(best, out[k]);     // the lags the term leaves alone
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        float b = red[0];
        for (int w = 1; w < kStatThreads / 64; ++w) b = fmaxf(b, red[w]);
        rowmax[blockIdx.x] = b;
    }
}

// =============================================================================================
// matched filter: direct-form correlation, NT templates fused over one read of x
//   y_t[c][k] = g[c] * sum_n (x[c][n+k] - m[c]) * taps[t][n]
// VALU-bound (2*(L_0 + L_1) flop per 4 + 4*NT bytes).  On gfx950 a v_pk_fma_f32 issues
// in the same 4 cycles as a scalar v_fma_f32, so the kernel is built around packed FMAs, and a
// ds_read_b128 of a full wave occupies the CU's LDS pipe for 8 cycles, so the register blocking is
// chosen to keep LDS traffic far below the FMA issue rate:
//   * a workgroup owns TWO rows x (256 R) lags; the de-meaned pieces sit in LDS interleaved as
//     (xA[j], xB[j]) pairs, so every v_pk_fma_f32 advances the same lag of both rows and the tap is
//     one broadcast SGPR (taps are wave-uniform scalar loads);
//   * a thread owns R CONSECUTIVE lags and slides an (R + 2)-sample register window over the taps:
//     one ds_read_b128 (two new samples of both rows) per 2 taps = per 2 R NT packed FMAs
//     (R = 8, NT = 2: 32 FMAs, i.e. 128 issue cycles per 8 LDS cycles);
//   * the window of thread t starts at float4 index (R/2) t, a 16 R-byte lane stride that would be
//     an (R/2)-way bank conflict; the tile is therefore stored residue-major: float4 index
//     f = (R/2) q + r lives at r * PITCH + q, which turns every window read into consecutive
//     lanes -> consecutive 16-byte words (PITCH = 4 mod 16 keeps the staging writes conflict-free);
//   * the two templates may have different supports: taps beyond the shorter one are only
//     applied to the longer one (HF 136 / LF 156 taps: 6 % fewer FMAs than padding both to 156).
// =============================================================================================
constexpr int kXcThreads = 256;
constexpr int kXcTapBlock = 256;                      // taps per LDS staging round

template <int R>
struct XcGeom {
    static constexpr int H = R / 2;                                   // float4 per R samples
    static constexpr int TILE = kXcThreads * R;                       // lags per workgroup
    static constexpr int NQ = kXcThreads + kXcTapBlock / R + 1;       // float4 columns per residue row
    static constexpr int PITCH = ((NQ + 15) / 16) * 16 + 4;
};

template <int NT, int R>
__global__ __launch_bounds__(kXcThreads) void xcorr_fir(const float* __restrict__ x, int nx, int ns,
                                                        const double* __restrict__ mean,
                                                        const float* __restrict__ maxabs,
                                                        const float* __restrict__ taps0,
                                                        const float* __restrict__ taps1, int l_both, int l_long,
                                                        float* __restrict__ y0, float* __restrict__ y1) {
    typedef XcGeom<R> G;
    constexpr int H = G::H, PITCH = G::PITCH;
    __shared__ float4 xs4[H * PITCH];
    float2* xs2 = reinterpret_cast<float2*>(xs4);
    const int tid = threadIdx.x;
    const int rowA = 2 * blockIdx.y;
    const bool hasB = (rowA + 1 < nx);
    const int rowB = hasB ? rowA + 1 : rowA;
    const int k0 = blockIdx.x * G::TILE;
    const float* pa = x + (size_t)rowA * ns;
    const float* pb = x + (size_t)rowB * ns;
    const Mean2 ma = mean2_load(mean, rowA), mb = mean2_load(mean, rowB);
    float ga = 1.f, gb = 1.f;
    if (maxabs) {
        const float a = maxabs[rowA], b = maxabs[rowB];
        ga = (a > 0.f) ? 1.0f / a : 0.f;
        gb = (b > 0.f) ? 1.0f / b : 0.f;
    }
    v2f acc[NT][R];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[t][r] = v2_make(0.f, 0.f);

    const int ltot = (NT == 2) ? l_long : l_both;
    for (int n0 = 0; n0 < ltot; n0 += kXcTapBlock) {
        const int nb = min(kXcTapBlock, ltot - n0);           // taps in this round (even)
        const int need = G::TILE + nb;                        // samples k0+n0 .. k0+n0+need-1
        for (int j = tid; j < need; j += kXcThreads) {
            const int i = k0 + n0 + j;
            float2 v = make_float2(0.f, 0.f);                 // beyond the row: zero padding
            if (i < ns) v = make_float2(demean(pa[i], ma), demean(pb[i], mb));
            const int f = j >> 1, q = f / H, rr = f - q * H;
            xs2[2 * (rr * PITCH + q) + (j & 1)] = v;
        }
        __syncthreads();
        v2f w[R + 2];
#pragma unroll
        for (int s = 0; s < H; ++s) {
            const float4 c = lds_read4(xs4 + s * PITCH + tid);
            w[2 * s] = v2_make(c.x, c.y);
            w[2 * s + 1] = v2_make(c.z, c.w);
        }
        const float4* wbase = xs4 + tid;
        const int steps = nb / 2;
        const int steps_both = (NT == 2) ? max(0, min(nb, l_both - n0)) / 2 : steps;
        // one step = two taps: fetch the two samples that enter the window, 2 R FMAs per template
        auto fetch = [&](int st) {
            const int s = H + st;
            const float4 c = lds_read4(wbase + (s % H) * PITCH + s / H);
            w[R] = v2_make(c.x, c.y);
            w[R + 1] = v2_make(c.z, c.w);
        };
        auto slide = [&]() {
#pragma unroll
            for (int i = 0; i < R; ++i) w[i] = w[i + 2];
        };
        int st = 0;
#pragma unroll R / 2 + 1
        for (; st < steps_both; ++st) {
            fetch(st);
            const float a0 = taps0[n0 + 2 * st], a1 = taps0[n0 + 2 * st + 1];
            float b0 = 0.f, b1 = 0.f;
            if (NT == 2) { b0 = taps1[n0 + 2 * st]; b1 = taps1[n0 + 2 * st + 1]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[0][r] = v2_fma(w[r], a0, acc[0][r]);
                acc[0][r] = v2_fma(w[r + 1], a1, acc[0][r]);
                if (NT == 2) {
                    acc[NT - 1][r] = v2_fma(w[r], b0, acc[NT - 1][r]);
                    acc[NT - 1][r] = v2_fma(w[r + 1], b1, acc[NT - 1][r]);
                }
            }
            slide();
        }
        if (NT == 2) {
#pragma unroll R / 2 + 1
            for (; st < steps; ++st) {                        // the longer template's extra taps
                fetch(st);
                const float b0 = taps1[n0 + 2 * st], b1 = taps1[n0 + 2 * st + 1];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[NT - 1][r] = v2_fma(w[r], b0, acc[NT - 1][r]);
                    acc[NT - 1][r] = v2_fma(w[r + 1], b1, acc[NT - 1][r]);
                }
                slide();
            }
        }
        __syncthreads();
    }
    const int k = k0 + R * tid;
    if (k >= ns) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* base = (t == 0 ? y0 : y1);
#pragma unroll
        for (int rowsel = 0; rowsel < 2; ++rowsel) {
            if (rowsel == 1 && !hasB) break;
            const size_t off = (size_t)(rowsel ? rowB : rowA) * ns + k;
            const float g = rowsel ? gb : ga;
            float o[R];
#pragma unroll
            for (int r = 0; r < R; ++r) o[r] = (rowsel ? v2_y(acc[t][r]) : v2_x(acc[t][r])) * g;
            float* dst = base + off;
            if (k + R <= ns && (off & 3) == 0) {
#pragma unroll
                for (int r = 0; r < R; r += 4)
                    *reinterpret_cast<float4*>(dst + r) = make_float4(o[r], o[r + 1], o[r + 2], o[r + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (k + r < ns) dst[r] = o[r];
            }
        }
    }
}

}  // namespace d4w

using namespace d4w;

template <bool REV, typename T>
static int sos_launch(int nsec, dim3 grid, void* stream, const SosArgsT<T>& A, const float* src,
                      const float* edge_in, float* dst, float* edge_out, int nx, int ns, int padlen,
                      int S, int W, const float* xorig, float dc_gain2) {
    switch (nsec) {
#define D4W_SOS_CASE(N)                                                                               \
    case N:                                                                                           \
        hipLaunchKernelGGL((sos_pass<N, REV, T>), grid, dim3(kSosRows), 0, (hipStream_t)stream, A, src, \
                           edge_in, dst, edge_out, nx, ns, padlen, S, W, xorig, dc_gain2);            \
        break;
        D4W_SOS_CASE(1) D4W_SOS_CASE(2) D4W_SOS_CASE(3) D4W_SOS_CASE(4) D4W_SOS_CASE(5)
        D4W_SOS_CASE(6) D4W_SOS_CASE(7) D4W_SOS_CASE(8) D4W_SOS_CASE(9) D4W_SOS_CASE(10)
#undef D4W_SOS_CASE
        default:
            return fail(D4W_EINVAL, "nsec = %d not in 1..%d", nsec, kSosMaxSec);
    }
    D4W_HIP(hipGetLastError());
    return D4W_OK;
}

template <int R>
static int xcorr_launch(const float* x, int nx, int ns, const double* mean, const float* maxabs, const float* taps0,
                        const float* taps1, int ntpl, int l_both, int l_long, float* y0, float* y1, void* stream) {
    const dim3 grid(ceil_div(ns, XcGeom<R>::TILE), ceil_div(nx, 2));
    if (grid.y > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 131070", nx);
    if (ntpl == 1)
        D4W_LAUNCH((xcorr_fir<1, R>), grid, dim3(kXcThreads), 0, stream, x, nx, ns, mean, maxabs, taps0, taps1, l_both,
                   l_long, y0, y1);
    else
        D4W_LAUNCH((xcorr_fir<2, R>), grid, dim3(kXcThreads), 0, stream, x, nx, ns, mean, maxabs, taps0, taps1, l_both,
                   l_long, y0, y1);
    return D4W_OK;
}

// coefficients of the cascade in both precisions, the precision the recursion needs, the squared DC gain
static int sos_prepare(const double* sos, const double* zi, int nsec, SosArgs& A, SosArgsT<double>& Ad, bool& precise, double& dcg2) {
    memset(&A, 0, sizeof(A));
    memset(&Ad, 0, sizeof(Ad));
    double dmin = 1e30;                                     // smallest |A_s(e^jw)| over the sections and w
    double dcg = 1.0;                                       // DC gain H(1) of the cascade
    for (int s = 0; s < nsec; ++s) {
        const double* c = sos + 6 * s;
        if (c[3] == 0.0) return fail(D4W_EINVAL, "section %d has a0 = 0", s);
        const double a0 = c[3], a1 = c[4] / a0, a2 = c[5] / a0;
        A.s[s] = SosCoef{(float)(c[0] / a0), (float)(c[1] / a0), (float)(c[2] / a0), (float)a1, (float)a2,
                         (float)zi[2 * s], (float)zi[2 * s + 1], 0.f};
        Ad.s[s] = SosCoefT<double>{c[0] / a0, c[1] / a0, c[2] / a0, a1, a2, zi[2 * s], zi[2 * s + 1], 0.0};
        dcg *= ((c[0] + c[1] + c[2]) / a0) / (1.0 + a1 + a2);
        for (int k = 0; k <= 1024; ++k) {                   // |1 + a1 z^-1 + a2 z^-2| on the unit circle
            const double w = M_PI * k / 1024.0;
            const double re = 1.0 + a1 * cos(w) + a2 * cos(2.0 * w), im = -a1 * sin(w) - a2 * sin(2.0 * w);
            dmin = std::min(dmin, sqrt(re * re + im * im));
        }
    }
    // The float32 recursion's rounding noise grows like 1 / dmin^2 (dmin ~ (1 - r) 2 sin(theta) for a pole
    // r e^(j theta): low-frequency poles are the ill-conditioned ones).  Order-8 14-30 Hz at 200 Hz:
    // dmin = 0.027, error 1.5e-6 of the output; 5-38 Hz: dmin = 0.0078, 2e-5.  Below 0.015: float64 states.
    static const int f64_env = [] { const char* v = getenv("D4W_SOS_F64"); return v ? atoi(v) : -1; }();
    precise = (f64_env >= 0) ? (f64_env > 0) : (dmin < 0.015);
    dcg2 = dcg * dcg;                                       // forward and backward pass
    return D4W_OK;
}

// forward + backward launch of sos_pass_lanes: src rows -> t rows (all outputs, + the right-extension outputs in `edge`),
// t rows -> the kept windows of the dst rows
static int sos_lanes_both(int nsec, bool precise, const SosArgs& A, const SosArgsT<double>& Ad, SlRows src, SlRows tr, SlRows dst,
                          int ka0, int ka1, int kb0, int kb1, float* edge, int nrows, int n, int padlen, SlRows piv, float dcg2,
                          void* stream, int phase = 0) {
    const int G = nsec <= 8 ? 8 : 16;
    const dim3 lgrid(ceil_div(nrows, (kSlThreads / 64) * (64 / G)));
#define D4W_SL_LAUNCH(GG, TT, AA)                                                                                            \
    do {                                                                                                                     \
        if (phase != 2)                                                                                                      \
            D4W_LAUNCH((sos_pass_lanes<GG, false, TT>), lgrid, dim3(kSlThreads), 0, stream, AA, nsec, src, (const float*)nullptr, tr, \
                       0, n, 0, n, edge, nrows, n, padlen, piv, 0.f);                                                        \
        if (phase != 1)                                                                                                      \
            D4W_LAUNCH((sos_pass_lanes<GG, true, TT>), lgrid, dim3(kSlThreads), 0, stream, AA, nsec, tr, (const float*)edge, dst, \
                       ka0, ka1, kb0, kb1, (float*)nullptr, nrows, n, padlen, piv, dcg2);                                    \
    } while (0)
    if (precise) { if (G == 8) D4W_SL_LAUNCH(8, double, Ad); else D4W_SL_LAUNCH(16, double, Ad); }
    else { if (G == 8) D4W_SL_LAUNCH(8, float, A); else D4W_SL_LAUNCH(16, float, A); }
#undef D4W_SL_LAUNCH
    return D4W_OK;
}

extern "C" {

size_t d4w_sosfiltfilt_ws_bytes(int nx, int ns, int padlen) {
    if (nx < 1 || ns < 1 || padlen < 0) return 0;
    return ((size_t)nx * ns + (size_t)nx * std::max(padlen, 1) + (size_t)nx) * sizeof(float);
}

int d4w_sosfiltfilt_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi,
                        int nsec, int padlen, int seg_len, int warm, void* ws, void* stream) {
    if (!x || !y || !sos || !zi || !ws) return fail(D4W_EINVAL, "NULL argument");
    if (nx < 1 || ns < 1 || padlen < 0) return fail(D4W_EINVAL, "bad shape %d x %d (padlen %d)", nx, ns, padlen);
    if (nsec < 1 || nsec > kSosMaxSec) return fail(D4W_EINVAL, "nsec = %d not in 1..%d", nsec, kSosMaxSec);
    if (ns <= padlen)
        return fail(D4W_EINVAL, "The length of the input vector x must be greater than padlen, which is %d.", padlen);
    SosArgs A;
    SosArgsT<double> Ad;
    bool precise = false;
    double dcg2 = 1.0;
    if (int rcp = sos_prepare(sos, zi, nsec, A, Ad, precise, dcg2)) return rcp;
    int S = seg_len, W = warm;
    if (S <= 0 || W <= 0 || S >= ns) { S = ns; W = ns; }           // one exact segment per row
    S = ((S + kSosChunk - 1) / kSosChunk) * kSosChunk;
    const int nseg = ceil_div(ns, S);
    float* t = (float*)ws;
    float* edge = t + (size_t)nx * ns;
    const dim3 grid(ceil_div(nx, kSosRows), nseg);
    // x may alias y: the backward pass needs the rows' first samples after y[.][0] may have been written
    // by another workgroup, so they are saved in the workspace's edge area tail first
    float* first = edge + (size_t)nx * std::max(padlen, 1);
    D4W_LAUNCH(sos_first_samples, dim3(ceil_div(nx, 256)), dim3(256), 0, stream, x, nx, ns, first);
    // One exact segment per row: the sections of a row on adjacent lanes (sos_pass_lanes) -- 8 x the waves of the
    // lane-per-row kernel, a step of a few dependent operations instead of 3 nsec.  D4W_SOS_LANES=0: the lane-per-row kernel.
    static const int lanes_env = [] { const char* v = getenv("D4W_SOS_LANES"); return v ? atoi(v) : 1; }();
    if (nseg == 1 && lanes_env) {
        const SlRows xr{x, (size_t)ns, 0, nx}, tr{t, (size_t)ns, 0, nx}, yr{y, (size_t)ns, 0, nx}, pr{first, 1, 0, nx};
        return sos_lanes_both(nsec, precise, A, Ad, xr, tr, yr, 0, ns, 0, ns, edge, nx, ns, padlen, pr, (float)dcg2, stream);
    }
    if (precise) {
        int rcd = sos_launch<false, double>(nsec, grid, stream, Ad, x, nullptr, t, edge, nx, ns, padlen, S, W, first, 0.f);
        if (rcd) return rcd;
        return sos_launch<true, double>(nsec, grid, stream, Ad, t, edge, y, nullptr, nx, ns, padlen, S, W, first, (float)dcg2);
    }
    int rc = sos_launch<false, float>(nsec, grid, stream, A, x, nullptr, t, edge, nx, ns, padlen, S, W, first, 0.f);
    if (rc) return rc;
    return sos_launch<true, float>(nsec, grid, stream, A, t, edge, y, nullptr, nx, ns, padlen, S, W, first, (float)dcg2);
}

size_t d4w_sosfiltfilt_ends_ws_bytes(int nx, int piece, int padlen) {
    if (nx < 1 || piece < 1 || padlen < 0) return 0;
    return ((size_t)2 * nx * piece + (size_t)2 * nx * std::max(padlen, 1)) * sizeof(float);
}

int d4w_sosfiltfilt_ends_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi, int nsec, int padlen,
                             int piece, int keep, int phase, void* ws, void* stream) {
    return d4w_sosfiltfilt_ends_sides_f32(x, y, nx, ns, sos, zi, nsec, padlen, piece, keep, phase, 3, ws, stream);
}

int d4w_sosfiltfilt_ends_sides_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi, int nsec,
                                   int padlen, int piece, int keep, int phase, int sides, void* ws, void* stream) {
    if (!x || !y || !sos || !zi || !ws) return fail(D4W_EINVAL, "NULL argument");
    if (sides < 1 || sides > 3) return fail(D4W_EINVAL, "sides = %d (1 left, 2 right, 3 both)", sides);
    if (x == y) return fail(D4W_EINVAL, "the row ends are read from x while y is written: x and y must not alias");
    if (nx < 1 || ns < 1 || padlen < 0) return fail(D4W_EINVAL, "bad shape %d x %d (padlen %d)", nx, ns, padlen);
    if (nsec < 1 || nsec > kSosMaxSec) return fail(D4W_EINVAL, "nsec = %d not in 1..%d", nsec, kSosMaxSec);
    if (phase < 0 || phase > 2) return fail(D4W_EINVAL, "phase = %d (0 both passes, 1 forward, 2 backward)", phase);
    if (piece <= padlen || piece > ns || keep < 1 || keep > piece)
        return fail(D4W_EINVAL, "pieces of %d samples (keep %d) need padlen %d < piece <= ns = %d and 1 <= keep <= piece", piece, keep, padlen, ns);
    SosArgs A;
    SosArgsT<double> Ad;
    bool precise = false;
    double dcg2 = 1.0;
    if (int rcp = sos_prepare(sos, zi, nsec, A, Ad, precise, dcg2)) return rcp;
    float* t = (float*)ws;
    float* edge = t + (size_t)2 * nx * piece;
    // rows 0 .. nx-1: the left pieces x[r][0 .. piece), rows nx .. 2 nx - 1: the right pieces x[r][ns - piece .. ns); the
    // forward outputs of all 2 nx pieces side by side in the workspace; the backward pass writes the `keep` outer samples of
    // each piece straight into y (left: [0, keep), right: piece samples [piece - keep, piece) = y[r][ns - keep .. ns))
    if (sides != 3) {
        // one end only (a file whose other end continues into a neighbour: das4whales_amd/stream.py): nx pieces, all of one kind
        const long long off = (sides == 2) ? (long long)ns - piece : 0;
        const SlRows xr1{x + off, (size_t)ns, 0, nx}, tr1{t, (size_t)piece, 0, nx}, yr1{y + off, (size_t)ns, 0, nx};
        const int k0 = (sides == 2) ? piece - keep : 0, k1 = (sides == 2) ? piece : keep;
        return sos_lanes_both(nsec, precise, A, Ad, xr1, tr1, yr1, k0, k1, k0, k1, edge, nx, piece, padlen, xr1, (float)dcg2, stream, phase);
    }
    const SlRows xr{x, (size_t)ns, (long long)ns - piece, nx}, tr{t, (size_t)piece, (long long)nx * piece, nx};
    const SlRows yr{y, (size_t)ns, (long long)ns - piece, nx};
    return sos_lanes_both(nsec, precise, A, Ad, xr, tr, yr, 0, keep, piece - keep, piece, edge, 2 * nx, piece, padlen, xr, (float)dcg2,
                          stream, phase);
}

int d4w_copy_cols_f32(const float* src, size_t ld_src, float* dst, size_t ld_dst, int nrows, int ncols, void* stream) {
    if (!src || !dst || nrows < 1 || ncols < 1) return fail(D4W_EINVAL, "bad argument");
    if (nrows > 65535) {                                          // grid.y limit: rows in slabs
        for (int r0 = 0; r0 < nrows; r0 += 65535) {
            const int nr = std::min(65535, nrows - r0);
            D4W_LAUNCH(copy_cols, dim3(std::min(8, ceil_div(ncols, 256)), nr), dim3(256), 0, stream, src + (size_t)r0 * ld_src,
                       ld_src, dst + (size_t)r0 * ld_dst, ld_dst, ncols);
        }
        return D4W_OK;
    }
    D4W_LAUNCH(copy_cols, dim3(std::min(8, ceil_div(ncols, 256)), nrows), dim3(256), 0, stream, src, ld_src, dst, ld_dst, ncols);
    return D4W_OK;
}

int d4w_raw2strain_f32(const void* raw, int raw_dtype, int ns, int c0, int cstep, int nx_out, double scale_factor,
                       float* y, void* stream) {
    if (!raw || !y || ns < 1 || nx_out < 1 || c0 < 0 || cstep < 1) return fail(D4W_EINVAL, "bad argument");
    // 16-byte loads when every selected row (and the output rows) starts on 16 bytes and holds whole vectors
    const size_t esz = raw_dtype == 1 ? 2 : raw_dtype == 3 ? 8 : 4;
    const bool vec = raw_dtype >= 0 && raw_dtype <= 3 && ((size_t)ns * esz) % 16 == 0 && (ns & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(raw) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (vec) {
        switch (raw_dtype) {
            case 0: D4W_LAUNCH(raw2strain_rows_vec<int32_t>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const int32_t*)raw, ns, c0, cstep, scale_factor, y); break;
            case 1: D4W_LAUNCH(raw2strain_rows_vec<int16_t>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const int16_t*)raw, ns, c0, cstep, scale_factor, y); break;
            case 2: D4W_LAUNCH(raw2strain_rows_vec<float>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const float*)raw, ns, c0, cstep, scale_factor, y); break;
            default: D4W_LAUNCH(raw2strain_rows_vec<double>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const double*)raw, ns, c0, cstep, scale_factor, y); break;
        }
        return D4W_OK;
    }
    switch (raw_dtype) {
        case 0: D4W_LAUNCH(raw2strain_rows<int32_t>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const int32_t*)raw, ns, c0, cstep, scale_factor, y); break;
        case 1: D4W_LAUNCH(raw2strain_rows<int16_t>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const int16_t*)raw, ns, c0, cstep, scale_factor, y); break;
        case 2: D4W_LAUNCH(raw2strain_rows<float>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const float*)raw, ns, c0, cstep, scale_factor, y); break;
        case 3: D4W_LAUNCH(raw2strain_rows<double>, dim3(nx_out), dim3(kStatThreads), 0, stream, (const double*)raw, ns, c0, cstep, scale_factor, y); break;
        default: return fail(D4W_EINVAL, "raw_dtype = %d (0 int32, 1 int16, 2 float32, 3 float64)", raw_dtype);
    }
    return D4W_OK;
}

int d4w_xcorr_dc_tail_rows_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs, double coef,
                               int support, float* y, const float* pmax, float* rowmax, double eps, void* stream) {
    if (!x || !y || nx < 1 || ns < 1 || support < 1) return fail(D4W_EINVAL, "bad argument");
    if (pmax && (!rowmax || !(eps >= 0.0))) return fail(D4W_EINVAL, "the per-row decision needs pmax, rowmax and eps >= 0");
    if (coef == 0.0 || support >= ns) return D4W_OK;
    D4W_LAUNCH(xcorr_dc_tail, dim3(nx), dim3(kStatThreads), 0, stream, x, ns, mean, maxabs, (float)coef, support, y, pmax,
               rowmax, (float)eps);
    return D4W_OK;
}

int d4w_xcorr_dc_tail_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs, double coef,
                          int support, float* y, void* stream) {
    return d4w_xcorr_dc_tail_rows_f32(x, nx, ns, mean, maxabs, coef, support, y, nullptr, nullptr, 0.0, stream);
}

int d4w_row_prefix_max_f32(const float* x, int nx, int ns, const double* mean, float* pmax, void* stream) {
    if (!x || !pmax || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_prefix_max, dim3(nx), dim3(kStatThreads), 0, stream, x, ns, mean, pmax);
    return D4W_OK;
}

int d4w_row_stats_f32(const float* x, int nx, int ns, double* mean, float* maxabs, void* stream) {
    if (!x || !mean || !maxabs || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_stats, dim3(nx), dim3(kStatThreads), 0, stream, x, ns, mean, maxabs, (float*)nullptr);
    return D4W_OK;
}

int d4w_row_stats_prefix_f32(const float* x, int nx, int ns, double* mean, float* maxabs, float* pmax, void* stream) {
    if (!x || !mean || !maxabs || !pmax || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_stats, dim3(nx), dim3(kStatThreads), 0, stream, x, ns, mean, maxabs, pmax);
    return D4W_OK;
}

int d4w_xcorr_lens_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs, const float* taps,
                       int ntpl, int ltaps, int len0, int len1, float* y0, float* y1, void* stream) {
    if (!x || !taps || !y0 || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    if (ntpl < 1 || ntpl > 2 || (ntpl == 2 && !y1)) return fail(D4W_EINVAL, "ntpl = %d (1 or 2 templates per call)", ntpl);
    if (ltaps < 4 || (ltaps & 3)) return fail(D4W_EINVAL, "ltaps = %d must be a positive multiple of 4", ltaps);
    if (len0 < 1 || len0 > ltaps || (ntpl == 2 && (len1 < 1 || len1 > ltaps)))
        return fail(D4W_EINVAL, "template supports (%d, %d) must lie in 1..ltaps = %d", len0, len1, ltaps);
    // taps are zero padded to ltaps (a multiple of 4), so rounding a support up to even stays in range
    int l0 = (len0 + 1) & ~1, l1 = (len1 + 1) & ~1;
    const float *t0 = taps, *t1 = taps + ltaps;
    if (ntpl == 1) { l1 = l0; t1 = t0; }
    else if (l0 > l1) {                                   // kernel convention: template 0 is the shorter one
        std::swap(l0, l1); std::swap(t0, t1); std::swap(y0, y1);
    }
    static const int r_env = [] { const char* v = getenv("D4W_XC_R"); return v ? atoi(v) : 0; }();   // tuning knob
    if (r_env == 4) return xcorr_launch<4>(x, nx, ns, mean, maxabs, t0, t1, ntpl, l0, l1, y0, y1, stream);
    return xcorr_launch<8>(x, nx, ns, mean, maxabs, t0, t1, ntpl, l0, l1, y0, y1, stream);
}

int d4w_xcorr_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs, const float* taps,
                  int ntpl, int ltaps, float* y0, float* y1, void* stream) {
    return d4w_xcorr_lens_f32(x, nx, ns, mean, maxabs, taps, ntpl, ltaps, ltaps, ltaps, y0, y1, stream);
}

}  // extern "C"
