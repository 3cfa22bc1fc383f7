// Row-independent spectral operators on MI355X (gfx950), every transform running in LDS with the
// mixed-radix stages of fft_lds.h:
//   * analytic signal along time (scipy.signal.hilbert): envelope, Hilbert transform, envelope SNR,
//     instantaneous frequency   -- reference dsp.py:830-856 (instant_freq), dsp.py:956-976
//     (snr_tr_array), detect.py:192,217 (pick_times_env / process_corr);
//   * STFT magnitude as librosa.stft is called by the reference (dsp.py:66-68 get_spectrogram,
//     detect.py:382 get_sliced_nspectrogram), with the per-channel maximum the reference
//     normalises by (dsp.py:76, detect.py:387);
//   * spectrogram x kernel correlation (detect.py:579-602 xcorr2d, :605-647 xcorr) and the median
//     it divides by (a radix select, not a reduction);
//   * peak picking: scipy.signal.find_peaks(x, prominence=thr) (detect.py:192,217,271);
//   * dsp.get_fx (dsp.py:18-38).
// All of it is HBM-streaming work; no MFMA.
#include <map>
#include <mutex>

#include "fft_host.h"

namespace d4w {

constexpr int kSpThreads = 256;
constexpr size_t kSpLdsMax = 150 * 1024;      // dynamic LDS a single-row transform may use

// ---------------------------------------------------------------------------------------------
// per-(device, length) transform tables, cached for the life of the process
// ---------------------------------------------------------------------------------------------
struct RowFftDev {
    AxisDesc ax;
    const int* pos;       // [L] frequency -> LDS position after the forward (DIF) transform
    const int* p2f;       // [L] position -> frequency
    const float2* wpack;  // [L] exp(-2 pi i f / (2L)): real-packing twiddle of a 2L-point real row
    const float* hann;    // [L] periodic Hann window (scipy get_window('hann', L, fftbins=True))
    const float2* wfull;  // [L] exp(-2 pi i m / L)
    // Bluestein form for a length with a prime factor > 31 (single-row transforms only): the L-point DFT as a
    // circular convolution of length bs_L = 2^a 3^b 5^c >= 2 L - 1 with the chirp exp(-i pi n^2 / L); pos / p2f are the
    // identity then.  bs_L = 0: the mixed-radix transform of `ax`.
    int bs_L;
    AxisDesc ax_bs;
    const float2* bs_chirp;   // [L]
    const float2* bs_filt;    // [bs_L]  FFT of the wrapped conjugate chirp / bs_L, at the DIF positions of ax_bs
};

struct RowFftHost {
    RowFftDev dev;
    bool generic;
    std::vector<void*> allocs;
};

static std::mutex g_rowfft_mu;
static std::map<std::pair<int, int>, RowFftHost*> g_rowfft;

template <typename T>
static int sp_upload(RowFftHost* h, const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    D4W_HIP(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    h->allocs.push_back(p);
    D4W_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return D4W_OK;
}

static int row_fft_get(int L, const RowFftHost** out) {
    int devid = 0;
    D4W_HIP(hipGetDevice(&devid));
    std::lock_guard<std::mutex> lk(g_rowfft_mu);
    auto it = g_rowfft.find({devid, L});
    if (it != g_rowfft.end()) { *out = it->second; return D4W_OK; }
    std::vector<int> rad;
    const bool bluestein = !factor_radices(L, rad);
    int bs_L = 0;
    if (bluestein) {
        bs_L = smooth_len_235(2L * L - 1);
        if (((size_t)bs_L + kTwLo + (size_t)(bs_L + kTwLo - 1) / kTwLo) * sizeof(float2) > kSpLdsMax)
            return fail(D4W_EINVAL, "transform length %d has a prime factor > 31 and is too long for the Bluestein form "
                        "(dsp.supported_length(n) gives the nearest shorter length with a direct kernel)", L);
        rad.clear();
    }
    RowFftHost* h = new RowFftHost();
    memset(&h->dev, 0, sizeof(h->dev));
    AxisDesc& ax = h->dev.ax;
    ax.L = L;
    ax.nstage = (L == 1) ? 0 : (int)rad.size();
    for (int i = 0; i < kMaxStages; ++i) ax.radix[i] = (i < (int)rad.size() && L > 1) ? rad[i] : 1;
    if (L == 1) rad.clear();
    std::vector<int> p2f = pos_to_freq(L, rad);
    if (bluestein)
        for (int i = 0; i < L; ++i) p2f[i] = i;                   // natural order
    std::vector<int> pos(L);
    for (int p = 0; p < L; ++p) pos[p2f[p]] = p;
    std::vector<float2> wp(L);
    for (int f = 0; f < L; ++f) wp[f] = wexp(f, 2LL * L);
    std::vector<float> hann(L);
    for (int n = 0; n < L; ++n) hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)L));
    int rc = sp_upload(h, twiddle_table2(L, &ax.nhi), &ax.tw2);
    if (!rc && bluestein) {
        std::vector<int> radL;
        factor_radices(bs_L, radL);
        AxisDesc& ab = h->dev.ax_bs;
        ab.L = bs_L;
        ab.nstage = (int)radL.size();
        for (int i = 0; i < kMaxStages; ++i) ab.radix[i] = (i < (int)radL.size()) ? radL[i] : 1;
        const std::vector<int> fL = pos_to_freq(bs_L, radL);
        std::vector<float2> chirp(L), filt(bs_L);
        std::vector<double> bre(bs_L, 0.0), bim(bs_L, 0.0);
        for (int n = 0; n < L; ++n) {
            const double ph = M_PI * (double)(((long long)n * n) % (2LL * L)) / (double)L;
            chirp[n] = make_float2((float)cos(ph), (float)-sin(ph));
            bre[n] = cos(ph); bim[n] = sin(ph);
            if (n) { bre[bs_L - n] = cos(ph); bim[bs_L - n] = sin(ph); }
        }
        host_dft_any(bre, bim);
        for (int q = 0; q < bs_L; ++q) filt[q] = make_float2((float)(bre[fL[q]] / bs_L), (float)(bim[fL[q]] / bs_L));
        rc = sp_upload(h, twiddle_table2(bs_L, &ab.nhi), &ab.tw2);
        if (!rc) rc = sp_upload(h, chirp, &h->dev.bs_chirp);
        if (!rc) rc = sp_upload(h, filt, &h->dev.bs_filt);
        h->dev.bs_L = bs_L;
    }
    if (!rc) rc = sp_upload(h, pos, &h->dev.pos);
    if (!rc) rc = sp_upload(h, p2f, &h->dev.p2f);
    if (!rc) rc = sp_upload(h, wp, &h->dev.wpack);
    if (!rc) rc = sp_upload(h, hann, &h->dev.hann);
    {
        std::vector<float2> wf(L);
        for (int m = 0; m < L; ++m) wf[m] = wexp(m, L);
        if (!rc) rc = sp_upload(h, wf, &h->dev.wfull);
    }
    if (rc) {
        for (void* p : h->allocs) (void)hipFree(p);
        delete h;
        return rc;
    }
    h->generic = !bluestein && axis_needs_generic(ax);
    g_rowfft[{devid, L}] = h;
    *out = h;
    return D4W_OK;
}

// ---------------------------------------------------------------------------------------------
// analytic signal of every row  (scipy.signal.hilbert(x, axis=1) = ifft(fft(x) * h))
//
// Even ns (every shape the reference processes): the real row is read as M = ns/2 packed complex
// samples, one M-point FFT gives the half spectrum X(f) after the real-spectrum untangle, the
// Hilbert transform's spectrum is -i X(f) (0 < f < M; 0 at DC and Nyquist, scipy's h = 1 there
// belongs to the real part), and the packed inverse returns H[x] itself; z = x + i H[x].
// Odd ns: plain ns-point complex transform of (x, 0), times h, inverse.
//   mode 0: |z|                               (envelope, detect.py:192)
//   mode 1: imag(z) = H[x]
//   mode 2: 10 log10(|z|^2 / var[row])        (dsp.py:975)
//   mode 4: |z| / std[row]                    (improcess.trace2image before scaling, improcess.py:60)
//   mode 3: arg(z[i+1] conj(z[i])) * fscale   (diff(unwrap(angle z)) / 2 pi * fs, dsp.py:846-855),
//           ns - 1 outputs per row
// ---------------------------------------------------------------------------------------------
enum { kAnEnvelope = 0, kAnHilbert = 1, kAnSnr = 2, kAnIfreq = 3, kAnEnvStd = 4 };

// elements / twiddle axis of a single-row transform tile
__device__ __forceinline__ int row_tile_elems(const RowFftDev& F) { return F.bs_L ? F.bs_L : F.ax.L; }
__device__ __forceinline__ const AxisDesc& row_tw_axis(const RowFftDev& F) { return F.bs_L ? F.ax_bs : F.ax; }

// DFT (INV = false) or unnormalised inverse DFT of the F.ax.L values at the start of `tile`, in place; result at
// positions F.pos[f].  The tile is synchronised on entry and on return.  Bluestein form when F.bs_L != 0 (the tile
// then holds F.bs_L elements): x w -> FFT -> x filter -> IFFT -> x w, w = exp(-i pi n^2 / L), conjugated for INV.
template <bool INV, bool GENERIC>
__device__ __forceinline__ void row_dft(float2* tile, const RowFftDev& F, const TwLds tw, int tid, int nthr) {
    if (F.bs_L == 0) {
        lds_fft<INV, false, GENERIC>(tile, F.ax, tw, 1, 1, 0, 1, 0, tid, nthr);
        return;
    }
    const int L = F.ax.L, BL = F.bs_L;
    for (int i = tid; i < BL; i += nthr) {
        float2 v = make_float2(0.f, 0.f);
        if (i < L) v = INV ? c_mulc(tile[i], F.bs_chirp[i]) : c_mul(tile[i], F.bs_chirp[i]);
        tile[i] = v;
    }
    lds_barrier();
    lds_fft<false, false, false>(tile, F.ax_bs, tw, 1, 1, 0, 1, 0, tid, nthr);
    for (int i = tid; i < BL; i += nthr) tile[i] = INV ? c_mulc(tile[i], F.bs_filt[i]) : c_mul(tile[i], F.bs_filt[i]);
    lds_barrier();
    lds_fft<true, false, false>(tile, F.ax_bs, tw, 1, 1, 0, 1, 0, tid, nthr);
    for (int i = tid; i < L; i += nthr) tile[i] = INV ? c_mulc(tile[i], F.bs_chirp[i]) : c_mul(tile[i], F.bs_chirp[i]);
    lds_barrier();
}

__device__ __forceinline__ float an_ifreq(float2 z0, float2 z1, float fscale) {
    const float2 p = c_mulc(z1, z0);
    return atan2f(p.y, p.x) * fscale;
}

constexpr int kAnMaxThreads = 512;
template <bool PACKED, bool GENERIC>
__global__ __launch_bounds__(kAnMaxThreads) void analytic_rows(RowFftDev F, const float* __restrict__ x, int ns,
                                                            float* __restrict__ y, int mode,
                                                            const float* __restrict__ var, float fscale) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int L = F.ax.L;
    const TwLds tw = tw_stage(row_tw_axis(F), tile + row_tile_elems(F), tid, nthr);
    const float* xr = x + (size_t)blockIdx.x * ns;
    constexpr int kAhead = 8;                                         // global loads in flight per lane (the sweeps are latency-bound)
    if (PACKED) {
        const float2* x2 = reinterpret_cast<const float2*>(xr);       // ns even: 8-byte aligned rows
        for (int m0 = tid; m0 < L; m0 += kAhead * nthr) {
            float2 q[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int m = m0 + k * nthr;
                q[k] = (m < L) ? x2[m] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int m = m0 + k * nthr;
                if (m < L) tile[m] = q[k];
            }
        }
    } else {
        for (int n = tid; n < L; n += nthr) tile[n] = make_float2(xr[n], 0.f);
    }
    lds_barrier();
    row_dft<false, GENERIC>(tile, F, tw, tid, nthr);
    if (PACKED) {
        const int M = L;
        for (int f = tid; f <= M / 2; f += nthr) {
            const int g = (f == 0) ? 0 : M - f;
            const int pa = F.pos[f], pb = F.pos[g];
            const float2 a = tile[pa], bc = c_conj(tile[pb]);
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
            tile[pa] = out_a;
            if (pb != pa) tile[pb] = out_b;
        }
    } else {
        const int half = (L + 1) / 2;                                 // first negative frequency
        for (int p = tid; p < L; p += nthr) {
            const int f = F.p2f[p];
            float h = (f < half) ? 2.f : 0.f;
            if (f == 0 || (2 * f == L)) h = 1.f;
            tile[p] = c_scale(tile[p], h);
        }
    }
    lds_barrier();
    row_dft<true, GENERIC>(tile, F, tw, tid, nthr);
    const float scale = 1.0f / (float)L;
    auto zat = [&](int i) -> float2 {
        if (PACKED) {
            const float2 h = tile[i >> 1];
            return make_float2(xr[i], ((i & 1) ? h.y : h.x) * scale);
        }
        return c_scale(tile[i], scale);
    };
    if (mode == kAnIfreq) {
        float* yr = y + (size_t)blockIdx.x * (ns - 1);
        for (int i = tid; i < ns - 1; i += nthr) yr[i] = an_ifreq(zat(i), zat(i + 1), fscale);
        return;
    }
    float* yr = y + (size_t)blockIdx.x * ns;
    const float inv_var = (mode == kAnSnr || mode == kAnEnvStd) ? 1.0f / var[blockIdx.x] : 0.f;
    if (PACKED) {
        // two samples per lane: the pair of inputs again (8 bytes, kAhead loads in flight), the pair of Hilbert values from
        // the tile, one 8-byte store
        const float2* x2 = reinterpret_cast<const float2*>(xr);
        float2* y2 = reinterpret_cast<float2*>(yr);
        auto val = [&](float re, float im) -> float {
            const float p = fmaf(re, re, im * im);
            if (mode == kAnEnvelope) return sqrtf(p);
            if (mode == kAnHilbert) return im;
            if (mode == kAnEnvStd) return sqrtf(p * inv_var);
            return 10.0f * log10f(p * inv_var);
        };
        for (int m0 = tid; m0 < L; m0 += kAhead * nthr) {
            float2 q[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int m = m0 + k * nthr;
                q[k] = (m < L) ? x2[m] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int m = m0 + k * nthr;
                if (m < L) {
                    const float2 h = tile[m];
                    y2[m] = make_float2(val(q[k].x, h.x * scale), val(q[k].y, h.y * scale));
                }
            }
        }
        return;
    }
    for (int i = tid; i < ns; i += nthr) {
        const float2 z = zat(i);
        float v;
        if (mode == kAnEnvelope) v = sqrtf(fmaf(z.x, z.x, z.y * z.y));
        else if (mode == kAnHilbert) v = z.y;
        else if (mode == kAnEnvStd) v = sqrtf(fmaf(z.x, z.x, z.y * z.y) * inv_var);
        else v = 10.0f * log10f(fmaf(z.x, z.x, z.y * z.y) * inv_var);
        yr[i] = v;
    }
}

// population variance of every row (np.std(x, axis=1)**2, dsp.py:975-976) in ONE sweep.  Every lane sums d = x - c and d^2 in
// float64 over its share of the row, c = the first sample it meets (its own shift: the shifted-data form is exact enough as
// long as the lane's ~ns / 256 samples are not many orders of magnitude closer to each other than to c); the lanes' (count,
// mean, M2) triples are then merged pairwise with the parallel update of Chan et al., M2 = M2a + M2b + delta^2 na nb / (na + nb),
// which has no cancellation at all.  Rounds 1-5 swept twice (mean, then centred squares: 2.2 ms per sweep of a
// 20 000 x 120 000 block -- the second sweep of a 480-KB row does not come out of L2).
struct VarAcc { double n, mean, m2; };
__device__ __forceinline__ VarAcc var_merge(VarAcc a, VarAcc b) {
    const double n = a.n + b.n;
    if (n == 0.0) return VarAcc{0.0, 0.0, 0.0};
    const double delta = b.mean - a.mean;
    return VarAcc{n, a.mean + delta * (b.n / n), a.m2 + b.m2 + delta * delta * (a.n * b.n / n)};
}
__global__ __launch_bounds__(kSpThreads) void row_var(const float* __restrict__ x, int ns, float* __restrict__ var) {
    __shared__ double red[3][kSpThreads / 64];
    const float* row = x + (size_t)blockIdx.x * ns;
    const int tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, c = 0.0, cnt = 0.0;
    auto take = [&](float v) {
        const double d = (double)v - c;
        s1 += d;
        s2 = fma(d, d, s2);
    };
    if ((ns & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        const int n4 = ns >> 2;
        if (tid < n4) c = (double)r4[tid].x;
        constexpr int kAhead = 4;                                   // 16-byte loads in flight per lane
        for (int i0 = tid; i0 < n4; i0 += kAhead * kSpThreads) {
            float4 q[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const int i = i0 + k * kSpThreads;
                if (i < n4) q[k] = r4[i];
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                if (i0 + k * kSpThreads < n4) { take(q[k].x); take(q[k].y); take(q[k].z); take(q[k].w); cnt += 4.0; }
        }
    } else {
        if (tid < ns) c = (double)row[tid];
        for (int i = tid; i < ns; i += kSpThreads) { take(row[i]); cnt += 1.0; }
    }
    VarAcc a{cnt, cnt > 0.0 ? c + s1 / cnt : 0.0, cnt > 0.0 ? fmax(s2 - s1 * s1 / cnt, 0.0) : 0.0};
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        a = var_merge(a, VarAcc{__shfl_xor(a.n, off), __shfl_xor(a.mean, off), __shfl_xor(a.m2, off)});
    if ((tid & 63) == 0) { red[0][tid / 64] = a.n; red[1][tid / 64] = a.mean; red[2][tid / 64] = a.m2; }
    __syncthreads();
    if (tid == 0) {
        VarAcc t{red[0][0], red[1][0], red[2][0]};
        for (int w = 1; w < kSpThreads / 64; ++w) t = var_merge(t, VarAcc{red[0][w], red[1][w], red[2][w]});
        var[blockIdx.x] = (float)(t.m2 / (double)ns);
    }
}

// 10 log10(x^2 / var[row])   (dsp.py:976)
__global__ __launch_bounds__(kSpThreads) void snr_rows(const float* __restrict__ x, int ns,
                                                       const float* __restrict__ var, float* __restrict__ y) {
    const size_t base = (size_t)blockIdx.y * ns;
    const float inv_var = 1.0f / var[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const float v = x[base + i];
        y[base + i] = 10.0f * log10f(v * v * inv_var);
    }
}

// ---------------------------------------------------------------------------------------------
// dsp.get_fx: 2 |fftshift(fft(x, nfft), axes=1)| / nfft * 1e9   (dsp.py:35-37)
// np.fft.fft(x, nfft) crops or zero-pads the row to nfft samples.
// ---------------------------------------------------------------------------------------------
template <bool GENERIC>
__global__ __launch_bounds__(kSpThreads) void fx_rows(RowFftDev F, const float* __restrict__ x, int ns,
                                                      float* __restrict__ y) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int L = F.ax.L;
    const TwLds tw = tw_stage(row_tw_axis(F), tile + row_tile_elems(F), tid, nthr);
    const float* xr = x + (size_t)blockIdx.x * ns;
    for (int n = tid; n < L; n += nthr) tile[n] = make_float2(n < ns ? xr[n] : 0.f, 0.f);
    lds_barrier();
    row_dft<false, GENERIC>(tile, F, tw, tid, nthr);
    const float scale = (float)(2.0e9 / (double)L);
    float* yr = y + (size_t)blockIdx.x * L;
    const int sh = L / 2;                                             // fftshift: out[j] = F[(j - L//2) mod L]
    for (int j = tid; j < L; j += nthr) {
        int f = j - sh;
        if (f < 0) f += L;
        const float2 v = tile[F.pos[f]];
        yr[j] = sqrtf(fmaf(v.x, v.x, v.y * v.y)) * scale;
    }
}

// ---------------------------------------------------------------------------------------------
// |librosa.stft(y, n_fft, hop_length=hop)|: periodic Hann, center=True with zero padding,
// frame t covers samples [t*hop - n_fft/2, t*hop + n_fft/2), n_frames = 1 + ns/hop, bins 0..n_fft/2
// (SURVEY.md A.1).  A workgroup owns FT consecutive frames of one channel: the samples they cover
// are staged once in LDS, two real frames ride one complex n_fft-point transform, the bins
// [b_lo, b_hi] are written as S[c][bin - b_lo][t] (t fastest) and the maximum over ALL bins and
// frames goes to rowmax[c] (the reference normalises by the full spectrogram's max before it
// slices, detect.py:387,390).
// ---------------------------------------------------------------------------------------------
struct StftDims {
    int ns, n_fft, hop, nframes, FT, b_lo, b_hi;
};

template <bool GENERIC>
__global__ __launch_bounds__(kSpThreads) void stft_mag(RowFftDev F, StftDims d, const float* __restrict__ x,
                                                       float* __restrict__ S, unsigned* __restrict__ rowmax) {
    D4W_DYN_LDS(smem_raw);
    float2* tile = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n_fft = d.n_fft, nb = d.FT / 2;
    // TL: LDS elements per transform -- the frame itself, or the Bluestein convolution length when n_fft has a prime factor
    // > 31 (x chirp -> FFT -> x filter -> inverse FFT -> x chirp, as row_dft; spectra in natural order then)
    const int TL = F.bs_L ? F.bs_L : n_fft;
    const AxisDesc& axt = row_tw_axis(F);
    const TwLds tw = tw_stage(axt, tile + nb * TL, tid, nthr);
    float* seg = reinterpret_cast<float*>(tile + nb * TL + tw_lds_elems(axt));
    const int seg_len = (d.FT - 1) * d.hop + n_fft;
    const int t0 = blockIdx.x * d.FT;
    const int s0 = t0 * d.hop - n_fft / 2;
    const float* xr = x + (size_t)blockIdx.y * d.ns;
    for (int j = tid; j < seg_len; j += nthr) {
        const int s = s0 + j;
        seg[j] = (s >= 0 && s < d.ns) ? xr[s] : 0.f;
    }
    lds_barrier();
    const FDiv dn(TL), dnb(nb);
    for (int w = tid; w < nb * TL; w += nthr) {
        const int b = dn.div(w), n = w - b * TL;
        float2 v = make_float2(0.f, 0.f);
        if (n < n_fft) {
            const float wn = F.hann[n];
            v = make_float2(seg[2 * b * d.hop + n] * wn, seg[(2 * b + 1) * d.hop + n] * wn);
            if (F.bs_L) v = c_mul(v, F.bs_chirp[n]);
        }
        tile[w] = v;
    }
    lds_barrier();
    if (F.bs_L) {
        lds_fft<false, false, false>(tile, F.ax_bs, tw, 1, nb, TL, 1, 0, tid, nthr);
        for (int w = tid; w < nb * TL; w += nthr) tile[w] = c_mul(tile[w], F.bs_filt[w - dn.div(w) * TL]);
        lds_barrier();
        lds_fft<true, false, false>(tile, F.ax_bs, tw, 1, nb, TL, 1, 0, tid, nthr);
        for (int w = tid; w < nb * n_fft; w += nthr) {
            const int b = w / n_fft, k = w - b * n_fft;
            tile[b * TL + k] = c_mul(tile[b * TL + k], F.bs_chirp[k]);
        }
        lds_barrier();
    } else {
        lds_fft<false, false, GENERIC>(tile, F.ax, tw, 1, nb, n_fft, 1, 0, tid, nthr);
    }
    const int nbins = n_fft / 2 + 1, nkeep = d.b_hi - d.b_lo + 1;
    float mx = 0.f;
    for (int w = tid; w < nbins * nb; w += nthr) {
        const int k = dnb.div(w), b = w - k * nb;
        const int tA = t0 + 2 * b;
        if (tA >= d.nframes) continue;
        const float2* tb = tile + b * TL;
        const float2 zk = tb[F.pos[k]], zm = c_conj(tb[F.pos[(k == 0) ? 0 : n_fft - k]]);
        const float2 A = c_scale(c_add(zk, zm), 0.5f);
        const float2 B = c_mul_mi(c_scale(c_sub(zk, zm), 0.5f));
        const float ma = sqrtf(fmaf(A.x, A.x, A.y * A.y)), mb = sqrtf(fmaf(B.x, B.x, B.y * B.y));
        const bool hasB = (tA + 1 < d.nframes);
        mx = fmaxf(mx, ma);
        if (hasB) mx = fmaxf(mx, mb);
        if (k >= d.b_lo && k <= d.b_hi) {
            float* o = S + ((size_t)blockIdx.y * nkeep + (k - d.b_lo)) * d.nframes + tA;
            o[0] = ma;
            if (hasB) o[1] = mb;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) atomicMax(rowmax + blockIdx.y, __float_as_uint(mx));   // mx >= 0: bit order = value order
}

// The same for frame lengths n_fft = RA * RB with both factors register butterflies (160 = 10 x 16, the detector's
// 0.8-s window; 128, 256, 512): one DIF split n = j + RB a, k = k1 + RA k2 --
//   S1 item (pair b, j)  : the RA windowed samples of two real frames (one complex value each), radix RA, x W_N^(j k1) -> LDS
//   S2 item (pair b, k1) : radix RB over j -> Z[k1 + RA k2], back in place
//   then the magnitudes of the kept bins (all bins when the row maximum is wanted) from Z[k] and conj Z[N - k].
// Two LDS round trips per transform instead of the generic kernel's staging pass + three runtime-radix stages:
// 5.5 -> see DESIGN.md 3.4 at 11020 x 12000, n_fft 160, hop 8.
template <int RA, int RB>
__global__ __launch_bounds__(kSpThreads) void stft_fat(RowFftDev F, StftDims d, const float* __restrict__ x,
                                                       float* __restrict__ S, unsigned* __restrict__ rowmax) {
    D4W_DYN_LDS(smem_raw);
    constexpr int N = RA * RB, PB = RB + 1, TP = RA * PB;           // LDS pitch of a k1 row / of a transform
    float2* buf = reinterpret_cast<float2*>(smem_raw);                // [nb][RA][RB + 1]
    const int tid = threadIdx.x, nb = d.FT / 2;
    float2* twl = buf + nb * TP;                                      // [RA][RB] W_N^(j k1)
    float* win = reinterpret_cast<float*>(twl + N);                   // [N]
    float* seg = win + N;
    const int seg_len = (d.FT - 1) * d.hop + N;
    const float* xr = x + (size_t)blockIdx.y * d.ns;
    for (int i = tid; i < N; i += kSpThreads) {
        const int k1 = i / RB, j = i - k1 * RB;
        twl[i] = F.wfull[(j * k1) % N];
        win[i] = F.hann[i];
    }
    float mx = 0.f;
    // a workgroup walks several frame tiles of its row: the twiddle / window tables are staged once, and a long record
    // is a few thousand workgroups instead of half a million
    const int ntiles = (d.nframes + d.FT - 1) / d.FT;
    // the samples of the NEXT tile travel in registers while this one is transformed (kSegPer per lane; longer segments --
    // large hops -- load the remainder directly)
    constexpr int kSegPer = 4;
    float nxt[kSegPer];
    auto seg_fetch = [&](int tile) {
        const int s0 = tile * d.FT * d.hop - N / 2;
#pragma unroll
        for (int k = 0; k < kSegPer; ++k) {
            const int j = tid + k * kSpThreads, sidx = s0 + j;
            nxt[k] = (j < seg_len && sidx >= 0 && sidx < d.ns) ? xr[sidx] : 0.f;
        }
    };
    if ((int)blockIdx.x < ntiles) seg_fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * d.FT;
    const int s0 = t0 * d.hop - N / 2;
#pragma unroll
    for (int k = 0; k < kSegPer; ++k) {
        const int j = tid + k * kSpThreads;
        if (j < seg_len) seg[j] = nxt[k];
    }
    for (int j = tid + kSegPer * kSpThreads; j < seg_len; j += kSpThreads) {
        const int sidx = s0 + j;
        seg[j] = (sidx >= 0 && sidx < d.ns) ? xr[sidx] : 0.f;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) seg_fetch(tile + gridDim.x);
    for (int it = tid; it < nb * RB; it += kSpThreads) {              // S1
        const int b = it / RB, j = it - b * RB;
        const float* sa = seg + 2 * b * d.hop + j;
        const float* sb = sa + d.hop;
        float2 z[RA];
        static_for<RA>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            const float wn = win[j + RB * a];
            z[a] = make_float2(sa[RB * a] * wn, sb[RB * a] * wn);
        });
        dft<RA>(z);
        float2* o = buf + b * TP + j;
        static_for<RA>([&](auto kk) {
            constexpr int k1 = decltype(kk)::value;
            o[k1 * PB] = (k1 == 0) ? z[0] : c_mul(z[k1], twl[k1 * RB + j]);
        });
    }
    __syncthreads();
    for (int it = tid; it < nb * RA; it += kSpThreads) {              // S2
        const int b = it / RA, k1 = it - b * RA;
        float2* r = buf + b * TP + k1 * PB;
        float2 v[RB];
        static_for<RB>([&](auto jj) { v[decltype(jj)::value] = r[decltype(jj)::value]; });
        dft<RB>(v);
        static_for<RB>([&](auto kk) { r[decltype(kk)::value] = v[decltype(kk)::value]; });
    }
    __syncthreads();
    // Z[k] of transform b sits at buf[b][(k % RA)][k / RA]
    const int klo = rowmax ? 0 : d.b_lo, khi = rowmax ? N / 2 : d.b_hi, nk = khi - klo + 1, nkeep = d.b_hi - d.b_lo + 1;
    for (int w = tid; w < nk * nb; w += kSpThreads) {
        const int kk = w / nb, b = w - kk * nb, k = klo + kk;
        const int tA = t0 + 2 * b;
        if (tA >= d.nframes) continue;
        const float2* tb = buf + b * TP;
        const int km = (k == 0) ? 0 : N - k;
        const float2 zk = tb[(k % RA) * PB + k / RA], zm = c_conj(tb[(km % RA) * PB + km / RA]);
        const float2 A = c_scale(c_add(zk, zm), 0.5f);
        const float2 B = c_mul_mi(c_scale(c_sub(zk, zm), 0.5f));
        const float ma = sqrtf(fmaf(A.x, A.x, A.y * A.y)), mb = sqrtf(fmaf(B.x, B.x, B.y * B.y));
        const bool hasB = (tA + 1 < d.nframes);
        mx = fmaxf(mx, ma);
        if (hasB) mx = fmaxf(mx, mb);
        if (k >= d.b_lo && k <= d.b_hi) {
            float* o = S + ((size_t)blockIdx.y * nkeep + (k - d.b_lo)) * d.nframes + tA;
            o[0] = ma;
            if (hasB) o[1] = mb;
        }
    }
    __syncthreads();                                              // seg / buf are refilled by the next tile
    }
    if (rowmax) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((tid & 63) == 0) atomicMax(rowmax + blockIdx.y, __float_as_uint(mx));   // mx >= 0: bit order = value order
    }
}

// S[c][i] /= denom[c]  (mode 0, detect.py:387)   or   20 log10(S[c][i] / denom[c])  (mode 1, dsp.py:76)
__global__ __launch_bounds__(kSpThreads) void scale_rows(float* __restrict__ S, size_t per_row,
                                                         const float* __restrict__ denom, int mode) {
    float* row = S + (size_t)blockIdx.y * per_row;
    const float dv = denom[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_row; i += (size_t)gridDim.x * blockDim.x) {
        const float v = row[i] / dv;
        row[i] = mode ? 20.0f * log10f(v) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// np.median of every row (detect.py:600 divides by the median of the sliced spectrogram):
// radix select on order-preserving keys in THREE sweeps of the row -- digits of 11, 11 and 10 bits with an LDS histogram of
// 2048 bins (8 per thread).  The sweeps are what the kernel costs: the detector's rows (13 bins x 1501 frames, 78 KB) do
// not stay in L2 between sweeps (FETCH_SIZE of the former 4 x 8-bit + 1 form: 3.4 GB for an 857-MB spectrogram,
// profiles/r04h/pmc_stream_11020x12000.txt; staging the row in LDS, a sampled one-sweep form and one wave per row were
// all slower, profiles/r04h/README.txt).  For an even count the upper middle value comes out of the last sweep as well: it
// is the lower one again when duplicates cover the next rank, else the next non-empty bin of the last histogram, else the
// smallest key above the selected 22-bit prefix, which the last sweep tracks.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned med_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float med_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int kMedBins = 2048, kMedPer = kMedBins / kSpThreads;
__global__ __launch_bounds__(kSpThreads) void row_median(const float* __restrict__ v, size_t n,
                                                         float* __restrict__ med) {
    __shared__ unsigned hist[kMedBins];
    __shared__ unsigned wtot[kSpThreads / 64];
    __shared__ unsigned s_bin, s_k, s_excl, s_next, s_above;
    static_assert(kMedPer * kSpThreads == kMedBins, "whole bins per thread");
    const float* row = v + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    unsigned prefix = 0u, mask = 0u, kk = (unsigned)((n - 1) / 2);
    unsigned above = 0xFFFFFFFFu;                                  // smallest key beyond the selected prefix (last sweep)
    unsigned bin = 0u, excl_sel = 0u, h_sel = 0u;
    constexpr int kAhead = 8;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = (pass == 0) ? 21 : (pass == 1) ? 10 : 0;
        const unsigned dmask = (pass == 2) ? 1023u : 2047u;
        for (int i = tid; i < kMedBins; i += kSpThreads) hist[i] = 0u;
        if (tid == 0) { s_next = 0xFFFFFFFFu; s_above = 0xFFFFFFFFu; }
        __syncthreads();
        // first sweep: every value takes part and the top digit (sign, exponent, two mantissa bits) of a row of magnitudes
        // falls into a handful of bins -- as LDS atomics a many-way same-address conflict per wave.  Four bins around the first
        // value's are counted in registers and added once per wave.
        const unsigned hot = (pass == 0) ? (med_key(row[0]) >> 21) - 1u : 0xFFFFF000u;
        unsigned priv[4] = {0u, 0u, 0u, 0u};
        for (size_t i0 = tid; i0 < n; i0 += (size_t)kAhead * kSpThreads) {
            float q[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const size_t i = i0 + (size_t)j * kSpThreads;
                q[j] = (i < n) ? row[i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                if (i0 + (size_t)j * kSpThreads < n) {
                    const unsigned k = med_key(q[j]);
                    if ((k & mask) == prefix) {
                        const unsigned bq = (k >> shift) & dmask, d = bq - hot;
                        if (d < 4u) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) priv[t] += (d == (unsigned)t) ? 1u : 0u;
                        } else {
                            atomicAdd(&hist[bq], 1u);
                        }
                    } else if (pass == 2 && (k & mask) > prefix) {
                        above = min(above, k);
                    }
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned c = priv[t];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
                if ((tid & 63) == 0 && c && hot + (unsigned)t < (unsigned)kMedBins) atomicAdd(&hist[hot + (unsigned)t], c);
            }
        }
        __syncthreads();
        // the bin holding rank kk: kMedPer bins per thread, inclusive scan of the threads' sums
        unsigned h[kMedPer], sum = 0u;
#pragma unroll
        for (int t = 0; t < kMedPer; ++t) { h[t] = hist[kMedPer * tid + t]; sum += h[t]; }
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            if ((tid & 63) >= off) incl += up;
        }
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < (tid >> 6); ++w) before += wtot[w];
        unsigned run = incl + before - sum;                        // keys in the bins before this thread's
#pragma unroll
        for (int t = 0; t < kMedPer; ++t) {
            if (kk >= run && kk < run + h[t]) {                   // exactly one (thread, t): the counts sum to more than kk
                s_bin = (unsigned)(kMedPer * tid + t);
                s_k = kk - run;
                s_excl = run;
            }
            run += h[t];
        }
        __syncthreads();
        bin = s_bin;
        if (pass == 2) {
            excl_sel = s_excl;
            h_sel = hist[bin];
            // the next non-empty bin of the last digit, and the smallest key beyond the prefix
#pragma unroll
            for (int t = 0; t < kMedPer; ++t) {
                const unsigned bq = (unsigned)(kMedPer * tid + t);
                if (bq > bin && h[t]) atomicMin(&s_next, bq);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) above = min(above, __shfl_xor(above, off));
            if ((tid & 63) == 0) atomicMin(&s_above, above);
        }
        kk = s_k;
        prefix |= bin << shift;
        mask |= dmask << shift;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned a = prefix;
        unsigned b = a;
        if ((n & 1) == 0) {
            // rank (n - 1) / 2 is key a, the kk-th of its h_sel copies; rank n / 2 is a again unless that was the last copy
            (void)excl_sel;
            if (kk + 1 >= h_sel) b = (s_next != 0xFFFFFFFFu) ? ((prefix & ~1023u) | s_next) : s_above;
        }
        med[blockIdx.x] = (n & 1) ? med_unkey(a) : 0.5f * (med_unkey(a) + med_unkey(b));
    }
}

// ---------------------------------------------------------------------------------------------
// spectrogram x kernel correlation along time, summed over frequency:
//   raw[c][t] = sum_f sum_j S[c][f][t + j - off] K[f][j]   (S = 0 outside [0, nt)),  t < nout
//   out = max(raw, 0) / (med[c] * nk)                      (detect.py:597-600)
// off = nk/2 reproduces fftconvolve(S, flip(K, 1), 'same', axes=1); off = 0 with nout = nt-nk+1
// and zero_ends reproduces detect.xcorr (detect.py:632-644: first and last value forced to 0).
// ---------------------------------------------------------------------------------------------
constexpr int kScThreads = 128, kScPer = 4, kScTile = kScThreads * kScPer;      // 512 correlation lags per workgroup
constexpr int kScLdsFloats = 2560;          // a few strip rows at a time: a small footprint keeps ~30 waves on a CU
__host__ __device__ constexpr int sc_width(int nk) { return (kScTile + nk + 2 + 3) & ~3; }

// Four consecutive lags per lane: a lane reads its window of a strip row in 16-byte pieces (conflict-free) and every piece
// feeds 16 FMAs -- a quarter of the LDS reads of one lag per lane -- with the 7 kernel taps a piece meets as wave-uniform
// scalars; the strip itself is filled with eight global loads in flight per lane.
// The strip of the NEXT frequency rows is already in registers while this one is worked on (one exposed load latency per
// workgroup instead of one per chunk: the kernel was bound by those waits and by an integer division per staged sample, 0.44 ms
// for a 0.86-GB spectrogram at 11020 x 12000; round 5).
// FL strip rows of M pieces per lane and chunk (FL x M x kScThreads >= FL x width staged samples: 4 x 5 for kernels of up to
// ~120 frames, 2 x 10, 1 x 20 beyond), chosen by the host from the kernel's length.
template <int FL, int M>
__global__ __launch_bounds__(kScThreads) void spectro_corr(const float* __restrict__ S, int nf, int nt,
                                                           const float* __restrict__ K, int nk, int off, int nout,
                                                           const float* __restrict__ med, int zero_ends,
                                                           float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float strip[kScLdsFloats];
    static_assert(FL * M * kScThreads <= kScLdsFloats + FL * kScThreads, "a chunk fits the strip");
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * kScTile;
    const int width = sc_width(nk);
    constexpr int fchunk = FL;
    const float* Sc = S + (size_t)blockIdx.y * nf * nt;
    float acc[kScPer] = {0.f, 0.f, 0.f, 0.f};
    float q[FL][M];
    auto fetch = [&](int f0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int j = tid + m * kScThreads, sidx = t0 + j - off;
            const bool in = j < width && sidx >= 0 && sidx < nt;
#pragma unroll
            for (int fl = 0; fl < FL; ++fl) q[fl][m] = (in && f0 + fl < nf) ? Sc[(size_t)(f0 + fl) * nt + sidx] : 0.f;
        }
    };
    auto stash = [&](int f0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int j = tid + m * kScThreads;
#pragma unroll
            for (int fl = 0; fl < FL; ++fl)
                if (j < width && f0 + fl < nf) strip[fl * width + j] = q[fl][m];
        }
    };
    fetch(0);
    for (int f0 = 0; f0 < nf; f0 += fchunk) {
        const int fn = min(fchunk, nf - f0);
        stash(f0);
        __syncthreads();
        if (f0 + fchunk < nf) fetch(f0 + fchunk);                 // in flight during the products below
        for (int fl = 0; fl < fn; ++fl) {
            const float* kr = K + (size_t)(f0 + fl) * nk;
            const float4* sr4 = reinterpret_cast<const float4*>(strip + fl * width) + tid;
            for (int c = 0; 4 * c < nk + 3; ++c) {
                const float4 v4 = sr4[c];
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                // sample e of the piece sits at lag offset 4 c + e: tap j = 4 c + e - q for lag q of this lane.  Taps outside
                // [0, nk) are skipped (wave-uniform branch), not multiplied by zero: a NaN next to the window stays out
#pragma unroll
                for (int d = 0; d < 7; ++d) {
                    const int j = 4 * c - 3 + d;
                    if (j >= 0 && j < nk) {
                        const float tap = kr[j];
#pragma unroll
                        for (int qq = 0; qq < kScPer; ++qq) {
                            const int e = d - 3 + qq;
                            if (e >= 0 && e < 4) acc[qq] = fmaf(v[e], tap, acc[qq]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int qq = 0; qq < kScPer; ++qq) {
        const int t = t0 + kScPer * tid + qq;
        if (t < nout) {
            float v = acc[qq] / (med[blockIdx.y] * (float)nk);
            if (zero_ends && (t == 0 || t == nout - 1)) v = 0.f;
            if (v < 0.f) v = 0.f;                                    // NaN (0/0 on an all-zero row) passes through
            out[(size_t)blockIdx.y * nout + t] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scipy.signal.find_peaks(x, prominence=thr)[0] per row  (detect.py:192,217,271)
//   _local_maxima_1d: a plateau whose left neighbour is lower and whose right neighbour is lower is
//   a peak at the plateau's middle sample (floor); the first and last sample never are.
//   _peak_prominences (wlen=None): walk left / right from the peak while samples are <= the peak,
//   tracking the minimum; prominence = peak - max(left_min, right_min); keep prominence >= thr.
// One workgroup per row; a thread owns a candidate left edge, accepted peaks are written in time
// order through a workgroup prefix sum.  idx[row][0..min(count, cap)) ; counts[row] = count.
// ---------------------------------------------------------------------------------------------
// Prominence walks use a two-level (max, min) summary of the whole row held in LDS (blocks of 2^b
// samples and super-blocks of 32 blocks): a walk steps sample by sample only to the edge of its block,
// block by block to the edge of its super-block, then over super-blocks while their maximum does not
// exceed the peak, and descends into the one block that stops it -- a few dozen dependent LDS reads
// instead of O(row) dependent loads (a smooth envelope of 120 000 samples has walks of thousands of
// samples, and in SIMT the longest walk of a wave is what every lane pays).  Rows of up to kFpRowLds
// samples are staged in LDS once (the sample steps are LDS reads); longer rows are read in place
// (L2-resident after the summary sweep).  Candidates are processed without barriers (a thread strides
// the row and marks its accepted peaks in an LDS bitmap); the time-ordered index list is produced at
// the end by a popcount prefix sum over the bitmap.
constexpr int kFpMaxBlocks = 4096;
constexpr int kFpRowLds = 16384;
constexpr int kFpThreads = 512;                // two 70-KB rows fit a CU: 16 waves instead of 8
constexpr int kFpFan = 32;                      // blocks per super-block (the walk code shifts by 5)
static_assert(kFpFan == 32, "fp_walk shifts by 5");

// min over the walk from q in direction DIR until the first sample > v (exclusive) or the row end.
// Every phase reads a BATCH of operands with independent loads before it looks at them: a walk is a
// chain of dependent decisions, and one LDS round trip per step is what it would otherwise cost.
constexpr int kFpBatch = 8;        // samples / summary entries read per dependent LDS round trip of a walk
constexpr int kFpBlkBatch = 4;

// Only the decision prominence >= thr is needed, not the prominence itself: a side is SATISFIED as soon
// as the running minimum reaches lim = the largest float <= v - thr, and the walk may stop there
// (the minimum can only decrease further).  Prominent peaks therefore walk only to their first deep
// dip and small ones to the next higher sample -- nobody walks across the row.
// scan `n` samples from q in direction DIR: 1 = a sample > v stopped it, 2 = satisfied, 0 = ran out
template <int DIR>
__device__ __forceinline__ int fp_scan_samples(const float* __restrict__ r, int ns, int& q, int n, float v, float lim,
                                               float& lmin) {
    for (int done = 0; done < n; done += kFpBatch) {
        float u[kFpBatch];
#pragma unroll
        for (int k = 0; k < kFpBatch; ++k) u[k] = r[min(max(q + DIR * k, 0), ns - 1)];   // clamped: speculative tail
        const int m = min(kFpBatch, n - done);
#pragma unroll
        for (int k = 0; k < kFpBatch; ++k) {
            if (k < m) {
                if (u[k] > v) { q += DIR * k; return 1; }
                lmin = fminf(lmin, u[k]);
                if (lmin <= lim) return 2;
            }
        }
        q += DIR * m;
    }
    return 0;
}

// scan `n` summary entries (stride `step` samples each) from q: 1 = an entry's max > v stopped it (q at
// that entry), 2 = satisfied, 0 = ran out
template <int DIR>
__device__ __forceinline__ int fp_scan_blocks(const float2* __restrict__ sm, int nent, int shift, int& q, int n,
                                              int step, float v, float lim, float& lmin) {
    for (int done = 0; done < n; done += kFpBlkBatch) {
        float2 e[kFpBlkBatch];
#pragma unroll
        for (int k = 0; k < kFpBlkBatch; ++k) e[k] = sm[min(max((q + DIR * k * step) >> shift, 0), nent - 1)];
        const int m = min(kFpBlkBatch, n - done);
#pragma unroll
        for (int k = 0; k < kFpBlkBatch; ++k) {
            if (k < m) {
                if (e[k].x > v) { q += DIR * k * step; return 1; }
                lmin = fminf(lmin, e[k].y);
                if (lmin <= lim) return 2;
            }
        }
        q += DIR * m * step;
    }
    return 0;
}

// running minimum of the walk from q in direction DIR up to the first sample > v, the row end, or the
// point where it reaches lim (whichever comes first)
template <int DIR>
__device__ __forceinline__ float fp_walk(const float* __restrict__ r, const float2* __restrict__ s1,
                                         const float2* __restrict__ s2, int ns, int nb, int nb2, int bshift, int q,
                                         float v, float lim) {
    const int BS = 1 << bshift, SB = BS * kFpFan;
    float lmin = v;
    // samples to the edge of the block (for DIR < 0 the block's first sample is included)
    int n = (DIR < 0) ? ((q + 1) & (BS - 1)) : ((BS - (q & (BS - 1))) & (BS - 1));
    if (DIR > 0) n = min(n, ns - q);
    if (fp_scan_samples<DIR>(r, ns, q, n, v, lim, lmin)) return lmin;
    if ((DIR < 0) ? (q < 0) : (q >= ns)) return lmin;
    // blocks to the edge of the super-block
    const int blk = q >> bshift;
    n = (DIR < 0) ? ((blk + 1) & (kFpFan - 1)) : ((kFpFan - (blk & (kFpFan - 1))) & (kFpFan - 1));
    if (DIR > 0) n = min(n, nb - blk);
    int st = fp_scan_blocks<DIR>(s1, nb, bshift, q, n, BS, v, lim, lmin);
    if (st == 2) return lmin;
    if (st == 0) {
        if ((DIR < 0) ? (q < 0) : (q >= ns)) return lmin;
        // super-blocks to the row end
        const int sb = q >> (bshift + 5);
        n = (DIR < 0) ? sb + 1 : nb2 - sb;
        if (fp_scan_blocks<DIR>(s2, nb2, bshift + 5, q, n, SB, v, lim, lmin) != 1) return lmin;
        // blocks of the stopping super-block (one of them has a larger sample)
        if (fp_scan_blocks<DIR>(s1, nb, bshift, q, kFpFan, BS, v, lim, lmin) == 2) return lmin;
    }
    // samples of the stopping block
    (void)fp_scan_samples<DIR>(r, ns, q, BS, v, lim, lmin);
    return lmin;
}

// The same decision taken by a whole WAVE for one side of one candidate: lane l looks at entry l of the phase (sample,
// block summary or super-block summary), two ballots give the first entry that stops the walk (max > v) and the first one
// that satisfies it (min <= lim), and the earlier of the two decides -- one LDS round trip and ~30 instructions per phase
// where the lane-per-side walk takes a round trip and ~150 instructions per 4-8 entries.  A row whose candidates are few
// (an envelope with thr a fraction of its strongest peak: ~10 per row) leaves 60 of 64 lanes idle in fp_walk and still
// pays its issue slots in every wave of the workgroup: 23 of 78 thousand cycles per row at 11020 x 12000
// (scripts/probe/fp_timing.py); fp_scan runs this form when the sides are few.  Returns whether the running minimum
// reaches lim before a sample > v or the row end (fp_walk(...) <= lim); q, v, lim are wave-uniform.
// W lanes per side (W = 16: four sides per wave advance together; sides in different phases diverge and reconverge
// after every phase): gl = lane within the group, gs = the group's first lane.
template <int W>
__device__ __forceinline__ int fp_group_phase(const float hi, const float lo, bool in, float v, float lim, int gs, int& first) {
    constexpr unsigned long long kMask = (W == 64) ? ~0ull : ((1ull << (W & 63)) - 1ull);
    const unsigned long long h = (__ballot(in && hi > v) >> gs) & kMask, sfy = (__ballot(in && lo <= lim) >> gs) & kMask;
    const int fh = h ? __builtin_ctzll(h) : 64, fs = sfy ? __builtin_ctzll(sfy) : 64;
    first = fh;
    return fs < fh ? 2 : (h ? 1 : 0);
}

template <int W>
__device__ __forceinline__ bool fp_walk_wave(const float* __restrict__ r, const float2* __restrict__ s1,
                                             const float2* __restrict__ s2, int ns, int nb, int nb2, int bshift, int DIR,
                                             int q, float v, float lim, int lane) {
    const int BS = 1 << bshift, SB = BS * kFpFan;
    const int gl = lane & (W - 1), gs = lane & ~(W - 1);
    if (v <= lim) return true;                                // thr <= 0: the peak itself is its base
    auto samples = [&](int n) -> int {                        // 2 satisfied, 1 stopped (q at the stopper), 0 ran out (q past them)
        for (int done = 0; done < n; done += W) {
            const int j = q + DIR * gl, m = min(W, n - done);
            const bool in = gl < m && j >= 0 && j < ns;
            const float u = r[min(max(j, 0), ns - 1)];
            int first;
            const int st = fp_group_phase<W>(u, u, in, v, lim, gs, first);
            if (st == 2) return 2;
            if (st == 1) { q += DIR * first; return 1; }
            q += DIR * m;
        }
        return 0;
    };
    auto blocks = [&](const float2* sm, int nent, int shift, int n, int step) -> int {
        for (int done = 0; done < n; done += W) {
            const int e = (q + DIR * gl * step) >> shift, m = min(W, n - done);
            const bool in = gl < m && e >= 0 && e < nent;
            const float2 sv = sm[min(max(e, 0), nent - 1)];
            int first;
            const int st = fp_group_phase<W>(sv.x, sv.y, in, v, lim, gs, first);
            if (st == 2) return 2;
            if (st == 1) { q += DIR * first * step; return 1; }
            q += DIR * m * step;
        }
        return 0;
    };
    // the phases of fp_walk, entry for entry
    int n = (DIR < 0) ? ((q + 1) & (BS - 1)) : ((BS - (q & (BS - 1))) & (BS - 1));
    if (DIR > 0) n = min(n, ns - q);
    int st = samples(n);
    if (st) return st == 2;
    if ((DIR < 0) ? (q < 0) : (q >= ns)) return false;
    const int blk = q >> bshift;
    n = (DIR < 0) ? ((blk + 1) & (kFpFan - 1)) : ((kFpFan - (blk & (kFpFan - 1))) & (kFpFan - 1));
    if (DIR > 0) n = min(n, nb - blk);
    st = blocks(s1, nb, bshift, n, BS);
    if (st == 2) return true;
    if (st == 0) {
        if ((DIR < 0) ? (q < 0) : (q >= ns)) return false;
        const int sb = q >> (bshift + 5);
        n = (DIR < 0) ? sb + 1 : nb2 - sb;
        st = blocks(s2, nb2, bshift + 5, n, SB);
        if (st != 1) return st == 2;                                        // satisfied, or the row end without a base
        if (blocks(s1, nb, bshift, kFpFan, BS) == 2) return true;         // blocks of the stopping super-block
    }
    return samples(BS) == 2;
}

// LDS tables of the picker for one row (after `lead` bytes the caller uses itself)
struct FpLds {
    float2* s1;        // [nb]  (max, min) of every block
    float2* s2;        // [nb2] (max, min) of every super-block
    unsigned* bits;    // [nwords] accepted peaks
    unsigned* cand;    // [nwords] rising edges of the maxima worth a walk
    int* clist;        // [kFpList] their positions, one round of bitmap words at a time
    float* rowl;       // [ns] the row (STAGED)
};

__device__ __forceinline__ void fp_summaries2(const FpLds& T, int nb, int nb2, int tid) {
    // one block summary per lane, 32-lane shuffle reduction (kFpFan = 32 = half a wave)
    for (int base = 0; base < nb; base += kFpThreads) {                  // wave-uniform trip count: every lane shuffles
        const int k = base + tid;
        float mx = -INFINITY, mn = INFINITY;
        if (k < nb) {
            const float2 e = T.s1[k];
            mx = e.x;
            mn = e.y;
        }
#pragma unroll
        for (int off = 1; off < kFpFan; off <<= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, off));
            mn = fminf(mn, __shfl_xor(mn, off));
        }
        if ((k & (kFpFan - 1)) == 0 && k < nb) T.s2[k / kFpFan] = make_float2(mx, mn);
    }
    (void)nb2;
}

// block summaries of a row held in `r` (LDS or global): eight blocks per wave iteration (their loads are independent
// and in flight together), lanes stride a block, wave shuffle reduction
__device__ __forceinline__ void fp_summaries1(const float* __restrict__ r, const FpLds& T, int ns, int nb, int bshift, int tid) {
    const int BS = 1 << bshift, lane = tid & 63, wave = tid >> 6;
    constexpr int kBatch = 8;
    for (int bk0 = wave * kBatch; bk0 < nb; bk0 += (kFpThreads / 64) * kBatch) {
        float mx[kBatch], mn[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            mx[j] = -INFINITY;
            mn[j] = INFINITY;
            const int lo = (bk0 + j) << bshift, hi = min(lo + BS, ns);
            for (int i = lo + lane; i < hi; i += 64) {
                const float u = r[i];
                mx[j] = fmaxf(mx[j], u);
                mn[j] = fminf(mn[j], u);
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], off));
                mn[j] = fminf(mn[j], __shfl_xor(mn[j], off));
            }
            if (lane == 0 && bk0 + j < nb) T.s1[bk0 + j] = make_float2(mx[j], mn[j]);
        }
    }
}

// stage the row into LDS and form the 32-sample block summaries in the same sweep: a lane loads four consecutive
// samples (16 bytes), eight neighbouring lanes cover one block and reduce with three shuffle steps
template <bool STORE>     // STORE: keep the row in T.rowl (rows that fit); otherwise only the summaries are formed
__device__ __forceinline__ void fp_stage_rows4(const float* __restrict__ rg, const FpLds& T, int ns, int nb, int tid) {
    const int ns4 = ns >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(rg);
    float4* l4 = reinterpret_cast<float4*>(T.rowl);
    constexpr int kAhead = 12;                                          // loads in flight per lane: the sweep is latency-bound
    for (int base = 0; base < 8 * nb; base += kAhead * kFpThreads) {    // wave-uniform trip count: every lane shuffles
        float4 q[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int v4 = base + k * kFpThreads + tid;
            q[k] = (v4 < ns4) ? g4[v4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int v4 = base + k * kFpThreads + tid;
            float mx = -INFINITY, mn = INFINITY;
            if (v4 < ns4) {
                if (STORE) l4[v4] = q[k];
                mx = fmaxf(fmaxf(q[k].x, q[k].y), fmaxf(q[k].z, q[k].w));
                mn = fminf(fminf(q[k].x, q[k].y), fminf(q[k].z, q[k].w));
            }
#pragma unroll
            for (int off = 1; off <= 4; off <<= 1) {
                mx = fmaxf(mx, __shfl_xor(mx, off));
                mn = fminf(mn, __shfl_xor(mn, off));
            }
            if ((v4 & 7) == 0 && (v4 >> 3) < nb) T.s1[v4 >> 3] = make_float2(mx, mn);
        }
    }
}

// Rows too long for LDS, after the summary sweep: the maxima that can reach the threshold at all (v - thr not below the
// row minimum) are looked for in a second sweep that passes the row through LDS in windows of kFpSegW samples (a core of
// kFpSegS plus kFpSegH either side, in the space the candidate list occupies later; the next windows are already in
// registers while this one is worked on).  A window with a handful of such maxima (an envelope with thr a fraction of its
// strongest peak) just marks them in T.cand for the summary walks of fp_scan.  A window full of them (a raw correlogram
// has a maximum every ~9 samples, 14 000 per 120 000-sample row, and its minimum is as deep as its maximum is high) is
// settled on the spot: the window is COMPACTED to its turning points E (samples not below both or not above both
// neighbours -- the minimum of any stretch and the first sample above any level survive the compaction, so a walk over E
// decides exactly what a walk over the samples decides), the maxima of the core are listed, and lane t takes side t & 1 of
// maximum t >> 1 and walks E: two entries per oscillation instead of its ~9 samples, both sides of a maximum at once, one
// maximum per lane, LDS reads only; almost all are settled within two or three periods.  Accepted peaks go to T.bits
// directly; a maximum whose walk leaves the window or runs past kFpSegSteps entries, a plateau, and every maximum of a
// window with more turning points than the lists hold is marked in T.cand.  Samples outside the row read as +inf: a walk
// stops there, as at the row end.
constexpr int kFpSegH = 64;
constexpr int kFpSegW = 2048;
constexpr int kFpSegS = kFpSegW - 2 * kFpSegH;          // 1920 samples
constexpr int kFpSegE = 1024;                           // turning points held per window
constexpr int kFpSegC = 512;                            // maxima held per window
constexpr int kFpSegSteps = 8;                          // entries of E a side may walk
constexpr int kFpSegFew = 24;                           // lanes with a maximum up to which a window is not worth compacting
constexpr int kFpSegMax = (kFpMaxBlocks * 32 + kFpSegS - 1) / kFpSegS;   // windows of the longest row with 32-sample blocks
static_assert(kFpSegS % 32 == 0, "a window core is whole summary blocks");
static_assert(kFpSegW == 4 * kFpThreads && kFpSegH % 4 == 0, "window geometry: one 16-byte load per lane");

// one side of maximum v over the turning points: 2 = the running minimum reached lim before a higher entry, 1 = a higher
// entry (or the row end) came first, 0 = not decided inside the window
template <int DIR>
__device__ __forceinline__ int fp_turning_side(const float* __restrict__ E, int nE, int idx, float v, float lim) {
    float lmin = INFINITY;
    for (int done = 0; done < kFpSegSteps; done += 4) {
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = E[min(max(idx + DIR * k, 0), nE - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = idx + DIR * k;
            if (j < 0 || j >= nE) return 0;
            if (e[k] > v) return 1;
            lmin = fminf(lmin, e[k]);
            if (lmin <= lim) return 2;
        }
        idx += DIR * 4;
    }
    return 0;
}

// Returns the row position from which the maxima have NOT been looked at: the first two windows with something in them
// decide for the row -- when both hold only a handful of maxima the rest is left to the barrier-free marking sweep of fp_scan.
__device__ __forceinline__ int fp_sweep_segments(const float* __restrict__ rg, const FpLds& T, int ns, int nb, double thr,
                                                 float gmin, int* wave_tot, unsigned char* hot, int tid) {
    float* seg = reinterpret_cast<float*>(T.clist);
    float4* l4 = reinterpret_cast<float4*>(T.clist);
    float* E = seg + kFpSegW;                                              // [kFpSegE]
    unsigned* C = reinterpret_cast<unsigned*>(E + kFpSegE);               // [kFpSegC]  window position | plateau << 15 | index in E << 16
    unsigned char* res = reinterpret_cast<unsigned char*>(C + kFpSegC);   // [2 kFpSegC]
    int* seg_tot = reinterpret_cast<int*>(res + 2 * kFpSegC);             // [kFpThreads / 64]
    const float4* g4 = reinterpret_cast<const float4*>(rg);
    const int ns4 = ns >> 2, lane = tid & 63, wave = tid >> 6;
    const int nseg = (ns + kFpSegS - 1) / kFpSegS;
    const float4 inf4 = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    // windows whose core holds no sample that could reach the threshold are neither loaded nor looked at
    for (int k = tid; k < kFpSegMax; k += kFpThreads) hot[k] = 0;
    __syncthreads();
    for (int bk = tid; bk < nb; bk += kFpThreads)
        if (!((double)T.s1[bk].x - thr < (double)gmin)) hot[bk / (kFpSegS / 32)] = 1;
    __syncthreads();
    auto fetch = [&](int s) -> float4 {
        const int v4 = (s * kFpSegS - kFpSegH) / 4 + tid;
        return (s < nseg && hot[s] && v4 >= 0 && v4 < ns4) ? g4[v4] : inf4;
    };
    float4 r0 = fetch(0), r1 = fetch(1), r2 = fetch(2);        // three windows in flight per lane
    int few = 0, full = 0, resume = ns;
    for (int s = 0; s < nseg; ++s) {
        const int w0 = s * kFpSegS - kFpSegH;                  // row position of the window's first sample
        const float4 cur = r0;
        r0 = r1;
        r1 = r2;
        r2 = fetch(s + 3);
        if (!hot[s]) continue;
        const int c = 4 * tid, i0 = w0 + c;
        const bool core = tid >= kFpSegH / 4 && tid < (kFpSegH + kFpSegS) / 4;
        l4[tid] = cur;
        __syncthreads();
        // maxima worth a walk among the lane's four samples
        const float u[6] = {c ? seg[c - 1] : INFINITY, cur.x, cur.y, cur.z, cur.w, (c + 4 < kFpSegW) ? seg[c + 4] : INFINITY};
        unsigned m = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            const float v = u[k + 1];
            if (core && i >= 1 && i < ns - 1 && u[k] < v && !(u[k + 2] > v) && !((double)v - thr < (double)gmin)) m |= 1u << k;
        }
        const unsigned long long busy = __ballot(m != 0u);
        if (lane == 0) wave_tot[wave] = __popcll(busy);
        __syncthreads();
        int nbusy = 0;
        for (int k = 0; k < kFpThreads / 64; ++k) nbusy += wave_tot[k];
        if (nbusy <= kFpSegFew) {                                // (uniform) nothing to gain from the lists
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((m >> k) & 1u) atomicOr(&T.cand[(i0 + k) >> 5], 1u << ((i0 + k) & 31));
            if (++few == 2 && !full) {
                resume = (s + 1) * kFpSegS;
                break;
            }
            continue;                                            // (the window and wave_tot are next written behind a barrier each)
        }
        ++full;
        unsigned em = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            const float v = u[k + 1];
            const bool hi = v >= u[k] && v >= u[k + 2], lo = v <= u[k] && v <= u[k + 2];
            if (hi || lo || i == 0 || i == ns - 1) em |= 1u << k;
        }
        const int mine = __popc(em) | (__popc(m) << 16);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(incl, off);
            if (lane >= off) incl += n;
        }
        if (lane == 63) seg_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < kFpThreads / 64; ++k) {
            if (k < wave) before += seg_tot[k];
            total += seg_tot[k];
        }
        const int nE = total & 0xFFFF, nC = total >> 16;
        const bool listed = nE <= kFpSegE && nC <= kFpSegC;
        if (listed) {
            int pe = (before + incl - mine) & 0xFFFF, pc = (before + incl - mine) >> 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((em >> k) & 1u) {
                    if ((m >> k) & 1u) C[pc++] = (unsigned)(c + k) | ((unsigned)pe << 16) | ((u[k + 2] == u[k + 1]) ? 0x8000u : 0u);
                    E[pe++] = u[k + 1];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((m >> k) & 1u) atomicOr(&T.cand[(i0 + k) >> 5], 1u << ((i0 + k) & 31));
        }
        __syncthreads();
        if (listed) {
            for (int t = tid; t < 2 * nC; t += kFpThreads) {
                const unsigned cd = C[t >> 1];
                const int e = (int)(cd >> 16);
                int r = 0;
                if (!(cd & 0x8000u)) {                               // plateaus are left to fp_scan
                    const float v = E[e];
                    const double dl = (double)v - thr;               // as in fp_scan: lim = the largest float u with v - u >= thr
                    float lim = (float)dl;
                    if ((double)lim > dl) lim = nextafterf(lim, -INFINITY);
                    r = (t & 1) ? fp_turning_side<+1>(E, nE, e + 1, v, lim) : fp_turning_side<-1>(E, nE, e - 1, v, lim);
                }
                res[t] = (unsigned char)r;
            }
        }
        __syncthreads();
        if (listed) {
            for (int q = tid; q < nC; q += kFpThreads) {
                const int a = res[2 * q], b = res[2 * q + 1];
                if (a == 1 || b == 1) continue;                      // a higher sample before a deep enough base: not prominent
                const int i = w0 + (int)(C[q] & 0x7FFFu);
                if (a == 2 && b == 2) atomicOr(&T.bits[i >> 5], 1u << (i & 31));
                else atomicOr(&T.cand[i >> 5], 1u << (i & 31));
            }
        }
        // no barrier here: the window, the counts and the lists are each next written behind a later barrier than their last read
    }
    __syncthreads();
    return min(resume, ns);
}

// candidates -> accepted-peak bitmap, in two phases so that the prominence walks run with every lane busy:
//   (1) every thread marks the rising edges of maxima that can reach the threshold at all in the `cand` bitmap -- no
//       base can lie below the row minimum, so a maximum with v - thr below it is rejected here, without a walk (with
//       thr = a fraction of the strongest peak that is almost every maximum);
//   (2) per round of kFpThreads bitmap words the marked positions are listed (popcount prefix sum) and lane c walks
//       candidate c.  Walking inside the marking loop instead made a wave pay the SUM of its lanes' walks (one lane
//       walking, 63 masked): 0.72 of 0.97 ms at 11020 x 12000 with ~15 picks per row.
// r: the row (LDS or global).  Ends with the accepted peaks in T.bits; the last barrier is the caller's.
#ifdef D4W_FP_TIMING      // probe builds only (scripts/probe/fp_timing.py): cycles of workgroup 0's thread 0 between phase marks
__device__ unsigned long long g_fp_t[16];
#define FP_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_fp_t[k], t_ - fp_t0); fp_t0 = t_; } } while (0)
#define FP_T0 unsigned long long fp_t0 = clock64()
#else
#define FP_MARK(k) do {} while (0)
#define FP_T0 do {} while (0)
#endif

constexpr int kFpList = 4096;          // candidates per walk round
constexpr int kFpWaveSides = 256;      // up to this many walk sides per round a group of lanes takes a side (fp_walk_wave)
#ifndef D4W_FP_GROUP
#define D4W_FP_GROUP 16
#endif
constexpr int kFpGroup = D4W_FP_GROUP; // lanes per side
static_assert(kFpSegW + kFpSegE + kFpSegC + kFpSegC / 2 + kFpThreads / 64 <= kFpList, "the window and its lists live in the candidate list");

__device__ __forceinline__ void fp_scan(const float* __restrict__ r, const FpLds& T, int ns, int nb, int nb2, int bshift,
                                        double thr, int nwords, int* wave_tot, unsigned* cfail, bool vec4, int mark_from,
                                        int tid) {
    // mark_from: first sample whose maxima are still to be marked (0: the whole row; > 0: fp_sweep_segments has dealt with
    // the samples before it -- a multiple of 4 -- and left in T.cand only the maxima it could not settle inside their window)
    const int lane = tid & 63, wave = tid >> 6;
    FP_T0;
    float gmin = INFINITY;
    for (int k = 0; k < nb2; ++k) gmin = fminf(gmin, T.s2[k].y);
    int* ccount = wave_tot + kFpThreads / 64;                 // (zeroed by the kernel before its first barrier)
    // 32-sample blocks, the whole row still to mark: a block whose maximum cannot reach the threshold (almost all of them
    // when thr is a fraction of the strongest peak) costs one summary read; the others are looked at by half a wave each, a
    // lane per sample, their bitmap word is stored whole (no zeroing pass, no atomics) and their candidates go straight
    // into the list -- the sample-by-sample marking sweep of the whole row was 8 of 28 thousand cycles of this function at
    // 11020 x 12000 (scripts/probe/fp_timing.py), the separate counting pass 2.5.  (A lane per hot block, 32 samples
    // each, cost the same 8: every wave with one hot lane pays the whole scan.)
    bool by_block = vec4 && bshift == 5 && mark_from == 0;
    if (by_block) {
        // the threshold test in float32: vcut = the smallest float v with !((double)v - thr < (double)gmin) (the test is
        // monotone in v: float -> double is exact, the float64 subtraction rounds monotonically)
        float vcut = (float)((double)gmin + thr);
        for (int k = 0; k < 4 && (double)vcut - thr < (double)gmin; ++k) vcut = nextafterf(vcut, INFINITY);
        for (int k = 0; k < 4; ++k) {
            const float below = nextafterf(vcut, -INFINITY);
            if ((double)below - thr < (double)gmin || below == vcut) break;
            vcut = below;
        }
        const bool vcut_ok = !((double)vcut - thr < (double)gmin) && ((double)nextafterf(vcut, -INFINITY) - thr < (double)gmin);
        // (1) the blocks worth a look, listed from the END of the candidate list (a row whose candidates and hot blocks
        // together exceed the list takes the round-by-round path below, which lists from the bitmap again)
        int* hcount = ccount + 1;
        for (int bk = tid; bk < nwords; bk += kFpThreads) {                     // nwords == nb
            if (!((double)T.s1[bk].x - thr < (double)gmin)) T.clist[kFpList - 1 - atomicAdd(hcount, 1)] = bk;
            else T.cand[bk] = 0u;
        }
        __syncthreads();
        // (2) half a wave per hot block, a lane per sample: one ballot gives the block's bitmap word
        // (a row where a quarter of the blocks are hot -- thr small against the row's range -- is cheaper in the sweep of
        // all samples below: 0.36 against 0.48 ms at thr = 0)
        const int nhot = *hcount, l = lane & 31;
        if (4 * nhot > nb) by_block = false;
        for (int h0 = 0; by_block && h0 < nhot; h0 += kFpThreads / 32) {      // wave-uniform trip count
            const int h = h0 + (tid >> 5);
            const int bk = T.clist[kFpList - 1 - min(h, nhot - 1)];
            const int i = (bk << 5) + l;
            const bool in = h < nhot && i >= 1 && i < ns - 1;
            const float v = r[min(i, ns - 1)], a = r[max(i - 1, 0)], b = r[min(i + 1, ns - 1)];
            const bool pk = in && a < v && !(b > v) && (vcut_ok ? v >= vcut : !((double)v - thr < (double)gmin));
            unsigned m = (unsigned)(__ballot(pk) >> (lane & 32));
            if (l == 0 && h < nhot) {
                T.cand[bk] = m;
                if (m) {
                    const int cnt = __popc(m);
                    int p = atomicAdd(ccount, cnt);
                    if (p + cnt + nhot <= kFpList) {
                        while (m) {
                            const int bit = __builtin_ctz(m);
                            m &= m - 1u;
                            T.clist[p++] = (bk << 5) + bit;
                        }
                    }
                }
            }
        }
    }
    if (!by_block && mark_from == 0) {
        for (int w = tid; w < nwords; w += kFpThreads) T.cand[w] = 0u;
        __syncthreads();
    }
    if (by_block) {
    } else if (vec4) {
        // four samples per lane: one 16-byte read and the two neighbours instead of three reads per sample; four such
        // groups in flight per lane (rows too long for LDS are read from global memory here)
        const float4* r4 = reinterpret_cast<const float4*>(r);
        constexpr int kAhead = 4;
        const int ns4 = ns >> 2;
        for (int g0 = tid + (mark_from >> 2); g0 < ns4; g0 += kAhead * kFpThreads) {
            float4 q[kAhead];
            float lo[kAhead], hi[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int v4 = g0 + j * kFpThreads, i0 = 4 * v4;
                const bool in = v4 < ns4;
                q[j] = in ? r4[v4] : make_float4(0.f, 0.f, 0.f, 0.f);
                lo[j] = (in && i0) ? r[i0 - 1] : INFINITY;
                hi[j] = (in && i0 + 4 < ns) ? r[i0 + 4] : INFINITY;
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int v4 = g0 + j * kFpThreads, i0 = 4 * v4;
                if (v4 >= ns4) continue;
                const float u[6] = {lo[j], q[j].x, q[j].y, q[j].z, q[j].w, hi[j]};
                unsigned m = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = u[k + 1];
                    const int i = i0 + k;
                    if (i >= 1 && i < ns - 1 && u[k] < v && !(u[k + 2] > v) && !((double)v - thr < (double)gmin)) m |= 1u << k;
                }
                if (m) atomicOr(&T.cand[i0 >> 5], m << (i0 & 31));
            }
        }
    } else {
        for (int i = max(1, mark_from) + tid; i < ns - 1; i += kFpThreads) {
            const float v = r[i];
            if (r[i - 1] < v && !(r[i + 1] > v) && !((double)v - thr < (double)gmin)) atomicOr(&T.cand[i >> 5], 1u << (i & 31));
        }
    }
    __syncthreads();
    FP_MARK(4);                                               // marking sweep
    // walks + acceptance of the `total` candidates listed in T.clist (any order: accepted peaks go to a bitmap)
    auto walk_listed = [&](int total) {
        FP_MARK(5);                                           // counting + listing
        for (int w2 = tid; w2 < kFpList / 32; w2 += kFpThreads) cfail[w2] = 0u;
        __syncthreads();
        FP_MARK(6);
        // the left and the right walk of a candidate run on two lanes (a walk is a chain of dependent LDS reads: with
        // ~20 candidates per row most lanes idle anyway); a direction whose base is too high marks the candidate failed
        // ... dealt round-robin to the waves: a wave pays the longest of its lanes' walks in every phase of fp_walk
        if (2 * total <= kFpWaveSides) {
            // few sides: kFpGroup lanes per side (fp_walk_wave), everything uniform within a group
            constexpr int kPerWave = 64 / kFpGroup;
            for (int c2 = wave * kPerWave + lane / kFpGroup; c2 < 2 * total; c2 += (kFpThreads / 64) * kPerWave) {
                const int c = c2 >> 1;
                const int i = T.clist[c];
                const float v = r[i];
                int ia = i + 1;
                while (ia < ns - 1 && r[ia] == v) ++ia;
                bool ok = r[ia] < v && !((double)v - thr < (double)gmin);
                if (ok) {
                    const double dl = (double)v - thr;
                    float lim = (float)dl;
                    if ((double)lim > dl) lim = nextafterf(lim, -INFINITY);
                    ok = fp_walk_wave<kFpGroup>(r, T.s1, T.s2, ns, nb, nb2, bshift, (c2 & 1) ? 1 : -1, (c2 & 1) ? ia : i - 1, v, lim,
                                                lane);
                }
                if (!ok && (lane & (kFpGroup - 1)) == 0) atomicOr(&cfail[c >> 5], 1u << (c & 31));
            }
        } else
        for (int c2 = (tid & 63) * (kFpThreads / 64) + (tid >> 6); c2 < 2 * total; c2 += kFpThreads) {
            const int c = c2 >> 1;
            const int i = T.clist[c];
            const float v = r[i];
            int ia = i + 1;
            while (ia < ns - 1 && r[ia] == v) ++ia;                  // plateau: reported at its middle sample
            bool ok = r[ia] < v && !((double)v - thr < (double)gmin);     // (the second test: maxima fp_sweep_segments passed on)
            if (ok) {
                // float64 like scipy (float32 samples are exact in float64, a float32 subtraction is not):
                // lim = the largest float u with (double)v - (double)u >= thr
                const double dl = (double)v - thr;
                float lim = (float)dl;
                if ((double)lim > dl) lim = nextafterf(lim, -INFINITY);
                if (c2 & 1) ok = !(fp_walk<+1>(r, T.s1, T.s2, ns, nb, nb2, bshift, ia, v, lim) > lim);
                else ok = !(fp_walk<-1>(r, T.s1, T.s2, ns, nb, nb2, bshift, i - 1, v, lim) > lim);    // left base too high
            }
            if (!ok) atomicOr(&cfail[c >> 5], 1u << (c & 31));
        }
        FP_MARK(7);                                           // thread 0's own walk
        __syncthreads();
        FP_MARK(8);                                           // ... and the wait for the longest one
        for (int c = tid; c < total; c += kFpThreads) {
            if ((cfail[c >> 5] >> (c & 31)) & 1u) continue;
            const int i = T.clist[c];
            const float v = r[i];
            int ia = i + 1;
            while (ia < ns - 1 && r[ia] == v) ++ia;
            const int mid = (i + ia - 1) / 2;
            atomicOr(&T.bits[mid >> 5], 1u << (mid & 31));
        }
        __syncthreads();
        FP_MARK(9);                                           // acceptance
    };
    // all candidates of the row in ONE list when they fit (the usual case: a walk phase costs its longest walk, however
    // few lanes walk), else one round of kFpList / 16 bitmap words at a time
    int total_all = 0;
    if (by_block) {
        total_all = *ccount;                                  // listed already when they fit ...
        if (total_all + ccount[1] > kFpList) total_all = kFpList + 1;        // ... beside the hot-block list
    } else {
        int mine = 0;
        for (int w = tid; w < nwords; w += kFpThreads) mine += __popc(T.cand[w]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) wave_tot[wave] = mine;
        if (tid == 0) *ccount = 0;
        __syncthreads();
        for (int k = 0; k < kFpThreads / 64; ++k) total_all += wave_tot[k];
        if (total_all <= kFpList) {
            for (int w = tid; w < nwords; w += kFpThreads) {
                unsigned word = T.cand[w];
                if (!word) continue;
                int p = atomicAdd(ccount, __popc(word));
                while (word) {
                    const int bit = __builtin_ctz(word);
                    word &= word - 1u;
                    T.clist[p++] = (w << 5) + bit;
                }
            }
        }
    }
    if (total_all <= kFpList) {
        walk_listed(total_all);
        return;
    }
    __syncthreads();                                          // wave_tot is reused below
    constexpr int kWordsPerRound = kFpList / 16;              // a word marks at most 16 maxima
    for (int w0 = 0; w0 < nwords; w0 += kWordsPerRound) {
        const int w = w0 + tid;
        unsigned word = (tid < kWordsPerRound && w < nwords) ? T.cand[w] : 0u;
        const int cnt = __popc(word);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(incl, off);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < kFpThreads / 64; ++k) {
            if (k < wave) before += wave_tot[k];
            total += wave_tot[k];
        }
        int p = before + incl - cnt;
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1u;
            T.clist[p++] = (w << 5) + bit;
        }
        walk_listed(total);
    }
}

// time-ordered index list: popcount prefix sum over the bitmap, kFpThreads words per round
__device__ __forceinline__ void fp_emit(const FpLds& T, int nwords, int* __restrict__ orow, int* __restrict__ count, int cap,
                                        int* wave_tot, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int w0 = 0; w0 < nwords; w0 += kFpThreads) {
        const int w = w0 + tid;
        unsigned word = (w < nwords) ? T.bits[w] : 0u;
        const int cnt = __popc(word);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(incl, off);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < kFpThreads / 64; ++k) {
            if (k < wave) before += wave_tot[k];
            total += wave_tot[k];
        }
        int p = base + before + incl - cnt;
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1u;
            if (p < cap) orow[p] = (w << 5) + bit;
            ++p;
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

__device__ __forceinline__ FpLds fp_lds(unsigned char* p, int nb, int nb2, int nwords) {
    FpLds T;
    T.s1 = reinterpret_cast<float2*>(p);
    T.s2 = T.s1 + nb;
    const int nw4 = (nwords + 3) & ~3;
    T.bits = reinterpret_cast<unsigned*>(T.s1 + ((nb + nb2 + 1) & ~1));   // every table starts on 16 bytes:
    T.cand = T.bits + nw4;                                                // the row is staged with 16-byte accesses
    T.clist = reinterpret_cast<int*>(T.cand + nw4);
    T.rowl = reinterpret_cast<float*>(T.clist + kFpList);
    return T;
}

template <bool STAGED>
__global__ __launch_bounds__(kFpThreads) void find_peaks_prom(const float* __restrict__ x, int ns, double thr0,
                                                              int bshift, int* __restrict__ idx,
                                                              int* __restrict__ counts, int cap,
                                                              const float* __restrict__ thr_dev) {
    // thr_dev: the threshold is thr0 x a DEVICE value (d4w_find_peaks_dthr_f32: "0.45 of the largest correlation" without the
    // host waiting for that maximum), formed in float64 like the host's 0.45 * float(c.max())
    const double thr = thr_dev ? thr0 * (double)*thr_dev : thr0;
    D4W_DYN_LDS(smem_raw);
    __shared__ int wave_tot[kFpThreads / 64 + 2];              // + the candidate and hot-block counters of fp_scan
    __shared__ unsigned cfail[kFpList / 32];
    __shared__ unsigned char hot[kFpSegMax];                   // fp_sweep_segments: windows worth loading
    const int BS = 1 << bshift, nb = (ns + BS - 1) >> bshift, nb2 = (nb + kFpFan - 1) / kFpFan;
    const int nwords = (ns + 31) >> 5;
    const FpLds T = fp_lds(smem_raw, nb, nb2, nwords);
    const float* rg = x + (size_t)blockIdx.x * ns;
    const int tid = threadIdx.x;
    FP_T0;
    if (tid == 0) wave_tot[kFpThreads / 64] = wave_tot[kFpThreads / 64 + 1] = 0;     // fp_scan's candidate / hot-block counters
    for (int w = tid; w < nwords; w += kFpThreads) T.bits[w] = 0u;
    const bool al16 = (ns & 3) == 0 && ((reinterpret_cast<size_t>(rg) & 15) == 0);
    const bool vec4 = bshift == 5 && al16;
    const bool sweep = !STAGED && vec4;
    if (vec4) {
        fp_stage_rows4<STAGED>(rg, T, ns, nb, tid);
    } else {
        if (STAGED) {
            for (int i = tid; i < ns; i += kFpThreads) T.rowl[i] = rg[i];
            __syncthreads();
        }
        fp_summaries1(STAGED ? T.rowl : rg, T, ns, nb, bshift, tid);
    }
    __syncthreads();
    FP_MARK(0);                                               // staged + block summaries
    fp_summaries2(T, nb, nb2, tid);
    if (sweep)
        for (int w = tid; w < nwords; w += kFpThreads) T.cand[w] = 0u;
    __syncthreads();
    FP_MARK(1);
    {
        // A peak's prominence is its height less the higher of two minima of the row: never more than (row maximum - row minimum).
        // A row that cannot reach the threshold -- most channels of a file carry no call above 0.45 of the file's largest
        // correlation -- is done here: no candidates, no walks, no bitmap scan.  (Exact: the difference is formed in float64 like the walks', and the
        // test is false for a NaN, which then takes the full path.)
        float gmax = -INFINITY, gmin2 = INFINITY;
        for (int k = 0; k < nb2; ++k) { const float2 e = T.s2[k]; gmax = fmaxf(gmax, e.x); gmin2 = fminf(gmin2, e.y); }
        if ((double)gmax - (double)gmin2 < thr) {
            if (tid == 0) counts[blockIdx.x] = 0;
            return;
        }
    }
    int mark_from = 0;
    if (sweep) {
        float gmin = INFINITY;
        for (int k = 0; k < nb2; ++k) gmin = fminf(gmin, T.s2[k].y);
        mark_from = fp_sweep_segments(rg, T, ns, nb, thr, gmin, wave_tot, hot, tid);
    }
    fp_scan(STAGED ? T.rowl : rg, T, ns, nb, nb2, bshift, thr, nwords, wave_tot, cfail, STAGED ? (ns & 3) == 0 : al16,
            mark_from, tid);
    __syncthreads();
    FP_MARK(2);                                               // marking + listing + walks
    fp_emit(T, nwords, idx + (size_t)blockIdx.x * cap, counts + blockIdx.x, cap, wave_tot, tid);
    FP_MARK(3);
}

// picks of all rows as ONE packed 2 x K table (detect.convert_pick_times, detect.py:277-303: row 0 = channel,
// row 1 = time index): row c's cnt[c] indices go to columns [off[c] - cnt[c], off[c]) (off = inclusive prefix sum)
__global__ __launch_bounds__(kSpThreads) void pack_picks(const int* __restrict__ idx, const int* __restrict__ cnt,
                                                          const long long* __restrict__ off, int nx, int cap, long long total,
                                                          long long* __restrict__ out) {
    for (int c = blockIdx.x; c < nx; c += gridDim.x) {
        const int n = min(cnt[c], cap);
        const long long base = off[c] - cnt[c];
        for (int j = threadIdx.x; j < n; j += kSpThreads) {
            out[base + j] = c;
            out[total + base + j] = idx[(size_t)c * cap + j];
        }
    }
}

// Inclusive prefix sums of the per-row pick counts (what pack_picks places the rows by), their total and the largest count,
// in ONE small launch: one workgroup, a run of consecutive rows per thread.  (torch.cumsum + max + stack: a rocprim scan, two
// reductions and a concatenation per picker call before round 5.)
constexpr int kOffThreads = 1024;
__global__ __launch_bounds__(kOffThreads) void pick_offsets(const int* __restrict__ cnt, int nx, long long* __restrict__ off,
                                                            long long* __restrict__ summary) {
    __shared__ long long wsum[kOffThreads / 64];
    __shared__ int pmax[kOffThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nx + kOffThreads - 1) / kOffThreads;
    const int a = min(tid * per, nx), b = min(a + per, nx);
    long long s = 0;
    int m = 0;
    for (int i = a; i < b; ++i) {
        const int c = cnt[i];
        s += c;
        m = max(m, c);
    }
    // inclusive scan of the threads' sums: inside a wave by shuffles, across the 16 waves through LDS (two barriers in all;
    // the Hillis-Steele form over 1024 threads took twenty)
    long long incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) pmax[wave] = m;
    __syncthreads();
    long long before = 0, total = 0;
    for (int w = 0; w < kOffThreads / 64; ++w) {
        if (w < wave) before += wsum[w];
        total += wsum[w];
    }
    long long run = before + incl - s;                     // exclusive prefix of this thread's rows
    for (int i = a; i < b; ++i) {
        run += cnt[i];
        off[i] = run;
    }
    if (tid == 0) {
        int mm = 0;
        for (int w = 0; w < kOffThreads / 64; ++w) mm = max(mm, pmax[w]);
        summary[0] = mm;
        summary[1] = total;
    }
}

}  // namespace d4w

using namespace d4w;

template <typename K>
static void sp_allow_lds(K kern, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

extern "C" {

int d4w_row_var_f32(const float* x, int nx, int ns, float* var, void* stream) {
    if (!x || !var || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_var, dim3(nx), dim3(kSpThreads), 0, stream, x, ns, var);
    return D4W_OK;
}

static size_t row_tile_lds(const RowFftDev& F) {
    const int n = F.bs_L ? F.bs_L : F.ax.L;
    return ((size_t)n + kTwLo + (F.bs_L ? F.ax_bs.nhi : F.ax.nhi)) * sizeof(float2);
}

int d4w_analytic_row_fits_lds(int ns) {
    if (ns < 2) return 0;
    int L = (ns % 2 == 0) ? ns / 2 : ns;
    std::vector<int> rad;
    if (!factor_radices(L, rad)) L = smooth_len_235(2L * L - 1);      // Bluestein tile
    return ((size_t)L + kTwLo + (size_t)(L + kTwLo - 1) / kTwLo) * sizeof(float2) <= kSpLdsMax ? 1 : 0;
}

int d4w_analytic_f32(const float* x, float* y, int nx, int ns, int mode, const float* var, double fs,
                     void* stream) {
    if (!x || !y || nx < 1 || ns < 2) return fail(D4W_EINVAL, "bad argument");
    if (mode < 0 || mode > 4) return fail(D4W_EINVAL, "mode = %d not in 0..4", mode);
    if ((mode == kAnSnr || mode == kAnEnvStd) && !var) return fail(D4W_EINVAL, "modes 2 and 4 need the row variances");
    const bool packed = (ns % 2 == 0);
    const int L = packed ? ns / 2 : ns;
    const RowFftHost* h = nullptr;
    int rc = row_fft_get(L, &h);
    if (rc) return rc;
    const size_t lds = row_tile_lds(h->dev);
    if (lds > kSpLdsMax)
        return fail(D4W_EINVAL, "rows of %d samples exceed the single-workgroup transform (max %d even / %d odd)",
                    ns, (int)(2 * (kSpLdsMax / 8 - 512)), (int)(kSpLdsMax / 8 - 512));
    const float fscale = (float)(fs / (2.0 * M_PI));
    static const int an_env = [] { const char* v = getenv("D4W_AN_THREADS"); return v ? atoi(v) : 0; }();
    // long rows: 512 threads per row (three 48-KB tiles fit a CU either way: 24 waves instead of 12; 0.69 -> 0.56 ms at
    // 11020 x 12000)
    const int an_threads = (an_env >= 64 && an_env <= kAnMaxThreads && an_env % 64 == 0) ? an_env
                           : (L >= 2048 ? kAnMaxThreads : kSpThreads);
#define D4W_AN(P, G)                                                                              \
    do {                                                                                          \
        sp_allow_lds(analytic_rows<P, G>, lds);                                                   \
        D4W_LAUNCH((analytic_rows<P, G>), dim3(nx), dim3(an_threads), lds, stream, h->dev, x, ns, \
                   y, mode, var, fscale);                                                         \
    } while (0)
    if (packed) { if (h->generic) D4W_AN(true, true); else D4W_AN(true, false); }
    else { if (h->generic) D4W_AN(false, true); else D4W_AN(false, false); }
#undef D4W_AN
    return D4W_OK;
}

int d4w_snr_f32(const float* x, float* y, int nx, int ns, int env, float* var_ws, void* stream) {
    if (!x || !y || !var_ws || nx < 1 || ns < 2) return fail(D4W_EINVAL, "bad argument");
    int rc = d4w_row_var_f32(x, nx, ns, var_ws, stream);
    if (rc) return rc;
    if (env) return d4w_analytic_f32(x, y, nx, ns, kAnSnr, var_ws, 0.0, stream);
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    D4W_LAUNCH(snr_rows, dim3(std::min(ceil_div(ns, kSpThreads), 64), nx), dim3(kSpThreads), 0, stream, x, ns,
               (const float*)var_ws, y);
    return D4W_OK;
}

int d4w_fx_f32(const float* x, float* y, int nx, int ns, int nfft, void* stream) {
    if (!x || !y || nx < 1 || ns < 1 || nfft < 1) return fail(D4W_EINVAL, "bad argument");
    const RowFftHost* h = nullptr;
    int rc = row_fft_get(nfft, &h);
    if (rc) return rc;
    const size_t lds = row_tile_lds(h->dev);
    if (lds > kSpLdsMax) return fail(D4W_EINVAL, "nfft = %d exceeds the single-workgroup transform", nfft);
    if (h->generic) {
        sp_allow_lds(fx_rows<true>, lds);
        D4W_LAUNCH(fx_rows<true>, dim3(nx), dim3(kSpThreads), lds, stream, h->dev, x, ns, y);
    } else {
        sp_allow_lds(fx_rows<false>, lds);
        D4W_LAUNCH(fx_rows<false>, dim3(nx), dim3(kSpThreads), lds, stream, h->dev, x, ns, y);
    }
    return D4W_OK;
}

int d4w_stft_frames(int ns, int hop) { return (hop > 0 && ns >= 0) ? 1 + ns / hop : 0; }

int d4w_stft_mag_f32(const float* x, float* S, float* rowmax, int nx, int ns, int n_fft, int hop, int bin_lo,
                     int bin_hi, void* stream) {
    if (!x || !S || nx < 1 || ns < 1) return fail(D4W_EINVAL, "bad argument");
    // a few kept bins of a heavily overlapped transform and no row maximum wanted (the detector's call): frames x DFT rows
    // as a matrix product on the matrix cores (stft_mm.hip) instead of one FFT per frame
    if (!rowmax && d4w_stft_mm_eligible(n_fft, hop, bin_lo, bin_hi))
        return d4w_stft_mag_mm_f32(x, S, nx, ns, n_fft, hop, bin_lo, bin_hi, stream);
    if (n_fft < 2 || (n_fft & 1) || hop < 1) return fail(D4W_EINVAL, "n_fft = %d must be even and >= 2, hop = %d >= 1", n_fft, hop);
    if (bin_lo < 0 || bin_hi > n_fft / 2 || bin_lo > bin_hi) return fail(D4W_EINVAL, "bin range [%d, %d] outside 0..%d", bin_lo, bin_hi, n_fft / 2);
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    const RowFftHost* h = nullptr;
    int rc = row_fft_get(n_fft, &h);
    if (rc) return rc;
    StftDims d;
    d.ns = ns; d.n_fft = n_fft; d.hop = hop; d.nframes = 1 + ns / hop; d.b_lo = bin_lo; d.b_hi = bin_hi;
    const int tl = h->dev.bs_L ? h->dev.bs_L : n_fft;        // LDS elements per transform (Bluestein: the convolution length)
    int nb = std::max(1, 6144 / tl);                         // complex transforms (frame pairs) per workgroup
    // two-factor frame lengths: 16 pairs make the first stage exactly one item per thread (RB = 16) and keep the tile at
    // 22-35 KB, i.e. 16-24 waves on a CU instead of 8 (measured at 11020 x 12000: n_fft 160 kept bins 2.61 -> 1.80 ms, all bins
    // 3.98 -> 2.77 ms, n_fft 256 5.05 -> 3.70 ms)
    if (n_fft == 128 || n_fft == 160 || n_fft == 256) nb = 16;
    else if (n_fft == 512) nb = 8;
    {
        static const int nb_env = [] { const char* v = getenv("D4W_STFT_NB"); return v ? atoi(v) : 0; }();
        if (nb_env > 0) nb = nb_env;
    }
    nb = std::min(nb, std::max(1, (d.nframes + 1) / 2));
    d.FT = 2 * nb;
    // two-factor register transforms for the common frame lengths (rowmax may be NULL there: only the kept bins are formed)
    {
        static const int fat_env = [] { const char* v = getenv("D4W_STFT_FAT"); return v ? atoi(v) : 1; }();
        int RA = 0, RB = 0;
        if (n_fft == 160) { RA = 10; RB = 16; }
        else if (n_fft == 128) { RA = 8; RB = 16; }
        else if (n_fft == 256) { RA = 16; RB = 16; }
        else if (n_fft == 512) { RA = 32; RB = 16; }
        if (RA && fat_env) {
            const size_t ldsf = ((size_t)nb * RA * (RB + 1) + n_fft) * sizeof(float2) +
                                ((size_t)n_fft + (size_t)(d.FT - 1) * hop + n_fft) * sizeof(float);
            if (ldsf <= kSpLdsMax) {
                if (rowmax) D4W_HIP(hipMemsetAsync(rowmax, 0, (size_t)nx * sizeof(float), (hipStream_t)stream));
                // tiles per row walked by min(tiles, ~8192 / nx) workgroups (one per row for a whole file, all tiles in parallel
                // for a single channel)
                const int ntile = ceil_div(d.nframes, d.FT);
                const dim3 gridf(std::max(1, std::min(ntile, 8192 / std::max(nx, 1))), nx);
#define D4W_FAT(A_, B_)                                                                                             \
    do {                                                                                                            \
        sp_allow_lds(stft_fat<A_, B_>, ldsf);                                                                       \
        D4W_LAUNCH((stft_fat<A_, B_>), gridf, dim3(kSpThreads), ldsf, stream, h->dev, d, x, S, (unsigned*)rowmax);  \
    } while (0)
                if (n_fft == 160) D4W_FAT(10, 16);
                else if (n_fft == 128) D4W_FAT(8, 16);
                else if (n_fft == 256) D4W_FAT(16, 16);
                else D4W_FAT(32, 16);
#undef D4W_FAT
                return D4W_OK;
            }
        }
    }
    if (!rowmax) return fail(D4W_EINVAL, "rowmax is NULL (only the two-factor frame lengths 128, 160, 256, 512 form the kept bins alone)");
    const size_t lds = ((size_t)nb * tl + kTwLo + (h->dev.bs_L ? h->dev.ax_bs.nhi : h->dev.ax.nhi)) * sizeof(float2) +
                       ((size_t)(d.FT - 1) * hop + n_fft) * sizeof(float);
    if (lds > kSpLdsMax) return fail(D4W_EINVAL, "n_fft = %d / hop = %d exceed the LDS frame tile", n_fft, hop);
    D4W_HIP(hipMemsetAsync(rowmax, 0, (size_t)nx * sizeof(float), (hipStream_t)stream));
    const dim3 grid(ceil_div(d.nframes, d.FT), nx);
    if (h->generic) {
        sp_allow_lds(stft_mag<true>, lds);
        D4W_LAUNCH(stft_mag<true>, grid, dim3(kSpThreads), lds, stream, h->dev, d, x, S, (unsigned*)rowmax);
    } else {
        sp_allow_lds(stft_mag<false>, lds);
        D4W_LAUNCH(stft_mag<false>, grid, dim3(kSpThreads), lds, stream, h->dev, d, x, S, (unsigned*)rowmax);
    }
    return D4W_OK;
}

int d4w_scale_rows_f32(float* S, int nx, size_t per_row, const float* denom, int mode, void* stream) {
    if (!S || !denom || nx < 1 || per_row < 1) return fail(D4W_EINVAL, "bad argument");
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    const int bx = (int)std::min<size_t>((per_row + kSpThreads - 1) / kSpThreads, 256);
    D4W_LAUNCH(scale_rows, dim3(bx, nx), dim3(kSpThreads), 0, stream, S, per_row, denom, mode);
    return D4W_OK;
}

int d4w_row_median_f32(const float* v, int nx, size_t per_row, float* med, void* stream) {
    if (!v || !med || nx < 1 || per_row < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(row_median, dim3(nx), dim3(kSpThreads), 0, stream, v, per_row, med);
    return D4W_OK;
}

int d4w_spectrocorr_f32(const float* S, int nx, int nf, int nt, const float* K, int nk, int off, int nout,
                        const float* med, int zero_ends, float* out, void* stream) {
    if (!S || !K || !med || !out || nx < 1 || nf < 1 || nt < 1 || nk < 1 || nout < 1)
        return fail(D4W_EINVAL, "bad argument");
    if (sc_width(nk) > kScLdsFloats) return fail(D4W_EINVAL, "kernel of %d frames is too long", nk);
    if (nx > 65535) return fail(D4W_EINVAL, "nx = %d exceeds the grid limit 65535", nx);
    const dim3 grid(ceil_div(nout, kScTile), nx);
    const int width = sc_width(nk);
    if (width <= 5 * kScThreads)
        D4W_LAUNCH((spectro_corr<4, 5>), grid, dim3(kScThreads), 0, stream, S, nf, nt, K, nk, off, nout, med, zero_ends, out);
    else if (width <= 10 * kScThreads)
        D4W_LAUNCH((spectro_corr<2, 10>), grid, dim3(kScThreads), 0, stream, S, nf, nt, K, nk, off, nout, med, zero_ends, out);
    else
        D4W_LAUNCH((spectro_corr<1, 20>), grid, dim3(kScThreads), 0, stream, S, nf, nt, K, nk, off, nout, med, zero_ends, out);
    return D4W_OK;
}

static int find_peaks_launch(const float* x, int nx, int ns, double prominence, const float* thr_dev, int32_t* idx,
                             int32_t* counts, int cap, void* stream) {
    if (!x || !idx || !counts || nx < 1 || ns < 1 || cap < 1) return fail(D4W_EINVAL, "bad argument");
    int bshift = 5;                                            // 32-sample blocks, larger for very long rows
    while (((ns + (1 << bshift) - 1) >> bshift) > kFpMaxBlocks) ++bshift;
    const int nb = (ns + (1 << bshift) - 1) >> bshift;
    static const int staged_env = [] { const char* v = getenv("D4W_FP_STAGED"); return v ? atoi(v) : -1; }();   // A/B switch
    const bool staged = staged_env >= 0 ? (staged_env > 0 && ns <= kFpRowLds) : (ns <= kFpRowLds);
    const int nb2 = (nb + kFpFan - 1) / kFpFan;
    const size_t lds = ((size_t)2 * ((nb + nb2 + 1) & ~1) + 2 * (size_t)((((ns + 31) >> 5) + 3) & ~3) + (size_t)kFpList +
                        (staged ? (size_t)ns : 0)) * sizeof(float);
    if (lds > kSpLdsMax) return fail(D4W_EINVAL, "rows of %d samples exceed the peak-picking LDS tables", ns);
    if (staged) {
        sp_allow_lds(find_peaks_prom<true>, lds);
        D4W_LAUNCH(find_peaks_prom<true>, dim3(nx), dim3(kFpThreads), lds, stream, x, ns, prominence, bshift, (int*)idx,
                   (int*)counts, cap, thr_dev);
    } else {
        sp_allow_lds(find_peaks_prom<false>, lds);
        D4W_LAUNCH(find_peaks_prom<false>, dim3(nx), dim3(kFpThreads), lds, stream, x, ns, prominence, bshift, (int*)idx,
                   (int*)counts, cap, thr_dev);
    }
    return D4W_OK;
}

int d4w_find_peaks_f32(const float* x, int nx, int ns, double prominence, int32_t* idx, int32_t* counts, int cap,
                       void* stream) {
    return find_peaks_launch(x, nx, ns, prominence, nullptr, idx, counts, cap, stream);
}

int d4w_find_peaks_dthr_f32(const float* x, int nx, int ns, const float* value, double scale, int32_t* idx, int32_t* counts,
                            int cap, void* stream) {
    if (!value) return fail(D4W_EINVAL, "NULL argument");
    return find_peaks_launch(x, nx, ns, scale, value, idx, counts, cap, stream);
}

#ifdef D4W_FP_TIMING
int d4w_fp_timing_read(unsigned long long* host16, int reset) {
    D4W_HIP(hipDeviceSynchronize());
    D4W_HIP(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_fp_t), sizeof(g_fp_t)));
    if (reset) {
        unsigned long long z[16] = {};
        D4W_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fp_t), z, sizeof(z)));
    }
    return D4W_OK;
}
#endif

int d4w_pick_offsets_i64(const int32_t* counts, int nx, int64_t* offsets, int64_t* summary2, void* stream) {
    if (!counts || !offsets || !summary2 || nx < 1) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(pick_offsets, dim3(1), dim3(kOffThreads), 0, stream, (const int*)counts, nx, (long long*)offsets, (long long*)summary2);
    return D4W_OK;
}

int d4w_pack_picks_i64(const int32_t* idx, const int32_t* counts, const int64_t* offsets, int nx, int cap, int64_t total,
                       int64_t* out, void* stream) {
    if (!idx || !counts || !offsets || nx < 1 || cap < 1 || total < 0) return fail(D4W_EINVAL, "bad argument");
    if (total == 0) return D4W_OK;
    if (!out) return fail(D4W_EINVAL, "bad argument");
    D4W_LAUNCH(pack_picks, dim3(std::min(nx, 4096)), dim3(kSpThreads), 0, stream, (const int*)idx, (const int*)counts,
               (const long long*)offsets, nx, cap, (long long)total, (long long*)out);
    return D4W_OK;
}

}  // extern "C"
