// Experiment, not built (round 5): detect.pick_times_env in ONE launch -- the envelope |hilbert(row)| formed in the LDS row the
// peak picker works on (analytic_rows' packed transform in place, then find_peaks_prom<true> past its staging), so that the
// envelope is never written to memory.  Correct (identical picks on the CPU test build and on the GPU: 37 GPU tests green with it
// as the default) and SLOWER: 1446 us per 11 020 x 12 000 correlogram against 492 us (analytic_rows) + 524 us (find_peaks_prom) as
// two launches (profiles/r05f/stream_kernels.txt).  Both halves are chains of LDS round trips and barriers on one workgroup per
// row; fused, a workgroup holds the picker's tables AND the transform tile (75 KB: two per compute unit instead of three) and
// runs the two chains one after the other, where the two launches overlap rows of different phases across workgroups.  The
// 1 GB of envelope traffic it saves per correlogram is ~0.2 ms of HBM time -- less than the occupancy it costs.
// To try it again: paste into csrc/spectral.hip before pack_picks, add the entry points (d4w_find_peaks_env_fits / _f32).
// detect.pick_times_env in ONE launch: find_peaks(|hilbert(row)|, prominence) with the envelope formed in the LDS row the
// picker works on (analytic_rows' packed transform, in place: the pair (H[2m], H[2m+1]) becomes (|z[2m]|, |z[2m+1]|) at the
// same 8 bytes) -- the envelope is never written to memory and never read back: 4 B read per sample instead of 4 + 4 + 4.
// Same arithmetic as analytic_rows<true, .> followed by find_peaks_prom<true>: identical picks.
template <bool GENERIC>
__global__ __launch_bounds__(kFpThreads) void find_peaks_env(RowFftDev F, const float* __restrict__ x, int ns, double thr,
                                                             int bshift, int* __restrict__ idx, int* __restrict__ counts,
                                                             int cap) {
    D4W_DYN_LDS(smem_raw);
    __shared__ int wave_tot[kFpThreads / 64 + 1];
    __shared__ unsigned cfail[kFpList / 32];
    const int BS = 1 << bshift, nb = (ns + BS - 1) >> bshift, nb2 = (nb + kFpFan - 1) / kFpFan;
    const int nwords = (ns + 31) >> 5;
    const FpLds T = fp_lds(smem_raw, nb, nb2, nwords);
    const int tid = threadIdx.x, nthr = kFpThreads;
    const int L = F.ax.L;                                              // ns / 2 packed samples
    float2* tile = reinterpret_cast<float2*>(T.rowl);                  // the picker's row IS the transform tile
    const TwLds tw = tw_stage(row_tw_axis(F), tile + row_tile_elems(F), tid, nthr);
    const float* xr = x + (size_t)blockIdx.x * ns;
    const float2* x2 = reinterpret_cast<const float2*>(xr);
    for (int w = tid; w < nwords; w += kFpThreads) T.bits[w] = 0u;
    constexpr int kAhead = 8;
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
    lds_barrier();
    row_dft<false, GENERIC>(tile, F, tw, tid, nthr);
    {
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
                const float2 Yp = c_mul_mi(c_add(E, tO));
                const float2 Ym = c_mul_pi(c_sub(E, tO));
                const float2 S = c_scale(c_add(Yp, Ym), 0.5f);
                const float2 D = c_mul_pi(c_mulc(c_scale(c_sub(Yp, Ym), 0.5f), w));
                out_a = c_add(S, D);
                out_b = c_conj(c_sub(S, D));
            }
            tile[pa] = out_a;
            if (pb != pa) tile[pb] = out_b;
        }
    }
    lds_barrier();
    row_dft<true, GENERIC>(tile, F, tw, tid, nthr);
    const float scale = 1.0f / (float)L;
    for (int m0 = tid; m0 < L; m0 += kAhead * nthr) {                  // the envelope, in place
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
                const float hx = h.x * scale, hy = h.y * scale;
                tile[m] = make_float2(sqrtf(fmaf(q[k].x, q[k].x, hx * hx)), sqrtf(fmaf(q[k].y, q[k].y, hy * hy)));
            }
        }
    }
    __syncthreads();
    // ---- the picker on the LDS row (find_peaks_prom<true> past its staging)
    fp_summaries1(T.rowl, T, ns, nb, bshift, tid);
    __syncthreads();
    fp_summaries2(T, nb, nb2, tid);
    __syncthreads();
    fp_scan(T.rowl, T, ns, nb, nb2, bshift, thr, nwords, wave_tot, cfail, (ns & 3) == 0, 0, tid);
    __syncthreads();
    fp_emit(T, nwords, idx + (size_t)blockIdx.x * cap, counts + blockIdx.x, cap, wave_tot, tid);
}

