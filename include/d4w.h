/* d4w.h -- C ABI of the MI355X-native DAS4Whales hot path (libd4w.so, built for gfx950).
 *
 * This is the drop-in boundary: plain C, device pointers and sizes only, no torch / numpy types.
 * The reference (leabouffaut/DAS4Whales @ 2024_08_07) has no FFI of its own -- its hot path is a
 * set of Python module functions over NumPy arrays -- so each entry point below cites the
 * reference function (file:line under src/das4whales/) whose array arithmetic it replaces.  The
 * Python mirror of the reference namespace (das4whales_amd/dsp.py, detect.py) and any other host
 * language bind these symbols (see INTEGRATION.md for the ctypes stub a reference maintainer
 * would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless its name ends in _host;
 *   - matrices are row-major [nx channels][ns time samples], float32, time fastest;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     with respect to the host and ordered on that stream;
 *   - return value: 0 = D4W_OK, negative = error; d4w_last_error() gives the message of the
 *     last failure on the calling thread.
 */
#ifndef D4W_H
#define D4W_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D4W_OK 0
#define D4W_EINVAL (-1)  /* bad argument / unsupported shape */
#define D4W_ENOMEM (-2)  /* device or host allocation failed */
#define D4W_EHIP (-3)    /* HIP runtime error (message has the call) */

const char* d4w_last_error(void);
/* "d4w <version> gfx950" -- lets the host side assert that the native library is the one loaded */
const char* d4w_version(void);

/* ------------------------------------------------------------------------------------------
 * f-k filter application:  y = real(ifft2(ifftshift(fftshift(fft2(x)) * M)))
 * replaces dsp.fk_filter_filt (dsp.py:725-756) and dsp.fk_filter_sparsefilt (dsp.py:759-786);
 * also the apply half of dsp.fk_filt (dsp.py:919,948-953).
 *
 * A plan owns the factorisation of the packed 2-D transform (nx x ns/2 complex), its twiddle /
 * index tables and the folded, permuted mask.  Requirements: ns even.  nx, ns / 2: any (prime factors
 * > 31 run Bluestein convolutions: inside the LDS tiles of passes C / B while that part of nx is
 * <= 4096 and that part of ns / 2 <= 2048, in global memory -- scratch <= 1 GiB per axis owned by the
 * plan, several times slower -- beyond).  Plans are immutable after set_mask and may be shared by streams that serialise
 * their own calls.
 * ------------------------------------------------------------------------------------------ */
typedef struct d4w_fk_plan d4w_fk_plan;

int d4w_fk_plan_create(int nx, int ns, d4w_fk_plan** plan);
/* opts: NULL or 6 ints {C1, C2, N1, N2, TA, TC}; any entry <= 0 keeps the planner's choice */
int d4w_fk_plan_create_ex(int nx, int ns, const int* opts_host, d4w_fk_plan** plan);
int d4w_fk_plan_destroy(d4w_fk_plan* plan);
/* 1 when [nx][ns] runs shape-specialised kernels (every sub-transform a compile-time register butterfly), 0 when it
 * runs the generic runtime-radix passes.  Specialised configurations for further shapes are compiled on demand by the
 * host side (das4whales_amd/fkjit.py: same templates, one translation unit per shape) and handed over with
 * d4w_fk_register_shape (entry = the d4w::FkFastEntry of csrc/fk_entry.h, entry_size = its sizeof as a build guard). */
int d4w_fk_shape_is_specialised(int nx, int ns);
int d4w_fk_register_shape(const void* entry, size_t entry_size);
/* fills {nx, ns, C1, C2, N1, N2, TA, TC} */
int d4w_fk_plan_info(const d4w_fk_plan* plan, int* info8_host);

/* Mask as the reference hands it over: dense [nx][ns] float32 on the fftshift-ed (k, f) grid
 * (dsp.py:129-130,137).  Folds it to its Hermitian part M_h = (M(k,f)+M(-k,-f))/2 -- which is
 * what `.real` at dsp.py:756,786 keeps -- and stores it in the plan's internal order.
 * Synchronises `stream` (the row-liveness scan of the folded mask is read back by the host). */
int d4w_fk_set_mask_dense_f32(d4w_fk_plan* plan, const float* mask_shifted, void* stream);
/* Same, with the OPT-IN tail pruning: wavenumber rows whose folded gains all stay at or below
 * prune_eps * max|M_h| (together with their Hermitian partner row's) are treated as zero, i.e. dead
 * (see d4w_fk_plan_live_rows).  hybrid_ninf_filter_design (dsp.py:348-406) multiplies every
 * wavenumber row by Butterworth tails that never reach zero (7.7e-7 at fmax + 14 Hz), so its rows are
 * all formally alive; prune_eps = 4e-6 keeps only the rows the speed band touches.  Not exact: the
 * output changes by at most the pruned gain times the input's energy in those bins.  0 = exact. */
int d4w_fk_set_mask_dense_pruned_f32(d4w_fk_plan* plan, const float* mask_shifted, double prune_eps, void* stream);
/* A reference design written straight into the plan, with no dense mask in between: what
 * d4w_design_mask_f32(mode, ...) followed by d4w_fk_set_mask_dense_pruned_f32 leaves in the plan, bit for bit
 * (same closed forms, same float32 rounding before the fold), for the three designs that have a closed
 * form -- mode 0 dsp.fk_filter_design (dsp.py:85-171), 1 dsp.hybrid_filter_design (dsp.py:174-305),
 * 2 dsp.hybrid_ninf_filter_design (dsp.py:308-454); arguments as d4w_design_mask_f32 below.  The Gaussian-blurred
 * designs (modes 3-5) need the dense grid and are refused.  Saves the nx*ns*4-byte mask (9.6 GB at
 * 20 000 x 120 000) and its write + read.  Synchronises `stream`. */
/* dsp.fk_filt's min-max normalisation (dsp.py:945) folded into the mask upload: every value enters as
 * mask * scale + offset (scale = 1 / (max - min), offset = -min * scale, min and max from d4w_minmax_f32), exactly what
 * d4w_minmax_normalise_f32 followed by d4w_fk_set_mask_dense_f32 leaves, without the extra read and write of the mask. */
int d4w_fk_set_mask_dense_affine_f32(d4w_fk_plan* plan, const float* mask_shifted, float scale, float offset,
                                     void* stream);
int d4w_fk_set_mask_design_f32(d4w_fk_plan* plan, int mode, double k_spacing, double t_spacing,
                               const double* params8_host, int i0, int i1, const double* hrow_dev,
                               double prune_eps, void* stream);

/* Number of wavenumber rows (0..nx) the current mask keeps alive.  A row whose folded mask -- and
 * whose Hermitian partner's -- is identically zero is multiplied by zero whatever it holds; the
 * shape-specialised kernels skip such rows in the three middle passes (exact, the analogue of the
 * reference's sparse.COO product at dsp.py:782).  nx = nothing is skipped (dense mask, or a shape
 * that runs the generic kernels).  D4W_FK_NOPRUNE=1 in the environment disables the skipping. */
int d4w_fk_plan_live_rows(const d4w_fk_plan* plan);

/* Order of the five passes for the current mask (shape-specialised kernels).  The c2 and n2 sub-transforms commute, so
 * besides the channel-first order A, C, B, C', A' (which skips dead WAVENUMBER rows) the filter can run time-first:
 * A, Bf, Cm, Bi, A' with the half spectrum compacted between Bf and Bi -- frequency columns whose folded gain is zero are
 * never written, columns whose gain is the same for every wavenumber (the Butterworth skirts hybrid_ninf_filter_design
 * keeps outside its looped columns, dsp.py:348-360) are scaled in Bf and skip the channel transform, the others go
 * through Cm (c2 forward x mask x c2 inverse in one visit).  Exact like the channel-first order: both treat as zero only
 * gains below max(prune_eps, 2^-24 / sqrt(nx ns)) * max|M_h| -- at prune_eps = 0 at most half a float32 ulp of the
 * input's RMS in any output sample (D4W_FK_ROUND_EPS=0: exact zeros only).  The plan picks the order that moves fewer
 * bytes (D4W_FK_ORDER=cf|tf forces one).  info6 = {1 time-first / 0 channel-first, band columns kept per row, tail
 * columns kept per row, ns / 2, live wavenumber rows, nx}; bytes2 = modelled bytes per channel-sample {channel-first,
 * time-first}. */
int d4w_fk_plan_order(const d4w_fk_plan* plan, int* info6_host, double* bytes2_host);

/* x -> y (may alias).  taper != 0 multiplies x by tukey(ns, 0.03) first (dsp.taper_data,
 * dsp.py:705-722, fused into the first pass; x itself is not modified). */
int d4w_fk_apply_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, void* stream);

/* d4w_fk_apply_f32 that also returns what the matched filter normalises the filtered rows by
 * (detect.py:157): row_mean[c] = mean(y[c,:]) DEVICE float64 [nx] (the reference de-means in float64; the correlators
 * take it as a two-float value), row_maxabs[c] = max|y[c,:]| DEVICE float32 [nx].
 * The shape-specialised kernels form them in the epilogue of the last pass (no extra read of y);
 * other shapes run d4w_row_stats_f32 afterwards.  Feed them to d4w_xcorr_fft_f32 / d4w_xcorr_lens_f32. */
int d4w_fk_apply_stats_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, double* row_mean,
                           float* row_maxabs, void* stream);
/* 1 when d4w_fk_apply_stats_f32 on this plan forms the statistics in the last pass's epilogue, 0 when it sweeps y afterwards
 * (small pass tiles on large blocks, the 60-s file shapes; unspecialised shapes): a caller that also wants the rows' prefix
 * maxima then runs d4w_fk_apply_f32 + d4w_row_stats_prefix_f32 for the same traffic. */
int d4w_fk_stats_in_epilogue(const d4w_fk_plan* plan);

/* Same as d4w_fk_apply_f32 but brackets each of the five passes (A, C, B, C', A') with HIP events
 * on `stream`, synchronises, and returns their durations in milliseconds in ms5_host[0..4].
 * Measurement aid for bench.py's roofline report; not for production loops. */
int d4w_fk_apply_timed_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, void* stream,
                           float* ms5_host);
/* timed variant of d4w_fk_apply_stats_f32 (row_mean / row_maxabs may both be NULL) */
int d4w_fk_apply_timed_stats_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, double* row_mean,
                                 float* row_maxabs, void* stream, float* ms5_host);

/* ------------------------------------------------------------------------------------------
 * Distributed f-k filter: ONE [nx][ns] block sharded by contiguous channel block over `world`
 * GPUs (rank r owns rows [row_begin, row_end), balanced like np.array_split).  Exactly the same
 * filter as d4w_fk_apply_f32 -- the 2-D transform couples all channels, so the packed spectrum is
 * re-sharded once each way (pencil decomposition):
 *
 *   d4w_fkd_time_fwd_f32   local rows: x_loc [nxl][ns] real -> z_loc [nxl][N1][N2] complex
 *                          (time-axis transform of the packed rows; q1-major sub-rows of N2 bins)
 *   -- all-to-all: sub-row q1 of every channel goes to rank owner[q1] (d4w_fkd_plan_q1_owner);
 *      the receiver concatenates the senders' pieces in rank order = channel order, which gives
 *      slab [nx][nq][N2] (nq = sub-rows owned, in ascending q1)
 *   d4w_fkd_chan_apply_f32 channel-axis transform, real-spectrum pair op x folded mask, inverse
 *                          channel-axis transform (x 1/(nx ns/2)), in place on the slab
 *   -- all-to-all back (the exact reverse)
 *   d4w_fkd_time_inv_f32   z_loc -> filtered rows, in place (read the buffer as float [nxl][ns])
 *
 * The exchange itself (RCCL all_to_all_single over xGMI, gloo in the CPU tests) is host plumbing:
 * das4whales_amd/shard.py fk_filter_sharded.  All buffers are DEVICE float32 (complex = 2 floats).
 * info12 = {nx, ns, world, rank, row_begin, row_end, N1, N2, nq, C1, C2, packed}.
 *
 * PACKED plans (d4w_fkd_plan_is_packed; every shape with shape-specialised kernels, e.g. 20000 x 120000 and the
 * 60-s file shapes) need no repacking around the exchanges: the time phase writes straight into the send buffer,
 * destination rank major --
 *   packed = [dest rank s][local row l][jq < nq(s)][N2] complex, block s holding nxl * nq(s) * N2 elements,
 * so that all_to_all_single delivers the slab [nx][nq][N2] as it stands (senders in rank order = channel order), and
 * the second exchange delivers the same layout back:
 *   d4w_fkd_time_fwd_packed_f32   x_loc [nxl][ns] -> packed            (n1 sub-transform + four-step twiddle)
 *   d4w_fkd_chan_apply_f32        slab, in place: c1 transform, c2 transform, n2 transform + pair op x mask +
 *                                 inverse n2, inverse c2, inverse c1    (five launches of the fk_fast.h kernels)
 *   d4w_fkd_time_inv_packed_f32   packed -> y_loc [nxl][ns]
 * Seven block passes per rank over its share instead of the single-device five; the generic plan (any other
 * shape) keeps the z_loc / index-packing protocol above.  Generic plan: nx, ns / 2 any (a prime factor > 31 turns
 * the slab's channel transform / the rows' time transform into a Bluestein convolution in global memory: scratch
 * <= 1 GiB each owned by the plan, D4W_FKD_BZ_CHUNK = columns / D4W_FKD_BT_CHUNK = rows per chunk; with such an
 * ns / 2 the half spectrum is one natural-order class, N1 = 1, owned by rank 0).
 * ------------------------------------------------------------------------------------------ */
typedef struct d4w_fkd_plan d4w_fkd_plan;
int d4w_fkd_plan_create(int nx, int ns, int world, int rank, d4w_fkd_plan** plan);
int d4w_fkd_plan_destroy(d4w_fkd_plan* plan);
int d4w_fkd_plan_info(const d4w_fkd_plan* plan, int* info12_host);
int d4w_fkd_plan_q1_owner(const d4w_fkd_plan* plan, int* owner_host /* [N1] */);
/* The Bluestein channel phase transforms only the slab columns whose folded gains -- or whose Hermitian partner column's --
 * are not all zero with the mask set last (the others leave the pair op as zeros): info2 = {those, all slab columns},
 * {0, 0} when the channel phase is not the global-memory Bluestein form.  D4W_FKD_BZ_BAND=0 transforms every column. */
int d4w_fkd_plan_live_columns(const d4w_fkd_plan* plan, int* info2_host);
/* mask: the full dense [nx][ns] float32 mask on the fftshift-ed grid (as d4w_fk_set_mask_dense_f32);
 * only the owned sub-rows are folded and kept */
int d4w_fkd_set_mask_dense_f32(d4w_fkd_plan* plan, const float* mask_shifted, void* stream);
/* the closed-form designs straight into the plan (see d4w_fk_set_mask_design_f32): a rank evaluates only the
 * gains of the sub-rows it owns, so no rank ever holds the dense mask.  Bit-identical to the dense route. */
int d4w_fkd_set_mask_design_f32(d4w_fkd_plan* plan, int mode, double k_spacing, double t_spacing,
                                const double* params8_host, int i0, int i1, const double* hrow_dev, void* stream);
int d4w_fkd_time_fwd_f32(d4w_fkd_plan* plan, const float* x_loc, float* z_loc, int taper, void* stream);
int d4w_fkd_chan_apply_f32(d4w_fkd_plan* plan, float* slab, void* stream);
int d4w_fkd_time_inv_f32(d4w_fkd_plan* plan, float* z_loc, void* stream);
int d4w_fkd_plan_is_packed(const d4w_fkd_plan* plan);
int d4w_fkd_time_fwd_packed_f32(d4w_fkd_plan* plan, const float* x_loc, float* packed, int taper, void* stream);
int d4w_fkd_time_inv_packed_f32(d4w_fkd_plan* plan, const float* packed, float* y_loc, void* stream);
/* the same on local rows [l0, l1) only (l0 a multiple of C1): row chunks whose transfer overlaps the next chunk's
 * transform (das4whales_amd/shard.py) */
int d4w_fkd_time_fwd_packed_rows_f32(d4w_fkd_plan* plan, const float* x_loc, float* packed, int taper, int l0, int l1,
                                     void* stream);
int d4w_fkd_time_inv_packed_rows_f32(d4w_fkd_plan* plan, const float* packed, float* y_loc, int l0, int l1, void* stream);
/* ... also leaving mean and max|.| of the filtered local rows (what detect.compute_cross_correlogram normalises by,
 * detect.py:157) in row_mean / row_maxabs [nxl], which the caller zeroes before the first chunk */
int d4w_fkd_time_inv_packed_rows_stats_f32(d4w_fkd_plan* plan, const float* packed, float* y_loc, int l0, int l1,
                                           double* row_mean, float* row_maxabs, void* stream);

/* dsp.taper_data (dsp.py:705-722): x *= tukey(ns, 0.03) in place, every row */
int d4w_taper_f32(float* x, int nx, int ns, void* stream);

/* ------------------------------------------------------------------------------------------
 * Zero-phase IIR filtering along time (rows independent):
 * replaces dsp.bp_filt (dsp.py:859-880: butter(8,'bp') + scipy.signal.filtfilt) and the user-side
 * scipy.signal.sosfiltfilt(sos, x, axis=1) applied to dsp.butterworth_filter designs
 * (dsp.py:789-827; Example.py:55, DAS4Whales_ExampleNotebook.md:292).
 *
 * SciPy semantics kept: odd extension by `padlen` samples on both sides, every section started
 * in its steady state for the first (extended) sample (sosfilt_zi * ext[0]), forward pass,
 * backward pass started at sosfilt_zi * y_fwd[last], crop.  The cascade runs as float32
 * second-order sections (the reference's 16-pole `ba` recursion is float64-only).
 *
 * Rows are cut into time segments of `seg_len` samples that run concurrently; a segment that
 * does not start at the row edge is warmed up over `warm` samples (host-chosen from the decay
 * of the cascade's impulse response, so the truncation error is below float32 resolution).
 * seg_len <= 0 or warm <= 0: one segment per row (exact recursion over the whole row).
 *
 *   sos_host [nsec][6]  = b0 b1 b2 a0(=1) a1 a2 (scipy layout), float64, HOST memory
 *   zi_host  [nsec][2]  = scipy.signal.sosfilt_zi(sos), float64, HOST memory
 *   ws: device workspace of d4w_sosfiltfilt_ws_bytes() bytes;  y may alias x.
 * Errors: D4W_EINVAL if ns <= padlen (SciPy raises ValueError there), nsec < 1 or nsec > 10.
 * ------------------------------------------------------------------------------------------ */
size_t d4w_sosfiltfilt_ws_bytes(int nx, int ns, int padlen);
int d4w_sosfiltfilt_f32(const float* x, float* y, int nx, int ns, const double* sos_host,
                        const double* zi_host, int nsec, int padlen, int seg_len, int warm,
                        void* ws, void* stream);
/* The two row-end pieces of every row filtered exactly and written into y -- what the overlap-save form of the band-pass
 * (d4w_fir_fft_f32: the interior) leaves to the recursion: the left piece x[r][0 .. piece) and the right piece
 * x[r][ns - piece .. ns) of every row run d4w_sosfiltfilt_f32's arithmetic as rows of `piece` samples (filtfilt's edge rule at
 * the true row end; the artificial cut at the inner end has decayed after piece - keep samples), read in place from x, and
 * their `keep` outer outputs go straight to y[r][0 .. keep) and y[r][ns - keep .. ns): no gathered copy of the pieces, no
 * scattered copy of the results.  The sections of a piece run on adjacent lanes (csrc/rowops.hip: sos_pass_lanes).
 * phase 0: both passes; 1: the forward pass only (reads x, fills ws); 2: the backward pass only (reads ws, writes y) -- a
 * caller that runs the interior kernel meanwhile orders phase 2 behind it (the interior writes columns [K, ns - K), the
 * pieces own [0, keep) and [ns - keep, ns), keep > K).  x and y must not alias; ws: d4w_sosfiltfilt_ends_ws_bytes. */
size_t d4w_sosfiltfilt_ends_ws_bytes(int nx, int piece, int padlen);
int d4w_sosfiltfilt_ends_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi, int nsec,
                             int padlen, int piece, int keep, int phase, void* ws, void* stream);
/* The same for ONE row end: sides = 1 the left pieces only, 2 the right pieces only (3 = both = the call above) -- a file
 * whose other end continues into a neighbouring file (das4whales_amd/stream.py: the first / last file of a record takes
 * d4w_fir_fft_halo_f32 with a stand-in halo on its free side and this call for that side's `keep` columns, after it). */
int d4w_sosfiltfilt_ends_sides_f32(const float* x, float* y, int nx, int ns, const double* sos, const double* zi, int nsec,
                                   int padlen, int piece, int keep, int phase, int sides, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused ingest (the step in front of the path): data_handle.load_das_data's channel selection and
 * float conversion + data_handle.raw2strain (data_handle.py:157-176,213-214):
 *   y[r][:] = (float64(raw[c0 + r*cstep][:]) - mean_r) * scale_factor,   r < nx_out
 * raw: DEVICE [nch][ns] row-major of raw_dtype 0 = int32 (OptaSense RawData), 1 = int16,
 * 2 = float32, 3 = float64; y: float32 [nx_out][ns].  One read of the selected raw rows, one write.
 * ------------------------------------------------------------------------------------------ */
int d4w_raw2strain_f32(const void* raw, int raw_dtype, int ns, int c0, int cstep, int nx_out,
                       double scale_factor, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Matched filter (time-domain cross-correlation, positive lags), rows independent:
 * replaces detect.compute_cross_correlogram (detect.py:140-166), detect.shift_xcorr
 * (detect.py:96-112) and the numerator of detect.shift_nxcorr (detect.py:115-137).
 *
 *   d4w_row_stats_f32:  mean[c] = mean(x[c,:]) as FLOAT64,  maxabs[c] = max|x[c,:]|   (detect.py:157).  The mean is
 *       accumulated about a pivot sample and kept in float64; every correlator splits it into hi + lo float32 parts and
 *       de-means as (x - hi) - lo, so a row whose offset is 10^3..10^5 x its signal is still de-meaned to float32 rounding
 *       of the signal (a float32 mean leaves |mean| 2^-24 in every sample).
 *   d4w_xcorr_f32:      for each template t < ntpl (1 or 2 per call, fused: x is read once)
 *       y_t[c][k] = g[c] * sum_{n < L_t, n+k < ns} (x[c][n+k] - m[c]) * taps[t][n]
 *       with m = mean (or 0 if mean == NULL) and g = 1/maxabs (or 1 if maxabs == NULL; 0 for an
 *       all-zero row, where the reference divides by zero).
 *   taps: DEVICE float32 [ntpl][ltaps] (ltaps multiple of 4, zero padded), the template's
 *   support already normalised by the host exactly as detect.py:158 does.
 * The de-meaned zero-padded template's DC tail (-mean/max on the padded part, detect.py:158) is a
 * separate entry point, d4w_xcorr_dc_tail_f32.
 * ------------------------------------------------------------------------------------------ */
int d4w_row_stats_f32(const float* x, int nx, int ns, double* mean, float* maxabs, void* stream);

/* dst[r][0 .. ncols) = src[r][0 .. ncols), r < nrows, rows ld_src / ld_dst floats apart (DEVICE pointers): the strided piece
 * copies around the band-pass (row ends to the recursion and back, dsp.py:859-880) in one launch each. */
int d4w_copy_cols_f32(const float* src, size_t ld_src, float* dst, size_t ld_dst, int nrows, int ncols, void* stream);
int d4w_xcorr_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs,
                  const float* taps, int ntpl, int ltaps, float* y0, float* y1, void* stream);
/* Same, with the true support of each template (len_t <= ltaps, taps[t][len_t..ltaps) == 0): taps
 * beyond the shorter support are only applied to the longer template (HF 136 / LF 156 samples). */
int d4w_xcorr_lens_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs,
                       const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0,
                       float* y1, void* stream);

/* The constant the reference leaves on the zero-padded part of the de-meaned template (detect.py:158):
 *   y[c][k] += coef * g[c] * sum_{i < k + support} (x[c][i] - m[c]),  k + support < ns,
 * coef = mean(template) / max|template| over the zero-padded length, support = length of the non-zero
 * part.  Added in place to a correlogram produced by d4w_xcorr_*_f32 with the same mean / maxabs.
 * |coef| ~ 5e-7 for the fin-whale templates: a 3-5e-6 effect on a 60-s file of WHITE rows, 3e-8 on band-passed rows, 1e-3 on
 * rows that drift (the term is |coef| g times a PREFIX SUM of the de-meaned row: its weight is a property of the data).
 *
 * d4w_row_prefix_max_f32:  pmax[c] = max_j |sum_{i < j} (x[c][i] - m[c])|  -- the most the term of row c can reach is
 *     |coef| g[c] pmax[c]; one read of x.
 * d4w_xcorr_dc_tail_rows_f32: the same update with a per-row decision: with pmax / rowmax given (rowmax[c] = max over the
 *     lags of y[c][:], d4w_xcorr_mm_rowmax_f32's epilogue) a row with |coef| g pmax <= eps max(rowmax, 0) is left alone --
 *     the omitted term is below eps of that row's largest correlation, whatever the data --, every other row receives the
 *     term and its rowmax entry is formed again.  pmax = rowmax = NULL: every row (d4w_xcorr_dc_tail_f32).  The Python
 *     mirror runs this with eps = 1e-6 (a tenth of the parity bar) whenever a zero-padded template has a non-zero mean. */
int d4w_xcorr_dc_tail_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs,
                          double coef, int support, float* y, void* stream);
int d4w_row_prefix_max_f32(const float* x, int nx, int ns, const double* mean, float* pmax, void* stream);
/* d4w_row_stats_f32 and d4w_row_prefix_max_f32 in one launch (the second sweep of a 60-s row is served by L2) */
int d4w_row_stats_prefix_f32(const float* x, int nx, int ns, double* mean, float* maxabs, float* pmax, void* stream);
int d4w_xcorr_dc_tail_rows_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs,
                               double coef, int support, float* y, const float* pmax, float* rowmax, double eps,
                               void* stream);

/* Overlap-save FFT form of the same correlation for short templates (support <=
 * d4w_xcorr_fft_max_support() = 161 samples; the fin-whale templates have 136 / 156): blocks of
 * 4096 samples, one forward transform per block and one inverse per template, ~5x fewer flops than
 * the direct form, bound by HBM instead of the vector ALUs.  Same arguments and result as
 * d4w_xcorr_lens_f32 (agreement to float32 rounding); ws = DEVICE scratch of
 * d4w_xcorr_fft_ws_bytes() bytes (template spectra, rebuilt per call on `stream`). */
/* taps may be NULL in d4w_xcorr_fft_cont_f32 / d4w_fir_fft_f32 / d4w_fir_fft_halo_f32 when `ws` still holds the tables a
 * previous call on the same stream built from the same taps (same supports / half width): consecutive files of a record
 * then pay the template spectra once. */
int d4w_xcorr_fft_max_support(void);
size_t d4w_xcorr_fft_ws_bytes(void);
int d4w_xcorr_fft_f32(const float* x, int nx, int ns, const double* mean, const float* maxabs,
                      const float* taps, int ntpl, int ltaps, int len0, int len1, float* y0,
                      float* y1, void* ws, void* stream);
/* The same for a record that continues: lags whose window runs past sample ns - 1 read the first n_next samples of
 * xnext [nx][ld_next] (the head of the next file of the same cable, filtered the same way; de-meaned with THIS file's
 * row means like the row's own samples) instead of zeros -- what das4whales_amd/stream.py used to do by concatenating
 * the two files first (no reference counterpart, parity target = the reference run on the concatenated
 * record).  Two templates per call (the fused kernel).  xnext = NULL: identical to d4w_xcorr_fft_f32. */
int d4w_xcorr_fft_cont_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next,
                           const double* mean, const float* maxabs, const float* taps, int ntpl, int ltaps,
                           int len0, int len1, float* y0, float* y1, void* ws, void* stream);

/* The same correlation as a banded-Toeplitz product on the matrix cores -- the form detect.compute_cross_correlogram
 * (detect.py:140-166) runs by default since round 4 (das4whales_amd/csrc/xcorr_mm.hip):
 *   C[i][a] = sum_u t[u - i] x[16 a + u]  per 256 lags, v_mfma_f32_16x16x32_f16 on binary16 hi / lo splits of the float32
 *   operands (three of the four partial products, float32 accumulation; agreement with d4w_xcorr_lens_f32 and a float64
 *   correlation to float32 rounding), one read of x and one write per correlogram, no workspace.
 * supports <= d4w_xcorr_mm_max_support() = 7936; ntpl = 1 or 2 (fused into one launch while both supports are <= 177; beyond,
 * one launch per template up to 497 taps -- the 450-sample template of scripts/main_mfdetect.py:70 --, and longer templates in
 * sections of 496 taps, each section one launch that reads x shifted by its first tap and accumulates into y); non-finite samples spread over the 256 lags of their tile in both templates.
 * mean / maxabs must be the rows' OWN statistics (|x - mean| / maxabs <= 2): with a continuation (xnext), whose head is scaled by
 * the row's statistics, and under D4W_MM_CLAMP=1 scaled samples beyond binary16's range are clamped to +-65504; elsewhere
 * statistics of another tensor can turn tiles into NaN (round 6: the clamp cost 2 % of the kernel on every call);
 * taps DEVICE [ntpl][ltaps], always needed (the
 * Toeplitz fragments are built in the kernel); mean / maxabs as d4w_xcorr_lens_f32 (maxabs == NULL: every 4096-lag chunk
 * is scaled by its own power of two); xnext / ld_next / n_next as d4w_xcorr_fft_cont_f32 (NULL, 0, 0: zeros behind the row). */
int d4w_xcorr_mm_max_support(void);
int d4w_xcorr_mm_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next,
                     const double* mean, const float* maxabs, const float* taps, int ntpl, int ltaps,
                     int len0, int len1, float* y0, float* y1, void* stream);
/* The same, also leaving rowmax_t[c] = max_k y_t[c][k] (DEVICE float32 [nx] per template; rowmax1 only with ntpl = 2) from the
 * kernel's epilogue: what a detection threshold is set from (scripts/main_mfdetect.py:82,95: 0.5 * max of the correlograms)
 * without reading the correlograms again.  NULL, NULL: d4w_xcorr_mm_f32. */
int d4w_xcorr_mm_rowmax_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next,
                            const double* mean, const float* maxabs, const float* taps, int ntpl, int ltaps,
                            int len0, int len1, float* y0, float* y1, float* rowmax0, float* rowmax1, void* stream);
/* The same, with the constant TAIL of the de-meaned zero-padded template added in the kernel's epilogue -- the whole of
 * detect.compute_cross_correlogram (detect.py:156-166: the template is normalised over its zero-padded length, detect.py:158, which
 * leaves -mean(t) / max|t| on the padding) in ONE pass over x, exact on every row:
 *   y_t[c][k] += tail_t * g[c] * sum_{i < k + len_t} (x[c][i] - m[c]),  k + len_t < ns;   tail_t = mean(t) / max|t| (0: no term).
 * (d4w_xcorr_dc_tail_rows_f32 adds the same term in a second pass over x and y, decided per row; this entry replaces the pair for
 * supports up to d4w_xcorr_mm_tail_max_support() = 497: one launch per template.)  Inside a block of 16 lags the term is part of
 * the Toeplitz product itself (the constant added to the staged taps); per block one prefix from a scan in the sample-conversion
 * phase; the prefix at a chunk's start carried along the row in float64 -- a workgroup walks whole rows (csrc/xcorr_mm.hip).
 * Needs mean and maxabs.  tail0 == tail1 == 0: d4w_xcorr_mm_rowmax_f32. */
int d4w_xcorr_mm_tail_max_support(void);
int d4w_xcorr_mm_tail_f32(const float* x, int nx, int ns, const float* xnext, int ld_next, int n_next,
                          const double* mean, const float* maxabs, const float* taps, int ntpl, int ltaps,
                          int len0, int len1, double tail0, double tail1, float* y0, float* y1,
                          float* rowmax0, float* rowmax1, void* stream);

/* Zero-phase FIR along time by overlap-save FFT blocks -- the INTERIOR of dsp.bp_filt / scipy.signal.sosfiltfilt
 * (dsp.py:859-880): away from the row ends a zero-phase IIR filter is the convolution with its two-sided response
 * g = h * h(-t), truncated at half width K where it has decayed below the tolerance (north star: "overlap-save FIR").
 *   y[r][n] = sum_{j <= 2K} taps[j] x[r][n - K + j]   for K <= n < ns - K;  the K columns at either row end are not
 *   written (the caller fills them with the exact recursion, d4w_sosfiltfilt_f32 on short pieces).
 * first[r] (device, e.g. the row's first sample) is subtracted before the transform and first[r] * dc_gain added
 * back, which keeps a large offset out of the float32 transform (dc_gain = |H(1)|^2 of the exact filter).
 * One read and one write of the block (8 B per sample).  K even, <= d4w_fir_fft_max_halfwidth(); ws as d4w_xcorr_fft_f32. */
int d4w_fir_fft_max_halfwidth(void);
int d4w_fir_fft_f32(const float* x, int nx, int ns, const float* taps, int K, const float* first, double dc_gain,
                    float* y, void* ws, void* stream);
/* The same filter writing only the output columns [col0, col1) (K <= col0 < col1 <= ns - K): the interior kernel of a band-pass
 * whose row-end columns are owned by d4w_sosfiltfilt_ends_f32 then touches none of them, and the two run side by side. */
int d4w_fir_fft_cols_f32(const float* x, int nx, int ns, const float* taps, int K, const float* first,
                         double dc_gain, float* y, int col0, int col1, void* ws, void* stream);
/* The same for a row that has neighbours on both sides (consecutive files of one record, das4whales_amd/stream.py):
 * left [nx][ld_left] holds the n_left samples before every row, right [nx][ld_right] the n_right samples after it
 * (n_left, n_right >= K), read in place -- no concatenated copy -- and ALL ns columns of y are written:
 *   y[r][n] = sum_j taps[j] xv[r][n - K + j],   xv = [left | x | right].
 * The row ends of a record (no neighbour) keep filtfilt's edge rule: d4w_sosfiltfilt_f32. */
int d4w_fir_fft_halo_f32(const float* x, int nx, int ns, const float* left, int ld_left, int n_left,
                         const float* right, int ld_right, int n_right, const float* taps, int K,
                         const float* first, double dc_gain, float* y, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * f-k mask design on the device (one-off per shape), float32 masks on the fftshift-ed (k, f)
 * grid, row-major [nx][ns] -- what d4w_fk_set_mask_dense_f32 takes.  Closed forms of the
 * reference's row / column loops; axis values in float64 as NumPy forms them.
 *   mode 0  dsp.fk_filter_design          dsp.py:85-171    params {cs_min, cp_min, cp_max, cs_max}
 *   mode 1  dsp.hybrid_filter_design      dsp.py:174-305   params {cs_min, cp_min, fmin, fmax}
 *   mode 2  dsp.hybrid_ninf_filter_design dsp.py:308-454   params {cs_min, cp_min, cp_max, cs_max};
 *           hrow_dev = DEVICE float64 [ns] band-pass row (zeros ++ |freqz(butter)|^2, dsp.py:348-349)
 *   mode 3  hybrid_gs_filter_design      before the blur (dsp.py:508-539)  params {-, cp_min, fmin, fmax}
 *   mode 4  hybrid_ninf_gs_filter_design before the blur (dsp.py:633-653)  params {-, cp_min, cp_max, -, fmin, fmax}
 *   mode 5  dsp.fk_filt wedge            before the blur (dsp.py:930-936)  params {c_min, c_max}
 * k_spacing = selected_channels[2]*dx (dsp.py:130), t_spacing = 1/fs; [i0, i1) = the half-open
 * column range the reference's loop runs over (np.argmax(f >= ...), host).
 * d4w_gaussian_filter_f32 = scipy.ndimage.gaussian_filter(., sigma) (truncate 4, reflect), used at
 * dsp.py:540,659,940; d4w_flip_sum_f32 = M + fliplr(M), then + flipud (dsp.py:660-661);
 * d4w_minmax_normalise_f32 = (g - min)/(max - min) in place (dsp.py:945), synchronises `stream`; a constant g
 * gives NaN everywhere like NumPy's 0 / 0.
 * ------------------------------------------------------------------------------------------ */
int d4w_design_mask_f32(int mode, int nx, int ns, double k_spacing, double t_spacing,
                        const double* params8_host, int i0, int i1, const double* hrow_dev,
                        float* mask, void* stream);
int d4w_flip_sum_f32(const float* in, float* out, int nx, int ns, void* stream);
int d4w_gaussian_filter_f32(const float* in, float* out, float* tmp, int nx, int ns, double sigma,
                            void* stream);
int d4w_minmax_normalise_f32(float* x, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row spectral operators (rows independent; every transform runs inside one workgroup's LDS).
 *
 * d4w_analytic_f32: scipy.signal.hilbert(x, axis=1) = ifft(fft(x) * h) per row, z = x + i H[x]
 *   mode 0  y = |z|                      envelope        (detect.py:192,217; scripts/main_mfdetect.py:58)
 *   mode 1  y = imag(z) = H[x]
 *   mode 2  y = 10 log10(|z|^2 / var[c]) var = DEVICE [nx] row variances   (dsp.py:975)
 *   mode 3  y[c][i] = diff(unwrap(angle z))[i] / (2 pi) * fs, [nx][ns-1]   (dsp.instant_freq, dsp.py:830-856)
 *   mode 4  y = |z| / sqrt(var[c])       improcess.trace2image before scaling   (improcess.py:60)
 *   Rows whose transform (ns / 2 packed values for even ns, ns for odd; the Bluestein convolution
 *   length when that has a prime factor > 31) fits one workgroup's LDS: d4w_analytic_row_fits_lds(ns);
 *   D4W_EINVAL otherwise -- those rows go through d4w_analytic_long_f32.
 * d4w_row_var_f32: var[c] = np.std(x[c])**2 (population).
 * d4w_snr_f32: dsp.snr_tr_array (dsp.py:956-976): 10 log10(x^2 / std^2) (env = 0) or the envelope
 *   form (env != 0); var_ws = DEVICE [nx] scratch.
 * d4w_fx_f32: dsp.get_fx (dsp.py:18-38): y[c][j] = 2 |fftshift(fft(x[c], nfft))|[j] / nfft * 1e9,
 *   y is [nx][nfft]; rows are cropped / zero-padded to nfft like np.fft.fft(x, nfft).
 * ------------------------------------------------------------------------------------------ */
int d4w_analytic_f32(const float* x, float* y, int nx, int ns, int mode, const float* var, double fs,
                     void* stream);
/* Rows too long for one workgroup's LDS (d4w_analytic_row_fits_lds(ns) == 0, e.g. 120 000 samples):
 * same modes through the four-step time-axis transform of the distributed f-k plan (two passes
 * each way over HBM) + the Hilbert pair op + one combine pass; any ns (odd ns: the rows as complex
 * sequences of their own length, scipy's one-sided multiplier on the spectrum; a prime factor > 31
 * of the transform length: the global-memory Bluestein form); ws = DEVICE scratch of
 * d4w_analytic_long_ws_bytes(nx, ns) bytes (4 nx ns for even ns, 8 nx ns for odd).  Shapes [nx][ns] with shape-specialised f-k kernels (built in or registered
 * through d4w_fk_register_shape) run that plan's time phase, its pass B with the Hilbert pair operation on real rows and
 * the inverse time phase instead (three fast passes + the combine pass; the ns / 2 prime-factor limit applies). */
int d4w_analytic_row_fits_lds(int ns);
size_t d4w_analytic_long_ws_bytes(int nx, int ns);
int d4w_analytic_long_f32(const float* x, float* y, int nx, int ns, int mode, const float* var, double fs,
                          void* ws, void* stream);
/* Frees the plans d4w_analytic_long_f32 keeps per (device, nx, ns) (it synchronises the device first; they are also dropped
 * once more than 16 shapes have been seen, so a stream of varying channel selections does not accumulate device tables). */
int d4w_analytic_long_clear(void);
int d4w_row_var_f32(const float* x, int nx, int ns, float* var, void* stream);
int d4w_snr_f32(const float* x, float* y, int nx, int ns, int env, float* var_ws, void* stream);
int d4w_fx_f32(const float* x, float* y, int nx, int ns, int nfft, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spectrograms and spectrogram correlation: replaces dsp.get_spectrogram (dsp.py:41-78),
 * detect.get_sliced_nspectrogram (detect.py:334-408), detect.xcorr2d (detect.py:579-602),
 * detect.xcorr (detect.py:605-647) and the per-channel loop of
 * detect.compute_cross_correlogram_spectrocorr (detect.py:650-709).
 *
 * d4w_stft_mag_f32: S[c][b - bin_lo][t] = |librosa.stft(x[c], n_fft, hop_length=hop)|[b][t] for
 *   bin_lo <= b <= bin_hi, t < d4w_stft_frames(ns, hop) = 1 + ns/hop (periodic Hann, center=True
 *   with zero padding: librosa >= 0.10 defaults).  rowmax[c] = max over ALL bins and frames (what
 *   detect.py:387 / dsp.py:76 normalise by).  n_fft even, <= 6144; a window with a prime factor > 31 runs its frame
 *   transform as a Bluestein convolution.  rowmax may be NULL for n_fft = 128, 160, 256, 512
 *   (two-factor register transforms): then only the kept bins are formed (detect.compute_cross_correlogram_spectrocorr,
 *   where the normalisation cancels against the median).
 * d4w_scale_rows_f32: S[c][:] /= denom[c] (mode 0) or 20 log10(S[c][:] / denom[c]) (mode 1).
 * d4w_row_median_f32: med[c] = np.median(v[c][0:per_row]).
 * d4w_spectrocorr_f32:
 *   raw[c][t] = sum_f sum_j S[c][f][t + j - off] K[f][j]  (zero outside the spectrogram), t < nout
 *   out[c][t] = max(raw, 0) / (med[c] * nk);  zero_ends != 0 also forces out[c][0] = out[c][nout-1] = 0.
 *   off = nk/2, nout = nt: detect.xcorr2d;  off = 0, nout = nt-nk+1, zero_ends: detect.xcorr.
 *   S [nx][nf][nt], K [nf][nk], med [nx], out [nx][nout], all DEVICE float32.
 * ------------------------------------------------------------------------------------------ */
int d4w_stft_frames(int ns, int hop);
int d4w_stft_mag_f32(const float* x, float* S, float* rowmax, int nx, int ns, int n_fft, int hop,
                     int bin_lo, int bin_hi, void* stream);
/* The same magnitudes for the detector's call shape -- a few kept bins of a heavily overlapped transform, no row maximum
 * (detect.compute_cross_correlogram_spectrocorr, detect.py:650-709: n_fft = 160, hop = 8, the 13 bins between 14 and 30 Hz) --
 * as ONE matrix product per 16 frames on the matrix cores: X[b][t] = sum_u (w[u] e^{-2 pi i b u / N}) x[hop t - N/2 + u],
 * operands as binary16 hi / lo pairs with float32 accumulation (das4whales_amd/csrc/stft_mm.hip).  d4w_stft_mag_f32 takes this
 * route by itself when rowmax == NULL and d4w_stft_mm_eligible(...) = 1: n_fft % 32 == 0 <= 160, hop % 8 == 0 <= 32, at most
 * 16 kept bins (D4W_STFT_MM=0 switches it off). */
int d4w_stft_mm_eligible(int n_fft, int hop, int bin_lo, int bin_hi);
int d4w_stft_mag_mm_f32(const float* x, float* S, int nx, int ns, int n_fft, int hop, int bin_lo, int bin_hi,
                        void* stream);
int d4w_scale_rows_f32(float* S, int nx, size_t per_row, const float* denom, int mode, void* stream);
int d4w_row_median_f32(const float* v, int nx, size_t per_row, float* med, void* stream);
int d4w_spectrocorr_f32(const float* S, int nx, int nf, int nt, const float* K, int nk, int off,
                        int nout, const float* med, int zero_ends, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Peak picking: scipy.signal.find_peaks(x[c], prominence=thr)[0] per row, replaces the loops of
 * detect.pick_times / pick_times_env / process_corr / pick_times_par (detect.py:169-274; the
 * envelope of the *_env variants is d4w_analytic_f32 mode 0).  Strict local maxima with plateaus
 * reported at their middle sample, prominence with wlen=None, kept when prominence >= thr
 * (compared in float64, as SciPy does on the float64 view of the same samples).
 *   idx [nx][cap] int32: the first min(counts[c], cap) peak positions of row c in time order;
 *   counts[c] = number of peaks found (may exceed cap: call again with a larger cap; ns/2 always suffices).
 * ------------------------------------------------------------------------------------------ */
int d4w_find_peaks_f32(const float* x, int nx, int ns, double prominence, int32_t* idx,
                       int32_t* counts, int cap, void* stream);
/* The same with the threshold formed on the device: prominence = scale * value[0] (float64 product of the float32 DEVICE
 * scalar, exactly the host's 0.45 * float(np.max(corr)) of scripts/main_mfdetect.py:82,95) -- a chain that takes its
 * threshold from the block's largest correlation (d4w_minmax_f32, d4w_xcorr_mm_rowmax_f32) need not wait for it. */
int d4w_find_peaks_dthr_f32(const float* x, int nx, int ns, const float* value, double scale, int32_t* idx,
                            int32_t* counts, int cap, void* stream);
/* offsets[c] = counts[0] + ... + counts[c] (int64, what d4w_pack_picks_i64 places the rows by) and
 * summary2 = {max_c counts[c], sum_c counts[c]} (DEVICE int64[2]: the capacity check and the table size of one picker call). */
int d4w_pick_offsets_i64(const int32_t* counts, int nx, int64_t* offsets, int64_t* summary2, void* stream);
/* The picks of all rows as one packed table out[2][total] int64, row 0 = channel index, row 1 = time index
 * -- what detect.convert_pick_times builds from the ragged lists (detect.py:277-303; the reference's docstring
 * has the two rows swapped, :289 vs :297-299).  offsets[c] = counts[0] + ... + counts[c] (inclusive prefix sum,
 * device), total = offsets[nx - 1]; every counts[c] must be <= cap. */
int d4w_pack_picks_i64(const int32_t* idx, const int32_t* counts, const int64_t* offsets, int nx, int cap,
                       int64_t total, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image operators of the Gabor detector: replaces improcess.scale_pixels /
 * trace2image (improcess.py:23-62), improcess.binning (improcess.py:395-420, torchvision Resize =
 * antialiased bilinear interpolation), cv2.filter2D and the thresholds of
 * scripts/main_gabordetect.py:109-134, improcess.apply_smooth_mask (improcess.py:423-454).
 * Images are row-major float32 [h][w] on the DEVICE.
 *
 * d4w_minmax_f32: minmax[0] = min(x), minmax[1] = max(x) over n values; minmax = DEVICE [2].
 * d4w_scale_pixels_f32: y = (x - minmax[0]) / (minmax[1] - minmax[0]) * gain; y may alias x.
 * d4w_threshold_f32: y = (x > thr) ? 1 : 0, compared in float64.
 * d4w_mask_mul_f32: y = x where mask != 0, else 0 (array * bool mask).
 * d4w_resize_bilinear_aa_f32: torch.nn.functional.interpolate(x, (oh, ow), mode="bilinear",
 *   align_corners=False, antialias=True) -- what torchvision 0.17 Resize runs; horizontal pass, then
 *   vertical pass; ws = DEVICE scratch of d4w_resize_ws_bytes(h, w, oh, ow) bytes.
 * d4w_filter2d_f32: cv2.filter2D(img, CV_64F, kernel): out[y][x] = sum K[ky][kx] *
 *   img[y + ky - kh/2][x + kx - kw/2] with BORDER_REFLECT_101; kernel = DEVICE [kh][kw];
 *   accumulate != 0 adds to out; ws = DEVICE scratch of d4w_filter2d_ws_bytes(kh, kw) bytes.
 *   (64 + kw - 1)(32 + kh - 1) floats must fit in 160 KiB of LDS (101 x 101: 85 KiB).
 * ------------------------------------------------------------------------------------------ */
int d4w_minmax_f32(const float* x, size_t n, float* minmax, void* stream);
int d4w_scale_pixels_f32(const float* x, float* y, size_t n, const float* minmax, double gain, void* stream);
int d4w_threshold_f32(const float* x, float* y, size_t n, double thr, void* stream);
int d4w_mask_mul_f32(const float* x, const float* mask, float* y, size_t n, void* stream);
size_t d4w_resize_ws_bytes(int h, int w, int oh, int ow);
int d4w_resize_bilinear_aa_f32(const float* x, int h, int w, float* y, int oh, int ow, void* ws, void* stream);
size_t d4w_filter2d_ws_bytes(int kh, int kw);
int d4w_filter2d_f32(const float* img, int h, int w, const float* kernel, int kh, int kw, float* out,
                     int accumulate, void* ws, void* stream);
/* The same correlation on the matrix cores for kernels of <= 113 columns (the detector's 101 x 101 Gabor kernels): kernel row
 * by kernel row a banded-Toeplitz product, out[y][16 a + i] += sum_u K[j][u - i] P[y + j][16 a + u], operands as binary16
 * hi / lo pairs with float32 accumulation (das4whales_amd/csrc/filter2d_mm.hip).  d4w_filter2d_f32 takes this route by itself
 * when d4w_filter2d_mm_eligible(kh, kw) = 1 (D4W_F2D_MM=0 switches it off); ws of the stand-alone entry point: DEVICE scratch of
 * d4w_filter2d_mm_ws_bytes(kh, kw) bytes. */
int d4w_filter2d_mm_eligible(int kh, int kw);
size_t d4w_filter2d_mm_ws_bytes(int kh, int kw);
int d4w_filter2d_mm_f32(const float* img, int h, int w, const float* kernel, int kh, int kw, float* out,
                        int accumulate, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D4W_H */
