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
 * index tables and the folded, permuted mask.  Requirements: ns even; nx and ns/2 factor into
 * primes <= 31.  Plans are immutable after set_mask and may be shared by streams that serialise
 * their own calls.
 * ------------------------------------------------------------------------------------------ */
typedef struct d4w_fk_plan d4w_fk_plan;

int d4w_fk_plan_create(int nx, int ns, d4w_fk_plan** plan);
/* opts: NULL or 6 ints {C1, C2, N1, N2, TA, TC}; any entry <= 0 keeps the planner's choice */
int d4w_fk_plan_create_ex(int nx, int ns, const int* opts_host, d4w_fk_plan** plan);
int d4w_fk_plan_destroy(d4w_fk_plan* plan);
/* fills {nx, ns, C1, C2, N1, N2, TA, TC} */
int d4w_fk_plan_info(const d4w_fk_plan* plan, int* info8_host);

/* Mask as the reference hands it over: dense [nx][ns] float32 on the fftshift-ed (k, f) grid
 * (dsp.py:129-130,137).  Folds it to its Hermitian part M_h = (M(k,f)+M(-k,-f))/2 -- which is
 * what `.real` at dsp.py:756,786 keeps -- and stores it in the plan's internal order. */
int d4w_fk_set_mask_dense_f32(d4w_fk_plan* plan, const float* mask_shifted, void* stream);

/* x -> y (may alias).  taper != 0 multiplies x by tukey(ns, 0.03) first (dsp.taper_data,
 * dsp.py:705-722, fused into the first pass; x itself is not modified). */
int d4w_fk_apply_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, void* stream);

/* Same as d4w_fk_apply_f32 but brackets each of the five passes (A, C, B, C', A') with HIP events
 * on `stream`, synchronises, and returns their durations in milliseconds in ms5_host[0..4].
 * Measurement aid for bench.py's roofline report; not for production loops. */
int d4w_fk_apply_timed_f32(d4w_fk_plan* plan, const float* x, float* y, int taper, void* stream,
                           float* ms5_host);

/* dsp.taper_data (dsp.py:705-722): x *= tukey(ns, 0.03) in place, every row */
int d4w_taper_f32(float* x, int nx, int ns, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D4W_H */
