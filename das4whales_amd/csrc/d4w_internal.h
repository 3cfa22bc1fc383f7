// das4whales_amd -- internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#ifndef D4W_EMU
#include <hip/hip_runtime.h>
#endif
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/d4w.h"

// ---------------------------------------------------------------------------------------------
// error plumbing: every C entry point returns an int status and leaves a message behind
// ---------------------------------------------------------------------------------------------
namespace d4w {
inline thread_local char g_err[512] = "";
// Two kernel families that must not be resident on a compute unit together (round 5, scripts/probe/stream_race2.py): with the
// matrix-core STFT (stft_mm_rows) running from ANOTHER HIP stream, the overlap-save FFT kernels (xcorr_fft.hip: band-pass,
// FFT-form matched filter) returned whole blocks 1-10 % off in a few workgroups per launch.  Round 5 serialised the two families
// across streams with this fence; round 6 found a FOREIGN kernel with the same effect (a rocBLAS GEMM), which no fence can know,
// and made the overlap-save kernels claim their compute unit's LDS instead (xcorr_fft.hip: xf_lds_claim, SUBS) -- nothing that
// needs LDS can be resident beside them.  The fence is OFF by default now (D4W_HAZARD_FENCE=1 turns it on): a launch of one
// family then first waits (on the device) for the last launch of the other, whichever stream that was on.  hazard_enter returns
// HOLDING the fence's host mutex and hazard_leave releases it: wait, launches and record are one critical section (two host
// threads cannot both pass the wait before either has recorded).  Limits: per process; devices 0..63 (beyond: the call fails).
//   rc = hazard_enter(self, stream); if (rc) return rc; ... launches ...; hazard_leave(self, stream);   (always paired)
//   self: 0 = overlap-save FFT kernels, 1 = stft_mm
int hazard_enter(int self, void* stream);
int hazard_leave(int self, void* stream);

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace d4w

#define D4W_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return d4w::fail(D4W_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define D4W_LAUNCH(kernel, grid, block, shmem, stream, ...)                                    \
    do {                                                                                       \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__);    \
        D4W_HIP(hipGetLastError());                                                            \
    } while (0)

// dynamic LDS declaration usable by both hipcc and the emulator.  (The switches between the gfx950 build and the CPU test
// build of the same sources -- tests/emu, -DD4W_EMU -- live in this header and in mm_common.h only; the kernels and their
// launch code are the same text in both, the test build supplying the HIP runtime calls and builtins they use.)
#ifdef D4W_EMU
#define D4W_DYN_LDS(name) unsigned char* name = hipemu::dyn_smem
#define D4W_BUILD_TAG "d4w 0.1 emu"
#else
#define D4W_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define D4W_BUILD_TAG "d4w 0.1 gfx950"
#endif

// ---------------------------------------------------------------------------------------------
// complex arithmetic on float2 (x = re, y = im)
// ---------------------------------------------------------------------------------------------
namespace d4w {
__device__ __forceinline__ float2 c_make(float re, float im) { return make_float2(re, im); }
__device__ __forceinline__ float2 c_conj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by -i : (x, y) -> (y, -x)
__device__ __forceinline__ float2 c_mul_mi(float2 a) { return make_float2(a.y, -a.x); }
// multiply by +i : (x, y) -> (-y, x)
__device__ __forceinline__ float2 c_mul_pi(float2 a) { return make_float2(-a.y, a.x); }
#if defined(D4W_PKMATH) && !defined(D4W_EMU)
// Packed-math variant: a complex value is one 64-bit VGPR pair and add / sub / scale / multiply are
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (half the VALU issue slots of the scalar forms; the FFT
// butterflies are add-dominated and VALU-bound).
typedef float d4w_pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d4w_pk2 pk_(float2 a) { d4w_pk2 r; r.x = a.x; r.y = a.y; return r; }
__device__ __forceinline__ float2 unpk_(d4w_pk2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return unpk_(pk_(a) + pk_(b)); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return unpk_(pk_(a) - pk_(b)); }
__device__ __forceinline__ float2 c_scale(float2 a, float s) { d4w_pk2 sv; sv.x = s; sv.y = s; return unpk_(pk_(a) * sv); }
// a * b = (a.x, a.x) * b + (-a.y, a.y) * (b.y, b.x)
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
    d4w_pk2 ax; ax.x = a.x; ax.y = a.x;
    d4w_pk2 ay; ay.x = -a.y; ay.y = a.y;
    d4w_pk2 bs; bs.x = b.y; bs.y = b.x;
    return unpk_(__builtin_elementwise_fma(ax, pk_(b), ay * bs));
}
// a * conj(b) = (b.x, b.x) * a + (a.y, -a.x) * (b.y, b.y)
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {
    d4w_pk2 bx; bx.x = b.x; bx.y = b.x;
    d4w_pk2 by; by.x = b.y; by.y = b.y;
    d4w_pk2 as; as.x = a.y; as.y = -a.x;
    return unpk_(__builtin_elementwise_fma(bx, pk_(a), as * by));
}
#else
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_scale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// a * b
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
#endif

// Streaming (non-temporal) store for write-once outputs far larger than L2 + MALL (the correlograms):
// fused matched filter 8.05 -> 7.78 ms at 20000 x 120000.  (No effect on the f-k passes, which keep plain stores.)
__device__ __forceinline__ void st_stream(float2* p, float2 v) {
#ifdef D4W_EMU
    *p = v;
#else
    typedef float d4w_f2v __attribute__((ext_vector_type(2)));
    d4w_f2v t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<d4w_f2v*>(p));
#endif
}

// 24-bit integer multiply (full-rate v_mul_i32_i24; v_mul_lo_u32 is quarter rate) and fast reciprocal
#ifdef D4W_EMU
__device__ __forceinline__ int d4w_mul24(int a, int b) { return a * b; }
__device__ __forceinline__ float d4w_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ int d4w_mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ float d4w_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// Cache policy of the f-k passes' streaming accesses to the block (round 6).  A 9.6-GB block is read once and written once per
// pass and never fits a cache; scripts/probe/copy_ceiling.hip: non-temporal loads read at up to 7.4 TB/s against 6.5 plain, an
// in-place sweep with nt loads + nt stores ran 5.68 against 5.38 TB/s.  In the passes themselves (variant builds on one box,
// profiles/r06c/fk_policy_ab.txt) it pays where a pass reads and writes the same tile in place AND nothing reads the result
// back soon: pass A' (3.57-3.75 ms against 3.80-3.86) and pass C forward (2.43-2.45 against 2.52-2.53); on pass A forward nt
// STORES cost 8 % (4.18 against 3.87-3.92 ms: its output is pass C's input) and pass C inverse is a wash.  NT = 1: non-temporal.
template <int NT>
__device__ __forceinline__ float2 fk_ldg(const float2* p) {
#ifndef D4W_EMU
    if constexpr (NT == 1) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
        return make_float2(v.x, v.y);
    }
#endif
    return *p;
}
template <int NT>
__device__ __forceinline__ void fk_stg(float2* p, float2 v) {
#ifndef D4W_EMU
    if constexpr (NT == 1) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v t; t.x = v.x; t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<f2v*>(p));
        return;
    }
#endif
    *p = v;
}
#ifndef D4W_FK_NT_AINV
#define D4W_FK_NT_AINV 1
#endif
#ifndef D4W_FK_NT_CFWD
#define D4W_FK_NT_CFWD 1
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() on gfx9 lowers to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, which would drain the global prefetch loads and the
// tile stores that the pipelined pass kernels deliberately keep in flight across the FFT stages.
__device__ __forceinline__ void lds_barrier() {
#ifdef D4W_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// 16-byte LDS read that stays one ds_read_b128: left to itself hipcc splits a float4 LDS load into
// ds_read2_b32 pairs, whose banking (dword index mod 32) turns a 16-byte lane stride into a 4-way
// conflict.  The volatile ext-vector access is not split.
__device__ __forceinline__ float4 lds_read4(const float4* p) {
#ifdef D4W_EMU
    return *p;
#else
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef const volatile f4_t __attribute__((address_space(3))) * lds_f4_ptr;
    const f4_t v = *(lds_f4_ptr)(p);
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}

// two-lane packed float (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950)
#ifdef D4W_EMU
struct v2f { float x, y; };
__device__ __forceinline__ v2f v2_make(float x, float y) { return v2f{x, y}; }
__device__ __forceinline__ float v2_x(v2f a) { return a.x; }
__device__ __forceinline__ float v2_y(v2f a) { return a.y; }
// a * s + c  (s broadcast)
__device__ __forceinline__ v2f v2_fma(v2f a, float s, v2f c) { return v2f{fmaf(a.x, s, c.x), fmaf(a.y, s, c.y)}; }
__device__ __forceinline__ v2f v2_fnma(v2f a, float s, v2f c) { return v2f{fmaf(-a.x, s, c.x), fmaf(-a.y, s, c.y)}; }
__device__ __forceinline__ v2f v2_add(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f v2_sub(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f v2_mul(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f v2_muls(v2f a, float s) { return v2f{a.x * s, a.y * s}; }
__device__ __forceinline__ v2f v2_neg(v2f a) { return v2f{-a.x, -a.y}; }
#else
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f v2_make(float x, float y) { v2f r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ float v2_x(v2f a) { return a.x; }
__device__ __forceinline__ float v2_y(v2f a) { return a.y; }
__device__ __forceinline__ v2f v2_fma(v2f a, float s, v2f c) {
    v2f sv; sv.x = s; sv.y = s;
    return __builtin_elementwise_fma(a, sv, c);
}
// c - a * s
__device__ __forceinline__ v2f v2_fnma(v2f a, float s, v2f c) {
    v2f sv; sv.x = -s; sv.y = -s;
    return __builtin_elementwise_fma(a, sv, c);
}
__device__ __forceinline__ v2f v2_add(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f v2_sub(v2f a, v2f b) { return a - b; }
__device__ __forceinline__ v2f v2_mul(v2f a, v2f b) { return a * b; }
__device__ __forceinline__ v2f v2_muls(v2f a, float s) { v2f sv; sv.x = s; sv.y = s; return a * sv; }
__device__ __forceinline__ v2f v2_neg(v2f a) { return -a; }
#endif

// Row mean of the matched filter's normalisation (detect.py:157 de-means in float64).  The producers (d4w_row_stats_f32,
// the f-k filter's last pass) leave it as a float64 per row; a consumer splits it once per row into a two-float value
// hi + lo and de-means a float32 sample as (x - hi) - lo: x - hi is exact whenever the row's offset dominates its signal
// (Sterbenz), so the de-meaned sample is good to float32 rounding of the DEVIATION, not of the offset -- a float32 mean
// alone leaves an error of |mean| 2^-24 in every sample (rows whose offset is >~ 300 x their signal left the 1e-5 bar).
struct Mean2 { float hi, lo; };
__device__ __forceinline__ Mean2 mean2_load(const double* __restrict__ mean, int row) {
    Mean2 m{0.f, 0.f};
    if (mean) {
        const double d = mean[row];
        m.hi = (float)d;
        m.lo = (float)(d - (double)m.hi);
    }
    return m;
}
__device__ __forceinline__ float demean(float v, Mean2 m) { return (v - m.hi) - m.lo; }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
}  // namespace d4w
