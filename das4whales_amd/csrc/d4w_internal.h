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

// dynamic LDS declaration usable by both hipcc and the emulator
#ifdef D4W_EMU
#define D4W_DYN_LDS(name) unsigned char* name = hipemu::dyn_smem
#else
#define D4W_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// ---------------------------------------------------------------------------------------------
// complex arithmetic on float2 (x = re, y = im)
// ---------------------------------------------------------------------------------------------
namespace d4w {
__device__ __forceinline__ float2 c_make(float re, float im) { return make_float2(re, im); }
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_scale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 c_conj(float2 a) { return make_float2(a.x, -a.y); }
// a * b
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
// multiply by -i : (x, y) -> (y, -x)
__device__ __forceinline__ float2 c_mul_mi(float2 a) { return make_float2(a.y, -a.x); }
// multiply by +i : (x, y) -> (-y, x)
__device__ __forceinline__ float2 c_mul_pi(float2 a) { return make_float2(-a.y, a.x); }

// 24-bit integer multiply (full-rate v_mul_i32_i24; v_mul_lo_u32 is quarter rate) and fast reciprocal
#ifdef D4W_EMU
__device__ __forceinline__ int d4w_mul24(int a, int b) { return a * b; }
__device__ __forceinline__ float d4w_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ int d4w_mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ float d4w_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
}  // namespace d4w
