// TEST INFRASTRUCTURE ONLY.
//
// A tiny single-process emulator for the subset of HIP the das4whales_amd kernels use, so that
// the *logic* of the HIP sources (index maps, LDS staging, barriers, wave shuffles, plan tables)
// can be debugged in the build container, which has no GPU.  The kernel sources in
// das4whales_amd/csrc are compiled unchanged with `g++ -x c++ -DD4W_EMU -include hip_emu.h`
// into tests/emu/_build/libd4w_emu.so and driven through the same C ABI with host pointers
// (tests/test_emu_*.py).  It is NOT a product path: the Python package never loads the emu
// library, nothing is timed on it, and the GPU parity tests always go through the gfx950 build.
//
// Model: one OS thread; each HIP thread of a workgroup is a ucontext fiber.  __syncthreads()
// blocks a fiber until every live fiber of the block has arrived; wave-level operations
// (__shfl*) block until every live fiber of the 64-lane wave has arrived; __ballot until every live fiber of the wave is at
// a ballot or parked at a barrier (divergent code: lanes that do not take part read as 0, like lanes masked out of EXEC).
// Workgroups run one after another (HIP promises no inter-block ordering anyway).
#pragma once
#ifndef D4W_EMU
#error "hip_emu.h is only for -DD4W_EMU builds"
#endif

#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

// ------------------------------------------------------------------------------------------
// qualifiers
// ------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_KERNEL_NAME(...) __VA_ARGS__

// ------------------------------------------------------------------------------------------
// vector types
// ------------------------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// ------------------------------------------------------------------------------------------
// runtime API stubs (device memory == host memory)
// ------------------------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;
    if (posix_memalign(p, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    memset(*p, 0xFF, n);  // NaN pattern: catches reads of never-written device memory
    return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) {
    for (size_t i = 0; i < count; ++i) static_cast<int*>(d)[i] = v;
    return hipSuccess;
}
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
// launch attributes, device queries and cross-stream ordering: accepted and ignored (one host thread runs everything in order)
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
struct hipFuncAttributes { int numRegs = 0; size_t sharedSizeBytes = 0; };
template <typename F> static inline hipError_t hipFuncGetAttributes(hipFuncAttributes*, F) { return hipErrorInvalidValue; }
struct hipDeviceProp_t { int multiProcessorCount = 3; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// a value every lane of the wave holds alike
#define __builtin_amdgcn_readfirstlane(v) (v)

// ------------------------------------------------------------------------------------------
// fiber scheduler
// ------------------------------------------------------------------------------------------
namespace hipemu {
enum { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3, WAIT_BALLOT = 4 };
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = DONE;
    dim3 tid;
};
struct uint3_ { unsigned x, y, z; };

inline uint3_ g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
inline ucontext_t g_sched;
inline Fiber* g_cur = nullptr;
inline std::vector<Fiber> g_fibers;
inline std::function<void()> g_body;
inline unsigned char* dyn_smem = nullptr;
inline uint64_t g_exch[2][1024];     // wave exchange slots (two generations)
inline int g_lin_tid = 0;
inline unsigned g_shfl_gen[16];      // per-wave shuffle generation
constexpr size_t STACK = 256 * 1024;

inline void fiber_entry() {
    g_body();
    g_cur->state = DONE;
    swapcontext(&g_cur->ctx, &g_sched);
}
inline void yield(int st) {
    Fiber* f = g_cur;
    f->state = st;
    swapcontext(&f->ctx, &g_sched);
}
inline void run_block(unsigned nthreads, dim3 block) {
    if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        if (!f.stack) f.stack = (char*)malloc(STACK);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.state = RUNNABLE;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    }
    for (auto& g : g_shfl_gen) g = 0;
    unsigned ndone = 0;
    const unsigned nwaves = (nthreads + 63) / 64;
    while (ndone < nthreads) {
        bool ran = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.state != RUNNABLE) continue;
            g_cur = &f;
            g_lin_tid = (int)t;
            g_threadIdx = {f.tid.x, f.tid.y, f.tid.z};
            swapcontext(&g_sched, &f.ctx);
            ran = true;
            if (f.state == DONE) ++ndone;
        }
        bool released = false;
        for (unsigned w = 0; w < nwaves; ++w) {
            unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
            unsigned live = 0, waiting = 0;
            for (unsigned t = lo; t < hi; ++t) {
                if (g_fibers[t].state != DONE) ++live;
                if (g_fibers[t].state == WAIT_WAVE) ++waiting;
            }
            if (waiting && waiting == live) {
                for (unsigned t = lo; t < hi; ++t)
                    if (g_fibers[t].state == WAIT_WAVE) g_fibers[t].state = RUNNABLE;
                released = true;
                continue;
            }
            // __ballot in divergent code (lane groups of a wave in different loop trips, lanes that have left for the next
            // barrier): the lanes that are not at a ballot do not take part and read as 0, like lanes masked out of EXEC.
            // (Shuffles keep the strict rule above: every live lane of the wave must arrive.)
            unsigned balloting = 0, parked = 0;
            for (unsigned t = lo; t < hi; ++t) {
                if (g_fibers[t].state == WAIT_BALLOT) ++balloting;
                if (g_fibers[t].state == WAIT_BLOCK) ++parked;
            }
            if (balloting && balloting + parked == live) {
                const unsigned gen = g_shfl_gen[w] & 1u;
                for (unsigned t = lo; t < hi; ++t) {
                    if (g_fibers[t].state == WAIT_BALLOT) g_fibers[t].state = RUNNABLE;
                    else g_exch[gen][t] = 0;
                }
                released = true;
            }
        }
        if (!released) {
            unsigned live = 0, waiting = 0;
            for (unsigned t = 0; t < nthreads; ++t) {
                if (g_fibers[t].state != DONE) ++live;
                if (g_fibers[t].state == WAIT_BLOCK) ++waiting;
            }
            if (waiting && waiting == live) {
                for (unsigned t = 0; t < nthreads; ++t)
                    if (g_fibers[t].state == WAIT_BLOCK) g_fibers[t].state = RUNNABLE;
                released = true;
            }
        }
        if (!ran && !released && ndone < nthreads) {
            fprintf(stderr, "hipemu: barrier deadlock (divergent __syncthreads / shuffle)\n");
            abort();
        }
    }
}

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) { fprintf(stderr, "hipemu: bad block size %u\n", nthreads); abort(); }
    if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: dynamic LDS %zu > 160 KiB\n", shmem); abort(); }
    unsigned char* smem = (unsigned char*)aligned_alloc(256, ((shmem + 255) / 256 + 1) * 256);
    dyn_smem = smem;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    g_body = [=]() { kernel(args...); };
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = {bx, by, bz};
                memset(smem, 0xFF, shmem);  // LDS is uninitialised on hardware: poison it
                run_block(nthreads, block);
            }
    free(smem);
    dyn_smem = nullptr;
}

template <typename T>
inline T wave_exchange(T v, int src_lane_rel /* lane within wave to read from */) {
    static_assert(sizeof(T) <= 8, "shuffle payload > 8 bytes");
    int wave = g_lin_tid / 64;
    int base = wave * 64;
    unsigned gen = g_shfl_gen[wave] & 1u;  // same for every lane of the wave at this point
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    g_exch[gen][g_lin_tid] = raw;
    int me = g_lin_tid;
    yield(WAIT_WAVE);
    // first lane to resume flips the generation for the next shuffle
    if ((g_shfl_gen[wave] & 1u) == gen) g_shfl_gen[wave]++;
    int src = base + src_lane_rel;
    unsigned nthreads = g_blockDim.x * g_blockDim.y * g_blockDim.z;
    T out = v;
    if (src_lane_rel >= 0 && src_lane_rel < 64 && (unsigned)src < nthreads) memcpy(&out, &g_exch[gen][src], sizeof(T));
    (void)me;
    return out;
}
// predicate of every lane of the wave as a 64-bit mask (lanes parked at a barrier or finished read as 0)
inline unsigned long long wave_ballot(int pred) {
    int wave = g_lin_tid / 64;
    int base = wave * 64;
    unsigned gen = g_shfl_gen[wave] & 1u;
    g_exch[gen][g_lin_tid] = pred ? 1u : 0u;
    yield(WAIT_BALLOT);
    if ((g_shfl_gen[wave] & 1u) == gen) g_shfl_gen[wave]++;
    unsigned nthreads = g_blockDim.x * g_blockDim.y * g_blockDim.z;
    unsigned long long out = 0ull;
    for (int l = 0; l < 64; ++l)
        if ((unsigned)(base + l) < nthreads && g_exch[gen][base + l]) out |= 1ull << l;
    return out;
}
}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
#define warpSize 64

static inline void __syncthreads() { hipemu::yield(hipemu::WAIT_BLOCK); }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::g_lin_tid & 63;
    (void)width;
    return hipemu::wave_exchange(v, lane ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::g_lin_tid & 63;
    int src = lane + (int)d;
    if ((src / width) != (lane / width)) src = lane;
    return hipemu::wave_exchange(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::g_lin_tid & 63;
    int src = lane - (int)d;
    if (src < 0 || (src / width) != (lane / width)) src = lane;
    return hipemu::wave_exchange(v, src);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::g_lin_tid & 63;
    return hipemu::wave_exchange(v, (lane / width) * width + (src % width));
}

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned long long __ballot(int pred) { return hipemu::wave_ballot(pred); }
// (used in wave-uniform code of full waves only: every lane of the wave votes)
static inline int __any(int pred) { return hipemu::wave_ballot(pred) != 0ull; }
static inline int __all(int pred) { return hipemu::wave_ballot(!pred) == 0ull; }

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline void sincospif(float x, float* s, float* c) { *s = (float)sin(M_PI * (double)x); *c = (float)cos(M_PI * (double)x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
using std::max;
using std::min;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

// ------------------------------------------------------------------------------------------
// binary16 values and the 16x16x32 f16 matrix instruction (csrc/xcorr_mm.hip)
// ------------------------------------------------------------------------------------------
namespace hipemu {
// float -> binary16, round to nearest even, subnormals kept (v_cvt_f16_f32)
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((x > 0x7F800000u) ? 0x200u : 0u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);              // rounds to infinity
    if (x < 0x33000001u) return (uint16_t)sign;                           // below half the smallest subnormal
    const int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                    // subnormal result
    else { shift = 13; base = (uint32_t)(e + 15) << 10; man &= 0x7FFFFFu; }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;                     // a carry into the exponent is the right answer
    return (uint16_t)(sign | (base + q));
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3FFu;
    float r;
    if (e == 0) r = ldexpf((float)m, -24);
    else if (e == 31) r = m ? NAN : INFINITY;
    else r = ldexpf((float)(m | 0x400u), e - 25);
    uint32_t x;
    memcpy(&x, &r, 4);
    x |= sign;
    memcpy(&r, &x, 4);
    return r;
}
inline uint16_t g_mfma_a[1024][8], g_mfma_b[1024][8];
// D = A B + C of v_mfma_f32_16x16x32_f16: lane l holds A[l & 15][8 (l >> 4) + j], B[8 (l >> 4) + j][l & 15], j < 8,
// and C / D[4 (l >> 4) + r][l & 15], r < 4.  Products are exact in float; the sum is kept in double and rounded once.
inline void mfma_f32_16x16x32_f16(const uint16_t (&a)[8], const uint16_t (&b)[8], float (&c)[4]) {
    const int me = g_lin_tid, base = (me / 64) * 64, lane = me & 63;
    memcpy(g_mfma_a[me], a, 16);
    memcpy(g_mfma_b[me], b, 16);
    yield(WAIT_WAVE);
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        double s = 0.0;
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j)
                s += (double)f16_to_f32(g_mfma_a[base + row + 16 * g][j]) * (double)f16_to_f32(g_mfma_b[base + col + 16 * g][j]);
        c[r] = (float)((double)c[r] + s);
    }
    yield(WAIT_WAVE);                                                     // nobody overwrites the slots before all have read
}
}  // namespace hipemu
