// Split-binary16 operands for the matrix cores (gfx950 v_mfma_f32_16x16x32_f16), shared by the kernels that run a
// float32 contraction as matrix products: xcorr_mm.hip (matched filter as a banded-Toeplitz product) and stft_mm.hip
// (short-time Fourier transform as frames x DFT-rows).  A float32 value v is carried as hi + lo / 2048 (hi = rn16(v),
// lo = rn16((v - hi) 2048): 22-23 significant bits); a product keeps hi hi + (hi lo + lo hi) / 2048, each exact in the
// instruction's float32 accumulator.  On the CPU emulator (tests/emu/hip_emu.h) the same names map to software binary16.
#pragma once
#include "fft_radix.h"

namespace d4w {

constexpr float kMmLoScale = 2048.f, kMmLoInv = 1.0f / 2048.f;

#ifdef D4W_EMU
typedef uint16_t mm_half;
struct alignas(16) mm_h8 { uint16_t v[8]; };
struct mm_f4 { float v[4]; };
__device__ __forceinline__ mm_half mm_to_half(float x) { return hipemu::f32_to_f16(x); }
__device__ __forceinline__ float mm_to_float(mm_half h) { return hipemu::f16_to_f32(h); }
__device__ __forceinline__ mm_f4 mm_zero() { return mm_f4{{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ mm_f4 mm_mfma(const mm_h8& a, const mm_h8& b, mm_f4 c) {
    hipemu::mfma_f32_16x16x32_f16(a.v, b.v, c.v);
    return c;
}
__device__ __forceinline__ float mm_get(const mm_f4& c, int r) { return c.v[r]; }
__device__ __forceinline__ void mm_set(mm_h8& a, int j, mm_half h) { a.v[j] = h; }
__device__ __forceinline__ int mm_uniform(int v) { return v; }
__device__ __forceinline__ void mm_sched_fence() {}
__device__ __forceinline__ void mm_store4(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
__device__ __forceinline__ void mm_store1(float* p, float v) { *p = v; }
__device__ __forceinline__ float4 mm_load4_stream(const float4* p) { return *p; }
__device__ __forceinline__ void mm_put4(mm_half* p, const mm_half (&h)[4]) { p[0] = h[0]; p[1] = h[1]; p[2] = h[2]; p[3] = h[3]; }
__device__ __forceinline__ float mm_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ void mm_wave_sync() { (void)__shfl_xor(0, 1); }     // every lane of the wave arrives before any goes on
__device__ __forceinline__ float mm_clamp_half(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
static inline int mm_num_cus() { return 2; }
#else
typedef _Float16 mm_half;
typedef _Float16 mm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 mm_h4 __attribute__((ext_vector_type(4)));
typedef float mm_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mm_half mm_to_half(float x) { return (_Float16)x; }
__device__ __forceinline__ float mm_to_float(mm_half h) { return (float)h; }
__device__ __forceinline__ mm_f4 mm_zero() { mm_f4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ mm_f4 mm_mfma(mm_h8 a, mm_h8 b, mm_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float mm_get(const mm_f4& c, int r) { return c[r]; }
__device__ __forceinline__ void mm_set(mm_h8& a, int j, mm_half h) { a[j] = h; }
__device__ __forceinline__ int mm_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void mm_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void mm_store4(float* p, float a, float b, float c, float d) {
    mm_f4 t = {a, b, c, d};
    __builtin_nontemporal_store(t, reinterpret_cast<mm_f4*>(p));
}
__device__ __forceinline__ void mm_store1(float* p, float v) { __builtin_nontemporal_store(v, p); }     // write-once outputs: streaming
__device__ __forceinline__ float4 mm_load4_stream(const float4* p) {                                   // read-once inputs
    const mm_f4 v = __builtin_nontemporal_load(reinterpret_cast<const mm_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void mm_put4(mm_half* p, const mm_half (&h)[4]) {                            // one 8-byte LDS store
    mm_h4 v = {h[0], h[1], h[2], h[3]};
    *reinterpret_cast<mm_h4*>(p) = v;
}
// v_sqrt_f32 (1 ulp) without the denormal rescaling and refinement sqrtf() wraps around it
__device__ __forceinline__ float mm_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
// orders a wave's LDS stores before its own later LDS loads (and the other way round): the LDS serves a wave's accesses in
// order, this only keeps the compiler from moving them across
__device__ __forceinline__ void mm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// into binary16's finite range (NaN stays NaN)
__device__ __forceinline__ float mm_clamp_half(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
// compute units of the current device, asked once per device (two runtime calls per launch before)
static inline int mm_num_cus() {
    static int cached[64] = {0};
    int devid = 0, v = 0;
    if (hipGetDevice(&devid) != hipSuccess) return 256;
    if (devid >= 0 && devid < 64 && cached[devid] > 0) return cached[devid];
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, devid) == hipSuccess && v > 0) {
        if (devid >= 0 && devid < 64) cached[devid] = v;
        return v;
    }
    return 256;
}
#endif

// *p = max(*p, v) for floats with the two native integer atomics (no float max atomic on this part; no compare-and-swap
// loop): a non-negative v orders like its bits as a signed integer, a negative one like the REVERSE of its bits as an
// unsigned integer, and a non-negative value always beats the bit pattern of a negative one in either view.  *p starts at -inf.
// A NaN enters as the quiet-NaN pattern 0x7FC00000 and STAYS (np.max propagates NaN): as a signed integer it beats every
// non-negative float and every negative one's pattern; as an unsigned integer it lies below every negative float's pattern,
// so the later atomicMin of a negative value leaves it alone.
__device__ __forceinline__ void mm_atomic_fmax(float* p, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v + 0.f));      // -0 enters as +0
    else if (v == v) atomicMin(reinterpret_cast<unsigned*>(p), __float_as_uint(v));
    else atomicMax(reinterpret_cast<int*>(p), 0x7FC00000);
}

// v = hi + lo / 2048 to 22-23 significant bits.  v is pinned to ONE float32 value first: left alone, hipcc contracts the
// caller's multiply into v_fma_mixlo_f16 for the residual (hi rounded once from the exact product) while the stored hi comes
// from v_cvt_pk_f16_f32 of the rounded product -- the two differ by a binary16 ulp for one sample in ~10^4 (a 1e-4 error).
__device__ __forceinline__ void mm_split(float v, mm_half& hi, mm_half& lo) {
#ifndef D4W_EMU
    asm volatile("" : "+v"(v));
#endif
    hi = mm_to_half(v);
    lo = mm_to_half((v - mm_to_float(hi)) * kMmLoScale);
}

// Four consecutive samples split and stored: ph[0..3] = hi, pl[0..3] = lo (8-byte LDS stores).  The same values as four
// mm_split calls, in 10 vector instructions instead of 20 (plus the waits the partial writes ask for) (round 6: the conversion phase of xcorr_mm_rows is a third of that
// kernel's time): v_cvt_pk_f16_f32 rounds two samples into one packed register, and v_fma_mixlo / mixhi_f16 form
// rn16(fma(hi, -2048, 2048 v)) = rn16((v - hi) 2048) straight from the binary16 half they read (both products are exact, so
// this is the residual of mm_split bit for bit) into the two halves of the packed lo register -- no float32 copy of hi, no
// separate subtract / multiply / convert / pack.  (s_nop: a partial register write must not be followed at once by a reader.)
__device__ __forceinline__ void mm_split_put4(const float (&v)[4], mm_half* ph, mm_half* pl) {
#ifdef D4W_EMU
    mm_half h[4], l[4];
    for (int e = 0; e < 4; ++e) mm_split(v[e], h[e], l[e]);
    mm_put4(ph, h);
    mm_put4(pl, l);
#else
    const float m = -kMmLoScale;
    auto pair = [&](float a, float b, unsigned& h, unsigned& l) {      // two samples per block: few registers live at once
        const float ka = a * kMmLoScale, kb = b * kMmLoScale;
        asm volatile(
            "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
            "s_nop 0\n\t"
            "v_fma_mixlo_f16 %1, %0, %6, %4 op_sel_hi:[1,0,0]\n\t"
            "s_nop 0\n\t"
            "v_fma_mixhi_f16 %1, %0, %6, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(h), "=&v"(l)
            : "v"(a), "v"(b), "v"(ka), "v"(kb), "s"(m));
    };
    unsigned h01, h23, l01, l23;
    pair(v[0], v[1], h01, l01);
    pair(v[2], v[3], h23, l23);
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    u2_t hv, lv;
    hv.x = h01; hv.y = h23; lv.x = l01; lv.y = l23;
    *reinterpret_cast<u2_t*>(ph) = hv;
    *reinterpret_cast<u2_t*>(pl) = lv;
#endif
}

// power of two >= a (a >= 0, finite): the scale that keeps a block of values inside [-1, 1] without rounding them
__device__ __forceinline__ void mm_pow2_scale(float a, float& up, float& down) {
    int e = 0;
    if (a > 0.f) (void)frexpf(a, &e);              // a = f 2^e, 0.5 <= f < 1
    e = min(max(e, -100), 100);
    up = ldexpf(1.0f, e);
    down = ldexpf(1.0f, -e);
}

}  // namespace d4w
