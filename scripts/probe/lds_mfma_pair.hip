// Minimal pair for the hazard of DESIGN.md section 1: does a kernel that feeds matrix instructions straight from LDS
// (ds_read_b128 -> v_mfma, 3 x 256 threads per CU, persistent) disturb the 16-byte LDS reads of another kernel's workgroups
// (128 threads, 38 KB, many short workgroups) on the same CUs, launched from another stream?
// victim: every lane writes a pattern with 16-byte stores, then reads it back `reads` times with 16-byte reads and compares.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/lds_mfma_pair.hip -o scripts/probe/lds_mfma_pair && scripts/probe/lds_mfma_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 3) void aggressor(float* out, int iters, int halves, int from_lds) {
    extern __shared__ _Float16 lh[];
    for (int i = threadIdx.x; i < halves; i += blockDim.x) lh[i] = (_Float16)(0.001f * (i & 1023));
    __syncthreads();
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.002f * ((threadIdx.x ^ i) & 63));
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const int base = ((lane & 15) * 8 + (lane >> 4) * 8 + it * 128) % (halves - 256);
        h8 xh = a, xl = a;
        if (from_lds) {
            xh = *reinterpret_cast<const h8*>(lh + (base & ~7));
            xl = *reinterpret_cast<const h8*>(lh + ((base + 128) & ~7));
        }
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xh, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xl, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, a, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xh, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xl, c5, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1];
}

__global__ __launch_bounds__(128, 2) void victim(int words4, int reads, unsigned long long* bad, unsigned* where) {
    extern __shared__ float4 lv[];
    const unsigned salt = blockIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < words4; i += blockDim.x) {
        const unsigned k = salt + 4u * i;
        lv[i] = make_float4(__uint_as_float(k & 0x3FFFFFFFu), __uint_as_float((k + 1) & 0x3FFFFFFFu), __uint_as_float((k + 2) & 0x3FFFFFFFu),
                            __uint_as_float((k + 3) & 0x3FFFFFFFu));
    }
    __syncthreads();
    unsigned long long mine = 0;
    for (int r = 0; r < reads; ++r) {
        const int i = (threadIdx.x * 9 + r * 131) % words4;                   // a swizzle-like spread over the tile
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v vv = *(const volatile f4v __attribute__((address_space(3)))*)(lv + i);      // one ds_read_b128 (C cast, as d4w_internal.h lds_read4)
        const float4 v = make_float4(vv.x, vv.y, vv.z, vv.w);
        const unsigned k = salt + 4u * i;
        const unsigned g[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        for (int q = 0; q < 4; ++q)
            if (g[q] != ((k + q) & 0x3FFFFFFFu)) { ++mine; atomicAdd(&where[q], 1u); }
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
    const int from_lds = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 40;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    float* out;
    unsigned long long* bad;
    unsigned* where;
    CHECK(hipMalloc(&out, (size_t)ncu * 3 * 256 * 4));
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&where, 16));
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipMemset(where, 0, 16));
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(aggressor, dim3(ncu * 3), dim3(256), 42608, sa, out, 12000, 42608 / 2, from_lds);
        hipLaunchKernelGGL(victim, dim3(22040), dim3(128), 38912, sb, 38912 / 16, 640, bad, where);
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h;
    unsigned w[4];
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(w, where, 16, hipMemcpyDeviceToHost));
    printf("{\"aggressor operands from LDS\": %d, \"launch pairs\": %d, \"victim 16-byte reads\": %.3g, \"wrong dwords\": %llu, \"by dword\": [%u, %u, %u, %u]}\n",
           from_lds, reps, (double)reps * 22040 * 128 * 640, h, w[0], w[1], w[2], w[3]);
    return 0;
}
