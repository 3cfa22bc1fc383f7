// Synthetic neighbours for scripts/probe/stream_race2.py (LOAD=M / V / L): what kind of co-resident work makes the overlap-save
// FFT kernels go wrong -- matrix instructions alone, vector FMAs alone, or LDS traffic alone?  Persistent grids (3 workgroups of
// 256 threads per CU), no memory traffic beyond one store at the end.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/probe/burners.hip -o scripts/probe/libburners.so
#include <hip/hip_runtime.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void burn_mfma(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// the same with ~150 live vector registers per lane (32 accumulators + operands), as the STFT kernel has
__global__ __launch_bounds__(256, 3) void burn_mfma_fat(float* out, int iters) {
    h8 a[4], b[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) { a[k][i] = (_Float16)(0.001f * (threadIdx.x + i + k)); b[k][i] = (_Float16)(0.002f * ((threadIdx.x ^ i) + k)); }
    f4 c[32];
    for (int k = 0; k < 32; ++k) c[k] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k & 3], b[(k >> 2) & 3], c[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += c[k][k & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void burn_valu(float* out, int iters) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 1e-9f * v[(i + 1) & 15]);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void burn_lds(float* out, int iters, int words) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int i = (threadIdx.x * 4 + it * 1024) % (words - 4);
        const float4 q = *reinterpret_cast<const float4*>(lds + (i & ~3));
        s += q.x + q.w;
        lds[(i + 2048) % words] = s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int burn(int kind, void* stream, float* out, int grid, int iters) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(burn_mfma, dim3(grid), dim3(256), 0, st, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(burn_valu, dim3(grid), dim3(256), 0, st, out, iters);
    else if (kind == 3) hipLaunchKernelGGL(burn_mfma_fat, dim3(grid), dim3(256), 42608, st, out, iters / 8);     // (+ the STFT's LDS footprint, unused)
    else hipLaunchKernelGGL(burn_lds, dim3(grid), dim3(256), 42608, st, out, iters, 42608 / 4);
    return (int)hipGetLastError();
}
