// Does a workgroup barrier hold while another kernel's workgroups (LDS-fed matrix instructions, their own barriers, 3 x 256
// threads per CU) are resident on the same CU?  Victim: 128 threads (two waves), in a loop: every lane writes (round, lane) to
// its LDS slot, barrier, reads the slot of its partner lane in the OTHER wave and checks the round, barrier.  A barrier that
// lets a wave through early shows as a stale round.  (Follow-up of the cross-stream hazard)
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/barrier_pair.hip -o scripts/probe/barrier_pair && scripts/probe/barrier_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256, 3) void aggressor(float* out, int iters, int halves) {
    extern __shared__ _Float16 lh[];
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.002f * ((threadIdx.x ^ i) & 63));
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if ((it & 15) == 0) {                                    // re-stage the operands now and then, barriers around it (as the STFT does per chunk)
            lds_barrier();
            for (int i = threadIdx.x; i < halves; i += blockDim.x) lh[i] = (_Float16)(0.001f * ((i + it) & 1023));
            lds_barrier();
        }
        const int base = ((lane & 15) * 8 + (lane >> 4) * 8 + it * 128) % (halves - 256);
        const h8 xh = *reinterpret_cast<const h8*>(lh + (base & ~7));
        const h8 xl = *reinterpret_cast<const h8*>(lh + ((base + 128) & ~7));
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xh, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xl, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, a, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <bool ASM_BARRIER>
__global__ __launch_bounds__(128, 2) void victim(int rounds, unsigned long long* bad) {
    extern __shared__ unsigned slot[];                          // [128] used, the rest only reserves the footprint
    unsigned long long mine = 0;
    const int tid = threadIdx.x, partner = (tid + 64) & 127;
    for (int r = 1; r <= rounds; ++r) {
        slot[tid] = ((unsigned)r << 8) | (unsigned)tid;
        if (ASM_BARRIER) lds_barrier(); else __syncthreads();
        const unsigned v = *(volatile unsigned*)&slot[partner];
        if (v != (((unsigned)r << 8) | (unsigned)partner)) ++mine;
        if (ASM_BARRIER) lds_barrier(); else __syncthreads();
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 40;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    float* out;
    unsigned long long* bad;
    CHECK(hipMalloc(&out, (size_t)ncu * 3 * 256 * 4));
    CHECK(hipMalloc(&bad, 16));
    CHECK(hipMemset(bad, 0, 16));
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(aggressor, dim3(ncu * 3), dim3(256), 42608, sa, out, 12000, 42608 / 2);
        hipLaunchKernelGGL(victim<true>, dim3(22040), dim3(128), 38912, sb, 300, bad);
        hipLaunchKernelGGL(victim<false>, dim3(22040), dim3(128), 38912, sb, 300, bad + 1);
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2];
    CHECK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("{\"launch pairs\": %d, \"barrier rounds checked per form\": %.3g, \"stale partner reads, s_waitcnt lgkmcnt(0) + s_barrier\": %llu, \"stale partner reads, __syncthreads\": %llu}\n",
           reps, (double)reps * 22040 * 128 * 300, h[0], h[1]);
    return 0;
}
