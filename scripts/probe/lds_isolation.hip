// Do workgroups of two kernels that run on two HIP streams keep their LDS to themselves?  (Follow-up of stream_race2.py: the
// overlap-save FFT kernels' LDS tile changed under them while the matrix-core STFT was resident on the same CU.)
// Kernel A: few long-lived workgroups (a persistent grid: 3 per CU, 256 threads, SA bytes of dynamic LDS); kernel B: many short
// workgroups (128 threads, SB bytes).  Each workgroup fills ALL of its dynamic LDS with a pattern derived from its own id, waits
// a little, reads it back and counts the words that changed; rounds of fill / check until the time is up.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/lds_isolation.hip -o scripts/probe/lds_isolation && scripts/probe/lds_isolation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void lds_owner(unsigned tag, int words, int rounds, int spin, unsigned long long* bad, unsigned* sample) {
    extern __shared__ unsigned lds[];
    const unsigned me = tag ^ (blockIdx.x * 2654435761u);
    unsigned long long mine = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned salt = me + 0x9E3779B9u * r;
        for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = salt ^ (unsigned)i * 40503u;
        __syncthreads();
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned v = lds[i], want = salt ^ (unsigned)i * 40503u;
            if (v != want) {
                ++mine;
                if (atomicAdd(&sample[0], 1u) < 8u) { const unsigned k = atomicAdd(&sample[1], 4u); sample[2 + k] = tag; sample[3 + k] = i; sample[4 + k] = v; sample[5 + k] = want; }
            }
        }
        __syncthreads();
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
    const int SA = argc > 1 ? atoi(argv[1]) : 42608, SB = argc > 2 ? atoi(argv[2]) : 38912;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    unsigned long long* bad;
    unsigned* sample;
    CHECK(hipMalloc(&bad, 2 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sample, 64 * sizeof(unsigned)));
    CHECK(hipMemset(bad, 0, 2 * sizeof(unsigned long long)));
    CHECK(hipMemset(sample, 0, 64 * sizeof(unsigned)));
    CHECK(hipFuncSetAttribute((const void*)lds_owner, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < reps; ++rep) {
        // A: 3 workgroups per CU, long rounds; B: 40 000 short workgroups
        hipLaunchKernelGGL(lds_owner, dim3(ncu * 3), dim3(256), SA, sa, 0xA0000000u + rep, SA / 4, 60, 40, bad, sample);
        hipLaunchKernelGGL(lds_owner, dim3(40000), dim3(128), SB, sb, 0xB0000000u + rep, SB / 4, 2, 4, bad + 1, sample);
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2];
    unsigned hs[64];
    CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hs, sample, sizeof(hs), hipMemcpyDeviceToHost));
    printf("{\"cus\": %d, \"lds_A\": %d, \"lds_B\": %d, \"launch_pairs\": %d, \"changed_words_A\": %llu, \"changed_words_B\": %llu", ncu, SA, SB, reps, h[0], h[1]);
    const unsigned n = hs[0] < 8 ? hs[0] : 8;
    for (unsigned k = 0; k < n; ++k) printf(", \"sample%u\": [\"%08x\", %u, \"%08x\", \"%08x\"]", k, hs[2 + 4 * k], hs[3 + 4 * k], hs[4 + 4 * k], hs[5 + 4 * k]);
    printf("}\n");
    return 0;
}
