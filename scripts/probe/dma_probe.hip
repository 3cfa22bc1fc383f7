// Copy-ceiling probe with LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without a VGPR round trip) against register
// copies, for the two questions round 3's review left open:
//   (1) f-k passes A / A': does a DMA read -> LDS -> 16-byte store stream beat the 4.8-5.1 TB/s of register copies,
//       flat and in pass A's strip geometry ([C1 rows, stride C2*M][N1 blocks, stride N2][128-byte strips])?
//   (2) matched filter: what does a 1 : 2 read : write stream (4 B read, 8 B written per sample) reach?
//       hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int C1 = 25, C2 = 800, N1 = 25, N2 = 2400, M = N1 * N2;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// every wave owns a ring of DEPTH 1-KiB slots; chunk = 64 lanes x 16 B; addr(chunk, lane) -> element offset in f4 units
template <int DEPTH, bool NT, class AddrF>
__device__ __forceinline__ void dma_stream(const f4* __restrict__ s, f4* __restrict__ d, size_t nchunks, AddrF addr) {
    extern __shared__ f4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f4* ring = lds + (size_t)wave * DEPTH * 64;
    const size_t w0 = (size_t)blockIdx.x * nw + wave, tw = (size_t)gridDim.x * nw;
    size_t c = w0;
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k) {
        const size_t cc = c + (size_t)k * tw;
        if (cc < nchunks) dma16(s + addr(cc, lane), ring + k * 64);
    }
    int slot = 0;
    for (; c < nchunks; c += tw) {
        const size_t ca = c + (size_t)(DEPTH - 1) * tw;
        int sa = slot + DEPTH - 1;
        if (sa >= DEPTH) sa -= DEPTH;
        if (ca < nchunks) dma16(s + addr(ca, lane), ring + sa * 64);
        else asm volatile("s_nop 0");
        // issued after this chunk's DMA: DEPTH - 1 DMAs and DEPTH - 1 stores (tail iterations issue fewer DMAs: waiting
        // for a smaller count than outstanding only waits longer)
        if (ca < nchunks) wait_vm<2 * DEPTH - 2>();
        else wait_vm<0>();
        // the read goes through asm: hipcc orders any LDS read it can see behind vmcnt(0) while an LDS-DMA is in flight
        f4 v;
        {
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ring + slot * 64 + lane);
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(la) : "memory");
        }
        if (NT) __builtin_nontemporal_store(v, d + addr(c, lane));
        else d[addr(c, lane)] = v;
        if (++slot == DEPTH) slot = 0;
    }
}

template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void dma_flat(const f4* s, f4* d, size_t n4) {
    dma_stream<DEPTH, NT>(s, d, n4 / 64, [](size_t c, int lane) { return c * 64 + lane; });
}

// pass A's strips: chunk = 8 strips of 128 B; strips of a tile (c2, block b): (c1, n1) -> ((c1 C2 + c2) M + n1 N2 + b TA) complex
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void dma_strips(const f4* s, f4* d) {
    constexpr int TA = 16, NBX = N2 / TA, SPT = C1 * N1;            // strips per tile 625
    const size_t nstrips = (size_t)C2 * NBX * SPT;
    dma_stream<DEPTH, NT>(s, d, nstrips / 8, [](size_t c, int lane) {
        const size_t st = c * 8 + (lane >> 3);
        const size_t tile = st / SPT;
        const int r = (int)(st - tile * SPT), c1 = r / N1, n1 = r - c1 * N1;
        const int c2 = (int)(tile / NBX), b = (int)(tile - (size_t)c2 * NBX);
        const size_t cplx = ((size_t)c1 * C2 + c2) * M + (size_t)n1 * N2 + (size_t)b * TA;
        return cplx / 2 + (lane & 7);                                // f4 = 2 complex
    });
}

// register streams for reference
__global__ void flat_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
template <bool NT>
__global__ void r1w2(const f4* __restrict__ s, f4* __restrict__ d0, f4* __restrict__ d1, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4 v = s[i];
        if (NT) { __builtin_nontemporal_store(v, d0 + i); __builtin_nontemporal_store(v + 1.f, d1 + i); }
        else { d0[i] = v; d1[i] = v + 1.f; }
    }
}
// the matched filter's chunk order: 16 KiB chunks, persistent workgroups, XCD-contiguous ranges, 4 loads then 8 stores per lane
template <bool NT>
__global__ __launch_bounds__(256) void r1w2_chunks(const f4* __restrict__ s, f4* __restrict__ d0, f4* __restrict__ d1, size_t nchunks) {
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, nq = gridDim.x >> 3;
    const size_t lo = nchunks * xcd / 8, hi = nchunks * (xcd + 1) / 8;
    f4 v[4], w[4];
    size_t c = lo + wq;
    if (c < hi)
        for (int q = 0; q < 4; ++q) v[q] = s[c * 1024 + q * 256 + threadIdx.x];
    for (; c < hi; c += nq) {
        for (int q = 0; q < 4; ++q) w[q] = v[q];
        if (c + nq < hi)
            for (int q = 0; q < 4; ++q) v[q] = s[(c + nq) * 1024 + q * 256 + threadIdx.x];
        for (int q = 0; q < 4; ++q) {
            const size_t o = c * 1024 + q * 256 + threadIdx.x;
            if (NT) { __builtin_nontemporal_store(w[q], d0 + o); __builtin_nontemporal_store(w[q] + 1.f, d1 + o); }
            else { d0[o] = w[q]; d1[o] = w[q] + 1.f; }
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F>
static float timeit(F f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / 5;
}

template <int DEPTH, bool NT>
static void run_dma(const f4* s, f4* d, size_t n4, int wgs_per_cu, bool strips) {
    const size_t lds = (size_t)4 * DEPTH * 1024;
    if (strips) {
        CK(hipFuncSetAttribute((const void*)dma_strips<DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float ms = timeit([&] { hipLaunchKernelGGL((dma_strips<DEPTH, NT>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, s, d); });
        printf("DMA strips 128 B  depth %2d  nt %d  %d WG/CU: %7.3f ms  %6.2f TB/s\n", DEPTH, (int)NT, wgs_per_cu, ms, 2.0 * n4 * 16 / ms * 1e-9);
    } else {
        CK(hipFuncSetAttribute((const void*)dma_flat<DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float ms = timeit([&] { hipLaunchKernelGGL((dma_flat<DEPTH, NT>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, s, d, n4); });
        printf("DMA flat          depth %2d  nt %d  %d WG/CU: %7.3f ms  %6.2f TB/s\n", DEPTH, (int)NT, wgs_per_cu, ms, 2.0 * n4 * 16 / ms * 1e-9);
    }
}

int main() {
    const size_t n4 = (size_t)C1 * C2 * M * 2 / 4;                  // 9.6 GB block as f4
    f4 *s, *d, *d2;
    CK(hipMalloc(&s, n4 * 16)); CK(hipMalloc(&d, n4 * 16)); CK(hipMalloc(&d2, n4 * 16));
    CK(hipMemset(s, 0, n4 * 16)); CK(hipMemset(d, 0, n4 * 16)); CK(hipMemset(d2, 0, n4 * 16));
    for (int g : {4, 8}) {
        float ms = timeit([&] { hipLaunchKernelGGL(flat_copy, dim3(256 * g), dim3(256), 0, 0, s, d, n4); });
        printf("register flat copy, grid %4d: %7.3f ms  %6.2f TB/s\n", 256 * g, ms, 2.0 * n4 * 16 / ms * 1e-9);
    }
    for (int w : {1, 2, 4}) {
        run_dma<4, false>(s, d, n4, w, false);
        run_dma<8, false>(s, d, n4, w, false);
        run_dma<16, false>(s, d, n4, w, false);
    }
    run_dma<8, true>(s, d, n4, 2, false);
    run_dma<16, true>(s, d, n4, 2, false);
    for (int w : {1, 2, 4}) {
        run_dma<8, false>(s, d, n4, w, true);
        run_dma<16, false>(s, d, n4, w, true);
    }
    run_dma<16, true>(s, d, n4, 2, true);
    // 1 : 2 read : write on a third of the block per array (the matched filter's 4 + 8 B per sample), total 28.8 GB
    {
        const size_t n = n4;
        for (int g : {4, 8, 16}) {
            float ms = timeit([&] { hipLaunchKernelGGL(r1w2<false>, dim3(256 * g), dim3(256), 0, 0, s, d, d2, n); });
            printf("register r1w2 plain stores, grid %4d: %7.3f ms  %6.2f TB/s\n", 256 * g, ms, 3.0 * n * 16 / ms * 1e-9);
            ms = timeit([&] { hipLaunchKernelGGL(r1w2<true>, dim3(256 * g), dim3(256), 0, 0, s, d, d2, n); });
            printf("register r1w2 nt stores,    grid %4d: %7.3f ms  %6.2f TB/s\n", 256 * g, ms, 3.0 * n * 16 / ms * 1e-9);
        }
        for (int w : {2, 3, 4, 8}) {
            float ms = timeit([&] { hipLaunchKernelGGL(r1w2_chunks<true>, dim3(256 * w), dim3(256), 0, 0, s, d, d2, n / 1024); });
            printf("r1w2 16-KiB chunks (matched-filter order) nt, %d WG/CU: %7.3f ms  %6.2f TB/s\n", w, ms, 3.0 * n * 16 / ms * 1e-9);
            ms = timeit([&] { hipLaunchKernelGGL(r1w2_chunks<false>, dim3(256 * w), dim3(256), 0, 0, s, d, d2, n / 1024); });
            printf("r1w2 16-KiB chunks (matched-filter order) plain, %d WG/CU: %7.3f ms  %6.2f TB/s\n", w, ms, 3.0 * n * 16 / ms * 1e-9);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
