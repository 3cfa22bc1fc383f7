// Two questions about the memory system of this part, answered with plain streaming kernels:
//  (1) what does a read + write stream reach (the guide quotes 6.29 TB/s for a float4
//      copy, round 1 measured 4.8 TB/s)?  Sweep: out-of-place / in-place, grid, bytes in flight per
//      thread, non-temporal loads / stores, buffer size.
//  (2) does the 256 MiB Infinity Cache keep a kernel's OUTPUT for the next kernel (write-allocate,
//      write-back)?  K in-place read-modify-write launches run back to back on one slab of S bytes,
//      slab after slab over the whole 9.6 GB block, against K full passes.  If dirty lines stay in
//      the cache the slab-ordered form moves 1 read + 1 write of HBM traffic instead of K of each.
// hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float vec4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_k(const vec4* __restrict__ src, vec4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        vec4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(r[u], dst + i + u * stride);
            else dst[i + u * stride] = r[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// contiguous chunk per workgroup (each block walks its own contiguous range: DRAM page locality)
template <int U, bool NTS>
__global__ __launch_bounds__(256) void copy_chunk(const vec4* __restrict__ src, vec4* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = (b + per < n) ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        vec4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(r[u], dst + i + u * 256);
            else dst[i + u * 256] = r[u];
        }
    }
    for (; i < e; i += 256) dst[i] = src[i];
}

template <int U>
__global__ __launch_bounds__(256) void rmw_k(vec4* __restrict__ x, size_t n, float a) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        vec4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) x[i + u * stride] = r[u] * a;
    }
    for (; i < n; i += stride) x[i] = x[i] * a;
}

__global__ __launch_bounds__(256) void read_k(const vec4* __restrict__ x, size_t n, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    vec4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += x[i];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void write_k(vec4* __restrict__ x, size_t n, float a) {
    const size_t stride = (size_t)gridDim.x * 256;
    const vec4 v = {a, a, a, a};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = v;
}

static hipEvent_t e0, e1;
template <class F>
static float timeit(F f, int reps = 3) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const size_t total = (size_t)20000 * 120000 * 4;      // 9.6 GB
    vec4 *a, *b;
    float* flag;
    CK(hipMalloc(&a, total));
    CK(hipMalloc(&b, total));
    CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 0, total));
    CK(hipMemset(b, 0, total));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t n = total / 16;

    printf("== (1) read + write streams, 9.6 GB -> 9.6 GB (TB/s counts read + written bytes)\n");
#define RUNCOPY(U, NTL, NTS, GRID)                                                                         \
    {                                                                                                        \
        float ms = timeit([&] { hipLaunchKernelGGL((copy_k<U, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, a, b, n); }); \
        printf("copy  U=%d ntload=%d ntstore=%d grid=%6d : %7.3f ms  %5.2f TB/s\n", U, NTL, NTS, GRID, ms, 2.0 * total / ms / 1e9); \
    }
    for (int grid : {1024, 2048, 4096, 16384, 65536}) {
        RUNCOPY(1, false, false, grid);
        RUNCOPY(4, false, false, grid);
    }
    RUNCOPY(2, false, false, 4096);
    RUNCOPY(8, false, false, 4096);
    RUNCOPY(4, true, false, 4096);
    RUNCOPY(4, false, true, 4096);
    RUNCOPY(4, true, true, 4096);
    RUNCOPY(8, true, true, 2048);
    RUNCOPY(4, true, true, 16384);
    {
        for (int grid : {256, 512, 1024, 2048, 8192}) {
            float ms = timeit([&] { hipLaunchKernelGGL((copy_chunk<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("chunk U=4 ntstore=0 grid=%6d : %7.3f ms  %5.2f TB/s\n", grid, ms, 2.0 * total / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("chunk U=4 ntstore=1 grid=%6d : %7.3f ms  %5.2f TB/s\n", grid, ms, 2.0 * total / ms / 1e9);
        }
    }
    {
        float ms = timeit([&] { hipLaunchKernelGGL((rmw_k<4>), dim3(4096), dim3(256), 0, 0, a, n, 1.0f); });
        printf("in-place rmw U=4 grid=4096       : %7.3f ms  %5.2f TB/s\n", ms, 2.0 * total / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(read_k, dim3(4096), dim3(256), 0, 0, a, n, flag); });
        printf("read only                        : %7.3f ms  %5.2f TB/s\n", ms, 1.0 * total / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(write_k, dim3(4096), dim3(256), 0, 0, a, n, 0.f); });
        printf("write only                       : %7.3f ms  %5.2f TB/s\n", ms, 1.0 * total / ms / 1e9);
        CK(hipMemcpyAsync(b, a, total, hipMemcpyDeviceToDevice, 0));
        ms = timeit([&] { CK(hipMemcpyAsync(b, a, total, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D                    : %7.3f ms  %5.2f TB/s\n", ms, 2.0 * total / ms / 1e9);
    }
    for (size_t mb : {256, 1024, 4096}) {
        const size_t nn = mb * 1024 * 1024 / 16;
        float ms = timeit([&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL((copy_k<4, false, false>), dim3(4096), dim3(256), 0, 0, a, b, nn); });
        printf("copy U=4 of %5zu MiB x8           : %7.3f ms  %5.2f TB/s\n", mb, ms, 8 * 2.0 * nn * 16 / ms / 1e9);
    }

    printf("== (2) K in-place read-modify-write launches per slab, slab after slab over 9.6 GB\n");
    printf("       (effective TB/s = K x 2 x 9.6 GB / time; K full passes reach the copy rate)\n");
    for (int K : {2, 3}) {
        for (size_t mb : {8, 16, 32, 64, 96, 128, 192, 256, 384, 1024}) {
            const size_t sn = mb * 1024 * 1024 / 16;
            const int grid = (int)((sn + 256 * 4 - 1) / (256 * 4) < 4096 ? (sn + 256 * 4 - 1) / (256 * 4) : 4096);
            float ms = timeit([&] {
                for (size_t off = 0; off < n; off += sn) {
                    const size_t m = (off + sn <= n) ? sn : n - off;
                    for (int k = 0; k < K; ++k) hipLaunchKernelGGL((rmw_k<4>), dim3(grid), dim3(256), 0, 0, a + off, m, 1.0f);
                }
            });
            printf("K=%d slab %5zu MiB (%4zu slabs): %7.3f ms  eff %5.2f TB/s\n", K, mb, (n + sn - 1) / sn, ms, K * 2.0 * total / ms / 1e9);
        }
    }
    printf("== (3) producer writes slab (from a second array), consumer reads it\n");
    for (size_t mb : {32, 64, 128, 256, 1024}) {
        const size_t sn = mb * 1024 * 1024 / 16;
        const int grid = (int)((sn + 256 * 4 - 1) / (256 * 4) < 4096 ? (sn + 256 * 4 - 1) / (256 * 4) : 4096);
        float ms = timeit([&] {
            for (size_t off = 0; off < n; off += sn) {
                const size_t m = (off + sn <= n) ? sn : n - off;
                hipLaunchKernelGGL((copy_k<4, false, false>), dim3(grid), dim3(256), 0, 0, b + off, a + off, m);
                hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a + off, m, flag);
            }
        });
        printf("copy->read slab %5zu MiB: %7.3f ms  (copy alone ~ 2 x 9.6 GB, + read 9.6 GB) eff %5.2f TB/s\n", mb, ms, 3.0 * total / ms / 1e9);
    }
    printf("== (4) two streams: rmw #1 on slab s+1 overlaps rmw #2 on slab s\n");
    {
        hipStream_t s1, s2;
        CK(hipStreamCreate(&s1));
        CK(hipStreamCreate(&s2));
        for (size_t mb : {32, 64, 128}) {
            const size_t sn = mb * 1024 * 1024 / 16;
            const int grid = 2048;
            const int nslab = (int)((n + sn - 1) / sn);
            std::vector<hipEvent_t> ev(nslab);
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s1));
                for (int s = 0; s < nslab; ++s) {
                    const size_t off = (size_t)s * sn, m = (off + sn <= n) ? sn : n - off;
                    hipLaunchKernelGGL((rmw_k<4>), dim3(grid), dim3(256), 0, s1, a + off, m, 1.0f);
                    CK(hipEventRecord(ev[s], s1));
                    CK(hipStreamWaitEvent(s2, ev[s], 0));
                    hipLaunchKernelGGL((rmw_k<4>), dim3(grid), dim3(256), 0, s2, a + off, m, 1.0f);
                }
                CK(hipEventRecord(ev[0], s2));
                CK(hipStreamWaitEvent(s1, ev[0], 0));
                CK(hipEventRecord(e1, s1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("2 streams K=2 slab %5zu MiB: %7.3f ms  eff %5.2f TB/s\n", mb, best, 2 * 2.0 * total / best / 1e9);
        }
    }
    return 0;
}
