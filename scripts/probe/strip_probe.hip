// Access-pattern probe for the f-k pass A / A' tiles: a pure copy that reads and writes the same
// [C1 rows, stride C2*M][N1 blocks, stride N2][TA complex] strips as fkf_passA_*, for several strip
// widths and vector widths.  Answers: what does the 128-byte-strip pattern cost against HBM copy
// speed, and would wider strips pay?      hipcc --offload-arch=gfx950 -O3 strip_probe.hip -o strip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int C1 = 25, C2 = 800, N1 = 25, N2 = 2400, M = N1 * N2;

// V = floats per thread access (2 = float2, 4 = float4); TA = complex elements per strip
template <int TA, int V, int THR>
__global__ __launch_bounds__(THR) void strip_copy(const float* __restrict__ src, float* __restrict__ dst, int ntiles) {
    constexpr int NBX = N2 / TA;
    constexpr int LPS = TA * 2 / V;                 // lanes per strip
    constexpr int ITEMS = N1 * LPS;                 // (n1, lane) items, each copies C1 values
    typedef float vec __attribute__((ext_vector_type(V)));
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int c2 = t / NBX, b0 = (t % NBX) * TA;
        for (int it = threadIdx.x; it < ITEMS; it += THR) {
            const int n1 = it / LPS, l = it % LPS;
            const size_t off = ((size_t)c2 * M + (size_t)n1 * N2 + b0) * 2 + (size_t)l * V;
            vec r[C1];
#pragma unroll
            for (int c1 = 0; c1 < C1; ++c1) r[c1] = *reinterpret_cast<const vec*>(src + off + (size_t)c1 * C2 * M * 2);
#pragma unroll
            for (int c1 = 0; c1 < C1; ++c1) *reinterpret_cast<vec*>(dst + off + (size_t)c1 * C2 * M * 2) = r[c1];
        }
    }
}

// pass-A thread mapping (one column per thread, THR = N1 * TA) but 16-byte lanes: the lanes of a column
// pair load float4 = both columns, the even lane for the even c1 rows, the odd lane for the odd rows
// (a DPP swap between the two lanes would then give each its own column for every row)
template <int TA, int THR>
__global__ __launch_bounds__(THR) void strip_copy_pair(const float* __restrict__ src, float* __restrict__ dst, int ntiles) {
    constexpr int NBX = N2 / TA;
    typedef float vec __attribute__((ext_vector_type(4)));
    const int hi = threadIdx.x / TA, tt = threadIdx.x % TA;
    if (hi >= N1) return;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int c2 = t / NBX, b0 = (t % NBX) * TA;
        const size_t off = ((size_t)c2 * M + (size_t)hi * N2 + b0 + (tt & ~1)) * 2;
        vec r[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int c1 = 2 * i + (tt & 1);
            if (c1 < C1) r[i] = *reinterpret_cast<const vec*>(src + off + (size_t)c1 * C2 * M * 2);
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int c1 = 2 * i + (tt & 1);
            if (c1 < C1) *reinterpret_cast<vec*>(dst + off + (size_t)c1 * C2 * M * 2) = r[i];
        }
    }
}

// the same lane-pair pattern with the pass kernels' structure: ONE workgroup per CU (dynamic LDS
// reservation), persistent, the loads of tile i+1 issued before the stores of tile i (two register sets)
template <int TA, int THR, int DEPTH>
__global__ __launch_bounds__(THR) void strip_copy_pipelined(const float* __restrict__ src, float* __restrict__ dst, int ntiles) {
    extern __shared__ float lds_dummy[];
    constexpr int NBX = N2 / TA;
    typedef float vec __attribute__((ext_vector_type(4)));
    const int hi = threadIdx.x / TA, tt = threadIdx.x % TA;
    if (threadIdx.x == 0) lds_dummy[0] = 0.f;
    if (hi >= N1) return;
    vec r[DEPTH][13];
    auto off_of = [&](int t) {
        const int c2 = t / NBX, b0 = (t % NBX) * TA;
        return ((size_t)c2 * M + (size_t)hi * N2 + b0 + (tt & ~1)) * 2;
    };
    auto issue = [&](vec (&q)[13], int t) {
        const size_t off = off_of(t);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int c1 = 2 * i + (tt & 1);
            if (c1 < C1) q[i] = *reinterpret_cast<const vec*>(src + off + (size_t)c1 * C2 * M * 2);
        }
    };
    auto store = [&](vec (&q)[13], int t) {
        const size_t off = off_of(t);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int c1 = 2 * i + (tt & 1);
            if (c1 < C1) *reinterpret_cast<vec*>(dst + off + (size_t)c1 * C2 * M * 2) = q[i];
        }
    };
    const int g = gridDim.x;
    int t = blockIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (t + d * g < ntiles) issue(r[d], t + d * g);
    for (; t < ntiles; t += DEPTH * g) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int tc = t + d * g;
            if (tc < ntiles) {
                if (tc + (DEPTH - 1) * g < ntiles) issue(r[(d + DEPTH - 1) % DEPTH], tc + (DEPTH - 1) * g);
                store(r[d], tc);
            }
        }
    }
}

__global__ void flat_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ void flat_read(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = s[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.f) d[0] = acc;          // never true: keeps the loads alive
}

__global__ void flat_write(float4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// read a fraction of the rows (every row whose index mod 100 < pct), write all rows: pass C' traffic mix
__global__ void part_read_full_write(const float4* __restrict__ s, float4* __restrict__ d, size_t rows, size_t row4, int pct) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const bool live = (int)(r % 100) < pct;
        for (size_t i = threadIdx.x; i < row4; i += blockDim.x)
            d[r * row4 + i] = live ? s[r * row4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
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
    return ms / 5;
}

template <int TA, int V, int THR>
static void run(const float* s, float* d, int grid_mult) {
    const int ntiles = C2 * (N2 / TA);
    const int grid = 256 * grid_mult;
    float ms = timeit([&] { hipLaunchKernelGGL((strip_copy<TA, V, THR>), dim3(grid), dim3(THR), 0, 0, s, d, ntiles); });
    const double bytes = 2.0 * C1 * C2 * (double)M * 8;
    printf("strip %4d B  vec %2d B  threads %4d  grid %5d : %7.3f ms  %6.2f TB/s\n", TA * 8, V * 4, THR, grid, ms, bytes / ms * 1e-9);
}

int main() {
    const size_t n = (size_t)C1 * C2 * M * 2;       // floats
    float *s, *d;
    CK(hipMalloc(&s, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMemset(s, 0, n * 4)); CK(hipMemset(d, 0, n * 4));
    float ms = timeit([&] { hipLaunchKernelGGL(flat_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4*)s, (float4*)d, n / 4); });
    printf("flat float4 copy: %7.3f ms  %6.2f TB/s\n", ms, 2.0 * n * 4 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(flat_read, dim3(256 * 8), dim3(256), 0, 0, (const float4*)s, (float4*)d, n / 4); });
    printf("flat float4 read : %7.3f ms  %6.2f TB/s\n", ms, n * 4.0 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(flat_write, dim3(256 * 8), dim3(256), 0, 0, (float4*)d, n / 4); });
    printf("flat float4 write: %7.3f ms  %6.2f TB/s\n", ms, n * 4.0 / ms * 1e-9);
    for (int pct : {27, 50, 100}) {
        const size_t rows = (size_t)C1 * C2, row4 = (size_t)M * 2 / 4;
        ms = timeit([&] { hipLaunchKernelGGL(part_read_full_write, dim3(256 * 8), dim3(256), 0, 0, (const float4*)s, (float4*)d, rows, row4, pct); });
        printf("read %3d %% of rows, write all: %7.3f ms  %6.2f TB/s moved\n", pct, ms, n * 4.0 * (1.0 + pct / 100.0) / ms * 1e-9);
    }
    for (int gm : {1, 2, 4}) {
        const int ntiles = C2 * (N2 / 16);
        ms = timeit([&] { hipLaunchKernelGGL((strip_copy_pair<16, 400>), dim3(256 * gm), dim3(400), 0, 0, s, d, ntiles); });
        printf("strip 128 B, lane pairs share float4 rows, threads 400 grid %4d: %7.3f ms  %6.2f TB/s\n", 256 * gm, ms, 2.0 * n * 4 / ms * 1e-9);
    }
    {
        const int ntiles = C2 * (N2 / 16);
        for (int lds : {0, 86 * 1024, 50 * 1024}) {
            hipFuncSetAttribute((const void*)strip_copy_pipelined<16, 400, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            hipFuncSetAttribute((const void*)strip_copy_pipelined<16, 400, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            const int grid = lds > 60 * 1024 ? 256 : (lds ? 768 : 1024);
            ms = timeit([&] { hipLaunchKernelGGL((strip_copy_pipelined<16, 400, 2>), dim3(grid), dim3(400), lds, 0, s, d, ntiles); });
            printf("pipelined pair copy depth 2, LDS %3d KiB, grid %4d: %7.3f ms  %6.2f TB/s\n", lds / 1024, grid, ms, 2.0 * n * 4 / ms * 1e-9);
            ms = timeit([&] { hipLaunchKernelGGL((strip_copy_pipelined<16, 400, 3>), dim3(grid), dim3(400), lds, 0, s, d, ntiles); });
            printf("pipelined pair copy depth 3, LDS %3d KiB, grid %4d: %7.3f ms  %6.2f TB/s\n", lds / 1024, grid, ms, 2.0 * n * 4 / ms * 1e-9);
        }
    }
    for (int gm : {1}) {
        run<16, 2, 400>(s, d, gm);
        run<16, 4, 256>(s, d, gm);
        run<32, 2, 512>(s, d, gm);
        run<32, 4, 400>(s, d, gm);
        run<64, 4, 512>(s, d, gm);
        run<96, 4, 512>(s, d, gm);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
