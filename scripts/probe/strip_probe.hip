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

__global__ void flat_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
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
    for (int gm : {1, 2, 4}) {
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
