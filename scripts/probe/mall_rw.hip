// What does the 256 MiB Infinity Cache absorb: re-reads only, or writes too?  A buffer of S bytes is swept K times by
// (r) a read-only kernel, (w) a write-only kernel, (rw) an in-place read-modify-write kernel; S from 32 MiB (fits) to 4 GiB
// (does not).  If the cache held dirty lines (write-back), (w) and (rw) on a resident buffer would run far above the HBM write
// rate; if it is write-through ("caches the contents of memory"), only (r) gains.
//     hipcc --offload-arch=gfx950 -O3 mall_rw.hip -o mall_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float vec4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rd_k(const vec4* __restrict__ x, size_t n, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    vec4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const vec4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        acc += (a + b) + (c + d);
    }
    for (; i < n; i += stride) acc += x[i];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void wr_k(vec4* __restrict__ x, size_t n, float a) {
    const size_t stride = (size_t)gridDim.x * 256;
    const vec4 v = {a, a, a, a};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = v;
}
__global__ __launch_bounds__(256) void rmw_k(vec4* __restrict__ x, size_t n, float a) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const vec4 p = x[i], q = x[i + stride], r = x[i + 2 * stride], s = x[i + 3 * stride];
        x[i] = p * a; x[i + stride] = q * a; x[i + 2 * stride] = r * a; x[i + 3 * stride] = s * a;
    }
    for (; i < n; i += stride) x[i] = x[i] * a;
}
// producer -> consumer: a kernel writes S bytes, the next one reads them (out of place: src is a second resident buffer)
__global__ __launch_bounds__(256) void cp_k(const vec4* __restrict__ s, vec4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}

int main() {
    const size_t maxb = (size_t)4 << 30;
    char *a, *b;
    float* out;
    CK(hipMalloc(&a, maxb));
    CK(hipMalloc(&b, maxb));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, maxb));
    CK(hipMemset(b, 0, maxb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 4096;
    printf("%8s  %10s %10s %10s %12s   (TB/s; rw counts read + written bytes, w->r counts the read only)\n", "S", "read", "write", "rmw", "write->read");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t S = mb << 20, n = S / 16;
        const int K = (int)((mb <= 512) ? 4096 / mb * 4 : 8);
        float t[4];
        for (int kind = 0; kind < 4; ++kind) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                // warm: one sweep so that the first timed one sees what the steady state sees
                if (kind == 0) rd_k<<<grid, 256>>>((const vec4*)a, n, out);
                else if (kind == 1) wr_k<<<grid, 256>>>((vec4*)a, n, 1.f);
                else if (kind == 2) rmw_k<<<grid, 256>>>((vec4*)a, n, 1.f);
                float ms = 0.f;
                if (kind < 3) {
                    CK(hipEventRecord(e0, 0));
                    for (int k = 0; k < K; ++k) {
                        if (kind == 0) rd_k<<<grid, 256>>>((const vec4*)a, n, out);
                        else if (kind == 1) wr_k<<<grid, 256>>>((vec4*)a, n, 1.f);
                        else rmw_k<<<grid, 256>>>((vec4*)a, n, 1.f);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                } else {
                    // write S (fresh values), then read it back: time the reads only (K pairs, events around each read)
                    for (int k = 0; k < K; ++k) {
                        wr_k<<<grid, 256>>>((vec4*)a, n, (float)k);
                        CK(hipEventRecord(e0, 0));
                        rd_k<<<grid, 256>>>((const vec4*)a, n, out);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float m1;
                        CK(hipEventElapsedTime(&m1, e0, e1));
                        ms += m1;
                    }
                }
                if (ms < best) best = ms;
            }
            const double bytes = (double)S * K * (kind == 2 ? 2 : 1);
            t[kind] = (float)(bytes / best / 1e9);
        }
        printf("%5zu MiB  %10.2f %10.2f %10.2f %12.2f\n", mb, t[0], t[1], t[2], t[3]);
        fflush(stdout);
    }
    // copy between two resident buffers: both streams cached?
    for (size_t mb : {32, 64, 128, 1024}) {
        const size_t S = mb << 20, n = S / 16;
        const int K = 32;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            cp_k<<<grid, 256>>>((const vec4*)a, (vec4*)b, n);
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < K; ++k) cp_k<<<grid, 256>>>((const vec4*)a, (vec4*)b, n);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("copy %5zu MiB -> %5zu MiB, both resident: %6.2f TB/s (read + written)\n", mb, mb, 2.0 * S * K / best / 1e9);
    }
    return 0;
}
