// Follow-up to mall_probe.hip: does the Infinity Cache keep a kernel's output when the slab is NOT contiguous but has the
// f-k passes' shape -- every row of the 20000 x 480000-byte block x 25 strips (stride 19200 B) of W bytes?  Two in-place
// read-modify-write launches per slab, slab after slab, against the same launches in the non-chained order (all first
// launches, then all second ones: same launch overheads, no residency).    hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float vec4 __attribute__((ext_vector_type(4)));
constexpr int NX = 20000, N1 = 25;
constexpr size_t PITCH = 480000, SUB = 19200;      // bytes

// slab = column window [off, off + W) of every n1 sub-row of every channel; one workgroup walks whole (row, strip) pieces
__global__ __launch_bounds__(256) void rmw_strided(char* base, int W, size_t off, float a) {
    const int per = W / 16;                                   // vec4 per piece
    const long npiece = (long)NX * N1;
    const long total = npiece * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long piece = i / per;
        const int j = (int)(i - piece * per);
        const long row = piece / N1;
        const int n1 = (int)(piece - row * N1);
        vec4* p = reinterpret_cast<vec4*>(base + row * PITCH + n1 * SUB + off) + j;
        *p = *p * a;
    }
}

int main() {
    const size_t total = (size_t)NX * PITCH;
    char* a;
    CK(hipMalloc(&a, total));
    CK(hipMemset(a, 0, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int W : {128, 256, 512, 1280, 3840}) {
        const int nslab = SUB / W;
        const double slab_mb = (double)NX * N1 * W / 1e6;
        for (int chained = 1; chained >= 0; --chained) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (chained) {
                    for (int s = 0; s < nslab; ++s)
                        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(rmw_strided, dim3(2048), dim3(256), 0, 0, a, W, (size_t)s * W, 1.0f);
                } else {
                    for (int k = 0; k < 2; ++k)
                        for (int s = 0; s < nslab; ++s) hipLaunchKernelGGL(rmw_strided, dim3(2048), dim3(256), 0, 0, a, W, (size_t)s * W, 1.0f);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("strip %5d B  slab %6.1f MB x %3d  %s: %7.3f ms  eff %5.2f TB/s\n", W, slab_mb, nslab, chained ? "chained    " : "not chained",
                   best, 2 * 2.0 * total / best / 1e9);
        }
    }
    return 0;
}
