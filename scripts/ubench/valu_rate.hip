// micro-benchmark: issue cost of scalar vs packed f32 VALU on gfx950 (cycles per wave instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2f sv = {s, s * 0.5f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 2) {   // v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 3) {   // v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 4) {   // v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 5) {   // v_fmac_f32 with SGPR operand
            asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                         "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
        } else if (MODE == 6) {   // v_pk_fma_f32 with SGPR pair + op_sel broadcast
            asm volatile("v_pk_fma_f32 %0, %0, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %3 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %4, %4, %8, %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %7 op_sel_hi:[1,0,1]"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sv));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, float* d, int waves_per_simd) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", d, w);
        run<1>("v_pk_fma_f32", d, w);
        run<2>("v_add_f32", d, w);
        run<3>("v_pk_add_f32", d, w);
        run<4>("v_pk_mul_f32", d, w);
        run<5>("v_fmac_f32 (sgpr)", d, w);
        run<6>("v_pk_fma_f32 (sgpr,bcast)", d, w);
    }
    return 0;
}
