// Gate for chaining the three MIDDLE passes of the f-k filter through the 256 MiB Infinity Cache.
//
// After pass A the 20 000 x 60 000 complex block factors into C1 N1 = 625 independent slabs (q, q1) of C2 = 800 rows x one
// sub-row of N2 = 2400 complex (19.2 KB runs, row pitch 480 KB): 15.4 MB each.  The middle passes touch a slab like this:
//   channel-first  C  : tiles of 800 rows x 128 B, read all rows, write the live ones
//                  B  : live rows, 19.2 KB runs read + written, + 9.6 KB of mask per row (HBM, read once)
//                  C' : tiles of 800 rows x 128 B, read the live rows, write all
//   time-first     Bf : rows read (19.2 KB) from the block, the kept columns written to the compact workspace W
//                  Cm : tiles of 800 rows x 128 B of W's band columns read + written, + 64 B of mask per row strip
//                  Bi : W rows read, block rows written
// This probe moves exactly those bytes with trivial arithmetic (so that a stale hand-off shows as a wrong checksum) in four
// orders:  (a) three full passes (today);  (b) the same kernels launched slab group by slab group;  (c) ONE persistent
// launch that deals tiles from an in-order ticket list [P1(g), P2(g-1), P3(g-2) interleaved] with an arrive counter per
// (phase, group): payload by plain stores + one-lane agent release per tile, consumer one relaxed poll + one agent acquire;
// (d) as (c) with write-through (sc1) 16-byte payload stores and no release fence.
//     hipcc --offload-arch=gfx950 -O3 mall_chain.hip -o mall_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float vec4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int NX = 20000, C1 = 25, C2 = 800, N1 = 25;
constexpr size_t PITCH = 480000, SUB = 19200;      // bytes: block row, one sub-row
constexpr int NSLAB = C1 * N1;                     // 625
constexpr int STRIPS = (int)(SUB / 128);           // 150 column tiles per slab
constexpr int THR = 256;

struct Chain {
    char* D;            // block [NX][PITCH]
    char* W;            // time-first workspace [NX][wpitch]
    const char* M;      // mask bytes (read-only stream)
    size_t wpitch;      // bytes
    size_t wsub;        // bytes of a sub-row block of W
    int wstrips;        // band strips (128 B) per sub-row block that Cm touches
    const int* live;    // [nlive] live row indices inside a slab (channel-first)
    int nlive;
    const unsigned* livebits;   // [C2 / 32]
    int tf;             // 1: time-first chain
};

struct Tile { int phase, slab, idx, group; };

template <bool SC1>
__device__ __forceinline__ void st16(vec4* p, vec4 v) {
    if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}

// ---- tile bodies (256 threads) ----
// column tile of the block: rows x 128 B = C2 * 8 pieces of 16 B, 25 per thread
template <bool SC1>
__device__ void col_tile(const Chain& c, char* base, size_t pitch, size_t coloff, bool read_live_only, bool write_live_only, float add,
                         const char* mask, size_t mpitch) {
    const int tid = threadIdx.x;
    const int j = tid & 7;
    char* colp = base + coloff + j * 16;
    const char* mcol = mask ? mask + coloff / 2 + j * 16 : nullptr;
#pragma unroll 1
    for (int i0 = 0; i0 < 25; i0 += 5) {           // 5 pieces per thread in flight, 16 waves per CU
        vec4 r[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int row = ((i0 + i) * THR + tid) >> 3;
            const bool lv = (c.livebits[row >> 5] >> (row & 31)) & 1u;
            if (!read_live_only || lv) r[i] = *reinterpret_cast<const vec4*>(colp + (size_t)row * pitch);
            else r[i] = vec4{0.f, 0.f, 0.f, 0.f};
        }
        if (mask && j < 4) {                       // 64 B of mask per row strip
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = ((i0 + i) * THR + tid) >> 3;
                r[i] += *reinterpret_cast<const vec4*>(mcol + (size_t)row * mpitch) * 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int row = ((i0 + i) * THR + tid) >> 3;
            const bool lv = (c.livebits[row >> 5] >> (row & 31)) & 1u;
            if (!write_live_only || lv) st16<SC1>(reinterpret_cast<vec4*>(colp + (size_t)row * pitch), r[i] + add);
        }
    }
}

// row tile: two rows; read nin pieces of each from src, (mask: nm pieces), write nout pieces to dst
template <bool SC1>
__device__ void row_tile(const char* s0, const char* s1, int nin, char* d0, char* d1, int nout, const char* m0, const char* m1, int nm, float add) {
    const int tid = threadIdx.x;
    vec4 a[5], b[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = i * THR + tid;
        a[i] = (p < nin) ? reinterpret_cast<const vec4*>(s0)[p] : vec4{0.f, 0.f, 0.f, 0.f};
        b[i] = (p < nin) ? reinterpret_cast<const vec4*>(s1)[p] : vec4{0.f, 0.f, 0.f, 0.f};
    }
    if (m0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = i * THR + tid;
            if (p < nm) {
                a[i] += reinterpret_cast<const vec4*>(m0)[p] * 0.0f;
                b[i] += reinterpret_cast<const vec4*>(m1)[p] * 0.0f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = i * THR + tid;
        if (p < nout) {
            st16<SC1>(reinterpret_cast<vec4*>(d0) + p, a[i] + add);
            st16<SC1>(reinterpret_cast<vec4*>(d1) + p, b[i] + add);
        }
    }
}

template <bool SC1>
__device__ void run_tile(const Chain& c, int phase, int slab, int idx) {
    const int q = slab / N1, q1 = slab % N1;
    const size_t row0 = (size_t)q * C2;
    if (!c.tf) {
        if (phase == 0) col_tile<SC1>(c, c.D + row0 * PITCH, PITCH, q1 * SUB + (size_t)idx * 128, false, true, 1.0f, nullptr, 0);
        else if (phase == 1) {
            const int r0 = c.live[2 * idx], r1 = c.live[min(2 * idx + 1, c.nlive - 1)];
            char* p0 = c.D + (row0 + r0) * PITCH + q1 * SUB;
            char* p1 = c.D + (row0 + r1) * PITCH + q1 * SUB;
            row_tile<SC1>(p0, p1, 1200, p0, p1, 1200, c.M + (row0 + r0) * (PITCH / 2) + q1 * (SUB / 2), c.M + (row0 + r1) * (PITCH / 2) + q1 * (SUB / 2), 600, 2.0f);
        } else col_tile<SC1>(c, c.D + row0 * PITCH, PITCH, q1 * SUB + (size_t)idx * 128, true, false, 4.0f, nullptr, 0);
    } else {
        const int nw = (int)(c.wsub / 16);
        if (phase == 0) {
            const char* p0 = c.D + (row0 + 2 * idx) * PITCH + q1 * SUB;
            char* w0 = c.W + (row0 + 2 * idx) * c.wpitch + q1 * c.wsub;
            row_tile<SC1>(p0, p0 + PITCH, 1200, w0, w0 + c.wpitch, nw, nullptr, nullptr, 0, 1.0f);
        } else if (phase == 1) {
            col_tile<SC1>(c, c.W + row0 * c.wpitch, c.wpitch, q1 * c.wsub + (size_t)idx * 128, false, false, 2.0f, c.M + row0 * (c.wpitch / 2), c.wpitch / 2);
        } else {
            char* p0 = c.D + (row0 + 2 * idx) * PITCH + q1 * SUB;
            const char* w0 = c.W + (row0 + 2 * idx) * c.wpitch + q1 * c.wsub;
            // every piece of the block row is rewritten; pieces beyond the kept columns carry the constant
            row_tile<SC1>(w0, w0 + c.wpitch, nw, p0, p0 + PITCH, 1200, nullptr, nullptr, 0, 4.0f);
        }
    }
}

// ---- (a), (b): one phase over a tile range, static grid-stride ----
__global__ __launch_bounds__(THR, 4) void phase_k(Chain c, const Tile* tiles, int t0, int t1) {
    for (int t = t0 + blockIdx.x; t < t1; t += gridDim.x) {
        const Tile tl = tiles[t];
        run_tile<false>(c, tl.phase, tl.slab, tl.idx);
    }
}

// ---- (c), (d): persistent, ticket order, arrive / wait counters ----
template <bool SC1>
__global__ __launch_bounds__(THR, 4) void chain_k(Chain c, const Tile* sched, int nsched, unsigned* ticket, unsigned* cnt, const unsigned* need,
                                                  int ngroups, unsigned* err) {
    __shared__ int s_t;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_t = (int)atomicAdd(ticket, 1u);
        __syncthreads();
        const int t = s_t;
        if (t >= nsched) break;
        const Tile tl = sched[t];
        if (tl.phase > 0) {
            if (threadIdx.x == 0) {
                gu32* f = (gu32*)(cnt + (size_t)(tl.phase - 1) * ngroups + tl.group);
                const unsigned want = need[(size_t)(tl.phase - 1) * ngroups + tl.group];
                unsigned spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) { atomicAdd(err, 1u); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        run_tile<SC1>(c, tl.phase, tl.slab, tl.idx);
        if (tl.phase < 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                if (!SC1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add((gu32*)(cnt + (size_t)tl.phase * ngroups + tl.group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void fill_k(vec4* x, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = (float)(i % 977);
        x[i] = vec4{v, v + 0.25f, v + 0.5f, v + 0.75f};
    }
}

// order-independent checksum: the bit patterns summed as integers
__global__ void sum_k(const vec4* x, size_t n, unsigned long long* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const vec4 v = x[i];
        acc += (unsigned long long)__float_as_uint(v.x) + 3ull * __float_as_uint(v.y) + 5ull * __float_as_uint(v.z) + 7ull * __float_as_uint(v.w);
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
    const size_t total = (size_t)NX * PITCH;
    char *D, *W, *M;
    CK(hipMalloc(&D, total));
    const size_t wsub = 12000, wpitch = wsub * N1;           // 0.625 of the half spectrum kept
    CK(hipMalloc(&W, (size_t)NX * wpitch));
    CK(hipMalloc(&M, total / 2));
    CK(hipMemset(M, 0, total / 2));
    CK(hipMemset(W, 0, (size_t)NX * wpitch));
    unsigned long long* dsum;
    CK(hipMalloc(&dsum, 8));
    unsigned *ticket, *cnt, *need, *err;
    CK(hipMalloc(&ticket, 4));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&cnt, 3 * NSLAB * 4));
    CK(hipMalloc(&need, 3 * NSLAB * 4));
    int* dlive;
    unsigned* dbits;
    CK(hipMalloc(&dlive, C2 * 4));
    CK(hipMalloc(&dbits, (C2 / 32) * 4));
    Tile* dtiles;
    CK(hipMalloc(&dtiles, sizeof(Tile) * 3 * (size_t)NSLAB * 400));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int ncu = 256;
    {
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        ncu = pr.multiProcessorCount;
    }
    const int NWG = ncu * 4;

    struct Case { const char* name; int tf; double f; int wstrips; };
    const Case cases[] = {{"channel-first, 27 % of the rows live (classic fan)", 0, 0.27, 0},
                          {"channel-first, every row live (dense mask)", 0, 1.0, 0},
                          {"time-first, 62.5 % of the columns kept, 40 % band (hybrid_ninf)", 1, 1.0, 60}};
    for (const Case& cs : cases) {
        std::vector<int> live;
        std::vector<unsigned> bits(C2 / 32, 0u);
        for (int r = 0; r < C2; ++r)
            if (((r * 37) % 100) < (int)(cs.f * 100 + 0.5)) { live.push_back(r); bits[r >> 5] |= 1u << (r & 31); }
        CK(hipMemcpy(dlive, live.data(), live.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
        Chain c{D, W, M, wpitch, wsub, cs.wstrips, dlive, (int)live.size(), dbits, cs.tf};
        // tiles of a slab per phase
        int per[3];
        if (!cs.tf) { per[0] = STRIPS; per[1] = ((int)live.size() + 1) / 2; per[2] = STRIPS; }
        else { per[0] = C2 / 2; per[1] = cs.wstrips; per[2] = C2 / 2; }
        // HBM bytes an ideal chain moves / the three full passes move
        double b_full, b_chain;
        if (!cs.tf) {
            const double fl = (double)live.size() / C2;
            b_full = total * (1 + fl) + total * (2 * fl + 0.5 * fl) + total * (fl + 1);
            b_chain = total * (1 + 0.5 * fl + 1);
        } else {
            const double fk = (double)wsub / SUB, fb = cs.wstrips * 128.0 / SUB;
            b_full = total * (1 + fk) + total * (2 * fb + 0.5 * fb) + total * (fk + 1);
            b_chain = total * (1 + 0.5 * fb + 1);
        }
        printf("== %s: three full passes move %.1f GB, an ideal chain %.1f GB of HBM traffic\n", cs.name, b_full / 1e9, b_chain / 1e9);
        unsigned long long ref_sum = 0;
        for (int gs : {0, 4, 8, 16, 32}) {                 // slabs per group (0: unchained)
            for (int mode = (gs == 0 ? 0 : 1); mode <= (gs == 0 ? 0 : 3); ++mode) {
                const int S = gs ? gs : NSLAB;
                const int ngroups = (NSLAB + S - 1) / S;
                // phase-major tile list ordered by group (modes a, b) and the interleaved ticket schedule (c, d)
                std::vector<Tile> tl;
                std::vector<int> start(3 * (ngroups + 1), 0);
                for (int p = 0; p < 3; ++p)
                    for (int g = 0; g < ngroups; ++g) {
                        start[p * (ngroups + 1) + g] = (int)tl.size();
                        for (int s = g * S; s < std::min(NSLAB, (g + 1) * S); ++s)
                            for (int i = 0; i < per[p]; ++i) tl.push_back(Tile{p, s, i, g});
                        start[p * (ngroups + 1) + g + 1] = (int)tl.size();
                    }
                std::vector<unsigned> hneed(3 * ngroups, 0u);
                for (const Tile& t : tl) hneed[t.phase * ngroups + t.group]++;
                std::vector<Tile> sched;
                if (mode >= 2) {
                    for (int step = 0; step < ngroups + 2; ++step) {
                        // merge the three (phase, group = step - phase) lists by fractional position
                        int pos[3] = {0, 0, 0}, len[3], off[3];
                        for (int p = 0; p < 3; ++p) {
                            const int g = step - p;
                            if (g < 0 || g >= ngroups) { len[p] = 0; off[p] = 0; continue; }
                            off[p] = start[p * (ngroups + 1) + g];
                            len[p] = start[p * (ngroups + 1) + g + 1] - off[p];
                        }
                        for (;;) {
                            int best = -1;
                            double bf = 2.0;
                            for (int p = 0; p < 3; ++p)
                                if (pos[p] < len[p]) {
                                    const double fr = (pos[p] + 0.5) / len[p];
                                    if (fr < bf) { bf = fr; best = p; }
                                }
                            if (best < 0) break;
                            sched.push_back(tl[off[best] + pos[best]++]);
                        }
                    }
                    CK(hipMemcpy(dtiles, sched.data(), sched.size() * sizeof(Tile), hipMemcpyHostToDevice));
                } else
                    CK(hipMemcpy(dtiles, tl.data(), tl.size() * sizeof(Tile), hipMemcpyHostToDevice));
                CK(hipMemcpy(need, hneed.data(), hneed.size() * 4, hipMemcpyHostToDevice));
                float best = 1e30f;
                unsigned long long sum = 0;
                unsigned herr = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    fill_k<<<4096, 256>>>((vec4*)D, total / 16);
                    CK(hipMemsetAsync(ticket, 0, 4, 0));
                    CK(hipMemsetAsync(err, 0, 4, 0));
                    CK(hipMemsetAsync(cnt, 0, 3 * NSLAB * 4, 0));
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0) {
                        for (int p = 0; p < 3; ++p)
                            phase_k<<<NWG, THR>>>(c, dtiles, start[p * (ngroups + 1)], start[p * (ngroups + 1) + ngroups]);
                    } else if (mode == 1) {
                        for (int g = 0; g < ngroups; ++g)
                            for (int p = 0; p < 3; ++p) {
                                const int a = start[p * (ngroups + 1) + g], b = start[p * (ngroups + 1) + g + 1];
                                phase_k<<<std::min(NWG, b - a), THR>>>(c, dtiles, a, b);
                            }
                    } else if (mode == 2)
                        chain_k<false><<<NWG, THR>>>(c, dtiles, (int)sched.size(), ticket, cnt, need, ngroups, err);
                    else
                        chain_k<true><<<NWG, THR>>>(c, dtiles, (int)sched.size(), ticket, cnt, need, ngroups, err);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                    CK(hipMemset(dsum, 0, 8));
                    sum_k<<<4096, 256>>>((const vec4*)D, total / 16, dsum);
                    CK(hipMemcpy(&sum, dsum, 8, hipMemcpyDeviceToHost));
                    unsigned e;
                    CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
                    herr += e;
                    if (mode == 0) ref_sum = sum;
                }
                const char* mn[] = {"(a) three full passes          ", "(b) launches per slab group    ", "(c) persistent, release fences ", "(d) persistent, sc1 stores     "};
                printf("%s group %2d slabs = %5.1f MB: %7.3f ms   %5.2f TB/s of the full-pass bytes   checksum %s%s\n", mn[mode], gs,
                       gs * C2 * (double)SUB / 1e6, best, b_full / best / 1e9, sum == ref_sum ? "ok" : "DIFFERS", herr ? "  SPIN TIMEOUT" : "");
                fflush(stdout);
            }
        }
    }
    return 0;
}
