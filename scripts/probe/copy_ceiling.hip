// Where does this pool's copy rate come from?  MI355X_MICROARCH.md records ~6.3 TB/s for a float4 copy; every streaming pass of
// this library (f-k passes A / C / C' / A', the matched filter's 1 : 2 stream) sits at 4.8-5.3 TB/s read + write on the boxes of
// this pool (profiles/r04a/dma_probe.txt, r05a/mall_rw.txt).  One table over the things that could separate the two:
//   * how the grid walks the buffer: naive (one 16-byte element per thread, grid = n / 256), persistent workgroups with chunks
//     dealt round-robin, XCD-contiguous ranges (the matched filter's order), one contiguous range per workgroup;
//   * the run a workgroup moves per step (4 / 8 / 16 / 32 KiB) and the workgroups per CU (1 ... 8);
//   * the cache-policy bits of the loads and of the stores (plain, sc0, nt, sc1, sc0 sc1, sc0 sc1 nt) through buffer instructions;
//   * the distance between source and destination modulo the HBM channel interleave (read / write turnaround on one channel);
//   * the buffer size (1 / 4 / 9.6 GB: what the 256-MiB Infinity Cache still absorbs);
//   * read-only, write-only, in-place (read a run, write it back: what passes C, B, C', A' do) and write-after-read of the
//     same lines (does a store to a line that has just been read cost less than a store to a cold line?).
//       hipcc --offload-arch=gfx950 -O3 copy_ceiling.hip -o copy_ceiling ; ./copy_ceiling [quick]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Mode { COPY = 0, READ = 1, WRITE = 2, INPLACE = 3, R1W2 = 4 };
enum Map { ROUND_ROBIN = 0, XCD_RANGES = 1, WG_RANGES = 2 };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// persistent workgroups of 256 threads; a chunk = U x 4 KiB contiguous; loads of chunk i + 1 are issued before the stores of chunk i
template <int MODE, int MAPK, int U, int AUXL, int AUXS>
__global__ __launch_bounds__(256) void stream_k(const char* __restrict__ s, char* __restrict__ d, char* __restrict__ d2,
                                                size_t nchunks, float* sink) {
    constexpr unsigned CB = U * 4096u;
    const int wg = blockIdx.x, nwg = gridDim.x;
    size_t c, cend, cstep;
    if (MAPK == ROUND_ROBIN) { c = wg; cend = nchunks; cstep = nwg; }
    else if (MAPK == XCD_RANGES) {
        const int xcd = wg & 7, wq = wg >> 3, nq = nwg >> 3;
        const size_t lo = nchunks * xcd / 8, hi = nchunks * (xcd + 1) / 8;
        c = lo + wq; cend = hi; cstep = nq;
    } else {
        const size_t lo = nchunks * wg / nwg, hi = nchunks * (wg + 1) / nwg;
        c = lo; cend = hi; cstep = 1;
    }
    const unsigned voff = threadIdx.x * 16u;
    f4 v[U], w[U];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](size_t cc, f4 (&r)[U]) {
        const auto rs = rsrc_of(s + cc * CB, CB);
#pragma unroll
        for (int q = 0; q < U; ++q) r[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + q * 4096u, 0, AUXL));
    };
    if (MODE != WRITE) { if (c < cend) load(c, v); }
    else {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, (float)q};
    }
    for (; c < cend; c += cstep) {
#pragma unroll
        for (int q = 0; q < U; ++q) w[q] = v[q];
        if (MODE != WRITE && c + cstep < cend) load(c + cstep, v);
        if (MODE == READ) {
#pragma unroll
            for (int q = 0; q < U; ++q) acc += w[q];
        } else {
            const auto rd = rsrc_of(d + c * CB, CB);
#pragma unroll
            for (int q = 0; q < U; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w[q]), rd, voff + q * 4096u, 0, AUXS);
            if (MODE == R1W2) {
                const auto r2 = rsrc_of(d2 + c * CB, CB);
#pragma unroll
                for (int q = 0; q < U; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w[q] + 1.f), r2, voff + q * 4096u, 0, AUXS);
            }
        }
    }
    if (MODE == READ && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// persistent WAVES that claim their chunks from a ticket counter (one global counter, or one per XCD range) instead of a static
// stride: a chunk = U x 1 KiB per wave; the ticket of the chunk after the next one is requested at the top of an iteration, so
// the atomic's latency is off the path.  The point: statically dealt persistent workgroups drift apart over a pass, the window
// of addresses in flight widens and the DRAM pages opened per byte go up; tickets keep the window as tight as the hardware's
// in-order dispatch of a naive grid does.
template <int MODE, int U, int AUXL, int AUXS, bool PERXCD>
__global__ __launch_bounds__(256) void stream_ticket(const char* __restrict__ s, char* __restrict__ d, char* __restrict__ d2,
                                                     size_t nchunks, unsigned* tickets, float* sink) {
    constexpr unsigned CB = U * 1024u;
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    unsigned* tk = tickets + (PERXCD ? xcd * 32 : 0);              // counters 128 B apart
    const size_t lo = PERXCD ? nchunks * xcd / 8 : 0, hi = PERXCD ? nchunks * (xcd + 1) / 8 : nchunks;
    // the returned value is looked at as late as possible: memory operations of a wave return in order, so asking for the
    // ticket at once would drain the loads and stores issued before it
    auto ask = [&]() -> unsigned {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(tk, 1u);
        return t;
    };
    auto claim = [&]() -> size_t { return lo + (size_t)__builtin_amdgcn_readfirstlane(ask()); };
    const unsigned voff = lane * 16u;
    f4 v[U], w[U];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](size_t cc, f4 (&r)[U]) {
        const auto rs = rsrc_of(s + cc * CB, CB);
#pragma unroll
        for (int q = 0; q < U; ++q) r[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + q * 1024u, 0, AUXL));
    };
    size_t c = claim(), cn = claim();
    if (MODE != WRITE) { if (c < hi) load(c, v); }
    else {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, (float)q};
    }
    while (c < hi) {
        const unsigned traw = ask();                                 // two ahead; consumed at the end of the iteration
#pragma unroll
        for (int q = 0; q < U; ++q) w[q] = v[q];
        if (MODE != WRITE && cn < hi) load(cn, v);
        if (MODE == READ) {
#pragma unroll
            for (int q = 0; q < U; ++q) acc += w[q];
        } else {
            const auto rd = rsrc_of(d + c * CB, CB);
#pragma unroll
            for (int q = 0; q < U; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w[q]), rd, voff + q * 1024u, 0, AUXS);
            if (MODE == R1W2) {
                const auto r2 = rsrc_of(d2 + c * CB, CB);
#pragma unroll
                for (int q = 0; q < U; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w[q] + 1.f), r2, voff + q * 1024u, 0, AUXS);
            }
        }
        c = cn; cn = lo + (size_t)__builtin_amdgcn_readfirstlane(traw);
    }
    if (MODE == READ && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// the naive form: one element per thread
__global__ __launch_bounds__(256) void naive_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int PER>
__global__ __launch_bounds__(256) void naive_copy_n(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
    const size_t b = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
    f4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) v[q] = s[b + q * 256];
#pragma unroll
    for (int q = 0; q < PER; ++q) d[b + q * 256] = v[q];
}
__global__ void grid_stride_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static hipEvent_t ea, eb;
template <class F>
static float timeit(F f, int reps = 5) {
    for (int i = 0; i < 2; ++i) f();
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(ea));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(eb));
        CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        t.push_back(ms / reps);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    return t[1];
}

static const char* aux_name(int a) {
    switch (a) { case 0: return "plain"; case 1: return "sc0"; case 2: return "nt"; case 16: return "sc1"; case 17: return "sc0 sc1";
                 case 18: return "sc1 nt"; case 19: return "sc0 sc1 nt"; case 3: return "sc0 nt"; }
    return "?";
}
static const char* mode_name(int m) { static const char* n[] = {"copy", "read", "write", "in-place", "r1w2"}; return n[m]; }
static const char* map_name(int m) { static const char* n[] = {"round-robin", "XCD ranges", "WG ranges"}; return n[m]; }
static double bytes_of(int mode, size_t bytes) { return mode == READ || mode == WRITE ? (double)bytes : mode == R1W2 ? 3.0 * bytes : 2.0 * bytes; }

template <int MODE, int MAPK, int U, int AUXL, int AUXS>
static float run(const char* s, char* d, char* d2, size_t bytes, int wgs_per_cu, float* sink, const char* note = "") {
    const size_t nchunks = bytes / (U * 4096u);
    const float ms = timeit([&] { hipLaunchKernelGGL((stream_k<MODE, MAPK, U, AUXL, AUXS>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, s, d, d2, nchunks, sink); });
    printf("%-8s | %-11s | %2d KiB | %d WG/CU | ld %-10s st %-10s | %5.2f GB | %7.3f ms | %5.2f TB/s %s\n", mode_name(MODE), map_name(MAPK), U * 4,
           wgs_per_cu, MODE == WRITE ? "-" : aux_name(AUXL), MODE == READ ? "-" : aux_name(AUXS), bytes * 1e-9, ms, bytes_of(MODE, bytes) / ms * 1e-9, note);
    fflush(stdout);
    return ms;
}

template <int MODE, int U, int AUXL, int AUXS, bool PERXCD>
static float run_ticket(const char* s, char* d, char* d2, size_t bytes, int wgs_per_cu, unsigned* tickets, float* sink) {
    const size_t nchunks = bytes / (U * 1024u);
    const float ms = timeit([&] {
        (void)hipMemsetAsync(tickets, 0, 8 * 128, 0);
        hipLaunchKernelGGL((stream_ticket<MODE, U, AUXL, AUXS, PERXCD>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, s, d, d2, nchunks, tickets, sink);
    });
    printf("%-8s | tickets %-7s | %2d KiB per wave | %d WG/CU | ld %-10s st %-10s | %5.2f GB | %7.3f ms | %5.2f TB/s\n", mode_name(MODE),
           PERXCD ? "per XCD" : "global", U, wgs_per_cu, MODE == WRITE ? "-" : aux_name(AUXL), MODE == READ ? "-" : aux_name(AUXS), bytes * 1e-9, ms,
           bytes_of(MODE, bytes) / ms * 1e-9);
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t big = (size_t)20000 * 120000 * 4;                      // 9.6 GB, the bench block
    const size_t slack = (size_t)64 << 20;
    char *s, *d, *d2; float* sink;
    CK(hipMalloc(&s, big + slack)); CK(hipMalloc(&d, big + slack)); CK(hipMalloc(&d2, big + slack)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(s, 1, big + slack)); CK(hipMemset(d, 0, big + slack)); CK(hipMemset(d2, 0, big + slack));
    printf("base addresses: s %p d %p d2 %p (d - s = %zd MiB)\n", (void*)s, (void*)d, (void*)d2, (ssize_t)(d - s) >> 20);
    const size_t n4 = big / 16;

    printf("# 1. the naive forms at 9.6 GB (one launch = the whole buffer)\n");
    { float ms = timeit([&] { hipLaunchKernelGGL(naive_copy, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); });
      printf("naive 1 x 16 B per thread, grid %zu: %7.3f ms %5.2f TB/s\n", n4 / 256, ms, 2.0 * big / ms * 1e-9); }
    { float ms = timeit([&] { hipLaunchKernelGGL(naive_copy_n<4>, dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); });
      printf("naive 4 x 16 B per thread, grid %zu: %7.3f ms %5.2f TB/s\n", n4 / 1024, ms, 2.0 * big / ms * 1e-9); }
    { float ms = timeit([&] { hipLaunchKernelGGL(naive_copy_n<8>, dim3((unsigned)(n4 / 2048)), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); });
      printf("naive 8 x 16 B per thread, grid %zu: %7.3f ms %5.2f TB/s\n", n4 / 2048, ms, 2.0 * big / ms * 1e-9); }
    for (int g : {2, 4, 8, 16}) {
        float ms = timeit([&] { hipLaunchKernelGGL(grid_stride_copy, dim3(256 * g), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); });
        printf("grid-stride copy, %2d WG/CU: %7.3f ms %5.2f TB/s\n", g, ms, 2.0 * big / ms * 1e-9);
    }
    { float ms = timeit([&] { CK(hipMemcpyAsync(d, s, big, hipMemcpyDeviceToDevice, 0)); });
      printf("hipMemcpyAsync D2D: %7.3f ms %5.2f TB/s\n", ms, 2.0 * big / ms * 1e-9); }

    if (argc > 1 && !strcmp(argv[1], "tickets")) {
        unsigned* tickets; CK(hipMalloc(&tickets, 8 * 128));
        printf("# 8. persistent waves with TICKETED chunks (global counter / one per XCD range), against the static maps of section 2\n");
        for (int w : {1, 2, 4, 8}) {
            run_ticket<COPY, 1, 0, 0, false>(s, d, d2, big, w, tickets, sink); run_ticket<COPY, 2, 0, 0, false>(s, d, d2, big, w, tickets, sink);
            run_ticket<COPY, 4, 0, 0, false>(s, d, d2, big, w, tickets, sink); run_ticket<COPY, 8, 0, 0, false>(s, d, d2, big, w, tickets, sink);
            run_ticket<COPY, 1, 0, 0, true>(s, d, d2, big, w, tickets, sink); run_ticket<COPY, 2, 0, 0, true>(s, d, d2, big, w, tickets, sink);
            run_ticket<COPY, 4, 0, 0, true>(s, d, d2, big, w, tickets, sink); run_ticket<COPY, 8, 0, 0, true>(s, d, d2, big, w, tickets, sink);
        }
        for (int w : {2, 4, 8}) {
            run_ticket<COPY, 4, 2, 2, false>(s, d, d2, big, w, tickets, sink); run_ticket<COPY, 4, 2, 16, false>(s, d, d2, big, w, tickets, sink);
            run_ticket<INPLACE, 4, 0, 0, false>(s, s, d2, big, w, tickets, sink); run_ticket<INPLACE, 4, 2, 2, false>(s, s, d2, big, w, tickets, sink);
            run_ticket<INPLACE, 4, 0, 0, true>(s, s, d2, big, w, tickets, sink); run_ticket<INPLACE, 4, 2, 2, true>(s, s, d2, big, w, tickets, sink);
            run_ticket<R1W2, 4, 0, 0, false>(s, d, d2, big, w, tickets, sink); run_ticket<R1W2, 4, 0, 2, false>(s, d, d2, big, w, tickets, sink);
            run_ticket<R1W2, 4, 0, 2, true>(s, d, d2, big, w, tickets, sink);
            run_ticket<READ, 4, 0, 0, false>(s, d, d2, big, w, tickets, sink); run_ticket<READ, 4, 2, 0, false>(s, d, d2, big, w, tickets, sink);
            run_ticket<WRITE, 4, 0, 0, false>(s, d, d2, big, w, tickets, sink); run_ticket<WRITE, 4, 0, 2, false>(s, d, d2, big, w, tickets, sink);
        }
        // the static reference points again, in this process
        run<COPY, ROUND_ROBIN, 4, 0, 0>(s, d, d2, big, 2, sink); run<COPY, XCD_RANGES, 4, 0, 0>(s, d, d2, big, 1, sink);
        { float ms = timeit([&] { hipLaunchKernelGGL(naive_copy, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); });
          printf("naive 1 x 16 B per thread, grid %zu: %7.3f ms %5.2f TB/s\n", n4 / 256, ms, 2.0 * big / ms * 1e-9); }
        return 0;
    }
    printf("# 2. walk order x run length x workgroups per CU (copy, plain policy, 9.6 GB)\n");
#define SWEEP_MAP(MAPK) \
    for (int w : {1, 2, 4, 8}) { run<COPY, MAPK, 1, 0, 0>(s, d, d2, big, w, sink); run<COPY, MAPK, 2, 0, 0>(s, d, d2, big, w, sink); \
                                 run<COPY, MAPK, 4, 0, 0>(s, d, d2, big, w, sink); run<COPY, MAPK, 8, 0, 0>(s, d, d2, big, w, sink); }
    SWEEP_MAP(ROUND_ROBIN)
    SWEEP_MAP(XCD_RANGES)
    if (!quick) { SWEEP_MAP(WG_RANGES) }

    printf("# 3. cache-policy bits (copy, round-robin and XCD ranges, 16 KiB runs, 2 WG/CU, 9.6 GB)\n");
#define POL(L, S) run<COPY, XCD_RANGES, 4, L, S>(s, d, d2, big, 2, sink); run<COPY, ROUND_ROBIN, 4, L, S>(s, d, d2, big, 2, sink);
    POL(0, 0) POL(0, 2) POL(0, 16) POL(0, 17) POL(0, 19) POL(0, 1)
    POL(2, 0) POL(2, 2) POL(2, 16) POL(2, 17) POL(2, 19)
    POL(16, 0) POL(16, 2) POL(16, 16) POL(17, 17) POL(19, 19) POL(17, 2) POL(19, 2)

    printf("# 4. read-only / write-only / in-place / 1 : 2, by policy (XCD ranges, 16 KiB, 2 WG/CU, 9.6 GB)\n");
    for (int w : {1, 2, 4}) { run<READ, XCD_RANGES, 4, 0, 0>(s, d, d2, big, w, sink); run<READ, XCD_RANGES, 4, 2, 0>(s, d, d2, big, w, sink); run<READ, XCD_RANGES, 4, 17, 0>(s, d, d2, big, w, sink); }
    for (int w : {1, 2, 4}) { run<WRITE, XCD_RANGES, 4, 0, 0>(s, d, d2, big, w, sink); run<WRITE, XCD_RANGES, 4, 0, 2>(s, d, d2, big, w, sink);
                              run<WRITE, XCD_RANGES, 4, 0, 16>(s, d, d2, big, w, sink); run<WRITE, XCD_RANGES, 4, 0, 17>(s, d, d2, big, w, sink); run<WRITE, XCD_RANGES, 4, 0, 19>(s, d, d2, big, w, sink); }
    for (int w : {1, 2, 4}) { run<INPLACE, XCD_RANGES, 4, 0, 0>(s, s, d2, big, w, sink, "(d == s)"); run<INPLACE, XCD_RANGES, 4, 0, 2>(s, s, d2, big, w, sink, "(d == s)");
                              run<INPLACE, XCD_RANGES, 4, 2, 2>(s, s, d2, big, w, sink, "(d == s)"); run<INPLACE, XCD_RANGES, 4, 0, 17>(s, s, d2, big, w, sink, "(d == s)");
                              run<INPLACE, XCD_RANGES, 4, 17, 17>(s, s, d2, big, w, sink, "(d == s)"); }
    for (int w : {2, 4}) { run<R1W2, XCD_RANGES, 4, 0, 0>(s, d, d2, big, w, sink); run<R1W2, XCD_RANGES, 4, 0, 2>(s, d, d2, big, w, sink); run<R1W2, XCD_RANGES, 4, 2, 2>(s, d, d2, big, w, sink);
                           run<R1W2, XCD_RANGES, 4, 0, 17>(s, d, d2, big, w, sink); run<R1W2, XCD_RANGES, 4, 0, 19>(s, d, d2, big, w, sink); }

    printf("# 5. destination offset modulo the channel interleave (copy, XCD ranges and round-robin, 16 KiB, 2 WG/CU, 9.6 GB)\n");
    for (size_t off : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)3 * 4096, (size_t)65536 + 4096, ((size_t)1 << 20) + 8192, ((size_t)32 << 20) + 16384}) {
        char note[64]; snprintf(note, sizeof note, "(d + %zu B)", off);
        run<COPY, XCD_RANGES, 4, 0, 0>(s, d + off, d2, big, 2, sink, note);
        run<COPY, ROUND_ROBIN, 4, 0, 0>(s, d + off, d2, big, 2, sink, note);
    }

    printf("# 6. buffer size (copy / read / write / in-place, XCD ranges, 16 KiB, 2 WG/CU, plain and nt stores)\n");
    for (size_t bytes : {(size_t)1 << 30, (size_t)4 << 30, big}) {
        run<COPY, XCD_RANGES, 4, 0, 0>(s, d, d2, bytes, 2, sink); run<COPY, XCD_RANGES, 4, 0, 2>(s, d, d2, bytes, 2, sink); run<COPY, ROUND_ROBIN, 4, 0, 0>(s, d, d2, bytes, 2, sink);
        run<READ, XCD_RANGES, 4, 0, 0>(s, d, d2, bytes, 2, sink); run<WRITE, XCD_RANGES, 4, 0, 0>(s, d, d2, bytes, 2, sink); run<INPLACE, XCD_RANGES, 4, 0, 0>(s, s, d2, bytes, 2, sink, "(d == s)");
    }

    printf("# 7. write after a read of the same lines: [read d, then write d] as two launches against a cold write (9.6 GB and 128 MiB = MALL-resident)\n");
    for (size_t bytes : {(size_t)128 << 20, big}) {
        const size_t nch = bytes / 16384;
        float cold = timeit([&] { hipLaunchKernelGGL((stream_k<WRITE, XCD_RANGES, 4, 0, 0>), dim3(512), dim3(256), 0, 0, s, d, d2, nch, sink); });
        float rd = timeit([&] { hipLaunchKernelGGL((stream_k<READ, XCD_RANGES, 4, 0, 0>), dim3(512), dim3(256), 0, 0, d, d, d2, nch, sink); });
        float both = timeit([&] { hipLaunchKernelGGL((stream_k<READ, XCD_RANGES, 4, 0, 0>), dim3(512), dim3(256), 0, 0, d, d, d2, nch, sink);
                                  hipLaunchKernelGGL((stream_k<WRITE, XCD_RANGES, 4, 0, 0>), dim3(512), dim3(256), 0, 0, s, d, d2, nch, sink); });
        printf("%6.2f GB: write alone %7.3f ms (%5.2f TB/s), read alone %7.3f ms (%5.2f TB/s), read then write %7.3f ms (write part %7.3f ms = %5.2f TB/s)\n",
               bytes * 1e-9, cold, bytes / cold * 1e-9, rd, bytes / rd * 1e-9, both, both - rd, bytes / (both - rd) * 1e-9);
    }
    return 0;
}
