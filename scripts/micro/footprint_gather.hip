// footprint_gather.hip -- diagnostic microbenchmark (not part of the library): the rate of scattered
// whole-row reads against the size of the table they are scattered over, by ACCESS SHAPE.
//   lane-per-row   every lane reads its own row, 16 bytes per load: one wave instruction touches 64
//                  different cache lines for 16 bytes each (the shape of the search kernels' wide-row loop)
//   group-per-row  LPR lanes read LPR x 16 contiguous bytes of one row per instruction: 8 lanes = one
//                  128-byte line per group, 64 lanes = 1 KiB of one row per instruction (wave-cooperative)
// Every variant reads the same rows (64 pseudo-random rows per wave and iteration, whole rows).  Run under
// rocprofv3 --pmc FETCH_SIZE to see what each shape makes the L2 fetch (the kernel names carry the shape).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/footprint_gather.hip -o /tmp/footprint_gather
//   /tmp/footprint_gather [max_GB] [min_GB] [iters]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const uint8_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return *reinterpret_cast<const u32x4 *>(p);
}

__device__ __forceinline__ uint64_t next_row(uint64_t &x, uint64_t nrows) {
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return x % nrows;
}

// lane-per-row: the old shape.  IN_FLIGHT pieces of the lane's row are requested before they are used.
template <int ROW, bool NT>
__global__ void __launch_bounds__(64) k_lane_per_row(const uint8_t *tab, uint64_t nrows, uint32_t iters, uint32_t *sink) {
    constexpr int P = ROW / 16;
    constexpr int F = P < 32 ? P : 32;
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint64_t x = 0x9E3779B97F4A7C15ull * (gid + 1);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint8_t *src = tab + next_row(x, nrows) * ROW;
        for (int p0 = 0; p0 < P; p0 += F) {
            u32x4 w[F];
#pragma unroll
            for (int p = 0; p < F; p++) w[p] = ld16<NT>(src + (p0 + p) * 16);
#pragma unroll
            for (int p = 0; p < F; p++) acc += w[p].x ^ w[p].y ^ w[p].z ^ w[p].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

constexpr int pow2_le(int v) { int r = 1; while (r * 2 <= v) r *= 2; return r; }

// group-per-row: LPR lanes share a row.  The wave still draws 64 rows per iteration (one per lane) and
// hands them out: round j gives group g the row drawn by lane j * G + g.
template <int ROW, int LPR, bool NT>
__global__ void __launch_bounds__(64) k_group_per_row(const uint8_t *tab, uint64_t nrows, uint32_t iters, uint32_t *sink) {
    constexpr int G = 64 / LPR;              // groups per wave
    constexpr int SEG = ROW / (16 * LPR);    // instructions per row
    static_assert(SEG >= 1, "row shorter than one group instruction");
    constexpr int ROUNDS = 64 / G;           // = LPR
    constexpr int F = 32;                    // loads in flight per lane
    constexpr int RPB = pow2_le(F / SEG);    // rounds per batch of loads (a power of two: divides ROUNDS)
    const uint32_t lane = threadIdx.x;
    const uint32_t gid = blockIdx.x * 64 + lane;
    const uint32_t g = lane / LPR, l = lane % LPR;
    uint64_t x = 0x9E3779B97F4A7C15ull * (gid + 1);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint64_t mine = next_row(x, nrows);
        const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
        for (int j0 = 0; j0 < ROUNDS; j0 += RPB) {
            u32x4 w[RPB * SEG];
#pragma unroll
            for (int jj = 0; jj < RPB; jj++) {
                const int srcl = (j0 + jj) * G + g;
                const uint64_t row = ((uint64_t)__shfl(hi, srcl) << 32) | __shfl(lo, srcl);
                const uint8_t *src = tab + row * ROW + l * 16;
#pragma unroll
                for (int s = 0; s < SEG; s++) w[jj * SEG + s] = ld16<NT>(src + s * (16 * LPR));
            }
#pragma unroll
            for (int k = 0; k < RPB * SEG; k++) acc += w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K>
static void time_kernel(K kern, const char *what, const uint8_t *tab, uint64_t bytes, int row, uint32_t waves, uint32_t iters, uint32_t *sink) {
    const uint64_t nrows = bytes / row;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, tab, nrows, 4u, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, tab, nrows, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = (double)waves * 64 * iters * row / 1e9;
    printf("table %7.2f GB, rows of %4d B, %5u waves, %-28s: %8.3f ms  %7.0f GB/s  (%.1f MB read)\n", bytes / 1e9, row, waves, what, ms, gb / (ms / 1e3), gb * 1e3);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const double max_gb = argc > 1 ? atof(argv[1]) : 64.0;
    const double min_gb = argc > 2 ? atof(argv[2]) : 0.0;
    const uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 32u;
    uint32_t *sink;
    CK(hipMalloc(&sink, 4));
    for (double gbs : {0.25, 1.0, 4.0, 16.0, 64.0}) {
        if (gbs > max_gb || gbs < min_gb) continue;
        const uint64_t bytes = (uint64_t)(gbs * (1ull << 30));
        uint8_t *tab;
        CK(hipMalloc(&tab, bytes));
        CK(hipMemset(tab, 1, bytes));
        CK(hipDeviceSynchronize());
        for (uint32_t waves : {1024u, 8192u}) {
#define RUN(K, WHAT, ROW) time_kernel(K, WHAT, tab, bytes, ROW, waves, iters, sink)
            RUN((k_lane_per_row<512, false>), "lane per row", 512);
            RUN((k_lane_per_row<512, true>), "lane per row, nt", 512);
            RUN((k_group_per_row<512, 8, false>), "8 lanes per row (128-B line)", 512);
            RUN((k_group_per_row<512, 8, true>), "8 lanes per row, nt", 512);
            RUN((k_group_per_row<512, 32, false>), "32 lanes per row (whole row)", 512);
            RUN((k_group_per_row<512, 32, true>), "32 lanes per row, nt", 512);
            RUN((k_lane_per_row<3072, false>), "lane per row", 3072);
            RUN((k_lane_per_row<3072, true>), "lane per row, nt", 3072);
            RUN((k_group_per_row<3072, 8, false>), "8 lanes per row (128-B line)", 3072);
            RUN((k_group_per_row<3072, 8, true>), "8 lanes per row, nt", 3072);
            RUN((k_group_per_row<3072, 64, false>), "64 lanes per row (1 KiB)", 3072);
            RUN((k_group_per_row<3072, 64, true>), "64 lanes per row, nt", 3072);
#undef RUN
        }
        CK(hipFree(tab));
    }
    return 0;
}
