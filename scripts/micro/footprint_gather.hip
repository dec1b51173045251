// footprint_gather.hip -- diagnostic microbenchmark (not part of the library): the rate of scattered
// row reads against the size of the table they are scattered over.  Every lane reads whole rows
// (ROW bytes, 16 bytes per load, all loads of a row in flight) at pseudo-random row indices; the grid
// fills the chip several waves deep.  Answers whether the ~1.3 TB/s the search kernels see on 30-65 GB
// indexes is the memory system's or theirs.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/footprint_gather.hip -o /tmp/footprint_gather && /tmp/footprint_gather
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

template <int ROW, bool NT>
__global__ void __launch_bounds__(64) k_gather(const uint8_t *tab, uint64_t nrows, uint32_t iters, uint32_t *sink) {
    constexpr int P = ROW / 16;
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint64_t x = 0x9E3779B97F4A7C15ull * (gid + 1);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        const uint64_t row = x % nrows;
        const uint4 *src = reinterpret_cast<const uint4 *>(tab + row * ROW);
        uint4 w[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            if (NT) {  // non-temporal: the row is not going to be read again
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + p));
                w[p] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                w[p] = src[p];
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++) acc += w[p].x ^ w[p].y ^ w[p].z ^ w[p].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int ROW, bool NT>
static void run(const uint8_t *tab, uint64_t bytes, uint32_t waves, uint32_t *sink) {
    const uint64_t nrows = bytes / ROW;
    const uint32_t iters = 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<ROW, NT>), dim3(waves), dim3(64), 0, 0, tab, nrows, 8u, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<ROW, NT>), dim3(waves), dim3(64), 0, 0, tab, nrows, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = (double)waves * 64 * iters * ROW / 1e9;
    printf("table %7.2f GB, rows of %4d B, %5u waves%s: %8.3f ms  %7.0f GB/s\n", bytes / 1e9, ROW, waves, NT ? ", non-temporal loads" : "", ms, gb / (ms / 1e3));
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double max_gb = argc > 1 ? atof(argv[1]) : 64.0;
    uint32_t *sink;
    CK(hipMalloc(&sink, 4));
    for (double gbs : {0.25, 1.0, 4.0, 16.0, 64.0}) {
        if (gbs > max_gb) break;
        const uint64_t bytes = (uint64_t)(gbs * (1ull << 30));
        uint8_t *tab;
        CK(hipMalloc(&tab, bytes));
        CK(hipMemset(tab, 1, bytes));
        CK(hipDeviceSynchronize());
        for (uint32_t waves : {1024u, 8192u}) {
            run<512, false>(tab, bytes, waves, sink);
            run<512, true>(tab, bytes, waves, sink);
            run<3072, false>(tab, bytes, waves, sink);
            run<3072, true>(tab, bytes, waves, sink);
        }
        CK(hipFree(tab));
    }
    return 0;
}
