// gather_latency.hip -- diagnostic microbenchmark (not part of the library): what one dependent
// "expansion" costs in the memory system under the concurrency of a 1024-query launch, for the
// candidate layouts of the f32 rows.  Every wave runs ITER dependent steps; a step = read one (or two)
// adjacency rows of 32 ids, then fetch the 400-byte vector rows of those ids, in one of several forms.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_latency.hip -o /tmp/gather_latency && /tmp/gather_latency
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

struct Args {
    const uint8_t *rows;   // N x 400 B
    const uint32_t *adj;   // N x 32 ids
    const uint8_t *blocks; // N x 12800 B (inline rows), may be null
    uint32_t N, iters;
    unsigned long long *cyc;  // [nwaves][4]: adjacency wait, rows wait, consume, total
    uint32_t *sink;
};

__device__ __forceinline__ void dma16(const void *gsrc, uint32_t lds_dst) {
    lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// VAR 0: lane per row, 64 rows (two candidates)     1: lane per row, 32 rows (lanes 32..63 idle)
//     2: lane PAIR per row, 32 rows                 3: DMA image (row-major in LDS), 32 rows
//     4: DMA image, 64 rows                         5: contiguous 12.8-KB block to registers
//     6: contiguous block by DMA into LDS + row-lane reads
template <int VAR>
__global__ void __launch_bounds__(64) k_gather(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x;
    uint32_t *ids_lds = reinterpret_cast<uint32_t *>(smem);         // 64 ids
    unsigned char *img = smem + 256;                                 // 64 x 400 B
    const uint32_t img_lds = __builtin_amdgcn_groupstaticsize() + 256;
    uint32_t c = (wave * 2654435761u) % a.N, p = (wave * 40503u + 12345u) % a.N;
    unsigned long long t_adj = 0, t_rows = 0, t_use = 0;
    uint32_t acc = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < a.iters; it++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        uint32_t nb = 0;
        constexpr bool TWO = (VAR == 0 || VAR == 4 || VAR == 8);
        constexpr bool BLOCK = (VAR == 5 || VAR == 6);
        if (!BLOCK) {
            if (TWO || lane < 32 || VAR == 2) {
                const uint32_t cand = (TWO && lane >= 32) ? p : c;
                const uint32_t slot = (VAR == 2) ? (uint32_t)(lane >> 1) : (uint32_t)(lane & 31);
                nb = a.adj[(size_t)cand * 32 + slot];
            }
            wait_vm0();
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        uint32_t x = 0;
        unsigned long long t2;
        if (VAR == 0 || VAR == 1) {
            uint4 w[25];
            if (VAR == 0 || lane < 32) {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.rows + (size_t)nb * 400);
#pragma unroll
                for (int q = 0; q < 25; q++) w[q] = src[q];
            } else {
#pragma unroll
                for (int q = 0; q < 25; q++) w[q] = make_uint4(0, 0, 0, 0);
            }
            wait_vm0();
            t2 = __builtin_readcyclecounter();
#pragma unroll
            for (int q = 0; q < 25; q++) x ^= w[q].x ^ w[q].w;
        } else if (VAR == 2) {
            uint4 w[13];
            const int h = lane & 1;
            const uint4 *src = reinterpret_cast<const uint4 *>(a.rows + (size_t)nb * 400) + 12 * h;
#pragma unroll
            for (int q = 0; q < 13; q++) w[q] = src[q];
            wait_vm0();
            t2 = __builtin_readcyclecounter();
#pragma unroll
            for (int q = 0; q < 13; q++) x ^= w[q].x ^ w[q].w;
        } else if (VAR == 3 || VAR == 4) {
            constexpr int NR = (VAR == 3) ? 32 : 64;
            constexpr int NI = (NR * 25 + 63) / 64;
            ids_lds[lane] = nb;
            wait_lgkm0();
            uint32_t r = (uint32_t)lane / 25u, pc = (uint32_t)lane % 25u;
#pragma unroll
            for (int k = 0; k < NI; k++) {
                const uint32_t rr = r < NR ? r : NR - 1;
                const uint32_t id = ids_lds[rr];
                dma16(a.rows + (size_t)id * 400 + 16 * pc, img_lds + 1024u * k);
                pc += 14;
                r += 2;
                if (pc >= 25) {
                    pc -= 25;
                    r += 1;
                }
            }
            wait_vm0();
            t2 = __builtin_readcyclecounter();
            if (lane < NR) {
                const uint4 *mine = reinterpret_cast<const uint4 *>(img + 400 * lane);
#pragma unroll
                for (int q = 0; q < 25; q++) {
                    const uint4 w = mine[q];
                    x ^= w.x ^ w.w;
                }
            }
            wait_lgkm0();
        } else if (VAR == 7 || VAR == 8) {
            // image order through registers: lane l of load k fetches piece 64 k + l of the row-major image
            // (a row's 25 pieces are contiguous lanes: ~3 rows per instruction instead of 64), writes it to
            // LDS, then the row lanes read their rows back
            constexpr int NR = (VAR == 7) ? 32 : 64;
            constexpr int NI = (NR * 25 + 63) / 64;
            ids_lds[lane] = nb;
            wait_lgkm0();
            uint32_t r = (uint32_t)lane / 25u, pc = (uint32_t)lane % 25u;
            uint4 w[NI];
#pragma unroll
            for (int k = 0; k < NI; k++) {
                const uint32_t rr = r < NR ? r : NR - 1;
                const uint32_t id = ids_lds[rr];
                w[k] = *reinterpret_cast<const uint4 *>(a.rows + (size_t)id * 400 + 16 * pc);
                pc += 14;
                r += 2;
                if (pc >= 25) {
                    pc -= 25;
                    r += 1;
                }
            }
            wait_vm0();
#pragma unroll
            for (int k = 0; k < NI; k++) *reinterpret_cast<uint4 *>(img + 1024 * k + 16 * lane) = w[k];
            wait_lgkm0();
            t2 = __builtin_readcyclecounter();
            if (lane < NR) {
                const uint4 *mine = reinterpret_cast<const uint4 *>(img + 400 * lane);
#pragma unroll
                for (int q = 0; q < 25; q++) {
                    const uint4 ww = mine[q];
                    x ^= ww.x ^ ww.w;
                }
            }
            wait_lgkm0();
        } else if (VAR == 5) {
            uint4 w[13];
            const uint4 *src = reinterpret_cast<const uint4 *>(a.blocks + (size_t)c * 12800);
#pragma unroll
            for (int q = 0; q < 13; q++) {
                const uint32_t o = 64u * q + lane;
                w[q] = src[o < 800 ? o : 799];
            }
            wait_vm0();
            t2 = __builtin_readcyclecounter();
#pragma unroll
            for (int q = 0; q < 13; q++) x ^= w[q].x ^ w[q].w;
        } else {
            const uint8_t *src = a.blocks + (size_t)c * 12800;
#pragma unroll
            for (int q = 0; q < 13; q++) {
                const uint32_t o = 64u * q + lane;
                dma16(src + 16 * (o < 800 ? o : 799), img_lds + 1024u * q);
            }
            wait_vm0();
            t2 = __builtin_readcyclecounter();
            if (lane < 32) {
                const uint4 *mine = reinterpret_cast<const uint4 *>(img + 400 * lane);
#pragma unroll
                for (int q = 0; q < 25; q++) {
                    const uint4 w = mine[q];
                    x ^= w.x ^ w.w;
                }
            }
            wait_lgkm0();
        }
        acc ^= x * 0x9E3779B1u + it;
        // the next candidates depend on what was loaded
        const uint32_t pick = (uint32_t)__shfl((int)acc, (int)((it * 7u) & 31u));
        const uint32_t pick2 = (uint32_t)__shfl((int)acc, (int)(32u + ((it * 5u) & 31u)));
        c = (pick * 2654435761u) % a.N;
        p = (pick2 * 2246822519u + 1u) % a.N;
        const unsigned long long t3 = __builtin_readcyclecounter();
        t_adj += t1 - t0;
        t_rows += t2 - t1;
        t_use += t3 - t2;
    }
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (lane == 0) {
        a.cyc[wave * 4 + 0] = t_adj;
        a.cyc[wave * 4 + 1] = t_rows;
        a.cyc[wave * 4 + 2] = t_use;
        a.cyc[wave * 4 + 3] = t_end - t_begin;
    }
    if (acc == 0x12345678u) a.sink[wave] = acc;
}

template <int VAR>
static void run(const Args &a0, uint32_t nwaves, const char *name, double bytes_per_step) {
    Args a = a0;
    const size_t lds = 256 + 64 * 400;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather<VAR>, dim3(nwaves), dim3(64), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;
        std::vector<unsigned long long> c(nwaves * 4);
        CK(hipMemcpy(c.data(), a.cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        for (uint32_t w = 0; w < nwaves; w++)
            for (int j = 0; j < 4; j++) s[j] += (double)c[w * 4 + j];
        for (int j = 0; j < 4; j++) s[j] /= (double)nwaves * a.iters;
        printf("%-44s waves %5u: %.3f ms, %6.2f us/step; cycles/step: adj %6.0f rows %6.0f use %5.0f total %6.0f; %.2f TB/s\n",
               name, nwaves, ms, ms * 1e3 / a.iters, s[0], s[1], s[2], s[3],
               bytes_per_step * nwaves * a.iters / (ms * 1e-3) / 1e12);
    }
}

int main(int argc, char **argv) {
    const uint32_t N = argc > 2 ? (uint32_t)atoi(argv[2]) : 1000000, ITERS = 60;
    const bool with_blocks = argc < 2 || atoi(argv[1]) != 0;
    uint8_t *rows;
    uint32_t *adj;
    uint8_t *blocks = nullptr;
    CK(hipMalloc(&rows, (size_t)N * 400));
    CK(hipMalloc(&adj, (size_t)N * 32 * 4));
    if (with_blocks) CK(hipMalloc(&blocks, (size_t)N * 12800));
    {
        std::vector<uint32_t> h((size_t)N * 100);
        uint64_t s = 88172645463325252ull;
        for (auto &x : h) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            x = (uint32_t)s;
        }
        CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> g((size_t)N * 32);
        for (auto &x : g) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            x = (uint32_t)(s % N);
        }
        CK(hipMemcpy(adj, g.data(), g.size() * 4, hipMemcpyHostToDevice));
        if (blocks) {
            for (size_t off = 0; off < (size_t)N * 12800; off += h.size() * 4) {
                const size_t n = std::min(h.size() * 4, (size_t)N * 12800 - off);
                CK(hipMemcpy(blocks + off, h.data(), n, hipMemcpyHostToDevice));
            }
        }
    }
    Args a{};
    a.rows = rows;
    a.adj = adj;
    a.blocks = blocks;
    a.N = N;
    a.iters = ITERS;
    CK(hipMalloc(&a.cyc, 8192 * 4 * 8));
    CK(hipMalloc(&a.sink, 8192 * 4));
    printf("table of %u rows\n", N);
    for (uint32_t nw : {256u, 1024u, 2048u}) {
        run<0>(a, nw, "0 lane/row, 64 rows (2 cand)", 64 * 400 + 256);
        run<1>(a, nw, "1 lane/row, 32 rows", 32 * 400 + 128);
        run<2>(a, nw, "2 lane-pair/row, 32 rows", 32 * 400 + 128);
        run<3>(a, nw, "3 DMA image, 32 rows + LDS reads", 32 * 400 + 128);
        run<4>(a, nw, "4 DMA image, 64 rows + LDS reads", 64 * 400 + 256);
        run<7>(a, nw, "7 image order via registers, 32 rows", 32 * 400 + 128);
        run<8>(a, nw, "8 image order via registers, 64 rows", 64 * 400 + 256);
        if (blocks) {
            run<5>(a, nw, "5 contiguous 12.8 KB block -> regs", 12800);
            run<6>(a, nw, "6 contiguous block DMA -> LDS + reads", 12800);
        }
    }
    return 0;
}
