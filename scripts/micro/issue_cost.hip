// issue_cost.hip -- diagnostic microbenchmark (not part of the library): what one instruction of the
// search loop costs a wave that is ALONE on its SIMD (a 1024-query launch), in cycles, by kind.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/issue_cost.hip -o scripts/micro/bin/issue_cost
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

struct Out {
    unsigned long long cyc[32];
};

__device__ __forceinline__ unsigned long long now() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

__global__ void __launch_bounds__(64) k_issue(Out *out, float *sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    float a0 = lane * 0.5f, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float b0 = 1.5f, b1 = 2.5f;
    unsigned long long t[32];
    int k = 0;
#define MEASURE(body)                         \
    {                                         \
        const unsigned long long s0 = now();  \
        for (int it = 0; it < iters; it++) {  \
            body                              \
        }                                     \
        t[k++] = now() - s0;                  \
    }
    // 0: dependent v_add_f32 chain
    MEASURE(asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(a0) : "v"(b0));)
    // 1: independent v_add_f32 (8 accumulators)
    MEASURE(asm volatile(REP16("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0));)
    // 2: independent v_pk_add_f32
    {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
        MEASURE(asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                             : "v"(q));)
        // 3: independent v_pk_mul_f32
        MEASURE(asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                             : "v"(q));)
        // 4: the distance chain pattern, packed: pk_add, pk_mul, add, add (16 element pairs)
        MEASURE(asm volatile(REP16("v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %1, %1, %1\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %0, %0, %5\n\t")
                             : "+v"(a0), "+v"(p0)
                             : "v"(p1), "v"(q), "v"(a6), "v"(a7));)
        // 5: the same, scalar: sub, sub, mul, mul, add, add
        MEASURE(asm volatile(REP16("v_sub_f32 %1, %3, %5\n\tv_sub_f32 %2, %4, %5\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2)
                             : "v"(a3), "v"(a4), "v"(b0));)
        a1 += p0.x + p1.y + p2.x + p3.y;
    }
    // 6: v_cmp_lt_u64 + s_bcnt1 (ballot + popcount)
    {
        unsigned long long x = (unsigned long long)lane * 77ull, y = 1000ull;
        uint32_t cnt = 0;
        MEASURE(asm volatile(REP16("v_cmp_lt_u64 vcc, %1, %2\n\ts_bcnt1_i32_b64 s20, vcc\n\ts_add_u32 %0, %0, s20\n\t")
                             : "+s"(cnt)
                             : "v"(x), "v"(y)
                             : "vcc", "s20", "scc");)
        // 7: v_cmp_lt_u32 + s_bcnt1
        uint32_t x32 = lane * 77u, y32 = 1000u;
        MEASURE(asm volatile(REP16("v_cmp_lt_u32 vcc, %1, %2\n\ts_bcnt1_i32_b64 s20, vcc\n\ts_add_u32 %0, %0, s20\n\t")
                             : "+s"(cnt)
                             : "v"(x32), "v"(y32)
                             : "vcc", "s20", "scc");)
        a2 += (float)cnt;
    }
    // 8: v_readlane_b32 x 64
    {
        uint32_t v = lane * 3u, s = 0;
        MEASURE(asm volatile(REP64("v_readlane_b32 s20, %1, 5\n\t") "s_mov_b32 %0, s20\n\t" : "=s"(s) : "v"(v) : "s20");)
        a3 += (float)s;
    }
    // 9: SALU chain: 64 dependent s_add
    {
        uint32_t s = 1;
        MEASURE(asm volatile(REP64("s_add_u32 %0, %0, 3\n\t") : "+s"(s) : : "scc");)
        a3 += (float)s;
    }
    // 10: v_cndmask x 64 (independent-ish)
    {
        uint32_t v = lane, w = lane * 2;
        MEASURE(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\t" REP64("v_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(v) : "v"(w) : "vcc");)
        a4 += (float)v;
    }
    // 11: v_mov_b32 dpp wave_shr:1 x 64
    {
        uint32_t v = lane, w = lane * 2;
        MEASURE(asm volatile(REP64("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(v) : "v"(w));)
        a4 += (float)v;
    }
    // 12: ds_read_b128 dependent round trips x 16
    {
        uint32_t addr = (lane * 16) & 0x3FF0;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 r;
        MEASURE(asm volatile(REP16("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x3ff0, %1\n\t")
                             : "=&v"(r), "+v"(addr)::"memory");
                a5 += (float)r.x;)
    }
    // 13: ds_cmpst_rtn_b32 dependent round trips x 16
    {
        uint32_t addr = (lane * 4) & 0x3FFC, cmp = 0xFFFFFFFFu, val = lane, old = 0;
        MEASURE(asm volatile(REP16("ds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\t")
                             : "=&v"(old)
                             : "v"(addr), "v"(cmp), "v"(val)
                             : "memory");
                a5 += (float)old;)
    }
    // 14: ds_write_b64 + ds_read_b64 round trip x 16
    {
        uint32_t addr = (lane * 8) & 0x3FF8;
        unsigned long long v = lane, r = 0;
        MEASURE(asm volatile(REP16("ds_write_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t")
                             : "=&v"(r)
                             : "v"(addr), "v"(v)
                             : "memory");
                a6 += (float)r;)
    }
    // 15: v_sqrt_f32 x 16 (the fix-up sequence around it is ordinary VALU)
    MEASURE(asm volatile(REP16("v_sqrt_f32 %0, %0\n\t") : "+v"(a7));)
    // 16: s_and_saveexec / s_or exec pair x 16 (the cost of a predicated block)
    {
        uint32_t v = lane;
        MEASURE(asm volatile(REP16("v_cmp_gt_u32 vcc, 32, %0\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, 1, %0\n\ts_or_b64 exec, exec, s[20:21]\n\t")
                             : "+v"(v)::"vcc", "s20", "s21", "scc");)
        a6 += (float)v;
    }
    // 17: s_cbranch taken x 16 (wave-uniform branch over nothing)
    {
        uint32_t s = 0;
        MEASURE(asm volatile(REP16("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : "+s"(s)::"scc");)
    }
    // 18: v_cndmask independent destinations, vcc
    {
        uint32_t v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, w = lane * 2;
        MEASURE(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\t" REP16("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc\n\t")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w) : "vcc");)
        a4 += (float)(v0 + v1 + v2 + v3);
    }
    // 19: v_cndmask with an SGPR-pair condition (e64), independent destinations
    {
        uint32_t v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, w = lane * 2;
        MEASURE(asm volatile("v_cmp_lt_u32 s[20:21], %0, %4\n\t" REP16("v_cndmask_b32 %0, %0, %4, s[20:21]\n\tv_cndmask_b32 %1, %1, %4, s[20:21]\n\tv_cndmask_b32 %2, %2, %4, s[20:21]\n\tv_cndmask_b32 %3, %3, %4, s[20:21]\n\t")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w) : "s20", "s21");)
        a4 += (float)(v0 + v1 + v2 + v3);
    }
    // 20: v_cmp + v_cndmask pairs (compare writes vcc, select reads it)
    {
        uint32_t v0 = lane, w = lane * 2;
        MEASURE(asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(v0) : "v"(w) : "vcc");)
        a4 += (float)v0;
    }
    // 21: v_cmp_lt_u64 alone (no scalar consumer)
    {
        unsigned long long x = (unsigned long long)lane * 77ull, y = 1000ull;
        MEASURE(asm volatile(REP64("v_cmp_lt_u64 vcc, %0, %1\n\t") : : "v"(x), "v"(y) : "vcc");)
    }
    // 22: v_cmp_lt_u32 alone
    {
        uint32_t x = lane * 77u, y = 1000u;
        MEASURE(asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n\t") : : "v"(x), "v"(y) : "vcc");)
    }
    // 23: v_cmp -> s_and_b64 -> v_cndmask (mask logic on the scalar side between compare and select)
    {
        uint32_t v0 = lane, w = lane * 2;
        MEASURE(asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 s[20:21], vcc, exec\n\tv_cndmask_b32 %0, %0, %1, s[20:21]\n\t") : "+v"(v0) : "v"(w) : "vcc", "s20", "s21", "scc");)
        a4 += (float)v0;
    }
    // 24: v_readlane -> s_cmp -> s_cselect (a scalar decision on a lane's value)
    {
        uint32_t v = lane * 3u, s = 0;
        MEASURE(asm volatile(REP16("v_readlane_b32 s20, %1, 5\n\ts_cmp_eq_u32 s20, 15\n\ts_cselect_b32 %0, 1, 2\n\t") : "+s"(s) : "v"(v) : "s20", "scc");)
        a3 += (float)s;
    }
    // 25: v_and_b32 / v_or_b32 / v_lshl (plain integer VALU, dependent)
    {
        uint32_t v = lane;
        MEASURE(asm volatile(REP64("v_xor_b32 %0, 0x55, %0\n\t") : "+v"(v));)
        a4 += (float)v;
    }
    // 26: two independent v_cmp then two s_bcnt1 (overlapped scalar consumers)
    {
        uint32_t x = lane * 77u, y = 1000u, cnt = 0;
        MEASURE(asm volatile(REP16("v_cmp_lt_u32 s[20:21], %1, %2\n\tv_cmp_gt_u32 s[22:23], %1, %2\n\ts_bcnt1_i32_b64 s24, s[20:21]\n\ts_bcnt1_i32_b64 s25, s[22:23]\n\ts_add_u32 %0, %0, s24\n\ts_add_u32 %0, %0, s25\n\t")
                             : "+s"(cnt) : "v"(x), "v"(y) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");)
        a2 += (float)cnt;
    }
    // 27: uniform branch NOT taken
    {
        uint32_t s = 1;
        MEASURE(asm volatile(REP16("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : "+s"(s)::"scc");)
    }
    if (lane == 0 && blockIdx.x == 0)
        for (int i = 0; i < k; i++) out->cyc[i] = t[i];
    sink[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    Out *d_out;
    float *d_sink;
    const int nb = 1024, iters = 20;
    CK(hipMalloc(&d_out, sizeof(Out)));
    CK(hipMalloc(&d_sink, nb * 64 * 4));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_issue, dim3(nb), dim3(64), 0, 0, d_out, d_sink, iters);
        CK(hipDeviceSynchronize());
    }
    Out h;
    CK(hipMemcpy(&h, d_out, sizeof(Out), hipMemcpyDeviceToHost));
    const char *names[] = {"v_add_f32 dependent chain", "v_add_f32 independent", "v_pk_add_f32 independent", "v_pk_mul_f32 independent",
                           "chain packed (pk_add,pk_mul,add,add = 2 elements)", "chain scalar (6 instr = 2 elements)",
                           "v_cmp_lt_u64 + s_bcnt1 + s_add", "v_cmp_lt_u32 + s_bcnt1 + s_add", "v_readlane_b32", "s_add_u32 dependent",
                           "v_cndmask_b32", "v_mov_b32_dpp wave_shr:1", "ds_read_b128 round trip", "ds_cmpst_rtn_b32 round trip",
                           "ds_write_b64 + ds_read_b64 round trips", "v_sqrt_f32", "saveexec block (cmp, saveexec, add, or)",
                           "uniform branch taken (cmp, cbranch)", "v_cndmask vcc, independent", "v_cndmask s[pair], independent",
                           "v_cmp + v_cndmask pair", "v_cmp_lt_u64 alone", "v_cmp_lt_u32 alone", "v_cmp, s_and_b64, v_cndmask", "v_readlane, s_cmp, s_cselect",
                           "v_xor_b32 dependent", "2 x v_cmp then 2 x s_bcnt1 + 2 s_add", "uniform branch not taken (cmp, cbranch, nop)"};
    const int per_iter[] = {64, 64, 64, 64, 16, 16, 16, 16, 64, 64, 64, 64, 16, 16, 16, 16, 16, 16, 64, 64, 16, 64, 64, 16, 16, 64, 16, 16};
    for (int i = 0; i < 28; i++)
        printf("%-52s %8.1f cycles per unit (%d units x %d iterations)\n", names[i], (double)h.cyc[i] / (per_iter[i] * iters), per_iter[i], iters);
    return 0;
}
