// search_kernels.hip -- the HNSW search hot path as hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What runs here replaces, for a whole batch of queries at once,
//   HNSW::ann_by_vector          hnsw/src/template.rs:306-335
//   Searcher::search_layer       hnsw/src/template/searcher.rs:23-103
//   Results (ordered sets)       hnsw/src/template/results.rs:26-33,96-116,148-180
//   QuantVec::new / distance_unrolled   vectors/src/quant.rs:41-66,14-37
//   FullVec::distance            vectors/src/full.rs:23-29
//   Dist ordering                graph/src/dist.rs:30-38
// and returns bit-identical ids (and distances) for the same index and query.
//
// Execution model: ONE 64-lane wavefront per query (one single-wave workgroup), everything a
// query needs while it runs lives on chip:
//   - `selected` and `candidates` (two BTreeSets in the reference) collapse into ONE sorted list
//     of <= ef keys with an "expanded" bit: every element enters both sets together
//     (searcher.rs:79-80,86-87), leaves `candidates` only by being expanded, and an element
//     evicted from `selected` is > selected.last() forever, so popping it could only hit the
//     `break` (searcher.rs:41-44).  key = dist_bits << 32 | id orders exactly like Dist::cmp for
//     the non-negative, non-NaN distances a sqrt produces; bit 63 (never set by such a float) is
//     the expanded flag.  The list is held in registers (lane l, register r = list[64 r + l]) and
//     re-sorted through an LDS permutation buffer when a batch of neighbours is merged.
//   - `visited` (an IntSet, cleared per layer, searcher.rs:101) is an open-addressing hash table
//     in LDS filled with ds_cmpst (atomicCAS).
//   - the (dequantised) query is staged in LDS / registers once.
// Per expansion the wave loads one adjacency row (one coalesced 128-B load at m = 16), filters
// it through the visited table, gathers the vector rows of the fresh neighbours (QUANT8: a lane
// PAIR per neighbour, each lane streaming 4 of distance_unrolled's 8 running sums from its own
// contiguous half row; F32: one lane per neighbour because FullVec's sum is one sequential
// chain), and merges the batch into the list by rank (ballot + popcount), which is the
// reference's streaming top-ef (searcher.rs:74-94) evaluated for the whole batch at once: the
// final `selected` does not depend on the order in which a batch is applied.
//
// Float fidelity: compiled with -ffp-contract=off; no FMA may fuse `code * delta + min` or
// `acc += t * t`; sqrt and the quantiser's division are the correctly rounded forms hipcc emits
// by default.  Accumulation order is the reference's (quant.rs:23-36, full.rs:24-28).

#include <cstdlib>

#include "device_index.h"
#include "search_common.h"

namespace hx {

static constexpr u64 KEY_INVALID = ~0ull;
static constexpr u64 KEY_MASK = 0x7FFFFFFFFFFFFFFFull;  // drops the expanded flag
static constexpr u64 KEY_EXPANDED = 1ull << 63;

#define HX_MAX_R 8  // ef <= 64 * HX_MAX_R on the specialised kernels and in the on-device build
#define HX_MAX_R_WIDE 16  // ef <= 1024 on the any-dimension search kernel

// Diagnostic build only (make stamps): per-phase cycle shares of the inline-rows expansion loop,
// written to a side buffer nothing else reads.  The shipped library is built without HX_STAMPS.
#ifdef HX_STAMPS
#define STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(slot, a, b) dbg_acc[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

// workgroup barrier that does not drain VMEM (LDS-DMA prefetches stay in flight across it)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// value held by lane q (0..3) of this lane's quad
template <int Q>
__device__ __forceinline__ float quad_bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x),
                                                                  Q * 0x55, 0xF, 0xF, true));
}
template <int Q>
__device__ __forceinline__ int quad_bcast_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, Q * 0x55, 0xF, 0xF, true);
}
// ---------------------------------------------------------------------------------------------
// Four lanes per candidate (two-wave kernel): lane (h, sub) of the quad owns running sums
// 4h + 2 sub and 4h + 2 sub + 1 of distance_unrolled, i.e. bytes 2 sub and 2 sub + 1 of every chunk
// dword of half h; the d % 8 tail belongs to lane (0, 0) alone.  qc[2 c + kk] is this lane's query
// value for chunk dword c, byte kk; qt[r] the tail values (lane (0,0) only).
// ---------------------------------------------------------------------------------------------
template <int P, int DS, typename QC, typename QT>
__device__ __forceinline__ void quant_pair_sums(const uint4 (&w)[P], const QC &qc, const QT &qt,
                                                int h, int sub, uint32_t nch4, uint32_t rem,
                                                float (&acc)[2]) {
    const float mn = __builtin_bit_cast(float, w[0].x);
    const float delta = __builtin_bit_cast(float, w[0].y);
    const bool tail_lane = (h == 0) && (sub == 0);
    const uint32_t sh = 16u * (uint32_t)sub;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (p == 0 && j < 2) continue;  // header
            const int e0 = 16 * p + 4 * j - 8;  // element index of byte 0 of this dword
            const int c = e0 / 4;               // chunk dword number
            bool is_chunk, is_tail;
            if (DS > 0) {
                constexpr int N4 = 4 * (DS / 8), RM = DS % 8;
                is_chunk = e0 < N4;
                is_tail = !is_chunk && e0 < N4 + RM;
                if (!is_chunk && !is_tail) continue;
            } else {
                is_chunk = (uint32_t)e0 < nch4;
                is_tail = !is_chunk && (uint32_t)e0 < nch4 + rem;
            }
            if (DS > 0 ? is_chunk : true) {
                const uint32_t u = dw[j] >> sh;
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const float x = ((float)((u >> (8 * kk)) & 0xFFu) * delta) + mn;
                    const float t = x - qc[2 * c + kk];
                    const float t2 = t * t;
                    acc[kk] += (DS > 0 || is_chunk) ? t2 : 0.0f;
                }
            }
            if (DS > 0 ? is_tail : true) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = e0 + k;
                    bool in;
                    if (DS > 0) {
                        constexpr int N4 = 4 * (DS / 8), RM = DS % 8;
                        if (e >= N4 + RM) continue;
                        in = true;
                    } else {
                        in = is_tail && (uint32_t)e < nch4 + rem;
                    }
                    const float x = ((float)((dw[j] >> (8 * k)) & 0xFFu) * delta) + mn;
                    const int r = DS > 0 ? e - 4 * (DS / 8) : (in ? e - (int)nch4 : 0);
                    const float t = x - qt[r];
                    const float t2 = t * t;
                    acc[0] += (in && tail_lane) ? t2 : 0.0f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage a STORED point as the query (build path: Point::dist2other between two stored points,
// points/src/points.rs:86-93).  QUANT8: the packed row already is in the half-row element order, so
// yq[h * nq_half + e] = code * delta + min straight from the row.  F32: the row's floats.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void stage_row(const DevView &v, uint32_t id, float *yq, int lane) {
    if (KIND == HNSW_VEC_QUANT8) {
        const uint32_t nq_half = v.half_bytes - 8;
        const uint8_t *row = v.rows + (size_t)id * v.row_stride;
        for (uint32_t i = lane; i < 2 * nq_half; i += 64) {
            const uint32_t hh = i >= nq_half ? 1u : 0u, e = i - hh * nq_half;
            const uint8_t *half = row + hh * v.half_bytes;
            const float mn = *reinterpret_cast<const float *>(half);
            const float delta = *reinterpret_cast<const float *>(half + 4);
            const bool used = e < v.nch4 || (hh == 0 && e < v.nch4 + v.rem);
            yq[i] = used ? ((float)half[8 + e] * delta) + mn : 0.0f;
        }
    } else {
        const float *row = reinterpret_cast<const float *>(v.rows + (size_t)id * v.row_stride);
        for (uint32_t e = lane; e < v.dim; e += 64) yq[e] = row[e];
    }
    wave_fence();
}

// ---------------------------------------------------------------------------------------------
// QUANT8, any dimension: the pieces of a half row that hold nothing but chunk elements, without
// per-element predicates, in stages of CH 16-byte pieces through two register buffers (the next
// stage in flight while the current one is summed).  Starts at piece `first` >= 1 (piece 0 carries the
// header); returns the number of pieces consumed (a multiple of CH).  Element e of the half sits at
// byte 8 + e.
// ---------------------------------------------------------------------------------------------
template <int CH>
__device__ __forceinline__ uint32_t quant_bulk_stages(const uint4 *src, const float *yh, float delta, float mn,
                                                      uint32_t first, uint32_t n_pure, float (&acc)[4]) {
    const uint32_t nst = n_pure / CH;
    if (nst == 0) return 0;
    uint4 a[CH], b[CH];
    auto fetch_at = [&](uint4 (&w)[CH], uint32_t st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; i++) w[i] = src[first + st * CH + i];
    };
    auto consume_at = [&](const uint4 (&w)[CH], uint32_t st) __attribute__((always_inline)) {
        const float *y = yh + 16 * (first + st * CH) - 8;  // query value of the stage's first element
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const uint32_t dw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const f32x2 c = {(float)((dw[j] >> (8 * k)) & 0xFFu), (float)((dw[j] >> (8 * (k + 1))) & 0xFFu)};
                    const f32x2 x = c * delta + mn;
                    const f32x2 yy = {y[16 * i + 4 * j + k], y[16 * i + 4 * j + k + 1]};
                    const f32x2 t = x - yy;
                    const f32x2 t2 = t * t;
                    acc[k] += t2.x;
                    acc[k + 1] += t2.y;
                }
            }
        }
    };
    fetch_at(a, 0);
#pragma unroll 1
    for (uint32_t st = 0; st < nst; st += 2) {
        if (st + 1 < nst) fetch_at(b, st + 1);
        consume_at(a, st);
        if (st + 2 < nst) fetch_at(a, st + 2);
        if (st + 1 < nst) consume_at(b, st + 1);
    }
    return nst * CH;
}

// ---------------------------------------------------------------------------------------------
// Distance of one stored point to the staged query for ANY dimension (runtime loops): used by
// the test-seam and brute-force kernels, and by the search kernel when no specialised variant
// fits.  QUANT8: valid on the even lane of the pair; F32: per lane.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float dist_any_dim(const DevView &v, uint32_t id, bool active, int h,
                                              const float *yq) {
    if (KIND == HNSW_VEC_QUANT8) {
        const float *yh = yq + h * (v.half_bytes - 8);
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) {
            const uint4 *src = reinterpret_cast<const uint4 *>(
                v.rows + (size_t)id * v.row_stride + (size_t)h * v.half_bytes);
            const uint32_t np = v.half_bytes >> 4;
            const uint4 w0 = src[0];
            const float mn = __builtin_bit_cast(float, w0.x);
            const float delta = __builtin_bit_cast(float, w0.y);
            // Pieces [p_lo, p_hi) through the predicated element loop, 4 pieces (64 B) per group.
            auto consume_pred = [&](uint32_t p_lo, uint32_t p_hi) __attribute__((always_inline)) {
                for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 4) {
                    uint4 w[4];
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        w[p] = (p0 + p < p_hi) ? src[p0 + p] : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        if (p0 + p >= p_hi) continue;  // wave-uniform: a piece outside the range costs nothing
                        const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int e = 16 * (int)(p0 + p) + 4 * j + k - 8;
                                const float x = ((float)((dw[j] >> (8 * k)) & 0xFFu) * delta) + mn;
                                const bool chunk = e >= 0 && (uint32_t)e < v.nch4;
                                const bool tail = e >= 0 && !chunk &&
                                                  (uint32_t)e < v.nch4 + v.rem && h == 0;
                                const float y = (chunk || tail) ? yh[e] : 0.0f;
                                const float t = x - y;
                                const float t2 = t * t;
                                if (k == 0) {
                                    acc[0] += (chunk || tail) ? t2 : 0.0f;
                                } else {
                                    acc[k] += chunk ? t2 : 0.0f;
                                    acc[0] += tail ? t2 : 0.0f;
                                }
                            }
                        }
                    }
                }
            };
            // Pieces 1 .. that hold nothing but chunk elements go through the predicate-free stages
            // (quant_bulk_stages); piece 0 (header) before, the remainder after -- every running sum
            // still sees its elements in ascending order.
            const uint32_t n_pure = v.nch4 >= 24 ? (v.nch4 - 8) / 16 : 0;  // pieces [1, 1 + n_pure)
            if (n_pure < 4) {
                consume_pred(0, np);
            } else {
                consume_pred(0, 1);
                // widest stages first, then narrower ones over what is left of the pure pieces
                uint32_t used = 0;
                if (n_pure >= 16) used += quant_bulk_stages<8>(src, yh, delta, mn, 1, n_pure, acc);
                used += quant_bulk_stages<4>(src, yh, delta, mn, 1 + used, n_pure - used, acc);
                used += quant_bulk_stages<1>(src, yh, delta, mn, 1 + used, n_pure - used, acc);
                consume_pred(1 + used, np);
            }
        }
        // acc.iter().sum(): ((((((a0+a1)+a2)+a3)+a4)+a5)+a6)+a7 with a4..a7 on the odd lane
        const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]), b2 = pair_swap(acc[2]),
                    b3 = pair_swap(acc[3]);
        float s = 0.0f;
        s += acc[0];
        s += acc[1];
        s += acc[2];
        s += acc[3];
        s += b0;
        s += b1;
        s += b2;
        s += b3;
        return __builtin_sqrtf(s);
    } else {
        // FullVec: one sequential sum per candidate, one candidate per lane (full.rs:24-28)
        float s = 0.0f;
        if (active) {
            const uint4 *src = reinterpret_cast<const uint4 *>(v.rows + (size_t)id * v.row_stride);
            const uint32_t np = v.row_stride >> 4, d = v.dim;
            // bulk: whole pairs of 16-piece stages, two register buffers, the next stage in flight
            // while the current one is summed (one exposed round trip per 512 B instead of per 128 B);
            // only pieces that lie entirely inside the row's d floats take this path
            constexpr uint32_t CH = 16;
            const uint32_t full_pieces = d >> 2;                      // pieces without padding floats
            const uint32_t pairs = full_pieces / (2 * CH);
            uint32_t p_done = 0;
            if (pairs > 0) {
                uint4 a[CH], b[CH];
                auto fetch_at = [&](uint4 (&w)[CH], const uint4 *p) __attribute__((always_inline)) {
#pragma unroll
                    for (uint32_t i = 0; i < CH; i++) w[i] = p[i];
                };
                auto consume_at = [&](const uint4 (&w)[CH], const float *y) __attribute__((always_inline)) {
#pragma unroll
                    for (uint32_t i = 0; i < CH; i++) {
                        const uint32_t dw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const f32x2 x = {__builtin_bit_cast(float, dw[j]), __builtin_bit_cast(float, dw[j + 1])};
                            const f32x2 yy = {y[4 * i + j], y[4 * i + j + 1]};
                            const f32x2 t = x - yy;
                            const f32x2 t2 = t * t;
                            s += t2.x;
                            s += t2.y;
                        }
                    }
                };
                fetch_at(a, src);
#pragma unroll 1
                for (uint32_t pr = 0; pr < pairs; pr++) {
                    const uint32_t st = 2 * pr;
                    fetch_at(b, src + (st + 1) * CH);
                    consume_at(a, yq + 4 * st * CH);
                    if (pr + 1 < pairs) fetch_at(a, src + (st + 2) * CH);
                    consume_at(b, yq + 4 * (st + 1) * CH);
                }
                p_done = pairs * 2 * CH;
            }
            for (uint32_t p0 = p_done; p0 < np; p0 += 8) {
                uint4 w[8];
#pragma unroll
                for (int p = 0; p < 8; p++)
                    w[p] = (p0 + p < np) ? src[p0 + p] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t e = 4 * (p0 + p) + j;
                        const bool in = e < d;
                        const float y = in ? yq[e] : 0.0f;
                        const float t = __builtin_bit_cast(float, dw[j]) - y;
                        const float t2 = t * t;
                        s += in ? t2 : 0.0f;  // +0.0 leaves a non-negative sum unchanged
                    }
                }
            }
        }
        return __builtin_sqrtf(s);
    }
}

// Asynchronous global -> LDS copy of one 1-KiB piece (64 lanes x 16 bytes): lane l's 16 bytes at
// `gsrc` land at LDS byte address lds_dst + 16 l.  No VGPR destination, and -- being inline asm --
// not part of the compiler's s_waitcnt bookkeeping, so the copy stays in flight across the loops
// and LDS atomics of the expansion body (hipcc drains vmcnt(0) at every loop it cannot see
// through).  The consumer issues its own `s_waitcnt vmcnt(0)` before reading the bytes back
// (cdna_hip_programming.md section 5.7: M0 is written in the same statement that reads it).
__device__ __forceinline__ void dma_piece_to_lds(const void *gsrc, uint32_t lds_dst) {
    lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);  // provably wave-uniform
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

// Loads that the compiler's s_waitcnt bookkeeping does not see, each with its own wait.  They serve
// the rare degree > 32 rows inside the inline-rows loops: a single compiler-visible VMEM load
// anywhere in that loop nest makes hipcc drain vmcnt(0) at the loop header on EVERY iteration,
// which would serialise the block prefetch (measured: the prefetch then gains nothing).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t asm_ld32(const void *p) {
    uint32_t r;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint4 asm_ld128(const void *p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return make_uint4(r.x, r.y, r.z, r.w);
}

// FullVec row against the staged query when the dimension is a compile-time constant: all P
// 16-byte pieces of the row are loaded up front (P x 16 bytes in flight per lane), the sum is the
// reference's single left-to-right chain (full.rs:24-28).
template <int P, int DS>
__device__ __forceinline__ float f32_row_sum(const uint4 (&w)[P], const float *yq) {
    // x - y and the square run two elements per instruction (v_pk_add_f32 / v_pk_mul_f32: each
    // element is the same correctly rounded IEEE operation); the sum stays the one serial chain
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const int e = 4 * p + j;
            if (e >= DS) continue;
            if (e + 1 < DS) {
                const f32x2 x = {__builtin_bit_cast(float, dw[j]), __builtin_bit_cast(float, dw[j + 1])};
                const f32x2 y = {yq[e], yq[e + 1]};
                const f32x2 t = x - y;
                const f32x2 t2 = t * t;
                s += t2.x;
                s += t2.y;
            } else {
                const float t = __builtin_bit_cast(float, dw[j]) - yq[e];
                s += t * t;
            }
        }
    }
    return s;
}

// Wide f32 rows (d > 192): the same single chain, with the row streamed through two register buffers
// of CH 16-byte pieces each.  The stage loop has a compile-time trip count and is fully unrolled, so
// the code is straight-line: the loads of stage s + 1 are in flight while stage s is summed and the
// compiler's s_waitcnt counts are exact (a rolled loop drains them at its header).
#ifndef HX_WIDE_CH
#define HX_WIDE_CH 16  // 16-byte pieces per stage buffer of the wide-row loop (two buffers in flight per lane)
#endif
template <int DS, int CH, bool ROLLED>
__device__ __forceinline__ float f32_row_sum_staged(const uint4 *src, const float *yq) {
    constexpr int NP = (DS + 3) / 4, NST = (NP + CH - 1) / CH;
    static_assert(!ROLLED || (NP % (2 * CH) == 0), "the rolled form needs whole stage pairs");
    uint4 a[CH], b[CH];
    float s = 0.0f;
    auto fetch = [&](uint4 (&w)[CH], int st) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < CH; p++)
            if (st * CH + p < NP) w[p] = src[st * CH + p];
    };
    auto consume = [&](const uint4 (&w)[CH], int st) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < CH; p++) {
            if (st * CH + p >= NP) continue;
            const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int e = 4 * (st * CH + p) + j;
                if (e >= DS) continue;
                if (e + 1 < DS) {
                    const f32x2 x = {__builtin_bit_cast(float, dw[j]), __builtin_bit_cast(float, dw[j + 1])};
                    const f32x2 y = {yq[e], yq[e + 1]};
                    const f32x2 t = x - y;
                    const f32x2 t2 = t * t;
                    s += t2.x;
                    s += t2.y;
                } else {
                    const float t = __builtin_bit_cast(float, dw[j]) - yq[e];
                    s += t * t;
                }
            }
        }
    };
    if constexpr (ROLLED) {
        // very wide rows: the unrolled form outgrows the instruction cache (d = 768: 3.9 ms against
        // 1.9 ms for the plain loop), so the stage pairs stay a loop -- one exposed round trip per
        // 2 CH pieces instead of one per 8
        auto fetch_at = [&](uint4 (&w)[CH], const uint4 *p) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CH; i++) w[i] = p[i];
        };
        auto consume_at = [&](const uint4 (&w)[CH], const float *y) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const uint32_t dw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 x = {__builtin_bit_cast(float, dw[j]), __builtin_bit_cast(float, dw[j + 1])};
                    const f32x2 yy = {y[4 * i + j], y[4 * i + j + 1]};
                    const f32x2 t = x - yy;
                    const f32x2 t2 = t * t;
                    s += t2.x;
                    s += t2.y;
                }
            }
        };
        static_assert(DS % 4 == 0, "whole pieces");
        fetch_at(a, src);
#pragma unroll 1
        for (int st = 0; st < NST; st += 2) {
            fetch_at(b, src + (st + 1) * CH);
            consume_at(a, yq + 4 * st * CH);
            if (st + 2 < NST) fetch_at(a, src + (st + 2) * CH);
            consume_at(b, yq + 4 * (st + 1) * CH);
        }
        return s;
    }
    fetch(a, 0);
#pragma unroll
    for (int st = 0; st < NST; st += 2) {
        if (st + 1 < NST) fetch(b, st + 1);
        consume(a, st);
        if (st + 2 < NST) fetch(a, st + 2);
        if (st + 1 < NST) consume(b, st + 1);
    }
    return s;
}

#include "coop_rows.inc"

// ---------------------------------------------------------------------------------------------
// Distance of a stored point to the staged row on the build path.  DS > 0: the dimension is a
// compile-time constant (all row pieces in flight, dead elements vanish), otherwise the runtime
// loops of dist_any_dim.  QUANT8: valid on the even lane of the pair; F32: per lane.
// ---------------------------------------------------------------------------------------------
// COOP (insert kernel, f32 rows of whole lines): the cooperative gather of coop_rows.inc through ids_s
// (64 words) and img (4 KiB) -- the insertion searches of a 50-100M point build read rows scattered over
// tens of GB, where the lane-per-row shape tops out at 1.2 TB/s (profiles/r03_gather_shapes_*.txt).
template <int KIND, int DS, bool COOP = false>
__device__ __forceinline__ float dist_build(const DevView &v, uint32_t id, bool active, int h, const float *yq,
                                            uint32_t *ids_s = nullptr, unsigned char *img = nullptr, int lane = 0) {
    if constexpr (COOP && coop_rows<KIND, DS>()) {
        return __builtin_sqrtf(f32_rows_coop<(DS > 0 ? DS : 32), HX_COOP_K>(v.rows, id, active, yq, ids_s, img, lane));
    } else if constexpr (DS > 0 && KIND == HNSW_VEC_QUANT8) {
        constexpr int NQ = 4 * (DS / 8) + DS % 8;  // elements of half 0 (half 1 has DS % 8 fewer)
        constexpr int P = (8 + NQ + 15) / 16;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) {
            const uint4 *src = reinterpret_cast<const uint4 *>(
                v.rows + (size_t)id * v.row_stride + (size_t)h * v.half_bytes);
            uint4 w[P];
#pragma unroll
            for (int p = 0; p < P; p++) w[p] = src[p];
            __builtin_amdgcn_sched_barrier(0);  // every piece requested before the arithmetic
            const QLds q{yq + h * (v.half_bytes - 8)};
            quant_half_sums<P, DS>(w, q, h, v.nch4, v.rem, acc);
        }
        const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]), b2 = pair_swap(acc[2]),
                    b3 = pair_swap(acc[3]);
        float s = 0.0f;
        s += acc[0];
        s += acc[1];
        s += acc[2];
        s += acc[3];
        s += b0;
        s += b1;
        s += b2;
        s += b3;
        return __builtin_sqrtf(s);
    } else if constexpr (DS > 0 && KIND == HNSW_VEC_F32) {
        constexpr int P = (DS + 3) / 4;
        float sm = 0.0f;
        if (active) {
            const uint4 *src = reinterpret_cast<const uint4 *>(v.rows + (size_t)id * v.row_stride);
            uint4 w[P];
#pragma unroll
            for (int p = 0; p < P; p++) w[p] = src[p];
            __builtin_amdgcn_sched_barrier(0);
            sm = f32_row_sum<P, DS>(w, yq);
        }
        return __builtin_sqrtf(sm);
    } else {
        return dist_any_dim<KIND>(v, id, active, h, yq);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-wave search state and the pieces of search_layer
// ---------------------------------------------------------------------------------------------
template <int R>
struct WaveList {
    u64 L[R];         // list[64 r + lane]; KEY_INVALID beyond n_cur
    uint32_t n_cur;   // wave-uniform
    u64 last_key;     // key (flag dropped) of position ef - 1 when the list is full

    // Merge the wave's candidate keys (KEY_INVALID = none) into the sorted list, keeping the ef
    // smallest: streaming top-ef of searcher.rs:74-94 for a whole batch (order-independent).
    // new_flag (0 or KEY_EXPANDED) is OR-ed into every key that enters the list
    __device__ __forceinline__ void merge(u64 key, uint32_t ef, u64 *perm, int lane, u64 new_flag = 0) {
        const bool full = n_cur >= ef;
        const bool surv = key != KEY_INVALID && (!full || key < last_key);
        u64 smask = __ballot(surv);
        if (smask == 0) return;
        const uint32_t m = (uint32_t)__popcll(smask);
        if (m <= (R == 1 ? 4u : 0u)) {
            // few survivors (the usual case once the list is full): insert them one at a time by
            // shifting the tail of the register-resident list one lane to the right (DPP
            // wave_shr:1, no LDS round trip).  Insertion order is irrelevant (N2).  Only for one-
            // register lists: with R > 1 every insert shifts R registers with a carry, and the
            // rank-scatter below is cheaper even for a single survivor (f32 efSearch 68: 0.266 ->
            // 0.255 ms; SQ counters had shown +33 % VALU instructions per query for R = 2 vs R = 1).
            u64 it = smask;
            while (it) {
                const int j = __ffsll((long long)it) - 1;
                it &= it - 1;
                const u64 e = readlane64(key, j);
                if (n_cur >= ef && !(e < last_key)) continue;  // an earlier insert tightened the bound
                uint32_t pos = 0;
#pragma unroll
                for (int r = 0; r < R; r++)
                    pos += (uint32_t)__popcll(__ballot((L[r] & KEY_MASK) < e));
                u64 carry = 0;  // lane 63 of the previous register feeds lane 0 of the next
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t idx = 64u * r + lane;
                    const uint32_t lo = (uint32_t)L[r], hi = (uint32_t)(L[r] >> 32);
                    uint32_t slo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x138, 0xF, 0xF, false);
                    uint32_t shi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xF, 0xF, false);
                    u64 sh = ((u64)shi << 32) | slo;
                    if (lane == 0) sh = carry;
                    if (R > 1) carry = readlane64(L[r], 63);
                    if (idx == pos)
                        L[r] = e | new_flag;
                    else if (idx > pos)
                        L[r] = sh;
                }
                n_cur = min(n_cur + 1, ef);
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (64u * r + lane >= n_cur) L[r] = KEY_INVALID;
                refresh_last(ef);
            }
            return;
        }
        uint32_t shift[R];
#pragma unroll
        for (int r = 0; r < R; r++) shift[r] = 0;
        uint32_t my_rank = 0;
        u64 it = smask;
        while (it) {  // wave-uniform loop over the survivors
            const int j = __ffsll((long long)it) - 1;
            it &= it - 1;
            const u64 e = readlane64(key, j);
            uint32_t below = 0;  // list entries smaller than e
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool lt = (L[r] & KEY_MASK) < e;  // invalid entries are the maximum
                below += (uint32_t)__popcll(__ballot(lt));
                shift[r] += lt ? 0u : 1u;
            }
            if (surv && e < key) my_rank++;
            if (lane == j) my_rank += below;
        }
        // scatter to the new positions through LDS (all reads of L happened above)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            const uint32_t np = idx + shift[r];
            if (idx < n_cur && np < ef) perm[np] = L[r];
        }
        if (surv && my_rank < ef) perm[my_rank] = key | new_flag;
        n_cur = min(n_cur + m, ef);
        wave_fence();  // single-wave workgroup: orders the LDS writes before the reads
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            L[r] = idx < n_cur ? perm[idx] : KEY_INVALID;
        }
        wave_fence();
        refresh_last(ef);
    }

    __device__ __forceinline__ void refresh_last(uint32_t ef) {
        if (n_cur >= ef) {
            const uint32_t pos = ef - 1;
            u64 k = 0;
#pragma unroll
            for (int r = 0; r < R; r++)
                if ((pos >> 6) == (uint32_t)r) k = readlane64(L[r], pos & 63);
            last_key = k & KEY_MASK;
        } else {
            last_key = KEY_INVALID;
        }
    }

    // position of the smallest entry not expanded yet, -1 if none (the loop of searcher.rs:35)
    __device__ __forceinline__ int first_unexpanded(int lane) const {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            const u64 mk = __ballot(idx < n_cur && (L[r] & KEY_EXPANDED) == 0);
            if (mk) return 64 * r + (__ffsll((long long)mk) - 1);
        }
        return -1;
    }
};

// merge over the first RR registers of a wider list (every entry at 64 RR and beyond is invalid before and after)
template <int RR, int R>
__device__ __forceinline__ void merge_prefix(WaveList<R> &wl, u64 key, uint32_t ef, u64 *perm, int lane, u64 new_flag) {
    static_assert(RR <= R, "prefix of the list");
    WaveList<RR> t;
#pragma unroll
    for (int r = 0; r < RR; r++) t.L[r] = wl.L[r];
    t.n_cur = wl.n_cur;
    t.last_key = wl.last_key;
    t.merge(key, ef, perm, lane, new_flag);
#pragma unroll
    for (int r = 0; r < RR; r++) wl.L[r] = t.L[r];
    wl.n_cur = t.n_cur;
    wl.last_key = t.last_key;
}

// LDS visited table (IntSet::insert, results.rs:101-103): open addressing over BUCKETS of four
// 32-bit slots.  One ds_read_b128 fetches the home bucket, the four compares run in registers and
// a single ds_cmpst claims the first empty slot, so an insert is two LDS round trips whatever the
// load; the classic one-slot linear probe needed one round trip per probe and the wave iterated as
// long as its unluckiest lane (4-5 rounds at 30 % load).  Slots fill left to right and never empty,
// ids within one adjacency row are distinct, so "absent from the bucket, first empty slot claimed"
// is an exact insert.  Returns true when id was not present.
// read-only membership test of the same table (speculative evaluation: nothing may be inserted yet)
__device__ __forceinline__ bool visited_contains(const uint32_t *tab, uint32_t hmask, uint32_t slots_log2,
                                                 uint32_t id) {
    const uint32_t bmask = hmask >> 2;
    uint32_t b = (id * 0x9E3779B1u) >> (32 - (slots_log2 - 2));
    while (true) {
        const uint4 bk = *reinterpret_cast<const uint4 *>(tab + 4 * b);
        if (bk.x == id || bk.y == id || bk.z == id || bk.w == id) return true;
        if (bk.x == HX_EMPTY_SLOT || bk.y == HX_EMPTY_SLOT || bk.z == HX_EMPTY_SLOT || bk.w == HX_EMPTY_SLOT)
            return false;
        b = (b + 1) & bmask;
    }
}

__device__ __forceinline__ bool visited_insert(uint32_t *tab, uint32_t hmask, uint32_t slots_log2,
                                               uint32_t id) {
    const uint32_t bmask = hmask >> 2;
    uint32_t b = (id * 0x9E3779B1u) >> (32 - (slots_log2 - 2));
    while (true) {
        const uint4 bk = *reinterpret_cast<const uint4 *>(tab + 4 * b);
        if (bk.x == id || bk.y == id || bk.z == id || bk.w == id) return false;
        int j = -1;
        if (bk.x == HX_EMPTY_SLOT)
            j = 0;
        else if (bk.y == HX_EMPTY_SLOT)
            j = 1;
        else if (bk.z == HX_EMPTY_SLOT)
            j = 2;
        else if (bk.w == HX_EMPTY_SLOT)
            j = 3;
        if (j < 0) {
            b = (b + 1) & bmask;  // bucket full: next bucket
            continue;
        }
        const uint32_t old = atomicCAS(&tab[4 * b + j], HX_EMPTY_SLOT, id);
        if (old == HX_EMPTY_SLOT) return true;
        // another lane of this wave took that slot in the same round: look at the bucket again
    }
}

// bytes of the region the merge buffer shares with the cooperative gather's image (4 KiB + 64 rank words)
template <int KIND, int DS, int R>
__host__ __device__ constexpr uint32_t scratch_region_bytes() {
    return coop_rows<KIND, DS>() && 64u * R * 8u < HX_COOP_IMG_BYTES + 256u ? (uint32_t)HX_COOP_IMG_BYTES + 256u : 64u * R * 8u;
}

template <int KIND, int P, int DS, int R, bool FAT>
__global__ void __launch_bounds__(64)
hx_search_kernel(const DevView v, const SearchArgs a, const uint32_t slots_log2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t q = a.qsel ? a.qsel[blockIdx.x] : blockIdx.x;
    uint32_t *htab = reinterpret_cast<uint32_t *>(smem);
    const uint32_t hslots = 1u << slots_log2, hmask = hslots - 1;
    u64 *perm = reinterpret_cast<u64 *>(smem + 4ull * hslots);
    // the merge's permutation buffer (64 R keys) and the cooperative gather's stage image + rank words are
    // never live together (a distance pass ends before its merge starts), so they share one region: with a
    // 32-KiB visited table the wave stays within a quarter of the CU's LDS for every list width up to 512
    constexpr uint32_t PERM_BYTES = scratch_region_bytes<KIND, DS, R>();
    float *yq = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(perm) + PERM_BYTES);
    // the visited set: an LDS table at 75 % load at most, and -- lists of eight / sixteen registers, round 4 -- a second
    // level in HBM for the ids beyond it, so that the table stays at 32 KiB and four waves per CU stay resident where a
    // 64- / 128-KiB table left two / one (the scheme of search_lean.hip's Visited::look2: the LDS level is CLOSED once it
    // holds its limit, later ids are claimed by one compare-and-swap in the HBM table, an id is in the set iff it is in
    // either level and is only inserted after both were found not to hold it)
    constexpr bool SPILLV = R >= 8;
    uint32_t lds_limit = hslots - (hslots >> 2);
    uint32_t *gtab = nullptr;
    uint32_t gmask = 0, gshift = 0;
    bool spill = false;  // wave-uniform: the LDS level is closed (reset at every layer)
    if (SPILLV && a.spill_tab != nullptr) {
        gtab = a.spill_tab + ((size_t)blockIdx.x << a.spill_log2);
        gmask = (1u << a.spill_log2) - 1;
        gshift = 32 - a.spill_log2;
        if (a.lds_limit != 0) lds_limit = min(lds_limit, max(128u, a.lds_limit));
    }
    const uint32_t vis_limit = lds_limit + (gtab != nullptr ? (gmask + 1) / 2 : 0u);
    // FAT: two 64 x 16 x P byte buffers for the prefetched block (after yq, 16-byte aligned)
    const uint32_t yq_bytes =
        ((KIND == HNSW_VEC_QUANT8 ? 2u * (v.half_bytes - 8) * 4u : v.dim * 4u) + 15u) & ~15u;
    unsigned char *spec_buf = reinterpret_cast<unsigned char *>(yq) + yq_bytes;
    const uint32_t spec_lds = __builtin_amdgcn_groupstaticsize() + 4u * hslots + PERM_BYTES + yq_bytes;
    unsigned char *coop_img = reinterpret_cast<unsigned char *>(perm);
    uint32_t *coop_ids = reinterpret_cast<uint32_t *>(coop_img + HX_COOP_IMG_BYTES);

    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;  // lanes per candidate
    constexpr int CHUNK = 64 / LPC;                         // adjacency slots per pass
    const int h = (LPC == 2) ? (lane & 1) : 0;
    const int cslot = lane / LPC;

    const uint32_t d = v.dim;
    const float *qv = a.Q + (size_t)q * d;
    uint32_t n_dist = 0, n_exp = 0, sum_deg = 0;
    int32_t status = HNSW_OK;
#ifdef HX_STAMPS
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_mid = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

    // ---- stage the query (Point::new -> QuantVec::new for QUANT8, template.rs:313) ----
    const uint32_t nq_half = (KIND == HNSW_VEC_QUANT8) ? (v.half_bytes - 8) : 0;
    if (!stage_query<KIND>(v, qv, yq, lane)) status = HNSW_ERR_NAN_INPUT;

    // query values of this lane in registers when the dimension is a compile-time constant
    // (wide compile-time dimensions keep the query in LDS: 4 d / 8 registers would not fit)
    constexpr bool QREG = (KIND == HNSW_VEC_QUANT8 && DS > 0 && DS <= 160);
    constexpr int NQR = QREG ? (4 * (DS / 8) + DS % 8) : 1;
    QRegs<NQR> qreg;
    if (QREG) {
#pragma unroll
        for (int e = 0; e < NQR; e++) qreg.v[e] = yq[h * nq_half + e];
    }
    const QLds qlds{yq + h * nq_half};

    WaveList<R> wl;
#pragma unroll
    for (int r = 0; r < R; r++) wl.L[r] = KEY_INVALID;
    wl.n_cur = 0;
    wl.last_key = KEY_INVALID;

    // ---- distance of one candidate per lane group; returns the key on the group's first lane
    auto eval_dist = [&](uint32_t id, bool active, bool hidden_loads) __attribute__((always_inline)) -> float {
        float dist = 0.0f;
        if (KIND == HNSW_VEC_QUANT8 && P > 0) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (active) {
                const uint4 *src = reinterpret_cast<const uint4 *>(
                    v.rows + (size_t)id * v.row_stride + (size_t)h * v.half_bytes);
                uint4 w[P > 0 ? P : 1];
                if (FAT && hidden_loads) {
#pragma unroll
                    for (int p = 0; p < P; p++) w[p] = asm_ld128(src + p);
                } else {
#pragma unroll
                    for (int p = 0; p < P; p++) w[p] = src[p];
                    __builtin_amdgcn_sched_barrier(0);  // every piece requested before the arithmetic
                }
                if (QREG)
                    quant_half_sums<(P > 0 ? P : 1), DS>(w, qreg, h, v.nch4, v.rem, acc);
                else
                    quant_half_sums<(P > 0 ? P : 1), DS>(w, qlds, h, v.nch4, v.rem, acc);
            }
            // acc.iter().sum(): ((((((a0+a1)+a2)+a3)+a4)+a5)+a6)+a7 with a4..a7 on the odd lane
            const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]), b2 = pair_swap(acc[2]),
                        b3 = pair_swap(acc[3]);
            float s = 0.0f;
            s += acc[0];
            s += acc[1];
            s += acc[2];
            s += acc[3];
            s += b0;
            s += b1;
            s += b2;
            s += b3;
            dist = __builtin_sqrtf(s);
        } else if (KIND == HNSW_VEC_F32 && P > 0 && DS > 0) {
            float sm = 0.0f;
            if constexpr (coop_rows<KIND, DS>()) {
                // whole-line rows: eight lanes to a 128-byte line, owners sum out of an LDS image
                sm = f32_rows_coop<(DS > 0 ? DS : 32), HX_COOP_K>(v.rows, id, active, yq, coop_ids, coop_img, lane);
            } else if (active) {
                const uint4 *src = reinterpret_cast<const uint4 *>(v.rows + (size_t)id * v.row_stride);
                if constexpr (coop_rows<KIND, DS>()) {
                    // (never reached: the cooperative gather below runs outside the lane mask)
                } else if constexpr (P > 48) {
                    sm = f32_row_sum_staged<(DS > 0 ? DS : 1), HX_WIDE_CH, true>(src, yq);
                } else {
                    uint4 w[P > 0 ? P : 1];
#pragma unroll
                    for (int p = 0; p < P; p++) w[p] = src[p];
                    __builtin_amdgcn_sched_barrier(0);  // every piece requested before the chain starts
#ifdef HX_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    dbg_mid = __builtin_readcyclecounter();
#endif
                    sm = f32_row_sum<(P > 0 ? P : 1), (DS > 0 ? DS : 1)>(w, yq);
                }
            }
            dist = __builtin_sqrtf(sm);
        } else {
            dist = dist_any_dim<KIND>(v, id, active, h, yq);
        }
        return dist;
    };
    auto eval_key = [&](uint32_t id, bool active, bool hidden_loads) -> u64 {
        const float dist = eval_dist(id, active, hidden_loads);
        const bool first = (LPC == 1) || (h == 0);
        if (!(active && first)) return KEY_INVALID;
        if (dist != dist) {
            status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
            return KEY_INVALID;
        }
        return ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | id;
    };

    // ---- one pass over up to CHUNK neighbour ids (one per lane group) ----
    // entries the visited table holds (exact: the overflow check adds the row's worst case before a pass)
    uint32_t n_vis = 0;
    // IntSet::insert / contains over both levels (true = id was absent / is present)
    auto vins = [&](uint32_t id) __attribute__((always_inline)) -> bool {
        if constexpr (SPILLV) {
            if (spill) {
                if (visited_contains(htab, hmask, slots_log2, id)) return false;
                uint32_t s2 = ((id * 0x9E3779B1u) >> gshift) & gmask;
                while (true) {  // the compare-and-swap is the probe: it returns what the slot holds
                    const uint32_t old = atomicCAS(gtab + s2, HX_EMPTY_SLOT, id);
                    if (old == HX_EMPTY_SLOT) return true;
                    if (old == id) return false;
                    s2 = (s2 + 1) & gmask;
                }
            }
        }
        return visited_insert(htab, hmask, slots_log2, id);
    };
    auto vhas = [&](uint32_t id) __attribute__((always_inline)) -> bool {
        if (visited_contains(htab, hmask, slots_log2, id)) return true;
        if constexpr (SPILLV) {
            if (spill) {
                uint32_t s2 = ((id * 0x9E3779B1u) >> gshift) & gmask;
                while (true) {
                    const uint32_t cur = __hip_atomic_load(gtab + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == id) return true;
                    if (cur == HX_EMPTY_SLOT) return false;
                    s2 = (s2 + 1) & gmask;
                }
            }
        }
        return false;
    };
    // room for cnt more ids?  Closes the LDS level (and empties the HBM one) when they would take it past its limit:
    // checked BEFORE every chunk of ids is inserted, so the LDS table never holds more than its limit
    auto room = [&](uint32_t cnt) __attribute__((always_inline)) -> bool {
        if constexpr (SPILLV) {
            if (gtab != nullptr && !spill && n_vis + cnt > lds_limit) {
                for (uint32_t s2 = lane; s2 < ((gmask + 1) >> 2); s2 += 64)
                    reinterpret_cast<uint4 *>(gtab)[s2] = make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                spill = true;
            }
        }
        return n_vis + cnt <= vis_limit;
    };
    auto process = [&](uint32_t id, bool valid, bool visit, uint32_t ef_l, bool hidden_loads = false) {
        bool fresh = valid;
        if (visit) {
            bool f = false;
            if (valid && h == 0) f = vins(id);
            if (LPC == 2) f = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
            fresh = f;
        }
        const u64 fm = __ballot(fresh && h == 0);
        if (visit) n_vis += (uint32_t)__popcll(fm);
        if (fm == 0) return;
        n_dist += (uint32_t)__popcll(fm);
        u64 key = eval_key(id, fresh, hidden_loads);
        if (__ballot(status != HNSW_OK)) status = HNSW_ERR_NAN_INPUT;
        wl.merge(key, ef_l, perm, lane);
    };

    if (status == HNSW_OK) {
        // ---- entry set: {ep} (template.rs:316-319) or the caller's (search_layer seam) ----
        const uint32_t n_entry = a.entries ? a.n_entry : 1;
        const uint32_t ef_first = max(1u, (a.layer_hi > a.layer_lo) ? a.ef_upper : a.ef_bottom);
        for (uint32_t base = 0; base < n_entry; base += CHUNK) {
            const uint32_t i = base + cslot;
            const bool valid = i < n_entry;
            uint32_t id = 0;
            if (valid) id = a.entries ? a.entries[i] : v.ep;
            if (__ballot(valid && id >= v.n_points)) {
                status = HNSW_ERR_ARG;
                break;
            }
            // the reference keeps every entry it is given; an entry set larger than ef is
            // rejected by the host side
            process(id, valid, false, max(ef_first, n_entry));
        }
    }

    for (int layer = a.layer_hi; status == HNSW_OK && layer >= a.layer_lo; layer--) {
        const uint32_t ef_l = max(1u, layer > a.layer_lo ? a.ef_upper : a.ef_bottom);
        // visited.clear() (searcher.rs:101) / fresh Results: empty table
        for (uint32_t s = lane; s < (hslots >> 2); s += 64)
            reinterpret_cast<uint4 *>(htab)[s] =
                make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
        wave_fence();
        n_vis = 0;
        spill = false;  // an empty LDS table: open again
        // candidates ∪= selected, visited ∪= ids(selected)  (searcher.rs:32-33)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            if (idx < wl.n_cur) {
                wl.L[r] &= KEY_MASK;
                vins((uint32_t)wl.L[r]);
            }
        }
        n_vis = wl.n_cur;
        wl.refresh_last(ef_l);
        const uint32_t S = layer == 0 ? v.S0 : v.S1;

        if (FAT && layer == 0) {
            // ---- layer 0 over the inline-rows blocks: ONE dependent, coalesced read per expansion
            // (the 32 neighbour rows of the candidate, ids embedded), and the block of the PREDICTED
            // next candidate -- the smallest unexpanded entry once the current one is marked -- is
            // already in flight while the current block is filtered, evaluated and merged.  The
            // prediction is a prefetch only: what is evaluated and merged, and in which order, is
            // exactly what the loop below does on the compact layout.
            constexpr int PP = P > 0 ? P : 1;
            constexpr uint32_t BLK = 1024u * PP;  // bytes of one block image in LDS
            uint4 w[PP];
            bool have_spec = false;
            uint32_t spec_id = 0, spec_sel = 0;
            const size_t lane_off = (size_t)cslot * v.row_stride + (size_t)h * v.half_bytes;
            // The hot loop below holds no compiler-visible VMEM load: hipcc drains vmcnt(0) at the
            // header of any loop with such a load somewhere inside, which would also drain the
            // prefetch.  The rare degree > 32 rows leave the hot loop, are served from the compact
            // layout with plain loads, and re-enter it.
            uint32_t ovf_pending = HX_EMPTY_SLOT;
            bool done = false;
            while (!done && status == HNSW_OK) {
            while (true) {
                STAMP(t0);
                const int cpos = wl.first_unexpanded(lane);
                if (cpos < 0) {
                    done = true;
                    break;
                }
                uint32_t cid = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((cpos >> 6) == r) {
                        cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], cpos & 63);
                        if (lane == (cpos & 63)) wl.L[r] |= KEY_EXPANDED;
                    }
                }
                n_exp++;
#ifdef HX_STAMPS
                const bool dbg_hit = have_spec && spec_id == cid;
                if (dbg_hit) dbg_acc[6]++;
#endif
                // Every block is staged through LDS by DMA (no compiler-visible VMEM load in this
                // loop, so hipcc inserts no vmcnt wait that would also drain the prefetch): the
                // current block is either the image prefetched one expansion ago or is fetched now;
                // then the predicted next block is started into the other buffer.
                const bool hit = have_spec && spec_id == cid;
                uint32_t cur_sel = spec_sel;
                if (!hit) {
                    cur_sel = spec_sel ^ 1u;  // a stale prefetch may still be landing in spec_sel
                    const unsigned char *src = v.fat + (size_t)cid * v.fat_stride + lane_off;
#pragma unroll
                    for (int p = 0; p < PP; p++)
                        dma_piece_to_lds(src + 16 * p, spec_lds + cur_sel * BLK + 1024u * p);
                }
                have_spec = false;
                {
                    const int ppos = wl.first_unexpanded(lane);
                    if (ppos >= 0) {
                        uint32_t pid = 0;
#pragma unroll
                        for (int r = 0; r < R; r++)
                            if ((ppos >> 6) == r)
                                pid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], ppos & 63);
                        spec_sel = cur_sel ^ 1u;
                        const unsigned char *src = v.fat + (size_t)pid * v.fat_stride + lane_off;
#pragma unroll
                        for (int p = 0; p < PP; p++)
                            dma_piece_to_lds(src + 16 * p, spec_lds + spec_sel * BLK + 1024u * p);
                        spec_id = pid;
                        have_spec = true;
                    }
                }
                // the current block has landed once all but the PP youngest DMA pieces are done
                if (have_spec)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                {
                    const uint4 *img = reinterpret_cast<const uint4 *>(spec_buf + cur_sel * BLK) + lane;
#pragma unroll
                    for (int p = 0; p < PP; p++) w[p] = img[64 * p];
                }
                STAMP(t1);
                STAMP_ADD(0, t0, t1);
                // the neighbour id travels in the last 4 bytes of half 0
                const uint32_t raw = w[PP - 1].w;
                const uint32_t nb = h ? (uint32_t)pair_swap_i((int)raw) : raw;
                const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
                const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
                uint32_t ovf = HX_EMPTY_SLOT;
                const u64 pm = __ballot(is_ptr);
                if (pm) ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                if (cnt != 0) {
                    sum_deg += cnt;
                    if (!room(cnt)) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    bool f = false;
                    if (valid && h == 0) f = vins(nb);
                    const bool fresh = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
                    const u64 fm = __ballot(fresh && h == 0);
                    n_vis += (uint32_t)__popcll(fm);  // what the table really holds
                    STAMP(t2);
#ifdef HX_STAMPS
                    if (dbg_hit)
                        dbg_acc[1] += t2 - t1;
                    else
                        dbg_acc[5] += t2 - t1;
#endif
                    if (fm != 0) {
                        n_dist += (uint32_t)__popcll(fm);
                        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (fresh) {
                            if (DS > 0)
                                quant_half_sums<PP, DS>(w, qreg, h, v.nch4, v.rem, acc);
                            else
                                quant_half_sums<PP, 0>(w, qlds, h, v.nch4, v.rem, acc);
                        }
                        const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]),
                                    b2 = pair_swap(acc[2]), b3 = pair_swap(acc[3]);
                        float s = 0.0f;
                        s += acc[0];
                        s += acc[1];
                        s += acc[2];
                        s += acc[3];
                        s += b0;
                        s += b1;
                        s += b2;
                        s += b3;
                        const float dist = __builtin_sqrtf(s);
                        u64 key = KEY_INVALID;
                        bool nan = false;
                        if (fresh && h == 0) {
                            nan = dist != dist;
                            if (!nan) key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | nb;
                        }
                        if (__ballot(nan)) {
                            status = HNSW_ERR_NAN_INPUT;
                            break;
                        }
                        STAMP(t3);
                        STAMP_ADD(2, t2, t3);
                        wl.merge(key, ef_l, perm, lane);
                        STAMP(t4);
                        STAMP_ADD(3, t3, t4);
                    }
                }
                if (ovf != HX_EMPTY_SLOT) {
                    ovf_pending = ovf;
                    break;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no prefetch in flight outside the hot loop
            have_spec = false;
            if (done || status != HNSW_OK) break;
            {  // degree > S0: the rest of the row, from the compact rows
                const uint32_t ovf = ovf_pending;
                ovf_pending = HX_EMPTY_SLOT;
                {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)asm_ld32(v.ovf_off + ovf));
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)asm_ld32(v.ovf_off + ovf + 1));
                    for (uint32_t base = lo; base < hi; base += CHUNK) {
                        const uint32_t i = base + cslot;
                        const bool ov = i < hi;
                        const uint32_t onb = ov ? asm_ld32(v.ovf_nbrs + i) : HX_EMPTY_SLOT;
                        const uint32_t ocnt = (uint32_t)__popcll(__ballot(ov && h == 0));
                        sum_deg += ocnt;
                        if (!room(ocnt)) {
                            status = HNSW_ERR_OVERFLOW;
                            break;
                        }
                        process(onb, ov, true, ef_l, true);
                        if (status != HNSW_OK) break;
                    }
                }
            }
            }
            continue;
        }

        if (KIND == HNSW_VEC_F32 && !FAT && layer == 0 && S <= 32 && !(a.flags & 1u)) {
            // ---- layer 0, one lane per neighbour, rows of at most 32 slots: TWO rows per pass.
            // Lanes 0..31 take the row of the candidate c being expanded, lanes 32..63 the row of the
            // runner-up p (the smallest unexpanded entry once c is marked).  Both adjacency rows and
            // both sets of vector rows are fetched together, one pass evaluates all of them.  c is
            // committed (visited insert, merge); if p is then still the smallest unexpanded entry --
            // measured: 2 times out of 3 -- it is committed from the distances already in registers,
            // i.e. that expansion costs no memory round trip and no distance pass.  Otherwise p's
            // results are dropped.  Nothing of p touches the visited set or the list before its
            // commit, and the commit filters against everything c inserted, so the expanded nodes,
            // the fresh sets, the counters and the result are those of the one-at-a-time loop.
            int cpos = wl.first_unexpanded(lane);  // carried: the pick after a merge is the next c
            while (status == HNSW_OK) {
                if (cpos < 0) break;
                STAMP(f0);
                uint32_t cid = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((cpos >> 6) == r) {
                        cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], cpos & 63);
                        if (lane == (cpos & 63)) wl.L[r] |= KEY_EXPANDED;
                    }
                }
                n_exp++;
                const int ppos = wl.first_unexpanded(lane);
                uint32_t pid = cid;
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (ppos >= 0 && (ppos >> 6) == r)
                        pid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], ppos & 63);
                const bool upper = lane >= 32;
                const uint32_t slot = (uint32_t)lane & 31u;
                uint32_t nb = HX_EMPTY_SLOT;
                if (slot < S && (!upper || ppos >= 0)) nb = v.adj0[(size_t)(upper ? pid : cid) * S + slot];
#ifdef HX_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP(f1);
                STAMP_ADD(0, f0, f1);
                dbg_acc[7]++;
#endif
                const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
                const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
                const u64 pm = __ballot(is_ptr);
                // a runner-up row with an overflow pointer is not speculated on
                const bool spec_ok = ppos >= 0 && (pm >> 32) == 0;
                uint32_t ovf = HX_EMPTY_SLOT;
                if (pm & 0xFFFFFFFFull)
                    ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)(pm & 0xFFFFFFFFull)) - 1) &
                          ~HX_OVF_FLAG;
                // ---- c: filter now; p: read-only look-up (its insert happens at its commit)
                const u64 vmask = __ballot(valid);
                const uint32_t cnt_c = (uint32_t)__popcll(vmask & 0xFFFFFFFFull);
                const uint32_t cnt_p = (uint32_t)__popcll(vmask >> 32);
                sum_deg += cnt_c;
                if (!room(cnt_c)) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                bool want = false;
                if (valid && !upper) want = vins(nb);
                n_vis += (uint32_t)__popcll(__ballot(want && !upper));  // what the table really holds
                // (the look-up runs after the inserts of this pass: what c just claimed is skipped)
                wave_fence();
                if (valid && upper && spec_ok) want = !vhas(nb);
                n_dist += (uint32_t)__popcll(__ballot(want && !upper));
                STAMP(f2);
                STAMP_ADD(1, f1, f2);
#ifdef HX_STAMPS
                dbg_mid = f2;
#endif
                float dist = 0.0f;
                if (__ballot(want)) dist = eval_dist(nb, want, false);
                STAMP(f3);
                STAMP_ADD(2, f2, dbg_mid);
                STAMP_ADD(3, dbg_mid, f3);
                const bool nan = want && dist != dist;
                if (__ballot(nan && !upper)) {
                    status = HNSW_ERR_NAN_INPUT;
                    break;
                }
                const u64 key = (want && !nan) ? (((u64)__builtin_bit_cast(uint32_t, dist) << 32) | nb) : KEY_INVALID;
                wl.merge(upper ? KEY_INVALID : key, ef_l, perm, lane);
                if (ovf != HX_EMPTY_SLOT) {  // degree > S: the rest of c's row
                    const uint32_t lo = v.ovf_off[ovf], hi = v.ovf_off[ovf + 1];
                    for (uint32_t base = lo; base < hi; base += CHUNK) {
                        const uint32_t i = base + cslot;
                        const bool ov = i < hi;
                        const uint32_t onb = ov ? v.ovf_nbrs[i] : HX_EMPTY_SLOT;
                        const uint32_t ocnt = (uint32_t)__popcll(__ballot(ov && h == 0));
                        sum_deg += ocnt;
                        if (!room(ocnt)) {
                            status = HNSW_ERR_OVERFLOW;
                            break;
                        }
                        process(onb, ov, true, ef_l);
                        if (status != HNSW_OK) break;
                    }
                    if (status != HNSW_OK) break;
                }
                // ---- is p the next candidate?  then commit it from the registers
                STAMP(f4);
                STAMP_ADD(5, f3, f4);
                const int npos = wl.first_unexpanded(lane);
                cpos = npos;
                if (npos < 0) break;
                if (!spec_ok) continue;
                uint32_t nid = 0;
#pragma unroll
                for (int r = 0; r < R; r++)
                    if ((npos >> 6) == r)
                        nid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], npos & 63);
                if (nid != pid) continue;
#pragma unroll
                for (int r = 0; r < R; r++)
                    if ((npos >> 6) == r && lane == (npos & 63)) wl.L[r] |= KEY_EXPANDED;
                n_exp++;
                sum_deg += cnt_p;
                if (!room(cnt_p)) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                // every valid neighbour of p goes through the filter now: one that c's commit inserted
                // meanwhile is dropped, one that was skipped above was in the set already
                bool fresh = false;
                if (valid && upper) fresh = vins(nb);
                n_vis += (uint32_t)__popcll(__ballot(fresh));
                n_dist += (uint32_t)__popcll(__ballot(fresh));
                if (__ballot(fresh && nan)) {
                    status = HNSW_ERR_NAN_INPUT;
                    break;
                }
                wl.merge((fresh && upper) ? key : KEY_INVALID, ef_l, perm, lane);
                cpos = wl.first_unexpanded(lane);
                STAMP(f5);
                STAMP_ADD(6, f4, f5);
            }
            continue;
        }

        while (true) {
            const int cpos = wl.first_unexpanded(lane);
            if (cpos < 0) break;  // candidates exhausted / only worse ones left
            // pop it: mark expanded, fetch its id
            uint32_t cid = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((cpos >> 6) == r) {
                    cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], cpos & 63);
                    if (lane == (cpos & 63)) wl.L[r] |= KEY_EXPANDED;
                }
            }
            n_exp++;
            const uint32_t *row;
            if (layer == 0) {
                row = v.adj0 + (size_t)cid * S;
            } else {
                const uint32_t ub = v.upper_base[cid];
                if (ub == HX_EMPTY_SLOT) {  // Graph::neighbors_vec -> NodeNotInGraph
                    status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                    break;
                }
                row = v.adj_up + ((size_t)ub + layer - 1) * S;
            }
            uint32_t ovf = HX_EMPTY_SLOT;
            for (uint32_t c0 = 0; c0 < S; c0 += CHUNK) {
                const uint32_t slot = c0 + cslot;
                uint32_t nb = HX_EMPTY_SLOT;
                if (slot < S) nb = row[slot];
                const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
                const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
                const u64 pm = __ballot(is_ptr);
                if (pm) ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                if (cnt == 0) continue;
                sum_deg += cnt;
                if (!room(cnt)) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                process(nb, valid, true, ef_l);
                if (status != HNSW_OK) break;
            }
            if (status == HNSW_OK && ovf != HX_EMPTY_SLOT) {  // degree > S: the rest of the row
                const uint32_t lo = v.ovf_off[ovf], hi = v.ovf_off[ovf + 1];
                for (uint32_t base = lo; base < hi; base += CHUNK) {
                    const uint32_t i = base + cslot;
                    const bool valid = i < hi;
                    const uint32_t nb = valid ? v.ovf_nbrs[i] : HX_EMPTY_SLOT;
                    const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                    sum_deg += cnt;
                    if (!room(cnt)) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    process(nb, valid, true, ef_l);
                    if (status != HNSW_OK) break;
                }
            }
            if (status != HNSW_OK) break;
        }
    }

    // ---- get_top_selected(n) (results.rs:59-61): the first n of the ascending list ----
    const uint32_t count = status == HNSW_OK ? min(a.n, wl.n_cur) : 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t idx = 64u * r + lane;
        if (idx < a.n) {
            const bool have = idx < count;
            a.out_ids[(size_t)q * a.n + idx] = have ? (uint32_t)wl.L[r] : HX_EMPTY_SLOT;
            if (a.out_dists)
                a.out_dists[(size_t)q * a.n + idx] =
                    have ? __builtin_bit_cast(float, (uint32_t)((wl.L[r] & KEY_MASK) >> 32))
                         : __builtin_inff();
        }
    }
    for (uint32_t idx = 64u * R + lane; idx < a.n; idx += 64) {  // n > list capacity: padding
        a.out_ids[(size_t)q * a.n + idx] = HX_EMPTY_SLOT;
        if (a.out_dists) a.out_dists[(size_t)q * a.n + idx] = __builtin_inff();
    }
#ifdef HX_STAMPS
    if (lane == 0 && a.dbg) {
        dbg_acc[4] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 8; i++) a.dbg[(size_t)q * 8 + i] = dbg_acc[i];
    }
#endif
    if (lane == 0) {
        if (a.out_counts) a.out_counts[q] = count;
        hnsw_query_stats st;
        st.n_dist = n_dist;
        st.n_exp = n_exp;
        st.sum_deg = sum_deg;
        st.status = status;
        a.out_stats[q] = st;
    }
}

// =============================================================================================
// Two-wave search kernel (QUANT8, inline-rows layout, m <= 16): ONE QUERY PER 128-THREAD
// WORKGROUP.  A batch of 1024 queries is only one wave per SIMD for the one-wave kernel, and a
// lone wave issues at most one vector instruction every four cycles; giving every query two
// waves doubles the waves per SIMD and halves each wave's share of an expansion:
//   - wave w owns adjacency slots 16w .. 16w+15, FOUR lanes per slot (lane (h, sub) of the quad:
//     running sums 4h + 2 sub, 4h + 2 sub + 1 of distance_unrolled), DMA-stages only its half of
//     the 4-KiB block, filters its 16 ids through the SHARED visited table;
//   - both waves keep an identical copy of the sorted list in registers: after the distance step
//     they exchange their <= 16 keys through LDS (one workgroup barrier per expansion) and each
//     merges all 32, so pick / predict / merge never need another word of communication.
// Wave 0 alone walks the entry point and the upper layers (ef = 1) on the compact layout, then
// hands its list to wave 1.  Results are those of the one-wave kernel bit for bit (the batch a
// merge sees is the same set of keys, and merging is order-independent).
// =============================================================================================
template <int N>
struct QRegs2 {
    float v[N];
    __device__ __forceinline__ float operator[](int i) const { return v[i]; }
};
struct QcLds {  // chunk values of one (h, sub) lane: element 4 (i / 2) + 2 sub + i % 2 of the half
    const float *p;
    __device__ __forceinline__ float operator[](int i) const { return p[4 * (i >> 1) + (i & 1)]; }
};

template <int P, int DS, int R>
__global__ void __launch_bounds__(128)
hx_search2_kernel(const DevView v, const SearchArgs a, const uint32_t slots_log2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KIND = HNSW_VEC_QUANT8;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t q = a.qsel ? a.qsel[blockIdx.x] : blockIdx.x;
    const uint32_t hslots = 1u << slots_log2, hmask = hslots - 1;
    const uint32_t vis_limit = hslots - (hslots >> 2);
    const uint32_t yq_bytes = ((2u * (v.half_bytes - 8) * 4u) + 15u) & ~15u;
    constexpr uint32_t HALF_BLK = 16u * 32u * P;  // bytes of one wave's half block (16 rows)
    // LDS carve (all dynamic): visited table | per-wave perm | exchange | per-wave yq | per-wave images
    uint32_t off = 0;
    uint32_t *htab = reinterpret_cast<uint32_t *>(smem);
    off += 4u * hslots;
    u64 *perm = reinterpret_cast<u64 *>(smem + off) + 64 * R * wv;
    u64 *perm0 = reinterpret_cast<u64 *>(smem + off);
    off += 2u * 64u * R * 8u;
    u64 *xkeys = reinterpret_cast<u64 *>(smem + off);  // [2 parities][32 slots]
    off += 2u * 32u * 8u;
    uint32_t *xmeta = reinterpret_cast<uint32_t *>(smem + off);  // [2 parities][2 waves][4]
    off += 2u * 2u * 4u * 4u;
    float *yq = reinterpret_cast<float *>(smem + off + yq_bytes * wv);
    off += 2u * yq_bytes;
    unsigned char *img = smem + off + 2u * HALF_BLK * wv;  // this wave's two half-block images
    const uint32_t img_lds = __builtin_amdgcn_groupstaticsize() + off + 2u * HALF_BLK * wv;

    const uint32_t d = v.dim;
    const float *qv = a.Q + (size_t)q * d;
    uint32_t n_dist = 0, n_exp = 0, sum_deg = 0, n_vis = 0;
    int32_t status = HNSW_OK;
#ifdef HX_STAMPS
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_mid = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

    // every wave stages its own copy of the query
    const uint32_t nq_half = v.half_bytes - 8;
    if (!stage_query<KIND>(v, qv, yq, lane)) status = HNSW_ERR_NAN_INPUT;

    WaveList<R> wl;
#pragma unroll
    for (int r = 0; r < R; r++) wl.L[r] = KEY_INVALID;
    wl.n_cur = 0;
    wl.last_key = KEY_INVALID;

    // ------------------------------------------------------------------------------------------
    // wave 0: entry point + upper layers (ef = 1), two lanes per candidate on the compact layout
    // ------------------------------------------------------------------------------------------
    if (wv == 0 && status == HNSW_OK) {
        const int h2 = lane & 1, cslot2 = lane >> 1;
        constexpr int NQR = DS > 0 ? (4 * (DS / 8) + DS % 8) : 1;
        QRegs<NQR> qreg;
        if (DS > 0) {
#pragma unroll
            for (int e = 0; e < NQR; e++) qreg.v[e] = yq[h2 * nq_half + e];
        }
        const QLds qlds{yq + h2 * nq_half};
        auto eval2 = [&](uint32_t id, bool active) -> u64 {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (active) {
                const uint4 *src = reinterpret_cast<const uint4 *>(
                    v.rows + (size_t)id * v.row_stride + (size_t)h2 * v.half_bytes);
                uint4 w[P];
#pragma unroll
                for (int p = 0; p < P; p++) w[p] = src[p];
                if (DS > 0)
                    quant_half_sums<P, DS>(w, qreg, h2, v.nch4, v.rem, acc);
                else
                    quant_half_sums<P, 0>(w, qlds, h2, v.nch4, v.rem, acc);
            }
            const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]), b2 = pair_swap(acc[2]),
                        b3 = pair_swap(acc[3]);
            float s = 0.0f;
            s += acc[0];
            s += acc[1];
            s += acc[2];
            s += acc[3];
            s += b0;
            s += b1;
            s += b2;
            s += b3;
            const float dist = __builtin_sqrtf(s);
            if (!(active && h2 == 0)) return KEY_INVALID;
            if (dist != dist) {
                status = HNSW_ERR_NAN_INPUT;
                return KEY_INVALID;
            }
            return ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | id;
        };
        auto process2 = [&](uint32_t id, bool valid, bool visit) {
            bool fresh = valid;
            if (visit) {
                bool f = false;
                if (valid && h2 == 0) f = visited_insert(htab, hmask, slots_log2, id);
                fresh = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
            }
            const u64 fm = __ballot(fresh && h2 == 0);
            if (fm == 0) return;
            n_dist += (uint32_t)__popcll(fm);
            u64 key = eval2(id, fresh);
            if (__ballot(status != HNSW_OK)) status = HNSW_ERR_NAN_INPUT;
            wl.merge(key, 1u, perm, lane);
        };
        // entry: {ep} (template.rs:316-319)
        process2(v.ep, lane < 2, false);
        for (int layer = (int)v.nb_layers - 1; status == HNSW_OK && layer >= 1; layer--) {
            for (uint32_t s = lane; s < (hslots >> 2); s += 64)
                reinterpret_cast<uint4 *>(htab)[s] =
                    make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
            wave_fence();
            if (lane == 0 && wl.n_cur > 0) {
                wl.L[0] &= KEY_MASK;
                visited_insert(htab, hmask, slots_log2, (uint32_t)wl.L[0]);
            }
            n_vis = wl.n_cur;
            wl.refresh_last(1u);
            const uint32_t S = v.S1;
            while (true) {
                const int cpos = wl.first_unexpanded(lane);
                if (cpos < 0) break;
                const uint32_t cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[0], 0);
                if (lane == 0) wl.L[0] |= KEY_EXPANDED;
                n_exp++;
                const uint32_t ub = v.upper_base[cid];
                if (ub == HX_EMPTY_SLOT) {
                    status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                    break;
                }
                const uint32_t *row = v.adj_up + ((size_t)ub + layer - 1) * S;
                uint32_t ovf = HX_EMPTY_SLOT;
                for (uint32_t c0 = 0; c0 < S; c0 += 32) {
                    const uint32_t slot = c0 + cslot2;
                    uint32_t nb = HX_EMPTY_SLOT;
                    if (slot < S) nb = row[slot];
                    const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
                    const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
                    const u64 pm = __ballot(is_ptr);
                    if (pm) ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                    const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h2 == 0));
                    if (cnt == 0) continue;
                    sum_deg += cnt;
                    if (n_vis + cnt > vis_limit) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    n_vis += cnt;
                    process2(nb, valid, true);
                    if (status != HNSW_OK) break;
                }
                if (status == HNSW_OK && ovf != HX_EMPTY_SLOT) {
                    const uint32_t lo = v.ovf_off[ovf], hi = v.ovf_off[ovf + 1];
                    for (uint32_t base = lo; base < hi; base += 32) {
                        const uint32_t i = base + cslot2;
                        const bool valid = i < hi;
                        const uint32_t nb = valid ? v.ovf_nbrs[i] : HX_EMPTY_SLOT;
                        const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h2 == 0));
                        sum_deg += cnt;
                        if (n_vis + cnt > vis_limit) {
                            status = HNSW_ERR_OVERFLOW;
                            break;
                        }
                        n_vis += cnt;
                        process2(nb, valid, true);
                        if (status != HNSW_OK) break;
                    }
                }
                if (status != HNSW_OK) break;
            }
        }
    }

    // ------------------------------------------------------------------------------------------
    // hand-over: wave 0 publishes (status, n_cur, list); both clear the visited table
    // ------------------------------------------------------------------------------------------
    wg_barrier();  // wave 1 must not touch the visited table while wave 0 walks the upper layers
    if (wv == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) perm0[64 * r + lane] = wl.L[r] & KEY_MASK;  // candidates ∪= selected
        if (lane == 0) {
            xmeta[0] = (uint32_t)status;
            xmeta[1] = wl.n_cur;
        }
    }
    for (uint32_t s = threadIdx.x; s < (hslots >> 2); s += 128)
        reinterpret_cast<uint4 *>(htab)[s] =
            make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
    wg_barrier();
    {
        status = (int32_t)xmeta[0];  // wave 0's verdict is the workgroup's
        wl.n_cur = xmeta[1];
#pragma unroll
        for (int r = 0; r < R; r++) wl.L[r] = perm0[64 * r + lane];
    }
    const uint32_t ef = max(1u, a.ef_bottom);
    if (wv == 0) {  // visited ∪= ids(selected)  (searcher.rs:33)
#pragma unroll
        for (int r = 0; r < R; r++)
            if (64u * r + lane < wl.n_cur) visited_insert(htab, hmask, slots_log2, (uint32_t)wl.L[r]);
    }
    n_vis = wl.n_cur;
    wl.refresh_last(ef);
    wg_barrier();  // also: xmeta / perm0 free for reuse

    // ------------------------------------------------------------------------------------------
    // layer 0: both waves, four lanes per slot, inline-rows blocks staged by LDS-DMA
    // ------------------------------------------------------------------------------------------
    if (status == HNSW_OK) {
        const int g = lane >> 2, j4 = lane & 3, h = j4 >> 1, sub = j4 & 1;
        const bool leader = j4 == 0;
        constexpr int NCD = DS > 0 ? (DS / 8) : 1;    // chunk dwords per half
        constexpr int NTL = DS > 0 ? (DS % 8) : 1;    // tail elements
        QRegs2<2 * NCD> qc;
        QRegs2<(NTL > 0 ? NTL : 1)> qt;
        if (DS > 0) {
#pragma unroll
            for (int i = 0; i < 2 * NCD; i++) qc.v[i] = yq[h * nq_half + 4 * (i >> 1) + 2 * sub + (i & 1)];
#pragma unroll
            for (int r = 0; r < NTL; r++) qt.v[r] = yq[v.nch4 + r];
        }
        const QcLds qc_lds{yq + h * nq_half + 2 * sub};
        const QLds qt_lds{yq + v.nch4};
        uint4 w[P];
        bool have_spec = false;
        uint32_t spec_id = 0, spec_sel = 0, par = 0;
        // DMA source of this lane within a block: this wave's 16 rows, 1-KiB pieces
        const size_t dma_off = (size_t)wv * HALF_BLK + (size_t)lane * 16;
        // where lane (g, h) reads its half row in an image
        const uint32_t rd_off = (uint32_t)g * v.row_stride + (uint32_t)h * v.half_bytes;
        constexpr int NPIECE = HALF_BLK / 1024;
        // (the hot loop holds no compiler-visible VMEM load, see the one-wave kernel)
        uint32_t ovf_pending = HX_EMPTY_SLOT;
        bool done = false;
        while (!done && status == HNSW_OK) {
        while (true) {
            STAMP(t0);
            const int cpos = wl.first_unexpanded(lane);
            if (cpos < 0) {
                done = true;
                break;
            }
            uint32_t cid = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((cpos >> 6) == r) {
                    cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], cpos & 63);
                    if (lane == (cpos & 63)) wl.L[r] |= KEY_EXPANDED;
                }
            }
            n_exp++;
            const bool hit = have_spec && spec_id == cid;
#ifdef HX_STAMPS
            if (hit) dbg_acc[6]++;
#endif
            uint32_t cur_sel = spec_sel;
            if (!hit) {
                cur_sel = spec_sel ^ 1u;
                const unsigned char *src = v.fat + (size_t)cid * v.fat_stride + dma_off;
#pragma unroll
                for (int p = 0; p < NPIECE; p++)
                    dma_piece_to_lds(src + 1024 * p, img_lds + cur_sel * HALF_BLK + 1024u * p);
            }
            have_spec = false;
            {
                const int ppos = wl.first_unexpanded(lane);
                if (ppos >= 0) {
                    uint32_t pid = 0;
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if ((ppos >> 6) == r)
                            pid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], ppos & 63);
                    spec_sel = cur_sel ^ 1u;
                    const unsigned char *src = v.fat + (size_t)pid * v.fat_stride + dma_off;
#pragma unroll
                    for (int p = 0; p < NPIECE; p++)
                        dma_piece_to_lds(src + 1024 * p, img_lds + spec_sel * HALF_BLK + 1024u * p);
                    spec_id = pid;
                    have_spec = true;
                }
            }
            if (have_spec)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(img + cur_sel * HALF_BLK + rd_off);
#pragma unroll
                for (int p = 0; p < P; p++) w[p] = src[p];
            }
            STAMP(t1);
            STAMP_ADD(0, t0, t1);
            // neighbour id: last 4 bytes of half 0, held by lanes h == 0 of the quad
            const uint32_t nb = (uint32_t)quad_bcast_i<0>((int)w[P - 1].w);
            const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
            const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
            uint32_t ovf = HX_EMPTY_SLOT;
            {
                const u64 pm = __ballot(is_ptr && leader);
                if (pm) ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
            }
            const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && leader));
            sum_deg += cnt;
            int32_t lstat = HNSW_OK;
            if (n_vis + 32 > vis_limit) lstat = HNSW_ERR_OVERFLOW;  // both waves reach the same verdict
            u64 key = KEY_INVALID;
            if (lstat == HNSW_OK && cnt != 0) {
                bool f = false;
                if (valid && leader) f = visited_insert(htab, hmask, slots_log2, nb);
                const bool fresh = quad_bcast_i<0>(f ? 1 : 0) != 0;
                const u64 fm = __ballot(fresh && leader);
                STAMP(t2);
                STAMP_ADD(1, t1, t2);
                if (fm != 0) {
                    n_dist += (uint32_t)__popcll(fm);
                    float acc[2] = {0.0f, 0.0f};
                    if (fresh) {
                        if (DS > 0)
                            quant_pair_sums<P, DS>(w, qc, qt, h, sub, v.nch4, v.rem, acc);
                        else
                            quant_pair_sums<P, 0>(w, qc_lds, qt_lds, h, sub, v.nch4, v.rem, acc);
                    }
                    // a0 .. a7 in order: lane 0 holds a0 a1, lane 1 a2 a3, lane 2 a4 a5, lane 3 a6 a7
                    float s = 0.0f;
                    s += quad_bcast<0>(acc[0]);
                    s += quad_bcast<0>(acc[1]);
                    s += quad_bcast<1>(acc[0]);
                    s += quad_bcast<1>(acc[1]);
                    s += quad_bcast<2>(acc[0]);
                    s += quad_bcast<2>(acc[1]);
                    s += quad_bcast<3>(acc[0]);
                    s += quad_bcast<3>(acc[1]);
                    const float dist = __builtin_sqrtf(s);
                    if (fresh && leader) {
                        if (dist != dist)
                            lstat = HNSW_ERR_NAN_INPUT;
                        else
                            key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | nb;
                    }
                }
            }
            if (__ballot(lstat == HNSW_ERR_NAN_INPUT)) lstat = HNSW_ERR_NAN_INPUT;
            STAMP(t3);
            STAMP_ADD(2, t1, t3);
            // ---- exchange: keys of my 16 slots, my counts / verdict / overflow pointer ----
            u64 *xk = xkeys + 32 * par;
            uint32_t *xm = xmeta + 8 * par;
            if (leader) xk[16 * wv + g] = key;
            if (lane == 0) {
                xm[4 * wv + 0] = cnt;
                xm[4 * wv + 1] = (uint32_t)lstat;
                xm[4 * wv + 2] = ovf;
            }
            wg_barrier();
            const uint32_t ow = 1u - (uint32_t)wv;
            const u64 okey = (j4 == 1) ? xk[16 * ow + g] : KEY_INVALID;
            const uint32_t ocnt = xm[4 * ow + 0];
            const int32_t ostat = (int32_t)xm[4 * ow + 1];
            const uint32_t oovf = xm[4 * ow + 2];
            par ^= 1u;
            if (lstat == HNSW_OK) lstat = ostat;
            if (lstat != HNSW_OK) {
                status = lstat;
                break;
            }
            n_vis += cnt + ocnt;
            STAMP(t4);
            STAMP_ADD(3, t3, t4);
            wl.merge(leader ? key : okey, ef, perm, lane);
            STAMP(t5);
            STAMP_ADD(5, t4, t5);
            if (ovf == HX_EMPTY_SLOT) ovf = oovf;
            if (ovf != HX_EMPTY_SLOT) {
                ovf_pending = ovf;
                break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no prefetch in flight outside the hot loop
        have_spec = false;
        if (done || status != HNSW_OK) break;
            {
                // degree > 32: the rest of the row lives in the overflow CSR (compact rows).  Wave 0
                // evaluates it, two lanes per neighbour, and publishes the keys; both waves merge.
                const uint32_t ovf = ovf_pending;
                ovf_pending = HX_EMPTY_SLOT;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)asm_ld32(v.ovf_off + ovf));
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)asm_ld32(v.ovf_off + ovf + 1));
                for (uint32_t base = lo; base < hi; base += 32) {
                    u64 *xk2 = xkeys + 32 * par;
                    uint32_t *xm2 = xmeta + 8 * par;
                    if (wv == 0) {
                        const int h2 = lane & 1;
                        const uint32_t i = base + (lane >> 1);
                        const bool ov = i < hi;
                        const uint32_t onb = ov ? asm_ld32(v.ovf_nbrs + i) : HX_EMPTY_SLOT;
                        const uint32_t c2 = (uint32_t)__popcll(__ballot(ov && h2 == 0));
                        int32_t st2 = HNSW_OK;
                        u64 k2 = KEY_INVALID;
                        if (n_vis + c2 > vis_limit) {
                            st2 = HNSW_ERR_OVERFLOW;
                        } else {
                            bool f = false;
                            if (ov && h2 == 0) f = visited_insert(htab, hmask, slots_log2, onb);
                            const bool fresh = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
                            n_dist += (uint32_t)__popcll(__ballot(fresh && h2 == 0));
                            float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                            if (fresh) {
                                const uint4 *src = reinterpret_cast<const uint4 *>(
                                    v.rows + (size_t)onb * v.row_stride + (size_t)h2 * v.half_bytes);
                                uint4 w2[P];
#pragma unroll
                                for (int p = 0; p < P; p++) w2[p] = asm_ld128(src + p);
                                const QLds ql{yq + h2 * nq_half};
                                quant_half_sums<P, 0>(w2, ql, h2, v.nch4, v.rem, acc4);
                            }
                            const float c0 = pair_swap(acc4[0]), c1 = pair_swap(acc4[1]),
                                        c2b = pair_swap(acc4[2]), c3 = pair_swap(acc4[3]);
                            float sm = 0.0f;
                            sm += acc4[0];
                            sm += acc4[1];
                            sm += acc4[2];
                            sm += acc4[3];
                            sm += c0;
                            sm += c1;
                            sm += c2b;
                            sm += c3;
                            const float dist = __builtin_sqrtf(sm);
                            if (fresh && h2 == 0) {
                                if (dist != dist)
                                    st2 = HNSW_ERR_NAN_INPUT;
                                else
                                    k2 = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | onb;
                            }
                            if (__ballot(st2 != HNSW_OK)) st2 = HNSW_ERR_NAN_INPUT;
                        }
                        sum_deg += c2;
                        if (h2 == 0) xk2[lane >> 1] = k2;
                        if (lane == 0) {
                            xm2[0] = c2;
                            xm2[1] = (uint32_t)st2;
                        }
                    }
                    wg_barrier();
                    const u64 k3 = ((lane & 1) == 0) ? xk2[lane >> 1] : KEY_INVALID;
                    const uint32_t c3 = xm2[0];
                    const int32_t st3 = (int32_t)xm2[1];
                    par ^= 1u;
                    if (st3 != HNSW_OK) {
                        status = st3;
                        break;
                    }
                    n_vis += c3;
                    wl.merge(k3, ef, perm, lane);
                }
            }
        }
    }
    // drain any prefetch still in flight before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ------------------------------------------------------------------------------------------
    // results: wave 1 hands its counters over, wave 0 writes (results.rs:59-61)
    // ------------------------------------------------------------------------------------------
    wg_barrier();
    if (wv == 1 && lane == 0) {
        xmeta[0] = n_dist;
        xmeta[1] = sum_deg;
    }
    wg_barrier();
#ifdef HX_STAMPS
    if (wv == 0 && lane == 0 && a.dbg) {
        dbg_acc[4] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 8; i++) a.dbg[(size_t)q * 8 + i] = dbg_acc[i];
    }
#endif
    if (wv == 0) {
        n_dist += xmeta[0];
        sum_deg += xmeta[1];
        const uint32_t count = status == HNSW_OK ? min(a.n, wl.n_cur) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            if (idx < a.n) {
                const bool have = idx < count;
                a.out_ids[(size_t)q * a.n + idx] = have ? (uint32_t)wl.L[r] : HX_EMPTY_SLOT;
                if (a.out_dists)
                    a.out_dists[(size_t)q * a.n + idx] =
                        have ? __builtin_bit_cast(float, (uint32_t)((wl.L[r] & KEY_MASK) >> 32))
                             : __builtin_inff();
            }
        }
        for (uint32_t idx = 64u * R + lane; idx < a.n; idx += 64) {
            a.out_ids[(size_t)q * a.n + idx] = HX_EMPTY_SLOT;
            if (a.out_dists) a.out_dists[(size_t)q * a.n + idx] = __builtin_inff();
        }
        if (lane == 0) {
            if (a.out_counts) a.out_counts[q] = count;
            hnsw_query_stats st;
            st.n_dist = n_dist;
            st.n_exp = n_exp;
            st.sum_deg = sum_deg;
            st.status = status;
            a.out_stats[q] = st;
        }
    }
}

// =============================================================================================
// On-device insertion search (HNSW::insert's first half: Inserter::build_insertion_results,
// hnsw/src/template/inserter.rs:40-126) for a BATCH of already stored points against the current
// HBM graph, one wave per point:
//   setup_insert            selected = {(ep, d(ep, p))}                      inserter.rs:53-68
//   traverse_layers_above   search_layer(ef = 1) for layers > p.level       inserter.rs:70-89
//   traverse_layers_below   per layer l <= p.level: search_layer(ef_cons), select_heuristic(m,
//                           extend_cands = keep_pruned = true), save, and the selection seeds the
//                           next layer                                       inserter.rs:91-126
// The graph is read-only during a launch: points of one batch do not see each other (the reference's
// multi-threaded insert_bulk is racy in the same way, template.rs:403-440), so a graph built this
// way is judged by recall, not by identity with the sequential build.  Deviations, all documented
// in DESIGN.md: the heuristic's candidate set is capped at the 512 nearest (the reference keeps
// all of selected ∪ their neighbours); the un-popped heuristic candidates do not leak into the
// next layer's frontier.  The edges themselves are applied by hx_connect_kernel /
// hx_remove_kernel below from the edge records this kernel files (or, in the hybrid build, on the host
// with the reference's make_connections / prune_connections / make_pruned_connections).
// =============================================================================================
template <int KIND, int DS>
__global__ void __launch_bounds__(64)
hx_insert_kernel(const DevView v, const InsertArgs a, const uint32_t slots_log2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = HX_MAX_R;  // list capacity 512: the heuristic's candidate set lives in it
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t p = a.point_ids[b];
    uint32_t *htab = reinterpret_cast<uint32_t *>(smem);
    const uint32_t hslots = 1u << slots_log2, hmask = hslots - 1;
    const uint32_t vis_limit = hslots - (hslots >> 2);
    u64 *perm = reinterpret_cast<u64 *>(smem + 4ull * hslots);
    u64 *selk = perm + 64 * R;                                   // [128] selected keys (m <= 128)
    const uint32_t yq_bytes =
        ((KIND == HNSW_VEC_QUANT8 ? 2u * (v.half_bytes - 8) * 4u : v.dim * 4u) + 15u) & ~15u;
    float *yq = reinterpret_cast<float *>(selk + 128);
    float *yqe = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(yq) + yq_bytes);
    // cooperative row gather (f32 rows of whole lines): the stage image lives in perm (4 KiB, used by the
    // merges only, never during a distance pass), the rank -> id words behind the staged rows
    unsigned char *coop_img = reinterpret_cast<unsigned char *>(perm);
    uint32_t *coop_ids = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(yqe) + yq_bytes);
    static_assert(!coop_rows<KIND, DS>() || 64 * R * 8 >= (int)HX_COOP_IMG_BYTES, "perm holds the stage image");

    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;
    constexpr int CHUNK = 64 / LPC;
    const int h = (LPC == 2) ? (lane & 1) : 0;
    const int cslot = lane / LPC;
    const bool first = (LPC == 1) || (h == 0);
    int32_t status = HNSW_OK;
    uint32_t n_vis = 0;
    // what this point's insertion reads (wave-uniform; summed into a.counters at the end): vector rows
    // (distance evaluations + staged rows), adjacency rows and the ids in them -- the build's algorithmic bytes
    uint32_t c_rows = 1, c_adj = 0, c_ids = 0;

    const uint32_t level = min((uint32_t)a.levels[p], v.nb_layers - 1);
    const uint32_t m = a.m, ef_cons = max(1u, a.ef_cons);
    // outputs of this point: [max_layers][m], padded
    uint32_t *o_ids = a.out_ids + (size_t)b * a.max_layers * m;
    float *o_d = a.out_dists + (size_t)b * a.max_layers * m;
    for (uint32_t i = lane; i < a.max_layers * m; i += 64) {
        o_ids[i] = HX_EMPTY_SLOT;
        o_d[i] = __builtin_inff();
    }

    stage_row<KIND>(v, p, yq, lane);

    WaveList<R> wl;
#pragma unroll
    for (int r = 0; r < R; r++) wl.L[r] = KEY_INVALID;
    wl.n_cur = 0;
    wl.last_key = KEY_INVALID;

    // one pass over up to CHUNK ids: optional visited filter, distance to the staged row, merge
    auto process = [&](uint32_t id, bool valid, bool visit, uint32_t ef_l, u64 new_flag) __attribute__((always_inline)) {
        bool fresh = valid;
        if (visit) {
            bool f = false;
            if (valid && first) f = visited_insert(htab, hmask, slots_log2, id);
            if (LPC == 2) f = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
            fresh = f;
        }
        const u64 fm = __ballot(fresh && first);
        if (visit) n_vis += (uint32_t)__popcll(fm);  // what the table really holds
        if (fm == 0) return;
        c_rows += (uint32_t)__popcll(fm);
        const float dist = dist_build<KIND, DS, true>(v, id, fresh, h, yq, coop_ids, coop_img, lane);
        u64 key = KEY_INVALID;
        bool nan = false;
        if (fresh && first) {
            nan = dist != dist;
            if (!nan) key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | id;
        }
        if (__ballot(nan)) status = HNSW_ERR_NAN_INPUT;
        // The list has eight registers for the heuristic's 512 candidates, and a merge pays its rank / scatter
        // work per register.  Entries beyond min(n_cur + batch, ef) cannot exist before or after this merge, so it
        // runs over the registers that can hold something: one for the searches (ef = 1 above the point's level,
        // ef_cons <= 64 below), two or four while the candidate set is filling.
        const uint32_t reach = min(wl.n_cur + (uint32_t)__popcll(fm), ef_l);
        if (reach <= 64u)
            merge_prefix<1>(wl, key, ef_l, perm, lane, new_flag);
        else if (reach <= 128u)
            merge_prefix<2>(wl, key, ef_l, perm, lane, new_flag);
        else if (reach <= 256u)
            merge_prefix<4>(wl, key, ef_l, perm, lane, new_flag);
        else
            wl.merge(key, ef_l, perm, lane, new_flag);
    };
    // expand every unexpanded entry of the list on `layer` (search_layer's loop, searcher.rs:35-95)
    auto expand_all = [&](int layer, uint32_t ef_l, u64 new_flag) __attribute__((always_inline)) {
        const uint32_t S = layer == 0 ? v.S0 : v.S1;
        // the entry at cpos: marked expanded, its id returned
        auto take = [&](int cpos) __attribute__((always_inline)) -> uint32_t {
            uint32_t cid = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((cpos >> 6) == r) {
                    cid = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl.L[r], cpos & 63);
                    if (lane == (cpos & 63)) wl.L[r] |= KEY_EXPANDED;
                }
            }
            return cid;
        };
        auto row_of = [&](uint32_t cid) __attribute__((always_inline)) -> const uint32_t * {
            if (layer == 0) return v.adj0 + (size_t)cid * S;
            const uint32_t ub = v.upper_base[cid];
            if (ub == HX_EMPTY_SLOT) {
                status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                return nullptr;
            }
            return v.adj_up + ((size_t)ub + layer - 1) * S;
        };
        // The heuristic's extension (new entries are born expanded) expands a FIXED set -- the entries the search
        // left -- and keeps the union of their neighbours: the order does not matter, so with one lane per row
        // (f32) and rows of up to 32 slots TWO entries go through one pass, one per half wave, instead of
        // leaving the upper half idle (the visited insert settles an id both rows hold).
        const bool two = LPC == 1 && new_flag != 0 && S <= 32;
        while (status == HNSW_OK) {
            const int cpos = wl.first_unexpanded(lane);
            if (cpos < 0) break;
            const uint32_t cid = take(cpos);
            const uint32_t *row = row_of(cid);
            if (row == nullptr) break;
            c_adj++;
            if (two) {
                const int cpos2 = wl.first_unexpanded(lane);
                const uint32_t *row2 = nullptr;
                if (cpos2 >= 0) {
                    row2 = row_of(take(cpos2));
                    if (row2 == nullptr) break;
                    c_adj++;
                }
                const bool upper = lane >= 32;
                const uint32_t slot = (uint32_t)lane & 31u;
                uint32_t nb = HX_EMPTY_SLOT;
                if (slot < S && (!upper || row2 != nullptr)) nb = (upper ? row2 : row)[slot];
                const bool valid = nb != HX_EMPTY_SLOT && !(nb & HX_OVF_FLAG) && nb != p;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
                c_ids += cnt;
                if (cnt == 0) continue;
                if (n_vis + cnt > vis_limit) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                process(nb, valid, true, ef_l, new_flag);
                continue;
            }
            for (uint32_t c0 = 0; c0 < S && status == HNSW_OK; c0 += CHUNK) {
                const uint32_t slot = c0 + cslot;
                uint32_t nb = HX_EMPTY_SLOT;
                if (slot < S) nb = row[slot];
                // during a build the device rows never carry overflow pointers (rows are truncated
                // to the stride when they are scattered); a flagged id is skipped
                const bool valid = nb != HX_EMPTY_SLOT && !(nb & HX_OVF_FLAG) && nb != p;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && first));
                c_ids += cnt;
                if (cnt == 0) continue;
                if (n_vis + cnt > vis_limit) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                process(nb, valid, true, ef_l, new_flag);
            }
        }
    };
    // start a layer: visited.clear(), candidates ∪= selected, visited ∪= ids(selected)
    auto begin_layer = [&](uint32_t ef_l) __attribute__((always_inline)) {
        for (uint32_t s = lane; s < (hslots >> 2); s += 64)
            reinterpret_cast<uint4 *>(htab)[s] =
                make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
        wave_fence();
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (64u * r + lane < wl.n_cur) {
                wl.L[r] &= KEY_MASK;
                visited_insert(htab, hmask, slots_log2, (uint32_t)wl.L[r]);
            }
        }
        n_vis = wl.n_cur;
        wl.refresh_last(ef_l);
    };

    if (p == v.ep || p >= v.n_points) status = HNSW_ERR_ARG;  // the host never sends the entry point
    if (status == HNSW_OK) process(v.ep, lane < LPC, false, 1u, 0);  // setup_insert

    for (int layer = (int)v.nb_layers - 1; status == HNSW_OK && layer >= 0; layer--) {
        if ((uint32_t)layer > level) {  // traverse_layers_above
            begin_layer(1u);
            expand_all(layer, 1u, 0);
            continue;
        }
        // ---- search_layer(ef_cons) ----
        begin_layer(ef_cons);
        expand_all(layer, ef_cons, 0);
        if (status != HNSW_OK) break;
        // ---- select_heuristic: candidates = selected ∪ neighbours(selected), distances to p
        // (results.rs:105-146).  Every current entry is expanded once more, this time keeping ALL
        // distinct neighbours (cap 512 nearest); entries that arrive now are born expanded.
        begin_layer(64u * R);
        expand_all(layer, 64u * R, KEY_EXPANDED);
        if (status != HNSW_OK) break;
        // The reference pops the candidates in ascending order and accepts e iff (d(e,p), e) <
        // (d(e,s), s) for every s selected so far (searcher.rs:128-139).  Equivalent, and parallel:
        // whenever a candidate s is selected, every LATER candidate e with (d(s,e), s) < (d(e,p), e)
        // is marked rejected (d is bit-symmetric); the next selection is the first unmarked one.
        // Each round stages s once and evaluates up to 64 / LPC candidates per pass.
        const uint32_t n_c = wl.n_cur;
        uint32_t ns = 0;
        uint32_t selbits = 0;  // bit r: the candidate at position 64 r + lane was selected
#pragma unroll
        for (int r = 0; r < R; r++) wl.L[r] &= KEY_MASK;  // the flag now means "rejected"
        uint32_t cursor = 0;
        // One sweep: the staged selected point (yqe, id sid) against the open candidates at positions [lo, hi):
        // those it dominates -- (d(s, e), s) < (d(e, p), e) -- are marked rejected.
        // (a rolled loop over the list registers with static selects: one copy of the distance
        // code instead of R x LPC, and the list stays in registers)
        auto sweep = [&](uint32_t sid, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
#pragma unroll 1
            for (int rr = 0; rr < R; rr++) {
                if (64u * rr + 64u <= lo || 64u * rr >= hi) continue;
                u64 mine = KEY_INVALID;
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (r == rr) mine = wl.L[r];
                const uint32_t idx = 64u * rr + lane;
                const bool open = idx >= lo && idx < hi && (mine & KEY_EXPANDED) == 0;
                if (__ballot(open) == 0) continue;
                const uint32_t my_id = (uint32_t)mine, my_db = (uint32_t)(mine >> 32);
                bool mark = false;
#pragma unroll
                for (int half = 0; half < LPC; half++) {
                    const int src = half * CHUNK + cslot;  // the lane that owns this pass's candidate
                    const uint32_t cid = (uint32_t)__shfl((int)my_id, src);
                    const uint32_t cdb = (uint32_t)__shfl((int)my_db, src);
                    const bool act = __shfl(open ? 1 : 0, src) != 0;
                    if (__ballot(act) == 0) continue;
                    c_rows += (uint32_t)__popcll(__ballot(act && first));
                    const float dist = dist_build<KIND, DS, true>(v, cid, act, h, yqe, coop_ids, coop_img, lane);
                    bool rej = false;
                    if (act && first) {
                        if (dist != dist) status = HNSW_ERR_NAN_INPUT;
                        const u64 sk = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | sid;
                        const u64 ck = ((u64)cdb << 32) | cid;
                        rej = sk < ck;
                    }
                    const u64 rm = __ballot(rej);  // bit LPC * cslot of the pass <-> owner lane src
                    const int own = lane - half * CHUNK;
                    if (own >= 0 && own < CHUNK && ((rm >> (LPC * own)) & 1)) mark = true;
                }
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (r == rr && mark) wl.L[r] |= KEY_EXPANDED;
            }
        };
        // The candidate set holds up to 512 entries but m selections usually come out of the first hundred:
        // a selected point sweeps only the WINDOW [0, win_end) of candidates; when the window holds nothing
        // unpopped and fewer than m are selected, it grows by 64 and the points selected so far sweep the new
        // part first.  Every candidate is still judged against every point selected before it is popped, so the
        // selection is the one the whole-set sweep made (round 2: every selection swept all 512 -- eight distance
        // passes per selection, most of them for candidates that are never reached).
        uint32_t win_end = min(n_c, 128u);
        while (ns < m && status == HNSW_OK) {
            int pos = -1;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t idx = 64u * r + lane;
                const u64 mk = __ballot(idx >= cursor && idx < win_end && (wl.L[r] & KEY_EXPANDED) == 0);
                if (pos < 0 && mk) pos = 64 * r + (__ffsll((long long)mk) - 1);
            }
            if (pos < 0) {
                if (win_end >= n_c) break;  // every candidate was popped
                const uint32_t new_end = min(n_c, win_end + 64u);
                for (uint32_t k2 = 0; k2 < ns && status == HNSW_OK; k2++) {  // catch up: the new part against the selected
                    const uint32_t sid2 = (uint32_t)selk[k2];
                    c_rows++;
                    stage_row<KIND>(v, sid2, yqe, lane);
                    sweep(sid2, win_end, new_end);
                }
                cursor = win_end;
                win_end = new_end;
                if (__ballot(status != HNSW_OK)) status = HNSW_ERR_NAN_INPUT;
                continue;
            }
            u64 sk_sel = 0;
#pragma unroll
            for (int r = 0; r < R; r++)
                if ((pos >> 6) == r) {
                    sk_sel = readlane64(wl.L[r], pos & 63);
                    if (lane == (pos & 63)) selbits |= 1u << r;
                }
            if (lane == 0) selk[ns] = sk_sel;
            ns++;
            cursor = (uint32_t)pos + 1;
            wave_fence();
            if (ns >= m || (cursor >= win_end && win_end >= n_c)) continue;  // nothing left to decide
            const uint32_t sid = (uint32_t)sk_sel;
            c_rows++;
            stage_row<KIND>(v, sid, yqe, lane);
            sweep(sid, cursor, win_end);
            if (__ballot(status != HNSW_OK)) status = HNSW_ERR_NAN_INPUT;
        }
        // keep_pruned: fill up from the rejected candidates in ascending order (searcher.rs:141-146);
        // only reached with ns < m when every candidate was popped
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = 64u * r + lane;
            u64 mk = __ballot(idx < n_c && (wl.L[r] & KEY_EXPANDED) != 0 && ((selbits >> r) & 1u) == 0);
            while (mk && ns < m && status == HNSW_OK) {
                const int j = __ffsll((long long)mk) - 1;
                mk &= mk - 1;
                const u64 ek = readlane64(wl.L[r], j) & KEY_MASK;
                if (lane == 0) selk[ns] = ek;
                ns++;
            }
        }
        wave_fence();
        // save_layer_results + the selection seeds the next layer (up to 128 selected: two rounds of 64)
#pragma unroll
        for (int r = 0; r < R; r++) wl.L[r] = KEY_INVALID;
        wl.n_cur = 0;
        wl.last_key = KEY_INVALID;
        for (uint32_t j0 = 0; j0 < max(ns, 1u); j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const u64 mine = j < ns ? selk[j] : KEY_INVALID;
            if (j < ns) {
                o_ids[(size_t)layer * m + j] = (uint32_t)mine;
                o_d[(size_t)layer * m + j] = __builtin_bit_cast(float, (uint32_t)(mine >> 32));
            }
            wl.merge(mine, max(ns, 1u), perm, lane);
        }
    }
    if (__ballot(status != HNSW_OK)) {
        int32_t st = status;
        for (int o = 32; o > 0; o >>= 1) st = min(st, __shfl_xor(st, o));
        status = st;
    }
    if (a.req_keys != nullptr && status == HNSW_OK) {
        // on-device connect: only a point whose every layer succeeded writes its own rows (nobody can
        // reach p yet) and files one reverse-edge request per selected neighbour
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t total = 0;
        for (uint32_t l = 0; l <= level; l++) {
            for (uint32_t j0 = 0; j0 < m; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane;
                const uint32_t id = j < m ? o_ids[(size_t)l * m + j] : HX_EMPTY_SLOT;
                total += (uint32_t)__popcll(__ballot(id != HX_EMPTY_SLOT));
            }
        }
        // emit_own (sharded build): the point's own rows travel as records too -- (layer, p <- n) next
        // to (layer, n <- p) -- so that the record list alone carries the whole batch to every replica
        const uint32_t per_edge = a.emit_own ? 2u : 1u;
        total *= per_edge;
        // Reserve `total` record slots with ONE atomic add.  Round 2 reserved by compare-and-swap so that the counter
        // never passed the last written record; 8192 waves retrying on one word made that loop 85 % of the insert
        // kernel (1M points: 2.21 s against 0.34 s; 90 % of a wave's life in SQ_WAIT_ANY).  The add keeps the
        // guarantee another way: once a reservation does not fit, the counter is beyond the capacity for good and
        // every later one fails too, so the records written are exactly the prefix [0, B) where B is the base of the
        // first failing reservation -- the smallest failing base, kept in *req_fail_base (atomic min; the host
        // starts it at 0xFFFFFFFF and takes min(counter, B) as the record count).
        uint32_t base = 0xFFFFFFFFu;
        if (lane == 0) {
            base = atomicAdd(a.req_count, total);
            if ((uint64_t)base + total > a.req_cap) {
                atomicMin(a.req_fail_base, base);
                base = 0xFFFFFFFFu;
            }
        }
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base == 0xFFFFFFFFu) {
            status = HNSW_ERR_OVERFLOW;  // nothing written, nothing reserved: the point takes the CPU path
        } else {
            for (uint32_t l = 0; l <= level; l++) {
                const uint32_t S = l == 0 ? v.S0 : v.S1;
                // the layer's selection is a prefix of its m output slots: cnt of them are filled
                uint32_t cnt = 0;
                for (uint32_t j0 = 0; j0 < m; j0 += 64) {
                    const uint32_t j = j0 + (uint32_t)lane;
                    cnt += (uint32_t)__popcll(__ballot(j < m && o_ids[(size_t)l * m + j] != HX_EMPTY_SLOT));
                }
                for (uint32_t j0 = 0; j0 < max(S, m); j0 += 64) {
                    const uint32_t j = j0 + (uint32_t)lane;
                    const uint32_t id = j < m ? o_ids[(size_t)l * m + j] : HX_EMPTY_SLOT;
                    if (!a.emit_own) {
                        const size_t at = l == 0 ? (size_t)p * S : ((size_t)v.upper_base[p] + l - 1) * S;
                        uint32_t *row = (l == 0 ? a.adj0_mut : a.adj_up_mut) + at;
                        if (j < S) row[j] = id;
                        uint32_t *rowd = l == 0 ? a.adjd0_mut : a.adjd_up_mut;
                        if (rowd != nullptr && j < S)  // the edge's distance travels with it (ConnectArgs)
                            rowd[at + j] = id != HX_EMPTY_SLOT ? __builtin_bit_cast(uint32_t, o_d[(size_t)l * m + j]) : 0xFFFFFFFFu;
                    }
                    if (id != HX_EMPTY_SLOT) {
                        const uint32_t db = __builtin_bit_cast(uint32_t, o_d[(size_t)l * m + j]);
                        a.req_keys[base + j] = hx_edge_key(l, id, p);
                        a.req_vals[base + j] = db;
                        if (a.emit_own) {
                            a.req_keys[base + cnt + j] = hx_edge_key(l, p, id);
                            a.req_vals[base + cnt + j] = db;
                        }
                    }
                }
                base += cnt * per_edge;
            }
        }
    }
    if (lane == 0) a.out_status[b] = status;
    if (a.counters != nullptr && lane == 0) {
        atomicAdd(a.counters + 0, (unsigned long long)c_rows);
        atomicAdd(a.counters + 1, (unsigned long long)c_adj);
        atomicAdd(a.counters + 2, (unsigned long long)c_ids);
    }
}

// rows[row_index[i]] = data[i] for whole adjacency rows of S slots (dirty rows after a build batch)
__global__ void __launch_bounds__(64)
hx_scatter_rows_kernel(uint32_t *dst, uint32_t S, const uint32_t *row_index, const uint32_t *data,
                       uint32_t n) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    uint32_t *out = dst + (size_t)row_index[i] * S;
    for (uint32_t k = threadIdx.x; k < S; k += 64) out[k] = data[(size_t)i * S + k];
}

// First-attempt size of the insert kernel's visited table, relative to the standard one: - 1 (2048 slots, 8 KiB)
// for ef_construction <= 32 on 32-slot rows.  LDS is what limits the insert kernel's waves per CU (6 with the
// standard 16-KiB table at d = 256, 10 with 8 KiB): 16M x 256d, insert kernel 12.3 -> 10.1 s with 29 points of
// 16M filling the small table (they run again with adjust + 1; round 3).
int insert_table_first_adjust(const DevView &v, const InsertArgs &a) {
    if (const char *e = getenv("HNSW_MI355X_INSERT_TABLE_ADJUST")) return atoi(e);  // A/B runs
    return (a.ef_cons <= 32 && v.S0 <= 32) ? -1 : 0;
}

int launch_insert(const DevView &v, const InsertArgs &a, uint32_t nblocks, hipStream_t stream, int table_adjust) {
    if (nblocks == 0) return HNSW_OK;
    if (a.m > 128 || a.m == 0 || a.ef_cons > 64 * HX_MAX_R) {
        set_error("on-device build supports m <= 128 and ef_construction <= 512");
        return HNSW_ERR_ARG;
    }
    // visited table: what ef_cons list entries with rows of S0 slots visit (m <= 32: 4096 / 8192 / 16384 slots as
    // before; the 128- and 256-slot rows of m = 64 / 128 take up to 32768 slots = 128 KiB, one wave per CU -- a
    // point that still fills it takes the CPU path after the build, like every point whose search fails)
    uint32_t slots_log2 = 12 + (a.ef_cons > 64 ? 1 : 0) + (a.ef_cons > 160 ? 1 : 0);
    if (v.S0 > 64) slots_log2 = std::max(slots_log2, std::min(15u, default_slots_log2(a.ef_cons, v.S0)));
    // table_adjust: the device-connect build first runs a batch with HALF the table where that buys waves per
    // CU (insert_table_first_adjust) and runs the few points that fill it again with a larger one
    slots_log2 = (uint32_t)std::min(15, std::max(9, (int)slots_log2 + table_adjust));
    const size_t yq_bytes =
        ((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull;
    const size_t lds = (4ull << slots_log2) + 64ull * HX_MAX_R * 8 + 128 * 8 + 2 * yq_bytes + 256 /* rank -> id words */;
    if (lds > 160 * 1024) {
        set_error("insert kernel needs %zu bytes of LDS", lds);
        return HNSW_ERR_ARG;
    }
    // the configs[1] dimension gets compile-time row loops
    void (*kfn)(const DevView, const InsertArgs, const uint32_t);
    if (v.kind == HNSW_VEC_QUANT8)
        kfn = v.dim == 100   ? hx_insert_kernel<HNSW_VEC_QUANT8, 100>
              : v.dim == 128 ? hx_insert_kernel<HNSW_VEC_QUANT8, 128>
              : v.dim == 256 ? hx_insert_kernel<HNSW_VEC_QUANT8, 256>
              : v.dim == 768 ? hx_insert_kernel<HNSW_VEC_QUANT8, 768>
                             : hx_insert_kernel<HNSW_VEC_QUANT8, 0>;
    else
        kfn = v.dim == 100                           ? hx_insert_kernel<HNSW_VEC_F32, 100>
              : v.dim == 128                         ? hx_insert_kernel<HNSW_VEC_F32, 128>
              : v.dim == 256 && v.row_stride == 1024 ? hx_insert_kernel<HNSW_VEC_F32, 256>  // the configs[4] dimension
              : v.dim == 768 && v.row_stride == 3072 ? hx_insert_kernel<HNSW_VEC_F32, 768>  // the configs[2] dimension
                                                     : hx_insert_kernel<HNSW_VEC_F32, 0>;
    const void *kern = reinterpret_cast<const void *>(kfn);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(64), lds, stream, v, a, slots_log2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("insert kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

// ---------------------------------------------------------------------------------------------
// On-device connect.  Edge records hx_edge_key(layer, row node, other node) arrive radix-sorted, so
// the records of one adjacency row are adjacent; one wave is launched per record, the wave of a
// row's first record owns the row for the phase and the others exit at once.  Rows are owned by
// exactly one wave per phase: no locks, no atomics on rows, deterministic for a given batch.
//
// Phase 2 (hx_connect_kernel): make_connections adds the sources to the row (template.rs:196-207); a
// row that would exceed the layer's cap is pruned to its `cap` nearest by (dist, id)
// (prune_connections / select_simple, template.rs:209-238,614-621) -- distances of the existing
// neighbours are evaluated here, the sources bring d(p, n) = d(n, p).  Every dropped neighbour x (and
// every source that did not make it) is reported so that phase 3 removes the reverse edge
// (remove_edge is symmetric, graph.rs:72-83).
// ---------------------------------------------------------------------------------------------
static constexpr uint64_t HX_EDGE_ID_MASK = (1ull << HX_EDGE_ID_BITS) - 1;

// number of records of the row that starts at record i (0 if i is not the first of its row)
__device__ __forceinline__ uint32_t edge_group_size(const uint64_t *keys, uint32_t count, uint32_t i, int lane) {
    const uint64_t prefix = keys[i] >> HX_EDGE_ID_BITS;
    if (i > 0 && (keys[i - 1] >> HX_EDGE_ID_BITS) == prefix) return 0;
    uint32_t k = 1;
    for (;;) {
        const uint32_t j = i + k + lane;
        const bool same = j < count && (keys[j] >> HX_EDGE_ID_BITS) == prefix;
        const u64 diff = ~__ballot(same);
        if (diff) return k + (uint32_t)(__ffsll((long long)diff) - 1);
        k += 64;
    }
}

// RS = registers per lane that hold one adjacency row (slot 64 r + lane): 1 for rows of up to 64 slots
// (m <= 32), 2 / 4 for the 128- / 256-slot layer-0 rows of m = 64 / 128 (the reference's own build benches
// use M in {32, 64, 128}, hnsw/benches/hnsw_benchmarks.rs:7)
template <int KIND, int DS, int RS>
__global__ void __launch_bounds__(64)
hx_connect_kernel(const DevView v, const ConnectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *perm = reinterpret_cast<u64 *>(smem);       // [64 RS]
    u64 *ekeys = perm + 64 * RS;                      // [64 RS] keys of the existing neighbours
    float *yq = reinterpret_cast<float *>(ekeys + 64 * RS);
    const int lane = threadIdx.x;
    const uint32_t lo = blockIdx.x;
    if (lo >= a.count) return;
    const uint32_t k = edge_group_size(a.keys, a.count, lo, lane);
    if (k == 0) return;
    const uint64_t head = a.keys[lo];
    const uint32_t n = (uint32_t)((head >> HX_EDGE_ID_BITS) & HX_EDGE_ID_MASK);
    const uint32_t layer = (uint32_t)(head >> (2 * HX_EDGE_ID_BITS));
    const uint32_t S = layer == 0 ? v.S0 : v.S1;
    const uint32_t cap = layer == 0 ? 2 * a.m : a.m;
    if (n >= v.n_points || layer >= v.nb_layers || (layer > 0 && v.upper_base[n] == HX_EMPTY_SLOT) || S > 64u * RS) {
        *a.status = HNSW_ERR_NODE_NOT_IN_GRAPH;  // a malformed record: never touch memory for it
        return;
    }
    if (a.own_world > 1) {  // sharded build: this row has ONE owner among the ranks
        if (n % a.own_world != a.own_rank) return;
        if (lane == 0) {     // its new contents will travel to the other replicas (64 lists: one counter would serialise)
            const uint32_t seg = blockIdx.x & (HX_CHG_LISTS - 1), at = atomicAdd(a.chg_count + seg, 1u);
            if (at < a.chg_cap)
                a.chg_keys[(size_t)seg * a.chg_cap + at] = hx_edge_key(layer, n, 0);
            else
                *a.status = HNSW_ERR_OVERFLOW;
        }
    }
    const size_t row_at = layer == 0 ? (size_t)n * S : ((size_t)v.upper_base[n] + layer - 1) * S;
    uint32_t *row = (layer == 0 ? a.adj0_mut : a.adj_up_mut) + row_at;
    uint32_t *rowd = layer == 0 ? a.adjd0_mut : a.adjd_up_mut;  // the edges' distances, or null
    if (rowd != nullptr) rowd += row_at;
    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;
    constexpr int CHUNK = 64 / LPC;
    const int h = (LPC == 2) ? (lane & 1) : 0;
    const int cslot = lane / LPC;
    const bool first = (LPC == 1) || (h == 0);

    uint32_t cur[RS], curd[RS];
    u64 hm[RS];
    uint32_t deg = 0;
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const uint32_t slot = 64u * r + (uint32_t)lane;
        cur[r] = slot < S ? row[slot] : HX_EMPTY_SLOT;
        curd[r] = (rowd != nullptr && slot < S) ? rowd[slot] : 0xFFFFFFFFu;
        hm[r] = __ballot(cur[r] != HX_EMPTY_SLOT);
        deg += (uint32_t)__popcll(hm[r]);
    }
    if (deg + k <= cap && deg + k <= S) {  // room for every source: append
        uint32_t before = 0;               // (all reads of the row happened above)
#pragma unroll
        for (int r = 0; r < RS; r++) {
            if (cur[r] != HX_EMPTY_SLOT) {
                const uint32_t at = before + (uint32_t)__popcll(hm[r] & ((1ull << lane) - 1));
                row[at] = cur[r];
                if (rowd != nullptr) rowd[at] = curd[r];
            }
            before += (uint32_t)__popcll(hm[r]);
        }
        for (uint32_t j = lane; j < k; j += 64) {
            row[deg + j] = (uint32_t)(a.keys[lo + j] & HX_EDGE_ID_MASK);
            if (rowd != nullptr) rowd[deg + j] = a.vals[lo + j];  // d(source, n) = d(n, source)
        }
        for (uint32_t j = deg + k + lane; j < S; j += 64) {
            row[j] = HX_EMPTY_SLOT;
            if (rowd != nullptr) rowd[j] = 0xFFFFFFFFu;
        }
        return;
    }
    // ---- prune: keep the `cap` nearest of existing ∪ sources ----
    // the node's own row is staged only when some existing neighbour's distance is not known yet
    bool any_unknown = false;
#pragma unroll
    for (int r = 0; r < RS; r++) any_unknown |= __ballot(cur[r] != HX_EMPTY_SLOT && curd[r] == 0xFFFFFFFFu) != 0;
    if (any_unknown) stage_row<KIND>(v, n, yq, lane);
    WaveList<RS> wl;
#pragma unroll
    for (int r = 0; r < RS; r++) {
        wl.L[r] = KEY_INVALID;
        ekeys[64 * r + lane] = KEY_INVALID;
    }
    wl.n_cur = 0;
    wl.last_key = KEY_INVALID;
    wave_fence();
    for (uint32_t c0 = 0; c0 < S; c0 += CHUNK) {  // existing neighbours, CHUNK at a time
        const uint32_t slot = c0 + cslot;
        uint32_t id = HX_EMPTY_SLOT, dbits = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < RS; r++) {
            const uint32_t t = (uint32_t)__shfl((int)cur[r], (int)(slot & 63));
            const uint32_t td = (uint32_t)__shfl((int)curd[r], (int)(slot & 63));
            if ((slot >> 6) == (uint32_t)r) {
                id = t;
                dbits = td;
            }
        }
        const bool act = slot < S && id < v.n_points;
        if (slot < S && id != HX_EMPTY_SLOT && id >= v.n_points) *a.status = HNSW_ERR_NODE_NOT_IN_GRAPH;
        const bool need = act && dbits == 0xFFFFFFFFu;  // (an edge that predates the build: evaluated once, kept from here on)
        if (__ballot(need) != 0) {
            const float dist = dist_build<KIND, DS>(v, id, need, h, yq);
            if (need) dbits = __builtin_bit_cast(uint32_t, dist);
        }
        u64 key = KEY_INVALID;
        if (act && first) {
            key = ((u64)dbits << 32) | id;
            ekeys[slot] = key;
        }
        wl.merge(key, cap, perm, lane);
    }
    auto source_key = [&](uint32_t j) -> u64 {
        return j < k ? ((u64)a.vals[lo + j] << 32) | (uint32_t)(a.keys[lo + j] & HX_EDGE_ID_MASK) : KEY_INVALID;
    };
    for (uint32_t j0 = 0; j0 < k; j0 += 64) wl.merge(source_key(j0 + lane), cap, perm, lane);
    wave_fence();
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const uint32_t slot = 64u * r + (uint32_t)lane;
        if (slot < S) {
            row[slot] = slot < wl.n_cur ? (uint32_t)wl.L[r] : HX_EMPTY_SLOT;
            if (rowd != nullptr) rowd[slot] = slot < wl.n_cur ? (uint32_t)(wl.L[r] >> 32) : 0xFFFFFFFFu;
        }
    }
    // report what fell out: key > the last kept key (keys are distinct)
    const u64 lastk = wl.n_cur >= cap ? wl.last_key : KEY_INVALID;
    auto emit = [&](u64 key) {
        const bool drop = key != KEY_INVALID && key > lastk;
        const u64 dm = __ballot(drop);
        if (dm == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.out_count, (uint32_t)__popcll(dm));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t at = base + (uint32_t)__popcll(dm & ((1ull << lane) - 1));
        if (drop) {
            if (at < a.out_cap)
                a.out_keys[at] = hx_edge_key(layer, (uint32_t)key, n);
            else
                *a.status = HNSW_ERR_OVERFLOW;
        }
    };
#pragma unroll
    for (int r = 0; r < RS; r++) emit(ekeys[64 * r + lane]);
    for (uint32_t j0 = 0; j0 < k; j0 += 64) emit(source_key(j0 + lane));
}

// ---------------------------------------------------------------------------------------------
// Phase 3 (hx_remove_kernel): the row of x drops the neighbours that dropped x in phase 2.  An edge
// to x's LAST neighbour is kept (isolate_node, graph.rs:85-94): such a refusal is reported and the
// host restores the reverse direction after the build.
// ---------------------------------------------------------------------------------------------
template <int RS>
__global__ void __launch_bounds__(64)
hx_remove_kernel(const DevView v, const ConnectArgs a) {
    const int lane = threadIdx.x;
    const uint32_t lo = blockIdx.x;
    if (lo >= a.count) return;
    const uint32_t k = edge_group_size(a.keys, a.count, lo, lane);
    if (k == 0) return;
    const uint64_t head = a.keys[lo];
    const uint32_t x = (uint32_t)((head >> HX_EDGE_ID_BITS) & HX_EDGE_ID_MASK);
    const uint32_t layer = (uint32_t)(head >> (2 * HX_EDGE_ID_BITS));
    const uint32_t S = layer == 0 ? v.S0 : v.S1;
    if (x >= v.n_points || layer >= v.nb_layers || (layer > 0 && v.upper_base[x] == HX_EMPTY_SLOT) || S > 64u * RS) {
        *a.status = HNSW_ERR_NODE_NOT_IN_GRAPH;
        return;
    }
    if (a.own_world > 1) {  // sharded build: x's row is dropped from by its owner only, and shipped afterwards
        if (x % a.own_world != a.own_rank) return;
        if (lane == 0) {
            const uint32_t seg = blockIdx.x & (HX_CHG_LISTS - 1), at = atomicAdd(a.chg_count + seg, 1u);
            if (at < a.chg_cap)
                a.chg_keys[(size_t)seg * a.chg_cap + at] = hx_edge_key(layer, x, 0);
            else
                *a.status = HNSW_ERR_OVERFLOW;
        }
    }
    const size_t row_at = layer == 0 ? (size_t)x * S : ((size_t)v.upper_base[x] + layer - 1) * S;
    uint32_t *row = (layer == 0 ? a.adj0_mut : a.adj_up_mut) + row_at;
    uint32_t *rowd = layer == 0 ? a.adjd0_mut : a.adjd_up_mut;  // the edges' distances move with their ids
    if (rowd != nullptr) rowd += row_at;
    uint32_t cur[RS], curd[RS];
    uint32_t deg = 0;
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const uint32_t slot = 64u * r + (uint32_t)lane;
        cur[r] = slot < S ? row[slot] : HX_EMPTY_SLOT;
        curd[r] = (rowd != nullptr && slot < S) ? rowd[slot] : 0xFFFFFFFFu;
        deg += (uint32_t)__popcll(__ballot(cur[r] != HX_EMPTY_SLOT));
    }
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t nb = (uint32_t)(a.keys[lo + j] & HX_EDGE_ID_MASK);
        u64 hit = 0;
#pragma unroll
        for (int r = 0; r < RS; r++) hit |= __ballot(cur[r] == nb);
        if (hit == 0) continue;
        if (deg == 1) {  // the last edge stays
            if (lane == 0) {
                const uint32_t at = atomicAdd(a.out_count, 1u);
                if (at < a.out_cap)
                    a.out_keys[at] = hx_edge_key(layer, x, nb);
                else
                    *a.status = HNSW_ERR_OVERFLOW;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < RS; r++)
            if (cur[r] == nb) cur[r] = HX_EMPTY_SLOT;
        deg--;
    }
    // compact: survivors to the front, every slot written by exactly one lane
    uint32_t before = 0;
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const u64 hm = __ballot(cur[r] != HX_EMPTY_SLOT);
        if (cur[r] != HX_EMPTY_SLOT) {
            const uint32_t at = before + (uint32_t)__popcll(hm & ((1ull << lane) - 1));
            row[at] = cur[r];
            if (rowd != nullptr) rowd[at] = curd[r];
        }
        before += (uint32_t)__popcll(hm);
    }
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const uint32_t slot = 64u * r + (uint32_t)lane;
        if (slot >= before && slot < S) {
            row[slot] = HX_EMPTY_SLOT;
            if (rowd != nullptr) rowd[slot] = 0xFFFFFFFFu;
        }
    }
}

template <int RS>
static int launch_connect_rs(const DevView &v, const ConnectArgs &a, hipStream_t stream) {
    const size_t yq_bytes =
        ((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull;
    const size_t lds = 2 * 64 * RS * 8 + yq_bytes;
    void (*kfn)(const DevView, const ConnectArgs);
    if (v.kind == HNSW_VEC_QUANT8)
        kfn = v.dim == 100   ? hx_connect_kernel<HNSW_VEC_QUANT8, 100, RS>
              : v.dim == 128 ? hx_connect_kernel<HNSW_VEC_QUANT8, 128, RS>
              : v.dim == 256 ? hx_connect_kernel<HNSW_VEC_QUANT8, 256, RS>
              : v.dim == 768 ? hx_connect_kernel<HNSW_VEC_QUANT8, 768, RS>
                             : hx_connect_kernel<HNSW_VEC_QUANT8, 0, RS>;
    else
        kfn = v.dim == 100   ? hx_connect_kernel<HNSW_VEC_F32, 100, RS>
              : v.dim == 128 ? hx_connect_kernel<HNSW_VEC_F32, 128, RS>
                             : hx_connect_kernel<HNSW_VEC_F32, 0, RS>;
    hipLaunchKernelGGL(kfn, dim3(a.count), dim3(64), lds, stream, v, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("connect kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

// the wide-row instantiations (m = 64 / 128) exist for the dimension-generic loops only: one compile-time
// dimension per row width would triple the build time of this file for shapes nobody has measured
template <int RS>
static int launch_connect_wide(const DevView &v, const ConnectArgs &a, hipStream_t stream) {
    const size_t yq_bytes =
        ((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull;
    const size_t lds = 2 * 64 * RS * 8 + yq_bytes;
    if (v.kind == HNSW_VEC_QUANT8)
        hipLaunchKernelGGL((hx_connect_kernel<HNSW_VEC_QUANT8, 0, RS>), dim3(a.count), dim3(64), lds, stream, v, a);
    else
        hipLaunchKernelGGL((hx_connect_kernel<HNSW_VEC_F32, 0, RS>), dim3(a.count), dim3(64), lds, stream, v, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("connect kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_connect(const DevView &v, const ConnectArgs &a, hipStream_t stream) {
    if (a.count == 0) return HNSW_OK;
    const uint32_t S = std::max(v.S0, v.S1);
    if (S <= 64) return launch_connect_rs<1>(v, a, stream);
    if (S <= 128) return launch_connect_wide<2>(v, a, stream);
    if (S <= 256) return launch_connect_wide<4>(v, a, stream);
    set_error("on-device build: adjacency rows of %u slots (m > 128)", S);
    return HNSW_ERR_ARG;
}

int launch_remove(const DevView &v, const ConnectArgs &a, hipStream_t stream) {
    if (a.count == 0) return HNSW_OK;
    const uint32_t S = std::max(v.S0, v.S1);
    if (S <= 64)
        hipLaunchKernelGGL(hx_remove_kernel<1>, dim3(a.count), dim3(64), 0, stream, v, a);
    else if (S <= 128)
        hipLaunchKernelGGL(hx_remove_kernel<2>, dim3(a.count), dim3(64), 0, stream, v, a);
    else if (S <= 256)
        hipLaunchKernelGGL(hx_remove_kernel<4>, dim3(a.count), dim3(64), 0, stream, v, a);
    else {
        set_error("on-device build: adjacency rows of %u slots (m > 128)", S);
        return HNSW_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("remove kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_scatter_rows(uint32_t *dst, uint32_t S, const uint32_t *d_row_index, const uint32_t *d_data,
                        uint32_t n, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_scatter_rows_kernel, dim3(n), dim3(64), 0, stream, dst, S, d_row_index, d_data, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("scatter kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

// Visited-table size for a search with list size ef on rows of up to s0 neighbours.  The table holds
// every id whose distance is computed in one layer; measured on 10240 queries of the 1M x 100d index
// (32-slot rows): efSearch 64 mean 1190 / max 1769, 96: 1593 / 2235, 128: 1966 / 2810, 288: 3593 / 5599
// (scripts/nvis_probe.py).  The limit (75 % of the slots) stays >= 1.1 x the observed maximum plus one
// row.  Too small is safe (status OVERFLOW; the host-pointer API retries with the next size), too
// large costs occupancy: 64 KiB tables leave 2 waves per CU, i.e. a 1024-query launch no longer fits
// the chip in one round.
uint32_t default_slots_log2(uint32_t ef, uint32_t s0) {
    const uint64_t e = (uint64_t)ef * std::max(s0, 8u) / 32u;  // ef in units of 32-slot rows
    if (e <= 112) return 12;  // 4096 slots, 16 KiB: limit 3072
    // 8192 slots, 32 KiB: limit 6144.  Up to ef 320: a search visits ~ 11 ef + 500 ids on average and up to 1.4 x that
    // (1M x 100d: 3.7 k at ef 300, no query of 8192 fills the table; at 384 one in twenty does and would run again with
    // the next size), and four waves per CU -- a whole batch of 1024 at once -- need the table to stay at 32 KiB
    // (ef 300, batch 1024: 0.86 ms against 1.50 ms with 64 KiB)
    if (e <= 320) return 13;
    if (e <= 576) return 14;  // 64 KiB: limit 12288
    if (ef <= 64 * HX_MAX_R_WIDE) return 15;  // 128 KiB
    // the HBM-resident table of hx_search_spill_kernel (limit: one half): about 30 visited ids per list entry
    uint32_t l = 16;
    while (l < 30 && (1ull << l) < 64ull * e) l++;
    return l;
}
uint32_t default_slots_log2(uint32_t ef) { return default_slots_log2(ef, 32); }
uint32_t max_slots_log2(uint32_t ef) { return ef <= 64 * HX_MAX_R_WIDE ? 15 : 31; }  // (the spill kernel caps its table at 4 N slots)

template <int KIND, int P, int DS, int R, bool FAT>
static int launch_one(const DevView &v, const SearchArgs &a_in, uint32_t nblocks, uint32_t slots_log2,
                      hipStream_t stream) {
    const size_t yq_bytes =
        (KIND == HNSW_VEC_QUANT8) ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4;
    SearchArgs a = a_in;
    // lists of eight / sixteen registers (ef > 320 asks for a 64- / 128-KiB table): 32 KiB of LDS + a second level in
    // HBM (stream-ordered scratch), see the kernel
    struct Scratch {
        void *p = nullptr;
        hipStream_t st = nullptr;
        ~Scratch() {
            if (p) (void)hipFreeAsync(p, st);
        }
    } sp;
    static const bool two_level = !(getenv("HNSW_MI355X_VISITED_2L") && atoi(getenv("HNSW_MI355X_VISITED_2L")) == 0);
    if (R >= 8 && two_level && slots_log2 >= 14 && a.spill_tab == nullptr) {
        const uint32_t glog2 = std::max(15u, slots_log2 + 1);
        sp.st = stream;
        if (hipMallocAsync(&sp.p, ((size_t)nblocks << glog2) * 4, stream) != hipSuccess) {
            (void)hipGetLastError();
            sp.p = nullptr;  // no scratch: the one-level table serves
        } else {
            a.spill_tab = static_cast<uint32_t *>(sp.p);
            a.spill_log2 = glog2;
            slots_log2 = 13;
            if (const char *e = getenv("HNSW_MI355X_VISITED_2L_LIMIT")) a.lds_limit = (uint32_t)atoi(e);
        }
    }
    size_t lds = (4ull << slots_log2) + scratch_region_bytes<KIND, DS, R>() + ((yq_bytes + 15) & ~15ull);
    if (FAT) lds += 2ull * 1024 * (P > 0 ? P : 1);
    auto kern = hx_search_kernel<KIND, P, DS, R, FAT>;
    if (lds > 160 * 1024) {
        set_error("search needs %zu bytes of LDS (> 160 KiB)", lds);
        return HNSW_ERR_ARG;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), lds, stream, v, a, slots_log2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("search kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

template <int P, int DS, int R>
static int launch_two(const DevView &v, const SearchArgs &a, uint32_t nblocks, uint32_t slots_log2,
                      hipStream_t stream) {
    const size_t yq_bytes = ((2ull * (v.half_bytes - 8) * 4) + 15) & ~15ull;
    const size_t lds = (4ull << slots_log2) + 2ull * 64 * R * 8 + 2 * 32 * 8 + 2 * 2 * 4 * 4 +
                       2 * yq_bytes + 2ull * 2 * (16ull * 32 * P);
    auto kern = hx_search2_kernel<P, DS, R>;
    if (lds > 160 * 1024) return HNSW_ERR_ARG;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(128), lds, stream, v, a, slots_log2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("two-wave search kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

// The two-wave kernel is opt-in (HNSW_MI355X_WAVES=2): measured on MI355X at 1M x 100d, batch 1024
// it takes 0.207 ms per batch against 0.185 ms for the one-wave kernel -- halving the distance work
// per wave does not pay for the key exchange + barrier and the replicated merge, because the loop
// is bound by instruction latency (about 750 instructions at ~5 cycles each per expansion), not by
// issue bandwidth.  Kept because it is parity-tested and documents the design point.
static bool want_two_waves(uint32_t) {
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("HNSW_MI355X_WAVES");
        forced = e ? atoi(e) : 0;
    }
    return forced == 2;
}

template <int KIND, int P, int DS>
static int launch_r(const DevView &v, const SearchArgs &a, uint32_t nblocks, uint32_t slots_log2,
                    hipStream_t stream, uint32_t ef_max) {
    // the inline-rows variant needs one pass to cover a whole layer-0 row
    // (block images of rows wider than 5 pieces per half would not leave 4 waves per CU: not built)
    constexpr bool CAN_FAT = (KIND == HNSW_VEC_QUANT8 && P > 0 && P <= 5);
    if constexpr (CAN_FAT) {
        // 16 rows per wave must be whole 1-KiB DMA pieces: row_stride * 16 % 1024 == 0
        if (v.fat != nullptr && v.S0 == 32 && a.layer_lo == 0 && a.entries == nullptr &&
            a.layer_hi == (int32_t)v.nb_layers - 1 && a.ef_upper == 1 && (16u * v.row_stride) % 1024u == 0 &&
            v.row_stride == 32u * P && want_two_waves(nblocks)) {
            if (ef_max <= 64) return launch_two<P, DS, 1>(v, a, nblocks, slots_log2, stream);
            if (ef_max <= 128) return launch_two<P, DS, 2>(v, a, nblocks, slots_log2, stream);
            if (ef_max <= 256) return launch_two<P, DS, 4>(v, a, nblocks, slots_log2, stream);
            if (ef_max <= 512) return launch_two<P, DS, 8>(v, a, nblocks, slots_log2, stream);
        }
    }
    // the inline-rows loop stages two block images in LDS; it is used while a wave still needs no more
    // than a quarter of the CU's LDS, so that 1024 waves fit the chip in one round
    const uint32_t r_list = ef_max <= 64 ? 1 : ef_max <= 128 ? 2 : ef_max <= 256 ? 4 : 8;
    const size_t fat_lds = (4ull << slots_log2) + 64ull * r_list * 8 + 2ull * 1024 * (P > 0 ? P : 1) + 1024;
    // ... and only while the launch is small enough to be latency-bound: with more than four waves per CU
    // the compact layout wins (measured, 1M x 100d quant8, efSearch 68: 2048 queries 0.359 ms inline
    // rows vs 0.256 ms compact; 32768 queries 8.4 vs 10.6 M q/s) -- the block images halve the
    // resident waves and every slot of a block is read whether it is needed or not
    static const uint32_t n_cu = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return (uint32_t)cus;
    }();
    if (CAN_FAT && v.fat != nullptr && v.S0 == 32 && a.layer_lo == 0 && fat_lds <= 40 * 1024 &&
        nblocks <= 4 * n_cu) {
        if (ef_max <= 64) return launch_one<KIND, P, DS, 1, CAN_FAT>(v, a, nblocks, slots_log2, stream);
        if (ef_max <= 128) return launch_one<KIND, P, DS, 2, CAN_FAT>(v, a, nblocks, slots_log2, stream);
        if (ef_max <= 256) return launch_one<KIND, P, DS, 4, CAN_FAT>(v, a, nblocks, slots_log2, stream);
        if (ef_max <= 512) return launch_one<KIND, P, DS, 8, CAN_FAT>(v, a, nblocks, slots_log2, stream);
    }
    if (ef_max <= 64) return launch_one<KIND, P, DS, 1, false>(v, a, nblocks, slots_log2, stream);
    if (ef_max <= 128) return launch_one<KIND, P, DS, 2, false>(v, a, nblocks, slots_log2, stream);
    if (ef_max <= 256) return launch_one<KIND, P, DS, 4, false>(v, a, nblocks, slots_log2, stream);
    if (ef_max <= 512) return launch_one<KIND, P, DS, 8, false>(v, a, nblocks, slots_log2, stream);
    set_error("ef = %u is above the supported maximum of %d", ef_max, 64 * HX_MAX_R);
    return HNSW_ERR_ARG;
}


// =============================================================================================
// ef beyond what a wave's registers hold (> 1024).  The reference's ann_by_vector has no limit on ef
// (template.rs:306-311: `selected` is a BTreeSet); the register-resident list of the kernels above stops
// at sixteen registers per lane.  This kernel keeps the SAME single sorted list (key = dist_bits << 32 |
// id, bit 63 = expanded) and the visited set in HBM scratch instead, one wave per query, and applies a
// batch of neighbour distances the way the reference literally does -- one element at a time
// (searcher.rs:74-94): position by a wave-wide count, the tail moved up by one from the top, the element
// stored.  Slow (every insertion is a few dependent HBM round trips) but exact and without a limit other
// than memory: the list holds ef entries of 8 bytes, the table a power of two of slots that the host
// doubles and re-runs when a query fills it to one half (HNSW_ERR_OVERFLOW), which ends at 4 N slots.
// List and table are read and written through L2 (agent-scope relaxed atomics: the lanes of the wave
// hand entries to each other through memory, and the table's compare-and-swap executes there).
// =============================================================================================
struct SpillScratch {
    u64 *lists;        // nblocks x list_cap
    uint32_t *tabs;    // nblocks << tab_log2
    uint32_t list_cap;
    uint32_t tab_log2;
    uint32_t q_first;  // launch block b serves query (selection entry) q_first + b
};

__device__ __forceinline__ uint32_t spill_ld(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 spill_ld(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void spill_st(u64 *p, u64 x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void spill_st(uint32_t *p, uint32_t x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a wave's own stores have reached L2 before its next loads are issued
__device__ __forceinline__ void spill_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// IntSet::insert on the HBM table (linear probing, load <= 1/2): true when id was absent
__device__ __forceinline__ bool spill_visited_insert(uint32_t *tab, uint32_t mask, uint32_t shift, uint32_t id) {
    uint32_t s = ((id * 0x9E3779B1u) >> shift) & mask;
    while (true) {
        const uint32_t cur = spill_ld(tab + s);
        if (cur == id) return false;
        if (cur == HX_EMPTY_SLOT) {
            const uint32_t old = atomicCAS(tab + s, HX_EMPTY_SLOT, id);
            if (old == HX_EMPTY_SLOT) return true;
            if (old == id) return false;
        }
        s = (s + 1) & mask;  // taken (by another id, possibly of this very pass): next slot
    }
}

template <int KIND>
__global__ void __launch_bounds__(64)
hx_search_spill_kernel(const DevView v, const SearchArgs a, const SpillScratch sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *yq = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x;
    const uint32_t qi = sp.q_first + blockIdx.x;
    const uint32_t q = a.qsel ? a.qsel[qi] : qi;
    u64 *L = sp.lists + (size_t)blockIdx.x * sp.list_cap;
    uint32_t *tab_all = sp.tabs + ((size_t)blockIdx.x << sp.tab_log2);
    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;
    constexpr int CHUNK = 64 / LPC;
    const int h = (LPC == 2) ? (lane & 1) : 0;
    const int cslot = lane / LPC;
    const uint32_t d = v.dim;
    uint32_t n_dist = 0, n_exp = 0, sum_deg = 0;
    int32_t status = HNSW_OK;
    if (!stage_query<KIND>(v, a.Q + (size_t)q * d, yq, lane)) status = HNSW_ERR_NAN_INPUT;

    uint32_t n_cur = 0, hint = 0, n_vis = 0;  // hint: every entry before it is expanded
    u64 last_key = KEY_INVALID;                // key of the last entry once the list holds ef of them
    uint32_t tmask = 0, tshift = 0, vis_limit = 0;

    // `selected.insert(e)` + `pop_last` when over ef (searcher.rs:77-92), e wave-uniform
    auto insert_one = [&](u64 e, uint32_t ef_l) __attribute__((always_inline)) {
        if (n_cur >= ef_l && !(e < last_key)) return;
        uint32_t pos = 0;
        for (uint32_t i = 0; i < n_cur; i += 64) {  // entries below e: the list is sorted, the first chunk that holds a larger one ends the count
            const uint32_t idx = i + (uint32_t)lane;
            const u64 k = idx < n_cur ? (spill_ld(L + idx) & KEY_MASK) : KEY_INVALID;
            const u64 below = __ballot(k < e);
            pos += (uint32_t)__popcll(below);
            if (below != ~0ull) break;
        }
        const uint32_t n_new = n_cur < ef_l ? n_cur + 1 : n_cur;  // full: the last entry falls off
        for (uint32_t hi = n_new; hi > pos + 1;) {                // entries pos .. n_new - 2 move up by one, top chunk first
            const bool mv = hi >= 1u + (uint32_t)lane && hi - 1u - (uint32_t)lane > pos;
            const uint32_t idx = hi - 1u - (uint32_t)lane;
            u64 t = 0;
            if (mv) t = spill_ld(L + idx - 1);
            if (mv) spill_st(L + idx, t);  // the chunk's loads have returned before its stores issue
            if (hi <= 64) break;
            hi -= 64;
        }
        if (lane == 0) spill_st(L + pos, e);
        spill_fence();
        n_cur = n_new;
        if (pos < hint) hint = pos;
        last_key = n_cur >= ef_l ? (readlane64(spill_ld(L + (n_cur - 1)), 0) & KEY_MASK) : KEY_INVALID;
    };
    // one pass over up to CHUNK ids: visited filter, distance, then the batch applied in lane order
    auto process = [&](uint32_t id, bool valid, bool visit, uint32_t ef_l) __attribute__((always_inline)) {
        bool fresh = valid;
        if (visit) {
            bool f = false;
            if (valid && h == 0) f = spill_visited_insert(tab_all, tmask, tshift, id);
            if (LPC == 2) f = (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
            fresh = f;
        }
        const u64 fm = __ballot(fresh && h == 0);
        if (visit) n_vis += (uint32_t)__popcll(fm);
        if (fm == 0) return;
        n_dist += (uint32_t)__popcll(fm);
        const float dist = dist_any_dim<KIND>(v, id, fresh, h, yq);
        const bool mine = fresh && h == 0;
        if (__ballot(mine && dist != dist)) {
            status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
            return;
        }
        const u64 key = mine ? (((u64)__builtin_bit_cast(uint32_t, dist) << 32) | id) : KEY_INVALID;
        u64 it = __ballot(mine);
        while (it) {
            const int j = __ffsll((long long)it) - 1;
            it &= it - 1;
            insert_one(readlane64(key, j), ef_l);
        }
    };

    // the table of a layer: all of it for a wide list, 4096 slots of it for the greedy upper layers
    auto use_table = [&](uint32_t ef_l) __attribute__((always_inline)) {
        const uint32_t log2 = ef_l <= 64 ? min(sp.tab_log2, 12u) : sp.tab_log2;
        tmask = (1u << log2) - 1;
        tshift = 32 - log2;
        vis_limit = 1u << (log2 - 1);
        for (uint32_t s2 = lane; s2 <= tmask; s2 += 64) spill_st(tab_all + s2, HX_EMPTY_SLOT);
        spill_fence();
    };

    if (status == HNSW_OK) {
        // ---- entry set: {ep} (template.rs:316-319) or the caller's (search_layer seam) ----
        const uint32_t n_entry = a.entries ? a.n_entry : 1;
        const uint32_t ef_first = max(1u, (a.layer_hi > a.layer_lo) ? a.ef_upper : a.ef_bottom);
        for (uint32_t base = 0; base < n_entry; base += CHUNK) {
            const uint32_t i = base + cslot;
            const bool valid = i < n_entry;
            uint32_t id = 0;
            if (valid) id = a.entries ? a.entries[i] : v.ep;
            if (__ballot(valid && id >= v.n_points)) {
                status = HNSW_ERR_ARG;
                break;
            }
            process(id, valid, false, max(ef_first, n_entry));
        }
    }
    for (int layer = a.layer_hi; status == HNSW_OK && layer >= a.layer_lo; layer--) {
        const uint32_t ef_l = max(1u, layer > a.layer_lo ? a.ef_upper : a.ef_bottom);
        use_table(ef_l);
        // candidates ∪= selected, visited ∪= ids(selected)  (searcher.rs:32-33)
        for (uint32_t i = 0; i < n_cur; i += 64) {
            const uint32_t idx = i + (uint32_t)lane;
            if (idx < n_cur) {
                const u64 k = spill_ld(L + idx) & KEY_MASK;
                spill_st(L + idx, k);
                spill_visited_insert(tab_all, tmask, tshift, (uint32_t)k);
            }
        }
        spill_fence();
        n_vis = n_cur;
        hint = 0;
        last_key = n_cur >= ef_l ? (readlane64(spill_ld(L + (n_cur - 1)), 0) & KEY_MASK) : KEY_INVALID;
        const uint32_t S = layer == 0 ? v.S0 : v.S1;
        while (status == HNSW_OK) {
            // ---- candidates.pop_first(): the smallest entry not expanded yet (searcher.rs:35-44) ----
            int pos = -1;
            u64 ck = 0;
            for (uint32_t i = hint; i < n_cur; i += 64) {
                const uint32_t idx = i + (uint32_t)lane;
                const u64 k = idx < n_cur ? spill_ld(L + idx) : KEY_INVALID;
                const u64 un = __ballot((k >> 63) == 0);
                if (un) {
                    const int j = __ffsll((long long)un) - 1;
                    pos = (int)i + j;
                    ck = readlane64(k, j);
                    break;
                }
            }
            if (pos < 0) break;
            if (lane == 0) spill_st(L + pos, ck | KEY_EXPANDED);
            spill_fence();
            hint = (uint32_t)pos + 1;
            const uint32_t cid = (uint32_t)ck;
            n_exp++;
            const uint32_t *row;
            if (layer == 0) {
                row = v.adj0 + (size_t)cid * S;
            } else {
                const uint32_t ub = v.upper_base[cid];
                if (ub == HX_EMPTY_SLOT) {  // Graph::neighbors_vec -> NodeNotInGraph
                    status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                    break;
                }
                row = v.adj_up + ((size_t)ub + layer - 1) * S;
            }
            uint32_t ovf = HX_EMPTY_SLOT;
            for (uint32_t c0 = 0; c0 < S && status == HNSW_OK; c0 += CHUNK) {
                const uint32_t slot = c0 + cslot;
                uint32_t nb = HX_EMPTY_SLOT;
                if (slot < S) nb = row[slot];
                const bool is_ptr = nb != HX_EMPTY_SLOT && (nb & HX_OVF_FLAG);
                const bool valid = nb != HX_EMPTY_SLOT && !is_ptr;
                const u64 pm = __ballot(is_ptr);
                if (pm) ovf = (uint32_t)__builtin_amdgcn_readlane((int)nb, __ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                if (cnt == 0) continue;
                sum_deg += cnt;
                if (n_vis + cnt > vis_limit) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                process(nb, valid, true, ef_l);
            }
            if (status == HNSW_OK && ovf != HX_EMPTY_SLOT) {  // degree > S: the rest of the row
                const uint32_t lo = v.ovf_off[ovf], hi = v.ovf_off[ovf + 1];
                for (uint32_t base = lo; base < hi && status == HNSW_OK; base += CHUNK) {
                    const uint32_t i = base + cslot;
                    const bool valid = i < hi;
                    const uint32_t nb = valid ? v.ovf_nbrs[i] : HX_EMPTY_SLOT;
                    const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                    sum_deg += cnt;
                    if (n_vis + cnt > vis_limit) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    process(nb, valid, true, ef_l);
                }
            }
        }
    }

    // ---- get_top_selected(n) (results.rs:59-61) ----
    const uint32_t count = status == HNSW_OK ? min(a.n, n_cur) : 0;
    for (uint32_t idx = lane; idx < a.n; idx += 64) {
        const bool have = idx < count;
        const u64 k = have ? spill_ld(L + idx) : KEY_INVALID;
        a.out_ids[(size_t)q * a.n + idx] = have ? (uint32_t)k : HX_EMPTY_SLOT;
        if (a.out_dists)
            a.out_dists[(size_t)q * a.n + idx] =
                have ? __builtin_bit_cast(float, (uint32_t)((k & KEY_MASK) >> 32)) : __builtin_inff();
    }
    if (lane == 0) {
        if (a.out_counts) a.out_counts[q] = count;
        hnsw_query_stats st;
        st.n_dist = n_dist;
        st.n_exp = n_exp;
        st.sum_deg = sum_deg;
        st.status = status;
        a.out_stats[q] = st;
    }
}

static int launch_spill(const DevView &v, const SearchArgs &a, uint32_t nblocks, uint32_t slots_log2,
                        hipStream_t stream, uint32_t ef_max) {
    // table: what the caller asks for (it doubles on HNSW_ERR_OVERFLOW), never more than 4 N slots -- a
    // layer visits every id at most once, so a table of 4 N cannot fill to its limit of one half
    uint32_t cap_log2 = 12;
    while (cap_log2 < 31 && (1ull << cap_log2) < 4ull * v.n_points) cap_log2++;
    const uint32_t tab_log2 = std::min(std::max(slots_log2, 12u), cap_log2);
    const uint64_t list_cap = ef_max;
    const uint64_t per_q = list_cap * 8 + (4ull << tab_log2);
    const uint64_t budget = 1ull << 30;  // scratch per launch
    const uint32_t group = (uint32_t)std::min<uint64_t>(nblocks, std::max<uint64_t>(1, budget / per_q));
    const size_t yq_bytes =
        ((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull;
    for (uint32_t first = 0; first < nblocks; first += group) {
        const uint32_t n = std::min(group, nblocks - first);
        void *mem = nullptr;
        bool async = true;
        if (hipMallocAsync(&mem, n * per_q, stream) != hipSuccess) {  // no stream-ordered pool: plain allocation
            (void)hipGetLastError();
            async = false;
            hipError_t e = hipMalloc(&mem, n * per_q);
            if (e != hipSuccess) {
                set_error("search with ef = %u needs %llu bytes of scratch: %s", ef_max, (unsigned long long)(n * per_q),
                          hipGetErrorString(e));
                return HNSW_ERR_OOM;
            }
        }
        SpillScratch sp{};
        sp.lists = static_cast<u64 *>(mem);
        sp.tabs = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(mem) + (size_t)n * list_cap * 8);
        sp.list_cap = (uint32_t)list_cap;
        sp.tab_log2 = tab_log2;
        sp.q_first = first;
        if (v.kind == HNSW_VEC_QUANT8)
            hipLaunchKernelGGL(hx_search_spill_kernel<HNSW_VEC_QUANT8>, dim3(n), dim3(64), yq_bytes, stream, v, a, sp);
        else
            hipLaunchKernelGGL(hx_search_spill_kernel<HNSW_VEC_F32>, dim3(n), dim3(64), yq_bytes, stream, v, a, sp);
        hipError_t e = hipGetLastError();
        if (async) {
            (void)hipFreeAsync(mem, stream);
        } else {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(mem);
        }
        if (e != hipSuccess) {
            set_error("search kernel launch (ef = %u): %s", ef_max, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    return HNSW_OK;
}

int launch_search(const DevView &v, const SearchArgs &a_in, uint32_t nblocks, uint32_t slots_log2,
                  hipStream_t stream) {
    if (nblocks == 0) return HNSW_OK;
    SearchArgs a = a_in;
    static const bool one_row = getenv("HNSW_MI355X_ONE_ROW") && atoi(getenv("HNSW_MI355X_ONE_ROW")) != 0;
    if (one_row) a.flags |= 1u;
    uint32_t ef_max = std::max(1u, a.ef_bottom);
    if (a.layer_hi > a.layer_lo) ef_max = std::max(ef_max, a.ef_upper);
    if (a.entries) ef_max = std::max(ef_max, a.n_entry);
    if (slots_log2 == 0) slots_log2 = default_slots_log2(ef_max, v.S0);
    if (lean_applicable(v, a, ef_max)) return launch_lean(v, a, nblocks, slots_log2, stream);
    if (ef_max > 64 * HX_MAX_R) {
        // beyond 512 entries: the any-dimension kernel with sixteen list registers per lane (ef <= 1024;
        // one wave per CU: the visited table takes 128 KiB).  The reference has no limit
        // (template.rs:306-311); this is as far as a wave-resident list goes.
        // beyond 1024: list and visited table in HBM scratch, exact and slow (the reference has no limit)
        if (ef_max > 64 * HX_MAX_R_WIDE) return launch_spill(v, a, nblocks, slots_log2, stream, ef_max);
        if (v.kind == HNSW_VEC_QUANT8)
            return launch_one<HNSW_VEC_QUANT8, 0, 0, HX_MAX_R_WIDE, false>(v, a, nblocks, slots_log2, stream);
        return launch_one<HNSW_VEC_F32, 0, 0, HX_MAX_R_WIDE, false>(v, a, nblocks, slots_log2, stream);
    }
    if (v.kind == HNSW_VEC_QUANT8) {
        const uint32_t P = v.half_bytes / 16;
        if (v.dim == 100) return launch_r<HNSW_VEC_QUANT8, 4, 100>(v, a, nblocks, slots_log2, stream, ef_max);
        if (v.dim == 128 && P == 5) return launch_r<HNSW_VEC_QUANT8, 5, 128>(v, a, nblocks, slots_log2, stream, ef_max);
        if (v.dim == 256 && P == 9) return launch_r<HNSW_VEC_QUANT8, 9, 256>(v, a, nblocks, slots_log2, stream, ef_max);
        if (v.dim == 768 && P == 25) return launch_r<HNSW_VEC_QUANT8, 25, 768>(v, a, nblocks, slots_log2, stream, ef_max);
        switch (P) {
            case 1: return launch_r<HNSW_VEC_QUANT8, 1, 0>(v, a, nblocks, slots_log2, stream, ef_max);
            case 2: return launch_r<HNSW_VEC_QUANT8, 2, 0>(v, a, nblocks, slots_log2, stream, ef_max);
            case 3: return launch_r<HNSW_VEC_QUANT8, 3, 0>(v, a, nblocks, slots_log2, stream, ef_max);
            case 4: return launch_r<HNSW_VEC_QUANT8, 4, 0>(v, a, nblocks, slots_log2, stream, ef_max);
            case 5: return launch_r<HNSW_VEC_QUANT8, 5, 0>(v, a, nblocks, slots_log2, stream, ef_max);
            default: return launch_r<HNSW_VEC_QUANT8, 0, 0>(v, a, nblocks, slots_log2, stream, ef_max);
        }
    }
    if (v.dim == 100 && v.row_stride == 400)
        return launch_r<HNSW_VEC_F32, 25, 100>(v, a, nblocks, slots_log2, stream, ef_max);
    if (v.dim == 128 && v.row_stride == 512)
        return launch_r<HNSW_VEC_F32, 32, 128>(v, a, nblocks, slots_log2, stream, ef_max);
    if (v.dim == 256 && v.row_stride == 1024)
        return launch_r<HNSW_VEC_F32, 64, 256>(v, a, nblocks, slots_log2, stream, ef_max);
    if (v.dim == 768 && v.row_stride == 3072)
        return launch_r<HNSW_VEC_F32, 192, 768>(v, a, nblocks, slots_log2, stream, ef_max);
    return launch_r<HNSW_VEC_F32, 0, 0>(v, a, nblocks, slots_log2, stream, ef_max);
}

// ---------------------------------------------------------------------------------------------
// distance_batch: VecBase::dist2many (vectors/src/lib.rs:17-22) for one query -- the search
// kernel restricted to "evaluate these ids": every id is an entry, results come back in list
// order, so this launcher runs the kernel with ef = n = k and then un-sorts on the host side.
// (Kept simple on purpose: it is a test seam, not a hot path.)
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(64)
hx_distance_kernel(const DevView v, const float *q, const uint32_t *ids, uint64_t k, float *out,
                   int32_t *status_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *yq = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x;
    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;
    constexpr int CHUNK = 64 / LPC;
    const int h = (LPC == 2) ? (lane & 1) : 0;
    int32_t status = stage_query<KIND>(v, q, yq, lane) ? HNSW_OK : HNSW_ERR_NAN_INPUT;
    for (uint64_t base = (uint64_t)blockIdx.x * CHUNK; base < k;
         base += (uint64_t)gridDim.x * CHUNK) {
        const uint64_t i = base + lane / LPC;
        const bool active = i < k;
        const uint32_t id = active ? ids[i] : 0;
        const bool ok = active && id < v.n_points;
        const float dist = dist_any_dim<KIND>(v, id, ok, h, yq);
        if (active && h == 0) {
            if (!ok) status = HNSW_ERR_ARG;
            out[i] = ok ? dist : __builtin_nanf("");
        }
    }
    if (status != HNSW_OK) atomicMin(status_out, status);
}

// ---------------------------------------------------------------------------------------------
// brute force: exact top-k of every query over ALL points under the index's own metric (the
// reference's ground truth: helpers/glove.rs:94-109, template.rs:531-541).  Block (seg, q) scans
// one contiguous segment of the ids and keeps its k best in the same sorted list the search
// uses; the host merges the nseg partial lists of a query.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(64)
hx_brute_kernel(const DevView v, const float *Q, uint32_t k, uint32_t nseg, uint32_t *part_ids,
                float *part_dists, int32_t *status_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *perm = reinterpret_cast<u64 *>(smem);
    float *yq = reinterpret_cast<float *>(perm + 64);
    const int lane = threadIdx.x;
    const uint32_t seg = blockIdx.x, q = blockIdx.y;
    constexpr int LPC = (KIND == HNSW_VEC_QUANT8) ? 2 : 1;
    constexpr int CHUNK = 64 / LPC;
    const int h = (LPC == 2) ? (lane & 1) : 0;
    int32_t status = stage_query<KIND>(v, Q + (size_t)q * v.dim, yq, lane) ? HNSW_OK : HNSW_ERR_NAN_INPUT;
    WaveList<1> wl;
    wl.L[0] = KEY_INVALID;
    wl.n_cur = 0;
    wl.last_key = KEY_INVALID;
    const uint64_t per = ((uint64_t)v.n_points + nseg - 1) / nseg;
    const uint64_t lo = per * seg, hi = min((uint64_t)v.n_points, lo + per);
    for (uint64_t base = lo; base < hi; base += CHUNK) {
        const uint64_t i = base + lane / LPC;
        const bool active = i < hi;
        const float dist = dist_any_dim<KIND>(v, (uint32_t)i, active, h, yq);
        u64 key = KEY_INVALID;
        if (active && h == 0) {
            if (dist != dist)
                status = HNSW_ERR_NAN_INPUT;
            else
                key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | (uint32_t)i;
        }
        wl.merge(key, k, perm, lane);
    }
    if ((uint32_t)lane < k) {
        const size_t o = ((size_t)q * nseg + seg) * k + lane;
        const bool have = (uint32_t)lane < wl.n_cur;
        part_ids[o] = have ? (uint32_t)wl.L[0] : HX_EMPTY_SLOT;
        part_dists[o] = have ? __builtin_bit_cast(float, (uint32_t)(wl.L[0] >> 32)) : __builtin_inff();
    }
    if (__ballot(status != HNSW_OK) && lane == 0) atomicMin(status_out, HNSW_ERR_NAN_INPUT);
}

int launch_brute_force(const DevView &v, const float *d_Q, uint64_t nq, uint32_t k, uint32_t nseg,
                       uint32_t *part_ids, float *part_dists, int32_t *d_status,
                       hipStream_t stream) {
    if (nq == 0) return HNSW_OK;
    if (k == 0 || k > 64 || nq > 65535) {
        set_error("brute force supports 1 <= k <= 64 and at most 65535 queries per call");
        return HNSW_ERR_ARG;
    }
    const size_t lds =
        64 * 8 + (((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull);
    if (v.kind == HNSW_VEC_QUANT8)
        hipLaunchKernelGGL(hx_brute_kernel<HNSW_VEC_QUANT8>, dim3(nseg, (uint32_t)nq), dim3(64), lds,
                           stream, v, d_Q, k, nseg, part_ids, part_dists, d_status);
    else
        hipLaunchKernelGGL(hx_brute_kernel<HNSW_VEC_F32>, dim3(nseg, (uint32_t)nq), dim3(64), lds,
                           stream, v, d_Q, k, nseg, part_ids, part_dists, d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("brute force kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_distance_batch(const DevView &v, const float *d_q, const uint32_t *d_ids, uint64_t k,
                          float *d_out, int32_t *d_status, hipStream_t stream) {
    if (k == 0) return HNSW_OK;
    const size_t lds =
        ((v.kind == HNSW_VEC_QUANT8 ? 2ull * (v.half_bytes - 8) * 4 : (size_t)v.dim * 4) + 15) & ~15ull;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (k + 31) / 32);
    if (v.kind == HNSW_VEC_QUANT8)
        hipLaunchKernelGGL(hx_distance_kernel<HNSW_VEC_QUANT8>, dim3(grid), dim3(64), lds, stream, v,
                           d_q, d_ids, k, d_out, d_status);
    else
        hipLaunchKernelGGL(hx_distance_kernel<HNSW_VEC_F32>, dim3(grid), dim3(64), lds, stream, v,
                           d_q, d_ids, k, d_out, d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("distance kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

}  // namespace hx
