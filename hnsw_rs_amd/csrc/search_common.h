// search_common.h -- device helpers shared by the search kernels (search_kernels.hip, search_lean.hip):
// the reference's quantiser and QuantVec::distance_unrolled in the lane-pair form, query staging.
// gfx950 only; include from .hip files.
#pragma once

#include "device_index.h"

namespace hx {

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}
// value held by the other lane of this lane's pair (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float pair_swap(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ int pair_swap_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
}
// Orders this wave's LDS traffic (one wave's DS operations execute in issue order; the clobber keeps
// the compiler from moving accesses across).  The per-query state of a wave is private to it, so no
// workgroup barrier is needed -- and none may be used where only some waves of a workgroup run.
__device__ __forceinline__ void wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Rust `f32 as u8`: saturating, NaN -> 0
__device__ __forceinline__ uint32_t f32_as_u8(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 255.0f) return 255;
    return (uint32_t)x;
}

// ---------------------------------------------------------------------------------------------
// distance of this lane's share of one QUANT8 row (quant.rs:14-37).
// P 16-byte pieces per half row; element e of the half sits at byte 8 + e.  Elements below nch4
// are chunk elements (running sum e & 3 of this lane), elements [nch4, nch4 + rem) are the tail
// and all go to running sum 0 of lane h == 0, in order, after its chunk elements.
// yq: this half's dequantised query values in the same element order (LDS or registers).
// ---------------------------------------------------------------------------------------------
template <int P, int DS, typename QSrc>
__device__ __forceinline__ void quant_half_sums(const uint4 (&w)[P], const QSrc &yq, int h,
                                                uint32_t nch4, uint32_t rem, float (&acc)[4]) {
    const float mn = __builtin_bit_cast(float, w[0].x);
    const float delta = __builtin_bit_cast(float, w[0].y);
#pragma unroll
    for (int p = 0; p < P; p++) {
        const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (p == 0 && j < 2) continue;  // header
            if (DS > 0) {
                // compile-time dimension: dead elements vanish; the four bytes of a chunk dword feed
                // running sums 0..3 as two packed pairs (v_pk_mul_f32 / v_pk_add_f32: per element the
                // same correctly rounded operations as the scalar form)
                constexpr int N4 = 4 * (DS / 8);
                const int e0 = 16 * p + 4 * j - 8;
                if (e0 + 3 < N4) {
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const f32x2 c = {(float)((dw[j] >> (8 * k)) & 0xFFu),
                                         (float)((dw[j] >> (8 * (k + 1))) & 0xFFu)};
                        const f32x2 x = c * delta + mn;
                        const f32x2 y = {yq[e0 + k], yq[e0 + k + 1]};
                        const f32x2 t = x - y;
                        const f32x2 t2 = t * t;
                        acc[k] += t2.x;
                        acc[k + 1] += t2.y;
                    }
                    continue;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = 16 * p + 4 * j + k - 8;
                if (DS > 0) {  // the tail of a compile-time dimension
                    constexpr int N4 = 4 * (DS / 8), RM = DS % 8;
                    if (e >= N4 + RM) continue;
                    const float x = ((float)((dw[j] >> (8 * k)) & 0xFFu) * delta) + mn;
                    const float t = x - yq[e];
                    const float t2 = t * t;
                    if (e < N4)
                        acc[k] += t2;
                    else
                        acc[0] += (h == 0) ? t2 : 0.0f;  // +0.0 leaves a non-negative sum as is
                } else {
                    const float x = ((float)((dw[j] >> (8 * k)) & 0xFFu) * delta) + mn;
                    const float t = x - yq[e];
                    const float t2 = t * t;
                    const bool chunk = (uint32_t)e < nch4;
                    const bool tail = !chunk && (uint32_t)e < nch4 + rem && h == 0;
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
}

// ---------------------------------------------------------------------------------------------
// Stage one query in LDS.  QUANT8: the query is quantised with its own min / delta exactly like
// a stored vector (Point::new -> QuantVec::new, template.rs:313, quant.rs:41-66) and kept
// dequantised, y = code * delta + min, split into the two half-row element orders:
// yq[h * nq_half + i].  F32: the raw values.  Returns false when the query holds a NaN
// (partial_cmp().unwrap() panics in the reference).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ bool stage_query(const DevView &v, const float *qv, float *yq, int lane) {
    const uint32_t d = v.dim;
    bool bad = false;
    if (KIND == HNSW_VEC_QUANT8) {
        const uint32_t nq_half = v.half_bytes - 8;
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (uint32_t e = lane; e < d; e += 64) {
            const float x = qv[e];
            bad |= (x != x);
            lo = fminf(lo, x);
            hi = fmaxf(hi, x);
        }
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o));
            hi = fmaxf(hi, __shfl_xor(hi, o));
        }
        const float delta = (hi - lo) / 255.0f;  // (ub - lb) / (2^8 - 1)
        for (uint32_t e = lane; e < 2 * nq_half; e += 64) yq[e] = 0.0f;
        wave_fence();
        const uint32_t full = d & ~7u;
        for (uint32_t e = lane; e < d; e += 64) {
            float b = (qv[e] - lo) / delta;
            b += 0.5f;
            const float y = ((float)f32_as_u8(floorf(b)) * delta) + lo;
            uint32_t hh, i;
            if (e < full) {
                hh = (e & 7) >> 2;
                i = 4 * (e >> 3) + (e & 3);
            } else {
                hh = 0;
                i = v.nch4 + (e - full);
            }
            yq[hh * nq_half + i] = y;
        }
    } else {
        for (uint32_t e = lane; e < d; e += 64) {
            const float x = qv[e];
            bad |= (x != x);
            yq[e] = x;
        }
    }
    wave_fence();
    return __ballot(bad) == 0;
}

// LDS-resident query values of one half
struct QLds {
    const float *p;
    __device__ __forceinline__ float operator[](int e) const { return p[e]; }
};
// register-resident query values (compile-time dimension)
template <int N>
struct QRegs {
    float v[N];
    __device__ __forceinline__ float operator[](int e) const { return v[e]; }
};


}  // namespace hx
