// brute_mfma.hip -- exhaustive ground truth on the matrix cores (gfx950, MI355X): the reference's brute
// force (helpers/glove.rs:94-109, template.rs:531-541: every query against every stored point, full sort)
// as a dense query x points contraction.
//
// NOT the exact path: the inner products are accumulated by v_mfma_f32_32x32x2_f32 in its own order, not
// in FullVec::distance's single left-to-right chain (full.rs:23-29), so a score can differ from the
// reference's distance in the last bits.  It is therefore used as a SCREEN: per query the k + 8 best
// points by score s(x) = |x|^2 - 2 x.q (the query's own norm does not change the order) are kept, and
// their distances are then recomputed in the reference's exact arithmetic and order (hx_pair_distance_kernel)
// and sorted by (dist, id).  The result equals the exact scan's unless rounding moved a true top-k member
// below k + 8 others, which the tests check does not happen on the test sets; hnsw_brute_force stays the
// exact, bit-for-bit one.  FullVec (f32) rows only.
//
// Shape: a 256-thread workgroup owns a tile of 32 queries (staged once in LDS, rows padded by 16 B so that
// the 16-byte operand reads are bank-conflict free) and one segment of the points; each of its four waves
// walks its own quarter of the segment in tiles of 32 points.  Per 32 x 32 tile the wave issues d / 2
// MFMAs (A = points, one row per lane modulo 32; B = queries), so a lane ends up with 16 scores of ONE
// query (column l % 32), which it filters against that query's running threshold in registers.
// MFMA-bound by design: 157 TFLOP/s f32 peak against ~4 TFLOP/s of the one-wave-per-segment VALU scan.

#include <hip/hip_runtime.h>

#include "device_index.h"

namespace hx {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

static constexpr int MF_K2 = 20;   // candidates kept per lane (two lanes and 4 x nseg waves per query)
static constexpr int MF_QT = 32;   // queries per workgroup
static constexpr int MF_CH = 8;    // 16-byte pieces per lane in flight per stage

__global__ void __launch_bounds__(256) hx_row_norms_kernel(const float *X, uint32_t N, uint32_t d, float *xn) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float *r = X + (size_t)i * d;
    float s = 0.0f;
    for (uint32_t e = 0; e < d; e++) s += r[e] * r[e];
    xn[i] = s;
}

__global__ void __launch_bounds__(256)
hx_brute_mfma_kernel(const float *X, const float *xn, uint32_t N, uint32_t d, const float *Q, uint32_t nq,
                     uint32_t nseg, float *out_s, uint32_t *out_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t ldq = d + 4;                                   // padded query row (floats)
    float *Qs = reinterpret_cast<float *>(smem);                  // [32][ldq]
    float *xns = Qs + (size_t)MF_QT * ldq;                        // [4 waves][32] norms of the current tile
    const uint32_t tile = blockIdx.y, seg = blockIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t j = lane & 31, hi = lane >> 5;
    // ---- stage the query tile (rows beyond nq are zero) ----
    for (uint32_t e = threadIdx.x; e < MF_QT * ldq; e += 256) {
        const uint32_t r = e / ldq, c = e % ldq;
        const uint32_t qi = tile * MF_QT + r;
        Qs[e] = (qi < nq && c < d) ? Q[(size_t)qi * d + c] : 0.0f;
    }
    __syncthreads();
    // ---- this wave's points ----
    const uint64_t per_seg = ((uint64_t)N + nseg - 1) / nseg;
    const uint64_t s_lo = per_seg * seg, s_hi = min((uint64_t)N, s_lo + per_seg);
    const uint64_t per_w = (((s_hi > s_lo ? s_hi - s_lo : 0) + 3) / 4 + 31) / 32 * 32;
    const uint64_t w_lo = s_lo + per_w * wave, w_hi = min(s_hi, w_lo + per_w);
    float bs[MF_K2];
    uint32_t bi[MF_K2];
#pragma unroll
    for (int t = 0; t < MF_K2; t++) {
        bs[t] = __builtin_inff();
        bi[t] = HX_EMPTY_SLOT;
    }
    float thr = __builtin_inff();
    const uint32_t npieces = d >> 2;                 // 16-byte pieces of a row
    const uint32_t nmine = (npieces + 1 - hi) >> 1;  // pieces 2 t + hi < npieces
    const uint32_t nmax = (npieces + 1) >> 1;
    const float *qrow = Qs + (size_t)j * ldq;
    for (uint64_t p0 = w_lo; p0 < w_hi; p0 += 32) {
        const uint64_t row = p0 + j;
        const bool rv = row < w_hi;
        const float *xr = X + (size_t)(rv ? row : w_lo) * d;
        if (lane < 32) xns[wave * 32 + lane] = rv ? xn[row] : __builtin_inff();
        v16f acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.0f;
        // stream this lane's pieces of its point row, MF_CH at a time, the next stage in flight while
        // the current one feeds the MFMAs
        v4f a_cur[MF_CH], a_nxt[MF_CH];
        auto fetch = [&](v4f (&w)[MF_CH], uint32_t t0) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < MF_CH; c++) {
                const uint32_t t = t0 + c;
                w[c] = (t < nmine && rv) ? *reinterpret_cast<const v4f *>(xr + 4 * (2 * t + hi)) : v4f{0.f, 0.f, 0.f, 0.f};
            }
        };
        fetch(a_cur, 0);
        // both halves of the wave run the same nmax steps: MFMA is a whole-wave instruction.  A half that
        // has no piece left (odd number of pieces) feeds zeros: its row fetch returns 0 and its query
        // read falls into the zero padding behind the row.
        for (uint32_t t0 = 0; t0 < nmax; t0 += MF_CH) {
            if (t0 + MF_CH < nmax) fetch(a_nxt, t0 + MF_CH);
#pragma unroll
            for (int c = 0; c < MF_CH; c++) {
                const uint32_t t = t0 + c;
                if (t < nmax) {  // wave-uniform
                    const v4f b = *reinterpret_cast<const v4f *>(qrow + 4 * (2 * t + hi));
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[c][u], b[u], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < MF_CH; c++) a_cur[c] = a_nxt[c];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile's norms are in LDS (own wave)
        // acc[v] = x_m . q_j for m = (v % 4) + 8 (v / 4) + 4 hi; score = |x_m|^2 - 2 x_m . q_j
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4f n4 = *reinterpret_cast<const v4f *>(xns + wave * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float sc = n4[u] - 2.0f * acc[4 * g + u];
                if (sc < thr) {
                    const uint32_t id = (uint32_t)(p0 + 8 * g + 4 * hi + u);
                    // replace the current worst, then find the new worst
                    bool done = false;
#pragma unroll
                    for (int t = 0; t < MF_K2; t++) {
                        if (!done && bs[t] == thr) {
                            bs[t] = sc;
                            bi[t] = id;
                            done = true;
                        }
                    }
                    float mx = bs[0];
#pragma unroll
                    for (int t = 1; t < MF_K2; t++) mx = fmaxf(mx, bs[t]);
                    thr = mx;
                }
            }
        }
    }
    const size_t o = ((((size_t)tile * nseg + seg) * 4 + wave) * 64 + lane) * MF_K2;
#pragma unroll
    for (int t = 0; t < MF_K2; t++) {
        out_s[o + t] = bs[t];
        out_i[o + t] = bi[t];
    }
}

// exact FullVec::distance (full.rs:23-29) of (query, point) pairs: one lane per pair, the single
// left-to-right chain, -ffp-contract=off
__global__ void __launch_bounds__(256)
hx_pair_distance_kernel(const float *X, uint32_t d, const float *Q, const uint32_t *qidx, const uint32_t *pidx,
                        uint64_t n, float *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pidx[i];
    if (p == HX_EMPTY_SLOT) {
        out[i] = __builtin_inff();
        return;
    }
    const float *x = X + (size_t)p * d, *q = Q + (size_t)qidx[i] * d;
    float s = 0.0f;
    for (uint32_t e = 0; e < d; e++) {
        const float t = x[e] - q[e];
        s += t * t;
    }
    out[i] = __builtin_sqrtf(s);
}

int launch_row_norms(const DevView &v, float *d_xn, hipStream_t stream) {
    hipLaunchKernelGGL(hx_row_norms_kernel, dim3((v.n_points + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const float *>(v.rows), v.n_points, v.dim, d_xn);
    return hipGetLastError() == hipSuccess ? HNSW_OK : HNSW_ERR_HIP;
}

uint32_t brute_mfma_k2() { return MF_K2; }

// out_s / out_i: [ntiles][nseg][4][64][MF_K2]
int launch_brute_mfma(const DevView &v, const float *d_xn, const float *d_Q, uint32_t nq, uint32_t nseg,
                      float *out_s, uint32_t *out_i, hipStream_t stream) {
    if (v.kind != HNSW_VEC_F32 || (v.dim & 3u) || v.row_stride != 4 * v.dim) {
        set_error("the MFMA scan serves f32 rows whose dimension is a multiple of 4");
        return HNSW_ERR_ARG;
    }
    const size_t lds = ((size_t)MF_QT * (v.dim + 4) + 4 * 32) * 4;
    if (lds > 160 * 1024) {
        set_error("the MFMA scan stages 32 queries in LDS: dimension %u is too large", v.dim);
        return HNSW_ERR_ARG;
    }
    auto kern = hx_brute_mfma_kernel;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    const uint32_t ntiles = (nq + MF_QT - 1) / MF_QT;
    hipLaunchKernelGGL(kern, dim3(nseg, ntiles), dim3(256), lds, stream, reinterpret_cast<const float *>(v.rows), d_xn,
                       v.n_points, v.dim, d_Q, nq, nseg, out_s, out_i);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("MFMA scan launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_pair_distance(const DevView &v, const float *d_Q, const uint32_t *d_qidx, const uint32_t *d_pidx, uint64_t n,
                         float *d_out, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_pair_distance_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float *>(v.rows), v.dim, d_Q, d_qidx, d_pidx, n, d_out);
    return hipGetLastError() == hipSuccess ? HNSW_OK : HNSW_ERR_HIP;
}

}  // namespace hx
