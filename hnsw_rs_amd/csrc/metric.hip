// metric.hip -- the cosine option ("metric_cosine", an EXTENSION: the reference has only Euclidean distance,
// vectors/src/lib.rs:10-27, vectors/src/full.rs:23-29).  Cosine order is L2 order on unit vectors, so the
// option is a normalisation at the library's edge -- rows as they are inserted (host, capi.cpp) and queries
// as they arrive (here, in place on the device copy) -- and everything behind it is the reference's
// arithmetic unchanged.  Host and device normalise with the same operations in the same order: the sum of
// squares is one left-to-right f32 chain (no FMA), the square root and the divisions are the correctly
// rounded forms, so a query normalised here equals the same vector normalised on the host bit for bit.
// A vector without a direction (zero, or a norm outside f32's range) becomes NaN, which the search reports as
// HNSW_ERR_NAN_INPUT.
#include <hip/hip_runtime.h>

#include "device_index.h"

namespace hx {

__global__ void __launch_bounds__(64) hx_normalise_rows_kernel(float *rows, uint64_t n, uint32_t d) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float *x = rows + i * d;
    float s = 0.0f;
    for (uint32_t e = 0; e < d; e++) {
        const float t = x[e] * x[e];
        s += t;
    }
    float nrm = __builtin_sqrtf(s);
    // no direction (zero, or a sum of squares outside f32's range): the row becomes NaN, which the search reports as
    // HNSW_ERR_NAN_INPUT -- an overflowed norm would otherwise turn the query into an all-zero vector silently
    if (!(nrm > 0.0f) || nrm == __builtin_inff()) nrm = __builtin_nanf("");
    for (uint32_t e = 0; e < d; e++) x[e] = x[e] / nrm;
}

int launch_normalise_rows(float *d_rows, uint64_t n, uint32_t d, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_normalise_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_rows, n, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("normalise kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

}  // namespace hx
