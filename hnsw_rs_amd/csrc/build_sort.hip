// build_sort.hip -- device radix sorts of the edge records of the on-device build (rocPRIM).
// Kept in its own translation unit: the rocPRIM headers are heavy and the search kernels do not
// need them.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "device_index.h"
#include "host_index.h"
#include "shard_exchange.h"

namespace hx {

static unsigned edge_key_bits(uint32_t nb_layers) {
    unsigned lb = 1;
    while ((1u << lb) < nb_layers) lb++;
    return 2 * HX_EDGE_ID_BITS + lb;
}

size_t sort_temp_bytes(uint32_t max_n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, max_n, 0u, 64u);
    (void)rocprim::radix_sort_keys(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr, max_n, 0u, 64u);
    return (a > b ? a : b) + 256;
}

int sort_edge_pairs(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                    const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, uint32_t nb_layers,
                    hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u,
                                             edge_key_bits(nb_layers), stream);
    if (e != hipSuccess) {
        set_error("radix_sort_pairs: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int sort_edge_keys(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, uint32_t n,
                   uint32_t nb_layers, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipError_t e =
        rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, 0u, edge_key_bits(nb_layers), stream);
    if (e != hipSuccess) {
        set_error("radix_sort_keys: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

// ---- sharded build: a rank's share of the records every rank received (shard_exchange.h) ----
__global__ void __launch_bounds__(256) hx_filter_records_kernel(const uint64_t *keys, const uint32_t *vals, uint32_t n, uint32_t rank,
                                                                uint32_t world, uint64_t *out_keys, uint32_t *out_vals,
                                                                uint32_t *out_count, uint32_t out_cap, int32_t *status) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint64_t key = 0;
    bool keep = false;
    if (i < n) {
        key = keys[i];
        keep = (uint32_t)((key >> HX_EDGE_ID_BITS) & ((1ull << HX_EDGE_ID_BITS) - 1)) % world == rank;
    }
    const unsigned long long m = __ballot(keep);
    if (m == 0) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(out_count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
    if (keep) {
        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pos < out_cap) {
            out_keys[pos] = key;
            if (vals) out_vals[pos] = vals[i];
        } else {
            *status = HNSW_ERR_OVERFLOW;
        }
    }
}

int filter_edge_records(const uint64_t *keys, const uint32_t *vals, uint32_t n, uint32_t rank, uint32_t world,
                        uint64_t *out_keys, uint32_t *out_vals, uint32_t *out_count, uint32_t out_cap, int32_t *status,
                        hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_filter_records_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, keys, vals, n, rank, world, out_keys,
                       out_vals, out_count, out_cap, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("record filter kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

}  // namespace hx
