// build_sort.hip -- device radix sorts of the edge records of the on-device build (rocPRIM).
// Kept in its own translation unit: the rocPRIM headers are heavy and the search kernels do not
// need them.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "device_index.h"
#include "host_index.h"

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

}  // namespace hx
