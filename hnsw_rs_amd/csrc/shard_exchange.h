// shard_exchange.h -- helpers of the sharded on-device build (capi.cpp, gpu_insert_bulk_full with a ShardCtx) that are
// not kernels of the search or of the single-GPU build.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace hx {

// Of n edge records (hx_edge_key(layer, target row's node, other node), optional value), keep those whose target row
// belongs to `rank` of `world` (node id % world, the ownership rule of ConnectArgs) and append them to out_keys /
// out_vals at the device counter *out_count (order arbitrary: the radix sort that follows makes it canonical).
// Every rank receives every record; sorting and launching over its own share alone divides the sort and takes the
// workgroups of the other owners' rows out of the connect / drop launches.  *status: HNSW_ERR_OVERFLOW past out_cap.
int filter_edge_records(const uint64_t *keys, const uint32_t *vals, uint32_t n, uint32_t rank, uint32_t world,
                        uint64_t *out_keys, uint32_t *out_vals, uint32_t *out_count, uint32_t out_cap, int32_t *status,
                        hipStream_t stream);

}  // namespace hx
