// patch.hip -- keeping the HBM snapshot current across HNSW::insert_vec (hnsw/src/template.rs:165-173) without
// uploading it again.  The reference's callers insert one vector and search straight away
// (eval_glove/src/main.rs:37-41); one insertion touches the new point's row and a few dozen adjacency rows, so
// the host packs exactly those into one staging buffer and one launch of hx_patch_kernel writes them where they
// belong (device_index.cpp, DeviceIndex::append_point).  Where the optional inline-rows copy of layer 0 exists,
// hx_fat_rebuild_kernel re-derives the blocks of the touched nodes from the patched rows and adjacency.
#include <hip/hip_runtime.h>

#include "device_index.h"

namespace hx {

// one workgroup per descriptor: n_words 32-bit words from the staging buffer to their place in the snapshot
__global__ void __launch_bounds__(64) hx_patch_kernel(const PatchDesc *desc, const uint32_t *staging, uint32_t n) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const PatchDesc dsc = desc[i];
    uint32_t *dst = reinterpret_cast<uint32_t *>(dsc.dst);
    const uint32_t *src = staging + dsc.src_word;
    for (uint32_t k = threadIdx.x; k < dsc.n_words; k += 64) dst[k] = src[k];
}

// one workgroup per touched layer-0 node: slot k of its block = a copy of the k-th neighbour's vector row with
// that neighbour's id in the last 4 bytes of half 0 (device_index.h, "fat")
__global__ void __launch_bounds__(64) hx_fat_rebuild_kernel(DevView v, uint8_t *fat, const uint32_t *nodes, uint32_t n) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t node = nodes[i];
    const uint32_t words = v.row_stride / 4, idpos = (v.half_bytes - 4) / 4;
    for (uint32_t k = 0; k < v.S0; k++) {
        const uint32_t nb = v.adj0[(size_t)node * v.S0 + k];
        uint32_t *o = reinterpret_cast<uint32_t *>(fat + (size_t)node * v.fat_stride + (size_t)k * v.row_stride);
        const bool real = nb != HX_EMPTY_SLOT && !(nb & HX_OVF_FLAG);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(v.rows + (size_t)(real ? nb : 0) * v.row_stride);
        for (uint32_t w = threadIdx.x; w < words; w += 64) o[w] = w == idpos ? nb : (real ? src[w] : 0u);
    }
}

// ---- sharded build: the rows their owner changed travel to the other replicas (ConnectArgs, device_index.h) ----
static constexpr uint64_t PK_ID_MASK = (1ull << HX_EDGE_ID_BITS) - 1;

__global__ void __launch_bounds__(64) hx_pack_rows_kernel(DevView v, const uint32_t *adj0, const uint32_t *adj_up,
                                                          const uint64_t *keys, const uint32_t *counts, uint32_t list_cap,
                                                          uint32_t ship, unsigned char *out) {
    static_assert(HX_CHG_LISTS == 64, "one lane per list below");
    const uint32_t list = blockIdx.y, i = blockIdx.x;
    const uint32_t mine = counts[threadIdx.x];
    if (i >= __shfl(mine, (int)list)) return;
    uint32_t before = threadIdx.x < list ? mine : 0;  // entries of the lists in front of this one
    for (int o = 32; o; o >>= 1) before += __shfl_xor(before, o);
    const uint64_t key = keys[(size_t)list * list_cap + i];
    const uint32_t node = (uint32_t)((key >> HX_EDGE_ID_BITS) & PK_ID_MASK), layer = (uint32_t)(key >> (2 * HX_EDGE_ID_BITS));
    const uint32_t S = layer == 0 ? v.S0 : v.S1;
    const uint32_t *row = layer == 0 ? adj0 + (size_t)node * S : adj_up + ((size_t)v.upper_base[node] + layer - 1) * S;
    uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)(before + i) * (8 + 4ull * ship));
    if (threadIdx.x == 0) {
        o[0] = (uint32_t)key;
        o[1] = (uint32_t)(key >> 32);
    }
    for (uint32_t k = threadIdx.x; k < ship; k += 64) o[2 + k] = k < S ? row[k] : HX_EMPTY_SLOT;
}

__global__ void __launch_bounds__(64) hx_apply_rows_kernel(DevView v, uint32_t *adj0, uint32_t *adj_up,
                                                           const unsigned char *entries, uint32_t n, uint32_t ship, int32_t *status) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t *e = reinterpret_cast<const uint32_t *>(entries + (size_t)i * (8 + 4ull * ship));
    const uint64_t key = ((uint64_t)e[1] << 32) | e[0];
    const uint32_t node = (uint32_t)((key >> HX_EDGE_ID_BITS) & PK_ID_MASK), layer = (uint32_t)(key >> (2 * HX_EDGE_ID_BITS));
    if (node >= v.n_points || layer >= v.nb_layers || (layer > 0 && v.upper_base[node] == HX_EMPTY_SLOT)) {
        if (threadIdx.x == 0) *status = HNSW_ERR_NODE_NOT_IN_GRAPH;  // a malformed entry: never touch memory for it
        return;
    }
    const uint32_t S = layer == 0 ? v.S0 : v.S1;
    uint32_t *row = layer == 0 ? adj0 + (size_t)node * S : adj_up + ((size_t)v.upper_base[node] + layer - 1) * S;
    for (uint32_t k = threadIdx.x; k < S && k < ship; k += 64) row[k] = e[2 + k];
}

int launch_pack_rows(const DevView &v, const uint32_t *adj0, const uint32_t *adj_up, const uint64_t *keys,
                     const uint32_t *counts, uint32_t list_cap, uint32_t max_count, uint32_t ship_slots, unsigned char *out,
                     hipStream_t stream) {
    if (max_count == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_pack_rows_kernel, dim3(max_count, HX_CHG_LISTS), dim3(64), 0, stream, v, adj0, adj_up, keys, counts,
                       list_cap, ship_slots, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("pack-rows kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_apply_rows(const DevView &v, uint32_t *adj0, uint32_t *adj_up, const unsigned char *entries, uint32_t n,
                      uint32_t ship_slots, int32_t *status, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_apply_rows_kernel, dim3(n), dim3(64), 0, stream, v, adj0, adj_up, entries, n, ship_slots, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("apply-rows kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_patch(const PatchDesc *d_desc, const uint32_t *d_staging, uint32_t n, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_patch_kernel, dim3(n), dim3(64), 0, stream, d_desc, d_staging, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("patch kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int launch_fat_rebuild(const DevView &v, uint8_t *fat, const uint32_t *d_nodes, uint32_t n, hipStream_t stream) {
    if (n == 0) return HNSW_OK;
    hipLaunchKernelGGL(hx_fat_rebuild_kernel, dim3(n), dim3(64), 0, stream, v, fat, d_nodes, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("inline-rows rebuild kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

}  // namespace hx
