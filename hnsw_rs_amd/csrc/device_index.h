// device_index.h -- the HBM-resident snapshot of a HostIndex and the launchers of the search
// kernels (search_kernels.hip).  gfx950 only.
//
// HBM layout (one allocation per array, all read-only during search):
//
//   rows      N x row_stride bytes.
//             QUANT8 (reference QuantVec, vectors/src/quant.rs:6-11): a row is two halves of
//             half_bytes each; half h (0/1) belongs to lane h of the lane pair that evaluates the
//             row and holds [min f32][delta f32][the 4 codes (8c+4h .. 8c+4h+3) of every full
//             8-chunk c, in chunk order][h == 0 only: the d % 8 tail codes][zero pad to 16 B].
//             Lane h therefore streams its four running sums of distance_unrolled
//             (quant.rs:14-37: sum j takes elements 8c + j) from contiguous 16-byte pieces.
//             d = 100: half = 8 + 48 + 4 -> 64 B, row = 128 B = one cache line.
//             F32 (reference FullVec, vectors/src/full.rs): d floats, padded to 16 B.
//   adj0      N x S0 u32: layer-0 adjacency in fixed-stride rows (S0 = pow2 >= 2m, 128 B at
//             m = 16), row index = node id, neighbour ids ascending, empty slots 0xFFFFFFFF.
//   adj_up    upper-layer rows, S1 = pow2 >= max(m, 8) slots each; the rows of one node for
//             layers 1..level are contiguous: row(id, l) = upper_base[id] + l - 1.
//   upper_base N u32 (0xFFFFFFFF for level-0 nodes).
//   fat       (optional, "inline rows") layer-0 blocks of S0 x row_stride bytes, one per node: slot k
//             of node i holds a COPY of the vector row of its k-th neighbour with that neighbour's
//             id in the last 4 (padding) bytes of half 0.  One expansion then costs ONE dependent,
//             fully coalesced 4-KiB read (m = 16, d = 100) instead of an adjacency row followed by
//             a 32-row gather, and the block of the predicted next candidate can be prefetched
//             while the current one is evaluated.  Costs S0 x the row bytes in HBM (4.1 GB per 1M
//             points at d = 100 -- the card has 288 GB); built when it fits the configured budget,
//             otherwise the compact rows + adj0 path is used.  Empty slots: id 0xFFFFFFFF.
//   ovf_off / ovf_nbrs  CSR of the neighbours that do not fit a row (degree > S happens:
//             SURVEY.md H6).  Such a row keeps S - 1 ids and its last slot holds
//             0x80000000 | overflow row.  Node ids are < 2^31 (enforced by the host index).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <vector>

#include "host_index.h"

namespace hx {

#define HX_EMPTY_SLOT 0xFFFFFFFFu
#define HX_OVF_FLAG 0x80000000u

struct DevView {  // kernel argument, passed by value
    const uint8_t *rows;
    const uint32_t *adj0;
    const uint32_t *adj_up;
    const uint32_t *upper_base;
    const uint32_t *ovf_off;
    const uint32_t *ovf_nbrs;
    const uint8_t *fat;   // inline-rows blocks of layer 0, or null
    uint64_t fat_stride;  // bytes per node block = S0 * row_stride
    uint32_t row_stride;  // bytes
    uint32_t half_bytes;  // QUANT8: bytes per half row
    uint32_t S0, S1;      // slots per adjacency row (powers of two)
    uint32_t n_points, dim;
    uint32_t nch4, rem;   // QUANT8: 4 * (dim / 8), dim % 8
    uint32_t nb_layers, ep;
    int32_t kind;
    uint32_t _pad;
};

struct SearchArgs {
    const float *Q;           // nq x dim (device)
    const uint32_t *qsel;     // optional: launch block b serves query qsel[b]
    const uint32_t *entries;  // optional explicit entry set (search_layer seam), else {ep}
    uint32_t n_entry;
    int32_t layer_hi, layer_lo;  // layers traversed: hi .. lo (inclusive, descending)
    uint32_t ef_upper, ef_bottom;  // ef for layers > layer_lo, and for layer_lo
    uint32_t n;                  // results per query
    uint32_t *out_ids;           // nq x n
    float *out_dists;            // nq x n or null
    uint32_t *out_counts;        // nq or null
    hnsw_query_stats *out_stats; // nq
    unsigned long long *dbg;     // diagnostic builds only (HX_STAMPS): nq x 6 cycle sums, else null
    uint32_t flags;              // bit 0: one row per pass (disables the two-row loops; HNSW_MI355X_ONE_ROW=1)
    // second level of the visited set (lists of eight / sixteen registers, ef > 320; set by the launcher): 1 << spill_log2
    // words of HBM per launched query, or null; lds_limit: ids the LDS level takes before it is closed (0 = 75 % of its slots)
    uint32_t *spill_tab;
    uint32_t spill_log2, lds_limit;
};

// one batch of the on-device build: insertion searches for point_ids[0..n) (search_kernels.hip)
struct InsertArgs {
    const uint32_t *point_ids;  // device
    const uint8_t *levels;      // device, level of every stored point
    uint32_t ef_cons, m;
    uint32_t max_layers;        // layer slots in the outputs
    uint32_t *out_ids;          // [n][max_layers][m], padded with 0xFFFFFFFF
    float *out_dists;           // [n][max_layers][m]
    int32_t *out_status;        // [n]
    // full on-device connect (optional, all null otherwise): a point whose searches all succeeded
    // writes its own rows and appends one reverse-edge request per selected neighbour
    uint32_t *adj0_mut, *adj_up_mut;
    // optional (round 4): the distance of every edge beside its id, same shape as the adjacency arrays, 0xFFFFFFFF =
    // not known yet -- what lets the connect phase prune without evaluating a single distance (see ConnectArgs)
    uint32_t *adjd0_mut, *adjd_up_mut;
    uint64_t *req_keys;         // hx_edge_key(layer, target, source)
    uint32_t *req_vals;         // bits of d(source, target)
    uint32_t *req_count;        // records reserved (may pass req_cap: see req_fail_base)
    uint32_t *req_fail_base;    // smallest base of a reservation that did not fit (host: 0xFFFFFFFF before the launch);
                                // the records written are [0, min(*req_count, *req_fail_base))
    uint32_t req_cap;
    uint32_t emit_own;          // 1: no own-row writes, records in both directions (sharded build)
    // optional (null otherwise): what the insertion searches read, summed over the launch -- [0] vector rows
    // (distance evaluations + staged rows), [1] adjacency rows, [2] the ids in them.  The build's algorithmic bytes.
    unsigned long long *counters;
};

// An edge record of the on-device connect, made to be radix-sorted: records of one adjacency row
// (layer, row node) become adjacent and ordered by the other node.  Needs ids < 2^30, layers < 16.
static constexpr uint32_t HX_EDGE_ID_BITS = 30;
static inline __host__ __device__ uint64_t hx_edge_key(uint32_t layer, uint32_t row_node, uint32_t other) {
    return ((uint64_t)layer << (2 * HX_EDGE_ID_BITS)) | ((uint64_t)row_node << HX_EDGE_ID_BITS) | other;
}

// phase 2 / 3 of the on-device connect: sorted edge records, one wave per record; the wave of the
// first record of a row handles the row, the others exit
struct ConnectArgs {
    const uint64_t *keys;   // sorted
    const uint32_t *vals;   // phase 2: distance bits in the same order; phase 3: unused
    uint32_t count;
    uint32_t m;
    uint32_t *adj0_mut, *adj_up_mut;
    // Edge distances kept beside the adjacency for the length of a build (optional, null = evaluate as before).  A prune
    // needs d(x, n) for every neighbour n of the row's node x (select_simple, template.rs:614-621); each of them was
    // computed when its edge was made -- by the insertion search of one endpoint, d being bit-symmetric -- so it is
    // stored with the edge (0xFFFFFFFF = not known: rows that predate the build; evaluated once, then kept) and a
    // prune becomes a sort of 33 keys instead of a gather of 32 vector rows (a third of a 16M x 256d build's
    // sort + connect + remove time was those gathers, through the any-dimension distance loop at that).
    uint32_t *adjd0_mut, *adjd_up_mut;
    uint64_t *out_keys;     // phase 2: removals hx_edge_key(layer, x, n); phase 3: refusals (same form)
    uint32_t *out_count;
    uint32_t out_cap;
    int32_t *status;        // single word: set when `out` is full or a record is malformed
    // Sharded build (round 4): with own_world > 1 a wave handles only the rows whose node id % own_world == own_rank --
    // every rank sees every record, each row has ONE owner that appends / prunes / drops -- and files the row in
    // chg_keys (hx_edge_key(layer, node, 0); duplicates allowed) so that its new contents can be shipped to the other
    // replicas (hx_pack_rows_kernel / hx_apply_rows_kernel, patch.hip).  The file is HX_CHG_LISTS lists of chg_cap
    // keys each with a counter each, chosen by the workgroup's index: millions of appends per batch through ONE
    // counter cost as much as the connect itself.  own_world <= 1: every row, nothing filed.
    uint32_t own_rank, own_world;
    uint64_t *chg_keys;   // [HX_CHG_LISTS][chg_cap]
    uint32_t *chg_count;  // [HX_CHG_LISTS]
    uint32_t chg_cap;
};
constexpr uint32_t HX_CHG_LISTS = 64;

// rows that changed on their owner, packed for the exchange: entry i = [key u64][ship_slots ids u32], and back.
// `keys` / `counts`: the HX_CHG_LISTS lists of list_cap keys of ConnectArgs; max_count = the longest list (the host
// has read the counters); the entries leave in list order, sum(counts) of them.
int launch_pack_rows(const DevView &v, const uint32_t *adj0, const uint32_t *adj_up, const uint64_t *keys,
                     const uint32_t *counts, uint32_t list_cap, uint32_t max_count, uint32_t ship_slots, unsigned char *out,
                     hipStream_t stream);
int launch_apply_rows(const DevView &v, uint32_t *adj0, uint32_t *adj_up, const unsigned char *entries, uint32_t n,
                      uint32_t ship_slots, int32_t *status, hipStream_t stream);

// device radix sorts of the edge records (build_sort.hip, rocPRIM)
size_t sort_temp_bytes(uint32_t max_n);
int sort_edge_pairs(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                    const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, uint32_t nb_layers,
                    hipStream_t stream);
int sort_edge_keys(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, uint32_t n,
                   uint32_t nb_layers, hipStream_t stream);

// one piece of a snapshot patch (patch.hip): n_words 32-bit words of the staging buffer, from word src_word on,
// are written to dst
struct PatchDesc {
    void *dst;
    uint32_t src_word, n_words;
};
int launch_patch(const PatchDesc *d_desc, const uint32_t *d_staging, uint32_t n, hipStream_t stream);
int launch_fat_rebuild(const DevView &v, uint8_t *fat, const uint32_t *d_nodes, uint32_t n, hipStream_t stream);

class DeviceIndex {
  public:
    ~DeviceIndex() { release(); }
    // meanwhile (optional) runs on another thread while the vector rows -- the bulk of a snapshot, and immutable once
    // stored -- are on the wire; the adjacency is packed after it has returned (the on-device build seeds its graph
    // on the host there: 0.8 s beside a 0.7-s upload at 16M x 256d)
    int upload(const HostIndex &idx, int device, const std::function<int()> &meanwhile = {});
    // inline-rows layout: -1 = auto (build it when it fits fat_budget_bytes), 0 = never, 1 = always
    int inline_rows = -1;
    uint64_t fat_budget_bytes = 64ull << 30;
    void release();
    bool current(const HostIndex &idx) const { return valid && (replica || version_seen == idx.version); }
    // After an on-device build the adjacency in HBM IS the graph the host just read back, except for the few
    // rows the host touched afterwards (kept-last-edge records): those rows are re-packed and copied, and the
    // snapshot is declared current for idx.version -- instead of uploading tens of GB again.  Returns false
    // (snapshot left as it was, the next search uploads) when that is not possible: inline rows wanted, or
    // overflow lists already present.  layer_row: (layer << 32) | node id.
    bool refresh_rows(const HostIndex &idx, const std::vector<uint64_t> &layer_row);
    // HNSW::insert_vec on a live snapshot (template.rs:165-173; its callers search right after it,
    // eval_glove/src/main.rs:37-41): point `id` = the last one of idx has just been stored and connected on the host
    // and layer_row lists the adjacency rows that insertion touched.  The new vector row, its upper_base word and
    // the touched rows (with their overflow lists, and the inline-rows blocks where that copy exists) are packed
    // into one staging buffer and written by one kernel (patch.hip); the arrays have room for the next points or
    // are grown by a device-to-device copy (capacity + 1/8), so an insertion costs O(rows touched), not O(N).
    // Returns false -- snapshot left stale, the next search uploads -- when the snapshot is not the state before
    // this insertion or a HIP call fails.
    bool append_point(const HostIndex &idx, NodeID id, const std::vector<uint64_t> &layer_row);
    // The adjacency array of layer 0 (which = 0) or of the upper layers (1) back to the host, in pieces through two
    // pinned buffers: consume(row_lo, row_hi, slots) sees rows [row_lo, row_hi) of S slots each while the next piece is
    // on the wire (the read-back at the end of an on-device build: 12.8 GB at 100M points).
    int read_adjacency(int which, uint64_t n_rows,
                       const std::function<void(uint64_t, uint64_t, const uint32_t *)> &consume);
    bool wants_inline_rows(const HostIndex &idx) const;

    // mutable views of the adjacency arrays (the on-device build scatters dirty rows into them)
    uint32_t *adj0_mut() { return static_cast<uint32_t *>(bufs_[1]); }
    uint32_t *adj_up_mut() { return static_cast<uint32_t *>(bufs_[2]); }

    // replication (hnsw_snapshot_*): the seven flat arrays of a valid snapshot; an empty DeviceIndex takes
    // arrays of given sizes (adopt_alloc), the caller fills them, adopt_commit installs the view
    void describe(uint64_t (&nbytes)[7], void *(&ptrs)[7]) const;
    int adopt_alloc(int device, const uint64_t (&nbytes)[7], void *(&ptrs)[7]);
    void adopt_commit(const DevView &scalars);

    bool valid = false;
    bool replica = false;  // adopted from another rank: there is no host index behind it
    int device = -1;
    uint64_t version_seen = 0;
    uint64_t bytes = 0;
    DevView view{};

  private:
    void *bufs_[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t sizes_[7] = {0, 0, 0, 0, 0, 0, 0};  // bytes in use (what a replica receives)
    uint64_t caps_[7] = {0, 0, 0, 0, 0, 0, 0};   // bytes allocated (append_point leaves room behind the arrays)
    uint64_t n_ovf_nbrs_ = 0;                    // ids in the overflow lists (array 5 holds one dummy word when none)
    bool grow(int i, uint64_t need_bytes, int fill_byte);
    // staging of append_point: one pinned buffer, one device buffer, one stream
    void *pin_ = nullptr, *stage_ = nullptr;
    size_t pin_cap_ = 0, stage_cap_ = 0;
    hipStream_t pstream_ = nullptr;
};

// row packing helpers shared by upload and tests
uint32_t quant_half_bytes(uint32_t dim);
uint32_t f32_row_stride(uint32_t dim);
uint32_t adj_stride(uint64_t cap, uint32_t min_slots);

// Launch the search kernel for `nblocks` queries on `stream`.  slots_log2 = log2 of the visited
// hash table size (LDS), 0 = choose from ef.  Returns HNSW_OK or HNSW_ERR_ARG / HNSW_ERR_HIP.
int launch_search(const DevView &v, const SearchArgs &a, uint32_t nblocks, uint32_t slots_log2,
                  hipStream_t stream);
// the latency-critical form of the same search (search_lean.hip): whole ann_by_vector descents over
// f32 rows; launch_search routes to it when it applies
bool lean_applicable(const DevView &v, const SearchArgs &a, uint32_t ef_max);
int launch_lean(const DevView &v, const SearchArgs &a, uint32_t nblocks, uint32_t slots_log2, hipStream_t stream);
uint32_t default_slots_log2(uint32_t ef);
uint32_t default_slots_log2(uint32_t ef, uint32_t s0);
uint32_t max_slots_log2(uint32_t ef);

int launch_insert(const DevView &v, const InsertArgs &a, uint32_t nblocks, hipStream_t stream, int table_adjust = 0);
int insert_table_first_adjust(const DevView &v, const InsertArgs &a);
int launch_connect(const DevView &v, const ConnectArgs &a, hipStream_t stream);
int launch_remove(const DevView &v, const ConnectArgs &a, hipStream_t stream);
int launch_scatter_rows(uint32_t *dst, uint32_t S, const uint32_t *d_row_index, const uint32_t *d_data,
                        uint32_t n, hipStream_t stream);

// exhaustive scan on the matrix cores (brute_mfma.hip): a screen by MFMA scores, then exact distances of
// the survivors.  f32 rows, dimension a multiple of 4.
uint32_t brute_mfma_k2();
int launch_row_norms(const DevView &v, float *d_xn, hipStream_t stream);
int launch_brute_mfma(const DevView &v, const float *d_xn, const float *d_Q, uint32_t nq, uint32_t nseg,
                      float *out_s, uint32_t *out_i, hipStream_t stream);
int launch_pair_distance(const DevView &v, const float *d_Q, const uint32_t *d_qidx, const uint32_t *d_pidx, uint64_t n,
                         float *d_out, hipStream_t stream);

// the cosine option: rows (queries) normalised in place to unit length (metric.hip)
int launch_normalise_rows(float *d_rows, uint64_t n, uint32_t d, hipStream_t stream);

// out[i] = dist(point ids[i], query) for one query; d_q is the raw query (dim floats, device)
int launch_distance_batch(const DevView &v, const float *d_q, const uint32_t *d_ids, uint64_t k,
                          float *d_out, int32_t *d_status, hipStream_t stream);

// exhaustive scan: every query against every point; partial top-k per (query, segment):
// part_ids / part_dists are nq x nseg x k.  The host merges the segments.
int launch_brute_force(const DevView &v, const float *d_Q, uint64_t nq, uint32_t k, uint32_t nseg,
                       uint32_t *part_ids, float *part_dists, int32_t *d_status,
                       hipStream_t stream);

}  // namespace hx
