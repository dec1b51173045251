/*
 * hnsw_mi355x.h -- C ABI of libhnsw_mi355x.so: an MI355X-native (gfx950 / CDNA4) HNSW engine that
 * drops in behind the public API of the Rust `hnsw` crate of Gumo-A/hnsw_rs.
 *
 * The reference has no FFI seam of its own: callers link the `hnsw` crate and use
 * hnsw::template::HNSW directly.  The seam preserved here is that public Rust API; a same-named
 * Rust shim (shim-rust/, source only) binds these entry points -- see INTEGRATION.md.
 * Every entry point cites the reference item it replaces (paths relative to the reference
 * repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every in/out buffer, nothing is retained
 *     after return; an opaque handle owns the host index and its HBM-resident snapshot.
 *   - every fallible call returns an int status (0 = HNSW_OK, < 0 = error); the text of the
 *     last error on the calling thread is available from hnsw_last_error().  Nothing unwinds or
 *     aborts across the ABI; where the reference panics (dim mismatch, NaN distance) the shim
 *     maps the status back to a panic, where it returns Err(String) to Err(String).
 *   - vectors are row-major float32; ids are the reference's NodeID = u32 (graph/src/lib.rs:1),
 *     dense 0..N-1 in insertion order (points/src/points.rs:64-73); UINT32_MAX pads id outputs.
 *   - search runs ONLY on the GPU (hand-written HIP, gfx950).  There is no CPU fallback: without
 *     a usable device the search entry points fail with HNSW_ERR_NO_DEVICE / HNSW_ERR_HIP.
 *   - concurrent searches on one handle are safe (the reference's ann_by_vector takes &self);
 *     insert_* must not run concurrently with anything else on the same handle.
 */
#ifndef HNSW_MI355X_H
#define HNSW_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define HNSW_OK 0
#define HNSW_ERR_BAD_DIM (-1)           /* template.rs:253-262 panics */
#define HNSW_ERR_NAN_INPUT (-2)         /* graph/src/dist.rs:32, vectors/src/quant.rs:44,48 panic */
#define HNSW_ERR_NODE_NOT_IN_GRAPH (-3) /* searcher.rs:45-50 Err(String) */
#define HNSW_ERR_IO (-4)                /* template.rs:75-93 Err(String) / save panics */
#define HNSW_ERR_HIP (-5)
#define HNSW_ERR_RCCL (-6)
#define HNSW_ERR_OOM (-7)
#define HNSW_ERR_ARG (-8)
#define HNSW_ERR_EMPTY (-9)             /* index has no points (template.rs:318 unwrap panics) */
#define HNSW_ERR_NO_DEVICE (-10)
#define HNSW_ERR_OVERFLOW (-11)         /* internal scratch exhausted even after the retry path */
#define HNSW_ERR_SELF_CONNECTION (-12)  /* graph/src/graph.rs:38-40 */

/* ---- vector kinds ------------------------------------------------------------------------- */
/* points/src/point.rs:4  `type VecType = QuantVec` is what the reference ships: 8-bit scalar
 * quantisation with on-the-fly dequantised f32 L2 (vectors/src/quant.rs).  HNSW_VEC_F32 is the
 * reference's alternate `VecType = FullVec` (vectors/src/full.rs). */
#define HNSW_VEC_QUANT8 0
#define HNSW_VEC_F32 1

typedef struct hnsw_index hnsw_index;

/* hnsw/src/params.rs:5-13 `pub struct Params` (a public field of HNSW, template.rs:37) */
typedef struct hnsw_params {
    uint32_t ep;
    uint32_t vec_kind;
    uint64_t m;
    uint64_t mmax;
    uint64_t mmax0;
    float ml;
    uint32_t _pad;
    uint64_t ef_cons;
    uint64_t dim;
} hnsw_params;

/* per-query traversal counters (the reference has none; they define the algorithmic bytes of
 * the roofline report): n_dist = distance evaluations (incl. the entry point), n_exp = expanded
 * candidates, sum_deg = sum of the degrees of the expanded adjacency rows. */
typedef struct hnsw_query_stats {
    uint32_t n_dist;
    uint32_t n_exp;
    uint32_t sum_deg;
    int32_t status; /* per-query status (HNSW_OK or an error code) */
} hnsw_query_stats;

const char *hnsw_last_error(void);
const char *hnsw_version(void);

/* ---- construction -------------------------------------------------------------------------- */
/* HNSW::new(m, ef_cons: Option<usize>, dim), template.rs:133-144; ef_cons == 0 means None
 * (defaults ef_cons = 2m, mmax = m, mmax0 = 2m, ml = 1/ln(m): params.rs:20-42). */
int hnsw_create(uint32_t m, uint32_t ef_cons, uint32_t dim, int vec_kind, hnsw_index **out);
void hnsw_free(hnsw_index *h);
/* #[derive(Clone)] on HNSW, template.rs:35 (benches clone the index, hnsw_benchmarks.rs:23) */
int hnsw_clone(const hnsw_index *h, hnsw_index **out);

int hnsw_get_params(const hnsw_index *h, hnsw_params *out);
/* params.ep is a public field; the reference re-picks it in hash order at every store
 * (template.rs:283-290) -- here the default is the smallest id on the top layer. */
int hnsw_set_ep(hnsw_index *h, uint32_t ep);

/* ---- build --------------------------------------------------------------------------------- */
/* HNSW::insert_bulk(self, vectors: Vec<Vec<f32>>, nb_threads, verbose), template.rs:388-444.
 * rows is n x dim row-major (the shim flattens Vec<Vec<f32>> and checks every row length,
 * returning HNSW_ERR_BAD_DIM where template.rs:253-262 panics).  May be called repeatedly. */
int hnsw_insert_bulk(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose);
/* same, with the level of every new point given explicitly (levels[n]; NULL = draw them).  The
 * reference draws levels from rand's StdRng re-seeded with 0 at every store
 * (points/src/points.rs:39-48,148-160); rand is not part of the reference tree, so
 * reproducible tests pass levels in. */
int hnsw_insert_bulk_levels(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads,
                            int verbose, const uint8_t *levels);
/* insert_bulk on the GPU (on-device build), batch-synchronous.  Per batch one
 * wave per point runs Inserter::build_insertion_results (inserter.rs:40-126: entry point, greedy
 * descent, search_layer(ef_cons) + select_heuristic per layer) against the graph in HBM and files one
 * reverse-edge request per selected neighbour; the requests are radix-sorted by target row and one
 * wave per row applies make_connections / prune_connections (template.rs:196-238: append, or keep
 * the cap's nearest), a third pass drops the reverse edges of what was pruned (graph.rs:72-94, a
 * node's last edge stays).  The host graph is read back once at the end.  Points of one batch do not
 * see each other (like the racing threads of the reference's multi-threaded insert_bulk), so the graph
 * is judged by recall and invariants, not by identity with the sequential build.  Needs m <= 128 and
 * ef_construction <= 512; nb_threads is used for the host-side parts (store, seed, read-back).
 * hnsw_set_option(h, "gpu_build", 2) routes hnsw_insert_bulk here; "gpu_build" = 1 selects the older
 * hybrid form (GPU searches, connect / prune on nb_threads host threads with the reference's locks). */
int hnsw_insert_bulk_device(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads,
                            int verbose, const uint8_t *levels);
/* The on-device build sharded over the GPUs of a node (BASELINE configs[4]): every rank calls this with
 * the SAME rows / levels on its own replica.  The insertion searches of each batch are split over the
 * ranks by position and what they produce travels as edge records through an all-gather; the connect / prune /
 * drop phases are split by row ownership (node id % world: a row's outcome depends on that row and its records
 * alone), the removals and the rows each owner changed travelling through further all-gathers of the size the
 * batch needs; the replicas are identical after every batch.  The collective is the caller's: d_send (one slot)
 * and d_recv (world slots) are device buffers of hnsw_sharded_slot_bytes() per slot, and
 * `allgather(ctx, bytes_per_rank)` (bytes_per_rank <= the slot size, a multiple of 64) must all-gather the FIRST
 * bytes_per_rank bytes of every rank's d_send into d_recv, rank r's at offset r * bytes_per_rank, and return 0
 * once d_recv is complete (RCCL through torch.distributed in hnsw_rs_amd/hnsw.py).  The library has synchronised
 * the device before it calls.  Every rank makes the same sequence of calls with the same sizes. */
typedef int (*hnsw_allgather_fn)(void *ctx, uint64_t bytes_per_rank);
uint64_t hnsw_sharded_slot_bytes(const hnsw_index *h, uint32_t world);
int hnsw_insert_bulk_sharded(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose,
                             const uint8_t *levels, uint32_t rank, uint32_t world, void *d_send, void *d_recv,
                             uint64_t slot_bytes, hnsw_allgather_fn allgather, void *ctx);
/* HNSW::insert_vec(&mut self, &Vec<f32>) -> Result<NodeID, String>, template.rs:165-173.  The reference's callers
 * search right after it (eval_glove/src/main.rs:37-41): when the HBM snapshot is current it is PATCHED -- the new
 * vector row, its upper-layer base and the adjacency rows the insertion touched go to the device in one staging
 * copy and one kernel, the arrays growing by a device-to-device copy when they are full -- so an insertion costs
 * O(rows touched) and the next search finds the snapshot current (hnsw_get_stat "point_patches" / "uploads"). */
int hnsw_insert_vec(hnsw_index *h, const float *v, uint32_t *out_id);
int hnsw_insert_vec_level(hnsw_index *h, const float *v, int level /* < 0: draw */, uint32_t *out_id);

/* Adopt a prebuilt graph (e.g. one built by another implementation of the reference) instead of
 * building: points first, then layers 0,1,2,... as CSR over ascending node ids, then the ep. */
int hnsw_import_points(hnsw_index *h, const float *rows, uint64_t n, const uint8_t *levels);
int hnsw_import_layer(hnsw_index *h, uint32_t layer, uint64_t n_nodes, const uint32_t *node_ids,
                      const uint64_t *offsets, const uint32_t *nbrs);

/* ---- query (the hot path; HIP kernels) ------------------------------------------------------ */
/* HNSW::ann_by_vector(&self, &Vec<f32>, n, ef) -> Result<Vec<NodeID>, String>, template.rs:306-335.
 * ids[n]; *count = number of ids returned (< n when ef < n or the index is tiny).
 * No limit on ef, like the reference: up to 1024 the candidate list lives in one wave's registers; beyond that
 * list and visited set live in HBM scratch (hx_search_spill_kernel: the same results, one insertion at a time,
 * orders of magnitude slower -- meant for correctness at the reference's contract, not for throughput). */
/* Concurrent calls on one handle are COALESCED (the reference takes &self, so its callers are many threads each
 * blocked in its own call): the first caller to arrive leads a batch, callers arriving with the same (n, ef) before
 * it is launched park their query in the batch's pinned staging area and sleep; the leader launches one kernel for
 * all of them and wakes them with their ids.  Every query is still answered by its own wave, so a call returns
 * exactly what it would return alone.  A lone caller launches at once; a leader that has seen concurrency waits up
 * to "coalesce_us" (hnsw_set_option; default 30, 0 = never wait, < 0 = coalescing off) for the callers that were
 * woken together to come back; at most "coalesce_depth" (2) batches are on the GPU at a time and a batch holds at
 * most "coalesce_max" (1024) queries. */
int hnsw_search(hnsw_index *h, const float *q, uint32_t n, uint32_t ef, uint32_t *ids,
                uint32_t *count);
/* Batched form (new; the reference answers one query per call): Q is nq x dim host memory,
 * ids nq x n (pad UINT32_MAX), dists nq x n or NULL (the distances the reference discards, pad
 * +inf), counts nq or NULL, stats nq or NULL.  Returns the first per-query error, if any. */
int hnsw_search_batch(hnsw_index *h, const float *Q, uint64_t nq, uint32_t n, uint32_t ef,
                      uint32_t *ids, float *dists, uint32_t *counts, hnsw_query_stats *stats);
/* Same with every buffer already resident in HBM on the handle's device; enqueues on `stream`
 * (a hipStream_t, NULL = default stream) and returns without synchronising.  d_stats is
 * required (its status field carries per-query errors); d_dists / d_counts may be NULL. */
int hnsw_search_batch_device(hnsw_index *h, const float *d_Q, uint64_t nq, uint32_t n, uint32_t ef,
                             uint32_t *d_ids, float *d_dists, uint32_t *d_counts,
                             hnsw_query_stats *d_stats, void *stream);

/* Completes a hnsw_search_batch_device call (same arguments): waits for `stream`, reads the per-query
 * statuses, re-runs the queries whose visited table filled up with a larger table (the results are those a
 * larger table would have given from the start) and returns the first remaining per-query error -- the
 * error behaviour of the reference's `Result` (template.rs:306, 323) for callers that keep everything in
 * HBM.  The device entry above never synchronises; a caller that skips this call must inspect
 * d_stats[i].status itself. */
int hnsw_search_batch_device_finish(hnsw_index *h, const float *d_Q, uint64_t nq, uint32_t n, uint32_t ef,
                                    uint32_t *d_ids, float *d_dists, uint32_t *d_counts,
                                    hnsw_query_stats *d_stats, void *stream);

/* Test seams that mirror the reference's own units:
 * VecBase::dist2many (vectors/src/lib.rs:17-22): the query (quantised like ann_by_vector does,
 * template.rs:313) against stored ids, on the device, exact accumulation order. */
int hnsw_distance_batch(hnsw_index *h, const float *q, const uint32_t *ids, uint64_t k, float *out);
/* Searcher::search_layer (searcher.rs:23-103) on one layer from an explicit entry set. */
int hnsw_search_layer(hnsw_index *h, uint32_t layer, const float *q, const uint32_t *entry_ids,
                      uint32_t n_entry, uint32_t ef, uint32_t *out_ids, float *out_dists,
                      uint32_t *out_count, hnsw_query_stats *stats);
/* Exact top-k under the index's own metric by exhaustive scan on the device (the reference's
 * brute force: helpers/glove.rs:94-109, template.rs:531-541). */
int hnsw_brute_force(hnsw_index *h, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids,
                     float *dists);
/* The same ground truth on the matrix cores, for sizes where the exact scan takes minutes (f32 rows,
 * dimension a multiple of 4, k <= 12): every point is screened by an MFMA score |x|^2 - 2 x.q
 * (v_mfma_f32_32x32x2_f32: f32 products and sums, but not FullVec::distance's summation order,
 * vectors/src/full.rs:23-29), the k + 8 best per query are re-evaluated in the reference's exact
 * arithmetic and sorted by (dist, id).  NOT bit-exact by construction -- rounding in the screen could in
 * principle lose a true neighbour; hnsw_brute_force is the exact scan.  An extension: the reference has
 * no counterpart. */
int hnsw_brute_force_fast(hnsw_index *h, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids,
                          float *dists);

/* ---- accessors ----------------------------------------------------------------------------- */
uint64_t hnsw_len(const hnsw_index *h);                                   /* template.rs:146 */
/* HNSW::distance(a, b) -> Option<f32>, template.rs:150-152: HNSW_ERR_ARG stands for None */
int hnsw_distance(const hnsw_index *h, uint32_t a, uint32_t b, float *out);
/* get_point(id).get_vals(): the (dequantised) values, template.rs:154, vectors/src/lib.rs:24-26 */
int hnsw_get_vector(const hnsw_index *h, uint32_t id, float *out);
int hnsw_get_level(const hnsw_index *h, uint32_t id, uint32_t *out);
/* raw QuantVec fields (vectors/src/quant.rs:6-11); HNSW_ERR_ARG for an F32 index */
int hnsw_get_quant(const hnsw_index *h, uint32_t id, uint8_t *codes, float *min_out, float *delta_out);
uint32_t hnsw_layer_count(const hnsw_index *h);                           /* layers.rs:21-23 */
uint64_t hnsw_layer_nb_nodes(const hnsw_index *h, uint32_t layer);        /* graph.rs:157-159 */
uint32_t hnsw_layer_m(const hnsw_index *h, uint32_t layer);               /* Graph.m, layers.rs:50 */
/* Graph::iter_nodes (graph.rs:27-29), ascending id; *n receives the node count */
int hnsw_layer_nodes(const hnsw_index *h, uint32_t layer, uint32_t *out, uint64_t cap, uint64_t *n);
/* Graph::neighbors / neighbors_vec / degree (graph.rs:96-113,150-155), ascending id */
int hnsw_neighbors(const hnsw_index *h, uint32_t layer, uint32_t id, uint32_t *buf, uint32_t cap,
                   uint32_t *deg);
/* whole layer as CSR over ascending node ids: node_ids[n_nodes], offsets[n_nodes + 1], nbrs[nnz];
 * pass NULL buffers to query the sizes. */
int hnsw_export_layer(const hnsw_index *h, uint32_t layer, uint32_t *node_ids, uint64_t *offsets,
                      uint32_t *nbrs, uint64_t *n_nodes, uint64_t *nnz);
/* HNSW::assert_param_compliance, template.rs:341-370: *ok = 1 when every degree <= ceil(1.1 *
 * limit) and no node of a multi-node layer is isolated */
int hnsw_check_param_compliance(const hnsw_index *h, int *ok);

/* ---- persistence ---------------------------------------------------------------------------- */
/* HNSW::save / HNSW::load, template.rs:43-131: directory with `points`, `params`, `layers/<n>`,
 * all big-endian, byte-compatible with the reference's Serializer impls (params.rs:78-114,
 * points.rs:124-145, point.rs:57-75, quant.rs:102-124, graph.rs:168-251). */
int hnsw_save(const hnsw_index *h, const char *dir);
int hnsw_load(const char *dir, hnsw_index **out);

/* ---- device management ---------------------------------------------------------------------- */
int hnsw_device_count(int *count);
/* bind the handle to a HIP device (default: the current device at first upload) */
int hnsw_set_device(hnsw_index *h, int device);
/* make the HBM snapshot current now (otherwise done lazily by the first search after a mutation) */
int hnsw_upload(hnsw_index *h);
int hnsw_device_bytes(const hnsw_index *h, uint64_t *bytes);
/* tuning knobs of the HBM snapshot (take effect at the next upload):
 *   "inline_rows"      -1 auto (default) / 0 never / 1 always: the layer-0 "inline rows" layout (a
 *                      copy of every neighbour's vector row next to the adjacency slot, so that one
 *                      expansion is one coalesced read; costs 2m x the row bytes of HBM)
 *   "inline_budget_mb" largest inline-rows allocation the auto mode accepts (default 65536)
 *   "gpu_build"        0 (default): hnsw_insert_bulk(_levels) is the CPU build; 2: it runs the on-device
 *                      build; 1: on-device searches with connect / prune on host threads (also what
 *                      hnsw_insert_bulk_device does while this is 1)
 *   "metric_cosine"    0 (default) / 1.  AN EXTENSION: the reference has Euclidean distance only
 *                      (vectors/src/lib.rs:10-27), so there is nothing to be bit-identical to -- parity unpinned.
 *                      With 1, every row is normalised to unit length as it is inserted and every query as it
 *                      arrives (one left-to-right f32 sum of squares, correctly rounded sqrt and division, the
 *                      same on host and device); behind that everything is the reference's L2 arithmetic, whose
 *                      order on unit vectors is the cosine order.  Distances returned are Euclidean distances
 *                      of the unit vectors (d^2 = 2 - 2 cos); hnsw_get_vector returns the stored unit rows; a
 *                      zero vector is HNSW_ERR_NAN_INPUT.  Set it before the first insert; save / load do not
 *                      carry it (the reference's file format has no such field): set it again after a load.
 *   "gpu_build_batch_max", "gpu_build_batch_div"
 *                      the on-device build inserts min(max, connected / div) points at a time (defaults
 *                      8192 and 8; max is capped at 32768); 256 and 64 stand closer to the reference's
 *                      one-at-a-time insertion (recall@10 + 0.0006 on the bench's index) for ~0.3 s more per
 *                      1M points; 32768 fills the machine better on indexes of tens of millions of points
 *                      (16M x 256d: 17.1 -> 15.4 s, recall unchanged) and when the insertion searches are
 *                      sharded over several GPUs
 *   "coalesce_us", "coalesce_depth", "coalesce_max"
 *                      the gathering of concurrent hnsw_search calls into one launch, see hnsw_search */
int hnsw_set_option(hnsw_index *h, const char *key, int64_t value);
/* counters of the handle: "uploads" (whole-snapshot uploads), "point_patches" (insert_vec calls that patched the
 * live snapshot), "patch_fallbacks" (those that could not: the next search uploads), "coalesced_batches" /
 * "coalesced_queries" / "coalesced_max_batch" (launches made for hnsw_search calls, the calls they answered, the
 * largest batch), "coalesce_ns_window" / "_turn" / "_gpu" / "_handout" (where the batch leaders' time went);
 * the on-device builds of the handle, summed: "build_points", "build_batches", "build_rows_read" (vector rows the
 * insertion searches and the heuristic read: distance evaluations + staged rows), "build_adj_rows" / "build_adj_ids"
 * (adjacency rows read and the ids in them), "build_records" / "build_removals" (edge records filed, reverse edges
 * dropped), "build_insert_kernel_us" (hx_insert_kernel, HIP events), "build_insert_phase_us" / "build_connect_us"
 * (host clock: phase 1 with its copies, sort + connect + remove), "build_connect_kernel_us" (hx_connect_kernel +
 * hx_remove_kernel, HIP events), and for the sharded build "build_rows_owned" /
 * "build_rows_received" (rows this rank changed as their owner and shipped; rows it received from the other owners),
 * "build_exchange_bytes" / "build_exchange_us" (the variable-size all-gathers: bytes received, host clock) */
int hnsw_get_stat(const hnsw_index *h, const char *key, uint64_t *out);

/* ---- replication of the HBM snapshot over the GPUs of a node ----------------------------------- */
/* The search path replicates the index per GPU (the reference keeps one index in
 * one process's RAM, template.rs:35-40).  The snapshot is a handful of flat device arrays plus a small
 * scalar header, so a replica is made by broadcasting them -- the caller's collective (ncclBroadcast over
 * xGMI; torch.distributed in hnsw_rs_amd/hnsw.py), the library only names the buffers:
 *   source rank:       hnsw_snapshot_describe(h, &d)   uploads if needed; d.bytes / d.ptr / d.header
 *   every other rank:  hnsw_create(same m, ef_cons, dim, kind), receive d.bytes and d.header,
 *                      hnsw_snapshot_adopt(h, &d)      allocates the arrays on the handle's device, fills d.ptr
 *                      <broadcast every array into d.ptr[i]>
 *                      hnsw_snapshot_commit(h)         the handle now answers searches
 * A handle made this way is a DEVICE-ONLY replica: it serves the search, brute-force and test-seam entry
 * points; it holds no host copy, so insert_*, save and the per-point / per-layer accessors fail with
 * HNSW_ERR_ARG (hnsw_len, hnsw_layer_count and hnsw_get_params answer from the header). */
#define HNSW_SNAPSHOT_ARRAYS 7
typedef struct hnsw_snapshot_desc {
    uint64_t bytes[HNSW_SNAPSHOT_ARRAYS]; /* size of each array, 0 = absent */
    void *ptr[HNSW_SNAPSHOT_ARRAYS];      /* device pointers on this handle's device */
    uint32_t header[32];                  /* scalar part, opaque: broadcast it verbatim */
} hnsw_snapshot_desc;
int hnsw_snapshot_describe(hnsw_index *h, hnsw_snapshot_desc *out);
int hnsw_snapshot_adopt(hnsw_index *h, hnsw_snapshot_desc *inout);
int hnsw_snapshot_commit(hnsw_index *h);

/* ---- harness helpers (not part of the reference's API) --------------------------------------- */
/* Synthetic "GloVe-shaped" data, counter-based so any row can be generated independently:
 * recipe 0 = low intrinsic dimension clusters (A), 1 = isotropic mixture (B), 2 = U[0,1)
 * (the reference's make_rand_vectors, template.rs:630-638).  out is n x d. */
int hnsw_synth_rows(int recipe, uint64_t seed, uint64_t first_row, uint64_t n, uint32_t d, float *out,
                    uint32_t nb_threads);
/* the level sampler used when levels are not given: believed-equivalent restatement of rand
 * 0.8.5 StdRng::seed_from_u64(0) -> gen::<f32>() -> floor(-ln(r) * ml) (points.rs:39-48,148-160) */
int hnsw_draw_levels(uint32_t m, uint64_t n, uint8_t *out);
/* The reference's call pattern as a load: `threads` host threads, each blocked in its own hnsw_search call
 * (ann_by_vector(&self, ...), template.rs:306-335, one query per call).  Thread t answers queries t, t + T, ... of
 * Q (nq x dim) again and again until `seconds` have passed and every query has been answered at least once.
 * ids (nq x n) / counts (nq, may be NULL) receive each query's last answer, *calls the number of completed calls,
 * *wall_s the elapsed time, lat_us[7] = {p50, p90, p99, max, mean} of the per-call latency in microseconds, then the
 * process's user and system CPU seconds over the run. */
int hnsw_bench_search_threads(hnsw_index *h, const float *Q, uint64_t nq, uint32_t n, uint32_t ef, uint32_t threads,
                              double seconds, uint32_t *ids, uint32_t *counts, uint64_t *calls, double *wall_s,
                              double *lat_us);

/* `callers` host threads, each calling hnsw_search_batch(nq queries, host pointers) `calls` times on its own slice
 * of Q (total x dim) into its own result buffers; *wall_s = first call to last return (two untimed calls per caller
 * first).  What concurrent batch callers of the C ABI see. */
int hnsw_bench_batch_threads(hnsw_index *h, const float *Q, uint64_t total, uint64_t nq, uint32_t n, uint32_t ef,
                             uint32_t callers, uint32_t calls, double *wall_s);

#ifdef __cplusplus
}
#endif
#endif
