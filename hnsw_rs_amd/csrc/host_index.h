// host_index.h -- host side of libhnsw_mi355x: the index a caller mutates through the C ABI
// (points + layered graph + params) and the build path (insert_bulk / insert_vec).
//
// The search hot path does NOT live here: it runs on the GPU from an HBM-resident snapshot of
// this structure (device_index.h, search_kernels.hip).  Citations are relative to the reference
// repository root.
#pragma once

#include <atomic>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/hnsw_mi355x.h"

namespace hx {

typedef uint32_t NodeID;  // graph/src/lib.rs:1

// graph/src/dist.rs:4-38 -- total order: dist, then id
struct Dist {
    NodeID id;
    float dist;
};
inline bool dist_lt(const Dist &a, const Dist &b) {
    return a.dist < b.dist || (a.dist == b.dist && a.id < b.id);
}
inline bool dist_eq(const Dist &a, const Dist &b) { return a.dist == b.dist && a.id == b.id; }

// hnsw/src/params.rs:5-13
struct Params {
    NodeID ep = 0;
    uint64_t m = 0, mmax = 0, mmax0 = 0;
    float ml = 0.0f;
    uint64_t ef_cons = 0, dim = 0;
};

// ---- arithmetic (vectors crate), host side: used by the build path and the accessors ---------
int quantize(const float *v, uint32_t d, float *min_out, float *delta_out, uint8_t *codes);
float dist_quant(uint32_t d, const uint8_t *cx, float delta_x, float min_x, const uint8_t *cy,
                 float delta_y, float min_y);
float dist_full(uint32_t d, const float *x, const float *y);
inline float default_ml(uint64_t m) { return 1.0f / std::log((float)m); }  // params.rs:15-17

// a borrowed Point (points/src/point.rs:6-10)
struct PointView {
    NodeID id = 0;
    const uint8_t *codes = nullptr;
    float delta = 0.0f, min = 0.0f;
    const float *vals = nullptr;
};

class Inserter;

// std::vector storage that is NOT zero-filled by resize(): the stored rows of a bulk insert are first touched
// by the threads that copy them in (a 51-GB row table is otherwise one thread's memset before the copy)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <class U>
    NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U, class... A>
    void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0)
            ::new ((void *)p) U;
        else
            ::new ((void *)p) U(std::forward<A>(a)...);
    }
};

class HostIndex {
  public:
    HostIndex(uint32_t m, uint32_t ef_cons, uint32_t dim, int vec_kind);
    HostIndex(const HostIndex &o);

    // ---- points (points/src/points.rs SimplePoints, stored SoA) ----
    int kind;
    uint32_t dim;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> codes;  // QUANT8: N x dim
    std::vector<float> mins, deltas;                   // QUANT8: N
    std::vector<float, NoInitAlloc<float>> vals;       // F32: N x dim
    std::vector<uint8_t> levels;    // N
    uint64_t len() const { return levels.size(); }
    bool get_point(NodeID id, PointView *p) const;
    float dist2other(const PointView &a, const PointView &b) const;
    void dist2many(const PointView &a, const NodeID *ids, size_t n, float *out) const;

    // ---- layered graph (graph/src/layers.rs, graph/src/graph.rs) ----
    // A node of level L belongs to layers 0..L (layers.rs:63-70).  Layer 0 rows are indexed by
    // id; a node's upper-layer rows are contiguous: row(id, l >= 1) = upper_base[id] + l - 1.
    Params params;
    std::vector<std::vector<NodeID>> layer_nodes;  // per layer, ascending id
    std::vector<std::vector<NodeID>> adj0;
    std::vector<uint32_t> upper_base;
    std::vector<std::vector<NodeID>> adj_up;
    uint64_t version = 0;  // bumped by every mutation; the device snapshot records what it saw

    uint32_t nb_layers() const { return (uint32_t)layer_nodes.size(); }
    uint64_t layer_m(uint32_t layer) const { return layer == 0 ? params.m * 2 : params.m; }
    bool in_layer(uint32_t layer, NodeID id) const {
        return id < len() && layer < nb_layers() && levels[id] >= layer;
    }
    std::vector<NodeID> &row(uint32_t layer, NodeID id) {
        return layer == 0 ? adj0[id] : adj_up[upper_base[id] + layer - 1];
    }
    const std::vector<NodeID> &row(uint32_t layer, NodeID id) const {
        return layer == 0 ? adj0[id] : adj_up[upper_base[id] + layer - 1];
    }
    // neighbours copied under the row's lock (graph.rs:103-113 locks the Mutex, copies the set)
    bool neighbors_vec(uint32_t layer, NodeID id, std::vector<NodeID> *out) const;
    bool degree(uint32_t layer, NodeID id, size_t *out) const;
    int add_edge(uint32_t layer, NodeID a, NodeID b);
    int remove_edge(uint32_t layer, NodeID a, NodeID b);
    int isolate_node(uint32_t layer, NodeID node);
    int replace_neighbors(uint32_t layer, NodeID node, const std::vector<NodeID> &nb);

    // ---- build (hnsw/src/template.rs) ----
    // reserve_rows: give every new layer-0 row its capacity up front (the CPU build appends under a lock).  Tens of
    // millions of small allocations -- the heap growing by 4-KiB pages -- were most of what storing the points cost
    // (16M points: 2.1 of 2.5 s): the on-device build passes false and calls reserve_layer0_rows on other threads
    // while the GPU runs its batches (nobody touches the host graph then), before it reads the graph back
    int store_points(const float *rows, uint64_t n, const uint8_t *levels_in,
                     std::vector<NodeID> *ids_out, uint32_t nb_threads = 1, bool reserve_rows = true);
    void reserve_layer0_rows(NodeID first, uint64_t n, uint32_t nb_threads);
    int insert(NodeID point_id, Inserter &ins);
    // second half of insert (template.rs:185-187): make_connections, prune_connections,
    // make_pruned_connections for the results held by `ins`
    int apply_insertion_results(Inserter &ins);
    // Connect one point from neighbour lists computed elsewhere (the on-device build): nbrs[l] holds
    // the heuristic's selection for layer l (Dist of each neighbour to the point).  Rows touched are
    // appended to `dirty` (layer << 32 | id) when it is not null.
    int connect_point(NodeID point_id, const std::vector<std::vector<Dist>> &nbrs,
                      std::vector<uint64_t> *dirty = nullptr, class DirtyStamps *stamps = nullptr);
    void prepare_build() { ensure_locks(); }
    int insert_bulk(const float *rows, uint64_t n, uint32_t nb_threads, bool verbose,
                    const uint8_t *levels_in);
    int insert_vec(const float *v, int level, NodeID *out_id);
    int import_points(const float *rows, uint64_t n, const uint8_t *levels_in);
    int import_layer(uint32_t layer, uint64_t n_nodes, const NodeID *node_ids,
                     const uint64_t *offsets, const NodeID *nbrs);
    bool check_param_compliance() const;

    // level draws when the caller gives none
    void draw_levels(uint64_t n, uint8_t *out) const;

  private:
    void lock_row(uint32_t layer, NodeID id) const;
    void unlock_row(uint32_t layer, NodeID id) const;
    void ensure_locks();
    mutable std::unique_ptr<std::atomic<uint8_t>[]> lock0_, lock_up_;
    mutable size_t lock0_n_ = 0, lock_up_n_ = 0;
    std::shared_ptr<Inserter> single_ins_;  // insert_vec's Inserter (never shared between indexes: see the copy constructor)
};

// opaque Inserter handles for code outside host_index.cpp
Inserter *new_inserter(uint64_t n_points);
void free_inserter(Inserter *);
// routes the rows touched by this thread's add_edge / remove_edge into `dirty` while in scope
class DirtyStamps;
struct DirtyScope {
    explicit DirtyScope(std::vector<uint64_t> *dirty, DirtyStamps *stamps = nullptr);
    ~DirtyScope();
};

// one per on-device build (owned by that build, reached by its threads through DirtyScope): a stamp per
// adjacency row so that a row is reported dirty once per batch
class DirtyStamps {
  public:
    DirtyStamps(size_t n_rows0, size_t n_rows_up);
    ~DirtyStamps();
    void next_batch();
    bool seen(uint32_t layer, size_t row);  // true when the row was already reported in this batch

  private:
    std::unique_ptr<std::atomic<uint32_t>[]> s0_, sup_;
    size_t n0_, nup_;
    uint32_t epoch_ = 1;
};

// persistence (template.rs:43-131)
int save_index(const HostIndex &idx, const std::string &dir);
int load_index(const std::string &dir, std::unique_ptr<HostIndex> *out);

// rand 0.8.5 StdRng (ChaCha12) level sampler, believed-equivalent restatement
void stdrng_levels(uint64_t seed, float ml, uint64_t n, uint8_t *out);

// synthetic data
int synth_rows(int recipe, uint64_t seed, uint64_t first_row, uint64_t n, uint32_t d, float *out,
               uint32_t nb_threads);

// thread-local error text
void set_error(const char *fmt, ...);
const char *get_error();

}  // namespace hx
