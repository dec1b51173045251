// host_index.cpp -- host index + build path of libhnsw_mi355x (see host_index.h).
//
// The build follows the reference's insertion algorithm step for step (citations inline) with
// flat containers instead of BTreeSet / IntSet / IntMap: sorted vectors with set semantics,
// epoch-stamped visited arrays, one spin lock per adjacency row.  With nb_threads == 1 the graph
// it produces is a pure function of (vectors, levels) -- tests/test_host_build.py checks it
// edge for edge against the CPU oracle's literal restatement.
//
// Float semantics: compiled with -ffp-contract=off, never fast-math; every op rounds once, in
// the reference's order.

#include "host_index.h"

#include <sys/mman.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

namespace hx {

// ---------------------------------------------------------------------------------------------
// error text
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// ---------------------------------------------------------------------------------------------
// arithmetic
// ---------------------------------------------------------------------------------------------

// Rust `f32 as u8` (saturating, NaN -> 0)
static inline uint8_t f32_as_u8(float x) {
    if (!(x > 0.0f)) return 0;  // NaN, negatives, zero
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

// QuantVec::new, vectors/src/quant.rs:41-66
int quantize(const float *v, uint32_t d, float *min_out, float *delta_out, uint8_t *codes) {
    if (d == 0) return HNSW_ERR_EMPTY;
    for (uint32_t i = 0; i < d; i++)
        if (std::isnan(v[i])) return HNSW_ERR_NAN_INPUT;  // partial_cmp().unwrap() panics
    float ub = v[0], lb = v[0];
    for (uint32_t i = 1; i < d; i++) {
        if (!(ub > v[i])) ub = v[i];  // max_by: last of equal maxima
        if (lb > v[i]) lb = v[i];     // min_by: first of equal minima
    }
    const float delta = (ub - lb) / 255.0f;  // 2^8 - 1
    for (uint32_t i = 0; i < d; i++) {
        float b = (v[i] - lb) / delta;
        b += 0.5f;
        codes[i] = f32_as_u8(std::floor(b));
    }
    *min_out = lb;
    *delta_out = delta;
    return HNSW_OK;
}

// QuantVec::distance_unrolled, vectors/src/quant.rs:14-37: eight running sums (lane j takes
// elements 8c + j), the d % 8 tail goes to sum 0 afterwards, left fold, sqrt.
float dist_quant(uint32_t d, const uint8_t *cx, float delta_x, float min_x, const uint8_t *cy,
                 float delta_y, float min_y) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t full = d & ~7u;
    for (uint32_t c = 0; c < full; c += 8) {
        for (int j = 0; j < 8; j++) {
            const float x = ((float)cx[c + j] * delta_x) + min_x;
            const float y = ((float)cy[c + j] * delta_y) + min_y;
            const float t = x - y;
            acc[j] += t * t;
        }
    }
    for (uint32_t i = full; i < d; i++) {
        const float x = ((float)cx[i] * delta_x) + min_x;
        const float y = ((float)cy[i] * delta_y) + min_y;
        const float t = x - y;
        acc[0] += t * t;
    }
    float s = 0.0f;
    for (int j = 0; j < 8; j++) s += acc[j];
    return std::sqrt(s);
}

// FullVec::distance, vectors/src/full.rs:23-29: one left-to-right sum
float dist_full(uint32_t d, const float *x, const float *y) {
    float s = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
        const float t = x[i] - y[i];
        s += t * t;
    }
    return std::sqrt(s);
}

// ---------------------------------------------------------------------------------------------
// HostIndex: points and graph
// ---------------------------------------------------------------------------------------------

HostIndex::HostIndex(uint32_t m, uint32_t ef_cons, uint32_t dim_, int vec_kind)
    : kind(vec_kind), dim(dim_) {
    // Params::from_m / from_m_efcons, hnsw/src/params.rs:20-42
    params.ep = 0;
    params.m = m;
    params.mmax = m;
    params.mmax0 = (uint64_t)m * 2;
    params.ml = default_ml(m);
    params.ef_cons = ef_cons ? ef_cons : (uint64_t)m * 2;
    params.dim = dim_;
}

HostIndex::HostIndex(const HostIndex &o)
    : kind(o.kind), dim(o.dim), codes(o.codes), mins(o.mins), deltas(o.deltas), vals(o.vals),
      levels(o.levels), params(o.params), layer_nodes(o.layer_nodes), adj0(o.adj0),
      upper_base(o.upper_base), adj_up(o.adj_up), version(o.version) {}

bool HostIndex::get_point(NodeID id, PointView *p) const {
    if ((uint64_t)id >= len()) return false;
    p->id = id;
    if (kind == HNSW_VEC_QUANT8) {
        p->codes = &codes[(size_t)id * dim];
        p->delta = deltas[id];
        p->min = mins[id];
        p->vals = nullptr;
    } else {
        p->vals = &vals[(size_t)id * dim];
        p->codes = nullptr;
    }
    return true;
}

float HostIndex::dist2other(const PointView &a, const PointView &b) const {
    return kind == HNSW_VEC_QUANT8
               ? dist_quant(dim, a.codes, a.delta, a.min, b.codes, b.delta, b.min)
               : dist_full(dim, a.vals, b.vals);
}

// VecBase::dist2many (vectors/src/lib.rs:17-22): a's distance to each of n stored points.  Every distance is
// its own left-to-right chain exactly as in dist_full (full.rs:23-29); for f32 rows eight of them are
// interleaved, so that the dependent adds of one chain do not wait for each other (a 256-d distance is 256
// serial adds: the sequential inserts of a build spend their time there).
template <int K>
static inline void dist_full_chains(uint32_t dim, const float *x, const float *const *r, float *out) {
    float acc[K];
    for (int k = 0; k < K; k++) acc[k] = 0.0f;
    for (uint32_t e = 0; e < dim; e++) {
        const float xe = x[e];
        for (int k = 0; k < K; k++) {
            const float t = xe - r[k][e];
            acc[k] += t * t;
        }
    }
    for (int k = 0; k < K; k++) out[k] = std::sqrt(acc[k]);
}
void HostIndex::dist2many(const PointView &a, const NodeID *ids, size_t n, float *out) const {
    if (kind != HNSW_VEC_F32) {
        for (size_t i = 0; i < n; i++) {
            PointView b;
            get_point(ids[i], &b);
            out[i] = dist2other(a, b);
        }
        return;
    }
    const float *r[8];
    size_t i = 0;
    auto rows_at = [&](size_t cnt) {
        for (size_t k = 0; k < cnt; k++) r[k] = &vals[(size_t)ids[i + k] * dim];
    };
    for (; i + 8 <= n; i += 8) {
        rows_at(8);
        dist_full_chains<8>(dim, a.vals, r, out + i);
    }
    if (i + 4 <= n) {
        rows_at(4);
        dist_full_chains<4>(dim, a.vals, r, out + i);
        i += 4;
    }
    if (i + 2 <= n) {
        rows_at(2);
        dist_full_chains<2>(dim, a.vals, r, out + i);
        i += 2;
    }
    if (i < n) {
        rows_at(1);
        dist_full_chains<1>(dim, a.vals, r, out + i);
    }
}

void HostIndex::ensure_locks() {
    if (lock0_n_ != adj0.size()) {
        lock0_n_ = adj0.size();
        lock0_.reset(new std::atomic<uint8_t>[lock0_n_ ? lock0_n_ : 1]);
        for (size_t i = 0; i < lock0_n_; i++) lock0_[i].store(0, std::memory_order_relaxed);
    }
    if (lock_up_n_ != adj_up.size()) {
        lock_up_n_ = adj_up.size();
        lock_up_.reset(new std::atomic<uint8_t>[lock_up_n_ ? lock_up_n_ : 1]);
        for (size_t i = 0; i < lock_up_n_; i++) lock_up_[i].store(0, std::memory_order_relaxed);
    }
}

void HostIndex::lock_row(uint32_t layer, NodeID id) const {
    std::atomic<uint8_t> *l;
    if (layer == 0) {
        if (!lock0_ || id >= lock0_n_) return;
        l = &lock0_[id];
    } else {
        const size_t r = (size_t)upper_base[id] + layer - 1;
        if (!lock_up_ || r >= lock_up_n_) return;
        l = &lock_up_[r];
    }
    uint8_t exp = 0;
    while (!l->compare_exchange_weak(exp, 1, std::memory_order_acquire)) {
        exp = 0;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
void HostIndex::unlock_row(uint32_t layer, NodeID id) const {
    if (layer == 0) {
        if (!lock0_ || id >= lock0_n_) return;
        lock0_[id].store(0, std::memory_order_release);
    } else {
        const size_t r = (size_t)upper_base[id] + layer - 1;
        if (!lock_up_ || r >= lock_up_n_) return;
        lock_up_[r].store(0, std::memory_order_release);
    }
}

bool HostIndex::neighbors_vec(uint32_t layer, NodeID id, std::vector<NodeID> *out) const {
    if (!in_layer(layer, id)) return false;
    lock_row(layer, id);
    *out = row(layer, id);
    unlock_row(layer, id);
    return true;
}
bool HostIndex::degree(uint32_t layer, NodeID id, size_t *out) const {
    if (!in_layer(layer, id)) return false;
    lock_row(layer, id);
    *out = row(layer, id).size();
    unlock_row(layer, id);
    return true;
}

// rows touched by the current thread's connect_point (on-device build): layer << 32 | id.  A
// per-row stamp (set with an atomic exchange) keeps every row at most once per batch across threads.
// The stamps belong to ONE build (DirtyStamps, owned by the build that created it); a thread finds the
// stamps of the build it works for through a thread-local pointer set together with its dirty list, so
// two handles building at the same time never see each other's arrays.
static thread_local std::vector<uint64_t> *tl_dirty = nullptr;
static thread_local DirtyStamps *tl_stamps = nullptr;
static inline void mark_dirty_row(uint32_t layer, NodeID id, size_t up_row) {
    if (!tl_dirty) return;
    if (tl_stamps && tl_stamps->seen(layer, layer == 0 ? (size_t)id : up_row)) return;
    tl_dirty->push_back(((uint64_t)layer << 32) | id);
}
#define mark_dirty(layer, id) mark_dirty_row(layer, id, (layer) == 0 ? 0 : (size_t)upper_base[id] + (layer) - 1)

static inline void row_insert(std::vector<NodeID> &r, NodeID x) {
    if (std::find(r.begin(), r.end(), x) == r.end()) r.push_back(x);
}
static inline void row_remove(std::vector<NodeID> &r, NodeID x) {
    auto it = std::find(r.begin(), r.end(), x);
    if (it != r.end()) {
        *it = r.back();
        r.pop_back();
    }
}

// Graph::add_edge, graph/src/graph.rs:37-52 (each endpoint locked on its own, like the reference)
int HostIndex::add_edge(uint32_t layer, NodeID a, NodeID b) {
    if (a == b) {
        set_error("self connection on node %u", a);
        return HNSW_ERR_SELF_CONNECTION;
    }
    if (!in_layer(layer, a) || !in_layer(layer, b)) {
        set_error("node %u not in graph (layer %u)", in_layer(layer, a) ? b : a, layer);
        return HNSW_ERR_NODE_NOT_IN_GRAPH;
    }
    lock_row(layer, a);
    row_insert(row(layer, a), b);
    unlock_row(layer, a);
    lock_row(layer, b);
    row_insert(row(layer, b), a);
    unlock_row(layer, b);
    mark_dirty(layer, a);
    mark_dirty(layer, b);
    return HNSW_OK;
}
// Graph::remove_edge, graph.rs:72-83
int HostIndex::remove_edge(uint32_t layer, NodeID a, NodeID b) {
    if (!in_layer(layer, a) || !in_layer(layer, b)) return HNSW_ERR_NODE_NOT_IN_GRAPH;
    lock_row(layer, a);
    row_remove(row(layer, a), b);
    unlock_row(layer, a);
    lock_row(layer, b);
    row_remove(row(layer, b), a);
    unlock_row(layer, b);
    mark_dirty(layer, a);
    mark_dirty(layer, b);
    return HNSW_OK;
}
// Graph::isolate_node, graph.rs:85-94: edges to degree-1 neighbours survive.  The reference reads
// the neighbour's degree and removes the edge in two unsynchronised steps; with several threads
// two of them can each see degree 2 and together strip a node of its last edges.  Here the check
// and the removal happen under both row locks (taken in id order, the only nested locking in the
// build), which is the same thing on one thread and keeps "min degree > 0" (template.rs:570) true
// on many.
int HostIndex::isolate_node(uint32_t layer, NodeID node) {
    std::vector<NodeID> nb;
    if (!neighbors_vec(layer, node, &nb)) return HNSW_ERR_NODE_NOT_IN_GRAPH;
    for (NodeID n : nb) {
        if (!in_layer(layer, n)) return HNSW_ERR_NODE_NOT_IN_GRAPH;
        const NodeID lo = std::min(node, n), hi = std::max(node, n);
        lock_row(layer, lo);
        if (hi != lo) lock_row(layer, hi);
        if (row(layer, n).size() != 1) {
            row_remove(row(layer, node), n);
            row_remove(row(layer, n), node);
            mark_dirty(layer, node);
            mark_dirty(layer, n);
        }
        if (hi != lo) unlock_row(layer, hi);
        unlock_row(layer, lo);
    }
    return HNSW_OK;
}
// Graph::replace_neighbors, graph.rs:128-138
int HostIndex::replace_neighbors(uint32_t layer, NodeID node, const std::vector<NodeID> &nb) {
    int rc = isolate_node(layer, node);
    if (rc != HNSW_OK) return rc;
    for (NodeID n : nb) {
        rc = add_edge(layer, node, n);
        if (rc != HNSW_OK) return rc;
    }
    return HNSW_OK;
}

// HNSW::assert_param_compliance, template.rs:341-370
bool HostIndex::check_param_compliance() const {
    bool ok = true;
    for (uint32_t l = 0; l < nb_layers(); l++) {
        const uint64_t max_degree = l > 0 ? params.mmax : params.mmax0;
        const size_t lim = (size_t)std::ceil((float)max_degree * 1.1f);
        for (NodeID id : layer_nodes[l]) {
            const size_t deg = row(l, id).size();
            if (deg > lim) ok = false;
            if (deg == 0 && layer_nodes[l].size() > 1) ok = false;
        }
    }
    return ok;
}

void HostIndex::draw_levels(uint64_t n, uint8_t *out) const {
    // SimplePoints::new re-seeds StdRng with 0 at every call (points.rs:40) and store_points
    // passes get_default_ml(m), not params.ml (template.rs:270)
    stdrng_levels(0, default_ml(params.m), n, out);
}

// ---------------------------------------------------------------------------------------------
// Inserter: Results (results.rs) + Searcher (searcher.rs) + Inserter (inserter.rs), flat
// ---------------------------------------------------------------------------------------------

class Inserter {
  public:
    // Results, results.rs:26-33
    std::vector<Dist> selected;  // BTreeSet<Dist>, ascending
    std::vector<Dist> cand;      // BTreeSet<Dist>, ascending from cand_head
    size_t cand_head = 0;
    std::vector<uint32_t> vstamp;  // visited: IntSet<NodeID> as epoch stamps
    uint32_t epoch = 1;
    std::vector<Dist> visited_h;
    struct LayerRes {
        NodeID point;
        std::vector<Dist> nbrs;
    };
    std::map<size_t, LayerRes> insertion_results;                     // one point per layer
    std::map<size_t, std::map<NodeID, std::vector<Dist>>> prune_results;
    // scratch
    std::vector<NodeID> nb, fresh;
    std::vector<float> dists;
    std::vector<Dist> batch, tmp;

    static bool set_insert(std::vector<Dist> &v, size_t head, const Dist &d) {
        auto it = std::lower_bound(v.begin() + head, v.end(), d, dist_lt);
        if (it != v.end() && dist_eq(*it, d)) return false;
        v.insert(it, d);
        return true;
    }
    bool cand_empty() const { return cand_head >= cand.size(); }
    void cand_clear() {
        cand.clear();
        cand_head = 0;
    }
    bool visit(NodeID id) {
        if (vstamp.size() <= id) vstamp.resize((size_t)id + 1, 0);
        if (vstamp[id] == epoch) return false;
        vstamp[id] = epoch;
        return true;
    }
    void visited_clear() {
        if (++epoch == 0) {
            std::fill(vstamp.begin(), vstamp.end(), 0);
            epoch = 1;
        }
    }
    void clear_all() {  // results.rs:182-190
        selected.clear();
        cand_clear();
        visited_clear();
        visited_h.clear();
        insertion_results.clear();
        prune_results.clear();
    }

    // Searcher::search_layer, searcher.rs:23-103
    int search_layer(const HostIndex &idx, uint32_t layer, const PointView &point, size_t ef) {
        for (const Dist &d : selected) set_insert(cand, cand_head, d);  // results.rs:148-157
        for (const Dist &d : selected) visit(d.id);                     // results.rs:159-168
        while (!cand_empty()) {
            const Dist c = cand[cand_head++];  // pop_first
            const Dist f = selected.back();    // selected.last()
            if (dist_lt(f, c)) break;          // cand_dist > furthest2q_dist
            if (!idx.neighbors_vec(layer, c.id, &nb)) {
                set_error("Error in search_layer: %u not in Graph", c.id);
                return HNSW_ERR_NODE_NOT_IN_GRAPH;
            }
            batch.clear();
            fresh.clear();
            for (NodeID n : nb) {
                if (!visit(n)) continue;
                if (n >= idx.len()) {
                    set_error("Could get point %u", n);
                    return HNSW_ERR_ARG;
                }
                fresh.push_back(n);
            }
            dists.resize(fresh.size());
            idx.dist2many(point, fresh.data(), fresh.size(), dists.data());
            for (size_t k = 0; k < fresh.size(); k++) {
                if (std::isnan(dists[k])) {
                    set_error("NaN distance to point %u", fresh[k]);
                    return HNSW_ERR_NAN_INPUT;
                }
                batch.push_back(Dist{fresh[k], dists[k]});
            }
            for (const Dist &e : batch) {
                const Dist f2 = selected.back();
                if (selected.size() < ef) {
                    set_insert(selected, 0, e);
                    set_insert(cand, cand_head, e);
                    continue;
                }
                if (dist_lt(e, f2)) {
                    set_insert(selected, 0, e);
                    set_insert(cand, cand_head, e);
                    if (selected.size() > ef) selected.pop_back();
                }
            }
        }
        cand_clear();     // searcher.rs:100
        visited_clear();  // searcher.rs:101
        return HNSW_OK;
    }

    // Searcher::select_heuristic, searcher.rs:109-153 (extend_cands = keep_pruned = true is the
    // only way the reference calls it, inserter.rs:114-122)
    int select_heuristic(const HostIndex &idx, uint32_t layer, const PointView &point, size_t m) {
        // select_setup, results.rs:105-111
        visited_h.clear();
        cand = selected;
        cand_head = 0;
        selected.clear();
        // extend_candidates_with_neighbors, results.rs:122-146.  The reference evaluates every neighbour of every
        // candidate and lets the set drop the repeats; an id's distance to the point is the same each time, so
        // each id is evaluated once here (the visited stamps are free between two searches): the same set, a
        // third of the distances and of the sort
        tmp.clear();
        for (const Dist &c0 : cand) visit(c0.id);
        for (const Dist &node : cand) {
            if (!idx.neighbors_vec(layer, node.id, &nb)) {
                set_error("Node %u is not in the Graph", node.id);
                return HNSW_ERR_NODE_NOT_IN_GRAPH;
            }
            fresh.clear();
            for (NodeID n : nb) {
                if (n >= idx.len()) return HNSW_ERR_ARG;
                if (visit(n)) fresh.push_back(n);
            }
            dists.resize(fresh.size());
            idx.dist2many(point, fresh.data(), fresh.size(), dists.data());  // points.distance(point.id, neighbor)
            for (size_t k = 0; k < fresh.size(); k++) {
                if (std::isnan(dists[k])) return HNSW_ERR_NAN_INPUT;
                tmp.push_back(Dist{fresh[k], dists[k]});
            }
        }
        visited_clear();
        cand.insert(cand.end(), tmp.begin(), tmp.end());
        std::sort(cand.begin(), cand.end(), dist_lt);  // (ids are distinct now: BTreeSet order)
        if (cand_empty()) return HNSW_ERR_EMPTY;
        selected.push_back(cand[cand_head++]);
        while (!cand_empty() && selected.size() < m) {
            const Dist e = cand[cand_head++];
            PointView ep;
            if (!idx.get_point(e.id, &ep)) return HNSW_ERR_ARG;
            // get_nearest_from_selected, results.rs:69-77, and `e < nearest` (searcher.rs:128-139): e is kept
            // iff it is below Dist(s, d(e, s)) for EVERY selected s, so the scan stops at the first s that is
            // not above it (most candidates are dropped by one of the nearest few); four distances at a time
            bool keep = true;
            fresh.clear();
            for (const Dist &s : selected) fresh.push_back(s.id);
            for (size_t k0 = 0; k0 < fresh.size() && keep; k0 += 4) {
                const size_t cnt = std::min<size_t>(4, fresh.size() - k0);
                float d4[4];
                idx.dist2many(ep, fresh.data() + k0, cnt, d4);
                for (size_t k = 0; k < cnt; k++)
                    if (!dist_lt(e, Dist{fresh[k0 + k], d4[k]})) {
                        keep = false;
                        break;
                    }
            }
            if (keep) {
                set_insert(selected, 0, e);
            } else {
                set_insert(visited_h, 0, e);
            }
        }
        size_t vh = 0;
        while (vh < visited_h.size() && selected.size() < m) set_insert(selected, 0, visited_h[vh++]);
        visited_h.erase(visited_h.begin(), visited_h.begin() + vh);
        // the un-popped candidates stay where they are: the next layer's search_layer starts from
        // them too
        return HNSW_OK;
    }

    // Inserter::build_insertion_results, inserter.rs:40-126
    int build_insertion_results(const HostIndex &idx, const PointView &point, uint32_t level) {
        if (point.id == idx.params.ep) return HNSW_OK;  // inserter.rs:42-45: results stay as they were
        clear_all();                                     // setup_insert, inserter.rs:53-68
        PointView ep;
        if (!idx.get_point(idx.params.ep, &ep)) return HNSW_ERR_ARG;
        const float d0 = idx.dist2other(ep, point);
        if (std::isnan(d0)) return HNSW_ERR_NAN_INPUT;
        selected.push_back(Dist{idx.params.ep, d0});
        const uint32_t layers_len = idx.nb_layers();
        for (uint32_t l = layers_len; l-- > level + 1;) {  // traverse_layers_above
            int rc = search_layer(idx, l, point, 1);
            if (rc != HNSW_OK) return rc;
        }
        const uint32_t bound = std::min(level, layers_len - 1);  // traverse_layers_below
        for (uint32_t l = bound + 1; l-- > 0;) {
            int rc = search_layer(idx, l, point, idx.params.ef_cons);
            if (rc != HNSW_OK) return rc;
            rc = select_heuristic(idx, l, point, idx.params.m);
            if (rc != HNSW_OK) return rc;
            insertion_results[l] = LayerRes{point.id, selected};  // save_layer_results
        }
        return HNSW_OK;
    }
};

// HNSW::insert, template.rs:177-190
int HostIndex::insert(NodeID point_id, Inserter &ins) {
    PointView point;
    if (!get_point(point_id, &point)) {
        set_error("Point %u not found in collection.", point_id);
        return HNSW_ERR_ARG;
    }
    int rc = ins.build_insertion_results(*this, point, levels[point_id]);
    if (rc != HNSW_OK) return rc;
    return apply_insertion_results(ins);
}

int HostIndex::apply_insertion_results(Inserter &ins) {
    int rc = HNSW_OK;
    // make_connections, template.rs:196-207
    for (auto &lr : ins.insertion_results) {
        for (const Dist &n : lr.second.nbrs) {
            rc = add_edge((uint32_t)lr.first, lr.second.point, n.id);
            if (rc != HNSW_OK) return rc;
        }
    }
    // prune_connections, template.rs:209-238: lists are computed from one snapshot ...
    ins.prune_results.clear();
    for (auto &lr : ins.insertion_results) {
        const uint32_t l = (uint32_t)lr.first;
        const uint64_t lm = layer_m(l);
        for (const Dist &x : lr.second.nbrs) {
            size_t deg;
            if (!degree(l, x.id, &deg)) return HNSW_ERR_NODE_NOT_IN_GRAPH;
            if (!(deg > lm)) continue;
            if (!neighbors_vec(l, x.id, &ins.nb)) return HNSW_ERR_NODE_NOT_IN_GRAPH;
            PointView a;
            get_point(x.id, &a);
            std::vector<Dist> ds;
            ds.reserve(ins.nb.size());
            for (NodeID n : ins.nb)
                if (n >= len()) return HNSW_ERR_ARG;
            ins.dists.resize(ins.nb.size());
            dist2many(a, ins.nb.data(), ins.nb.size(), ins.dists.data());
            for (size_t k = 0; k < ins.nb.size(); k++) ds.push_back(Dist{ins.nb[k], ins.dists[k]});
            std::sort(ds.begin(), ds.end(), dist_lt);  // select_simple, template.rs:614-621
            if (ds.size() > lm) ds.resize(lm);
            ins.prune_results[l][x.id] = std::move(ds);
        }
    }
    // ... make_pruned_connections, template.rs:240-251: and applied one node at a time
    for (auto &lr : ins.prune_results) {
        for (auto &nd : lr.second) {
            std::vector<NodeID> ids;
            ids.reserve(nd.second.size());
            for (const Dist &n : nd.second) ids.push_back(n.id);
            rc = replace_neighbors((uint32_t)lr.first, nd.first, ids);
            if (rc != HNSW_OK) return rc;
        }
    }
    return HNSW_OK;
}

Inserter *new_inserter(uint64_t n_points) {
    Inserter *ins = new Inserter();
    ins->vstamp.assign(n_points, 0);
    return ins;
}
void free_inserter(Inserter *ins) { delete ins; }
DirtyScope::DirtyScope(std::vector<uint64_t> *dirty, DirtyStamps *stamps) {
    tl_dirty = dirty;
    tl_stamps = stamps;
}
DirtyScope::~DirtyScope() {
    tl_dirty = nullptr;
    tl_stamps = nullptr;
}

DirtyStamps::DirtyStamps(size_t n0, size_t n_up) : n0_(n0), nup_(n_up) {
    s0_.reset(new std::atomic<uint32_t>[n0 ? n0 : 1]);
    sup_.reset(new std::atomic<uint32_t>[n_up ? n_up : 1]);
    for (size_t i = 0; i < n0; i++) s0_[i].store(0, std::memory_order_relaxed);
    for (size_t i = 0; i < n_up; i++) sup_[i].store(0, std::memory_order_relaxed);
}
DirtyStamps::~DirtyStamps() {}
void DirtyStamps::next_batch() { epoch_++; }
bool DirtyStamps::seen(uint32_t layer, size_t row) {
    if (row >= (layer == 0 ? n0_ : nup_)) return false;  // a row added after the build began: always reported
    std::atomic<uint32_t> &st = layer == 0 ? s0_[row] : sup_[row];
    return st.exchange(epoch_, std::memory_order_relaxed) == epoch_;
}

int HostIndex::connect_point(NodeID point_id, const std::vector<std::vector<Dist>> &nbrs,
                             std::vector<uint64_t> *dirty, DirtyStamps *stamps) {
    if (point_id >= len()) return HNSW_ERR_ARG;
    Inserter ins;
    for (size_t l = 0; l < nbrs.size(); l++) {
        if (nbrs[l].empty()) continue;
        Inserter::LayerRes lr{point_id, nbrs[l]};
        std::sort(lr.nbrs.begin(), lr.nbrs.end(), dist_lt);  // BTreeSet<Dist> order
        ins.insertion_results[l] = std::move(lr);
    }
    DirtyScope scope(dirty, stamps);
    return apply_insertion_results(ins);
}

// HNSW::store_points, template.rs:269-293
// Fresh storage for a bulk insert is first touched by the copying threads: with 2-MiB pages (where the kernel
// grants them on madvise) that is one fault per 2 MiB instead of one per 4 KiB.  Advice only.
static void advise_huge_pages(void *p, size_t nbytes) {
    const uintptr_t H = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)p + H - 1) & ~(H - 1), e = ((uintptr_t)p + nbytes) & ~(H - 1);
    if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
}

int HostIndex::store_points(const float *rows, uint64_t n, const uint8_t *levels_in,
                            std::vector<NodeID> *ids_out, uint32_t nb_threads, bool reserve_rows) {
    if (n == 0) {
        set_error("no vectors given");
        return HNSW_ERR_EMPTY;
    }
    if (len() + n > (uint64_t)0x7FFFFFFF) {
        set_error("index would exceed 2^31 - 1 points");
        return HNSW_ERR_ARG;
    }
    std::vector<uint8_t> drawn;
    if (!levels_in) {
        drawn.resize(n);
        draw_levels(n, drawn.data());
        levels_in = drawn.data();
    }
    // rows are independent: quantise / copy them on nb_threads threads straight into the stored arrays (each
    // thread first-touches its own part), report the first bad row and put the arrays back as they were
    const NodeID first = (NodeID)len();
    const size_t old_codes = codes.size(), old_vals = vals.size(), old_n = mins.size();
    std::atomic<uint64_t> bad{UINT64_MAX};
    auto mark_bad = [&](uint64_t i) {
        uint64_t cur = bad.load();
        while (i < cur && !bad.compare_exchange_weak(cur, i)) {
        }
    };
    auto run = [&](auto &&work) {
        const unsigned nt = (unsigned)std::min<uint64_t>(std::max(1u, nb_threads), std::max<uint64_t>(1, n / 4096));
        if (nt <= 1) {
            work(0, n);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
            for (auto &t : th) t.join();
        }
    };
    if (kind == HNSW_VEC_QUANT8) {
        codes.resize(old_codes + (size_t)n * dim);
        advise_huge_pages(codes.data() + old_codes, (size_t)n * dim);
        mins.resize(old_n + n);
        deltas.resize(old_n + n);
        run([&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi && i < bad.load(std::memory_order_relaxed); i++) {  // (rows below a bad one are still checked: the FIRST bad row is reported)
                int rc = quantize(rows + i * dim, dim, &mins[old_n + i], &deltas[old_n + i],
                                  &codes[old_codes + (size_t)i * dim]);
                if (rc == HNSW_OK && !std::isfinite(deltas[old_n + i])) rc = HNSW_ERR_NAN_INPUT;
                if (rc != HNSW_OK) {
                    mark_bad(i);
                    return;
                }
            }
        });
    } else {
        vals.resize(old_vals + (size_t)n * dim);
        advise_huge_pages(vals.data() + old_vals, (size_t)n * dim * sizeof(float));
        run([&](uint64_t lo, uint64_t hi) {
            // blocks of rows small enough to stay in cache between the NaN scan and the copy
            const uint64_t step = std::max<uint64_t>(1, 16384 / dim);
            for (uint64_t i = lo; i < hi; i += step) {
                const uint64_t e = std::min(hi, i + step);
                const float *src = rows + i * dim;
                const size_t cnt = (size_t)(e - i) * dim;
                bool nan = false;
                for (size_t k = 0; k < cnt; k++) nan |= src[k] != src[k];
                if (nan) {
                    for (size_t k = 0; k < cnt; k++)
                        if (src[k] != src[k]) {
                            mark_bad(i + k / dim);
                            break;
                        }
                    return;
                }
                memcpy(&vals[old_vals + (size_t)i * dim], src, cnt * sizeof(float));
            }
        });
    }
    if (bad.load() != UINT64_MAX) {
        codes.resize(old_codes);
        mins.resize(old_n);
        deltas.resize(old_n);
        vals.resize(old_vals);
        if (kind == HNSW_VEC_QUANT8) {
            set_error("row %llu: NaN / non-finite range cannot be quantised", (unsigned long long)bad.load());
        } else {
            set_error("row %llu contains NaN", (unsigned long long)bad.load());
        }
        return HNSW_ERR_NAN_INPUT;
    }
    levels.insert(levels.end(), levels_in, levels_in + n);
    // graph rows of the new nodes (Layers::add_node + add_level, graph/src/layers.rs:48-70): the per-node bookkeeping is serial (ids arrive ascending), the allocation
    // of the layer-0 rows -- one small block per node -- runs on the threads
    const size_t total = len();
    adj0.resize(total);
    upper_base.resize(total, UINT32_MAX);
    for (uint64_t i = 0; i < n; i++) {
        const NodeID id = first + (NodeID)i;
        const uint32_t level = levels[id];
        while (layer_nodes.size() <= level) layer_nodes.emplace_back();
        if (level >= 1 && upper_base[id] == UINT32_MAX) {
            upper_base[id] = (uint32_t)adj_up.size();
            adj_up.resize(adj_up.size() + level);
        }
        for (uint32_t l = 0; l <= level; l++) layer_nodes[l].push_back(id);
    }
    if (reserve_rows) reserve_layer0_rows(first, n, nb_threads);
    // template.rs:283-290: ep = first key of the top layer (hash order there; smallest id here)
    params.ep = layer_nodes.back().front();
    if (ids_out) {
        ids_out->resize(n);
        for (uint64_t i = 0; i < n; i++) (*ids_out)[i] = first + (NodeID)i;
    }
    version++;
    return HNSW_OK;
}

void HostIndex::reserve_layer0_rows(NodeID first, uint64_t n, uint32_t nb_threads) {
    const size_t cap = layer_m(0) + 2;
    const unsigned nt = (unsigned)std::min<uint64_t>(std::max(1u, nb_threads), std::max<uint64_t>(1, n / 4096));
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) adj0[first + i].reserve(cap);
    };
    if (nt <= 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
}

// HNSW::insert_bulk, template.rs:388-444
int HostIndex::insert_bulk(const float *rows, uint64_t n, uint32_t nb_threads, bool verbose,
                           const uint8_t *levels_in) {
    if (nb_threads == 0) nb_threads = 1;
    std::vector<NodeID> ids;
    int rc = store_points(rows, n, levels_in, &ids, nb_threads);
    if (rc != HNSW_OK) return rc;
    const NodeID first = ids.front();  // stored ids are contiguous
    ensure_locks();
    std::atomic<uint64_t> done{0};
    const uint64_t total = n;
    std::mutex err_mu;
    int first_rc = HNSW_OK;
    std::string first_msg;
    for (uint32_t layer_nb = nb_layers(); layer_nb-- > 0;) {
        // template.rs:406-416
        const size_t nb_nodes = layer_nodes[layer_nb].size();
        const size_t chunk = (nb_nodes + nb_threads - 1) / nb_threads;
        std::vector<NodeID> lids;
        for (NodeID id : layer_nodes[layer_nb])
            if (id >= first && levels[id] == (uint8_t)layer_nb) lids.push_back(id);
        if (lids.empty()) continue;
        auto work = [&](size_t lo, size_t hi) {
            Inserter ins;  // one fresh Inserter per chunk, template.rs:427
            ins.vstamp.assign(len(), 0);
            for (size_t i = lo; i < hi; i++) {
                int r = insert(lids[i], ins);
                if (r != HNSW_OK) {
                    std::lock_guard<std::mutex> g(err_mu);
                    if (first_rc == HNSW_OK) {
                        first_rc = r;
                        first_msg = get_error();
                    }
                    return;
                }
                const uint64_t k = done.fetch_add(1) + 1;
                if (verbose && (k % ((total / 20) + 1) == 0 || k == total))
                    fprintf(stderr, "\rBuilding HNSW index %llu/%llu", (unsigned long long)k,
                            (unsigned long long)total);
            }
        };
        if (nb_threads == 1 || lids.size() <= chunk) {
            work(0, lids.size());
        } else {
            std::vector<std::thread> th;
            for (size_t lo = 0; lo < lids.size(); lo += chunk)
                th.emplace_back(work, lo, std::min(lids.size(), lo + chunk));
            for (auto &t : th) t.join();
        }
        if (first_rc != HNSW_OK) break;
    }
    if (verbose) fprintf(stderr, "\n");
    version++;
    if (first_rc != HNSW_OK) {
        set_error("%s", first_msg.c_str());
        return first_rc;
    }
    return HNSW_OK;
}

// HNSW::insert_vec, template.rs:165-173.  The reference makes a fresh Inserter per call (template.rs:171); its
// state is cleared at the start of every insertion (inserter.rs:53-68), so ONE kept on the index behaves the same
// and an insertion does not pay an O(N) allocation of visited stamps (nor a new lock array: nothing else may
// touch the index during an insert_vec, the rows' locks are only for insert_bulk's threads).
int HostIndex::insert_vec(const float *v, int level, NodeID *out_id) {
    std::vector<NodeID> ids;
    uint8_t lv = (uint8_t)level;
    int rc = store_points(v, 1, level < 0 ? nullptr : &lv, &ids);
    if (rc != HNSW_OK) return rc;
    if (!single_ins_) single_ins_.reset(new_inserter(0), free_inserter);
    single_ins_->clear_all();  // a fresh Inserter holds no results (they would be re-applied if the point is the new entry point, inserter.rs:42-45)
    rc = insert(ids[0], *single_ins_);
    version++;
    if (rc != HNSW_OK) return rc;
    if (out_id) *out_id = ids[0];
    return HNSW_OK;
}

int HostIndex::import_points(const float *rows, uint64_t n, const uint8_t *levels_in) {
    std::vector<uint8_t> zeros;
    if (!levels_in) {
        zeros.assign(n, 0);
        levels_in = zeros.data();
    }
    return store_points(rows, n, levels_in, nullptr);
}

int HostIndex::import_layer(uint32_t layer, uint64_t n_nodes, const NodeID *node_ids,
                            const uint64_t *offsets, const NodeID *nbrs) {
    if (layer >= nb_layers() || n_nodes != layer_nodes[layer].size()) {
        set_error("import_layer %u: %llu nodes given, the points' levels imply %llu", layer,
                  (unsigned long long)n_nodes,
                  (unsigned long long)(layer < nb_layers() ? layer_nodes[layer].size() : 0));
        return HNSW_ERR_ARG;
    }
    for (uint64_t i = 0; i < n_nodes; i++) {
        const NodeID id = node_ids[i];
        if (!in_layer(layer, id)) {
            set_error("import_layer %u: node %u has a lower level", layer, id);
            return HNSW_ERR_NODE_NOT_IN_GRAPH;
        }
        std::vector<NodeID> &r = row(layer, id);
        r.clear();
        for (uint64_t k = offsets[i]; k < offsets[i + 1]; k++) {
            if (!in_layer(layer, nbrs[k])) {
                set_error("import_layer %u: neighbour %u of %u not in layer", layer, nbrs[k], id);
                return HNSW_ERR_NODE_NOT_IN_GRAPH;
            }
            row_insert(r, nbrs[k]);
        }
    }
    version++;
    return HNSW_OK;
}

// ---------------------------------------------------------------------------------------------
// rand 0.8.5 StdRng = ChaCha12 (rand_chacha 0.3.1), seed_from_u64 via PCG32 (rand_core 0.6).
// rand is not part of the reference tree, so this restatement is unverifiable offline: it is the
// DEFAULT level source only; reproducible runs pass levels explicitly.
// ---------------------------------------------------------------------------------------------
namespace {
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline void qr(uint32_t *s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl32(s[d], 16);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl32(s[b], 12);
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl32(s[d], 8);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl32(s[b], 7);
}
struct ChaCha12 {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    explicit ChaCha12(uint64_t seed) {
        // SeedableRng::seed_from_u64: PCG32 fills the 32-byte seed, 4 bytes (LE) per step
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
        uint64_t state = seed;
        for (int i = 0; i < 8; i++) {
            state = state * MUL + INC;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    void block() {
        uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2],
                          key[3], key[4], key[5], key[6], key[7], (uint32_t)counter,
                          (uint32_t)(counter >> 32), 0, 0};
        uint32_t w[16];
        memcpy(w, s, sizeof(w));
        for (int r = 0; r < 6; r++) {  // 12 rounds = 6 double rounds
            qr(w, 0, 4, 8, 12); qr(w, 1, 5, 9, 13); qr(w, 2, 6, 10, 14); qr(w, 3, 7, 11, 15);
            qr(w, 0, 5, 10, 15); qr(w, 1, 6, 11, 12); qr(w, 2, 7, 8, 13); qr(w, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf[i] = w[i] + s[i];
        counter++;
        idx = 0;
    }
    uint32_t next_u32() {
        if (idx >= 16) block();
        return buf[idx++];
    }
    float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }  // Standard: [0,1)
};
}  // namespace

void stdrng_levels(uint64_t seed, float ml, uint64_t n, uint8_t *out) {
    ChaCha12 rng(seed);
    for (uint64_t i = 0; i < n; i++) {
        float r = 0.0f;  // new_layer, points/src/points.rs:148-160
        while (r == 0.0f || r == 1.0f) r = rng.gen_f32();
        const float lv = std::floor(-std::log(r) * ml);
        out[i] = (uint8_t)((uint64_t)lv & 0xFF);  // `as usize`, then point.rs:15 `as u8`
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic data: counter-based, so any row is a pure function of
// (recipe, seed, row index) and the store / query sets share one cluster structure.
// ---------------------------------------------------------------------------------------------
namespace {
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
inline uint64_t h3(uint64_t a, uint64_t b, uint64_t c) {
    return splitmix64(splitmix64(splitmix64(a) ^ b) ^ c);
}
inline double u01(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
// standard normal #k of stream (a, b): Box-Muller on two hashed uniforms
inline double normal(uint64_t a, uint64_t b, uint64_t k) {
    const double u1 = u01(h3(a, b, 2 * k)), u2 = u01(h3(a, b, 2 * k + 1));
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}
const uint64_t STRUCT_CENTRES = 0x5EED0A01, STRUCT_PROJ = 0x5EED0004, STRUCT_CENTRES_B = 0x5EED0B01;
}  // namespace

int synth_rows(int recipe, uint64_t seed, uint64_t first_row, uint64_t n, uint32_t d, float *out,
               uint32_t nb_threads) {
    if (recipe < 0 || recipe > 2 || d == 0) return HNSW_ERR_ARG;
    if (nb_threads == 0) nb_threads = 1;
    const int R = 16, KA = 256, KB = 4096;
    std::vector<double> P;
    if (recipe == 0) {
        P.resize((size_t)R * d);
        for (int r = 0; r < R; r++)
            for (uint32_t j = 0; j < d; j++) P[(size_t)r * d + j] = normal(STRUCT_PROJ, r, j) / 4.0;
    }
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t row = first_row + i;
            float *x = out + i * d;
            if (recipe == 0) {
                const uint64_t k = h3(seed, row, 0xC1) % KA;
                double z[R];
                for (int r = 0; r < R; r++)
                    z[r] = 0.60 * normal(STRUCT_CENTRES, k, r) + 0.45 * normal(seed, row, r);
                for (uint32_t j = 0; j < d; j++) {
                    double s = 0.067;
                    for (int r = 0; r < R; r++) s += z[r] * P[(size_t)r * d + j];
                    s += 0.05 * normal(seed, row, R + j);
                    x[j] = (float)s;
                }
            } else if (recipe == 1) {
                const uint64_t k = h3(seed, row, 0xC2) % KB;
                for (uint32_t j = 0; j < d; j++)
                    x[j] = (float)(0.067 + 0.60 * normal(STRUCT_CENTRES_B, k, j) +
                                   0.45 * normal(seed, row, j));
            } else {
                for (uint32_t j = 0; j < d; j++)
                    x[j] = (float)(h3(seed, row, j) >> 40) * (1.0f / 16777216.0f);
            }
        }
    };
    if (nb_threads == 1 || n < 1024) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < nb_threads; t++)
            th.emplace_back(work, n * t / nb_threads, n * (t + 1) / nb_threads);
        for (auto &t : th) t.join();
    }
    return HNSW_OK;
}

}  // namespace hx
