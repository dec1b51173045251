// oracle.cpp -- CPU ORACLE (test infrastructure, not product code; see oracle.h).
//
// A literal restatement of the hot path of Gumo-A/hnsw_rs (and of the build path that is needed
// to obtain a graph at all), written so that each function can be read side by side with the
// Rust it follows.  Containers mirror the reference's: BTreeSet<Dist> -> std::set<Dist>,
// IntSet<NodeID> -> std::unordered_set / a sorted vector used as a set, IntMap -> std::map
// (ascending-key iteration is this oracle's documented stand-in for hashbrown's iteration order,
// SURVEY.md section 8c / Appendix B).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off (never -ffast-math); see oracle/Makefile.
// All citations are relative to /root/reference/.

#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

typedef uint32_t NodeID;  // graph/src/lib.rs:1

// ---------------------------------------------------------------------------------------------
// vectors crate
// ---------------------------------------------------------------------------------------------

// Rust `f32 as u8`: saturating, NaN -> 0.
inline uint8_t f32_as_u8(float x) {
    if (std::isnan(x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;  // truncation toward zero
}

// vectors/src/quant.rs:41-66  QuantVec::new
int quantize(const float *v, uint32_t d, float *min_out, float *delta_out, uint8_t *codes) {
    if (d == 0) return ORC_ERR_EMPTY;  // .max_by(..).unwrap() on an empty iterator panics
    for (uint32_t i = 0; i < d; i++)
        if (std::isnan(v[i])) return ORC_ERR_NAN;  // partial_cmp().unwrap() panics
    // Iterator::max_by keeps the LAST of several equal maxima, min_by the FIRST of equal minima.
    float upper_bound = v[0];
    for (uint32_t i = 1; i < d; i++)
        if (!(upper_bound > v[i])) upper_bound = v[i];
    float lower_bound = v[0];
    for (uint32_t i = 1; i < d; i++)
        if (lower_bound > v[i]) lower_bound = v[i];
    // 2.0f32.powi(BITS) - 1.0 with BITS = 8
    const float levels = 256.0f - 1.0f;
    const float delta = (upper_bound - lower_bound) / levels;
    for (uint32_t i = 0; i < d; i++) {
        float buffer = (v[i] - lower_bound) / delta;
        buffer += 0.5f;
        codes[i] = f32_as_u8(std::floor(buffer));
    }
    *min_out = lower_bound;
    *delta_out = delta;
    return ORC_OK;
}

// vectors/src/quant.rs:14-37  QuantVec::distance_unrolled (CHUNK_SIZE = 8)
float dist_quant(uint32_t d, const uint8_t *cx, float delta_x, float min_x, const uint8_t *cy,
                 float delta_y, float min_y) {
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t nchunks = d / 8;
    for (uint32_t c = 0; c < nchunks; c++) {
        for (uint32_t idx = 0; idx < 8; idx++) {
            const float x_f32 = ((float)cx[8 * c + idx] * delta_x) + min_x;
            const float y_f32 = ((float)cy[8 * c + idx] * delta_y) + min_y;
            const float t = x_f32 - y_f32;
            acc[idx] += t * t;  // powi(2) == one rounded multiply
        }
    }
    for (uint32_t i = 8 * nchunks; i < d; i++) {  // remainder, all into acc[0]
        const float x_f32 = (float)cx[i] * delta_x + min_x;
        const float y_f32 = (float)cy[i] * delta_y + min_y;
        const float t = x_f32 - y_f32;
        acc[0] += t * t;
    }
    float sum = 0.0f;  // acc.iter().sum::<f32>()
    for (int j = 0; j < 8; j++) sum += acc[j];
    return std::sqrt(sum);
}

// vectors/src/full.rs:23-29  FullVec::distance
float dist_full(uint32_t d, const float *x, const float *y) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
        const float t = x[i] - y[i];
        sum += t * t;
    }
    return std::sqrt(sum);
}

// ---------------------------------------------------------------------------------------------
// graph crate
// ---------------------------------------------------------------------------------------------

// graph/src/dist.rs:4-38
struct Dist {
    NodeID id;
    float dist;
};
inline int dist_cmp(const Dist &a, const Dist &b) {
    if (a.dist < b.dist) return -1;
    if (a.dist > b.dist) return 1;
    if (a.dist == b.dist) return a.id < b.id ? -1 : (a.id > b.id ? 1 : 0);
    return -2;  // NaN: partial_cmp().unwrap() panics
}
struct DistLess {
    bool operator()(const Dist &a, const Dist &b) const { return dist_cmp(a, b) == -1; }
};
typedef std::set<Dist, DistLess> OrderedDists;  // BTreeSet<Dist>

// IntSet<NodeID> as a sorted vector with set semantics (iteration = ascending id)
struct IdSet {
    std::vector<NodeID> v;
    bool insert(NodeID x) {
        auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it != v.end() && *it == x) return false;
        v.insert(it, x);
        return true;
    }
    bool remove(NodeID x) {
        auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it == v.end() || *it != x) return false;
        v.erase(it);
        return true;
    }
    size_t len() const { return v.size(); }
};

enum GraphErr { G_OK = 0, G_NODE_NOT_IN_GRAPH, G_SELF_CONNECTION };

// graph/src/graph.rs:9-16
struct Graph {
    std::unordered_map<NodeID, IdSet> nodes;
    size_t level;
    size_t m;

    // graph.rs:31-35
    void add_node(NodeID id) { nodes.emplace(id, IdSet()); }
    // graph.rs:37-52
    GraphErr add_edge(NodeID a, NodeID b, NodeID *bad = nullptr) {
        if (a == b) {
            if (bad) *bad = a;
            return G_SELF_CONNECTION;
        }
        auto ia = nodes.find(a), ib = nodes.find(b);
        if (ia == nodes.end() || ib == nodes.end()) {  // graph.rs:54-70
            if (bad) *bad = (ia != nodes.end()) ? b : a;
            return G_NODE_NOT_IN_GRAPH;
        }
        ia->second.insert(b);
        ib->second.insert(a);
        return G_OK;
    }
    // graph.rs:72-83
    GraphErr remove_edge(NodeID a, NodeID b) {
        auto ia = nodes.find(a), ib = nodes.find(b);
        if (ia == nodes.end() || ib == nodes.end()) return G_NODE_NOT_IN_GRAPH;
        ia->second.remove(b);
        ib->second.remove(a);
        return G_OK;
    }
    // graph.rs:150-155
    bool degree(NodeID n, size_t *out) const {
        auto it = nodes.find(n);
        if (it == nodes.end()) return false;
        *out = it->second.len();
        return true;
    }
    // graph.rs:103-113
    bool neighbors_vec(NodeID n, std::vector<NodeID> *out) const {
        auto it = nodes.find(n);
        if (it == nodes.end()) return false;
        *out = it->second.v;
        return true;
    }
    // graph.rs:85-94
    GraphErr isolate_node(NodeID node) {
        std::vector<NodeID> nb;
        if (!neighbors_vec(node, &nb)) return G_NODE_NOT_IN_GRAPH;
        for (NodeID neighbor : nb) {
            size_t deg;
            if (!degree(neighbor, &deg)) return G_NODE_NOT_IN_GRAPH;
            if (deg == 1) continue;
            GraphErr e = remove_edge(node, neighbor);
            if (e != G_OK) return e;
        }
        return G_OK;
    }
    // graph.rs:140-148
    template <class It>
    GraphErr add_neighbors(NodeID node, It begin, It end) {
        for (It it = begin; it != end; ++it) {
            GraphErr e = add_edge(node, *it);
            if (e != G_OK) return e;
        }
        return G_OK;
    }
    // graph.rs:128-138
    template <class It>
    GraphErr replace_neighbors(NodeID node, It begin, It end) {
        GraphErr e = isolate_node(node);
        if (e != G_OK) return e;
        return add_neighbors(node, begin, end);
    }
    // iter_nodes (graph.rs:27-29) in this oracle's documented order: ascending id
    std::vector<NodeID> iter_nodes() const {
        std::vector<NodeID> ids;
        ids.reserve(nodes.size());
        for (auto &kv : nodes) ids.push_back(kv.first);
        std::sort(ids.begin(), ids.end());
        return ids;
    }
};

// graph/src/layers.rs:7-70
struct Layers {
    std::vector<Graph> levels;
    size_t m;
    size_t len() const { return levels.size(); }
    void add_level(size_t level) {  // layers.rs:48-59
        while (len() <= level) {
            Graph g;
            g.level = len();
            g.m = (len() == 0) ? m * 2 : m;
            levels.push_back(std::move(g));
        }
    }
    void add_node(NodeID id, size_t level) {  // layers.rs:63-70
        add_level(level);
        for (size_t l = 0; l <= level && l < levels.size(); l++) levels[l].add_node(id);
    }
};

// ---------------------------------------------------------------------------------------------
// points crate (SimplePoints as SoA; arithmetic unchanged)
// ---------------------------------------------------------------------------------------------

struct PointRef {  // points/src/point.rs:6-10 (a view)
    NodeID id = 0;
    uint8_t level = 0;
    const uint8_t *codes = nullptr;
    float delta = 0.0f, min = 0.0f;
    const float *vals = nullptr;
};

struct Points {
    int kind;
    uint32_t dim;
    std::vector<uint8_t> codes;
    std::vector<float> mins, deltas;
    std::vector<float> vals;
    std::vector<uint8_t> levels;
    size_t len() const { return levels.size(); }
    bool get_point(NodeID idx, PointRef *p) const {  // points.rs:75-77
        if ((size_t)idx >= len()) return false;
        p->id = idx;
        p->level = levels[idx];
        if (kind == ORC_VEC_QUANT8) {
            p->codes = &codes[(size_t)idx * dim];
            p->delta = deltas[idx];
            p->min = mins[idx];
            p->vals = nullptr;
        } else {
            p->codes = nullptr;
            p->delta = p->min = 0.0f;
            p->vals = &vals[(size_t)idx * dim];
        }
        return true;
    }
};

// Point::dist2other -> VecType::dist2other (points/src/point.rs:35-37)
inline float dist2other(const Points &pts, const PointRef &a, const PointRef &b) {
    if (pts.kind == ORC_VEC_QUANT8)
        return dist_quant(pts.dim, a.codes, a.delta, a.min, b.codes, b.delta, b.min);
    return dist_full(pts.dim, a.vals, b.vals);
}

// a free-standing Point (the query): Point::new, points/src/point.rs:24-30
struct OwnedPoint {
    std::vector<uint8_t> codes;
    std::vector<float> vals;
    float delta = 0.0f, min = 0.0f;
    PointRef ref;
};
int make_point(const Points &pts, const float *v, OwnedPoint *p) {
    p->ref.id = 0;
    p->ref.level = 0;
    if (pts.kind == ORC_VEC_QUANT8) {
        p->codes.resize(pts.dim);
        int rc = quantize(v, pts.dim, &p->min, &p->delta, p->codes.data());
        if (rc != ORC_OK) return rc;
        p->ref.codes = p->codes.data();
        p->ref.delta = p->delta;
        p->ref.min = p->min;
        p->ref.vals = nullptr;
    } else {
        for (uint32_t i = 0; i < pts.dim; i++)
            if (std::isnan(v[i])) return ORC_ERR_NAN;
        p->vals.assign(v, v + pts.dim);
        p->ref.codes = nullptr;
        p->ref.vals = p->vals.data();
        p->ref.delta = p->ref.min = 0.0f;
    }
    return ORC_OK;
}

// ---------------------------------------------------------------------------------------------
// hnsw crate
// ---------------------------------------------------------------------------------------------

// hnsw/src/params.rs:5-42
struct Params {
    NodeID ep;
    size_t m, mmax, mmax0;
    float ml;
    size_t ef_cons, dim;
};

typedef std::map<NodeID, OrderedDists> LayerResult;    // IntMap<NodeID, OrderedDists>
typedef std::map<size_t, LayerResult> LayersResults;   // IntMap<usize, LayerResult>

struct Counters {
    uint64_t n_dist = 0, n_exp = 0, sum_deg = 0;
};

// hnsw/src/template/results.rs:26-45
struct Results {
    OrderedDists selected, candidates;
    std::unordered_set<NodeID> visited;
    OrderedDists visited_h;
    LayersResults insertion_results, prune_results;
    void clear_all() {  // results.rs:182-190
        selected.clear();
        candidates.clear();
        visited.clear();
        visited_h.clear();
        insertion_results.clear();
        prune_results.clear();
    }
};

}  // namespace

struct orc_index {
    Params params;
    Layers layers;
    Points points;
};

namespace {

typedef orc_index HNSW;

// hnsw/src/template/searcher.rs:23-103  Searcher::search_layer
int search_layer(Results &results, const Graph &layer, const PointRef &point, const HNSW &index,
                 size_t ef, Counters *ctr) {
    // results.rs:148-157 extend_candidates_with_selected
    for (const Dist &node : results.selected) results.candidates.insert(node);
    // results.rs:159-168 extend_visited_with_selected
    for (const Dist &node : results.selected) results.visited.insert(node.id);

    while (!results.candidates.empty()) {
        const Dist cand_dist = *results.candidates.begin();  // pop_first
        results.candidates.erase(results.candidates.begin());
        const Dist furthest2q_dist = *results.selected.rbegin();  // selected.last()
        if (dist_cmp(cand_dist, furthest2q_dist) > 0) break;
        std::vector<NodeID> cand_neighbors;
        if (!layer.neighbors_vec(cand_dist.id, &cand_neighbors)) return ORC_ERR_NODE_NOT_IN_GRAPH;
        if (ctr) {
            ctr->n_exp += 1;
            ctr->sum_deg += cand_neighbors.size();
        }
        std::vector<Dist> q2cand_neighbors_dists;
        for (NodeID node : cand_neighbors) {
            if (!results.visited.insert(node).second) continue;  // results.rs:101-103
            PointRef p;
            if (!index.points.get_point(node, &p)) return ORC_ERR_ARG;  // .expect(..) panics
            const float dist = dist2other(index.points, p, point);
            if (std::isnan(dist)) return ORC_ERR_NAN;  // Dist::cmp would panic
            if (ctr) ctr->n_dist += 1;
            q2cand_neighbors_dists.push_back(Dist{node, dist});
        }
        for (const Dist &n2q_dist : q2cand_neighbors_dists) {
            const Dist f2q_dist = *results.selected.rbegin();
            if (results.selected.size() < ef) {
                results.selected.insert(n2q_dist);
                results.candidates.insert(n2q_dist);
                continue;
            }
            if (dist_cmp(n2q_dist, f2q_dist) < 0) {
                results.selected.insert(n2q_dist);
                results.candidates.insert(n2q_dist);
                if (results.selected.size() > ef) {
                    results.selected.erase(std::prev(results.selected.end()));  // pop_last
                }
            }
        }
    }
    results.candidates.clear();  // searcher.rs:100
    results.visited.clear();     // searcher.rs:101
    return ORC_OK;
}

// results.rs:69-77 get_nearest_from_selected
Dist get_nearest_from_selected(const Results &results, const PointRef &point, const Points &pts) {
    bool first = true;
    Dist best{0, 0.0f};
    for (const Dist &s : results.selected) {
        PointRef sp;
        pts.get_point(s.id, &sp);
        Dist d{s.id, dist2other(pts, point, sp)};
        if (first || dist_cmp(d, best) < 0) {
            best = d;
            first = false;
        }
    }
    return best;
}

// searcher.rs:109-153 select_heuristic (+ results.rs:105-146 helpers)
int select_heuristic(Results &results, const Graph &layer, const PointRef &point,
                     const Points &points, size_t m, bool extend_cands, bool keep_pruned) {
    // results.rs:105-111 select_setup
    results.visited_h.clear();
    results.candidates.clear();
    for (const Dist &d : results.selected) results.candidates.insert(d);
    results.selected.clear();
    if (extend_cands) {
        // results.rs:122-146 extend_candidates_with_neighbors
        std::vector<NodeID> neighbors;
        for (const Dist &node : results.candidates) {
            auto it = layer.nodes.find(node.id);
            if (it == layer.nodes.end()) return ORC_ERR_NODE_NOT_IN_GRAPH;  // panic!
            for (NodeID nb : it->second.v) neighbors.push_back(nb);
        }
        for (NodeID nb : neighbors) {
            PointRef np;
            if (!points.get_point(nb, &np)) return ORC_ERR_ARG;
            // points.distance(point.id, neighbor): a = point, b = neighbor (points.rs:86-93)
            Dist d{nb, dist2other(points, point, np)};
            if (std::isnan(d.dist)) return ORC_ERR_NAN;
            results.candidates.insert(d);
        }
    }
    if (results.candidates.empty()) return ORC_ERR_EMPTY;  // pop_first().unwrap() panics
    {
        Dist node_e = *results.candidates.begin();
        results.candidates.erase(results.candidates.begin());
        results.selected.insert(node_e);
    }
    while (!results.candidates.empty() && results.selected.size() < m) {
        Dist node_e = *results.candidates.begin();
        results.candidates.erase(results.candidates.begin());
        PointRef e_point;
        if (!points.get_point(node_e.id, &e_point)) return ORC_ERR_ARG;
        Dist nearest_selected = get_nearest_from_selected(results, e_point, points);
        if (dist_cmp(node_e, nearest_selected) < 0) {
            results.selected.insert(node_e);
        } else if (keep_pruned) {
            results.visited_h.insert(node_e);
        }
    }
    if (keep_pruned) {
        while (!results.visited_h.empty() && results.selected.size() < m) {
            Dist node_e = *results.visited_h.begin();
            results.visited_h.erase(results.visited_h.begin());
            results.selected.insert(node_e);
        }
    }
    return ORC_OK;  // NB: candidates is NOT cleared
}

// hnsw/src/template/inserter.rs:19-127
struct Inserter {
    Results results;

    int build_insertion_results(const HNSW &index, const PointRef &point) {
        if (point.id == index.params.ep) return ORC_OK;  // inserter.rs:42-45 (results left stale)
        // setup_insert, inserter.rs:53-68
        results.clear_all();
        PointRef ep;
        if (!index.points.get_point(index.params.ep, &ep)) return ORC_ERR_ARG;
        const float dist2ep = dist2other(index.points, ep, point);  // index.distance(ep, point.id)
        if (std::isnan(dist2ep)) return ORC_ERR_NAN;
        results.selected.insert(Dist{index.params.ep, dist2ep});
        // traverse_layers_above, inserter.rs:70-89
        const size_t layers_len = index.layers.len();
        for (size_t layer_nb = layers_len; layer_nb-- > (size_t)point.level + 1;) {
            int rc = search_layer(results, index.layers.levels[layer_nb], point, index, 1, nullptr);
            if (rc != ORC_OK) return rc;
        }
        // traverse_layers_below, inserter.rs:91-126
        const size_t bound = std::min((size_t)point.level, layers_len - 1);
        for (size_t layer_nb = bound + 1; layer_nb-- > 0;) {
            const Graph &layer = index.layers.levels[layer_nb];
            int rc = search_layer(results, layer, point, index, index.params.ef_cons, nullptr);
            if (rc != ORC_OK) return rc;
            rc = select_heuristic(results, layer, point, index.points, index.params.m, true, true);
            if (rc != ORC_OK) return rc;
            // results.rs:79-84 save_layer_results
            results.insertion_results[layer_nb][point.id] = results.selected;
        }
        return ORC_OK;
    }
};

// template.rs:614-621 select_simple
OrderedDists select_simple(std::vector<Dist> cands, size_t m) {
    std::sort(cands.begin(), cands.end(), DistLess());
    OrderedDists out;
    for (size_t i = 0; i < cands.size() && i < m; i++) out.insert(cands[i]);
    return out;
}

// template.rs:177-190 insert (+196-251)
int insert(HNSW &index, NodeID point_id, Inserter &inserter) {
    PointRef point;
    if (!index.points.get_point(point_id, &point)) return ORC_ERR_ARG;
    int rc = inserter.build_insertion_results(index, point);
    if (rc != ORC_OK) return rc;
    Results &results = inserter.results;
    // make_connections, template.rs:196-207
    for (auto &lr : results.insertion_results) {
        Graph &layer = index.layers.levels[lr.first];
        for (auto &nd : lr.second) {
            for (const Dist &n : nd.second) {
                if (layer.add_edge(nd.first, n.id) != G_OK) return ORC_ERR_NODE_NOT_IN_GRAPH;
            }
        }
    }
    // prune_connections, template.rs:209-238
    results.prune_results.clear();
    {
        LayersResults snapshot = results.insertion_results;  // .clone()
        for (auto &lr : snapshot) {
            const size_t layer_nb = lr.first;
            Graph &layer = index.layers.levels[layer_nb];
            for (auto &nd : lr.second) {
                for (const Dist &to_prune : nd.second) {
                    size_t deg;
                    if (!layer.degree(to_prune.id, &deg)) return ORC_ERR_NODE_NOT_IN_GRAPH;
                    if (!(deg > layer.m)) continue;
                    std::vector<NodeID> nbrs;
                    layer.neighbors_vec(to_prune.id, &nbrs);
                    std::vector<Dist> dists;
                    PointRef a;
                    index.points.get_point(to_prune.id, &a);
                    for (NodeID n : nbrs) {
                        PointRef b;
                        if (!index.points.get_point(n, &b)) return ORC_ERR_ARG;
                        dists.push_back(Dist{n, dist2other(index.points, a, b)});
                    }
                    results.prune_results[layer_nb][to_prune.id] = select_simple(dists, layer.m);
                }
            }
        }
    }
    // make_pruned_connections, template.rs:240-251
    for (auto &lr : results.prune_results) {
        Graph &layer = index.layers.levels[lr.first];
        for (auto &nd : lr.second) {
            std::vector<NodeID> ids;
            for (const Dist &n : nd.second) ids.push_back(n.id);
            if (layer.replace_neighbors(nd.first, ids.begin(), ids.end()) != G_OK)
                return ORC_ERR_NODE_NOT_IN_GRAPH;
        }
    }
    return ORC_OK;
}

// template.rs:269-293 store_points (levels explicit: points.rs:39-48,148-160 draw them from
// rand's StdRng, which is not under /root/reference)
int store_points(HNSW &index, const float *rows, uint64_t n, const uint8_t *levels,
                 std::vector<NodeID> *ids_out) {
    if (n == 0) return ORC_ERR_EMPTY;
    Points &pts = index.points;
    const uint32_t d = pts.dim;
    std::vector<NodeID> ids;
    for (uint64_t i = 0; i < n; i++) {
        const float *v = rows + i * d;
        if (pts.kind == ORC_VEC_QUANT8) {
            std::vector<uint8_t> codes(d);
            float mn, dl;
            int rc = quantize(v, d, &mn, &dl, codes.data());
            if (rc != ORC_OK) return rc;
            pts.codes.insert(pts.codes.end(), codes.begin(), codes.end());
            pts.mins.push_back(mn);
            pts.deltas.push_back(dl);
        } else {
            for (uint32_t j = 0; j < d; j++)
                if (std::isnan(v[j])) return ORC_ERR_NAN;
            pts.vals.insert(pts.vals.end(), v, v + d);
        }
        pts.levels.push_back(levels[i]);
        ids.push_back((NodeID)(pts.len() - 1));  // points.rs:64-73 push: id = position
    }
    for (NodeID id : ids) index.layers.add_node(id, pts.levels[id]);
    const size_t max_layer_nb = index.layers.len() - 1;
    // template.rs:284: first key of the top layer in hash order -> documented default: smallest id
    NodeID new_ep = 0;
    bool first = true;
    for (auto &kv : index.layers.levels[max_layer_nb].nodes) {
        if (first || kv.first < new_ep) new_ep = kv.first;
        first = false;
    }
    index.params.ep = new_ep;
    if (ids_out) *ids_out = ids;
    return ORC_OK;
}

// template.rs:306-335 ann_by_vector
int ann_by_vector(const HNSW &index, const float *vector, uint32_t n, uint32_t ef,
                  std::vector<Dist> *out, Counters *ctr) {
    if (index.points.len() == 0 || index.layers.len() == 0) return ORC_ERR_EMPTY;
    OwnedPoint point;
    int rc = make_point(index.points, vector, &point);
    if (rc != ORC_OK) return rc;
    Results results;
    PointRef ep;
    if (!index.points.get_point(index.params.ep, &ep)) return ORC_ERR_ARG;  // .unwrap() panics
    const float d0 = dist2other(index.points, point.ref, ep);  // distance2point(&point, ep)
    if (std::isnan(d0)) return ORC_ERR_NAN;
    if (ctr) ctr->n_dist += 1;
    results.selected.insert(Dist{index.params.ep, d0});
    const size_t nb_layers = index.layers.len();
    for (size_t layer_nb = nb_layers; layer_nb-- > 1;) {
        rc = search_layer(results, index.layers.levels[layer_nb], point.ref, index, 1, ctr);
        if (rc != ORC_OK) return rc;
    }
    rc = search_layer(results, index.layers.levels[0], point.ref, index, ef, ctr);
    if (rc != ORC_OK) return rc;
    out->clear();
    for (const Dist &d : results.selected) {  // results.rs:59-61 get_top_selected(n)
        if (out->size() >= n) break;
        out->push_back(d);
    }
    return ORC_OK;
}

void write_result(const std::vector<Dist> &res, uint32_t n, uint32_t *ids, float *dists,
                  uint32_t *count) {
    for (uint32_t i = 0; i < n; i++) {
        if (i < res.size()) {
            ids[i] = res[i].id;
            if (dists) dists[i] = res[i].dist;
        } else {
            ids[i] = UINT32_MAX;
            if (dists) dists[i] = INFINITY;
        }
    }
    if (count) *count = (uint32_t)res.size();
}

}  // namespace

// =============================================================================================
// C interface
// =============================================================================================
extern "C" {

int orc_quantize(const float *v, uint32_t d, float *min_out, float *delta_out, uint8_t *codes) {
    return quantize(v, d, min_out, delta_out, codes);
}
float orc_dist_quant(uint32_t d, const uint8_t *cx, float delta_x, float min_x, const uint8_t *cy,
                     float delta_y, float min_y) {
    return dist_quant(d, cx, delta_x, min_x, cy, delta_y, min_y);
}
float orc_dist_full(uint32_t d, const float *x, const float *y) { return dist_full(d, x, y); }

// vectors/src/quant.rs:67-73 over iter_vals (quant.rs:79-83)
float orc_dist_generic_qq(uint32_t d, const uint8_t *cx, float delta_x, float min_x,
                          const uint8_t *cy, float delta_y, float min_y) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
        const float x = ((float)cx[i] * delta_x) + min_x;
        const float y = ((float)cy[i] * delta_y) + min_y;
        const float t = x - y;
        sum += t * t;
    }
    return std::sqrt(sum);
}
float orc_dist_generic_qf(uint32_t d, const uint8_t *cx, float delta_x, float min_x,
                          const float *y) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
        const float x = ((float)cx[i] * delta_x) + min_x;
        const float t = x - y[i];
        sum += t * t;
    }
    return std::sqrt(sum);
}
int orc_dist_cmp(uint32_t id_a, float d_a, uint32_t id_b, float d_b) {
    return dist_cmp(Dist{id_a, d_a}, Dist{id_b, d_b});
}
uint8_t orc_level_from_uniform(float r, float ml) {
    // points/src/points.rs:158: (-rand_nb.ln() * ml).floor() as usize ; point.rs:15 `level as u8`
    const float lv = std::floor(-std::log(r) * ml);
    if (std::isnan(lv) || lv <= 0.0f) return 0;
    if (lv >= 1.8446744e19f) return 255;  // usize saturates, then truncates to u8 (0xFF)
    return (uint8_t)((uint64_t)lv & 0xFF);
}
float orc_default_ml(uint32_t m) { return 1.0f / std::log((float)m); }  // params.rs:15-17

orc_index *orc_new(uint32_t m, uint32_t ef_cons, uint32_t dim, int vec_kind) {
    orc_index *h = new orc_index();
    h->params.ep = 0;
    h->params.m = m;
    h->params.mmax = m;
    h->params.mmax0 = (size_t)m * 2;
    h->params.ml = orc_default_ml(m);
    h->params.ef_cons = ef_cons ? ef_cons : (size_t)m * 2;
    h->params.dim = dim;
    h->layers.m = m;
    h->points.kind = vec_kind;
    h->points.dim = dim;
    return h;
}
void orc_free(orc_index *h) { delete h; }
orc_index *orc_clone(const orc_index *h) { return new orc_index(*h); }

int orc_insert_bulk(orc_index *h, const float *rows, uint64_t n, const uint8_t *levels) {
    std::vector<NodeID> ids;
    int rc = store_points(*h, rows, n, levels, &ids);
    if (rc != ORC_OK) return rc;
    std::unordered_set<NodeID> stored_ids(ids.begin(), ids.end());
    // template.rs:403-440 with nb_threads == 1: one chunk per layer, one fresh Inserter per chunk
    for (size_t layer_nb = h->layers.len(); layer_nb-- > 0;) {
        std::vector<NodeID> layer_ids;
        for (NodeID id : h->layers.levels[layer_nb].iter_nodes())
            if (stored_ids.count(id) && h->points.levels[id] == (uint8_t)layer_nb)
                layer_ids.push_back(id);
        Inserter inserter;
        for (NodeID id : layer_ids) {
            rc = insert(*h, id, inserter);
            if (rc != ORC_OK) return rc;
        }
    }
    return ORC_OK;
}

int orc_insert_vec(orc_index *h, const float *v, uint8_t level, uint32_t *out_id) {
    std::vector<NodeID> ids;
    int rc = store_points(*h, v, 1, &level, &ids);  // template.rs:167
    if (rc != ORC_OK) return rc;
    const NodeID point_id = ids[0];
    h->layers.add_node(point_id, h->points.levels[point_id]);  // template.rs:170 (idempotent)
    Inserter inserter;                                          // template.rs:171
    rc = insert(*h, point_id, inserter);
    if (rc != ORC_OK) return rc;
    if (out_id) *out_id = point_id;
    return ORC_OK;
}

int orc_import_points(orc_index *h, const float *rows, uint64_t n, const uint8_t *levels) {
    Points &pts = h->points;
    const uint32_t d = pts.dim;
    for (uint64_t i = 0; i < n; i++) {
        const float *v = rows + i * d;
        if (pts.kind == ORC_VEC_QUANT8) {
            std::vector<uint8_t> codes(d);
            float mn, dl;
            int rc = quantize(v, d, &mn, &dl, codes.data());
            if (rc != ORC_OK) return rc;
            pts.codes.insert(pts.codes.end(), codes.begin(), codes.end());
            pts.mins.push_back(mn);
            pts.deltas.push_back(dl);
        } else {
            pts.vals.insert(pts.vals.end(), v, v + d);
        }
        pts.levels.push_back(levels ? levels[i] : 0);
    }
    return ORC_OK;
}
int orc_import_points_quant(orc_index *h, const uint8_t *codes, const float *mins,
                            const float *deltas, uint64_t n, const uint8_t *levels) {
    Points &pts = h->points;
    if (pts.kind != ORC_VEC_QUANT8) return ORC_ERR_ARG;
    pts.codes.insert(pts.codes.end(), codes, codes + n * pts.dim);
    pts.mins.insert(pts.mins.end(), mins, mins + n);
    pts.deltas.insert(pts.deltas.end(), deltas, deltas + n);
    for (uint64_t i = 0; i < n; i++) pts.levels.push_back(levels ? levels[i] : 0);
    return ORC_OK;
}
int orc_import_layer(orc_index *h, uint32_t layer, uint64_t n_nodes, const uint32_t *node_ids,
                     const uint64_t *offsets, const uint32_t *nbrs) {
    if (layer != h->layers.len()) return ORC_ERR_ARG;
    h->layers.add_level(layer);
    Graph &g = h->layers.levels[layer];
    g.nodes.reserve(n_nodes);
    for (uint64_t i = 0; i < n_nodes; i++) {
        IdSet s;
        s.v.assign(nbrs + offsets[i], nbrs + offsets[i + 1]);
        std::sort(s.v.begin(), s.v.end());
        g.nodes.emplace(node_ids[i], std::move(s));
    }
    return ORC_OK;
}
void orc_set_ep(orc_index *h, uint32_t ep) { h->params.ep = ep; }

uint64_t orc_len(const orc_index *h) { return h->points.len(); }
uint32_t orc_ep(const orc_index *h) { return h->params.ep; }
uint32_t orc_nb_layers(const orc_index *h) { return (uint32_t)h->layers.len(); }
uint64_t orc_layer_nb_nodes(const orc_index *h, uint32_t layer) {
    return layer < h->layers.len() ? h->layers.levels[layer].nodes.size() : 0;
}
uint32_t orc_layer_m(const orc_index *h, uint32_t layer) {
    return layer < h->layers.len() ? (uint32_t)h->layers.levels[layer].m : 0;
}
uint64_t orc_layer_nodes(const orc_index *h, uint32_t layer, uint32_t *out, uint64_t cap) {
    if (layer >= h->layers.len()) return 0;
    std::vector<NodeID> ids = h->layers.levels[layer].iter_nodes();
    for (uint64_t i = 0; i < ids.size() && i < cap; i++) out[i] = ids[i];
    return ids.size();
}
int64_t orc_neighbors(const orc_index *h, uint32_t layer, uint32_t id, uint32_t *out,
                      uint64_t cap) {
    if (layer >= h->layers.len()) return -1;
    auto it = h->layers.levels[layer].nodes.find(id);
    if (it == h->layers.levels[layer].nodes.end()) return -1;
    const std::vector<NodeID> &v = it->second.v;
    for (uint64_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return (int64_t)v.size();
}
int orc_distance(const orc_index *h, uint32_t a, uint32_t b, float *out) {
    PointRef pa, pb;
    if (!h->points.get_point(a, &pa) || !h->points.get_point(b, &pb)) return ORC_ERR_ARG;
    *out = dist2other(h->points, pa, pb);
    return ORC_OK;
}
int orc_get_vals(const orc_index *h, uint32_t id, float *out) {
    PointRef p;
    if (!h->points.get_point(id, &p)) return ORC_ERR_ARG;
    for (uint32_t i = 0; i < h->points.dim; i++)
        out[i] = (h->points.kind == ORC_VEC_QUANT8) ? ((float)p.codes[i] * p.delta) + p.min
                                                    : p.vals[i];
    return ORC_OK;
}
int orc_get_quant(const orc_index *h, uint32_t id, uint8_t *codes, float *min_out, float *delta_out,
                  uint8_t *level) {
    PointRef p;
    if (h->points.kind != ORC_VEC_QUANT8 || !h->points.get_point(id, &p)) return ORC_ERR_ARG;
    if (codes) memcpy(codes, p.codes, h->points.dim);
    if (min_out) *min_out = p.min;
    if (delta_out) *delta_out = p.delta;
    if (level) *level = p.level;
    return ORC_OK;
}

int orc_ann_by_vector(const orc_index *h, const float *q, uint32_t n, uint32_t ef, uint32_t *ids,
                      float *dists, uint32_t *count, uint64_t *stats) {
    std::vector<Dist> res;
    Counters ctr;
    int rc = ann_by_vector(*h, q, n, ef, &res, &ctr);
    if (rc != ORC_OK) return rc;
    write_result(res, n, ids, dists, count);
    if (stats) {
        stats[0] = ctr.n_dist;
        stats[1] = ctr.n_exp;
        stats[2] = ctr.sum_deg;
    }
    return ORC_OK;
}

int orc_search_batch(const orc_index *h, const float *Q, uint64_t nq, uint32_t n, uint32_t ef,
                     uint32_t *ids, float *dists, uint32_t *counts, uint64_t *stats,
                     int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<int> rcs(nthreads, ORC_OK);
    auto work = [&](int t) {
        const uint64_t lo = nq * t / nthreads, hi = nq * (t + 1) / nthreads;
        for (uint64_t i = lo; i < hi; i++) {
            int rc = orc_ann_by_vector(h, Q + i * h->points.dim, n, ef, ids + i * n,
                                       dists ? dists + i * n : nullptr,
                                       counts ? counts + i : nullptr,
                                       stats ? stats + i * 3 : nullptr);
            if (rc != ORC_OK) rcs[t] = rc;
        }
    };
    if (nthreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (int rc : rcs)
        if (rc != ORC_OK) return rc;
    return ORC_OK;
}

int orc_search_layer(const orc_index *h, uint32_t layer, const float *q, const uint32_t *entry_ids,
                     uint32_t n_entry, uint32_t ef, uint32_t *out_ids, float *out_dists,
                     uint32_t *out_count, uint64_t *stats) {
    if (layer >= h->layers.len()) return ORC_ERR_ARG;
    OwnedPoint point;
    int rc = make_point(h->points, q, &point);
    if (rc != ORC_OK) return rc;
    Results results;
    Counters ctr;
    for (uint32_t i = 0; i < n_entry; i++) {
        PointRef p;
        if (!h->points.get_point(entry_ids[i], &p)) return ORC_ERR_ARG;
        const float d = dist2other(h->points, point.ref, p);
        if (std::isnan(d)) return ORC_ERR_NAN;
        ctr.n_dist += 1;
        results.selected.insert(Dist{entry_ids[i], d});
    }
    if (results.selected.empty()) return ORC_ERR_EMPTY;
    rc = search_layer(results, h->layers.levels[layer], point.ref, *h, ef, &ctr);
    if (rc != ORC_OK) return rc;
    uint32_t k = 0;
    for (const Dist &d : results.selected) {
        out_ids[k] = d.id;
        if (out_dists) out_dists[k] = d.dist;
        k++;
    }
    *out_count = k;
    if (stats) {
        stats[0] = ctr.n_dist;
        stats[1] = ctr.n_exp;
        stats[2] = ctr.sum_deg;
    }
    return ORC_OK;
}

int orc_distance_batch(const orc_index *h, const float *q, const uint32_t *ids, uint64_t k,
                       float *out) {
    OwnedPoint point;
    int rc = make_point(h->points, q, &point);
    if (rc != ORC_OK) return rc;
    for (uint64_t i = 0; i < k; i++) {
        PointRef p;
        if (!h->points.get_point(ids[i], &p)) return ORC_ERR_ARG;
        out[i] = dist2other(h->points, p, point.ref);  // searcher.rs:66-69 operand order
    }
    return ORC_OK;
}

int orc_brute_force(const orc_index *h, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids,
                    float *dists, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const uint64_t N = h->points.len();
    std::vector<int> rcs(nthreads, ORC_OK);
    auto work = [&](int t) {
        const uint64_t lo = nq * t / nthreads, hi = nq * (t + 1) / nthreads;
        std::vector<Dist> all(N);
        for (uint64_t qi = lo; qi < hi; qi++) {
            OwnedPoint point;
            int rc = make_point(h->points, Q + qi * h->points.dim, &point);
            if (rc != ORC_OK) {
                rcs[t] = rc;
                return;
            }
            for (uint64_t i = 0; i < N; i++) {
                PointRef p;
                h->points.get_point((NodeID)i, &p);
                all[i] = Dist{(NodeID)i, dist2other(h->points, point.ref, p)};
            }
            const uint64_t kk = std::min<uint64_t>(k, N);
            // full sort in the reference (glove.rs:107); partial_sort gives the same prefix
            std::partial_sort(all.begin(), all.begin() + kk, all.end(), DistLess());
            for (uint32_t j = 0; j < k; j++) {
                ids[qi * k + j] = j < kk ? all[j].id : UINT32_MAX;
                if (dists) dists[qi * k + j] = j < kk ? all[j].dist : INFINITY;
            }
        }
    };
    if (nthreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (int rc : rcs)
        if (rc != ORC_OK) return rc;
    return ORC_OK;
}

}  // extern "C"
