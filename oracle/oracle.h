/*
 * oracle.h -- C interface of the CPU ORACLE for the HNSW search path of Gumo-A/hnsw_rs.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a literal CPU restatement of the
 * reference's Rust semantics, used only as the checker by tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py.  Nothing under hnsw_rs_amd/ may include, link or call it.
 *
 * Parity pin: the restatement is checked against every known-answer the reference's own tests
 * hold for this path (tests/test_oracle_kat.py):
 *   - distance KATs   vectors/src/quant.rs:154-201, vectors/src/full.rs:99-146
 *   - quant error<1%  vectors/tests/full_lvq_tests.rs:4-27
 *   - Dist ordering   hnsw/src/template/results.rs:209-231, graph/src/dist.rs:30-38
 *   - recall@10>0.99  hnsw/src/template.rs:518-572 (test-data, M=12, ef=100, n=10)
 *   - graph invariants graph/src/graph.rs:305-432
 * The reference itself (Rust) cannot be compiled here (no rustc/cargo in the image), and it
 * stores no expected id lists, so id-level goldens under tests/golden/ are produced by this
 * oracle and cross-checked by an independent numpy restatement (oracle/oracle_np.py).
 * What stays UNPINNED (third-party behaviour absent from /root/reference): rand 0.8.5 StdRng
 * level draws and hashbrown iteration order -- levels / insertion order / entry point are
 * therefore explicit inputs here.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef HNSW_ORACLE_H
#define HNSW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_VEC_QUANT8 0 /* points/src/point.rs:4  type VecType = QuantVec (as shipped) */
#define ORC_VEC_F32 1    /* the alternate VecType = FullVec (vectors/src/full.rs)        */

#define ORC_OK 0
#define ORC_ERR_BAD_DIM (-1)
#define ORC_ERR_NAN (-2)
#define ORC_ERR_NODE_NOT_IN_GRAPH (-3)
#define ORC_ERR_EMPTY (-4)
#define ORC_ERR_ARG (-5)

typedef struct orc_index orc_index;

/* ---- arithmetic (vectors crate) ------------------------------------------------------ */
/* QuantVec::new, vectors/src/quant.rs:41-66.  Returns ORC_ERR_NAN where Rust would panic
 * (partial_cmp().unwrap() on NaN), ORC_ERR_EMPTY for d == 0. */
int orc_quantize(const float *v, uint32_t d, float *min_out, float *delta_out, uint8_t *codes);
/* QuantVec::distance_unrolled == dist2other, vectors/src/quant.rs:14-37,75-77 */
float orc_dist_quant(uint32_t d, const uint8_t *cx, float delta_x, float min_x, const uint8_t *cy,
                     float delta_y, float min_y);
/* FullVec::distance == dist2other, vectors/src/full.rs:23-33 */
float orc_dist_full(uint32_t d, const float *x, const float *y);
/* generic VecBase::distance over dequantised iterators, vectors/src/quant.rs:67-73,79-83
 * (used by the reference's tests only); kinds: 0 = quant, 1 = full, per operand. */
float orc_dist_generic_qq(uint32_t d, const uint8_t *cx, float delta_x, float min_x,
                          const uint8_t *cy, float delta_y, float min_y);
float orc_dist_generic_qf(uint32_t d, const uint8_t *cx, float delta_x, float min_x,
                          const float *y);
/* Dist::cmp, graph/src/dist.rs:30-38: -1 / 0 / +1; -2 if either distance is NaN (Rust panics) */
int orc_dist_cmp(uint32_t id_a, float d_a, uint32_t id_b, float d_b);
/* new_layer's arithmetic, points/src/points.rs:158: floor(-ln(r) * ml) as usize, then `as u8` */
uint8_t orc_level_from_uniform(float r, float ml);
/* get_default_ml, hnsw/src/params.rs:15-17 */
float orc_default_ml(uint32_t m);

/* ---- index (hnsw crate) ---------------------------------------------------------------- */
/* HNSW::new, hnsw/src/template.rs:133-144 + Params::from_m / from_m_efcons params.rs:20-42.
 * ef_cons == 0 means None (default 2*m). */
orc_index *orc_new(uint32_t m, uint32_t ef_cons, uint32_t dim, int vec_kind);
void orc_free(orc_index *);
orc_index *orc_clone(const orc_index *);

/* HNSW::insert_bulk with nb_threads == 1, hnsw/src/template.rs:388-444.
 * levels[n] are the explicit level draws (see header note); insertion order inside a level is
 * ascending id; the entry point is the smallest id on the top layer (documented defaults for
 * the reference's hash-iteration order). */
int orc_insert_bulk(orc_index *, const float *rows, uint64_t n, const uint8_t *levels);
/* HNSW::insert_vec, hnsw/src/template.rs:165-173 */
int orc_insert_vec(orc_index *, const float *v, uint8_t level, uint32_t *out_id);

/* Import a prebuilt index instead of building (points are quantised by the oracle itself). */
int orc_import_points(orc_index *, const float *rows, uint64_t n, const uint8_t *levels);
int orc_import_points_quant(orc_index *, const uint8_t *codes, const float *mins,
                            const float *deltas, uint64_t n, const uint8_t *levels);
/* nodes of one layer with CSR adjacency; layers must be imported in order 0,1,2,... */
int orc_import_layer(orc_index *, uint32_t layer, uint64_t n_nodes, const uint32_t *node_ids,
                     const uint64_t *offsets, const uint32_t *nbrs);
void orc_set_ep(orc_index *, uint32_t ep);

/* accessors (template.rs:146-156,192; graph.rs:103-113,150-163) */
uint64_t orc_len(const orc_index *);
uint32_t orc_ep(const orc_index *);
uint32_t orc_nb_layers(const orc_index *);
uint64_t orc_layer_nb_nodes(const orc_index *, uint32_t layer);
uint32_t orc_layer_m(const orc_index *, uint32_t layer);
/* fills node ids ascending; returns count */
uint64_t orc_layer_nodes(const orc_index *, uint32_t layer, uint32_t *out, uint64_t cap);
/* neighbours ascending; returns degree or -1 if the node is not in the layer */
int64_t orc_neighbors(const orc_index *, uint32_t layer, uint32_t id, uint32_t *out, uint64_t cap);
/* HNSW::distance(a,b), template.rs:150-152: returns 0 and sets *out, or ORC_ERR_ARG (None) */
int orc_distance(const orc_index *, uint32_t a, uint32_t b, float *out);
/* Point::get_vals (dequantised values), vectors/src/lib.rs:24-26 */
int orc_get_vals(const orc_index *, uint32_t id, float *out);
int orc_get_quant(const orc_index *, uint32_t id, uint8_t *codes, float *min_out, float *delta_out,
                  uint8_t *level);

/* HNSW::ann_by_vector, hnsw/src/template.rs:306-335.  ids[n] padded with UINT32_MAX; dists
 * (optional) are the distances the reference discards; stats (optional, 3 x u64): n_dist (all
 * dist2other calls incl. the entry point), n_exp (expanded candidates), sum_deg (sum of the
 * degrees of the expanded adjacency rows). */
int orc_ann_by_vector(const orc_index *, const float *q, uint32_t n, uint32_t ef, uint32_t *ids,
                      float *dists, uint32_t *count, uint64_t *stats);
/* loop of the above over nq queries on nthreads OS threads (queries statically partitioned) */
int orc_search_batch(const orc_index *, const float *Q, uint64_t nq, uint32_t n, uint32_t ef,
                     uint32_t *ids, float *dists, uint32_t *counts, uint64_t *stats, int nthreads);
/* Searcher::search_layer seam, searcher.rs:23-103: entry set -> selected after the layer */
int orc_search_layer(const orc_index *, uint32_t layer, const float *q, const uint32_t *entry_ids,
                     uint32_t n_entry, uint32_t ef, uint32_t *out_ids, float *out_dists,
                     uint32_t *out_count, uint64_t *stats);
/* VecBase::dist2many seam, vectors/src/lib.rs:17-22: query (quantised like ann_by_vector does)
 * against stored ids */
int orc_distance_batch(const orc_index *, const float *q, const uint32_t *ids, uint64_t k,
                       float *out);
/* exact top-k by full sort of Dist under the index's own metric,
 * hnsw/src/helpers/glove.rs:94-109 / template.rs:531-541 */
int orc_brute_force(const orc_index *, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids,
                    float *dists, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
