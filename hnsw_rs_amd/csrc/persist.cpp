// persist.cpp -- HNSW::save / HNSW::load (hnsw/src/template.rs:43-131) in the reference's own
// on-disk format, so that an index written by either implementation loads in the other.
//
// Everything is big-endian.
//   <dir>/params   52 bytes: m u64, mmax u64, mmax0 u64, ml f32, ef_cons u64, dim u64, ep u64
//                  (hnsw/src/params.rs:78-114; Serializer::size() says 58 but 52 are written)
//   <dir>/points   len u64, point_size u64, then per point: level u8 + vector
//                  (points/src/points.rs:124-145, points/src/point.rs:57-75)
//                  QuantVec = min f32, delta f32, dim codes   (vectors/src/quant.rs:102-124)
//                  FullVec  = dim x f32                         (vectors/src/full.rs:54-69)
//   <dir>/layers/<n>  level u8, nb_nodes u32, m u16, then per node: id u32 + m x u32 neighbour
//                  slots padded with 0xFFFFFFFF (graph/src/graph.rs:168-251)
// Deviation, documented: the reference writes deg > m rows longer than m slots, which corrupts
// the file (graph.rs:172-178).  The writer here stores max(m, max degree) in the m field so
// that every row fits and the file stays readable by the reference's own reader.

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "host_index.h"

namespace hx {
namespace {

void put_u64(std::vector<uint8_t> &b, uint64_t v) {
    for (int i = 7; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i)));
}
void put_u32(std::vector<uint8_t> &b, uint32_t v) {
    for (int i = 3; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i)));
}
void put_u16(std::vector<uint8_t> &b, uint16_t v) {
    b.push_back((uint8_t)(v >> 8));
    b.push_back((uint8_t)v);
}
void put_f32(std::vector<uint8_t> &b, float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    put_u32(b, u);
}
uint64_t get_u64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
uint32_t get_u32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
float get_f32(const uint8_t *p) {
    const uint32_t u = get_u32(p);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

bool write_file(const std::string &path, const std::vector<uint8_t> &b) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    return fclose(f) == 0 && ok;
}
bool read_file(const std::string &path, std::vector<uint8_t> *b) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    b->resize(n > 0 ? (size_t)n : 0);
    const bool ok = n <= 0 || fread(b->data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
bool is_dir(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

}  // namespace

int save_index(const HostIndex &idx, const std::string &dir) {
    if (idx.len() == 0) {
        set_error("cannot save an empty index");  // points.rs:127 get_point(0).unwrap() panics
        return HNSW_ERR_EMPTY;
    }
    if (!is_dir(dir) && mkdir(dir.c_str(), 0777) != 0) {
        set_error("Could not create dir %s", dir.c_str());
        return HNSW_ERR_IO;
    }
    std::vector<uint8_t> b;
    // points
    const uint64_t point_size =
        1 + (idx.kind == HNSW_VEC_QUANT8 ? 8 + (uint64_t)idx.dim : 4 * (uint64_t)idx.dim);
    put_u64(b, idx.len());
    put_u64(b, point_size);
    for (uint64_t i = 0; i < idx.len(); i++) {
        b.push_back(idx.levels[i]);
        if (idx.kind == HNSW_VEC_QUANT8) {
            put_f32(b, idx.mins[i]);
            put_f32(b, idx.deltas[i]);
            b.insert(b.end(), &idx.codes[i * idx.dim], &idx.codes[i * idx.dim] + idx.dim);
        } else {
            for (uint32_t j = 0; j < idx.dim; j++) put_f32(b, idx.vals[i * idx.dim + j]);
        }
    }
    if (!write_file(dir + "/points", b)) {
        set_error("Could not write bytes to point file");
        return HNSW_ERR_IO;
    }
    // params
    b.clear();
    put_u64(b, idx.params.m);
    put_u64(b, idx.params.mmax);
    put_u64(b, idx.params.mmax0);
    put_f32(b, idx.params.ml);
    put_u64(b, idx.params.ef_cons);
    put_u64(b, idx.params.dim);
    put_u64(b, idx.params.ep);
    if (!write_file(dir + "/params", b)) {
        set_error("Could not write bytes to params file");
        return HNSW_ERR_IO;
    }
    // layers
    const std::string ldir = dir + "/layers";
    if (mkdir(ldir.c_str(), 0777) != 0) {  // fs::create_dir fails when it exists, template.rs:64
        set_error("Could not create layers dir");
        return HNSW_ERR_IO;
    }
    for (uint32_t l = 0; l < idx.nb_layers(); l++) {
        size_t m = idx.layer_m(l);
        for (NodeID id : idx.layer_nodes[l]) m = std::max(m, idx.row(l, id).size());
        if (m > 0xFFFF || l > 0xFF) {
            set_error("layer %u does not fit the format (m %zu)", l, m);
            return HNSW_ERR_IO;
        }
        b.clear();
        b.push_back((uint8_t)l);
        put_u32(b, (uint32_t)idx.layer_nodes[l].size());
        put_u16(b, (uint16_t)m);
        for (NodeID id : idx.layer_nodes[l]) {
            put_u32(b, id);
            std::vector<NodeID> nb = idx.row(l, id);
            std::sort(nb.begin(), nb.end());
            for (size_t k = 0; k < m; k++) put_u32(b, k < nb.size() ? nb[k] : UINT32_MAX);
        }
        if (!write_file(ldir + "/" + std::to_string(l), b)) {
            set_error("Could not write bytes to layer %u", l);
            return HNSW_ERR_IO;
        }
    }
    return HNSW_OK;
}

int load_index(const std::string &dir, std::unique_ptr<HostIndex> *out) {
    if (!is_dir(dir)) {
        set_error("%s does not exist", dir.c_str());
        return HNSW_ERR_IO;
    }
    std::vector<uint8_t> pb, qb;
    if (!read_file(dir + "/points", &pb) || pb.size() < 16) {
        set_error("Problem reading points file");
        return HNSW_ERR_IO;
    }
    if (!read_file(dir + "/params", &qb) || qb.size() < 52) {
        set_error("Problem reading params file");
        return HNSW_ERR_IO;
    }
    Params p;
    p.m = get_u64(&qb[0]);
    p.mmax = get_u64(&qb[8]);
    p.mmax0 = get_u64(&qb[16]);
    p.ml = get_f32(&qb[24]);
    p.ef_cons = get_u64(&qb[28]);
    p.dim = get_u64(&qb[36]);
    p.ep = (NodeID)get_u64(&qb[44]);
    const uint64_t len = get_u64(&pb[0]), point_size = get_u64(&pb[8]);
    int kind;
    if (point_size == 9 + p.dim)
        kind = HNSW_VEC_QUANT8;
    else if (point_size == 1 + 4 * p.dim)
        kind = HNSW_VEC_F32;
    else {
        set_error("points file: point_size %llu matches neither vector kind for dim %llu",
                  (unsigned long long)point_size, (unsigned long long)p.dim);
        return HNSW_ERR_IO;
    }
    if (pb.size() < 16 + len * point_size || p.m == 0 || p.dim == 0 || len >= 0x7FFFFFFF) {
        set_error("points file truncated or params invalid");
        return HNSW_ERR_IO;
    }
    std::unique_ptr<HostIndex> idx(new HostIndex((uint32_t)p.m, (uint32_t)p.ef_cons,
                                                 (uint32_t)p.dim, kind));
    idx->params = p;
    const uint32_t d = (uint32_t)p.dim;
    idx->levels.resize(len);
    if (kind == HNSW_VEC_QUANT8) {
        idx->codes.resize(len * d);
        idx->mins.resize(len);
        idx->deltas.resize(len);
    } else {
        idx->vals.resize(len * d);
    }
    for (uint64_t i = 0; i < len; i++) {
        const uint8_t *q = &pb[16 + i * point_size];
        idx->levels[i] = q[0];
        if (kind == HNSW_VEC_QUANT8) {
            idx->mins[i] = get_f32(q + 1);
            idx->deltas[i] = get_f32(q + 5);
            memcpy(&idx->codes[i * d], q + 9, d);
        } else {
            for (uint32_t j = 0; j < d; j++) idx->vals[i * d + j] = get_f32(q + 1 + 4 * j);
        }
    }
    // layers: files named by number, sorted numerically, level must equal position (template.rs:103-121)
    const std::string ldir = dir + "/layers";
    DIR *dp = opendir(ldir.c_str());
    if (!dp) {
        set_error("There was a problem reading layers");
        return HNSW_ERR_IO;
    }
    std::vector<uint64_t> files;
    while (struct dirent *e = readdir(dp)) {
        if (e->d_name[0] == '.') continue;
        char *end = nullptr;
        const unsigned long long v = strtoull(e->d_name, &end, 10);
        if (end == e->d_name || *end != 0) {
            closedir(dp);
            set_error("unexpected file %s in layers/", e->d_name);
            return HNSW_ERR_IO;
        }
        files.push_back(v);
    }
    closedir(dp);
    std::sort(files.begin(), files.end());
    // membership: the reference keeps whatever node set each file lists; here a node's layers
    // are implied by its level, so the two must agree
    idx->upper_base.assign(len, UINT32_MAX);
    idx->adj0.resize(len);
    for (uint64_t i = 0; i < len; i++) {
        const uint32_t lv = idx->levels[i];
        while (idx->layer_nodes.size() <= lv) idx->layer_nodes.emplace_back();
        if (lv >= 1) {
            idx->upper_base[i] = (uint32_t)idx->adj_up.size();
            idx->adj_up.resize(idx->adj_up.size() + lv);
        }
        for (uint32_t l = 0; l <= lv; l++) idx->layer_nodes[l].push_back((NodeID)i);
    }
    if (files.size() != idx->layer_nodes.size()) {
        set_error("%zu layer files, the points' levels imply %zu", files.size(),
                  idx->layer_nodes.size());
        return HNSW_ERR_IO;
    }
    for (size_t pos = 0; pos < files.size(); pos++) {
        std::vector<uint8_t> lb;
        if (!read_file(ldir + "/" + std::to_string(files[pos]), &lb) || lb.size() < 7) {
            set_error("Problem reading layer file %llu", (unsigned long long)files[pos]);
            return HNSW_ERR_IO;
        }
        const uint32_t level = lb[0], nb_nodes = get_u32(&lb[1]);
        const size_t m = ((size_t)lb[5] << 8) | lb[6];
        if (level != pos) {  // assert_eq!(layers.len(), layer.level), template.rs:119
            set_error("layer file %llu holds level %u", (unsigned long long)files[pos], level);
            return HNSW_ERR_IO;
        }
        if (lb.size() != 7 + (size_t)nb_nodes * 4 * (m + 1) ||
            nb_nodes != idx->layer_nodes[pos].size()) {
            set_error("layer %u: size mismatch (a degree > m row corrupts the reference's format)",
                      level);
            return HNSW_ERR_IO;
        }
        size_t off = 7;
        for (uint32_t i = 0; i < nb_nodes; i++) {
            const NodeID id = get_u32(&lb[off]);
            off += 4;
            if (!idx->in_layer(level, id)) {
                set_error("layer %u lists node %u whose level is lower", level, id);
                return HNSW_ERR_IO;
            }
            std::vector<NodeID> &r = idx->row(level, id);
            for (size_t k = 0; k < m; k++) {
                const NodeID n = get_u32(&lb[off + 4 * k]);
                if (n == UINT32_MAX) break;  // graph.rs:192: stop at the first sentinel
                // the search kernels index the row arrays with what is stored here
                if (!idx->in_layer(level, n) || n == id) {
                    set_error("layer %u: node %u lists neighbour %u, which is not a node of that layer", level, id, n);
                    return HNSW_ERR_IO;
                }
                if (std::find(r.begin(), r.end(), n) == r.end()) r.push_back(n);
            }
            off += 4 * m;
        }
    }
    // the entry point must be a stored point of the top layer (params.rs:13, template.rs:283-290), and the
    // caps must be the ones the layers were built with (params.rs:20-42): both size device structures
    if (len > 0 && (p.ep >= len || idx->levels[p.ep] + 1u != idx->layer_nodes.size())) {
        set_error("params: entry point %u is not a node of the top layer", p.ep);
        return HNSW_ERR_IO;
    }
    // Params::from accepts any mmax / mmax0, also below m (params.rs:44-62: they are only read by
    // assert_param_compliance); refused here is only what would mis-size a device structure: a cap beyond the
    // layer files' 16-bit row width, and ef_cons = 0 (the build searches with it)
    if (p.mmax > 65535 || p.mmax0 > 65535 || p.ef_cons == 0) {
        set_error("params: mmax %llu / mmax0 %llu beyond the 16-bit row width of the layer files, or ef_cons %llu = 0",
                  (unsigned long long)p.mmax, (unsigned long long)p.mmax0, (unsigned long long)p.ef_cons);
        return HNSW_ERR_IO;
    }
    idx->version = 1;
    *out = std::move(idx);
    return HNSW_OK;
}

}  // namespace hx
