// device_index.cpp -- packs a HostIndex into the HBM layout described in device_index.h and
// uploads it.  Host code only (HIP runtime API); the kernels are in search_kernels.hip.

#include "device_index.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hx {

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s", #expr, hipGetErrorString(e_));                    \
            return e_ == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP;              \
        }                                                                                \
    } while (0)

uint32_t quant_half_bytes(uint32_t dim) {
    const uint32_t need = 8 + 4 * (dim / 8) + (dim % 8);
    return (need + 15) & ~15u;
}
uint32_t f32_row_stride(uint32_t dim) { return (4 * dim + 15) & ~15u; }
uint32_t adj_stride(uint64_t cap, uint32_t min_slots) {
    uint32_t s = min_slots;
    while (s < cap) s <<= 1;
    return s;
}

void DeviceIndex::release() {
    if (device >= 0) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        for (void *&b : bufs_) {
            if (b) (void)hipFree(b);
            b = nullptr;
        }
        if (pin_) (void)hipHostFree(pin_);
        if (stage_) (void)hipFree(stage_);
        if (pstream_) (void)hipStreamDestroy(pstream_);
        pin_ = stage_ = nullptr;
        pstream_ = nullptr;
        pin_cap_ = stage_cap_ = 0;
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    valid = false;
    replica = false;
    bytes = 0;
    n_ovf_nbrs_ = 0;
    for (uint64_t &z : sizes_) z = 0;
    for (uint64_t &z : caps_) z = 0;
}

void DeviceIndex::describe(uint64_t (&nbytes)[7], void *(&ptrs)[7]) const {
    for (int i = 0; i < 7; i++) {
        nbytes[i] = bufs_[i] ? sizes_[i] : 0;
        ptrs[i] = bufs_[i];
    }
}

int DeviceIndex::adopt_alloc(int dev, const uint64_t (&nbytes)[7], void *(&ptrs)[7]) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        set_error("no HIP device available (search runs on the GPU only)");
        return HNSW_ERR_NO_DEVICE;
    }
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    release();
    HIP_TRY(hipSetDevice(dev));
    device = dev;
    bytes = 0;
    for (int i = 0; i < 7; i++) {
        ptrs[i] = nullptr;
        if (nbytes[i] == 0) continue;
        HIP_TRY(hipMalloc(&bufs_[i], nbytes[i]));
        sizes_[i] = caps_[i] = nbytes[i];
        ptrs[i] = bufs_[i];
        bytes += nbytes[i];
    }
    return HNSW_OK;
}

void DeviceIndex::adopt_commit(const DevView &scalars) {
    DevView v = scalars;
    v.rows = (const uint8_t *)bufs_[0];
    v.adj0 = (const uint32_t *)bufs_[1];
    v.adj_up = (const uint32_t *)bufs_[2];
    v.upper_base = (const uint32_t *)bufs_[3];
    v.ovf_off = (const uint32_t *)bufs_[4];
    v.ovf_nbrs = (const uint32_t *)bufs_[5];
    v.fat = (const uint8_t *)bufs_[6];
    if (!bufs_[6]) v.fat_stride = 0;
    view = v;
    replica = true;
    valid = true;
}

template <class F>
static void parallel_rows(uint64_t n, F f) {
    unsigned nt = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 65536) nt = 1;
    if (nt == 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(f, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
}

// inline rows ("fat" layer-0 blocks, device_index.h): only small 8-bit indexes have them
bool DeviceIndex::wants_inline_rows(const HostIndex &idx) const {
    if (idx.kind != HNSW_VEC_QUANT8) return false;
    int want = inline_rows;
    if (const char *e = getenv("HNSW_MI355X_INLINE_ROWS")) want = atoi(e);
    const uint32_t half = quant_half_bytes(idx.dim), S0 = adj_stride(idx.layer_m(0), 32);
    const uint32_t used = 8 + 4 * (idx.dim / 8) + idx.dim % 8;
    const bool room = S0 == 32 && half >= used + 4;
    // d = 100 is served by the lean compact-layout kernel (search_lean.hip), faster at every launch size
    // than the inline-rows loop: the 4-GB copy is only built there when asked for explicitly
    if (want < 0 && idx.dim == 100 && !(getenv("HNSW_MI355X_LEAN_Q8") && atoi(getenv("HNSW_MI355X_LEAN_Q8")) == 0) &&
        !(getenv("HNSW_MI355X_LEAN") && atoi(getenv("HNSW_MI355X_LEAN")) == 0))
        want = 0;
    const uint64_t need = (uint64_t)idx.len() * S0 * (2ull * half);
    return room && (want == 1 || (want < 0 && need <= fat_budget_bytes));
}

// sorted ids into the row's S slots; true: the row has more than S ids, its last slot is left for the
// overflow pointer
static bool pack_adj_row(const std::vector<NodeID> &src, uint32_t *dst, uint32_t S, std::vector<NodeID> &tmp) {
    tmp = src;
    std::sort(tmp.begin(), tmp.end());
    const size_t take = tmp.size() <= S ? tmp.size() : S - 1;
    std::copy(tmp.begin(), tmp.begin() + take, dst);
    std::fill(dst + take, dst + S, HX_EMPTY_SLOT);
    return tmp.size() > S;
}

bool DeviceIndex::refresh_rows(const HostIndex &idx, const std::vector<uint64_t> &layer_row) {
    if (!valid || replica || view.n_points != idx.len() || view.nb_layers != idx.nb_layers()) return false;
    if (wants_inline_rows(idx) || view.fat_stride != 0) return false;
    if (sizes_[4] != 4) return false;  // overflow lists already there: a fresh upload rebuilds them
    if (hipSetDevice(device) != hipSuccess) return false;
    // rows in the order upload() files overflow lists: layer 0 by id, then the upper rows by row index
    std::vector<std::pair<uint64_t, uint32_t>> rows;  // (row index, 0 = adj0 / 1 = adj_up)
    for (uint64_t key : layer_row) {
        const uint32_t layer = (uint32_t)(key >> 32);
        const NodeID id = (NodeID)key;
        if (!idx.in_layer(layer, id)) return false;
        rows.push_back(layer == 0 ? std::make_pair((uint64_t)id, 0u)
                                  : std::make_pair((uint64_t)idx.upper_base[id] + layer - 1, 1u));
    }
    std::sort(rows.begin(), rows.end(), [](const auto &x, const auto &y) {
        return x.second != y.second ? x.second < y.second : x.first < y.first;
    });
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    std::vector<uint32_t> ovf_off(1, 0), ovf_nbrs, slots;
    std::vector<NodeID> tmp;
    for (const auto &r : rows) {
        const bool up = r.second != 0;
        const uint32_t S = up ? view.S1 : view.S0;
        const std::vector<NodeID> &src = up ? idx.adj_up[r.first] : idx.adj0[r.first];
        slots.assign(S, HX_EMPTY_SLOT);
        if (pack_adj_row(src, slots.data(), S, tmp)) {
            slots[S - 1] = HX_OVF_FLAG | (uint32_t)(ovf_off.size() - 1);
            ovf_nbrs.insert(ovf_nbrs.end(), tmp.begin() + (S - 1), tmp.end());
            ovf_off.push_back((uint32_t)ovf_nbrs.size());
        }
        uint32_t *dst = static_cast<uint32_t *>(bufs_[up ? 2 : 1]) + r.first * S;
        if (hipMemcpy(dst, slots.data(), (size_t)S * 4, hipMemcpyHostToDevice) != hipSuccess) {
            valid = false;  // half-written: the next search uploads
            return false;
        }
    }
    if (!ovf_nbrs.empty()) {
        void *off = nullptr, *nb = nullptr;
        if (hipMalloc(&off, ovf_off.size() * 4) != hipSuccess || hipMalloc(&nb, ovf_nbrs.size() * 4) != hipSuccess ||
            hipMemcpy(off, ovf_off.data(), ovf_off.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(nb, ovf_nbrs.data(), ovf_nbrs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (off) (void)hipFree(off);
            if (nb) (void)hipFree(nb);
            valid = false;
            return false;
        }
        (void)hipFree(bufs_[4]);
        (void)hipFree(bufs_[5]);
        bytes += ovf_off.size() * 4 + ovf_nbrs.size() * 4 - sizes_[4] - sizes_[5];
        bufs_[4] = off;
        bufs_[5] = nb;
        sizes_[4] = caps_[4] = ovf_off.size() * 4;
        sizes_[5] = caps_[5] = ovf_nbrs.size() * 4;
        n_ovf_nbrs_ = ovf_nbrs.size();
        view.ovf_off = (const uint32_t *)off;
        view.ovf_nbrs = (const uint32_t *)nb;
    }
    view.ep = idx.params.ep;
    version_seen = idx.version;
    return true;
}

// array i to at least need_bytes, keeping its contents; what is added is filled with fill_byte (0xFF = empty
// adjacency slots / "no upper rows"), so that rows handed out later start empty
bool DeviceIndex::grow(int i, uint64_t need_bytes, int fill_byte) {
    if (need_bytes <= caps_[i]) return true;
    const uint64_t cap = std::max<uint64_t>(need_bytes, caps_[i] + caps_[i] / 8 + 4096);
    void *nb = nullptr;
    if (hipMalloc(&nb, cap) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // on the patch stream, and finished before the old array goes (a memset on the null stream would not be
    // ordered against the patch kernel that follows on pstream_)
    bool ok = true;
    if (bufs_[i] && sizes_[i]) ok = hipMemcpyAsync(nb, bufs_[i], sizes_[i], hipMemcpyDeviceToDevice, pstream_) == hipSuccess;
    ok = ok && hipMemsetAsync(static_cast<unsigned char *>(nb) + sizes_[i], fill_byte, cap - sizes_[i], pstream_) == hipSuccess;
    ok = ok && hipStreamSynchronize(pstream_) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)hipFree(nb);
        return false;
    }
    if (bufs_[i]) (void)hipFree(bufs_[i]);
    bytes += cap - caps_[i];
    bufs_[i] = nb;
    caps_[i] = cap;
    return true;
}

bool DeviceIndex::append_point(const HostIndex &idx, NodeID id, const std::vector<uint64_t> &layer_row) {
    const uint64_t N = idx.len();
    if (!valid || replica || N == 0 || id != N - 1 || view.n_points + 1 != N || idx.kind != view.kind ||
        idx.dim != view.dim || idx.nb_layers() < view.nb_layers)
        return false;
    if (view.S0 != adj_stride(idx.layer_m(0), 32) || view.S1 != adj_stride(idx.params.m, 8)) return false;
    if (hipSetDevice(device) != hipSuccess) return false;
    const bool q8 = idx.kind == HNSW_VEC_QUANT8, fat = view.fat_stride != 0 && bufs_[6];
    const uint32_t S0 = view.S0, S1 = view.S1;

    // ---- what is written, as (array, byte offset, words) over one staging buffer ----
    struct Piece {
        int array;
        uint64_t offset;
        uint32_t src_word, n_words;
    };
    std::vector<Piece> pieces;
    std::vector<uint32_t> st;  // the staging words
    auto add = [&](int array, uint64_t offset, uint32_t n_words) -> uint32_t * {
        pieces.push_back(Piece{array, offset, (uint32_t)st.size(), n_words});
        st.resize(st.size() + n_words, 0);
        return &st[st.size() - n_words];
    };
    {  // the vector row, in the layout of upload()
        uint8_t *o = reinterpret_cast<uint8_t *>(add(0, (uint64_t)id * view.row_stride, view.row_stride / 4));
        if (q8) {
            const uint32_t d = idx.dim, nch = d / 8, rem = d % 8, half = view.half_bytes;
            const uint8_t *c = &idx.codes[(size_t)id * d];
            for (uint32_t h = 0; h < 2; h++) {
                uint8_t *oh = o + h * half;
                memcpy(oh, &idx.mins[id], 4);
                memcpy(oh + 4, &idx.deltas[id], 4);
                for (uint32_t ch = 0; ch < nch; ch++) memcpy(oh + 8 + 4 * ch, c + 8 * ch + 4 * h, 4);
                if (h == 0) memcpy(oh + 8 + 4 * nch, c + 8 * nch, rem);
            }
        } else {
            memcpy(o, &idx.vals[(size_t)id * idx.dim], 4 * (size_t)idx.dim);
        }
    }
    *add(3, (uint64_t)id * 4, 1) = idx.upper_base[id];
    // the touched adjacency rows and the new point's own (a row without edges stays empty, but it must exist)
    std::vector<std::pair<uint64_t, uint32_t>> rows;  // (row index, 0 = adj0 / 1 = adj_up)
    for (uint64_t key : layer_row) {
        const uint32_t layer = (uint32_t)(key >> 32);
        const NodeID node = (NodeID)key;
        if (!idx.in_layer(layer, node)) return false;
        rows.push_back(layer == 0 ? std::make_pair((uint64_t)node, 0u)
                                  : std::make_pair((uint64_t)idx.upper_base[node] + layer - 1, 1u));
    }
    for (uint32_t l = 0; l <= idx.levels[id]; l++)
        rows.push_back(l == 0 ? std::make_pair((uint64_t)id, 0u) : std::make_pair((uint64_t)idx.upper_base[id] + l - 1, 1u));
    std::sort(rows.begin(), rows.end(), [](const auto &x, const auto &y) {
        return x.second != y.second ? x.second < y.second : x.first < y.first;
    });
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    uint64_t n_ovf_rows = sizes_[4] / 4 - 1, n_ovf_nbrs = n_ovf_nbrs_;
    std::vector<NodeID> tmp;
    std::vector<uint32_t> fat_nodes;
    for (const auto &r : rows) {
        const bool up = r.second != 0;
        const uint32_t S = up ? S1 : S0;
        const std::vector<NodeID> &src = up ? idx.adj_up[r.first] : idx.adj0[r.first];
        uint32_t *slots = add(up ? 2 : 1, r.first * S * 4ull, S);
        if (pack_adj_row(src, slots, S, tmp)) {  // more ids than slots: the rest goes to a NEW overflow list
            slots[S - 1] = HX_OVF_FLAG | (uint32_t)n_ovf_rows;
            const uint32_t extra = (uint32_t)(tmp.size() - (S - 1));
            uint32_t *nb = add(5, n_ovf_nbrs * 4, extra);  // (`slots` is stale from here on: st may have moved)
            std::copy(tmp.begin() + (S - 1), tmp.end(), nb);
            n_ovf_nbrs += extra;
            n_ovf_rows++;
            *add(4, n_ovf_rows * 4, 1) = (uint32_t)n_ovf_nbrs;
        }
        if (!up) fat_nodes.push_back((uint32_t)r.first);
    }
    const uint32_t n_fat = fat ? (uint32_t)fat_nodes.size() : 0;
    const uint32_t fat_word = (uint32_t)st.size();
    if (n_fat) st.insert(st.end(), fat_nodes.begin(), fat_nodes.end());

    // ---- room in the arrays ----
    const uint64_t need[7] = {N * view.row_stride,
                              N * S0 * 4ull,
                              std::max<uint64_t>(1, idx.adj_up.size()) * S1 * 4ull,
                              N * 4ull,
                              (n_ovf_rows + 1) * 4ull,
                              std::max<uint64_t>(1, n_ovf_nbrs) * 4ull,
                              fat ? N * view.fat_stride : 0};
    const int fill[7] = {0, 0xFF, 0xFF, 0xFF, 0, 0xFF, 0};
    if (!pstream_ && hipStreamCreateWithFlags(&pstream_, hipStreamNonBlocking) != hipSuccess) return false;
    for (int i = 0; i < 7; i++)
        if (need[i] && !grow(i, need[i], fill[i])) {
            valid = false;  // (an array that did grow has moved: the view's pointers are no longer all good -- the next search uploads)
            return false;
        }

    // ---- staging: [descriptors | words] through one pinned buffer and one copy ----
    const size_t desc_bytes = (pieces.size() * sizeof(PatchDesc) + 255) & ~(size_t)255;
    const size_t total = desc_bytes + st.size() * 4;
    if (pin_cap_ < total) {
        if (pin_) (void)hipHostFree(pin_);
        if (stage_) (void)hipFree(stage_);
        pin_ = stage_ = nullptr;
        pin_cap_ = stage_cap_ = 0;
        const size_t cap = std::max<size_t>(2 * total, 64 << 10);
        if (hipHostMalloc(&pin_, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&stage_, cap) != hipSuccess) {
            (void)hipGetLastError();
            if (pin_) (void)hipHostFree(pin_);
            if (stage_) (void)hipFree(stage_);
            pin_ = stage_ = nullptr;
            return false;
        }
        pin_cap_ = stage_cap_ = cap;
    }
    PatchDesc *pd = static_cast<PatchDesc *>(pin_);
    for (size_t i = 0; i < pieces.size(); i++) {
        const Piece &pc = pieces[i];
        pd[i].dst = static_cast<unsigned char *>(bufs_[pc.array]) + pc.offset;
        pd[i].src_word = pc.src_word;
        pd[i].n_words = pc.n_words;
    }
    memcpy(static_cast<unsigned char *>(pin_) + desc_bytes, st.data(), st.size() * 4);
    const uint32_t *d_words = reinterpret_cast<const uint32_t *>(static_cast<unsigned char *>(stage_) + desc_bytes);
    DevView v = view;
    v.rows = (const uint8_t *)bufs_[0];
    v.adj0 = (const uint32_t *)bufs_[1];
    v.adj_up = (const uint32_t *)bufs_[2];
    v.upper_base = (const uint32_t *)bufs_[3];
    v.ovf_off = (const uint32_t *)bufs_[4];
    v.ovf_nbrs = (const uint32_t *)bufs_[5];
    v.fat = (const uint8_t *)bufs_[6];
    v.n_points = (uint32_t)N;
    v.nb_layers = idx.nb_layers();
    v.ep = idx.params.ep;
    bool ok = hipMemcpyAsync(stage_, pin_, total, hipMemcpyHostToDevice, pstream_) == hipSuccess &&
              launch_patch(static_cast<const PatchDesc *>(stage_), d_words, (uint32_t)pieces.size(), pstream_) == HNSW_OK &&
              (!n_fat || launch_fat_rebuild(v, static_cast<uint8_t *>(bufs_[6]), d_words + fat_word, n_fat, pstream_) == HNSW_OK) &&
              hipStreamSynchronize(pstream_) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        valid = false;  // half-written: the next search uploads
        return false;
    }
    for (int i = 0; i < 7; i++)
        if (need[i]) sizes_[i] = need[i];
    n_ovf_nbrs_ = n_ovf_nbrs;
    view = v;
    version_seen = idx.version;
    return true;
}

// [lo, hi) on up to 16 threads
template <class F>
static void parallel_span(uint64_t lo, uint64_t hi, F f) {
    const uint64_t n = hi - lo;
    unsigned nt = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 16384) nt = 1;
    if (nt == 1) {
        f(lo, hi);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(f, lo + n * t / nt, lo + n * (t + 1) / nt);
    for (auto &t : th) t.join();
}

// Host rows -> device array in pieces through two pinned buffers: piece k + 1 is packed by the host threads
// while piece k is on the wire.  (A pageable hipMemcpy of a row table of tens of GB runs at a fraction of the
// link, and packing the whole table first doubles the host memory and adds a pass over it.)
// fill(row_lo, row_hi, dst): writes rows [row_lo, row_hi) of `unit` bytes each.
namespace {
constexpr size_t PIECE_BYTES = 64ull << 20;
inline size_t piece_limit() {  // (HNSW_MI355X_UPLOAD_PIECE_MB: A/B runs of the staging piece size)
    static const size_t v = [] {
        const char *e = getenv("HNSW_MI355X_UPLOAD_PIECE_MB");
        const long mb = e ? atol(e) : 0;
        return mb > 0 ? (size_t)mb << 20 : PIECE_BYTES;
    }();
    return v;
}
struct PinnedPair {
    size_t piece_bytes;  // of each buffer: 64 MiB, less for a small index (pinning memory costs time too)
    explicit PinnedPair(size_t largest_array) : piece_bytes(std::max<size_t>(4096, std::min(piece_limit(), largest_array))) {}
    void *buf[2] = {nullptr, nullptr};
    bool pinned = true;  // false: the host refused to pin memory (a small memlock limit); plain buffers, blocking copies
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    ~PinnedPair() {
        for (int i = 0; i < 2; i++) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (buf[i]) {
                if (pinned)
                    (void)hipHostFree(buf[i]);
                else
                    free(buf[i]);
            }
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};
}  // namespace

template <class Fill>
static int upload_pieces(PinnedPair &pp, void *dst, uint64_t n_rows, size_t unit, Fill fill) {
    if (!pp.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&pp.ev[i], hipEventDisableTiming));
        const bool refuse = getenv("HNSW_MI355X_NO_PINNED") && atoi(getenv("HNSW_MI355X_NO_PINNED")) != 0;  // (tests)
        if (refuse || hipHostMalloc(&pp.buf[0], pp.piece_bytes, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&pp.buf[1], pp.piece_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (pp.buf[0]) (void)hipHostFree(pp.buf[0]);
            pp.pinned = false;
            pp.buf[0] = malloc(pp.piece_bytes);
            pp.buf[1] = nullptr;
            if (!pp.buf[0]) {
                set_error("upload: no memory for a %zu-byte staging buffer", pp.piece_bytes);
                return HNSW_ERR_OOM;
            }
        }
    }
    if (!pp.pinned) {  // one plain buffer, one blocking copy per piece
        const uint64_t per1 = std::max<uint64_t>(1, pp.piece_bytes / unit);
        for (uint64_t lo = 0; lo < n_rows; lo += per1) {
            const uint64_t hi = std::min(n_rows, lo + per1);
            fill(lo, hi, static_cast<unsigned char *>(pp.buf[0]));
            HIP_TRY(hipMemcpy(static_cast<unsigned char *>(dst) + lo * unit, pp.buf[0], (hi - lo) * unit, hipMemcpyHostToDevice));
        }
        return HNSW_OK;
    }
    const uint64_t per = std::max<uint64_t>(1, pp.piece_bytes / unit);
    int k = 0;
    for (uint64_t lo = 0; lo < n_rows; lo += per, k ^= 1) {
        const uint64_t hi = std::min(n_rows, lo + per);
        HIP_TRY(hipEventSynchronize(pp.ev[k]));  // the copy that last read this buffer
        fill(lo, hi, static_cast<unsigned char *>(pp.buf[k]));
        HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(dst) + lo * unit, pp.buf[k], (hi - lo) * unit,
                               hipMemcpyHostToDevice, pp.stream));
        HIP_TRY(hipEventRecord(pp.ev[k], pp.stream));
    }
    HIP_TRY(hipStreamSynchronize(pp.stream));
    return HNSW_OK;
}

int DeviceIndex::read_adjacency(int which, uint64_t n_rows,
                                const std::function<void(uint64_t, uint64_t, const uint32_t *)> &consume) {
    if (!valid || n_rows == 0) return HNSW_OK;
    const uint32_t S = which == 0 ? view.S0 : view.S1;
    const size_t unit = (size_t)S * 4;
    const unsigned char *src = static_cast<const unsigned char *>(bufs_[which == 0 ? 1 : 2]);
    HIP_TRY(hipSetDevice(device));
    PinnedPair pp(n_rows * unit);
    HIP_TRY(hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&pp.ev[i], hipEventDisableTiming));
    const bool refuse = getenv("HNSW_MI355X_NO_PINNED") && atoi(getenv("HNSW_MI355X_NO_PINNED")) != 0;
    if (refuse || hipHostMalloc(&pp.buf[0], pp.piece_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&pp.buf[1], pp.piece_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (pp.buf[0]) (void)hipHostFree(pp.buf[0]);
        pp.pinned = false;
        pp.buf[0] = malloc(pp.piece_bytes);
        pp.buf[1] = nullptr;
        if (!pp.buf[0]) {
            set_error("read-back: no memory for a %zu-byte staging buffer", pp.piece_bytes);
            return HNSW_ERR_OOM;
        }
    }
    const uint64_t per = std::max<uint64_t>(1, pp.piece_bytes / unit);
    if (!pp.pinned) {
        for (uint64_t lo = 0; lo < n_rows; lo += per) {
            const uint64_t hi = std::min(n_rows, lo + per);
            HIP_TRY(hipMemcpy(pp.buf[0], src + lo * unit, (hi - lo) * unit, hipMemcpyDeviceToHost));
            consume(lo, hi, static_cast<const uint32_t *>(pp.buf[0]));
        }
        return HNSW_OK;
    }
    // piece k + 1 is on the wire while the host consumes piece k
    auto request = [&](uint64_t lo, int k) -> int {
        const uint64_t hi = std::min(n_rows, lo + per);
        HIP_TRY(hipMemcpyAsync(pp.buf[k], src + lo * unit, (hi - lo) * unit, hipMemcpyDeviceToHost, pp.stream));
        HIP_TRY(hipEventRecord(pp.ev[k], pp.stream));
        return HNSW_OK;
    };
    int rc = request(0, 0), k = 0;
    if (rc != HNSW_OK) return rc;
    for (uint64_t lo = 0; lo < n_rows; lo += per, k ^= 1) {
        const uint64_t hi = std::min(n_rows, lo + per);
        if (hi < n_rows && (rc = request(hi, k ^ 1)) != HNSW_OK) return rc;
        HIP_TRY(hipEventSynchronize(pp.ev[k]));
        consume(lo, hi, static_cast<const uint32_t *>(pp.buf[k]));
    }
    return HNSW_OK;
}

int DeviceIndex::upload(const HostIndex &idx, int dev, const std::function<int()> &meanwhile) {
    const uint64_t N = idx.len();
    if (N == 0) {
        set_error("index is empty");
        return HNSW_ERR_EMPTY;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        set_error("no HIP device available (search runs on the GPU only)");
        return HNSW_ERR_NO_DEVICE;
    }
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    release();
    HIP_TRY(hipSetDevice(dev));
    device = dev;

    // the caller's side job: started now, joined before anything but the vector rows is read from idx (and on every
    // way out of this function)
    struct Side {
        std::thread t;
        int rc = HNSW_OK;
        std::string err;
        void join() {
            if (t.joinable()) t.join();
        }
        ~Side() { join(); }
    } side;
    if (meanwhile)
        side.t = std::thread([&side, &meanwhile] {
            side.rc = meanwhile();
            if (side.rc != HNSW_OK) side.err = get_error();
        });

    DevView v{};
    v.kind = idx.kind;
    v.n_points = (uint32_t)N;
    v.dim = idx.dim;
    v.nch4 = 4 * (idx.dim / 8);
    v.rem = idx.dim % 8;
    const bool q8 = idx.kind == HNSW_VEC_QUANT8;
    v.half_bytes = q8 ? quant_half_bytes(idx.dim) : 0;
    v.row_stride = q8 ? 2 * v.half_bytes : f32_row_stride(idx.dim);
    v.S0 = adj_stride(idx.layer_m(0), 32);
    v.S1 = adj_stride(idx.params.m, 8);

    // ---- one vector row / one adjacency row in the device layout ----
    auto pack_vector = [&](uint64_t i, uint8_t *o) {
        if (q8) {
            const uint32_t d = idx.dim, nch = d / 8, rem = d % 8, half = v.half_bytes;
            const uint8_t *c = &idx.codes[i * d];
            memset(o, 0, v.row_stride);
            for (uint32_t h = 0; h < 2; h++) {
                uint8_t *oh = o + h * half;
                memcpy(oh, &idx.mins[i], 4);
                memcpy(oh + 4, &idx.deltas[i], 4);
                for (uint32_t ch = 0; ch < nch; ch++) memcpy(oh + 8 + 4 * ch, c + 8 * ch + 4 * h, 4);
                if (h == 0) memcpy(oh + 8 + 4 * nch, c + 8 * nch, rem);
            }
        } else {
            const size_t nb = 4 * (size_t)idx.dim;
            memcpy(o, &idx.vals[i * idx.dim], nb);
            if (nb < v.row_stride) memset(o + nb, 0, v.row_stride - nb);
        }
    };

    // ---- inline rows: those (small) indexes also keep whole-table host copies to build the blocks from ----
    const bool want_fat = wants_inline_rows(idx);

    bytes = 0;
    auto dev_alloc = [&](int i, size_t nbytes) -> int {
        HIP_TRY(hipMalloc(&bufs_[i], nbytes));
        sizes_[i] = caps_[i] = nbytes;
        bytes += nbytes;
        return HNSW_OK;
    };
    int rc;
    std::vector<uint64_t> over0;  // layer-0 rows with more than S0 ids (rare)
    std::vector<uint8_t> rows_h;  // whole-table host copies: only when the inline rows are built from them
    std::vector<uint32_t> adj0_h;
    if ((rc = dev_alloc(0, (size_t)N * v.row_stride)) != HNSW_OK) return rc;
    if ((rc = dev_alloc(1, (size_t)N * v.S0 * 4)) != HNSW_OK) return rc;
    {
        PinnedPair pp(std::max((size_t)N * v.row_stride, (size_t)N * v.S0 * 4));
        if (want_fat) rows_h.resize((size_t)N * v.row_stride);
        rc = upload_pieces(pp, bufs_[0], N, v.row_stride, [&](uint64_t lo, uint64_t hi, unsigned char *dst) {
            parallel_span(lo, hi, [&](uint64_t a, uint64_t b) {
                for (uint64_t i = a; i < b; i++) pack_vector(i, dst + (i - lo) * v.row_stride);
            });
            if (want_fat) memcpy(&rows_h[lo * v.row_stride], dst, (hi - lo) * v.row_stride);
        });
        if (rc != HNSW_OK) return rc;
        side.join();
        if (side.rc != HNSW_OK) {
            set_error("%s", side.err.c_str());
            return side.rc;
        }
        v.nb_layers = idx.nb_layers();
        v.ep = idx.params.ep;
        if (want_fat) adj0_h.resize((size_t)N * v.S0);
        std::mutex over_mu;
        rc = upload_pieces(pp, bufs_[1], N, (size_t)v.S0 * 4, [&](uint64_t lo, uint64_t hi, unsigned char *dst) {
            parallel_span(lo, hi, [&](uint64_t a, uint64_t b) {
                std::vector<NodeID> tmp;
                std::vector<uint64_t> mine;
                for (uint64_t i = a; i < b; i++)
                    if (pack_adj_row(idx.adj0[i], reinterpret_cast<uint32_t *>(dst) + (i - lo) * v.S0, v.S0, tmp)) mine.push_back(i);
                if (!mine.empty()) {
                    std::lock_guard<std::mutex> g(over_mu);
                    over0.insert(over0.end(), mine.begin(), mine.end());
                }
            });
            if (want_fat) memcpy(&adj0_h[lo * v.S0], dst, (hi - lo) * (size_t)v.S0 * 4);
        });
        if (rc != HNSW_OK) return rc;
    }
    std::sort(over0.begin(), over0.end());

    // ---- upper layers, overflow lists (small) ----
    std::vector<uint32_t> adj_up(std::max<size_t>(1, idx.adj_up.size()) * v.S1, HX_EMPTY_SLOT);
    std::vector<uint32_t> ovf_off(1, 0), ovf_nbrs;
    auto overflow_slot = [&](const std::vector<NodeID> &src, uint32_t S) -> uint32_t {
        std::vector<NodeID> tmp = src;
        std::sort(tmp.begin(), tmp.end());
        const uint32_t slot = HX_OVF_FLAG | (uint32_t)(ovf_off.size() - 1);
        ovf_nbrs.insert(ovf_nbrs.end(), tmp.begin() + (S - 1), tmp.end());
        ovf_off.push_back((uint32_t)ovf_nbrs.size());
        return slot;
    };
    for (uint64_t i : over0) {
        const uint32_t slot = overflow_slot(idx.adj0[i], v.S0);
        HIP_TRY(hipMemcpy(static_cast<uint32_t *>(bufs_[1]) + i * v.S0 + (v.S0 - 1), &slot, 4, hipMemcpyHostToDevice));
        if (want_fat) adj0_h[i * v.S0 + (v.S0 - 1)] = slot;
    }
    {
        std::vector<NodeID> tmp;
        for (size_t r = 0; r < idx.adj_up.size(); r++)
            if (pack_adj_row(idx.adj_up[r], &adj_up[r * v.S1], v.S1, tmp))
                adj_up[r * v.S1 + (v.S1 - 1)] = overflow_slot(idx.adj_up[r], v.S1);
    }
    n_ovf_nbrs_ = ovf_nbrs.size();
    if (ovf_nbrs.empty()) ovf_nbrs.push_back(HX_EMPTY_SLOT);
    std::vector<uint32_t> ub(idx.upper_base.begin(), idx.upper_base.end());
    ub.resize(N, UINT32_MAX);

    std::vector<uint8_t> fat;
    if (want_fat) {
        v.fat_stride = (uint64_t)v.S0 * v.row_stride;
        fat.assign((uint64_t)N * v.fat_stride, 0);
        const uint32_t idpos = v.half_bytes - 4;
        parallel_rows(N, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                for (uint32_t k = 0; k < v.S0; k++) {
                    const uint32_t nb = adj0_h[i * v.S0 + k];
                    uint8_t *o = &fat[i * v.fat_stride + (uint64_t)k * v.row_stride];
                    if (nb != HX_EMPTY_SLOT && !(nb & HX_OVF_FLAG)) memcpy(o, &rows_h[(size_t)nb * v.row_stride], v.row_stride);
                    memcpy(o + idpos, &nb, 4);
                }
            }
        });
    }

    struct Up {
        const void *src;
        size_t nbytes;
    } ups[7] = {{nullptr, 0},
                {nullptr, 0},
                {adj_up.data(), adj_up.size() * 4},
                {ub.data(), ub.size() * 4},
                {ovf_off.data(), ovf_off.size() * 4},
                {ovf_nbrs.data(), ovf_nbrs.size() * 4},
                {fat.data(), fat.size()}};
    for (int i = 2; i < 7; i++) {
        if (ups[i].nbytes == 0) continue;
        if (i == 6) {  // the inline-rows copy is optional: without room for it the compact path serves
            if (hipMalloc(&bufs_[i], ups[i].nbytes) != hipSuccess) {
                (void)hipGetLastError();
                bufs_[i] = nullptr;
                v.fat_stride = 0;
                continue;
            }
        } else {
            HIP_TRY(hipMalloc(&bufs_[i], ups[i].nbytes));
        }
        HIP_TRY(hipMemcpy(bufs_[i], ups[i].src, ups[i].nbytes, hipMemcpyHostToDevice));
        sizes_[i] = caps_[i] = ups[i].nbytes;
        bytes += ups[i].nbytes;
    }
    v.rows = (const uint8_t *)bufs_[0];
    v.adj0 = (const uint32_t *)bufs_[1];
    v.adj_up = (const uint32_t *)bufs_[2];
    v.upper_base = (const uint32_t *)bufs_[3];
    v.ovf_off = (const uint32_t *)bufs_[4];
    v.ovf_nbrs = (const uint32_t *)bufs_[5];
    v.fat = (const uint8_t *)bufs_[6];
    view = v;
    version_seen = idx.version;
    valid = true;
    return HNSW_OK;
}

}  // namespace hx
