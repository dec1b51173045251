// device_index.cpp -- packs a HostIndex into the HBM layout described in device_index.h and
// uploads it.  Host code only (HIP runtime API); the kernels are in search_kernels.hip.

#include "device_index.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
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
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    valid = false;
    replica = false;
    bytes = 0;
    for (uint64_t &z : sizes_) z = 0;
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
        sizes_[i] = nbytes[i];
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

int DeviceIndex::upload(const HostIndex &idx, int dev) {
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

    DevView v{};
    v.kind = idx.kind;
    v.n_points = (uint32_t)N;
    v.dim = idx.dim;
    v.nch4 = 4 * (idx.dim / 8);
    v.rem = idx.dim % 8;
    v.nb_layers = idx.nb_layers();
    v.ep = idx.params.ep;

    // ---- vector rows ----
    std::vector<uint8_t> rows;
    if (idx.kind == HNSW_VEC_QUANT8) {
        const uint32_t half = quant_half_bytes(idx.dim);
        v.half_bytes = half;
        v.row_stride = 2 * half;
        rows.assign((size_t)N * v.row_stride, 0);
        const uint32_t d = idx.dim, nch = d / 8, rem = d % 8;
        parallel_rows(N, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                const uint8_t *c = &idx.codes[i * d];
                for (uint32_t h = 0; h < 2; h++) {
                    uint8_t *o = &rows[i * v.row_stride + h * half];
                    memcpy(o, &idx.mins[i], 4);
                    memcpy(o + 4, &idx.deltas[i], 4);
                    for (uint32_t ch = 0; ch < nch; ch++) memcpy(o + 8 + 4 * ch, c + 8 * ch + 4 * h, 4);
                    if (h == 0) memcpy(o + 8 + 4 * nch, c + 8 * nch, rem);
                }
            }
        });
    } else {
        v.half_bytes = 0;
        v.row_stride = f32_row_stride(idx.dim);
        rows.assign((size_t)N * v.row_stride, 0);
        parallel_rows(N, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++)
                memcpy(&rows[i * v.row_stride], &idx.vals[i * idx.dim], 4 * (size_t)idx.dim);
        });
    }

    // ---- adjacency ----
    v.S0 = adj_stride(idx.layer_m(0), 32);
    v.S1 = adj_stride(idx.params.m, 8);
    std::vector<uint32_t> adj0((size_t)N * v.S0, HX_EMPTY_SLOT);
    std::vector<uint32_t> adj_up(std::max<size_t>(1, idx.adj_up.size()) * v.S1, HX_EMPTY_SLOT);
    std::vector<uint32_t> ovf_off(1, 0), ovf_nbrs;
    auto pack_row = [&](const std::vector<NodeID> &src, uint32_t *dst, uint32_t S,
                        std::vector<NodeID> &tmp) -> bool {
        tmp = src;
        std::sort(tmp.begin(), tmp.end());
        if (tmp.size() <= S) {
            std::copy(tmp.begin(), tmp.end(), dst);
            return false;
        }
        std::copy(tmp.begin(), tmp.begin() + (S - 1), dst);
        return true;  // the caller appends the overflow (serially)
    };
    // rows that fit are packed in parallel; the rare overflow rows are fixed up serially
    std::vector<uint64_t> over0;
    {
        std::vector<std::vector<uint64_t>> over_t(64);
        std::atomic<unsigned> slot{0};
        parallel_rows(N, [&](uint64_t lo, uint64_t hi) {
            const unsigned me = slot.fetch_add(1);
            std::vector<NodeID> tmp;
            for (uint64_t i = lo; i < hi; i++)
                if (pack_row(idx.adj0[i], &adj0[i * v.S0], v.S0, tmp)) over_t[me].push_back(i);
        });
        for (auto &o : over_t) over0.insert(over0.end(), o.begin(), o.end());
        std::sort(over0.begin(), over0.end());
    }
    auto add_overflow = [&](const std::vector<NodeID> &src, uint32_t *dst, uint32_t S) {
        std::vector<NodeID> tmp = src;
        std::sort(tmp.begin(), tmp.end());
        dst[S - 1] = HX_OVF_FLAG | (uint32_t)(ovf_off.size() - 1);
        ovf_nbrs.insert(ovf_nbrs.end(), tmp.begin() + (S - 1), tmp.end());
        ovf_off.push_back((uint32_t)ovf_nbrs.size());
    };
    for (uint64_t i : over0) add_overflow(idx.adj0[i], &adj0[i * v.S0], v.S0);
    {
        std::vector<NodeID> tmp;
        for (size_t r = 0; r < idx.adj_up.size(); r++)
            if (pack_row(idx.adj_up[r], &adj_up[r * v.S1], v.S1, tmp))
                add_overflow(idx.adj_up[r], &adj_up[r * v.S1], v.S1);
    }
    if (ovf_nbrs.empty()) ovf_nbrs.push_back(HX_EMPTY_SLOT);
    std::vector<uint32_t> ub(idx.upper_base.begin(), idx.upper_base.end());
    ub.resize(N, UINT32_MAX);

    // ---- inline rows ("fat" layer-0 blocks), see device_index.h ----
    std::vector<uint8_t> fat;
    {
        int want = inline_rows;
        if (const char *e = getenv("HNSW_MI355X_INLINE_ROWS")) want = atoi(e);
        const uint32_t used = 8 + v.nch4 + v.rem;
        const bool room = idx.kind == HNSW_VEC_QUANT8 && v.S0 == 32 && v.half_bytes >= used + 4;
        // d = 100 is served by the lean compact-layout kernel (search_lean.hip), faster at every launch size
        // than the inline-rows loop: the 4-GB copy is only built there when asked for explicitly
        if (want < 0 && idx.dim == 100 && !(getenv("HNSW_MI355X_LEAN_Q8") && atoi(getenv("HNSW_MI355X_LEAN_Q8")) == 0) &&
            !(getenv("HNSW_MI355X_LEAN") && atoi(getenv("HNSW_MI355X_LEAN")) == 0))
            want = 0;
        const uint64_t need = (uint64_t)N * v.S0 * v.row_stride;
        if (room && (want == 1 || (want < 0 && need <= fat_budget_bytes))) {
            v.fat_stride = (uint64_t)v.S0 * v.row_stride;
            fat.assign(need, 0);
            const uint32_t idpos = v.half_bytes - 4;
            parallel_rows(N, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; i++) {
                    for (uint32_t k = 0; k < v.S0; k++) {
                        const uint32_t nb = adj0[i * v.S0 + k];
                        uint8_t *o = &fat[i * v.fat_stride + (uint64_t)k * v.row_stride];
                        if (nb != HX_EMPTY_SLOT && !(nb & HX_OVF_FLAG))
                            memcpy(o, &rows[(size_t)nb * v.row_stride], v.row_stride);
                        memcpy(o + idpos, &nb, 4);
                    }
                }
            });
        }
    }

    struct Up {
        const void *src;
        size_t nbytes;
    } ups[7] = {{rows.data(), rows.size()},
                {adj0.data(), adj0.size() * 4},
                {adj_up.data(), adj_up.size() * 4},
                {ub.data(), ub.size() * 4},
                {ovf_off.data(), ovf_off.size() * 4},
                {ovf_nbrs.data(), ovf_nbrs.size() * 4},
                {fat.data(), fat.size()}};
    bytes = 0;
    for (int i = 0; i < 7; i++) {
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
        sizes_[i] = ups[i].nbytes;
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
