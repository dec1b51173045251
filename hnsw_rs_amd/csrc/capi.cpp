// capi.cpp -- the extern "C" surface of libhnsw_mi355x.so (include/hnsw_mi355x.h).
// Host logic only; the search entry points upload the index snapshot to HBM on demand and
// launch the HIP kernels of search_kernels.hip.  There is no CPU search path.

#include <hip/hip_runtime.h>

#include <linux/futex.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "device_index.h"
#include "shard_exchange.h"
#include "host_index.h"

using hx::set_error;

// Per-call scratch of the host-pointer search entry points: one device arena, one pinned host arena
// and a stream, kept in a pool on the handle so that a call costs no allocation, one H2D and one D2H
// copy.  Concurrent callers each take their own scratch (hnsw_search* stays re-entrant).
struct SearchScratch {
    void *dev = nullptr, *pin = nullptr;
    size_t dev_cap = 0, pin_cap = 0;
    hipStream_t stream = nullptr;
    int device = -1;
    ~SearchScratch() {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// ---- coalescing of concurrent one-query calls (hnsw_search = the shim's ann_by_vector) ------------------------
// The reference answers ONE query per call and takes &self, so its callers are many threads each blocked in its
// own call (template.rs:306-335).  A lone query is a lone wave: ~130 us on a machine that answers 1024 queries in
// the same time.  Concurrent calls on one handle are therefore gathered: a caller claims a slot of the open batch
// (one compare-and-swap on the batch's word: no lock on this path -- hundreds of callers taking turns on a mutex
// that each holds for 100 ns spend their time in futex hand-offs, measured: 256 callers, 15 cores of system time),
// copies its query into the batch's pinned staging area and sleeps on one of the batch's futex words; the caller
// that claimed slot 0 is the batch's LEADER: it closes the batch, launches ONE kernel for everything that arrived,
// hands every caller its ids and wakes them.  Every query of a batch is answered by its own wave exactly as a lone
// query would be, so the result of a call does not depend on what it was batched with.
//   window:  a leader that has seen concurrency (the previous batch held more than one query) waits up to
//            `window_us` for the callers that were woken together with it to come back; a lone caller never waits.
//   depth:   at most `depth` batches are on the GPU at once; leaders beyond that keep collecting arrivals.
struct SpinLock {  // the slow paths' lock: a short spin, then sleep on the word (free / held / held with sleepers)
    std::atomic<uint32_t> v{0};
    void lock() {
        for (int spins = 0; spins < 128; spins++) {
            uint32_t exp = 0;
            if (v.load(std::memory_order_relaxed) == 0 && v.compare_exchange_weak(exp, 1, std::memory_order_acquire)) return;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        while (v.exchange(2, std::memory_order_acquire) != 0)
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&v), FUTEX_WAIT_PRIVATE, 2, nullptr, nullptr, 0);
    }
    void unlock() {
        if (v.exchange(0, std::memory_order_release) == 2)
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&v), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    }
};

struct CoBatch {
    SearchScratch s;
    size_t p_q = 0, p_out = 0;  // pinned arena offsets (HostSearchPlan for `cap` queries)
    // (n, ef, dim, cap) of this incarnation; written before the word's generation is bumped, read by joiners
    std::atomic<uint32_t> cap{0}, n{0}, ef{0}, dim{0};
    // bits 0..15: slots claimed; bit 16: closed (no more joins); bits 32..63: generation (a batch is reused).
    // A joiner's compare-and-swap succeeds only on the word it read its parameters under.
    static constexpr uint64_t COUNT = 0xFFFFull, CLOSED = 1ull << 16, GEN = 1ull << 32;
    std::atomic<uint64_t> word{CLOSED};
    std::atomic<uint32_t> filed{0};  // claimed slots whose query and request are in place
    struct Req {
        uint32_t *ids, *count;
    };
    std::vector<Req> reqs;
    // futex words, 0 = collecting / running, 1 = results handed out; callers spread over them by slot
    struct alignas(64) Word {
        std::atomic<uint32_t> v{0};
    };
    static constexpr uint32_t WORDS = 16;
    Word done[WORDS];
    std::atomic<uint32_t> readers{0};  // followers that have not picked up their status yet
    int rc = HNSW_OK;                  // batch-level failure (launch, copy), with its text
    std::string err;
    std::vector<int32_t> status;       // per query
};
struct Coalescer {
    std::atomic<CoBatch *> fast{nullptr};  // the open batch callers try first (the latest parameters seen)
    SpinLock mu;                           // everything below; callers on the fast path never take it
    std::condition_variable_any cv;        // leaders wait here for a place on the GPU
    std::vector<CoBatch *> open;           // every open batch, `fast` included
    std::vector<std::unique_ptr<CoBatch>> all;
    std::vector<CoBatch *> idle;
    uint32_t in_flight = 0;
    std::atomic<uint32_t> last_size{1};
    // options "coalesce_us" (< 0: off, every call launches by itself), "coalesce_depth", "coalesce_max"
    std::atomic<int64_t> window_us{30};
    uint32_t depth = 3, cap = 1024;
    std::atomic<uint64_t> n_batches{0}, n_queries{0}, max_batch{0};
    // where a leader's time goes, in ns (hnsw_get_stat "coalesce_ns_window" / "_turn" / "_gpu" / "_handout")
    std::atomic<uint64_t> ns_window{0}, ns_turn{0}, ns_gpu{0}, ns_handout{0};
};

struct hnsw_index {
    std::unique_ptr<hx::HostIndex> host;
    hx::DeviceIndex dev;
    int device = -1;
    int gpu_build = 0;  // option "gpu_build": insert_bulk runs the on-device build (1 host connect, 2 device connect)
    // The on-device build inserts its points in batches of min(build_batch_max, max(64, connected /
    // build_batch_div)): the points of a batch do not see one another.  The defaults
    // build 1M points in 0.7 s; smaller batches stand closer to the reference's one-at-a-time insertion
    // (options "gpu_build_batch_max", "gpu_build_batch_div": 256 and 64 take 4 s per 1M points and lift
    // recall@10 at efSearch 64 from 0.9894 to 0.9901 on the bench's index).
    uint32_t build_batch_max = 8192, build_batch_div = 8;
    // set when an on-device build stopped half way (HIP error, failed exchange): the new points are stored
    // but not all of them are connected, so every later search or build on this handle fails loudly
    // instead of answering from an incomplete graph
    bool incomplete_build = false;
    // option "metric_cosine" (an extension, the reference is Euclidean only): rows are normalised to unit
    // length as they are inserted and queries as they arrive, so the L2 order behind is the cosine order
    bool cosine = false;
    std::mutex mu;
    std::mutex pool_mu;
    std::vector<std::unique_ptr<SearchScratch>> pool;
    Coalescer co;
    // counters behind hnsw_get_stat
    std::atomic<uint64_t> n_uploads{0}, n_point_patches{0}, n_patch_fallbacks{0};
    // the on-device builds of this handle, summed (hnsw_get_stat "build_*"): what the insert kernel read -- the
    // build's algorithmic bytes -- and how long it and the connect phases ran
    struct BuildStats {
        uint64_t points = 0, batches = 0, rows_read = 0, adj_rows = 0, adj_ids = 0, records = 0, removals = 0;
        uint64_t rows_owned = 0, rows_received = 0, exchange_bytes = 0;  // sharded build, phases 2 / 3 by row ownership
        double insert_kernel_s = 0, insert_phase_s = 0, connect_s = 0, exchange_s = 0, connect_kernel_s = 0;
    } build;
};

namespace {

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            set_error("%s failed: %s", #expr, hipGetErrorString(e_));         \
            return e_ == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP;   \
        }                                                                     \
    } while (0)

struct DevBuf {  // RAII device allocation
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        HIP_TRY(hipMalloc(&p, n ? n : 1));
        return HNSW_OK;
    }
    template <class T>
    T *as() {
        return static_cast<T *>(p);
    }
};

// a device-only replica (hnsw_snapshot_adopt / _commit) has no host index behind its snapshot
inline bool is_replica(const hnsw_index *h) { return h->dev.replica; }
inline uint64_t index_len(const hnsw_index *h) { return is_replica(h) ? h->dev.view.n_points : h->host->len(); }
int reject_replica(const hnsw_index *h, const char *what) {
    if (!is_replica(h)) return HNSW_OK;
    set_error("%s: this handle is a device-only replica (hnsw_snapshot_adopt); it holds no host copy of the index", what);
    return HNSW_ERR_ARG;
}

// The cosine option on the way in: a unit-length copy of n rows, by the same operations in the same order as
// hx_normalise_rows_kernel (metric.hip) -- one left-to-right f32 sum of squares, correctly rounded sqrt and
// division, no FMA (this file is compiled with -ffp-contract=off).  Returns rows itself when the option is off.
int cosine_rows(const hnsw_index *h, const float *&rows, uint64_t n, std::vector<float> &keep, uint32_t nb_threads = 1) {
    if (!h->cosine || !rows) return HNSW_OK;
    const uint32_t d = h->host->dim;
    keep.resize((size_t)n * d);
    std::atomic<uint64_t> bad{UINT64_MAX};
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const float *x = rows + i * d;
            float s = 0.0f;
            for (uint32_t e = 0; e < d; e++) {
                const float t = x[e] * x[e];
                s += t;
            }
            const float nrm = sqrtf(s);
            // a row without a direction (all zero, or a sum of squares that under- / overflows f32) cannot be put on
            // the unit sphere: refused here by name instead of poisoning distances with inf - inf later
            if (!(nrm > 0.0f) || !std::isfinite(nrm)) {
                uint64_t cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {
                }
                return;
            }
            float *y = &keep[(size_t)i * d];
            for (uint32_t e = 0; e < d; e++) y[e] = x[e] / nrm;
        }
    };
    const unsigned nt = (unsigned)std::min<uint64_t>(std::max(1u, nb_threads), std::max<uint64_t>(1, n / 4096));
    if (nt <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto &t : th) t.join();
    }
    if (bad.load() != UINT64_MAX) {
        set_error("row %llu has no direction (zero, NaN, or a norm outside f32's range): the cosine metric cannot place it",
                  (unsigned long long)bad.load());
        return HNSW_ERR_NAN_INPUT;
    }
    rows = keep.data();
    return HNSW_OK;
}
// ... and for queries already copied to the device
int cosine_queries(const hnsw_index *h, void *d_Q, uint64_t nq, hipStream_t stream) {
    if (!h->cosine) return HNSW_OK;
    return hx::launch_normalise_rows(static_cast<float *>(d_Q), nq, h->dev.view.dim, stream);
}

// ... and for queries the caller keeps in HBM (const to us): a stream-ordered unit-length copy
struct DeviceQueries {
    const float *q = nullptr;
    void *tmp = nullptr;
    hipStream_t st = nullptr;
    int prepare(const hnsw_index *h, const float *d_Q, uint64_t nq, hipStream_t stream) {
        q = d_Q;
        st = stream;
        if (!h->cosine) return HNSW_OK;
        const size_t bytes = (size_t)nq * h->dev.view.dim * 4;
        HIP_TRY(hipMallocAsync(&tmp, bytes, stream));
        HIP_TRY(hipMemcpyAsync(tmp, d_Q, bytes, hipMemcpyDeviceToDevice, stream));
        q = static_cast<const float *>(tmp);
        return hx::launch_normalise_rows(static_cast<float *>(tmp), nq, h->dev.view.dim, stream);
    }
    ~DeviceQueries() {
        if (tmp) (void)hipFreeAsync(tmp, st);
    }
};

int ensure_uploaded(hnsw_index *h) {
    std::lock_guard<std::mutex> g(h->mu);
    if (is_replica(h) && !h->dev.valid) {
        set_error("replica snapshot not committed (hnsw_snapshot_commit)");
        return HNSW_ERR_ARG;
    }
    if (!h->dev.current(*h->host)) {
        int rc = h->dev.upload(*h->host, h->device);
        if (rc != HNSW_OK) return rc;
        h->device = h->dev.device;
        h->n_uploads.fetch_add(1, std::memory_order_relaxed);
        // searches above ef 320 take their visited set's second level from the device's stream-ordered pool, launch after
        // launch: the pool keeps what is given back instead of returning it to the driver at every synchronisation
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, h->dev.device) == hipSuccess && pool != nullptr) {
            uint64_t keep = 1ull << 30;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    hipError_t e = hipSetDevice(h->dev.device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d): %s", h->dev.device, hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

int check_search_args(const hnsw_index *h, uint32_t ef) {
    if (!h) {
        set_error("null handle");
        return HNSW_ERR_ARG;
    }
    if (h->incomplete_build) {
        set_error("an on-device build on this handle failed half way; the index is incomplete, discard it");
        return HNSW_ERR_ARG;
    }
    if (index_len(h) == 0) {
        set_error("index is empty");
        return HNSW_ERR_EMPTY;
    }
    // no limit on ef (template.rs:306-311): up to 1024 the list lives in a wave's registers, beyond that in HBM
    // scratch (hx_search_spill_kernel: exact, slow).  2^26 entries only bounds the scratch arithmetic.
    if (ef > (1u << 26)) {
        set_error("ef = %u is above 2^26", ef);
        return HNSW_ERR_ARG;
    }
    return HNSW_OK;
}

hx::SearchArgs ann_args(const hx::DevView &v, const float *dQ, uint32_t n, uint32_t ef,
                        uint32_t *ids, float *dists, uint32_t *counts, hnsw_query_stats *stats) {
    hx::SearchArgs a{};
    a.Q = dQ;
    a.qsel = nullptr;
    a.entries = nullptr;
    a.n_entry = 1;
    a.layer_hi = (int32_t)v.nb_layers - 1;  // template.rs:322-326: layers L-1..1 with ef = 1,
    a.layer_lo = 0;                         // then layer 0 with ef
    a.ef_upper = 1;
    a.ef_bottom = ef;
    a.n = n;
    a.out_ids = ids;
    a.out_dists = dists;
    a.out_counts = counts;
    a.out_stats = stats;
    return a;
}

struct ScratchLease {  // takes a scratch from the handle's pool, gives it back at scope exit
    hnsw_index *h;
    std::unique_ptr<SearchScratch> s;
    explicit ScratchLease(hnsw_index *hh) : h(hh) {
        std::lock_guard<std::mutex> g(h->pool_mu);
        if (!h->pool.empty()) {
            s = std::move(h->pool.back());
            h->pool.pop_back();
        }
    }
    ~ScratchLease() {
        if (!s) return;
        std::lock_guard<std::mutex> g(h->pool_mu);
        if (h->pool.size() < 16) h->pool.push_back(std::move(s));
    }
    int prepare(int device, size_t dev_bytes, size_t pin_bytes) {
        if (s && s->device != device) s.reset();
        if (!s) {
            s.reset(new SearchScratch());
            s->device = device;
            HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        }
        if (s->dev_cap < dev_bytes) {
            if (s->dev) (void)hipFree(s->dev);
            s->dev = nullptr;
            s->dev_cap = 0;
            const size_t cap = dev_bytes + dev_bytes / 4 + 4096;
            HIP_TRY(hipMalloc(&s->dev, cap));
            s->dev_cap = cap;
        }
        if (s->pin_cap < pin_bytes) {
            if (s->pin) (void)hipHostFree(s->pin);
            s->pin = nullptr;
            s->pin_cap = 0;
            const size_t cap = pin_bytes + pin_bytes / 4 + 4096;
            HIP_TRY(hipHostMalloc(&s->pin, cap, hipHostMallocDefault));
            s->pin_cap = cap;
        }
        return HNSW_OK;
    }
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- the host-pointer search path --------------------------------------------------------------------------
// One search of nq queries in a leased scratch.  Pinned arena: [queries | result block]; device arena:
// [queries | selection | entries | result block]; the result block is [ids | dists | counts | stats], the same
// layout on both sides, so it comes back in ONE copy.  The queries reach the device by a true asynchronous
// copy out of pinned memory (a hipMemcpyAsync out of pageable user memory is staged by the runtime and does
// not overlap anything).
struct HostSearchPlan {
    size_t o_q, o_sel, o_ent, o_out, dev_bytes;           // device arena
    size_t r_ids, r_dists, r_counts, r_stats, out_bytes;  // result block
    size_t p_q, p_out, pin_bytes;                         // pinned arena
};
HostSearchPlan plan_host_search(uint64_t nq, uint32_t d, uint32_t n, uint32_t n_entry) {
    HostSearchPlan p{};
    p.o_q = 0;
    p.o_sel = p.o_q + align256(nq * d * 4);
    p.o_ent = p.o_sel + align256(nq * 4);
    p.o_out = p.o_ent + align256((size_t)n_entry * 4);
    p.r_ids = 0;
    p.r_dists = p.r_ids + align256(nq * n * 4);
    p.r_counts = p.r_dists + align256(nq * n * 4);
    p.r_stats = p.r_counts + align256(nq * 4);
    p.out_bytes = p.r_stats + align256(nq * sizeof(hnsw_query_stats));
    p.dev_bytes = p.o_out + p.out_bytes;
    p.p_q = 0;
    p.p_out = align256(nq * d * 4);
    p.pin_bytes = p.p_out + p.out_bytes;
    return p;
}

// text of a per-query failure on the calling thread; returns the status
int query_status_error(uint64_t i, int32_t status) {
    switch (status) {
        case HNSW_OK:
            break;
        case HNSW_ERR_NAN_INPUT:
            set_error("query %llu: NaN in the query or in a distance", (unsigned long long)i);
            break;
        case HNSW_ERR_NODE_NOT_IN_GRAPH:
            set_error("Error in search_layer: node not in Graph (query %llu)", (unsigned long long)i);
            break;
        case HNSW_ERR_OVERFLOW:
            set_error("query %llu: visited table exhausted at its largest size", (unsigned long long)i);
            break;
        default:
            set_error("query %llu failed with status %d", (unsigned long long)i, status);
    }
    return status;
}

// The search itself: the queries are in s.pin + p.p_q (or, for a large call, still in the caller's memory: Q_user),
// the results are left in s.pin + p.p_out.  Queries whose visited table filled up are run again with a table twice
// the size.  Returns launch-level errors only; per-query statuses stay in the result block.
int search_staged(hnsw_index *h, SearchScratch &s, const HostSearchPlan &p, hx::SearchArgs a_host, uint64_t nq,
                  const uint32_t *entries, const float *Q_user) {
    const hx::DevView &v = h->dev.view;
    unsigned char *dv = static_cast<unsigned char *>(s.dev), *hv = static_cast<unsigned char *>(s.pin);
    int rc;
    // Small calls skip both copies: pinned host memory is mapped into the device's address space, the kernel reads
    // each query once (400 B per wave over the link) and writes its few result words straight into the pinned
    // result block.  Measured on the 1M x 100d index: a lone 1024-query call 225 us against 232 us with the copies, but
    // 2 / 3 concurrent 1024-query callers 5.2 / 7.5 M q/s against 5.9 / 8.0 M (the copy engines overlap with the other
    // caller's kernel, reads over the link from a busy kernel do not) -- so calls of up to 512 queries (every coalesced
    // batch of up to 512 callers) go without copies, larger ones, and calls whose queries are normalised on the
    // device first (the cosine option), keep them.
    static const bool zc_allowed = !(getenv("HNSW_MI355X_ZERO_COPY") && atoi(getenv("HNSW_MI355X_ZERO_COPY")) == 0);
    static const uint64_t zc_max = getenv("HNSW_MI355X_ZERO_COPY_MAX") ? strtoull(getenv("HNSW_MI355X_ZERO_COPY_MAX"), nullptr, 0) : 512;
    const bool zc = zc_allowed && !Q_user && !h->cosine && nq <= zc_max;
    hx::SearchArgs a = a_host;
    unsigned char *ob = zc ? hv + p.p_out : dv + p.o_out;  // where the kernel writes the result block
    if (zc) {
        a.Q = reinterpret_cast<const float *>(hv + p.p_q);
    } else {
        HIP_TRY(hipMemcpyAsync(dv + p.o_q, Q_user ? (const void *)Q_user : (const void *)(hv + p.p_q), nq * v.dim * 4,
                               hipMemcpyHostToDevice, s.stream));
        if ((rc = cosine_queries(h, dv + p.o_q, nq, s.stream))) return rc;
        a.Q = reinterpret_cast<const float *>(dv + p.o_q);
    }
    a.out_ids = reinterpret_cast<uint32_t *>(ob + p.r_ids);
    a.out_dists = reinterpret_cast<float *>(ob + p.r_dists);
    a.out_counts = reinterpret_cast<uint32_t *>(ob + p.r_counts);
    a.out_stats = reinterpret_cast<hnsw_query_stats *>(ob + p.r_stats);
    if (entries) {
        HIP_TRY(hipMemcpyAsync(dv + p.o_ent, entries, (size_t)a.n_entry * 4, hipMemcpyHostToDevice, s.stream));
        a.entries = reinterpret_cast<const uint32_t *>(dv + p.o_ent);
    }
    const hnsw_query_stats *st = reinterpret_cast<const hnsw_query_stats *>(hv + p.p_out + p.r_stats);
    uint32_t ef_max = std::max(a.ef_bottom, a.ef_upper);
    uint32_t slots = hx::default_slots_log2(ef_max, v.S0);
    uint64_t nrun = nq;
    std::vector<uint32_t> sel;
    while (true) {
        rc = hx::launch_search(v, a, (uint32_t)nrun, slots, s.stream);
        if (rc != HNSW_OK) return rc;
        if (!zc) HIP_TRY(hipMemcpyAsync(hv + p.p_out, dv + p.o_out, p.out_bytes, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
        sel.clear();
        for (uint64_t i = 0; i < nq; i++)
            if (st[i].status == HNSW_ERR_OVERFLOW) sel.push_back((uint32_t)i);
        if (sel.empty() || slots >= hx::max_slots_log2(ef_max)) break;
        slots++;
        HIP_TRY(hipMemcpyAsync(dv + p.o_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));  // `sel` is reused by the next round
        a.qsel = reinterpret_cast<const uint32_t *>(dv + p.o_sel);
        nrun = sel.size();
    }
    return HNSW_OK;
}

// host-pointer search (hnsw_search_batch, hnsw_search_layer): user buffers in, user buffers out
int search_host(hnsw_index *h, hx::SearchArgs a_host, const float *Q, uint64_t nq, uint32_t *ids,
                float *dists, uint32_t *counts, hnsw_query_stats *stats, const uint32_t *entries) {
    int rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    const hx::DevView &v = h->dev.view;
    const uint32_t n = a_host.n, d = v.dim;
    const HostSearchPlan p = plan_host_search(nq, d, n, entries ? a_host.n_entry : 0);
    // queries go through the pinned arena up to 8 MiB (a batch of 1024 x 100d is 400 KB); beyond that the
    // runtime's own pageable staging serves, and the pinned arena holds the result block only
    const bool stage_q = nq * (size_t)d * 4 <= (8u << 20);
    ScratchLease lease(h);
    if ((rc = lease.prepare(h->dev.device, p.dev_bytes, stage_q ? p.pin_bytes : p.out_bytes))) return rc;
    SearchScratch &s = *lease.s;
    HostSearchPlan pp = p;
    if (!stage_q) pp.p_out = 0;
    unsigned char *hv = static_cast<unsigned char *>(s.pin);
    if (stage_q) memcpy(hv + pp.p_q, Q, nq * (size_t)d * 4);
    if ((rc = search_staged(h, s, pp, a_host, nq, entries, stage_q ? nullptr : Q))) return rc;
    const unsigned char *ob = hv + pp.p_out;
    const hnsw_query_stats *st = reinterpret_cast<const hnsw_query_stats *>(ob + pp.r_stats);
    memcpy(ids, ob + pp.r_ids, nq * n * 4);
    if (dists) memcpy(dists, ob + pp.r_dists, nq * n * 4);
    if (counts) memcpy(counts, ob + pp.r_counts, nq * 4);
    if (stats) memcpy(stats, st, nq * sizeof(hnsw_query_stats));
    for (uint64_t i = 0; i < nq; i++)
        if (st[i].status != HNSW_OK) return query_status_error(i, st[i].status);
    return HNSW_OK;
}


// ---- hnsw_search through the coalescer ---------------------------------------------------------------------------
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t while_equals) {
    while (w->load(std::memory_order_acquire) == while_equals)
        (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, while_equals, nullptr, nullptr, 0);
}
// Waking n sleepers from one thread costs that thread n wake-ups one after the other (a hundred microseconds for
// a hundred callers): the leader wakes two, and every caller that wakes up wakes two more.  The word is already 1
// by then, so a caller that was not asleep yet never goes to sleep and no wake-up can be lost.
inline void futex_wake(std::atomic<uint32_t> *w, int n) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

// a batch from the pool made ready for a new incarnation with `claimed` slots already taken (1: the caller leads
// it; 0: a leaderless successor whose first joiner will); called under Coalescer::mu
int cobatch_open(hnsw_index *h, CoBatch &b, uint32_t cap, uint32_t n, uint32_t ef, uint32_t claimed) {
    const uint32_t d = h->dev.view.dim;
    const HostSearchPlan p = plan_host_search(cap, d, n, 0);
    SearchScratch &s = b.s;
    if (s.device != h->dev.device) {  // (a handle moved to another device: start over)
        if (s.dev) (void)hipFree(s.dev);
        if (s.pin) (void)hipHostFree(s.pin);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        s.dev = s.pin = nullptr;
        s.stream = nullptr;
        s.dev_cap = s.pin_cap = 0;
        s.device = h->dev.device;
    }
    if (!s.stream) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (s.dev_cap < p.dev_bytes) {
        if (s.dev) (void)hipFree(s.dev);
        s.dev = nullptr;
        s.dev_cap = 0;
        HIP_TRY(hipMalloc(&s.dev, p.dev_bytes));
        s.dev_cap = p.dev_bytes;
    }
    if (s.pin_cap < p.pin_bytes) {
        if (s.pin) (void)hipHostFree(s.pin);
        s.pin = nullptr;
        s.pin_cap = 0;
        HIP_TRY(hipHostMalloc(&s.pin, p.pin_bytes, hipHostMallocDefault));
        s.pin_cap = p.pin_bytes;
    }
    b.p_q = p.p_q;
    b.p_out = p.p_out;
    b.cap.store(cap, std::memory_order_relaxed);
    b.n.store(n, std::memory_order_relaxed);
    b.ef.store(ef, std::memory_order_relaxed);
    b.dim.store(d, std::memory_order_relaxed);
    b.filed.store(0, std::memory_order_relaxed);
    for (auto &w : b.done) w.v.store(0, std::memory_order_relaxed);
    b.readers.store(0, std::memory_order_relaxed);
    b.rc = HNSW_OK;
    b.err.clear();
    if (b.reqs.size() < cap) b.reqs.resize(cap);
    if (b.status.size() < cap) b.status.resize(cap);
    const uint64_t gen = (b.word.load(std::memory_order_relaxed) >> 32) + 1;
    b.word.store((gen << 32) | claimed, std::memory_order_release);  // open
    return HNSW_OK;
}

// claim a slot of an open batch with these parameters: the slot, or -1 (closed, full, other parameters)
inline int cobatch_join(CoBatch *b, uint32_t n, uint32_t ef, uint32_t d) {
    uint64_t w = b->word.load(std::memory_order_acquire);
    while (true) {
        if ((w & CoBatch::CLOSED) || (w & CoBatch::COUNT) >= b->cap.load(std::memory_order_relaxed)) return -1;
        if (b->n.load(std::memory_order_relaxed) != n || b->ef.load(std::memory_order_relaxed) != ef ||
            b->dim.load(std::memory_order_relaxed) != d)
            return -1;
        // succeeds only if the word is still the one the parameters were read under (same generation, still open)
        if (b->word.compare_exchange_weak(w, w + 1, std::memory_order_acq_rel, std::memory_order_acquire))
            return (int)(w & CoBatch::COUNT);
    }
}

// the snapshot is what the host index holds (read without the handle's lock: nothing may mutate an index while
// it is being searched, include/hnsw_mi355x.h)
inline bool snapshot_current(const hnsw_index *h) {
    return h->dev.valid && (h->dev.replica || h->dev.version_seen == h->host->version);
}

int search_coalesced(hnsw_index *h, const float *q, uint32_t n, uint32_t ef, uint32_t *ids, uint32_t *count) {
    int rc;
    if (!snapshot_current(h) && (rc = ensure_uploaded(h)) != HNSW_OK) return rc;
    Coalescer &co = h->co;
    const uint32_t d = h->dev.view.dim;
    // ---- claim a slot: the open batch everybody looks at first, else (under the lock) any open batch with these
    // parameters, else a new batch which this caller leads ----
    CoBatch *b = co.fast.load(std::memory_order_acquire);
    int slot = b ? cobatch_join(b, n, ef, d) : -1;
    if (slot < 0) {
        std::lock_guard<SpinLock> g(co.mu);
        for (CoBatch *o : co.open)
            if ((slot = cobatch_join(o, n, ef, d)) >= 0) {
                b = o;
                break;
            }
        if (slot < 0) {
            // a leaderless batch nobody joined (other parameters) is taken out of circulation rather than left open
            for (size_t i = 0; i < co.open.size();) {
                CoBatch *o = co.open[i];
                uint64_t w = o->word.load(std::memory_order_acquire);
                if ((w & CoBatch::COUNT) == 0 && !(w & CoBatch::CLOSED) &&
                    o->word.compare_exchange_strong(w, w | CoBatch::CLOSED, std::memory_order_acq_rel)) {
                    co.open.erase(co.open.begin() + i);
                    co.idle.push_back(o);
                    if (co.fast.load(std::memory_order_relaxed) == o) co.fast.store(nullptr, std::memory_order_release);
                } else {
                    i++;
                }
            }
            if (!co.idle.empty()) {
                b = co.idle.back();
                co.idle.pop_back();
            } else {
                co.all.emplace_back(new CoBatch());
                b = co.all.back().get();
            }
            if ((rc = cobatch_open(h, *b, co.cap, n, ef, 1)) != HNSW_OK) {
                co.idle.push_back(b);
                return rc;
            }
            slot = 0;
            co.open.push_back(b);
            co.fast.store(b, std::memory_order_release);
        }
    }
    memcpy(static_cast<unsigned char *>(b->s.pin) + b->p_q + (size_t)slot * d * 4, q, (size_t)d * 4);
    b->reqs[slot] = CoBatch::Req{ids, count};
    b->filed.fetch_add(1, std::memory_order_release);

    if (slot != 0) {
        std::atomic<uint32_t> *word = &b->done[slot % CoBatch::WORDS].v;
        futex_wait(word, 0);
        futex_wake(word, 2);
        int my = b->rc;
        if (my != HNSW_OK)
            set_error("%s", b->err.c_str());
        else
            my = query_status_error(0, b->status[slot]);
        if (b->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // the last one out returns the batch
            std::lock_guard<SpinLock> g(co.mu);
            co.idle.push_back(b);
        }
        return my;
    }

    // ---- leader (slot 0) ----
    using sclk = std::chrono::steady_clock;
    auto ns_since = [](sclk::time_point t) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(sclk::now() - t).count(); };
    const auto t_lead = sclk::now();
    const int64_t window_us = co.window_us.load(std::memory_order_relaxed);
    const uint32_t last = co.last_size.load(std::memory_order_relaxed);
    if (window_us > 0 && last > 1) {
        // callers woken together come back together: wait for as many as the previous batch held, at most the
        // window (spinning: a timed sleep of tens of microseconds wakes up 50 us late)
        const uint32_t target = std::min(last, b->cap.load(std::memory_order_relaxed));
        const auto deadline = t_lead + std::chrono::microseconds(window_us);
        while ((b->word.load(std::memory_order_acquire) & CoBatch::COUNT) < target && sclk::now() < deadline) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    co.ns_window.fetch_add(ns_since(t_lead), std::memory_order_relaxed);
    const auto t_turn = sclk::now();
    {
        std::unique_lock<SpinLock> lk(co.mu);
        while (co.in_flight >= co.depth) co.cv.wait(lk);
        co.in_flight++;
        co.open.erase(std::find(co.open.begin(), co.open.end(), b));
        if (co.fast.load(std::memory_order_relaxed) == b) {
            // the successor is published BEFORE this batch closes, so that arrivals always find an open batch without
            // the lock; it has no leader yet: whoever claims its slot 0 will be
            CoBatch *nx = nullptr;
            if (!co.idle.empty()) {
                nx = co.idle.back();
                co.idle.pop_back();
            } else {
                co.all.emplace_back(new CoBatch());
                nx = co.all.back().get();
            }
            if (cobatch_open(h, *nx, co.cap, n, ef, 0) == HNSW_OK) {
                co.open.push_back(nx);
                co.fast.store(nx, std::memory_order_release);
            } else {
                co.idle.push_back(nx);
                co.fast.store(nullptr, std::memory_order_release);
            }
        }
    }
    co.ns_turn.fetch_add(ns_since(t_turn), std::memory_order_relaxed);
    const uint32_t nq = (uint32_t)(b->word.fetch_or(CoBatch::CLOSED, std::memory_order_acq_rel) & CoBatch::COUNT);
    for (uint32_t spins = 0; b->filed.load(std::memory_order_acquire) != nq; spins++) {
        // joiners between their claim and their copy (~100 ns) -- unless one of them was descheduled right there
        // (a throttled CPU quota can hold a thread for a whole period): then stop burning the core it needs
        if (spins < 2000) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            std::this_thread::yield();
        }
    }
    co.last_size.store(nq, std::memory_order_relaxed);
    co.n_batches.fetch_add(1, std::memory_order_relaxed);
    co.n_queries.fetch_add(nq, std::memory_order_relaxed);
    uint64_t mb = co.max_batch.load(std::memory_order_relaxed);
    while (nq > mb && !co.max_batch.compare_exchange_weak(mb, nq)) {
    }
    HostSearchPlan p = plan_host_search(nq, d, n, 0);  // the device arena and the result block are laid out for nq
    p.p_q = b->p_q;
    p.p_out = b->p_out;  // (the pinned result block starts where the batch's capacity put it)
    hx::DevView dummy{};
    dummy.nb_layers = h->dev.view.nb_layers;
    hx::SearchArgs a = ann_args(dummy, nullptr, n, ef, nullptr, nullptr, nullptr, nullptr);
    const auto t_gpu = sclk::now();
    rc = hipSetDevice(h->dev.device) == hipSuccess ? HNSW_OK : HNSW_ERR_HIP;
    if (rc != HNSW_OK) set_error("hipSetDevice(%d) failed", h->dev.device);
    if (rc == HNSW_OK) rc = search_staged(h, b->s, p, a, nq, nullptr, nullptr);
    co.ns_gpu.fetch_add(ns_since(t_gpu), std::memory_order_relaxed);
    const auto t_hand = sclk::now();
    {
        std::lock_guard<SpinLock> g(co.mu);
        co.in_flight--;
    }
    co.cv.notify_all();
    int my;
    if (rc != HNSW_OK) {
        b->rc = rc;
        b->err = hx::get_error();
        my = rc;
    } else {
        const unsigned char *ob = static_cast<const unsigned char *>(b->s.pin) + p.p_out;
        const uint32_t *o_ids = reinterpret_cast<const uint32_t *>(ob + p.r_ids);
        const uint32_t *o_cnt = reinterpret_cast<const uint32_t *>(ob + p.r_counts);
        const hnsw_query_stats *st = reinterpret_cast<const hnsw_query_stats *>(ob + p.r_stats);
        for (uint32_t i = 0; i < nq; i++) {
            memcpy(b->reqs[i].ids, o_ids + (size_t)i * n, (size_t)n * 4);
            if (b->reqs[i].count) *b->reqs[i].count = o_cnt[i];
            b->status[i] = st[i].status;
        }
        my = query_status_error(0, b->status[0]);
    }
    if (nq > 1) {
        b->readers.store(nq - 1, std::memory_order_release);
        for (uint32_t w = 0; w < std::min(nq, CoBatch::WORDS); w++) {
            b->done[w].v.store(1, std::memory_order_release);
            futex_wake(&b->done[w].v, 2);
        }
    } else {
        std::lock_guard<SpinLock> g(co.mu);
        co.idle.push_back(b);
    }
    co.ns_handout.fetch_add(ns_since(t_hand), std::memory_order_relaxed);
    return my;
}

// ---------------------------------------------------------------------------------------------
// On-device index build: batch-synchronous insert_bulk.
//   per batch:  GPU  hx_insert_kernel -- one wave per point: entry point, greedy descent, and for
//                    every layer of the point search_layer(ef_cons) + select_heuristic
//                    (inserter.rs:40-126) against the graph as it stands in HBM
//               host connect_point   -- the reference's make_connections / prune_connections /
//                    make_pruned_connections (template.rs:196-251) on `nb_threads` threads with the
//                    per-row locks of the CPU build
//               GPU  hx_scatter_rows -- the adjacency rows that changed go back to HBM
// The first points (and any point whose search reports an error) take the CPU path, batches grow
// with the graph (a batch never exceeds 1/8 of the points already connected, at most 4096): points
// of one batch do not see each other, like the racing threads of the reference's own multi-threaded
// insert_bulk.  The result is a valid HNSW graph judged by recall, not by identity.
// ---------------------------------------------------------------------------------------------
// Insertion order of the reference: layers top-down, ids ascending inside a level (template.rs:403-416); the
// entry point is already in.  Levels are bytes: one counting pass instead of a sort of tens of millions of ids.
std::vector<hx::NodeID> insertion_order(const hx::HostIndex &host, const std::vector<hx::NodeID> &ids) {
    size_t count[257] = {0};
    for (hx::NodeID id : ids)
        if (id != host.params.ep) count[host.levels[id]]++;
    size_t start[256], at = 0;
    for (int l = 255; l >= 0; l--) {
        start[l] = at;
        at += count[l];
    }
    std::vector<hx::NodeID> order(at);
    for (hx::NodeID id : ids)
        if (id != host.params.ep) order[start[host.levels[id]]++] = id;
    return order;
}

int gpu_insert_bulk(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose,
                    const uint8_t *levels) {
    using hx::NodeID;
    hx::HostIndex &host = *h->host;
    if (nb_threads == 0) nb_threads = 1;
    if (host.params.m > 128 || host.params.ef_cons > 512) {
        set_error("on-device build supports m <= 128 and ef_construction <= 512");
        return HNSW_ERR_ARG;
    }
    const uint64_t n_before = host.len();
    std::vector<NodeID> ids;
    int rc = host.store_points(rows, n, levels, &ids, nb_threads);
    if (rc != HNSW_OK) return rc;
    host.prepare_build();
    // insertion order of the reference: layers top-down, ids ascending inside a level (template.rs:403-416)
    const std::vector<NodeID> order = insertion_order(host, ids);

    // ---- seed on the CPU: the first points must be inserted one after the other ----
    const uint64_t SEED = 2048;
    size_t pos = 0;
    if (n_before < SEED) {
        const size_t take = std::min<size_t>(order.size(), SEED - n_before);
        std::vector<NodeID> seed(order.begin(), order.begin() + take);
        // sequential (one Inserter) so that the seed graph is the reference's single-thread graph
        std::unique_ptr<hx::Inserter, void (*)(hx::Inserter *)> ins(hx::new_inserter(host.len()),
                                                                     hx::free_inserter);
        for (NodeID id : seed) {
            rc = host.insert(id, *ins);
            if (rc != HNSW_OK) return rc;
        }
        pos = take;
    }
    if (pos == order.size()) {
        host.version++;
        return HNSW_OK;
    }

    // ---- device snapshot without the search-only extras ----
    const int saved_inline = h->dev.inline_rows;
    h->dev.inline_rows = 0;
    h->dev.release();
    rc = h->dev.upload(host, h->device);
    h->dev.inline_rows = saved_inline;
    if (rc != HNSW_OK) return rc;
    h->device = h->dev.device;
    HIP_TRY(hipSetDevice(h->dev.device));
    hx::DevView v = h->dev.view;
    const uint32_t m = (uint32_t)host.params.m, L = host.nb_layers();
    const uint32_t BMAX = 4096;
    DevBuf dLevels, dIds, dOutIds, dOutD, dStatus, dRowIdx, dRowData;
    if ((rc = dLevels.alloc(host.len())) || (rc = dIds.alloc(BMAX * 4)) ||
        (rc = dOutIds.alloc((size_t)BMAX * L * m * 4)) || (rc = dOutD.alloc((size_t)BMAX * L * m * 4)) ||
        (rc = dStatus.alloc(BMAX * 4)))
        return rc;
    HIP_TRY(hipMemcpy(dLevels.p, host.levels.data(), host.len(), hipMemcpyHostToDevice));
    std::vector<uint32_t> o_ids((size_t)BMAX * L * m);
    std::vector<float> o_d((size_t)BMAX * L * m);
    std::vector<int32_t> o_st(BMAX);
    size_t row_cap = 0;
    std::vector<uint32_t> row_idx, row_data;
    std::vector<std::vector<uint64_t>> dirty_t(nb_threads);
    hx::DirtyStamps stamps(host.adj0.size(), host.adj_up.size());
    uint64_t connected = n_before + pos;
    const auto t_start = std::chrono::steady_clock::now();
    double t_gpu = 0, t_host = 0, t_sync = 0;
    size_t n_fallback = 0, n_batches = 0;

    while (pos < order.size()) {
        const size_t B = std::min<size_t>(order.size() - pos,
                                          std::min<uint64_t>(std::min<uint64_t>(BMAX, h->build_batch_max), std::max<uint64_t>(64, connected / h->build_batch_div)));
        const NodeID *batch = &order[pos];
        auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipMemcpy(dIds.p, batch, B * 4, hipMemcpyHostToDevice));
        hx::InsertArgs a{};
        a.point_ids = dIds.as<uint32_t>();
        a.levels = dLevels.as<uint8_t>();
        a.ef_cons = (uint32_t)host.params.ef_cons;
        a.m = m;
        a.max_layers = L;
        a.out_ids = dOutIds.as<uint32_t>();
        a.out_dists = dOutD.as<float>();
        a.out_status = dStatus.as<int32_t>();
        rc = hx::launch_insert(v, a, (uint32_t)B, nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(o_ids.data(), dOutIds.p, B * L * m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(o_d.data(), dOutD.p, B * L * m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(o_st.data(), dStatus.p, B * 4, hipMemcpyDeviceToHost));
        auto t1 = std::chrono::steady_clock::now();

        // ---- host: connect the batch (reference semantics), collect the rows that changed ----
        std::atomic<size_t> next{0};
        std::atomic<int> err{HNSW_OK};
        std::vector<NodeID> fallback;
        std::mutex fb_mu;
        for (auto &dv : dirty_t) dv.clear();
        stamps.next_batch();
        auto work = [&](unsigned t) {
            std::vector<std::vector<hx::Dist>> nbrs(L);
            for (size_t i = next.fetch_add(1); i < B && err.load() == HNSW_OK; i = next.fetch_add(1)) {
                const NodeID p = batch[i];
                if (o_st[i] != HNSW_OK) {
                    std::lock_guard<std::mutex> g(fb_mu);
                    fallback.push_back(p);
                    continue;
                }
                for (uint32_t l = 0; l < L; l++) {
                    nbrs[l].clear();
                    for (uint32_t k = 0; k < m; k++) {
                        const uint32_t id = o_ids[(i * L + l) * m + k];
                        if (id != UINT32_MAX) nbrs[l].push_back(hx::Dist{id, o_d[(i * L + l) * m + k]});
                    }
                }
                const int r = host.connect_point(p, nbrs, &dirty_t[t], &stamps);
                if (r != HNSW_OK) err.store(r);
            }
        };
        {
            const unsigned nt = (unsigned)std::min<size_t>(nb_threads, B);
            if (nt <= 1) {
                work(0);
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
                for (auto &t : th) t.join();
            }
        }
        if (err.load() != HNSW_OK) return err.load();
        if (!fallback.empty()) {  // e.g. visited-table overflow: the CPU path serves those points
            n_fallback += fallback.size();
            std::sort(fallback.begin(), fallback.end());
            std::unique_ptr<hx::Inserter, void (*)(hx::Inserter *)> ins(hx::new_inserter(host.len()),
                                                                         hx::free_inserter);
            hx::DirtyScope scope(&dirty_t[0], &stamps);
            for (NodeID p : fallback) {
                rc = host.insert(p, *ins);
                if (rc != HNSW_OK) return rc;
            }
        }
        auto t2 = std::chrono::steady_clock::now();

        // ---- changed rows back to HBM (truncated to the stride; the final upload is exact) ----
        std::vector<uint64_t> dirty0, dirty_up;  // already unique (per-row stamps)
        for (auto &dv : dirty_t)
            for (uint64_t key : dv) ((key >> 32) == 0 ? dirty0 : dirty_up).push_back(key);
        for (int pass = 0; pass < 2; pass++) {  // pass 0: layer 0 rows, pass 1: upper-layer rows
            const std::vector<uint64_t> &dirty = pass == 0 ? dirty0 : dirty_up;
            if (dirty.empty()) continue;
            const uint32_t S = pass == 0 ? v.S0 : v.S1;
            row_idx.resize(dirty.size());
            row_data.resize(dirty.size() * (size_t)S);
            auto pack = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++) {
                    const uint32_t layer = (uint32_t)(dirty[i] >> 32);
                    const NodeID id = (NodeID)dirty[i];
                    const std::vector<NodeID> &r = host.row(layer, id);
                    row_idx[i] = layer == 0 ? id : host.upper_base[id] + layer - 1;
                    uint32_t *o = &row_data[i * (size_t)S];
                    const size_t k = std::min<size_t>(r.size(), S);
                    std::copy(r.begin(), r.begin() + k, o);
                    std::fill(o + k, o + S, UINT32_MAX);
                }
            };
            const unsigned nt = (unsigned)std::min<size_t>(nb_threads, std::max<size_t>(1, dirty.size() / 4096));
            if (nt <= 1) {
                pack(0, dirty.size());
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++)
                    th.emplace_back(pack, dirty.size() * t / nt, dirty.size() * (t + 1) / nt);
                for (auto &t : th) t.join();
            }
            if (row_idx.size() > row_cap) {
                row_cap = row_idx.size() * 2;
                if (dRowIdx.p) (void)hipFree(dRowIdx.p);
                if (dRowData.p) (void)hipFree(dRowData.p);
                dRowIdx.p = dRowData.p = nullptr;
                if ((rc = dRowIdx.alloc(row_cap * 4)) ||
                    (rc = dRowData.alloc(row_cap * (size_t)std::max(v.S0, v.S1) * 4)))
                    return rc;
            }
            HIP_TRY(hipMemcpy(dRowIdx.p, row_idx.data(), row_idx.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dRowData.p, row_data.data(), row_data.size() * 4, hipMemcpyHostToDevice));
            rc = hx::launch_scatter_rows(pass == 0 ? h->dev.adj0_mut() : h->dev.adj_up_mut(), S,
                                         dRowIdx.as<uint32_t>(), dRowData.as<uint32_t>(),
                                         (uint32_t)row_idx.size(), nullptr);
            if (rc != HNSW_OK) return rc;
            HIP_TRY(hipDeviceSynchronize());  // the staging buffers are reused by the next pass
        }
        HIP_TRY(hipDeviceSynchronize());
        auto t3 = std::chrono::steady_clock::now();
        t_gpu += std::chrono::duration<double>(t1 - t0).count();
        t_host += std::chrono::duration<double>(t2 - t1).count();
        t_sync += std::chrono::duration<double>(t3 - t2).count();
        pos += B;
        connected += B;
        n_batches++;
        if (verbose && (n_batches % 16 == 0 || pos == order.size()))
            fprintf(stderr, "\rBuilding HNSW index on the GPU %zu/%zu", pos, order.size());
    }
    if (verbose) {
        const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        fprintf(stderr,
                "\non-device build: %zu batches in %.2f s (insert kernel + copies %.2f s, host connect %.2f s, "
                "row scatter %.2f s), %zu points took the CPU path\n",
                n_batches, tot, t_gpu, t_host, t_sync, n_fallback);
    }
    host.version++;  // the search snapshot (overflow CSR, inline rows) is rebuilt by the next upload
    return HNSW_OK;
}

// ---------------------------------------------------------------------------------------------
// On-device build, connect step on the GPU as well (option "gpu_build" = 2).  Per batch:
//   phase 1  hx_insert_kernel  -- as above; additionally writes the new point's own rows and appends
//                                 one reverse-edge request (target n, source p, layer, d) per selected
//                                 neighbour
//   host     sort the requests by (layer, target)                       [a few ms per batch]
//   phase 2  hx_connect_kernel -- one wave per target row: append, or prune to the cap's nearest;
//                                 reports the edges that fell out
//   host     sort the removals by (layer, x)
//   phase 3  hx_remove_kernel  -- one wave per row that lost a reverse edge (keeps a last edge)
// Every adjacency row is owned by one wave per phase: no locks, deterministic for a given batch
// schedule.  The host graph is rebuilt from the device arrays once, at the end.
// ---------------------------------------------------------------------------------------------
// The device rows of the full on-device build hold at most `cap` neighbours.  A row the CPU path left
// longer than that (the reference's transient overflow) is pruned here the way the next
// prune_connections would: nearest `cap` by (dist, id), reverse edges removed.  Where the dropped
// edge is the other node's last one it stays on that side (graph.rs:85-94); the pruned side gets it
// back after the build (`restore`: hx_edge_key(layer, x, node)), exactly like a refusal of hx_remove_kernel.
void clamp_rows_to_cap(hx::HostIndex &host, std::vector<uint64_t> *restore) {
    using hx::NodeID;
    for (uint32_t l = 0; l < host.nb_layers(); l++) {
        const size_t cap = (size_t)host.layer_m(l);
        for (NodeID id : host.layer_nodes[l]) {
            std::vector<NodeID> &row = host.row(l, id);
            if (row.size() <= cap) continue;
            hx::PointView a, b;
            host.get_point(id, &a);
            std::vector<hx::Dist> ds;
            for (NodeID x : row) {
                host.get_point(x, &b);
                ds.push_back(hx::Dist{x, host.dist2other(a, b)});
            }
            std::sort(ds.begin(), ds.end(), hx::dist_lt);
            for (size_t i = cap; i < ds.size(); i++) {
                std::vector<NodeID> &back = host.row(l, ds[i].id);
                if (back.size() == 1 && back[0] == id)
                    restore->push_back(hx::hx_edge_key(l, ds[i].id, id));
                else
                    back.erase(std::remove(back.begin(), back.end(), id), back.end());
            }
            row.clear();
            for (size_t i = 0; i < cap; i++) row.push_back(ds[i].id);
        }
    }
}

// Sharded build (BASELINE configs[4]): every rank holds the full replica.  The insertion searches of a batch are
// split over the ranks by position, and what they produce travels as edge records through an all-gather (the
// caller's collective, RCCL in production); the record list carries the whole batch (own rows included,
// InsertArgs::emit_own) and its sort makes the order canonical.  Phases 2 / 3 are split by ROW: every rank sees
// every record, the rank that owns a row (node id % world) appends / prunes / drops in it -- each row's outcome
// depends on that row and its records alone, so the split changes nothing -- the removals phase 2 files are
// all-gathered between the two phases, and the rows an owner changed travel to the other replicas as whole rows of
// ids at the end of the batch (hx_pack_rows_kernel / hx_apply_rows_kernel).  Five collectives per batch (records;
// removal counts + removals; row counts + rows -- the count exchanges are 64 B per rank and carry the rank's status,
// so the ranks stop together), the replicas identical after each.  HNSW_MI355X_SHARD_CONNECT=0: phases 2 / 3 on
// every rank in full, as before round 4 (one collective per batch).
struct ShardCtx {
    uint32_t rank, world;
    unsigned char *d_send, *d_recv;  // one slot / world slots of slot_bytes
    uint64_t slot_bytes;
    hnsw_allgather_fn allgather;
    void *ctx;
};
// SH_BCAP: the largest batch (option gpu_build_batch_max, default 8192); buffers and exchange slots are sized for it
constexpr uint32_t SH_HEADER = 64, SH_FAILCAP = 1024, SH_BCAP = 32768;
inline uint32_t shard_slot_records(uint32_t m, uint32_t world) {
    return ((SH_BCAP + world - 1) / world) * m * 4;  // both directions, 2 x slack for upper layers
}
inline uint64_t shard_record_bytes(uint32_t m, uint32_t world) {  // the records' part of a slot: [header][failed ids][keys][vals]
    return ((uint64_t)SH_HEADER + SH_FAILCAP * 4 + (uint64_t)shard_slot_records(m, world) * 12 + 255) & ~255ull;
}
// rows one rank may change in a batch (its share of the targets of every rank's records, plus the rows it drops from)
inline uint32_t shard_slot_rows(uint32_t m, uint32_t world) { return 2 * shard_slot_records(m, world); }
inline uint32_t shard_ship_slots(uint32_t m) { return hx::adj_stride(2ull * m, 32); }  // ids per shipped row: a layer-0 row
inline uint64_t shard_slot_bytes(uint32_t m, uint32_t world) {
    const uint64_t rows = (uint64_t)SH_HEADER + (uint64_t)shard_slot_rows(m, world) * (8 + 4ull * shard_ship_slots(m));
    return (std::max(shard_record_bytes(m, world), rows) + 255) & ~255ull;
}

int gpu_insert_bulk_full(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose,
                         const uint8_t *levels, const ShardCtx *sh = nullptr) {
    using hx::NodeID;
    hx::HostIndex &host = *h->host;
    if (nb_threads == 0) nb_threads = 1;
    if (host.params.m > 128 || host.params.ef_cons > 512) {
        set_error("on-device build supports m <= 128 and ef_construction <= 512");
        return HNSW_ERR_ARG;
    }
    if (host.len() + n >= (1ull << hx::HX_EDGE_ID_BITS)) {  // edge records carry 30-bit ids
        if (sh) {
            set_error("sharded build: ids must stay below 2^30");
            return HNSW_ERR_ARG;
        }
        return gpu_insert_bulk(h, rows, n, nb_threads, verbose, levels);
    }
    const uint64_t n_before = host.len();
    std::vector<NodeID> ids;
    const auto t_enter = std::chrono::steady_clock::now();
    int rc = host.store_points(rows, n, levels, &ids, nb_threads, /*reserve_rows=*/false);
    if (rc != HNSW_OK) return rc;
    const auto t_stored = std::chrono::steady_clock::now();
    host.prepare_build();
    const auto t_prep = std::chrono::steady_clock::now();
    const std::vector<NodeID> order = insertion_order(host, ids);
    const auto t_ordered = std::chrono::steady_clock::now();
    const uint64_t SEED = 2048;
    size_t pos = 0;
    std::unique_ptr<hx::Inserter, void (*)(hx::Inserter *)> ins(hx::new_inserter(host.len()), hx::free_inserter);
    if (verbose)
        fprintf(stderr, "host phases before the seed: store_points %.2f s, locks %.2f s, order %.2f s, inserter %.2f s\n",
                std::chrono::duration<double>(t_stored - t_enter).count(), std::chrono::duration<double>(t_prep - t_stored).count(),
                std::chrono::duration<double>(t_ordered - t_prep).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ordered).count());
    // The first points are inserted one after the other on the host (the graph depends on it).  Unless they are all
    // there is to insert, that runs WHILE the vector rows travel to HBM (DeviceIndex::upload's side job): the rows are
    // immutable once stored, the adjacency -- which the seed writes -- is packed after the seed has finished.
    const size_t take = n_before < SEED ? std::min<size_t>(order.size(), SEED - n_before) : 0;
    std::vector<uint64_t> restore;
    auto seed_and_clamp = [&]() -> int {
        for (size_t i = 0; i < take; i++) {
            const int r = host.insert(order[i], *ins);
            if (r != HNSW_OK) return r;
        }
        clamp_rows_to_cap(host, &restore);
        return HNSW_OK;
    };
    if (take == order.size() || host.nb_layers() > 16) {  // nothing for the device (edge records carry 4-bit layers)
        for (; pos < order.size(); pos++) {
            rc = host.insert(order[pos], *ins);
            if (rc != HNSW_OK) return rc;
        }
        host.version++;
        return HNSW_OK;
    }
    pos = take;
    const auto t_start = std::chrono::steady_clock::now();
    const auto t_clamped = t_start;
    const int saved_inline = h->dev.inline_rows;
    h->dev.inline_rows = 0;
    h->dev.release();
    rc = h->dev.upload(host, h->device, seed_and_clamp);
    h->dev.inline_rows = saved_inline;
    if (rc != HNSW_OK) return rc;
    h->device = h->dev.device;
    HIP_TRY(hipSetDevice(h->dev.device));
    const auto t_uploaded = std::chrono::steady_clock::now();
    hx::DevView v = h->dev.view;
    const uint32_t m = (uint32_t)host.params.m, L = host.nb_layers();
    const uint32_t BMAX = SH_BCAP;
    // a point has 1 + 1/(m-1) layers on average; sharded: records in both directions
    const uint32_t W = sh ? sh->world : 1, SLOT_REC = sh ? shard_slot_records(m, W) : 0;
    // Record capacity: what the largest batch of this build can file, (level + 1) * m per point (twice
    // that with records in both directions) -- not an average: a batch of high-level points files more than
    // 2 m each.  Sharded: the caller's slots are sized by m and the world alone; a point whose records do
    // not fit its rank's slot fails cleanly on the device (nothing reserved) and takes the CPU path.
    uint64_t need_max = 0;
    if (!sh) {
        uint64_t c = n_before + pos;
        for (size_t q = pos; q < order.size();) {
            const size_t Bq = std::min<size_t>(order.size() - q, std::min<uint64_t>(std::min<uint64_t>(BMAX, h->build_batch_max), std::max<uint64_t>(64, c / h->build_batch_div)));
            uint64_t need = 0;
            for (size_t i = 0; i < Bq; i++) need += ((uint64_t)host.levels[order[q + i]] + 1) * m;
            need_max = std::max(need_max, need);
            q += Bq;
            c += Bq;
        }
        if (need_max >= (1ull << 31)) {
            set_error("on-device build: a batch would file %llu edge records", (unsigned long long)need_max);
            return HNSW_ERR_ARG;
        }
    }
    const uint32_t REQ_CAP = sh ? W * SLOT_REC : (uint32_t)std::max<uint64_t>(need_max, (uint64_t)BMAX * m * 2);
    if (sh && (sh->slot_bytes < shard_slot_bytes(m, W) || sh->rank >= W || !sh->d_send || !sh->d_recv || !sh->allgather)) {
        set_error("sharded build: exchange buffers too small or bad rank / world");
        return HNSW_ERR_ARG;
    }
    const uint32_t REF_CAP = 1u << 20;      // kept-last-edge records of the whole build
    const size_t temp_bytes = hx::sort_temp_bytes(REQ_CAP);
    const uint64_t REC_BYTES = sh ? shard_record_bytes(m, W) : 0;
    // phases 2 / 3 split by row ownership (ShardCtx above)
    static const bool shard_connect_on = !(getenv("HNSW_MI355X_SHARD_CONNECT") && atoi(getenv("HNSW_MI355X_SHARD_CONNECT")) == 0);
    const bool own_rows = sh && W > 1 && shard_connect_on;
    const uint32_t CHG_CAP = own_rows ? shard_slot_rows(m, W) : 0, SHIP = shard_ship_slots(m);
    const uint64_t SHIP_UNIT = 8 + 4ull * SHIP;
    if (own_rows && (v.S0 > SHIP || v.S1 > SHIP)) {
        set_error("sharded build: adjacency rows of %u / %u slots, exchange entries of %u", v.S0, v.S1, SHIP);
        return HNSW_ERR_ARG;
    }
    // the file of changed rows: HX_CHG_LISTS lists (ConnectArgs), each with room for twice its even share
    const uint32_t CHG_LIST_CAP = own_rows ? 2 * ((CHG_CAP + hx::HX_CHG_LISTS - 1) / hx::HX_CHG_LISTS) : 0;
    DevBuf dLevels, dIds, dOutIds, dOutD, dStatus, dCnt, dKeyA, dKeyB, dValA, dValB, dTemp, dRef, dRead, dChg, dChgCnt;
    if (own_rows && ((rc = dChg.alloc((size_t)CHG_LIST_CAP * hx::HX_CHG_LISTS * 8)) || (rc = dChgCnt.alloc(hx::HX_CHG_LISTS * 4)))) return rc;
    // One variable-size exchange: 64 B per rank first ([count, status]: every rank learns every count and stops
    // with the others when one of them failed), then the largest count's worth of bytes per rank.  `d_src` is copied
    // behind the header (nullptr: the data is in the slot already).  Rank r's data: d_recv + r * stride + SH_HEADER.
    std::vector<uint32_t> x_counts(W);
    uint64_t x_stride = 0;
    double t_exchange = 0;
    uint64_t x_bytes = 0;
    auto exchange = [&](uint32_t count, uint64_t unit, int32_t status, const void *d_src, const char *what) -> int {
        const auto tx0 = std::chrono::steady_clock::now();
        if ((uint64_t)SH_HEADER + count * unit > sh->slot_bytes && status == 0) status = HNSW_ERR_OVERFLOW;
        uint32_t hdr[SH_HEADER / 4] = {0};
        hdr[0] = status ? 0 : count;
        hdr[1] = (uint32_t)status;
        HIP_TRY(hipMemcpy(sh->d_send, hdr, SH_HEADER, hipMemcpyHostToDevice));
        if (d_src && hdr[0])
            HIP_TRY(hipMemcpyAsync(sh->d_send + SH_HEADER, d_src, hdr[0] * unit, hipMemcpyDeviceToDevice, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        int r = sh->allgather(sh->ctx, SH_HEADER);
        if (r != 0) {
            set_error("sharded build: the all-gather callback failed (%d) on the %s counts", r, what);
            return HNSW_ERR_RCCL;
        }
        std::vector<uint32_t> all((size_t)W * SH_HEADER / 4);
        HIP_TRY(hipMemcpy(all.data(), sh->d_recv, (size_t)W * SH_HEADER, hipMemcpyDeviceToHost));
        uint32_t maxc = 0;
        for (uint32_t k = 0; k < W; k++) {
            if (all[k * (SH_HEADER / 4) + 1] != 0) {
                set_error("sharded build: rank %u reported status %d in the %s phase", k, (int)all[k * (SH_HEADER / 4) + 1], what);
                return (int)all[k * (SH_HEADER / 4) + 1];
            }
            x_counts[k] = all[k * (SH_HEADER / 4)];
            maxc = std::max(maxc, x_counts[k]);
        }
        x_stride = (SH_HEADER + maxc * unit + 63) & ~63ull;
        if (x_stride > sh->slot_bytes) {  // cannot happen with honest peers (each checked its own count above)
            set_error("sharded build: a rank announced %u %s, beyond the slot", maxc, what);
            return HNSW_ERR_OVERFLOW;
        }
        if (maxc) {
            r = sh->allgather(sh->ctx, x_stride);
            if (r != 0) {
                set_error("sharded build: the all-gather callback failed (%d) on the %s", r, what);
                return HNSW_ERR_RCCL;
            }
            x_bytes += x_stride * W;
        }
        t_exchange += std::chrono::duration<double>(std::chrono::steady_clock::now() - tx0).count();
        return HNSW_OK;
    };
    struct EvPair {  // the insert kernel's launches are timed with HIP events on their stream
        hipEvent_t a = nullptr, b = nullptr;
        ~EvPair() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev, ev_conn, ev_rem;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventCreate(&ev_conn.a));
    HIP_TRY(hipEventCreate(&ev_conn.b));
    HIP_TRY(hipEventCreate(&ev_rem.a));
    HIP_TRY(hipEventCreate(&ev_rem.b));
    double t_kernel_ms = 0, t_conn_kernel_ms = 0;
    bool rem_pending = false;
    auto collect_remove_time = [&]() -> int {  // the drop kernel of the previous batch (nothing waits for it in-batch)
        if (!rem_pending) return HNSW_OK;
        HIP_TRY(hipEventSynchronize(ev_rem.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev_rem.a, ev_rem.b));
        t_conn_kernel_ms += ms;
        rem_pending = false;
        return HNSW_OK;
    };
    if ((rc = dRead.alloc(32))) return rc;
    HIP_TRY(hipMemset(dRead.p, 0, 32));
    if ((rc = dLevels.alloc(host.len())) || (rc = dIds.alloc(BMAX * 4)) ||
        (rc = dOutIds.alloc((size_t)BMAX * L * m * 4)) || (rc = dOutD.alloc((size_t)BMAX * L * m * 4)) ||
        (rc = dStatus.alloc(BMAX * 4)) || (rc = dCnt.alloc(64)) || (rc = dKeyA.alloc((size_t)REQ_CAP * 8)) ||
        (rc = dKeyB.alloc((size_t)REQ_CAP * 8)) || (rc = dValA.alloc((size_t)REQ_CAP * 4)) ||
        (rc = dValB.alloc((size_t)REQ_CAP * 4)) || (rc = dTemp.alloc(temp_bytes)) ||
        (rc = dRef.alloc((size_t)REF_CAP * 8)))
        return rc;
    HIP_TRY(hipMemcpy(dLevels.p, host.levels.data(), host.len(), hipMemcpyHostToDevice));
    // the edges' distances beside the adjacency for the length of this build (ConnectArgs: a prune then evaluates
    // nothing); 0xFFFFFFFF = not known yet (the rows that predate this build: evaluated at their first prune).  128 B
    // per point at m = 16; without the memory for it the build runs as before
    DevBuf dAdjD0, dAdjDUp;
    uint32_t *adjd0 = nullptr, *adjd_up = nullptr;
    {
        static const bool keep_dists = !(getenv("HNSW_MI355X_BUILD_EDGE_DISTS") && atoi(getenv("HNSW_MI355X_BUILD_EDGE_DISTS")) == 0);
        const size_t b0 = (size_t)host.len() * v.S0 * 4, b1 = std::max<size_t>(1, host.adj_up.size()) * v.S1 * 4;
        if (keep_dists && hipMalloc(&dAdjD0.p, b0) == hipSuccess && hipMalloc(&dAdjDUp.p, b1) == hipSuccess &&
            hipMemset(dAdjD0.p, 0xFF, b0) == hipSuccess && hipMemset(dAdjDUp.p, 0xFF, b1) == hipSuccess) {
            adjd0 = dAdjD0.as<uint32_t>();
            adjd_up = dAdjDUp.as<uint32_t>();
        } else {
            (void)hipGetLastError();
        }
    }
    // counters: [0] requests, [1] removals, [2] refusals (accumulate over the build), [3] status
    uint32_t *cnt = dCnt.as<uint32_t>();
    HIP_TRY(hipMemset(dCnt.p, 0, 64));
    std::vector<int32_t> o_st(BMAX);
    std::vector<NodeID> failed;
    uint64_t connected = n_before + pos;
    double t_ins = 0, t_conn = 0;
    size_t n_batches = 0, n_req = 0, n_rem = 0, n_again = 0, n_shipped = 0, n_owned = 0;
    uint32_t counts[4];
    // the capacity of the new points' host rows (one small allocation each) is reserved by other threads while the
    // GPU runs the batches: nothing touches the host graph until the read-back below
    struct RowReserve {
        std::thread t;
        ~RowReserve() {
            if (t.joinable()) t.join();
        }
    } row_reserve;
    {
        hx::HostIndex *hp = &host;
        const NodeID first_new = (NodeID)n_before;
        const uint32_t threads = std::max(1u, nb_threads / 2);
        row_reserve.t = std::thread([hp, first_new, n, threads] { hp->reserve_layer0_rows(first_new, n, threads); });
    }
    const auto t_loop0 = std::chrono::steady_clock::now();

    while (pos < order.size()) {
        const size_t B = std::min<size_t>(order.size() - pos,
                                          std::min<uint64_t>(std::min<uint64_t>(BMAX, h->build_batch_max), std::max<uint64_t>(64, connected / h->build_batch_div)));
        const NodeID *batch = &order[pos];
        auto t0 = std::chrono::steady_clock::now();
        if ((rc = collect_remove_time())) return rc;
        // this rank's slice of the batch (everything when not sharded)
        const size_t s_lo = sh ? B * sh->rank / W : 0, s_hi = sh ? B * (sh->rank + 1) / W : B, nb = s_hi - s_lo;
        if (nb) HIP_TRY(hipMemcpy(dIds.p, batch + s_lo, nb * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(dCnt.p, 0, 8));  // requests, removals
        if (own_rows) HIP_TRY(hipMemset(dChgCnt.p, 0, hx::HX_CHG_LISTS * 4));  // rows this rank changed
        // ---- phase 1: searches + heuristic, own rows, requests ----
        hx::InsertArgs a{};
        a.point_ids = dIds.as<uint32_t>();
        a.levels = dLevels.as<uint8_t>();
        a.ef_cons = (uint32_t)host.params.ef_cons;
        a.m = m;
        a.max_layers = L;
        a.out_ids = dOutIds.as<uint32_t>();
        a.out_dists = dOutD.as<float>();
        a.out_status = dStatus.as<int32_t>();
        a.adj0_mut = h->dev.adj0_mut();
        a.adj_up_mut = h->dev.adj_up_mut();
        a.adjd0_mut = adjd0;
        a.adjd_up_mut = adjd_up;
        a.counters = dRead.as<unsigned long long>();
        uint32_t nreq = 0, nreq_all = 0;  // records this rank sorts and applies; records of the batch
        auto timed_insert = [&](uint32_t nblocks, int adjust) -> int {
            HIP_TRY(hipEventRecord(ev.a, nullptr));
            const int r = hx::launch_insert(v, a, nblocks, nullptr, adjust);
            if (r != HNSW_OK) return r;
            HIP_TRY(hipEventRecord(ev.b, nullptr));
            HIP_TRY(hipEventSynchronize(ev.b));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
            t_kernel_ms += ms;
            return HNSW_OK;
        };
        if (!sh) {
            a.req_keys = dKeyA.as<uint64_t>();
            a.req_vals = dValA.as<uint32_t>();
            a.req_count = cnt + 0;
            a.req_fail_base = cnt + 4;
            a.req_cap = REQ_CAP;
            HIP_TRY(hipMemset(cnt + 4, 0xFF, 4));  // no reservation has failed yet
            const int first_adjust = hx::insert_table_first_adjust(v, a);
            rc = timed_insert((uint32_t)nb, first_adjust);
            if (rc != HNSW_OK) return rc;
            uint32_t c5[5];
            HIP_TRY(hipMemcpy(c5, dCnt.p, 20, hipMemcpyDeviceToHost));  // synchronises
            HIP_TRY(hipMemcpy(o_st.data(), dStatus.p, nb * 4, hipMemcpyDeviceToHost));
            // A point that filled its visited table filed nothing: it runs again in this batch with a larger
            // table (the result of a search does not depend on the table's size, and the points of a batch do
            // not see each other anyway).  Not when a reservation failed: then the counter is past the capacity.
            std::vector<NodeID> again;
            for (size_t i = 0; i < nb; i++) {
                if (o_st[i] == HNSW_OK) continue;
                if (o_st[i] == HNSW_ERR_OVERFLOW && c5[4] == UINT32_MAX) {
                    again.push_back(batch[i]);
                } else {
                    failed.push_back(batch[i]);  // filed nothing; CPU path after the build
                }
            }
            if (!again.empty()) {
                HIP_TRY(hipMemcpy(dIds.p, again.data(), again.size() * 4, hipMemcpyHostToDevice));
                rc = timed_insert((uint32_t)again.size(), std::max(first_adjust, 0) + 1);
                if (rc != HNSW_OK) return rc;
                HIP_TRY(hipMemcpy(c5, dCnt.p, 20, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(o_st.data(), dStatus.p, again.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < again.size(); i++)
                    if (o_st[i] != HNSW_OK) failed.push_back(again[i]);
                n_again += again.size();
            }
            memcpy(counts, c5, 16);
            // the records written: everything reserved, or the prefix below the first reservation that did not fit
            nreq = std::min(c5[0], c5[4]);
            if (nreq > REQ_CAP) {
                set_error("on-device build: record counter %u beyond the capacity %u", nreq, REQ_CAP);
                return HNSW_ERR_OVERFLOW;
            }
            nreq_all = nreq;
        } else {
            // the slot: [count, nfail, ...64 B][failed ids][keys][vals]
            unsigned char *slot = sh->d_send;
            uint32_t *hdr = reinterpret_cast<uint32_t *>(slot);
            uint32_t *fail_ids = reinterpret_cast<uint32_t *>(slot + SH_HEADER);
            const size_t o_keys = SH_HEADER + SH_FAILCAP * 4, o_vals = o_keys + (size_t)SLOT_REC * 8;
            HIP_TRY(hipMemset(slot, 0, SH_HEADER));
            HIP_TRY(hipMemset(hdr + 2, 0xFF, 4));  // header: [count, nfail, first failing base, ...]
            a.req_keys = reinterpret_cast<uint64_t *>(slot + o_keys);
            a.req_vals = reinterpret_cast<uint32_t *>(slot + o_vals);
            a.req_count = hdr;
            a.req_fail_base = hdr + 2;
            a.req_cap = SLOT_REC;
            a.emit_own = 1;
            const int first_adjust = hx::insert_table_first_adjust(v, a);
            if (nb) rc = timed_insert((uint32_t)nb, first_adjust);
            if (rc != HNSW_OK) return rc;
            if (nb) HIP_TRY(hipMemcpy(o_st.data(), dStatus.p, nb * 4, hipMemcpyDeviceToHost));  // synchronises
            std::vector<uint32_t> myfail, again;
            uint32_t h3a[3] = {0, 0, UINT32_MAX};
            if (nb) HIP_TRY(hipMemcpy(h3a, hdr, 12, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nb; i++) {
                if (o_st[i] == HNSW_OK) continue;
                if (o_st[i] == HNSW_ERR_OVERFLOW && h3a[2] == UINT32_MAX) {
                    again.push_back(batch[s_lo + i]);  // filled its visited table: once more with a larger one
                } else {
                    myfail.push_back(batch[s_lo + i]);
                }
            }
            if (!again.empty()) {
                HIP_TRY(hipMemcpy(dIds.p, again.data(), again.size() * 4, hipMemcpyHostToDevice));
                rc = timed_insert((uint32_t)again.size(), std::max(first_adjust, 0) + 1);
                if (rc != HNSW_OK) return rc;
                HIP_TRY(hipMemcpy(o_st.data(), dStatus.p, again.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < again.size(); i++)
                    if (o_st[i] != HNSW_OK) myfail.push_back(again[i]);
                n_again += again.size();
            }
            if (myfail.size() > SH_FAILCAP) {
                set_error("sharded build: %zu points of one batch failed on the device", myfail.size());
                return HNSW_ERR_OVERFLOW;
            }
            const uint32_t nf = (uint32_t)myfail.size();
            {  // the count the other ranks read: the records really written (see hx_insert_kernel's reservation)
                uint32_t h3[3];
                HIP_TRY(hipMemcpy(h3, hdr, 12, hipMemcpyDeviceToHost));
                const uint32_t written = std::min(h3[0], h3[2]);
                if (written != h3[0]) HIP_TRY(hipMemcpy(hdr, &written, 4, hipMemcpyHostToDevice));
            }
            HIP_TRY(hipMemcpy(hdr + 1, &nf, 4, hipMemcpyHostToDevice));
            if (nf) HIP_TRY(hipMemcpy(fail_ids, myfail.data(), nf * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipDeviceSynchronize());
            // ---- the one exchange of the batch ----
            rc = sh->allgather(sh->ctx, REC_BYTES);
            if (rc != 0) {
                set_error("sharded build: the all-gather callback failed (%d)", rc);
                return HNSW_ERR_RCCL;
            }
            // concatenate the slots' records; every rank sees the same list
            for (uint32_t r = 0; r < W; r++) {
                const unsigned char *rs = sh->d_recv + (size_t)r * REC_BYTES;
                uint32_t rh[2];
                HIP_TRY(hipMemcpy(rh, rs, 8, hipMemcpyDeviceToHost));
                if (rh[0] > SLOT_REC || rh[1] > SH_FAILCAP || nreq + rh[0] > REQ_CAP) {
                    set_error("sharded build: malformed slot from rank %u", r);
                    return HNSW_ERR_ARG;
                }
                if (rh[1]) {
                    std::vector<uint32_t> f(rh[1]);
                    HIP_TRY(hipMemcpy(f.data(), rs + SH_HEADER, rh[1] * 4, hipMemcpyDeviceToHost));
                    failed.insert(failed.end(), f.begin(), f.end());
                }
                if (rh[0] && own_rows) {
                    // this rank's share of the slot's records: the rows it owns (the others' never reach its sort)
                    rc = hx::filter_edge_records(reinterpret_cast<const uint64_t *>(rs + o_keys), reinterpret_cast<const uint32_t *>(rs + o_vals),
                                                 rh[0], sh->rank, W, dKeyA.as<uint64_t>(), dValA.as<uint32_t>(), cnt + 0, REQ_CAP,
                                                 reinterpret_cast<int32_t *>(cnt + 3), nullptr);
                    if (rc != HNSW_OK) return rc;
                } else if (rh[0]) {
                    HIP_TRY(hipMemcpyAsync(dKeyA.as<uint64_t>() + nreq, rs + o_keys, (size_t)rh[0] * 8,
                                           hipMemcpyDeviceToDevice, nullptr));
                    HIP_TRY(hipMemcpyAsync(dValA.as<uint32_t>() + nreq, rs + o_vals, (size_t)rh[0] * 4,
                                           hipMemcpyDeviceToDevice, nullptr));
                }
                nreq += rh[0];
            }
            nreq_all = nreq;
            if (own_rows) HIP_TRY(hipMemcpy(&nreq, cnt + 0, 4, hipMemcpyDeviceToHost));  // (the counter was zeroed with the batch)
        }
        auto t1 = std::chrono::steady_clock::now();
        // ---- phase 2: group by target row (radix sort), append / prune ----
        rc = hx::sort_edge_pairs(dTemp.p, temp_bytes, dKeyA.as<uint64_t>(), dKeyB.as<uint64_t>(),
                                 dValA.as<uint32_t>(), dValB.as<uint32_t>(), nreq, L, nullptr);
        if (rc != HNSW_OK) return rc;
        hx::ConnectArgs ca{};
        ca.keys = dKeyB.as<uint64_t>();
        ca.vals = dValB.as<uint32_t>();
        ca.count = nreq;
        ca.m = m;
        ca.adj0_mut = h->dev.adj0_mut();
        ca.adj_up_mut = h->dev.adj_up_mut();
        ca.adjd0_mut = adjd0;
        ca.adjd_up_mut = adjd_up;
        ca.out_keys = dKeyA.as<uint64_t>();  // the unsorted requests are dead by now
        ca.out_count = cnt + 1;
        ca.out_cap = REQ_CAP;
        ca.status = reinterpret_cast<int32_t *>(cnt + 3);
        if (own_rows) {
            ca.own_rank = sh->rank;
            ca.own_world = W;
            ca.chg_keys = dChg.as<uint64_t>();
            ca.chg_count = dChgCnt.as<uint32_t>();
            ca.chg_cap = CHG_LIST_CAP;
        }
        HIP_TRY(hipEventRecord(ev_conn.a, nullptr));
        rc = hx::launch_connect(v, ca, nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipEventRecord(ev_conn.b, nullptr));
        HIP_TRY(hipMemcpy(counts, dCnt.p, 16, hipMemcpyDeviceToHost));
        {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev_conn.a, ev_conn.b));
            t_conn_kernel_ms += ms;
        }
        if (counts[3] != 0 && !own_rows) {
            set_error("on-device build: connect kernel reported status %d in batch %zu", (int)counts[3], n_batches);
            return (int)counts[3];
        }
        uint32_t nrem = counts[1], nrem_all = counts[1];
        if (own_rows) {
            // the removals of every owner's prunes, in every rank's list (phase 3 filters by the owner of the row
            // that loses the edge); the sort below makes the order canonical
            const int32_t st = counts[3] != 0 ? (int32_t)counts[3] : (nrem > REQ_CAP ? HNSW_ERR_OVERFLOW : 0);
            rc = exchange(st ? 0 : nrem, 8, st, dKeyA.p, "removals");
            if (rc != HNSW_OK) return rc;
            uint64_t tot = 0;
            for (uint32_t r = 0; r < W; r++) tot += x_counts[r];
            if (tot > REQ_CAP) {
                set_error("sharded build: %llu removals in one batch, room for %u", (unsigned long long)tot, REQ_CAP);
                return HNSW_ERR_OVERFLOW;
            }
            // ... of which this rank sorts and applies those that drop from a row it owns
            HIP_TRY(hipMemset(cnt + 1, 0, 4));
            for (uint32_t r = 0; r < W; r++) {
                if (x_counts[r] == 0) continue;
                rc = hx::filter_edge_records(reinterpret_cast<const uint64_t *>(sh->d_recv + (size_t)r * x_stride + SH_HEADER), nullptr,
                                             x_counts[r], sh->rank, W, dKeyA.as<uint64_t>(), nullptr, cnt + 1, REQ_CAP,
                                             reinterpret_cast<int32_t *>(cnt + 3), nullptr);
                if (rc != HNSW_OK) return rc;
            }
            nrem_all = (uint32_t)tot;
            HIP_TRY(hipMemcpy(&nrem, cnt + 1, 4, hipMemcpyDeviceToHost));
        }
        // ---- phase 3: group the removals by row, drop the reverse edges ----
        rc = hx::sort_edge_keys(dTemp.p, temp_bytes, dKeyA.as<uint64_t>(), dKeyB.as<uint64_t>(), nrem, L, nullptr);
        if (rc != HNSW_OK) return rc;
        ca.keys = dKeyB.as<uint64_t>();
        ca.vals = nullptr;
        ca.count = nrem;
        ca.out_keys = dRef.as<uint64_t>();
        ca.out_count = cnt + 2;
        ca.out_cap = REF_CAP;
        HIP_TRY(hipEventRecord(ev_rem.a, nullptr));
        rc = hx::launch_remove(v, ca, nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipEventRecord(ev_rem.b, nullptr));
        rem_pending = true;
        if (own_rows) {
            // ---- the rows this rank changed, to the other replicas ----
            uint32_t c4[4], lists[hx::HX_CHG_LISTS];
            HIP_TRY(hipMemcpy(lists, dChgCnt.p, sizeof(lists), hipMemcpyDeviceToHost));  // synchronises
            HIP_TRY(hipMemcpy(c4, dCnt.p, 16, hipMemcpyDeviceToHost));
            int32_t st = (int32_t)c4[3];
            uint64_t nchg64 = 0;
            uint32_t longest = 0;
            for (uint32_t c : lists) {
                nchg64 += c;
                longest = std::max(longest, c);
            }
            if (st == 0 && (longest > CHG_LIST_CAP || nchg64 > CHG_CAP || (uint64_t)SH_HEADER + nchg64 * SHIP_UNIT > sh->slot_bytes))
                st = HNSW_ERR_OVERFLOW;
            const uint32_t nchg = st ? 0 : (uint32_t)nchg64;
            if (st == 0) {
                rc = hx::launch_pack_rows(v, h->dev.adj0_mut(), h->dev.adj_up_mut(), dChg.as<uint64_t>(), dChgCnt.as<uint32_t>(),
                                          CHG_LIST_CAP, longest, SHIP, sh->d_send + SH_HEADER, nullptr);
                if (rc != HNSW_OK) st = rc;
            }
            rc = exchange(st ? 0 : nchg, SHIP_UNIT, st, nullptr, "changed rows");
            if (rc != HNSW_OK) return rc;
            n_owned += nchg;
            for (uint32_t r = 0; r < W; r++) {
                if (r == sh->rank || x_counts[r] == 0) continue;
                rc = hx::launch_apply_rows(v, h->dev.adj0_mut(), h->dev.adj_up_mut(), sh->d_recv + (size_t)r * x_stride + SH_HEADER,
                                           x_counts[r], SHIP, reinterpret_cast<int32_t *>(cnt + 3), nullptr);
                if (rc != HNSW_OK) return rc;
                n_shipped += x_counts[r];
            }
            // the receive buffer is read by those launches: they finish before the next batch's exchange overwrites it
            HIP_TRY(hipDeviceSynchronize());
        }
        if (verbose) HIP_TRY(hipDeviceSynchronize());  // only to attribute the time
        auto t2 = std::chrono::steady_clock::now();
        t_ins += std::chrono::duration<double>(t1 - t0).count();
        t_conn += std::chrono::duration<double>(t2 - t1).count();
        n_req += nreq_all;
        n_rem += nrem_all;
        pos += B;
        connected += B;
        n_batches++;
        if (verbose && (n_batches % 64 == 0 || pos == order.size()))
            fprintf(stderr, "\rBuilding HNSW index on the GPU %zu/%zu", pos, order.size());
    }

    if ((rc = collect_remove_time())) return rc;
    {  // what the build read and how long its kernels ran (hnsw_get_stat "build_*")
        unsigned long long rd[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(rd, dRead.p, 32, hipMemcpyDeviceToHost));
        hnsw_index::BuildStats &bs = h->build;
        bs.points += order.size();
        bs.batches += n_batches;
        bs.rows_read += rd[0];
        bs.adj_rows += rd[1];
        bs.adj_ids += rd[2];
        bs.records += n_req;
        bs.removals += n_rem;
        bs.insert_kernel_s += t_kernel_ms * 1e-3;
        bs.connect_kernel_s += t_conn_kernel_ms * 1e-3;
        bs.insert_phase_s += t_ins;
        bs.connect_s += t_conn;
        bs.rows_owned += n_owned;
        bs.rows_received += n_shipped;
        bs.exchange_bytes += x_bytes;
        bs.exchange_s += t_exchange;
    }
    // ---- the host graph from the device arrays ----
    if (row_reserve.t.joinable()) row_reserve.t.join();
    const auto t_sync0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpy(counts, dCnt.p, 16, hipMemcpyDeviceToHost));
    if (counts[3] != 0 || counts[2] > REF_CAP) {
        set_error("on-device build: status %d, %u kept-last-edge records", (int)counts[3], counts[2]);
        return HNSW_ERR_OVERFLOW;
    }
    std::vector<uint64_t> refusals(counts[2]);
    if (counts[2]) HIP_TRY(hipMemcpy(refusals.data(), dRef.p, (size_t)counts[2] * 8, hipMemcpyDeviceToHost));
    if (own_rows) {  // every owner's kept-last-edge records, on every rank (the host restores the mirror edges below)
        rc = exchange(counts[2], 8, 0, dRef.p, "kept-last-edge records");
        if (rc != HNSW_OK) return rc;
        refusals.clear();
        for (uint32_t r = 0; r < W; r++) {
            if (x_counts[r] == 0) continue;
            const size_t at = refusals.size();
            refusals.resize(at + x_counts[r]);
            HIP_TRY(hipMemcpy(refusals.data() + at, sh->d_recv + (size_t)r * x_stride + SH_HEADER, (size_t)x_counts[r] * 8,
                              hipMemcpyDeviceToHost));
        }
        if (verbose)
            fprintf(stderr, "\nsharded build, rank %u of %u: phases 2 / 3 on the rows it owns; %zu rows received, %.1f MB through the "
                            "variable-size exchanges in %.2f s\n", sh->rank, W, n_shipped, x_bytes / 1e6, t_exchange);
    }
    for (int pass = 0; pass < 2; pass++) {
        std::vector<std::vector<NodeID>> &rowsv = pass == 0 ? host.adj0 : host.adj_up;
        const uint32_t S = pass == 0 ? v.S0 : v.S1;
        const size_t R = rowsv.size();
        if (R == 0) continue;
        // pieces of the array arrive through pinned buffers; the threads turn each into the host's rows while
        // the next one is on the wire
        rc = h->dev.read_adjacency(pass, R, [&](uint64_t plo, uint64_t phi, const uint32_t *data) {
            auto fill = [&](uint64_t lo, uint64_t hi) {
                std::vector<NodeID> ids_of_row(S);
                for (uint64_t r = lo; r < hi; r++) {
                    const uint32_t *src = data + (r - plo) * (size_t)S;
                    uint32_t deg = 0;
                    for (uint32_t k = 0; k < S; k++)
                        if (src[k] != UINT32_MAX) ids_of_row[deg++] = src[k];
                    rowsv[r].assign(ids_of_row.begin(), ids_of_row.begin() + deg);  // one allocation of the row's size
                }
            };
            const uint64_t cnt = phi - plo;
            const unsigned nt = (unsigned)std::min<uint64_t>(nb_threads, std::max<uint64_t>(1, cnt / 16384));
            if (nt <= 1) {
                fill(plo, phi);
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++) th.emplace_back(fill, plo + cnt * t / nt, plo + cnt * (t + 1) / nt);
                for (auto &t : th) t.join();
            }
        });
        if (rc != HNSW_OK) return rc;
    }
    // an edge x -> nb that stayed because it was x's last one: restore nb -> x (graph.rs:85-94 keeps both)
    refusals.insert(refusals.end(), restore.begin(), restore.end());
    const uint64_t id_mask = (1ull << hx::HX_EDGE_ID_BITS) - 1;
    std::vector<uint64_t> touched;  // rows the host changes after the read-back: (layer << 32) | id
    for (uint64_t key : refusals) {
        const uint32_t layer = (uint32_t)(key >> (2 * hx::HX_EDGE_ID_BITS));
        const NodeID x = (NodeID)((key >> hx::HX_EDGE_ID_BITS) & id_mask), nb = (NodeID)(key & id_mask);
        std::vector<NodeID> &row = host.row(layer, nb);
        const std::vector<NodeID> &back = host.row(layer, x);
        if (std::find(back.begin(), back.end(), nb) != back.end() &&
            std::find(row.begin(), row.end(), x) == row.end()) {
            row.push_back(x);
            touched.push_back(((uint64_t)layer << 32) | nb);
        }
    }
    const bool device_is_the_graph = failed.empty();
    // points the kernel could not serve take the CPU path
    if (!failed.empty()) {
        std::sort(failed.begin(), failed.end());
        for (NodeID p : failed) {
            rc = host.insert(p, *ins);
            if (rc != HNSW_OK) return rc;
        }
    }
    if (verbose) {
        const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        const double t_sync = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sync0).count();
        fprintf(stderr,
                "\non-device build (device connect): %zu batches in %.2f s (insert kernel %.2f s, sort + connect + "
                "remove %.2f s, graph read-back %.2f s); %zu requests, %zu removals, %u kept-last-edge, %zu points "
                "ran again with a larger visited table, %zu took the CPU path\n",
                n_batches, tot, t_ins, t_conn, t_sync, n_req, n_rem, counts[2], n_again, failed.size());
        auto secs = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
            return std::chrono::duration<double>(y - x).count();
        };
        fprintf(stderr,
                "host phases: store_points %.2f s, levels + order %.2f s, (row clamp %.2f s,) first %llu points on the CPU beside the "
                "upload %.2f s, buffers %.2f s, batch loop %.2f s\n",
                secs(t_enter, t_stored), secs(t_stored, t_start), secs(t_start, t_clamped), (unsigned long long)SEED,
                secs(t_clamped, t_uploaded), secs(t_uploaded, t_loop0), secs(t_loop0, t_sync0));
    }
    host.version++;
    // the adjacency in HBM is the graph just read back: patch the few rows changed since and keep the snapshot
    // (a build of tens of GB is otherwise followed by an upload of the same tens of GB)
    if (device_is_the_graph && !(getenv("HNSW_MI355X_REUPLOAD") && atoi(getenv("HNSW_MI355X_REUPLOAD")) != 0))
        (void)h->dev.refresh_rows(host, touched);
    return HNSW_OK;
}

// An on-device build that returns an error after it stored the points leaves some of them unconnected
// (and, in the device-connect form, the graph only in HBM): the handle is marked and refuses further use.
// Errors raised before anything was stored (bad rows, bad arguments) leave the index as it was.
template <class F>
int device_build_guard(hnsw_index *h, F &&build) {
    const uint64_t n_before = h->host->len();
    const int rc = build();
    if (rc != HNSW_OK && h->host->len() != n_before) h->incomplete_build = true;
    return rc;
}

}  // namespace

extern "C" {

const char *hnsw_last_error(void) { return hx::get_error(); }
const char *hnsw_version(void) { return "hnsw_mi355x 0.1 (gfx950)"; }

int hnsw_create(uint32_t m, uint32_t ef_cons, uint32_t dim, int vec_kind, hnsw_index **out) {
    if (!out || m < 2 || dim == 0 || (vec_kind != HNSW_VEC_QUANT8 && vec_kind != HNSW_VEC_F32)) {
        set_error("hnsw_create: need m >= 2, dim >= 1 and a valid vector kind");
        return HNSW_ERR_ARG;
    }
    hnsw_index *h = new (std::nothrow) hnsw_index();
    if (!h) return HNSW_ERR_OOM;
    h->host.reset(new hx::HostIndex(m, ef_cons, dim, vec_kind));
    *out = h;
    return HNSW_OK;
}

void hnsw_free(hnsw_index *h) { delete h; }

int hnsw_clone(const hnsw_index *h, hnsw_index **out) {
    if (!h || !out) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_clone");
    hnsw_index *c = new (std::nothrow) hnsw_index();
    if (!c) return HNSW_ERR_OOM;
    c->host.reset(new hx::HostIndex(*h->host));
    c->device = h->device;
    // the handle's options travel with the clone (a cosine index that forgot its metric would stop normalising
    // its queries)
    c->cosine = h->cosine;
    c->gpu_build = h->gpu_build;
    c->build_batch_max = h->build_batch_max;
    c->build_batch_div = h->build_batch_div;
    c->dev.inline_rows = h->dev.inline_rows;
    c->dev.fat_budget_bytes = h->dev.fat_budget_bytes;
    c->co.window_us.store(h->co.window_us.load());
    c->co.depth = h->co.depth;
    c->co.cap = h->co.cap;
    *out = c;
    return HNSW_OK;
}

int hnsw_get_params(const hnsw_index *h, hnsw_params *out) {
    if (!h || !out) return HNSW_ERR_ARG;
    const hx::Params &p = h->host->params;
    memset(out, 0, sizeof(*out));
    out->ep = is_replica(h) ? h->dev.view.ep : p.ep;
    out->vec_kind = (uint32_t)h->host->kind;
    out->m = p.m;
    out->mmax = p.mmax;
    out->mmax0 = p.mmax0;
    out->ml = p.ml;
    out->ef_cons = p.ef_cons;
    out->dim = p.dim;
    return HNSW_OK;
}

int hnsw_set_ep(hnsw_index *h, uint32_t ep) {
    if (h && is_replica(h)) return reject_replica(h, "hnsw_set_ep");
    if (!h || ep >= h->host->len()) {
        set_error("entry point %u out of range", ep);
        return HNSW_ERR_ARG;
    }
    h->host->params.ep = ep;
    h->host->version++;
    return HNSW_OK;
}

int hnsw_insert_bulk(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose) {
    return hnsw_insert_bulk_levels(h, rows, n, nb_threads, verbose, nullptr);
}
int hnsw_insert_bulk_levels(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads,
                            int verbose, const uint8_t *levels) {
    if (!h || !rows) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_insert_bulk");
    if (h->incomplete_build) return check_search_args(h, 1);
    std::vector<float> unit;
    if (int crc = cosine_rows(h, rows, n, unit, nb_threads)) return crc;
    if (h->gpu_build == 2) return device_build_guard(h, [&] { return gpu_insert_bulk_full(h, rows, n, nb_threads, verbose, levels); });
    if (h->gpu_build) return device_build_guard(h, [&] { return gpu_insert_bulk(h, rows, n, nb_threads, verbose, levels); });
    return h->host->insert_bulk(rows, n, nb_threads, verbose != 0, levels);
}
uint64_t hnsw_sharded_slot_bytes(const hnsw_index *h, uint32_t world) {
    if (!h || world == 0) return 0;
    return shard_slot_bytes((uint32_t)h->host->params.m, world);
}
int hnsw_insert_bulk_sharded(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads, int verbose,
                             const uint8_t *levels, uint32_t rank, uint32_t world, void *d_send, void *d_recv,
                             uint64_t slot_bytes, hnsw_allgather_fn allgather, void *ctx) {
    if (!h || !rows || world == 0 || rank >= world) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_insert_bulk_sharded");
    std::vector<float> unit;
    if (int crc = cosine_rows(h, rows, n, unit, nb_threads)) return crc;
    ShardCtx sh{rank, world, static_cast<unsigned char *>(d_send), static_cast<unsigned char *>(d_recv), slot_bytes,
                allgather, ctx};
    if (h->incomplete_build) return check_search_args(h, 1);
    return device_build_guard(h, [&] { return gpu_insert_bulk_full(h, rows, n, nb_threads, verbose, levels, &sh); });
}
int hnsw_insert_bulk_device(hnsw_index *h, const float *rows, uint64_t n, uint32_t nb_threads,
                            int verbose, const uint8_t *levels) {
    if (!h || !rows) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_insert_bulk_device");
    if (h->incomplete_build) return check_search_args(h, 1);
    std::vector<float> unit;
    if (int crc = cosine_rows(h, rows, n, unit, nb_threads)) return crc;
    if (h->gpu_build == 1) return device_build_guard(h, [&] { return gpu_insert_bulk(h, rows, n, nb_threads, verbose, levels); });
    return device_build_guard(h, [&] { return gpu_insert_bulk_full(h, rows, n, nb_threads, verbose, levels); });
}
int hnsw_insert_vec(hnsw_index *h, const float *v, uint32_t *out_id) {
    return hnsw_insert_vec_level(h, v, -1, out_id);
}
int hnsw_insert_vec_level(hnsw_index *h, const float *v, int level, uint32_t *out_id) {
    if (!h || !v || level > 255) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_insert_vec");
    if (h->incomplete_build) return check_search_args(h, 1);
    std::vector<float> unit;
    if (int crc = cosine_rows(h, v, 1, unit)) return crc;
    // The reference's callers search right after an insert_vec (eval_glove/src/main.rs:37-41).  When the HBM snapshot
    // was current before the insertion it is patched -- the new row and the adjacency rows the insertion touched --
    // instead of being thrown away and uploaded again by the next search (DeviceIndex::append_point).
    std::lock_guard<std::mutex> g(h->mu);
    const bool live = h->dev.valid && h->dev.current(*h->host) &&
                      !(getenv("HNSW_MI355X_REUPLOAD") && atoi(getenv("HNSW_MI355X_REUPLOAD")) != 0);
    std::vector<uint64_t> touched;
    uint32_t id = 0;
    int rc;
    {
        hx::DirtyScope scope(live ? &touched : nullptr);
        rc = h->host->insert_vec(v, level, &id);
    }
    if (rc != HNSW_OK) return rc;
    if (out_id) *out_id = id;
    if (live) {
        if (h->dev.append_point(*h->host, id, touched))
            h->n_point_patches.fetch_add(1, std::memory_order_relaxed);
        else
            h->n_patch_fallbacks.fetch_add(1, std::memory_order_relaxed);
    }
    return HNSW_OK;
}
int hnsw_import_points(hnsw_index *h, const float *rows, uint64_t n, const uint8_t *levels) {
    if (!h || !rows) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_import_points");
    std::vector<float> unit;
    if (int crc = cosine_rows(h, rows, n, unit)) return crc;
    return h->host->import_points(rows, n, levels);
}
int hnsw_import_layer(hnsw_index *h, uint32_t layer, uint64_t n_nodes, const uint32_t *node_ids,
                      const uint64_t *offsets, const uint32_t *nbrs) {
    if (!h || !node_ids || !offsets) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_import_layer");
    return h->host->import_layer(layer, n_nodes, node_ids, offsets, nbrs);
}

// ---- query -------------------------------------------------------------------------------------
int hnsw_search(hnsw_index *h, const float *q, uint32_t n, uint32_t ef, uint32_t *ids,
                uint32_t *count) {
    // concurrent callers are gathered into one launch (the coalescer above); ef beyond the register-resident list
    // (the HBM-spill kernel) and result lists of thousands of ids go by themselves
    if (h && q && ids && n > 0 && n <= 1024 && ef <= 1024 && h->co.window_us.load(std::memory_order_relaxed) >= 0) {
        const int rc = check_search_args(h, ef);
        if (rc != HNSW_OK) return rc;
        return search_coalesced(h, q, n, ef, ids, count);
    }
    return hnsw_search_batch(h, q, 1, n, ef, ids, nullptr, count, nullptr);
}

int hnsw_search_batch(hnsw_index *h, const float *Q, uint64_t nq, uint32_t n, uint32_t ef,
                      uint32_t *ids, float *dists, uint32_t *counts, hnsw_query_stats *stats) {
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if (nq == 0) return HNSW_OK;
    if (!Q || !ids || nq > 0x7FFFFFFFull) return HNSW_ERR_ARG;
    if (n == 0) {
        if (counts) memset(counts, 0, nq * 4);
        return HNSW_OK;
    }
    hx::DevView dummy{};
    dummy.nb_layers = hnsw_layer_count(h);  // the host index's, or the adopted snapshot's for a replica
    hx::SearchArgs a = ann_args(dummy, nullptr, n, ef, nullptr, nullptr, nullptr, nullptr);
    return search_host(h, a, Q, nq, ids, dists, counts, stats, nullptr);
}

int hnsw_search_batch_device(hnsw_index *h, const float *d_Q, uint64_t nq, uint32_t n, uint32_t ef,
                             uint32_t *d_ids, float *d_dists, uint32_t *d_counts,
                             hnsw_query_stats *d_stats, void *stream) {
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if (nq == 0 || n == 0) return HNSW_OK;
    if (!d_Q || !d_ids || !d_stats || nq > 0x7FFFFFFFull) return HNSW_ERR_ARG;
    rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    DeviceQueries dq;
    if ((rc = dq.prepare(h, d_Q, nq, static_cast<hipStream_t>(stream)))) return rc;
    hx::SearchArgs a = ann_args(h->dev.view, dq.q, n, ef, d_ids, d_dists, d_counts, d_stats);
#ifdef HX_STAMPS
    a.dbg = reinterpret_cast<unsigned long long *>(getenv("HX_DBG_PTR") ? strtoull(getenv("HX_DBG_PTR"), nullptr, 0) : 0);
#endif
    return hx::launch_search(h->dev.view, a, (uint32_t)nq, 0, static_cast<hipStream_t>(stream));
}

// Completes a hnsw_search_batch_device call: waits for the stream, reads the per-query statuses, re-runs
// the queries whose visited table filled up with a table twice the size (same arithmetic, same result as
// if the larger table had been used from the start) and reports the first remaining per-query error.
int hnsw_search_batch_device_finish(hnsw_index *h, const float *d_Q, uint64_t nq, uint32_t n, uint32_t ef,
                                    uint32_t *d_ids, float *d_dists, uint32_t *d_counts,
                                    hnsw_query_stats *d_stats, void *stream_v) {
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if (nq == 0 || n == 0) return HNSW_OK;
    if (!d_Q || !d_ids || !d_stats || nq > 0x7FFFFFFFull) return HNSW_ERR_ARG;
    rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const hx::DevView &v = h->dev.view;
    const size_t st_bytes = nq * sizeof(hnsw_query_stats);
    ScratchLease lease(h);
    if ((rc = lease.prepare(h->dev.device, align256(nq * 4), st_bytes))) return rc;
    SearchScratch &s = *lease.s;
    hnsw_query_stats *st = static_cast<hnsw_query_stats *>(s.pin);
    DeviceQueries dq;  // a re-run reads the queries again: the unit-length copy under the cosine option
    if ((rc = dq.prepare(h, d_Q, nq, stream))) return rc;
    hx::SearchArgs a = ann_args(v, dq.q, n, ef, d_ids, d_dists, d_counts, d_stats);
    uint32_t slots = hx::default_slots_log2(ef, v.S0);
    std::vector<uint32_t> sel;
    while (true) {
        HIP_TRY(hipMemcpyAsync(st, d_stats, st_bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        sel.clear();
        for (uint64_t i = 0; i < nq; i++)
            if (st[i].status == HNSW_ERR_OVERFLOW) sel.push_back((uint32_t)i);
        if (sel.empty() || slots >= hx::max_slots_log2(ef)) break;
        slots++;
        HIP_TRY(hipMemcpyAsync(s.dev, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // `sel` is reused by the next round
        a.qsel = static_cast<const uint32_t *>(s.dev);
        rc = hx::launch_search(v, a, (uint32_t)sel.size(), slots, stream);
        if (rc != HNSW_OK) return rc;
    }
    for (uint64_t i = 0; i < nq; i++) {
        if (st[i].status != HNSW_OK) {
            set_error("query %llu failed with status %d%s", (unsigned long long)i, st[i].status,
                      st[i].status == HNSW_ERR_NAN_INPUT    ? " (NaN in the query or in a distance)"
                      : st[i].status == HNSW_ERR_OVERFLOW   ? " (visited table exhausted at its largest size)"
                                                             : "");
            return st[i].status;
        }
    }
    return HNSW_OK;
}

int hnsw_distance_batch(hnsw_index *h, const float *q, const uint32_t *ids, uint64_t k, float *out) {
    int rc = check_search_args(h, 1);
    if (rc != HNSW_OK) return rc;
    if (k == 0) return HNSW_OK;
    if (!q || !ids || !out) return HNSW_ERR_ARG;
    rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    const hx::DevView &v = h->dev.view;
    DevBuf dq, dids, dout, dst;
    if ((rc = dq.alloc(v.dim * 4)) || (rc = dids.alloc(k * 4)) || (rc = dout.alloc(k * 4)) ||
        (rc = dst.alloc(4)))
        return rc;
    HIP_TRY(hipMemcpy(dq.p, q, v.dim * 4, hipMemcpyHostToDevice));
    if ((rc = cosine_queries(h, dq.p, 1, nullptr))) return rc;
    HIP_TRY(hipMemcpy(dids.p, ids, k * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dst.p, 0, 4));
    rc = hx::launch_distance_batch(v, dq.as<float>(), dids.as<uint32_t>(), k, dout.as<float>(),
                                   dst.as<int32_t>(), nullptr);
    if (rc != HNSW_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    int32_t st = 0;
    HIP_TRY(hipMemcpy(&st, dst.p, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, dout.p, k * 4, hipMemcpyDeviceToHost));
    if (st != HNSW_OK) {
        set_error(st == HNSW_ERR_NAN_INPUT ? "NaN in the query" : "point id out of range");
        return st;
    }
    return HNSW_OK;
}

int hnsw_search_layer(hnsw_index *h, uint32_t layer, const float *q, const uint32_t *entry_ids,
                      uint32_t n_entry, uint32_t ef, uint32_t *out_ids, float *out_dists,
                      uint32_t *out_count, hnsw_query_stats *stats) {
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if (!q || !entry_ids || !out_ids || !out_count || n_entry == 0 || ef == 0) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_search_layer (the seam checks its entry set against the host graph)");
    if (layer >= hnsw_layer_count(h)) {
        set_error("Layer %u not found in the structure.", layer);  // layers.rs:25-30 panics
        return HNSW_ERR_ARG;
    }
    if (n_entry > ef) {
        set_error("search_layer seam: the entry set (%u) may not be larger than ef (%u)", n_entry, ef);
        return HNSW_ERR_ARG;
    }
    for (uint32_t i = 0; i < n_entry; i++) {
        if (!h->host->in_layer(layer, entry_ids[i])) {
            set_error("entry %u is not in layer %u", entry_ids[i], layer);
            return HNSW_ERR_NODE_NOT_IN_GRAPH;
        }
        for (uint32_t j = 0; j < i; j++)
            if (entry_ids[j] == entry_ids[i]) {
                set_error("duplicate entry %u", entry_ids[i]);
                return HNSW_ERR_ARG;
            }
    }
    hx::SearchArgs a{};
    a.n_entry = n_entry;
    a.layer_hi = a.layer_lo = (int32_t)layer;
    a.ef_upper = 1;
    a.ef_bottom = ef;
    a.n = ef;
    std::vector<uint32_t> ids(ef);
    std::vector<float> dists(ef);
    uint32_t count = 0;
    hnsw_query_stats st{};
    rc = search_host(h, a, q, 1, ids.data(), dists.data(), &count, &st, entry_ids);
    if (rc != HNSW_OK) return rc;
    for (uint32_t i = 0; i < count; i++) {
        out_ids[i] = ids[i];
        if (out_dists) out_dists[i] = dists[i];
    }
    *out_count = count;
    if (stats) *stats = st;
    return HNSW_OK;
}

int hnsw_brute_force(hnsw_index *h, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids,
                     float *dists) {
    int rc = check_search_args(h, 1);
    if (rc != HNSW_OK) return rc;
    if (nq == 0) return HNSW_OK;
    if (!Q || !ids || k == 0 || k > 64) {
        set_error("brute force supports 1 <= k <= 64");
        return HNSW_ERR_ARG;
    }
    rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    const hx::DevView &v = h->dev.view;
    const uint32_t nseg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(512, (v.n_points + 2047) / 2048));
    const uint64_t batch = 2048;
    DevBuf dQ, dIds, dDists, dSt;
    if ((rc = dQ.alloc(batch * v.dim * 4)) || (rc = dIds.alloc(batch * nseg * k * 4)) ||
        (rc = dDists.alloc(batch * nseg * k * 4)) || (rc = dSt.alloc(4)))
        return rc;
    std::vector<uint32_t> pid(batch * nseg * k);
    std::vector<float> pd(batch * nseg * k);
    std::vector<std::pair<float, uint32_t>> cand;
    for (uint64_t q0 = 0; q0 < nq; q0 += batch) {
        const uint64_t nb = std::min(batch, nq - q0);
        HIP_TRY(hipMemcpy(dQ.p, Q + q0 * v.dim, nb * v.dim * 4, hipMemcpyHostToDevice));
        if ((rc = cosine_queries(h, dQ.p, nb, nullptr))) return rc;
        HIP_TRY(hipMemset(dSt.p, 0, 4));
        rc = hx::launch_brute_force(v, dQ.as<float>(), nb, k, nseg, dIds.as<uint32_t>(),
                                    dDists.as<float>(), dSt.as<int32_t>(), nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        int32_t st = 0;
        HIP_TRY(hipMemcpy(&st, dSt.p, 4, hipMemcpyDeviceToHost));
        if (st != HNSW_OK) {
            set_error("NaN in a query or a distance");
            return st;
        }
        HIP_TRY(hipMemcpy(pid.data(), dIds.p, nb * nseg * k * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(pd.data(), dDists.p, nb * nseg * k * 4, hipMemcpyDeviceToHost));
        for (uint64_t qi = 0; qi < nb; qi++) {
            cand.clear();
            for (uint64_t j = 0; j < (uint64_t)nseg * k; j++) {
                const uint32_t id = pid[qi * nseg * k + j];
                if (id != UINT32_MAX) cand.emplace_back(pd[qi * nseg * k + j], id);
            }
            std::sort(cand.begin(), cand.end());  // (dist, id): Dist::cmp for non-NaN distances
            for (uint32_t j = 0; j < k; j++) {
                const bool have = j < cand.size();
                ids[(q0 + qi) * k + j] = have ? cand[j].second : UINT32_MAX;
                if (dists) dists[(q0 + qi) * k + j] = have ? cand[j].first : INFINITY;
            }
        }
    }
    return HNSW_OK;
}

// Ground truth on the matrix cores (brute_mfma.hip): MFMA scores screen every point, the k + 8 best per
// query are re-evaluated in the reference's exact arithmetic and sorted by (dist, id).  Not bit-exact by
// construction (the screen could in principle lose a true neighbour to rounding): hnsw_brute_force is the
// exact scan; this one is for ground truth at sizes where the exact scan takes minutes.
int hnsw_brute_force_fast(hnsw_index *h, const float *Q, uint64_t nq, uint32_t k, uint32_t *ids, float *dists) {
    int rc = check_search_args(h, 1);
    if (rc != HNSW_OK) return rc;
    if (nq == 0) return HNSW_OK;
    const uint32_t K2 = hx::brute_mfma_k2();
    if (!Q || !ids || k == 0 || k + 8 > K2) {
        set_error("the MFMA scan supports 1 <= k <= %u", K2 - 8);
        return HNSW_ERR_ARG;
    }
    rc = ensure_uploaded(h);
    if (rc != HNSW_OK) return rc;
    const hx::DevView &v = h->dev.view;
    if (v.kind != HNSW_VEC_F32 || (v.dim & 3u)) {
        set_error("the MFMA scan serves f32 rows whose dimension is a multiple of 4");
        return HNSW_ERR_ARG;
    }
    // A NaN in a query makes every screen score NaN, and `score < threshold` is false for NaN: the screen would
    // keep nothing and the call would return padding ids with status OK where the exact scan reports
    // HNSW_ERR_NAN_INPUT (stored rows cannot hold one: insert rejects them)
    for (uint64_t i = 0; i < nq * (uint64_t)v.dim; i++)
        if (Q[i] != Q[i]) {
            set_error("query %llu: NaN in the query", (unsigned long long)(i / v.dim));
            return HNSW_ERR_NAN_INPUT;
        }
    const uint64_t batch = 2048;  // queries per launch: 64 tiles of 32
    // (buffers sized for the tiles a launch really has: a 32-query call on a multi-million-point index has one
    // tile and up to 768 segments, not 64 tiles of them)
    const uint32_t ntile_max = (uint32_t)((std::min(batch, nq) + 31) / 32);
    // enough workgroups for the chip (256 CUs, one 98-KB query tile each at d = 768) without cutting the
    // points into segments shorter than a few tiles per wave
    // about 768 workgroups per launch (256 CUs, up to three 32-query tiles of a small dimension each), but
    // no segment shorter than a few point tiles per wave; fewer segments = fewer partial lists to merge
    const uint32_t ntile_first = (uint32_t)((std::min(batch, nq) + 31) / 32);
    const uint32_t nseg = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>(std::max<uint32_t>(1, 768 / ntile_first), (uint64_t)v.n_points / 4096));
    const uint32_t M = k + 8;
    const size_t per_tile = (size_t)nseg * 4 * 64 * K2;
    DevBuf dQ, dXn, dS, dI, dQi, dPi, dOut;
    if ((rc = dQ.alloc(batch * v.dim * 4)) || (rc = dXn.alloc((size_t)v.n_points * 4)) ||
        (rc = dS.alloc(ntile_max * per_tile * 4)) || (rc = dI.alloc(ntile_max * per_tile * 4)) ||
        (rc = dQi.alloc(batch * M * 4)) || (rc = dPi.alloc(batch * M * 4)) || (rc = dOut.alloc(batch * M * 4)))
        return rc;
    rc = hx::launch_row_norms(v, dXn.as<float>(), nullptr);
    if (rc != HNSW_OK) return rc;
    std::vector<float> hs(ntile_max * per_tile), hd(batch * M);
    std::vector<uint32_t> hi(ntile_max * per_tile), qidx(batch * M), pidx(batch * M);
    std::vector<std::pair<float, uint32_t>> cand;
    for (uint64_t q0 = 0; q0 < nq; q0 += batch) {
        const uint64_t nb = std::min(batch, nq - q0);
        const uint32_t ntile = (uint32_t)((nb + 31) / 32);
        HIP_TRY(hipMemcpy(dQ.p, Q + q0 * v.dim, nb * v.dim * 4, hipMemcpyHostToDevice));
        if ((rc = cosine_queries(h, dQ.p, nb, nullptr))) return rc;
        rc = hx::launch_brute_mfma(v, dXn.as<float>(), dQ.as<float>(), (uint32_t)nb, nseg, dS.as<float>(),
                                   dI.as<uint32_t>(), nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipMemcpy(hs.data(), dS.p, ntile * per_tile * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hi.data(), dI.p, ntile * per_tile * 4, hipMemcpyDeviceToHost));
        // per query: the lists of lanes j and j + 32 of every (segment, wave); keep the M best scores
        for (uint64_t qi = 0; qi < nb; qi++) {
            const uint32_t tile = (uint32_t)(qi / 32), j = (uint32_t)(qi % 32);
            cand.clear();
            for (uint32_t sw = 0; sw < nseg * 4; sw++)
                for (uint32_t half = 0; half < 2; half++) {
                    const size_t o = (((size_t)tile * nseg * 4 + sw) * 64 + j + 32 * half) * K2;
                    for (uint32_t t = 0; t < K2; t++)
                        if (hi[o + t] != UINT32_MAX) cand.emplace_back(hs[o + t], hi[o + t]);
                }
            const size_t keep = std::min<size_t>(M, cand.size());
            std::partial_sort(cand.begin(), cand.begin() + keep, cand.end());
            for (uint32_t t = 0; t < M; t++) {
                qidx[qi * M + t] = (uint32_t)qi;
                pidx[qi * M + t] = t < keep ? cand[t].second : UINT32_MAX;
            }
        }
        // exact distances of the survivors, in the reference's arithmetic
        HIP_TRY(hipMemcpy(dQi.p, qidx.data(), nb * M * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dPi.p, pidx.data(), nb * M * 4, hipMemcpyHostToDevice));
        rc = hx::launch_pair_distance(v, dQ.as<float>(), dQi.as<uint32_t>(), dPi.as<uint32_t>(), nb * M,
                                      dOut.as<float>(), nullptr);
        if (rc != HNSW_OK) return rc;
        HIP_TRY(hipMemcpy(hd.data(), dOut.p, nb * M * 4, hipMemcpyDeviceToHost));
        for (uint64_t qi = 0; qi < nb; qi++) {
            cand.clear();
            for (uint32_t t = 0; t < M; t++) {
                const uint32_t id = pidx[qi * M + t];
                if (id == UINT32_MAX) continue;
                const float dd = hd[qi * M + t];
                if (dd != dd) {
                    set_error("NaN in a query or a distance");
                    return HNSW_ERR_NAN_INPUT;
                }
                cand.emplace_back(dd, id);
            }
            std::sort(cand.begin(), cand.end());  // (dist, id): Dist::cmp for non-NaN distances
            for (uint32_t t = 0; t < k; t++) {
                const bool have = t < cand.size();
                ids[(q0 + qi) * k + t] = have ? cand[t].second : UINT32_MAX;
                if (dists) dists[(q0 + qi) * k + t] = have ? cand[t].first : INFINITY;
            }
        }
    }
    return HNSW_OK;
}

// ---- accessors -----------------------------------------------------------------------------------
uint64_t hnsw_len(const hnsw_index *h) { return h ? index_len(h) : 0; }

int hnsw_distance(const hnsw_index *h, uint32_t a, uint32_t b, float *out) {
    if (!h || !out) return HNSW_ERR_ARG;
    hx::PointView pa, pb;
    if (!h->host->get_point(a, &pa) || !h->host->get_point(b, &pb)) return HNSW_ERR_ARG;  // None
    *out = h->host->dist2other(pa, pb);
    return HNSW_OK;
}

int hnsw_get_vector(const hnsw_index *h, uint32_t id, float *out) {
    if (!h || !out) return HNSW_ERR_ARG;
    hx::PointView p;
    if (!h->host->get_point(id, &p)) return HNSW_ERR_ARG;
    const uint32_t d = h->host->dim;
    if (h->host->kind == HNSW_VEC_QUANT8)
        for (uint32_t i = 0; i < d; i++) out[i] = ((float)p.codes[i] * p.delta) + p.min;  // quant.rs:79-83
    else
        memcpy(out, p.vals, 4 * (size_t)d);
    return HNSW_OK;
}

int hnsw_get_level(const hnsw_index *h, uint32_t id, uint32_t *out) {
    if (!h || !out || id >= h->host->len()) return HNSW_ERR_ARG;
    *out = h->host->levels[id];
    return HNSW_OK;
}

int hnsw_get_quant(const hnsw_index *h, uint32_t id, uint8_t *codes, float *min_out, float *delta_out) {
    if (!h || h->host->kind != HNSW_VEC_QUANT8 || id >= h->host->len()) return HNSW_ERR_ARG;
    if (codes) memcpy(codes, &h->host->codes[(size_t)id * h->host->dim], h->host->dim);
    if (min_out) *min_out = h->host->mins[id];
    if (delta_out) *delta_out = h->host->deltas[id];
    return HNSW_OK;
}

uint32_t hnsw_layer_count(const hnsw_index *h) {
    if (!h) return 0;
    return is_replica(h) ? h->dev.view.nb_layers : h->host->nb_layers();
}
uint64_t hnsw_layer_nb_nodes(const hnsw_index *h, uint32_t layer) {
    return (h && layer < h->host->nb_layers()) ? h->host->layer_nodes[layer].size() : 0;
}
uint32_t hnsw_layer_m(const hnsw_index *h, uint32_t layer) {
    return (h && layer < h->host->nb_layers()) ? (uint32_t)h->host->layer_m(layer) : 0;
}

int hnsw_layer_nodes(const hnsw_index *h, uint32_t layer, uint32_t *out, uint64_t cap, uint64_t *n) {
    if (!h || layer >= h->host->nb_layers()) return HNSW_ERR_ARG;
    const std::vector<hx::NodeID> &ids = h->host->layer_nodes[layer];
    if (n) *n = ids.size();
    if (out)
        for (uint64_t i = 0; i < ids.size() && i < cap; i++) out[i] = ids[i];
    return HNSW_OK;
}

int hnsw_neighbors(const hnsw_index *h, uint32_t layer, uint32_t id, uint32_t *buf, uint32_t cap,
                   uint32_t *deg) {
    if (!h) return HNSW_ERR_ARG;
    if (!h->host->in_layer(layer, id)) {
        set_error("node %u not in graph (layer %u)", id, layer);
        return HNSW_ERR_NODE_NOT_IN_GRAPH;
    }
    std::vector<hx::NodeID> nb = h->host->row(layer, id);
    std::sort(nb.begin(), nb.end());
    if (deg) *deg = (uint32_t)nb.size();
    if (buf)
        for (uint32_t i = 0; i < nb.size() && i < cap; i++) buf[i] = nb[i];
    return HNSW_OK;
}

int hnsw_export_layer(const hnsw_index *h, uint32_t layer, uint32_t *node_ids, uint64_t *offsets,
                      uint32_t *nbrs, uint64_t *n_nodes, uint64_t *nnz) {
    if (!h || layer >= h->host->nb_layers()) return HNSW_ERR_ARG;
    const std::vector<hx::NodeID> &ids = h->host->layer_nodes[layer];
    uint64_t total = 0;
    for (hx::NodeID id : ids) total += h->host->row(layer, id).size();
    if (n_nodes) *n_nodes = ids.size();
    if (nnz) *nnz = total;
    if (!node_ids || !offsets || !nbrs) return HNSW_OK;
    uint64_t off = 0;
    for (uint64_t i = 0; i < ids.size(); i++) {
        node_ids[i] = ids[i];
        offsets[i] = off;
        const std::vector<hx::NodeID> &r = h->host->row(layer, ids[i]);
        std::copy(r.begin(), r.end(), nbrs + off);
        std::sort(nbrs + off, nbrs + off + r.size());
        off += r.size();
    }
    offsets[ids.size()] = off;
    return HNSW_OK;
}

int hnsw_check_param_compliance(const hnsw_index *h, int *ok) {
    if (!h || !ok) return HNSW_ERR_ARG;
    *ok = h->host->check_param_compliance() ? 1 : 0;
    return HNSW_OK;
}

// ---- persistence -----------------------------------------------------------------------------------
int hnsw_save(const hnsw_index *h, const char *dir) {
    if (!h || !dir) return HNSW_ERR_ARG;
    if (is_replica(h)) return reject_replica(h, "hnsw_save");
    if (h->incomplete_build) {  // an index whose on-device build stopped half way must not be written as if it were whole
        set_error("an on-device build on this handle failed half way; the index is incomplete, not saved");
        return HNSW_ERR_ARG;
    }
    return hx::save_index(*h->host, dir);
}
int hnsw_load(const char *dir, hnsw_index **out) {
    if (!dir || !out) return HNSW_ERR_ARG;
    std::unique_ptr<hx::HostIndex> idx;
    int rc = hx::load_index(dir, &idx);
    if (rc != HNSW_OK) return rc;
    hnsw_index *h = new (std::nothrow) hnsw_index();
    if (!h) return HNSW_ERR_OOM;
    h->host = std::move(idx);
    *out = h;
    return HNSW_OK;
}

// ---- device management -------------------------------------------------------------------------------
int hnsw_device_count(int *count) {
    if (!count) return HNSW_ERR_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return HNSW_OK;
}
int hnsw_set_device(hnsw_index *h, int device) {
    if (!h) return HNSW_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (is_replica(h) && device != h->device) return reject_replica(h, "hnsw_set_device (a replica stays on the device it was received on)");
    if (device != h->device) {
        h->dev.release();
        // a leaderless batch the coalescer keeps open was made ready for the old device (stream, device arena): retire
        // it; the next caller opens one on the new device (no search may be in flight during this call)
        Coalescer &co = h->co;
        std::lock_guard<SpinLock> cg(co.mu);
        for (size_t i = 0; i < co.open.size();) {
            CoBatch *o = co.open[i];
            uint64_t w = o->word.load(std::memory_order_acquire);
            if ((w & CoBatch::COUNT) == 0 && !(w & CoBatch::CLOSED) &&
                o->word.compare_exchange_strong(w, w | CoBatch::CLOSED, std::memory_order_acq_rel)) {
                co.open.erase(co.open.begin() + i);
                co.idle.push_back(o);
            } else {
                i++;
            }
        }
        co.fast.store(nullptr, std::memory_order_release);
    }
    h->device = device;
    return HNSW_OK;
}
int hnsw_upload(hnsw_index *h) {
    if (!h) return HNSW_ERR_ARG;
    return ensure_uploaded(h);
}
int hnsw_set_option(hnsw_index *h, const char *key, int64_t value) {
    if (!h || !key) return HNSW_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (!strcmp(key, "inline_rows")) {
        h->dev.inline_rows = (int)value;
    } else if (!strcmp(key, "inline_budget_mb")) {
        h->dev.fat_budget_bytes = (uint64_t)value << 20;
    } else if (!strcmp(key, "gpu_build")) {
        h->gpu_build = (int)value;
        return HNSW_OK;
    } else if (!strcmp(key, "metric_cosine")) {
        // points already stored stay as they are: set it before the first insert (or on a loaded index whose
        // rows were stored under it -- the reference's file format has no field for a metric)
        h->cosine = value != 0;
        return HNSW_OK;
    } else if (!strcmp(key, "gpu_build_batch_max")) {
        if (value < 1) {
            set_error("gpu_build_batch_max must be positive");
            return HNSW_ERR_ARG;
        }
        h->build_batch_max = (uint32_t)std::min<int64_t>(value, 1 << 20);
        return HNSW_OK;
    } else if (!strcmp(key, "gpu_build_batch_div")) {
        if (value < 1) {
            set_error("gpu_build_batch_div must be positive");
            return HNSW_ERR_ARG;
        }
        h->build_batch_div = (uint32_t)std::min<int64_t>(value, 1 << 20);
        return HNSW_OK;
    } else if (!strcmp(key, "coalesce_us") || !strcmp(key, "coalesce_depth") || !strcmp(key, "coalesce_max")) {
        std::lock_guard<SpinLock> cg(h->co.mu);
        if (!strcmp(key, "coalesce_us")) {
            h->co.window_us.store(std::min<int64_t>(value, 100000));
        } else if (value < 1) {
            set_error("%s must be positive", key);
            return HNSW_ERR_ARG;
        } else if (!strcmp(key, "coalesce_depth")) {
            h->co.depth = (uint32_t)std::min<int64_t>(value, 64);
        } else {
            h->co.cap = (uint32_t)std::min<int64_t>(value, 65536);
        }
        return HNSW_OK;
    } else {
        set_error("unknown option %s", key);
        return HNSW_ERR_ARG;
    }
    if (!is_replica(h)) h->dev.release();  // rebuilt by the next upload
    return HNSW_OK;
}
int hnsw_device_bytes(const hnsw_index *h, uint64_t *bytes) {
    if (!h || !bytes) return HNSW_ERR_ARG;
    *bytes = h->dev.valid ? h->dev.bytes : 0;
    return HNSW_OK;
}
int hnsw_get_stat(const hnsw_index *h, const char *key, uint64_t *out) {
    if (!h || !key || !out) return HNSW_ERR_ARG;
    if (!strcmp(key, "uploads")) {
        *out = h->n_uploads.load();
    } else if (!strcmp(key, "point_patches")) {
        *out = h->n_point_patches.load();
    } else if (!strcmp(key, "patch_fallbacks")) {
        *out = h->n_patch_fallbacks.load();
    } else if (!strncmp(key, "build_", 6)) {
        const hnsw_index::BuildStats &bs = h->build;
        const char *k = key + 6;
        if (!strcmp(k, "points")) *out = bs.points;
        else if (!strcmp(k, "batches")) *out = bs.batches;
        else if (!strcmp(k, "rows_read")) *out = bs.rows_read;
        else if (!strcmp(k, "adj_rows")) *out = bs.adj_rows;
        else if (!strcmp(k, "adj_ids")) *out = bs.adj_ids;
        else if (!strcmp(k, "records")) *out = bs.records;
        else if (!strcmp(k, "removals")) *out = bs.removals;
        else if (!strcmp(k, "insert_kernel_us")) *out = (uint64_t)(bs.insert_kernel_s * 1e6);
        else if (!strcmp(k, "insert_phase_us")) *out = (uint64_t)(bs.insert_phase_s * 1e6);
        else if (!strcmp(k, "connect_us")) *out = (uint64_t)(bs.connect_s * 1e6);
        else if (!strcmp(k, "connect_kernel_us")) *out = (uint64_t)(bs.connect_kernel_s * 1e6);
        else if (!strcmp(k, "rows_owned")) *out = bs.rows_owned;
        else if (!strcmp(k, "rows_received")) *out = bs.rows_received;
        else if (!strcmp(k, "exchange_bytes")) *out = bs.exchange_bytes;
        else if (!strcmp(k, "exchange_us")) *out = (uint64_t)(bs.exchange_s * 1e6);
        else {
            set_error("unknown statistic %s", key);
            return HNSW_ERR_ARG;
        }
    } else if (!strcmp(key, "coalesced_batches")) {
        *out = h->co.n_batches.load();
    } else if (!strcmp(key, "coalesced_queries")) {
        *out = h->co.n_queries.load();
    } else if (!strcmp(key, "coalesced_max_batch")) {
        *out = h->co.max_batch.load();
    } else if (!strcmp(key, "coalesce_ns_window")) {
        *out = h->co.ns_window.load();
    } else if (!strcmp(key, "coalesce_ns_turn")) {
        *out = h->co.ns_turn.load();
    } else if (!strcmp(key, "coalesce_ns_gpu")) {
        *out = h->co.ns_gpu.load();
    } else if (!strcmp(key, "coalesce_ns_handout")) {
        *out = h->co.ns_handout.load();
    } else {
        set_error("unknown statistic %s", key);
        return HNSW_ERR_ARG;
    }
    return HNSW_OK;
}

// ---- snapshot replication ------------------------------------------------------------------------------
namespace {
struct SnapHeader {  // what travels in hnsw_snapshot_desc.header (32 words)
    uint32_t magic, version;
    int32_t kind;
    uint32_t dim, n_points, nb_layers, ep, S0, S1, row_stride, half_bytes, nch4, rem;
    uint32_t fat_stride_lo, fat_stride_hi;
    uint32_t m, ef_cons;
    uint32_t flags;  // bit 0: the cosine option (queries are normalised on arrival)
    uint32_t reserved[14];
};
static_assert(sizeof(SnapHeader) == 32 * 4, "snapshot header is 32 words");
constexpr uint32_t SNAP_MAGIC = 0x48584E53u;  // "SNXH"
}  // namespace

int hnsw_snapshot_describe(hnsw_index *h, hnsw_snapshot_desc *out) {
    if (!h || !out) return HNSW_ERR_ARG;
    int rc = check_search_args(h, 1);
    if (rc != HNSW_OK) return rc;
    if ((rc = ensure_uploaded(h))) return rc;
    const hx::DevView &v = h->dev.view;
    memset(out, 0, sizeof(*out));
    uint64_t nb[7];
    void *pp[7];
    h->dev.describe(nb, pp);
    for (int i = 0; i < HNSW_SNAPSHOT_ARRAYS; i++) {
        out->bytes[i] = nb[i];
        out->ptr[i] = pp[i];
    }
    SnapHeader hd{};
    hd.magic = SNAP_MAGIC;
    hd.version = 1;
    hd.kind = v.kind;
    hd.dim = v.dim;
    hd.n_points = v.n_points;
    hd.nb_layers = v.nb_layers;
    hd.ep = v.ep;
    hd.S0 = v.S0;
    hd.S1 = v.S1;
    hd.row_stride = v.row_stride;
    hd.half_bytes = v.half_bytes;
    hd.nch4 = v.nch4;
    hd.rem = v.rem;
    hd.fat_stride_lo = (uint32_t)v.fat_stride;
    hd.fat_stride_hi = (uint32_t)(v.fat_stride >> 32);
    hd.m = (uint32_t)h->host->params.m;
    hd.ef_cons = (uint32_t)h->host->params.ef_cons;
    hd.flags = h->cosine ? 1u : 0u;
    memcpy(out->header, &hd, sizeof(hd));
    return HNSW_OK;
}

int hnsw_snapshot_adopt(hnsw_index *h, hnsw_snapshot_desc *d) {
    if (!h || !d) return HNSW_ERR_ARG;
    SnapHeader hd;
    memcpy(&hd, d->header, sizeof(hd));
    if (hd.magic != SNAP_MAGIC || hd.version != 1) {
        set_error("hnsw_snapshot_adopt: not a snapshot header");
        return HNSW_ERR_ARG;
    }
    if (h->host->len() != 0 || is_replica(h)) {
        set_error("hnsw_snapshot_adopt: the receiving handle must be empty (fresh from hnsw_create)");
        return HNSW_ERR_ARG;
    }
    if (hd.kind != h->host->kind || hd.dim != h->host->dim || hd.m != (uint32_t)h->host->params.m) {
        set_error("hnsw_snapshot_adopt: snapshot of a %ud kind-%d m=%u index offered to a %ud kind-%d m=%u handle", hd.dim,
                  hd.kind, hd.m, h->host->dim, h->host->kind, (uint32_t)h->host->params.m);
        return HNSW_ERR_BAD_DIM;
    }
    // Every stride in the header is recomputed from the receiving handle's own m, dim and kind and must agree (the
    // kernels index the arrays by id without a range check, and a zero stride must never reach a division), then
    // the array sizes must be what the header implies.
    const bool q8 = h->host->kind == HNSW_VEC_QUANT8;
    const uint32_t half = q8 ? hx::quant_half_bytes(hd.dim) : 0, stride = q8 ? 2 * half : hx::f32_row_stride(hd.dim);
    const uint32_t S0 = hx::adj_stride(h->host->layer_m(0), 32), S1 = hx::adj_stride(h->host->params.m, 8);
    const uint64_t fat_stride = ((uint64_t)hd.fat_stride_hi << 32) | hd.fat_stride_lo;
    if (hd.S0 != S0 || hd.S1 != S1 || hd.row_stride != stride || hd.half_bytes != half || hd.nch4 != 4 * (hd.dim / 8) ||
        hd.rem != hd.dim % 8 || (fat_stride != 0 && fat_stride != (uint64_t)S0 * stride)) {
        set_error("hnsw_snapshot_adopt: the header's strides are not those of a %ud kind-%d m=%u index", hd.dim, hd.kind, hd.m);
        return HNSW_ERR_ARG;
    }
    const uint64_t N = hd.n_points;
    if (N == 0 || N > 0x7FFFFFFFull || d->bytes[0] != N * stride || d->bytes[1] != N * S0 * 4ull || d->bytes[3] != N * 4ull ||
        d->bytes[2] == 0 || d->bytes[2] % (S1 * 4ull) != 0 || d->bytes[4] < 4 || d->bytes[4] % 4 != 0 || d->bytes[5] < 4 ||
        (d->bytes[6] != 0) != (fat_stride != 0) || (fat_stride != 0 && d->bytes[6] != N * fat_stride) || hd.ep >= N ||
        hd.nb_layers == 0) {
        set_error("hnsw_snapshot_adopt: array sizes do not match the header");
        return HNSW_ERR_ARG;
    }
    std::lock_guard<std::mutex> g(h->mu);
    uint64_t nb[7];
    void *pp[7];
    for (int i = 0; i < 7; i++) nb[i] = d->bytes[i];
    int rc = h->dev.adopt_alloc(h->device, nb, pp);
    if (rc != HNSW_OK) {
        h->dev.release();
        return rc;
    }
    h->device = h->dev.device;
    for (int i = 0; i < 7; i++) d->ptr[i] = pp[i];
    hx::DevView v{};
    v.kind = hd.kind;
    v.dim = hd.dim;
    v.n_points = hd.n_points;
    v.nb_layers = hd.nb_layers;
    v.ep = hd.ep;
    v.S0 = hd.S0;
    v.S1 = hd.S1;
    v.row_stride = hd.row_stride;
    v.half_bytes = hd.half_bytes;
    v.nch4 = hd.nch4;
    v.rem = hd.rem;
    v.fat_stride = ((uint64_t)hd.fat_stride_hi << 32) | hd.fat_stride_lo;
    h->dev.view = v;     // scalars now, pointers at commit
    h->dev.replica = true;
    h->cosine = (hd.flags & 1u) != 0;
    return HNSW_OK;
}

int hnsw_snapshot_commit(hnsw_index *h) {
    if (!h) return HNSW_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->dev.replica || h->dev.valid) {
        set_error("hnsw_snapshot_commit: no adopted snapshot waiting");
        return HNSW_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(h->dev.device));
    HIP_TRY(hipDeviceSynchronize());  // whatever filled the arrays has finished
    h->dev.adopt_commit(h->dev.view);
    h->host->params.ep = h->dev.view.ep;
    return HNSW_OK;
}

// ---- harness helpers -----------------------------------------------------------------------------------
int hnsw_synth_rows(int recipe, uint64_t seed, uint64_t first_row, uint64_t n, uint32_t d, float *out,
                    uint32_t nb_threads) {
    if (!out) return HNSW_ERR_ARG;
    return hx::synth_rows(recipe, seed, first_row, n, d, out, nb_threads);
}
// T host threads, each blocked in its own hnsw_search call like the reference's callers (ann_by_vector(&self), one
// query per call): thread t answers queries t, t + T, t + 2T, ... of Q, again and again until `seconds` have passed
// and every query has been answered at least once.  ids receives each query's LAST answer (so the caller can hold
// the run to the oracle), lat_us = {p50, p90, p99, max, mean} of the per-call latencies.
int hnsw_bench_search_threads(hnsw_index *h, const float *Q, uint64_t nq, uint32_t n, uint32_t ef, uint32_t threads,
                              double seconds, uint32_t *ids, uint32_t *counts, uint64_t *calls_out, double *wall_s,
                              double *lat_us) {
    if (!h || !Q || !ids || nq == 0 || n == 0 || threads == 0 || threads > 4096) return HNSW_ERR_ARG;
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if ((rc = ensure_uploaded(h))) return rc;
    const uint32_t d = h->dev.view.dim;
    const uint32_t T = (uint32_t)std::min<uint64_t>(threads, nq);
    std::vector<std::vector<float>> lat(T);
    std::atomic<int> first_rc{HNSW_OK};
    std::string first_msg;
    std::mutex msg_mu;
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    using clk = std::chrono::steady_clock;
    clk::time_point t_start;
    auto work = [&](uint32_t t) {
        std::vector<float> &L = lat[t];
        L.reserve(1 << 16);
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        const auto deadline = t_start + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(seconds));
        bool full_pass = false;
        while (first_rc.load(std::memory_order_relaxed) == HNSW_OK) {
            for (uint64_t i = t; i < nq; i += T) {
                uint32_t cnt = 0;
                const auto a = clk::now();
                const int r = hnsw_search(h, Q + i * d, n, ef, ids + i * n, &cnt);
                const auto b = clk::now();
                if (counts) counts[i] = cnt;
                if (r != HNSW_OK) {
                    std::lock_guard<std::mutex> g(msg_mu);
                    if (first_rc.load() == HNSW_OK) {
                        first_msg = hx::get_error();
                        first_rc.store(r);
                    }
                    return;
                }
                L.push_back(std::chrono::duration<float, std::micro>(b - a).count());
                if (full_pass && b >= deadline) return;
            }
            full_pass = true;
            if (clk::now() >= deadline) return;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; t++) th.emplace_back(work, t);
    while (ready.load() < T) std::this_thread::yield();
    struct rusage ru0;
    getrusage(RUSAGE_SELF, &ru0);
    t_start = clk::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    const double wall = std::chrono::duration<double>(clk::now() - t_start).count();
    if (first_rc.load() != HNSW_OK) {
        set_error("%s", first_msg.c_str());
        return first_rc.load();
    }
    std::vector<float> all;
    for (auto &L : lat) all.insert(all.end(), L.begin(), L.end());
    std::sort(all.begin(), all.end());
    if (calls_out) *calls_out = all.size();
    if (wall_s) *wall_s = wall;
    if (lat_us) {
        struct rusage ru1;
        getrusage(RUSAGE_SELF, &ru1);
        lat_us[5] = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + 1e-6 * (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec);
        lat_us[6] = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
    }
    if (lat_us && !all.empty()) {
        auto pct = [&](double p) { return (double)all[std::min(all.size() - 1, (size_t)(p * all.size()))]; };
        double sum = 0;
        for (float x : all) sum += x;
        lat_us[0] = pct(0.50);
        lat_us[1] = pct(0.90);
        lat_us[2] = pct(0.99);
        lat_us[3] = all.back();
        lat_us[4] = sum / all.size();
    }
    return HNSW_OK;
}
// `callers` host threads, each calling hnsw_search_batch (host pointers in and out) `calls` times on its own slice
// of Q (caller t takes queries [t * nq, (t + 1) * nq) modulo total) into its own result buffers: what concurrent
// batch callers of the C ABI see, without an interpreter in the loop.  *wall_s = the time from the first call to the
// last return (every caller's stream and staging exist before the clock starts: two untimed calls each).
int hnsw_bench_batch_threads(hnsw_index *h, const float *Q, uint64_t total, uint64_t nq, uint32_t n, uint32_t ef,
                             uint32_t callers, uint32_t calls, double *wall_s) {
    if (!h || !Q || !wall_s || nq == 0 || total < nq || n == 0 || callers == 0 || callers > 64 || calls == 0) return HNSW_ERR_ARG;
    int rc = check_search_args(h, ef);
    if (rc != HNSW_OK) return rc;
    if ((rc = ensure_uploaded(h))) return rc;
    const uint32_t d = h->dev.view.dim;
    std::atomic<int> first_rc{HNSW_OK};
    std::string first_msg;
    std::mutex msg_mu;
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    const uint64_t slices = total / nq;
    auto work = [&](uint32_t t) {
        std::vector<uint32_t> ids(nq * n), counts(nq);
        std::vector<float> dists(nq * n);
        std::vector<hnsw_query_stats> st(nq);
        auto one = [&](uint32_t i) {
            const float *q = Q + ((t + (uint64_t)i * callers) % slices) * nq * d;
            return hnsw_search_batch(h, q, nq, n, ef, ids.data(), dists.data(), counts.data(), st.data());
        };
        int r = one(0);
        if (r == HNSW_OK) r = one(1);
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (uint32_t i = 0; i < calls && r == HNSW_OK && first_rc.load(std::memory_order_relaxed) == HNSW_OK; i++) r = one(i + 2);
        if (r != HNSW_OK) {
            std::lock_guard<std::mutex> g(msg_mu);
            if (first_rc.load() == HNSW_OK) {
                first_msg = hx::get_error();
                first_rc.store(r);
            }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < callers; t++) th.emplace_back(work, t);
    while (ready.load() < callers) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    *wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (first_rc.load() != HNSW_OK) {
        set_error("%s", first_msg.c_str());
        return first_rc.load();
    }
    return HNSW_OK;
}
int hnsw_draw_levels(uint32_t m, uint64_t n, uint8_t *out) {
    if (!out || m < 2) return HNSW_ERR_ARG;
    hx::stdrng_levels(0, hx::default_ml(m), n, out);
    return HNSW_OK;
}

}  // extern "C"
