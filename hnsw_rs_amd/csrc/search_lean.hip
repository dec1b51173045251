// search_lean.hip -- the latency-critical form of the search hot path for gfx950 (MI355X, CDNA4):
// HNSW::ann_by_vector (hnsw/src/template.rs:306-335) over FullVec rows (vectors/src/full.rs:23-29)
// for the launch sizes where a query's own dependent chain, not the memory system, sets the time
// (a 1024-query batch is one wave per SIMD; measured in round 2 with in-kernel cycle stamps: the
// generic kernel of search_kernels.hip spent 2.9 k of its 10 k cycles per pass in the distance chain
// because the compiler fetched every query value through its own LDS round trip with the address
// spilled to a VGPR lane, 1.6 k in the visited filter's nested divergent loops, and 22 % of the whole
// query in the handful of upper-layer expansions, each three dependent round trips).
//
// Same algorithm, same results, same counters as hx_search_kernel (the parity tests run both):
//   - one 64-lane wave per query; `selected` + `candidates` (results.rs:26-33) are ONE sorted list of
//     <= ef keys (dist_bits << 32 | id, bit 63 = expanded), searcher.rs:35-94 is "expand the smallest
//     unexpanded entry until none is left";
//   - layer 0 evaluates TWO adjacency rows per pass: lanes 0..31 the candidate c, lanes 32..63 the
//     runner-up p, which is committed from registers when it is still next after c's merge.
// What is different:
//   - the query lives in VGPRs (100 registers at d = 100): the chain is v_pk_add / v_pk_mul / v_add
//     on registers, nothing else;
//   - the visited filter is one straight-line round (bucket read, compare, one ds_cmpst) for the whole
//     wave -- c's lanes insert, p's lanes only look -- with a loop only for the lanes that lost a slot
//     or met a full bucket;
//   - the list is interleaved (entry i = lane i / R, register i % R), so an insertion is a one-lane DPP
//     shift without a carry between registers; one or two survivors are shift-inserted, larger batches
//     go through LDS without a vector -> scalar hand-off per survivor (list entries count their shift,
//     survivors their rank, the holes the scattered list leaves take the survivors in order);
//   - the row request, the visited claims and the distance chain of a pass sit under ONE exec mask
//     (split over several, the allocator parked the row registers in AGPRs: one wave per SIMD and a
//     v_accvgpr_read per element), and the pass's uniform branches carry likely / unlikely hints;
//   - the upper layers (ef = 1, searcher.rs:23-103 degenerates to a greedy walk) keep just the best
//     key, and the next node's adjacency base (upper_base) is fetched together with the vector rows of
//     the neighbours, so an expansion is two dependent round trips instead of three.
//
// Float fidelity: -ffp-contract=off; the sum is FullVec's single left-to-right chain; sqrt is the
// correctly rounded one.

#include <algorithm>
#include <cstdlib>

#include "device_index.h"
#include "search_common.h"

namespace hx {
#include "coop_rows.inc"
namespace {

// block placement hints for the f32 kernel's pass (a taken branch costs a lone wave ~30 cycles of
// instruction refetch: SQ_WAIT_INST_ANY was 12 % of the wave's time); measured + 2 % there, nothing for
// the quant8 kernel, whose loop is left as the compiler lays it out
#define HX_LIKELY(x) __builtin_expect(!!(x), 1)
#define HX_UNLIKELY(x) __builtin_expect(!!(x), 0)

constexpr u64 LK_INVALID = ~0ull;
constexpr u64 LK_MASK = 0x7FFFFFFFFFFFFFFFull;
constexpr u64 LK_EXPANDED = 1ull << 63;

struct LeanArgs {
    const float *rows;           // F32: N x DS floats; QUANT8: the packed half rows (device_index.h)
    const uint32_t *adj0;        // N x S0
    const uint32_t *adj_up;      // rows of S1
    const uint32_t *upper_base;  // N
    const uint32_t *ovf_off, *ovf_nbrs;
    const float *Q;
    const uint32_t *qsel;
    uint32_t *out_ids;
    float *out_dists;
    uint32_t *out_counts;
    hnsw_query_stats *out_stats;
    uint32_t n_points, ep, nb_layers, S0, S1, ef, n, slots_log2;
    unsigned long long *dbg;  // diagnostic builds only (HX_STAMPS)
    // second level of the visited set (eight-register lists, ef > 320): 1 << spill_log2 words of HBM per launched
    // query, or null
    uint32_t *spill_tab;
    uint32_t spill_log2;
    uint32_t lds_limit;  // ids the LDS level takes before it is closed; 0 = 75 % of its slots (tests lower it)
};

// Diagnostic build only (make stamps): per-phase cycle sums, written to a side buffer nothing else reads.
#ifdef HX_STAMPS
// a stamp drains the wave's memory counters and is a scheduling barrier, so that what lies between two
// stamps is exactly the code written between them
__device__ __forceinline__ unsigned long long stamp_now() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp_now()
#define STAMP_ADD(slot, a, b) dbg_acc[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ u64 rdlane64(u64 v, uint32_t l) {
    return ((u64)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or_max(uint32_t x) {  // lanes without a source keep 0xFFFFFFFF
    return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, 0xF, 0xF, false);
}
// minimum over the wave (wave-uniform result)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_or_max<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min(v, dpp_or_max<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min(v, dpp_or_max<0x141>(v));  // row_half_mirror
    v = min(v, dpp_or_max<0x140>(v));  // row_mirror: every lane holds its row's minimum
    return min(min(rdlane(v, 0), rdlane(v, 16)), min(rdlane(v, 32), rdlane(v, 48)));
}

// ---------------------------------------------------------------------------------------------
// visited set (IntSet::insert, results.rs:101-103): open addressing over buckets of four slots.
// Slots of a bucket fill left to right and never empty again; ids are < 2^31 and an empty slot is
// 0xFFFFFFFF, so the number of slots with a clear sign bit is the index of the first empty one.
// Everything a lane decides stays in vector registers (a wave alone on its SIMD pays ~20 cycles for
// every vector -> scalar hand-off and ~25 for every branch: scripts/micro/issue_cost.hip).
// ---------------------------------------------------------------------------------------------
struct Visited {
    uint32_t *tab;
    uint32_t bshift, bmask;  // bucket = (id * K) >> bshift; bmask = buckets - 1

    __device__ __forceinline__ uint32_t home(uint32_t id) const { return (id * 0x9E3779B1u) >> bshift; }

    // state of bucket b with respect to id: 0..3 = id absent, this is the first empty slot;
    // 4 = id present; 5 = id absent and the bucket is full
    __device__ __forceinline__ uint32_t look(uint32_t id, uint32_t b) const {
        const uint4 bk = *reinterpret_cast<const uint4 *>(tab + 4 * b);
        const uint32_t m = min(min(bk.x ^ id, bk.y ^ id), min(bk.z ^ id, bk.w ^ id));
        const uint32_t nf = 4u + (uint32_t)(((int32_t)bk.x >> 31) + ((int32_t)bk.y >> 31) + ((int32_t)bk.z >> 31) +
                                            ((int32_t)bk.w >> 31));
        return m == 0 ? 4u : (nf == 4u ? 5u : nf);
    }
    // claims slot t of bucket b for id where t < 4; returns what the slot held (0 for t >= 4)
    __device__ __forceinline__ uint32_t claim(uint32_t id, uint32_t b, uint32_t t) const {
        uint32_t old = 0;
        if (t < 4u) old = atomicCAS(tab + 4 * b + t, HX_EMPTY_SLOT, id);
        return old;
    }
    // the lanes that could not finish in their first round: a slot lost to another lane (look again at
    // the same bucket) or a full bucket (the next one).  Rare; the only loop of the filter.
    __device__ __forceinline__ bool finish(uint32_t id, uint32_t b, uint32_t t, bool pend, bool fresh) const {
        while (__ballot(pend)) {
            if (pend) {
                if (t == 5u) b = (b + 1) & bmask;
                t = look(id, b);
                const uint32_t old = claim(id, b, t);
                const bool won = old == HX_EMPTY_SLOT;
                fresh |= won;
                pend = (t != 4u) & !won;
            }
        }
        return fresh;
    }
    // exact insert of the lanes with ins == true (the ids of one call are distinct); true = was absent
    __device__ __forceinline__ bool insert(uint32_t id, bool ins) const {
        const uint32_t b = home(id);
        uint32_t t = look(id, b);
        t = ins ? t : 4u;
        const uint32_t old = claim(id, b, t);
        const bool fresh = old == HX_EMPTY_SLOT;
        return finish(id, b, t, (t != 4u) & !fresh, fresh);
    }
    __device__ __forceinline__ void clear(uint32_t nslots, int lane) const {
        for (uint32_t s = lane; s < (nslots >> 2); s += 64)
            reinterpret_cast<uint4 *>(tab)[s] = make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
        lds_fence();
    }

    // ---- second level (round 4; the eight-register kernels, 320 < ef <= 512) ----------------------------------
    // A 64-KiB table leaves two waves per CU, i.e. a batch of 1024 runs in two rounds (4.7 us per expansion
    // against 2.8 at ef 320).  The LDS table stays at 32 KiB instead; once it holds its limit it is CLOSED (looked
    // at, never written again) and the ids that arrive afterwards go to a table in HBM that only the queries
    // which get that far ever touch: one slot per id, linear probing, agent-scope loads and a compare-and-swap in
    // L2 (the scheme of hx_search_spill_kernel, search_kernels.hip).  An id is in the set iff it is in either
    // level, and an id is only ever inserted after both levels were found not to hold it: the set -- and with it
    // the fresh sets, the counters and the result -- is the one-level table's (IntSet, results.rs:101-103).
    uint32_t *gtab = nullptr;
    uint32_t gshift = 0, gmask = 0;
    bool spill = false;  // wave-uniform: the LDS level is closed

    __device__ __forceinline__ bool g_contains(uint32_t id) const {
        uint32_t s = ((id * 0x9E3779B1u) >> gshift) & gmask;
        while (true) {
            const uint32_t cur = __hip_atomic_load(gtab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == id) return true;
            if (cur == HX_EMPTY_SLOT) return false;
            s = (s + 1) & gmask;
        }
    }
    // exact insert into the HBM level: HX_EMPTY_SLOT when id was absent (now inserted), id when it was there.
    // The compare-and-swap IS the probe (it returns what the slot holds): one round trip per slot looked at.
    __device__ __forceinline__ uint32_t g_claim(uint32_t id) const {
        uint32_t s = ((id * 0x9E3779B1u) >> gshift) & gmask;
        while (true) {
            const uint32_t old = atomicCAS(gtab + s, HX_EMPTY_SLOT, id);
            if (old == HX_EMPTY_SLOT) return HX_EMPTY_SLOT;
            if (old == id) return id;
            s = (s + 1) & gmask;  // taken by another id (possibly of this very pass): next slot
        }
    }
    // close the LDS level: from here on new ids go to the HBM level, emptied now
    __device__ __forceinline__ void open_second_level(int lane) {
        const uint32_t nslots = gmask + 1;
        for (uint32_t s = lane; s < (nslots >> 2); s += 64)
            reinterpret_cast<uint4 *>(gtab)[s] = make_uint4(HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT, HX_EMPTY_SLOT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached L2 before the first probe
        spill = true;
    }
    // look over both levels.  While the LDS level is open: 0..3 / 4 / 5 as look().  Closed: 4 = present in either
    // level; 6 = absent from both (to be claimed in the HBM level later: the runner-up's lanes, which only look);
    // 7 = was absent from both and has just been claimed in the HBM level -- the lanes with claim_now (the candidate's:
    // they insert whatever is absent, and nothing else touches the set between their look and their claim, so
    // claiming at the look is the same insert one HBM round trip earlier; a runner-up lane that does or does not see
    // it yet is right either way, its own insert at its commit is exact).  probe == false: 4 (a lane without an id).
    __device__ __forceinline__ uint32_t look2(uint32_t id, uint32_t b, bool probe, bool claim_now) const {
        uint32_t t = look(id, b);
        if (!spill) return t;
        if (!probe) return 4u;
        while (t == 5u) {  // a full bucket: the id may sit in a later one
            b = (b + 1) & bmask;
            t = look(id, b);
        }
        if (t == 4u) return 4u;
        if (claim_now) return g_claim(id) == HX_EMPTY_SLOT ? 7u : 4u;
        return g_contains(id) ? 4u : 6u;
    }
    __device__ __forceinline__ uint32_t claim2(uint32_t id, uint32_t b, uint32_t t) const {
        uint32_t old = 0;
        if (t < 4u)
            old = atomicCAS(tab + 4 * b + t, HX_EMPTY_SLOT, id);
        else if (t == 6u)
            old = g_claim(id);
        else if (t == 7u)
            old = HX_EMPTY_SLOT;  // claimed at the look
        return old;
    }
    __device__ __forceinline__ bool finish2(uint32_t id, uint32_t b, uint32_t t, bool pend, bool fresh) const {
        while (__ballot(pend)) {
            if (pend) {
                if (t == 5u) b = (b + 1) & bmask;
                t = look2(id, b, true, false);
                const uint32_t old = claim2(id, b, t);
                const bool won = old == HX_EMPTY_SLOT;
                fresh |= won;
                pend = (t != 4u) & !won;  // (a lost HBM claim means the id is there: the next look says 4)
            }
        }
        return fresh;
    }
};

// ---------------------------------------------------------------------------------------------
// the sorted list, interleaved over the wave: entry i is register i % R of lane i / R
// ---------------------------------------------------------------------------------------------
#ifndef HX_MERGE_SHIFT_MAX
#define HX_MERGE_SHIFT_MAX 2  // survivors merged one at a time by shifting the list; more go through LDS
#endif
template <int R>
struct Lst {
    static constexpr int NR = R;  // list registers per lane
    struct Masks {
        u64 U[R];
    };
    u64 L[R];
    uint32_t n_cur;  // wave-uniform
    u64 last_key;    // key (flag dropped) of entry ef - 1 when the list is full, else LK_INVALID

    static __device__ __forceinline__ uint32_t idx_of(int r, int lane) { return (uint32_t)(R * lane + r); }
    __device__ __forceinline__ void set_first(u64 key, int lane) {
        if (lane == 0) L[0] = key;
        n_cur = 1;
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = LK_INVALID;
        n_cur = 0;
        last_key = LK_INVALID;
    }
    // index of the smallest entry not expanded yet, -1 if none (invalid entries have bit 63 set)
    __device__ __forceinline__ int first_unexp() const {
        uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u64 U = __ballot((int32_t)(uint32_t)(L[r] >> 32) >= 0);
            const uint32_t i = U ? (uint32_t)(R * (__ffsll((long long)U) - 1) + r) : 0xFFFFFFFFu;
            pos = min(pos, i);
        }
        return (int)pos;
    }
    // ballots of the entries not expanded yet, and "take the smallest of them" on those masks: several
    // picks cost one vector -> scalar hand-off
    __device__ __forceinline__ void unexp_masks(Masks &mk) const {
#pragma unroll
        for (int r = 0; r < R; r++) mk.U[r] = __ballot((int32_t)(uint32_t)(L[r] >> 32) >= 0);
    }
    __device__ __forceinline__ int take_first(Masks &mk) const {
        u64(&U)[R] = mk.U;
        uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = U[r] ? (uint32_t)(R * (__ffsll((long long)U[r]) - 1) + r) : 0xFFFFFFFFu;
            pos = min(pos, i);
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (pos != 0xFFFFFFFFu && (pos % R) == (uint32_t)r) U[r] &= U[r] - 1;  // its lowest set bit
        return (int)pos;
    }
    __device__ __forceinline__ uint32_t id_at(uint32_t pos) const {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t t = rdlane((uint32_t)L[r], pos / R);
            if ((pos % R) == (uint32_t)r) v = t;
        }
        return v;
    }
    __device__ __forceinline__ u64 key_at(uint32_t pos) const {
        u64 v = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u64 t = rdlane64(L[r], pos / R);
            if ((pos % R) == (uint32_t)r) v = t;
        }
        return v;
    }
    __device__ __forceinline__ void mark(uint32_t pos, int lane) {
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((pos % R) == (uint32_t)r && (uint32_t)lane == pos / R) L[r] |= LK_EXPANDED;
    }
    __device__ __forceinline__ void refresh_last(uint32_t ef) {
        last_key = n_cur >= ef ? (key_at(ef - 1) & LK_MASK) : LK_INVALID;
    }

    // Merge the wave's keys (LK_INVALID = none) into the list, keeping the ef smallest: the streaming
    // top-ef of searcher.rs:74-94 applied to a whole batch (order-independent).
    __device__ __forceinline__ void merge(u64 key, uint32_t ef, u64 *perm, int lane) {
        const bool surv = key < last_key;  // last_key is LK_INVALID while the list is not full
        const u64 smask = __ballot(surv);
        if (smask == 0) return;
        const uint32_t m = (uint32_t)__popcll(smask);
        if (m <= HX_MERGE_SHIFT_MAX) {
            u64 it = smask;
            while (it) {
                const uint32_t j = (uint32_t)__ffsll((long long)it) - 1;
                it &= it - 1;
                const u64 e = rdlane64(key, j);
                if (!(e < last_key)) continue;  // the first insert tightened the bound
                uint32_t pos = 0;
#pragma unroll
                for (int r = 0; r < R; r++) pos += (uint32_t)__popcll(__ballot((L[r] & LK_MASK) < e));
                // entries from pos on move up by one: entry i takes entry i - 1, i.e. register r takes
                // register r - 1 of the same lane, register 0 takes register R - 1 of the lane below
                const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)L[R - 1], 0x138, 0xF, 0xF, false);
                const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(L[R - 1] >> 32), 0x138, 0xF, 0xF, false);
                u64 below = ((u64)phi << 32) | plo;
#pragma unroll
                for (int r = R - 1; r >= 0; r--) {
                    const uint32_t idx = (uint32_t)(R * lane + r);
                    const u64 from = r > 0 ? L[r > 0 ? r - 1 : 0] : below;
                    L[r] = idx < pos ? L[r] : (idx == pos ? e : from);
                }
                n_cur = min(n_cur + 1, ef);
                if (ef < 64u * R) {
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if ((uint32_t)(R * lane + r) >= ef) L[r] = LK_INVALID;
                }
                refresh_last(ef);
            }
            return;
        }
        // Three or more.  One loop over the survivors, no vector -> scalar hand-off in it: every list
        // entry counts the survivors that sort before it (its shift), every survivor its rank among the
        // survivors.  The list entries then go to their final places through LDS; the places nobody wrote
        // are the survivors', in order: the s-th hole takes the survivor of rank s (sv[], filled on the
        // way).  perm holds 64 R entries, sv the 64 behind them.
        u64 *sv = perm + 64 * R;
        u64 Lm[R];
        uint32_t shift[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            Lm[r] = L[r] & LK_MASK;
            shift[r] = 0;
        }
        uint32_t srank = 0;
        u64 it = smask;
        while (it) {
            const uint32_t j = (uint32_t)__ffsll((long long)it) - 1;
            it &= it - 1;
            const u64 e = rdlane64(key, j);
#pragma unroll
            for (int r = 0; r < R; r++) shift[r] += e < Lm[r] ? 1u : 0u;
            srank += e < key ? 1u : 0u;
        }
        const uint32_t n_new = min(n_cur + m, ef);
#pragma unroll
        for (int r = 0; r < R; r++) perm[R * lane + r] = LK_INVALID;
        asm volatile("" ::: "memory");  // LDS serves a wave's requests in order
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = (uint32_t)(R * lane + r);
            const uint32_t np = idx + shift[r];
            if (idx < n_cur && np < n_new) perm[np] = L[r];
        }
        if (surv) sv[srank] = key;
        asm volatile("" ::: "memory");
        u64 v[R], H[R];
        bool hole[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = perm[R * lane + r];
#pragma unroll
        for (int r = 0; r < R; r++) {
            hole[r] = v[r] == LK_INVALID && (uint32_t)(R * lane + r) < n_new;
            H[r] = __ballot(hole[r]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t hr = 0;  // holes at places before R * lane + r
#pragma unroll
            for (int r2 = 0; r2 < R; r2++) {
                hr += __builtin_amdgcn_mbcnt_hi((uint32_t)(H[r2] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H[r2], 0u));
                if (r2 < r && hole[r2]) hr++;
            }
            if (hole[r]) v[r] = sv[hr];
        }
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = v[r];
        n_cur = n_new;
        lds_fence();
        refresh_last(ef);
    }
};

// ---------------------------------------------------------------------------------------------
// The list for 64 < ef <= 128 as HEAD + TAIL: the 64 smallest entries in one register (entry i = lane
// i), entries 64 .. ef - 1 in a second one (entry 64 + i = lane i).  With the interleaved two-register
// list every pick, every mark and every merge touched both registers, and the metric's efSearch 68 --
// four entries past one register -- paid 8.7 % for 1.3 % more work (profiles/r02_ef_sweep.txt).  Here the
// candidates are picked and marked in the head alone (the tail is looked at only once the head holds
// nothing unexpanded, i.e. in a query's last picks), a merge that inserts nothing touches neither, an
// insertion into the head hands the entry that falls off its end to the tail (one whole-register shift),
// an insertion beyond entry 63 touches the tail alone.  Same contents, same order, same results as Lst<2>.
// ---------------------------------------------------------------------------------------------
struct LstHT {
    static constexpr int NR = 2;
    struct Masks {
        u64 head, tail;
        bool tail_known;
    };
    u64 L[2];  // L[0] = head, L[1] = tail
    uint32_t n_cur;
    u64 last_key;

    static __device__ __forceinline__ uint32_t idx_of(int r, int lane) { return (uint32_t)(64 * r + lane); }
    __device__ __forceinline__ void init() {
        L[0] = L[1] = LK_INVALID;
        n_cur = 0;
        last_key = LK_INVALID;
    }
    __device__ __forceinline__ void set_first(u64 key, int lane) {
        if (lane == 0) L[0] = key;
        n_cur = 1;
    }
    __device__ __forceinline__ void unexp_masks(Masks &mk) const {
        mk.head = __ballot((int32_t)(uint32_t)(L[0] >> 32) >= 0);
        mk.tail = 0;
        mk.tail_known = false;
    }
    // smallest unexpanded entry: in the head while it has one
    __device__ __forceinline__ int take_first(Masks &mk) const {
        if (HX_LIKELY(mk.head != 0)) {
            const int pos = __ffsll((long long)mk.head) - 1;
            mk.head &= mk.head - 1;
            return pos;
        }
        if (!mk.tail_known) {
            mk.tail = n_cur > 64 ? __ballot((int32_t)(uint32_t)(L[1] >> 32) >= 0) : 0ull;
            mk.tail_known = true;
        }
        if (mk.tail == 0) return -1;
        const int pos = __ffsll((long long)mk.tail) - 1;
        mk.tail &= mk.tail - 1;
        return 64 + pos;
    }
    __device__ __forceinline__ uint32_t id_at(uint32_t pos) const {
        const uint32_t a = rdlane((uint32_t)L[0], pos & 63u), b = rdlane((uint32_t)L[1], pos & 63u);
        return pos < 64u ? a : b;
    }
    __device__ __forceinline__ u64 key_at(uint32_t pos) const {
        const u64 a = rdlane64(L[0], pos & 63u), b = rdlane64(L[1], pos & 63u);
        return pos < 64u ? a : b;
    }
    __device__ __forceinline__ void mark(uint32_t pos, int lane) {
        if ((uint32_t)lane == pos) L[0] |= LK_EXPANDED;
        if ((uint32_t)lane + 64u == pos) L[1] |= LK_EXPANDED;
    }
    __device__ __forceinline__ void refresh_last(uint32_t ef) {
        // ef > 64: entry ef - 1 lives in the tail
        last_key = n_cur >= ef ? (rdlane64(L[1], (ef - 65u) & 63u) & LK_MASK) : LK_INVALID;
    }
    static __device__ __forceinline__ u64 shr1(u64 v) {  // lane i takes lane i - 1's value (lane 0: 0)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
        return ((u64)hi << 32) | lo;
    }

    __device__ __forceinline__ void merge(u64 key, uint32_t ef, u64 *perm, int lane) {
        const bool surv = key < last_key;  // last_key is LK_INVALID while the list is not full
        const u64 smask = __ballot(surv);
        if (smask == 0) return;
        const uint32_t m = (uint32_t)__popcll(smask);
        if (m <= HX_MERGE_SHIFT_MAX) {
            u64 it = smask;
            while (it) {
                const uint32_t j = (uint32_t)__ffsll((long long)it) - 1;
                it &= it - 1;
                const u64 e = rdlane64(key, j);
                if (!(e < last_key)) continue;  // the first insert tightened the bound
                const uint32_t pos = (uint32_t)__popcll(__ballot((L[0] & LK_MASK) < e));
                if (pos < 64u) {
                    // into the head: its last entry moves to the front of the tail
                    const u64 carry = rdlane64(L[0], 63);
                    const u64 below = shr1(L[0]);
                    L[0] = (uint32_t)lane < pos ? L[0] : ((uint32_t)lane == pos ? e : below);
                    const u64 tb = shr1(L[1]);
                    L[1] = lane == 0 ? carry : tb;
                } else {
                    const uint32_t tpos = (uint32_t)__popcll(__ballot((L[1] & LK_MASK) < e));
                    const u64 tb = shr1(L[1]);
                    L[1] = (uint32_t)lane < tpos ? L[1] : ((uint32_t)lane == tpos ? e : tb);
                }
                n_cur = min(n_cur + 1, ef);
                if (64u + (uint32_t)lane >= ef) L[1] = LK_INVALID;
                refresh_last(ef);
            }
            return;
        }
        // Three or more: the rank / scatter merge of Lst<R> over the two registers (perm: 128 entries,
        // sv: the 64 behind them)
        u64 *sv = perm + 128;
        const u64 Mm = L[0] & LK_MASK, Tm = L[1] & LK_MASK;
        uint32_t sh0 = 0, sh1 = 0, srank = 0;
        u64 it = smask;
        while (it) {
            const uint32_t j = (uint32_t)__ffsll((long long)it) - 1;
            it &= it - 1;
            const u64 e = rdlane64(key, j);
            sh0 += e < Mm ? 1u : 0u;
            sh1 += e < Tm ? 1u : 0u;
            srank += e < key ? 1u : 0u;
        }
        const uint32_t n_new = min(n_cur + m, ef);
        perm[lane] = LK_INVALID;
        perm[64 + lane] = LK_INVALID;
        asm volatile("" ::: "memory");  // LDS serves a wave's requests in order
        {
            const uint32_t i0 = (uint32_t)lane, i1 = 64u + (uint32_t)lane;
            if (i0 < n_cur && i0 + sh0 < n_new) perm[i0 + sh0] = L[0];
            if (i1 < n_cur && i1 + sh1 < n_new) perm[i1 + sh1] = L[1];
        }
        if (surv) sv[srank] = key;
        asm volatile("" ::: "memory");
        u64 v0 = perm[lane], v1 = perm[64 + lane];
        const bool hole0 = v0 == LK_INVALID && (uint32_t)lane < n_new;
        const bool hole1 = v1 == LK_INVALID && 64u + (uint32_t)lane < n_new;
        const u64 H0 = __ballot(hole0), H1 = __ballot(hole1);
        const uint32_t hr0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(H0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H0, 0u));
        const uint32_t hr1 = (uint32_t)__popcll(H0) +
                             __builtin_amdgcn_mbcnt_hi((uint32_t)(H1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H1, 0u));
        if (hole0) v0 = sv[hr0];
        if (hole1) v1 = sv[hr1];
        L[0] = v0;
        L[1] = v1;
        n_cur = n_new;
        lds_fence();
        refresh_last(ef);
    }
};

// FullVec::distance of one row per lane against the query held in registers (full.rs:23-29): x - y and
// the square two elements per instruction (each element the same single-rounded IEEE operations), the
// sum one serial chain in element order.  No exec mask of its own: the callers request the row's DS / 4
// pieces, run this and keep the result under ONE mask (row registers defined under one mask and consumed
// under another are live across the join for the allocator: 256 VGPRs + 84 AGPRs instead of 234).
template <int DS>
__device__ __forceinline__ float chain_sum(const uint4 (&w)[DS / 4], const float (&qv)[DS]) {
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < DS / 4; p++) {
        const uint32_t dw[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const f32x2 x = {__builtin_bit_cast(float, dw[j]), __builtin_bit_cast(float, dw[j + 1])};
            const f32x2 y = {qv[4 * p + j], qv[4 * p + j + 1]};
            const f32x2 t = x - y;
            const f32x2 t2 = t * t;
            s += t2.x;
            s += t2.y;
        }
    }
    return s;
}
template <int DS>
__device__ __forceinline__ float row_dist(const float *rows, uint32_t id, bool want, const float (&qv)[DS]) {
    float s = 0.0f;
    if (want) {
        const uint4 *src = reinterpret_cast<const uint4 *>(rows + (size_t)id * DS);
        uint4 w[DS / 4];
#pragma unroll
        for (int p = 0; p < DS / 4; p++) w[p] = src[p];
        __builtin_amdgcn_sched_barrier(0);  // every piece requested before the chain starts
        s = chain_sum<DS>(w, qv);
    }
    return __builtin_sqrtf(s);
}

// CK: stages of the cooperative gather in flight (d = 128 only).  The query's 128 values, the stage ring and
// both forms of the gather do not fit 256 registers with four stages: that build parks a few values in AGPRs
// and runs one wave per SIMD -- right for launches of up to four waves per CU (4M x 128d, efSearch 64, batch
// 1024: 0.256 ms against 0.264 ms with two stages), wrong for larger ones (8192 queries: 1.63 against 1.23 ms),
// which take the two-stage build (252 registers, two waves per SIMD).
template <int DS, class LT, int CK = 4>
__global__ void __launch_bounds__(64) hx_lean_f32_kernel(const LeanArgs a) {
    constexpr int R = LT::NR;
    static_assert(DS % 4 == 0, "whole 16-byte pieces");
    // rows of whole 128-byte lines (d = 128: the configs[3] dimension) are gathered cooperatively, eight lanes
    // to a line, and summed by their owner lanes out of an LDS image (coop_rows.inc): no row registers, so the
    // query's 128 values still fit beside the stage ring
    constexpr bool COOP = coop_rows<HNSW_VEC_F32, DS>();
    constexpr bool SPILLV = R >= 6;  // ef > 320: the visited set may continue in HBM (Visited::look2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t q = a.qsel ? a.qsel[blockIdx.x] : blockIdx.x;
    const uint32_t hslots = 1u << a.slots_log2;
    u64 *perm = reinterpret_cast<u64 *>(smem + 4ull * hslots);
    // COOP: the gather's stage image (4 KiB + 64 rank words) SHARES the merge's permutation buffer -- a distance pass
    // ends before its merge starts, nothing in either region lives across a pass -- so that the eight-register list
    // at d = 128 stays within a quarter of the CU's LDS (32 KiB table + 4.5 KiB: four waves per CU; with separate
    // regions it was 41.5 KB, three waves, a batch of 1024 in two rounds: 7.4 us per expansion)
    unsigned char *coop_img = reinterpret_cast<unsigned char *>(perm);
    uint32_t *coop_ids = reinterpret_cast<uint32_t *>(coop_img + HX_COOP_IMG_BYTES);
    Visited vis;
    vis.tab = reinterpret_cast<uint32_t *>(smem);

    uint32_t n_dist = 0, n_exp = 0, sum_deg = 0, n_vis = 0;
    int32_t status = HNSW_OK;
#ifdef HX_STAMPS
    unsigned long long dbg_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

    // ---- the query: every lane holds all DS values (Point::new of a FullVec is the vector itself) ----
    const float *qp = a.Q + (size_t)q * DS;
    QueryRegs<DS> qreg;
    float(&qv)[DS] = qreg.v;
#pragma unroll
    for (int e = 0; e < DS; e++) qv[e] = qp[e];
    {
        bool bad = false;
        for (int e = lane; e < DS; e += 64) {
            const float x = qp[e];
            bad |= (x != x);
        }
        if (__ballot(bad)) status = HNSW_ERR_NAN_INPUT;  // partial_cmp().unwrap() panics in the reference
    }
#pragma unroll
    for (int e = 0; e < DS; e++) asm volatile("" : "+v"(qv[e]));  // pinned to VGPRs for the whole kernel

    // distance of one row per wanting lane (every lane calls it: the cooperative form needs the whole wave)
    auto dist_of = [&](uint32_t id, bool want) __attribute__((always_inline)) -> float {
        if constexpr (COOP) {
            return __builtin_sqrtf(f32_rows_coop<DS, CK, true, true>(reinterpret_cast<const uint8_t *>(a.rows), id, want, qreg,
                                                                    coop_ids, coop_img, lane));
        } else {
            return row_dist<DS>(a.rows, id, want, qv);
        }
    };

    // ---- entry point (template.rs:316-319) ----
    u64 best = LK_INVALID;
    uint32_t cur = a.ep;
    if (status == HNSW_OK) {
        if (cur >= a.n_points) {
            status = HNSW_ERR_ARG;
        } else {
            const float d0 = dist_of(cur, lane == 0);
            const uint32_t bits = rdlane(__builtin_bit_cast(uint32_t, d0), 0);
            n_dist = 1;
            if (__builtin_bit_cast(float, bits) != __builtin_bit_cast(float, bits))
                status = HNSW_ERR_NAN_INPUT;
            else
                best = ((u64)bits << 32) | cur;
        }
    }

    // ---- upper layers, ef = 1 (template.rs:322-324): a greedy walk.  With one slot in `selected`
    // every fresh neighbour below the running best replaces it (searcher.rs:74-94), the next pop is
    // the best found so far, and once an expansion improves nothing the next pop is a leftover above
    // the best and the loop breaks (searcher.rs:41-44).  visited (cleared per layer, searcher.rs:101)
    // only keeps the distance counter honest here. ----
    const uint32_t up_slots = max(hslots >> 2, 64u);
    uint32_t ub_cur = HX_EMPTY_SLOT;
    bool have_ub = false;
    for (int layer = (int)a.nb_layers - 1; layer >= 1 && status == HNSW_OK; layer--) {
        vis.bshift = 32 - (__ffs((int)up_slots) - 1 - 2);
        vis.bmask = (up_slots >> 2) - 1;
        const uint32_t vis_limit = up_slots - (up_slots >> 2);
        vis.clear(up_slots, lane);
        vis.insert(cur, lane == 0);
        n_vis = 1;
        if (!have_ub) {
            ub_cur = uni(a.upper_base[cur]);
            have_ub = true;
        }
        while (status == HNSW_OK) {
            if (ub_cur == HX_EMPTY_SLOT) {  // Graph::neighbors_vec -> NodeNotInGraph (searcher.rs:45-50)
                status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                break;
            }
            n_exp++;
            bool improved = false;
            uint32_t new_cur = cur, new_ub = ub_cur;
            const uint32_t *row = a.adj_up + ((size_t)ub_cur + (uint32_t)layer - 1) * a.S1;
            uint32_t ovf_lo = 0, ovf_hi = 0;
            bool first = true;
            while (true) {
                uint32_t nb = HX_EMPTY_SLOT;
                if (first) {
                    if ((uint32_t)lane < a.S1) nb = row[lane];
                } else {
                    if (ovf_lo + lane < ovf_hi) nb = a.ovf_nbrs[ovf_lo + lane];
                    ovf_lo += 64;
                }
                const bool valid = (int32_t)nb >= 0;
                if (first) {
                    const u64 pm = __ballot((int32_t)nb < 0 && nb != HX_EMPTY_SLOT);
                    if (HX_UNLIKELY(pm != 0)) {
                        const uint32_t o = rdlane(nb, (uint32_t)__ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                        ovf_lo = uni(a.ovf_off[o]);
                        ovf_hi = uni(a.ovf_off[o + 1]);
                    }
                    first = false;
                }
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
                sum_deg += cnt;
                if (HX_LIKELY(cnt != 0)) {
                    if (HX_UNLIKELY(n_vis + cnt > vis_limit)) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    const bool fresh = vis.insert(nb, valid);
                    const u64 fm = __ballot(fresh);
                    const uint32_t nf = (uint32_t)__popcll(fm);
                    n_vis += nf;
                    n_dist += nf;
                    if (HX_LIKELY(fm != 0)) {
                        uint32_t ubn = HX_EMPTY_SLOT;
                        if (fresh) ubn = a.upper_base[nb];  // in flight together with the vector row
                        const float dist = dist_of(nb, fresh);
                        const bool nan = fresh && dist != dist;
                        if (HX_UNLIKELY(__ballot(nan) != 0)) {
                            status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
                            break;
                        }
                        const uint32_t db = fresh ? __builtin_bit_cast(uint32_t, dist) : 0xFFFFFFFFu;
                        const uint32_t mn = wave_min_u32(db);
                        u64 tie = __ballot(fresh && db == mn);
                        uint32_t j = (uint32_t)__ffsll((long long)tie) - 1;
                        uint32_t bid = rdlane(nb, j);
                        tie &= tie - 1;
                        while (tie) {  // equal distances: the smaller id wins (dist.rs:30-38)
                            const uint32_t j2 = (uint32_t)__ffsll((long long)tie) - 1;
                            tie &= tie - 1;
                            const uint32_t id2 = rdlane(nb, j2);
                            if (id2 < bid) {
                                bid = id2;
                                j = j2;
                            }
                        }
                        const u64 k = ((u64)mn << 32) | bid;
                        if (k < best) {
                            best = k;
                            new_cur = bid;
                            new_ub = rdlane(ubn, j);
                            improved = true;
                        }
                    }
                }
                if (HX_LIKELY(ovf_lo >= ovf_hi)) break;
            }
            if (!improved || status != HNSW_OK) break;
            cur = new_cur;
            ub_cur = new_ub;
        }
    }

    // ---- layer 0 with ef (template.rs:326) ----
#ifdef HX_STAMPS
    dbg_acc[7] = __builtin_readcyclecounter() - t_begin;  // query staging + entry point + upper layers
#endif
    const uint32_t ef = max(1u, a.ef);
    LT lst;
    lst.init();
    if (status == HNSW_OK) {
        vis.bshift = 32 - (a.slots_log2 - 2);
        vis.bmask = (hslots >> 2) - 1;
        uint32_t lds_limit = hslots - (hslots >> 2);  // 75 % load at most
        if constexpr (SPILLV) {
            if (a.spill_tab != nullptr && a.lds_limit != 0) lds_limit = min(lds_limit, max(128u, a.lds_limit));
        }
        uint32_t vis_limit = lds_limit;
        if constexpr (SPILLV) {
            if (a.spill_tab != nullptr) {  // a second level in HBM takes what the LDS table cannot (Visited, above)
                vis.gtab = a.spill_tab + ((size_t)blockIdx.x << a.spill_log2);
                vis.gmask = (1u << a.spill_log2) - 1;
                vis.gshift = 32 - a.spill_log2;
                vis_limit = lds_limit + (1u << a.spill_log2) / 2;  // the HBM level at half load
            }
        }
        vis.clear(hslots, lane);
        vis.insert(cur, lane == 0);
        n_vis = 1;
        lst.set_first(best, lane);
        lst.refresh_last(ef);
        const uint32_t S0 = a.S0;
        const bool upper = lane >= 32;
        const uint32_t slot = (uint32_t)lane & 31u;
        // the adjacency rows fetched ahead for the pass after this one (see below)
        uint32_t pre_nb = HX_EMPTY_SLOT, pre_c = HX_EMPTY_SLOT, pre_p = HX_EMPTY_SLOT;

        // the part of c's row that did not fit its S0 slots (degree > S0, rare), 32 ids per pass
        uint32_t ovf_lo = 0, ovf_hi = 0;

        while (true) {
            STAMP(f0);
            if constexpr (SPILLV) {
                // the LDS table has reached its limit (it may hold 64 more: one pass): closed from this pass on
                if (HX_UNLIKELY(!vis.spill && vis.gtab != nullptr && n_vis + 64u > lds_limit)) vis.open_second_level(lane);
            }
            int ppos = -1;
            uint32_t pid = HX_EMPTY_SLOT;
            uint32_t nb = HX_EMPTY_SLOT;
            const bool ovf_pass = ovf_lo < ovf_hi;
            if (HX_UNLIKELY(ovf_pass)) {
                // ---- more of the last candidate's neighbours: the same pass without a pick and without
                // a runner-up (one body for both keeps a single set of row registers in the loop) ----
                if (lane < 32 && ovf_lo + (uint32_t)lane < ovf_hi) nb = a.ovf_nbrs[ovf_lo + (uint32_t)lane];
                ovf_lo += 32;
            } else {
                // ---- pick c (the smallest unexpanded entry) and the runner-up p: one set of ballots ----
                typename LT::Masks U;
                lst.unexp_masks(U);
                const int cpos = lst.take_first(U);
                if (HX_UNLIKELY(cpos < 0)) break;  // candidates exhausted / only worse ones left (searcher.rs:35,41-44)
                ppos = lst.take_first(U);
                const uint32_t cid = lst.id_at((uint32_t)cpos);
                if (ppos >= 0) pid = lst.id_at((uint32_t)ppos);
                lst.mark((uint32_t)cpos, lane);
                n_exp++;
                if (HX_LIKELY(pre_c == cid && pre_p == pid)) {
                    nb = pre_nb;  // both rows were requested during the previous pass
                } else {
                    if (slot < S0 && (!upper || ppos >= 0)) nb = a.adj0[(size_t)(upper ? pid : cid) * S0 + slot];
                }
                pre_c = HX_EMPTY_SLOT;
            }
#ifdef HX_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_acc[6]++;
#endif
            STAMP(f1);
            STAMP_ADD(0, f0, f1);
            // ---- classify the slots.  A row with an overflow pointer (degree > S0) is rare: c's is
            // finished after the merge, a runner-up with one is not speculated on ----
            const u64 pm = ovf_pass ? 0ull : __ballot((int32_t)nb < -1);  // 0x80000000 | overflow row
            const bool spec_ok = ppos >= 0 && (pm >> 32) == 0;
            const bool valid = (int32_t)nb >= 0 && (!upper || spec_ok);
            // ---- visited: every lane looks at its home bucket; c's lanes claim a slot, p's lanes only
            // look (p's real insert happens at its commit, after everything c inserted; a stale "absent"
            // merely evaluates a distance for nothing) ----
            const uint32_t vb = vis.home(nb);
            uint32_t vt0;
            if constexpr (SPILLV)
                vt0 = vis.look2(nb, vb, valid, valid && !upper);  // (6 / 7: absent from both levels, see Visited)
            else
                vt0 = vis.look(nb, vb);
            const uint32_t vt = valid ? vt0 : 4u;       // 0..3 claimable slot, 4 nothing to do, 5 bucket full
            // The rows are requested as soon as the look says "absent" -- for c's lanes that is exactly the
            // set that ends up inserted, barring the rare id that sits in a later bucket of a full home
            // bucket -- and the claims (a second LDS round trip and the loop of the lanes that lost a slot)
            // run while the rows are on their way.
            const bool want_pre = vt != 4u;
            const uint32_t ct = upper ? 4u : vt;
            bool fresh = false;  // c's lanes that inserted their id
            float dist = 0.0f;
            // Request, claims and chain under ONE exec mask -- the lanes whose look said "absent"; the
            // claiming lanes are among them.  (Row registers defined under one mask and consumed under
            // another are live across the join for the allocator: it parked 68 of them in AGPRs, the chain
            // paid a v_accvgpr_read per element and the kernel fell to one wave per SIMD.)  A lane of c
            // that turns out not to be fresh -- its id sat in a later bucket -- evaluates a distance for
            // nothing.
            if constexpr (COOP) {
                // claims first (c's lanes), then one cooperative gather for c's fresh rows and the rows p's
                // look called absent
                const uint32_t old = SPILLV ? vis.claim2(nb, vb, ct) : vis.claim(nb, vb, ct);
                fresh = old == HX_EMPTY_SLOT;
                const bool pend = (ct != 4u) & !fresh;
                if (HX_UNLIKELY(__ballot(pend) != 0)) fresh = SPILLV ? vis.finish2(nb, vb, ct, pend, fresh) : vis.finish(nb, vb, ct, pend, fresh);
                dist = dist_of(nb, upper ? want_pre : fresh);
            } else if (want_pre) {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.rows + (size_t)nb * DS);
                uint4 w[DS / 4];
#pragma unroll
                for (int p = 0; p < DS / 4; p++) w[p] = src[p];
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t old = SPILLV ? vis.claim2(nb, vb, ct) : vis.claim(nb, vb, ct);
                fresh = old == HX_EMPTY_SLOT;
                const bool pend = (ct != 4u) & !fresh;
                if (HX_UNLIKELY(__ballot(pend) != 0)) {
                    fresh = SPILLV ? vis.finish2(nb, vb, ct, pend, fresh) : vis.finish(nb, vb, ct, pend, fresh);
#ifdef HX_STAMPS
                    dbg_acc[11]++;
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef HX_STAMPS
                // (diagnostic build) what is left of the row round trip once the claims are done, then the chain alone
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long g0 = __builtin_readcyclecounter();  // (not stamp_now: that would wait for the rows)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long g1 = stamp_now();
                dbg_acc[2] += g1 - g0;
#endif
                const float s = chain_sum<DS>(w, qv);
                dist = __builtin_sqrtf(s);
#ifdef HX_STAMPS
                asm volatile("" ::"v"(dist) : "memory");
                dbg_acc[3] += stamp_now() - g1;
#endif
            }
            const bool want = upper ? want_pre : fresh;
            const u64 vmask = __ballot(valid), fmask = __ballot(fresh);
            const uint32_t cnt_c = (uint32_t)__popcll(vmask & 0xFFFFFFFFull);
            const uint32_t cnt_p = (uint32_t)__popcll(vmask >> 32);
            const uint32_t nf_c = (uint32_t)__popcll(fmask);
            sum_deg += cnt_c;
            n_vis += nf_c;
            n_dist += nf_c;
            // the table holds at most vis_limit + 64 ids (checked after the inserts: one hand-off less)
            if (HX_UNLIKELY(n_vis > vis_limit)) {
                status = HNSW_ERR_OVERFLOW;
                break;
            }
            const bool nan = want && dist != dist;
            u64 key = LK_INVALID;
            if (want && !nan) key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | nb;
            STAMP(f2);
            STAMP_ADD(1, f1, f2);
            STAMP(f3);
#ifdef HX_STAMPS
            {
                const uint32_t mm = (uint32_t)__popcll(__ballot(!upper && key < lst.last_key));
                if (mm == 0) dbg_acc[12]++; else if (mm <= 2) dbg_acc[13]++; else dbg_acc[14]++;  // static slots: a computed index would move the array to scratch
            }
#endif
            const u64 nanm = __ballot(nan);
            if (HX_UNLIKELY((nanm & 0xFFFFFFFFull) != 0)) {
                status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
                break;
            }
            lst.merge(upper ? LK_INVALID : key, ef, perm, lane);
            STAMP(f3b);
            STAMP_ADD(8, f3, f3b);
            if (HX_UNLIKELY((pm & 0xFFFFFFFFull) != 0)) {  // degree > S0: the rest of c's row goes through the next passes (rare)
                const uint32_t c_ovf = rdlane(nb, (uint32_t)__ffsll((long long)(pm & 0xFFFFFFFFull)) - 1) & ~HX_OVF_FLAG;
                ovf_lo = uni(a.ovf_off[c_ovf]);
                ovf_hi = uni(a.ovf_off[c_ovf + 1]);
                if (ovf_lo < ovf_hi) continue;  // (the runner-up is picked again afterwards)
            }
            // ---- is p the next candidate?  then commit it from the registers ----
            if (HX_UNLIKELY(!spec_ok)) continue;
            typename LT::Masks V;
            lst.unexp_masks(V);
            const int npos = lst.take_first(V);
            if (HX_UNLIKELY(npos < 0)) break;
            if (HX_UNLIKELY(lst.id_at((uint32_t)npos) != pid)) continue;
            // The pass after this one will most likely expand the two entries that follow p (measured:
            // three times out of four when p is committed).  Their adjacency rows are requested now and
            // land while p is being committed; if the next pick is a different pair they are dropped.
            {
                const int apos = lst.take_first(V);
                if (HX_LIKELY(apos >= 0)) {
                    const int bpos = lst.take_first(V);
                    pre_c = lst.id_at((uint32_t)apos);
                    pre_p = bpos >= 0 ? lst.id_at((uint32_t)bpos) : HX_EMPTY_SLOT;
                    pre_nb = HX_EMPTY_SLOT;
                    if (slot < S0 && (!upper || bpos >= 0)) pre_nb = a.adj0[(size_t)(upper ? pre_p : pre_c) * S0 + slot];
                }
            }
            STAMP(f4);
            STAMP_ADD(9, f3b, f4);
            lst.mark((uint32_t)npos, lane);
            n_exp++;
            // Every valid neighbour of p goes through the filter now.  The bucket state seen while
            // speculating is still good for a direct claim: slots fill left to right, so if slot vt is
            // still empty nothing entered the bucket since and the id is absent; if it now holds this very
            // id, c's commit inserted it; anything else sends the lane through the full insert.
            const uint32_t pt = upper ? vt : 4u;
            const uint32_t pold = SPILLV ? vis.claim2(nb, vb, pt) : vis.claim(nb, vb, pt);  // (pt == 6: the HBM claim is exact by itself)
            bool pfresh = pold == HX_EMPTY_SLOT;
            const bool ppend = (pt == 5u) | ((pt < 4u) & !pfresh & (pold != nb));
            if (HX_UNLIKELY(__ballot(ppend) != 0))
                pfresh = SPILLV ? vis.finish2(nb, vb, pt == 5u ? 5u : 0u, ppend, pfresh) : vis.finish(nb, vb, pt == 5u ? 5u : 0u, ppend, pfresh);
            const uint32_t nf_p = (uint32_t)__popcll(__ballot(pfresh));
            sum_deg += cnt_p;
            n_vis += nf_p;
            n_dist += nf_p;
            if (HX_UNLIKELY(n_vis > vis_limit)) {
                status = HNSW_ERR_OVERFLOW;
                break;
            }
            if (HX_UNLIKELY(__ballot(pfresh && nan) != 0)) {
                status = HNSW_ERR_NAN_INPUT;
                break;
            }
            STAMP(f4b);
            STAMP_ADD(5, f4, f4b);
#ifdef HX_STAMPS
            {
                const uint32_t mm = (uint32_t)__popcll(__ballot(pfresh && key < lst.last_key));
                if (mm == 0) dbg_acc[12]++; else if (mm <= 2) dbg_acc[13]++; else dbg_acc[14]++;  // static slots: a computed index would move the array to scratch
                dbg_acc[15]++;
            }
#endif
            lst.merge(pfresh ? key : LK_INVALID, ef, perm, lane);
            STAMP(f5);
            STAMP_ADD(10, f4b, f5);
        }
    }

    // ---- get_top_selected(n) (results.rs:59-61): the first n entries of the ascending list ----
    const uint32_t count = status == HNSW_OK ? min(a.n, lst.n_cur) : 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t idx = LT::idx_of(r, lane);
        if (idx < a.n) {
            const bool have = idx < count;
            a.out_ids[(size_t)q * a.n + idx] = have ? (uint32_t)lst.L[r] : HX_EMPTY_SLOT;
            if (a.out_dists)
                a.out_dists[(size_t)q * a.n + idx] =
                    have ? __builtin_bit_cast(float, (uint32_t)((lst.L[r] & LK_MASK) >> 32)) : __builtin_inff();
        }
    }
    for (uint32_t idx = 64u * R + lane; idx < a.n; idx += 64) {  // n beyond the list capacity: padding
        a.out_ids[(size_t)q * a.n + idx] = HX_EMPTY_SLOT;
        if (a.out_dists) a.out_dists[(size_t)q * a.n + idx] = __builtin_inff();
    }
#ifdef HX_STAMPS
    if (lane == 0 && a.dbg) {
        dbg_acc[4] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 16; i++) a.dbg[(size_t)q * 16 + i] = dbg_acc[i];
    }
#endif
    if (lane == 0) {
        if (a.out_counts) a.out_counts[q] = count;
        hnsw_query_stats st;
        st.n_dist = n_dist;
        st.n_exp = n_exp;
        st.sum_deg = sum_deg;
        st.status = status;
        a.out_stats[q] = st;
    }
}

// =============================================================================================
// QuantVec rows (the reference's shipped VecType, points/src/point.rs:4), d = 100, compact layout: a
// row is one 128-byte line, a lane PAIR evaluates a neighbour (lane h streams half h: four of
// distance_unrolled's eight running sums, quant.rs:14-37).  One adjacency row (32 slots) fills the wave,
// so a pass expands one candidate; the adjacency row of the runner-up is requested at the pick and is
// there when the runner-up is next (three times out of four).  Same list, visited set and upper-layer
// walk as the f32 kernel above.  Against the inline-rows layout of hx_search_kernel this reads 128 B per
// evaluated neighbour instead of a whole 4-KiB block per expansion (measured traffic / algorithmic bytes
// 2.1 there) and needs no 4-GB copy of the rows.
// =============================================================================================
template <class LT>
__global__ void __launch_bounds__(64) hx_lean_q8_kernel(const LeanArgs a) {
    constexpr int R = LT::NR;
    constexpr bool SPILLV = R >= 6;  // ef > 320: the visited set may continue in HBM (Visited::look2)
    constexpr int DS = 100, P = 4, NQ = 4 * (DS / 8) + DS % 8;  // 52 query values per half (half 1 uses 48)
    constexpr uint32_t HALF = 64, ROW = 128, NCH4 = 4 * (DS / 8), REM = DS % 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int h = lane & 1;
    const uint32_t cslot = (uint32_t)lane >> 1;
    const uint32_t q = a.qsel ? a.qsel[blockIdx.x] : blockIdx.x;
    const uint32_t hslots = 1u << a.slots_log2;
    u64 *perm = reinterpret_cast<u64 *>(smem + 4ull * hslots);
    float *yq = reinterpret_cast<float *>(perm + 64 * R + 64);  // behind the merge's two areas
    Visited vis;
    vis.tab = reinterpret_cast<uint32_t *>(smem);
    const uint8_t *rows8 = reinterpret_cast<const uint8_t *>(a.rows);

    uint32_t n_dist = 0, n_exp = 0, sum_deg = 0, n_vis = 0;
    int32_t status = HNSW_OK;
#ifdef HX_STAMPS
    unsigned long long dbg_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

    // ---- the query goes through the same quantiser as a stored vector (Point::new -> QuantVec::new,
    // template.rs:313, quant.rs:41-66) and is kept dequantised, in this lane's half order, in registers ----
    QRegs<NQ> qreg;
    {
        DevView dv{};
        dv.dim = DS;
        dv.half_bytes = HALF;
        dv.nch4 = NCH4;
        dv.rem = REM;
        if (!stage_query<HNSW_VEC_QUANT8>(dv, a.Q + (size_t)q * DS, yq, lane)) status = HNSW_ERR_NAN_INPUT;
#pragma unroll
        for (int e = 0; e < NQ; e++) qreg.v[e] = yq[h * (HALF - 8) + e];
    }
    // distance of the neighbour of this lane pair; the result is on the even lane
    auto eval = [&](uint32_t id, bool want) __attribute__((always_inline)) -> float {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (want) {
            const uint4 *src = reinterpret_cast<const uint4 *>(rows8 + (size_t)id * ROW + (size_t)h * HALF);
            uint4 w[P];
#pragma unroll
            for (int p = 0; p < P; p++) w[p] = src[p];
            __builtin_amdgcn_sched_barrier(0);
            quant_half_sums<P, DS>(w, qreg, h, NCH4, REM, acc);
        }
        // acc.iter().sum(): ((((((a0+a1)+a2)+a3)+a4)+a5)+a6)+a7 with a4..a7 on the odd lane
        const float b0 = pair_swap(acc[0]), b1 = pair_swap(acc[1]), b2 = pair_swap(acc[2]), b3 = pair_swap(acc[3]);
        float s = 0.0f;
        s += acc[0];
        s += acc[1];
        s += acc[2];
        s += acc[3];
        s += b0;
        s += b1;
        s += b2;
        s += b3;
        return __builtin_sqrtf(s);
    };
    // exact insert of the even lanes' ids; both lanes of a pair learn the outcome
    auto pair_insert = [&](uint32_t id, bool ins) __attribute__((always_inline)) -> bool {
        bool f;
        if constexpr (SPILLV) {  // (both levels once the LDS level is closed)
            const bool mine = ins && h == 0;
            const uint32_t b0 = vis.home(id);
            const uint32_t t0 = vis.look2(id, b0, mine, mine);
            const uint32_t t = mine ? t0 : 4u;
            const uint32_t old = vis.claim2(id, b0, t);
            f = old == HX_EMPTY_SLOT;
            f = vis.finish2(id, b0, t, (t != 4u) & !f, f);
        } else {
            f = vis.insert(id, ins && h == 0);
        }
        return (pair_swap_i(f ? 1 : 0) | (f ? 1 : 0)) != 0;
    };

    // ---- entry point (template.rs:316-319) ----
    u64 best = LK_INVALID;
    uint32_t cur = a.ep;
    if (status == HNSW_OK) {
        if (cur >= a.n_points) {
            status = HNSW_ERR_ARG;
        } else {
            const float d0 = eval(cur, lane < 2);
            const uint32_t bits = rdlane(__builtin_bit_cast(uint32_t, d0), 0);
            n_dist = 1;
            if (__builtin_bit_cast(float, bits) != __builtin_bit_cast(float, bits))
                status = HNSW_ERR_NAN_INPUT;
            else
                best = ((u64)bits << 32) | cur;
        }
    }

    // ---- upper layers, ef = 1: the greedy walk of the f32 kernel, a lane pair per neighbour ----
    const uint32_t up_slots = max(hslots >> 2, 64u);
    uint32_t ub_cur = HX_EMPTY_SLOT;
    bool have_ub = false;
    for (int layer = (int)a.nb_layers - 1; layer >= 1 && status == HNSW_OK; layer--) {
        vis.bshift = 32 - (__ffs((int)up_slots) - 1 - 2);
        vis.bmask = (up_slots >> 2) - 1;
        const uint32_t vis_limit = up_slots - (up_slots >> 2);
        vis.clear(up_slots, lane);
        vis.insert(cur, lane == 0);
        n_vis = 1;
        if (!have_ub) {
            ub_cur = uni(a.upper_base[cur]);
            have_ub = true;
        }
        while (status == HNSW_OK) {
            if (ub_cur == HX_EMPTY_SLOT) {  // Graph::neighbors_vec -> NodeNotInGraph (searcher.rs:45-50)
                status = HNSW_ERR_NODE_NOT_IN_GRAPH;
                break;
            }
            n_exp++;
            bool improved = false;
            uint32_t new_cur = cur, new_ub = ub_cur;
            const uint32_t *row = a.adj_up + ((size_t)ub_cur + (uint32_t)layer - 1) * a.S1;
            uint32_t ovf_lo = 0, ovf_hi = 0;
            for (uint32_t c0 = 0; c0 < a.S1 || ovf_lo < ovf_hi; c0 += 32) {
                uint32_t nb = HX_EMPTY_SLOT;
                if (c0 < a.S1) {
                    if (c0 + cslot < a.S1) nb = row[c0 + cslot];
                    const u64 pm = __ballot((int32_t)nb < -1);
                    if (HX_UNLIKELY(pm != 0)) {
                        const uint32_t o = rdlane(nb, (uint32_t)__ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                        ovf_lo = uni(a.ovf_off[o]);
                        ovf_hi = uni(a.ovf_off[o + 1]);
                    }
                } else {
                    if (ovf_lo + cslot < ovf_hi) nb = a.ovf_nbrs[ovf_lo + cslot];
                    ovf_lo += 32;
                }
                const bool valid = (int32_t)nb >= 0;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(valid && h == 0));
                sum_deg += cnt;
                if (cnt == 0) continue;
                if (n_vis + cnt > vis_limit) {
                    status = HNSW_ERR_OVERFLOW;
                    break;
                }
                const bool fresh = pair_insert(nb, valid);
                const u64 fm = __ballot(fresh && h == 0);
                const uint32_t nf = (uint32_t)__popcll(fm);
                n_vis += nf;
                n_dist += nf;
                if (fm == 0) continue;
                uint32_t ubn = HX_EMPTY_SLOT;
                if (fresh && h == 0) ubn = a.upper_base[nb];  // in flight together with the vector row
                const float dist = eval(nb, fresh);
                const bool mine = fresh && h == 0;
                if (__ballot(mine && dist != dist)) {
                    status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
                    break;
                }
                const uint32_t db = mine ? __builtin_bit_cast(uint32_t, dist) : 0xFFFFFFFFu;
                const uint32_t mn = wave_min_u32(db);
                u64 tie = __ballot(mine && db == mn);
                uint32_t j = (uint32_t)__ffsll((long long)tie) - 1;
                uint32_t bid = rdlane(nb, j);
                tie &= tie - 1;
                while (tie) {  // equal distances: the smaller id wins (dist.rs:30-38)
                    const uint32_t j2 = (uint32_t)__ffsll((long long)tie) - 1;
                    tie &= tie - 1;
                    const uint32_t id2 = rdlane(nb, j2);
                    if (id2 < bid) {
                        bid = id2;
                        j = j2;
                    }
                }
                const u64 k = ((u64)mn << 32) | bid;
                if (k < best) {
                    best = k;
                    new_cur = bid;
                    new_ub = rdlane(ubn, j);
                    improved = true;
                }
            }
            if (!improved || status != HNSW_OK) break;
            cur = new_cur;
            ub_cur = new_ub;
        }
    }

    // ---- layer 0 with ef (template.rs:326) ----
#ifdef HX_STAMPS
    dbg_acc[7] = __builtin_readcyclecounter() - t_begin;
#endif
    const uint32_t ef = max(1u, a.ef);
    LT lst;
    lst.init();
    if (status == HNSW_OK) {
        vis.bshift = 32 - (a.slots_log2 - 2);
        vis.bmask = (hslots >> 2) - 1;
        uint32_t lds_limit = hslots - (hslots >> 2);  // 75 % load at most
        if constexpr (SPILLV) {
            if (a.spill_tab != nullptr && a.lds_limit != 0) lds_limit = min(lds_limit, max(128u, a.lds_limit));
        }
        uint32_t vis_limit = lds_limit;
        if constexpr (SPILLV) {
            if (a.spill_tab != nullptr) {  // a second level in HBM takes what the LDS table cannot (Visited)
                vis.gtab = a.spill_tab + ((size_t)blockIdx.x << a.spill_log2);
                vis.gmask = (1u << a.spill_log2) - 1;
                vis.gshift = 32 - a.spill_log2;
                vis_limit = lds_limit + (1u << a.spill_log2) / 2;
            }
        }
        vis.clear(hslots, lane);
        vis.insert(cur, lane == 0);
        n_vis = 1;
        lst.set_first(best, lane);
        lst.refresh_last(ef);
        const uint32_t S0 = a.S0;
        uint32_t pre_nb = HX_EMPTY_SLOT, pre_c = HX_EMPTY_SLOT;  // adjacency row requested ahead

        // One candidate per pass.  (Evaluating the runner-up as well -- its rows fetched together with
        // c's, its chain run when it is committed -- was built, parity-tested and measured slower:
        // 0.196 against 0.180 ms per batch at efSearch 68.  So was requesting the rows right after the
        // look, before the claim, which pays for the f32 rows' 25 loads: 0.186 against 0.180 ms here, and
        // still 0.1788 against 0.1765 ms with request, claim and chain under one exec mask.  So was the
        // f32 kernel's two-candidate pass with the wanted rows packed onto the pairs: + 6 to + 10 %.)
        while (true) {
            STAMP(f0);
            if constexpr (SPILLV) {
                // the LDS table has reached its limit (it may hold 32 more: one pass): closed from this pass on
                if (HX_UNLIKELY(!vis.spill && vis.gtab != nullptr && n_vis + 64u > lds_limit)) vis.open_second_level(lane);
            }
            typename LT::Masks U;
            lst.unexp_masks(U);
            const int cpos = lst.take_first(U);
            if (cpos < 0) break;  // candidates exhausted / only worse ones left (searcher.rs:35,41-44)
            const int ppos = lst.take_first(U);
            const uint32_t cid = lst.id_at((uint32_t)cpos);
            lst.mark((uint32_t)cpos, lane);
            n_exp++;
            uint32_t nb;
            if (pre_c == cid) {
                nb = pre_nb;
            } else {
                nb = HX_EMPTY_SLOT;
                if (cslot < S0) nb = a.adj0[(size_t)cid * S0 + cslot];
            }
            // the runner-up is the next candidate three times out of four: its adjacency row is requested
            // now and lands while this pass filters, evaluates and merges
            pre_c = HX_EMPTY_SLOT;
            if (ppos >= 0) {
                pre_c = lst.id_at((uint32_t)ppos);
                pre_nb = HX_EMPTY_SLOT;
                if (cslot < S0) pre_nb = a.adj0[(size_t)pre_c * S0 + cslot];
            }
            const u64 pm = __ballot((int32_t)nb < -1);  // 0x80000000 | overflow row (degree > S0, rare)
#ifdef HX_STAMPS
            dbg_acc[6]++;
#endif
            STAMP(f1);
            STAMP_ADD(0, f0, f1);
            const bool valid = (int32_t)nb >= 0;
            // ---- visited: the pair looks at the id's home bucket, the even lane claims a slot ----
            const uint32_t vb = vis.home(nb);
            uint32_t vt0;
            if constexpr (SPILLV)
                vt0 = vis.look2(nb, vb, valid && h == 0, valid && h == 0);  // (7: claimed in the HBM level at the look)
            else
                vt0 = vis.look(nb, vb);
            const uint32_t ct = (valid && h == 0) ? vt0 : 4u;
            const uint32_t old = SPILLV ? vis.claim2(nb, vb, ct) : vis.claim(nb, vb, ct);
            bool fresh0 = old == HX_EMPTY_SLOT;
            const bool pend = (ct != 4u) & !fresh0;
            if (__ballot(pend)) fresh0 = SPILLV ? vis.finish2(nb, vb, ct, pend, fresh0) : vis.finish(nb, vb, ct, pend, fresh0);
            const bool fresh = (pair_swap_i(fresh0 ? 1 : 0) | (fresh0 ? 1 : 0)) != 0;
            const u64 vmask = __ballot(valid && h == 0), fmask = __ballot(fresh0);
            const uint32_t nf = (uint32_t)__popcll(fmask);
            sum_deg += (uint32_t)__popcll(vmask);
            n_vis += nf;
            n_dist += nf;
            if (HX_UNLIKELY(n_vis > vis_limit)) {  // the table holds at most vis_limit + 32 ids
                status = HNSW_ERR_OVERFLOW;
                break;
            }
            STAMP(f2);
            STAMP_ADD(1, f1, f2);
            u64 key = LK_INVALID;
            if (fmask) {
                const float dist = eval(nb, fresh);
                const bool mine = fresh0;
                if (__ballot(mine && dist != dist)) {
                    status = HNSW_ERR_NAN_INPUT;  // Dist::cmp would panic (dist.rs:32)
                    break;
                }
                if (mine) key = ((u64)__builtin_bit_cast(uint32_t, dist) << 32) | nb;
            }
            STAMP(f3);
            STAMP_ADD(3, f2, f3);
            lst.merge(key, ef, perm, lane);
            STAMP(f4);
            STAMP_ADD(8, f3, f4);
            if (pm) {  // degree > S0: the rest of the row
                const uint32_t c_ovf = rdlane(nb, (uint32_t)__ffsll((long long)pm) - 1) & ~HX_OVF_FLAG;
                const uint32_t lo = uni(a.ovf_off[c_ovf]), hi = uni(a.ovf_off[c_ovf + 1]);
                for (uint32_t base = lo; base < hi && status == HNSW_OK; base += 32) {
                    const uint32_t i = base + cslot;
                    const bool ov = i < hi;
                    const uint32_t onb = ov ? a.ovf_nbrs[i] : HX_EMPTY_SLOT;
                    const uint32_t ocnt = (uint32_t)__popcll(__ballot(ov && h == 0));
                    sum_deg += ocnt;
                    if (n_vis + ocnt > vis_limit) {
                        status = HNSW_ERR_OVERFLOW;
                        break;
                    }
                    if constexpr (SPILLV) {
                        // a long overflow list must not run the LDS table full between two passes: the same switch
                        // as at the top of a pass, before every 32 ids
                        if (HX_UNLIKELY(!vis.spill && vis.gtab != nullptr && n_vis + 64u > lds_limit)) vis.open_second_level(lane);
                    }
                    const bool ofresh = pair_insert(onb, ov);
                    const uint32_t onf = (uint32_t)__popcll(__ballot(ofresh && h == 0));
                    n_vis += onf;
                    n_dist += onf;
                    if (onf == 0) continue;
                    const float odist = eval(onb, ofresh);
                    const bool omine = ofresh && h == 0;
                    if (__ballot(omine && odist != odist)) {
                        status = HNSW_ERR_NAN_INPUT;
                        break;
                    }
                    lst.merge(omine ? (((u64)__builtin_bit_cast(uint32_t, odist) << 32) | onb) : LK_INVALID, ef, perm,
                              lane);
                }
                if (status != HNSW_OK) break;
            }
        }
    }

    // ---- get_top_selected(n) (results.rs:59-61): the first n entries of the ascending list ----
    const uint32_t count = status == HNSW_OK ? min(a.n, lst.n_cur) : 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t idx = LT::idx_of(r, lane);
        if (idx < a.n) {
            const bool have = idx < count;
            a.out_ids[(size_t)q * a.n + idx] = have ? (uint32_t)lst.L[r] : HX_EMPTY_SLOT;
            if (a.out_dists)
                a.out_dists[(size_t)q * a.n + idx] =
                    have ? __builtin_bit_cast(float, (uint32_t)((lst.L[r] & LK_MASK) >> 32)) : __builtin_inff();
        }
    }
    for (uint32_t idx = 64u * R + lane; idx < a.n; idx += 64) {  // n beyond the list capacity: padding
        a.out_ids[(size_t)q * a.n + idx] = HX_EMPTY_SLOT;
        if (a.out_dists) a.out_dists[(size_t)q * a.n + idx] = __builtin_inff();
    }
#ifdef HX_STAMPS
    if (lane == 0 && a.dbg) {
        dbg_acc[4] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 16; i++) a.dbg[(size_t)q * 16 + i] = dbg_acc[i];
    }
#endif
    if (lane == 0) {
        if (a.out_counts) a.out_counts[q] = count;
        hnsw_query_stats st;
        st.n_dist = n_dist;
        st.n_exp = n_exp;
        st.sum_deg = sum_deg;
        st.status = status;
        a.out_stats[q] = st;
    }
}

#include "pair_kernel.inc"

template <class LT>
int launch_lean_q8(const LeanArgs &a_in, uint32_t nblocks, hipStream_t stream) {
    constexpr int R = LT::NR;
    LeanArgs a = a_in;
    // six- to eight-register lists (320 < ef <= 512): 32 KiB of LDS table + a second level in HBM, as launch_lean_one
    struct Scratch {
        void *p = nullptr;
        hipStream_t st = nullptr;
        ~Scratch() {
            if (p) (void)hipFreeAsync(p, st);
        }
    } sp;
    static const bool two_level = !(getenv("HNSW_MI355X_VISITED_2L") && atoi(getenv("HNSW_MI355X_VISITED_2L")) == 0);
    if (R >= 6 && two_level && a.slots_log2 >= 13) {  // (also where 32 KiB would do for most queries: the few that fill it go on in HBM instead of being run again)
        const uint32_t glog2 = std::max(15u, a.slots_log2 + 1);
        sp.st = stream;
        if (hipMallocAsync(&sp.p, ((size_t)nblocks << glog2) * 4, stream) != hipSuccess) {
            (void)hipGetLastError();
            sp.p = nullptr;
        } else {
            a.spill_tab = static_cast<uint32_t *>(sp.p);
            a.spill_log2 = glog2;
            a.slots_log2 = 13;
            if (const char *e = getenv("HNSW_MI355X_VISITED_2L_LIMIT")) a.lds_limit = (uint32_t)atoi(e);
        }
    }
    const size_t lds = (4ull << a.slots_log2) + (64ull * R + 64) * 8 + 2 * 56 * 4;
    auto kern = hx_lean_q8_kernel<LT>;
    if (lds > 160 * 1024) {
        set_error("search needs %zu bytes of LDS (> 160 KiB)", lds);
        return HNSW_ERR_ARG;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("search kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

template <int DS, class LT, int CK = 4>
int launch_lean_one(const LeanArgs &a_in, uint32_t nblocks, hipStream_t stream) {
    constexpr int R = LT::NR;
    LeanArgs a = a_in;
    // eight-register lists (320 < ef <= 512): the LDS table stays at 32 KiB -- four waves per CU, a batch of 1024 in
    // one round -- and a second level in HBM (stream-ordered scratch, 128 KiB per query) takes the ids beyond it
    struct Scratch {
        void *p = nullptr;
        hipStream_t st = nullptr;
        ~Scratch() {
            if (p) (void)hipFreeAsync(p, st);
        }
    } sp;
    static const bool two_level = !(getenv("HNSW_MI355X_VISITED_2L") && atoi(getenv("HNSW_MI355X_VISITED_2L")) == 0);
    if (R >= 6 && two_level && a.slots_log2 >= 13) {  // (also where 32 KiB would do for most queries: the few that fill it go on in HBM instead of being run again)
        const uint32_t glog2 = std::max(15u, a.slots_log2 + 1);
        sp.st = stream;
        if (hipMallocAsync(&sp.p, ((size_t)nblocks << glog2) * 4, stream) != hipSuccess) {
            (void)hipGetLastError();
            sp.p = nullptr;  // no scratch: the one-level table serves (two waves per CU)
        } else {
            a.spill_tab = static_cast<uint32_t *>(sp.p);
            a.spill_log2 = glog2;
            a.slots_log2 = 13;
            if (const char *e = getenv("HNSW_MI355X_VISITED_2L_LIMIT")) a.lds_limit = (uint32_t)atoi(e);  // (tests: close the LDS level early)
        }
    }
    const size_t lds = (4ull << a.slots_log2) +
                       std::max<size_t>((64ull * R + 64) * 8, coop_rows<HNSW_VEC_F32, DS>() ? HX_COOP_IMG_BYTES + 256 : 0);
    auto kern = hx_lean_f32_kernel<DS, LT, CK>;
    if (lds > 160 * 1024) {
        set_error("search needs %zu bytes of LDS (> 160 KiB)", lds);
        return HNSW_ERR_ARG;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return HNSW_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("search kernel launch: %s", hipGetErrorString(e));
        return HNSW_ERR_HIP;
    }
    return HNSW_OK;
}

}  // namespace

// Whether the lean kernel serves this search: the whole ann_by_vector descent over f32 rows of a
// dimension it is built for, lists of up to four registers (ef <= 256).  HNSW_MI355X_LEAN=0 turns it off (A/B).
bool lean_applicable(const DevView &v, const SearchArgs &a, uint32_t ef_max) {
    static const bool enabled = !(getenv("HNSW_MI355X_LEAN") && atoi(getenv("HNSW_MI355X_LEAN")) == 0);
    if (!enabled) return false;
    static const bool q8_enabled = !(getenv("HNSW_MI355X_LEAN_Q8") && atoi(getenv("HNSW_MI355X_LEAN_Q8")) == 0);
    static const bool coop128 = !(getenv("HNSW_MI355X_LEAN_128") && atoi(getenv("HNSW_MI355X_LEAN_128")) == 0);
    const bool f32_ok = v.kind == HNSW_VEC_F32 && ((v.dim == 100 && v.row_stride == 400) ||
                                                     (coop128 && v.dim == 128 && v.row_stride == 512));
    const bool q8_ok = q8_enabled && v.kind == HNSW_VEC_QUANT8 && v.dim == 100 && v.row_stride == 128 && v.half_bytes == 64;
    if (!f32_ok && !q8_ok) return false;
    if (a.entries != nullptr || a.layer_lo != 0 || a.layer_hi != (int32_t)v.nb_layers - 1) return false;
    if (a.layer_hi > 0 && a.ef_upper != 1) return false;
    // lists: one register (ef <= 64), head + tail (<= 128), four interleaved registers (<= 256, HNSW_MI355X_LEAN_WIDE=0
    // sends those to the generic kernel, for A/B runs)
    static const bool wide = !(getenv("HNSW_MI355X_LEAN_WIDE") && atoi(getenv("HNSW_MI355X_LEAN_WIDE")) == 0);
    // (five to eight registers, 256 < ef <= 512: d = 100, both kinds -- round 3: f32 only, x 1.28 against the generic kernel,
    // quant8 gained nothing with the one-level 64-KiB table; round 4: the two-level visited set keeps four waves per CU)
    static const bool q8_wide = !(getenv("HNSW_MI355X_LEAN_Q8_WIDE") && atoi(getenv("HNSW_MI355X_LEAN_Q8_WIDE")) == 0);
    const bool to512 = (v.dim == 100 && (v.kind == HNSW_VEC_F32 || q8_wide)) || (v.dim == 128 && v.kind == HNSW_VEC_F32);
    if (v.S0 > 32 || v.S1 > 64 || ef_max > (wide ? (to512 ? 512u : 256u) : 128u) || (a.flags & 1u)) return false;
    return true;
}

int launch_lean(const DevView &v, const SearchArgs &s, uint32_t nblocks, uint32_t slots_log2, hipStream_t stream) {
    LeanArgs a{};
    a.rows = reinterpret_cast<const float *>(v.rows);
    a.adj0 = v.adj0;
    a.adj_up = v.adj_up;
    a.upper_base = v.upper_base;
    a.ovf_off = v.ovf_off;
    a.ovf_nbrs = v.ovf_nbrs;
    a.Q = s.Q;
    a.qsel = s.qsel;
    a.out_ids = s.out_ids;
    a.out_dists = s.out_dists;
    a.out_counts = s.out_counts;
    a.out_stats = s.out_stats;
    a.n_points = v.n_points;
    a.ep = v.ep;
    a.nb_layers = v.nb_layers;
    a.S0 = v.S0;
    a.S1 = v.S1;
    a.ef = s.ef_bottom;
    a.n = s.n;
    a.slots_log2 = slots_log2;
    a.dbg = s.dbg;
    // 64 < ef <= 128: head + tail list (HNSW_MI355X_LIST=interleaved: round 2's two-register list, for A/B runs); 128 < ef <= 256: four interleaved registers
    static const bool interleaved = getenv("HNSW_MI355X_LIST") && getenv("HNSW_MI355X_LIST")[0] == 'i';
    if (v.kind == HNSW_VEC_QUANT8) {
        if (a.ef <= 64) return launch_lean_q8<Lst<1>>(a, nblocks, stream);
        if (a.ef > 448) return launch_lean_q8<Lst<8>>(a, nblocks, stream);  // (round 4: 256 < ef <= 512 with the two-level visited set)
        if (a.ef > 384) return launch_lean_q8<Lst<7>>(a, nblocks, stream);
        if (a.ef > 320) return launch_lean_q8<Lst<6>>(a, nblocks, stream);
        if (a.ef > 256) return launch_lean_q8<Lst<5>>(a, nblocks, stream);
        if (a.ef > 128) return launch_lean_q8<Lst<4>>(a, nblocks, stream);
        if (interleaved) return launch_lean_q8<Lst<2>>(a, nblocks, stream);
        return launch_lean_q8<LstHT>(a, nblocks, stream);
    }
    if (v.dim == 128) {  // whole-line rows: the cooperative gather (HNSW_MI355X_LEAN_128=0: the generic kernel, for A/B runs)
        static const uint32_t n_cu = [] {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            return (uint32_t)cus;
        }();
        const bool few = nblocks <= 4 * n_cu;  // at most one wave per SIMD: the four-stage build (see the kernel)
        if (a.ef <= 64) return few ? launch_lean_one<128, Lst<1>, 4>(a, nblocks, stream) : launch_lean_one<128, Lst<1>, 2>(a, nblocks, stream);
        if (a.ef > 384) return launch_lean_one<128, Lst<8>, 2>(a, nblocks, stream);  // (round 4: 256 < ef <= 512, two-level visited set)
        if (a.ef > 256) return launch_lean_one<128, Lst<6>, 2>(a, nblocks, stream);
        if (a.ef > 128) return launch_lean_one<128, Lst<4>, 2>(a, nblocks, stream);
        return few ? launch_lean_one<128, LstHT, 4>(a, nblocks, stream) : launch_lean_one<128, LstHT, 2>(a, nblocks, stream);
    }
    // two waves per query (pair_kernel.inc): launches that leave the SIMDs a wave or two each
    static const int pair_mode = getenv("HNSW_MI355X_PAIR") ? atoi(getenv("HNSW_MI355X_PAIR")) : 0;
    // (a.qsel: a re-run of the queries a first launch gave up -- those take the one-wave kernel)
    if (pair_mode != 0 && a.ef <= 128 && !interleaved && a.qsel == nullptr) {
        if (a.ef <= 64) return launch_pair<100, Lst<1>>(a, nblocks, stream);
        return launch_pair<100, LstHT>(a, nblocks, stream);
    }
    if (a.ef <= 64) return launch_lean_one<100, Lst<1>>(a, nblocks, stream);
    // 256 < ef <= 512: as many interleaved registers as the list needs (round 4: five to eight; every register costs
    // each pick, mark and merge -- 2.1 us per expansion with four, 2.9 with eight at the same ef)
    if (a.ef > 448) return launch_lean_one<100, Lst<8>>(a, nblocks, stream);
    if (a.ef > 384) return launch_lean_one<100, Lst<7>>(a, nblocks, stream);
    if (a.ef > 320) return launch_lean_one<100, Lst<6>>(a, nblocks, stream);
    if (a.ef > 256) return launch_lean_one<100, Lst<5>>(a, nblocks, stream);
    if (a.ef > 128) return launch_lean_one<100, Lst<4>>(a, nblocks, stream);
    if (interleaved) return launch_lean_one<100, Lst<2>>(a, nblocks, stream);
    return launch_lean_one<100, LstHT>(a, nblocks, stream);
}

}  // namespace hx
