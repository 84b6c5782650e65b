// rlap_flow.hip -- the multi-CU ("dataflow") exact elimination for o_v = random on gfx950 (protocol and proof sketch: rlap_flow.h;
// the same steps run by randomly interleaved virtual waves on the CPU: tests/csrc/host_mirror.cc::mirror_flow_chol).
//
// Replaces the loop of RandomPreconditioner::getSchurComplement, /root/reference/rlap/csrc/preconditioner.cc:713-787, for one graph
// or a batch: every position of the (known) order is claimed by ONE wave -- a 64-thread workgroup, four to a compute unit, on
// every XCD -- which waits until the vertex's column is final (pend == 0), gathers it, sorts / merges / orders it with the exact
// std::sort restatements (rlap_wave_sort.h), publishes its uniform count for the look-back, samples and commits.
//
// Memory between workgroups (MI355X: the L1 of a CU is never refreshed by other CUs' stores, the L2s of the eight XCDs are
// not coherent with each other): every word that one workgroup writes and another reads in this launch -- entries, tags,
// directory words, counters, look-back words -- is written with an agent-scope (sc1, write-through) store or atomic and read
// with an agent-scope (sc1) load or atomic; a wave waits for its stores (s_waitcnt vmcnt(0)) before the atomic decrements that
// release its neighbours.  No fences, no plain loads of shared words.  What the launch only reads (colptr, order, positions,
// uniforms) is loaded normally.
// Compiled with -ffp-contract=off like the rest (the reference is built without FMA, setup.py:26-37).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>
#include <cstddef>
#include <algorithm>

#include "rlap_core.h"
#include "rlap_flow.h"
#include "rlap_kernels.h"
#include "rlap_wave_sort.h"

namespace rlap {

// ---------------------------------------------------------------------------
// agent-scope accesses
// ---------------------------------------------------------------------------
#define RLX __ATOMIC_RELAXED
#define AGT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ int32_t ag_ld(const int32_t* p) { return __hip_atomic_load(p, RLX, AGT); }
__device__ __forceinline__ void ag_st(int32_t* p, int32_t v) { __hip_atomic_store(p, v, RLX, AGT); }
__device__ __forceinline__ unsigned long long ag_ld64(const unsigned long long* p) { return __hip_atomic_load(p, RLX, AGT); }
__device__ __forceinline__ void ag_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, RLX, AGT); }
__device__ __forceinline__ int32_t ag_add(int32_t* p, int32_t v) { return __hip_atomic_fetch_add(p, v, RLX, AGT); }
// A 16-byte entry: ONE write-through / bypassing access (buffer_load/store_dwordx4 sc1) while the slot array is within a buffer
// resource's 4 GB, else two 8-byte words (no reader while a writer is at it either way: rlap_flow.h).  One fabric request per entry
// instead of two: under the load of a batch every request counts.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct SlotMem { __amdgpu_buffer_rsrc_t rsrc; bool wide; Slot* e; };
__device__ __forceinline__ SlotMem slot_mem(const Arrays& A) {
    SlotMem M;
    const unsigned long long bytes = (unsigned long long)(uint32_t)A.slot_cap * 16ull;
    M.wide = bytes < 0xFFFFFFF0ull;
    M.rsrc = __builtin_amdgcn_make_buffer_rsrc(A.e, 0, M.wide ? (int)(uint32_t)bytes : 0, 0x00020000);
    M.e = A.e;
    return M;
}
__device__ __forceinline__ Slot ag_ld_slot(const SlotMem& M, int32_t sidx) {
    Slot s;
    if (M.wide) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(M.rsrc, (int)((uint32_t)sidx * 16u), 0, 16);
        s.val = __longlong_as_double((long long)((unsigned long long)v.x | ((unsigned long long)v.y << 32)));
        s.nbr = (int32_t)v.z; s.twin = (int32_t)v.w;
    } else {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(M.e + sidx);
        const unsigned long long w0 = ag_ld64(q), w1 = ag_ld64(q + 1);
        s.val = __longlong_as_double((long long)w0);
        s.nbr = (int32_t)(uint32_t)(w1 & 0xFFFFFFFFull);
        s.twin = (int32_t)(uint32_t)(w1 >> 32);
    }
    return s;
}
__device__ __forceinline__ void ag_st_slot(const SlotMem& M, int32_t sidx, double val, int32_t nbr, int32_t twin) {
    const unsigned long long w0 = (unsigned long long)__double_as_longlong(val);
    if (M.wide) {
        u32x4 v;
        v.x = (uint32_t)w0; v.y = (uint32_t)(w0 >> 32); v.z = (uint32_t)nbr; v.w = (uint32_t)twin;
        __builtin_amdgcn_raw_buffer_store_b128(v, M.rsrc, (int)((uint32_t)sidx * 16u), 0, 16);
    } else {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(M.e + sidx);
        ag_st64(q, w0);
        ag_st64(q + 1, (unsigned long long)(uint32_t)nbr | ((unsigned long long)(uint32_t)twin << 32));
    }
}
__device__ __forceinline__ void ag_st_slot_val(Slot* p, double val) { ag_st64(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(val)); }
#define DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")


// ---------------------------------------------------------------------------
// working storage of one column: LDS arrays (short columns), or arrays in the bump-allocated scratch with the sort's records,
// stop lists and tables in the wave's LDS while they fit (long ones)
// ---------------------------------------------------------------------------
// long columns: sort records, stop lists and tables in the wave's LDS while they fit (false: everything in global scratch)
constexpr bool FLOW_BIG_IN_LDS = true;
template <int EC>
struct FlowLds {
    static constexpr bool SMALL = true;
    static constexpr int CAP = EC;
    SRec rec[EC];                 // sort records; after the ordering: cum[EC] | newv[EC]
    double a_val[EC], b_val[EC];
    alignas(16) double skey[EC + 8];
    int32_t a_nbr[EC], a_twin[EC], a_tag[EC];      // a_tag: tags while the list order is restored, then the twins of merged duplicates
    int32_t b_nbr[EC], b_twin[EC], b_dup[EC], b_pos[EC];
    int32_t f_dup[EC], f_pos[EC];                  // f_dup doubles as the register sort's scratch and the level sort's second table
    int32_t ksel[EC];                              // doubles as the level sort's first table
    uint16_t ulist[EC + 2], dlist[EC + 2];
    uint32_t segmark[(EC + 31) / 32 + 1];
    int32_t stk[3 * 48];
    int32_t tmp[160];                              // register sort's scratch in the long-column form
    __device__ __forceinline__ double* cum() { return reinterpret_cast<double*>(rec); }
    __device__ __forceinline__ double* newv() { return reinterpret_cast<double*>(rec) + EC; }
    __device__ __forceinline__ SRec& R(int i) { return rec[i]; }
};
// the same LDS block seen by a long column: only what the sorts touch (and afterwards the cumulative weights / new weights)
constexpr int FLOW_LVL_BIG = 512;    // long columns up to this many keys take the level-synchronous sort (8 elements per lane: more live in scratch memory)
template <int EC, int NW>
struct FlowSortLds {
    static constexpr size_t AVAIL = offsetof(FlowLds<EC>, stk);
    static constexpr int LVL = (AVAIL >= (size_t)FLOW_LVL_BIG * 26 + 4096) ? FLOW_LVL_BIG : 0;
    static constexpr int NT = LVL > 0 ? NW : 1;                                   // one pair of tables per sorting wave
    static constexpr int BIGL = (int)((AVAIL - (size_t)LVL * 6 * NT - 512) / 23) & ~7;   // records + two 16-bit lists + marks + queue share per entry
    static constexpr int QCAP = NW > 1 ? BIGL / 17 + 8 : 1;
    SRec rec[BIGL];
    uint16_t ulist[BIGL + 2], dlist[BIGL + 2];
    uint32_t segmark[BIGL / 32 + 2];
    uint32_t tab[LVL > 0 ? NT * LVL : 1];
    uint16_t tab2[LVL > 0 ? NT * LVL : 2];
    int32_t segq[2 * 3 * QCAP];
    int32_t qcnt[4];
};
// ... and by a column too long for that: the sort moves 16-bit indices, the keys stay where they are (8 + 2 + 4 bytes per entry)
template <int EC, int NW>
struct FlowIdxLds {
    static constexpr size_t AVAIL = offsetof(FlowLds<EC>, stk);
    static constexpr int LVL = FlowSortLds<EC, NW>::LVL;
    static constexpr int NT = LVL > 0 ? NW : 1;
    static constexpr int IDXL = (int)((AVAIL - (size_t)LVL * 6 * NT - 512) / 17) & ~7;
    static constexpr int QCAP = NW > 1 ? IDXL / 17 + 8 : 1;
    double key[IDXL];
    uint16_t idx[IDXL + 2], ulist[IDXL + 2], dlist[IDXL + 2];
    uint32_t segmark[IDXL / 32 + 2];
    uint32_t tab[LVL > 0 ? NT * LVL : 1];
    uint16_t tab2[LVL > 0 ? NT * LVL : 2];
    int32_t segq[2 * 3 * QCAP];
    int32_t qcnt[4];
};
// (the keys of an index sort are in the workgroup's LDS, FlowIdxLds::key: said so, the comparisons inside wave_lvl_sort read them with ds_ instructions)
struct IdxKeyLess { const double* key; __device__ bool operator()(uint16_t a, uint16_t b) const { __builtin_assume(RLAP_IS_LDS(key)); return key[a] < key[b]; } };
struct IdxKeyGreater { const double* key; __device__ bool operator()(uint16_t a, uint16_t b) const { __builtin_assume(RLAP_IS_LDS(key)); return key[a] > key[b]; } };
#ifndef RLAP_FLOW_MW_SORT
#define RLAP_FLOW_MW_SORT 1
#endif
enum { FCMD_SORT = 1, FCMD_EXIT = 2, FCMD_RECUR = 3 };
// MW: the owner's sorts of more than FLOW_LVL_BIG keys go to all waves of the workgroup (flow_block_sort).  That function is
// inlined into its callers on purpose: as a called function that itself calls wave_lvl_sort it left the elimination kernel reading
// wrong LDS data afterwards (a loop-invariant address prepared at kernel entry held other values in most lanes; a freshly computed
// address, or a flat load of the same generic address, read the right ones) -- DESIGN.md 8.6.  RLAP_FLOW_MW_SORT=0 builds the
// kernel with single-wave sorts only.
template <int EC, int NW, bool MW = false>
struct FlowBig {
    static constexpr bool SMALL = false;
    static constexpr int EC_ = EC, NW_ = NW;
    static constexpr bool MWSORT = MW;
    int gtab_words;          // words in a row from b_pos on that may hold the id set (flow_has_dup)
    long long* xdbg;   // experiment builds (RLAP_FLOW_MW_SORT): profile buffer + 12
    FlowSortLds<EC, NW>* S;  // derived from the kernel's __shared__ block (the compiler sees LDS addresses)
    FlowIdxLds<EC, NW>* X;   // the same block, index-sort view
    int32_t* stk; int32_t* tmp;
    int32_t* cmd;            // NW > 1: the workgroup's helper waves are told here (FCMD_*, kind, count, descending)
    int kind;                // the sort's records: 1 in LDS, 2 indices + keys in LDS, 0 in the scratch (grec, gulist, gdlist, gsegmark)
    SRec* grec; uint16_t *gulist, *gdlist; uint32_t* gsegmark;
    double *a_val, *b_val, *skey, *cum_, *newv_;
    int32_t *a_nbr, *a_twin, *a_tag, *b_nbr, *b_twin, *b_dup, *b_pos, *f_dup, *f_pos, *ksel;
    __device__ __forceinline__ double* cum() { return cum_; }
    __device__ __forceinline__ double* newv() { return newv_; }
    __device__ __forceinline__ SRec R(int i) const {
        if (kind == 1) return S->rec[i];
        if (kind == 2) { SRec r; const int q = X->idx[i]; r.key = X->key[q]; r.idx = q; r.aux = 0; return r; }
        return grec[i];
    }
};
// The multi-wave sort of a long column whose records (kind 1) or indices + keys (kind 2) are staged in the workgroup's LDS:
// called by ALL NW waves (the owner and its helpers) with the same arguments.
// (inlined into its callers: called as a function that itself calls wave_lvl_sort, the elimination kernel read wrong LDS data
// afterwards -- DESIGN.md 8.6)
#define FLOW_BLOCK_SORT_INLINE __forceinline__
template <int EC, int NW>
__device__ FLOW_BLOCK_SORT_INLINE bool flow_block_sort(FlowSortLds<EC, NW>* S, FlowIdxLds<EC, NW>* X, int kind, int cnt, int greater, int wave, int lane) {
    constexpr int LREG = FLOW_LVL_BIG / 64;
    if (kind == 1) {
        if (greater) return block_std_sort_lvl<SRec, SRecGreaterKey, NW, LREG>(S->rec, cnt, SRecGreaterKey(), S->ulist, S->dlist, S->segmark, S->segq, S->qcnt, FlowSortLds<EC, NW>::QCAP, S->tab, S->tab2, wave, lane);
        return block_std_sort_lvl<SRec, SRecLessKey, NW, LREG>(S->rec, cnt, SRecLessKey(), S->ulist, S->dlist, S->segmark, S->segq, S->qcnt, FlowSortLds<EC, NW>::QCAP, S->tab, S->tab2, wave, lane);
    }
    if (greater) return block_std_sort_lvl<uint16_t, IdxKeyGreater, NW, LREG>(X->idx, cnt, IdxKeyGreater{X->key}, X->ulist, X->dlist, X->segmark, X->segq, X->qcnt, FlowIdxLds<EC, NW>::QCAP, X->tab, X->tab2, wave, lane);
    return block_std_sort_lvl<uint16_t, IdxKeyLess, NW, LREG>(X->idx, cnt, IdxKeyLess{X->key}, X->ulist, X->dlist, X->segmark, X->segq, X->qcnt, FlowIdxLds<EC, NW>::QCAP, X->tab, X->tab2, wave, lane);
}
static_assert(FLOW_SCR_BYTES >= 16 + 5 * 8 + 10 * 4 + 2 * 2 + 1 && FLOW_SCR_BYTES % 8 == 0, "records, five doubles, ten ints, two 16-bit lists and the segment marks per entry");

// std::sort order of the staged keys B.skey[0..cnt) into the records {key, source index} (rlap_wave_sort.h)
template <bool GREATER, int EC>
__device__ __forceinline__ void flow_sort(FlowLds<EC>& B, int cnt, int lane) {
    typedef typename std::conditional<GREATER, SRecGreaterKey, SRecLessKey>::type Cmp;
    if (cnt <= 64) {
        double key = lane < cnt ? B.skey[lane] : 0.0;
        int idx = lane, pos = lane;
        const bool ok = wave_sort64<GREATER>(key, idx, cnt, lane, B.f_dup, &pos);
        if (ok) {
            if (lane < cnt) { B.rec[pos].key = key; B.rec[pos].idx = idx; }
            WAVE_SYNC();
            return;
        }
    }
    WAVE_SYNC();
    for (int i = lane; i < cnt; i += 64) { B.rec[i].key = B.skey[i]; B.rec[i].idx = i; }
    WAVE_SYNC();
    bool ok;
    constexpr int REGMAX = (EC + 63) / 64;
    uint32_t* tab = reinterpret_cast<uint32_t*>(B.ksel);
    uint16_t* tab2 = reinterpret_cast<uint16_t*>(B.f_dup);
    const WaveSortPtrs WP = {B.ulist, B.dlist, B.segmark, B.stk};
    // (the level-synchronous form keeps 10 to 13 registers per element and lane: beyond 8 elements per lane it lives in scratch
    // memory -- 370 reloads at 16 -- so longer arrays take partitions on top and that form for the segments of 512 and less)
    if (cnt <= 128) ok = wave_lvl_sort<SRec, Cmp, 2>(B.rec, cnt, Cmp(), B.ulist, B.dlist, tab, tab2, lane);
    else if (REGMAX <= 4 || cnt <= 256) ok = wave_lvl_sort<SRec, Cmp, (REGMAX < 4 ? REGMAX : 4)>(B.rec, cnt, Cmp(), B.ulist, B.dlist, tab, tab2, lane);
    else if (REGMAX <= 8 || cnt <= 512) ok = wave_lvl_sort<SRec, Cmp, (REGMAX < 8 ? REGMAX : 8)>(B.rec, cnt, Cmp(), B.ulist, B.dlist, tab, tab2, lane);
    else ok = wave_std_sort_hyb<SRec, Cmp, 8>(B.rec, cnt, Cmp(), WP, tab, tab2, lane);
    if (!ok) {   // depth limit: std::sort heap-sorts there; start over with the form that follows it
        WAVE_SYNC();
        for (int i = lane; i < cnt; i += 64) { B.rec[i].key = B.skey[i]; B.rec[i].idx = i; }
        WAVE_SYNC();
        wave_std_sort<SRec, Cmp, 0>(B.rec, cnt, Cmp(), WP, lane);
    }
    WAVE_SYNC();
}
template <bool GREATER, int EC, int NW, bool MW>
__device__ __forceinline__ void flow_sort(FlowBig<EC, NW, MW>& B, int cnt, int lane) {
    typedef typename std::conditional<GREATER, SRecGreaterKey, SRecLessKey>::type Cmp;
    typedef typename std::conditional<GREATER, IdxKeyGreater, IdxKeyLess>::type ICmp;
    constexpr int LVL = FlowSortLds<EC, NW>::LVL;
    if (B.kind == 1 || B.kind == 2) {
        FlowSortLds<EC, NW>& S = *B.S;
        FlowIdxLds<EC, NW>& X = *B.X;
        auto stage = [&]() {
            if (B.kind == 1) { for (int i = lane; i < cnt; i += 64) { S.rec[i].key = B.skey[i]; S.rec[i].idx = i; } }
            else { for (int i = lane; i < cnt; i += 64) { X.key[i] = B.skey[i]; X.idx[i] = (uint16_t)i; } }
            WAVE_SYNC();
        };
        stage();
        bool ok = false;
        if constexpr (MW && NW > 1 && LVL > 0) {
            if (cnt > LVL) {   // worth the helpers: partitions level by level over the workgroup's waves, the level-synchronous form per short segment
                if (lane == 0) { B.cmd[1] = B.kind; B.cmd[2] = cnt; B.cmd[3] = GREATER ? 1 : 0; B.cmd[0] = FCMD_SORT; }
                __syncthreads();
                ok = flow_block_sort<EC, NW>(B.S, B.X, B.kind, cnt, GREATER ? 1 : 0, 0, lane);
                __syncthreads();
                if (!ok) stage();
                else { WAVE_SYNC(); return; }
            }
        }
        if (B.kind == 1) {
            const WaveSortPtrs WP = {S.ulist, S.dlist, S.segmark, B.stk};
            if constexpr (LVL > 0) {
                if (!ok && !(MW && NW > 1 && cnt > LVL)) {
                    ok = wave_std_sort_hyb<SRec, Cmp, LVL / 64>(S.rec, cnt, Cmp(), WP, S.tab, S.tab2, lane);
                    if (!ok) stage();
                }
            }
            if (!ok) wave_std_sort<SRec, Cmp, 0>(S.rec, cnt, Cmp(), WP, lane);
        } else {
            const ICmp less{X.key};
            const WaveSortPtrs WP = {X.ulist, X.dlist, X.segmark, B.stk};
            if constexpr (LVL > 0) {
                if (!ok && !(MW && NW > 1 && cnt > LVL)) {
                    ok = wave_std_sort_hyb<uint16_t, ICmp, LVL / 64>(X.idx, cnt, less, WP, X.tab, X.tab2, lane);
                    if (!ok) stage();
                }
            }
            if (!ok) wave_std_sort<uint16_t, ICmp, 0>(X.idx, cnt, less, WP, lane);
        }
        WAVE_SYNC();
    } else {
        for (int i = lane; i < cnt; i += 64) { B.grec[i].key = B.skey[i]; B.grec[i].idx = i; B.grec[i].aux = 0; }
        WAVE_SYNC();
        if (cnt <= 65000) {
            if constexpr (LVL > 0) {
                // partitions in global memory only while a segment is longer than the LDS record form holds; the rest is sorted through in LDS
                FlowSortLds<EC, NW>& S = *B.S;
                const WaveSortPtrs WL = {S.ulist, S.dlist, S.segmark, B.stk};
                wave_std_sort_staged<SRec, Cmp, LVL / 64>(B.grec, cnt, Cmp(), B.gulist, B.gdlist, B.stk, S.rec, FlowSortLds<EC, NW>::BIGL, WL, S.tab, S.tab2, lane);
            } else {
                const WaveSortPtrs WP = {B.gulist, B.gdlist, B.gsegmark, B.stk};
                wave_std_sort<SRec, Cmp, 0>(B.grec, cnt, Cmp(), WP, lane);
            }
        } else {
            if (lane == 0) gs_std_sort<SRec>(B.grec, cnt, Cmp());   // beyond the 16-bit stop lists: one lane
        }
        WAVE_SYNC();
    }
}

// The order of keys that are known to be distinct (ids: the set said so; tags: one per uniform): radix sort instead of the std::sort
// forms; ascending, or descending (DESC); the result is left where the ordinary sort leaves it (B.R(i)).  Returns false when the
// column is sorted the ordinary way instead (records in global memory; too few entries to matter).
template <bool DESC, class BUF>
__device__ __forceinline__ bool flow_sort_distinct(BUF& B, const int32_t* ids, int len0, int lane) {
    if (len0 <= 64) {
        // one key per lane: its position is the number of smaller keys (distinct keys: no ties to break)
        if constexpr (!BUF::SMALL) return false;
        else {
            const int32_t key = lane < len0 ? ids[lane] : 0x7FFFFFFF;
            int rank = 0;
            for (int l = 0; l < len0; ++l) rank += (__builtin_amdgcn_readlane(key, l) < key) ? 1 : 0;
            if (DESC) rank = len0 - 1 - rank;
            if (lane < len0) { B.rec[rank].key = (double)key; B.rec[rank].idx = lane; }
            WAVE_SYNC();
            return true;
        }
    }
    if constexpr (BUF::SMALL) {
        static_assert(sizeof(B.rec) >= (size_t)BUF::CAP * 12 && sizeof(B.f_pos) + sizeof(B.ksel) >= 1024 && offsetof(BUF, ksel) == offsetof(BUF, f_pos) + sizeof(B.f_pos),
                      "radix buffers in the records' LDS, counts in f_pos | ksel (one after the other, both idle until the o_n order)");
        uint32_t* k0 = reinterpret_cast<uint32_t*>(B.rec); uint32_t* k1 = k0 + BUF::CAP;
        uint16_t* i0 = reinterpret_cast<uint16_t*>(k1 + BUF::CAP); uint16_t* i1 = i0 + BUF::CAP;
        int32_t* fin = B.f_dup;
        wave_radix_distinct(ids, len0, lane, k0, k1, i0, i1, reinterpret_cast<uint32_t*>(B.f_pos), [&](int pos, uint32_t, int src) { fin[DESC ? len0 - 1 - pos : pos] = src; });
        WAVE_SYNC();
        for (int i = lane; i < len0; i += 64) { const int src = fin[i]; B.rec[i].key = (double)ids[src]; B.rec[i].idx = src; }
        WAVE_SYNC();
        return true;
    } else {
        if (B.kind == 1) {
            auto& S = *B.S;
            const int n8 = (len0 + 7) & ~7;
            if ((size_t)n8 * 12 > sizeof(S.rec)) return false;
            uint32_t* k0 = reinterpret_cast<uint32_t*>(S.rec); uint32_t* k1 = k0 + n8;
            uint16_t* i0 = reinterpret_cast<uint16_t*>(k1 + n8); uint16_t* i1 = i0 + n8;
            uint16_t* fin = S.ulist;
            wave_radix_distinct(ids, len0, lane, k0, k1, i0, i1, S.tab, [&](int pos, uint32_t, int src) { fin[DESC ? len0 - 1 - pos : pos] = (uint16_t)src; });
            WAVE_SYNC();
            for (int i = lane; i < len0; i += 64) { const int src = fin[i]; S.rec[i].key = (double)ids[src]; S.rec[i].idx = src; }
            WAVE_SYNC();
            return true;
        }
        if (B.kind == 2) {
            auto& X = *B.X;
            const int n8 = (len0 + 7) & ~7;
            if ((size_t)n8 * 12 > (size_t)(reinterpret_cast<char*>(X.dlist) - reinterpret_cast<char*>(&X))) return false;
            uint32_t* k0 = reinterpret_cast<uint32_t*>(&X); uint32_t* k1 = k0 + n8;
            uint16_t* i0 = reinterpret_cast<uint16_t*>(k1 + n8); uint16_t* i1 = i0 + n8;
            uint16_t* fin = X.dlist;
            wave_radix_distinct(ids, len0, lane, k0, k1, i0, i1, X.tab, [&](int pos, uint32_t, int src) { fin[DESC ? len0 - 1 - pos : pos] = (uint16_t)src; });
            WAVE_SYNC();
            for (int i = lane; i < len0; i += 64) { X.key[i] = (double)ids[i]; X.idx[i] = fin[i]; }
            WAVE_SYNC();
            return true;
        }
        {   // records in global memory (beyond the LDS sort forms): the radix buffers take 12 bytes per key, so somewhat longer columns still fit the block
            constexpr size_t AVAIL = FlowSortLds<BUF::EC_, BUF::NW_>::AVAIL;
            const int n8 = (len0 + 7) & ~7;
            if ((size_t)n8 * 12 + 1024 > AVAIL || len0 > 65000) return false;
            char* base = reinterpret_cast<char*>(B.S);
            uint32_t* k0 = reinterpret_cast<uint32_t*>(base); uint32_t* k1 = k0 + n8;
            uint16_t* i0 = reinterpret_cast<uint16_t*>(k1 + n8); uint16_t* i1 = i0 + n8;
            uint32_t* hist = reinterpret_cast<uint32_t*>(base + AVAIL - 1024);
            uint16_t* fin = B.gulist;
            wave_radix_distinct(ids, len0, lane, k0, k1, i0, i1, hist, [&](int pos, uint32_t, int src) { fin[DESC ? len0 - 1 - pos : pos] = (uint16_t)src; });
            WAVE_SYNC();
            for (int i = lane; i < len0; i += 64) { const int src = fin[i]; B.grec[i].key = (double)ids[src]; B.grec[i].idx = src; B.grec[i].aux = 0; }
            WAVE_SYNC();
            return true;
        }
    }
}

// Does an id repeat among B.a_nbr[0 .. len0)?  An open-addressing set in the LDS the first sort is about to use (1 yes, 0 no,
// -1 the column is too long for the set).  Exact: the answer decides whether the list order has to be restored.
template <class BUF>
__device__ __forceinline__ int flow_has_dup(BUF& B, int len0, int lane, int* distinct) {
    int32_t* tab; int cap;
    if constexpr (BUF::SMALL) { tab = reinterpret_cast<int32_t*>(B.rec); cap = (int)(sizeof(B.rec) / 4); }
    else { tab = reinterpret_cast<int32_t*>(B.S->rec); cap = (int)(sizeof(B.S->rec) / 4); }   // (the records' part of the sort overlay)
    int bits = 31 - __builtin_clz((unsigned)cap);        // largest power of two that fits
    if (2 * len0 > (1 << bits)) {
        // too long for the LDS: the set goes into four of the column's scratch arrays that are idle until the merge (b_pos, f_dup, f_pos,
        // ksel: 4 * (ext + 8) words in a row) -- a long column holds the look-back front until its count is known, and a sort costs milliseconds
        if constexpr (BUF::SMALL) return -1;
        else {
            if (B.gtab_words < 2 * len0) return -1;
            tab = B.b_pos;
            bits = 31 - __builtin_clz((unsigned)B.gtab_words);
        }
    }
    while (bits > 6 && (1 << (bits - 1)) >= 2 * len0) --bits;   // no larger than needed: the table is cleared first
    const int size = 1 << bits, mask = size - 1;
    for (int q = lane; q < size; q += 64) tab[q] = -1;
    if constexpr (!BUF::SMALL) DRAIN_STORES();   // (the table may be in global memory: the clearing stores land before the atomics)
    WAVE_SYNC();
    bool dup = false;
    int nnew = 0;
    for (int i = lane; i < len0; i += 64) {
        const int32_t id = B.a_nbr[i];
        uint32_t hpos = ((uint32_t)id * 2654435761u) >> (32 - bits);
        while (true) {
            const int32_t old = atomicCAS(&tab[hpos], -1, id);
            if (old == -1) { ++nnew; break; }
            if (old == id) { dup = true; break; }
            hpos = (hpos + 1) & (uint32_t)mask;
        }
    }
    for (int off = 32; off > 0; off >>= 1) nnew += __shfl_xor(nnew, off);
    *distinct = nnew;   // = the neighbours left after the merge (:646-659)
    const bool any = __ballot(dup) != 0ull;
    WAVE_SYNC();
    return any ? 1 : 0;
}

__device__ __forceinline__ bool flow_abort(const FlowArrays& F) { return ag_ld(&F.ctrl[FC_ABORT]) != 0; }
__device__ __forceinline__ void flow_fail(const FlowArrays& F, int32_t st) { __hip_atomic_fetch_max(&F.ctrl[FC_ABORT], st, RLX, AGT); }
// A wait may last as long as the elimination itself (the last position of a clique waits for all the others), so it is not the
// number of polls that is bounded but the time WITHOUT PROGRESS anywhere: ctrl[4] counts committed positions; a waiter gives up
// (ST_INTERNAL: never a result) when that counter has stood still for `stall_ticks` of the 100 MHz clock.  Called every 64 polls.
struct FlowWatch { int32_t seen; long long t0; };
__device__ __forceinline__ bool flow_stalled(const FlowArrays& F, const FlowParams& P, FlowWatch& w) {
    const int32_t now_p = ag_ld(&F.ctrl[FC_PROGRESS]);
    const long long now_t = wall_clock64();
    if (now_p != w.seen || ag_ld(&F.ctrl[FC_HEAVY]) > 0) { w.seen = now_p; w.t0 = now_t; return false; }   // (a hub of 10^5 entries takes one wave seconds: that is work, not a stall)
    return now_t - w.t0 > (long long)P.spin_limit * 100000ll;   // spin_limit: milliseconds
}

// chunk `c` of column k: its base from the directory, allocated by whoever asks first (others wait for the word)
__device__ __forceinline__ int32_t flow_chunk_base(const Arrays& A, const FlowArrays& F, const FlowParams& P, int32_t k, int c) {
    FlowWatch fw = {-1, 0};
    int32_t* word;
    if (c < FDIR - 1) word = F.cdir + (int64_t)k * FDIR + c;
    else {
        int32_t* ow = F.cdir + (int64_t)k * FDIR + (FDIR - 1);
        int32_t ob = ag_ld(ow);
        if (ob == FD_EMPTY) {
            int32_t seen = FD_EMPTY;
            if (__hip_atomic_compare_exchange_strong(ow, &seen, FD_BUSY, RLX, RLX, AGT)) {
                int32_t base = ag_add(A.pool_top, FDIR_OVF_SLOTS);
                if (base < 0 || base > A.slot_cap - FDIR_OVF_SLOTS) { flow_fail(F, ST_POOL_OVERFLOW); base = FD_FAIL; }
                else {
                    int32_t* o = reinterpret_cast<int32_t*>(A.e + base);
                    for (int q = 0; q < FDIR_OVF; ++q) ag_st(o + q, FD_EMPTY);
                    DRAIN_STORES();
                }
                ag_st(ow, base);
                ob = base;
            } else ob = seen;
        }
        for (int sp = 1; ob == FD_BUSY; ++sp) {
            __builtin_amdgcn_s_sleep(2);
            ob = ag_ld(ow);
            if ((sp & 63) == 0 && flow_stalled(F, P, fw)) { flow_fail(F, ST_INTERNAL); return FD_FAIL; }
        }
        if (ob < 0) return FD_FAIL;
        word = reinterpret_cast<int32_t*>(A.e + ob) + (c - (FDIR - 1));
    }
    int32_t b = ag_ld(word);
    if (b == FD_EMPTY) {
        int32_t seen = FD_EMPTY;
        if (__hip_atomic_compare_exchange_strong(word, &seen, FD_BUSY, RLX, RLX, AGT)) {
            const int32_t need = 1 + chunk_cap(c);
            int32_t base = ag_add(A.pool_top, need);
            if (base < 0 || base > A.slot_cap - need) { flow_fail(F, ST_POOL_OVERFLOW); base = FD_FAIL; }
            ag_st(word, base);
            b = base;
        } else b = seen;
    }
    for (int sp = 1; b == FD_BUSY; ++sp) {
        __builtin_amdgcn_s_sleep(2);
        b = ag_ld(word);
        if ((sp & 63) == 0 && flow_stalled(F, P, fw)) { flow_fail(F, ST_INTERNAL); return FD_FAIL; }
    }
    return b < 0 ? FD_FAIL : b;
}

// The two serial loops of an elimination (:728-779), by ONE lane -- the operation order is the result.  Values are fetched eight
// at a time (the loop's own latency is the floating-point chain, not the LDS).  `v` and `newv` may be the same array.
__device__ __forceinline__ double flow_cumsum(const double* v, double* cum, int m) {
    double csum = 0;
    int j0 = 0;
    for (; j0 + 8 <= m; j0 += 8) {
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = v[j0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { csum += vv[u]; cum[j0 + u] = csum; }
    }
    for (; j0 < m; ++j0) { csum += v[j0]; cum[j0] = csum; }
    return csum;
}
__device__ __forceinline__ void flow_recur(const double* v, double* newv, int m, double csum) {
    double wdeg = csum, colScale = 1;
    int j0 = 0;
    for (; j0 + 8 <= m - 1; j0 += 8) {
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = v[j0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double w = vv[u] * colScale;
            const double f = w / wdeg;
            const double omf = 1 - f;
            newv[j0 + u] = f * omf * wdeg;
            colScale = colScale * omf;
            wdeg = wdeg * omf * omf;
        }
    }
    for (; j0 < m - 1; ++j0) {
        const double w = v[j0] * colScale;
        const double f = w / wdeg;
        const double omf = 1 - f;
        newv[j0] = f * omf * wdeg;
        colScale = colScale * omf;
        wdeg = wdeg * omf * omf;
    }
}
constexpr int FLOW_HELP_MIN = 24;   // shorter columns keep the recurrence on the owning wave (the hand-over is two workgroup barriers)

#define FLOW_JITTER(tag) do { if (P.jitter > 0) { const uint32_t _h = ((uint32_t)idx * 2654435761u + (uint32_t)(tag) * 40503u) >> 9; \
        if ((_h & 3u) == 0u) for (int _q = 0; _q < P.jitter * (int)(1 + ((_h >> 2) & 7u)); ++_q) __builtin_amdgcn_s_sleep(8); } } while (0)

// One position of the order, from the gathered column to the release of the neighbours.  `ext` slots are read (appended + CSR,
// dead ones included); the live ones must fit `cap`.  Returns false when the launch is being abandoned.
template <int NWK, class BUF>
__device__ __forceinline__ bool flow_eliminate(int32_t* const cmd, const int ON, const Arrays& A, const FlowArrays& F, const FlowParams& P, BUF& B, const int32_t cap,
                                               const int32_t idx, const int32_t v, const int32_t cp0, const int32_t cp1, const int32_t acnt,
                                               const int32_t g, const int32_t vbase, const uint64_t gseed, int32_t* last_draws) {
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    const SlotMem M = slot_mem(A);
    long long ft_prev = P.prof ? wall_clock64() : 0;
    const long long ft_begin = ft_prev;
#define FTRACE(k) do { if (P.trace && lane == 0) P.trace[(int64_t)idx * 6 + (k)] = wall_clock64(); } while (0)
#define FSTAMP(k) do { if (P.prof) { const long long _t = wall_clock64(); if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + (BUF::SMALL ? 0 : 20) + (k), (unsigned long long)(_t - ft_prev)); ft_prev = _t; } } while (0)
    // ---- gather (:616-639) in physical order: appended index descending, then the CSR segment descending ----
    int len0 = 0, napp = 0;
    {
        int32_t dirbase = FD_EMPTY;
        if (acnt > 0) {
            const int ct = chunk_of(acnt - 1);
            if (lane <= ct) {
                if (lane < FDIR - 1) dirbase = ag_ld(F.cdir + (int64_t)v * FDIR + lane);
                else { const int32_t ob = ag_ld(F.cdir + (int64_t)v * FDIR + (FDIR - 1)); dirbase = ob >= 0 ? ag_ld(reinterpret_cast<const int32_t*>(A.e + ob) + (lane - (FDIR - 1))) : FD_FAIL; }
            }
        }
        for (int32_t i0 = acnt - 1; i0 >= 0; i0 -= 64) {
            const int32_t i = i0 - lane;
            const bool valid = i >= 0;
            const int c = valid ? chunk_of(i) : 0;
            const int32_t base = __shfl(dirbase, c);
            double val = 0; int32_t nb = 0, tw = 0, tg = 0;
            if (valid && base >= 0) {
                const int32_t s = base + 1 + (i - chunk_start(c));
                const Slot gsl = ag_ld_slot(M, s);
                val = gsl.val; nb = gsl.nbr; tw = gsl.twin; tg = ag_ld(F.atag + s);
            }
            const bool live = valid && val > 0;
            const uint64_t mask = __ballot(live);
            const int pos = len0 + popc64(mask & lt);
            if (live && pos < cap) { B.a_nbr[pos] = nb; B.a_val[pos] = val; B.a_twin[pos] = tw; B.a_tag[pos] = tg; B.b_dup[pos] = i; }   // (b_dup: the appended index, until the merge)
            len0 += popc64(mask);
        }
        napp = len0;
        if (acnt > (1 << 22)) { flow_fail(F, ST_INTERNAL); return false; }   // (the list-order key below holds 22 bits of appended index)
        for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
            const int32_t s = s0 - lane;
            const bool valid = s >= cp0;
            double val = 0; int32_t nb = 0, tw = 0;
            if (valid) { const Slot gsl = ag_ld_slot(M, s); val = gsl.val; nb = gsl.nbr; tw = gsl.twin; }
            const bool live = valid && val > 0;
            const uint64_t mask = __ballot(live);
            const int pos = len0 + popc64(mask & lt);
            if (live && pos < cap) { B.a_nbr[pos] = nb; B.a_val[pos] = val; B.a_twin[pos] = tw; B.a_tag[pos] = -1; }
            len0 += popc64(mask);
        }
    }
    if (len0 > cap) { flow_fail(F, ST_INTERNAL); return false; }
    WAVE_SYNC();
    FSTAMP(1);

    bool published = false;
    // ---- sort by neighbour id (std::sort semantics, :641-644).  Where ids repeat, the list order is part of the result: the appended
    //      entries are first put newest first (descending tag).  Whether they repeat is asked of a hash set, not of a sort ----
    {
        bool need_tag = false, sorted = false;
        int distinct = -1;
        // every position behind this one waits for its count (look-back): the set has it before any sort, so it is asked even where
        // the list order cannot matter (fewer than two appended entries) -- unless nobody waits (mode "frontier")
        const bool stream = A.rng_mode != RNG_COUNTER;
        const bool asked = napp > 1 || !BUF::SMALL || (stream && len0 > 1);
        const int dupk = asked ? flow_has_dup(B, len0, lane, &distinct) : 0;
        if (stream && dupk >= 0 && distinct >= 0) {
            published = true;
            if (lane == 0) ag_st64(F.lb + idx, LB_VALID | (unsigned long long)(distinct > 1 ? distinct - 1 : 0));
            FTRACE(2);
        }
        if (dupk < 0) {   // (too long for the set: sort, look, and sort again if need be)
            for (int i = lane; i < len0; i += 64) B.skey[i] = (double)B.a_nbr[i];
            WAVE_SYNC();
            flow_sort<false>(B, len0, lane);
            sorted = true;
            bool dup = false;
            for (int p = lane + 1; p < len0; p += 64) dup |= (B.R(p).key == B.R(p - 1).key);
            need_tag = __ballot(dup) != 0ull;
        } else need_tag = dupk > 0 && napp > 1;   // (one appended entry or none: the gather order is the list order)
        if (need_tag) {
            // newest first: by the tag, then -- entries one elimination pushed into this column share a tag in mode "frontier" -- by the appended index
            // (stream mode: a tag is the index of one uniform, no two entries share it -- any sort gives the order; the appended entries lead the column)
            if (!(stream && flow_sort_distinct<true>(B, B.a_tag, napp, lane))) {
                for (int i = lane; i < napp; i += 64) B.skey[i] = -((double)B.a_tag[i] * 4194304.0 + (double)B.b_dup[i]);
                WAVE_SYNC();
                flow_sort<false>(B, napp, lane);   // distinct keys
            }
            for (int i = lane; i < napp; i += 64) { const int s = B.R(i).idx; B.b_nbr[i] = B.a_nbr[s]; B.b_twin[i] = B.a_twin[s]; B.b_val[i] = B.a_val[s]; }
            WAVE_SYNC();
            for (int i = lane; i < napp; i += 64) { B.a_nbr[i] = B.b_nbr[i]; B.a_twin[i] = B.b_twin[i]; B.a_val[i] = B.b_val[i]; }
            WAVE_SYNC();
            sorted = false;
        }
        if (!sorted) {
            // no id repeats (the set was asked and said so): the sorted order is unique, any sort gives std::sort's permutation
            const bool nodup = asked && dupk == 0;
            if (!(nodup && flow_sort_distinct<false>(B, B.a_nbr, len0, lane))) {
                for (int i = lane; i < len0; i += 64) B.skey[i] = (double)B.a_nbr[i];
                WAVE_SYNC();
                flow_sort<false>(B, len0, lane);
            }
        }
    }

    FSTAMP(2);
    // ---- merge multi-edges (:646-659): the first of a run keeps its twin and takes the sum (in sorted order), the others' twins die ----
    int m = 0, nk = 0;
    for (int p0 = 0; p0 < len0; p0 += 64) {
        const int p = p0 + lane;
        const bool act = p < len0;
        const int src = act ? B.R(p).idx : 0;
        const int32_t nb = act ? B.a_nbr[src] : -1;
        const int32_t nbprev = (act && p > 0) ? B.a_nbr[B.R(p - 1).idx] : -2;
        const bool head = act && nb != nbprev;
        const uint64_t mask = __ballot(head), kmask = __ballot(act && !head);
        if (head) {
            const int x = m + popc64(mask & lt);
            double val = B.a_val[src];
            int d = 0;
            for (int q = p + 1; q < len0; ++q) { const int sq = B.R(q).idx; if (B.a_nbr[sq] != nb) break; val += B.a_val[sq]; ++d; }
            B.b_nbr[x] = nb; B.b_twin[x] = B.a_twin[src]; B.b_val[x] = val; B.b_dup[x] = d;
            B.b_pos[x] = A.vr[nb].pqpos;   // (positions never change in a launch: a plain load)
        } else if (act) {
            B.a_tag[nk + popc64(kmask & lt)] = B.a_twin[src];
        }
        m += popc64(mask); nk += popc64(kmask);
    }
    const int32_t cdraw = m > 1 ? m - 1 : 0;
    if (lane == 0 && !published && A.rng_mode != RNG_COUNTER) ag_st64(F.lb + idx, LB_VALID | (unsigned long long)cdraw);   // my count: later positions can look past me now
    if (!published) FTRACE(2);
    WAVE_SYNC();
    FSTAMP(3);
    FLOW_JITTER(1);

    // ---- order the neighbours by o_n (:661-673) ----
    if (ON == ON_RANDOM) {
        const uint64_t kb = keyed_order_base(gseed, v - vbase, 0);
        for (int i = lane; i < m; i += 64) B.skey[i] = keyed_order_dkey(kb, B.b_nbr[i] - vbase);
    } else {
        for (int i = lane; i < m; i += 64) B.skey[i] = B.b_val[i];
    }
    WAVE_SYNC();
    if (ON == ON_DESC) flow_sort<true>(B, m, lane); else flow_sort<false>(B, m, lane);
    if constexpr (!BUF::SMALL) {
        // a long column's sorted indices are checked before they are used as addresses: a bad one ends the launch with ST_INTERNAL (the
        // call then runs on the round kernel) instead of a memory fault -- the net under DESIGN.md 8.6
        bool bad = false;
        for (int j = lane; j < m; j += 64) { const int x = B.R(j).idx; bad |= (x < 0 || x >= m); }
        const uint64_t bm = __ballot(bad);
#ifdef RLAP_FLOW_MW_DIAG
        // (experiment builds: the same indices through an index register the compiler cannot have prepared before the calls above)
        bool bad3 = false;
        int lane2 = lane; asm volatile("" : "+v"(lane2));
        for (int j = lane2; j < m; j += 64) { const int x = B.R(j).idx; bad3 |= (x < 0 || x >= m); }
        const uint64_t bm3 = __ballot(bad3);
        if (bm && B.xdbg && lane == 0) { B.xdbg[3] += 1; B.xdbg[0] = 1; B.xdbg[1] = (long long)bm; B.xdbg[2] = (long long)bm3; }
#endif
        if (bm) { flow_fail(F, ST_INTERNAL); return false; }
    }
    for (int j = lane; j < m; j += 64) {
        const int x = B.R(j).idx;
        B.a_nbr[j] = B.b_nbr[x]; B.a_val[j] = B.b_val[x]; B.a_twin[j] = B.b_twin[x]; B.f_dup[j] = B.b_dup[x]; B.f_pos[j] = B.b_pos[x];
    }
    WAVE_SYNC();

    FSTAMP(4);
    // ---- cumulative weights and the f / colScale / wdeg recurrence (:728-779): one lane, the operation order is the result ----
    double* cum = B.cum();
    double* newv = B.newv();
    const double* vsrc = B.a_val;
    bool in_lds = BUF::SMALL;
    char* lbase;
    if constexpr (BUF::SMALL) lbase = reinterpret_cast<char*>(&B);
    else {
        lbase = reinterpret_cast<char*>(B.S);
        constexpr int LCAP = (int)(sizeof(*B.S) / 16);   // the sort's LDS is free now: both loops and the sampling run over it (cumulative sums | values, then new weights)
        if (m <= LCAP) {
            double* lc = reinterpret_cast<double*>(B.S);
            for (int j = lane; j < m; j += 64) lc[LCAP + j] = B.a_val[j];
            WAVE_SYNC();
            cum = lc; newv = lc + LCAP; vsrc = newv; in_lds = true;
        }
    }
    double csum_l0 = 0;
    if (lane == 0) csum_l0 = flow_cumsum(vsrc, cum, m);
    // the recurrence is needed at the commit only: the workgroup's helper wave runs it while this one waits for its uniforms and samples
    bool helped = false;
    if constexpr (NWK > 1) helped = in_lds && m >= FLOW_HELP_MIN;
    if (helped) {
        if (lane == 0) {
            const long long cb = __double_as_longlong(csum_l0);
            cmd[1] = m; cmd[2] = (int32_t)(reinterpret_cast<const char*>(vsrc) - lbase); cmd[3] = (int32_t)(reinterpret_cast<char*>(newv) - lbase);
            cmd[4] = (int32_t)(uint32_t)((unsigned long long)cb & 0xFFFFFFFFull); cmd[5] = (int32_t)(uint32_t)((unsigned long long)cb >> 32);
            cmd[0] = FCMD_RECUR;
        }
        __syncthreads();
    } else if (lane == 0) flow_recur(vsrc, newv, m, csum_l0);
    WAVE_SYNC();

    FSTAMP(5);
    // ---- uniform offset: decoupled look-back over the published counts (the stream is consumed in order, :729) ----
    long long D = 0;
    if (A.rng_mode != RNG_COUNTER) {
        long long sum = 0;
        int32_t hi = idx - 1;   // highest position not yet added
        int spins = 0;
        FlowWatch fw = {-1, 0};
        while (true) {
            const int32_t q = hi - lane;
            unsigned long long wv = LB_VALID | LB_PREFIX;   // below the array: never reached (a sentinel stops the walk)
            if (q >= 0) wv = ag_ld64(F.lb + q);
            const uint64_t pm = __ballot((wv & LB_PREFIX) != 0ull && (wv & LB_VALID) != 0ull);
            const uint64_t vm = __ballot((wv & LB_VALID) != 0ull);
            const int stop = pm ? __builtin_ctzll(pm) : 64;               // nearest inclusive prefix, in lanes
            const uint64_t need = stop >= 63 ? ~0ull : ((2ull << stop) - 1ull);
            if ((vm & need) == need) {
                long long part = (lane <= stop) ? (long long)(wv & LB_MASK) : 0ll;
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                sum += part;
                if (pm) break;
                hi -= 64;
                continue;
            }
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0) {
                if (flow_abort(F)) { if (helped) __syncthreads(); return false; }
                if (flow_stalled(F, P, fw)) { flow_fail(F, ST_INTERNAL); if (helped) __syncthreads(); return false; }
            }
        }
        D = sum;
    }
    if (A.rng_mode != RNG_COUNTER) {
        if (D + cdraw > A.rng_len) { flow_fail(F, ST_RNG_OVERFLOW); if (helped) __syncthreads(); return false; }
        if (lane == 0) ag_st64(F.lb + idx, LB_VALID | LB_PREFIX | (unsigned long long)(D + cdraw));
        *last_draws = (int32_t)(D + cdraw);
    } else {   // mode "frontier": no stream, no offsets, nobody to wait for -- the draws are only counted
        if (lane == 0 && cdraw > 0) atomicAdd(reinterpret_cast<unsigned long long*>(&P.gd[g].n_draws), (unsigned long long)cdraw);
        *last_draws = -1;
    }
    FSTAMP(6);
    FTRACE(3);
    FLOW_JITTER(2);

    // ---- sample k for every position but the last (:747-756) ----
    const double csum = m > 0 ? cum[m - 1] : 0.0;
    for (int j = lane; j < m - 1; j += 64) {
        const double u = (A.rng_mode == RNG_COUNTER) ? frontier_uniform(gseed, (int64_t)(v - vbase), (int64_t)j) : A.rng[D + j];
        const double cj = cum[j];
        const double r = u * (csum - cj) + cj;
        B.ksel[j] = upper_index(cum, m, r);
    }
    WAVE_SYNC();

    if (helped) __syncthreads();   // the helper's new weights are in place
    FSTAMP(7);
    // ---- commit (:766-776): the new entry is appended to column k, the twin rewritten in place; live pairs are counted at the end that
    //      comes later; then my last neighbour's entry and the merged duplicates' twins die (:791-792, :655) ----
    bool failed = false;
    // (a long column's arrays are in global memory: a turn of this loop is three dependent round trips -- operands, the append index,
    // the chunk base.  Four turns go through each step together there, so that their round trips overlap)
    constexpr int CU = BUF::SMALL ? 1 : 4;
    for (int j0 = 0; j0 < m - 1; j0 += 64 * CU) {
        bool act[CU];
        int32_t k[CU], a[CU], s_r[CU], pa[CU], pk[CU], ai[CU], base[CU];
        int c[CU];
        double nw[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int j = j0 + 64 * u + lane;
            act[u] = j < m - 1;
            k[u] = 0; a[u] = 0; s_r[u] = 0; pa[u] = 0; pk[u] = 0; ai[u] = 0; base[u] = FD_FAIL; c[u] = 0; nw[u] = 0;
            if (act[u]) {
                const int32_t ks = B.ksel[j];
                k[u] = B.a_nbr[ks]; pk[u] = B.f_pos[ks];
                a[u] = B.a_nbr[j]; pa[u] = B.f_pos[j]; s_r[u] = B.a_twin[j]; nw[u] = newv[j];
            }
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            if (A.rng_mode != RNG_COUNTER) {
                // mode "exact": the tag (index of the uniform) alone orders what is pushed into a column -- every lane takes its own index
                if (act[u]) { ai[u] = ag_add(&A.vr[k[u]].app_cnt, 1); c[u] = chunk_of(ai[u]); }
            } else {
                // mode "frontier": one elimination's pushes into a column share a tag and are ordered by their append index, so lanes that
                // drew the same target form a group: ONE returning atomic per group, its indices handed out in lane (= position) order
                // (the lanes with my target: one ballot per bit of the id instead of one pass per distinct target)
                uint64_t mymask = __ballot(act[u]);
#pragma unroll
                for (int b = 0; b < 31; ++b) { const bool bit = (k[u] >> b) & 1; const uint64_t bb = __ballot(act[u] && bit); mymask &= bit ? bb : ~bb; }
                if (act[u]) {
                    const int leader = __builtin_ctzll(mymask);
                    int32_t base0 = 0;
                    if (lane == leader) base0 = ag_add(&A.vr[k[u]].app_cnt, popc64(mymask));
                    base0 = __shfl(base0, leader);
                    ai[u] = base0 + popc64(mymask & lt);
                    c[u] = chunk_of(ai[u]);
                }
            }
        }
        FLOW_JITTER(3 + (lane & 3));
#pragma unroll
        for (int u = 0; u < CU; ++u) if (act[u]) base[u] = flow_chunk_base(A, F, P, k[u], c[u]);
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int j = j0 + 64 * u + lane;
            if (act[u] && base[u] >= 0) {
                const int32_t s_n = base[u] + 1 + (ai[u] - chunk_start(c[u]));
                ag_st_slot(M, s_n, nw[u], a[u], s_r[u]);
                ag_st(F.atag + s_n, (A.rng_mode == RNG_COUNTER) ? idx : (int32_t)(D + j));   // increasing in the sequential order either way
                ag_st_slot(M, s_r[u], nw[u], k[u], s_n);
                if (nw[u] > 0) {
                    if (pk[u] < pa[u] && pa[u] != FPOS_NONE) ag_add(&A.vr[a[u]].key, 1);
                    if (pa[u] < pk[u] && pk[u] != FPOS_NONE) ag_add(&A.vr[k[u]].key, 1);
                }
            } else if (act[u]) failed = true;
        }
    }
    if (lane == 0 && m >= 1) ag_st_slot_val(A.e + B.a_twin[m - 1], 0.0);
    for (int i = lane; i < nk; i += 64) ag_st_slot_val(A.e + B.a_tag[i], 0.0);
    FSTAMP(8);
    DRAIN_STORES();   // every store and increment of this wave has arrived before a neighbour is released
    FSTAMP(9);
    FLOW_JITTER(8);
    for (int j = lane; j < m; j += 64) {
        if (B.f_pos[j] != FPOS_NONE) ag_add(&A.vr[B.a_nbr[j]].key, -(1 + B.f_dup[j]));
    }
    if (lane == 0) __hip_atomic_fetch_add(&F.ctrl[FC_PROGRESS], 1, RLX, AGT);   // progress (flow_stalled)
    FTRACE(4);
    if (P.trace && lane == 0) P.trace[(int64_t)idx * 6 + 5] = len0;
    FSTAMP(10);
    if (P.prof && lane == 0 && !BUF::SMALL) { atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 36, 1ull); atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 37, (unsigned long long)len0); if (len0 > 900) atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 38, 1ull); if (len0 > 1888) atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 39, 1ull);
        const unsigned long long dt = (unsigned long long)(wall_clock64() - ft_begin); atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 34, dt); if (len0 > 3320) { atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 32, 1ull); atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 33, dt); } atomicMax(reinterpret_cast<unsigned long long*>(P.prof) + 35, (unsigned long long)len0); }
    if (P.prof && lane == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 16, 1ull); atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 17, (unsigned long long)len0); atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 18, (unsigned long long)(cp1 - cp0 + acnt)); }
#undef FSTAMP
#undef FTRACE
    (void)g;
    return __ballot(failed) == 0ull;
}

// ---------------------------------------------------------------------------
// The persistent kernel: one wave per workgroup; grid = as many as are wanted in flight (they need not all be resident: a
// workgroup that starts late claims later positions, and what a position waits for is always held by a running wave).
// ---------------------------------------------------------------------------
// One claimed position whose pend counter has reached 0: short columns in the workgroup's LDS, long ones in scratch memory.
// Returns false when the launch is being abandoned.
template <int EC, int NW>
__device__ __forceinline__ bool flow_position(FlowLds<EC>& L, int32_t* const s_cmd, const int ON, const Arrays& A, const FlowArrays& F, const FlowParams& P,
                                             const int32_t idx, const int32_t v, const int lane) {
    const int32_t g = F.qg[idx];
    const int32_t vbase = P.gd[g].vbase;
    const uint64_t gseed = A.shuffle_seed + (uint64_t)g;
    const int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
    int32_t acnt = 0;
    if (lane == 0) acnt = ag_ld(&A.vr[v].app_cnt);
    acnt = __builtin_amdgcn_readfirstlane(acnt);
    const int32_t ext = (cp1 - cp0) + acnt;
    int32_t draws = 0;
    bool ok;
    if (ext <= EC) {
        ok = flow_eliminate<NW>(s_cmd, ON, A, F, P, L, EC, idx, v, cp0, cp1, acnt, g, vbase, gseed, &draws);
    } else {
        // long column: the arrays come from the bump allocator; the sort's records, stop lists and tables stay in LDS while they fit
        int32_t b0 = 0;
        const int32_t want = ext + 8;
        if (lane == 0) b0 = ag_add(&F.ctrl[FC_SCR], want);
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (b0 < 0 || b0 > P.scr_entries - want) { if (lane == 0) flow_fail(F, ST_FLOW_SCRATCH); return false; }
        char* base = P.scr + (int64_t)b0 * FLOW_SCR_BYTES;
        FlowBig<EC, NW, RLAP_FLOW_MW_SORT != 0> Gb;
        Gb.cmd = s_cmd; Gb.xdbg = P.prof ? P.prof + 12 : nullptr;
        const int64_t n8 = want;
        double* d = reinterpret_cast<double*>(base);
        Gb.a_val = d; Gb.b_val = d + n8; Gb.skey = d + 2 * n8; Gb.cum_ = d + 3 * n8; Gb.newv_ = d + 4 * n8;
        int32_t* ip = reinterpret_cast<int32_t*>(d + 5 * n8);
        Gb.a_nbr = ip; Gb.a_twin = ip + n8; Gb.a_tag = ip + 2 * n8; Gb.b_nbr = ip + 3 * n8; Gb.b_twin = ip + 4 * n8; Gb.b_dup = ip + 5 * n8;
        Gb.b_pos = ip + 6 * n8; Gb.f_dup = ip + 7 * n8; Gb.f_pos = ip + 8 * n8; Gb.ksel = ip + 9 * n8;
        Gb.gtab_words = (int)(4 * n8);
        char* q = reinterpret_cast<char*>(ip + 10 * n8);   // 80 * n8 bytes in front: 8-byte aligned
        Gb.grec = reinterpret_cast<SRec*>(q);
        Gb.gulist = reinterpret_cast<uint16_t*>(q + 16 * n8);
        Gb.gdlist = Gb.gulist + n8;
        Gb.gsegmark = reinterpret_cast<uint32_t*>(Gb.gdlist + n8);
        Gb.S = reinterpret_cast<FlowSortLds<EC, NW>*>(&L);
        Gb.X = reinterpret_cast<FlowIdxLds<EC, NW>*>(&L);
        Gb.stk = L.stk; Gb.tmp = L.tmp;
        Gb.kind = !FLOW_BIG_IN_LDS ? 0 : (ext <= FlowSortLds<EC, NW>::BIGL ? 1 : (ext <= FlowIdxLds<EC, NW>::IDXL ? 2 : 0));
        const bool heavy = ext > 16384;
        if (heavy && lane == 0) __hip_atomic_fetch_add(&F.ctrl[FC_HEAVY], 1, RLX, AGT);
        ok = flow_eliminate<NW>(s_cmd, ON, A, F, P, Gb, ext, idx, v, cp0, cp1, acnt, g, vbase, gseed, &draws);
        if (heavy && lane == 0) __hip_atomic_fetch_add(&F.ctrl[FC_HEAVY], -1, RLX, AGT);
        if (lane == 0) __hip_atomic_fetch_add(&F.ctrl[FC_LONG], 1, RLX, AGT);
    }
    if (!ok) return false;
    // the last position of a graph files the graph's draw count
    if (lane == 0 && draws >= 0 && (idx + 1 >= F.Q || F.qv[idx + 1] < 0)) P.gd[g].n_draws = draws;
    return true;
}

// ---------------------------------------------------------------------------
// The persistent kernel: one eliminating wave per workgroup (+ helper waves); grid = as many as are wanted in flight (they need
// not all be resident: a workgroup that starts late claims later positions, and what a position waits for is always held by a
// running wave).
// ---------------------------------------------------------------------------
template <int EC, int MINW, int NW>
__global__ __launch_bounds__(64 * NW, MINW) void k_eliminate_flow(Arrays A, FlowArrays F, FlowParams P) {
    __shared__ FlowLds<EC> L;
    __shared__ int32_t s_cmd[8];
    static_assert(sizeof(FlowSortLds<EC, NW>) <= offsetof(FlowLds<EC>, stk) && sizeof(FlowIdxLds<EC, NW>) <= offsetof(FlowLds<EC>, stk), "the long-column sort does not fit the wave's LDS");
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int ON = A.o_n;
    constexpr bool MWS = RLAP_FLOW_MW_SORT != 0;
    if (P.poison >= 0) {
        uint32_t* const w = reinterpret_cast<uint32_t*>(&L);
        const uint32_t pat = 0x01010101u * (uint32_t)(P.poison & 0xFF);
        for (size_t q = threadIdx.x; q < sizeof(L) / 4; q += 64 * NW) w[q] = pat;
        __syncthreads();
    }
    {   // rejected input (the setup kernels' flags are read here, not on the host): nothing is eliminated
        int32_t bad = 0;
        if (P.in_flags[FLAG_RANGE]) bad = ST_INDEX_RANGE;
        else if (P.in_flags[FLAG_CROSS] || P.in_flags[FLAG_PERM]) bad = ST_BAD_ARG;
        else if (P.in_acc[2] != 0.0 || !(P.in_acc[0] <= 1e-24 * P.in_acc[1])) bad = ST_NOT_SYMMETRIC;
        if (bad) { if (threadIdx.x == 0 && blockIdx.x == 0) flow_fail(F, bad); return; }
    }
    if constexpr (NW > 1) {
        // waves 1 .. NW-1 only help wave 0 (the recurrence; the sorts of long columns): they sleep at the workgroup barrier until told
        if (wave > 0) {
            while (true) {
                __syncthreads();
                if (s_cmd[0] == FCMD_EXIT) break;
                if (s_cmd[0] == FCMD_RECUR) {
                    if (wave == 1 && lane == 0) {
                        const unsigned long long cb = (unsigned long long)(uint32_t)s_cmd[4] | ((unsigned long long)(uint32_t)s_cmd[5] << 32);
                        flow_recur(reinterpret_cast<const double*>(reinterpret_cast<char*>(&L) + s_cmd[2]), reinterpret_cast<double*>(reinterpret_cast<char*>(&L) + s_cmd[3]), s_cmd[1], __longlong_as_double((long long)cb));
                    }
                } else if constexpr (MWS)
                flow_block_sort<EC, NW>(reinterpret_cast<FlowSortLds<EC, NW>*>(&L), reinterpret_cast<FlowIdxLds<EC, NW>*>(&L), s_cmd[1], s_cmd[2], s_cmd[3], wave, lane);
                __syncthreads();
            }
            return;
        }
    }
    {
        while (true) {
            long long mt0 = (P.prof || P.trace) ? wall_clock64() : 0;
            int32_t idx = 0;
            if (lane == 0) idx = ag_add(&F.ctrl[FC_CLAIM], 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= F.Qclaim) break;
            if (F.qorder) idx = (int32_t)F.qorder[idx];   // batches: the graphs' positions interleaved (any order that keeps each graph's own is valid)
            const int32_t v = F.qv[idx];
            if (v < 0) continue;   // a graph's sentinel
            if (flow_abort(F)) break;
            // ---- wait until every earlier neighbour has committed ----
            {
                int spins = 0;
                int32_t pend = 1;
                FlowWatch fw = {-1, 0};
                while (true) {
                    if (lane == 0) pend = ag_ld(&A.vr[v].key);
                    pend = __builtin_amdgcn_readfirstlane(pend);
                    if (pend == 0) break;
                    __builtin_amdgcn_s_sleep(2);
                    ++spins;
                    if ((spins & 63) == 0) {
                        if (flow_abort(F)) { pend = -1; break; }
                        if (flow_stalled(F, P, fw)) { flow_fail(F, ST_INTERNAL); pend = -1; break; }
                    }
                }
                if (pend != 0) break;
            }
            if (P.prof && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.prof) + 0, (unsigned long long)(wall_clock64() - mt0));
            if (P.trace && lane == 0) { P.trace[(int64_t)idx * 6 + 0] = mt0 ? mt0 : wall_clock64(); P.trace[(int64_t)idx * 6 + 1] = wall_clock64(); }
            if (!flow_position<EC, NW>(L, s_cmd, ON, A, F, P, idx, v, lane)) break;
        }
    }
    if constexpr (NW > 1) {   // every way out of the loop ends here: the helpers are sent home
        if (lane == 0) s_cmd[0] = FCMD_EXIT;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// setup: positions, queue, sentinels, pending counters
// ---------------------------------------------------------------------------
__global__ void k_flow_queue(const int64_t* __restrict__ perm, const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd, int32_t N,
                             VRec* __restrict__ vr, int32_t* __restrict__ qv, int32_t* __restrict__ qg, unsigned long long* __restrict__ lb,
                             uint64_t* __restrict__ okey, uint32_t* __restrict__ oval, int gbits) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t g = vgraph[i];
    const GraphDesc& D = gd[g];
    const int32_t idx = i - D.vbase;             // place in the node_id vector
    const int64_t p = (int64_t)D.n - 1 - idx;    // pop number (:609 pops from the back)
    int64_t pl = perm[i];
    if (pl < 0 || pl >= (int64_t)D.n) pl = 0;    // (flagged by k_perm_check: nothing is eliminated then)
    const int32_t v = D.vbase + (int32_t)pl;
    if (p < D.n_elim) {
        const int32_t li = D.flow_base + 1 + (int32_t)p;
        qv[li] = v; qg[li] = g; lb[li] = 0ull;
        vr[v].pqpos = li;
        if (okey) { const int32_t pi = D.flow_base - g + (int32_t)p; okey[pi] = ((uint64_t)p << gbits) | (uint64_t)(uint32_t)g; oval[pi] = (uint32_t)li; }   // claim order: by position, then graph
    }
    if (idx == 0) { qv[D.flow_base] = -1; qg[D.flow_base] = g; lb[D.flow_base] = LB_VALID | LB_PREFIX; }
}
__global__ void k_flow_qinit(int32_t* __restrict__ qv, int32_t* __restrict__ qg, unsigned long long* __restrict__ lb, int32_t Q) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) { qv[i] = -1; qg[i] = 0; lb[i] = LB_VALID | LB_PREFIX; }   // every word starts as a sentinel (prefix 0)
}
__global__ void k_flow_vinit(VRec* __restrict__ vr, int32_t* __restrict__ cdir, int32_t N) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    vr[i].key = 0; vr[i].pqpos = FPOS_NONE;
    int4* d = reinterpret_cast<int4*>(cdir + (int64_t)i * FDIR);
    d[0] = make_int4(FD_EMPTY, FD_EMPTY, FD_EMPTY, FD_EMPTY); d[1] = make_int4(FD_EMPTY, FD_EMPTY, FD_EMPTY, FD_EMPTY);
    static_assert(FDIR == 8, "two 16-byte stores per directory");
}
// pend[a] = live entries (u -> a) in columns of vertices u that are eliminated before a
__global__ void k_flow_pending(const Slot* __restrict__ ent, const int32_t* __restrict__ slot_col, const int32_t* __restrict__ nnz_p, VRec* __restrict__ vr) {
    const int32_t nnz = *nnz_p;
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nnz; s += gridDim.x * blockDim.x) {
        const Slot e = ent[s];
        if (!(e.val > 0)) continue;
        const int32_t pu = vr[slot_col[s]].pqpos, pa = vr[e.nbr].pqpos;
        if (pu < pa && pa != FPOS_NONE) atomicAdd(&vr[e.nbr].key, 1);
    }
}
// afterwards: header links + VRec::app_chunk (the layout the output pass reads), per-graph status
__global__ void k_flow_finish(Arrays A, FlowArrays F, GraphDesc* __restrict__ gd, int32_t N, int32_t G) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t st = F.ctrl[FC_ABORT];
    // an abandoned launch leaves append counts ahead of their chunks (directory words EMPTY, BUSY or FAIL): no chain may be built
    // from them, and nothing after this kernel may follow one -- the columns are cut back to their CSR part; the call is retried
    // or fails with `st`, its rows are never handed out
    if (i < N) { if (st == 0) flow_finish_vertex(A, F, i); else { A.vr[i].app_cnt = 0; A.vr[i].app_chunk = -1; } }
    if (i < G) {
        if (st) gd[i].status = st;
        gd[i].pad0 = 0; gd[i].pad1 = (i == 0) ? F.ctrl[FC_LONG] : 0;   // ("rounds" has no meaning here; "singles" = long columns)
    }
}

// ---------------------------------------------------------------------------
// tag order of the surviving columns (rlap_flow.h): concurrent eliminations push into a column out of order; the output pass
// (:789-810 with compressColumnSC :678-711) reads the list newest first.  Columns whose appended entries are not in tag order are
// sorted: one (column, tag) radix sort over their entries, then a permutation through a copy.  (Twins are not kept: nothing
// reads them after the elimination.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t flow_slot_plain(const Arrays& A, const FlowArrays& F, int32_t v, int32_t i) {
    const int c = chunk_of(i);
    const int32_t base = c < FDIR - 1 ? F.cdir[(int64_t)v * FDIR + c] : reinterpret_cast<const int32_t*>(A.e + F.cdir[(int64_t)v * FDIR + FDIR - 1])[c - (FDIR - 1)];
    return base + 1 + (i - chunk_start(c));
}
__global__ void k_flow_ro_count(Arrays A, FlowArrays F, const uint32_t* __restrict__ order, int32_t S, int32_t* __restrict__ cnt) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > S) return;
    if (i == S) { cnt[S] = 0; return; }
    const int32_t v = (int32_t)order[i];
    const int32_t acnt = A.vr[v].app_cnt;
    int32_t c = 0;
    if (acnt >= 2) {
        c = acnt;
        if (acnt <= 48) {   // short: look (most columns are in order already)
            bool sorted = true;
            int32_t prev = F.atag[flow_slot_plain(A, F, v, 0)];
            for (int32_t q = 1; q < acnt; ++q) { const int32_t t = F.atag[flow_slot_plain(A, F, v, q)]; sorted &= t >= prev; prev = t; }
            if (sorted) c = 0;
        }
    }
    cnt[i] = c;
}
__global__ void k_flow_ro_emit(Arrays A, FlowArrays F, const uint32_t* __restrict__ order, int32_t S, const int32_t* __restrict__ off,
                               uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ home, int32_t cap, int32_t* __restrict__ status) {
    const int32_t M = off[S];
    if (M > cap) { if (blockIdx.x == 0 && threadIdx.x == 0 && *status == 0) *status = ST_FLOW_REORDER; return; }
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < cap; r += gridDim.x * blockDim.x) {
        if (r >= M) { keys[r] = ~0ull; vals[r] = 0u; continue; }   // (the sort runs over the whole buffer: its length is a host-side argument)
        int32_t lo = 0, hi = S;   // last i with off[i] <= r
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (off[mid] <= r) lo = mid; else hi = mid; }
        const int32_t v = (int32_t)order[lo];
        const int32_t s = flow_slot_plain(A, F, v, r - off[lo]);
        keys[r] = ((uint64_t)(uint32_t)lo << 32) | (uint32_t)F.atag[s];
        vals[r] = (uint32_t)s; home[r] = (uint32_t)s;
    }
}
__global__ void k_flow_ro_fetch(Arrays A, const int32_t* __restrict__ off, int32_t S, const uint32_t* __restrict__ sorted_slot, int32_t cap, Slot* __restrict__ tmp) {
    const int32_t M = off[S];
    if (M > cap) return;
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < M; r += gridDim.x * blockDim.x) tmp[r] = A.e[sorted_slot[r]];
}
__global__ void k_flow_ro_store(Arrays A, FlowArrays F, const int32_t* __restrict__ off, int32_t S, const uint32_t* __restrict__ home, const uint64_t* __restrict__ sorted_keys,
                                int32_t cap, const Slot* __restrict__ tmp) {
    const int32_t M = off[S];
    if (M > cap) return;
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < M; r += gridDim.x * blockDim.x) { A.e[home[r]] = tmp[r]; F.atag[home[r]] = (int32_t)(uint32_t)(sorted_keys[r] & 0xFFFFFFFFull); }
}

// test hook (rlap_debug_wave_sort, desc bit 5): the long-column sort of the dataflow kernel -- one workgroup sorts one array of
// doubles, records in LDS (level-synchronous / hybrid / partition forms by length) or, with desc bit 6 or beyond the LDS block, in
// global scratch; returns the permutation.  Compared with std::sort itself by tests/test_gpu_flow.py.
__global__ __launch_bounds__(64 * FLOW_NW) void k_debug_flow_sort(const double* __restrict__ keys, const int32_t* __restrict__ offs, int32_t narr, int32_t desc,
                                                                  int32_t* __restrict__ perm_out, char* __restrict__ scr) {
    __shared__ FlowLds<FLOW_EC> L;
    __shared__ int32_t s_cmd[8];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (wave > 0) {   // helpers, as in k_eliminate_flow
        while (true) {
            __syncthreads();
            if (s_cmd[0] == FCMD_EXIT) break;
            flow_block_sort<FLOW_EC, FLOW_NW>(reinterpret_cast<FlowSortLds<FLOW_EC, FLOW_NW>*>(&L), reinterpret_cast<FlowIdxLds<FLOW_EC, FLOW_NW>*>(&L), s_cmd[1], s_cmd[2], s_cmd[3], wave, lane);
            __syncthreads();
        }
        return;
    }
    for (int32_t arr = blockIdx.x; arr < narr; arr += gridDim.x) {
        const int32_t o = offs[arr], n = offs[arr + 1] - o;
        FlowBig<FLOW_EC, FLOW_NW, true> B;
        B.S = reinterpret_cast<FlowSortLds<FLOW_EC, FLOW_NW>*>(&L); B.X = reinterpret_cast<FlowIdxLds<FLOW_EC, FLOW_NW>*>(&L); B.stk = L.stk; B.tmp = L.tmp; B.cmd = s_cmd; B.xdbg = nullptr; B.gtab_words = 0;
        B.kind = (desc & 64) ? 0 : ((desc & 128) ? (n <= FlowIdxLds<FLOW_EC, FLOW_NW>::IDXL ? 2 : 0) : (n <= FlowSortLds<FLOW_EC, FLOW_NW>::BIGL ? 1 : (n <= FlowIdxLds<FLOW_EC, FLOW_NW>::IDXL ? 2 : 0)));
        B.skey = const_cast<double*>(keys) + o;
        char* base = scr + 32 * (int64_t)o + 64 * (int64_t)arr;
        B.grec = reinterpret_cast<SRec*>(base);
        B.gulist = reinterpret_cast<uint16_t*>(base + 16 * (int64_t)n);
        B.gdlist = B.gulist + (n + 2);
        B.gsegmark = reinterpret_cast<uint32_t*>(base + 20 * (int64_t)n + 8);
        const long long t0 = wall_clock64();
        bool done = false;
        if (desc & 512) {   // the keys are distinct non-negative integers: the radix form the elimination uses for ids without repeats and for tags
            int32_t* ids = reinterpret_cast<int32_t*>(base + 24 * (int64_t)n + 16);
            for (int q = lane; q < n; q += 64) ids[q] = (int32_t)B.skey[q];
            WAVE_SYNC();
            done = (desc & 1) ? flow_sort_distinct<true>(B, ids, n, lane) : flow_sort_distinct<false>(B, ids, n, lane);
        }
        if (!done) { if (desc & 1) flow_sort<true>(B, n, lane); else flow_sort<false>(B, n, lane); }
        const long long t1 = wall_clock64();
        for (int q = lane; q < n; q += 64) perm_out[o + q] = B.R(q).idx;
        if ((desc & 256) && lane == 0 && n > 0) perm_out[o] = (int32_t)(t1 - t0);   // diagnostic: the sort's duration in 10 ns ticks instead of the first index
        WAVE_SYNC();
    }
    if (lane == 0) s_cmd[0] = FCMD_EXIT;
    __syncthreads();
}
void launch_debug_flow_sort(hipStream_t s, const double* keys, const int32_t* offs, int32_t narr, int32_t desc, int32_t* perm_out, char* scr) {
    hipLaunchKernelGGL(k_debug_flow_sort, dim3((unsigned)std::min<int32_t>(narr, 512)), dim3(64 * FLOW_NW), 0, s, keys, offs, narr, desc, perm_out, scr);
}

void launch_flow_setup(hipStream_t s, const Arrays& A, const FlowArrays& F, const FlowParams& P, const int32_t* slot_col, const int32_t* nnz_p,
                       int32_t N, int64_t Eeff, uint64_t* okey, uint32_t* oval, int gbits) {
    const unsigned nb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_flow_qinit, dim3((unsigned)((F.Q + 255) / 256)), dim3(256), 0, s, const_cast<int32_t*>(F.qv), const_cast<int32_t*>(F.qg), F.lb, F.Q);
    hipLaunchKernelGGL(k_flow_vinit, dim3(nb), dim3(256), 0, s, A.vr, F.cdir, N);
    hipLaunchKernelGGL(k_flow_queue, dim3(nb), dim3(256), 0, s, A.perm, P.vgraph, P.gd, N, A.vr, const_cast<int32_t*>(F.qv), const_cast<int32_t*>(F.qg), F.lb, okey, oval, gbits);
    if (Eeff > 0) {
        const unsigned ne = (unsigned)std::min<int64_t>((Eeff + 255) / 256, 8192);
        hipLaunchKernelGGL(k_flow_pending, dim3(ne), dim3(256), 0, s, A.e, slot_col, nnz_p, A.vr);
    }
}

// two shapes: one wave with a large LDS block, two workgroups per CU (a single large graph: what bounds the run is how fast a long
// column is dealt with); or a small block, eight per CU (a batch of small graphs: what bounds the run is how many positions are in flight)
void launch_flow_eliminate(int shape, unsigned grid, hipStream_t stream, const Arrays& A, const FlowArrays& F, const FlowParams& P) {
    if (shape == 2) hipLaunchKernelGGL((k_eliminate_flow<FLOW_EC_SMALL, 2, 1>), dim3(grid), dim3(64), 0, stream, A, F, P);
    else if (shape == 3) hipLaunchKernelGGL((k_eliminate_flow<FLOW_EC_MID, 1, 1>), dim3(grid), dim3(64), 0, stream, A, F, P);
    else if (shape == 4) hipLaunchKernelGGL((k_eliminate_flow<FLOW_EC, 1, 1>), dim3(grid), dim3(64), 0, stream, A, F, P);
    else hipLaunchKernelGGL((k_eliminate_flow<FLOW_EC, 1, FLOW_NW>), dim3(grid), dim3(64 * FLOW_NW), 0, stream, A, F, P);
}

void launch_flow_finish(hipStream_t s, const Arrays& A, const FlowArrays& F, GraphDesc* gd, int32_t N, int32_t G) {
    const int32_t n = N > G ? N : G;
    hipLaunchKernelGGL(k_flow_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, F, gd, N, G);
}

void launch_flow_ro_count(hipStream_t s, const Arrays& A, const FlowArrays& F, const uint32_t* order, int32_t S, int32_t* cnt) {
    hipLaunchKernelGGL(k_flow_ro_count, dim3((unsigned)((S + 1 + 255) / 256)), dim3(256), 0, s, A, F, order, S, cnt);
}
void launch_flow_ro_emit(hipStream_t s, const Arrays& A, const FlowArrays& F, const uint32_t* order, int32_t S, const int32_t* off, uint64_t* keys, uint32_t* vals,
                         uint32_t* home, int32_t cap, int32_t* status) {
    hipLaunchKernelGGL(k_flow_ro_emit, dim3(2048), dim3(256), 0, s, A, F, order, S, off, keys, vals, home, cap, status);
}
void launch_flow_ro_permute(hipStream_t s, const Arrays& A, const FlowArrays& F, const int32_t* off, int32_t S, const uint32_t* sorted_slot, const uint64_t* sorted_keys,
                            const uint32_t* home, int32_t cap, Slot* tmp) {
    hipLaunchKernelGGL(k_flow_ro_fetch, dim3(2048), dim3(256), 0, s, A, off, S, sorted_slot, cap, tmp);
    hipLaunchKernelGGL(k_flow_ro_store, dim3(2048), dim3(256), 0, s, A, F, off, S, home, sorted_keys, cap, tmp);
}

}  // namespace rlap
