// rlap_core.h -- data-structure logic shared by the HIP kernels (device) and the
// host-side mirror used by the CPU unit tests (tests/csrc/host_mirror.cc).
// Everything here is plain sequential C++ on caller-provided arrays; the
// wave-parallel parts live in the .hip files.
//
// Reference behaviour restated (paths relative to /root/reference):
//   rlap/csrc/preconditioner.cc:125-246  bucket PQ (LIFO per bucket)  -> lazy bucket stacks
//   rlap/csrc/preconditioner.cc:273-345  std::sort calls               -> std_sort_emul()
//   rlap/csrc/preconditioner.cc:65-114,404-414 linked column lists     -> CSR segment + chunks
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RLAP_HD __host__ __device__ __forceinline__
#define RLAP_HD_NOINLINE __host__ __device__ __noinline__
#else
#define RLAP_HD inline
#define RLAP_HD_NOINLINE inline
#endif

namespace rlap {

enum { OV_RANDOM = 0, OV_DEGREE = 1, OV_COARSEN = 2 };
enum { ON_ASC = 0, ON_DESC = 1, ON_RANDOM = 2 };

// status codes returned through the C ABI (include/rlap_hip.h)
enum {
    ST_OK = 0,
    ST_NOT_SYMMETRIC = 1,
    ST_INDEX_RANGE = 2,
    ST_BAD_ARG = 3,
    ST_POOL_OVERFLOW = 4,
    ST_LOG_OVERFLOW = 5,
    ST_RNG_OVERFLOW = 6,
    ST_OUT_OVERFLOW = 7,
    ST_HIP_ERROR = 8,
    ST_TOO_LARGE = 9,
    ST_INTERNAL = 10,
    ST_FLOW_SCRATCH = 12,   // dataflow elimination: working storage of the long columns exhausted (the call repeats with more)
    ST_FLOW_REORDER = 13,   // ... more out-of-order appended entries than the reorder buffers hold (the call repeats with the round kernel)
};

// ---------------------------------------------------------------------------
// Keyed neighbour order: replaces std::shuffle(std::random_device) of
// preconditioner.cc:303-307,340-342 by an injected, reproducible order
// (ascending (key, nbr)).  Same definition in oracle/rlap_oracle.cc.
// `vertex` and `nbr` are ids LOCAL to their graph (global id - GraphDesc::vbase) and graph g of a
// batch uses the seed `shuffle_seed + g`, so that a batched call returns what separate calls on the
// single graphs return (include/rlap_hip.h).
// ---------------------------------------------------------------------------
RLAP_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RLAP_HD uint64_t keyed_order_base(uint64_t seed, int64_t vertex, int phase) {
    return mix64(seed ^ mix64(2ull * (uint64_t)vertex + (uint64_t)phase));
}
RLAP_HD uint64_t keyed_order_key(uint64_t base, int64_t nbr) { return mix64(base ^ mix64((uint64_t)nbr)); }

// ---------------------------------------------------------------------------
// mode = "frontier" (SURVEY section 7 step 7, H2): the j-th uniform an elimination of `vertex` draws is a function of
// (seed, vertex, j) instead of the next output of the ONE MT19937-64 stream the reference consumes in elimination order
// (preconditioner.cc:356-357).  Same distribution, no order in which the draws must be made; bit-exact against the oracle run
// in the same mode, NOT against the reference.  53 random bits -> [0, 1).  Ids are LOCAL to the graph, the seed is the graph's.
// ---------------------------------------------------------------------------
enum { RNG_STREAM = 0, RNG_COUNTER = 1 };
RLAP_HD double frontier_uniform(uint64_t seed, int64_t vertex, int64_t j) {
    const uint64_t z = mix64(mix64((seed ^ 0x66726F6E74696572ull) ^ mix64((uint64_t)vertex)) ^ mix64((uint64_t)j));
    return (double)(z >> 11) * 1.1102230246251565e-16;   // 2^-53
}

// ---------------------------------------------------------------------------
// Appended-entry chunks.  Column v = CSR segment [colptr[v], colptr[v+1]) plus
// `app_cnt[v]` appended entries kept in geometrically growing chunks:
// chunk c holds appended indices [CHUNK0*(2^c-1), CHUNK0*(2^(c+1)-1)).
// A chunk occupies (1 + capacity) slots: slot `base` is a header whose e_nbr
// field links to the previous (older) chunk's base, entries follow.
// Reference traversal order (head = newest push, then CSC order reversed,
// preconditioner.cc:84-97,409-411) = appended indices descending, then the CSR
// segment descending.
// ---------------------------------------------------------------------------
constexpr int CHUNK0 = 8;
RLAP_HD int ilog2_u32(unsigned q) {  // floor(log2(q)), q >= 1
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)q);
#else
    int c = 0;
    while (q > 1u) { q >>= 1; ++c; }
    return c;
#endif
}
RLAP_HD int chunk_of(int a) {  // chunk index of appended index a
    return ilog2_u32((unsigned)a / CHUNK0 + 1u);
}
RLAP_HD int chunk_start(int c) { return CHUNK0 * ((1 << c) - 1); }
RLAP_HD int chunk_cap(int c) { return CHUNK0 << c; }

// ---------------------------------------------------------------------------
// Bucket index of a PQ key: preconditioner.cc:222-223,238-239
// ---------------------------------------------------------------------------
RLAP_HD int pq_list_of(int key, int n) { return key <= n ? key : n + key / n; }

// One PQ op on a target, tracked as (key, mv): `mv` = number of the last op
// that changed the target's bucket (a move re-inserts it at the bucket head,
// preconditioner.cc:192-214), -1 if none so far.  delta=+1: DegreePQInc
// (:235-246); delta=-1: DegreePQDec (:216-233, a no-op at key 1).
RLAP_HD void pq_op(int& key, int& mv, int n, int delta, int opnum) {
    if (delta < 0 && key == 1) return;
    int ol = pq_list_of(key, n), nl = pq_list_of(key + delta, n);
    key += delta;
    if (ol != nl) mv = opnum;
}

// ---------------------------------------------------------------------------
// Sort records and an exact re-statement of libstdc++ 11 std::sort
// (/usr/include/c++/11/bits/stl_algo.h:1810-1960, stl_heap.h:128-430):
// introsort loop (median-of-3 to first, unguarded Hoare partition, threshold
// 16, depth limit 2*floor(lg n), heap-sort fallback) + final insertion sort.
// The reference sorts POINTERS with a strict-weak comparator on one field, so
// the resulting permutation under ties is a function of this algorithm only.
// ---------------------------------------------------------------------------
struct SRec {
    double key;
    int32_t idx;
    int32_t aux;
};

template <bool GREATER>
RLAP_HD bool srec_less(const SRec& a, const SRec& b) {
    return GREATER ? (a.key > b.key) : (a.key < b.key);
}

// Generic form: T = element type, less(const T&, const T&) strict weak order.
template <class T, class Less, class P>
RLAP_HD void gs_adjust_heap(P a, int first, int hole, int len, T value, Less less) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && less(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <class T, class Less, class P>
RLAP_HD void gs_heap_sort(P a, int first, int last, Less less) {  // __partial_sort(first,last,last)
    int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            T v = a[first + parent];
            gs_adjust_heap<T>(a, first, parent, len, v, less);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {
        --l;
        T v = a[l];
        a[l] = a[first];
        gs_adjust_heap<T>(a, first, 0, l - first, v, less);
    }
}

// introsort phase only (everything before __final_insertion_sort)
template <class T, class Less, class P>
RLAP_HD void gs_introsort_loop(P a, int n, Less less) {
    if (n <= 16) return;
    int depth0 = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
    depth0 *= 2;
    // explicit stack replaces the recursion on the right part
    int stk_first[72], stk_last[72], stk_depth[72];
    int sp = 0;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = depth0; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { gs_heap_sort<T>(a, first, last, less); break; }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1)
            int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
            int pick;
            if (less(a[ia], a[ib])) {
                if (less(a[ib], a[ic])) pick = ib;
                else if (less(a[ia], a[ic])) pick = ic;
                else pick = ia;
            } else if (less(a[ia], a[ic])) pick = ia;
            else if (less(a[ib], a[ic])) pick = ic;
            else pick = ib;
            { T t = a[first]; a[first] = a[pick]; a[pick] = t; }
            // __unguarded_partition(first+1, last, pivot=first)
            T pv = a[first];
            int f = first + 1, l = last;
            while (true) {
                while (less(a[f], pv)) ++f;
                --l;
                while (less(pv, a[l])) --l;
                if (!(f < l)) break;
                T t = a[f]; a[f] = a[l]; a[l] = t;
                ++f;
            }
            int cut = f;
            // recurse on [cut,last), continue with [first,cut)
            stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth; ++sp;
            last = cut;
        }
    }
}

// Same loop for n <= 32: after a partition at most one side can still exceed the
// threshold of 16, so no stack is needed (nothing lands in scratch memory).
template <class T, class Less, class P>
RLAP_HD void gs_introsort_loop_small(P a, int n, Less less) {
    if (n <= 16) return;
    int depth = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth;
    depth *= 2;
    int first = 0, last = n;
    while (last - first > 16) {
        if (depth == 0) { gs_heap_sort<T>(a, first, last, less); break; }
        --depth;
        int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
        int pick;
        if (less(a[ia], a[ib])) {
            if (less(a[ib], a[ic])) pick = ib;
            else if (less(a[ia], a[ic])) pick = ic;
            else pick = ia;
        } else if (less(a[ia], a[ic])) pick = ia;
        else if (less(a[ib], a[ic])) pick = ic;
        else pick = ib;
        { T t = a[first]; a[first] = a[pick]; a[pick] = t; }
        T pv = a[first];
        int f = first + 1, l = last;
        while (true) {
            while (less(a[f], pv)) ++f;
            --l;
            while (less(pv, a[l])) --l;
            if (!(f < l)) break;
            T t = a[f]; a[f] = a[l]; a[l] = t;
            ++f;
        }
        if (last - f > 16) first = f; else last = f;   // the other side is <= 16 and needs no more partitioning
    }
}
// __final_insertion_sort == stable insertion sort of the whole range (the
// unguarded part never runs past the sentinel left by the partitions).
template <class T, class Less, class P>
RLAP_HD void gs_insertion_sort(P a, int n, Less less) {
    for (int i = 1; i < n; ++i) {
        T v = a[i];
        int j = i - 1;
        while (j >= 0 && less(v, a[j])) { a[j + 1] = a[j]; --j; }
        a[j + 1] = v;
    }
}

template <class T, class Less, class P>
RLAP_HD void gs_std_sort_small(P a, int n, Less less) {   // n <= 32
    if (n < 2) return;
    gs_introsort_loop_small<T>(a, n, less);
    gs_insertion_sort<T>(a, n, less);
}

template <class T, class Less, class P>
RLAP_HD void gs_std_sort(P a, int n, Less less) {
    if (n < 2) return;
    gs_introsort_loop<T>(a, n, less);
    gs_insertion_sort<T>(a, n, less);
}

template <bool GREATER>
struct SRecLess {
    RLAP_HD bool operator()(const SRec& x, const SRec& y) const { return GREATER ? (x.key > y.key) : (x.key < y.key); }
};

template <bool GREATER, class P>
RLAP_HD void ss_heap_sort(P a, int first, int last) { gs_heap_sort<SRec>(a, first, last, SRecLess<GREATER>()); }

template <bool GREATER, class P>
RLAP_HD void std_sort_emul(P a, int n) { gs_std_sort<SRec>(a, n, SRecLess<GREATER>()); }

// Key of the injected "random" neighbour order: a 52-bit hash, exact in a double.
// The order itself is std::sort (emulated) ascending on this key, applied to the
// neighbours in post-merge (ascending id) order.
RLAP_HD double keyed_order_dkey(uint64_t base, int64_t nbr) { return (double)(keyed_order_key(base, nbr) >> 12); }

// ---------------------------------------------------------------------------
// Device-resident state (SoA, int32 indices, f64 weights).
// ---------------------------------------------------------------------------
struct GraphDesc {      // one per graph of a batch
    int32_t vbase;      // global id of local vertex 0
    int32_t n;          // vertices
    int64_t t;          // num_remove
    int32_t bucket_base;  // into bs_cnt/bs_alloc/bs_dir/ocur/oend, 2n+1 buckets
    int32_t pad0;
    int32_t pad1;
    int32_t scr_base;   // big-column scratch, in entries
    int32_t scr_cap;
    // running state / results
    int32_t minlist;
    int32_t push_cnt;   // ids handed to PQ pushes so far (pqpos values)
    int32_t n_elim;
    int32_t status;
    int64_t n_draws;
    int64_t out_rows;   // rows this graph emits (filled by the output pass)
    int32_t flow_base;  // dataflow elimination (rlap_flow.h): look-back index of this graph's sentinel; its positions follow
    int32_t pad2;
};

struct ColBuf {         // working storage for one column (LDS or global scratch)
    SRec* rec;
    int32_t* a_slot; int32_t* a_nbr; int32_t* a_twin; double* a_val;
    int32_t* b_slot; int32_t* b_nbr; int32_t* b_twin; double* b_val;
    double* cum; double* newv;
    int32_t* ksel; int32_t* t_key; int32_t* t_mv; int32_t* t_of;  // t_of[j] = b-index of final position j
};

// One directed entry = one 16-byte record: a candidate's gather is one load per slot and a rewire one store per slot
// (round 1 kept three arrays: three 64-byte lines touched per slot).
struct alignas(16) Slot {
    double val;     // weight; <= 0: dead (never unlinked, like the reference's nodes)
    int32_t nbr;    // neighbour (global id); in a chunk header: base of the previous chunk
    int32_t twin;   // slot of the transposed entry
};
// Per-vertex state in one 16-byte record: what a round reads about a candidate or a sampled target sits in one line.
struct alignas(16) VRec {
    int32_t key;        // PQ key (degree proxy, preconditioner.cc:125-246)
    int32_t pqpos;      // -1 = original place, >= 0 id of its latest push, -2 popped
    int32_t app_cnt;    // appended entries
    int32_t app_chunk;  // base of the newest chunk (-1: none)
};
struct Arrays {
    const int32_t* colptr;   // [N+1]
    Slot* e;                 // [slot_cap]; slots [0,nnz) = CSR, rest = chunk pool
    int32_t slot_cap;
    int32_t* pool_top;       // next free slot (device counter)
    VRec* vr;                // [N]
    int32_t* ocur; int32_t* oend;                    // [sum(2n+1)] never-moved members of each bucket
    // per-bucket LIFO stacks of moved vertices (geometric chunks, per-bucket directory)
    int32_t* bs_cnt; int32_t* bs_alloc;              // [sum(2n+1)] height, chunks allocated
    int32_t* bs_dir;                                 // [sum(2n+1) * BDIR] chunk base in the bs pool
    int32_t* bs_v; int32_t* bs_id;                   // [bs_pool_cap] entry = (vertex, push id)
    int32_t* bs_pool_top; int32_t bs_pool_cap;
    const int32_t* orig_order;                       // [N] global ids by (graph, deg asc, id desc)
    const double* rng; int64_t rng_len;
    const int64_t* perm;     // [N] local ids (o_v random), graph g at [vbase, vbase+n)
    int32_t o_v; int32_t o_n;
    uint64_t shuffle_seed;
    int32_t rng_mode;        // RNG_STREAM: uniforms from `rng` in elimination order (the reference); RNG_COUNTER: frontier_uniform(shuffle_seed, vertex, j)
    int32_t vbase;           // global id of the graph's local vertex 0 (the kernels set it per graph)
};
// the j-th uniform of the elimination of `vertex` (global id), draw number `draw` of the graph's stream
RLAP_HD double arr_uniform(const Arrays& A, int64_t draw, int32_t vertex, int32_t j) {
    return A.rng_mode == RNG_COUNTER ? frontier_uniform(A.shuffle_seed, (int64_t)(vertex - A.vbase), (int64_t)j) : A.rng[draw];
}

#if defined(__HIP_DEVICE_COMPILE__)
RLAP_HD int32_t pool_take(int32_t* top, int32_t cnt) { return atomicAdd(top, cnt); }
#else
RLAP_HD int32_t pool_take(int32_t* top, int32_t cnt) { int32_t r = *top; *top = r + cnt; return r; }
#endif

// Append one entry to column k; returns its slot or -1 on pool overflow.
RLAP_HD int32_t col_append(const Arrays& A, int32_t k) {
    int32_t a = A.vr[k].app_cnt;
    int c = chunk_of(a);
    int cs = chunk_start(c);
    if (a == cs) {
        int32_t need = 1 + chunk_cap(c);
        int32_t base = pool_take(A.pool_top, need);
        if (base < 0 || base > A.slot_cap - need) return -1;
        A.e[base].nbr = A.vr[k].app_chunk;
        A.vr[k].app_chunk = base;
    }
    A.vr[k].app_cnt = a + 1;
    return A.vr[k].app_chunk + 1 + (a - cs);
}

// preconditioner.cc:248-271 on the array layout. Returns the live count.
RLAP_HD int32_t serial_gather(const Arrays& A, int32_t v, const ColBuf& B, int32_t cap) {
    int32_t len = 0;
    int32_t a = A.vr[v].app_cnt;
    if (a > 0) {
        int32_t base = A.vr[v].app_chunk;
        int c = chunk_of(a - 1);
        int32_t idx = a - 1;
        while (idx >= 0) {
            int32_t cs = chunk_start(c);
            for (int32_t t = idx; t >= cs; --t) {
                int32_t s = base + 1 + (t - cs);
                double val = A.e[s].val;
                if (val > 0) {
                    if (len < cap) { B.a_slot[len] = s; B.a_nbr[len] = A.e[s].nbr; B.a_val[len] = val; B.a_twin[len] = A.e[s].twin; }
                    ++len;
                }
            }
            idx = cs - 1;
            base = A.e[base].nbr;
            --c;
        }
    }
    for (int32_t s = A.colptr[v + 1] - 1; s >= A.colptr[v]; --s) {
        double val = A.e[s].val;
        if (val > 0) {
            if (len < cap) { B.a_slot[len] = s; B.a_nbr[len] = A.e[s].nbr; B.a_val[len] = val; B.a_twin[len] = A.e[s].twin; }
            ++len;
        }
    }
    return len;
}

// Bucket stacks == the reference's doubly linked bucket lists
// (preconditioner.cc:125-246): a move re-inserts the vertex at the bucket head,
// so a bucket pops its moved vertices newest first, then its never-moved
// vertices in descending id.  An entry is stale once the vertex moved again or
// was popped (pqpos != id); stale entries are dropped when they reach the top.
constexpr int BCH0 = 16;   // capacity of a bucket stack's first chunk
constexpr int BDIR = 26;   // directory entries per bucket (covers 2^30 entries)
RLAP_HD int bs_chunk_of(int a) { return ilog2_u32((unsigned)a / BCH0 + 1u); }
RLAP_HD int bs_chunk_start(int c) { return BCH0 * ((1 << c) - 1); }
RLAP_HD int bs_chunk_cap(int c) { return BCH0 << c; }
// pool slot of entry `a` of bucket `b` (global bucket index); chunk must exist
RLAP_HD int32_t bs_slot(const Arrays& A, int32_t b, int32_t a) {
    int c = bs_chunk_of(a);
    return A.bs_dir[(int64_t)b * BDIR + c] + (a - bs_chunk_start(c));
}
// make sure the chunk holding entry `a` of bucket b exists; returns false on pool overflow
RLAP_HD bool bs_ensure(const Arrays& A, int32_t b, int32_t a) {
    int c = bs_chunk_of(a);
    int32_t have = A.bs_alloc[b];
    while (have <= c) {
        int32_t need = bs_chunk_cap(have);
        int32_t base = pool_take(A.bs_pool_top, need);
        if (base < 0 || base > A.bs_pool_cap - need) return false;
        A.bs_dir[(int64_t)b * BDIR + have] = base;
        ++have;
    }
    A.bs_alloc[b] = have;
    return true;
}

RLAP_HD int pq_push(const Arrays& A, GraphDesc& G, int32_t v, int32_t lst) {
    int32_t b = G.bucket_base + lst;
    int32_t a = A.bs_cnt[b];
    if (!bs_ensure(A, b, a)) return ST_LOG_OVERFLOW;
    int32_t s = bs_slot(A, b, a);
    int32_t id = G.push_cnt++;
    A.bs_v[s] = v; A.bs_id[s] = id;
    A.bs_cnt[b] = a + 1;
    A.vr[v].pqpos = id;
    if (lst < G.minlist) G.minlist = lst;
    return ST_OK;
}

RLAP_HD int32_t pq_pop(const Arrays& A, GraphDesc& G) {
    while (true) {
        int32_t b = G.bucket_base + G.minlist;
        int32_t cnt = A.bs_cnt[b];
        while (cnt > 0) {
            int32_t s = bs_slot(A, b, cnt - 1);
            --cnt;
            int32_t v = A.bs_v[s];
            if (A.vr[v].pqpos == A.bs_id[s]) { A.bs_cnt[b] = cnt; A.vr[v].pqpos = -2; return v; }
        }
        A.bs_cnt[b] = 0;
        int32_t oc = A.ocur[b], oe = A.oend[b];
        while (oc < oe) {
            int32_t v = A.orig_order[oc++];
            if (A.vr[v].pqpos == -1) { A.ocur[b] = oc; A.vr[v].pqpos = -2; return v; }
        }
        A.ocur[b] = oc;
        G.minlist += 1;
        if (G.minlist > 2 * G.n) return -1;  // empty queue (callers never pop then)
    }
}

// Commit the per-target PQ results of one elimination: B.t_key/t_mv indexed by
// merged index x in [0,m), vertex ids in B.b_nbr.  Moves are pushed in op order.
RLAP_HD int pq_commit_serial(const Arrays& A, GraphDesc& G, const ColBuf& B, int32_t m) {
    int32_t cnt = 0;
    for (int32_t x = 0; x < m; ++x) {
        A.vr[B.b_nbr[x]].key = B.t_key[x];
        if (B.t_mv[x] >= 0) { B.rec[cnt].key = (double)B.t_mv[x]; B.rec[cnt].idx = x; B.rec[cnt].aux = 0; ++cnt; }
    }
    std_sort_emul<false>(B.rec, cnt);  // distinct keys
    for (int32_t i = 0; i < cnt; ++i) {
        int32_t x = B.rec[i].idx;
        int rc = pq_push(A, G, B.b_nbr[x], pq_list_of(B.t_key[x], G.n));
        if (rc) return rc;
    }
    return ST_OK;
}

// Sort gathered entries by neighbour id exactly as std::sort would
// (preconditioner.cc:275-276), then merge equal neighbours (:278-293).
// elim=true additionally kills the twins of merged duplicates and records the
// PQ Decs.  Leaves survivors in B.b_* (ascending id); returns their count.
RLAP_HD int32_t serial_sort_merge(const Arrays& A, const GraphDesc& G, const ColBuf& B, int32_t len0, bool elim, bool use_pq) {
    for (int32_t i = 0; i < len0; ++i) { B.rec[i].key = (double)B.a_nbr[i]; B.rec[i].idx = i; B.rec[i].aux = 0; }
    std_sort_emul<false>(B.rec, len0);
    int32_t m = 0;
    for (int32_t i = 0; i < len0; ++i) {
        int32_t s = B.rec[i].idx;
        if (m == 0 || B.a_nbr[s] != B.b_nbr[m - 1]) {
            B.b_slot[m] = B.a_slot[s]; B.b_nbr[m] = B.a_nbr[s]; B.b_val[m] = B.a_val[s]; B.b_twin[m] = B.a_twin[s];
            if (elim && use_pq) { B.t_key[m] = A.vr[B.a_nbr[s]].key; B.t_mv[m] = -1; }
            ++m;
        } else {
            B.b_val[m - 1] += B.a_val[s];
            if (elim) {
                A.e[B.a_twin[s]].val = 0;
                if (use_pq) pq_op(B.t_key[m - 1], B.t_mv[m - 1], G.n, -1, i);
            }
        }
    }
    return m;
}

// Order the m merged neighbours by o_n (:295-307) into B.a_*; B.t_of[j] = merged index.
// `vertex` and B.b_nbr are global ids of the graph starting at `vbase`; A.shuffle_seed is that graph's seed.
RLAP_HD void serial_order(const Arrays& A, const ColBuf& B, int32_t m, int32_t vertex, int phase, int32_t vbase) {
    if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) {
        uint64_t base = keyed_order_base(A.shuffle_seed, vertex - vbase, phase);
        for (int32_t i = 0; i < m; ++i) { B.rec[i].key = keyed_order_dkey(base, B.b_nbr[i] - vbase); B.rec[i].idx = i; B.rec[i].aux = 0; }
        std_sort_emul<false>(B.rec, m);
    } else {
        for (int32_t i = 0; i < m; ++i) { B.rec[i].key = B.b_val[i]; B.rec[i].idx = i; B.rec[i].aux = 0; }
        if (A.o_n == ON_ASC) std_sort_emul<false>(B.rec, m); else std_sort_emul<true>(B.rec, m);
    }
    for (int32_t j = 0; j < m; ++j) {
        int32_t x = B.rec[j].idx;
        B.t_of[j] = x;
        B.a_slot[j] = B.b_slot[x]; B.a_nbr[j] = B.b_nbr[x]; B.a_val[j] = B.b_val[x]; B.a_twin[j] = B.b_twin[x];
    }
}

RLAP_HD int32_t upper_index(const double* cum, int32_t m, double r) {  // first a with cum[a] > r, else m-1 (:387-393)
    int32_t lo = 0, hi = m;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (cum[mid] > r) hi = mid; else lo = mid + 1;
    }
    return lo < m ? lo : m - 1;
}

RLAP_HD int serial_rewire(const Arrays& A, int32_t s_r, int32_t nbr_j, int32_t k, double w) {
    int32_t s_n = col_append(A, k);
    if (s_n < 0) return ST_POOL_OVERFLOW;
    A.e[s_r].nbr = k; A.e[s_r].val = w; A.e[s_r].twin = s_n;
    A.e[s_n].nbr = nbr_j; A.e[s_n].val = w; A.e[s_n].twin = s_r;
    return ST_OK;
}

// One elimination, sequential form.  `e1` = 1-based elimination number (the
// reference's `it` before its increment).  preconditioner.cc:358-433 (degree),
// :723-787 (random), :846-914 (coarsen).
RLAP_HD int serial_eliminate(const Arrays& A, GraphDesc& G, const ColBuf& B, int32_t cap, int32_t v, int64_t e1) {
    const bool use_pq = A.o_v != OV_RANDOM;
    int32_t len0 = serial_gather(A, v, B, cap);
    if (len0 > cap) return ST_INTERNAL;
    int32_t m = serial_sort_merge(A, G, B, len0, true, use_pq);
    serial_order(A, B, m, v, 0, G.vbase);
    if (A.o_v == OV_COARSEN) {
        if (m >= 1) {
            double csum = 0;
            for (int32_t j = 0; j < m; ++j) { csum += B.a_val[j]; B.cum[j] = csum; }
            if (A.rng_mode != RNG_COUNTER && G.n_draws >= A.rng_len) return ST_RNG_OVERFLOW;
            double u = arr_uniform(A, G.n_draws, v, 0); G.n_draws += 1;
            double r = u * csum;
            int32_t koff = upper_index(B.cum, m, r);
            int32_t k = B.a_nbr[koff];
            double wk = B.a_val[koff];
            A.e[B.a_twin[koff]].val = 0;
            int32_t xk = B.t_of[koff];
            pq_op(B.t_key[xk], B.t_mv[xk], G.n, -1, len0);
            for (int32_t j = 0; j < m; ++j) {
                if (j == koff) continue;
                pq_op(B.t_key[xk], B.t_mv[xk], G.n, +1, len0 + 1 + j);
                double w = B.a_val[j];
                double nw = (wk * w) / (wk + w);
                int rc = serial_rewire(A, B.a_twin[j], B.a_nbr[j], k, nw);
                if (rc) return rc;
            }
        }
    } else {
        double csum = 0;
        for (int32_t j = 0; j < m; ++j) { csum += B.a_val[j]; B.cum[j] = csum; }
        if (A.rng_mode != RNG_COUNTER && m > 1 && G.n_draws + (m - 1) > A.rng_len) return ST_RNG_OVERFLOW;
        double wdeg = csum, colScale = 1;
        for (int32_t j = 0; j < m - 1; ++j) {
            double w = B.a_val[j] * colScale;
            double f = w / wdeg;
            double u = arr_uniform(A, G.n_draws + j, v, j);
            double r = u * (csum - B.cum[j]) + B.cum[j];
            int32_t koff = upper_index(B.cum, m, r);
            B.ksel[j] = koff;
            double omf = 1 - f;
            B.newv[j] = f * omf * wdeg;
            colScale = colScale * omf;
            wdeg = wdeg * omf * omf;
        }
        if (m > 1) G.n_draws += m - 1;
        for (int32_t j = 0; j < m - 1; ++j) {
            int32_t koff = B.ksel[j];
            if (use_pq) { int32_t xk = B.t_of[koff]; pq_op(B.t_key[xk], B.t_mv[xk], G.n, +1, len0 + j); }
            int rc = serial_rewire(A, B.a_twin[j], B.a_nbr[j], B.a_nbr[koff], B.newv[j]);
            if (rc) return rc;
        }
        if (m > 0) {
            if (use_pq && e1 + 1 < (int64_t)G.n) { int32_t xl = B.t_of[m - 1]; pq_op(B.t_key[xl], B.t_mv[xl], G.n, -1, len0 + m); }
            A.e[B.a_twin[m - 1]].val = 0;
        }
    }
    if (use_pq) return pq_commit_serial(A, G, B, m);
    return ST_OK;
}

// Output of one surviving vertex (:440-453 + :312-345): rows (nbr, v, w) in o_n
// order written to out[3*row0 ...]; returns the row count. count_only skips writes.
RLAP_HD int32_t serial_output(const Arrays& A, const GraphDesc& G, const ColBuf& B, int32_t cap, int32_t v, double* out, int64_t row0, bool count_only) {
    int32_t len0 = serial_gather(A, v, B, cap);
    if (len0 > cap) return -1;
    int32_t m = serial_sort_merge(A, G, B, len0, false, false);
    if (count_only) return m;
    serial_order(A, B, m, v, 1, G.vbase);
    for (int32_t j = 0; j < m; ++j) {
        double* o = out + 3 * (row0 + j);
        o[0] = (double)B.a_nbr[j]; o[1] = (double)v; o[2] = B.a_val[j];
    }
    return m;
}


// ---------------------------------------------------------------------------
// Batch ("frontier") elimination: the next B pops of the sequential order are
// prepared side by side and committed as the longest prefix whose members are
// pairwise non-adjacent and that no PQ move pre-empts -- exactly the sequential
// result, with the RNG offsets handed out by a prefix sum.
// Per-candidate state lives in one record (LDS on the device).
// ---------------------------------------------------------------------------
constexpr int BCAP = 32;   // live entries a batch candidate may have; longer ones take the single-vertex path

struct TRes {     // per-target PQ replay result, stored over Ent::aux
    int32_t key_after;
    int16_t mv;   // op number of the target's last bucket move by this candidate, -1 none
    uint8_t c;    // pushes into the target's column (= Incs)
    uint8_t flags;
};
struct Ent {
    double val;   // weight; after sampling: the new edge weight of position j
    union {
        double aux;   // keyed-order key / cumulative weight ...
        TRes res;     // ... after sampling: the replay result
    };
    int32_t nbr;
    int32_t twin;
};
enum { CF_BIG = 1, CF_DUP = 2, CF_COMPLEX = 4, CF_DEP = 8, CF_TIE = 16, CF_NEQ = 32, CF_READY = 64, CF_PATCHED = 128 };
constexpr int DEPMAX = 4;   // dependent candidates with more earlier neighbours in the round than this cut it
enum { TF_CONTENDED = 1 };

// BC = live entries a candidate may hold (32: one half-wave per candidate, 64: one wave, 128: one wave, two entries per lane).
template <int BC> struct CandPad {};
template <> struct CandPad<64> { int64_t pad64; };   // keeps the LDS stride of CandT<64> off the 64-bank period
template <> struct CandPad<128> { int64_t pad128; };
template <int BC>
struct CandT : CandPad<BC> {
    static constexpr int CAP = BC;
    Ent e[BC];
    uint8_t ksel[BC];
    int32_t v;
    int32_t m;        // live (= distinct) neighbours
    int32_t flags;
    int32_t src;      // >= 0: bucket-stack index it was read from; < 0: ~(index into orig_order)
    int32_t ndraw;
    int32_t koff;     // coarsen: chosen position
    int32_t cp1;      // colptr[v+1]
    int32_t acnt;     // appended entries
    int32_t ext;      // slots to read (appended + CSR), dead ones included
    int32_t nmv;      // PQ moves this candidate causes (counted during the replay; bounds the round's move list)
    int32_t nkill;    // merged multi-edges (o_v = random, 64-slot form): e[m .. m+nkill) hold the twins that die (:289)
    int32_t cb[BC <= 32 ? 3 : (BC <= 64 ? 4 : 5)];   // bases of the appended chunks (chunk_of(BC - 1) + 1)
    int32_t ndep;     // earlier candidates of the round this one is adjacent to (cand_patch), at most DEPMAX
    uint8_t dep[4];
    int64_t draw0;    // first uniform
};   // 872 B (BC=32) / 1688 B (BC=64) / 3288 B (BC=128): word strides 218 / 422 / 822
typedef CandT<BCAP> Cand;

RLAP_HD TRes& ent_tres(Ent& e) { return e.res; }

struct EntLessNbr { RLAP_HD bool operator()(const Ent& x, const Ent& y) const { return x.nbr < y.nbr; } };
struct EntLessVal { RLAP_HD bool operator()(const Ent& x, const Ent& y) const { return x.val < y.val; } };
struct EntGreaterVal { RLAP_HD bool operator()(const Ent& x, const Ent& y) const { return x.val > y.val; } };
struct EntLessAux { RLAP_HD bool operator()(const Ent& x, const Ent& y) const { return x.aux < y.aux; } };

// (key, index) view over a candidate's Ent::aux / ksel[] arrays, so the o_n order can
// be found by moving 9 bytes per element instead of whole entries.  The permutation
// std::sort produces depends on the comparison outcomes only, not on the payload.
struct KI { double k; uint8_t idx; };
struct KIRef {
    Ent* e; uint8_t* ix; int i;
    RLAP_HD operator KI() const { KI v; v.k = e[i].aux; v.idx = ix[i]; return v; }
    RLAP_HD KIRef& operator=(const KI& v) { e[i].aux = v.k; ix[i] = v.idx; return *this; }
    RLAP_HD KIRef& operator=(const KIRef& o) { KI v = o; e[i].aux = v.k; ix[i] = v.idx; return *this; }
};
struct KIArr {
    Ent* e; uint8_t* ix;
    RLAP_HD KIRef operator[](int i) const { KIRef r; r.e = e; r.ix = ix; r.i = i; return r; }
};
struct KILess { RLAP_HD bool operator()(const KI& x, const KI& y) const { return x.k < y.k; } };
struct KIGreater { RLAP_HD bool operator()(const KI& x, const KI& y) const { return x.k > y.k; } };

// o_n order of a candidate whose entries are already sorted by id and carry the sort key
// in aux (weight, or the keyed-order key): leaves the source index of position j in ksel[j].
template <class CT>
RLAP_HD void cand_order_index(const Arrays& A, CT& C) {
    const int32_t m = C.m;
    for (int32_t j = 0; j < m; ++j) C.ksel[j] = (uint8_t)j;
    KIArr arr; arr.e = C.e; arr.ix = C.ksel;
    if (CT::CAP <= 32) {   // the stack-free variant is only valid up to 32 elements
        if (A.o_n == ON_DESC && A.o_v != OV_COARSEN) gs_std_sort_small<KI>(arr, m, KIGreater());
        else gs_std_sort_small<KI>(arr, m, KILess());
    } else {
        if (A.o_n == ON_DESC && A.o_v != OV_COARSEN) gs_std_sort<KI>(arr, m, KIGreater());
        else gs_std_sort<KI>(arr, m, KILess());
    }
    C.ndraw = (A.o_v == OV_COARSEN) ? (m >= 1 ? 1 : 0) : (m > 1 ? m - 1 : 0);
}

// Multi-edges inside a batch candidate: merged in place for o_v = random with 64-slot candidates (every
// fourth vertex met in random order on small dense graphs has one); the PQ orders send them to the
// single-vertex path, where the DegreePQDec per merged entry (:291) is replayed.
template <class CT>
RLAP_HD bool cand_merges_multi_edges(const Arrays& A) { return CT::CAP >= 64 && A.o_v == OV_RANDOM; }

// Candidate preparation in three steps so the loads can be spread over all
// threads: cand_meta (one per candidate), cand_load (one per slot), cand_finish.
template <class CT>
RLAP_HD void cand_meta(const Arrays& A, int32_t v, CT& C) {
    C.v = v; C.flags = 0; C.m = 0; C.ndraw = 0; C.koff = 0; C.nkill = 0; C.nmv = 0; C.ndep = 0;
    int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
    int32_t acnt = A.vr[v].app_cnt;
    int32_t base = A.vr[v].app_chunk;              // read with the counts (not after them): one dependent round trip less
    C.cp1 = cp1; C.acnt = acnt; C.ext = (cp1 - cp0) + acnt;
    if (C.ext > CT::CAP) { C.flags = CF_BIG; C.ext = 0; return; }
    if (acnt > 0) {
        int ct = chunk_of(acnt - 1);            // <= 2 (CAP 32) / <= 3 (CAP 64) / <= 4 (CAP 128) because acnt <= CAP
        for (int c = ct; c >= 0; --c) { C.cb[c] = base; if (c > 0) base = A.e[base].nbr; }
    }
}
// traversal position e of the column (:248-271): appended entries newest first, then the CSR segment backwards
template <class CT>
RLAP_HD int32_t cand_slot(const CT& C, int32_t e) {
    if (e < C.acnt) {
        int32_t a = C.acnt - 1 - e;
        int c = chunk_of(a);
        return C.cb[c] + 1 + (a - chunk_start(c));
    }
    return C.cp1 - 1 - (e - C.acnt);
}
template <class CT>
RLAP_HD void cand_load(const Arrays& A, CT& C, int32_t e) {
    int32_t s = cand_slot(C, e);
    C.e[e].val = A.e[s].val; C.e[e].nbr = A.e[s].nbr; C.e[e].twin = A.e[s].twin; C.e[e].aux = 0;
}
// drop dead entries -> sort by id (:275) -> multi-edges go to the single-vertex path -> order by o_n (:295-307)
template <class CT>
RLAP_HD void cand_finish(const Arrays& A, CT& C, int32_t vbase = 0) {
    if (C.flags & CF_BIG) return;
    int32_t len = 0;
    for (int32_t e = 0; e < C.ext; ++e) {
        if (C.e[e].val > 0) { if (len != e) C.e[len] = C.e[e]; ++len; }
    }
    C.m = len;
    gs_std_sort<Ent>(C.e, len, EntLessNbr());
    if (cand_merges_multi_edges<CT>(A)) {
        // merge equal ids in sorted order (:278-293): the first keeps its slot and twin and takes the sum,
        // the twins of the others die at commit (no PQ in this mode, so nothing else to replay)
        int32_t mm = 0, nk = 0;
        int32_t kill[CT::CAP];
        for (int32_t i = 0; i < len; ++i) {
            if (mm == 0 || C.e[i].nbr != C.e[mm - 1].nbr) { if (mm != i) C.e[mm] = C.e[i]; ++mm; }
            else { C.e[mm - 1].val += C.e[i].val; kill[nk++] = C.e[i].twin; }
        }
        for (int32_t q = 0; q < nk; ++q) C.e[mm + q].twin = kill[q];
        C.nkill = nk;
        len = mm;
        C.m = len;
    } else {
        for (int32_t i = 1; i < len; ++i) if (C.e[i].nbr == C.e[i - 1].nbr) { C.flags = CF_DUP; return; }
    }
    if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) {
        uint64_t kb = keyed_order_base(A.shuffle_seed, C.v - vbase, 0);
        for (int32_t i = 0; i < len; ++i) C.e[i].aux = keyed_order_dkey(kb, C.e[i].nbr - vbase);
        gs_std_sort<Ent>(C.e, len, EntLessAux());
    } else if (A.o_n == ON_ASC) {
        gs_std_sort<Ent>(C.e, len, EntLessVal());
    } else {
        gs_std_sort<Ent>(C.e, len, EntGreaterVal());
    }
    C.ndraw = (A.o_v == OV_COARSEN) ? (len >= 1 ? 1 : 0) : (len > 1 ? len - 1 : 0);
}
template <class CT>
RLAP_HD void cand_prepare(const Arrays& A, int32_t v, CT& C, int32_t vbase = 0) {
    cand_meta(A, v, C);
    for (int32_t e = 0; e < C.ext; ++e) cand_load(A, C, e);
    cand_finish(A, C, vbase);
}

// A candidate d adjacent to an EARLIER candidate j of the same round sees j eliminated first (preconditioner.cc:404-414):
// with d at position p of j's order, the twin of j's entry -- d's entry towards j -- is rewritten in place to
// (k, w_new) where k is the neighbour j sampled for position p, and its twin becomes the entry j appends to k's column.
// That is all that happens to d as long as d is neither sampled as a target by j (it would get an appended entry and an
// Inc: it leaves the predicted order) nor j's last neighbour (its entry dies, Dec: it pre-empts the order) and as long as
// k is not a neighbour of d already (a multi-edge: the PQ orders send those to the single-vertex path) and the new weight
// is positive (rounding can make it <= 0, and then the rewritten entry is dead: `val > 0`, :252).  Then d's
// record can be patched from j's sampled record without touching memory; the number of neighbours, hence the number of
// uniforms d draws, is unchanged.  The twin is recorded as ~(j * CAP + p): resolved from the round's slot table at commit.
// Sequential form (one thread); returns false when d cannot be patched (the round is cut before it).
// Must run after every earlier candidate has been sampled (ksel, new weights in e[].val) and before d is.
template <class CT>
RLAP_HD bool cand_patch(const Arrays& A, CT* cand, int32_t d, int32_t vbase) {
    CT& C = cand[d];
    // a record with merged multi-edges (64-slot candidates, o_v = random) cannot be patched: WHICH duplicate survived the merge
    // is std::sort's (unstable) choice over the whole gathered list, and that list would hold k instead of j
    if (C.nkill > 0) return false;
    for (int32_t q = 0; q < C.ndep; ++q) {
        const int32_t j = C.dep[q];
        const CT& J = cand[j];
        int32_t p = -1;
        for (int32_t z = 0; z < J.m; ++z) if (J.e[z].nbr == C.v) p = z;
        if (p < 0 || p == J.m - 1) return false;                        // (last neighbour: :420-432)
        for (int32_t z = 0; z < J.m - 1; ++z) if (J.ksel[z] == p) return false;   // sampled as a target (:394-399)
        // rounding can leave f > 1 near the end of a long column (wdeg ~ 1e-31): the new weight is then <= 0 and the rewritten
        // entry is dead for getColumnLength (:252 `val > 0`) -- d has one neighbour less, which only a fresh gather sees
        if (!(J.e[p].val > 0)) return false;
        const int32_t k = J.e[J.ksel[p]].nbr;
        int32_t qd = -1;
        for (int32_t z = 0; z < C.m; ++z) { if (C.e[z].nbr == J.v) qd = z; if (C.e[z].nbr == k) { C.flags |= CF_DUP; return false; } }
        if (qd < 0) return false;
        C.e[qd].nbr = k; C.e[qd].val = J.e[p].val; C.e[qd].twin = ~(j * CT::CAP + p);
    }
    C.flags |= CF_PATCHED;
    // order again: by id (:275; ids are distinct), then by o_n (:295-307) -- as cand_finish does
    const int32_t len = C.m;
    gs_std_sort<Ent>(C.e, len, EntLessNbr());
    if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) {
        uint64_t kb = keyed_order_base(A.shuffle_seed, C.v - vbase, 0);
        for (int32_t i = 0; i < len; ++i) C.e[i].aux = keyed_order_dkey(kb, C.e[i].nbr - vbase);
        gs_std_sort<Ent>(C.e, len, EntLessAux());
    } else if (A.o_n == ON_ASC) {
        gs_std_sort<Ent>(C.e, len, EntLessVal());
    } else {
        gs_std_sort<Ent>(C.e, len, EntGreaterVal());
    }
    return true;
}

RLAP_HD int32_t ent_upper_index(const Ent* e, int32_t m, double r) {  // first a with cum[a] > r, else m-1
    int32_t lo = 0, hi = m;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (e[mid].aux > r) hi = mid; else lo = mid + 1;
    }
    return lo < m ? lo : m - 1;
}

// cumulative weights (:366-373); coarsen also draws its single target and the
// collapsed weights here (:867-897)
template <class CT>
RLAP_HD void cand_cumsum(const Arrays& A, CT& C) {
    const int32_t m = C.m;
    double csum = 0;
    for (int32_t j = 0; j < m; ++j) { csum += C.e[j].val; C.e[j].aux = csum; }
    if (A.o_v == OV_COARSEN && m >= 1) {
        double u = arr_uniform(A, C.draw0, C.v, 0);
        double r = u * csum;
        int32_t koff = ent_upper_index(C.e, m, r);
        C.koff = koff;
        double wk = C.e[koff].val;
        for (int32_t j = 0; j < m; ++j) {
            if (j == koff) continue;
            double w = C.e[j].val;
            C.e[j].val = (wk * w) / (wk + w);
            C.ksel[j] = (uint8_t)koff;
        }
    }
}
// sampled target of position j < m-1 (:385-394); degree/random only
template <class CT>
RLAP_HD void cand_pick(const Arrays& A, CT& C, int32_t j) {
    const int32_t m = C.m;
    double csum = C.e[m - 1].aux;
    double u = arr_uniform(A, C.draw0 + j, C.v, j);
    double cj = C.e[j].aux;
    double r = u * (csum - cj) + cj;
    C.ksel[j] = (uint8_t)ent_upper_index(C.e, m, r);
}
// the f / colScale / wdeg recurrences (:374-417): leaves the new edge weight of position j in e[j].val
template <class CT>
RLAP_HD void cand_recur(const Arrays& A, CT& C) {
    if (A.o_v == OV_COARSEN) return;
    const int32_t m = C.m;
    if (m < 1) return;
    double wdeg = C.e[m - 1].aux, colScale = 1;
    for (int32_t j = 0; j < m - 1; ++j) {
        double w = C.e[j].val * colScale;
        double f = w / wdeg;
        double omf = 1 - f;
        C.e[j].val = f * omf * wdeg;
        colScale = colScale * omf;
        wdeg = wdeg * omf * omf;
    }
}
template <class CT>
RLAP_HD void cand_sample(const Arrays& A, CT& C) {
    cand_cumsum(A, C);
    if (A.o_v != OV_COARSEN) for (int32_t j = 0; j < C.m - 1; ++j) cand_pick(A, C, j);
    cand_recur(A, C);
}

// Picks that name target j among ksel[8w .. 8w+8) (positions below `limit` only): 0x80 in byte q of the result iff
// ksel[8w+q] == j.  One 8-byte load and a handful of ALU operations instead of eight byte loads and compares -- the replay
// and the slot phases ask this for every (candidate, target) pair.  (ksel is 8-byte aligned in every CandT.)
RLAP_HD int popcount_u64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
RLAP_HD int clz_u64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
RLAP_HD int ctz_u64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
RLAP_HD uint64_t ksel_matches(const uint8_t* ksel, int32_t w, int32_t j, int32_t limit) {
    const uint64_t lo7 = 0x7F7F7F7F7F7F7F7Full;
    const uint64_t x = reinterpret_cast<const uint64_t*>(ksel)[w] ^ (0x0101010101010101ull * (uint64_t)(uint8_t)j);
    uint64_t y = ~(((x & lo7) + lo7) | x | lo7);          // 0x80 in every byte of x that is zero
    const int32_t valid = limit - 8 * w;                   // bytes of this word that are positions below the limit
    if (valid < 8) y &= valid <= 0 ? 0ull : ((1ull << (8 * valid)) - 1ull);
    return y;
}
// Net PQ effect of candidate C on its target at position j, starting from key0
// (valid while every intermediate key stays <= n, where each change moves the
// vertex; otherwise *complex is set).  Returns the new key; *mv = op number of
// the last move or -1; *cnt = pushes into the target's column.
// Split in two: what does not depend on the key (pushes into the target's column, position of the last one) ...
template <class CT>
RLAP_HD void cand_replay_pre(const Arrays& A, const CT& C, int32_t j, int* cnt, int* last_inc) {
    const int32_t m = C.m;
    int32_t li = -1, c = 0;
    if (A.o_v == OV_COARSEN) {
        if (j == C.koff) c = m - 1;
    } else {
        // positions q < m-1 with ksel[q] == j, eight picks per load (ksel_matches)
        for (int32_t w = 0; 8 * w < m - 1; ++w) {
            const uint64_t y = ksel_matches(C.ksel, w, j, m - 1);
            if (y) { c += popcount_u64(y); li = 8 * w + ((63 - clz_u64(y)) >> 3); }
        }
    }
    *cnt = c; *last_inc = li;
}
// ... and the key arithmetic on top of it (a target shared by several candidates chains it through their records)
template <class CT>
RLAP_HD int32_t cand_replay_post(const Arrays& A, const CT& C, int32_t j, int32_t key0, int32_t n, bool allow_last_dec,
                                 int c, int last_inc, int* mv, bool* complex) {
    const int32_t m = C.m;
    int32_t key = key0;
    *mv = -1;
    if (A.o_v == OV_COARSEN) {
        if (j == C.koff) {
            if (key != 1) { key -= 1; *mv = m; }            // :882, op number m
            if (c > 0) {
                key += c;
                int32_t lastj = (C.koff == m - 1) ? m - 2 : m - 1;
                *mv = m + 1 + lastj;                           // :896
            }
        }
    } else {
        if (c > 0) { key += c; *mv = m + last_inc; }          // :399, op numbers m + q
        if (j == m - 1 && allow_last_dec && key != 1) { key -= 1; *mv = 2 * m; }  // :427
    }
    if (key0 > n || key0 + c > n) *complex = true;
    return key;
}
template <class CT>
RLAP_HD int32_t cand_replay(const Arrays& A, const CT& C, int32_t j, int32_t key0, int32_t n, bool allow_last_dec,
                            int* mv, int* cnt, bool* complex) {
    int c, li;
    cand_replay_pre(A, C, j, &c, &li);
    *cnt = c;
    return cand_replay_post(A, C, j, key0, n, allow_last_dec, c, li, mv, complex);
}

}  // namespace rlap
