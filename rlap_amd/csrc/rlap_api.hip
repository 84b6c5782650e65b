// rlap_api.hip -- C ABI (include/rlap_hip.h) and host orchestration of the kernels.
// Host code here only sizes buffers, launches kernels on the handle's stream and
// reads back a few scalars; all arithmetic of the path runs in rlap_kernels.hip.
// rocPRIM (AMD's native primitives, header-only) supplies the device radix sort
// and prefix sums used for plumbing (COO ordering, offsets).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/rlap_hip.h"
#include "rlap_core.h"
#include "rlap_kernels.h"

using namespace rlap;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::fprintf(stderr, "[rlap_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), __FILE__, __LINE__); return RLAP_E_HIP; } } while (0)
#define ENSURE(buf, bytes) HIPCHK((buf).ensure((size_t)(bytes)))

inline unsigned nblk(int64_t n, int bs) { return (unsigned)std::max<int64_t>(1, (n + bs - 1) / bs); }

}  // namespace

// Every per-call buffer is carved out of ONE arena (256-byte aligned pieces, the layout is a function of the host-known sizes E, n, G
// and the handle's growth factors).  The arena is either the caller's (rlap_set_workspace: torch's allocator in rlap_amd/ops.py -- the
// reference lets torch allocate what the op needs, py_api_binder.cc:42) or the handle's own (one hipMalloc, grown only when a call needs
// a larger size class than any before it).  The cached uniform table is a second, persistent buffer under the same rule.
struct Carver {
    char* base; size_t off;
    template <class T> T* take(int64_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * (size_t)(count > 0 ? count : 0);
        return p;
    }
};

struct rlap_handle_s {
    int device = 0;
    int n_cu = 0;                 // compute units of `device`
    bool big_attr_set = false;    // k_sc_merge_big's 128 KB dynamic-LDS opt-in (a per-device function attribute)
    std::mutex mu;                // one call at a time per handle (the workspace is the handle's); use one handle per host thread for concurrency
    hipStream_t stream = nullptr;
    bool timing = false;
    hipEvent_t ev[8];
    hipStream_t side[2] = {nullptr, nullptr};   // the output pass's independent tiers run side by side
    hipEvent_t fork_ev[3];
    // the arena: caller-provided (ext_ws) or owned
    void* ext_ws = nullptr; size_t ext_ws_bytes = 0;
    DevBuf own_ws;
    size_t ws_needed = 0;         // bytes the last call wanted (reported with RLAP_E_WORKSPACE)
    // the uniform table: caller-provided or owned; rng_len = entries generated so far in the current buffer
    double* ext_rng = nullptr; int64_t ext_rng_cap = 0;
    DevBuf own_rng;
    double* rng_ptr = nullptr;
    int64_t rng_len = 0;
    int64_t rng_needed = 0;       // entries the last call wanted
    DevBuf eqtab;   // k_eq_tables, allocated and built at rlap_create
    DevBuf small;   // two 64-bit words for rlap_approx_chol_from_edges' num_nodes reduction (max id + 1, negative-id flag)
    DevBuf prof;    // diagnostic phase profile (RLAP_PHASE_PROFILE=1), allocated at rlap_create
    DevBuf trace;   // diagnostic per-position time stamps of the dataflow kernel (RLAP_FLOW_TRACE=<file>)
    // pinned host mirror of what a call reads back
    void* h_results = nullptr; size_t h_results_cap = 0;
    // growth factors kept across calls
    double pool_factor = 1.0;
    double log_factor = 2.0;
    int64_t rng_min = 0;          // lower bound of the uniform table (grown after an overflow)
    int64_t scr_budget = 1 << 18; // output pass: entries of global scratch for columns beyond 8192 slots
    double flow_scr_factor = 1.0; // dataflow elimination: working storage of the long columns (grown after ST_FLOW_SCRATCH)
    bool flow_off_once = false;   // the next attempt uses the round kernel (set after ST_FLOW_REORDER)
    int rng_mode = 0;             // 0 the reference's one stream (mode "exact"), 1 counter-based uniforms (mode "frontier", rlap_set_rng_mode)
    // test hooks (rlap_debug_set_limits): tiny first sizes so that the retry path runs
    double dbg_pool = -1.0, dbg_log = -1.0; int64_t dbg_rng = -1, dbg_scr = -1;
    int64_t total_retries = 0;
    bool force_sort = false;      // the next attempt sorts the COO whatever order it is in (set when a skipped sort cannot be trusted)
    int jitter = 0;               // RLAP_DEBUG_JITTER=<n>: waves of the elimination kernel sleep n x 0.25 us behind its barriers (a different set each time)
    int poison = -1;              // RLAP_DEBUG_POISON=<byte>: the arena, the output buffer and the elimination kernel's LDS are filled with it before every attempt
};

namespace {

// device-side scalars of a call, zeroed by one memset
struct Scalars {
    int32_t flags[FLAG_COUNT];
    int32_t nnz; int32_t pad;
    double acc[4];
    unsigned long long counters[8];   // [0] sc scratch top, [4..6] tier counts (6 ints)
    unsigned long long maxid;
    unsigned long long live[LIVE_SLOTS * LIVE_STRIDE];   // live entries read by the output pass, partial sums
};

// The first `count` uniforms of the sampling stream (K10) in the handle's table.  The table lives in the caller's buffer
// (rlap_set_rng_table) or in the handle's own; it is generated from the start whenever the buffer changes or must grow.
int ensure_rng(rlap_handle h, int64_t count) {
    h->rng_needed = std::max<int64_t>(count, 1 << 16);
    if (h->rng_ptr && count <= h->rng_len) return RLAP_OK;
    int64_t want = h->rng_needed;
    double* buf = nullptr;
    if (h->ext_rng) {
        if (want > h->ext_rng_cap) return RLAP_E_WORKSPACE;
        buf = h->ext_rng; want = h->ext_rng_cap;
    } else {
        ENSURE(h->own_rng, sizeof(double) * (size_t)want);
        buf = h->own_rng.as<double>(); want = (int64_t)(h->own_rng.cap / sizeof(double));
    }
    hipLaunchKernelGGL(k_mt19937_64_table, dim3(1), dim3(320), 0, h->stream, buf, want);
    HIPCHK(hipGetLastError());
    h->rng_ptr = buf; h->rng_len = want;
    return RLAP_OK;
}

// temporary storage of the rocPRIM calls: sized up front (queries with a null pointer touch nothing) and carved from the arena
struct SortTmp { void* p; size_t bytes; };

template <class K, class V>
int sort_pairs(rlap_handle h, const SortTmp& T, K* k_in, K* k_out, V* v_in, V* v_out, int64_t n, unsigned begin_bit, unsigned end_bit) {
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit, h->stream));
    if (bytes > T.bytes) return RLAP_E_INTERNAL;   // (sort_tmp_bytes asks the same questions up front)
    bytes = T.bytes;
    HIPCHK(rocprim::radix_sort_pairs(T.p, bytes, k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit, h->stream));
    return RLAP_OK;
}

template <class In, class Out>
int excl_scan(rlap_handle h, const SortTmp& T, In* in, Out* out, int64_t n) {
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, (Out)0, (size_t)n, rocprim::plus<Out>(), h->stream));
    if (bytes > T.bytes) return RLAP_E_INTERNAL;
    bytes = T.bytes;
    HIPCHK(rocprim::exclusive_scan(T.p, bytes, in, out, (Out)0, (size_t)n, rocprim::plus<Out>(), h->stream));
    return RLAP_OK;
}

inline unsigned bits_for(uint64_t maxval) { unsigned b = 1; while (b < 64 && (maxval >> b) != 0) ++b; return b; }

// largest temporary any rocPRIM call of a call with these sizes asks for (the very calls run_once makes, asked with null storage)
int sort_tmp_bytes(int64_t Eeff, int64_t N, int64_t G, int64_t S, bool flow, size_t* out) {
    size_t mx = 256, b = 0;
    hipStream_t s0 = nullptr;
    const unsigned kbits = bits_for((uint64_t)(N > 1 ? N - 1 : 1)), gbits = bits_for((uint64_t)(G > 0 ? G - 1 : 0));
    if (Eeff > 0) {
        b = 0; HIPCHK(rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)Eeff, 0u, std::min(64u, 2 * kbits + 1), s0)); mx = std::max(mx, b);
        rocprim::counting_iterator<uint32_t> iota(0u);
        b = 0; HIPCHK(rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, iota, (uint32_t*)nullptr, (size_t)Eeff, 0u, std::min(32u, kbits + 1), s0)); mx = std::max(mx, b);
        b = 0; HIPCHK(rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(Eeff + 1), rocprim::plus<int32_t>(), s0)); mx = std::max(mx, b);
    }
    const size_t Nn = (size_t)std::max<int64_t>(N, 1);
    b = 0; HIPCHK(rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, Nn, 0u, 32u + gbits, s0)); mx = std::max(mx, b);
    b = 0; HIPCHK(rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, Nn, 0u, 64u, s0)); mx = std::max(mx, b);
    b = 0; HIPCHK(rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(S + 1), rocprim::plus<int64_t>(), s0)); mx = std::max(mx, b);
    if (flow && Eeff > 0) {   // tag order of the surviving columns (rlap_flow.hip): one (column, tag) sort over at most Eeff appended entries
        b = 0; HIPCHK(rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)Eeff, 0u, 64u, s0)); mx = std::max(mx, b);
        b = 0; HIPCHK(rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(S + 1), rocprim::plus<int32_t>(), s0)); mx = std::max(mx, b);
    }
    *out = mx;
    return RLAP_OK;
}

// o_v = random runs the dataflow kernel (rlap_flow.hip) unless RLAP_FLOW=0 asks for the round kernel.  Default: single graphs and
// pairs, and batches of up to 64 graphs of 1024 vertices or more on average -- there the waves of the whole device work on few graphs
// (BA(4096,8), ms per call, round kernel / dataflow: 4 graphs 9.5 / 4.6, 16: 10.1 / 6.0, 32: 10.6 / 7.4, 64: 9.7-15.8 / 9.2, 96: 11.4 /
// 11.1, 128: 11.7 / 12.4, 1024: 30 / 59).  A batch of many graphs is bound by the positions in flight, where one workgroup per graph
// with a round's loads all in flight together does better.  RLAP_FLOW=1 forces the dataflow kernel for any batch.
inline bool flow_wanted(const rlap_handle h, int o_v, int64_t G, int64_t n_total) {
    if (o_v != OV_RANDOM || h->flow_off_once) return false;
    const char* e = std::getenv("RLAP_FLOW");
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return true;
    return G <= 2 || (G <= 64 && n_total >= 1024 * G);
}

constexpr int64_t SORT_SKIP_MIN = 1 << 21;   // directed entries from which the order of the input is looked at before sorting it

struct Fills {
    FillJobs J; int64_t maxcount = 0;
    Fills() { J.n = 0; }
    void add(void* p, int64_t count_i32, int32_t v) { if (count_i32 <= 0) return; J.ptr[J.n] = (int32_t*)p; J.count[J.n] = count_i32; J.value[J.n] = v; ++J.n; maxcount = std::max(maxcount, count_i32); }
    void launch(hipStream_t s) {
        if (J.n == 0) return;
        unsigned gx = (unsigned)std::min<int64_t>((maxcount + 255) / 256, 2048);
        hipLaunchKernelGGL(k_fill_multi, dim3(gx, (unsigned)J.n), dim3(256), 0, s, J);
        J.n = 0; maxcount = 0;
    }
};

// keyed random permutation of each graph's local ids, drawn on the device (stands in for the
// std::shuffle(random_device) of preconditioner.cc:594-596 when the caller injects no node_id vector).
// Graph g of a batch is keyed by seed + g like the neighbour order (rlap_core.h): a batched or sharded call draws what
// single-graph calls with seed + g draw, whatever the sharding.  64 hash bits: no ties to break at any graph size.
__global__ void k_perm_keys(const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd, int32_t N, uint64_t seed, int gbits,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t g = vgraph[i];
    const uint32_t local = (uint32_t)(i - gd[g].vbase);
    const uint64_t hk = mix64(mix64((seed + (uint64_t)g) ^ 0x7065726D5F6B6579ull) ^ mix64(local));
    // one graph: the whole hash; a batch: the graph number on top of 64 - gbits hash bits (one 64-bit sort either way)
    keys[i] = gbits == 0 ? hk : (((uint64_t)(uint32_t)g << (64 - gbits)) | (hk >> gbits));
    vals[i] = local;
}
__global__ void k_widen_u32(const uint32_t* __restrict__ in, int32_t N, int64_t* __restrict__ out) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = (int64_t)in[i];
}


// Host-known sizes of one call (upper bounds; what only the device knows -- nnz, row counts -- never sizes a buffer).
struct Sizes {
    int64_t G, N, Eeff, slot_cap, bucket_total, log_total, scr_total, S, scr_budget;
    size_t sort_tmp, res_bytes;
    bool want_genperm;
    bool flow;                    // dataflow elimination (o_v = random): its arrays are part of the arena
    int64_t flow_Q, flow_scr;     // look-back words (positions + one sentinel per graph); entries of long-column working storage
};
struct WBuf { void* p = nullptr; template <class T> T* as() const { return reinterpret_cast<T*>(p); } };
struct WS {
    WBuf node_ptr_d, vgraph, scal, keys0, keys1, idx0, idx1, head, pos, sorttmp, colptr, slot_col, permchk, genperm, ent, vrec, ocur, oend, origpos, orig_order, gd_d, pool_top, bs_cnt, bs_alloc, bs_dir, bs_v, bs_id, bs_pool_top, batch_pos, skey0, skey1, sval0, sval1, scr_rec, scr_i32, scr_f64, surv_base_d, ext, tmp_off, tmp_nbr, tmp_val, cnt, row_off, sc_rec, sc_i32, sc_f64, biglist, hugelists, results,
         f_cdir, f_atag, f_lb, f_qv, f_qg, f_ctrl, f_scr, f_rocnt, f_rooff;
};
inline size_t host_block_bytes(int64_t G) { return 16 * (size_t)(G + 1) + sizeof(GraphDesc) * (size_t)G; }
// the arena's layout: run once with a null base to learn the size, once more to place the buffers
size_t carve(Carver& C, const Sizes& z, WS& W) {
    const int64_t G = z.G, N = z.N, S = z.S, Ealloc = std::max<int64_t>(z.Eeff, 1), BT = z.bucket_total;
    W.scal.p = C.take<Scalars>(1);
    W.results.p = C.take<char>((int64_t)z.res_bytes);
    {   // what the host hands over per call, one block (one H2D copy from the pinned staging buffer): node_ptr | surv_base | gd
        char* hp = C.take<char>((int64_t)host_block_bytes(G));
        W.node_ptr_d.p = hp;
        W.surv_base_d.p = hp ? hp + 8 * (G + 1) : nullptr;
        W.gd_d.p = hp ? hp + 16 * (G + 1) : nullptr;
    }
    W.pool_top.p = C.take<int32_t>(1);
    W.bs_pool_top.p = C.take<int32_t>(1);
    W.vgraph.p = C.take<int32_t>(N);
    W.keys0.p = C.take<uint64_t>(Ealloc); W.keys1.p = C.take<uint64_t>(Ealloc);
    W.idx0.p = C.take<uint32_t>(Ealloc); W.idx1.p = C.take<uint32_t>(Ealloc);
    W.head.p = C.take<int32_t>(Ealloc + 1); W.pos.p = C.take<int32_t>(Ealloc + 1);
    W.sorttmp.p = C.take<char>((int64_t)z.sort_tmp);
    W.ent.p = C.take<Slot>(z.slot_cap);
    W.slot_col.p = C.take<int32_t>(Ealloc);
    W.colptr.p = C.take<int32_t>(N + 1);
    W.vrec.p = C.take<VRec>(N);
    W.origpos.p = C.take<int32_t>(N);
    W.ocur.p = C.take<int32_t>(BT); W.oend.p = C.take<int32_t>(BT);
    W.bs_cnt.p = C.take<int32_t>(BT); W.bs_alloc.p = C.take<int32_t>(BT); W.bs_dir.p = C.take<int32_t>(BT * BDIR);
    W.bs_v.p = C.take<int32_t>(z.log_total); W.bs_id.p = C.take<int32_t>(z.log_total);
    W.batch_pos.p = C.take<int32_t>(N);
    W.skey0.p = C.take<uint64_t>(N); W.skey1.p = C.take<uint64_t>(N); W.sval0.p = C.take<uint32_t>(N); W.sval1.p = C.take<uint32_t>(N);
    W.orig_order.p = C.take<int32_t>(N);
    W.permchk.p = C.take<int32_t>(N);
    W.genperm.p = C.take<int64_t>(z.want_genperm ? N : 0);
    W.scr_rec.p = C.take<SRec>(z.scr_total); W.scr_i32.p = C.take<int32_t>(10 * z.scr_total); W.scr_f64.p = C.take<double>(4 * z.scr_total);
    W.ext.p = C.take<int32_t>(S + 1); W.cnt.p = C.take<int32_t>(S + 1); W.tmp_off.p = C.take<int64_t>(S + 1); W.row_off.p = C.take<int64_t>(S + 1);
    W.tmp_nbr.p = C.take<int32_t>(z.slot_cap); W.tmp_val.p = C.take<double>(z.slot_cap);   // staged rows: at most one per slot in use
    W.sc_rec.p = C.take<SRec>(z.scr_budget); W.sc_i32.p = C.take<int32_t>(7 * z.scr_budget); W.sc_f64.p = C.take<double>(2 * z.scr_budget);
    W.biglist.p = C.take<int32_t>(8 * (S + 1));
    W.hugelists.p = C.take<uint16_t>((int64_t)NHUGE * 2 * (HUGECAP + 2));
    if (z.flow) {
        W.f_ctrl.p = C.take<int32_t>(FC_WORDS);
        W.f_cdir.p = C.take<int32_t>(N * FDIR);
        W.f_atag.p = C.take<int32_t>(z.slot_cap);
        W.f_lb.p = C.take<unsigned long long>(z.flow_Q); W.f_qv.p = C.take<int32_t>(z.flow_Q); W.f_qg.p = C.take<int32_t>(z.flow_Q);
        W.f_scr.p = C.take<char>(z.flow_scr * FLOW_SCR_BYTES);
        W.f_rocnt.p = C.take<int32_t>(S + 1); W.f_rooff.p = C.take<int32_t>(S + 1);
    }
    return (C.off + 255) & ~(size_t)255;
}

// sizes of a call on (E directed input entries, N vertices, G graphs, S surviving vertices) under the handle's growth factors
int call_sizes(const rlap_handle h, int64_t Eeff, int64_t N, int64_t G, int64_t bucket_total, int64_t S, bool want_genperm, bool flow, int64_t nelim_total, Sizes* z) {
    const double pool_factor = h->dbg_pool >= 0 ? h->dbg_pool : h->pool_factor;
    const double log_factor = h->dbg_log >= 0 ? h->dbg_log : h->log_factor;
    const int64_t nnz_ub = Eeff;
    const int64_t pool = (int64_t)(pool_factor * nnz_ub) + (h->dbg_pool >= 0 ? 0 : 16 * N + 1024) + G * (G <= 512 ? (int64_t)POOL_GRAB_BIG : (int64_t)POOL_GRAB_SMALL);
    z->G = G; z->N = N; z->Eeff = Eeff; z->S = S; z->bucket_total = bucket_total;
    z->slot_cap = nnz_ub + pool;
    if (z->slot_cap >= ((int64_t)1 << 31) - 64) return RLAP_E_TOO_LARGE;
    z->log_total = (int64_t)(log_factor * nnz_ub) + (h->dbg_log >= 0 ? 64 : (int64_t)BCH0 * bucket_total + 64 * G);
    z->scr_total = nnz_ub / 2 + 8 * G + 8;
    if (bucket_total * BDIR >= ((int64_t)1 << 40) || bucket_total >= ((int64_t)1 << 31) || z->log_total >= ((int64_t)1 << 31)) return RLAP_E_TOO_LARGE;
    z->scr_budget = h->dbg_scr >= 0 ? h->dbg_scr : h->scr_budget;
    z->res_bytes = sizeof(CallResults) + 8 * (size_t)(G + 1);
    z->want_genperm = want_genperm;
    z->flow = flow;
    z->flow_Q = nelim_total + G;
    z->flow_scr = flow ? (int64_t)(h->flow_scr_factor * (double)std::max<int64_t>(Eeff / 16, 1 << 17)) : 0;
    if (flow && h->dbg_scr >= 0) z->flow_scr = std::max<int64_t>(h->dbg_scr, 16);   // (test hook: a tiny first size, so that the retry runs)
    if (z->flow_scr >= ((int64_t)1 << 31) - 64) z->flow_scr = ((int64_t)1 << 31) - 64;
    return sort_tmp_bytes(Eeff, N, G, S, flow, &z->sort_tmp);
}

// uniforms a call may draw: a graph rarely draws more than its own directed entry count; an overflow doubles the table
int64_t rng_guess(const rlap_handle h, int64_t Eeff, int64_t G) {
    int64_t guess = (G == 1) ? Eeff + 1024 : std::min<int64_t>(Eeff + 1024, 4 * (Eeff / G) + 65536);
    return std::max<int64_t>(std::max(guess, h->rng_min), 1 << 16);
}

struct Call {
    const int64_t* d_row; const int64_t* d_col; const double* d_w; int64_t E;
    int64_t G; const int64_t* h_node_ptr; const int64_t* h_t;
    int o_v, o_n; const int64_t* d_perm; uint64_t seed;
    double* d_out; int64_t out_cap; int64_t* h_out_row_ptr; rlap_stats* st;
    int symmetrize;
};

// One attempt.  No host synchronisation until the single read-back at the end: every size is an upper bound
// known on the host (E, N, G); what only the device knows (nnz, row counts, overflows, input checks) is read
// by the kernels from device memory and reported once.
int run_once(rlap_handle h, const Call& c, int* retry_kind, int64_t* retry_need) {
    *retry_kind = 0; *retry_need = 0;
    hipStream_t s = h->stream;
    const int64_t G = c.G, E = c.E;
    const int64_t N = c.h_node_ptr[G];
    if (c.h_node_ptr[0] != 0) return RLAP_E_BAD_ARG;
    for (int64_t g = 0; g < G; ++g) if (c.h_node_ptr[g + 1] < c.h_node_ptr[g]) return RLAP_E_BAD_ARG;
    const int64_t Eeff = c.symmetrize ? 2 * E : E;
    if (N >= (int64_t)1 << 30 || Eeff >= (int64_t)1 << 31 || G >= (int64_t)1 << 30) return RLAP_E_TOO_LARGE;
    if (h->timing) HIPCHK(hipEventRecord(h->ev[0], s));

    rlap_stats st;
    std::memset(&st, 0, sizeof(st));

    if (N == 0) {
        for (int64_t g = 0; g <= G; ++g) c.h_out_row_ptr[g] = 0;
        if (c.st) *c.st = st;
        return RLAP_OK;
    }

    // ---------------- sizes (host-known upper bounds) ----------------
    const int64_t nnz_ub = Eeff;
    std::vector<GraphDesc> gd((size_t)G);
    int64_t bucket_total = 0, nelim_total = 0;
    std::vector<int64_t> surv_base((size_t)G + 1, 0);
    for (int64_t g = 0; g < G; ++g) {
        GraphDesc& D = gd[g];
        std::memset(&D, 0, sizeof(D));
        int64_t n = c.h_node_ptr[g + 1] - c.h_node_ptr[g];
        D.vbase = (int32_t)c.h_node_ptr[g]; D.n = (int32_t)n; D.t = c.h_t[g];
        D.bucket_base = (int32_t)bucket_total; bucket_total += 2 * n + 1;
        int64_t ne = std::max<int64_t>(0, std::min<int64_t>(c.h_t[g], n - 1));
        D.n_elim = (int32_t)ne;
        D.flow_base = (int32_t)(nelim_total + g);
        nelim_total += ne;
        surv_base[g + 1] = surv_base[g] + (n - ne);
    }
    const int64_t S = surv_base[G];
    st.n_eliminated = nelim_total;
    Sizes z;
    const bool flow = flow_wanted(h, c.o_v, G, N);
    { int rc = call_sizes(h, Eeff, N, G, bucket_total, S, c.o_v == OV_RANDOM && !c.d_perm, flow, nelim_total, &z); if (rc) return rc; }
    const int64_t slot_cap = z.slot_cap, log_total = z.log_total, scr_total = z.scr_total, scr_budget = z.scr_budget;
    const size_t res_bytes = z.res_bytes;

    // ---------------- workspace: one arena, no allocation unless this call is of a larger size class than any before ----------------
    WS W;
    {
        Carver dry{nullptr, 0};
        const size_t need = carve(dry, z, W);
        h->ws_needed = need;
        char* base = nullptr;
        if (h->ext_ws) {
            if (need > h->ext_ws_bytes) return RLAP_E_WORKSPACE;   // the caller allocates rlap_workspace_needed() bytes and calls again
            base = static_cast<char*>(h->ext_ws);
        } else {
            ENSURE(h->own_ws, need);
            base = h->own_ws.as<char>();
        }
        Carver real{base, 0};
        carve(real, z, W);
        if (h->poison >= 0) {   // debug: whatever a kernel reads without having been given it is this byte, not the previous call's data
            HIPCHK(hipMemsetAsync(base, h->poison, need, s));
            if (c.d_out && c.out_cap > 0) HIPCHK(hipMemsetAsync(c.d_out, h->poison, sizeof(double) * 3 * (size_t)c.out_cap, s));
        }
    }
    {
        const size_t pinned_need = ((res_bytes + 255) & ~(size_t)255) + host_block_bytes(G);   // read-back block + staging of the host parameters
        if (pinned_need > h->h_results_cap) {
            if (h->h_results) (void)hipHostFree(h->h_results);
            h->h_results = nullptr; h->h_results_cap = 0;
            HIPCHK(hipHostMalloc(&h->h_results, pinned_need + 65536, hipHostMallocDefault));
            h->h_results_cap = pinned_need + 65536;
        }
    }
    { int rc = ensure_rng(h, rng_guess(h, Eeff, G)); if (rc) return rc; }
    const SortTmp ST{W.sorttmp.p, z.sort_tmp};
    // (test hook: the kernels are told the table is shorter than it is)
    const int64_t rng_len_eff = h->dbg_rng >= 0 ? std::min<int64_t>(h->rng_len, h->dbg_rng) : h->rng_len;

    Scalars* SC = W.scal.as<Scalars>();
    int32_t* flags = SC->flags;
    double* acc = SC->acc;
    unsigned long long* counters = SC->counters;   // [0] sc scratch top
    unsigned long long* live = SC->live;
    int32_t* nnz_p = &SC->nnz;

    // ---------------- setup: COO -> CSR ----------------
    {
        // node_ptr | surv_base | gd: staged in pinned host memory (behind the read-back block), ONE asynchronous copy -- three copies
        // from pageable memory were three blocking staging copies inside the runtime
        char* stage = static_cast<char*>(h->h_results) + ((res_bytes + 255) & ~(size_t)255);
        std::memcpy(stage, c.h_node_ptr, 8 * (size_t)(G + 1));
        std::memcpy(stage + 8 * (G + 1), surv_base.data(), 8 * (size_t)(G + 1));
        std::memcpy(stage + 16 * (G + 1), gd.data(), sizeof(GraphDesc) * (size_t)G);
        HIPCHK(hipMemcpyAsync(W.node_ptr_d.p, stage, host_block_bytes(G), hipMemcpyHostToDevice, s));
    }
    {
        Fills F;
        F.add(W.scal.p, sizeof(Scalars) / 4, 0);
        if (c.o_v != OV_RANDOM) {   // the bucket stacks and the never-moved lists: PQ orders only
            F.add(W.bs_cnt.p, bucket_total, 0); F.add(W.bs_alloc.p, bucket_total, 0);
            F.add(W.ocur.p, bucket_total, 0); F.add(W.oend.p, bucket_total, 0);
        }
        F.add(W.bs_pool_top.p, 1, 0);
        F.add(W.batch_pos.p, N, -1);
        F.add(W.ext.as<int32_t>() + S, 1, 0); F.add(W.cnt.as<int32_t>() + S, 1, 0);
        if (G == 1) F.add(W.vgraph.p, N, 0);
        F.launch(s);
    }
    if (G > 1) hipLaunchKernelGGL(k_vertex_graph, dim3(nblk(N, 256)), dim3(256), 0, s, W.node_ptr_d.as<int64_t>(), (int)G, W.vgraph.as<int32_t>(), N);

    const int kbits = (int)bits_for((uint64_t)(N > 1 ? N - 1 : 1));
    bool swapped = false;   // the input was in (row, col) order and is read transposed (same matrix when it is exactly symmetric)
    uint64_t* keys_sorted = W.keys1.as<uint64_t>();
    uint32_t* idx_sorted = W.idx1.as<uint32_t>();
    uint32_t* rowid = W.idx0.as<uint32_t>();   // dense row ids of the slots (key of the twin sort); a buffer the COO sort has left free
    if (Eeff > 0) {
        hipLaunchKernelGGL(k_edge_keys, dim3(nblk(Eeff, 256)), dim3(256), 0, s, c.d_row, c.d_col, c.d_w, E, N, G > 1 ? W.vgraph.as<int32_t>() : (const int32_t*)nullptr,
                           c.symmetrize, kbits, W.keys0.as<uint64_t>(), W.idx0.as<uint32_t>(), flags);
        // Large inputs: one early look at the order flags (the call's only other host synchronisation) -- a COO that is sorted by
        // (col, row) needs no sort, one sorted by (row, col) (PyG coalesce) is read transposed and needs none either; exact symmetry,
        // which makes the transposed reading the same matrix, is verified with the twins (else the call is repeated with the sort).
        bool skip_sort = false;
        int64_t skip_min = SORT_SKIP_MIN;
        if (const char* e = std::getenv("RLAP_SORT_SKIP_MIN")) skip_min = std::atoll(e);   // (tests: exercise the order check on small inputs)
        if (Eeff >= skip_min && !c.symmetrize && !h->force_sort) {
            int32_t fl[FLAG_COUNT];
            HIPCHK(hipMemcpyAsync(fl, flags, sizeof(fl), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (!fl[FLAG_UNSORTED_CR]) skip_sort = true;
            else if (!fl[FLAG_UNSORTED_RC] && !fl[FLAG_RANGE]) {
                hipLaunchKernelGGL(k_edge_keys, dim3(nblk(Eeff, 256)), dim3(256), 0, s, c.d_col, c.d_row, c.d_w, E, N, G > 1 ? W.vgraph.as<int32_t>() : (const int32_t*)nullptr,
                                   0, kbits, W.keys0.as<uint64_t>(), W.idx0.as<uint32_t>(), flags);
                skip_sort = true; swapped = true;
            }
        }
        st.reserved = skip_sort ? (swapped ? 2 : 1) : 0;
        if (skip_sort) {
            keys_sorted = W.keys0.as<uint64_t>(); idx_sorted = W.idx0.as<uint32_t>();
            rowid = W.idx1.as<uint32_t>();
        } else {
            // keys are (col << kbits | row) with ids < N, or all ones for dropped entries (bit 2 * kbits makes these sort last)
            int rc = sort_pairs(h, ST, W.keys0.as<uint64_t>(), W.keys1.as<uint64_t>(), W.idx0.as<uint32_t>(), W.idx1.as<uint32_t>(), Eeff, 0,
                                (unsigned)std::min(64, 2 * kbits + 1));
            if (rc) return rc;
        }
        { Fills F2; F2.add(rowid, Eeff, (int32_t)N); F2.launch(s); }   // dense row-id array: slots beyond nnz hold N (sorts last)
        hipLaunchKernelGGL(k_heads, dim3(nblk(Eeff + 1, 256)), dim3(256), 0, s, keys_sorted, Eeff, W.head.as<int32_t>());
        int rc = excl_scan(h, ST, W.head.as<int32_t>(), W.pos.as<int32_t>(), Eeff + 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_fill_csr, dim3(nblk(Eeff, 256)), dim3(256), 0, s, keys_sorted, idx_sorted, W.head.as<int32_t>(),
                           W.pos.as<int32_t>(), c.d_w, Eeff, (c.symmetrize && !c.d_w) ? 1 : 0, kbits, W.ent.as<Slot>(), W.slot_col.as<int32_t>(),
                           reinterpret_cast<int32_t*>(rowid));
    }
    // (nnz = pos[Eeff]: k_colptr reads it there and files it in the scalar block and as the pool's first free slot)
    hipLaunchKernelGGL(k_colptr, dim3(nblk(N + 1, 256)), dim3(256), 0, s, W.slot_col.as<int32_t>(), Eeff > 0 ? W.pos.as<int32_t>() + Eeff : (const int32_t*)&SC->pad, (int32_t)N,   // (no entries: a zeroed word of its own, not the output word -- both are __restrict__)
                       W.colptr.as<int32_t>(), nnz_p, W.pool_top.as<int32_t>());
    if (Eeff > 0) {
        // twins: stable sort of the slots by row id (keys0 is free again: sorted keys in its first half, the slot order T in its second)
        uint32_t* skeys = W.keys0.as<uint32_t>();
        uint32_t* T = skeys + Eeff;
        rocprim::counting_iterator<uint32_t> iota(0u);
        size_t bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, rowid, skeys, iota, T, (size_t)Eeff, 0u, (unsigned)std::min(32, kbits + 1), s));
        if (bytes > ST.bytes) return RLAP_E_INTERNAL;
        bytes = ST.bytes;
        HIPCHK(rocprim::radix_sort_pairs(ST.p, bytes, rowid, skeys, iota, T, (size_t)Eeff, 0u, (unsigned)std::min(32, kbits + 1), s));
        hipLaunchKernelGGL(k_twin_sorted, dim3((unsigned)std::min<int64_t>(nblk(Eeff, 256), 4096)), dim3(256), 0, s, W.ent.as<Slot>(), T, nnz_p, acc);
        hipLaunchKernelGGL(k_twin_store, dim3((unsigned)std::min<int64_t>(nblk(Eeff, 256), 4096)), dim3(256), 0, s, W.ent.as<Slot>(), T, nnz_p);
    }
    // per-graph scratch of the long-column fall-backs (needs the per-graph nnz: filled on the device)
    hipLaunchKernelGGL(k_gd_scratch, dim3(1), dim3(256), 0, s, W.colptr.as<int32_t>(), W.node_ptr_d.as<int64_t>(), (int32_t)G, W.gd_d.as<GraphDesc>());

    // ---------------- PQ init ----------------
    const unsigned gbits = bits_for((uint64_t)(G > 0 ? G - 1 : 0));
    hipLaunchKernelGGL(k_pq_init, dim3(nblk(N, 256)), dim3(256), 0, s, W.colptr.as<int32_t>(), W.vgraph.as<int32_t>(), (int32_t)N, W.vrec.as<VRec>(),
                       W.skey0.as<uint64_t>(), W.sval0.as<uint32_t>());
    if (c.o_v != OV_RANDOM) {   // (o_v = random has no queue: k_pq_init's vertex records are all it needs)
        // key = graph << 32 | degree: only the bits that can be set take part in the sort
        int rc = sort_pairs(h, ST, W.skey0.as<uint64_t>(), W.skey1.as<uint64_t>(), W.sval0.as<uint32_t>(), W.orig_order.as<uint32_t>(), N, 0, 32 + gbits);
        if (rc) return rc;
        hipLaunchKernelGGL(k_bucket_bounds, dim3(nblk(N, 256)), dim3(256), 0, s, W.orig_order.as<uint32_t>(), W.vrec.as<VRec>(), W.vgraph.as<int32_t>(),
                           W.gd_d.as<GraphDesc>(), (int32_t)N, W.ocur.as<int32_t>(), W.oend.as<int32_t>(), W.origpos.as<int32_t>());
    }

    // ---------------- o_v = random: the node_id vector ----------------
    const int64_t* d_perm = c.d_perm;
    if (c.o_v == OV_RANDOM) {
        if (!d_perm) {
            hipLaunchKernelGGL(k_perm_keys, dim3(nblk(N, 256)), dim3(256), 0, s, W.vgraph.as<int32_t>(), W.gd_d.as<GraphDesc>(), (int32_t)N, c.seed,
                               G > 1 ? (int)gbits : 0, W.skey0.as<uint64_t>(), W.sval0.as<uint32_t>());
            int rc = sort_pairs(h, ST, W.skey0.as<uint64_t>(), W.skey1.as<uint64_t>(), W.sval0.as<uint32_t>(), W.sval1.as<uint32_t>(), N, 0, 64);
            if (rc) return rc;
            hipLaunchKernelGGL(k_widen_u32, dim3(nblk(N, 256)), dim3(256), 0, s, W.sval1.as<uint32_t>(), (int32_t)N, W.genperm.as<int64_t>());
            d_perm = W.genperm.as<int64_t>();
        } else {
            HIPCHK(hipMemsetAsync(W.permchk.p, 0, 4 * N, s));
            hipLaunchKernelGGL(k_perm_check, dim3(nblk(N, 256)), dim3(256), 0, s, d_perm, W.vgraph.as<int32_t>(), W.gd_d.as<GraphDesc>(), (int32_t)N,
                               W.permchk.as<int32_t>(), flags);
        }
    }

    Arrays A;
    A.colptr = W.colptr.as<int32_t>();
    A.e = W.ent.as<Slot>();
    A.slot_cap = (int32_t)slot_cap; A.pool_top = W.pool_top.as<int32_t>();
    A.vr = W.vrec.as<VRec>();
    A.ocur = W.ocur.as<int32_t>(); A.oend = W.oend.as<int32_t>();
    A.orig_order = W.orig_order.as<int32_t>();
    A.bs_cnt = W.bs_cnt.as<int32_t>(); A.bs_alloc = W.bs_alloc.as<int32_t>(); A.bs_dir = W.bs_dir.as<int32_t>();
    A.bs_v = W.bs_v.as<int32_t>(); A.bs_id = W.bs_id.as<int32_t>();
    A.bs_pool_top = W.bs_pool_top.as<int32_t>(); A.bs_pool_cap = (int32_t)log_total;
    A.rng = h->rng_ptr; A.rng_len = rng_len_eff;
    A.perm = d_perm; A.o_v = c.o_v; A.o_n = c.o_n; A.shuffle_seed = c.seed;
    A.rng_mode = h->rng_mode ? RNG_COUNTER : RNG_STREAM; A.vbase = 0;
    if (A.rng_mode == RNG_COUNTER) A.rng_len = (int64_t)1 << 62;   // (no table is read: the stream-length checks of the kernels never fire)
    ElimScratch ES;
    ES.rec = W.scr_rec.as<SRec>(); ES.i32 = W.scr_i32.as<int32_t>(); ES.f64 = W.scr_f64.as<double>(); ES.cap = scr_total;
    ES.eqtab = h->eqtab.as<uint8_t>();
    ES.prof = nullptr;
    const char* prof_env = std::getenv("RLAP_PHASE_PROFILE");   // diagnostic only: per-phase clock sums of graph 0
    if (prof_env && prof_env[0] == '1' && h->prof.p) {
        HIPCHK(hipMemsetAsync(h->prof.p, 0, 8 * 40, s));
        ES.prof = h->prof.as<long long>();
    }
    ES.poison = h->poison;
    ES.jitter = h->jitter;

    if (h->timing) HIPCHK(hipEventRecord(h->ev[1], s));
    // ---------------- elimination ----------------
    // 128-slot candidates for o_v = random when the columns run long: 12 or more entries per vertex on average (from the input
    // sizes: no read-back).  Measured (ms per call, 64 -> 128 slots): BA(1M,10) 2950 -> 2180, BA(169343,7) 260 -> 248, BA(4096,8)
    // 10.2 -> 9.2; BA(20000,5) 23.0 -> 23.4 and sparser graphs lose (their rounds fill 64 candidates and get only 32)
    const bool wide = c.o_v == OV_RANDOM && nnz_ub >= 12 * N && N >= 512;
    FlowArrays FA;
    std::memset(&FA, 0, sizeof(FA));
    if (flow) {
        // dataflow elimination (rlap_flow.hip): every position of the order is one wave's, on any compute unit
        FA.cdir = W.f_cdir.as<int32_t>(); FA.atag = W.f_atag.as<int32_t>(); FA.lb = W.f_lb.as<unsigned long long>();
        FA.qv = W.f_qv.as<int32_t>(); FA.qg = W.f_qg.as<int32_t>(); FA.ctrl = W.f_ctrl.as<int32_t>(); FA.Q = (int32_t)z.flow_Q;
        FlowParams FP;
        FP.vgraph = W.vgraph.as<int32_t>(); FP.gd = W.gd_d.as<GraphDesc>(); FP.in_flags = flags; FP.in_acc = acc;
        FP.scr = W.f_scr.as<char>(); FP.scr_entries = (int32_t)z.flow_scr;
        FP.spin_limit = 4000; FP.jitter = h->jitter; FP.poison = h->poison;
        FP.prof = ES.prof;
        FP.trace = nullptr;
        const char* trace_env = std::getenv("RLAP_FLOW_TRACE");   // diagnostic only: per-position time stamps, written to the named file after the call
        if (trace_env && trace_env[0]) {
            ENSURE(h->trace, (size_t)z.flow_Q * 48);
            HIPCHK(hipMemsetAsync(h->trace.p, 0, (size_t)z.flow_Q * 48, s));
            FP.trace = h->trace.as<long long>();
        }
        if (const char* e = std::getenv("RLAP_FLOW_STALL_MS")) FP.spin_limit = std::max(1, std::atoi(e));
        HIPCHK(hipMemsetAsync(W.f_ctrl.p, 0, FC_WORDS * 4, s));
        FA.qorder = nullptr; FA.Qclaim = FA.Q;
        const bool interleave = G > 1 && nelim_total > 0;   // a batch: claims run over the graphs' positions interleaved, so that every graph has waves
        launch_flow_setup(s, A, FA, FP, W.slot_col.as<int32_t>(), nnz_p, (int32_t)N, Eeff, interleave ? W.skey0.as<uint64_t>() : nullptr,
                          interleave ? W.sval0.as<uint32_t>() : nullptr, (int)gbits);
        if (interleave) {
            int rc = sort_pairs(h, ST, W.skey0.as<uint64_t>(), W.skey1.as<uint64_t>(), W.sval0.as<uint32_t>(), W.sval1.as<uint32_t>(), nelim_total, 0, std::min(64u, 32u + gbits));
            if (rc) return rc;
            FA.qorder = W.sval1.as<uint32_t>(); FA.Qclaim = (int32_t)nelim_total;
        }
        int shape = G >= 48 ? 3 : 1;   // 1: 76 KB of LDS, two workgroups per CU, four waves each (three help with long sorts); 4: the same with one wave; 3: 40 KB, four; 2: 17 KB, eight
        if (const char* e = std::getenv("RLAP_FLOW_SHAPE")) { if (e[0] >= '1' && e[0] <= '4') shape = e[0] - '0'; }   // diagnostic override
        unsigned grid = (shape == 2 ? 8u : shape == 3 ? 4u : 2u) * (unsigned)h->n_cu;   // (workgroups: one position each)
        if (const char* e = std::getenv("RLAP_FLOW_WAVES")) grid = (unsigned)std::max(1, std::atoi(e));
        launch_flow_eliminate(shape, grid, s, A, FA, FP);
        launch_flow_finish(s, A, FA, W.gd_d.as<GraphDesc>(), (int32_t)N, (int32_t)G);
    } else
    launch_eliminate_batch(c.o_v, c.o_n, (unsigned)G, h->n_cu, s, A, W.gd_d.as<GraphDesc>(), ES, W.batch_pos.as<int32_t>(), flags, acc, wide);
    HIPCHK(hipGetLastError());
    if (h->timing) HIPCHK(hipEventRecord(h->ev[2], s));

    // ---------------- output ----------------
    uint32_t* order = nullptr;
    if (c.o_v == OV_RANDOM) {
        hipLaunchKernelGGL(k_sc_perm_order, dim3(nblk(N, 256)), dim3(256), 0, s, d_perm, W.vgraph.as<int32_t>(), W.gd_d.as<GraphDesc>(),
                           W.surv_base_d.as<int64_t>(), (int32_t)N, W.sval1.as<uint32_t>());
        order = W.sval1.as<uint32_t>();
        if (flow && S > 0 && Eeff > 0) {
            // tag order of the surviving columns: concurrent eliminations pushed into them out of order (rlap_flow.h)
            const int32_t ro_cap = (int32_t)Eeff;
            launch_flow_ro_count(s, A, FA, order, (int32_t)S, W.f_rocnt.as<int32_t>());
            { int rc = excl_scan(h, ST, W.f_rocnt.as<int32_t>(), W.f_rooff.as<int32_t>(), S + 1); if (rc) return rc; }
            uint32_t* home = reinterpret_cast<uint32_t*>(W.head.p);
            launch_flow_ro_emit(s, A, FA, order, (int32_t)S, W.f_rooff.as<int32_t>(), W.keys0.as<uint64_t>(), W.idx0.as<uint32_t>(), home, ro_cap, &W.gd_d.as<GraphDesc>()[0].status);
            const unsigned kb = bits_for((uint64_t)std::max<int64_t>(S - 1, 1)) + 32u;
            { int rc = sort_pairs(h, ST, W.keys0.as<uint64_t>(), W.keys1.as<uint64_t>(), W.idx0.as<uint32_t>(), W.idx1.as<uint32_t>(), Eeff, 0, std::min(64u, kb)); if (rc) return rc; }
            launch_flow_ro_permute(s, A, FA, W.f_rooff.as<int32_t>(), (int32_t)S, W.idx1.as<uint32_t>(), W.keys1.as<uint64_t>(), home, ro_cap, reinterpret_cast<Slot*>(W.tmp_val.p));
            HIPCHK(hipGetLastError());
        }
    } else {
        hipLaunchKernelGGL(k_sc_keys, dim3(nblk(N, 256)), dim3(256), 0, s, W.vrec.as<VRec>(), W.origpos.as<int32_t>(),
                           W.vgraph.as<int32_t>(), W.gd_d.as<GraphDesc>(), (int32_t)N, W.skey0.as<uint64_t>(), W.sval0.as<uint32_t>());
        // key = bucket << 32 | order, all ones for eliminated vertices
        int rc = sort_pairs(h, ST, W.skey0.as<uint64_t>(), W.skey1.as<uint64_t>(), W.sval0.as<uint32_t>(), W.sval1.as<uint32_t>(), N, 0, 64);
        if (rc) return rc;
        order = W.sval1.as<uint32_t>();
    }
    if (S > 0)
        hipLaunchKernelGGL(k_sc_ext, dim3(nblk(S, 256)), dim3(256), 0, s, order, W.colptr.as<int32_t>(), W.vrec.as<VRec>(), (int32_t)S, flags, acc, W.ext.as<int32_t>());
    { int rc = excl_scan(h, ST, W.ext.as<int32_t>(), W.tmp_off.as<int64_t>(), S + 1); if (rc) return rc; }
    ScScratch SS;
    SS.rec = W.sc_rec.as<SRec>(); SS.i32 = W.sc_i32.as<int32_t>(); SS.f64 = W.sc_f64.as<double>(); SS.cap = scr_budget; SS.top = counters + 0; SS.flags = flags;
    if (h->timing) HIPCHK(hipEventRecord(h->ev[3], s));
    if (S > 0) {
        int32_t* tiercounts = reinterpret_cast<int32_t*>(counters + 4);   // 8 ints: tiers 0..7
        if (!h->big_attr_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_merge_big), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_BYTES));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_merge_mw<8>), hipFuncAttributeMaxDynamicSharedMemorySize, MW_BIG_LDS_BYTES));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_merge_mw<4>), hipFuncAttributeMaxDynamicSharedMemorySize, MW_MID_LDS_BYTES));
            h->big_attr_set = true;
        }
        ScLaunch X;
        X.main = s; X.side[0] = h->side[0]; X.side[1] = h->side[1];
        for (int q = 0; q < 3; ++q) X.ev[q] = h->fork_ev[q];
        launch_sc_merge(X, A, W.gd_d.as<GraphDesc>(), W.vgraph.as<int32_t>(), order, W.ext.as<int32_t>(), W.tmp_off.as<int64_t>(), (int32_t)S,
                        W.tmp_nbr.as<int32_t>(), W.tmp_val.as<double>(), W.cnt.as<int32_t>(), SS, live, W.biglist.as<int32_t>(), tiercounts,
                        W.hugelists.as<uint16_t>());
        HIPCHK(hipGetLastError());
    }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[4], s));
    { int rc = excl_scan(h, ST, W.cnt.as<int32_t>(), W.row_off.as<int64_t>(), S + 1); if (rc) return rc; }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[5], s));
    if (S > 0 && c.out_cap > 0) {
        const int64_t rows_ub = std::min<int64_t>(c.out_cap, slot_cap);
        unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows_ub + 255) / 256, 256 * 8));
        hipLaunchKernelGGL(k_sc_compact, dim3(grid), dim3(256), 0, s, order, W.cnt.as<int32_t>(), W.row_off.as<int64_t>(), W.tmp_off.as<int64_t>(),
                           W.tmp_nbr.as<int32_t>(), W.tmp_val.as<double>(), (int32_t)S, c.d_out, c.out_cap);
        HIPCHK(hipGetLastError());
    }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[6], s));
    // ---------------- the one read-back ----------------
    CallResults* res_d = W.results.as<CallResults>();
    int64_t* out_ptr_d = reinterpret_cast<int64_t*>(res_d + 1);
    hipLaunchKernelGGL(k_graph_rows, dim3(nblk(G + 1, 256)), dim3(256), 0, s, W.surv_base_d.as<int64_t>(), W.row_off.as<int64_t>(), (int32_t)G, out_ptr_d);
    hipLaunchKernelGGL(k_collect, dim3(1), dim3(256), 0, s, flags, acc, nnz_p, counters, live, W.tmp_off.as<int64_t>(), W.row_off.as<int64_t>(), (int32_t)S,
                       W.gd_d.as<GraphDesc>(), (int32_t)G, W.pool_top.as<int32_t>(), W.bs_pool_top.as<int32_t>(), res_d);
    HIPCHK(hipMemcpyAsync(h->h_results, W.results.p, res_bytes, hipMemcpyDeviceToHost, s));
    if (h->timing) HIPCHK(hipEventRecord(h->ev[7], s));
    HIPCHK(hipStreamSynchronize(s));
    const CallResults& R = *reinterpret_cast<const CallResults*>(h->h_results);
    const int64_t* out_ptr_h = reinterpret_cast<const int64_t*>(reinterpret_cast<const CallResults*>(h->h_results) + 1);

    if (flow) if (const char* trace_env = std::getenv("RLAP_FLOW_TRACE")) if (trace_env[0] && h->trace.p) {
        std::vector<long long> tr((size_t)z.flow_Q * 6);
        HIPCHK(hipMemcpy(tr.data(), h->trace.p, tr.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(trace_env, "wb")) { std::fwrite(tr.data(), 8, tr.size(), f); std::fclose(f); }
    }
    if (ES.prof && flow) {
        long long pr[40];
        HIPCHK(hipMemcpy(pr, h->prof.p, sizeof(pr), hipMemcpyDeviceToHost));
        const char* names[11] = {"claim + wait for pend == 0", "gather", "sort by id (+ tag order)", "merge + publish", "o_n order", "cumsum + recurrence", "look-back wait", "sample", "commit (claims, chunks, stores)", "drain", "decrements"};
        std::fprintf(stderr, "[rlap flow profile] positions=%lld mean live=%.1f mean extent=%.1f; sums over all waves, us per position\n", pr[16], pr[16] ? (double)pr[17] / pr[16] : 0.0, pr[16] ? (double)pr[18] / pr[16] : 0.0);
        for (int k = 0; k < 11; ++k) std::fprintf(stderr, "  %-34s %10.3f ms total  %8.2f us/pos\n", names[k], pr[k] / 1e5, pr[16] ? pr[k] / 100.0 / (double)pr[16] : 0.0);
        if (pr[12] || pr[15]) std::fprintf(stderr, "[rlap flow] RLAP_FLOW_MW_SORT build: %lld sorted index sets out of range; lanes affected with the prepared address %llx, with a fresh one %llx\n", pr[15], pr[13], pr[14]);
        std::fprintf(stderr, "  long columns: %lld (mean live %.0f, max %lld; %lld beyond 900, %lld beyond 1888, %lld beyond 3320 taking %.3f of %.3f ms)\n", pr[36], pr[36] ? (double)pr[37] / pr[36] : 0.0, pr[35], pr[38], pr[39], pr[32], pr[33] / 1e5, pr[34] / 1e5);
        for (int k = 1; k < 11; ++k) std::fprintf(stderr, "    %-32s %10.3f ms total  %8.2f us/col\n", names[k], pr[20 + k] / 1e5, pr[36] ? pr[20 + k] / 100.0 / (double)pr[36] : 0.0);
    } else if (ES.prof) {
        long long pr[40];
        HIPCHK(hipMemcpy(pr, h->prof.p, sizeof(pr), hipMemcpyDeviceToHost));
        const char* names[20] = {"P0 select", "P1 permute + closing barrier", "P1b multi-edge cut + P2 offsets", "P3 sample", "P4 replay", "single path", "P5b slots + shared targets", "move order", "pushes", "P1 meta + loads + dependence", "P5a cursors + pool", "P5c rewire stores", "P1 id rank", "P1 (sync)", "P1 o_n order (sorts)", "P1 write pass", "P1 (sync)", "P1 o_n rank", "empty (cost of one stamp)", "P3b dependent candidates (patch)"};
        std::fprintf(stderr, "[rlap phase profile] rounds=%lld singles=%lld (100 MHz ticks); shader clock over the kernel: %.0f MHz\n", pr[22], pr[23], pr[21] > 0 ? 100.0 * (double)pr[20] / (double)pr[21] : 0.0);
        for (int k = 0; k < 20; ++k) std::fprintf(stderr, "  %-22s %10.3f ms\n", names[k], pr[k] / 1e5);
        const char* wnames[9] = {"gather", "sort by id", "merge", "meta loads + order", "cumsum + recurrence", "sample", "replay + slots", "rewire", "pq commit"};
        for (int k = 0; k < 9; ++k) std::fprintf(stderr, "  single/wave: %-20s %10.3f ms\n", wnames[k], pr[24 + k] / 1e5);
        const char* xnames[5] = {"P4.1 hash insert + key loads", "P4.2 replay (uncontended)", "P4.3 sort of the contended records", "P4.4 contended walk", "P5b uncontended slots (rest of P5b = contended walk)"};
        for (int k = 0; k < 5; ++k) std::fprintf(stderr, "  detail: %-52s %10.3f ms\n", xnames[k], pr[33 + k] / 1e5);
    }

    st.nnz = R.nnz;
    st.n_draws = R.n_draws;
    st.n_rounds = R.rounds; st.n_singles = R.singles;
    st.out_rows = R.m_total;
    st.live_entries = R.live_total;
    if (h->timing) {
        float t = 0;
        (void)hipEventElapsedTime(&t, h->ev[0], h->ev[1]); st.ms_setup = t;
        (void)hipEventElapsedTime(&t, h->ev[1], h->ev[2]); st.ms_elim = t;
        (void)hipEventElapsedTime(&t, h->ev[2], h->ev[7]); st.ms_output = t;
        (void)hipEventElapsedTime(&t, h->ev[3], h->ev[4]); st.ms_sc_merge = t;
        (void)hipEventElapsedTime(&t, h->ev[5], h->ev[6]); st.ms_sc_compact = t;
        (void)hipEventElapsedTime(&t, h->ev[0], h->ev[7]); st.ms_total = t;
    }
    if (c.st) *c.st = st;
    // input errors first (the elimination was skipped), then growth limits (the call is repeated), then the output size
    if (R.status == ST_INDEX_RANGE || R.flags[FLAG_RANGE]) return RLAP_E_INDEX_RANGE;
    if (R.status == ST_BAD_ARG || R.flags[FLAG_CROSS] || R.flags[FLAG_PERM]) return RLAP_E_BAD_ARG;
    if (R.status == ST_NOT_SYMMETRIC) return RLAP_E_NOT_SYMMETRIC;
    if (R.status) {
        if (R.status == ST_POOL_OVERFLOW) *retry_kind = 1;
        else if (R.status == ST_LOG_OVERFLOW) *retry_kind = 2;
        else if (R.status == ST_RNG_OVERFLOW) *retry_kind = 3;
        else if (R.status == ST_FLOW_SCRATCH) *retry_kind = 6;
        else if (R.status == ST_FLOW_REORDER) *retry_kind = 7;
        else if (R.status == ST_INTERNAL && flow) *retry_kind = 7;   // the dataflow kernel gave up (stall watchdog, index check): once more on the round kernel
        return R.status;
    }
    if (R.flags[FLAG_SCR]) { *retry_kind = 4; *retry_need = R.scr_need; return RLAP_E_INTERNAL; }
    if (swapped && R.acc[0] != 0.0) { *retry_kind = 5; return RLAP_E_INTERNAL; }   // symmetric only within the tolerance: repeat, reading the input as given
    for (int64_t g = 0; g <= G; ++g) c.h_out_row_ptr[g] = out_ptr_h[g];
    if (R.m_total > c.out_cap) { c.h_out_row_ptr[G] = R.m_total; return RLAP_E_OUT_OVERFLOW; }
    return RLAP_OK;
}

struct DeviceGuard {
    int prev = -1; bool changed = false;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess); }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

int run_call(rlap_handle h, const Call& c) {
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device);
    int rc = RLAP_E_INTERNAL;
    int retries = 0;
    for (int attempt = 0; attempt < 10; ++attempt) {
        int kind = 0; int64_t need = 0;
        rc = run_once(h, c, &kind, &need);
        if (kind == 0) break;
        ++retries;
        // a test hook's tiny limit applies to the first attempt only; afterwards the regular growth takes over
        if (kind == 1) { if (h->dbg_pool >= 0) h->dbg_pool = -1.0; else h->pool_factor = h->pool_factor * 2 + 1; }
        else if (kind == 2) { if (h->dbg_log >= 0) h->dbg_log = -1.0; else h->log_factor *= 2; }
        else if (kind == 3) { if (h->dbg_rng >= 0) h->dbg_rng = -1; else h->rng_min = std::max<int64_t>(2 * h->rng_len, 1 << 16); }
        else if (kind == 4) { if (h->dbg_scr >= 0) h->dbg_scr = -1; h->scr_budget = std::max<int64_t>(h->scr_budget, need + 8); }
        else if (kind == 5) h->force_sort = true;
        else if (kind == 6) { if (h->dbg_scr >= 0) h->dbg_scr = -1; else h->flow_scr_factor *= 4; }
        else if (kind == 7) h->flow_off_once = true;
    }
    h->force_sort = false; h->flow_off_once = false;
    h->total_retries += retries;
    if (c.st) c.st->n_retries = retries;
    return rc;
}

}  // namespace

extern "C" {

// the body of rlap_create: any failure leaves a partly built handle that rlap_create hands to rlap_destroy
static int create_into(rlap_handle h) {
    for (auto& e : h->ev) HIPCHK(hipEventCreate(&e));
    for (auto& e : h->fork_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& st : h->side) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // everything a call needs besides its arena and the uniform table is allocated here, once: the equal-key permutation tables
    // (built on one of the handle's own non-blocking streams -- not the null stream, which would serialise against every blocking
    // stream of the caller -- and finished before rlap_create returns), the pinned read-back block, the diagnostic profile
    ENSURE(h->small, 256);
    ENSURE(h->eqtab, EQTAB_BYTES);
    launch_eq_tables(h->side[0], h->eqtab.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->side[0]));
    HIPCHK(hipHostMalloc(&h->h_results, 1 << 17, hipHostMallocDefault));
    h->h_results_cap = 1 << 17;
    if (const char* e = std::getenv("RLAP_PHASE_PROFILE")) { if (e[0] == '1') ENSURE(h->prof, 8 * 40); }
    if (const char* e = std::getenv("RLAP_DEBUG_POISON")) { if (e[0]) h->poison = std::atoi(e) & 0xFF; }
    if (const char* e = std::getenv("RLAP_DEBUG_JITTER")) { if (e[0]) h->jitter = std::max(0, std::min(64, std::atoi(e))); }
    return RLAP_OK;
}

int rlap_create(rlap_handle* out) {
    if (!out) return RLAP_E_BAD_ARG;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    rlap_handle h = new rlap_handle_s();
    h->device = dev;
    if (hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || h->n_cu <= 0) h->n_cu = 256;
    for (auto& e : h->ev) e = nullptr;
    for (auto& e : h->fork_ev) e = nullptr;
    const int rc = create_into(h);
    if (rc != RLAP_OK) { (void)rlap_destroy(h); return rc; }
    *out = h;
    return RLAP_OK;
}

int rlap_destroy(rlap_handle h) {
    if (!h) return RLAP_OK;
    DeviceGuard dg(h->device);
    if (h->h_results) (void)hipHostFree(h->h_results);
    h->own_ws.release(); h->own_rng.release(); h->eqtab.release(); h->prof.release(); h->trace.release(); h->small.release();
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : h->fork_ev) if (e) (void)hipEventDestroy(e);
    for (auto& st : h->side) if (st) (void)hipStreamDestroy(st);
    delete h;
    return RLAP_OK;
}

// ---- workspace contract ("caller owns every buffer"; the reference lets torch allocate, py_api_binder.cc:42) ----
static int ws_query(const rlap_handle h, int64_t E, int64_t n_total, int64_t G, int symmetrize, size_t* ws_bytes, int64_t* rng_entries) {
    if (E < 0 || n_total < 0 || G < 1) return RLAP_E_BAD_ARG;
    const int64_t Eeff = symmetrize ? 2 * E : E;
    if (n_total >= (int64_t)1 << 30 || Eeff >= (int64_t)1 << 31 || G >= (int64_t)1 << 30) return RLAP_E_TOO_LARGE;
    rlap_handle_s defaults;   // (growth factors of a fresh handle when none is given)
    const rlap_handle hh = h ? h : &defaults;
    Sizes z;
    // bounds that hold for every split of n_total over G graphs and every num_remove: 2n+1 buckets per graph, S <= n_total
    int rc = call_sizes(hh, Eeff, n_total, G, 2 * n_total + G, n_total, true, flow_wanted(hh, OV_RANDOM, G, n_total), n_total, &z);
    if (rc) return rc;
    WS W; Carver dry{nullptr, 0};
    *ws_bytes = carve(dry, z, W);
    *rng_entries = rng_guess(hh, Eeff, G);
    return RLAP_OK;
}

int rlap_workspace_bytes(int64_t E, int64_t n_total, int64_t G, int symmetrize, size_t* ws_bytes, int64_t* rng_entries) {
    if (!ws_bytes || !rng_entries) return RLAP_E_BAD_ARG;
    return ws_query(nullptr, E, n_total, G, symmetrize, ws_bytes, rng_entries);
}

int rlap_workspace_query(rlap_handle h, int64_t E, int64_t n_total, int64_t G, int symmetrize, size_t* ws_bytes, int64_t* rng_entries) {
    if (!h || !ws_bytes || !rng_entries) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return ws_query(h, E, n_total, G, symmetrize, ws_bytes, rng_entries);
}

int rlap_set_workspace(rlap_handle h, void* d_ws, size_t ws_bytes, double* d_rng, int64_t rng_entries) {
    if (!h || (d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 255)) || rng_entries < 0) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device);
    h->ext_ws = d_ws; h->ext_ws_bytes = d_ws ? ws_bytes : 0;
    if (d_ws) h->own_ws.release();
    if (d_rng != h->ext_rng || rng_entries != h->ext_rng_cap || (!d_rng && h->rng_ptr != h->own_rng.as<double>())) { h->rng_ptr = nullptr; h->rng_len = 0; }   // a table in another buffer is generated anew
    h->ext_rng = d_rng; h->ext_rng_cap = d_rng ? rng_entries : 0;
    if (d_rng) h->own_rng.release();
    return RLAP_OK;
}

int rlap_workspace_needed(rlap_handle h, size_t* ws_bytes, int64_t* rng_entries) {
    if (!h || !ws_bytes || !rng_entries) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    *ws_bytes = h->ws_needed; *rng_entries = h->rng_needed;
    return RLAP_OK;
}

int rlap_set_stream(rlap_handle h, void* hip_stream) {
    if (!h) return RLAP_E_BAD_ARG;
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RLAP_OK;
}

int rlap_set_rng_mode(rlap_handle h, int mode) {
    if (!h || (mode != 0 && mode != 1)) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->rng_mode = mode;
    return RLAP_OK;
}

int rlap_set_timing(rlap_handle h, int enable) {
    if (!h) return RLAP_E_BAD_ARG;
    h->timing = enable != 0;
    return RLAP_OK;
}

const char* rlap_status_string(int status) {
    switch (status) {
        case RLAP_OK: return "ok";
        case RLAP_E_NOT_SYMMETRIC: return "adjacency matrix is not symmetric";
        case RLAP_E_INDEX_RANGE: return "edge index out of range [0, num_nodes)";
        case RLAP_E_BAD_ARG: return "bad argument";
        case RLAP_E_POOL_OVERFLOW: return "append pool exhausted";
        case RLAP_E_LOG_OVERFLOW: return "priority-queue log exhausted";
        case RLAP_E_RNG_OVERFLOW: return "uniform table exhausted";
        case RLAP_E_OUT_OVERFLOW: return "output buffer too small";
        case RLAP_E_HIP: return "HIP runtime error";
        case RLAP_E_TOO_LARGE: return "problem exceeds int32 slot ids";
        case RLAP_E_WORKSPACE: return "caller-provided workspace too small (see rlap_workspace_needed)";
        default: return "internal error";
    }
}

int rlap_identity(rlap_handle h, const double* d_in, double* d_tmp, double* d_out, int64_t rows, int64_t cols) {
    if (!h || rows < 0 || cols < 0) return RLAP_E_BAD_ARG;
    int64_t total = rows * cols;
    if (total == 0) return RLAP_OK;
    hipLaunchKernelGGL(k_transpose_copy, dim3(nblk(total, 256)), dim3(256), 0, h->stream, d_in, d_tmp, rows, cols, 1);
    hipLaunchKernelGGL(k_transpose_copy, dim3(nblk(total, 256)), dim3(256), 0, h->stream, (const double*)d_tmp, d_out, rows, cols, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RLAP_OK;
}

int rlap_unpack_edge_info(rlap_handle h, const double* d_edge_info, int64_t E, int64_t* d_row, int64_t* d_col, double* d_w) {
    if (!h || E < 0) return RLAP_E_BAD_ARG;
    if (E == 0) return RLAP_OK;
    hipLaunchKernelGGL(k_unpack_edge_info, dim3(nblk(E, 256)), dim3(256), 0, h->stream, d_edge_info, E, d_row, d_col, d_w);
    HIPCHK(hipGetLastError());
    return RLAP_OK;
}

int rlap_pack_rows(rlap_handle h, const double* d_sc, int64_t rows, void* d_packed) {
    if (!h || rows < 0) return RLAP_E_BAD_ARG;
    if (rows == 0) return RLAP_OK;
    DeviceGuard dg(h->device);
    hipLaunchKernelGGL(k_pack_rows, dim3(nblk(rows, 256)), dim3(256), 0, h->stream, d_sc, rows, reinterpret_cast<unsigned long long*>(d_packed));
    HIPCHK(hipGetLastError());
    return RLAP_OK;
}

int rlap_unpack_rows(rlap_handle h, const void* d_packed, int64_t rows, double* d_sc) {
    if (!h || rows < 0) return RLAP_E_BAD_ARG;
    if (rows == 0) return RLAP_OK;
    DeviceGuard dg(h->device);
    hipLaunchKernelGGL(k_unpack_rows, dim3(nblk(rows, 256)), dim3(256), 0, h->stream, reinterpret_cast<const unsigned long long*>(d_packed), rows, d_sc);
    HIPCHK(hipGetLastError());
    return RLAP_OK;
}

int rlap_approx_chol_batched(rlap_handle h, const int64_t* d_row, const int64_t* d_col, const double* d_w, int64_t E, int64_t G,
                             const int64_t* h_node_ptr, const int64_t* h_num_remove, int o_v, int o_n, const int64_t* d_perm,
                             uint64_t shuffle_seed, double* d_out, int64_t out_cap_rows, int64_t* h_out_row_ptr, rlap_stats* h_stats) {
    if (!h || G < 1 || E < 0 || !h_node_ptr || !h_num_remove || !h_out_row_ptr) return RLAP_E_BAD_ARG;
    if (o_v < 0 || o_v > 2 || o_n < 0 || o_n > 2) return RLAP_E_BAD_ARG;
    if (E > 0 && (!d_row || !d_col)) return RLAP_E_BAD_ARG;
    if (out_cap_rows > 0 && !d_out) return RLAP_E_BAD_ARG;
    Call c{d_row, d_col, d_w, E, G, h_node_ptr, h_num_remove, o_v, o_n, d_perm, shuffle_seed, d_out, out_cap_rows, h_out_row_ptr, h_stats, 0};
    return run_call(h, c);
}

int rlap_approx_chol(rlap_handle h, const int64_t* d_row, const int64_t* d_col, const double* d_w, int64_t E, int64_t n, int64_t t,
                     int o_v, int o_n, const int64_t* d_perm, uint64_t shuffle_seed, double* d_out, int64_t out_cap_rows,
                     int64_t* h_out_rows, rlap_stats* h_stats) {
    if (n < 0 || !h_out_rows) return RLAP_E_BAD_ARG;
    int64_t node_ptr[2] = {0, n};
    int64_t tt[1] = {t};
    int64_t rp[2] = {0, 0};
    int rc = rlap_approx_chol_batched(h, d_row, d_col, d_w, E, 1, node_ptr, tt, o_v, o_n, d_perm, shuffle_seed, d_out, out_cap_rows, rp, h_stats);
    *h_out_rows = rp[1];
    return rc;
}

int rlap_approx_chol_from_edges(rlap_handle h, const int64_t* d_src, const int64_t* d_dst, const double* d_w, int64_t E, int64_t n, int64_t t,
                                double remove_frac, int symmetrize, int o_v, int o_n, const int64_t* d_perm, uint64_t shuffle_seed,
                                double* d_out, int64_t out_cap_rows, int64_t* h_out_rows, int64_t* h_num_nodes, rlap_stats* h_stats) {
    if (!h || E < 0 || !h_out_rows) return RLAP_E_BAD_ARG;
    if (o_v < 0 || o_v > 2 || o_n < 0 || o_n > 2) return RLAP_E_BAD_ARG;
    if (E > 0 && (!d_src || !d_dst)) return RLAP_E_BAD_ARG;
    if (out_cap_rows > 0 && !d_out) return RLAP_E_BAD_ARG;
    if (n < 0 && d_perm) return RLAP_E_BAD_ARG;   // an injected node_id vector needs a known num_nodes
    if (n < 0) {
        // num_nodes = edge_index.max() + 1 (augmentor_benchmarks.py:77): one reduction + one 8-byte read-back
        std::lock_guard<std::mutex> lock(h->mu);
        DeviceGuard dg(h->device);
        unsigned long long mx[2] = {0ull, 0ull};   // max id + 1, any id negative
        if (E > 0) {
            unsigned long long* d_mx = h->small.as<unsigned long long>();
            HIPCHK(hipMemsetAsync(d_mx, 0, 16, h->stream));
            hipLaunchKernelGGL(k_max_id, dim3((unsigned)std::min<int64_t>((E + 255) / 256, 1024)), dim3(256), 0, h->stream, d_src, d_dst, E, d_mx);
            HIPCHK(hipMemcpyAsync(mx, d_mx, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        if (mx[1]) return RLAP_E_INDEX_RANGE;   // a negative id is out of range for every num_nodes (an explicit n reports the same)
        n = (int64_t)mx[0];
    }
    if (t < 0) t = (int64_t)(remove_frac * (double)n);   // int(frac * num_nodes), augmentor_benchmarks.py:78
    if (h_num_nodes) *h_num_nodes = n;
    int64_t node_ptr[2] = {0, n};
    int64_t tt[1] = {t};
    int64_t rp[2] = {0, 0};
    Call c{d_src, d_dst, d_w, E, 1, node_ptr, tt, o_v, o_n, d_perm, shuffle_seed, d_out, out_cap_rows, rp, h_stats, symmetrize ? 1 : 0};
    int rc = run_call(h, c);
    *h_out_rows = rp[1];
    return rc;
}

int rlap_debug_set_limits(rlap_handle h, double pool_factor, double log_factor, int64_t rng_len, int64_t scratch_entries) {
    if (!h) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->dbg_pool = pool_factor; h->dbg_log = log_factor; h->dbg_rng = rng_len; h->dbg_scr = scratch_entries;
    return RLAP_OK;
}

int rlap_debug_set_poison(rlap_handle h, int byte) {
    if (!h) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->poison = byte < 0 ? -1 : (byte & 0xFF);
    return RLAP_OK;
}

int rlap_debug_set_jitter(rlap_handle h, int quarter_us) {
    if (!h) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->jitter = std::max(0, std::min(64, quarter_us));
    return RLAP_OK;
}

int rlap_rng_uniforms(rlap_handle h, int64_t count, double* d_out) {
    if (!h || count < 0) return RLAP_E_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device);
    int rc = ensure_rng(h, count);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_out, h->rng_ptr, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RLAP_OK;
}

int rlap_debug_wave_sort(rlap_handle h, const double* d_keys, const int32_t* d_offs, int32_t narr, int32_t desc, int32_t* d_perm_out) {
    if (!h || narr < 0) return RLAP_E_BAD_ARG;
    if (narr == 0) return RLAP_OK;
    if (desc & 32) {   // the dataflow kernel's long-column sort (rlap_flow.hip)
        int32_t total = 0;
        HIPCHK(hipMemcpy(&total, d_offs + narr, 4, hipMemcpyDeviceToHost));
        DevBuf scr;
        ENSURE(scr, 32 * (size_t)std::max(total, 1) + 64 * (size_t)narr + 256);
        launch_debug_flow_sort(h->stream, d_keys, d_offs, narr, desc, d_perm_out, scr.as<char>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        scr.release();
        return RLAP_OK;
    }
    hipLaunchKernelGGL(k_debug_wave_sort, dim3((unsigned)std::min<int32_t>(narr, 2048)), dim3(64), 0, h->stream, d_keys, d_offs, narr, desc, d_perm_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RLAP_OK;
}

// networkx-style BA generator (repeated-endpoint list), own xorshift RNG.
int64_t rlap_util_ba_graph(int64_t n, int64_t m, uint64_t seed, int64_t* h_row, int64_t* h_col) {
    if (n <= m || m < 1) return 0;
    int64_t und = m * (n - m);
    if (!h_row || !h_col) return 2 * und;
    std::vector<int32_t> rep;
    rep.reserve((size_t)(2 * und));
    std::vector<int32_t> targets((size_t)m);
    for (int64_t i = 0; i < m; ++i) targets[i] = (int32_t)i;
    std::vector<uint64_t> keys;
    keys.reserve((size_t)(2 * und));
    uint64_t sst = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next = [&]() { sst = mix64(sst); return sst; };
    std::vector<int32_t> mark((size_t)n, -1);
    for (int64_t i = m; i < n; ++i) {
        for (int64_t j = 0; j < m; ++j) {
            int64_t tg = targets[j];
            keys.push_back(((uint64_t)tg << 32) | (uint64_t)i);  // (col=tg,row=i)
            keys.push_back(((uint64_t)i << 32) | (uint64_t)tg);
            rep.push_back((int32_t)tg);
        }
        for (int64_t j = 0; j < m; ++j) rep.push_back((int32_t)i);
        int64_t got = 0;
        while (got < m) {
            uint64_t r = next();
            int32_t cand = rep[(size_t)(((unsigned __int128)r * (unsigned __int128)rep.size()) >> 64)];
            if (mark[cand] != (int32_t)i) { mark[cand] = (int32_t)i; targets[got++] = cand; }
        }
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    for (size_t k = 0; k < keys.size(); ++k) { h_col[k] = (int64_t)(keys[k] >> 32); h_row[k] = (int64_t)(keys[k] & 0xFFFFFFFFull); }
    return (int64_t)keys.size();
}

}  // extern "C"
