// rlap_api.hip -- C ABI (include/rlap_hip.h) and host orchestration of the kernels.
// Host code here only sizes buffers, launches kernels on the handle's stream and
// reads back a few scalars; all arithmetic of the path runs in rlap_kernels.hip.
// rocPRIM (AMD's native primitives, header-only) supplies the device radix sort
// and prefix sums used for plumbing (COO ordering, offsets).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

struct rlap_handle_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    hipEvent_t ev[8];
    // setup
    DevBuf node_ptr_d, vgraph, flags, acc, keys0, keys1, idx0, idx1, head, pos, sorttmp, deg, colptr, slot_col, gcol;
    // graph state
    DevBuf e_nbr, e_val, e_twin, app_cnt, app_chunk, key, pqpos, ocur, oend, origpos, orig_order, gd_d, pool_top;
    DevBuf bs_cnt, bs_alloc, bs_dir, bs_v, bs_id, bs_pool_top, batch_pos, prof;
    DevBuf skey0, skey1, sval0, sval1;
    DevBuf rng;
    int64_t rng_len = 0;
    DevBuf scr_rec, scr_i32, scr_f64;
    // output
    DevBuf surv_base_d, ext, tmp_off, tmp_nbr, tmp_val, cnt, row_off, out_ptr_d, sc_rec, sc_i32, sc_f64, counters, biglist, biglists, hugelists;
    // growth factors kept across calls
    double pool_factor = 1.0;
    double log_factor = 2.0;
};

namespace {

int ensure_rng(rlap_handle h, int64_t count) {
    if (count <= h->rng_len) return RLAP_OK;
    int64_t want = std::max<int64_t>(count, 1 << 16);
    ENSURE(h->rng, sizeof(double) * (size_t)want);
    hipLaunchKernelGGL(k_mt19937_64_table, dim3(1), dim3(320), 0, h->stream, h->rng.as<double>(), want);
    HIPCHK(hipGetLastError());
    h->rng_len = want;
    return RLAP_OK;
}

template <class K, class V>
int sort_pairs(rlap_handle h, K* k_in, K* k_out, V* v_in, V* v_out, int64_t n, unsigned begin_bit, unsigned end_bit) {
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit, h->stream));
    ENSURE(h->sorttmp, bytes);
    bytes = h->sorttmp.cap;
    HIPCHK(rocprim::radix_sort_pairs(h->sorttmp.p, bytes, k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit, h->stream));
    return RLAP_OK;
}

template <class In, class Out>
int excl_scan(rlap_handle h, In* in, Out* out, int64_t n) {
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, (Out)0, (size_t)n, rocprim::plus<Out>(), h->stream));
    ENSURE(h->sorttmp, bytes);
    bytes = h->sorttmp.cap;
    HIPCHK(rocprim::exclusive_scan(h->sorttmp.p, bytes, in, out, (Out)0, (size_t)n, rocprim::plus<Out>(), h->stream));
    return RLAP_OK;
}

__global__ void k_gather_i32(const int32_t* __restrict__ src, const int64_t* __restrict__ index, int32_t n, int32_t* __restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

__global__ void k_ext_totals(const int32_t* __restrict__ ext, int32_t S, int32_t thresh, unsigned long long* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long big = 0;
    if (i < S) { int32_t e = ext[i]; if (e > thresh) big = (unsigned long long)e; }
    for (int off = 32; off > 0; off >>= 1) big += __shfl_down(big, off);
    if ((threadIdx.x & 63) == 0 && big) atomicAdd(out, big);
}

struct Call {
    const int64_t* d_row; const int64_t* d_col; const double* d_w; int64_t E;
    int64_t G; const int64_t* h_node_ptr; const int64_t* h_t;
    int o_v, o_n; const int64_t* d_perm; uint64_t seed;
    double* d_out; int64_t out_cap; int64_t* h_out_row_ptr; rlap_stats* st;
};

int run_once(rlap_handle h, const Call& c, int* retry_kind) {
    *retry_kind = 0;
    hipStream_t s = h->stream;
    const int64_t G = c.G, E = c.E;
    const int64_t N = c.h_node_ptr[G];
    if (c.h_node_ptr[0] != 0) return RLAP_E_BAD_ARG;
    for (int64_t g = 0; g < G; ++g) if (c.h_node_ptr[g + 1] < c.h_node_ptr[g]) return RLAP_E_BAD_ARG;
    if (N >= (int64_t)1 << 30 || E >= (int64_t)1 << 31 || G >= (int64_t)1 << 30) return RLAP_E_TOO_LARGE;
    if (h->timing) HIPCHK(hipEventRecord(h->ev[0], s));

    rlap_stats st;
    std::memset(&st, 0, sizeof(st));
    std::vector<int64_t> out_ptr_h((size_t)G + 1, 0);

    if (N == 0) {
        for (int64_t g = 0; g <= G; ++g) c.h_out_row_ptr[g] = 0;
        if (c.st) *c.st = st;
        return RLAP_OK;
    }

    // ---------------- setup: COO -> CSR ----------------
    ENSURE(h->node_ptr_d, sizeof(int64_t) * (G + 1));
    HIPCHK(hipMemcpyAsync(h->node_ptr_d.p, c.h_node_ptr, sizeof(int64_t) * (G + 1), hipMemcpyHostToDevice, s));
    ENSURE(h->vgraph, sizeof(int32_t) * N);
    hipLaunchKernelGGL(k_vertex_graph, dim3(nblk(N, 256)), dim3(256), 0, s, h->node_ptr_d.as<int64_t>(), (int)G, h->vgraph.as<int32_t>(), N);
    ENSURE(h->flags, sizeof(int32_t) * FLAG_COUNT);
    ENSURE(h->acc, sizeof(double) * 4);
    ENSURE(h->counters, sizeof(unsigned long long) * 8);
    HIPCHK(hipMemsetAsync(h->flags.p, 0, sizeof(int32_t) * FLAG_COUNT, s));
    HIPCHK(hipMemsetAsync(h->acc.p, 0, sizeof(double) * 4, s));
    HIPCHK(hipMemsetAsync(h->counters.p, 0, sizeof(unsigned long long) * 8, s));

    int32_t nnz = 0;
    const int64_t Ealloc = std::max<int64_t>(E, 1);
    ENSURE(h->keys0, 8 * Ealloc); ENSURE(h->keys1, 8 * Ealloc);
    ENSURE(h->idx0, 4 * Ealloc); ENSURE(h->idx1, 4 * Ealloc);
    ENSURE(h->head, 4 * (Ealloc + 1)); ENSURE(h->pos, 4 * (Ealloc + 1));
    if (E > 0) {
        hipLaunchKernelGGL(k_edge_keys, dim3(nblk(E, 256)), dim3(256), 0, s, c.d_row, c.d_col, c.d_w, E, N, h->vgraph.as<int32_t>(),
                           h->keys0.as<uint64_t>(), h->idx0.as<uint32_t>(), h->flags.as<int32_t>());
        int rc = sort_pairs(h, h->keys0.as<uint64_t>(), h->keys1.as<uint64_t>(), h->idx0.as<uint32_t>(), h->idx1.as<uint32_t>(), E, 0, 64);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->head.as<int32_t>() + E, 0, 4, s));
        hipLaunchKernelGGL(k_heads, dim3(nblk(E, 256)), dim3(256), 0, s, h->keys1.as<uint64_t>(), E, h->head.as<int32_t>());
        rc = excl_scan(h, h->head.as<int32_t>(), h->pos.as<int32_t>(), E + 1);
        if (rc) return rc;
        int32_t flags_h[FLAG_COUNT];
        HIPCHK(hipMemcpyAsync(&nnz, h->pos.as<int32_t>() + E, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(flags_h, h->flags.p, sizeof(flags_h), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (flags_h[FLAG_RANGE]) return RLAP_E_INDEX_RANGE;
        if (flags_h[FLAG_CROSS]) return RLAP_E_BAD_ARG;
    }
    st.nnz = nnz;

    // + what the workgroups may hold in reserve (one reservation each; the shape is chosen by graphs vs. CUs at launch)
    const int64_t pool = (int64_t)(h->pool_factor * nnz) + 16 * N + 1024 + G * (G <= 512 ? (int64_t)POOL_GRAB_BIG : (int64_t)POOL_GRAB_SMALL);
    const int64_t slot_cap = (int64_t)nnz + pool;
    if (slot_cap >= ((int64_t)1 << 31) - 64) return RLAP_E_TOO_LARGE;
    ENSURE(h->e_nbr, 4 * slot_cap); ENSURE(h->e_val, 8 * slot_cap); ENSURE(h->e_twin, 4 * slot_cap);
    ENSURE(h->slot_col, 4 * std::max<int64_t>(nnz, 1));
    ENSURE(h->deg, 4 * (N + 1)); ENSURE(h->colptr, 4 * (N + 1));
    HIPCHK(hipMemsetAsync(h->deg.p, 0, 4 * (N + 1), s));
    if (E > 0)
        hipLaunchKernelGGL(k_fill_csr, dim3(nblk(E, 256)), dim3(256), 0, s, h->keys1.as<uint64_t>(), h->idx1.as<uint32_t>(), h->head.as<int32_t>(),
                           h->pos.as<int32_t>(), c.d_w, E, h->e_nbr.as<int32_t>(), h->e_val.as<double>(), h->slot_col.as<int32_t>(), h->deg.as<int32_t>());
    { int rc = excl_scan(h, h->deg.as<int32_t>(), h->colptr.as<int32_t>(), N + 1); if (rc) return rc; }
    if (nnz > 0)
        hipLaunchKernelGGL(k_twin_sym, dim3(nblk(nnz, 256)), dim3(256), 0, s, h->colptr.as<int32_t>(), h->e_nbr.as<int32_t>(), h->e_val.as<double>(),
                           h->slot_col.as<int32_t>(), nnz, h->e_twin.as<int32_t>(), h->acc.as<double>());
    // per-graph nnz (colptr at graph boundaries) + symmetry verdict
    ENSURE(h->gcol, 4 * (G + 1));
    hipLaunchKernelGGL(k_gather_i32, dim3(nblk(G + 1, 256)), dim3(256), 0, s, h->colptr.as<int32_t>(), h->node_ptr_d.as<int64_t>(), (int32_t)(G + 1), h->gcol.as<int32_t>());
    std::vector<int32_t> gcol_h((size_t)G + 1);
    double acc_h[4];
    HIPCHK(hipMemcpyAsync(gcol_h.data(), h->gcol.p, 4 * (G + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(acc_h, h->acc.p, sizeof(acc_h), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (acc_h[2] != 0.0 || !(acc_h[0] <= 1e-24 * acc_h[1])) return RLAP_E_NOT_SYMMETRIC;

    // ---------------- graph descriptors ----------------
    std::vector<GraphDesc> gd((size_t)G);
    int64_t bucket_total = 0, log_total = 0, scr_total = 0, nelim_total = 0;
    std::vector<int64_t> surv_base((size_t)G + 1, 0);
    int64_t max_nnz_g = 0;
    for (int64_t g = 0; g < G; ++g) {
        GraphDesc& D = gd[g];
        std::memset(&D, 0, sizeof(D));
        int64_t n = c.h_node_ptr[g + 1] - c.h_node_ptr[g];
        int64_t nnz_g = gcol_h[g + 1] - gcol_h[g];
        max_nnz_g = std::max(max_nnz_g, nnz_g);
        D.vbase = (int32_t)c.h_node_ptr[g]; D.n = (int32_t)n; D.t = c.h_t[g];
        D.bucket_base = (int32_t)bucket_total; bucket_total += 2 * n + 1;
        log_total += (int64_t)(h->log_factor * nnz_g) + (int64_t)BCH0 * (2 * n + 1) + 64;
        int64_t sc = nnz_g / 2 + 8;
        D.scr_base = (int32_t)scr_total; D.scr_cap = (int32_t)sc; scr_total += sc;
        int64_t ne = std::max<int64_t>(0, std::min<int64_t>(c.h_t[g], n - 1));
        D.n_elim = (int32_t)ne;
        nelim_total += ne;
        surv_base[g + 1] = surv_base[g] + (n - ne);
    }
    if (bucket_total * BDIR >= ((int64_t)1 << 40) || bucket_total >= ((int64_t)1 << 31) || log_total >= ((int64_t)1 << 31) || scr_total >= ((int64_t)1 << 31)) return RLAP_E_TOO_LARGE;
    const int64_t S = surv_base[G];
    st.n_eliminated = nelim_total;
    ENSURE(h->gd_d, sizeof(GraphDesc) * G);
    HIPCHK(hipMemcpyAsync(h->gd_d.p, gd.data(), sizeof(GraphDesc) * G, hipMemcpyHostToDevice, s));

    // ---------------- PQ init ----------------
    ENSURE(h->app_cnt, 4 * N); ENSURE(h->app_chunk, 4 * N); ENSURE(h->key, 4 * N); ENSURE(h->pqpos, 4 * N);
    ENSURE(h->origpos, 4 * N);
    ENSURE(h->ocur, 4 * bucket_total); ENSURE(h->oend, 4 * bucket_total);
    ENSURE(h->bs_cnt, 4 * bucket_total); ENSURE(h->bs_alloc, 4 * bucket_total); ENSURE(h->bs_dir, 4 * bucket_total * BDIR);
    ENSURE(h->bs_v, 4 * log_total); ENSURE(h->bs_id, 4 * log_total); ENSURE(h->bs_pool_top, 4);
    ENSURE(h->batch_pos, 4 * N);
    ENSURE(h->skey0, 8 * N); ENSURE(h->skey1, 8 * N); ENSURE(h->sval0, 4 * N); ENSURE(h->sval1, 4 * N);
    HIPCHK(hipMemsetAsync(h->bs_cnt.p, 0, 4 * bucket_total, s));
    HIPCHK(hipMemsetAsync(h->bs_alloc.p, 0, 4 * bucket_total, s));
    HIPCHK(hipMemsetAsync(h->bs_pool_top.p, 0, 4, s));
    HIPCHK(hipMemsetAsync(h->batch_pos.p, 0xFF, 4 * N, s));
    HIPCHK(hipMemsetAsync(h->ocur.p, 0, 4 * bucket_total, s));
    HIPCHK(hipMemsetAsync(h->oend.p, 0, 4 * bucket_total, s));
    hipLaunchKernelGGL(k_pq_init, dim3(nblk(N, 256)), dim3(256), 0, s, h->colptr.as<int32_t>(), h->vgraph.as<int32_t>(), (int32_t)N, h->key.as<int32_t>(),
                       h->pqpos.as<int32_t>(), h->app_cnt.as<int32_t>(), h->app_chunk.as<int32_t>(), h->skey0.as<uint64_t>(), h->sval0.as<uint32_t>());
    {
        int rc = sort_pairs(h, h->skey0.as<uint64_t>(), h->skey1.as<uint64_t>(), h->sval0.as<uint32_t>(), h->sval1.as<uint32_t>(), N, 0, 64);
        if (rc) return rc;
    }
    ENSURE(h->orig_order, 4 * N);
    HIPCHK(hipMemcpyAsync(h->orig_order.p, h->sval1.p, 4 * N, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_bucket_bounds, dim3(nblk(N, 256)), dim3(256), 0, s, h->orig_order.as<uint32_t>(), h->key.as<int32_t>(), h->vgraph.as<int32_t>(),
                       h->gd_d.as<GraphDesc>(), (int32_t)N, h->ocur.as<int32_t>(), h->oend.as<int32_t>(), h->origpos.as<int32_t>());

    { int rc = ensure_rng(h, std::max<int64_t>(max_nnz_g + 1024, h->rng_len)); if (rc) return rc; }

    ENSURE(h->scr_rec, sizeof(SRec) * scr_total); ENSURE(h->scr_i32, 4 * 10 * scr_total); ENSURE(h->scr_f64, 8 * 4 * scr_total);
    ENSURE(h->pool_top, 4);
    HIPCHK(hipMemcpyAsync(h->pool_top.p, &nnz, 4, hipMemcpyHostToDevice, s));

    Arrays A;
    A.colptr = h->colptr.as<int32_t>();
    A.e_nbr = h->e_nbr.as<int32_t>(); A.e_val = h->e_val.as<double>(); A.e_twin = h->e_twin.as<int32_t>();
    A.slot_cap = (int32_t)slot_cap; A.pool_top = h->pool_top.as<int32_t>();
    A.app_cnt = h->app_cnt.as<int32_t>(); A.app_chunk = h->app_chunk.as<int32_t>();
    A.key = h->key.as<int32_t>(); A.pqpos = h->pqpos.as<int32_t>();
    A.ocur = h->ocur.as<int32_t>(); A.oend = h->oend.as<int32_t>();
    A.orig_order = h->orig_order.as<int32_t>();
    A.bs_cnt = h->bs_cnt.as<int32_t>(); A.bs_alloc = h->bs_alloc.as<int32_t>(); A.bs_dir = h->bs_dir.as<int32_t>();
    A.bs_v = h->bs_v.as<int32_t>(); A.bs_id = h->bs_id.as<int32_t>();
    A.bs_pool_top = h->bs_pool_top.as<int32_t>(); A.bs_pool_cap = (int32_t)log_total;
    A.rng = h->rng.as<double>(); A.rng_len = h->rng_len;
    A.perm = c.d_perm; A.o_v = c.o_v; A.o_n = c.o_n; A.shuffle_seed = c.seed;
    ElimScratch ES;
    ES.rec = h->scr_rec.as<SRec>(); ES.i32 = h->scr_i32.as<int32_t>(); ES.f64 = h->scr_f64.as<double>(); ES.cap = scr_total;
    ES.prof = nullptr;
    const char* prof_env = std::getenv("RLAP_PHASE_PROFILE");   // diagnostic only: per-phase clock sums of graph 0
    if (prof_env && prof_env[0] == '1') {
        ENSURE(h->prof, 8 * 40);
        HIPCHK(hipMemsetAsync(h->prof.p, 0, 8 * 40, s));
        ES.prof = h->prof.as<long long>();
    }

    if (h->timing) HIPCHK(hipEventRecord(h->ev[1], s));
    // ---------------- elimination ----------------
    launch_eliminate_batch(c.o_v, c.o_n, (unsigned)G, s, A, h->gd_d.as<GraphDesc>(), ES, h->batch_pos.as<int32_t>());
    HIPCHK(hipGetLastError());
    if (h->timing) HIPCHK(hipEventRecord(h->ev[2], s));
    HIPCHK(hipMemcpyAsync(gd.data(), h->gd_d.p, sizeof(GraphDesc) * G, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ES.prof) {
        long long pr[40];
        HIPCHK(hipMemcpy(pr, h->prof.p, sizeof(pr), hipMemcpyDeviceToHost));
        const char* names[19] = {"P0 select", "P1 permute + closing barrier", "P1b multi-edge cut + P2 offsets", "P3 sample", "P4 replay", "single path", "P5b slots + shared targets", "move order", "pushes", "P1 meta + loads + dependence", "P5a cursors + pool", "P5c rewire stores", "P1 id rank", "P1 (sync)", "P1 o_n order (sorts)", "P1 write pass", "P1 (sync)", "P1 o_n rank", "empty (cost of one stamp)"};
        std::fprintf(stderr, "[rlap phase profile] rounds=%lld singles=%lld (100 MHz ticks); shader clock over the kernel: %.0f MHz\n", pr[22], pr[23], pr[21] > 0 ? 100.0 * (double)pr[20] / (double)pr[21] : 0.0);
        for (int k = 0; k < 19; ++k) std::fprintf(stderr, "  %-22s %10.3f ms\n", names[k], pr[k] / 1e5);
        const char* wnames[9] = {"gather", "sort by id", "merge", "meta loads + order", "cumsum + recurrence", "sample", "replay + slots", "rewire", "pq commit"};
        for (int k = 0; k < 9; ++k) std::fprintf(stderr, "  single/wave: %-20s %10.3f ms\n", wnames[k], pr[24 + k] / 1e5);
    }
    for (int64_t g = 0; g < G; ++g) {
        if (gd[g].status) {
            int stt = gd[g].status;
            if (stt == ST_POOL_OVERFLOW) *retry_kind = 1;
            else if (stt == ST_LOG_OVERFLOW) *retry_kind = 2;
            else if (stt == ST_RNG_OVERFLOW) *retry_kind = 3;
            return stt;
        }
        st.n_draws = std::max<int64_t>(st.n_draws, gd[g].n_draws);
    }

    // ---------------- output ----------------
    ENSURE(h->surv_base_d, 8 * (G + 1));
    HIPCHK(hipMemcpyAsync(h->surv_base_d.p, surv_base.data(), 8 * (G + 1), hipMemcpyHostToDevice, s));
    uint32_t* order = nullptr;
    if (c.o_v == OV_RANDOM) {
        hipLaunchKernelGGL(k_sc_perm_order, dim3(nblk(N, 256)), dim3(256), 0, s, c.d_perm, h->vgraph.as<int32_t>(), h->gd_d.as<GraphDesc>(),
                           h->surv_base_d.as<int64_t>(), (int32_t)N, h->sval1.as<uint32_t>());
        order = h->sval1.as<uint32_t>();
    } else {
        hipLaunchKernelGGL(k_sc_keys, dim3(nblk(N, 256)), dim3(256), 0, s, h->key.as<int32_t>(), h->pqpos.as<int32_t>(), h->origpos.as<int32_t>(),
                           h->vgraph.as<int32_t>(), h->gd_d.as<GraphDesc>(), (int32_t)N, h->skey0.as<uint64_t>(), h->sval0.as<uint32_t>());
        int rc = sort_pairs(h, h->skey0.as<uint64_t>(), h->skey1.as<uint64_t>(), h->sval0.as<uint32_t>(), h->sval1.as<uint32_t>(), N, 0, 64);
        if (rc) return rc;
        order = h->sval1.as<uint32_t>();
    }
    ENSURE(h->ext, 4 * (S + 1)); ENSURE(h->cnt, 4 * (S + 1)); ENSURE(h->tmp_off, 8 * (S + 1)); ENSURE(h->row_off, 8 * (S + 1));
    HIPCHK(hipMemsetAsync(h->ext.as<int32_t>() + S, 0, 4, s));
    HIPCHK(hipMemsetAsync(h->cnt.as<int32_t>() + S, 0, 4, s));
    unsigned long long* counters = h->counters.as<unsigned long long>();  // [0] sc scratch top, [1] live total, [2] big extent total
    if (S > 0) {
        hipLaunchKernelGGL(k_sc_ext, dim3(nblk(S, 256)), dim3(256), 0, s, order, h->colptr.as<int32_t>(), h->app_cnt.as<int32_t>(), (int32_t)S, h->ext.as<int32_t>());
        hipLaunchKernelGGL(k_ext_totals, dim3(nblk(S, 256)), dim3(256), 0, s, h->ext.as<int32_t>(), (int32_t)S, (int32_t)SCAP, counters + 2);
    }
    { int rc = excl_scan(h, h->ext.as<int32_t>(), h->tmp_off.as<int64_t>(), S + 1); if (rc) return rc; }
    int64_t ext_total = 0;
    unsigned long long big_total = 0;
    HIPCHK(hipMemcpyAsync(&ext_total, h->tmp_off.as<int64_t>() + S, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&big_total, counters + 2, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    ENSURE(h->tmp_nbr, 4 * std::max<int64_t>(ext_total, 1)); ENSURE(h->tmp_val, 8 * std::max<int64_t>(ext_total, 1));
    const int64_t bigcap = (int64_t)big_total + 8;
    ENSURE(h->sc_rec, sizeof(SRec) * bigcap); ENSURE(h->sc_i32, 4 * 7 * bigcap); ENSURE(h->sc_f64, 8 * 2 * bigcap);
    ScScratch SS;
    SS.rec = h->sc_rec.as<SRec>(); SS.i32 = h->sc_i32.as<int32_t>(); SS.f64 = h->sc_f64.as<double>(); SS.cap = bigcap; SS.top = counters + 0;
    if (h->timing) HIPCHK(hipEventRecord(h->ev[3], s));
    if (S > 0) {
        ENSURE(h->biglist, 4 * 6 * (S + 1));
        int32_t* tiercounts = reinterpret_cast<int32_t*>(counters + 4);   // 6 ints: tiers 0..5
        launch_sc_merge(s, A, h->gd_d.as<GraphDesc>(), h->vgraph.as<int32_t>(), order, h->ext.as<int32_t>(), h->tmp_off.as<int64_t>(), (int32_t)S,
                        h->tmp_nbr.as<int32_t>(), h->tmp_val.as<double>(), h->cnt.as<int32_t>(), SS, counters + 1, h->biglist.as<int32_t>(), tiercounts);
        HIPCHK(hipGetLastError());
    }
    if (S > 0) {
        // long columns: whole column in LDS, one single-wave workgroup each
        int32_t* bigcount = reinterpret_cast<int32_t*>(counters + 4) + 3;
        int32_t* biglist_ptr = h->biglist.as<int32_t>() + 3 * (size_t)S;
        static bool attr_set = false;
        if (!attr_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_merge_big), hipFuncAttributeMaxDynamicSharedMemorySize, BIGCAP * 16));
            attr_set = true;
        }
        ENSURE(h->biglists, (size_t)1024 * 2 * (BIGCAP + 2) * sizeof(uint16_t));
        hipLaunchKernelGGL(k_sc_merge_big, dim3(1024), dim3(64), BIGCAP * 16, s, A, order, h->ext.as<int32_t>(), h->tmp_off.as<int64_t>(),
                           biglist_ptr, bigcount, h->tmp_nbr.as<int32_t>(), h->tmp_val.as<double>(), h->cnt.as<int32_t>(), counters + 1,
                           h->biglists.as<uint16_t>());
        HIPCHK(hipGetLastError());
        // longer than the LDS record array (hubs of weighted graphs): records in global scratch, a few workgroups
        constexpr unsigned NHUGE = 32;
        ENSURE(h->hugelists, (size_t)NHUGE * 2 * (HUGECAP + 2) * sizeof(uint16_t));
        hipLaunchKernelGGL(k_sc_merge_huge, dim3(NHUGE), dim3(64), 0, s, A, order, h->ext.as<int32_t>(), h->tmp_off.as<int64_t>(),
                           h->biglist.as<int32_t>() + 4 * (size_t)S, reinterpret_cast<int32_t*>(counters + 4) + 4, h->tmp_nbr.as<int32_t>(),
                           h->tmp_val.as<double>(), h->cnt.as<int32_t>(), counters + 1, h->hugelists.as<uint16_t>(), SS.rec, SS.top);
        HIPCHK(hipGetLastError());
    }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[4], s));
    { int rc = excl_scan(h, h->cnt.as<int32_t>(), h->row_off.as<int64_t>(), S + 1); if (rc) return rc; }
    int64_t m_total = 0;
    unsigned long long live_total = 0;
    HIPCHK(hipMemcpyAsync(&m_total, h->row_off.as<int64_t>() + S, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&live_total, counters + 1, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    st.out_rows = m_total;
    st.live_entries = (int64_t)live_total;
    if (m_total > c.out_cap) {
        c.h_out_row_ptr[G] = m_total;
        if (c.st) *c.st = st;
        return RLAP_E_OUT_OVERFLOW;
    }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[5], s));
    if (S > 0 && m_total > 0) {
        unsigned grid = (unsigned)std::min<int64_t>((m_total + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_sc_compact, dim3(grid), dim3(256), 0, s, order, h->cnt.as<int32_t>(), h->row_off.as<int64_t>(), h->tmp_off.as<int64_t>(),
                           h->tmp_nbr.as<int32_t>(), h->tmp_val.as<double>(), (int32_t)S, c.d_out);
        HIPCHK(hipGetLastError());
    }
    if (h->timing) HIPCHK(hipEventRecord(h->ev[6], s));
    ENSURE(h->out_ptr_d, 8 * (G + 1));
    hipLaunchKernelGGL(k_graph_rows, dim3(nblk(G + 1, 256)), dim3(256), 0, s, h->surv_base_d.as<int64_t>(), h->row_off.as<int64_t>(), (int32_t)G, h->out_ptr_d.as<int64_t>());
    HIPCHK(hipMemcpyAsync(out_ptr_h.data(), h->out_ptr_d.p, 8 * (G + 1), hipMemcpyDeviceToHost, s));
    if (h->timing) HIPCHK(hipEventRecord(h->ev[7], s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t g = 0; g <= G; ++g) c.h_out_row_ptr[g] = out_ptr_h[g];
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
    return RLAP_OK;
}

}  // namespace

extern "C" {

int rlap_create(rlap_handle* out) {
    if (!out) return RLAP_E_BAD_ARG;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    rlap_handle h = new rlap_handle_s();
    h->device = dev;
    for (auto& e : h->ev) HIPCHK(hipEventCreate(&e));
    *out = h;
    return RLAP_OK;
}

int rlap_destroy(rlap_handle h) {
    if (!h) return RLAP_OK;
    DevBuf* bufs[] = {&h->node_ptr_d, &h->vgraph, &h->flags, &h->acc, &h->keys0, &h->keys1, &h->idx0, &h->idx1, &h->head, &h->pos, &h->sorttmp,
                      &h->deg, &h->colptr, &h->slot_col, &h->gcol, &h->e_nbr, &h->e_val, &h->e_twin, &h->app_cnt, &h->app_chunk, &h->key, &h->pqpos,
                      &h->ocur, &h->oend, &h->origpos, &h->orig_order, &h->bs_cnt, &h->bs_alloc, &h->bs_dir, &h->bs_v, &h->bs_id, &h->bs_pool_top, &h->batch_pos, &h->prof, &h->gd_d, &h->pool_top, &h->skey0,
                      &h->skey1, &h->sval0, &h->sval1, &h->rng, &h->scr_rec, &h->scr_i32, &h->scr_f64, &h->surv_base_d, &h->ext, &h->tmp_off,
                      &h->tmp_nbr, &h->tmp_val, &h->cnt, &h->row_off, &h->out_ptr_d, &h->sc_rec, &h->sc_i32, &h->sc_f64, &h->counters, &h->biglist, &h->biglists, &h->hugelists};
    for (DevBuf* b : bufs) b->release();
    for (auto& e : h->ev) (void)hipEventDestroy(e);
    delete h;
    return RLAP_OK;
}

int rlap_set_stream(rlap_handle h, void* hip_stream) {
    if (!h) return RLAP_E_BAD_ARG;
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
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

int rlap_approx_chol_batched(rlap_handle h, const int64_t* d_row, const int64_t* d_col, const double* d_w, int64_t E, int64_t G,
                             const int64_t* h_node_ptr, const int64_t* h_num_remove, int o_v, int o_n, const int64_t* d_perm,
                             uint64_t shuffle_seed, double* d_out, int64_t out_cap_rows, int64_t* h_out_row_ptr, rlap_stats* h_stats) {
    if (!h || G < 1 || E < 0 || !h_node_ptr || !h_num_remove || !h_out_row_ptr) return RLAP_E_BAD_ARG;
    if (o_v < 0 || o_v > 2 || o_n < 0 || o_n > 2) return RLAP_E_BAD_ARG;
    if (o_v == RLAP_OV_RANDOM && !d_perm && h_node_ptr[G] > 0) return RLAP_E_BAD_ARG;
    if (E > 0 && (!d_row || !d_col)) return RLAP_E_BAD_ARG;
    HIPCHK(hipSetDevice(h->device));
    Call c{d_row, d_col, d_w, E, G, h_node_ptr, h_num_remove, o_v, o_n, d_perm, shuffle_seed, d_out, out_cap_rows, h_out_row_ptr, h_stats};
    int rc = RLAP_E_INTERNAL;
    for (int attempt = 0; attempt < 8; ++attempt) {
        int kind = 0;
        rc = run_once(h, c, &kind);
        if (kind == 1) { h->pool_factor = h->pool_factor * 2 + 1; continue; }
        if (kind == 2) { h->log_factor *= 2; continue; }
        if (kind == 3) { int r2 = ensure_rng(h, h->rng_len * 2); if (r2) return r2; continue; }
        break;
    }
    return rc;
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

int rlap_rng_uniforms(rlap_handle h, int64_t count, double* d_out) {
    if (!h || count < 0) return RLAP_E_BAD_ARG;
    int rc = ensure_rng(h, count);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_out, h->rng.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RLAP_OK;
}

int rlap_debug_wave_sort(rlap_handle h, const double* d_keys, const int32_t* d_offs, int32_t narr, int32_t desc, int32_t* d_perm_out) {
    if (!h || narr < 0) return RLAP_E_BAD_ARG;
    if (narr == 0) return RLAP_OK;
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
