// host_mirror.cc -- runs the SAME data-structure code the HIP kernels use
// (rlap_amd/csrc/rlap_core.h: chunked columns, lazy bucket-stack PQ, std::sort
// emulation, sequential elimination/output) on the CPU, so the design can be
// checked against the oracle without a GPU.  Test-only; not part of the product.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../rlap_amd/csrc/rlap_core.h"

using namespace rlap;

extern "C" {

// std_sort_emul permutation of doubles: perm_out[i] = original index ending at i.
void mirror_sort_perm(const double* keys, int64_t n, int desc, int64_t* perm_out) {
    std::vector<SRec> r((size_t)n);
    for (int64_t i = 0; i < n; ++i) { r[i].key = keys[i]; r[i].idx = (int32_t)i; r[i].aux = 0; }
    if (desc) std_sort_emul<true>(r.data(), (int)n); else std_sort_emul<false>(r.data(), (int)n);
    for (int64_t i = 0; i < n; ++i) perm_out[i] = r[i].idx;
}

void mirror_heapsort_perm(const double* keys, int64_t n, int desc, int64_t* perm_out) {
    std::vector<SRec> r((size_t)n);
    for (int64_t i = 0; i < n; ++i) { r[i].key = keys[i]; r[i].idx = (int32_t)i; r[i].aux = 0; }
    if (desc) ss_heap_sort<true>(r.data(), 0, (int)n); else ss_heap_sort<false>(r.data(), 0, (int)n);
    for (int64_t i = 0; i < n; ++i) perm_out[i] = r[i].idx;
}

// Coalesced, symmetric COO in (row[], col[], w[]) -> Schur complement rows.
// Single graph. Returns status; *out malloc'd (m,3).
int mirror_approx_chol(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t,
                       int o_v, int o_n, const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots,
                       double** out, int64_t* out_rows, int64_t* order_out) {
    // CSC by counting sort on (col,row); input assumed duplicate-free here
    std::vector<int32_t> colptr(n + 1, 0);
    for (int64_t p = 0; p < E; ++p) if (w[p] != 0) colptr[col[p] + 1]++;
    for (int64_t c = 0; c < n; ++c) colptr[c + 1] += colptr[c];
    int32_t nnz = colptr[n];
    int32_t slot_cap = nnz + pool_slots;
    std::vector<int32_t> e_nbr(slot_cap), e_twin(slot_cap, -1);
    std::vector<double> e_val(slot_cap, 0.0);
    {
        std::vector<std::pair<int64_t, int64_t>> keyed;  // (col*n+row, p)
        keyed.reserve(E);
        for (int64_t p = 0; p < E; ++p) if (w[p] != 0) keyed.push_back({col[p] * n + row[p], p});
        std::sort(keyed.begin(), keyed.end());
        for (int32_t s = 0; s < nnz; ++s) { e_nbr[s] = (int32_t)row[keyed[s].second]; e_val[s] = w[keyed[s].second]; }
        std::vector<int32_t> cur(colptr.begin(), colptr.end() - 1);
        for (int64_t c = 0; c < n; ++c)
            for (int32_t s = colptr[c]; s < colptr[c + 1]; ++s) e_twin[s] = cur[e_nbr[s]]++;
    }
    std::vector<int32_t> app_cnt(n, 0), app_chunk(n, -1), key(n), pqpos(n, -1);
    std::vector<int32_t> bhead(2 * n + 1, -1), ocur(2 * n + 1, 0), oend(2 * n + 1, 0), orig(n);
    for (int64_t v = 0; v < n; ++v) { key[v] = colptr[v + 1] - colptr[v]; orig[v] = (int32_t)v; }
    std::sort(orig.begin(), orig.end(), [&](int32_t a, int32_t b) { return key[a] != key[b] ? key[a] < key[b] : a > b; });
    for (int32_t i = 0; i < n;) {
        int32_t j = i;
        while (j < n && key[orig[j]] == key[orig[i]]) ++j;
        ocur[key[orig[i]]] = i; oend[key[orig[i]]] = j;
        i = j;
    }
    int32_t log_cap = 4 * nnz + 64;
    std::vector<int32_t> log_v(log_cap), log_prev(log_cap);
    int64_t rng_len = (int64_t)nnz * 4 + 1024;
    std::vector<double> rng((size_t)rng_len);
    {
        std::mt19937_64 g;
        for (auto& u : rng) {
            uint64_t raw = g();
            double d = (double)raw * 5.42101086242752217e-20;  // 2^-64
            if (d >= 1.0) d = 0.99999999999999989;
            u = d;
        }
    }
    std::vector<int64_t> perm_l(n);
    if (perm) for (int64_t i = 0; i < n; ++i) perm_l[i] = perm[i];
    int32_t pool_top = nnz;
    Arrays A;
    A.colptr = colptr.data(); A.e_nbr = e_nbr.data(); A.e_val = e_val.data(); A.e_twin = e_twin.data();
    A.slot_cap = slot_cap; A.pool_top = &pool_top; A.app_cnt = app_cnt.data(); A.app_chunk = app_chunk.data();
    A.key = key.data(); A.pqpos = pqpos.data(); A.bhead = bhead.data(); A.ocur = ocur.data(); A.oend = oend.data();
    A.orig_order = orig.data(); A.log_v = log_v.data(); A.log_prev = log_prev.data();
    A.rng = rng.data(); A.rng_len = rng_len; A.perm = perm_l.data(); A.o_v = o_v; A.o_n = o_n; A.shuffle_seed = shuffle_seed;
    GraphDesc G;
    std::memset(&G, 0, sizeof(G));
    G.vbase = 0; G.n = (int32_t)n; G.t = t; G.bucket_base = 0; G.log_base = 0; G.log_cap = log_cap;
    int32_t cap = nnz + 8;
    std::vector<SRec> rec(cap);
    std::vector<int32_t> i32(11 * (size_t)cap);
    std::vector<double> f64(4 * (size_t)cap);
    ColBuf B;
    B.rec = rec.data();
    int32_t* ip = i32.data();
    B.a_slot = ip; B.a_nbr = ip + cap; B.a_twin = ip + 2 * cap; B.b_slot = ip + 3 * cap; B.b_nbr = ip + 4 * cap; B.b_twin = ip + 5 * cap;
    B.ksel = ip + 6 * cap; B.t_key = ip + 7 * cap; B.t_mv = ip + 8 * cap; B.t_of = ip + 9 * cap;
    B.a_val = f64.data(); B.b_val = f64.data() + cap; B.cum = f64.data() + 2 * cap; B.newv = f64.data() + 3 * cap;

    int64_t nelim = std::min<int64_t>(t, n - 1);
    if (nelim < 0) nelim = 0;
    int64_t npop = 0;
    for (int64_t e1 = 1; e1 <= nelim; ++e1) {
        int32_t v = (o_v == OV_RANDOM) ? (int32_t)perm_l[n - e1] : pq_pop(A, G);
        if (order_out) order_out[npop] = v;
        ++npop;
        int rc = serial_eliminate(A, G, B, cap, v, e1);
        if (rc) return rc;
    }
    std::vector<double> res;
    for (int64_t q = nelim; q < n; ++q) {
        int32_t v = (o_v == OV_RANDOM) ? (int32_t)perm_l[n - 1 - q] : pq_pop(A, G);
        if (order_out) order_out[npop] = v;
        ++npop;
        int32_t m = serial_output(A, G, B, cap, v, nullptr, 0, true);
        size_t r0 = res.size() / 3;
        res.resize(res.size() + 3 * (size_t)m);
        serial_output(A, G, B, cap, v, res.data(), (int64_t)r0, false);
    }
    int64_t m = (int64_t)(res.size() / 3);
    double* o = (double*)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(3 * m, 1));
    if (m) std::memcpy(o, res.data(), sizeof(double) * 3 * (size_t)m);
    *out = o; *out_rows = m;
    return 0;
}

void mirror_free(double* p) { std::free(p); }
}
