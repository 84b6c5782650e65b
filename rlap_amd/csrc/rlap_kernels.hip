// rlap_kernels.hip -- hand-written gfx950 (wave64) kernels of the rLap augmentor.
//
//   K10 k_mt19937_64_table   sampling stream of preconditioner.cc:356-357
//   K1  k_edge_keys/k_heads/k_fill_csr   COO -> CSR  (reader.cc:42-61)
//   K2+K3 k_twin_sym         symmetry check + twin index (factorizers.cc:19-22,
//                            preconditioner.cc:22-49)
//   K4  k_pq_init/k_bucket_bounds  bucket queue (preconditioner.cc:125-165)
//   K5-K8 k_eliminate        one wave per graph, exact sequential semantics
//                            (preconditioner.cc:348-433 | 713-787 | 835-914)
//   K9  k_sc_keys/k_sc_ext/k_sc_merge/k_sc_compact  output (:435-457,312-345)
//
// Compiled with -ffp-contract=off: the reference is built without FMA
// (setup.py:26-37), and index selection depends on the exact roundings.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rlap_core.h"
#include "rlap_kernels.h"

namespace rlap {

// ---------------------------------------------------------------------------
// small wave helpers (wave64)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

// ---------------------------------------------------------------------------
// K10: MT19937-64 (default seed 5489) -> u = (double)raw / 2^64, clamped below 1
// (libstdc++ generate_canonical, bits/random.tcc:3348-3380).  One workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(320) void k_mt19937_64_table(double* __restrict__ out, int64_t count) {
    constexpr int NN = 312, MM = 156;
    constexpr uint64_t MATRIX_A = 0xB5026F5AA96619E9ull, UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull;
    __shared__ uint64_t x[NN];
    const int tid = threadIdx.x;
    if (tid == 0) {
        uint64_t s = 5489ull;
        x[0] = s;
        for (int i = 1; i < NN; ++i) { s = 6364136223846793005ull * (s ^ (s >> 62)) + (uint64_t)i; x[i] = s; }
    }
    __syncthreads();
    for (int64_t base = 0; base < count; base += NN) {
        // twist, first half: inputs are all old words
        uint64_t nv = 0;
        if (tid < MM) {
            uint64_t y = (x[tid] & UM) | (x[tid + 1] & LM);
            nv = x[tid + MM] ^ (y >> 1) ^ ((y & 1ull) ? MATRIX_A : 0ull);
        }
        __syncthreads();
        if (tid < MM) x[tid] = nv;
        __syncthreads();
        // second half: x[i-156] is new, x[i+1] old (x[0] new for i = 311)
        if (tid >= MM && tid < NN) {
            uint64_t nxt = x[(tid + 1) % NN];
            uint64_t y = (x[tid] & UM) | (nxt & LM);
            nv = x[tid - MM] ^ (y >> 1) ^ ((y & 1ull) ? MATRIX_A : 0ull);
        }
        __syncthreads();
        if (tid >= MM && tid < NN) x[tid] = nv;
        __syncthreads();
        if (tid < NN && base + tid < count) {
            uint64_t y = x[tid];
            y ^= (y >> 29) & 0x5555555555555555ull;
            y ^= (y << 17) & 0x71D67FFFEDA60000ull;
            y ^= (y << 37) & 0xFFF7EEE000000000ull;
            y ^= (y >> 43);
            double u = (double)y * 5.42101086242752217003726400434970855712890625e-20;  // 2^-64
            if (u >= 1.0) u = 0.99999999999999988897769753748434595763683319091796875;
            out[base + tid] = u;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// identity / edge_info helpers (py_api_binder.cc:10-51,71-76; ops.py:47)
// ---------------------------------------------------------------------------
__global__ void k_transpose_copy(const double* __restrict__ in, double* __restrict__ out, int64_t rows, int64_t cols, int to_colmajor) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = rows * cols;
    if (i >= total) return;
    int64_t r = i / cols, c = i % cols;
    if (to_colmajor) out[c * rows + r] = in[i]; else out[i] = in[c * rows + r];
}

__global__ void k_unpack_edge_info(const double* __restrict__ ei, int64_t E, int64_t* __restrict__ row, int64_t* __restrict__ col, double* __restrict__ w) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    row[p] = (int64_t)ei[3 * p];
    col[p] = (int64_t)ei[3 * p + 1];
    w[p] = ei[3 * p + 2];
}

// ---------------------------------------------------------------------------
// K1: COO -> CSR
// ---------------------------------------------------------------------------
__global__ void k_vertex_graph(const int64_t* __restrict__ node_ptr, int G, int32_t* __restrict__ vgraph, int64_t N) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    int lo = 0, hi = G;  // last g with node_ptr[g] <= v
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (node_ptr[mid] <= v) lo = mid; else hi = mid;
    }
    vgraph[v] = lo;
}

__global__ void k_edge_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col, const double* __restrict__ w,
                            int64_t E, int64_t N, const int32_t* __restrict__ vgraph, uint64_t* __restrict__ keys,
                            uint32_t* __restrict__ idx, int32_t* __restrict__ flags) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    int64_t r = row[p], c = col[p];
    double wv = w ? w[p] : 1.0;
    uint64_t k = ~0ull;
    if (r < 0 || r >= N || c < 0 || c >= N) {
        flags[FLAG_RANGE] = 1;
    } else if (wv != 0) {
        if (vgraph[r] != vgraph[c]) flags[FLAG_CROSS] = 1;
        k = ((uint64_t)c << 32) | (uint64_t)r;
    }
    keys[p] = k;
    idx[p] = (uint32_t)p;
}

__global__ void k_heads(const uint64_t* __restrict__ keys, int64_t E, int32_t* __restrict__ head) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    uint64_t k = keys[i];
    head[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

__global__ void k_fill_csr(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, const int32_t* __restrict__ head,
                           const int32_t* __restrict__ pos, const double* __restrict__ w, int64_t E, int32_t* __restrict__ e_nbr,
                           double* __restrict__ e_val, int32_t* __restrict__ slot_col, int32_t* __restrict__ deg) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E || !head[i]) return;
    uint64_t k = keys[i];
    int32_t s = pos[i];
    double sum = w ? w[idx[i]] : 1.0;
    for (int64_t q = i + 1; q < E && keys[q] == k; ++q) sum += w ? w[idx[q]] : 1.0;  // duplicates summed in input order
    int32_t c = (int32_t)(k >> 32);
    e_nbr[s] = (int32_t)(k & 0xFFFFFFFFull);
    e_val[s] = sum;
    slot_col[s] = c;
    atomicAdd(&deg[c], 1);
}

// ---------------------------------------------------------------------------
// K2 + K3: twin index by binary search in the neighbour's (sorted) segment and
// the isApprox(A^T) test: ||A-A^T||_F^2 <= 1e-24 ||A||_F^2 (Eigen default prec).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_twin_sym(const int32_t* __restrict__ colptr, const int32_t* __restrict__ e_nbr,
                                                  const double* __restrict__ e_val, const int32_t* __restrict__ slot_col,
                                                  int32_t nnz, int32_t* __restrict__ e_twin, double* __restrict__ acc) {
    int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    double d2 = 0, n2 = 0;
    if (p < nnz) {
        int32_t c = slot_col[p], r = e_nbr[p];
        double v = e_val[p];
        n2 = v * v;
        int32_t lo = colptr[r], hi = colptr[r + 1];
        while (lo < hi) {
            int32_t mid = (lo + hi) >> 1;
            if (e_nbr[mid] < c) lo = mid + 1; else hi = mid;
        }
        if (lo < colptr[r + 1] && e_nbr[lo] == c) {
            e_twin[p] = lo;
            double d = v - e_val[lo];
            d2 = d * d;
        } else {
            e_twin[p] = -1;
            d2 = 2 * v * v;
            acc[2] = 1.0;  // structurally asymmetric
        }
    }
    for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_down(d2, off); n2 += __shfl_down(n2, off); }
    __shared__ double sd[4], sn[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sd[w] = d2; sn[w] = n2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], sd[0] + sd[1] + sd[2] + sd[3]);
        atomicAdd(&acc[1], sn[0] + sn[1] + sn[2] + sn[3]);
    }
}

// ---------------------------------------------------------------------------
// K4: PQ init.  Keys = degree; never-moved vertices of a bucket are listed in
// descending id (= LIFO order after ascending insertion, :137-157).
// ---------------------------------------------------------------------------
__global__ void k_pq_init(const int32_t* __restrict__ colptr, const int32_t* __restrict__ vgraph, int32_t N,
                          int32_t* __restrict__ key, int32_t* __restrict__ pqpos, int32_t* __restrict__ app_cnt,
                          int32_t* __restrict__ app_chunk, uint64_t* __restrict__ skey, uint32_t* __restrict__ sval) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t v = N - 1 - i;  // descending id; the (stable) sort keeps this order inside a bucket
    int32_t d = colptr[v + 1] - colptr[v];
    key[v] = d;
    pqpos[v] = -1;
    app_cnt[v] = 0;
    app_chunk[v] = -1;
    skey[i] = ((uint64_t)(uint32_t)vgraph[v] << 32) | (uint32_t)d;
    sval[i] = (uint32_t)v;
}

__global__ void k_bucket_bounds(const uint32_t* __restrict__ order, const int32_t* __restrict__ key, const int32_t* __restrict__ vgraph,
                                const GraphDesc* __restrict__ gd, int32_t N, int32_t* __restrict__ ocur, int32_t* __restrict__ oend,
                                int32_t* __restrict__ origpos) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t v = (int32_t)order[i];
    origpos[v] = i;
    int32_t g = vgraph[v], d = key[v];
    int32_t b = gd[g].bucket_base + d;
    bool first = true, last = true;
    if (i > 0) { int32_t u = (int32_t)order[i - 1]; first = !(vgraph[u] == g && key[u] == d); }
    if (i < N - 1) { int32_t u = (int32_t)order[i + 1]; last = !(vgraph[u] == g && key[u] == d); }
    if (first) ocur[b] = i;
    if (last) oend[b] = i + 1;
}

// ---------------------------------------------------------------------------
// K5-K8: elimination, one wave (= one 64-thread workgroup) per graph.
// ---------------------------------------------------------------------------

struct ElimLds {
    SRec rec[ECAP];  // sort records; after ordering re-used as cum[ECAP] + newv[ECAP]
    double a_val[ECAP];
    double b_val[ECAP];
    int32_t a_slot[ECAP], a_nbr[ECAP], a_twin[ECAP];
    int32_t b_slot[ECAP], b_nbr[ECAP], b_twin[ECAP], b_pos[ECAP], b_dup[ECAP];
    int32_t ksel[ECAP], t_key[ECAP], t_mv[ECAP], t_of[ECAP], t_cnt[ECAP], t_chunk[ECAP], t_list[ECAP], t_rank[ECAP], pslot[ECAP];
};

// Rank sort == any stable sort. Exact w.r.t. std::sort when cnt <= 16 (pure
// insertion sort, stable) or when all keys are distinct (unique answer).
// Returns true if it wrote rec[rank] = {key, idx}; false if ties need the
// sequential emulation.
template <bool GREATER, class KeyF>
__device__ __forceinline__ bool wave_rank_sort(ElimLds& L, int cnt, KeyF keyf, int lane) {
    bool dup = false;
    for (int i = lane; i < cnt; i += 64) {
        double ki = keyf(i);
        int rank = 0;
        for (int j = 0; j < cnt; ++j) {
            double kj = keyf(j);
            bool before = GREATER ? (kj > ki) : (kj < ki);
            bool eq = (kj == ki);
            rank += (before || (eq && j < i)) ? 1 : 0;
            dup |= (eq && j != i);
        }
        L.t_rank[i] = rank;
    }
    bool anydup = __ballot(dup) != 0ull;
    if (cnt > 16 && anydup) return false;
    for (int i = lane; i < cnt; i += 64) {
        int r = L.t_rank[i];
        L.rec[r].key = keyf(i);
        L.rec[r].idx = i;
    }
    return true;
}

__device__ __forceinline__ int32_t alloc_in_column(const Arrays& A, int32_t& a, int32_t& chunk, int32_t* status) {
    int c = chunk_of(a);
    int cs = chunk_start(c);
    if (a == cs) {
        int32_t need = 1 + chunk_cap(c);
        int32_t base = atomicAdd(A.pool_top, need);
        if (base < 0 || base > A.slot_cap - need) { *status = ST_POOL_OVERFLOW; return 0; }
        A.e_nbr[base] = chunk;
        chunk = base;
    }
    int32_t s = chunk + 1 + (a - cs);
    a += 1;
    return s;
}

__device__ void wave_eliminate(const Arrays& A, GraphDesc& G, ElimLds& L, int32_t v, int64_t e1, int32_t cp0, int32_t cp1,
                               int32_t acnt, int32_t abase) {
    const int lane = lane_id();
    const bool use_pq = A.o_v != OV_RANDOM;
    const bool coarsen = A.o_v == OV_COARSEN;
    const uint64_t lt = lanemask_lt(lane);

    // ---- gather live entries in the reference's traversal order (:248-271) ----
    int len0 = 0;
    {
        int32_t idx = acnt - 1, base = abase;
        int c = idx >= 0 ? chunk_of(idx) : 0;
        while (idx >= 0) {
            int32_t cs = chunk_start(c);
            for (int32_t t0 = idx; t0 >= cs; t0 -= 64) {
                int32_t t = t0 - lane;
                bool valid = t >= cs;
                int32_t s = base + 1 + (t - cs);
                double val = 0; int32_t nb = 0, tw = 0;
                if (valid) { val = A.e_val[s]; nb = A.e_nbr[s]; tw = A.e_twin[s]; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live) { L.a_slot[pos] = s; L.a_nbr[pos] = nb; L.a_val[pos] = val; L.a_twin[pos] = tw; }
                len0 += popc64(mask);
            }
            int32_t prev = A.e_nbr[base];
            idx = cs - 1; base = prev; --c;
        }
        for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
            int32_t s = s0 - lane;
            bool valid = s >= cp0;
            double val = 0; int32_t nb = 0, tw = 0;
            if (valid) { val = A.e_val[s]; nb = A.e_nbr[s]; tw = A.e_twin[s]; }
            bool live = valid && val > 0;
            uint64_t mask = __ballot(live);
            int pos = len0 + popc64(mask & lt);
            if (live) { L.a_slot[pos] = s; L.a_nbr[pos] = nb; L.a_val[pos] = val; L.a_twin[pos] = tw; }
            len0 += popc64(mask);
        }
    }
    __syncthreads();

    // ---- sort by neighbour id (std::sort semantics, :275-276) ----
    {
        bool done = wave_rank_sort<false>(L, len0, [&](int i) { return (double)L.a_nbr[i]; }, lane);
        __syncthreads();
        if (!done) {
            if (lane == 0) {
                for (int i = 0; i < len0; ++i) { L.rec[i].key = (double)L.a_nbr[i]; L.rec[i].idx = i; }
                std_sort_emul<false>(L.rec, len0);
            }
            __syncthreads();
        }
    }

    // ---- merge multi-edges (:278-293) ----
    int m = 0;
    if (len0 <= 64) {
        bool act = lane < len0;
        int src = act ? L.rec[lane].idx : 0;
        int32_t nb = act ? L.a_nbr[src] : -1;
        int32_t nbprev = __shfl_up(nb, 1);
        bool head = act && (lane == 0 || nb != nbprev);
        uint64_t mask = __ballot(head);
        m = popc64(mask);
        int x = popc64(mask & (lt | (1ull << lane))) - 1;
        if (head) {
            uint64_t above = (lane == 63) ? 0ull : (mask >> (lane + 1));
            int nexthead = above ? (lane + 1 + __builtin_ctzll(above)) : len0;
            int d = nexthead - lane - 1;
            double val = L.a_val[src];
            for (int q = 1; q <= d; ++q) val += L.a_val[L.rec[lane + q].idx];
            L.b_slot[x] = L.a_slot[src]; L.b_nbr[x] = nb; L.b_twin[x] = L.a_twin[src]; L.b_val[x] = val;
            L.b_pos[x] = lane; L.b_dup[x] = d;
        } else if (act) {
            A.e_val[L.a_twin[src]] = 0;  // duplicate's twin dies (:289)
        }
    } else {
        if (lane == 0) {
            int mm = 0;
            for (int i = 0; i < len0; ++i) {
                int s = L.rec[i].idx;
                if (mm == 0 || L.a_nbr[s] != L.b_nbr[mm - 1]) {
                    L.b_slot[mm] = L.a_slot[s]; L.b_nbr[mm] = L.a_nbr[s]; L.b_twin[mm] = L.a_twin[s]; L.b_val[mm] = L.a_val[s];
                    L.b_pos[mm] = i; L.b_dup[mm] = 0;
                    ++mm;
                } else {
                    L.b_val[mm - 1] += L.a_val[s];
                    L.b_dup[mm - 1] += 1;
                    A.e_val[L.a_twin[s]] = 0;
                }
            }
            L.t_rank[0] = mm;
        }
        __syncthreads();
        m = L.t_rank[0];
    }
    __syncthreads();

    // ---- neighbour metadata (PQ key, append cursor): issued now, used after ordering ----
    for (int x = lane; x < m; x += 64) {
        int32_t nb = L.b_nbr[x];
        L.t_cnt[x] = A.app_cnt[nb];
        L.t_chunk[x] = A.app_chunk[nb];
        L.t_key[x] = use_pq ? A.key[nb] : 0;
    }

    // ---- order neighbours by o_n (:295-307) ----
    {
        bool done;
        if (A.o_n == ON_RANDOM || coarsen) {
            uint64_t kb = keyed_order_base(A.shuffle_seed, v, 0);
            done = wave_rank_sort<false>(L, m, [&](int i) { return keyed_order_dkey(kb, L.b_nbr[i]); }, lane);
            __syncthreads();
            if (!done) {
                if (lane == 0) {
                    for (int i = 0; i < m; ++i) { L.rec[i].key = keyed_order_dkey(kb, L.b_nbr[i]); L.rec[i].idx = i; }
                    std_sort_emul<false>(L.rec, m);
                }
                __syncthreads();
            }
        } else if (A.o_n == ON_ASC) {
            done = wave_rank_sort<false>(L, m, [&](int i) { return L.b_val[i]; }, lane);
            __syncthreads();
            if (!done) {
                if (lane == 0) {
                    for (int i = 0; i < m; ++i) { L.rec[i].key = L.b_val[i]; L.rec[i].idx = i; }
                    std_sort_emul<false>(L.rec, m);
                }
                __syncthreads();
            }
        } else {
            done = wave_rank_sort<true>(L, m, [&](int i) { return L.b_val[i]; }, lane);
            __syncthreads();
            if (!done) {
                if (lane == 0) {
                    for (int i = 0; i < m; ++i) { L.rec[i].key = L.b_val[i]; L.rec[i].idx = i; }
                    std_sort_emul<true>(L.rec, m);
                }
                __syncthreads();
            }
        }
        for (int j = lane; j < m; j += 64) {
            int x = L.rec[j].idx;
            L.t_of[j] = x;
            L.a_slot[j] = L.b_slot[x]; L.a_nbr[j] = L.b_nbr[x]; L.a_val[j] = L.b_val[x]; L.a_twin[j] = L.b_twin[x];
        }
    }
    __syncthreads();

    double* cum = reinterpret_cast<double*>(L.rec);
    double* newv = cum + ECAP;
    int32_t status = 0;
    int koff_c = 0;        // coarsen: chosen position
    double wk_c = 0;

    // ---- cumulative weights + the f / colScale / wdeg recurrences (:366-417) ----
    if (lane == 0) {
        double csum = 0;
        for (int j = 0; j < m; ++j) { csum += L.a_val[j]; cum[j] = csum; }
        if (!coarsen) {
            double wdeg = csum, colScale = 1;
            for (int j = 0; j < m - 1; ++j) {
                double w = L.a_val[j] * colScale;
                double f = w / wdeg;
                double omf = 1 - f;
                newv[j] = f * omf * wdeg;
                colScale = colScale * omf;
                wdeg = wdeg * omf * omf;
            }
        }
    }
    __syncthreads();
    const double csum = m > 0 ? cum[m - 1] : 0.0;
    const int64_t draws0 = G.n_draws;
    int ndraw = coarsen ? (m >= 1 ? 1 : 0) : (m > 1 ? m - 1 : 0);
    if (draws0 + ndraw > A.rng_len) { if (lane == 0) G.status = ST_RNG_OVERFLOW; __syncthreads(); return; }

    if (coarsen) {
        if (m >= 1) {
            double u = A.rng[draws0];
            double r = u * csum;
            koff_c = upper_index(cum, m, r);
            wk_c = L.a_val[koff_c];
        }
    } else {
        // ---- sample k for every position but the last (:385-394) ----
        for (int j = lane; j < m - 1; j += 64) {
            double u = A.rng[draws0 + j];
            double cj = cum[j];
            double r = u * (csum - cj) + cj;
            L.ksel[j] = upper_index(cum, m, r);
        }
    }
    __syncthreads();

    // ---- per-neighbour: PQ op replay (:291,:399,:427 | :882,:896) + slots for the pushes into its column ----
    const int xk_c = (coarsen && m >= 1) ? L.t_of[koff_c] : -1;
    for (int x = lane; x < m; x += 64) {
        int key = L.t_key[x], mv = -1;
        int32_t a = L.t_cnt[x], chunk = L.t_chunk[x];
        const int32_t a_before = a;
        if (use_pq) {
            int d = L.b_dup[x], p0 = L.b_pos[x];
            for (int q = 1; q <= d; ++q) pq_op(key, mv, G.n, -1, p0 + q);
        }
        if (coarsen) {
            if (x == xk_c) {
                pq_op(key, mv, G.n, -1, len0);
                for (int j = 0; j < m; ++j) {
                    if (j == koff_c) continue;
                    pq_op(key, mv, G.n, +1, len0 + 1 + j);
                    L.pslot[j] = alloc_in_column(A, a, chunk, &status);
                }
            }
        } else {
            for (int j = 0; j < m - 1; ++j) {
                if (L.t_of[L.ksel[j]] == x) {
                    if (use_pq) pq_op(key, mv, G.n, +1, len0 + j);
                    L.pslot[j] = alloc_in_column(A, a, chunk, &status);
                }
            }
            if (use_pq && x == L.t_of[m - 1] && e1 + 1 < (int64_t)G.n) pq_op(key, mv, G.n, -1, len0 + m);
        }
        if (a != a_before) { A.app_cnt[L.b_nbr[x]] = a; A.app_chunk[L.b_nbr[x]] = chunk; }
        if (use_pq) {
            if (key != L.t_key[x]) A.key[L.b_nbr[x]] = key;
            L.t_key[x] = key;
            L.t_mv[x] = mv;
            L.t_list[x] = mv >= 0 ? pq_list_of(key, G.n) : -1;
        }
    }
    if (__ballot(status != 0) != 0ull) { if (lane == 0) G.status = ST_POOL_OVERFLOW; __syncthreads(); return; }
    __syncthreads();

    // ---- rewire: twin rewritten in place, new entry appended to column k (:404-414) ----
    if (coarsen) {
        for (int j = lane; j < m; j += 64) {
            if (j == koff_c) continue;
            double w = L.a_val[j];
            double nw = (wk_c * w) / (wk_c + w);
            int32_t k = L.a_nbr[koff_c], s_r = L.a_twin[j], s_n = L.pslot[j];
            A.e_nbr[s_r] = k; A.e_val[s_r] = nw; A.e_twin[s_r] = s_n;
            A.e_nbr[s_n] = L.a_nbr[j]; A.e_val[s_n] = nw; A.e_twin[s_n] = s_r;
        }
        if (lane == 0 && m >= 1) A.e_val[L.a_twin[koff_c]] = 0;
    } else {
        for (int j = lane; j < m - 1; j += 64) {
            double nw = newv[j];
            int32_t k = L.a_nbr[L.ksel[j]], s_r = L.a_twin[j], s_n = L.pslot[j];
            A.e_nbr[s_r] = k; A.e_val[s_r] = nw; A.e_twin[s_r] = s_n;
            A.e_nbr[s_n] = L.a_nbr[j]; A.e_val[s_n] = nw; A.e_twin[s_n] = s_r;
        }
        if (lane == 0 && m >= 1) A.e_val[L.a_twin[m - 1]] = 0;  // :429-430
    }

    // ---- PQ commit: moved neighbours re-enter their bucket at the head, in op order ----
    if (use_pq) {
        int nmoved_part = 0;
        for (int x = lane; x < m; x += 64) {
            int mvx = L.t_mv[x];
            int rank = 0;
            if (mvx >= 0) {
                for (int y = 0; y < m; ++y) { int mvy = L.t_mv[y]; rank += (mvy >= 0 && mvy < mvx) ? 1 : 0; }
                nmoved_part += 1;
            }
            L.t_rank[x] = rank;
        }
        int nmoved = nmoved_part;
        for (int off = 32; off > 0; off >>= 1) nmoved += __shfl_xor(nmoved, off);
        __syncthreads();
        const int32_t log_cnt0 = G.log_cnt;
        if (log_cnt0 + nmoved > G.log_cap) { if (lane == 0) G.status = ST_LOG_OVERFLOW; __syncthreads(); return; }
        int minl = 0x7FFFFFFF;
        // pass 1: predecessor in the same bucket (or the bucket's old head); t_cnt/t_chunk are free now
        for (int x = lane; x < m; x += 64) {
            int mvx = L.t_mv[x];
            if (mvx < 0) continue;
            int lst = L.t_list[x];
            int pred = -1, predmv = -1;
            bool succ = false;
            for (int y = 0; y < m; ++y) {
                int mvy = L.t_mv[y];
                if (mvy < 0 || L.t_list[y] != lst || y == x) continue;
                if (mvy < mvx) { if (mvy > predmv) { predmv = mvy; pred = y; } } else succ = true;
            }
            L.t_cnt[x] = pred >= 0 ? (log_cnt0 + L.t_rank[pred]) : A.bhead[G.bucket_base + lst];
            L.t_chunk[x] = succ ? 1 : 0;
            minl = min(minl, lst);
        }
        __syncthreads();
        // pass 2: publish
        for (int x = lane; x < m; x += 64) {
            if (L.t_mv[x] < 0) continue;
            int32_t e = log_cnt0 + L.t_rank[x];
            int32_t nb = L.b_nbr[x];
            A.log_v[G.log_base + e] = nb;
            A.log_prev[G.log_base + e] = L.t_cnt[x];
            A.pqpos[nb] = e;
            if (!L.t_chunk[x]) A.bhead[G.bucket_base + L.t_list[x]] = e;
        }
        for (int off = 32; off > 0; off >>= 1) minl = min(minl, __shfl_xor(minl, off));
        if (lane == 0) {
            G.log_cnt = log_cnt0 + nmoved;
            if (minl < G.minlist) G.minlist = minl;
        }
    }
    if (lane == 0) G.n_draws = draws0 + ndraw;
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_eliminate(Arrays A, GraphDesc* __restrict__ gd, ElimScratch S) {
    __shared__ ElimLds L;
    __shared__ GraphDesc G;
    __shared__ int32_t s_v;
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (lane == 0) G = gd[g];
    __syncthreads();
    const int32_t n = G.n;
    int64_t nelim = G.t < (int64_t)(n - 1) ? G.t : (int64_t)(n - 1);
    if (nelim < 0) nelim = 0;
    for (int64_t e1 = 1; e1 <= nelim; ++e1) {
        int32_t v;
        if (A.o_v == OV_RANDOM) {
            v = G.vbase + (int32_t)A.perm[G.vbase + n - e1];
        } else {
            if (lane == 0) s_v = pq_pop(A, G);
            __syncthreads();
            v = s_v;
        }
        int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
        int32_t acnt = A.app_cnt[v], abase = A.app_chunk[v];
        if ((cp1 - cp0) + acnt > ECAP) {
            if (lane == 0) {
                ColBuf B = S.colbuf(G.scr_base);
                int rc = serial_eliminate(A, G, B, G.scr_cap, v, e1);
                if (rc) G.status = rc;
            }
            __syncthreads();
        } else {
            wave_eliminate(A, G, L, v, e1, cp0, cp1, acnt, abase);
        }
        if (G.status != 0) break;
    }
    __syncthreads();
    if (lane == 0) { G.n_elim = (int32_t)nelim; gd[g] = G; }
}

// ---------------------------------------------------------------------------
// K9: output.  sc_keys/sc_perm: pop order of the surviving vertices;
// sc_merge (pass A): per vertex gather -> sort -> merge -> order -> staging;
// sc_compact (pass B): prefix-sum compaction of the staged rows into (m,3) f64.
// ---------------------------------------------------------------------------
__global__ void k_sc_keys(const int32_t* __restrict__ key, const int32_t* __restrict__ pqpos, const int32_t* __restrict__ origpos,
                          const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd, int32_t N,
                          uint64_t* __restrict__ skey, uint32_t* __restrict__ sval) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    int32_t pp = pqpos[v];
    uint64_t k = ~0ull;
    if (pp != -2) {
        const GraphDesc& D = gd[vgraph[v]];
        uint32_t b = (uint32_t)(D.bucket_base + pq_list_of(key[v], D.n));
        uint32_t ord = pp >= 0 ? (0x7FFFFFFFu - (uint32_t)pp) : (0x80000000u + (uint32_t)origpos[v]);
        k = ((uint64_t)b << 32) | ord;
    }
    skey[v] = k;
    sval[v] = (uint32_t)v;
}

__global__ void k_sc_perm_order(const int64_t* __restrict__ perm, const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd,
                                const int64_t* __restrict__ surv_base, int32_t N, uint32_t* __restrict__ order) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t g = vgraph[i];
    const GraphDesc& D = gd[g];
    int32_t idx = i - D.vbase;              // position in the node_id vector
    int64_t q = (int64_t)D.n - 1 - idx;     // pop number (0-based)
    if (q >= D.n_elim) order[surv_base[g] + (q - D.n_elim)] = (uint32_t)(D.vbase + (int32_t)perm[i]);
}

__global__ void k_sc_ext(const uint32_t* __restrict__ order, const int32_t* __restrict__ colptr, const int32_t* __restrict__ app_cnt,
                         int32_t S, int32_t* __restrict__ ext) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    int32_t v = (int32_t)order[i];
    ext[i] = (colptr[v + 1] - colptr[v]) + app_cnt[v];
}

struct ScLds {
    SRec rec[SCAP];
    double a_val[SCAP];
    double b_val[SCAP];
    int32_t a_nbr[SCAP];
    int32_t b_nbr[SCAP];
    int32_t rank[SCAP];
};

template <bool GREATER, class KeyF>
__device__ __forceinline__ bool sc_rank_sort(ScLds& L, int cnt, KeyF keyf, int lane) {
    bool dup = false;
    for (int i = lane; i < cnt; i += 64) {
        double ki = keyf(i);
        int rank = 0;
        for (int j = 0; j < cnt; ++j) {
            double kj = keyf(j);
            bool before = GREATER ? (kj > ki) : (kj < ki);
            bool eq = (kj == ki);
            rank += (before || (eq && j < i)) ? 1 : 0;
            dup |= (eq && j != i);
        }
        L.rank[i] = rank;
    }
    bool anydup = __ballot(dup) != 0ull;
    if (cnt > 16 && anydup) return false;
    for (int i = lane; i < cnt; i += 64) {
        int r = L.rank[i];
        L.rec[r].key = keyf(i);
        L.rec[r].idx = i;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_sc_merge(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                 const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                 const int64_t* __restrict__ tmp_off, int32_t S, int32_t* __restrict__ tmp_nbr,
                                                 double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out, ScScratch SS,
                                                 unsigned long long* __restrict__ live_total) {
    __shared__ ScLds L;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    for (int32_t i = blockIdx.x; i < S; i += gridDim.x) {
        const int32_t v = (int32_t)order[i];
        const int64_t toff = tmp_off[i];
        const int32_t ex = ext[i];
        if (ex > SCAP) {
            // long column: sequential form in global scratch (one lane)
            if (lane == 0) {
                unsigned long long off = atomicAdd(SS.top, (unsigned long long)ex);
                ColBuf B = SS.colbuf((int64_t)off);
                GraphDesc D = gd[vgraph[v]];
                int32_t len0 = serial_gather(A, v, B, ex);
                int32_t m = serial_sort_merge(A, D, B, len0, false, false);
                serial_order(A, B, m, v, 1);
                for (int j = 0; j < m; ++j) { tmp_nbr[toff + j] = B.a_nbr[j]; tmp_val[toff + j] = B.a_val[j]; }
                cnt_out[i] = m;
                atomicAdd(live_total, (unsigned long long)len0);
            }
            __syncthreads();
            continue;
        }
        const int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
        const int32_t acnt = A.app_cnt[v];
        int len0 = 0;
        {
            int32_t idx = acnt - 1, base = A.app_chunk[v];
            int c = idx >= 0 ? chunk_of(idx) : 0;
            while (idx >= 0) {
                int32_t cs = chunk_start(c);
                for (int32_t t0 = idx; t0 >= cs; t0 -= 64) {
                    int32_t t = t0 - lane;
                    bool valid = t >= cs;
                    int32_t s = base + 1 + (t - cs);
                    double val = 0; int32_t nb = 0;
                    if (valid) { val = A.e_val[s]; nb = A.e_nbr[s]; }
                    bool live = valid && val > 0;
                    uint64_t mask = __ballot(live);
                    int pos = len0 + popc64(mask & lt);
                    if (live) { L.a_nbr[pos] = nb; L.a_val[pos] = val; }
                    len0 += popc64(mask);
                }
                int32_t prev = A.e_nbr[base];
                idx = cs - 1; base = prev; --c;
            }
            for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
                int32_t s = s0 - lane;
                bool valid = s >= cp0;
                double val = 0; int32_t nb = 0;
                if (valid) { val = A.e_val[s]; nb = A.e_nbr[s]; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live) { L.a_nbr[pos] = nb; L.a_val[pos] = val; }
                len0 += popc64(mask);
            }
        }
        __syncthreads();
        {
            bool done = sc_rank_sort<false>(L, len0, [&](int q) { return (double)L.a_nbr[q]; }, lane);
            __syncthreads();
            if (!done) {
                if (lane == 0) {
                    for (int q = 0; q < len0; ++q) { L.rec[q].key = (double)L.a_nbr[q]; L.rec[q].idx = q; }
                    std_sort_emul<false>(L.rec, len0);
                }
                __syncthreads();
            }
        }
        // merge (:317-329): head = first of its id group; sums in sorted order
        int m = 0;
        {
            int carry = 0;
            for (int i0 = 0; i0 < len0; i0 += 64) {
                int q = i0 + lane;
                bool act = q < len0;
                int32_t nb = act ? L.a_nbr[L.rec[q].idx] : -1;
                int32_t nbprev = (q > 0 && act) ? L.a_nbr[L.rec[q - 1].idx] : -2;
                bool head = act && nb != nbprev;
                uint64_t mask = __ballot(head);
                int x = carry + popc64(mask & lt);
                if (head) {
                    double val = L.a_val[L.rec[q].idx];
                    for (int z = q + 1; z < len0 && L.a_nbr[L.rec[z].idx] == nb; ++z) val += L.a_val[L.rec[z].idx];
                    L.b_nbr[x] = nb; L.b_val[x] = val;
                }
                carry += popc64(mask);
            }
            m = carry;
        }
        __syncthreads();
        {
            bool done;
            if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) {
                uint64_t kb = keyed_order_base(A.shuffle_seed, v, 1);
                done = sc_rank_sort<false>(L, m, [&](int q) { return keyed_order_dkey(kb, L.b_nbr[q]); }, lane);
                __syncthreads();
                if (!done) {
                    if (lane == 0) {
                        for (int q = 0; q < m; ++q) { L.rec[q].key = keyed_order_dkey(kb, L.b_nbr[q]); L.rec[q].idx = q; }
                        std_sort_emul<false>(L.rec, m);
                    }
                    __syncthreads();
                }
            } else if (A.o_n == ON_ASC) {
                done = sc_rank_sort<false>(L, m, [&](int q) { return L.b_val[q]; }, lane);
                __syncthreads();
                if (!done) {
                    if (lane == 0) {
                        for (int q = 0; q < m; ++q) { L.rec[q].key = L.b_val[q]; L.rec[q].idx = q; }
                        std_sort_emul<false>(L.rec, m);
                    }
                    __syncthreads();
                }
            } else {
                done = sc_rank_sort<true>(L, m, [&](int q) { return L.b_val[q]; }, lane);
                __syncthreads();
                if (!done) {
                    if (lane == 0) {
                        for (int q = 0; q < m; ++q) { L.rec[q].key = L.b_val[q]; L.rec[q].idx = q; }
                        std_sort_emul<true>(L.rec, m);
                    }
                    __syncthreads();
                }
            }
        }
        for (int j = lane; j < m; j += 64) {
            int x = L.rec[j].idx;
            tmp_nbr[toff + j] = L.b_nbr[x];
            tmp_val[toff + j] = L.b_val[x];
        }
        if (lane == 0) { cnt_out[i] = m; atomicAdd(live_total, (unsigned long long)len0); }
        __syncthreads();
    }
}

// Pass B: rows of vertex order[i] go to out[3*row_off[i] ...]; fully coalesced
// 8-byte stores (element d of the vertex's 3*cnt doubles -> row d/3, field d%3).
__global__ __launch_bounds__(256) void k_sc_compact(const uint32_t* __restrict__ order, const int32_t* __restrict__ cnt,
                                                    const int64_t* __restrict__ row_off, const int64_t* __restrict__ tmp_off,
                                                    const int32_t* __restrict__ tmp_nbr, const double* __restrict__ tmp_val,
                                                    int32_t S, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int32_t i = wave; i < S; i += nwaves) {
        const int32_t c = cnt[i];
        const double vcol = (double)order[i];
        const int64_t ro = row_off[i], to = tmp_off[i];
        double* o = out + 3 * ro;
        for (int32_t d = lane; d < 3 * c; d += 64) {
            int32_t r = d / 3, f = d - 3 * r;
            double val = f == 0 ? (double)tmp_nbr[to + r] : (f == 1 ? vcol : tmp_val[to + r]);
            o[d] = val;
        }
    }
}

// per-graph row pointers: first output position of each graph
__global__ void k_graph_rows(const int64_t* __restrict__ surv_base, const int64_t* __restrict__ row_off, int32_t G, int64_t* __restrict__ out_ptr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    out_ptr[g] = row_off[surv_base[g]];
}

}  // namespace rlap
