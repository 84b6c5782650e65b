// rlap_kernels.hip -- hand-written gfx950 (wave64) kernels of the rLap augmentor.
//
//   K10 k_mt19937_64_table   sampling stream of preconditioner.cc:356-357
//   K1  k_edge_keys/k_heads/k_fill_csr   COO -> CSR  (reader.cc:42-61)
//   K2+K3 k_twin_sorted      symmetry check + twin index (factorizers.cc:19-22,
//                            preconditioner.cc:22-49)
//   K4  k_pq_init/k_bucket_bounds  bucket queue (preconditioner.cc:125-165)
//   K5-K8 k_eliminate_batch_t<o_v,o_n,slots,threads>  one workgroup per graph, rounds of independent
//                            vertices with exact sequential semantics; wave_eliminate / wave_eliminate_big /
//                            serial_eliminate for what a round cannot take
//                            (preconditioner.cc:348-433 | 713-787 | 835-914)
//   K9  k_sc_keys/k_sc_ext/k_sc_tierlists/k_sc_merge_t<cap>/k_sc_merge_big/k_sc_merge_huge/k_sc_compact
//                            output (:435-457,312-345)
//   wave_std_sort / wave_sort64 / group_sort   libstdc++ std::sort's permutation, in parallel
//
// Compiled with -ffp-contract=off: the reference is built without FMA
// (setup.py:26-37), and index selection depends on the exact roundings.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>

#include "rlap_core.h"
#include "rlap_kernels.h"
#include "rlap_wave_sort.h"

namespace rlap {


// This file is compiled twice (Makefile).  rlap_kernels.o has everything but the priority-queue instantiations of the elimination
// kernel; rlap_kernels_pq.o (-DRLAP_ELIM_PQ_TU -mllvm -disable-machine-licm) has only those: with machine LICM the invariants that get
// hoisted out of the round loop (LDS base addresses, the reciprocal of n, pointers moved to VGPRs) are spilled in its preheader and
// reloaded from scratch inside it -- 79 reloads in <degree,asc,32,1024>, 12 without; the o_v = random kernels are better off with it.
#ifndef RLAP_ELIM_PQ_TU
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

// Exchange format of sc_edge_info between GPUs (the RCCL all-gather): a row [row, col, w] as two 64-bit
// words -- (row << 32 | col), the bits of w -- 16 bytes over xGMI instead of 24.  One thread per row, both directions.
__global__ void k_pack_rows(const double* __restrict__ sc, int64_t m, unsigned long long* __restrict__ packed) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const unsigned long long r = (unsigned long long)(long long)sc[3 * p], c = (unsigned long long)(long long)sc[3 * p + 1];
    packed[2 * p] = (r << 32) | (c & 0xFFFFFFFFull);
    packed[2 * p + 1] = (unsigned long long)__double_as_longlong(sc[3 * p + 2]);
}
__global__ void k_unpack_rows(const unsigned long long* __restrict__ packed, int64_t m, double* __restrict__ sc) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const unsigned long long ids = packed[2 * p];
    sc[3 * p] = (double)(ids >> 32);
    sc[3 * p + 1] = (double)(ids & 0xFFFFFFFFull);
    sc[3 * p + 2] = __longlong_as_double((long long)packed[2 * p + 1]);
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

// symmetrize != 0: the step before the path (PyG to_undirected, scripts/node_shared.py:326-327) fused in -- every
// input entry (a,b) also yields (b,a); the duplicates this creates are folded by k_heads / k_fill_csr.
__global__ void k_edge_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col, const double* __restrict__ w,
                            int64_t E, int64_t N, const int32_t* __restrict__ vgraph, int symmetrize, int kbits, uint64_t* __restrict__ keys,
                            uint32_t* __restrict__ idx, int32_t* __restrict__ flags) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t Eeff = symmetrize ? 2 * E : E;
    if (p >= Eeff) return;
    const int64_t q = p < E ? p : p - E;
    int64_t r = row[q], c = col[q];
    if (p >= E) { const int64_t t = r; r = c; c = t; }
    double wv = w ? w[q] : 1.0;
    uint64_t k = ~0ull;
    if (r < 0 || r >= N || c < 0 || c >= N) {
        flags[FLAG_RANGE] = 1;
    } else if (wv != 0) {
        if (vgraph && vgraph[r] != vgraph[c]) flags[FLAG_CROSS] = 1;
        k = ((uint64_t)c << kbits) | (uint64_t)r;   // kbits = bits of N - 1: the sort only runs over the 2 * kbits + 1 bits that can differ
    }
    keys[p] = k;
    idx[p] = (uint32_t)q;
    // is the input in (col, row) order already -- or in (row, col) order, what PyG's coalesce leaves?  (Then the radix sort
    // can be skipped, rlap_api.hip.)  Every entry is compared with its predecessor in both orders.
    if (!symmetrize && p > 0) {
        const int64_t r0 = row[p - 1], c0 = col[p - 1];
        const double w0 = w ? w[p - 1] : 1.0;
        const bool bad = (k == ~0ull) || w0 == 0 || r0 < 0 || r0 >= N || c0 < 0 || c0 >= N;
        if (bad || (((uint64_t)c0 << kbits) | (uint64_t)r0) > k) flags[FLAG_UNSORTED_CR] = 1;
        if (bad || (((uint64_t)r0 << kbits) | (uint64_t)c0) > (((uint64_t)r << kbits) | (uint64_t)c)) flags[FLAG_UNSORTED_RC] = 1;
    } else if (p == 0 && k == ~0ull) { flags[FLAG_UNSORTED_CR] = 1; flags[FLAG_UNSORTED_RC] = 1; }
}

// num_nodes = edge_index.max() + 1 (scripts/augmentor_benchmarks.py:77) without a torch reduction + .item()
__global__ void k_max_id(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E, unsigned long long* __restrict__ out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long m = -1;
    bool neg = false;
    for (; p < E; p += (int64_t)gridDim.x * blockDim.x) { long long r = row[p], c = col[p]; m = r > m ? r : m; m = c > m ? c : m; neg |= (r < 0) | (c < 0); }
    for (int off = 32; off > 0; off >>= 1) { long long o = __shfl_down(m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(out, (unsigned long long)(m + 1));
    if (neg) out[1] = 1ull;   // a negative id: out of range whatever num_nodes is
}

// several small fills in one launch (per-call state that used to take a dozen memsets)
__global__ void k_fill_multi(FillJobs J) {
    const int j = blockIdx.y;
    if (j >= J.n) return;
    int32_t* p = J.ptr[j];
    const int64_t cnt = J.count[j];
    const int32_t v = J.value[j];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// o_v = random: the injected node_id vector must hold, per graph, a permutation of its LOCAL ids
// (preconditioner.cc:588-601 builds 0..n-1 and shuffles it); anything else would index out of bounds.
__global__ void k_perm_check(const int64_t* __restrict__ perm, const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd,
                             int32_t N, int32_t* __restrict__ seen, int32_t* __restrict__ flags) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const GraphDesc& D = gd[vgraph[i]];
    const int64_t p = perm[i];
    if (p < 0 || p >= (int64_t)D.n) { flags[FLAG_PERM] = 1; return; }
    if (atomicExch(&seen[D.vbase + (int32_t)p], 1) != 0) flags[FLAG_PERM] = 1;
}

// per-graph scratch for the long-column fall-backs: nnz_g / 2 + 8 entries each, bases by a running sum
// (one workgroup; the per-graph nnz is only known on the device)
__global__ __launch_bounds__(256) void k_gd_scratch(const int32_t* __restrict__ colptr, const int64_t* __restrict__ node_ptr, int32_t G,
                                                    GraphDesc* __restrict__ gd) {
    __shared__ int32_t s_w[4];
    __shared__ int32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int32_t g0 = 0; g0 < G; g0 += 256) {
        const int32_t g = g0 + tid;
        int32_t sc = 0;
        if (g < G) sc = (colptr[node_ptr[g + 1]] - colptr[node_ptr[g]]) / 2 + 8;
        int32_t incl = sc;
        for (int off = 1; off < 64; off <<= 1) { int32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int32_t base = s_carry;
        for (int w = 0; w < wv; ++w) base += s_w[w];
        if (g < G) { gd[g].scr_base = base + incl - sc; gd[g].scr_cap = sc; }
        __syncthreads();
        if (tid == 255) s_carry = base + incl;
        __syncthreads();
    }
}

// everything the host wants to know about a call, gathered into one block (one D2H copy at the end)
__global__ __launch_bounds__(256) void k_collect(const int32_t* __restrict__ flags, const double* __restrict__ acc, const int32_t* __restrict__ nnz_p,
                          const unsigned long long* __restrict__ counters, const unsigned long long* __restrict__ live,
                          const int64_t* __restrict__ tmp_off, const int64_t* __restrict__ row_off,
                          int32_t S, const GraphDesc* __restrict__ gd, int32_t G, const int32_t* __restrict__ pool_top,
                          const int32_t* __restrict__ bs_pool_top, CallResults* __restrict__ out) {
    __shared__ int32_t s_st;
    __shared__ unsigned long long s_nd, s_rounds, s_singles;
    const int tid = threadIdx.x;
    if (tid == 0) { s_st = 0; s_nd = 0ull; s_rounds = 0ull; s_singles = 0ull; }
    __syncthreads();
    int32_t st = 0; unsigned long long nd = 0, rounds = 0, singles = 0;
    for (int32_t g = tid; g < G; g += blockDim.x) {
        const int32_t sg = gd[g].status;
        if (sg > st) st = sg;
        const unsigned long long d = (unsigned long long)gd[g].n_draws;
        nd = d > nd ? d : nd;
        rounds += (unsigned long long)gd[g].pad0; singles += (unsigned long long)gd[g].pad1;
    }
    if (st) atomicMax(&s_st, st);
    if (nd) atomicMax(&s_nd, nd);
    if (rounds) atomicAdd(&s_rounds, rounds);
    if (singles) atomicAdd(&s_singles, singles);
    __syncthreads();
    if (tid != 0) return;
    CallResults R;
    for (int q = 0; q < FLAG_COUNT; ++q) R.flags[q] = flags[q];
    for (int q = 0; q < 4; ++q) R.acc[q] = acc[q];
    R.nnz = *nnz_p;
    { unsigned long long lt = 0ull; for (int q = 0; q < LIVE_SLOTS; ++q) lt += live[q * LIVE_STRIDE]; R.live_total = (int64_t)lt; }
    R.scr_need = (int64_t)counters[0];
    R.ext_total = tmp_off[S];
    R.m_total = row_off[S];
    R.pool_used = *pool_top;
    R.log_used = *bs_pool_top;
    R.status = s_st; R.n_draws = (int64_t)s_nd; R.rounds = (int64_t)s_rounds; R.singles = (int64_t)s_singles;
    *out = R;
}

__global__ void k_heads(const uint64_t* __restrict__ keys, int64_t E, int32_t* __restrict__ head) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > E) return;
    if (i == E) { head[E] = 0; return; }
    uint64_t k = keys[i];
    head[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

__global__ void k_fill_csr(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, const int32_t* __restrict__ head,
                           const int32_t* __restrict__ pos, const double* __restrict__ w, int64_t E, int set_semantics, int kbits,
                           Slot* __restrict__ ent, int32_t* __restrict__ slot_col, int32_t* __restrict__ nbr32) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E || !head[i]) return;
    uint64_t k = keys[i];
    int32_t s = pos[i];
    double sum = w ? w[idx[i]] : 1.0;
    // duplicates summed in input order (setFromTriplets); unweighted symmetrised input is an edge SET (PyG coalesce): weight 1
    if (!set_semantics) for (int64_t q = i + 1; q < E && keys[q] == k; ++q) sum += w ? w[idx[q]] : 1.0;
    int32_t c = (int32_t)(k >> kbits);
    Slot g; g.val = sum; g.nbr = (int32_t)(k & ((1ull << kbits) - 1ull)); g.twin = c;   // (twin holds the COLUMN until k_twin_store: k_twin_sorted reads a partner's ids and weight with one 16-byte load)
    ent[s] = g;
    slot_col[s] = c;
    nbr32[s] = g.nbr;   // dense copy of the row ids: key of the sort that finds the twins (k_twin_sorted)
}

// colptr[c] = first slot whose column is >= c (slots are sorted by column): no atomics, empty columns included
// (thread 0 also files the entry count where the later kernels read it and starts the append pool behind the CSR image:
//  two 4-byte device-to-device copies less per call)
__global__ void k_colptr(const int32_t* __restrict__ slot_col, const int32_t* __restrict__ nnz_p, int32_t N, int32_t* __restrict__ colptr,
                         int32_t* __restrict__ nnz_out, int32_t* __restrict__ pool_top_out) {
    int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > N) return;
    if (c == 0) { const int32_t z = *nnz_p; *nnz_out = z; *pool_top_out = z; }
    int32_t lo = 0, hi = *nnz_p;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (slot_col[mid] < c) lo = mid + 1; else hi = mid;
    }
    colptr[c] = lo;
}

// ---------------------------------------------------------------------------
// K2 + K3: twin index (from one stable sort of the slots by row id) and
// the isApprox(A^T) test: ||A-A^T||_F^2 <= 1e-24 ||A||_F^2 (Eigen default prec).
// ---------------------------------------------------------------------------
// Twins from a sort instead of a search per entry.  The slots are in (col, row) order; a STABLE sort of the slots by
// row alone puts them in (row, col) order: T[k] = slot holding the k-th smallest (row, col) pair.  For a symmetric
// pattern that pair is the transpose of the k-th smallest (col, row) pair, i.e. of slot k: twin[k] = T[k].  Any slot whose
// partner does not hold the transposed ids proves the pattern asymmetric.  (One radix sort of 4-byte keys over bits_for(N)
// bits and two gathers per entry instead of log2(degree) dependent probes.)
__global__ __launch_bounds__(256) void k_twin_sorted(const Slot* __restrict__ ent, const uint32_t* __restrict__ T, const int32_t* __restrict__ nnz_p,
                                                     double* __restrict__ acc) {
    const int32_t nnz = *nnz_p;
    double d2 = 0, n2 = 0;
    bool asym = false;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += gridDim.x * blockDim.x) {
        const Slot me = ent[p];                  // (val, row, column)
        n2 += me.val * me.val;
        const uint32_t q = T[p];
        bool ok = q < (uint32_t)nnz;
        if (ok) {
            const Slot tw = ent[q];              // one 16-byte gather: the partner's weight, row and column
            ok = tw.nbr == me.twin && tw.twin == me.nbr;
            if (ok) { const double d = me.val - tw.val; d2 += d * d; }
        }
        if (!ok) { d2 += 2 * me.val * me.val; asym = true; }
    }
    if (asym) acc[2] = 1.0;  // structurally asymmetric
    for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_down(d2, off); n2 += __shfl_down(n2, off); }
    __shared__ double sd[4], sn[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sd[w] = d2; sn[w] = n2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double bd = sd[0] + sd[1] + sd[2] + sd[3], bn = sn[0] + sn[1] + sn[2] + sn[3];
        if (bd != 0.0) atomicAdd(&acc[0], bd);
        if (bn != 0.0) atomicAdd(&acc[1], bn);
    }
}
// ... and the twin indices stored once every partner has been read (the field held the column until now)
__global__ void k_twin_store(Slot* __restrict__ ent, const uint32_t* __restrict__ T, const int32_t* __restrict__ nnz_p) {
    const int32_t nnz = *nnz_p;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += gridDim.x * blockDim.x) ent[p].twin = (int32_t)T[p];
}

// ---------------------------------------------------------------------------
// K4: PQ init.  Keys = degree; never-moved vertices of a bucket are listed in
// descending id (= LIFO order after ascending insertion, :137-157).
// ---------------------------------------------------------------------------
__global__ void k_pq_init(const int32_t* __restrict__ colptr, const int32_t* __restrict__ vgraph, int32_t N,
                          VRec* __restrict__ vr, uint64_t* __restrict__ skey, uint32_t* __restrict__ sval) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t v = N - 1 - i;  // descending id; the (stable) sort keeps this order inside a bucket
    int32_t d = colptr[v + 1] - colptr[v];
    VRec r; r.key = d; r.pqpos = -1; r.app_cnt = 0; r.app_chunk = -1;
    vr[v] = r;
    skey[i] = ((uint64_t)(uint32_t)vgraph[v] << 32) | (uint32_t)d;
    sval[i] = (uint32_t)v;
}

__global__ void k_bucket_bounds(const uint32_t* __restrict__ order, const VRec* __restrict__ vr, const int32_t* __restrict__ vgraph,
                                const GraphDesc* __restrict__ gd, int32_t N, int32_t* __restrict__ ocur, int32_t* __restrict__ oend,
                                int32_t* __restrict__ origpos) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t v = (int32_t)order[i];
    origpos[v] = i;
    int32_t g = vgraph[v], d = vr[v].key;
    int32_t b = gd[g].bucket_base + d;
    bool first = true, last = true;
    if (i > 0) { int32_t u = (int32_t)order[i - 1]; first = !(vgraph[u] == g && vr[u].key == d); }
    if (i < N - 1) { int32_t u = (int32_t)order[i + 1]; last = !(vgraph[u] == g && vr[u].key == d); }
    if (first) ocur[b] = i;
    if (last) oend[b] = i + 1;
}

#endif  // !RLAP_ELIM_PQ_TU
// ---------------------------------------------------------------------------
// K5-K8: elimination.  One 1024-thread workgroup per graph runs rounds of the batch
// ("frontier") scheme; vertices it cannot take (long columns, multi-edges, keys
// beyond n) go through the single-vertex wave path or the sequential fallback.
// ---------------------------------------------------------------------------



template <int EC>
struct ElimLdsT {
    SRec rec[EC];  // sort records; after ordering re-used as cum[EC] + newv[EC]
    double a_val[EC];
    double b_val[EC];
    int32_t a_slot[EC], a_nbr[EC], a_twin[EC];
    int32_t b_slot[EC], b_nbr[EC], b_twin[EC], b_pos[EC], b_dup[EC];
    int32_t ksel[EC], t_key[EC], t_mv[EC], t_of[EC], t_cnt[EC], t_chunk[EC], t_list[EC], t_rank[EC], pslot[EC];
    WaveSortScratchT<EC> ws;
    alignas(16) double skey[EC + 8];   // sort keys as staged (kept: a sort that meets the depth limit starts over from them)
};

// std::sort order of the staged keys L.skey[0..cnt) into L.rec[] = {key, source index}: one element per lane
// up to 64 keys, the level-synchronous introsort restatement beyond.
template <bool GREATER, int EC>
__device__ __forceinline__ void wave_sort_staged(ElimLdsT<EC>& L, int cnt, int lane) {
    if (cnt <= 64) {
        double key = lane < cnt ? L.skey[lane] : 0.0;
        int idx = lane, pos = lane;
        const bool ok = wave_sort64<GREATER>(key, idx, cnt, lane, L.pslot, &pos);   // pslot is free until the slots are handed out
        if (ok) {
            if (lane < cnt) { L.rec[pos].key = key; L.rec[pos].idx = idx; }
            WAVE_SYNC();
            return;
        }
    }
    WAVE_SYNC();
    const WaveSortPtrs WP = {L.ws.ulist, L.ws.dlist, L.ws.segmark, L.ws.stk};
    for (int i = lane; i < cnt; i += 64) { L.rec[i].key = L.skey[i]; L.rec[i].idx = i; }
    WAVE_SYNC();
    // beyond 64 keys: the introsort restatement, one pass per recursion level (wave_lvl_sort; exact with or without ties -- a
    // stable rank costs as much and is only exact without them); its depth-limit exit starts over.  t_rank / pslot are free here.
    typedef typename std::conditional<GREATER, SRecGreaterKey, SRecLessKey>::type Cmp;
    bool ok;
    if (cnt <= 128) ok = wave_lvl_sort<SRec, Cmp, 2>(L.rec, cnt, Cmp(), L.ws.ulist, L.ws.dlist, reinterpret_cast<uint32_t*>(L.t_rank), reinterpret_cast<uint16_t*>(L.pslot), lane);
    else if (cnt <= 256) ok = wave_lvl_sort<SRec, Cmp, 4>(L.rec, cnt, Cmp(), L.ws.ulist, L.ws.dlist, reinterpret_cast<uint32_t*>(L.t_rank), reinterpret_cast<uint16_t*>(L.pslot), lane);
    else ok = wave_lvl_sort<SRec, Cmp, EC / 64>(L.rec, cnt, Cmp(), L.ws.ulist, L.ws.dlist, reinterpret_cast<uint32_t*>(L.t_rank), reinterpret_cast<uint16_t*>(L.pslot), lane);
    if (!ok) {
        WAVE_SYNC();
        for (int i = lane; i < cnt; i += 64) { L.rec[i].key = L.skey[i]; L.rec[i].idx = i; }
        WAVE_SYNC();
        wave_std_sort<SRec, Cmp, EC / 64>(L.rec, cnt, Cmp(), WP, lane);
    }
    WAVE_SYNC();
}

__device__ __forceinline__ int32_t alloc_in_column(const Arrays& A, int32_t& a, int32_t& chunk, int32_t* status) {
    int c = chunk_of(a);
    int cs = chunk_start(c);
    if (a == cs) {
        int32_t need = 1 + chunk_cap(c);
        int32_t base = atomicAdd(A.pool_top, need);
        if (base < 0 || base > A.slot_cap - need) { *status = ST_POOL_OVERFLOW; return 0; }
        A.e[base].nbr = chunk;
        chunk = base;
    }
    int32_t s = chunk + 1 + (a - cs);
    a += 1;
    return s;
}

__device__ __forceinline__ int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The f / colScale / wdeg recurrence (:396-417): new weight of every position but the last.  One lane; the operation order is the result.
template <int EC>
__device__ __forceinline__ void new_weights(const double* a_val, double* newv, int m, double csum) {
    double wdeg = csum, colScale = 1;
    for (int j0 = 0; j0 < m - 1; j0 += 8) {
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = a_val[j0 + u < EC ? j0 + u : EC - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j0 + u < m - 1) {
                double w = vv[u] * colScale;
                double f = w / wdeg;
                double omf = 1 - f;
                newv[j0 + u] = f * omf * wdeg;
                colScale = colScale * omf;
                wdeg = wdeg * omf * omf;
            }
        }
    }
}

constexpr int SPIN_LIMIT = 1 << 26;   // s_sleep(1) rounds a wave waits for another one of its workgroup before it gives up with ST_INTERNAL (about 2 s: the
                                      // helper wave also sits out the sequential fall-back of a very long column, tens of milliseconds)
constexpr int HELP_MIN = 24;   // columns shorter than this keep the recurrence on the eliminating wave (the hand-over costs about a microsecond)

// Wave 1 while wave 0 runs the single-vertex path: waits for the request, runs the recurrence over the ordered weights, reports.
// (csum is recomputed here: the same left-to-right sum the eliminating wave forms for the cumulative weights.)
template <int EC>
__device__ __noinline__ void single_helper(ElimLdsT<EC>& L, double* big_newv, int32_t* help) {
    const int lane = lane_id();
    int f = 0;
    if (lane == 0) {
        int sp = 0;
        while ((f = __hip_atomic_load(&help[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 1 && f != 3 && f != 4 && ++sp < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
        if (sp >= SPIN_LIMIT) help[2] = 1;   // gave up: the workgroup reports ST_INTERNAL (a request posted later would never be answered)
    }
    f = __shfl(f, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (f == 4) {   // the long-column path (wave_eliminate_big): the ordered weights are in its newv[], rewritten in place
        if (lane == 0) {
            const int m = help[1];
            double csum = 0;
            for (int j = 0; j < m; ++j) csum += big_newv[j];
            double wdeg = csum, colScale = 1;
            for (int j = 0; j < m - 1; ++j) {
                double w = big_newv[j] * colScale;
                double f2 = w / wdeg;
                double omf = 1 - f2;
                big_newv[j] = f2 * omf * wdeg;
                colScale = colScale * omf;
                wdeg = wdeg * omf * omf;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&help[0], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    if (f != 1) {
        if (lane == 0 && f == 3) __hip_atomic_store(&help[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    if (lane == 0) {
        const int m = help[1];
        double csum = 0;
        for (int j0 = 0; j0 < m; j0 += 8) {
            double vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = L.a_val[j0 + u < EC ? j0 + u : EC - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j0 + u < m) csum += vv[u];
        }
        new_weights<EC>(L.a_val, reinterpret_cast<double*>(L.rec) + EC, m, csum);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&help[0], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Single-vertex path, executed by ONE wave (columns up to EC entries, multi-edges, any key range).
template <int EC>
__device__ __noinline__ void wave_eliminate(const Arrays& A, GraphDesc& G, ElimLdsT<EC>& L, int32_t v, int64_t e1, int32_t cp0, int32_t cp1,
                               int32_t acnt, int32_t abase, long long* wprof, int32_t* help) {
    const int lane = lane_id();
    long long wt_prev = wprof ? wall_clock64() : 0;
#define WSTAMP(k) do { if (wprof && lane == 0) { long long _t = wall_clock64(); wprof[k] += _t - wt_prev; wt_prev = _t; } } while (0)
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
                if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; tw = g.twin; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live) { L.a_slot[pos] = s; L.a_nbr[pos] = nb; L.a_val[pos] = val; L.a_twin[pos] = tw; }
                len0 += popc64(mask);
            }
            int32_t prev = A.e[base].nbr;
            idx = cs - 1; base = prev; --c;
        }
        for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
            int32_t s = s0 - lane;
            bool valid = s >= cp0;
            double val = 0; int32_t nb = 0, tw = 0;
            if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; tw = g.twin; }
            bool live = valid && val > 0;
            uint64_t mask = __ballot(live);
            int pos = len0 + popc64(mask & lt);
            if (live) { L.a_slot[pos] = s; L.a_nbr[pos] = nb; L.a_val[pos] = val; L.a_twin[pos] = tw; }
            len0 += popc64(mask);
        }
    }
    WAVE_SYNC();
    WSTAMP(0);

    // ---- sort by neighbour id (std::sort semantics, :275-276) ----
    for (int i = lane; i < len0; i += 64) L.skey[i] = (double)L.a_nbr[i];
    WAVE_SYNC();
    wave_sort_staged<false>(L, len0, lane);

    WSTAMP(1);
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
            A.e[L.a_twin[src]].val = 0;  // duplicate's twin dies (:289)
        }
    } else {
        for (int p0 = 0; p0 < len0; p0 += 64) {
            const int p = p0 + lane;
            const bool act = p < len0;
            const int src = act ? L.rec[p].idx : 0;
            const int32_t nb = act ? L.a_nbr[src] : -1;
            const int32_t nbprev = (act && p > 0) ? L.a_nbr[L.rec[p - 1].idx] : -2;
            const bool head = act && nb != nbprev;
            const uint64_t mask = __ballot(head);
            const int x = m + popc64(mask & lt);
            if (head) {
                double val = L.a_val[src];
                int d = 0;
                for (int q = p + 1; q < len0 && L.a_nbr[L.rec[q].idx] == nb; ++q) { val += L.a_val[L.rec[q].idx]; ++d; }
                L.b_slot[x] = L.a_slot[src]; L.b_nbr[x] = nb; L.b_twin[x] = L.a_twin[src]; L.b_val[x] = val;
                L.b_pos[x] = p; L.b_dup[x] = d;
            } else if (act) {
                A.e[L.a_twin[src]].val = 0;  // duplicate's twin dies (:289)
            }
            m += popc64(mask);
        }
    }
    WAVE_SYNC();

    WSTAMP(2);
    // ---- neighbour metadata (PQ key, append cursor): issued now, used after ordering ----
    for (int x = lane; x < m; x += 64) {
        int32_t nb = L.b_nbr[x];
        L.t_cnt[x] = A.vr[nb].app_cnt;
        L.t_chunk[x] = A.vr[nb].app_chunk;
        L.t_key[x] = use_pq ? A.vr[nb].key : 0;
    }

    // ---- order neighbours by o_n (:295-307) ----
    {
        if (A.o_n == ON_RANDOM || coarsen) {
            uint64_t kb = keyed_order_base(A.shuffle_seed, v - G.vbase, 0);
            for (int i = lane; i < m; i += 64) L.skey[i] = keyed_order_dkey(kb, L.b_nbr[i] - G.vbase);
            WAVE_SYNC();
            wave_sort_staged<false>(L, m, lane);
        } else {
            for (int i = lane; i < m; i += 64) L.skey[i] = L.b_val[i];
            WAVE_SYNC();
            if (A.o_n == ON_ASC) wave_sort_staged<false>(L, m, lane);
            else wave_sort_staged<true>(L, m, lane);
        }
        for (int j = lane; j < m; j += 64) {
            int x = L.rec[j].idx;
            L.t_of[j] = x;
            L.a_slot[j] = L.b_slot[x]; L.a_nbr[j] = L.b_nbr[x]; L.a_val[j] = L.b_val[x]; L.a_twin[j] = L.b_twin[x];
        }
    }
    WAVE_SYNC();

    WSTAMP(3);
    double* cum = reinterpret_cast<double*>(L.rec);
    double* newv = cum + EC;
    int32_t status = 0;
    int koff_c = 0;        // coarsen: chosen position
    double wk_c = 0;

    // ---- cumulative weights + the f / colScale / wdeg recurrences (:366-417) ----
    // The second recurrence (the new weights) is not needed before the rewire: wave 1 of the workgroup, idle otherwise, runs it
    // (single_helper) while this wave samples and hands out the slots.  help[0]: 0 idle, 1 request (help[1] = m), 2 done, 3 no request.
    const bool helped = !coarsen && m >= HELP_MIN;
    if (helped && lane == 0) {
        help[1] = m;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&help[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (a wait that runs out is an internal error, never a result: the new weights would be read unfinished)
#define HELP_FINISH() do { if (helped) { if (lane == 0) { int _sp = 0; while (__hip_atomic_load(&help[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 2 && ++_sp < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1); \
        if (_sp >= SPIN_LIMIT) G.status = ST_INTERNAL; \
        __hip_atomic_store(&help[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); WAVE_SYNC(); } \
        else if (lane == 0) __hip_atomic_store(&help[0], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
    if (lane == 0) {
        // the two recurrences are the result's operation order: one lane, values fetched eight at a time
        // (nothing else runs on the CU now, so every LDS round trip would be waited for)
        double csum = 0;
        for (int j0 = 0; j0 < m; j0 += 8) {
            double vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = L.a_val[j0 + u < EC ? j0 + u : EC - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) { if (j0 + u < m) { csum += vv[u]; cum[j0 + u] = csum; } }
        }
        if (!coarsen && !helped) new_weights<EC>(L.a_val, newv, m, csum);
    }
    WAVE_SYNC();
    WSTAMP(4);
    const double csum = m > 0 ? cum[m - 1] : 0.0;
    const int64_t draws0 = G.n_draws;
    int ndraw = coarsen ? (m >= 1 ? 1 : 0) : (m > 1 ? m - 1 : 0);
    if (draws0 + ndraw > A.rng_len) { HELP_FINISH(); if (lane == 0) G.status = ST_RNG_OVERFLOW; WAVE_SYNC(); return; }

    if (coarsen) {
        if (m >= 1) {
            double u = arr_uniform(A, draws0, v, 0);
            double r = u * csum;
            koff_c = upper_index(cum, m, r);
            wk_c = L.a_val[koff_c];
        }
    } else {
        // ---- sample k for every position but the last (:385-394) ----
        for (int j = lane; j < m - 1; j += 64) {
            double u = arr_uniform(A, draws0 + j, v, j);
            double cj = cum[j];
            double r = u * (csum - cj) + cj;
            L.ksel[j] = upper_index(cum, m, r);
        }
    }
    WAVE_SYNC();

    WSTAMP(5);
    // ---- per-neighbour: PQ op replay (:291,:399,:427 | :882,:896) + slots for the pushes into its column ----
    const int xk_c = (coarsen && m >= 1) ? L.t_of[koff_c] : -1;
    if (!use_pq) {
        // no PQ to replay (o_v = random): hand out the slots 64 positions at a time, in position order.  Lanes
        // that drew the same target form a group; its first lane advances the target's append cursor (kept in
        // LDS between the chunks) once per member, in lane order.
        for (int x = lane; x < m; x += 64) L.t_list[x] = L.t_cnt[x];   // cursor before this vertex
        WAVE_SYNC();
        for (int j0 = 0; j0 < m - 1; j0 += 64) {
            const int j = j0 + lane;
            const bool act = j < m - 1;
            const int x = act ? L.t_of[L.ksel[j]] : 0;
            uint64_t mymask = __ballot(act);   // lanes that drew my target: one ballot per bit of its index (< EC <= 512)
            static_assert(EC <= 512 && EC % 64 == 0, "9 bits of target index");
#pragma unroll
            for (int bit = 0; bit < 9; ++bit) {
                const bool mybit = (x >> bit) & 1;
                const uint64_t bb = __ballot(act && mybit);
                mymask &= mybit ? bb : ~bb;
            }
            if (act && lane == __builtin_ctzll(mymask)) {
                int32_t a = L.t_cnt[x], chunk = L.t_chunk[x];
                uint64_t mm = mymask;
                while (mm) { const int l = __builtin_ctzll(mm); mm &= mm - 1; L.t_rank[l] = alloc_in_column(A, a, chunk, &status); }
                L.t_cnt[x] = a; L.t_chunk[x] = chunk;
            }
            WAVE_SYNC();
            if (act) L.pslot[j] = L.t_rank[lane];
            WAVE_SYNC();
        }
        for (int x = lane; x < m; x += 64) {
            if (L.t_cnt[x] != L.t_list[x]) { A.vr[L.b_nbr[x]].app_cnt = L.t_cnt[x]; A.vr[L.b_nbr[x]].app_chunk = L.t_chunk[x]; }
        }
    } else
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
        if (a != a_before) { A.vr[L.b_nbr[x]].app_cnt = a; A.vr[L.b_nbr[x]].app_chunk = chunk; }
        if (use_pq) {
            if (key != L.t_key[x]) A.vr[L.b_nbr[x]].key = key;
            L.t_key[x] = key;
            L.t_mv[x] = mv;
            L.t_list[x] = mv >= 0 ? pq_list_of(key, G.n) : -1;
        }
    }
    HELP_FINISH();   // (the new weights are in place from here on)
    if (__ballot(status != 0) != 0ull) { if (lane == 0) G.status = ST_POOL_OVERFLOW; WAVE_SYNC(); return; }
    WAVE_SYNC();

    WSTAMP(6);
    // ---- rewire: twin rewritten in place, new entry appended to column k (:404-414) ----
    if (coarsen) {
        for (int j = lane; j < m; j += 64) {
            if (j == koff_c) continue;
            double w = L.a_val[j];
            double nw = (wk_c * w) / (wk_c + w);
            int32_t k = L.a_nbr[koff_c], s_r = L.a_twin[j], s_n = L.pslot[j];
            A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
            A.e[s_n].nbr = L.a_nbr[j]; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
        }
        if (lane == 0 && m >= 1) A.e[L.a_twin[koff_c]].val = 0;
    } else {
        for (int j = lane; j < m - 1; j += 64) {
            double nw = newv[j];
            int32_t k = L.a_nbr[L.ksel[j]], s_r = L.a_twin[j], s_n = L.pslot[j];
            A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
            A.e[s_n].nbr = L.a_nbr[j]; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
        }
        if (lane == 0 && m >= 1) A.e[L.a_twin[m - 1]].val = 0;  // :429-430
    }

    WSTAMP(7);
    // ---- PQ commit: moved neighbours re-enter their bucket at the head, in op order ----
    if (use_pq) {
        for (int x = lane; x < m; x += 64) {
            int mvx = L.t_mv[x];
            int rank = -1;
            if (mvx >= 0) {
                rank = 0;
                for (int y = 0; y < m; ++y) { int mvy = L.t_mv[y]; rank += (mvy >= 0 && mvy < mvx) ? 1 : 0; }
            }
            L.t_rank[x] = rank;
        }
        WAVE_SYNC();
        for (int x = lane; x < m; x += 64) { int r = L.t_rank[x]; if (r >= 0) L.pslot[r] = x; }   // pslot is free now
        int nmoved_part = 0;
        for (int x = lane; x < m; x += 64) nmoved_part += (L.t_rank[x] >= 0) ? 1 : 0;
        int nmoved = nmoved_part;
        for (int off = 32; off > 0; off >>= 1) nmoved += __shfl_xor(nmoved, off);
        WAVE_SYNC();
        if (lane == 0) {
            for (int r = 0; r < nmoved; ++r) {
                int x = L.pslot[r];
                int rc = pq_push(A, G, L.b_nbr[x], L.t_list[x]);
                if (rc) { G.status = rc; break; }
            }
        }
        WAVE_SYNC();
        if (G.status != 0) return;
    }
    if (lane == 0) G.n_draws = draws0 + ndraw;
    WAVE_SYNC();
    WSTAMP(8);
#undef WSTAMP
#undef HELP_FINISH
}


// ---------------------------------------------------------------------------
// Single-vertex path for columns beyond ECAP entries, o_v = random (hubs met early in a random order;
// no PQ to replay).  One wave; the sort records, the stop lists and afterwards the cumulative weights live
// in the workgroup's LDS (the batch round's storage is free while a single vertex is eliminated), the column
// itself in the graph's global scratch (BIGE = 7168 entries with the 1024-thread workgroup, 1536 with the 256-thread one).  Every step is O(len/64) or the O(len log len / 64) sort; the only
// lane-serial parts are the two floating-point recurrences, whose order of operations is the result.
// Returns false (nothing changed) when the column does not fit: the caller falls back to the sequential form.
// ---------------------------------------------------------------------------
template <int BIGE>
struct BigElimLdsT {
    union {
        SRec rec[BIGE];
        struct { double cum[BIGE]; double newv[BIGE]; } c;
    };
    uint16_t ulist[BIGE + 2];
    uint16_t dlist[BIGE + 2];
    uint32_t segmark[BIGE / 32 + 2];
    int32_t stk[3 * 48];
    int32_t tmp[64];
};

template <int BIGE>
__device__ __noinline__ bool wave_eliminate_big(const Arrays& A, GraphDesc& G, BigElimLdsT<BIGE>& L, const ColBuf& B, int32_t cap, int32_t v,
                                                int32_t cp0, int32_t cp1, int32_t acnt, int32_t abase, int32_t* help) {
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    if (cap > BIGE) cap = BIGE;

    // ---- gather live entries in the reference's traversal order (:616-639) ----
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
                if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; tw = g.twin; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live && pos < cap) { B.a_slot[pos] = s; B.a_nbr[pos] = nb; B.a_val[pos] = val; B.a_twin[pos] = tw; }
                len0 += popc64(mask);
            }
            int32_t prev = A.e[base].nbr;
            idx = cs - 1; base = prev; --c;
        }
        for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
            int32_t s = s0 - lane;
            bool valid = s >= cp0;
            double val = 0; int32_t nb = 0, tw = 0;
            if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; tw = g.twin; }
            bool live = valid && val > 0;
            uint64_t mask = __ballot(live);
            int pos = len0 + popc64(mask & lt);
            if (live && pos < cap) { B.a_slot[pos] = s; B.a_nbr[pos] = nb; B.a_val[pos] = val; B.a_twin[pos] = tw; }
            len0 += popc64(mask);
        }
    }
    if (len0 > cap) return false;
    WAVE_SYNC();
    const WaveSortPtrs WP = {L.ulist, L.dlist, L.segmark, L.stk};

    // ---- sort by neighbour id (std::sort semantics, :641-644) ----
    for (int i = lane; i < len0; i += 64) { L.rec[i].key = (double)B.a_nbr[i]; L.rec[i].idx = i; L.rec[i].aux = 0; }
    WAVE_SYNC();
    wave_std_sort<SRec>(L.rec, len0, SRecLessKey(), WP, lane);
    WAVE_SYNC();

    // ---- merge multi-edges (:646-659): group heads by ballot, the sum runs in sorted order ----
    int m = 0;
    for (int p0 = 0; p0 < len0; p0 += 64) {
        const int p = p0 + lane;
        const bool act = p < len0;
        const int s = act ? L.rec[p].idx : 0;
        const int32_t nb = act ? B.a_nbr[s] : -1;
        const int32_t nbprev = (act && p > 0) ? B.a_nbr[L.rec[p - 1].idx] : -2;
        const bool head = act && nb != nbprev;
        const uint64_t mask = __ballot(head);
        const int x = m + popc64(mask & lt);
        if (head) {
            double val = B.a_val[s];
            for (int q = p + 1; q < len0 && B.a_nbr[L.rec[q].idx] == nb; ++q) val += B.a_val[L.rec[q].idx];
            B.b_slot[x] = B.a_slot[s]; B.b_nbr[x] = nb; B.b_twin[x] = B.a_twin[s]; B.b_val[x] = val;
        } else if (act) {
            A.e[B.a_twin[s]].val = 0;   // duplicate's twin dies (:655)
        }
        m += popc64(mask);
    }
    WAVE_SYNC();

    // ---- order neighbours by o_n (:661-673) ----
    if (A.o_n == ON_RANDOM) {
        const uint64_t kb = keyed_order_base(A.shuffle_seed, v - G.vbase, 0);
        for (int i = lane; i < m; i += 64) { L.rec[i].key = keyed_order_dkey(kb, B.b_nbr[i] - G.vbase); L.rec[i].idx = i; }
    } else {
        for (int i = lane; i < m; i += 64) { L.rec[i].key = B.b_val[i]; L.rec[i].idx = i; }
    }
    WAVE_SYNC();
    if (A.o_n == ON_DESC) wave_std_sort<SRec>(L.rec, m, SRecGreaterKey(), WP, lane);
    else wave_std_sort<SRec>(L.rec, m, SRecLessKey(), WP, lane);
    WAVE_SYNC();
    for (int j = lane; j < m; j += 64) {
        const int x = L.rec[j].idx;
        B.a_slot[j] = B.b_slot[x]; B.a_nbr[j] = B.b_nbr[x]; B.a_val[j] = B.b_val[x]; B.a_twin[j] = B.b_twin[x];
    }
    WAVE_SYNC();

    // ---- cumulative weights + the f / colScale / wdeg recurrences (:728-779), in LDS over the sort records ----
    double* cum = L.c.cum;
    double* newv = L.c.newv;
    for (int j = lane; j < m; j += 64) { const double w = B.a_val[j]; newv[j] = w; cum[j] = w; }
    WAVE_SYNC();
    // the new-weight recurrence runs on the workgroup's second wave (single_helper, request 4: in place over newv[]) while this
    // one forms the cumulative weights (in place over cum[]), samples and hands out the slots; it is needed at the rewire
    const bool helped = m >= HELP_MIN;
    if (helped && lane == 0) {
        help[1] = m;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&help[0], 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#define BIG_HELP_FINISH() do { if (helped) { if (lane == 0) { int _sp = 0; while (__hip_atomic_load(&help[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 2 && ++_sp < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1); \
        if (_sp >= SPIN_LIMIT) G.status = ST_INTERNAL; \
        __hip_atomic_store(&help[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); WAVE_SYNC(); } \
        else if (lane == 0) __hip_atomic_store(&help[0], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
    if (lane == 0) {
        double csum = 0;
        for (int j = 0; j < m; ++j) { csum += cum[j]; cum[j] = csum; }
        if (!helped) {
            double wdeg = csum, colScale = 1;
            for (int j = 0; j < m - 1; ++j) {
                double w = newv[j] * colScale;
                double f = w / wdeg;
                double omf = 1 - f;
                newv[j] = f * omf * wdeg;
                colScale = colScale * omf;
                wdeg = wdeg * omf * omf;
            }
        }
    }
    WAVE_SYNC();
    const double csum = m > 0 ? cum[m - 1] : 0.0;
    const int64_t draws0 = G.n_draws;
    const int ndraw = m > 1 ? m - 1 : 0;
    if (draws0 + ndraw > A.rng_len) { BIG_HELP_FINISH(); if (lane == 0) G.status = ST_RNG_OVERFLOW; WAVE_SYNC(); return true; }

    // ---- sample k for every position but the last (:747-756) ----
    for (int j = lane; j < m - 1; j += 64) {
        double u = arr_uniform(A, draws0 + j, v, j);
        double cj = cum[j];
        double r = u * (csum - cj) + cj;
        B.ksel[j] = upper_index(cum, m, r);
    }
    WAVE_SYNC();

    BIG_HELP_FINISH();   // (the new weights are in place from here on)
#undef BIG_HELP_FINISH
    // ---- rewire (:766-776), 64 positions at a time in position order: lanes that drew the same target form a
    //      group whose first lane hands out the target's next slots in lane (= position) order ----
    int32_t status = 0;
    for (int j0 = 0; j0 < m - 1; j0 += 64) {
        const int j = j0 + lane;
        const bool act = j < m - 1;
        const int32_t ks = act ? B.ksel[j] : 0;
        const int32_t k = act ? B.a_nbr[ks] : -1;
        uint64_t mymask = __ballot(act);   // lanes that drew my target: one ballot per bit of its position (< BIGE <= 8192)
        static_assert(BIGE <= 8192, "13 bits of target position");
#pragma unroll
        for (int bit = 0; bit < 13; ++bit) {
            const bool mybit = (ks >> bit) & 1;
            const uint64_t bb = __ballot(act && mybit);
            mymask &= mybit ? bb : ~bb;
        }
        if (act && lane == __builtin_ctzll(mymask)) {
            int32_t a = ld_agent(&A.vr[k].app_cnt), chunk = ld_agent(&A.vr[k].app_chunk);
            uint64_t mm = mymask;
            while (mm) { const int l = __builtin_ctzll(mm); mm &= mm - 1; L.tmp[l] = alloc_in_column(A, a, chunk, &status); }
            A.vr[k].app_cnt = a; A.vr[k].app_chunk = chunk;
        }
        __threadfence();
        WAVE_SYNC();
        if (act && status == 0) {
            const int32_t s_r = B.a_twin[j], s_n = L.tmp[lane];
            const double nw = newv[j];
            A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
            A.e[s_n].nbr = B.a_nbr[j]; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
        }
        WAVE_SYNC();
        if (__ballot(status != 0) != 0ull) break;
    }
    if (__ballot(status != 0) != 0ull) { if (lane == 0) G.status = ST_POOL_OVERFLOW; WAVE_SYNC(); return true; }
    if (lane == 0) {
        if (m >= 1) A.e[B.a_twin[m - 1]].val = 0;   // :791-792
        G.n_draws = draws0 + ndraw;
    }
    WAVE_SYNC();
    return true;
}


// ---------------------------------------------------------------------------
// Batch kernel
// ---------------------------------------------------------------------------
// Two shapes of the workgroup: 1024 threads (4096 slots per round, ~156 KB LDS, one workgroup per CU) for a few large
// graphs; 256 threads (1024 slots, 39 KB and 128 VGPRs: four workgroups per CU) when a batch holds more graphs than the device has
// CUs -- small graphs give short rounds anyway and four of them hide each other's latency.
constexpr int PASSES = 4;     // (candidate, slot) pairs per thread: slots per round = 4 x threads

struct CRec { int32_t x, i, j; };

template <int BC, int NTT>
struct BatchLdsT {
    CandT<BC> cand[4 * NTT / BC];
    union {
        struct {
            uint64_t mkey[NTT];   // PQ moves per round: one per thread, sorted in registers + LDS
            int32_t mval[NTT];
            int32_t hidx[NTT];
        };
        int32_t hcnt[4 * NTT];    // before the commit: counts of the target hash table (keys over pslot)
    };
    int32_t pslot[4 * NTT];   // slot of the entry position p of candidate i appends (commit phase)
    unsigned long long cmask[4 * NTT / BC];   // per candidate: which op numbers (mv - m) end in a PQ move -> order of the moves
    CRec cont[NTT / 2];       // contended (target, candidate) records per round
    CRec csorted[NTT / 2];
    int32_t scan[NTT / 64 + 8];
};

template <int BC, int NTT>
union ElimSharedT {
    BatchLdsT<BC, NTT> b;
    ElimLdsT<(NTT >= 1024 ? ECAP : ECAP_SMALL)> e;
    BigElimLdsT<(NTT >= 1024 ? 7168 : 1536)> g;
};
static_assert(sizeof(ElimSharedT<64, 1024>) == sizeof(BatchLdsT<64, 1024>) && sizeof(ElimSharedT<32, 1024>) == sizeof(BatchLdsT<32, 1024>),
              "the single-vertex paths borrow the batch round's LDS");

// exclusive block scan over the workgroup's NWAVE waves; returns exclusive prefix, *total = sum
template <int NWAVE>
__device__ __noinline__ int block_excl_scan(int val, int* scratch, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = val;
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < NWAVE; ++w) { int sw = scratch[w]; if (w < wave) base += sw; tot += sw; }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

// ordered block compaction helper: position of a flagged thread among flagged threads (thread order)
template <int NWAVE>
__device__ __forceinline__ int block_rank(bool flag, int* scratch, int* total) {
    return block_excl_scan<NWAVE>(flag ? 1 : 0, scratch, total);
}


// rewire stores of one push (preconditioner.cc:404-414)
__device__ __forceinline__ void rewire_store(const Arrays& A, int32_t s_r, int32_t s_n, int32_t nbr_j, int32_t k, double nw) {
    Slot a; a.val = nw; a.nbr = k; a.twin = s_n;
    Slot b; b.val = nw; b.nbr = nbr_j; b.twin = s_r;
    A.e[s_r] = a;     // one 16-byte store per slot
    A.e[s_n] = b;
}

// all pushes of candidate C into the column of its target at position j, in position order
template <class CT>
__device__ __forceinline__ void push_into_target(const Arrays& A, const CT& C, int32_t j, int32_t& a, int32_t& chunk, int32_t* status) {
    const int32_t x = C.e[j].nbr;
    if (A.o_v == OV_COARSEN) {
        if (j != C.koff) return;
        for (int32_t p = 0; p < C.m; ++p) {
            if (p == C.koff) continue;
            int32_t s_n = alloc_in_column(A, a, chunk, status);
            if (*status) return;
            rewire_store(A, C.e[p].twin, s_n, C.e[p].nbr, x, C.e[p].val);
        }
    } else {
        for (int32_t p = 0; p < C.m - 1; ++p) {
            if (C.ksel[p] != j) continue;
            int32_t s_n = alloc_in_column(A, a, chunk, status);
            if (*status) return;
            rewire_store(A, C.e[p].twin, s_n, C.e[p].nbr, x, C.e[p].val);
        }
    }
}

// slots for all pushes of candidate C into the column of its target at position j (position order)
template <class CT>
__device__ __forceinline__ void slots_into_target(const Arrays& A, const CT& C, int32_t j, int32_t& a, int32_t& chunk, int32_t* status, int32_t* pslot_row) {
    const bool co = A.o_v == OV_COARSEN;
    if (co && j != C.koff) return;
    const int32_t plast = co ? C.m : C.m - 1;
    for (int32_t p = 0; p < plast; ++p) {
        if (co ? (p == C.koff) : (C.ksel[p] != j)) continue;
        pslot_row[p] = alloc_in_column(A, a, chunk, status);
        if (*status) return;
    }
}

template <class CT>
__device__ __noinline__ void cand_order_index_call(const Arrays& A, CT& C) { cand_order_index(A, C); }

// rlap_core.h::cand_patch by one wave (32-slot candidates: one entry per lane of the lower half), followed by the sampling of
// the patched candidate.  d's entries live in registers while they are patched and ordered again: by id with a rank count
// (ids are distinct), by o_n with the half-wave std::sort restatement.  Returns false when d cannot be patched.
template <int ON, class CT>
__device__ __noinline__ bool wave_patch(const int32_t o_v, const uint64_t shuffle_seed, const double* __restrict__ rng, const int32_t rng_mode, CT* cand, int32_t d, int32_t vbase, int32_t* tmp_wave) {
    Arrays A;            // (by value, like cand_prepare_wide: what the sampling helpers read of the argument block; lives in registers)
    A.o_v = o_v; A.o_n = ON; A.shuffle_seed = shuffle_seed; A.rng = rng; A.rng_mode = rng_mode; A.vbase = vbase;
    static_assert(CT::CAP == 32 || CT::CAP == 64, "one entry per lane");
    const int lane = lane_id();
    CT& C = cand[d];
    const int32_t m = C.m, nd = C.ndep, v = C.v;
    if (C.nkill > 0) return false;   // merged multi-edges (64-slot candidates): see rlap_core.h::cand_patch
    int32_t my_nbr = lane < m ? C.e[lane].nbr : -1;
    double my_val = lane < m ? C.e[lane].val : 0.0;
    int32_t my_twin = lane < m ? C.e[lane].twin : 0;
    bool ok = true, dup = false;
    for (int32_t q = 0; q < nd && ok; ++q) {
        const int32_t j = C.dep[q];
        const CT& J = cand[j];
        const int32_t mj = J.m;
        const int32_t jn = lane < mj ? J.e[lane].nbr : -2;
        const uint64_t pm = __ballot(jn == v);
        if (pm == 0ull) { ok = false; break; }
        const int32_t p = __builtin_ctzll(pm);
        if (p == mj - 1) { ok = false; break; }                                   // last neighbour (:420-432)
        const int32_t ks = lane < mj - 1 ? (int32_t)J.ksel[lane] : -1;
        if (__ballot(ks == p) != 0ull) { ok = false; break; }                     // sampled as a target (:394-399)
        const int32_t k = J.e[J.ksel[p]].nbr;
        const double newv = J.e[p].val;
        if (!(newv > 0)) { ok = false; break; }                                   // dead on arrival (rounding: f > 1), see cand_patch
        if (__ballot(my_nbr == k) != 0ull) { ok = false; dup = true; break; }     // multi-edge: single-vertex path
        const uint64_t qm = __ballot(my_nbr == J.v);
        if (qm == 0ull) { ok = false; break; }
        if (lane == __builtin_ctzll(qm)) { my_nbr = k; my_val = newv; my_twin = ~(j * CT::CAP + p); }
    }
    if (!ok) {
        if (lane == 0 && dup) atomicOr(&C.flags, CF_DUP);
        return false;
    }
    // order by id (:275): rank = smaller ids (distinct)
    int32_t rank = 0;
    for (int32_t q = 0; q < m; ++q) { const int32_t nq = __shfl(my_nbr, q); rank += (nq < my_nbr) ? 1 : 0; }
    {
        const int target = lane < m ? rank : lane;
        my_nbr = __builtin_amdgcn_ds_permute(target << 2, my_nbr);
        my_twin = __builtin_amdgcn_ds_permute(target << 2, my_twin);
        const long long bv = __double_as_longlong(my_val);
        const int lo = __builtin_amdgcn_ds_permute(target << 2, (int)(uint32_t)bv), hi = __builtin_amdgcn_ds_permute(target << 2, (int)(uint32_t)((unsigned long long)bv >> 32));
        my_val = __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
    }
    // order by o_n (:295-307) with std::sort semantics
    constexpr bool keyed = (ON == ON_RANDOM);
    double key = keyed ? keyed_order_dkey(keyed_order_base(A.shuffle_seed, v - vbase, 0), my_nbr - vbase) : my_val;
    int idx = lane & (CT::CAP - 1), pos = lane & (CT::CAP - 1);
    if (CT::CAP == 32) {
        const bool want = lane < 32 && m > 1;
        const bool okg = (ON == ON_DESC) ? group_sort<true, 32>(key, idx, m, want, lane, tmp_wave, &pos) : group_sort<false, 32>(key, idx, m, want, lane, tmp_wave, &pos);
        if (__ballot(want && !okg) != 0ull) return false;   // depth limit of the introsort: leave it to the single-vertex path (conservative)
    } else if (m > 1) {
        const bool okg = (ON == ON_DESC) ? wave_sort64<true>(key, idx, m, lane, tmp_wave, &pos) : wave_sort64<false>(key, idx, m, lane, tmp_wave, &pos);
        if (!okg) return false;
    }
    {
        const int32_t nb = __shfl(my_nbr, idx), tw = __shfl(my_twin, idx);
        const double vv = __shfl(my_val, idx);
        WAVE_SYNC();
        if (lane < m) { Ent& E = C.e[pos]; E.nbr = nb; E.twin = tw; E.val = vv; E.aux = 0; }
    }
    WAVE_SYNC();
    // sampling of the patched candidate (P3 of the round, for this one candidate)
    if (lane == 0) cand_cumsum(A, C);
    WAVE_SYNC();
    if (lane < m - 1) cand_pick(A, C, lane);
    if (lane == 63) cand_recur(A, C);
    WAVE_SYNC();
    return true;
}

// ---------------------------------------------------------------------------
// Candidate preparation for 128-slot candidates (o_v = random on graphs whose columns run long: a column of 65..128 entries
// no longer ends the round).  ONE wave per candidate, two entries per lane: entry e = lane came in as (v0, n0, t0, a0),
// entry 64 + lane as (v1, ...); the caller has stored the raw (val, nbr) of every loaded slot at C.e[e].  Steps as in the
// 32/64-slot code: rank by id among the live entries (std::sort order where ids repeat), multi-edges folded (:278-293),
// o_n order with std::sort semantics (:295-307), permutation applied.  The sorts beyond 64 elements are the level-synchronous
// restatement over an index array (the keys stay in the record).  scr: WIDE_SCR_BYTES of LDS owned by this wave.
// ---------------------------------------------------------------------------
constexpr int WIDE_SCR_BYTES = 2048;

// Bitonic network over the 128 values a wave holds two per lane (element t*64 + lane in a[t]): ascending.  The exchanges
// with distance 64 stay inside the lane, the others are xor-shuffles.  Only the first KMAX elements (a power of two) matter
// to the caller, the rest is padding that compares last: the stages beyond KMAX are left out, and up to 64 the second register
// is left alone.  The second form sorts (key, source) pairs by key, then source -- a stable order.
template <int KMAX>
__device__ __forceinline__ void wave_bitonic_t(unsigned long long (&a)[2], int lane) {
#pragma unroll
    for (int kk = 2; kk <= KMAX; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            if (jj == 64) {
                if (a[1] < a[0]) { const unsigned long long t = a[0]; a[0] = a[1]; a[1] = t; }
            } else {
#pragma unroll
                for (int t = 0; t < (KMAX > 64 ? 2 : 1); ++t) {
                    const int e = t * 64 + lane;
                    const unsigned long long o = __shfl_xor(a[t], jj);
                    const bool keep_min = (((e & jj) == 0) == ((e & kk) == 0));
                    if (keep_min == (o < a[t])) a[t] = o;
                }
            }
        }
    }
}
template <int KMAX>
__device__ __forceinline__ void wave_bitonic_t(unsigned long long (&a)[2], int (&s)[2], int lane) {
#pragma unroll
    for (int kk = 2; kk <= KMAX; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            if (jj == 64) {
                if (a[1] < a[0] || (a[1] == a[0] && s[1] < s[0])) { const unsigned long long t = a[0]; a[0] = a[1]; a[1] = t; const int u = s[0]; s[0] = s[1]; s[1] = u; }
            } else {
#pragma unroll
                for (int t = 0; t < (KMAX > 64 ? 2 : 1); ++t) {
                    const int e = t * 64 + lane;
                    const unsigned long long o = __shfl_xor(a[t], jj);
                    const int os = __shfl_xor(s[t], jj);
                    const bool keep_min = (((e & jj) == 0) == ((e & kk) == 0));
                    const bool other_less = o < a[t] || (o == a[t] && os < s[t]);
                    if (keep_min == other_less) { a[t] = o; s[t] = os; }
                }
            }
        }
    }
}
__device__ __forceinline__ void wave_bitonic128(unsigned long long (&a)[2], int lane, int count) {   // count: elements that matter (wave-uniform)
    if (count <= 16) wave_bitonic_t<16>(a, lane);
    else if (count <= 32) wave_bitonic_t<32>(a, lane);
    else if (count <= 64) wave_bitonic_t<64>(a, lane);
    else wave_bitonic_t<128>(a, lane);
}
__device__ __forceinline__ void wave_bitonic128(unsigned long long (&a)[2], int (&s)[2], int lane, int count) {
    if (count <= 16) wave_bitonic_t<16>(a, s, lane);
    else if (count <= 32) wave_bitonic_t<32>(a, s, lane);
    else if (count <= 64) wave_bitonic_t<64>(a, s, lane);
    else wave_bitonic_t<128>(a, s, lane);
}
struct WideIdLess { const Ent* e; __device__ bool operator()(uint16_t a, uint16_t b) const { return e[a].nbr < e[b].nbr; } };
struct WideAuxLess { const Ent* e; __device__ bool operator()(uint16_t a, uint16_t b) const { return e[a].aux < e[b].aux; } };
struct WideAuxGreater { const Ent* e; __device__ bool operator()(uint16_t a, uint16_t b) const { return e[a].aux > e[b].aux; } };

// (the three mode scalars by value, not `const Arrays&`: a reference to the kernel's argument block makes every call store a 200-byte
// copy of it to the stack -- 24 scratch stores per candidate in the round loop -- for three fields)
__device__ __noinline__ void cand_prepare_wide(const int32_t o_v, const int32_t o_n, const uint64_t shuffle_seed, CandT<128>& C, int32_t vbase, double v0, int32_t n0, int32_t t0, bool a0,
                                               double v1, int32_t n1, int32_t t1, bool a1, uint8_t* scr, const uint8_t* __restrict__ eqg) {
    ASSUME_LDS(&C); ASSUME_LDS(scr);
    Arrays A;            // what the callees below read of it (o_v, o_n); lives in registers
    A.o_v = o_v; A.o_n = o_n; A.shuffle_seed = shuffle_seed;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    const bool keyed = (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN);
    const bool desc = (A.o_n == ON_DESC) && !keyed;
    uint16_t* const rec = reinterpret_cast<uint16_t*>(scr);
    uint16_t* const ulist = reinterpret_cast<uint16_t*>(scr + 256);
    uint16_t* const dlist = reinterpret_cast<uint16_t*>(scr + 520);
    uint32_t* const tab = reinterpret_cast<uint32_t*>(scr + 784);
    uint16_t* const tab2 = reinterpret_cast<uint16_t*>(scr + 1296);
    uint8_t* const posmap = scr + 1552;
    int32_t* const tmp64 = reinterpret_cast<int32_t*>(scr);   // wave_sort64's scratch (never live together with the arrays above)
    static_assert(1552 + 128 <= WIDE_SCR_BYTES && 256 * 4 <= WIDE_SCR_BYTES, "scratch layout");
    const int32_t ext = C.ext;   // slots loaded (dead ones included)
    const bool live0 = a0 && v0 > 0, live1 = a1 && v1 > 0;
    const uint64_t h0 = __ballot(live0), h1 = __ballot(live1);
    const int32_t nlive = popc64(h0) + popc64(h1);
    // ---- rank by id among the live entries (dead ones last); equal ids = multi-edges.  Key = (id, place in the traversal
    //      order): the network's order is the stable one, i.e. the insertion sort's ----
    // A column without appended entries whose CSR segment is still in order (in-place twin rewrites can break it: every slot,
    // live or dead, must lie strictly above its successor) is read in descending id: the rank is a count of the live entries behind.
    bool clean = false;
    if (C.acnt == 0 && ext <= 64) {   // (wave-uniform)
        const int32_t nxt = __shfl_down(n0, 1);
        clean = __ballot(a0 && (lane + 1 < ext) && !(n0 > nxt)) == 0ull;
    }
    int32_t r0, r1 = 0;
    bool anydup = false;
    if (clean) {
        r0 = popc64(h0 & ~((2ull << lane) - 1ull));
        if (lane == 63) r0 = 0;
        WAVE_SYNC();   // (the raw entries stored by the caller: not read on this path, but the stores below must come after them)
    } else {
    unsigned long long ik[2];
    ik[0] = ((unsigned long long)(live0 ? (uint32_t)n0 : 0x7FFFFFFFu) << 8) | (unsigned)lane;
    ik[1] = ((unsigned long long)(live1 ? (uint32_t)n1 : 0x7FFFFFFFu) << 8) | (unsigned)(64 + lane);
    wave_bitonic128(ik, lane, ext);   // (slots e >= ext hold nothing)
    {
        unsigned long long pv0 = __shfl_up(ik[0], 1), pv1 = __shfl_up(ik[1], 1);
        const unsigned long long edge = __shfl(ik[0], 63);
        if (lane == 0) { pv0 = ~0ull; pv1 = edge; }
        const bool du0 = (ik[0] >> 8) == (pv0 >> 8) && (ik[0] >> 8) != 0x7FFFFFFFull;
        const bool du1 = (ik[1] >> 8) == (pv1 >> 8) && (ik[1] >> 8) != 0x7FFFFFFFull;
        anydup = (__ballot(du0) | __ballot(du1)) != 0ull;
    }
    posmap[(int)(ik[0] & 0xFFull)] = (uint8_t)lane;
    posmap[(int)(ik[1] & 0xFFull)] = (uint8_t)(64 + lane);
    WAVE_SYNC();   // (also: the raw entries stored by the caller)
    r0 = posmap[lane]; r1 = posmap[64 + lane];
    WAVE_SYNC();
    }
    int32_t rk0 = r0, rk1 = r1;
    bool merge = false;
    if (anydup) {
        merge = true;
        if (nlive > 16) {   // (up to 16: insertion sort, the stable order above)
            // std::sort by id of the live entries in traversal order
            const int g0 = popc64(h0 & lt), g1 = popc64(h0) + popc64(h1 & lt);
            if (live0) rec[g0] = (uint16_t)lane;
            if (live1) rec[g1] = (uint16_t)(64 + lane);
            WAVE_SYNC();
            bool ok;
            if (nlive <= 64) {
                const int src = lane < nlive ? (int)rec[lane] : 0;
                WAVE_SYNC();
                double key = (double)C.e[src].nbr;
                int idx = src, pos = lane;
                ok = wave_sort64<false>(key, idx, nlive, lane, tmp64, &pos);
                WAVE_SYNC();
                if (ok && lane < nlive) posmap[idx] = (uint8_t)pos;
            } else {
                ok = wave_lvl_sort<uint16_t, WideIdLess, 2>(rec, nlive, WideIdLess{C.e}, ulist, dlist, tab, tab2, lane);
                if (ok) {
                    if (lane < nlive) posmap[rec[lane]] = (uint8_t)lane;
                    if (64 + lane < nlive) posmap[rec[64 + lane]] = (uint8_t)(64 + lane);
                }
            }
            WAVE_SYNC();
            if (!ok) {   // depth limit of the introsort: the single-vertex path takes this vertex
                if (lane == 0) atomicOr(&C.flags, CF_DUP);
                WAVE_SYNC();
                return;
            }
            if (live0) rk0 = posmap[lane];
            if (live1) rk1 = posmap[64 + lane];
        }
    }
    WAVE_SYNC();   // every read of the raw entries is done
    const uint64_t kb = keyed ? keyed_order_base(A.shuffle_seed, C.v - vbase, 0) : 0ull;
    if (live0) { Ent& E = C.e[rk0]; E.val = v0; E.nbr = n0; E.twin = t0; E.aux = keyed ? keyed_order_dkey(kb, n0 - vbase) : v0; }
    if (live1) { Ent& E = C.e[rk1]; E.val = v1; E.nbr = n1; E.twin = t1; E.aux = keyed ? keyed_order_dkey(kb, n1 - vbase) : v1; }
    WAVE_SYNC();
    int32_t m = nlive;
    if (merge) {
        // fold equal ids: the sum runs in sorted order (:284); the twins of the folded entries die at commit (:289)
        const bool c0 = lane < nlive, c1 = 64 + lane < nlive;
        const int32_t nb0 = c0 ? C.e[lane].nbr : -1, nb1 = c1 ? C.e[64 + lane].nbr : -1;
        const int32_t tw0 = c0 ? C.e[lane].twin : 0, tw1 = c1 ? C.e[64 + lane].twin : 0;
        double s0 = c0 ? C.e[lane].val : 0.0, s1 = c1 ? C.e[64 + lane].val : 0.0;
        const int32_t pv0 = (c0 && lane > 0) ? C.e[lane - 1].nbr : -2, pv1 = c1 ? C.e[63 + lane].nbr : -2;
        const bool hd0 = c0 && nb0 != pv0, hd1 = c1 && nb1 != pv1;
        const uint64_t hm0 = __ballot(hd0), hm1 = __ballot(hd1);
        m = popc64(hm0) + popc64(hm1);
        if (hd0) {
            const uint64_t above = (lane == 63) ? 0ull : (hm0 >> (lane + 1));
            const int nexthead = above ? (lane + 1 + __builtin_ctzll(above)) : (hm1 ? 64 + __builtin_ctzll(hm1) : nlive);
            for (int q = lane + 1; q < nexthead; ++q) s0 += C.e[q].val;
        }
        if (hd1) {
            const uint64_t above = (lane == 63) ? 0ull : (hm1 >> (lane + 1));
            const int nexthead = above ? (64 + lane + 1 + __builtin_ctzll(above)) : nlive;
            for (int q = 64 + lane + 1; q < nexthead; ++q) s1 += C.e[q].val;
        }
        const uint64_t nh0 = __ballot(c0 && !hd0), nh1 = __ballot(c1 && !hd1);
        const int32_t x0 = popc64(hm0 & lt), x1 = popc64(hm0) + popc64(hm1 & lt);
        const int32_t y0 = popc64(nh0 & lt), y1 = popc64(nh0) + popc64(nh1 & lt);
        WAVE_SYNC();
        if (hd0) { Ent& E = C.e[x0]; E.val = s0; E.nbr = nb0; E.twin = tw0; E.aux = keyed ? keyed_order_dkey(kb, nb0 - vbase) : s0; }
        else if (c0) C.e[m + y0].twin = tw0;
        WAVE_SYNC();   // (a lane's first entry may land where another lane's second one was read: all reads are above)
        if (hd1) { Ent& E = C.e[x1]; E.val = s1; E.nbr = nb1; E.twin = tw1; E.aux = keyed ? keyed_order_dkey(kb, nb1 - vbase) : s1; }
        else if (c1) C.e[m + y1].twin = tw1;
        if (lane == 0) { C.m = m; C.nkill = nlive - m; }
        WAVE_SYNC();
    } else if (lane == 0) C.m = nlive;
    if (m == 0) { WAVE_SYNC(); return; }
    // ---- o_n order: ksel[position] = source ----
    const bool c0 = lane < m, c1 = 64 + lane < m;
    const double k0 = c0 ? C.e[lane].aux : 0.0, k1 = c1 ? C.e[64 + lane].aux : 0.0;
    const double kfirst = C.e[0].aux;
    const bool alleq = (__ballot(c0 && k0 != kfirst) | __ballot(c1 && k1 != kfirst)) == 0ull;
    if (alleq) {   // std::sort of all-equal keys: identity up to 16, a fixed permutation above (k_eq_tables)
        if (c0) C.ksel[lane] = (m <= 16) ? (uint8_t)lane : eqg[(m - 17) * 128 + lane];
        if (c1) C.ksel[64 + lane] = eqg[(m - 17) * 128 + 64 + lane];
    } else {
        // stable rank by (key, position): a bitonic network over the keys' bit patterns (weights and keyed keys are positive
        // doubles: the patterns order like the values; complemented for the descending order); exact for m <= 16 (insertion
        // sort) and whenever no two keys are equal
        unsigned long long ok2[2];
        int os2[2] = {lane, 64 + lane};
        ok2[0] = c0 ? (desc ? ~(unsigned long long)__double_as_longlong(k0) : (unsigned long long)__double_as_longlong(k0)) : ~0ull;
        ok2[1] = c1 ? (desc ? ~(unsigned long long)__double_as_longlong(k1) : (unsigned long long)__double_as_longlong(k1)) : ~0ull;
        wave_bitonic128(ok2, os2, lane, m);
        bool anytie;
        {
            unsigned long long pv0 = __shfl_up(ok2[0], 1), pv1 = __shfl_up(ok2[1], 1);
            const unsigned long long edge = __shfl(ok2[0], 63);
            if (lane == 0) pv1 = edge;
            anytie = (__ballot(c0 && lane > 0 && ok2[0] == pv0) | __ballot(c1 && ok2[1] == pv1)) != 0ull;
        }
        if (m <= 16 || !anytie) {
            if (c0) C.ksel[lane] = (uint8_t)os2[0];
            if (c1) C.ksel[64 + lane] = (uint8_t)os2[1];
        } else {
            bool ok;
            if (m <= 64) {
                double key = k0;
                int idx = lane, pos = lane;
                ok = desc ? wave_sort64<true>(key, idx, m, lane, tmp64, &pos) : wave_sort64<false>(key, idx, m, lane, tmp64, &pos);
                if (ok && lane < m) C.ksel[pos] = (uint8_t)idx;
            } else {
                if (c0) rec[lane] = (uint16_t)lane;
                if (c1) rec[64 + lane] = (uint16_t)(64 + lane);
                WAVE_SYNC();
                ok = desc ? wave_lvl_sort<uint16_t, WideAuxGreater, 2>(rec, m, WideAuxGreater{C.e}, ulist, dlist, tab, tab2, lane)
                          : wave_lvl_sort<uint16_t, WideAuxLess, 2>(rec, m, WideAuxLess{C.e}, ulist, dlist, tab, tab2, lane);
                if (ok) { if (c0) C.ksel[lane] = (uint8_t)rec[lane]; if (c1) C.ksel[64 + lane] = (uint8_t)rec[64 + lane]; }
            }
            WAVE_SYNC();
            if (!ok && lane == 0) cand_order_index_call(A, C);   // depth limit: the sequential restatement
        }
    }
    WAVE_SYNC();
    // ---- apply the permutation ----
    double pv0 = 0, pv1 = 0; int32_t pn0 = 0, pn1 = 0, pt0 = 0, pt1 = 0;
    if (c0) { const Ent& Sx = C.e[C.ksel[lane]]; pv0 = Sx.val; pn0 = Sx.nbr; pt0 = Sx.twin; }
    if (c1) { const Ent& Sx = C.e[C.ksel[64 + lane]]; pv1 = Sx.val; pn1 = Sx.nbr; pt1 = Sx.twin; }
    WAVE_SYNC();
    if (c0) { Ent& E = C.e[lane]; E.val = pv0; E.nbr = pn0; E.twin = pt0; E.aux = 0; }
    if (c1) { Ent& E = C.e[64 + lane]; E.val = pv1; E.nbr = pn1; E.twin = pt1; E.aux = 0; }
    if (lane == 0) C.ndraw = (A.o_v == OV_COARSEN) ? 1 : (m > 1 ? m - 1 : 0);
    WAVE_SYNC();
}

__device__ __noinline__ void serial_eliminate_call(const Arrays& A, GraphDesc& G, const ElimScratch& S, int32_t v0, int64_t e1) {
    ColBuf Bf = S.colbuf(G.scr_base);
    int rc = serial_eliminate(A, G, Bf, G.scr_cap, v0, e1);
    if (rc) G.status = rc;
}

// Which positions of my candidate drew ME as their target (ksel[p] == j)?  A group of BC lanes = the positions of one candidate:
// lane p publishes ksel[p] bit by bit with ballots, lane j keeps the positions whose bits spell j.  Replaces the O(m) scan of
// ksel every (candidate, target) thread made in the replay and again when the slots are handed out.  Every lane of the wave calls.
template <int BC>
__device__ __forceinline__ uint64_t group_pick_mask(const uint8_t* ksel, int32_t m, bool cand_ok, int lane) {
    static_assert(BC == 32 || BC == 64, "one candidate per (half-)wave");
    constexpr int BITS = BC == 32 ? 5 : 6;
    const int gl = lane & (BC - 1), hb = lane & (64 - BC);
    const bool vp = cand_ok && gl < m - 1;       // position gl draws a target
    const int t = vp ? (int)ksel[gl] : 0;
    uint64_t mm = __ballot(vp);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint64_t bb = __ballot(vp && ((t >> b) & 1));
        mm &= ((gl >> b) & 1) ? bb : ~bb;
    }
    return (mm >> hb) & (BC == 64 ? ~0ull : 0xFFFFFFFFull);
}

// Specialised on (o_v, o_n): the mode tests fold away, which keeps the round loop's code (executed once
// per round by every wave) small enough for the instruction cache.
template <int OV, int ON, int BC, int NTT>
__global__ __launch_bounds__(NTT, (NTT >= 1024 ? 1 : 4)) void k_eliminate_batch_t(Arrays A_in, GraphDesc* __restrict__ gd, ElimScratch S,
                                                           int32_t* __restrict__ batch_pos, const int32_t* __restrict__ in_flags,
                                                           const double* __restrict__ in_acc) {
    constexpr int NT = NTT;            // threads per workgroup
    // Workgroup barrier.  Debug (RLAP_DEBUG_JITTER=<n>, S.jitter): behind every barrier some waves -- a different set each time -- sleep
    // for n x 0.25 us, so that a value read behind a barrier while another wave already rewrites it (the push-id race of round 2) shows
    // in every run instead of once in some thousand.  Off: one scalar compare per barrier.
    uint32_t jit_ctr = 0;
#define BSYNC() do { __syncthreads(); if (S.jitter > 0) { ++jit_ctr; if (((((uint32_t)threadIdx.x >> 6) * 2654435761u + jit_ctr * 40503u) >> 7 & 3u) == 0u) \
        for (int _q = 0; _q < S.jitter; ++_q) __builtin_amdgcn_s_sleep(8); } } while (0)
    constexpr int NWAVE = NT / 64;
    constexpr int SLOTS = PASSES * NT; // candidates x slots per round
    constexpr int MCAP = NT;           // PQ moves per round
    constexpr int CCAP = NT / 2;       // contended records per round
    constexpr int HBITS = (SLOTS == 4096) ? 12 : (SLOTS == 2048) ? 11 : 10;   // log2(SLOTS): size of the target hash table
    static_assert((1 << HBITS) == SLOTS, "hash table = one place per slot");
    // BC slots per candidate: a group of BC lanes (half a wave or a whole wave) works on one candidate
    constexpr int BCAP = BC;
    constexpr int BATCH = SLOTS / BC;
    typedef CandT<BC> Cand;
    typedef BatchLdsT<BC, NTT> BatchLds;
    constexpr bool MERGE = (BC >= 64 && OV == OV_RANDOM);           // multi-edges merged inside the candidate (rlap_core.h::cand_merges_multi_edges)
    // candidates adjacent to earlier ones of the round are patched in LDS (rlap_core.h::cand_patch).  Measured: pays for the degree
    // order (C3 369 -> 274 ms); with o_v = random (64-slot candidates; wave_patch handles them too) the rounds of config 5 get 6.5 %
    // fewer and no faster, and the coarsening order fills its rounds without it -- so only the degree order uses it.
    constexpr bool PATCH = (OV == OV_DEGREE);
    constexpr int GSH = BC >= 64 ? 0 : 64 - BC;                   // lane & GSH = first lane of my group
    constexpr uint64_t GMASK = BC >= 64 ? ~0ull : 0xFFFFFFFFull;   // a group's part of a 64-bit ballot
    Arrays A = A_in;
    A.o_v = OV;
    A.o_n = ON;
    A.shuffle_seed = A_in.shuffle_seed + (uint64_t)blockIdx.x;   // graph g of a batch: seed + g, ids local to the graph (rlap_core.h)
    A.vbase = gd[blockIdx.x].vbase;
    __shared__ ElimSharedT<BC, NTT> sh;
    __shared__ GraphDesc G;
    __shared__ int32_t s_help[3];      // hand-over between the eliminating wave and its helper (single_helper); [2]: the helper gave up waiting
    __shared__ int32_t s_nc, s_pmax, s_p, s_ncont, s_nmoves, s_status, s_nhigh, s_nlow, s_pool_cur, s_pool_end, s_anydep, s_npatched;
    // chunk directory of the bucket the pops come from, kept in LDS between rounds: the prediction's stack window then starts with
    // the entries themselves instead of a dependent look-up of their chunk base (one global round trip less per round).
    // s_dirb = that bucket (global index), -1 none; refreshed when the bucket changes, a push allocates a chunk for it (push phase,
    // same thread) or the single-vertex path has pushed (it allocates through bs_ensure)
    __shared__ int32_t s_dirb;
    __shared__ int32_t s_dir[BDIR];
    constexpr int32_t POOL_GRAB = NT >= 1024 ? POOL_GRAB_BIG : POOL_GRAB_SMALL;   // slots reserved at a time for the round's appends
    // std::sort's permutation of n = 17..BC all-equal keys and its inverse (final position of the entry with id-rank r):
    // in LDS for the 1024-thread shape; the 256-thread shape reads the handle's tables (k_eq_tables) through the L1,
    // which leaves its LDS at 39 KB -- four workgroups per CU
    constexpr bool EQG = NTT < 1024 || BC == 128;
    constexpr int EC1 = NTT >= 1024 ? ECAP : ECAP_SMALL;   // single-vertex path in LDS up to this extent
    __shared__ uint8_t s_eqperm[EQG ? 1 : BC - 16][EQG ? 1 : BC];
    __shared__ uint8_t s_eqinv[EQG ? 1 : BC - 16][EQG ? 1 : BC];
    const uint8_t* __restrict__ eqg = S.eqtab + (BC == 64 ? 0 : (BC == 32 ? EQTAB_OFF32 : EQTAB_OFF128));
    BatchLds& L = sh.b;
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    if (S.poison >= 0) {   // debug: LDS starts as this byte, not as what the previous workgroup on this CU left behind
        uint32_t* const w = reinterpret_cast<uint32_t*>(&sh);
        const uint32_t pat = 0x01010101u * (uint32_t)(S.poison & 0xFF);
        for (size_t q = tid; q < sizeof(sh) / 4; q += NT) w[q] = pat;
        BSYNC();
    }
    if (tid == 0) {
        G = gd[g]; s_nc = 0; s_pmax = BATCH + 1; s_p = BATCH + 1; s_ncont = 0; s_nmoves = 0; s_pool_cur = 0; s_pool_end = 0; s_anydep = 0; s_npatched = 0;
        s_help[0] = 0; s_help[1] = 0; s_help[2] = 0; s_dirb = -1;
        // the input checks of the setup kernels are read here, not on the host (no mid-call synchronisation): bad input -> nothing is eliminated
        int32_t bad = 0;
        if (in_flags[FLAG_RANGE]) bad = ST_INDEX_RANGE;
        else if (in_flags[FLAG_CROSS] || in_flags[FLAG_PERM]) bad = ST_BAD_ARG;
        else if (in_acc[2] != 0.0 || !(in_acc[0] <= 1e-24 * in_acc[1])) bad = ST_NOT_SYMMETRIC;   // isApprox(A^T), factorizers.cc:19-22
        s_status = bad;
    }
    if constexpr (!EQG) {
        for (int base = 0; base < BC - 16; base += BATCH) {   // the candidate records serve as scratch: BATCH lengths at a time
            const int row = base + tid;
            if (tid < BATCH && row < BC - 16) {
                Cand& C = L.cand[tid];
                const int nn = 17 + row;
                for (int q = 0; q < nn; ++q) { C.e[q].aux = 1.0; C.ksel[q] = (uint8_t)q; }
                C.m = nn;
                Arrays A3 = A;
                cand_order_index_call(A3, C);   // all keys equal: asc and desc compare alike
                for (int q = 0; q < BC; ++q) s_eqperm[row][q] = q < nn ? C.ksel[q] : (uint8_t)q;
                for (int q = 0; q < BC; ++q) s_eqinv[row][s_eqperm[row][q]] = (uint8_t)q;
            }
        }
    }
    BSYNC();
    const int32_t n = G.n;
    const bool use_pq = A.o_v != OV_RANDOM;
    int64_t nelim = G.t < (int64_t)(n - 1) ? G.t : (int64_t)(n - 1);
    if (nelim < 0 || s_status != 0) nelim = 0;
    int64_t done = 0;
    int32_t rounds = 0, singles = 0;
    bool pending_long = false;   // o_v = random: the last round was cut by a long column, which is therefore the next vertex
    long long t_prev = 0;
    __shared__ long long s_prof[40];   // diagnostic build only (S.prof != nullptr)
#define PHASE_STAMP(k) do { if (S.prof && tid == 0 && g == 0) { long long _t = wall_clock64(); s_prof[k] += _t - t_prev; t_prev = _t; } } while (0)
    long long clk0 = 0, wall0 = 0;
    if (S.prof && tid == 0 && g == 0) { for (int q = 0; q < 40; ++q) s_prof[q] = 0; t_prev = wall_clock64(); clk0 = clock64(); wall0 = t_prev; }

    while (done < nelim) {
        // per-round opaque copy of the thread id: keeps the compiler from hoisting the (cheap) per-pass index
        // and LDS-address arithmetic out of the round loop, where it would stay live across every barrier
        // and get spilled to scratch memory
        int tid_round = threadIdx.x;
        asm volatile("" : "+v"(tid_round));
        const int tid = tid_round;
        const int lane = tid & 63;
        // thread <-> (candidate, position) in the replay and commit phases.  64-slot candidates (o_v = random): a wave is one candidate,
        // so that its lanes can match positions and targets with ballots (group_pick_mask; config 5: 19.9 -> 18.6 ms).  The others keep
        // the interleaved layout (8 or 32 lanes per candidate and pass) and scan ksel: with 32-slot candidates the grouped layout keeps all
        // four passes busy where the interleaved one skips the last two for short candidates (measured: C3 285 -> 305 ms with it)
        constexpr bool GROUPED = (BC == 64);
        auto PI = [&](int k) -> int32_t { return GROUPED ? (k * NT + tid) / BCAP : tid / (NT / BATCH); };
        auto PJ = [&](int k) -> int32_t { return GROUPED ? (k * NT + tid) % BCAP : k * (NT / BATCH) + tid % (NT / BATCH); };
        if (OV == OV_RANDOM && pending_long) {
            // the vertex that cut the previous round goes straight to the single-vertex path: no round is set up around it
            pending_long = false;
            ++singles;
            const int32_t v0 = G.vbase + (int32_t)A.perm[G.vbase + n - (done + 1)];
            if (tid < 64) {
                Arrays A2 = A;
                ElimScratch S2 = S;
                int32_t cp0 = A2.colptr[v0], cp1 = A2.colptr[v0 + 1];
                int32_t acnt = A2.vr[v0].app_cnt, abase = A2.vr[v0].app_chunk;
                if ((cp1 - cp0) + acnt > EC1) {
                    ColBuf Bf = S2.colbuf(G.scr_base);
                    const bool handled = wave_eliminate_big(A2, G, sh.g, Bf, G.scr_cap, v0, cp0, cp1, acnt, abase, s_help);
                    if (!handled && lane == 0) {   // (a column the long-column path took has told the helper itself)
                        // no help wanted: said BEFORE the one-lane elimination, which may last longer than the helper's wait
                        __hip_atomic_store(&s_help[0], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        serial_eliminate_call(A2, G, S2, v0, done + 1);
                    }
                } else {
                    wave_eliminate(A2, G, sh.e, v0, done + 1, cp0, cp1, acnt, abase, (S.prof && g == 0) ? &s_prof[24] : nullptr, s_help);
                }
            } else if (tid < 128) {
                single_helper(sh.e, sh.g.c.newv, s_help);
            }
            BSYNC();
            if (G.status != 0 || s_help[2] != 0) { if (tid == 0) s_status = G.status ? G.status : ST_INTERNAL; BSYNC(); break; }
            done += 1;
            PHASE_STAMP(5);
            continue;
        }
        ++rounds;
        const int32_t Bcur = (int32_t)((nelim - done) < (int64_t)BATCH ? (nelim - done) : (int64_t)BATCH);
        // ================= P0: predict the next pops =================
        // (the round counters were reset before the barrier that ended the previous round)
        int32_t b = 0;
        if (use_pq) {
            while (true) {
                b = G.bucket_base + G.minlist;
                const int32_t cnt = A.bs_cnt[b];
                int32_t oc0 = A.ocur[b];
                const int32_t oe = A.oend[b];
                if (s_dirb != b) {   // (workgroup-uniform: s_dirb changes only in front of a barrier)
                    if (tid < BDIR) s_dir[tid] = A.bs_dir[(int64_t)b * BDIR + tid];
                    BSYNC();
                    if (tid == 0) s_dirb = b;
                }
                // moved members, newest first, then the never-moved members in descending id.  The window of
                // the stack that reaches its bottom is read together with the first window of the never-moved
                // members (both load chains in flight at once, one block scan for the two).
                int32_t top = cnt;   // entries [0,top) not yet scanned
                while (s_nc < Bcur && (top > 0 || oc0 < oe)) {
                    const int32_t a = top - 1 - tid;
                    bool vs = false, vo = false;
                    int32_t v_s = -1, v_o = -1;
                    if (a >= 0) {
                        const int cdir = bs_chunk_of(a);
                        const int32_t sl = s_dir[cdir] + (a - bs_chunk_start(cdir));
                        v_s = A.bs_v[sl];
                        vs = (A.vr[v_s].pqpos == A.bs_id[sl]);
                    }
                    const bool last_stack = top <= NT;
                    const int32_t oc = oc0 + tid;
                    if (last_stack && oc < oe) { v_o = A.orig_order[oc]; vo = (A.vr[v_o].pqpos == -1); }
                    int tot;
                    const int r = block_excl_scan<NWAVE>((vs ? 1 : 0) | (vo ? (1 << 16) : 0), L.scan, &tot);
                    const int rs = r & 0xFFFF, ro = r >> 16, ts = tot & 0xFFFF, to = tot >> 16;
                    const int base = s_nc;
                    BSYNC();
                    if (vs && base + rs < Bcur) { L.cand[base + rs].v = v_s; L.cand[base + rs].src = a; }
                    if (vo && base + ts + ro < Bcur) { L.cand[base + ts + ro].v = v_o; L.cand[base + ts + ro].src = ~oc; }
                    if (tid == 0) s_nc = (base + ts + to < Bcur) ? base + ts + to : Bcur;
                    BSYNC();
                    top = top > NT ? top - NT : 0;
                    if (last_stack) oc0 += NT;
                }
                if (s_nc > 0) break;
                BSYNC();
                if (tid == 0) {
                    A.bs_cnt[b] = 0; A.ocur[b] = oe;
                    G.minlist += 1;
                    if (G.minlist > 2 * n) s_status = ST_INTERNAL;
                }
                BSYNC();
                if (s_status) break;
            }
            if (s_status) break;
        } else {
            if (tid < Bcur) { L.cand[tid].v = G.vbase + (int32_t)A.perm[G.vbase + n - (done + tid + 1)]; L.cand[tid].src = 0; }
            if (tid == 0) s_nc = Bcur;
            BSYNC();
        }
        const int32_t nc = s_nc;
        PHASE_STAMP(0);

        // ================= P1: prepare =================
        // (a) per candidate: column extent + chunk bases; (b) per slot: the loads; (c) per candidate: filter + sorts
        if (tid < nc) {
            int32_t v = L.cand[tid].v, src = L.cand[tid].src;
            cand_meta(A, v, L.cand[tid]);
            L.cand[tid].src = src;
            batch_pos[v] = tid;
            if (L.cand[tid].flags & CF_BIG) atomicMin(&s_pmax, tid);   // a long column cuts the round: nothing is loaded behind it
        }
        BSYNC();
        const bool first_is_big = (L.cand[0].flags & CF_BIG) != 0;   // goes to the single-vertex path: skip this round's prepare
        if (!first_is_big) {
            // BATCH*BCAP = 4*NT slots: issue every load before the first LDS store
            static_assert(BATCH * BCAP == PASSES * NT && (BC == 32 || BC == 64 || BC == 128), "slot loops are unrolled for PASSES passes");
            // slot of pass k -> (candidate, entry).  128-slot candidates: a wave owns a candidate, its entries lane and 64 + lane are
            // the wave's passes 2c and 2c+1 (everything candidate-local stays inside one wave)
            auto SI = [&](int k) -> int32_t { return BC == 128 ? ((k >> 1) * NWAVE + (tid >> 6)) : (k * NT + tid) / BCAP; };
            auto SE = [&](int k) -> int32_t { return BC == 128 ? (((k & 1) << 6) + (tid & 63)) : (k * NT + tid) % BCAP; };
            double lv[PASSES]; int32_t ln[PASSES], lt[PASSES]; bool la[PASSES];
            const int32_t cut0 = (OV == OV_RANDOM) ? s_pmax : nc;   // o_v = random: the first long column (most rounds have one) is known already
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = SI(k), e = SE(k);
                la[k] = (i < nc) && (OV != OV_RANDOM || i < cut0) && (e < L.cand[i].ext);
                lv[k] = 0; ln[k] = 0; lt[k] = 0;
                if (la[k]) {
                    int32_t sl = cand_slot(L.cand[i], e);
                    const Slot g = A.e[sl]; lv[k] = g.val; ln[k] = g.nbr; lt[k] = g.twin;
                }
            }
            // dependence: is a live neighbour an earlier candidate of this round?  Looked up now (only the ids are needed), so
            // that the cut is known before the sorts: nothing more is prepared for the candidates behind it.
            int32_t bpk[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) { bpk[k] = -1; if (la[k] && lv[k] > 0) bpk[k] = batch_pos[ln[k]]; }
            // Shortcut for the common column: no appended entries, CSR segment still sorted, all live
            // weights equal, weight order.  Read backwards it is in descending id, so the id rank is a
            // popcount of the live mask; std::sort of all-equal keys is the identity up to 16 entries and a
            // fixed permutation above: every entry goes straight to its final position.
            const bool keyed = (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN);
            bool ready[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) ready[k] = false;
            if constexpr (BC <= 64) {
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t idx = k * NT + tid;
                const int32_t i = idx / BCAP, e = idx % BCAP;
                const int hb = lane & GSH;
                const bool live = la[k] && lv[k] > 0;
                const uint64_t half = (__ballot(live) >> hb) & GMASK;
                const int32_t nxt = __shfl_down(ln[k], 1);
                const bool cand_ok = (i < nc) && (OV != OV_RANDOM || i < cut0) && !(L.cand[i].flags & CF_BIG);
                const int32_t ext = cand_ok ? L.cand[i].ext : 0;
                const bool unsorted = la[k] && (e + 1 < ext) && !(ln[k] > nxt);
                const uint64_t hbad = (__ballot(unsorted) >> hb) & GMASK;
                const int f = half ? __builtin_ctzll(half) : 0;
                const double w0 = __shfl(lv[k], hb + f);
                const uint64_t hdw = (__ballot(live && lv[k] != w0) >> hb) & GMASK;
                ready[k] = cand_ok && !keyed && L.cand[i].acnt == 0 && hbad == 0 && hdw == 0;
                if (ready[k]) {
                    Cand& C = L.cand[i];
                    const int32_t m = __popcll(half);
                    if (live) {
                        const int32_t r = __popcll(half & ~((2ull << e) - 1ull));
                        const int32_t pos = (m <= 16) ? r : (int32_t)(EQG ? eqg[(BC - 16) * BC + (m - 17) * BC + r] : s_eqinv[EQG ? 0 : m - 17][EQG ? 0 : r]);
                        Ent& E = C.e[pos];
                        E.val = lv[k]; E.nbr = ln[k]; E.twin = lt[k]; E.aux = 0;
                    }
                    if (e == 0) {
                        C.m = m;
                        C.ndraw = m > 1 ? m - 1 : 0;
                        C.flags |= CF_READY;
                    }
                    la[k] = false;   // nothing left to do for this slot
                }
            }
            }
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                if (la[k]) {
                    Ent& E = L.cand[SI(k)].e[SE(k)];
                    E.val = lv[k]; E.nbr = ln[k];
                }
            }
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = SI(k);
                if (bpk[k] >= 0 && bpk[k] < i) {
                    if (PATCH) {
                        // adjacent to an earlier candidate of the round: remembered, patched after that one is sampled (wave_patch)
                        const int32_t q = atomicAdd(&L.cand[i].ndep, 1);
                        if (q < DEPMAX) L.cand[i].dep[q] = (uint8_t)bpk[k]; else atomicMin(&s_pmax, i);
                        s_anydep = 1;
                    } else atomicMin(&s_pmax, i);
                }
            }
            if (tid < nc && (L.cand[tid].flags & CF_BIG)) atomicMin(&s_pmax, tid);
            BSYNC();
            const int32_t ncp = s_pmax < nc ? s_pmax : nc;   // candidates [ncp, nc) are behind the cut
#pragma unroll
            for (int k = 0; k < PASSES; ++k) if (SI(k) >= ncp) { la[k] = false; ready[k] = false; }
            PHASE_STAMP(9);
            if constexpr (BC == 128) {
                // 128-slot candidates: the wave prepares its (at most two) candidates one after the other (cand_prepare_wide);
                // sort scratch in the hash / slot tables, which are idle until the barrier that ends P1
                uint8_t* const wscr = reinterpret_cast<uint8_t*>(L.hcnt) + (size_t)(tid >> 6) * WIDE_SCR_BYTES;
                static_assert((size_t)NWAVE * WIDE_SCR_BYTES <= sizeof(int32_t) * 8 * (size_t)NT, "scratch of the waves fits hcnt + pslot");
#pragma unroll
                for (int cc = 0; cc < PASSES / 2; ++cc) {
                    const int32_t i = cc * NWAVE + (tid >> 6);
                    if (i < ncp && !(L.cand[i].flags & CF_BIG)) {   // (wave-uniform)
                        cand_prepare_wide(A.o_v, A.o_n, A.shuffle_seed, L.cand[i], G.vbase, lv[2 * cc], ln[2 * cc], lt[2 * cc], la[2 * cc], lv[2 * cc + 1], ln[2 * cc + 1], lt[2 * cc + 1], la[2 * cc + 1], wscr, eqg);
                    }
                }
            } else {
            // rank of every live entry among its column's live entries by id: with distinct ids the
            // sorted order is unique, so no std::sort emulation is needed (equal ids -> single-vertex path).
            // One half-wave = one candidate.  A column without appended entries whose CSR segment is still
            // sorted is read in descending id, so its rank is a popcount of the live mask; others loop.
            int32_t rk[PASSES];
            bool merge_k[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                rk[k] = -1; merge_k[k] = false;
                if (__ballot(la[k]) == 0ull) continue;   // wave-uniform: nothing left in this pass for this wave
                const int32_t idx = k * NT + tid;
                const int32_t i = idx / BCAP, e = idx % BCAP;
                const bool live = la[k] && lv[k] > 0;
                const uint64_t half = (__ballot(live) >> (lane & GSH)) & GMASK;
                // in-place twin rewrites (:404-406) can leave a CSR segment unsorted: the shortcut needs
                // every slot (live or dead) strictly above its successor
                const int32_t nxt = __shfl_down(ln[k], 1);
                const bool unsorted = la[k] && (i < ncp) && (e + 1 < L.cand[i].ext) && !(ln[k] > nxt);
                const uint64_t hbad = (__ballot(unsorted) >> (lane & GSH)) & GMASK;
                rk[k] = -1;
                if (!ready[k] && i < ncp && L.cand[i].ext > 0) {
                    Cand& C = L.cand[i];
                    if (C.acnt == 0 && hbad == 0) {
                        if (live) rk[k] = __popcll(half & ~((2ull << e) - 1ull));
                        if (e == 0) C.m = __popcll(half);
                    } else if (!MERGE) {
                        // columns with appended entries (or an unsorted CSR segment): rank by id with a bitonic network over the
                        // half-wave's registers -- (id, lane) pairs, dead entries last; equal ids = multi-edge -> single-vertex path
                        uint32_t skey = live ? (uint32_t)ln[k] : 0x7FFFFFFFu;
                        int ssrc = e;
#pragma unroll
                        for (int kk = 2; kk <= BC; kk <<= 1) {
#pragma unroll
                            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                                const uint32_t okey = (uint32_t)__shfl_xor((int)skey, jj);
                                const int osrc = __shfl_xor(ssrc, jj);
                                const bool keep_min = (((e & jj) == 0) == ((e & kk) == 0));
                                const bool other_less = okey < skey || (okey == skey && osrc < ssrc);
                                if (keep_min == other_less) { skey = okey; ssrc = osrc; }
                            }
                        }
                        // lane p now holds the p-th pair: tell the source lane its rank (forward permute), look for equal neighbours
                        const uint32_t pkey = (uint32_t)__shfl_up((int)skey, 1);
                        const bool dupp = e > 0 && skey == pkey && skey != 0x7FFFFFFFu;
                        const int myrank = __builtin_amdgcn_ds_permute(((lane & GSH) + ssrc) << 2, e);
                        if (__ballot(dupp) >> (lane & GSH) & GMASK) { if (e == 0) atomicOr(&C.flags, CF_DUP); }
                        if (live) rk[k] = myrank;
                        if (e == 0) C.m = __popcll(half);
                    } else {
                        // one wave = this candidate.  Multi-edges are merged here (no PQ to replay in this mode): the
                        // position std::sort by id gives every live entry (stable rank up to 16 entries, the
                        // introsort restatement above), then merge_k[] below folds equal ids.
                        const int32_t ext = C.ext, me = ln[k];
                        int32_t r = 0, eqb = 0;
                        bool dup = false;
                        if (live) {
                            for (int32_t q = 0; q < ext; ++q) {
                                bool lq = C.e[q].val > 0;
                                int32_t nq = C.e[q].nbr;
                                r += (lq && nq < me) ? 1 : 0;
                                eqb += (lq && nq == me && q < e) ? 1 : 0;
                                dup |= (lq && nq == me && q != e);
                            }
                        }
                        const int32_t nlive = __popcll(half);
                        const bool anydup = __ballot(live && dup) != 0ull;
                        if (!anydup) {
                            if (live) rk[k] = r;
                        } else if (nlive <= 16) {
                            if (live) rk[k] = r + eqb;
                            merge_k[k] = true;
                        } else {
                            int32_t* tmp = L.pslot + (tid >> 6) * 256;   // pslot is not live before the commit
                            if (live) tmp[144 + __popcll(half & lanemask_lt(lane))] = lane;   // gather order = traversal order
                            WAVE_SYNC();
                            const int src = lane < nlive ? tmp[144 + lane] : 0;
                            double key = (double)__shfl(me, src);
                            int idx = src, pos = lane;
                            const WS64 wr = wave_sort64_call<false>(key, idx, nlive, lane, tmp);
                            key = wr.key; idx = wr.idx; pos = wr.pos;
                            const bool ok = wr.ok != 0;
                            WAVE_SYNC();
                            if (lane < nlive) tmp[144 + idx] = pos;
                            WAVE_SYNC();
                            if (ok) { if (live) rk[k] = tmp[144 + lane]; merge_k[k] = true; }
                            else if (lane == 0) atomicOr(&C.flags, CF_DUP);   // depth limit hit: single-vertex path
                            WAVE_SYNC();
                        }
                        if (lane == 0) C.m = nlive;
                    }
                }
            }
            WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave
            PHASE_STAMP(12);
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                if (__ballot(rk[k] >= 0) == 0ull) continue;
                if (rk[k] >= 0) {
                    const int32_t idx = k * NT + tid;
                    Cand& C = L.cand[idx / BCAP];
                    Ent& E = C.e[rk[k]];
                    E.val = lv[k]; E.nbr = ln[k]; E.twin = lt[k];
                    E.aux = keyed ? keyed_order_dkey(keyed_order_base(A.shuffle_seed, C.v - G.vbase, 0), ln[k] - G.vbase) : lv[k];
                }
            }
            if (MERGE) {
#pragma unroll
                for (int k = 0; k < PASSES; ++k) {
                    if (!merge_k[k]) continue;   // wave-uniform
                    WAVE_SYNC();
                    Cand& C = L.cand[k * NWAVE + (tid >> 6)];
                    const int32_t nl = C.m;
                    const bool act = lane < nl;
                    const int32_t nb = act ? C.e[lane].nbr : -1;
                    const int32_t tw = act ? C.e[lane].twin : 0;
                    double val = act ? C.e[lane].val : 0.0;
                    const int32_t nbprev = __shfl_up(nb, 1);
                    const bool head = act && (lane == 0 || nb != nbprev);
                    const uint64_t hm = __ballot(head), lt = lanemask_lt(lane);
                    const int32_t m = __popcll(hm);
                    if (head) {   // the sum runs in sorted order (:284)
                        const uint64_t above = (lane == 63) ? 0ull : (hm >> (lane + 1));
                        const int nexthead = above ? (lane + 1 + __builtin_ctzll(above)) : nl;
                        for (int q = lane + 1; q < nexthead; ++q) val += C.e[q].val;
                    }
                    const int32_t x = __popcll(hm & lt);
                    const int32_t y = __popcll(__ballot(act && !head) & lt);
                    WAVE_SYNC();
                    if (head) {
                        Ent& E = C.e[x];
                        E.val = val; E.nbr = nb; E.twin = tw;
                        E.aux = keyed ? keyed_order_dkey(keyed_order_base(A.shuffle_seed, C.v - G.vbase, 0), nb - G.vbase) : val;
                    } else if (act) {
                        C.e[m + y].twin = tw;   // dies at commit (:289)
                    }
                    if (lane == 0) { C.m = m; C.nkill = nl - m; }
                }
            }
            PHASE_STAMP(15);
            PHASE_STAMP(18);
            WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave
            PHASE_STAMP(16);
            // o_n order with std::sort semantics under ties.  All-equal keys (unit weights): identity for
            // m <= 16, a precomputed permutation above.  Otherwise a stable rank is exact for m <= 16
            // (pure insertion sort) and whenever the keys are distinct; the rest is emulated by one thread.
            const bool desc = (A.o_n == ON_DESC) && !keyed;
            int32_t r2[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                r2[k] = -1;
                const int32_t idx = k * NT + tid;
                const int32_t i = idx / BCAP, j = idx % BCAP;
                const bool valid = (i < ncp) && !(L.cand[i].flags & (CF_BIG | CF_DUP | CF_READY)) && (j < L.cand[i].m);
                if (__ballot(valid) == 0ull) continue;
                double kme = 0, k0 = 0;
                if (valid) { kme = L.cand[i].e[j].aux; k0 = L.cand[i].e[0].aux; }
                const uint64_t hdiff = (__ballot(valid && kme != k0) >> (lane & GSH)) & GMASK;
                if (valid) {
                    Cand& C = L.cand[i];
                    const int32_t m = C.m;
                    if (hdiff == 0) {
                        C.ksel[j] = (m <= 16) ? (uint8_t)j : (EQG ? eqg[(m - 17) * BC + j] : s_eqperm[EQG ? 0 : m - 17][EQG ? 0 : j]);
                    } else {
                        int32_t r = 0; bool tie = false;
                        for (int32_t q = 0; q < m; ++q) {
                            const double kq = C.e[q].aux;
                            const bool eq = (kq == kme);
                            const bool before = desc ? (kq > kme) : (kq < kme);
                            r += (before || (eq && q < j)) ? 1 : 0;
                            tie |= (eq && q != j);
                        }
                        r2[k] = r;
                        if (tie) atomicOr(&C.flags, CF_TIE);
                        if (j == 0) atomicOr(&C.flags, CF_NEQ);
                    }
                }
            }
            PHASE_STAMP(17);
            WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave
            PHASE_STAMP(13);
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                if (__ballot(r2[k] >= 0) == 0ull) continue;
                if (r2[k] >= 0) {
                    const int32_t idx = k * NT + tid;
                    Cand& C = L.cand[idx / BCAP];
                    if (C.m <= 16 || !(C.flags & CF_TIE)) C.ksel[r2[k]] = (uint8_t)(idx % BCAP);
                }
            }
            if (BC == 64) {
                // one wave = one candidate: std::sort's permutation with the elements in registers
                WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave   // rank writes of ksel above vs. the sort's own
                int32_t* tmp = L.pslot + (tid >> 6) * 256;   // pslot is not live before the commit
#pragma unroll 1
                for (int k = 0; k < PASSES; ++k) {
                    const int32_t i = k * NWAVE + (tid >> 6);
                    if (i >= ncp) break;
                    Cand& C = L.cand[i];
                    const int32_t fl = C.flags, m = C.m;
                    if ((fl & (CF_BIG | CF_DUP | CF_READY)) || !(m > 16 && (fl & CF_TIE) && (fl & CF_NEQ))) continue;
                    double key = lane < m ? C.e[lane].aux : 0.0;
                    int idx = lane, pos = lane;
                    const WS64 wr = desc ? wave_sort64_call<true>(key, idx, m, lane, tmp) : wave_sort64_call<false>(key, idx, m, lane, tmp);
                    key = wr.key; idx = wr.idx; pos = wr.pos;
                    const bool ok = wr.ok != 0;
                    if (ok) { if (lane < m) C.ksel[pos] = (uint8_t)idx; }
                    else if (lane == 0) { Arrays A3 = A; cand_order_index_call(A3, C); }
                    WAVE_SYNC();
                }
            } else {
                // half a wave = one candidate: the same with two independent sorts per wave
                WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave   // rank writes of ksel above vs. the sort's own
                int32_t* tmp = L.pslot + (tid >> 6) * 256;   // pslot is not live before the commit
#pragma unroll 1
                for (int k = 0; k < PASSES; ++k) {
                    const int32_t i = (k * NT + tid) / BCAP;
                    const int gl = lane & (BC - 1);
                    bool want = false;
                    int32_t m = 0;
                    if (i < ncp) {
                        const int32_t fl = L.cand[i].flags;
                        m = L.cand[i].m;
                        want = !(fl & (CF_BIG | CF_DUP | CF_READY)) && m > 16 && (fl & CF_TIE) && (fl & CF_NEQ);
                    }
                    if (__ballot(want) == 0ull) continue;
                    Cand& C = L.cand[i < ncp ? i : 0];
                    double key = (want && gl < m) ? C.e[gl].aux : 0.0;
                    int idx = gl, pos = gl;
                    const bool ok = desc ? group_sort<true, BC>(key, idx, m, want, lane, tmp, &pos) : group_sort<false, BC>(key, idx, m, want, lane, tmp, &pos);
                    if (want) {
                        if (ok) { if (gl < m) C.ksel[pos] = (uint8_t)idx; }
                        else if (gl == 0) { Arrays A3 = A; cand_order_index_call(A3, C); }
                    }
                    WAVE_SYNC();
                }
            }
            WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave
            PHASE_STAMP(14);
            // apply the permutation: every position fetches its source entry, then all store
            double pv[PASSES]; int32_t pn[PASSES], pt[PASSES]; bool pa[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t idx = k * NT + tid;
                const int32_t i = idx / BCAP, j = idx % BCAP;
                pa[k] = (i < ncp) && !(L.cand[i].flags & (CF_BIG | CF_DUP | CF_READY)) && (j < L.cand[i].m);
                pv[k] = 0; pn[k] = 0; pt[k] = 0;
                if (__ballot(pa[k]) == 0ull) continue;
                if (pa[k]) {
                    const Ent& Sx = L.cand[i].e[L.cand[i].ksel[j]]; pv[k] = Sx.val; pn[k] = Sx.nbr; pt[k] = Sx.twin;
                    if (j == 0) { const int32_t m = L.cand[i].m; L.cand[i].ndraw = (A.o_v == OV_COARSEN) ? 1 : (m > 1 ? m - 1 : 0); }   // m >= 1 here
                }
            }
            WAVE_SYNC();   // candidate-local: producer and consumer are the same (half-)wave
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                if (__ballot(pa[k]) == 0ull) continue;
                if (pa[k]) {
                    const int32_t idx = k * NT + tid;
                    Ent& E = L.cand[idx / BCAP].e[idx % BCAP];
                    E.val = pv[k]; E.nbr = pn[k]; E.twin = pt[k]; E.aux = 0;
                }
            }
            }   // BC <= 64
        }
        BSYNC();
        PHASE_STAMP(1);
        // targets shared by several candidates are found with an LDS hash table (keys over pslot, counts over the
        // move arrays: both idle until the commit); cleared here, filled and read in P4 behind later barriers
        int32_t* const hkeys = L.pslot;
        int32_t* const hcnt = L.hcnt;
#pragma unroll
        for (int k = 0; k < PASSES; ++k) { hkeys[k * NT + tid] = -1; hcnt[k * NT + tid] = 0; }
        if (tid < BATCH) L.cmask[tid] = 0ull;
        // ================= P1b: first candidate that depends on an earlier one =================
        if (tid < nc && (L.cand[tid].flags & (CF_BIG | CF_DUP))) atomicMin(&s_pmax, tid);   // (multi-edges show up during the sorts)
        // RNG offsets (P2): a prefix sum does not depend on what follows it, so it is taken over all candidates of the round
        int dtot;
        const int dex = block_excl_scan<NWAVE>((tid < nc && !first_is_big) ? L.cand[tid].ndraw : 0, L.scan, &dtot);
        BSYNC();
        int32_t Pmax = first_is_big ? 0 : (s_pmax < nc ? s_pmax : nc);
        int32_t P = 0;
        PHASE_STAMP(2);
        if (Pmax > 0) {
            // (dtot covers the whole round: at most 4k uniforms more than the prefix needs; the table has that slack)
            if (G.n_draws + dtot > A.rng_len) { if (tid == 0) s_status = ST_RNG_OVERFLOW; BSYNC(); break; }
            // ================= P3: sampling =================
            if (tid < Pmax) {
                L.cand[tid].draw0 = G.n_draws + dex;
                if (!PATCH || L.cand[tid].ndep == 0) cand_cumsum(A, L.cand[tid]);
            }
            BSYNC();
            if (A.o_v != OV_COARSEN) {
                #pragma unroll 1
                for (int k = 0; k < PASSES; ++k) {   // slot-major pairs, rolled (code size): waves whose candidates are short skip later passes
                    const int32_t i = tid / (NT / BATCH), j = k * (NT / BATCH) + tid % (NT / BATCH);
                    if (i >= Pmax || (PATCH && L.cand[i].ndep != 0)) continue;
                    if (j < L.cand[i].m - 1) cand_pick(A, L.cand[i], j);
                }
                // the recurrence (touches e[].val only) runs on the LAST waves, whose share of the picks above is
                // empty unless the round is nearly full: it overlaps with the other waves' picks
                if (tid >= NT - BATCH && tid - (NT - BATCH) < Pmax && (!PATCH || L.cand[tid - (NT - BATCH)].ndep == 0)) cand_recur(A, L.cand[tid - (NT - BATCH)]);
            }
            BSYNC();
            PHASE_STAMP(3);
            if constexpr (PATCH) { if (s_anydep) {
                // ---- dependent candidates: patched from the sampled records of the earlier candidates they are adjacent to, ordered
                //      again and sampled, one wave each; a candidate waits until the ones it depends on are through (a chain of
                //      dependent candidates takes one pass of this loop per link).  What cannot be patched cuts the round. ----
                int32_t* const ptmp = reinterpret_cast<int32_t*>(L.cont) + (tid >> 6) * 160;   // the record lists are idle until the replay
                static_assert(sizeof(CRec) * CCAP * 2 >= (size_t)NWAVE * 160 * 4, "sort scratch of the patching waves fits the record lists");
                // No workgroup barrier per link of a chain: a wave takes its candidates in increasing order and spins on the flags of
                // the (smaller) dependent candidates it needs; the smallest unresolved one can always proceed, so every wave gets through.
                #pragma unroll 1
                for (int32_t i = (tid >> 6); i < Pmax; i += NWAVE) {
                    Cand& C = L.cand[i];
                    const int32_t nd = C.ndep;
                    if (nd == 0) continue;     // (wave-uniform)
                    bool lost = false;
                    for (int32_t q = 0; q < nd && q < DEPMAX; ++q) {
                        Cand& J = L.cand[C.dep[q]];
                        if (J.ndep == 0) continue;
                        volatile int32_t* fp = &J.flags;
                        int32_t f = *fp, spins = 0;
                        while (!(f & (CF_PATCHED | CF_DEP))) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT) { f = CF_DEP; if (lane == 0) s_status = ST_INTERNAL; break; }   // (a stuck wave must not hang the device -- and must not pass for a result)
                            f = *fp;
                        }
                        if (f & CF_DEP) lost = true;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (i >= *(volatile int32_t*)&s_pmax) lost = true;     // behind a cut already
                    bool okp = false;
                    if (!lost) okp = wave_patch<ON>(A.o_v, A.shuffle_seed, A.rng, A.rng_mode, L.cand, i, G.vbase, ptmp);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) {
                        if (okp) { atomicOr(&C.flags, CF_PATCHED); s_npatched = 1; }
                        else { atomicOr(&C.flags, CF_DEP); atomicMin(&s_pmax, i); }
                    }
                }
                BSYNC();
                Pmax = s_pmax < Pmax ? s_pmax : Pmax;
            } }
            if constexpr (PATCH) { if (s_status) break; }   // (workgroup-uniform: read behind the barrier above or the one that ended P3)
            PHASE_STAMP(19);
            PHASE_STAMP(3);
            // ================= P4: PQ replay; targets shared by several candidates go in candidate order =================
            int32_t key0k[PASSES];   // the targets' keys: fetched here, used after the barrier (their latency hides behind the table)
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = PI(k), j = PJ(k);
                key0k[k] = 1;
                if (i >= Pmax) continue;
                if (j < L.cand[i].m) {
                    const int32_t x = L.cand[i].e[j].nbr;
                    if (use_pq) key0k[k] = A.vr[x].key;
                    uint32_t hh = ((uint32_t)x * 2654435761u) >> (32 - HBITS);
                    while (true) {   // linear probing; at most SLOTS distinct keys in SLOTS places
                        const int32_t cur = hkeys[hh];
                        if (cur == x) break;
                        if (cur == -1) { const int32_t old = atomicCAS(&hkeys[hh], -1, x); if (old == -1 || old == x) break; }
                        hh = (hh + 1) & (SLOTS - 1);
                    }
                    atomicAdd(&hcnt[hh], 1);
                }
            }
            BSYNC();
            PHASE_STAMP(33);
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = PI(k), j = PJ(k);
                uint64_t pm = 0ull;
                if constexpr (GROUPED) {
                    if (__ballot(i < Pmax) == 0ull) continue;   // (wave-uniform)
                    const bool cok = i < Pmax;
                    pm = group_pick_mask<BC <= 64 ? BC : 64>(L.cand[cok ? i : 0].ksel, cok ? L.cand[i].m : 0, cok, lane);
                }
                if (i >= Pmax) continue;
                Cand& C = L.cand[i];
                if (j >= C.m) continue;
                const int32_t x = C.e[j].nbr;
                uint32_t hh = ((uint32_t)x * 2654435761u) >> (32 - HBITS);
                while (hkeys[hh] != x) hh = (hh + 1) & (SLOTS - 1);
                const int32_t tc = hcnt[hh];
                const int32_t key0 = key0k[k];
                TRes& R = ent_tres(C.e[j]);
                if (tc > 1) {
                    // the key-independent half of the replay now, in parallel; the walk below chains the keys
                    int c0, li0;
                    if constexpr (GROUPED) { c0 = __popcll(pm); li0 = pm ? 63 - __builtin_clzll(pm) : -1; }
                    else cand_replay_pre(A, C, j, &c0, &li0);
                    R.flags = TF_CONTENDED; R.mv = (int16_t)li0; R.c = (uint8_t)c0; R.key_after = 0;
                    int32_t q = atomicAdd(&s_ncont, 1);
                    if (q < CCAP) { L.cont[q].x = x; L.cont[q].i = i; L.cont[q].j = j; }
                    else atomicMin(&s_p, i);     // record list full: stop the round before this candidate
                    continue;
                }
                const bool allow_last = (done + i + 1) + 1 < (int64_t)n;
                int mv, c; bool cx = false;
                int32_t k2;
                if constexpr (GROUPED) {
                    c = __popcll(pm);
                    k2 = cand_replay_post(A, C, j, key0, use_pq ? n : 0x7FFFFFFF, allow_last, c, pm ? 63 - __builtin_clzll(pm) : -1, &mv, &cx);
                } else k2 = cand_replay(A, C, j, key0, use_pq ? n : 0x7FFFFFFF, allow_last, &mv, &c, &cx);
                if (!use_pq) { mv = -1; cx = false; }
                R.key_after = k2; R.mv = (int16_t)mv; R.c = (uint8_t)c; R.flags = 0;
                if (cx) atomicMin(&s_p, i);   // keys beyond n: this candidate goes to the single-vertex path
                if (mv >= 0) {
                    atomicAdd(&C.nmv, 1);
                    if (pq_list_of(k2, n) <= G.minlist) atomicMin(&s_p, i + 1);   // pre-empts the predicted order: last of the round
                }
            }
            BSYNC();
            PHASE_STAMP(34);
            // contended records: order by (x, i) with a rank sort (keys are distinct), then replay each group in order
            const int32_t ncont = s_ncont < CCAP ? s_ncont : CCAP;
            CRec* csorted = L.csorted;
            for (int32_t q = tid; q < ncont; q += NT) {
                CRec me = L.cont[q];
                uint64_t kme = ((uint64_t)(uint32_t)me.x << 32) | (uint32_t)me.i;
                int rank = 0;
                for (int32_t r = 0; r < ncont; ++r) {
                    uint64_t kr = ((uint64_t)(uint32_t)L.cont[r].x << 32) | (uint32_t)L.cont[r].i;
                    rank += (kr < kme) ? 1 : 0;
                }
                csorted[rank] = me;
            }
            BSYNC();
            PHASE_STAMP(35);
            for (int32_t q = tid; q < ncont; q += NT) {
                if (q > 0 && csorted[q - 1].x == csorted[q].x) continue;   // group head only
                int32_t x = csorted[q].x;
                int32_t key = use_pq ? A.vr[x].key : 1;
                for (int32_t r = q; r < ncont && csorted[r].x == x; ++r) {
                    Cand& C = L.cand[csorted[r].i];
                    const bool allow_last = (done + csorted[r].i + 1) + 1 < (int64_t)n;
                    TRes& R = ent_tres(C.e[csorted[r].j]);
                    int mv; bool cx = false;
                    int32_t k2 = cand_replay_post(A, C, csorted[r].j, key, use_pq ? n : 0x7FFFFFFF, allow_last, (int)R.c, (int)R.mv, &mv, &cx);
                    if (!use_pq) { mv = -1; cx = false; }
                    R.key_after = k2; R.mv = (int16_t)mv;
                    if (cx) atomicMin(&s_p, csorted[r].i);
                    if (mv >= 0) {
                        atomicAdd(&C.nmv, 1);
                        if (pq_list_of(k2, n) <= G.minlist) atomicMin(&s_p, csorted[r].i + 1);
                    }
                    key = k2;
                }
            }
            BSYNC();
            // ================= P: the first pre-empting / complex candidate was noted during the replay; bound the
            PHASE_STAMP(36);
            // number of moves (candidates sit in the first two waves: a wave scan and one word through LDS) =================
            {
                const int mycnt = tid < Pmax ? L.cand[tid].nmv : 0;
                int incl = mycnt;
                for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
                if (tid == 63) L.scan[0] = incl;
                BSYNC();
                static_assert(BATCH <= 128, "candidates live in waves 0 and 1");
                if (tid < Pmax && incl + (tid >= 64 ? L.scan[0] : 0) > MCAP) atomicMin(&s_p, tid);
            }
            BSYNC();
            P = s_p < Pmax ? s_p : Pmax;
            PHASE_STAMP(4);
        }
        if (P == 0) {
            // ================= single-vertex path for candidate 0 =================
            ++singles;
            const int32_t v0 = L.cand[0].v, src0 = L.cand[0].src;
            BSYNC();
            if (tid < nc) batch_pos[L.cand[tid].v] = -1;
            if (tid == 0 && use_pq) {
                if (src0 >= 0) A.bs_cnt[b] = src0; else { A.bs_cnt[b] = 0; A.ocur[b] = (~src0) + 1; }
                A.vr[v0].pqpos = -2;
            }
            BSYNC();
            if (tid < 64) {
                // the out-of-line paths take the descriptors by reference: hand them private copies so that
                // the kernel's own copy never has its address taken (it would otherwise live in scratch
                // memory and every A.member access in the hot phases would be a scratch load)
                Arrays A2 = A;
                ElimScratch S2 = S;
                int32_t cp0 = A2.colptr[v0], cp1 = A2.colptr[v0 + 1];
                int32_t acnt = A2.vr[v0].app_cnt, abase = A2.vr[v0].app_chunk;
                if ((cp1 - cp0) + acnt > EC1) {
                    bool handled = false;
                    if (OV == OV_RANDOM) {
                        ColBuf Bf = S2.colbuf(G.scr_base);
                        handled = wave_eliminate_big(A2, G, sh.g, Bf, G.scr_cap, v0, cp0, cp1, acnt, abase, s_help);
                    }
                    if (!handled && lane == 0) {
                        __hip_atomic_store(&s_help[0], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (before, not after: see above)
                        serial_eliminate_call(A2, G, S2, v0, done + 1);
                    }
                } else {
                    wave_eliminate(A2, G, sh.e, v0, done + 1, cp0, cp1, acnt, abase, (S.prof && g == 0) ? &s_prof[24] : nullptr, s_help);
                }
            } else if (tid < 128) {
                single_helper(sh.e, sh.g.c.newv, s_help);
            }
            if (tid == 0) { s_nc = 0; s_pmax = BATCH + 1; s_p = BATCH + 1; s_ncont = 0; s_nmoves = 0; s_anydep = 0; s_npatched = 0; s_dirb = -1; }
            BSYNC();
            if (G.status != 0 || s_help[2] != 0) { if (tid == 0) s_status = G.status ? G.status : ST_INTERNAL; BSYNC(); break; }
            done += 1;
            PHASE_STAMP(5);
            continue;
        }

        // ================= P5: commit candidates [0,P) =================
        if (tid == 0 && use_pq) {
            int32_t src = L.cand[P - 1].src;
            if (src >= 0) A.bs_cnt[b] = src; else { A.bs_cnt[b] = 0; A.ocur[b] = (~src) + 1; }
        }
        int32_t status = 0;
        if (tid < P) {
            Cand& C = L.cand[tid];
            if (use_pq) A.vr[C.v].pqpos = -2;
            // the edge to the last neighbour (or to the coarsening target) dies (:429-430 | :880-881)
            if (C.m >= 1) {
                const int32_t tw = C.e[A.o_v == OV_COARSEN ? C.koff : C.m - 1].twin;
                if (!PATCH || tw >= 0) A.e[tw].val = 0;   // (a patched entry's twin is appended in this very round: second pass of the rewire stores)
            }
            if (MERGE) for (int32_t q = 0; q < C.nkill; ++q) A.e[C.e[C.m + q].twin].val = 0;   // merged multi-edges (:289)
        }
        // append cursors of the targets shared by several candidates: loaded now (one group head per thread), used
        // after the pool bump below
        const int32_t ncont_c = s_ncont < CCAP ? s_ncont : CCAP;
        bool chead = false;
        int32_t ca0 = 0, cchunk0 = 0;
        if (tid < ncont_c) {
            const CRec me = L.csorted[tid];
            chead = (tid == 0 || L.csorted[tid - 1].x != me.x) && me.i < P;
            if (chead) { ca0 = A.vr[me.x].app_cnt; cchunk0 = A.vr[me.x].app_chunk; }
            // which positions of candidate me.i push into this target: found here by every record's own thread, so that
            // the group walk below only steps through the set bits (the unsorted record list is free: used as scratch)
            uint64_t pm = 0ull, pm_hi = 0ull;
            if (me.i < P) {
                const Cand& C = L.cand[me.i];
                if (BC <= 64 && A.o_v == OV_COARSEN) { if (me.j == C.koff) pm = ((C.m >= 64 ? ~0ull : ((1ull << C.m) - 1ull)) & ~(1ull << C.koff)); }
                else for (int32_t w = 0; 8 * w < C.m - 1; ++w) {
                    for (uint64_t y = ksel_matches(C.ksel, w, me.j, C.m - 1); y; y &= y - 1) {
                        const int32_t p = 8 * w + (ctz_u64(y) >> 3);
                        if (BC <= 64 || p < 64) pm |= 1ull << (p & 63); else pm_hi |= 1ull << (p & 63);
                    }
                }
            }
            L.cont[tid].x = (int32_t)(uint32_t)pm; L.cont[tid].j = (int32_t)(uint32_t)(pm >> 32);
            if (BC == 128) L.mkey[tid] = pm_hi;   // (128-slot candidates: o_v = random only, no PQ moves -- the move list is free)
        }
        // targets touched by one candidate only: one thread per (candidate, target); two passes so that
        // all loads are in flight together and the pool is bumped once per round
        {
            TRes Rk[PASSES]; int32_t xk[PASSES], a0k[PASSES], chk[PASSES], needk[PASSES]; bool actk[PASSES];
            int32_t need_thread = 0;
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = PI(k), j = PJ(k);
                actk[k] = false; xk[k] = 0; a0k[k] = 0; chk[k] = -1; needk[k] = 0;
                Rk[k].key_after = 0; Rk[k].mv = -1; Rk[k].c = 0; Rk[k].flags = 0;
                if (i < P && j < L.cand[i].m) {
                    Rk[k] = ent_tres(L.cand[i].e[j]);
                    xk[k] = L.cand[i].e[j].nbr;
                    actk[k] = !(Rk[k].flags & TF_CONTENDED);
                }
                if (actk[k] && Rk[k].c > 0) { a0k[k] = A.vr[xk[k]].app_cnt; chk[k] = A.vr[xk[k]].app_chunk; }
            }
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                if (__ballot(actk[k] && Rk[k].c > 0) != 0ull && actk[k] && Rk[k].c > 0) {
                    for (int32_t t = 0; t < (int32_t)Rk[k].c; ++t) {
                        int32_t a = a0k[k] + t;
                        int c = chunk_of(a);
                        if (a == chunk_start(c)) needk[k] += 1 + chunk_cap(c);
                    }
                }
                need_thread += needk[k];
            }
            int ntot;
            int nex = block_excl_scan<NWAVE>(need_thread, L.scan, &ntot);
            if (tid == 0) {
                // slots come from a reservation this workgroup holds (refilled with one global atomic when it
                // runs out): the round does not wait for a device-wide atomic
                int32_t base = 0;
                if (ntot > 0) {
                    if (s_pool_cur + ntot > s_pool_end) {
                        const int32_t grab = ntot > POOL_GRAB ? ntot : POOL_GRAB;
                        const int32_t b0 = atomicAdd(A.pool_top, grab);
                        if (b0 < 0 || b0 > A.slot_cap - grab) { s_pool_cur = 0; s_pool_end = 0; base = -1; }
                        else { s_pool_cur = b0; s_pool_end = b0 + grab; }
                    }
                    if (base == 0) { base = s_pool_cur; s_pool_cur += ntot; }
                }
                L.scan[NWAVE + 1] = base;
            }
            BSYNC();
            const int32_t pbase = L.scan[NWAVE + 1];
            if (pbase < 0) status = ST_POOL_OVERFLOW;
            PHASE_STAMP(10);
            int32_t cursor = pbase + nex;
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = PI(k), j = PJ(k);
                if (__ballot(actk[k]) == 0ull) continue;
                uint64_t pm = 0ull;
                if constexpr (GROUPED) {
                    const bool cok = i < P;
                    pm = group_pick_mask<BC <= 64 ? BC : 64>(L.cand[cok ? i : 0].ksel, cok ? L.cand[i].m : 0, cok, lane);
                }
                if (actk[k] && Rk[k].c > 0 && status == 0) {
                    const Cand& C = L.cand[i];
                    int32_t a = a0k[k], chunk = chk[k];
                    if constexpr (GROUPED) {
                        for (uint64_t mm = pm; mm; mm &= mm - 1) {   // the positions that push into this target, ascending
                            const int p = __builtin_ctzll(mm);
                            int c = chunk_of(a);
                            int32_t cs = chunk_start(c);
                            if (a == cs) { A.e[cursor].nbr = chunk; chunk = cursor; cursor += 1 + chunk_cap(c); }
                            L.pslot[i * BCAP + p] = chunk + 1 + (a - cs);
                            ++a;
                        }
                    } else {
                    if (A.o_v == OV_COARSEN) {
                        for (int32_t p = 0; p < C.m; ++p) {
                            if (p == C.koff) continue;
                            int c = chunk_of(a);
                            int32_t cs = chunk_start(c);
                            if (a == cs) { A.e[cursor].nbr = chunk; chunk = cursor; cursor += 1 + chunk_cap(c); }
                            L.pslot[i * BCAP + p] = chunk + 1 + (a - cs);
                            ++a;
                        }
                    } else {
                        for (int32_t w = 0; 8 * w < C.m - 1; ++w) {   // the picks that name this target, eight per load, ascending
                            for (uint64_t y = ksel_matches(C.ksel, w, j, C.m - 1); y; y &= y - 1) {
                                const int32_t p = 8 * w + (ctz_u64(y) >> 3);
                                int c = chunk_of(a);
                                int32_t cs = chunk_start(c);
                                if (a == cs) { A.e[cursor].nbr = chunk; chunk = cursor; cursor += 1 + chunk_cap(c); }
                                L.pslot[i * BCAP + p] = chunk + 1 + (a - cs);
                                ++a;
                            }
                        }
                    }
                    }
                    A.vr[xk[k]].app_cnt = a; A.vr[xk[k]].app_chunk = chunk;
                }
                const bool mvd = use_pq && actk[k] && Rk[k].mv >= 0;
                const uint64_t mvmask = __ballot(mvd);
                int32_t qbase = 0;
                if (mvmask) {   // one LDS atomic per wave instead of one per move
                    if (lane == __builtin_ctzll(mvmask)) qbase = atomicAdd(&s_nmoves, __popcll(mvmask));
                    qbase = __shfl(qbase, __builtin_ctzll(mvmask));
                }
                if (mvd) {
                    A.vr[xk[k]].key = Rk[k].key_after;
                    atomicOr(&L.cmask[i], 1ull << (Rk[k].mv - L.cand[i].m));
                    int32_t q = qbase + __popcll(mvmask & lanemask_lt(lane));
                    if (q < MCAP) {
                        L.mkey[q] = ((uint64_t)(uint32_t)pq_list_of(Rk[k].key_after, n) << 32) | (uint32_t)((i << 8) | Rk[k].mv);
                        L.mval[q] = xk[k];
                    }
                }
            }
        }
        PHASE_STAMP(37);
        // targets shared by several candidates: one thread walks the target's records in candidate order
        {
            const int32_t ncont = ncont_c;   // <= CCAP < NT: one record per thread
            const CRec* csorted = L.csorted;
            if (chead) {
                const int32_t q = tid;
                const int32_t x = csorted[q].x;
                int32_t a = ca0, chunk = cchunk0;
                const int32_t a_before = a;
                int32_t key_final = 0, mvseq = -1;
                for (int32_t r = q; r < ncont && csorted[r].x == x && csorted[r].i < P; ++r) {
                    Cand& C = L.cand[csorted[r].i];
                    int32_t* pslot_row = &L.pslot[csorted[r].i * BCAP];
                    uint64_t pm = ((uint64_t)(uint32_t)L.cont[r].j << 32) | (uint32_t)L.cont[r].x;
                    while (pm && status == 0) { const int p = __builtin_ctzll(pm); pm &= pm - 1; pslot_row[p] = alloc_in_column(A, a, chunk, &status); }
                    if constexpr (BC == 128) {
                        uint64_t ph = L.mkey[r];
                        while (ph && status == 0) { const int p = __builtin_ctzll(ph); ph &= ph - 1; pslot_row[64 + p] = alloc_in_column(A, a, chunk, &status); }
                    }
                    TRes R = ent_tres(C.e[csorted[r].j]);
                    key_final = R.key_after;
                    if (R.mv >= 0) mvseq = (csorted[r].i << 8) | R.mv;
                }
                if (a != a_before) { A.vr[x].app_cnt = a; A.vr[x].app_chunk = chunk; }
                if (use_pq) {
                    A.vr[x].key = key_final;
                    if (mvseq >= 0) {
                        atomicOr(&L.cmask[mvseq >> 8], 1ull << ((mvseq & 0xFF) - L.cand[mvseq >> 8].m));
                        int32_t qq = atomicAdd(&s_nmoves, 1);
                        if (qq < MCAP) {
                            L.mkey[qq] = ((uint64_t)(uint32_t)pq_list_of(key_final, n) << 32) | (uint32_t)mvseq;
                            L.mval[qq] = x;
                        }
                    }
                }
            }
        }
        if (__syncthreads_or(status != 0)) { if (tid == 0) s_status = ST_POOL_OVERFLOW; BSYNC(); break; }
        if (S.jitter > 0 && ((threadIdx.x >> 6) & 1)) for (int _q = 0; _q < S.jitter; ++_q) __builtin_amdgcn_s_sleep(8);
        PHASE_STAMP(6);
        // one thread per appended entry: twin rewritten in place + the new entry (:404-414)
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
            const int32_t i = tid / (NT / BATCH), pp = k * (NT / BATCH) + tid % (NT / BATCH);
            if (i < P) {
                const Cand& C = L.cand[i];
                const bool co = A.o_v == OV_COARSEN;
                if (co ? (pp < C.m && pp != C.koff) : (pp < C.m - 1)) {
                    const int32_t kk = co ? C.koff : (int32_t)C.ksel[pp];
                    if (!PATCH || C.e[pp].twin >= 0) rewire_store(A, C.e[pp].twin, L.pslot[i * BCAP + pp], C.e[pp].nbr, C.e[kk].nbr, C.e[pp].val);
                }
            }
        }
        if (PATCH && s_npatched) {
            // entries of patched candidates whose twin is an entry an EARLIER candidate of this round appends (~twin = its place in the
            // slot table): that candidate's stores (first pass, above) come first, the rewrite in place (:404-406) or the kill (:429-430) second
            BSYNC();
#pragma unroll 1
            for (int k = 0; k < PASSES; ++k) {
                const int32_t i = tid / (NT / BATCH), pp = k * (NT / BATCH) + tid % (NT / BATCH);
                if (i < P) {
                    const Cand& C = L.cand[i];
                    if (pp < C.m && C.e[pp].twin < 0) {
                        const int32_t s_r = L.pslot[~C.e[pp].twin];
                        if (pp < C.m - 1) rewire_store(A, s_r, L.pslot[i * BCAP + pp], C.e[pp].nbr, C.e[(int32_t)C.ksel[pp]].nbr, C.e[pp].val);
                        else A.e[s_r].val = 0;
                    }
                }
            }
        }
        PHASE_STAMP(11);
        const int32_t nmoves = s_nmoves;
        if (nmoves > MCAP) { if (tid == 0) s_status = ST_INTERNAL; BSYNC(); break; }
        if (use_pq && nmoves > 0) {
            static_assert(BC == 32 || OV == OV_RANDOM, "op numbers mv - m fit a 64-bit mask for 32-slot candidates");
            // ---- order of the moves = (bucket, candidate, op).  No sort: a move's place in (candidate, op) order is
            //      the candidates' move counts summed before it plus the rank of its op bit in its candidate's mask;
            //      the moves are laid out in that order, then each finds its rank inside its bucket by counting
            //      (64 buckets above the lowest through a per-wave table, the few beyond -- hubs -- by comparing
            //      among themselves).  Falls back to the bitonic sort when more than 128 moves go beyond. ----
            const bool mine = tid < nmoves;
            int32_t list_my = 0, x_my = -1, rank_my = 0, pos_my = 0;
            bool tail_my = false;
            uint32_t bmin = 0xFFFFFFFFu;
            bool fallback = false;
            {
                const uint64_t kq0 = mine ? L.mkey[tid] : 0ull;
                const int32_t x0 = mine ? L.mval[tid] : -1;
                const int32_t seq = (int32_t)(kq0 & 0x7FFFull);
                const int32_t ci = seq >> 8, cmv = seq & 0xFF;
                if (mine) bmin = (uint32_t)(kq0 >> 32);
                for (int off = 32; off > 0; off >>= 1) bmin = min(bmin, (uint32_t)__shfl_xor((int)bmin, off));
                if (lane == 0) L.scan[tid >> 6] = (int32_t)bmin;
                const int cnt_i = tid < P ? __popcll(L.cmask[tid]) : 0;
                int incl = cnt_i;
                for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
                if (tid == 63) L.scan[NWAVE] = incl;
                BSYNC();
                for (int w = 0; w < NWAVE; ++w) bmin = min(bmin, (uint32_t)L.scan[w]);
                if (tid < P) L.hidx[tid] = incl - cnt_i + (tid >= 64 ? L.scan[NWAVE] : 0);
                // [NWAVE][RB] move counts per (wave, bucket above the lowest); pslot is idle now (SLOTS = NWAVE * 256 ints)
                constexpr int RB = 256, HCAP = 128;
                static_assert(NWAVE * RB <= SLOTS && 2 * RB + 2 * HCAP <= 3 * CCAP * 2, "tables fit pslot / the record lists");
                int32_t* const wtab = L.pslot;
#pragma unroll
                for (int k = 0; k < PASSES; ++k) wtab[k * NT + tid] = 0;
                if (tid == 0) s_nhigh = 0;
                BSYNC();
                uint32_t* const lst32 = reinterpret_cast<uint32_t*>(L.mkey);
                if (mine) {
                    const int32_t idx = L.hidx[ci] + __popcll(L.cmask[ci] & ((1ull << (cmv - L.cand[ci].m)) - 1ull));
                    lst32[idx] = (uint32_t)(kq0 >> 32);
                    L.mval[idx] = x0;
                }
                BSYNC();
                if (mine) { list_my = (int32_t)lst32[tid]; x_my = L.mval[tid]; }
                const uint32_t brel = (uint32_t)list_my - bmin;
                const bool low = mine && brel < (uint32_t)RB, high = mine && brel >= (uint32_t)RB;
                int32_t* const btot = reinterpret_cast<int32_t*>(L.cont);     // [RB] moves per low bucket (the record lists are idle now)
                int32_t* const bstart = btot + RB;                            // [RB] first sorted position of a low bucket
                int32_t* const hl_list = bstart + RB;                         // [HCAP] buckets of the moves beyond
                int32_t* const hl_idx = hl_list + HCAP;                       // [HCAP] their places in (candidate, op) order
                int rank_w = 0;
                {
                    // lanes of my wave with my bucket: one ballot per bit of the bucket number (8 bits) instead of a loop
                    // over the distinct buckets of the wave
                    static_assert(RB == 256, "8 bits of bucket number");
                    uint64_t eq = __ballot(low);
#pragma unroll
                    for (int bit = 0; bit < 8; ++bit) {
                        const bool mybit = (brel >> bit) & 1u;
                        const uint64_t bb = __ballot(low && mybit);
                        eq &= mybit ? bb : ~bb;
                    }
                    if (low) {
                        const uint64_t before = eq & lanemask_lt(lane);
                        rank_w = __popcll(before);
                        if (before == 0ull) wtab[(tid >> 6) * RB + brel] = __popcll(eq);
                    }
                }
                if (high) { const int32_t q = atomicAdd(&s_nhigh, 1); if (q < HCAP) { hl_list[q] = list_my; hl_idx[q] = tid; } }
                BSYNC();
                const int32_t nhigh = s_nhigh;
                fallback = nhigh > HCAP;
                if (!fallback) {
                    if (tid < RB) {   // per bucket: exclusive prefix over the waves, total
                        int acc = 0;
                        for (int w = 0; w < NWAVE; ++w) { const int t = wtab[w * RB + tid]; wtab[w * RB + tid] = acc; acc += t; }
                        btot[tid] = acc;
                    }
                    BSYNC();
                    if (tid < 64) {   // bucket starts: each lane of wave 0 takes RB/64 consecutive buckets
                        int tsum = 0;
#pragma unroll
                        for (int u = 0; u < RB / 64; ++u) tsum += btot[tid * (RB / 64) + u];
                        int inc2 = tsum;
                        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc2, off); if (lane >= off) inc2 += t; }
                        int run = inc2 - tsum;
#pragma unroll
                        for (int u = 0; u < RB / 64; ++u) { bstart[tid * (RB / 64) + u] = run; run += btot[tid * (RB / 64) + u]; }
                        if (tid == 63) s_nlow = inc2;
                    }
                    BSYNC();
                    if (low) {
                        rank_my = wtab[(tid >> 6) * RB + brel] + rank_w;
                        pos_my = bstart[brel] + rank_my;
                        tail_my = rank_my == btot[brel] - 1;
                    } else if (high) {
                        int32_t before = 0, same_b = 0, cntb = 0;
                        for (int32_t q = 0; q < nhigh; ++q) {
                            const int32_t l2 = hl_list[q], t2 = hl_idx[q];
                            before += (l2 < list_my || (l2 == list_my && t2 < tid)) ? 1 : 0;
                            if (l2 == list_my) { ++cntb; same_b += (t2 < tid) ? 1 : 0; }
                        }
                        rank_my = same_b;
                        pos_my = s_nlow + before;
                        tail_my = rank_my == cntb - 1;
                    }
                } else {
                    // the moves sit in (candidate, op) order: their place stands in for the (candidate, op) key
                    BSYNC();
                    L.mkey[tid] = mine ? (((uint64_t)(uint32_t)list_my << 32) | (uint32_t)tid) : ~0ull;
                    L.mval[tid] = x_my;
                    BSYNC();
                }
            }
            if (fallback) {
                // ---- sort the moves by (bucket, op order): bitonic, one element per thread; strides < 64 stay
                //      inside the wave (shuffles), only the larger strides go through LDS.  Keys are packed
                //      into 32 bits ((bucket - lowest bucket) << 15 | candidate << 8 | op) when they fit ----
                {
                    uint64_t kq = tid < nmoves ? L.mkey[tid] : ~0ull;
                    int32_t vq = tid < nmoves ? L.mval[tid] : -1;
                    uint32_t bmin = tid < nmoves ? (uint32_t)(kq >> 32) : 0xFFFFFFFFu, bmax = tid < nmoves ? (uint32_t)(kq >> 32) : 0u;
                    for (int off = 32; off > 0; off >>= 1) { bmin = min(bmin, (uint32_t)__shfl_xor((int)bmin, off)); bmax = max(bmax, (uint32_t)__shfl_xor((int)bmax, off)); }
                    if (lane == 0) { L.scan[tid >> 6] = (int32_t)bmin; L.hidx[tid >> 6] = (int32_t)bmax; }
                    BSYNC();
                    for (int w = 0; w < NWAVE; ++w) { bmin = min(bmin, (uint32_t)L.scan[w]); bmax = max(bmax, (uint32_t)L.hidx[w]); }
                    int32_t npow = 64;
                    while (npow < nmoves) npow <<= 1;
                    BSYNC();
                    if (bmax - bmin < (1u << 16)) {
                        uint32_t k32 = tid < nmoves ? ((((uint32_t)(kq >> 32) - bmin) << 15) | ((uint32_t)kq & 0x7FFFu)) : 0xFFFFFFFFu;
                        uint32_t* lk = reinterpret_cast<uint32_t*>(L.mkey);
                        for (int32_t k = 2; k <= npow; k <<= 1) {
                            for (int32_t jj = k >> 1; jj > 0; jj >>= 1) {
                                const bool take_min = (((tid & jj) == 0) == ((tid & k) == 0));
                                uint32_t ko; int32_t vo;
                                if (jj >= 64) {
                                    lk[tid] = k32; L.mval[tid] = vq;
                                    BSYNC();
                                    ko = lk[tid ^ jj]; vo = L.mval[tid ^ jj];
                                    BSYNC();
                                } else {
                                    ko = (uint32_t)__shfl_xor((int)k32, jj);
                                    vo = __shfl_xor(vq, jj);
                                }
                                const bool swap = take_min ? (ko < k32) : (ko > k32);
                                if (swap) { k32 = ko; vq = vo; }
                            }
                        }
                        BSYNC();
                        kq = (k32 == 0xFFFFFFFFu) ? ~0ull : ((((uint64_t)((k32 >> 15) + bmin)) << 32) | (uint64_t)(k32 & 0x7FFFu));
                    } else {
                        for (int32_t k = 2; k <= npow; k <<= 1) {
                            for (int32_t jj = k >> 1; jj > 0; jj >>= 1) {
                                const bool take_min = (((tid & jj) == 0) == ((tid & k) == 0));
                                uint64_t ko; int32_t vo;
                                if (jj >= 64) {
                                    L.mkey[tid] = kq; L.mval[tid] = vq;
                                    BSYNC();
                                    ko = L.mkey[tid ^ jj]; vo = L.mval[tid ^ jj];
                                    BSYNC();
                                } else {
                                    uint32_t lo32 = (uint32_t)kq, hi32 = (uint32_t)(kq >> 32);
                                    uint32_t olo = (uint32_t)__shfl_xor((int)lo32, jj), ohi = (uint32_t)__shfl_xor((int)hi32, jj);
                                    ko = ((uint64_t)ohi << 32) | olo;
                                    vo = __shfl_xor(vq, jj);
                                }
                                const bool swap = take_min ? (ko < kq) : (ko > kq);
                                if (swap) { kq = ko; vq = vo; }
                            }
                        }
                    }
                    L.mkey[tid] = kq; L.mval[tid] = vq;
                    BSYNC();
                }
                // ---- bucket-group head index of every move (inclusive max scan) ----
                {
                    int32_t carry = 0;
                    for (int32_t c0 = 0; c0 < nmoves; c0 += NT) {
                        int32_t r = c0 + tid;
                        int32_t hv = -1;
                        if (r < nmoves) {
                            uint32_t bk = (uint32_t)(L.mkey[r] >> 32);
                            bool head = (r == 0) || ((uint32_t)(L.mkey[r - 1] >> 32) != bk);
                            hv = head ? r : -1;
                        }
                        int32_t incl = hv;
                        for (int off = 1; off < 64; off <<= 1) { int32_t tt = __shfl_up(incl, off); if (lane >= off) incl = max(incl, tt); }
                        if (lane == 63) L.scan[tid >> 6] = incl;
                        BSYNC();
                        int32_t pre = carry;
                        for (int w = 0; w < (tid >> 6); ++w) pre = max(pre, L.scan[w]);
                        incl = max(incl, pre);
                        if (r < nmoves) L.hidx[r] = incl;
                        int32_t nc2 = carry;
                        for (int w = 0; w < NWAVE; ++w) nc2 = max(nc2, L.scan[w]);
                        carry = nc2;
                        BSYNC();
                    }
                }
                if (mine) {
                    list_my = (int32_t)(L.mkey[tid] >> 32);
                    x_my = L.mval[tid];
                    rank_my = tid - L.hidx[tid];
                    pos_my = tid;
                    tail_my = (tid == nmoves - 1) || ((uint32_t)(L.mkey[tid + 1] >> 32) != (uint32_t)(L.mkey[tid] >> 32));
                    if (tid == 0) bmin = (uint32_t)list_my;
                }
                bmin = (uint32_t)(L.mkey[0] >> 32);
            }
            PHASE_STAMP(7);
            // ---- allocate new stack chunks, then write the entries (one move per thread; the stack height and chunk
            //      count are read once and kept in registers) ----
            int32_t bk = 0, a_my = 0, c_my = 0, al_my = 0;
            if (mine) {
                bk = G.bucket_base + list_my;
                a_my = A.bs_cnt[bk] + rank_my;
                al_my = A.bs_alloc[bk];
                c_my = bs_chunk_of(a_my);
                if (c_my >= al_my && a_my == bs_chunk_start(c_my)) {
                    int32_t need = bs_chunk_cap(c_my);
                    int32_t base = atomicAdd(A.bs_pool_top, need);
                    if (base < 0 || base > A.bs_pool_cap - need) status = ST_LOG_OVERFLOW;
                    else { A.bs_dir[(int64_t)bk * BDIR + c_my] = base; if (bk == s_dirb) s_dir[c_my] = base; }
                }
            }
            // the push ids of this round start at G.push_cnt: read BEFORE the barrier -- thread 0 advances the counter behind it, and a wave
            // that came out of the barrier late used to read the advanced value (ids of two rounds overlapped: a wrong LIFO order inside a
            // bucket, once in some thousand batched calls -- the "graph 851" mismatch of round 2)
            const int32_t id0 = G.push_cnt;
            if (__syncthreads_or(status != 0)) { if (tid == 0) s_status = ST_LOG_OVERFLOW; BSYNC(); break; }
            if (S.jitter > 0 && ((threadIdx.x >> 6) & 1)) for (int _q = 0; _q < S.jitter; ++_q) __builtin_amdgcn_s_sleep(8);
            if (mine) {
                int32_t sl = A.bs_dir[(int64_t)bk * BDIR + c_my] + (a_my - bs_chunk_start(c_my));
                A.bs_v[sl] = x_my; A.bs_id[sl] = id0 + pos_my;
                A.vr[x_my].pqpos = id0 + pos_my;
                if (tail_my) {   // last move of its bucket group: the stack's new height (and chunk count)
                    A.bs_cnt[bk] = a_my + 1;
                    if (c_my + 1 > al_my) A.bs_alloc[bk] = c_my + 1;
                }
            }
            if (tid == 0) {
                if ((int32_t)bmin < G.minlist) G.minlist = (int32_t)bmin;
                G.push_cnt = id0 + nmoves;
            }
        }
        PHASE_STAMP(8);
        // ---- round epilogue ----
        if (tid < nc) batch_pos[L.cand[tid].v] = -1;
        if (tid == 0) {
            G.n_draws = L.cand[P - 1].draw0 + L.cand[P - 1].ndraw;   // offsets are a running sum (P2)
            s_nc = 0; s_pmax = BATCH + 1; s_p = BATCH + 1; s_ncont = 0; s_nmoves = 0; s_anydep = 0; s_npatched = 0;
        }
        done += P;
        pending_long = (OV == OV_RANDOM) && P < nc && (L.cand[P].flags & CF_BIG) != 0;
        BSYNC();
    }
    BSYNC();
    if (tid == 0) {
        G.n_elim = (int32_t)nelim;
        if (s_status) G.status = s_status;
        G.pad0 = rounds; G.pad1 = singles;
        if (S.prof && g == 0) { for (int q = 0; q < 20; ++q) S.prof[q] = s_prof[q]; S.prof[20] = clock64() - clk0; S.prof[21] = wall_clock64() - wall0; S.prof[22] = rounds; S.prof[23] = singles; for (int q = 24; q < 40; ++q) S.prof[q] = s_prof[q]; }
        gd[g] = G;
    }
}
#undef BSYNC

#ifndef RLAP_ELIM_PQ_TU
// The handle's tables of std::sort's permutation of 17..BC all-equal keys (layout: rlap_kernels.h EQTAB_*), built once.
template <int BC>
__global__ __launch_bounds__(64) void k_eq_tables(uint8_t* __restrict__ out) {
    __shared__ CandT<BC> c[16];
    const int tid = threadIdx.x;
    Arrays A = {};
    A.o_v = OV_RANDOM; A.o_n = ON_ASC;   // all keys equal: asc and desc compare alike
    for (int base = 0; base < BC - 16; base += 16) {
        const int row = base + tid;
        if (tid < 16 && row < BC - 16) {
            CandT<BC>& C = c[tid];
            const int nn = 17 + row;
            for (int q = 0; q < nn; ++q) { C.e[q].aux = 1.0; C.ksel[q] = (uint8_t)q; }
            C.m = nn;
            cand_order_index_call(A, C);
            uint8_t* perm = out + row * BC;
            uint8_t* inv = out + (BC - 16) * BC + row * BC;
            for (int q = 0; q < BC; ++q) perm[q] = q < nn ? C.ksel[q] : (uint8_t)q;
            for (int q = 0; q < BC; ++q) inv[perm[q]] = (uint8_t)q;
        }
        __syncthreads();
    }
}

void launch_eq_tables(hipStream_t stream, uint8_t* out) {
    hipLaunchKernelGGL(k_eq_tables<64>, dim3(1), dim3(64), 0, stream, out);
    hipLaunchKernelGGL(k_eq_tables<32>, dim3(1), dim3(64), 0, stream, out + EQTAB_OFF32);
    hipLaunchKernelGGL(k_eq_tables<128>, dim3(1), dim3(64), 0, stream, out + EQTAB_OFF128);
}

// the priority-queue instantiations live in the other translation unit (top of this file)
void launch_eliminate_pq(int o_v, int o_n, unsigned G, bool many, hipStream_t stream, const Arrays& A, GraphDesc* gd, const ElimScratch& S,
                         int32_t* batch_pos, const int32_t* flags, const double* acc);

void launch_eliminate_batch(int o_v, int o_n, unsigned G, int n_cu, hipStream_t stream, const Arrays& A, GraphDesc* gd, const ElimScratch& S,
                            int32_t* batch_pos, const int32_t* flags, const double* acc, bool wide) {
    // o_v = random meets long columns all the time (19 % of BA(1M,10) have more than 32 live entries when their
    // turn comes): it runs with 64 slots per candidate (64 candidates per round); the min-degree orders with 32.
    // more graphs than CUs: the 256-thread shape, four workgroups per CU (128 VGPRs, 39 KB of LDS; measured on 4096-node graphs: the same
    // time per graph as the 1024-thread shape, which is only ahead when one graph offers more than 32 independent vertices a round)
    if (n_cu <= 0) n_cu = 256;
    bool many = G > (unsigned)n_cu;
    if (const char* e = std::getenv("RLAP_BATCH_SHAPE")) { if (e[0] == '2') many = true; else if (e[0] == '1') many = false; }   // diagnostic override: 256 / 1024
    // o_v = random on graphs with long columns (`wide`, chosen by the caller from the entries per vertex): 128 slots per
    // candidate, 32 candidates per round -- a column of 65..128 entries no longer ends the round (BA(4096,8): 134 rounds + 13
    // single vertices instead of 148 + 60).  Sparse graphs fill their rounds with 64 slots and keep 64 candidates.
    if (const char* e = std::getenv("RLAP_WIDE")) { if (e[0] == '1') wide = true; else if (e[0] == '0') wide = false; }   // diagnostic override
    if (wide && !many && o_v == OV_RANDOM) {
        if (o_n == ON_ASC) hipLaunchKernelGGL((k_eliminate_batch_t<OV_RANDOM, ON_ASC, 128, 1024>), dim3(G), dim3(1024), 0, stream, A, gd, S, batch_pos, flags, acc);
        else if (o_n == ON_DESC) hipLaunchKernelGGL((k_eliminate_batch_t<OV_RANDOM, ON_DESC, 128, 1024>), dim3(G), dim3(1024), 0, stream, A, gd, S, batch_pos, flags, acc);
        else hipLaunchKernelGGL((k_eliminate_batch_t<OV_RANDOM, ON_RANDOM, 128, 1024>), dim3(G), dim3(1024), 0, stream, A, gd, S, batch_pos, flags, acc);
        return;
    }
#define RLAP_CASE(OV, ON, BC) if (o_v == OV && o_n == ON) { \
        if (many) hipLaunchKernelGGL((k_eliminate_batch_t<OV, ON, BC, 256>), dim3(G), dim3(256), 0, stream, A, gd, S, batch_pos, flags, acc); \
        else hipLaunchKernelGGL((k_eliminate_batch_t<OV, ON, BC, 1024>), dim3(G), dim3(1024), 0, stream, A, gd, S, batch_pos, flags, acc); \
        return; }
    RLAP_CASE(OV_RANDOM, ON_ASC, 64) RLAP_CASE(OV_RANDOM, ON_DESC, 64) RLAP_CASE(OV_RANDOM, ON_RANDOM, 64)
#undef RLAP_CASE
    launch_eliminate_pq(o_v, o_n, G, many, stream, A, gd, S, batch_pos, flags, acc);
}
#else   // RLAP_ELIM_PQ_TU
void launch_eliminate_pq(int o_v, int o_n, unsigned G, bool many, hipStream_t stream, const Arrays& A, GraphDesc* gd, const ElimScratch& S,
                         int32_t* batch_pos, const int32_t* flags, const double* acc) {
#define RLAP_CASE(OV, ON, BC) if (o_v == OV && o_n == ON) { \
        if (many) hipLaunchKernelGGL((k_eliminate_batch_t<OV, ON, BC, 256>), dim3(G), dim3(256), 0, stream, A, gd, S, batch_pos, flags, acc); \
        else hipLaunchKernelGGL((k_eliminate_batch_t<OV, ON, BC, 1024>), dim3(G), dim3(1024), 0, stream, A, gd, S, batch_pos, flags, acc); \
        return; }
    RLAP_CASE(OV_DEGREE, ON_ASC, 32) RLAP_CASE(OV_DEGREE, ON_DESC, 32) RLAP_CASE(OV_DEGREE, ON_RANDOM, 32)
    RLAP_CASE(OV_COARSEN, ON_ASC, 32) RLAP_CASE(OV_COARSEN, ON_DESC, 32) RLAP_CASE(OV_COARSEN, ON_RANDOM, 32)
#undef RLAP_CASE
}
#endif  // RLAP_ELIM_PQ_TU

#ifndef RLAP_ELIM_PQ_TU
// ---------------------------------------------------------------------------
// K9: output.  sc_keys/sc_perm: pop order of the surviving vertices;
// sc_merge (pass A): per vertex gather -> sort -> merge -> order -> staging;
// sc_compact (pass B): prefix-sum compaction of the staged rows into (m,3) f64.
// ---------------------------------------------------------------------------
__global__ void k_sc_keys(const VRec* __restrict__ vr, const int32_t* __restrict__ origpos,
                          const int32_t* __restrict__ vgraph, const GraphDesc* __restrict__ gd, int32_t N,
                          uint64_t* __restrict__ skey, uint32_t* __restrict__ sval) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const VRec me = vr[v];
    int32_t pp = me.pqpos;
    uint64_t k = ~0ull;
    if (pp != -2) {
        const GraphDesc& D = gd[vgraph[v]];
        uint32_t b = (uint32_t)(D.bucket_base + pq_list_of(me.key, D.n));
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
    int64_t pl = perm[i];
    if (pl < 0 || pl >= (int64_t)D.n) pl = 0;   // invalid entry (flagged by k_perm_check; the call fails): keep the index in range
    if (q >= D.n_elim) order[surv_base[g] + (q - D.n_elim)] = (uint32_t)(D.vbase + (int32_t)pl);
}

// Rejected input (range / cross-graph / node_id flags, asymmetry: the same tests the elimination kernel makes) gives every column
// the extent 0: the output pass then stages and writes nothing -- an invalid node_id vector can name one hub S times, which would
// run past the staging arrays (sized one row per slot in use).
__global__ void k_sc_ext(const uint32_t* __restrict__ order, const int32_t* __restrict__ colptr, const VRec* __restrict__ vr,
                         int32_t S, const int32_t* __restrict__ in_flags, const double* __restrict__ in_acc, int32_t* __restrict__ ext) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const bool bad = in_flags[FLAG_RANGE] || in_flags[FLAG_CROSS] || in_flags[FLAG_PERM] || in_acc[2] != 0.0 || !(in_acc[0] <= 1e-24 * in_acc[1]);
    int32_t v = (int32_t)order[i];
    ext[i] = bad ? 0 : (colptr[v + 1] - colptr[v]) + vr[v].app_cnt;
}

struct RecKeyLessDbg { const SRec* r; __device__ bool operator()(uint16_t a, uint16_t b) const { return r[a].key < r[b].key; } };
struct RecKeyGreaterDbg { const SRec* r; __device__ bool operator()(uint16_t a, uint16_t b) const { return r[a].key > r[b].key; } };
// test hook: one wave sorts one array of doubles, returns the permutation (tests/test_gpu_parity.py).
// CAUTION: an out-of-line function is compiled once for all its callers, with the loosest register budget among them.  This kernel
// has none: calling an instantiation the elimination kernels use (e.g. wave_lvl_sort<SRec, .., 5 or 6>) from here lifts that
// function beyond 128 VGPRs and takes the 256-thread shape from four workgroups per CU to two (tests/test_cabi_symbols.py checks the
// register counts of the built kernels).
__global__ __launch_bounds__(64) void k_debug_wave_sort(const double* __restrict__ keys, const int32_t* __restrict__ offs, int32_t narr,
                                                        int32_t desc, int32_t* __restrict__ perm_out) {
    __shared__ SRec rec[SCAP];
    __shared__ WaveSortScratch W;
    __shared__ int32_t tmp64[160];
    __shared__ uint32_t ltab[SCAP];
    __shared__ uint16_t ltab2[SCAP];
    const int lane = lane_id();
    for (int32_t arr = blockIdx.x; arr < narr; arr += gridDim.x) {
        const int32_t o = offs[arr], n = offs[arr + 1] - o;
        for (int q = lane; q < n; q += 64) { rec[q].key = keys[o + q]; rec[q].idx = q; rec[q].aux = 0; }
        __syncthreads();
        WaveSortPtrs WP = {W.ulist, W.dlist, W.segmark, W.stk};
        if (desc & 4) {   // half-wave variant: this block's wave sorts arrays 2*arr and 2*arr+1 side by side (those of <= 32 elements)
            const int32_t a2 = 2 * arr + (lane >> 5);
            const int gl = lane & 31;
            int32_t o2 = 0, n2 = 0;
            if (a2 < narr) { o2 = offs[a2]; n2 = offs[a2 + 1] - o2; }
            const bool want = a2 < narr && n2 <= 32 && n2 >= 1;
            double key = (want && gl < n2) ? keys[o2 + gl] : 0.0;
            int idx = gl, pos = gl;
            bool ok = (desc & 1) ? group_sort<true, 32>(key, idx, n2, want, lane, tmp64, &pos) : group_sort<false, 32>(key, idx, n2, want, lane, tmp64, &pos);
            if (want && gl < n2) perm_out[o2 + (ok ? pos : gl)] = ok ? idx : -1;
            __syncthreads();
            if (2 * arr + 2 >= narr) break;
            continue;
        }
        if ((desc & 2) && n <= 64) {   // register-resident variant (batch candidates of the 64-slot kernel)
            double key = lane < n ? rec[lane].key : 0.0;
            int idx = lane, pos = lane;
            bool ok = (desc & 1) ? wave_sort64<true>(key, idx, n, lane, tmp64, &pos) : wave_sort64<false>(key, idx, n, lane, tmp64, &pos);
            if (lane < n) perm_out[o + (ok ? pos : lane)] = ok ? idx : -1;
            __syncthreads();
            continue;
        }
        if ((desc & 16) && n <= 128) {   // level-synchronous variant over an INDEX array, keys looked up by the comparison (128-slot candidates)
            __shared__ uint16_t ord[128];
            for (int q = lane; q < n; q += 64) ord[q] = (uint16_t)q;
            __syncthreads();
            const bool ok = (desc & 1) ? wave_lvl_sort<uint16_t, RecKeyGreaterDbg, 2>(ord, n, RecKeyGreaterDbg{rec}, W.ulist, W.dlist, ltab, ltab2, lane)
                                       : wave_lvl_sort<uint16_t, RecKeyLessDbg, 2>(ord, n, RecKeyLessDbg{rec}, W.ulist, W.dlist, ltab, ltab2, lane);
            __syncthreads();
            for (int q = lane; q < n; q += 64) perm_out[o + q] = ok ? (int32_t)ord[q] : -1;
            __syncthreads();
            continue;
        }
        if (desc & 8) {   // level-synchronous variant (single-vertex path, long columns of the output pass)
            bool ok;
            if (n <= 128) ok = (desc & 1) ? wave_lvl_sort<SRec, SRecGreaterKey, 2>(rec, n, SRecGreaterKey(), W.ulist, W.dlist, ltab, ltab2, lane) : wave_lvl_sort<SRec, SRecLessKey, 2>(rec, n, SRecLessKey(), W.ulist, W.dlist, ltab, ltab2, lane);
            else if (n <= 256) ok = (desc & 1) ? wave_lvl_sort<SRec, SRecGreaterKey, 4>(rec, n, SRecGreaterKey(), W.ulist, W.dlist, ltab, ltab2, lane) : wave_lvl_sort<SRec, SRecLessKey, 4>(rec, n, SRecLessKey(), W.ulist, W.dlist, ltab, ltab2, lane);
            else ok = (desc & 1) ? wave_lvl_sort<SRec, SRecGreaterKey, 8>(rec, n, SRecGreaterKey(), W.ulist, W.dlist, ltab, ltab2, lane) : wave_lvl_sort<SRec, SRecLessKey, 8>(rec, n, SRecLessKey(), W.ulist, W.dlist, ltab, ltab2, lane);
            if (!ok) {   // depth limit: start over on the original order
                __syncthreads();
                for (int q = lane; q < n; q += 64) { rec[q].key = keys[o + q]; rec[q].idx = q; rec[q].aux = 0; }
                __syncthreads();
                if (desc & 1) wave_std_sort<SRec>(rec, n, SRecGreaterKey(), WP, lane); else wave_std_sort<SRec>(rec, n, SRecLessKey(), WP, lane);
            }
        } else if (desc & 1) wave_std_sort<SRec>(rec, n, SRecGreaterKey(), WP, lane); else wave_std_sort<SRec>(rec, n, SRecLessKey(), WP, lane);
        __syncthreads();
        for (int q = lane; q < n; q += 64) perm_out[o + q] = rec[q].idx;
        __syncthreads();
    }
}

template <int CAP>
struct ScLdsT {
    SRec rec[CAP];
    double a_val[CAP];
    double b_val[CAP];
    int32_t a_nbr[CAP];
    int32_t b_nbr[CAP];
    int32_t rank[CAP];
};

template <bool GREATER, class LDS, class KeyF>
__device__ __forceinline__ bool sc_rank_sort(LDS& L, int cnt, KeyF keyf, int lane) {
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

// CAP = LDS capacity per column; columns with extent in (LO, CAP] are taken, the others left to the other
// instantiation / the long-column kernel.  The small instantiation (CAP 64) keeps ~30 waves per CU resident.
template <int CAP, int LO>
__global__ __launch_bounds__(64) void k_sc_merge_t(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                   const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                   const int64_t* __restrict__ tmp_off, int32_t S, int32_t* __restrict__ tmp_nbr,
                                                   double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out, ScScratch SS,
                                                   unsigned long long* __restrict__ live_total, const int32_t* __restrict__ worklist,
                                                   const int32_t* __restrict__ workcount) {
    __shared__ ScLdsT<CAP> L;
    __shared__ WaveSortScratchT<CAP> WS;
    __shared__ int32_t s_tmp64[CAP == 64 ? 144 : 1];
    const WaveSortPtrs WP = {WS.ulist, WS.dlist, WS.segmark, WS.stk};
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    (void)S;
    // columns of at most 64 entries: std::sort's permutation with one element per lane (registers + shuffles);
    // leaves L.rec[position] = {key, source index} like the LDS paths.  False: depth limit hit, use those.
    auto sort64 = [&](const double keyv, const int cnt, const bool greater) -> bool {
        if (CAP != 64) return false;
        double key = keyv;
        int idx = lane, pos = lane;
        const bool ok = greater ? wave_sort64<true>(key, idx, cnt, lane, s_tmp64, &pos) : wave_sort64<false>(key, idx, cnt, lane, s_tmp64, &pos);
        if (ok && lane < cnt) { L.rec[pos].key = key; L.rec[pos].idx = idx; }
        __syncthreads();
        return ok;
    };
    const int32_t nwork = *workcount;
    unsigned long long live_acc = 0ull;   // lane 0: live entries seen by this workgroup (one atomic at the end, spread over LIVE_SLOTS counters)
    for (int32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const int32_t i = worklist[wi];   // lists are filled with atomics: heavy columns end up spread over the grid
        const int32_t v = (int32_t)order[i];
        const int64_t toff = tmp_off[i];
        const int32_t ex = ext[i];
        if (CAP == SCAP && ex > SCAP) {   // (only the last tier's list can hold such a column: the other instantiations stay lean)
            // long column: k_sc_merge_big / k_sc_merge_huge take it, unless it is too long even for their 16-bit stop lists:
            // then the sequential form in global scratch (one lane)
            if (lane == 0 && ex > HUGECAP) {
                unsigned long long off = atomicAdd(SS.top, (unsigned long long)ex);
                if ((int64_t)(off + (unsigned long long)ex) > SS.cap) { SS.flags[FLAG_SCR] = 1; cnt_out[i] = 0; }   // scratch budget too small: the call is repeated with more
                else {
                ColBuf B = SS.colbuf((int64_t)off);
                GraphDesc D = gd[vgraph[v]];
                int32_t len0 = serial_gather(A, v, B, ex);
                int32_t m = serial_sort_merge(A, D, B, len0, false, false);
                Arrays Ag = A;
                Ag.shuffle_seed = A.shuffle_seed + (uint64_t)vgraph[v];
                serial_order(Ag, B, m, v, 1, D.vbase);
                for (int j = 0; j < m; ++j) { tmp_nbr[toff + j] = B.a_nbr[j]; tmp_val[toff + j] = B.a_val[j]; }
                cnt_out[i] = m;
                live_acc += (unsigned long long)len0;
                }
            }
            __syncthreads();
            continue;
        }
        const int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
        const int32_t acnt = A.vr[v].app_cnt;
        int len0 = 0;
        {
            int32_t idx = acnt - 1, base = A.vr[v].app_chunk;
            int c = idx >= 0 ? chunk_of(idx) : 0;
            while (idx >= 0) {
                int32_t cs = chunk_start(c);
                for (int32_t t0 = idx; t0 >= cs; t0 -= 64) {
                    int32_t t = t0 - lane;
                    bool valid = t >= cs;
                    int32_t s = base + 1 + (t - cs);
                    double val = 0; int32_t nb = 0;
                    if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; }
                    bool live = valid && val > 0;
                    uint64_t mask = __ballot(live);
                    int pos = len0 + popc64(mask & lt);
                    if (live) { L.a_nbr[pos] = nb; L.a_val[pos] = val; }
                    len0 += popc64(mask);
                }
                int32_t prev = A.e[base].nbr;
                idx = cs - 1; base = prev; --c;
            }
            for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
                int32_t s = s0 - lane;
                bool valid = s >= cp0;
                double val = 0; int32_t nb = 0;
                if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live) { L.a_nbr[pos] = nb; L.a_val[pos] = val; }
                len0 += popc64(mask);
            }
        }
        __syncthreads();
        constexpr int RT = (CAP + 63) / 64;
        constexpr bool RANK_FIRST = CAP <= 64;   // longer columns: the stable rank is O(n^2/64), the restatement O(n log n / 64)
        if (!sort64(lane < len0 ? (double)L.a_nbr[lane] : 0.0, len0, false)) {
            bool done = RANK_FIRST ? sc_rank_sort<false>(L, len0, [&](int q) { return (double)L.a_nbr[q]; }, lane) : false;
            __syncthreads();
            if (!done) {
                for (int q = lane; q < len0; q += 64) { L.rec[q].key = (double)L.a_nbr[q]; L.rec[q].idx = q; }
                __syncthreads();
                wave_std_sort<SRec, SRecLessKey, RT>(L.rec, len0, SRecLessKey(), WP, lane);
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
                const int32_t gi = vgraph[v], vb = gd[gi].vbase;
                uint64_t kb = keyed_order_base(A.shuffle_seed + (uint64_t)gi, v - vb, 1);
                if (sort64(lane < m ? keyed_order_dkey(kb, L.b_nbr[lane] - vb) : 0.0, m, false)) done = true; else
                done = RANK_FIRST ? sc_rank_sort<false>(L, m, [&](int q) { return keyed_order_dkey(kb, L.b_nbr[q] - vb); }, lane) : false;
                __syncthreads();
                if (!done) {
                    for (int q = lane; q < m; q += 64) { L.rec[q].key = keyed_order_dkey(kb, L.b_nbr[q] - vb); L.rec[q].idx = q; }
                    __syncthreads();
                    wave_std_sort<SRec, SRecLessKey, RT>(L.rec, m, SRecLessKey(), WP, lane);
                    __syncthreads();
                }
            } else if (A.o_n == ON_ASC) {
                if (sort64(lane < m ? L.b_val[lane] : 0.0, m, false)) done = true; else
                done = RANK_FIRST ? sc_rank_sort<false>(L, m, [&](int q) { return L.b_val[q]; }, lane) : false;
                __syncthreads();
                if (!done) {
                    for (int q = lane; q < m; q += 64) { L.rec[q].key = L.b_val[q]; L.rec[q].idx = q; }
                    __syncthreads();
                    wave_std_sort<SRec, SRecLessKey, RT>(L.rec, m, SRecLessKey(), WP, lane);
                    __syncthreads();
                }
            } else {
                if (sort64(lane < m ? L.b_val[lane] : 0.0, m, true)) done = true; else
                done = RANK_FIRST ? sc_rank_sort<true>(L, m, [&](int q) { return L.b_val[q]; }, lane) : false;
                __syncthreads();
                if (!done) {
                    for (int q = lane; q < m; q += 64) { L.rec[q].key = L.b_val[q]; L.rec[q].idx = q; }
                    __syncthreads();
                    wave_std_sort<SRec, SRecGreaterKey, RT>(L.rec, m, SRecGreaterKey(), WP, lane);
                    __syncthreads();
                }
            }
        }
        for (int j = lane; j < m; j += 64) {
            int x = L.rec[j].idx;
            tmp_nbr[toff + j] = L.b_nbr[x];
            tmp_val[toff + j] = L.b_val[x];
        }
        if (lane == 0) { cnt_out[i] = m; live_acc += (unsigned long long)len0; }
        __syncthreads();
    }
    if (lane == 0 && live_acc) atomicAdd(&live_total[(blockIdx.x & (LIVE_SLOTS - 1)) * LIVE_STRIDE], live_acc);
}

// work lists per capacity tier (5: <=32 (two columns per wave), 0: <=64, 1: <=192, 2: <=512 and what the long-column kernels
// do not take, 3: k_sc_merge_big, 4: k_sc_merge_huge)
__global__ __launch_bounds__(1024) void k_sc_tierlists(const int32_t* __restrict__ ext, int32_t S, int32_t keyed, int32_t* __restrict__ lists, int32_t* __restrict__ counts) {
    __shared__ int32_t s_cnt[8], s_base[8];
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int tier = -1;
    (void)keyed;
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (i < S) {
        int32_t e = ext[i];
        tier = e <= 32 ? 5 : e <= 64 ? 0 : (e <= 192 ? 1 : ((e > SCAP && e <= HUGECAP) ? (e <= MID1CAP ? 7 : (e <= MIDCAP ? 6 : (e <= BIGCAP ? 3 : 4))) : 2));
    }
    // one LDS atomic per wave and tier, one global atomic per workgroup and tier (same-address atomics serialise):
    // neighbouring columns stay neighbours in the list (locality of the staged rows)
    int32_t my = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint64_t mk = __ballot(tier == t);
        if (mk == 0ull) continue;
        int32_t base = 0;
        const int leader = __builtin_ctzll(mk);
        if (lane == leader) base = atomicAdd(&s_cnt[t], __popcll(mk));
        base = __shfl(base, leader);
        if (tier == t) my = base + __popcll(mk & lanemask_lt(lane));
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (tier >= 0) lists[(size_t)tier * S + s_base[tier] + my] = i;
}

// Columns of at most 32 slots (most of them after half the vertices are gone): TWO per wave, one per half-wave, entirely
// in registers -- the column is read one slot per lane, compacted, ordered by id and by o_n with group_sort<32> (the
// std::sort restatement for two 32-lane groups), merged with shuffles, and stored from the lanes.  A column whose sort
// hits the depth limit is handed to the 64-entry kernel's list.
__global__ __launch_bounds__(64) void k_sc_merge_half(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                      const uint32_t* __restrict__ order, const int64_t* __restrict__ tmp_off,
                                                      int32_t* __restrict__ tmp_nbr, double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out,
                                                      unsigned long long* __restrict__ live_total, const int32_t* __restrict__ worklist,
                                                      const int32_t* __restrict__ workcount, int32_t* __restrict__ list64, int32_t* __restrict__ count64) {
    __shared__ int32_t s_tmp[2 * (2 * 34 + 12)];
    const int lane = lane_id();
    const int gl = lane & 31, gbase = lane & 32;
    const uint64_t gmask = 0xFFFFFFFFull << gbase;
    const uint64_t lt = lanemask_lt(lane) & gmask;
    const bool keyed = (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN);
    const bool desc = (A.o_n == ON_DESC) && !keyed;
    const int32_t nwork = *workcount;
    unsigned long long live_acc = 0ull;   // lanes 0 and 32
    auto to_lane = [&](int target, int v) { return __builtin_amdgcn_ds_permute((gbase + target) << 2, v); };   // forward permute inside the half
    auto to_lane_d = [&](int target, double v) {
        const long long b = __double_as_longlong(v);
        const int lo = to_lane(target, (int)(uint32_t)b), hi = to_lane(target, (int)(uint32_t)((unsigned long long)b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
    };
    for (int32_t w0 = 2 * blockIdx.x; w0 < nwork; w0 += 2 * gridDim.x) {
        const int32_t wi = w0 + (lane >> 5);
        const bool have = wi < nwork;
        int32_t i = 0, v = 0, cp1 = 0, acnt = 0, ext = 0, cb0 = 0, cb1 = 0, cb2 = 0;
        int64_t toff = 0;
        if (have) {
            i = worklist[wi];
            v = (int32_t)order[i];
            toff = tmp_off[i];
            const int32_t cp0 = A.colptr[v];
            cp1 = A.colptr[v + 1];
            acnt = A.vr[v].app_cnt;
            ext = (cp1 - cp0) + acnt;
            if (acnt > 0) {   // bases of the (at most three) appended chunks
                const int ct = chunk_of(acnt - 1);
                int32_t base = A.vr[v].app_chunk;
                if (ct == 2) { cb2 = base; base = A.e[base].nbr; }
                if (ct >= 1) { cb1 = base; base = A.e[base].nbr; }
                cb0 = base;
            }
        }
        // one slot per lane, in the traversal order of :248-271 (appended entries newest first, then the CSR segment backwards)
        double val = 0; int32_t nb = 0;
        if (gl < ext) {
            int32_t sl;
            if (gl < acnt) { const int32_t a = acnt - 1 - gl; const int c = chunk_of(a); sl = (c == 0 ? cb0 : c == 1 ? cb1 : cb2) + 1 + (a - chunk_start(c)); }
            else sl = cp1 - 1 - (gl - acnt);
            val = A.e[sl].val; nb = A.e[sl].nbr;
        }
        const bool live = gl < ext && val > 0;
        const uint64_t lm = __ballot(live) & gmask;
        const int nlive = __popcll(lm);
        // compact the live entries to the front of the half (a permutation: dead lanes go behind)
        {
            const int below = __popcll(lm & lt);
            const int target = live ? below : nlive + (gl - below);
            nb = to_lane(target, nb);
            val = to_lane_d(target, val);
        }
        // order by id (:314-315)
        bool ok = true;
        {
            double key = (double)nb; int idx = gl, pos = gl;
            const bool want = have && nlive > 1;
            const bool okg = group_sort<false, 32>(key, idx, nlive, want, lane, s_tmp, &pos);
            if (want) {
                ok = okg;
                const double ve = __shfl(val, gbase + idx);
                const int target = gl < nlive ? pos : gl;
                nb = to_lane(target, (int32_t)key);
                val = to_lane_d(target, ve);
            }
        }
        // merge (:317-329): the first of a run of equal ids takes the sum, in sorted order
        int m = nlive;
        {
            const int32_t nbp = __shfl_up(nb, 1);
            const bool head = gl < nlive && (gl == 0 || nb != nbp);
            const uint64_t hm = __ballot(head) & gmask;
            m = __popcll(hm);
            if (m != nlive) {   // (group-uniform)
                double acc = val;
                const uint64_t above = (gl == 31) ? 0ull : ((hm >> (lane + 1)) & (0xFFFFFFFFull >> (gl + 1)));
                const int nexthead = above ? (gl + 1 + __builtin_ctzll(above)) : nlive;
                for (int q = 1; q < 32; ++q) {   // shuffles must be executed by the whole half: fixed trip count, predicated adds
                    const double vq = __shfl(val, gbase + ((gl + q) & 31));
                    if (head && gl + q < nexthead) acc += vq;
                }
                const int below = __popcll(hm & lt);
                const int target = head ? below : (gl < nlive ? m + (gl - below) : gl);
                nb = to_lane(target, nb);
                val = to_lane_d(target, acc);
            }
        }
        // order by o_n (:331-343) and store from the lanes
        {
            double key = val;
            if (keyed && have) {
                const int32_t gi = vgraph[v], vb = gd[gi].vbase;
                key = keyed_order_dkey(keyed_order_base(A.shuffle_seed + (uint64_t)gi, v - vb, 1), nb - vb);
            }
            int idx = gl, pos = gl;
            const bool want = have && m > 1;
            const bool okg = desc ? group_sort<true, 32>(key, idx, m, want, lane, s_tmp, &pos) : group_sort<false, 32>(key, idx, m, want, lane, s_tmp, &pos);
            if (want) ok = ok && okg;
            const int32_t nbo = __shfl(nb, gbase + idx);
            const double vo = __shfl(val, gbase + idx);
            if (have && ok && gl < m) { tmp_nbr[toff + pos] = nbo; tmp_val[toff + pos] = vo; }
        }
        if (have && gl == 0) {
            if (ok) { cnt_out[i] = m; live_acc += (unsigned long long)nlive; }
            else list64[atomicAdd(count64, 1)] = i;   // depth limit hit: the 64-entry kernel (launched after this one) takes it
        }
    }
    live_acc += __shfl_down(live_acc, 32);
    if (lane == 0 && live_acc) atomicAdd(&live_total[(blockIdx.x & (LIVE_SLOTS - 1)) * LIVE_STRIDE], live_acc);
}

void launch_sc_merge(const ScLaunch& X, const Arrays& A, const GraphDesc* gd, const int32_t* vgraph, const uint32_t* order, const int32_t* ext,
                     const int64_t* tmp_off, int32_t S, int32_t* tmp_nbr, double* tmp_val, int32_t* cnt_out, const ScScratch& SS,
                     unsigned long long* live_total, int32_t* lists, int32_t* counts, uint16_t* hugelists) {
    const int keyed = (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) ? 1 : 0;
    hipStream_t stream = X.main;
    hipLaunchKernelGGL(k_sc_tierlists, dim3((S + 1023) / 1024), dim3(1024), 0, stream, ext, S, keyed, lists, counts);
    // the tiers are independent of each other (only the <=64 kernel follows the <=32 one, which hands it the columns whose sort
    // hit the depth limit): the long-column kernels and the LDS tiers run beside the two big tiers on side streams
    const bool fork = X.side[0] && X.side[1];
    hipStream_t s1 = fork ? X.side[0] : stream, s2 = fork ? X.side[1] : stream;
    if (fork) {
        (void)hipEventRecord(X.ev[0], stream);
        (void)hipStreamWaitEvent(s1, X.ev[0], 0);
        (void)hipStreamWaitEvent(s2, X.ev[0], 0);
    }
    // long columns: whole column in LDS, one single-wave workgroup each; longer than the LDS record array (hubs of
    // weighted graphs): records in global scratch, a few workgroups
    // (one workgroup of 8 waves per CU at ~156 KB of LDS: the few columns beyond MIDCAP slots; the tier below: 4 waves, three to a CU)
    // (the <=512 tier goes first on the first side stream: the 8-wave kernel behind it needs a whole CU's LDS per workgroup and waits for
    // one to drain anyway -- config 5 has no such column and the kernel sat there for 4.7 ms; on the main stream this tier was the end of
    // the longest chain, <=32 -> <=64 -> <=512: 4.8 ms)
    unsigned g2 = (unsigned)(S < 256 * 6 * 16 ? S : 256 * 6 * 16);
    hipLaunchKernelGGL((k_sc_merge_t<SCAP, 192>), dim3(g2), dim3(64), 0, s1, A, gd, vgraph, order, ext, tmp_off, S, tmp_nbr, tmp_val, cnt_out, SS, live_total, lists + 2 * (size_t)S, counts + 2);
    hipLaunchKernelGGL((k_sc_merge_mw<8>), dim3(512), dim3(512), MW_BIG_LDS_BYTES, s1, A, gd, vgraph, order, ext, tmp_off, lists + 3 * (size_t)S, counts + 3,
                       tmp_nbr, tmp_val, cnt_out, live_total, (int32_t)BIGCAP, (int32_t)MW_BIG_QCAP);
    hipLaunchKernelGGL((k_sc_merge_mw<4>), dim3(1024), dim3(256), MW_MID_LDS_BYTES, s2, A, gd, vgraph, order, ext, tmp_off, lists + 6 * (size_t)S, counts + 6,
                       tmp_nbr, tmp_val, cnt_out, live_total, (int32_t)MIDCAP, (int32_t)MW_MID_QCAP);
    hipLaunchKernelGGL(k_sc_merge_big, dim3(2048), dim3(64), MID1_LDS_BYTES, s2, A, gd, vgraph, order, ext, tmp_off, lists + 7 * (size_t)S, counts + 7,
                       tmp_nbr, tmp_val, cnt_out, live_total, nullptr, (int32_t)MID1CAP);   // (stop lists in LDS: no global list region)
    hipLaunchKernelGGL(k_sc_merge_huge, dim3(NHUGE), dim3(64), 0, s1, A, gd, vgraph, order, ext, tmp_off, lists + 4 * (size_t)S, counts + 4,
                       tmp_nbr, tmp_val, cnt_out, live_total, hugelists, SS.rec, SS.top, SS.cap, SS.flags);
    unsigned g1 = (unsigned)(S < 256 * 16 * 8 ? S : 256 * 16 * 8);
    hipLaunchKernelGGL((k_sc_merge_t<192, 64>), dim3(g1), dim3(64), 0, s2, A, gd, vgraph, order, ext, tmp_off, S, tmp_nbr, tmp_val, cnt_out, SS, live_total, lists + (size_t)S, counts + 1);
    unsigned gh = (unsigned)(S < 256 * 32 * 4 ? (S + 1) / 2 : 256 * 32 * 4);
    if (gh == 0) gh = 1;
    hipLaunchKernelGGL(k_sc_merge_half, dim3(gh), dim3(64), 0, stream, A, gd, vgraph, order, tmp_off, tmp_nbr, tmp_val, cnt_out, live_total, lists + 5 * (size_t)S, counts + 5, lists, counts);
    unsigned g0 = (unsigned)(S < 256 * 32 * 4 ? S : 256 * 32 * 4);
    hipLaunchKernelGGL((k_sc_merge_t<64, -1>), dim3(g0), dim3(64), 0, stream, A, gd, vgraph, order, ext, tmp_off, S, tmp_nbr, tmp_val, cnt_out, SS, live_total, lists, counts);
    if (fork) {
        (void)hipEventRecord(X.ev[1], s1);
        (void)hipEventRecord(X.ev[2], s2);
        (void)hipStreamWaitEvent(stream, X.ev[1], 0);
        (void)hipStreamWaitEvent(stream, X.ev[2], 0);
    }
}

// Long columns (SCAP < extent <= BIGCAP): one single-wave workgroup per column with the whole
// column in LDS as 16-byte records {id as double, weight}; both sorts are the wave-parallel std::sort
// restatement (exact under ties), gather / merge / stores by the wave.
struct Rec2 { double a; double b; };
struct Rec2LessA { __device__ bool operator()(const Rec2& x, const Rec2& y) const { return x.a < y.a; } };
struct Rec2LessB { __device__ bool operator()(const Rec2& x, const Rec2& y) const { return x.b < y.b; } };
struct Rec2GreaterB { __device__ bool operator()(const Rec2& x, const Rec2& y) const { return x.b > y.b; } };
// keyed neighbour order (o_n = random / coarsen): the key is a hash of the neighbour id, recomputed per comparison
struct Rec2LessKeyed {
    uint64_t kb;
    int32_t vbase;
    __device__ bool operator()(const Rec2& x, const Rec2& y) const { return keyed_order_dkey(kb, (int32_t)x.a - vbase) < keyed_order_dkey(kb, (int32_t)y.a - vbase); }
};

// HUGE = false: SCAP < extent <= BIGCAP, records in LDS.  HUGE = true: BIGCAP < extent <= HUGECAP (a hub of a
// weighted graph), records in global scratch (L2), both sorts by the wave-parallel restatement.
template <bool HUGE>
__device__ __forceinline__ void sc_merge_long_body(const Arrays& A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                   const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                   const int64_t* __restrict__ tmp_off, const int32_t* __restrict__ list,
                                                   const int32_t* __restrict__ count, int32_t* __restrict__ tmp_nbr,
                                                   double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out,
                                                   unsigned long long* __restrict__ live_total, uint16_t* __restrict__ lists,
                                                   Rec2* lds_R, int32_t lds_cap, Rec2* glob_R, unsigned long long* glob_top, int64_t glob_cap, int32_t* glob_flags) {
    constexpr int LCAP = HUGE ? HUGECAP : BIGCAP;
    __shared__ uint32_t s_segmark[LCAP / 32 + 2];
    __shared__ int32_t s_stk[3 * 48];
    __shared__ unsigned long long s_off;
    // HUGE: the sorts partition in global memory only down to segments of STG records, which are sorted through in LDS (wave_std_sort_staged)
    constexpr int STG = HUGE ? 3072 : 1;
    __shared__ Rec2 s_stg[STG];
    __shared__ uint16_t s_stg_ul[STG + 2], s_stg_dl[STG + 2];
    __shared__ uint32_t s_stg_mark[STG / 32 + 2];
    __shared__ uint32_t s_stg_tab[HUGE ? 512 : 1];
    __shared__ uint16_t s_stg_tab2[HUGE ? 512 : 2];
    WaveSortPtrs WL;   // (assigned, not brace-initialised: a constant aggregate of LDS addresses is rejected as a static initialiser)
    WL.ulist = s_stg_ul; WL.dlist = s_stg_dl; WL.segmark = s_stg_mark; WL.stk = s_stk;
    // stop lists of the partition emulation: in LDS behind the records (long columns), in global scratch, one region per
    // workgroup (huge columns)
    uint16_t* const mylists = HUGE ? lists + (size_t)blockIdx.x * 2 * (LCAP + 2) : reinterpret_cast<uint16_t*>(lds_R + lds_cap);
    const WaveSortPtrs WP = {mylists, mylists + ((HUGE ? LCAP : lds_cap) + 2), s_segmark, s_stk};
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt(lane);
    const int32_t nbig = *count;
    unsigned long long live_acc = 0ull;
    for (int32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const int32_t i = list[bi];
        const int32_t v = (int32_t)order[i];
        const int64_t toff = tmp_off[i];
        const int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
        const int32_t acnt = A.vr[v].app_cnt;
        Rec2* R = lds_R;
        if (HUGE) {
            if (lane == 0) {
                s_off = atomicAdd(glob_top, (unsigned long long)ext[i]);
                if ((int64_t)(s_off + (unsigned long long)ext[i]) > glob_cap) { s_off = ~0ull; glob_flags[FLAG_SCR] = 1; cnt_out[i] = 0; }
            }
            __syncthreads();
            if (s_off == ~0ull) { __syncthreads(); continue; }   // scratch budget too small: the call is repeated with more
            R = glob_R + s_off;
        }
        int len0 = 0;
        {
            int32_t idx = acnt - 1, base = A.vr[v].app_chunk;
            int c = idx >= 0 ? chunk_of(idx) : 0;
            while (idx >= 0) {
                int32_t cs = chunk_start(c);
                for (int32_t t0 = idx; t0 >= cs; t0 -= 64) {
                    int32_t t = t0 - lane;
                    bool valid = t >= cs;
                    int32_t s = base + 1 + (t - cs);
                    double val = 0; int32_t nb = 0;
                    if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; }
                    bool live = valid && val > 0;
                    uint64_t mask = __ballot(live);
                    int pos = len0 + popc64(mask & lt);
                    if (live) { R[pos].a = (double)nb; R[pos].b = val; }
                    len0 += popc64(mask);
                }
                int32_t prev = A.e[base].nbr;
                idx = cs - 1; base = prev; --c;
            }
            for (int32_t s0 = cp1 - 1; s0 >= cp0; s0 -= 64) {
                int32_t s = s0 - lane;
                bool valid = s >= cp0;
                double val = 0; int32_t nb = 0;
                if (valid) { const Slot g = A.e[s]; val = g.val; nb = g.nbr; }
                bool live = valid && val > 0;
                uint64_t mask = __ballot(live);
                int pos = len0 + popc64(mask & lt);
                if (live) { R[pos].a = (double)nb; R[pos].b = val; }
                len0 += popc64(mask);
            }
        }
        __syncthreads();
        // sort by id (:314-315), std::sort semantics (the gather order is the traversal order)
        if constexpr (HUGE) wave_std_sort_staged<Rec2, Rec2LessA, 8>(R, len0, Rec2LessA(), WP.ulist, WP.dlist, s_stk, s_stg, STG, WL, s_stg_tab, s_stg_tab2, lane);
        else wave_std_sort<Rec2>(R, len0, Rec2LessA(), WP, lane);
        __syncthreads();
        // merge (:317-329), 64 positions at a time: heads by ballot, sums in sorted order; all reads of a chunk
        // (its look-ahead included) come before its writes, which land at or below the chunk
        int m = 0;
        for (int32_t p0 = 0; p0 < len0; p0 += 64) {
            const int32_t p = p0 + lane;
            const bool act = p < len0;
            Rec2 me = {0.0, 0.0};
            double prev = -1.0;
            if (act) { me = R[p]; if (p > 0) prev = R[p - 1].a; }
            const bool head = act && me.a != prev;
            if (head) for (int32_t q = p + 1; q < len0 && R[q].a == me.a; ++q) me.b += R[q].b;
            const uint64_t mask = __ballot(head);
            const int32_t x = m + popc64(mask & lt);
            __syncthreads();
            if (head) R[x] = me;
            __syncthreads();
            m += popc64(mask);
        }
        if constexpr (HUGE) {
            if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) { const int32_t gi = vgraph[v]; Rec2LessKeyed lk; lk.vbase = gd[gi].vbase; lk.kb = keyed_order_base(A.shuffle_seed + (uint64_t)gi, v - lk.vbase, 1); wave_std_sort_staged<Rec2, Rec2LessKeyed, 8>(R, m, lk, WP.ulist, WP.dlist, s_stk, s_stg, STG, WL, s_stg_tab, s_stg_tab2, lane); }   // :339-343
            else if (A.o_n == ON_ASC) wave_std_sort_staged<Rec2, Rec2LessB, 8>(R, m, Rec2LessB(), WP.ulist, WP.dlist, s_stk, s_stg, STG, WL, s_stg_tab, s_stg_tab2, lane);
            else wave_std_sort_staged<Rec2, Rec2GreaterB, 8>(R, m, Rec2GreaterB(), WP.ulist, WP.dlist, s_stk, s_stg, STG, WL, s_stg_tab, s_stg_tab2, lane);   // :331-338
        } else {
        if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) { const int32_t gi = vgraph[v]; Rec2LessKeyed lk; lk.vbase = gd[gi].vbase; lk.kb = keyed_order_base(A.shuffle_seed + (uint64_t)gi, v - lk.vbase, 1); wave_std_sort<Rec2>(R, m, lk, WP, lane); }   // :339-343
        else if (A.o_n == ON_ASC) wave_std_sort<Rec2>(R, m, Rec2LessB(), WP, lane);
        else wave_std_sort<Rec2>(R, m, Rec2GreaterB(), WP, lane);   // :331-338
        }
        __syncthreads();
        for (int32_t j = lane; j < m; j += 64) { tmp_nbr[toff + j] = (int32_t)R[j].a; tmp_val[toff + j] = R[j].b; }
        if (lane == 0) { cnt_out[i] = m; live_acc += (unsigned long long)len0; }
        __syncthreads();
    }
    if (lane == 0 && live_acc) atomicAdd(&live_total[(blockIdx.x & (LIVE_SLOTS - 1)) * LIVE_STRIDE], live_acc);
}

__global__ __launch_bounds__(64) void k_sc_merge_big(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                     const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                     const int64_t* __restrict__ tmp_off, const int32_t* __restrict__ list,
                                                     const int32_t* __restrict__ count, int32_t* __restrict__ tmp_nbr,
                                                     double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out,
                                                     unsigned long long* __restrict__ live_total, uint16_t* __restrict__ lists, int32_t lcap) {
    extern __shared__ Rec2 R_lds[];   // lcap records, then the two stop lists of the sort
    sc_merge_long_body<false>(A, gd, vgraph, order, ext, tmp_off, list, count, tmp_nbr, tmp_val, cnt_out, live_total, lists, R_lds, lcap, nullptr, nullptr, 0, nullptr);
}

// ---------------------------------------------------------------------------
// The same restatement of std::sort for ONE array shared by several waves (long surviving columns).  The introsort loop is
// run level by level: every segment longer than 16 that exists at a level is partitioned by one wave (the wave-parallel
// Hoare partition of wave_std_sort, its stop lists kept at the segment's own offsets), its two parts are queued for the next
// level or marked as final; a workgroup barrier separates the levels.  The order in which segments are partitioned does not
// matter (they are disjoint and std::sort's recursion treats them independently), so the permutation is std::sort's.
// The final insertion sort (independent stable sorts of the marked segments) is spread over all threads.
// segq: two queues of QCAP segments (first, last, depth); qcnt[2]; segmark: (n+31)/32+1 words.
// ---------------------------------------------------------------------------
template <class T, class Less, int NW>
__device__ void block_std_sort(T* a, const int n, Less less, uint16_t* ulist, uint16_t* dlist, uint32_t* segmark, int32_t* segq, int32_t* qcnt, const int QCAP) {
    constexpr int NTB = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt = lanemask_lt(lane);
    for (int q = tid; q < (n + 31) / 32 + 1; q += NTB) segmark[q] = 0u;
    if (tid == 0) {
        int depth0 = 0;
        for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
        qcnt[0] = n > 16 ? 1 : 0; qcnt[1] = 0;
        segq[0] = 0; segq[1] = n; segq[2] = 2 * depth0;
    }
    __syncthreads();
    if (n < 2) return;
    if (n <= 16) {
        if (tid == 0) gs_insertion_sort<T>(a, n, less);
        __syncthreads();
        return;
    }
    int cur = 0;
    while (true) {
        const int ncur = qcnt[cur];
        if (ncur == 0) break;
        int32_t* const qin = segq + cur * 3 * QCAP;
        int32_t* const qout = segq + (cur ^ 1) * 3 * QCAP;
        for (int sidx = wave; sidx < ncur; sidx += NW) {
            const int first = qin[3 * sidx], last = qin[3 * sidx + 1];
            int depth = qin[3 * sidx + 2];
            if (depth == 0) {   // depth limit: heap sort (std::__partial_sort), by one lane; the segment stays one sorted run
                if (lane == 0) { gs_heap_sort<T>(a, first, last, less); atomicOr(&segmark[first >> 5], 1u << (first & 31)); }
                WAVE_SYNC();
                continue;
            }
            --depth;
            if (lane == 0) {   // __move_median_to_first(first, first+1, mid, last-1)
                int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                int pick;
                if (less(a[ia], a[ib])) {
                    if (less(a[ib], a[ic])) pick = ib;
                    else if (less(a[ia], a[ic])) pick = ic;
                    else pick = ia;
                } else if (less(a[ia], a[ic])) pick = ia;
                else if (less(a[ib], a[ic])) pick = ic;
                else pick = ib;
                T t = a[first]; a[first] = a[pick]; a[pick] = t;
            }
            WAVE_SYNC();
            const T pv = a[first];
            uint16_t* const ul = ulist + first;
            uint16_t* const dl = dlist + first;
            int nu = 0, nd = 0;
            for (int p0 = first + 1; p0 < last; p0 += 64) {
                int p = p0 + lane;
                bool stop = (p < last) && !less(a[p], pv);
                uint64_t mk = __ballot(stop);
                if (stop) ul[nu + popc64(mk & lt)] = (uint16_t)p;
                nu += popc64(mk);
            }
            for (int p0 = last - 1; p0 > first; p0 -= 64) {
                int p = p0 - lane;
                bool stop = (p > first) && !less(pv, a[p]);
                uint64_t mk = __ballot(stop);
                if (stop) dl[nd + popc64(mk & lt)] = (uint16_t)p;
                nd += popc64(mk);
            }
            if (lane == 0) dl[nd] = (uint16_t)first;   // the pivot itself stops the down-scan
            WAVE_SYNC();
            int k = 0;
            {
                const int tmax = nu < nd ? nu : nd;
                bool open = true;
                for (int t0 = 0; t0 < tmax && open; t0 += 64) {
                    int t = t0 + lane;
                    bool ok = (t < tmax) && (ul[t] < dl[t]);
                    uint64_t mk = __ballot(ok);
                    uint64_t inv = ~mk;
                    int run = inv ? __builtin_ctzll(inv) : 64;
                    k += run;
                    open = (run == 64);
                }
            }
            T xu, xd;
            for (int t0 = 0; t0 < k; t0 += 64) {
                int t = t0 + lane;
                if (t < k) { xu = a[ul[t]]; xd = a[dl[t]]; }
                WAVE_SYNC();
                if (t < k) { a[ul[t]] = xd; a[dl[t]] = xu; }
                WAVE_SYNC();
            }
            int cut;
            {
                int cu = (k < nu) ? (int)ul[k] : 0x7FFFFFFF;
                int cd = (k > 0) ? (int)dl[k - 1] : 0x7FFFFFFF;
                cut = cu < cd ? cu : cd;
            }
            WAVE_SYNC();
            if (lane == 0) {
                // [first, cut) and [cut, last): longer than 16 -> next level, else a final segment (marked at its start)
                if (cut - first > 16) { const int qi = atomicAdd(&qcnt[cur ^ 1], 1); qout[3 * qi] = first; qout[3 * qi + 1] = cut; qout[3 * qi + 2] = depth; }
                else atomicOr(&segmark[first >> 5], 1u << (first & 31));
                if (last - cut > 16) { const int qi = atomicAdd(&qcnt[cur ^ 1], 1); qout[3 * qi] = cut; qout[3 * qi + 1] = last; qout[3 * qi + 2] = depth; }
                else if (cut < last) atomicOr(&segmark[cut >> 5], 1u << (cut & 31));
            }
            WAVE_SYNC();
        }
        __syncthreads();
        if (tid == 0) qcnt[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    // final insertion sort: thread t takes the segments that start in the 32-position words t, t+NTB, ...
    for (int w0 = tid; w0 * 32 < n; w0 += NTB) {
        uint32_t bits = segmark[w0];
        while (bits) {
            const int s0 = w0 * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            int e0 = n;
            if (bits) e0 = w0 * 32 + __builtin_ctz(bits);
            else {
                for (int w1 = w0 + 1; w1 * 32 < n; ++w1) { uint32_t bb = segmark[w1]; if (bb) { e0 = w1 * 32 + __builtin_ctz(bb); break; } }
            }
            if (e0 > n) e0 = n;
            for (int i = s0 + 1; i < e0; ++i) {
                T v = a[i];
                int j = i - 1;
                while (j >= s0 && less(v, a[j])) { a[j + 1] = a[j]; --j; }
                a[j + 1] = v;
            }
        }
    }
    __syncthreads();
}

// Long surviving columns with several waves per column: gather (all threads, traversal order kept by a prefix count), sort by
// id, merge, order by o_n, store -- the two sorts are block_std_sort.  NW waves, records + stop lists + segment queues in
// dynamic LDS: lcap * 16 + 2 * (lcap + 2) * 2 + 2 * 3 * QCAP * 4 bytes.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_sc_merge_mw(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                          const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                          const int64_t* __restrict__ tmp_off, const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ count, int32_t* __restrict__ tmp_nbr,
                                                          double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out,
                                                          unsigned long long* __restrict__ live_total, int32_t lcap, int32_t qcap) {
    constexpr int NTB = NW * 64;
    extern __shared__ Rec2 R_mw[];
    Rec2* const R = R_mw;
    uint16_t* const ulist = reinterpret_cast<uint16_t*>(R + lcap);
    uint16_t* const dlist = ulist + (lcap + 2);
    int32_t* const segq = reinterpret_cast<int32_t*>(dlist + (lcap + 2));   // (lcap is even: 4-byte aligned)
    __shared__ uint32_t s_segmark[BIGCAP / 32 + 2];
    __shared__ int32_t s_qcnt[2];
    __shared__ int32_t s_seg_base[40], s_seg_first[40], s_seg_pref[41], s_nseg;   // the column as runs of slots in traversal order
    __shared__ int32_t s_wtot[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt = lanemask_lt(lane);
    const int32_t nbig = *count;
    unsigned long long live_acc = 0ull;
    for (int32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const int32_t i = list[bi];
        const int32_t v = (int32_t)order[i];
        const int64_t toff = tmp_off[i];
        if (tid == 0) {
            // traversal order (:248-271): appended chunks newest first (each backwards), then the CSR segment backwards
            const int32_t cp0 = A.colptr[v], cp1 = A.colptr[v + 1];
            const int32_t acnt = A.vr[v].app_cnt;
            int ns = 0, pref = 0;
            int32_t idx = acnt - 1, base = A.vr[v].app_chunk;
            int c = idx >= 0 ? chunk_of(idx) : 0;
            while (idx >= 0) {
                const int32_t cs = chunk_start(c);
                s_seg_base[ns] = base + 1; s_seg_first[ns] = idx - cs;   // slot of traversal position 0 of this run = base + 1 + (idx - cs), then descending
                s_seg_pref[ns] = pref; pref += idx - cs + 1; ++ns;
                const int32_t prev = A.e[base].nbr;
                idx = cs - 1; base = prev; --c;
            }
            s_seg_base[ns] = cp0; s_seg_first[ns] = cp1 - 1 - cp0; s_seg_pref[ns] = pref; pref += cp1 - cp0; ++ns;
            s_seg_pref[ns] = pref;
            s_nseg = ns;
        }
        __syncthreads();
        const int32_t extv = s_seg_pref[s_nseg];
        int32_t len0 = 0;
        for (int32_t e0 = 0; e0 < extv; e0 += NTB) {
            const int32_t e = e0 + tid;
            double val = 0; int32_t nb = 0;
            if (e < extv) {
                int sgi = 0;
                while (e >= s_seg_pref[sgi + 1]) ++sgi;
                const int32_t sl = s_seg_base[sgi] + s_seg_first[sgi] - (e - s_seg_pref[sgi]);
                const Slot g = A.e[sl]; val = g.val; nb = g.nbr;
            }
            const bool live = e < extv && val > 0;
            const uint64_t mask = __ballot(live);
            if (lane == 0) s_wtot[wave] = popc64(mask);
            __syncthreads();
            int32_t before = len0, tot = 0;
            for (int w = 0; w < NW; ++w) { const int32_t t = s_wtot[w]; if (w < wave) before += t; tot += t; }
            if (live) { Rec2 r; r.a = (double)nb; r.b = val; R[before + popc64(mask & lt)] = r; }
            len0 += tot;
            __syncthreads();
        }
        // sort by id (:314-315)
        block_std_sort<Rec2, Rec2LessA, NW>(R, len0, Rec2LessA(), ulist, dlist, s_segmark, segq, s_qcnt, qcap);
        // merge (:317-329), NTB positions at a time: heads by comparison with the predecessor, sums in sorted order; all reads of a
        // block (its look-ahead included) come before its writes, which land at or below the block
        int32_t m = 0;
        for (int32_t p0 = 0; p0 < len0; p0 += NTB) {
            const int32_t p = p0 + tid;
            const bool act = p < len0;
            Rec2 me = {0.0, 0.0};
            double prev = -1.0;
            if (act) { me = R[p]; if (p > 0) prev = R[p - 1].a; }
            const bool head = act && me.a != prev;
            if (head) for (int32_t q = p + 1; q < len0 && R[q].a == me.a; ++q) me.b += R[q].b;
            const uint64_t mask = __ballot(head);
            if (lane == 0) s_wtot[wave] = popc64(mask);
            __syncthreads();
            int32_t before = m, tot = 0;
            for (int w = 0; w < NW; ++w) { const int32_t t = s_wtot[w]; if (w < wave) before += t; tot += t; }
            if (head) R[before + popc64(mask & lt)] = me;
            m += tot;
            __syncthreads();
        }
        // order by o_n (:331-343)
        if (A.o_n == ON_RANDOM || A.o_v == OV_COARSEN) {
            const int32_t gi = vgraph[v];
            Rec2LessKeyed lk; lk.vbase = gd[gi].vbase; lk.kb = keyed_order_base(A.shuffle_seed + (uint64_t)gi, v - lk.vbase, 1);
            block_std_sort<Rec2, Rec2LessKeyed, NW>(R, m, lk, ulist, dlist, s_segmark, segq, s_qcnt, qcap);
        } else if (A.o_n == ON_ASC) block_std_sort<Rec2, Rec2LessB, NW>(R, m, Rec2LessB(), ulist, dlist, s_segmark, segq, s_qcnt, qcap);
        else block_std_sort<Rec2, Rec2GreaterB, NW>(R, m, Rec2GreaterB(), ulist, dlist, s_segmark, segq, s_qcnt, qcap);
        for (int32_t j = tid; j < m; j += NTB) { tmp_nbr[toff + j] = (int32_t)R[j].a; tmp_val[toff + j] = R[j].b; }
        if (tid == 0) { cnt_out[i] = m; live_acc += (unsigned long long)len0; }
        __syncthreads();
    }
    if (tid == 0 && live_acc) atomicAdd(&live_total[(blockIdx.x & (LIVE_SLOTS - 1)) * LIVE_STRIDE], live_acc);
}

__global__ __launch_bounds__(64) void k_sc_merge_huge(Arrays A, const GraphDesc* __restrict__ gd, const int32_t* __restrict__ vgraph,
                                                      const uint32_t* __restrict__ order, const int32_t* __restrict__ ext,
                                                      const int64_t* __restrict__ tmp_off, const int32_t* __restrict__ list,
                                                      const int32_t* __restrict__ count, int32_t* __restrict__ tmp_nbr,
                                                      double* __restrict__ tmp_val, int32_t* __restrict__ cnt_out,
                                                      unsigned long long* __restrict__ live_total, uint16_t* __restrict__ lists,
                                                      SRec* __restrict__ scratch, unsigned long long* __restrict__ scratch_top, int64_t scratch_cap,
                                                      int32_t* __restrict__ flags) {
    static_assert(sizeof(Rec2) == sizeof(SRec), "the long-column records borrow the output pass's record scratch");
    sc_merge_long_body<true>(A, gd, vgraph, order, ext, tmp_off, list, count, tmp_nbr, tmp_val, cnt_out, live_total, lists, nullptr, 0,
                             reinterpret_cast<Rec2*>(scratch), scratch_top, scratch_cap, flags);
}

// Pass B (the prefix-sum compaction): row r of the output belongs to the surviving vertex i with
// row_off[i] <= r < row_off[i+1].  One lane per ROW (tiles of 64 consecutive rows per wave, a
// contiguous run of tiles per wave), so lanes stay busy whatever the column lengths are; the
// owner of each row is found in the tile's window of row_off by a 6-step search over lane
// registers, and the 64x3 doubles are staged through LDS so that every store is a contiguous
// 512-byte wave store.
__global__ __launch_bounds__(256) void k_sc_compact(const uint32_t* __restrict__ order, const int32_t* __restrict__ cnt,
                                                    const int64_t* __restrict__ row_off, const int64_t* __restrict__ tmp_off,
                                                    const int32_t* __restrict__ tmp_nbr, const double* __restrict__ tmp_val,
                                                    int32_t S, double* __restrict__ out, int64_t out_cap) {
    (void)cnt;
    __shared__ double stage[4][192];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t m_total = row_off[S];
    if (m_total > out_cap) return;   // the caller's buffer is too small: nothing is written, the row count is reported
    const int64_t ntiles = (m_total + 63) >> 6;
    const int64_t per = (ntiles + nwaves - 1) / nwaves;
    int64_t t0 = wave * per, t1 = t0 + per;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 >= t1) return;
    // owner of the first row of this wave's run: last i with row_off[i] <= r
    int32_t i0;
    {
        const int64_t r = t0 << 6;
        int32_t lo = 0, hi = S;   // invariant: row_off[lo] <= r < row_off[hi] (row_off[S] = m_total > r)
        while (hi - lo > 1) {
            int32_t mid = (lo + hi) >> 1;
            if (row_off[mid] <= r) lo = mid; else hi = mid;
        }
        i0 = lo;
    }
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t r = (t << 6) + lane;
        const bool valid = r < m_total;
        int32_t owner = i0;
        int64_t obase = 0;
        // window search; windows advance by 63 while some lane's row lies beyond the window
        int32_t wbase = i0;
        bool found = false;
        while (true) {
            int32_t idx = wbase + lane;
            int64_t ro = row_off[idx <= S ? idx : S];
            // largest p in [0,63] with ro_p <= r
            int p = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                int q = p + step;
                int64_t rq = __shfl(ro, q);
                if (rq <= r) p = q;
            }
            int64_t rp = __shfl(ro, p);
            bool inside = (p < 63) || (wbase + 63 >= S);   // p == 63 may mean "further right"
            if (!found && (inside || !valid)) { owner = wbase + p; obase = rp; found = true; }
            if (__ballot(!found) == 0ull) break;
            wbase += 63;
        }
        if (owner >= S) owner = S - 1;
        double f0 = 0, f1 = 0, f2 = 0;
        if (valid) {
            const int64_t src = tmp_off[owner] + (r - obase);
            f0 = (double)tmp_nbr[src];
            f1 = (double)order[owner];
            f2 = tmp_val[src];
        }
        double* st = stage[wv];
        st[lane * 3 + 0] = f0; st[lane * 3 + 1] = f1; st[lane * 3 + 2] = f2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t o0 = (t << 6) * 3;
        const int64_t lim = m_total * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t d = o0 + lane + 64 * k;
            if (d < lim) out[d] = st[lane + 64 * k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        i0 = __shfl(owner, 63);
    }
}

// per-graph row pointers: first output position of each graph
__global__ void k_graph_rows(const int64_t* __restrict__ surv_base, const int64_t* __restrict__ row_off, int32_t G, int64_t* __restrict__ out_ptr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    out_ptr[g] = row_off[surv_base[g]];
}

#endif  // !RLAP_ELIM_PQ_TU
}  // namespace rlap
