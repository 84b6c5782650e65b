/* rlap_hip.h -- C ABI of librlap_hip.so, the MI355X (gfx950) implementation of
 * rLap's approximate-Cholesky / randomized Schur-complement augmentor.
 *
 * This is the drop-in boundary for ONE path of kvignesh1420/rlap:
 *   torch op  extension_cpp::approximate_cholesky(Tensor edge_info, int num_nodes,
 *             int num_remove, str o_v, str o_n) -> Tensor
 *             (reference: rlap/csrc/py_api_binder.cc:54-69,80-88)
 *   torch op  extension_cpp::identity(Tensor a) -> Tensor
 *             (reference: rlap/csrc/py_api_binder.cc:71-76)
 * which are what rlap/ops.py:52-58 and :61-63 call.  Plain pointers and sizes
 * only; every array pointer is DEVICE memory unless its name starts with h_.
 * All work is enqueued on the handle's HIP stream without any host synchronisation
 * in between (every workspace size is an upper bound computed from E, n and G; input
 * checks and growth limits are evaluated by the kernels); the call returns after ONE
 * read-back of the scalars it reports. (Inputs of >= 2^21 entries: one more, early, 32-byte read-back
 * decides whether the COO sort can be skipped because the input is sorted already.)
 * Threading: a handle owns its workspace and serialises the calls made on it (a mutex);
 * for concurrent calls give every host thread its own handle (and stream) -- the
 * reference builds a fresh ApproximateCholesky per call (py_api_binder.cc:57).
 */
#ifndef RLAP_HIP_H
#define RLAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rlap_handle_s* rlap_handle;

/* o_v: reference strings "random" / "degree" / "coarsen" (rlap/ops.py:49) */
enum { RLAP_OV_RANDOM = 0, RLAP_OV_DEGREE = 1, RLAP_OV_COARSEN = 2 };
/* o_n: reference strings "asc" / "desc" / "random" (rlap/ops.py:50) */
enum { RLAP_ON_ASC = 0, RLAP_ON_DESC = 1, RLAP_ON_RANDOM = 2 };

/* status codes */
enum {
    RLAP_OK = 0,
    RLAP_E_NOT_SYMMETRIC = 1, /* reference: prints + exit(0), factorizers.cc:19-22 */
    RLAP_E_INDEX_RANGE = 2,
    RLAP_E_BAD_ARG = 3,
    RLAP_E_POOL_OVERFLOW = 4, /* internal growth limits; the call retries with a larger */
    RLAP_E_LOG_OVERFLOW = 5,  /* workspace by itself, these only surface if that fails  */
    RLAP_E_RNG_OVERFLOW = 6,
    RLAP_E_OUT_OVERFLOW = 7,  /* out_cap_rows too small; *needed rows reported */
    RLAP_E_HIP = 8,
    RLAP_E_TOO_LARGE = 9,     /* nnz + growth pool exceeds int32 slot ids */
    RLAP_E_INTERNAL = 10,
    RLAP_E_WORKSPACE = 11     /* caller-provided workspace / uniform table too small: rlap_workspace_needed() says how much */
};

typedef struct {
    int64_t nnz;          /* directed entries after zero-drop + duplicate sum */
    int64_t n_eliminated; /* sum over graphs of min(t, n-1) (preconditioner.cc:358) */
    int64_t n_draws;      /* uniforms consumed (max over graphs)              */
    int64_t out_rows;     /* rows of sc_edge_info                             */
    int64_t live_entries; /* L: live directed entries read by the output pass */
    float ms_setup;       /* COO->CSR, symmetry, twins, PQ init (HIP events)  */
    float ms_elim;        /* elimination kernel                               */
    float ms_output;      /* sc_order + sc_merge + sc_compact                 */
    float ms_sc_merge;    /* output pass A alone                              */
    float ms_sc_compact;  /* output pass B alone (ballot/prefix compaction)   */
    float ms_total;
    int32_t n_retries;    /* times the call was repeated with a larger workspace (RLAP_E_*_OVERFLOW inside) */
    int32_t reserved;     /* COO sort: 0 done, 1 skipped (input in (col,row) order), 2 skipped, input in (row,col) order read transposed */
    int64_t n_rounds;     /* elimination: batch rounds, summed over graphs              */
    int64_t n_singles;    /* elimination: vertices that took the single-vertex path     */
} rlap_stats;

/* Lifetime.  A handle binds to the HIP device current at creation.  It owns a few KB of tables (allocated in
 * rlap_create); every per-call buffer is carved from ONE workspace arena and the cached MT19937-64 uniform table is a
 * second buffer -- the caller's (rlap_set_workspace) or, without one, the handle's own (one hipMalloc each, repeated only
 * when a call is of a larger size class than any before it; never a hipFree/hipMalloc in a call that fits). */
int rlap_create(rlap_handle* out);
int rlap_destroy(rlap_handle h);
int rlap_set_stream(rlap_handle h, void* hip_stream); /* hipStream_t; NULL = default */

/* Where the sampling uniforms come from (SURVEY section 7 step 7 / 8(b) `mode`).  0 = "exact" (default): the first outputs of
 * the default-seeded std::mt19937_64 in elimination order, preconditioner.cc:356-357,386 -- results equal the reference's.
 * 1 = "frontier": the j-th uniform of a vertex's elimination is a function of (seed, vertex, j); same distribution, no order
 * in which draws must be made (for o_v = random the multi-CU kernel then waits for nothing but its true neighbours);
 * bit-exact against the oracle in the same mode, NOT against the reference.  Applies to the handle's later calls. */
int rlap_set_rng_mode(rlap_handle h, int mode);

/* Workspace contract (SURVEY 8(b) "caller owns every buffer"; the reference's only allocation is the torch tensor of its
 * result, py_api_binder.cc:42).
 *   rlap_workspace_bytes : upper bound, for a fresh handle, of the arena bytes and of the uniform-table entries a call on
 *                          E directed input entries, n_total vertices (summed over the G graphs of a batch) needs, whatever
 *                          num_remove and the split of n_total over the graphs are.
 *   rlap_workspace_query : the same for THIS handle (its growth factors rise after an RLAP_E_*_OVERFLOW retry).
 *   rlap_set_workspace   : d_ws (256-byte aligned, ws_bytes) and d_rng (rng_entries doubles) are used by all later calls; the
 *                          library then allocates and frees nothing.  NULL gives a buffer back to the handle.  The table is
 *                          generated into d_rng by the first call that needs it (and again whenever d_rng changes).
 *   A call that does not fit returns RLAP_E_WORKSPACE without touching the device; rlap_workspace_needed reports what it
 *   wanted (the caller allocates that much and repeats the call). */
int rlap_workspace_bytes(int64_t E, int64_t n_total, int64_t G, int symmetrize, size_t* ws_bytes, int64_t* rng_entries);
int rlap_workspace_query(rlap_handle h, int64_t E, int64_t n_total, int64_t G, int symmetrize, size_t* ws_bytes, int64_t* rng_entries);
int rlap_set_workspace(rlap_handle h, void* d_ws, size_t ws_bytes, double* d_rng, int64_t rng_entries);
int rlap_workspace_needed(rlap_handle h, size_t* ws_bytes, int64_t* rng_entries);
int rlap_set_timing(rlap_handle h, int enable);       /* fill rlap_stats.ms_* with HIP events */
const char* rlap_status_string(int status);

/* identity: (rows, cols) f64 row-major -> column-major staging -> row-major.
 * Replaces extension_cpp::identity (py_api_binder.cc:71-76, tensorToEigen :10-31,
 * eigenToTensor :33-51).  d_tmp needs rows*cols doubles. */
int rlap_identity(rlap_handle h, const double* d_in, double* d_tmp, double* d_out, int64_t rows, int64_t cols);

/* Split the reference's packed edge_info (E,3) f64 row-major [row, col, w]
 * (rlap/ops.py:47) into the COO arrays the calls below take. */
int rlap_unpack_edge_info(rlap_handle h, const double* d_edge_info, int64_t E, int64_t* d_row, int64_t* d_col, double* d_w);

/* Exchange format of sc_edge_info for the batched multi-GPU mode (no reference counterpart: the
 * reference has no multi-process path): row [row, col, w] <-> two 64-bit words (row << 32 | col, bits of w), so
 * that the RCCL all-gather moves 16 instead of 24 bytes per row.  d_packed holds 2 * rows 64-bit words.  Stream-ordered. */
int rlap_pack_rows(rlap_handle h, const double* d_sc, int64_t rows, void* d_packed);
int rlap_unpack_rows(rlap_handle h, const void* d_packed, int64_t rows, double* d_sc);

/* The op.  Replaces ApproximateCholesky::setup + getSchurComplement
 * (factorizers.cc:46-73) for one graph:
 *   d_row/d_col/d_w : COO, E directed entries (d_w NULL = all ones; w==0 rows are
 *                     dropped, duplicates summed: reader.cc:42-61)
 *   n, t            : num_nodes, num_remove
 *   d_perm          : o_v=random only: the node_id vector (a permutation of
 *                     0..n-1) popped from the BACK (preconditioner.cc:588-613);
 *                     the reference draws it from std::random_device.  Checked on
 *                     the device: anything but a permutation gives RLAP_E_BAD_ARG.
 *                     NULL = drawn on the device from shuffle_seed (keyed shuffle)
 *   shuffle_seed    : o_n=random / o_v=coarsen only: seed of the keyed neighbour
 *                     order that stands in for std::shuffle(random_device)
 *   d_out           : (out_cap_rows,3) f64 row-major [row, col, w]
 *   h_out_rows      : rows written (or needed, with RLAP_E_OUT_OVERFLOW)      */
int rlap_approx_chol(rlap_handle h, const int64_t* d_row, const int64_t* d_col, const double* d_w, int64_t E,
                     int64_t n, int64_t t, int o_v, int o_n, const int64_t* d_perm, uint64_t shuffle_seed,
                     double* d_out, int64_t out_cap_rows, int64_t* h_out_rows, rlap_stats* h_stats);

/* Batched graphs (a disjoint union): graph g owns the node ids
 * [h_node_ptr[g], h_node_ptr[g+1]); no edge may cross graphs.  Each graph is
 * eliminated independently with its own num_remove and its own restart of the
 * sampling stream -- what G separate reference calls would do: the rows of graph g
 * are those of rlap_approx_chol on graph g alone (ids shifted by node_ptr[g]) with
 * its slice of d_perm and with shuffle_seed + g (the keyed neighbour order hashes
 * graph-local ids).  d_perm holds, for graph g, a permutation of LOCAL ids
 * 0..n_g-1 at [node_ptr[g], node_ptr[g+1]).
 * Rows come out grouped by graph, with global node ids; h_out_row_ptr[G+1]. */
int rlap_approx_chol_batched(rlap_handle h, const int64_t* d_row, const int64_t* d_col, const double* d_w, int64_t E,
                             int64_t G, const int64_t* h_node_ptr, const int64_t* h_num_remove, int o_v, int o_n,
                             const int64_t* d_perm, uint64_t shuffle_seed, double* d_out, int64_t out_cap_rows,
                             int64_t* h_out_row_ptr, rlap_stats* h_stats);

/* The op with the step BEFORE the path fused in (scripts/node_shared.py:326-327,
 * scripts/augmentor_benchmarks.py:77-78):
 *   symmetrize != 0 : the input holds every undirected edge in one or both directions; (b,a) is
 *                     added for every (a,b) and duplicates are folded inside the COO->CSR kernels
 *                     (PyG to_undirected + coalesce: an edge SET with unit weights when d_w is
 *                     NULL, summed weights otherwise).  out_cap_rows must allow 2*E rows.
 *   n < 0           : num_nodes = max id + 1, found on the device (one 8-byte read-back)
 *   t < 0           : num_remove = (int64)(remove_frac * num_nodes)
 *   h_num_nodes     : (nullable) the num_nodes used */
int rlap_approx_chol_from_edges(rlap_handle h, const int64_t* d_src, const int64_t* d_dst, const double* d_w, int64_t E,
                                int64_t n, int64_t t, double remove_frac, int symmetrize, int o_v, int o_n,
                                const int64_t* d_perm, uint64_t shuffle_seed, double* d_out, int64_t out_cap_rows,
                                int64_t* h_out_rows, int64_t* h_num_nodes, rlap_stats* h_stats);

/* Test hook: first-attempt limits for the next calls on this handle (negative = default): append-pool factor,
 * PQ-log factor, length of the uniform table the kernels may use, entries of the output pass's long-column
 * scratch.  A call that runs into one of them repeats itself with the regular sizes (rlap_stats.n_retries). */
int rlap_debug_set_limits(rlap_handle h, double pool_factor, double log_factor, int64_t rng_len, int64_t scratch_entries);

/* Debug aid (also: environment RLAP_DEBUG_POISON=<byte> at rlap_create): before every attempt of the next calls the whole
 * workspace arena, the caller's output buffer and the elimination kernel's LDS are filled with `byte` (0..255; negative = off),
 * so that a kernel reading something it was never given reads that byte and not the previous call's (or the previous
 * workgroup's) data -- a stale read changes the result deterministically instead of once in a thousand runs. */
int rlap_debug_set_poison(rlap_handle h, int byte);

/* Debug aid (also: environment RLAP_DEBUG_JITTER=<n> at rlap_create): behind every workgroup barrier of the elimination kernel
 * a changing subset of the waves sleeps for n x 0.25 us (0 = off, at most 64).  A value that one wave reads behind a barrier while
 * another wave already rewrites it -- a race that needs a wave to fall a microsecond behind and otherwise shows once in some
 * thousand calls -- then shows in every call.  Results must not depend on it. */
int rlap_debug_set_jitter(rlap_handle h, int quarter_us);

/* First `count` uniforms of the sampling stream (default-seeded std::mt19937_64
 * through uniform_real_distribution<double>(0,1), preconditioner.cc:356-357)
 * as generated on the device; for known-answer tests. */
int rlap_rng_uniforms(rlap_handle h, int64_t count, double* d_out);

/* Test hook: the wave-parallel std::sort emulation used by the kernels, on `narr` arrays of doubles
 * (array a = d_keys[d_offs[a] .. d_offs[a+1]), each at most 512 long).  d_perm_out[d_offs[a]+i] = index of the
 * element that ends at position i; compared with libstdc++'s std::sort by tests/test_gpu_parity.py.
 * desc bit 0 = descending; bit 1 = arrays of at most 64 elements use the register-resident variant of the batch kernel;
 * bit 2 = the half-wave variant (arrays of at most 32 elements, two per wave; longer ones are left untouched);
 * bit 5 = the long-column sort of the dataflow elimination (rlap_flow.hip; any length up to 65000), bit 6 with it = records in global memory, bit 7 with it = 16-bit indices sorted in LDS (the form columns beyond 3,400 entries take);
 * bit 8 with it = the sort's duration in 10 ns ticks instead of the first index (timing tool); bit 9 with it = the keys are distinct
 * non-negative integers: the radix form the elimination uses where keys cannot repeat (ids of a column without multi-edges, tags). */
int rlap_debug_wave_sort(rlap_handle h, const double* d_keys, const int32_t* d_offs, int32_t narr, int32_t desc, int32_t* d_perm_out);

/* Host-side synthetic input (bench/tests): Barabasi-Albert graph as a symmetric,
 * coalesced COO sorted by (col,row).  Returns the directed entry count; call with
 * NULL arrays to size them (upper bound 2*m*(n-m)). */
int64_t rlap_util_ba_graph(int64_t n, int64_t m, uint64_t seed, int64_t* h_row, int64_t* h_col);

#ifdef __cplusplus
}
#endif
#endif
