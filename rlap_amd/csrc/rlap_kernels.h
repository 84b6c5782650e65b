// rlap_kernels.h -- kernel declarations shared by rlap_kernels.hip and rlap_api.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rlap_core.h"
#include "rlap_flow.h"

namespace rlap {

enum { FLAG_RANGE = 0, FLAG_CROSS = 1, FLAG_PERM = 2, FLAG_SCR = 3, FLAG_UNSORTED_CR = 4, FLAG_UNSORTED_RC = 5, FLAG_COUNT = 8 };

// several (pointer, count, value) fills done by one launch
struct FillJobs { int32_t* ptr[16]; int64_t count[16]; int32_t value[16]; int n; };

// what the host reads back, once, at the end of a call
struct CallResults {
    int32_t flags[FLAG_COUNT];
    double acc[4];
    int32_t nnz, status;
    int64_t n_draws, rounds, singles;
    int64_t live_total, scr_need, ext_total, m_total;
    int32_t pool_used, log_used;
};
constexpr int ECAP = 384;  // elimination: column extent handled in LDS; longer -> sequential form in global scratch
constexpr int ECAP_SMALL = 320;   // the same for the 256-thread shape (four workgroups per CU: 40 KB of LDS each)
constexpr int EQTAB_OFF32 = 2 * 48 * 64;                 // equal-key permutation tables: 64-slot candidates first (perm, inverse), then 32-slot
constexpr int EQTAB_OFF128 = EQTAB_OFF32 + 2 * 16 * 32;     // ... then 128-slot
constexpr int EQTAB_BYTES = EQTAB_OFF128 + 2 * 112 * 128;
constexpr int SCAP = 512;  // output pass: same
constexpr int BIGCAP = 7168;  // output pass, long columns: one workgroup, 125 KB LDS record array + 31 KB LDS stop lists of the sort
constexpr int BIG_LDS_BYTES = BIGCAP * 16 + 2 * (BIGCAP + 2) * 2;
constexpr int MIDCAP = 2048;  // ... columns up to this many slots take a third of that LDS (three workgroups per CU)
constexpr int MID_LDS_BYTES = MIDCAP * 16 + 2 * (MIDCAP + 2) * 2;
// multi-wave variant (k_sc_merge_mw): records + stop lists + two segment queues of the level-by-level introsort
constexpr int MW_BIG_QCAP = BIGCAP / 17 + 8, MW_MID_QCAP = MIDCAP / 17 + 8;
constexpr int MW_BIG_LDS_BYTES = BIGCAP * 16 + 2 * (BIGCAP + 2) * 2 + 8 + 2 * 3 * MW_BIG_QCAP * 4;
constexpr int MW_MID_LDS_BYTES = MIDCAP * 16 + 2 * (MIDCAP + 2) * 2 + 8 + 2 * 3 * MW_MID_QCAP * 4;
constexpr int MID1CAP = 1024;  // ... and up to this many a sixth (seven per CU)
constexpr int MID1_LDS_BYTES = MID1CAP * 16 + 2 * (MID1CAP + 2) * 2;
constexpr int POOL_GRAB_BIG = 16384;   // append slots a 1024-thread workgroup reserves at a time (256-thread: POOL_GRAB_SMALL)
constexpr int POOL_GRAB_SMALL = 2048;
constexpr int LIVE_SLOTS = 64;   // the output pass sums its live-entry count into this many counters, LIVE_STRIDE words apart
constexpr int LIVE_STRIDE = 16;
constexpr int HUGECAP = 65535; // output pass, longer still: records in global scratch (uint16 stop lists)

// Global scratch for columns too long for LDS (sequential fallback).
struct ElimScratch {
    SRec* rec;
    int32_t* i32;   // 10 arrays of `cap` ints, interleaved per array
    double* f64;    // 4 arrays of `cap` doubles
    int64_t cap;    // total entries over all graphs
    const uint8_t* eqtab;   // k_eq_tables (per handle)
    long long* prof; // optional diagnostic build only: per-phase cycle sums of graph 0 (nullptr in production)
    int32_t poison;  // debug (RLAP_DEBUG_POISON): >= 0 -> the workgroup's LDS is filled with this byte before anything else
    int32_t jitter;  // debug (RLAP_DEBUG_JITTER): > 0 -> waves sleep behind the elimination kernel's barriers (schedule perturbation)
    __host__ __device__ ColBuf colbuf(int64_t base) const {
        ColBuf B;
        B.rec = rec + base;
        B.a_slot = i32 + base; B.a_nbr = i32 + cap + base; B.a_twin = i32 + 2 * cap + base;
        B.b_slot = i32 + 3 * cap + base; B.b_nbr = i32 + 4 * cap + base; B.b_twin = i32 + 5 * cap + base;
        B.ksel = i32 + 6 * cap + base; B.t_key = i32 + 7 * cap + base; B.t_mv = i32 + 8 * cap + base; B.t_of = i32 + 9 * cap + base;
        B.a_val = f64 + base; B.b_val = f64 + cap + base; B.cum = f64 + 2 * cap + base; B.newv = f64 + 3 * cap + base;
        return B;
    }
};

struct ScScratch {
    SRec* rec;
    int32_t* i32;   // 7 arrays (a_slot,a_nbr,a_twin,b_slot,b_nbr,b_twin,t_of)
    double* f64;    // 2 arrays (a_val,b_val)
    int64_t cap;
    unsigned long long* top;
    int32_t* flags;   // FLAG_SCR is raised when `cap` entries do not suffice
    __host__ __device__ ColBuf colbuf(int64_t base) const {
        ColBuf B;
        B.rec = rec + base;
        B.a_slot = i32 + base; B.a_nbr = i32 + cap + base; B.a_twin = i32 + 2 * cap + base;
        B.b_slot = i32 + 3 * cap + base; B.b_nbr = i32 + 4 * cap + base; B.b_twin = i32 + 5 * cap + base;
        B.t_of = i32 + 6 * cap + base;
        B.ksel = nullptr; B.t_key = nullptr; B.t_mv = nullptr;
        B.a_val = f64 + base; B.b_val = f64 + cap + base; B.cum = nullptr; B.newv = nullptr;
        return B;
    }
};

__global__ void k_mt19937_64_table(double* out, int64_t count);
__global__ void k_transpose_copy(const double* in, double* out, int64_t rows, int64_t cols, int to_colmajor);
__global__ void k_pack_rows(const double* sc, int64_t m, unsigned long long* packed);
__global__ void k_unpack_rows(const unsigned long long* packed, int64_t m, double* sc);
__global__ void k_unpack_edge_info(const double* ei, int64_t E, int64_t* row, int64_t* col, double* w);
__global__ void k_vertex_graph(const int64_t* node_ptr, int G, int32_t* vgraph, int64_t N);
__global__ void k_edge_keys(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t N, const int32_t* vgraph,
                            int symmetrize, int kbits, uint64_t* keys, uint32_t* idx, int32_t* flags);
__global__ void k_max_id(const int64_t* row, const int64_t* col, int64_t E, unsigned long long* out);
__global__ void k_fill_multi(FillJobs J);
__global__ void k_perm_check(const int64_t* perm, const int32_t* vgraph, const GraphDesc* gd, int32_t N, int32_t* seen, int32_t* flags);
__global__ void k_gd_scratch(const int32_t* colptr, const int64_t* node_ptr, int32_t G, GraphDesc* gd);
__global__ void k_collect(const int32_t* flags, const double* acc, const int32_t* nnz_p, const unsigned long long* counters,
                          const unsigned long long* live, const int64_t* tmp_off, const int64_t* row_off, int32_t S, const GraphDesc* gd, int32_t G,
                          const int32_t* pool_top, const int32_t* bs_pool_top, CallResults* out);
__global__ void k_heads(const uint64_t* keys, int64_t E, int32_t* head);
__global__ void k_fill_csr(const uint64_t* keys, const uint32_t* idx, const int32_t* head, const int32_t* pos, const double* w,
                           int64_t E, int set_semantics, int kbits, Slot* ent, int32_t* slot_col, int32_t* nbr32);
__global__ void k_twin_sorted(const Slot* ent, const uint32_t* T, const int32_t* nnz_p, double* acc);
__global__ void k_twin_store(Slot* ent, const uint32_t* T, const int32_t* nnz_p);
__global__ void k_colptr(const int32_t* slot_col, const int32_t* nnz_p, int32_t N, int32_t* colptr, int32_t* nnz_out, int32_t* pool_top_out);
__global__ void k_pq_init(const int32_t* colptr, const int32_t* vgraph, int32_t N, VRec* vr, uint64_t* skey, uint32_t* sval);
__global__ void k_bucket_bounds(const uint32_t* order, const VRec* vr, const int32_t* vgraph, const GraphDesc* gd, int32_t N,
                                int32_t* ocur, int32_t* oend, int32_t* origpos);
void launch_eq_tables(hipStream_t stream, uint8_t* out);
void launch_eliminate_batch(int o_v, int o_n, unsigned G, int n_cu, hipStream_t stream, const Arrays& A, GraphDesc* gd, const ElimScratch& S,
                            int32_t* batch_pos, const int32_t* flags, const double* acc, bool wide);
// ---- dataflow elimination for o_v = random (rlap_flow.hip) ----
constexpr int FLOW_EC = 896;              // column extent a wave handles in its own LDS (two one-wave workgroups per CU)
constexpr int FLOW_EC_SMALL = 192;        // ... in the shape for batches of small graphs (eight per CU)
constexpr int FLOW_EC_MID = 448;          // ... four per CU
constexpr int FLOW_NW = 2;                // waves per workgroup in the large shape: wave 0 owns the position, the others help with the sorts of its long columns
constexpr int FLOW_SCR_BYTES = 104;       // working storage per entry of a longer column (bump-allocated, rlap_flow.hip)
struct FlowParams {
    const int32_t* vgraph;
    GraphDesc* gd;
    const int32_t* in_flags;
    const double* in_acc;
    char* scr;               // long columns: bump-allocated working storage (FLOW_SCR_BYTES per entry)
    int32_t scr_entries;
    int32_t spin_limit;      // milliseconds without a committed position anywhere after which a waiting wave gives up (ST_INTERNAL)
    int32_t jitter;          // debug: waves sleep at the phase boundaries (schedule perturbation)
    int32_t poison;          // debug: LDS starts as this byte
    long long* prof;         // diagnostic (RLAP_PHASE_PROFILE=1): per-phase sums of the 100 MHz clock over all waves, nullptr in production
    long long* trace;        // diagnostic (RLAP_FLOW_TRACE=<file>): six words per position -- 100 MHz clock at the claim, at pend == 0, at the count's publication,
                             // after the look-back, at the end; live entries -- nullptr in production
};
void launch_flow_setup(hipStream_t s, const Arrays& A, const FlowArrays& F, const FlowParams& P, const int32_t* slot_col, const int32_t* nnz_p, int32_t N, int64_t Eeff,
                       uint64_t* okey, uint32_t* oval, int gbits);
void launch_flow_eliminate(int shape, unsigned grid, hipStream_t stream, const Arrays& A, const FlowArrays& F, const FlowParams& P);
void launch_debug_flow_sort(hipStream_t s, const double* keys, const int32_t* offs, int32_t narr, int32_t desc, int32_t* perm_out, char* scr);
void launch_flow_finish(hipStream_t s, const Arrays& A, const FlowArrays& F, GraphDesc* gd, int32_t N, int32_t G);
void launch_flow_ro_count(hipStream_t s, const Arrays& A, const FlowArrays& F, const uint32_t* order, int32_t S, int32_t* cnt);
void launch_flow_ro_emit(hipStream_t s, const Arrays& A, const FlowArrays& F, const uint32_t* order, int32_t S, const int32_t* off, uint64_t* keys, uint32_t* vals,
                         uint32_t* home, int32_t cap, int32_t* status);
void launch_flow_ro_permute(hipStream_t s, const Arrays& A, const FlowArrays& F, const int32_t* off, int32_t S, const uint32_t* sorted_slot, const uint64_t* sorted_keys,
                            const uint32_t* home, int32_t cap, Slot* tmp);
__global__ void k_sc_keys(const VRec* vr, const int32_t* origpos, const int32_t* vgraph, const GraphDesc* gd,
                          int32_t N, uint64_t* skey, uint32_t* sval);
__global__ void k_sc_perm_order(const int64_t* perm, const int32_t* vgraph, const GraphDesc* gd, const int64_t* surv_base, int32_t N,
                                uint32_t* order);
__global__ void k_sc_ext(const uint32_t* order, const int32_t* colptr, const VRec* vr, int32_t S, const int32_t* in_flags, const double* in_acc, int32_t* ext);
constexpr unsigned NHUGE = 32;   // workgroups of k_sc_merge_huge
struct ScLaunch { hipStream_t main; hipStream_t side[2]; hipEvent_t ev[3]; };   // side streams may be null: everything on `main`
template <int NW>
__global__ void k_sc_merge_mw(Arrays A, const GraphDesc* gd, const int32_t* vgraph, const uint32_t* order, const int32_t* ext, const int64_t* tmp_off,
                              const int32_t* list, const int32_t* count, int32_t* tmp_nbr, double* tmp_val, int32_t* cnt_out,
                              unsigned long long* live_total, int32_t lcap, int32_t qcap);
void launch_sc_merge(const ScLaunch& X, const Arrays& A, const GraphDesc* gd, const int32_t* vgraph, const uint32_t* order, const int32_t* ext,
                     const int64_t* tmp_off, int32_t S, int32_t* tmp_nbr, double* tmp_val, int32_t* cnt_out, const ScScratch& SS,
                     unsigned long long* live_total, int32_t* lists, int32_t* counts, uint16_t* hugelists);
__global__ void k_debug_wave_sort(const double* keys, const int32_t* offs, int32_t narr, int32_t desc, int32_t* perm_out);
__global__ void k_sc_merge_big(Arrays A, const GraphDesc* gd, const int32_t* vgraph, const uint32_t* order, const int32_t* ext, const int64_t* tmp_off, const int32_t* list, const int32_t* count,
                               int32_t* tmp_nbr, double* tmp_val, int32_t* cnt_out, unsigned long long* live_total, uint16_t* lists, int32_t lcap);
__global__ void k_sc_merge_huge(Arrays A, const GraphDesc* gd, const int32_t* vgraph, const uint32_t* order, const int32_t* ext, const int64_t* tmp_off, const int32_t* list, const int32_t* count,
                                int32_t* tmp_nbr, double* tmp_val, int32_t* cnt_out, unsigned long long* live_total, uint16_t* lists, SRec* scratch,
                                unsigned long long* scratch_top, int64_t scratch_cap, int32_t* flags);
__global__ void k_sc_compact(const uint32_t* order, const int32_t* cnt, const int64_t* row_off, const int64_t* tmp_off,
                             const int32_t* tmp_nbr, const double* tmp_val, int32_t S, double* out, int64_t out_cap);
__global__ void k_graph_rows(const int64_t* surv_base, const int64_t* row_off, int32_t G, int64_t* out_ptr);

}  // namespace rlap
