/* A caller of include/rlap_hip.h written in C: no torch, no C++ -- device memory from hipMalloc, the workspace contract
 * (rlap_workspace_query -> caller allocates -> rlap_set_workspace, with the RLAP_E_WORKSPACE retry loop), one call of the op,
 * the rows written to a file as raw doubles.  What a host in another language binds is exactly this sequence
 * (INTEGRATION.md section 2/3).  Built by __graft_entry__.build(); tests/test_gpu_round3.py runs it and compares the file with
 * the oracle.
 *   cabi_caller N M SEED NUM_REMOVE O_V O_N OUT_FILE        (O_V: 0 random 1 degree 2 coarsen; O_N: 0 asc 1 desc 2 random) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "rlap_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_RLAP(x) do { int r_ = (x); if (r_ != RLAP_OK) { fprintf(stderr, "%s: %s\n", #x, rlap_status_string(r_)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: %s N M SEED NUM_REMOVE O_V O_N OUT_FILE\n", argv[0]); return 1; }
    const int64_t n = atoll(argv[1]), m = atoll(argv[2]), t = atoll(argv[4]);
    const uint64_t seed = strtoull(argv[3], NULL, 10);
    const int o_v = atoi(argv[5]), o_n = atoi(argv[6]);

    /* synthetic input on the host (the library's own generator, so that the test can build the same graph) */
    const int64_t cap = rlap_util_ba_graph(n, m, seed, NULL, NULL);
    int64_t* h_row = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t* h_col = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
    const int64_t E = rlap_util_ba_graph(n, m, seed, h_row, h_col);

    int64_t *d_row = NULL, *d_col = NULL;
    double* d_out = NULL;
    CHECK_HIP(hipMalloc((void**)&d_row, sizeof(int64_t) * (size_t)E));
    CHECK_HIP(hipMalloc((void**)&d_col, sizeof(int64_t) * (size_t)E));
    CHECK_HIP(hipMalloc((void**)&d_out, sizeof(double) * 3 * (size_t)E));
    CHECK_HIP(hipMemcpy(d_row, h_row, sizeof(int64_t) * (size_t)E, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_col, h_col, sizeof(int64_t) * (size_t)E, hipMemcpyHostToDevice));

    rlap_handle h = NULL;
    CHECK_RLAP(rlap_create(&h));
    size_t ws_bytes = 0;
    int64_t rng_entries = 0;
    CHECK_RLAP(rlap_workspace_query(h, E, n, 1, 0, &ws_bytes, &rng_entries));
    void* d_ws = NULL;
    double* d_rng = NULL;
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMalloc((void**)&d_rng, sizeof(double) * (size_t)rng_entries));
    CHECK_RLAP(rlap_set_workspace(h, d_ws, ws_bytes, d_rng, rng_entries));

    /* o_v = random: d_perm NULL = the node_id vector is drawn on the device from the seed */
    int64_t rows = 0;
    rlap_stats st;
    memset(&st, 0, sizeof(st));
    int rc, grows = 0;
    while ((rc = rlap_approx_chol(h, d_row, d_col, NULL, E, n, t, o_v, o_n, NULL, seed, d_out, E, &rows, &st)) == RLAP_E_WORKSPACE) {
        /* a growth limit met inside the call: ask, allocate, hand over, call again (nothing was computed) */
        CHECK_RLAP(rlap_workspace_needed(h, &ws_bytes, &rng_entries));
        CHECK_HIP(hipFree(d_ws));
        CHECK_HIP(hipFree(d_rng));
        CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
        CHECK_HIP(hipMalloc((void**)&d_rng, sizeof(double) * (size_t)rng_entries));
        CHECK_RLAP(rlap_set_workspace(h, d_ws, ws_bytes, d_rng, rng_entries));
        if (++grows > 8) { fprintf(stderr, "workspace keeps growing\n"); return 4; }
    }
    CHECK_RLAP(rc);

    double* h_out = (double*)malloc(sizeof(double) * 3 * (size_t)(rows > 0 ? rows : 1));
    CHECK_HIP(hipMemcpy(h_out, d_out, sizeof(double) * 3 * (size_t)rows, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 5; }
    fwrite(h_out, sizeof(double), 3 * (size_t)rows, f);
    fclose(f);
    printf("rows=%lld eliminated=%lld draws=%lld workspace=%zu bytes, %lld uniforms, regrown %d times\n", (long long)rows,
           (long long)st.n_eliminated, (long long)st.n_draws, ws_bytes, (long long)rng_entries, grows);

    CHECK_RLAP(rlap_destroy(h));
    CHECK_HIP(hipFree(d_ws)); CHECK_HIP(hipFree(d_rng)); CHECK_HIP(hipFree(d_row)); CHECK_HIP(hipFree(d_col)); CHECK_HIP(hipFree(d_out));
    free(h_row); free(h_col); free(h_out);
    return 0;
}
