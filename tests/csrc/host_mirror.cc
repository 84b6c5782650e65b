// host_mirror.cc -- runs the SAME data-structure code the HIP kernels use
// (rlap_amd/csrc/rlap_core.h: chunked columns, bucket-stack PQ, std::sort
// emulation, sequential elimination/output, batch candidates) on the CPU, so the
// design can be checked against the oracle without a GPU.  Test-only.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../rlap_amd/csrc/rlap_core.h"
#include "../../rlap_amd/csrc/rlap_flow.h"

using namespace rlap;

namespace {

struct Setup {
    int64_t n, t;
    int32_t nnz, cap;
    std::vector<int32_t> colptr, ocur, oend, orig, bs_cnt, bs_alloc, bs_dir, bs_v, bs_id;
    std::vector<Slot> ent;
    std::vector<VRec> vrec;
    std::vector<double> rng;
    std::vector<int64_t> perm_l;
    int32_t pool_top, bs_pool_top;
    Arrays A;
    GraphDesc G;
    std::vector<SRec> rec;
    std::vector<int32_t> i32;
    std::vector<double> f64;
    ColBuf B;

    void build(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n_, int64_t t_, int o_v, int o_n,
               const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots) {
        n = n_; t = t_;
        colptr.assign(n + 1, 0);
        for (int64_t p = 0; p < E; ++p) if (w[p] != 0) colptr[col[p] + 1]++;
        for (int64_t c = 0; c < n; ++c) colptr[c + 1] += colptr[c];
        nnz = colptr[n];
        int32_t slot_cap = nnz + pool_slots;
        ent.assign(slot_cap, Slot{0.0, 0, -1});
        {
            std::vector<std::pair<int64_t, int64_t>> keyed;
            keyed.reserve(E);
            for (int64_t p = 0; p < E; ++p) if (w[p] != 0) keyed.push_back({col[p] * n + row[p], p});
            std::sort(keyed.begin(), keyed.end());
            for (int32_t s = 0; s < nnz; ++s) { ent[s].nbr = (int32_t)row[keyed[s].second]; ent[s].val = w[keyed[s].second]; }
            std::vector<int32_t> cur(colptr.begin(), colptr.end() - 1);
            for (int64_t c = 0; c < n; ++c)
                for (int32_t s = colptr[c]; s < colptr[c + 1]; ++s) ent[s].twin = cur[ent[s].nbr]++;
        }
        vrec.assign(n, VRec{0, -1, 0, -1});
        std::vector<int32_t> key((size_t)n, 0);
        ocur.assign(2 * n + 1, 0); oend.assign(2 * n + 1, 0); orig.assign(n, 0);
        bs_cnt.assign(2 * n + 1, 0); bs_alloc.assign(2 * n + 1, 0); bs_dir.assign((size_t)(2 * n + 1) * BDIR, -1);
        int32_t bs_pool_cap = 4 * (int32_t)E + BCH0 * (int32_t)(2 * n + 1) + 64;
        bs_pool_top = 0;
        bs_v.assign(bs_pool_cap, 0); bs_id.assign(bs_pool_cap, 0);
        for (int64_t v = 0; v < n; ++v) { key[v] = colptr[v + 1] - colptr[v]; vrec[v].key = key[v]; orig[v] = (int32_t)v; }
        std::sort(orig.begin(), orig.end(), [&](int32_t a, int32_t b) { return key[a] != key[b] ? key[a] < key[b] : a > b; });
        for (int32_t i = 0; i < n;) {
            int32_t j = i;
            while (j < n && key[orig[j]] == key[orig[i]]) ++j;
            ocur[key[orig[i]]] = i; oend[key[orig[i]]] = j;
            i = j;
        }
        int64_t rng_len = std::max<int64_t>((int64_t)nnz * 4 + 1024, (int64_t)pool_slots);   // (a draw per appended entry at most)
        rng.resize((size_t)rng_len);
        {
            std::mt19937_64 g;
            for (auto& u : rng) {
                uint64_t raw = g();
                double d = (double)raw * 5.42101086242752217e-20;  // 2^-64
                if (d >= 1.0) d = 0.99999999999999989;
                u = d;
            }
        }
        perm_l.assign(n, 0);
        if (perm) for (int64_t i = 0; i < n; ++i) perm_l[i] = perm[i];
        pool_top = nnz;
        A.colptr = colptr.data(); A.e = ent.data();
        A.slot_cap = slot_cap; A.pool_top = &pool_top; A.vr = vrec.data();
        A.ocur = ocur.data(); A.oend = oend.data();
        A.orig_order = orig.data();
        A.bs_cnt = bs_cnt.data(); A.bs_alloc = bs_alloc.data(); A.bs_dir = bs_dir.data(); A.bs_v = bs_v.data(); A.bs_id = bs_id.data();
        A.bs_pool_top = &bs_pool_top; A.bs_pool_cap = bs_pool_cap;
        A.rng = rng.data(); A.rng_len = rng_len; A.perm = perm_l.data(); A.o_v = o_v; A.o_n = o_n; A.shuffle_seed = shuffle_seed;
        A.rng_mode = RNG_STREAM; A.vbase = 0;
        std::memset(&G, 0, sizeof(G));
        G.vbase = 0; G.n = (int32_t)n; G.t = t; G.bucket_base = 0;
        cap = nnz + 8;
        rec.resize(cap); i32.resize(11 * (size_t)cap); f64.resize(4 * (size_t)cap);
        B.rec = rec.data();
        int32_t* ip = i32.data();
        B.a_slot = ip; B.a_nbr = ip + cap; B.a_twin = ip + 2 * cap; B.b_slot = ip + 3 * cap; B.b_nbr = ip + 4 * cap; B.b_twin = ip + 5 * cap;
        B.ksel = ip + 6 * cap; B.t_key = ip + 7 * cap; B.t_mv = ip + 8 * cap; B.t_of = ip + 9 * cap;
        B.a_val = f64.data(); B.b_val = f64.data() + cap; B.cum = f64.data() + 2 * cap; B.newv = f64.data() + 3 * cap;
    }

    // output phase (shared by both drivers)
    int finish(int64_t nelim, int64_t npop, int64_t* order_out, double** out, int64_t* out_rows) {
        std::vector<double> res;
        if (std::getenv("RLAP_MIRROR_EXT_HIST")) {   // diagnostic: slots per surviving column (what the output pass's tiers are sized by) and live entries
            std::vector<char> gone((size_t)n, 0);
            if (order_out) for (int64_t q = 0; q < npop; ++q) gone[order_out[q]] = 1;
            for (int64_t v = 0; v < n; ++v) if (!gone[v]) {
                int32_t ext = (A.colptr[v + 1] - A.colptr[v]) + A.vr[v].app_cnt;
                if (ext > 512) std::fprintf(stderr, "ext %d live %d\n", ext, serial_gather(A, (int32_t)v, B, cap));
            }
        }
        for (int64_t q = nelim; q < n; ++q) {
            int32_t v = (A.o_v == OV_RANDOM) ? (int32_t)perm_l[n - 1 - q] : pq_pop(A, G);
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
};

}  // namespace

// Batch driver: the round structure of the frontier kernel, executed with plain
// loops.  Every step uses only what the parallel kernel has at that point
// (pre-round global state + per-candidate records), so a logic error in the
// batching rules shows up here as a mismatch with the oracle.
// stats_out[0] = rounds, [1] = single-vertex fallbacks, [2] = contended targets.
template <int BC>
static int mirror_batch_impl(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t,
                             int o_v, int o_n, const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots, int32_t Bsz,
                             double** out, int64_t* out_rows, int64_t* order_out, int64_t* stats_out) {
    Setup S;
    S.build(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots);
    const Arrays& A = S.A;
    GraphDesc& G = S.G;
    const bool use_pq = o_v != OV_RANDOM;
    int64_t nelim = std::min<int64_t>(t, n - 1);
    if (nelim < 0) nelim = 0;
    int64_t done = 0, npop = 0, rounds = 0, singles = 0, contended_total = 0;
    typedef CandT<BC> Cand;
    std::vector<Cand> cand((size_t)Bsz);
    int64_t why[6] = {0, 0, 0, 0, 0, 0};
    int64_t adjcase[4] = {0, 0, 0, 0};
    int64_t npatch_try = 0, npatch_ok = 0, npatch_dead = 0;
    const int64_t patch_limit = std::getenv("RLAP_MIRROR_PATCH_LIMIT") ? std::atoll(std::getenv("RLAP_MIRROR_PATCH_LIMIT")) : -1;   // diagnostic: only the first N patches
    const bool patching = (A.o_v != OV_COARSEN) && (std::getenv("RLAP_MIRROR_NO_PATCH") == nullptr);   // first dependent candidate of a round: sampled by an earlier one (moves away), last neighbour (Dec), patchable, patch makes a multi-edge
    int64_t single_len = 0, single_gt384 = 0, single_gt384_len = 0, single_max = 0;  // adjacent, big, dup, complex, pre-empted, full
    std::vector<int32_t> batch_pos((size_t)n, -1), tcount((size_t)n, 0);
    struct CRec { int32_t x, i, j; };
    struct Move { uint64_t key; int32_t v; };
    while (done < nelim) {
        ++rounds;
        int32_t Bcur = (int32_t)std::min<int64_t>(Bsz, nelim - done);
        // ---- P0: predict the next pops ----
        int32_t nc = 0;
        int32_t b = 0;
        if (use_pq) {
            while (true) {
                b = G.bucket_base + G.minlist;
                int32_t cnt = A.bs_cnt[b];
                for (int32_t a = cnt - 1; a >= 0 && nc < Bcur; --a) {
                    int32_t s = bs_slot(A, b, a);
                    int32_t v = A.bs_v[s];
                    if (A.vr[v].pqpos == A.bs_id[s]) { cand[nc].v = v; cand[nc].src = a; ++nc; }
                }
                if (nc < Bcur) {
                    for (int32_t oc = A.ocur[b]; oc < A.oend[b] && nc < Bcur; ++oc) {
                        int32_t v = A.orig_order[oc];
                        if (A.vr[v].pqpos == -1) { cand[nc].v = v; cand[nc].src = ~oc; ++nc; }
                    }
                }
                if (nc > 0) break;
                A.bs_cnt[b] = 0; A.ocur[b] = A.oend[b];
                G.minlist += 1;
                if (G.minlist > 2 * G.n) return ST_INTERNAL;
            }
        } else {
            for (; nc < Bcur; ++nc) { cand[nc].v = (int32_t)S.perm_l[n - (done + nc + 1)]; cand[nc].src = 0; }
        }
        // ---- P1: prepare (thread per candidate) ----
        for (int32_t i = 0; i < nc; ++i) { int32_t v = cand[i].v, src = cand[i].src; cand_prepare(A, v, cand[i]); cand[i].src = src; batch_pos[v] = i; }
        // ---- P1b: dependence; Pmax ----
        // A candidate adjacent to EARLIER candidates of the round ("dependent") no longer cuts the round by itself: its
        // record is patched from theirs once they are sampled (cand_patch).  It still cuts when it has more than DEPMAX of them.
        int32_t Pmax = nc; int Pwhy = 5;
        for (int32_t i = 0; i < nc; ++i) {
            Cand& C = cand[i];
            bool bad = (C.flags & (CF_BIG | CF_DUP)) != 0;
            C.ndep = 0;
            if (!bad) for (int32_t j = 0; j < C.m; ++j) {
                int32_t bp = batch_pos[C.e[j].nbr];
                if (bp >= 0 && bp < i) { if (!patching || C.ndep >= DEPMAX) { bad = true; break; } C.dep[C.ndep++] = (uint8_t)bp; }
            }
            if (bad && i < Pmax) { Pmax = i; Pwhy = (C.flags & CF_BIG) ? 1 : (C.flags & CF_DUP) ? 2 : 0; }
        }
        auto consume = [&](int32_t P) {   // candidates [0,P) leave the queue
            if (!use_pq) return;
            int32_t src = cand[P - 1].src;
            if (src >= 0) A.bs_cnt[b] = src; else { A.bs_cnt[b] = 0; A.ocur[b] = (~src) + 1; }
            for (int32_t i = 0; i < P; ++i) A.vr[cand[i].v].pqpos = -2;
        };
        auto cleanup = [&]() { for (int32_t i = 0; i < nc; ++i) batch_pos[cand[i].v] = -1; };
        if (Pmax == 0) {
            // candidate 0 needs the single-vertex path
            ++why[Pwhy];
            ++singles;
            consume(1);
            cleanup();
            if (order_out) order_out[npop] = cand[0].v;
            ++npop;
            { int32_t l0 = serial_gather(A, cand[0].v, S.B, S.cap); single_len += l0; if (l0 > 384) { ++single_gt384; single_gt384_len += l0; } if (l0 > single_max) single_max = l0; }
            int rc = serial_eliminate(A, G, S.B, S.cap, cand[0].v, done + 1);
            if (rc) return rc;
            done += 1;
            continue;
        }
        // ---- P2: RNG offsets ----
        int64_t off = G.n_draws;
        for (int32_t i = 0; i < Pmax; ++i) { cand[i].draw0 = off; off += cand[i].ndraw; }
        if (off > A.rng_len) return ST_RNG_OVERFLOW;
        // ---- P3: sampling; dependent candidates are patched first (their earlier neighbours are sampled by then) ----
        for (int32_t i = 0; i < Pmax; ++i) {
            if (cand[i].ndep > 0) {
                ++npatch_try;
                if ((patch_limit >= 0 && npatch_ok >= patch_limit) || !cand_patch(A, cand.data(), i, G.vbase)) {
                    // (statistics) refused because an earlier candidate's new weight for this one came out <= 0?
                    for (int32_t q = 0; q < cand[i].ndep; ++q) { const Cand& J = cand[cand[i].dep[q]]; for (int32_t z = 0; z + 1 < J.m; ++z) if (J.e[z].nbr == cand[i].v && !(J.e[z].val > 0)) ++npatch_dead; }
                    Pmax = i; Pwhy = (cand[i].flags & CF_DUP) ? 2 : 0; break;
                }
                ++npatch_ok;
            }
            cand_sample(A, cand[i]);
        }
        if (false && Pwhy == 0 && Pmax < nc && A.o_v != OV_COARSEN) {
            // what would it take to carry on past the first dependent candidate d = cand[Pmax]?
            Cand& D = cand[Pmax];
            bool any_last = false, any_target = false, any_dup = false;
            for (int32_t q = 0; q < D.m; ++q) {
                int32_t bp = batch_pos[D.e[q].nbr];
                if (bp < 0 || bp >= Pmax) continue;
                Cand& J = cand[bp];
                int32_t p = -1;
                for (int32_t z = 0; z < J.m; ++z) if (J.e[z].nbr == D.v) p = z;
                if (p < 0) continue;
                if (p == J.m - 1) any_last = true;
                int32_t c = 0;
                for (int32_t z = 0; z < J.m - 1; ++z) if (J.ksel[z] == p) ++c;
                if (c > 0) any_target = true;
                if (p < J.m - 1) { int32_t k = J.e[J.ksel[p]].nbr; for (int32_t z = 0; z < D.m; ++z) if (D.e[z].nbr == k) any_dup = true; }
            }
            ++adjcase[any_last ? 1 : any_target ? 0 : any_dup ? 3 : 2];
        }
        // ---- P4: PQ replay with contended targets in candidate order ----
        for (int32_t i = 0; i < Pmax; ++i) for (int32_t j = 0; j < cand[i].m; ++j) tcount[cand[i].e[j].nbr]++;
        std::vector<CRec> cont;
        for (int32_t i = 0; i < Pmax; ++i) {
            Cand& C = cand[i];
            bool allow_last = (done + i + 1) + 1 < n;
            for (int32_t j = 0; j < C.m; ++j) {
                int32_t x = C.e[j].nbr;
                TRes& R = ent_tres(C.e[j]);
                if (tcount[x] > 1) { R.flags = TF_CONTENDED; cont.push_back({x, i, j}); continue; }
                int mv, c; bool cx = false;
                int32_t k2 = cand_replay(A, C, j, use_pq ? A.vr[x].key : 1, use_pq ? G.n : 0x7FFFFFFF, allow_last, &mv, &c, &cx);
                if (!use_pq) { mv = -1; cx = false; }
                R.key_after = k2; R.mv = (int16_t)mv; R.c = (uint8_t)c; R.flags = 0;
                if (cx) C.flags |= CF_COMPLEX;
            }
        }
        std::sort(cont.begin(), cont.end(), [](const CRec& p, const CRec& q) { return p.x != q.x ? p.x < q.x : p.i < q.i; });
        contended_total += (int64_t)cont.size();
        for (size_t q = 0; q < cont.size();) {
            size_t r = q;
            int32_t key = use_pq ? A.vr[cont[q].x].key : 1;
            while (r < cont.size() && cont[r].x == cont[q].x) {
                Cand& C = cand[cont[r].i];
                bool allow_last = (done + cont[r].i + 1) + 1 < n;
                int mv, c; bool cx = false;
                int32_t k2 = cand_replay(A, C, cont[r].j, key, use_pq ? G.n : 0x7FFFFFFF, allow_last, &mv, &c, &cx);
                if (!use_pq) { mv = -1; cx = false; }
                TRes& R = ent_tres(C.e[cont[r].j]);
                R.key_after = k2; R.mv = (int16_t)mv; R.c = (uint8_t)c;
                if (cx) C.flags |= CF_COMPLEX;
                key = k2;
                ++r;
            }
            q = r;
        }
        // ---- P: first pre-emption / complex candidate ----
        int32_t P = Pmax;
        for (int32_t i = 0; i < Pmax; ++i) {
            Cand& C = cand[i];
            if (C.flags & CF_COMPLEX) { P = std::min(P, i); Pwhy = 3; break; }
            bool pre = false;
            if (use_pq) for (int32_t j = 0; j < C.m; ++j) { TRes& R = ent_tres(C.e[j]); if (R.mv >= 0 && pq_list_of(R.key_after, G.n) <= G.minlist) pre = true; }
            if (pre) { if (i + 1 < P) Pwhy = 4; P = std::min(P, i + 1); break; }
        }
        for (int32_t i = 0; i < Pmax; ++i) for (int32_t j = 0; j < cand[i].m; ++j) tcount[cand[i].e[j].nbr] = 0;
        ++why[Pwhy];
        if (P == 0) {
            ++singles;
            consume(1);
            cleanup();
            if (order_out) order_out[npop] = cand[0].v;
            ++npop;
            { int32_t l0 = serial_gather(A, cand[0].v, S.B, S.cap); single_len += l0; if (l0 > 384) { ++single_gt384; single_gt384_len += l0; } if (l0 > single_max) single_max = l0; }
            int rc = serial_eliminate(A, G, S.B, S.cap, cand[0].v, done + 1);
            if (rc) return rc;
            done += 1;
            continue;
        }
        // ---- P5: commit candidates [0,P) ----
        consume(P);
        // column appends: per target in candidate order (contended ones share a cursor)
        // non-contended first
        std::vector<Move> moves;
        for (int32_t i = 0; i < P; ++i) {
            Cand& C = cand[i];
            if (order_out) order_out[npop] = C.v;
            ++npop;
        }
        // process per (target, candidate) in (x, i) order for contended; any order otherwise.
        // Build list of all records of committed candidates sorted by (x, i): simple & exact.
        std::vector<CRec> all;
        for (int32_t i = 0; i < P; ++i) for (int32_t j = 0; j < cand[i].m; ++j) all.push_back({cand[i].e[j].nbr, i, j});
        std::sort(all.begin(), all.end(), [](const CRec& p, const CRec& q) { return p.x != q.x ? p.x < q.x : p.i < q.i; });
        // slots for pushes, stored back into a side table
        std::vector<std::vector<int32_t>> pslot((size_t)P);
        for (int32_t i = 0; i < P; ++i) pslot[i].assign(BC, -1);
        for (size_t q = 0; q < all.size(); ++q) {
            Cand& C = cand[all[q].i];
            int32_t x = all[q].x, j = all[q].j;
            TRes R = ent_tres(C.e[j]);
            // pushes into column x from candidate i, in position order
            if (A.o_v == OV_COARSEN) {
                if (j == C.koff) for (int32_t p = 0; p < C.m; ++p) { if (p == C.koff) continue; int32_t s = col_append(A, x); if (s < 0) return ST_POOL_OVERFLOW; pslot[all[q].i][p] = s; }
            } else {
                for (int32_t p = 0; p < C.m - 1; ++p) if (C.ksel[p] == j) { int32_t s = col_append(A, x); if (s < 0) return ST_POOL_OVERFLOW; pslot[all[q].i][p] = s; }
            }
            bool last_of_x = (q + 1 == all.size()) || all[q + 1].x != x;
            if (use_pq && last_of_x) {
                // final state of x after the committed prefix; its last move may stem from an earlier record
                int32_t mvseq = -1;
                for (size_t r = q + 1; r-- > 0 && all[r].x == x;) {
                    TRes Rr = ent_tres(cand[all[r].i].e[all[r].j]);
                    if (Rr.mv >= 0) { mvseq = (all[r].i << 8) | Rr.mv; break; }
                }
                if (R.key_after != A.vr[x].key) A.vr[x].key = R.key_after;
                if (mvseq >= 0) moves.push_back({((uint64_t)(uint32_t)pq_list_of(R.key_after, G.n) << 32) | (uint32_t)mvseq, x});
            }
        }
        // rewires + kills
        for (int32_t i = 0; i < P; ++i) {
            Cand& C = cand[i];
            int32_t m = C.m;
            if (A.o_v == OV_COARSEN) {
                if (m >= 1) {
                    int32_t k = C.e[C.koff].nbr;
                    for (int32_t j = 0; j < m; ++j) {
                        if (j == C.koff) continue;
                        int32_t s_r = C.e[j].twin, s_n = pslot[i][j];
                        double nw = C.e[j].val;
                        A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
                        A.e[s_n].nbr = C.e[j].nbr; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
                    }
                    A.e[C.e[C.koff].twin].val = 0;
                }
            } else {
                for (int32_t j = 0; j < m - 1; ++j) {
                    int32_t k = C.e[C.ksel[j]].nbr, s_r = C.e[j].twin, s_n = pslot[i][j];
                    if (s_r < 0) s_r = pslot[(~s_r) / BC][(~s_r) % BC];   // patched entry: the slot an earlier candidate of this round appends
                    double nw = C.e[j].val;
                    A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
                    A.e[s_n].nbr = C.e[j].nbr; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
                }
                if (m >= 1) { int32_t s_l = C.e[m - 1].twin; if (s_l < 0) s_l = pslot[(~s_l) / BC][(~s_l) % BC]; A.e[s_l].val = 0; }
            }
            for (int32_t q = 0; q < C.nkill; ++q) A.e[C.e[m + q].twin].val = 0;   // merged multi-edges (:289)
            G.n_draws += C.ndraw;
        }
        // PQ pushes in (bucket, op) order
        std::sort(moves.begin(), moves.end(), [](const Move& p, const Move& q) { return p.key < q.key; });
        for (const Move& mvv : moves) { int rc = pq_push(A, G, mvv.v, (int32_t)(mvv.key >> 32)); if (rc) return rc; }
        cleanup();
        done += P;
    }
    if (stats_out) { stats_out[0] = rounds; stats_out[1] = singles; stats_out[2] = contended_total; for (int q = 0; q < 6; ++q) stats_out[3 + q] = why[q]; stats_out[9] = single_len; stats_out[10] = single_gt384; stats_out[11] = single_gt384_len; stats_out[12] = single_max; for (int q = 0; q < 4; ++q) stats_out[13 + q] = adjcase[q]; stats_out[17] = npatch_try; stats_out[18] = npatch_ok; stats_out[19] = npatch_dead; }
    return S.finish(nelim, npop, order_out, out, out_rows);
}

extern "C" {

// the keyed neighbour order as the kernels compile it (rlap_core.h), for tests/test_keyed_order.py
double mirror_keyed_key(uint64_t seed, int64_t vertex, int phase, int64_t nbr) {
    return keyed_order_dkey(keyed_order_base(seed, vertex, phase), nbr);
}

// std_sort_emul permutation of doubles: perm_out[i] = original index ending at i.
void mirror_sort_perm(const double* keys, int64_t n, int desc, int64_t* perm_out) {
    std::vector<SRec> r((size_t)n);
    for (int64_t i = 0; i < n; ++i) { r[i].key = keys[i]; r[i].idx = (int32_t)i; r[i].aux = 0; }
    if (desc) std_sort_emul<true>(r.data(), (int)n); else std_sort_emul<false>(r.data(), (int)n);
    for (int64_t i = 0; i < n; ++i) perm_out[i] = r[i].idx;
}

// n <= 32: the stack-free variant used for batch candidates, on the (key,index) proxy view
void mirror_sort_perm_small(const double* keys, int64_t n, int desc, int64_t* perm_out) {
    Cand C;
    for (int64_t i = 0; i < n; ++i) { C.e[i].aux = keys[i]; C.ksel[i] = (uint8_t)i; }
    KIArr arr; arr.e = C.e; arr.ix = C.ksel;
    if (desc) gs_std_sort_small<KI>(arr, (int)n, KIGreater()); else gs_std_sort_small<KI>(arr, (int)n, KILess());
    for (int64_t i = 0; i < n; ++i) perm_out[i] = C.ksel[i];
}

void mirror_heapsort_perm(const double* keys, int64_t n, int desc, int64_t* perm_out) {
    std::vector<SRec> r((size_t)n);
    for (int64_t i = 0; i < n; ++i) { r[i].key = keys[i]; r[i].idx = (int32_t)i; r[i].aux = 0; }
    if (desc) ss_heap_sort<true>(r.data(), 0, (int)n); else ss_heap_sort<false>(r.data(), 0, (int)n);
    for (int64_t i = 0; i < n; ++i) perm_out[i] = r[i].idx;
}

// Coalesced, symmetric COO in (row[], col[], w[]) -> Schur complement rows,
// sequential driver.  Single graph. Returns status; *out malloc'd (m,3).
int mirror_approx_chol(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t,
                       int o_v, int o_n, const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots,
                       double** out, int64_t* out_rows, int64_t* order_out) {
    Setup S;
    S.build(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots);
    int64_t nelim = std::min<int64_t>(t, n - 1);
    if (nelim < 0) nelim = 0;
    int64_t npop = 0;
    for (int64_t e1 = 1; e1 <= nelim; ++e1) {
        int32_t v = (o_v == OV_RANDOM) ? (int32_t)S.perm_l[n - e1] : pq_pop(S.A, S.G);
        if (order_out) order_out[npop] = v;
        ++npop;
        int rc = serial_eliminate(S.A, S.G, S.B, S.cap, v, e1);
        if (rc) return rc;
    }
    return S.finish(nelim, npop, order_out, out, out_rows);
}

// stats_out: [0] rounds, [1] single-vertex fallbacks, [2] contended targets, [3..8] what ended the rounds
// (adjacent candidate, long column, multi-edge, complex key, pre-empting move, nothing).
// Schedule model of the dataflow design of DESIGN.md section 8 item 7, for o_v = random (the order is known in advance): the
// sequential elimination is run once and, beside it, an event simulation for each worker count in workers[]: the next position
// of the order is claimed by the wave that is free first; it can gather once every EARLIER vertex that was adjacent to it when
// that one was eliminated has committed (touch[]); its neighbour count is published cost[0] + cost[1]*len later; the uniform
// offsets are a chain over the published counts (cost[4] per link); sampling + commit take cost[2] + cost[3]*len.  No
// speculation is modelled (a wave that has claimed a blocked position waits).  out[k] = makespan (same unit as cost[]) for
// workers[k]; out[nW + 0..3] = dependency levels (longest chain, in vertices), critical path (unbounded workers), mean and
// maximum gathered length.
int mirror_flow_model(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t, int o_n,
                      const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots, int32_t nW, const int32_t* workers,
                      const double* cost, double* out) {
    Setup S;
    S.build(row, col, w, E, n, t, OV_RANDOM, o_n, perm, shuffle_seed, pool_slots);
    int64_t nelim = std::min<int64_t>(t, n - 1);
    if (nelim < 0) nelim = 0;
    const int K = nW + 1;                                   // last model: unbounded workers
    std::vector<std::vector<double>> touch((size_t)K, std::vector<double>((size_t)n, 0.0));
    std::vector<std::vector<double>> freeq((size_t)nW);      // min-heaps of the waves' free times
    for (int k = 0; k < nW; ++k) freeq[k].assign((size_t)workers[k], 0.0);
    std::vector<double> off_prev((size_t)K, 0.0), makespan((size_t)K, 0.0);
    std::vector<int32_t> levtouch((size_t)n, 0);
    int32_t maxlev = 0, maxlen = 0;
    double sumlen = 0;
    for (int64_t e1 = 1; e1 <= nelim; ++e1) {
        int32_t v = (int32_t)S.perm_l[n - e1];
        int32_t len = serial_gather(S.A, v, S.B, S.cap);
        std::vector<int32_t> nb(S.B.a_nbr, S.B.a_nbr + len);
        sumlen += len; if (len > maxlen) maxlen = len;
        int32_t lev = levtouch[v] + 1;
        if (lev > maxlev) maxlev = lev;
        for (int32_t x : nb) if (levtouch[x] < lev) levtouch[x] = lev;
        const double tp = cost[0] + cost[1] * len, tc = cost[2] + cost[3] * len;
        for (int k = 0; k < K; ++k) {
            double wf = 0;
            if (k < nW) { std::pop_heap(freeq[k].begin(), freeq[k].end(), std::greater<double>()); wf = freeq[k].back(); }
            const double start = std::max(wf, touch[k][v]);
            const double pub = start + tp;
            const double off = std::max(pub, off_prev[k]) + cost[4];
            off_prev[k] = off;
            const double fin = off + tc;
            if (k < nW) { freeq[k].back() = fin; std::push_heap(freeq[k].begin(), freeq[k].end(), std::greater<double>()); }
            for (int32_t x : nb) if (touch[k][x] < fin) touch[k][x] = fin;
            if (fin > makespan[k]) makespan[k] = fin;
        }
        int rc = serial_eliminate(S.A, S.G, S.B, S.cap, v, e1);
        if (rc) return rc;
    }
    for (int k = 0; k < nW; ++k) out[k] = makespan[k];
    out[nW] = maxlev; out[nW + 1] = makespan[nW]; out[nW + 2] = nelim ? sumlen / (double)nelim : 0; out[nW + 3] = maxlen;
    return 0;
}

int mirror_approx_chol_batch(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t,
                             int o_v, int o_n, const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots, int32_t Bsz,
                             double** out, int64_t* out_rows, int64_t* order_out, int64_t* stats_out) {
    int64_t st[13];
    int rc = mirror_batch_impl<32>(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots, Bsz, out, out_rows, order_out, st);
    if (stats_out) for (int q = 0; q < 3; ++q) stats_out[q] = st[q];
    return rc;
}

// same with the candidate capacity of the kernel variant (32, 64 or 128) and the full statistics (13 values: + singles: total length, count and total length of those over 384 entries, max)
int mirror_approx_chol_batch_bc(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t,
                                int o_v, int o_n, const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots, int32_t Bsz,
                                int32_t bc, double** out, int64_t* out_rows, int64_t* order_out, int64_t* stats_out) {
    if (bc == 128) return mirror_batch_impl<128>(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots, Bsz, out, out_rows, order_out, stats_out);
    if (bc == 64) return mirror_batch_impl<64>(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots, Bsz, out, out_rows, order_out, stats_out);
    return mirror_batch_impl<32>(row, col, w, E, n, t, o_v, o_n, perm, shuffle_seed, pool_slots, Bsz, out, out_rows, order_out, stats_out);
}

void mirror_free(double* p) { std::free(p); }
}

// ---------------------------------------------------------------------------
// The dataflow ("flow") form of the elimination for o_v = random (rlap_flow.h), run by `nwaves` virtual waves that a seeded
// scheduler interleaves at random, one small step at a time: claim -> wait for pend == 0 -> gather / sort / merge / order /
// publish the count -> look-back for the uniform offset -> sample -> commit (one appended entry per step, in a shuffled order
// of the positions: concurrent eliminations reach a column out of order, and so do the lanes of one) -> decrements.
// Every step uses only what the kernel has at that point.  Then the header links, the tag order of the surviving columns and
// the ordinary output pass.  stats_out: [0] gathers that found equal ids, [1] surviving columns whose appended entries were out
// of order, [2] max positions in flight, [3] appended entries, [4] chunks beyond the inline directory.
// ---------------------------------------------------------------------------
namespace {
struct FEnt { double val; int32_t nbr, twin, tag; };
struct FWave {
    int state = 0;            // 0 claim, 1 wait pend, 2 look-back, 3 commit entries, 4 decrements, 5 done
    int32_t idx = -1, v = -1;
    std::vector<FEnt> fin;    // final (o_n) order, merged
    std::vector<int32_t> dupcnt;   // per merged entry: duplicates folded into it
    std::vector<int32_t> kill;     // twins of merged duplicates
    std::vector<int32_t> ksel;
    std::vector<double> newv;
    std::vector<int32_t> jorder;
    int64_t D = 0;
    size_t jnext = 0;
};
}

extern "C" int mirror_flow_chol(const int64_t* row, const int64_t* col, const double* w, int64_t E, int64_t n, int64_t t, int o_n,
                                const int64_t* perm, uint64_t shuffle_seed, int32_t pool_slots, int32_t nwaves, uint64_t sched_seed,
                                double** out, int64_t* out_rows, int64_t* stats_out) {
    Setup S;
    S.build(row, col, w, E, n, t, OV_RANDOM, o_n, perm, shuffle_seed, pool_slots);
    const Arrays& A = S.A;
    int64_t nelim = std::min<int64_t>(t, n - 1);
    if (nelim < 0) nelim = 0;
    // ---- flow state ----
    const int32_t Q = (int32_t)nelim + 1;
    std::vector<int32_t> cdir((size_t)n * FDIR, FD_EMPTY), atag((size_t)A.slot_cap, -1), qv((size_t)Q, -1), qg((size_t)Q, 0), ctrl(FC_WORDS, 0);
    std::vector<unsigned long long> lb((size_t)Q, 0ull);
    FlowArrays F;
    F.cdir = cdir.data(); F.atag = atag.data(); F.lb = lb.data(); F.qv = qv.data(); F.qg = qg.data(); F.ctrl = ctrl.data(); F.Q = Q;
    lb[0] = LB_VALID | LB_PREFIX;   // sentinel: prefix 0
    for (int64_t v = 0; v < n; ++v) { A.vr[v].key = 0; A.vr[v].pqpos = FPOS_NONE; }
    for (int64_t p = 0; p < nelim; ++p) { int32_t v = (int32_t)S.perm_l[n - 1 - p]; qv[1 + p] = v; A.vr[v].pqpos = (int32_t)(1 + p); }
    for (int64_t u = 0; u < n; ++u)
        for (int32_t s = A.colptr[u]; s < A.colptr[u + 1]; ++s) {
            const int32_t a = A.e[s].nbr;
            if (A.e[s].val > 0 && A.vr[u].pqpos < A.vr[a].pqpos && A.vr[a].pqpos != FPOS_NONE) A.vr[a].key += 1;
        }
    std::mt19937_64 sched(sched_seed);
    int64_t st_dupgather = 0, st_unsorted = 0, st_inflight_max = 0, st_appended = 0, st_ovf = 0;
    auto chunk_base = [&](int32_t k, int c) -> int32_t {   // directory look-up, allocating on first use
        int32_t* word;
        if (c < FDIR - 1) word = &cdir[(size_t)k * FDIR + c];
        else {
            int32_t& ow = cdir[(size_t)k * FDIR + FDIR - 1];
            if (ow == FD_EMPTY) {
                int32_t base = pool_take(A.pool_top, FDIR_OVF_SLOTS);
                if (base < 0 || base > A.slot_cap - FDIR_OVF_SLOTS) return FD_FAIL;
                int32_t* o = reinterpret_cast<int32_t*>(A.e + base);
                for (int q = 0; q < FDIR_OVF; ++q) o[q] = FD_EMPTY;
                ow = base; ++st_ovf;
            }
            word = reinterpret_cast<int32_t*>(A.e + ow) + (c - (FDIR - 1));
        }
        if (*word == FD_EMPTY) {
            int32_t need = 1 + chunk_cap(c);
            int32_t base = pool_take(A.pool_top, need);
            if (base < 0 || base > A.slot_cap - need) return FD_FAIL;
            *word = base;
        }
        return *word;
    };
    auto order_keyed = [&](std::vector<FEnt>& b, int32_t v) {   // o_n order of the merged, id-sorted entries (:661-673)
        const int32_t m = (int32_t)b.size();
        std::vector<SRec> rec((size_t)std::max(m, 1));
        if (o_n == ON_RANDOM) {
            uint64_t base = keyed_order_base(A.shuffle_seed, v, 0);
            for (int32_t i = 0; i < m; ++i) { rec[i].key = keyed_order_dkey(base, b[i].nbr); rec[i].idx = i; rec[i].aux = 0; }
            std_sort_emul<false>(rec.data(), m);
        } else {
            for (int32_t i = 0; i < m; ++i) { rec[i].key = b[i].val; rec[i].idx = i; rec[i].aux = 0; }
            if (o_n == ON_ASC) std_sort_emul<false>(rec.data(), m); else std_sort_emul<true>(rec.data(), m);
        }
        std::vector<int32_t> p((size_t)m);
        for (int32_t i = 0; i < m; ++i) p[i] = rec[i].idx;
        return p;
    };
    std::vector<FWave> W((size_t)std::max(nwaves, 1));
    int64_t ndone = 0;
    auto step = [&](FWave& X) -> int {
        switch (X.state) {
        case 0: {
            int32_t idx = ctrl[FC_CLAIM]++;
            while (idx < Q && qv[idx] < 0) idx = ctrl[FC_CLAIM]++;   // sentinels
            if (idx >= Q) { X.state = 5; ++ndone; return 0; }
            X.idx = idx; X.v = qv[idx]; X.state = 1;
            return 0;
        }
        case 1: {
            const int32_t v = X.v;
            if (A.vr[v].key != 0) return 0;
            // ---- gather in PHYSICAL order: appended index descending, then the CSR segment descending ----
            std::vector<FEnt> a;
            const int32_t acnt = A.vr[v].app_cnt;
            for (int32_t i = acnt - 1; i >= 0; --i) {
                const int c = chunk_of(i);
                const int32_t base = c < FDIR - 1 ? cdir[(size_t)v * FDIR + c] : reinterpret_cast<int32_t*>(A.e + cdir[(size_t)v * FDIR + FDIR - 1])[c - (FDIR - 1)];
                if (base < 0) return ST_INTERNAL;
                const int32_t s = base + 1 + (i - chunk_start(c));
                if (A.e[s].val > 0) a.push_back({A.e[s].val, A.e[s].nbr, A.e[s].twin, atag[s]});
            }
            const size_t napp = a.size();
            for (int32_t s = A.colptr[v + 1] - 1; s >= A.colptr[v]; --s) if (A.e[s].val > 0) a.push_back({A.e[s].val, A.e[s].nbr, A.e[s].twin, -1});
            for (const FEnt& e : a) if (A.vr[e.nbr].pqpos < X.idx) return ST_INTERNAL;   // an earlier vertex still in the column: the counters are wrong
            auto sort_by_id = [&](std::vector<FEnt>& x) {
                std::vector<SRec> rec(std::max<size_t>(x.size(), 1));
                for (size_t i = 0; i < x.size(); ++i) { rec[i].key = (double)x[i].nbr; rec[i].idx = (int32_t)i; rec[i].aux = 0; }
                std_sort_emul<false>(rec.data(), (int)x.size());
                std::vector<FEnt> y(x.size());
                for (size_t i = 0; i < x.size(); ++i) y[i] = x[rec[i].idx];
                return y;
            };
            std::vector<FEnt> sorted = sort_by_id(a);
            bool dup = false;
            for (size_t i = 1; i < sorted.size(); ++i) if (sorted[i].nbr == sorted[i - 1].nbr) dup = true;
            if (dup) {   // equal ids: the list order matters -- appended entries newest first (descending tag), and sort again
                ++st_dupgather;
                std::sort(a.begin(), a.begin() + (long)napp, [](const FEnt& p, const FEnt& q) { return p.tag > q.tag; });
                sorted = sort_by_id(a);
            }
            // ---- merge (:646-659): the first of a run survives with the sum, the others' twins die ----
            std::vector<FEnt> b;
            X.dupcnt.clear(); X.kill.clear();
            for (size_t i = 0; i < sorted.size(); ++i) {
                if (b.empty() || sorted[i].nbr != b.back().nbr) { b.push_back(sorted[i]); X.dupcnt.push_back(0); }
                else { b.back().val += sorted[i].val; X.kill.push_back(sorted[i].twin); X.dupcnt.back() += 1; }
            }
            std::vector<int32_t> p = order_keyed(b, v);
            const int32_t m = (int32_t)b.size();
            X.fin.resize((size_t)m);
            std::vector<int32_t> dc((size_t)m);
            for (int32_t j = 0; j < m; ++j) { X.fin[j] = b[p[j]]; dc[j] = X.dupcnt[p[j]]; }
            X.dupcnt = dc;
            const int64_t c = m > 1 ? m - 1 : 0;
            lb[X.idx] = LB_VALID | (unsigned long long)c;   // own count
            X.state = 2;
            return 0;
        }
        case 2: {
            // decoupled look-back: own count is published; walk back to the nearest inclusive prefix
            unsigned long long sum = 0;
            for (int32_t q = X.idx - 1;; --q) {
                const unsigned long long wv = lb[q];
                if (!(wv & LB_VALID)) return 0;   // not there yet
                sum += wv & LB_MASK;
                if (wv & LB_PREFIX) break;
            }
            const int32_t m = (int32_t)X.fin.size();
            const int64_t c = m > 1 ? m - 1 : 0;
            X.D = (int64_t)sum;
            if (X.D + c > A.rng_len) return ST_RNG_OVERFLOW;
            lb[X.idx] = LB_VALID | LB_PREFIX | (unsigned long long)(X.D + c);
            // ---- sample (:728-779) ----
            std::vector<double> cum((size_t)std::max(m, 1));
            double csum = 0;
            for (int32_t j = 0; j < m; ++j) { csum += X.fin[j].val; cum[j] = csum; }
            X.ksel.assign((size_t)std::max(m, 1), 0); X.newv.assign((size_t)std::max(m, 1), 0.0);
            double wdeg = csum, colScale = 1;
            for (int32_t j = 0; j < m - 1; ++j) {
                double ww = X.fin[j].val * colScale;
                double f = ww / wdeg;
                double u = A.rng[X.D + j];
                double r = u * (csum - cum[j]) + cum[j];
                X.ksel[j] = upper_index(cum.data(), m, r);
                double omf = 1 - f;
                X.newv[j] = f * omf * wdeg;
                colScale = colScale * omf;
                wdeg = wdeg * omf * omf;
            }
            X.jorder.resize((size_t)std::max(m - 1, 0));
            for (int32_t j = 0; j < m - 1; ++j) X.jorder[j] = j;
            std::shuffle(X.jorder.begin(), X.jorder.end(), sched);
            X.jnext = 0;
            X.state = 3;
            return 0;
        }
        case 3: {
            const int32_t m = (int32_t)X.fin.size();
            if (X.jnext < X.jorder.size()) {
                const int32_t j = X.jorder[X.jnext++];
                const int32_t k = X.fin[X.ksel[j]].nbr, a = X.fin[j].nbr, s_r = X.fin[j].twin;
                const double nw = X.newv[j];
                const int32_t ai = A.vr[k].app_cnt++;   // (atomic on the device)
                const int c = chunk_of(ai);
                const int32_t base = chunk_base(k, c);
                if (base < 0) return ST_POOL_OVERFLOW;
                const int32_t s_n = base + 1 + (ai - chunk_start(c));
                ++st_appended;
                A.e[s_n].nbr = a; A.e[s_n].val = nw; A.e[s_n].twin = s_r;
                atag[s_n] = (int32_t)(X.D + j);
                A.e[s_r].nbr = k; A.e[s_r].val = nw; A.e[s_r].twin = s_n;
                if (nw > 0) {   // a live pair: it waits at whichever end comes later for the one that comes earlier
                    const int32_t pa = A.vr[a].pqpos, pk = A.vr[k].pqpos;
                    if (pk < pa && pa != FPOS_NONE) A.vr[a].key += 1;
                    if (pa < pk && pk != FPOS_NONE) A.vr[k].key += 1;
                }
                return 0;
            }
            if (m >= 1) A.e[X.fin[m - 1].twin].val = 0;          // :791-792
            for (int32_t tw : X.kill) A.e[tw].val = 0;          // :655
            X.state = 4;
            return 0;
        }
        case 4: {
            for (size_t j = 0; j < X.fin.size(); ++j) {
                const int32_t a = X.fin[j].nbr;
                if (A.vr[a].pqpos != FPOS_NONE) { A.vr[a].key -= 1 + X.dupcnt[j]; if (A.vr[a].key < 0) return ST_INTERNAL; }
            }
            X.state = 0;
            return 0;
        }
        default: return 0;
        }
    };
    int64_t idle = 0;
    while (ndone < (int64_t)W.size()) {
        FWave& X = W[(size_t)(sched() % W.size())];
        const int before = X.state; const size_t jb = X.jnext;
        int rc = step(X);
        if (rc) return rc;
        if (X.state == before && X.jnext == jb) { if (++idle > 50000000) return ST_INTERNAL; } else idle = 0;
        int64_t infl = 0; for (const FWave& Y : W) infl += (Y.state >= 1 && Y.state <= 4); if (infl > st_inflight_max) st_inflight_max = infl;
    }
    for (int64_t v = 0; v < n; ++v) if (A.vr[v].pqpos != FPOS_NONE && A.vr[v].key != 0) return ST_INTERNAL;
    S.G.n_draws = (int64_t)(lb[(size_t)Q - 1] & LB_MASK);
    // ---- afterwards: header links, tag order of the surviving columns ----
    for (int64_t v = 0; v < n; ++v) flow_finish_vertex(A, F, (int32_t)v);
    for (int64_t q = nelim; q < n; ++q) {
        const int32_t v = (int32_t)S.perm_l[n - 1 - q];
        const int32_t acnt = A.vr[v].app_cnt;
        std::vector<int32_t> slots;
        for (int32_t i = 0; i < acnt; ++i) { const int c = chunk_of(i); const int32_t base = c < FDIR - 1 ? cdir[(size_t)v * FDIR + c] : reinterpret_cast<int32_t*>(A.e + cdir[(size_t)v * FDIR + FDIR - 1])[c - (FDIR - 1)]; slots.push_back(base + 1 + (i - chunk_start(c))); }
        bool sorted_already = true;
        for (int32_t i = 1; i < acnt; ++i) if (atag[slots[i]] < atag[slots[i - 1]]) sorted_already = false;
        if (sorted_already) continue;
        ++st_unsorted;
        std::vector<std::pair<int32_t, Slot>> items;
        for (int32_t i = 0; i < acnt; ++i) items.push_back({atag[slots[i]], A.e[slots[i]]});
        std::sort(items.begin(), items.end(), [](const std::pair<int32_t, Slot>& p, const std::pair<int32_t, Slot>& q) { return p.first < q.first; });
        for (int32_t i = 0; i < acnt; ++i) { A.e[slots[i]] = items[i].second; atag[slots[i]] = items[i].first; }
    }
    if (stats_out) { stats_out[0] = st_dupgather; stats_out[1] = st_unsorted; stats_out[2] = st_inflight_max; stats_out[3] = st_appended; stats_out[4] = st_ovf; }
    for (int64_t v = 0; v < n; ++v) { A.vr[v].key = 0; A.vr[v].pqpos = -1; }
    return S.finish(nelim, nelim, nullptr, out, out_rows);
}
