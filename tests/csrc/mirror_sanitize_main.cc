// mirror_sanitize_main.cc -- the device data structures and batch-round rules (rlap_amd/csrc/rlap_core.h, run on the CPU by
// host_mirror.cc) under AddressSanitizer + UndefinedBehaviorSanitizer, compared with the oracle on the shapes the 256-thread
// PATCH path meets: 32-slot candidates in rounds of 32 and 128, chains of dependent candidates (cliques: every candidate of a
// round is adjacent to all earlier ones -- DEPMAX and beyond), small dense graphs, hubs, multi-edges, all (o_v, o_n) modes,
// 64- and 128-slot candidates for the random order.  Test-only (host code under -fsanitize=address,undefined).
// Built and run by tests/test_sanitizers.py:  g++ -fsanitize=address,undefined -fno-sanitize-recover=all ...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

#include "host_mirror.cc"
#include "../../oracle/rlap_oracle.cc"

namespace {

struct Graph { int64_t n; std::vector<int64_t> row, col; std::vector<double> w; };

uint64_t rstate = 0x1234567ull;
uint64_t rnd() { rstate = rlap::mix64(rstate); return rstate; }

Graph from_pairs(int64_t n, const std::set<std::pair<int64_t, int64_t>>& und, bool weighted) {
    Graph g; g.n = n;
    std::vector<std::pair<int64_t, int64_t>> dir;
    for (auto& e : und) { dir.push_back({e.second, e.first}); dir.push_back({e.first, e.second}); }   // (col, row) pairs
    std::sort(dir.begin(), dir.end());
    for (auto& d : dir) { g.col.push_back(d.first); g.row.push_back(d.second); }
    g.w.assign(dir.size(), 1.0);
    if (weighted) {
        for (size_t k = 0; k < dir.size(); ++k) {
            int64_t a = std::min(dir[k].first, dir[k].second), b = std::max(dir[k].first, dir[k].second);
            g.w[k] = 0.5 + (double)(rlap::mix64((uint64_t)(a * 1000003 + b)) >> 11) * (1.0 / 9007199254740992.0);
        }
    }
    return g;
}

Graph clique(int64_t n, bool weighted) {
    std::set<std::pair<int64_t, int64_t>> und;
    for (int64_t a = 0; a < n; ++a) for (int64_t b = a + 1; b < n; ++b) und.insert({a, b});
    return from_pairs(n, und, weighted);
}

Graph ba(int64_t n, int64_t m, bool weighted) {
    std::set<std::pair<int64_t, int64_t>> und;
    std::vector<int64_t> rep, targets;
    for (int64_t i = 0; i < m; ++i) targets.push_back(i);
    for (int64_t i = m; i < n; ++i) {
        for (int64_t t : targets) { und.insert({std::min(t, i), std::max(t, i)}); rep.push_back(t); }
        for (int64_t j = 0; j < m; ++j) rep.push_back(i);
        std::set<int64_t> ch;
        while ((int64_t)ch.size() < m) ch.insert(rep[rnd() % rep.size()]);
        targets.assign(ch.begin(), ch.end());
    }
    return from_pairs(n, und, weighted);
}

// a ring of cliques joined by single edges: long chains of dependent candidates next to independent ones
Graph clique_ring(int64_t k, int64_t sz, bool weighted) {
    std::set<std::pair<int64_t, int64_t>> und;
    for (int64_t c = 0; c < k; ++c) {
        for (int64_t a = 0; a < sz; ++a) for (int64_t b = a + 1; b < sz; ++b) und.insert({c * sz + a, c * sz + b});
        int64_t u = c * sz + sz - 1, v = ((c + 1) % k) * sz;
        if (u != v) und.insert({std::min(u, v), std::max(u, v)});
    }
    return from_pairs(k * sz, und, weighted);
}

int failures = 0;

void check(const char* name, const Graph& g, int64_t t, int o_v, int o_n, int bc, int bsz) {
    const int64_t E = (int64_t)g.row.size();
    std::vector<double> info((size_t)E * 3);
    for (int64_t p = 0; p < E; ++p) { info[3 * p] = (double)g.row[p]; info[3 * p + 1] = (double)g.col[p]; info[3 * p + 2] = g.w[p]; }
    std::vector<int64_t> perm((size_t)g.n);
    for (int64_t i = 0; i < g.n; ++i) perm[i] = i;
    for (int64_t i = g.n - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (uint64_t)(i + 1)]);
    double* ref = nullptr; int64_t ref_rows = 0;
    rlap_oracle_stats st;
    int rc = rlap_oracle_approx_chol(info.data(), E, g.n, t, o_v, o_n, perm.data(), 77, 0, 1, &ref, &ref_rows, nullptr, &st);
    if (rc) { std::printf("FAIL %s: oracle rc %d\n", name, rc); ++failures; return; }
    double* got = nullptr; int64_t got_rows = 0;
    int64_t stats[24];
    const int32_t pool = (int32_t)(4 * E + 64 * g.n + 4096);
    if (bsz <= 0) rc = mirror_approx_chol(g.row.data(), g.col.data(), g.w.data(), E, g.n, t, o_v, o_n, perm.data(), 77, pool, &got, &got_rows, nullptr);
    else rc = mirror_approx_chol_batch_bc(g.row.data(), g.col.data(), g.w.data(), E, g.n, t, o_v, o_n, perm.data(), 77, pool, bsz, bc, &got, &got_rows, nullptr, stats);
    bool ok = rc == 0 && got_rows == ref_rows && (ref_rows == 0 || std::memcmp(got, ref, sizeof(double) * 3 * (size_t)ref_rows) == 0);
    if (!ok) { std::printf("FAIL %s: o_v %d o_n %d bc %d batch %d: rc %d rows %lld vs %lld\n", name, o_v, o_n, bc, bsz, rc, (long long)got_rows, (long long)ref_rows); ++failures; }
    if (got) mirror_free(got);
    rlap_oracle_free(ref);
}

}  // namespace

int main() {
    struct Named { const char* name; Graph g; };
    std::vector<Named> gs;
    gs.push_back({"K6", clique(6, false)});
    gs.push_back({"K6w", clique(6, true)});
    gs.push_back({"K40", clique(40, false)});           // every column beyond 32 entries: single-vertex path throughout
    gs.push_back({"K20w", clique(20, true)});
    gs.push_back({"ring12x7", clique_ring(12, 7, false)});
    gs.push_back({"ring30x5w", clique_ring(30, 5, true)});
    gs.push_back({"BA(600,3)", ba(600, 3, false)});
    gs.push_back({"BA(400,8)", ba(400, 8, false)});
    gs.push_back({"BA(400,8)w", ba(400, 8, true)});
    gs.push_back({"BA(150,40)", ba(150, 40, false)});   // README shape: dense, long columns
    int ncheck = 0;
    for (auto& G : gs) {
        const int64_t n = G.g.n;
        for (int o_v = 0; o_v < 3; ++o_v) for (int o_n = 0; o_n < 3; ++o_n) {
            for (int64_t t : {n / 2, n - 1}) {
                check(G.name, G.g, t, o_v, o_n, 32, 0);                       // sequential driver
                for (int bsz : {7, 32, 128}) { check(G.name, G.g, t, o_v, o_n, 32, bsz); ++ncheck; }
                if (o_v == 0) { check(G.name, G.g, t, o_v, o_n, 64, 64); check(G.name, G.g, t, o_v, o_n, 128, 32); ncheck += 2; }
                ++ncheck;
            }
        }
    }
    std::printf("%d checks, %d failures\n", ncheck, failures);
    return failures ? 1 : 0;
}
