// rlap_wave_sort.h -- wave64 helpers and the wave-parallel restatements of libstdc++ 11 std::sort (same permutation under ties as
// rlap_core.h::gs_std_sort): wave_std_sort (any length, data in LDS or global memory), wave_lvl_sort (level-synchronous, LDS),
// wave_sort64 (one element per lane, registers), group_sort (two 32-lane groups per wave).  Templates and inline device code only:
// included by every translation unit that sorts (rlap_kernels.hip, twice; rlap_flow.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "rlap_core.h"

namespace rlap {

// ---------------------------------------------------------------------------
// small wave helpers (wave64)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

#if defined(__HIP_DEVICE_COMPILE__)
#define ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))   // out-of-line functions see generic pointers otherwise
#else
#define ASSUME_LDS(p) ((void)0)
#endif
#define RLAP_IS_LDS(p) __builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p))
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---------------------------------------------------------------------------
// Wave-parallel restatement of libstdc++'s std::sort (same permutation under ties as
// rlap_core.h::gs_std_sort, which is the sequential restatement).  One wave, data in LDS.
//   * introsort loop: the unguarded Hoare partition is done by the whole wave.  The up-scan stops at
//     the positions with !(x < pivot), the down-scan at those with !(pivot < x); the t-th up-stop is
//     swapped with the t-th down-stop while it lies to its left, and swapped elements are never
//     looked at again, so both stop lists can be taken from the array as it is before the swaps.
//     With k swaps, the cut is min(u_k, d_{k-1}).
//   * final insertion sort == independent stable sorts of the <=16-element segments the loop leaves
//     (everything left of a cut is <= everything right of it), one lane per segment.
// Scratch per wave: two uint16 lists of n entries, n bits of segment marks, a small segment stack.
// ---------------------------------------------------------------------------
template <int CAP>
struct WaveSortScratchT {
    uint16_t ulist[CAP + 2];
    uint16_t dlist[CAP + 2];
    uint32_t segmark[(CAP + 31) / 32 + 1];
    int32_t stk[3 * 48];
};
typedef WaveSortScratchT<SCAP> WaveSortScratch;

struct WaveSortPtrs { uint16_t* ulist; uint16_t* dlist; uint32_t* segmark; int32_t* stk; };

// REGT > 0: arrays of at most 64*REGT elements run the final phase with one lane per ELEMENT, the elements held
// in registers between the rank computation and the stores (no second buffer).
template <class T, class Less, int REGT = 0>
__device__ void wave_std_sort(T* a, int n, Less less, const WaveSortPtrs W_, int lane, T* obuf = nullptr, int depth_in = -1) {
    struct { uint16_t* ulist; uint16_t* dlist; uint32_t* segmark; int32_t* stk; } W = {W_.ulist, W_.dlist, W_.segmark, W_.stk};
    if (n < 2) return;
    const uint64_t lt = lanemask_lt(lane);
    for (int q = lane; q < (n + 31) / 32 + 1; q += 64) W.segmark[q] = 0u;
    WAVE_SYNC();
    if (n <= 16) {
        if (lane == 0) gs_insertion_sort<T>(a, n, less);
        WAVE_SYNC();
        return;
    }
    int depth0 = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
    depth0 *= 2;
    if (depth_in >= 0) depth0 = depth_in;   // a segment of a larger array: the introsort loop's remaining depth limit
    int sp = 0;
    if (lane == 0) { W.stk[0] = 0; W.stk[1] = n; W.stk[2] = depth0; }
    sp = 1;
    WAVE_SYNC();
    while (sp > 0) {
        --sp;
        int first = W.stk[3 * sp], last = W.stk[3 * sp + 1], depth = W.stk[3 * sp + 2];
        bool heap_sorted = false;
        while (last - first > 16) {
            if (depth == 0) {
                if (lane == 0) gs_heap_sort<T>(a, first, last, less);
                WAVE_SYNC();
                heap_sorted = true;
                break;
            }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1), by one lane
            if (lane == 0) {
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
            // stop lists of the two scans over [first+1, last)
            int nu = 0, nd = 0;
            for (int p0 = first + 1; p0 < last; p0 += 64) {
                int p = p0 + lane;
                bool stop = (p < last) && !less(a[p], pv);
                uint64_t mk = __ballot(stop);
                if (stop) W.ulist[nu + popc64(mk & lt)] = (uint16_t)p;
                nu += popc64(mk);
            }
            for (int p0 = last - 1; p0 > first; p0 -= 64) {
                int p = p0 - lane;
                bool stop = (p > first) && !less(pv, a[p]);
                uint64_t mk = __ballot(stop);
                if (stop) W.dlist[nd + popc64(mk & lt)] = (uint16_t)p;
                nd += popc64(mk);
            }
            if (lane == 0) W.dlist[nd] = (uint16_t)first;   // the pivot itself stops the down-scan
            WAVE_SYNC();
            // number of swaps: pairs (u_t, d_t) with u_t < d_t form a prefix
            int k = 0;
            {
                const int tmax = nu < nd ? nu : nd;
                bool open = true;
                for (int t0 = 0; t0 < tmax && open; t0 += 64) {
                    int t = t0 + lane;
                    bool ok = (t < tmax) && (W.ulist[t] < W.dlist[t]);
                    uint64_t mk = __ballot(ok);
                    // ok is monotone (true...true,false...): count the leading run
                    uint64_t inv = ~mk;
                    int run = inv ? __builtin_ctzll(inv) : 64;
                    k += run;
                    open = (run == 64);
                }
            }
            T xu, xd;
            for (int t0 = 0; t0 < k; t0 += 64) {
                int t = t0 + lane;
                if (t < k) { xu = a[W.ulist[t]]; xd = a[W.dlist[t]]; }
                WAVE_SYNC();
                if (t < k) { a[W.ulist[t]] = xd; a[W.dlist[t]] = xu; }
                WAVE_SYNC();
            }
            int cut;
            {
                int cu = (k < nu) ? (int)W.ulist[k] : 0x7FFFFFFF;
                int cd = (k > 0) ? (int)W.dlist[k - 1] : 0x7FFFFFFF;
                cut = cu < cd ? cu : cd;
            }
            WAVE_SYNC();
            // recurse on [cut,last), continue with [first,cut)
            if (last - cut > 16) {
                if (lane == 0) { W.stk[3 * sp] = cut; W.stk[3 * sp + 1] = last; W.stk[3 * sp + 2] = depth; }
                ++sp;
            } else if (lane == 0) {
                atomicOr(&W.segmark[cut >> 5], 1u << (cut & 31));
            }
            WAVE_SYNC();
            last = cut;
        }
        if (lane == 0) atomicOr(&W.segmark[first >> 5], 1u << (first & 31));
        (void)heap_sorted;
        WAVE_SYNC();
    }
    if (REGT > 0 && n <= 64 * REGT) {
        T vv[REGT > 0 ? REGT : 1];
        int rr[REGT > 0 ? REGT : 1];
#pragma unroll
        for (int t = 0; t < REGT; ++t) {
            const int p = lane + 64 * t;
            rr[t] = -1;
            if (p < n) {
                const int w = p >> 5;
                const uint32_t here = W.segmark[w];
                const uint32_t lowm = here & (0xFFFFFFFFu >> (31 - (p & 31)));
                int s0;
                if (lowm) s0 = w * 32 + 31 - __builtin_clz(lowm);
                else { int w1 = w - 1; uint32_t bb = W.segmark[w1]; while (bb == 0u) { --w1; bb = W.segmark[w1]; } s0 = w1 * 32 + 31 - __builtin_clz(bb); }
                const uint32_t highm = ((p & 31) == 31) ? 0u : (here & (0xFFFFFFFFu << ((p & 31) + 1)));
                int e0 = n;
                if (highm) e0 = w * 32 + __builtin_ctz(highm);
                else { for (int w1 = w + 1; w1 * 32 < n; ++w1) { uint32_t bb = W.segmark[w1]; if (bb) { e0 = w1 * 32 + __builtin_ctz(bb); break; } } }
                const T v = a[p];
                int r = s0;
                for (int q = s0; q < e0; ++q) {
                    const T x = a[q];
                    r += (q != p && (less(x, v) || (!less(v, x) && q < p))) ? 1 : 0;
                }
                vv[t] = v; rr[t] = r;
            }
        }
        WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < REGT; ++t) if (rr[t] >= 0) a[rr[t]] = vv[t];
        WAVE_SYNC();
        return;
    }
    if (obuf) {
        // final insertion sort == stable sort of every marked segment (<= 16 elements): one lane per ELEMENT
        // computes its rank inside its segment and writes it to its place in obuf, then everything is copied back
        for (int p = lane; p < n; p += 64) {
            const int w = p >> 5;
            const uint32_t here = W.segmark[w];
            const uint32_t lowm = here & (0xFFFFFFFFu >> (31 - (p & 31)));
            int s0;
            if (lowm) s0 = w * 32 + 31 - __builtin_clz(lowm);
            else { int w1 = w - 1; uint32_t bb = W.segmark[w1]; while (bb == 0u) { --w1; bb = W.segmark[w1]; } s0 = w1 * 32 + 31 - __builtin_clz(bb); }
            const uint32_t highm = ((p & 31) == 31) ? 0u : (here & (0xFFFFFFFFu << ((p & 31) + 1)));
            int e0 = n;
            if (highm) e0 = w * 32 + __builtin_ctz(highm);
            else { for (int w1 = w + 1; w1 * 32 < n; ++w1) { uint32_t bb = W.segmark[w1]; if (bb) { e0 = w1 * 32 + __builtin_ctz(bb); break; } } }
            const T v = a[p];
            int r = s0;
            for (int q = s0; q < e0; ++q) {
                const T x = a[q];
                r += (q != p && (less(x, v) || (!less(v, x) && q < p))) ? 1 : 0;
            }
            obuf[r] = v;
        }
        WAVE_SYNC();
        for (int p = lane; p < n; p += 64) a[p] = obuf[p];
        WAVE_SYNC();
        return;
    }
    // final insertion sort: lane l takes the segments that start in the 32-position words l, l+64, ...
    for (int w0 = lane; w0 * 32 < n; w0 += 64) {
        uint32_t bits = W.segmark[w0];
        while (bits) {
            const int s0 = w0 * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            int e0 = n;   // end = next mark after s0, or n
            if (bits) e0 = w0 * 32 + __builtin_ctz(bits);
            else {
                for (int w1 = w0 + 1; w1 * 32 < n; ++w1) { uint32_t bb = W.segmark[w1]; if (bb) { e0 = w1 * 32 + __builtin_ctz(bb); break; } }
            }
            for (int i = s0 + 1; i < e0; ++i) {
                T v = a[i];
                int j = i - 1;
                while (j >= s0 && less(v, a[j])) { a[j + 1] = a[j]; --j; }
                a[j + 1] = v;
            }
        }
    }
    WAVE_SYNC();
}

// ---------------------------------------------------------------------------
// Level-synchronous form of the same restatement for n <= 64*REG elements in LDS (element p = lane p%64, turn p/64):
// every segment of one recursion level is partitioned in the same pass, so a sort costs as many passes as the introsort
// is deep (both halves of a partition run with the same decremented depth limit, and partitions of disjoint segments do
// not see each other) instead of one pass per partition.  The segment starts are a bit mask (wave-uniform); the stop
// lists of a segment are stored at the segment's own offset; ranks inside them are differences of per-position stop counts
// (ballot popcounts, tabulated in LDS so that a segment reads the counts at its two ends):
//   up-stop p, `cu` up-stops before it in its segment, `cd` down-stops behind it: it is u_cu and swaps iff d_cu > p
//   iff cd > cu; a down-stop is d_cd and swaps iff u_cd < p iff cu > cd; with k swaps the cut min(u_k, d_{k-1}) is the
//   up-stop with cu == k and cd >= k, or the down-stop with cd == k-1 and cu <= k.
// Returns false when a segment longer than 16 meets depth limit 0 (std::sort heap-sorts it): the array is then
// partly partitioned and the caller starts over with wave_std_sort on the original order.
// ---------------------------------------------------------------------------
template <class T, class Less, int REG>
__device__ __noinline__ bool wave_lvl_sort(T* a, int n, Less less, uint16_t* ulist, uint16_t* dlist, uint32_t* tab, uint16_t* tab2, int lane, int depth_in = -1) {
    if (n < 2) return true;
    // every caller's arrays are LDS, but a call passes generic pointers: flat accesses, about twice the latency of ds_ instructions and
    // waited for one by one.  Told so, the compiler infers the address space.
    if (!RLAP_IS_LDS(a) || !RLAP_IS_LDS(tab)) __builtin_trap();
    __builtin_assume(RLAP_IS_LDS(a)); __builtin_assume(RLAP_IS_LDS(ulist)); __builtin_assume(RLAP_IS_LDS(dlist));
    __builtin_assume(RLAP_IS_LDS(tab)); __builtin_assume(RLAP_IS_LDS(tab2));
    const uint64_t lt = lanemask_lt(lane);
    const uint64_t le = lt | (1ull << lane);
    const uint64_t gt = ~le;
    uint64_t seg[REG];
#pragma unroll
    for (int t = 0; t < REG; ++t) seg[t] = 0ull;
    seg[0] = 1ull;
    int depth = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth;
    depth *= 2;
    if (depth_in >= 0) depth = depth_in;   // a segment of a larger array (wave_std_sort_hyb): the introsort loop's remaining depth limit
    int first[REG], last[REG];
    while (true) {
        // segment of every element: nearest start at or below / above its position
        int prevw[REG], nextw[REG];
        {
            int run = 0;
#pragma unroll
            for (int t = 0; t < REG; ++t) { prevw[t] = run; if (seg[t]) run = 64 * t + 63 - __builtin_clzll(seg[t]); }
            run = n;
#pragma unroll
            for (int t = REG - 1; t >= 0; --t) { nextw[t] = run; if (seg[t]) run = 64 * t + __builtin_ctzll(seg[t]); }
        }
        bool act[REG];
        uint64_t anyact = 0ull;
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            const int p = lane + 64 * t;
            const uint64_t lo = seg[t] & le, hi = seg[t] & gt;
            first[t] = lo ? 64 * t + 63 - __builtin_clzll(lo) : prevw[t];
            last[t] = hi ? 64 * t + __builtin_ctzll(hi) : nextw[t];
            act[t] = p < n && last[t] - first[t] > 16;
            anyact |= __ballot(act[t]);
        }
        if (!anyact) break;
        if (depth == 0) return false;
        --depth;
        T xn[REG];
        bool su[REG], sd[REG], moved[REG];
        uint64_t MU[REG], MD[REG];
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            const int p = lane + 64 * t;
            su[t] = false; sd[t] = false; moved[t] = false;
            if (act[t]) {
                // __move_median_to_first(first, first+1, mid, last-1), then the pivot sits at `first`
                const int f = first[t], ia = f + 1, ib = f + (last[t] - f) / 2, ic = last[t] - 1;
                const T A = a[ia], B = a[ib], C = a[ic], F = a[f], X = a[p];
                int pick; T P;
                if (less(A, B)) {
                    if (less(B, C)) { pick = ib; P = B; }
                    else if (less(A, C)) { pick = ic; P = C; }
                    else { pick = ia; P = A; }
                } else if (less(A, C)) { pick = ia; P = A; }
                else if (less(B, C)) { pick = ic; P = C; }
                else { pick = ib; P = B; }
                xn[t] = (p == f) ? P : ((p == pick) ? F : X);
                moved[t] = (p == f) || (p == pick);
                su[t] = p > f && !less(xn[t], P);
                sd[t] = p > f && !less(P, xn[t]);
            }
            MU[t] = __ballot(su[t]);
            MD[t] = __ballot(sd[t]);
        }
        // stops in front of every position (up-stops in the low half, down-stops in the high half): a segment reads the
        // entries at its two ends instead of counting over the masks
        int cumU = 0, cumD = 0;
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            const int p = lane + 64 * t;
            if (p < n) tab[p] = (uint32_t)(cumU + popc64(MU[t] & lt)) | ((uint32_t)(cumD + popc64(MD[t] & lt)) << 16);
            cumU += popc64(MU[t]); cumD += popc64(MD[t]);
        }
        const uint32_t tab_n = (uint32_t)cumU | ((uint32_t)cumD << 16);
        WAVE_SYNC();
        int cu[REG], cd[REG];
        bool swu[REG], swd[REG];
        uint64_t SW[REG];
        cumU = 0; cumD = 0;
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            swu[t] = false; swd[t] = false; cu[t] = 0; cd[t] = 0;
            if (act[t]) {
                const uint32_t x1 = tab[first[t] + 1], xl = (last[t] == n) ? tab_n : tab[last[t]];
                cu[t] = cumU + popc64(MU[t] & lt) - (int)(x1 & 0xFFFFu);     // up-stops before me in my segment
                cd[t] = (int)(xl >> 16) - (cumD + popc64(MD[t] & le));        // down-stops behind me
                swu[t] = su[t] && cd[t] > cu[t];
                swd[t] = sd[t] && cu[t] > cd[t];
                if (su[t]) ulist[first[t] + cu[t]] = (uint16_t)(lane + 64 * t);
                if (sd[t]) dlist[first[t] + cd[t]] = (uint16_t)(lane + 64 * t);
            }
            cumU += popc64(MU[t]); cumD += popc64(MD[t]);
            SW[t] = __ballot(swu[t]);
        }
        int cumS = 0;
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            const int p = lane + 64 * t;
            if (p < n) tab2[p] = (uint16_t)(cumS + popc64(SW[t] & lt));
            cumS += popc64(SW[t]);
        }
        WAVE_SYNC();
        uint64_t CUT[REG];
#pragma unroll
        for (int t = 0; t < REG; ++t) {
            bool iscut = false;
            if (act[t]) {
                const int k = (int)((last[t] == n) ? (uint16_t)cumS : tab2[last[t]]) - (int)tab2[first[t] + 1];   // swaps of my segment
                iscut = (su[t] && cu[t] == k && cd[t] >= k) || (sd[t] && k >= 1 && cd[t] == k - 1 && cu[t] <= k);
                int dest = lane + 64 * t;
                if (swu[t]) dest = dlist[first[t] + cu[t]];
                if (swd[t]) dest = ulist[first[t] + cd[t]];
                if (swu[t] || swd[t] || moved[t]) a[dest] = xn[t];
            }
            CUT[t] = __ballot(iscut);
        }
        WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < REG; ++t) seg[t] |= CUT[t];
    }
    // final insertion sort == stable order inside every segment (all of them <= 16 elements now)
    T vv[REG];
    int rr[REG];
#pragma unroll
    for (int t = 0; t < REG; ++t) {
        const int p = lane + 64 * t;
        rr[t] = -1;
        if (p < n) {
            const T v = a[p];
            int r = first[t];
            for (int q = first[t]; q < last[t]; ++q) {
                const T x = a[q];
                r += (q != p && (less(x, v) || (!less(v, x) && q < p))) ? 1 : 0;
            }
            vv[t] = v; rr[t] = r;
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < REG; ++t) if (rr[t] >= 0) a[rr[t]] = vv[t];
    WAVE_SYNC();
    return true;
}

// ---------------------------------------------------------------------------
// Long arrays in LDS: wave_std_sort's partition loop while a segment is longer than 64*LREG elements, the level-synchronous form
// (with the loop's remaining depth limit) for every segment at or below that -- a partition of a short segment costs the whole
// wave as much as one of a long segment, and most partitions of a long array are short.  Same permutation as wave_std_sort.
// Returns false when a level-synchronous part meets the depth limit (the array is then partly partitioned: the caller starts
// over from the original order with wave_std_sort, which follows std::sort into its heap sort).
// ---------------------------------------------------------------------------
template <class T, class Less, int LREG>
__device__ bool wave_std_sort_hyb(T* a, int n, Less less, const WaveSortPtrs W_, uint32_t* tab, uint16_t* tab2, int lane, int depth_in = -1) {
    struct { uint16_t* ulist; uint16_t* dlist; uint32_t* segmark; int32_t* stk; } W = {W_.ulist, W_.dlist, W_.segmark, W_.stk};
    if (n < 2) return true;
    const uint64_t lt = lanemask_lt(lane);
    for (int q = lane; q < (n + 31) / 32 + 1; q += 64) W.segmark[q] = 0u;
    WAVE_SYNC();
    if (n <= 16) {
        if (lane == 0) gs_insertion_sort<T>(a, n, less);
        WAVE_SYNC();
        return true;
    }
    int depth0 = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
    depth0 *= 2;
    if (depth_in >= 0) depth0 = depth_in;
    int sp = 0;
    if (lane == 0) { W.stk[0] = 0; W.stk[1] = n; W.stk[2] = depth0; }
    sp = 1;
    WAVE_SYNC();
    while (sp > 0) {
        --sp;
        int first = W.stk[3 * sp], last = W.stk[3 * sp + 1], depth = W.stk[3 * sp + 2];
        bool whole = false;   // [first, last) was finished by the level-synchronous form
        while (last - first > 16) {
            if (last - first <= 64 * LREG) {
                if (!wave_lvl_sort<T, Less, LREG>(a + first, last - first, less, W.ulist + first, W.dlist + first, tab, tab2, lane, depth)) return false;
                // sorted through: any split of it is a valid set of final segments -- one mark every 16 elements
                for (int q = first + 16 * lane; q < last; q += 16 * 64) atomicOr(&W.segmark[q >> 5], 1u << (q & 31));
                WAVE_SYNC();
                whole = true;
                break;
            }
            if (depth == 0) {
                if (lane == 0) gs_heap_sort<T>(a, first, last, less);
                WAVE_SYNC();
                break;
            }
            --depth;
            if (lane == 0) {
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
            int nu = 0, nd = 0;
            for (int p0 = first + 1; p0 < last; p0 += 64) {
                int p = p0 + lane;
                bool stop = (p < last) && !less(a[p], pv);
                uint64_t mk = __ballot(stop);
                if (stop) W.ulist[nu + popc64(mk & lt)] = (uint16_t)p;
                nu += popc64(mk);
            }
            for (int p0 = last - 1; p0 > first; p0 -= 64) {
                int p = p0 - lane;
                bool stop = (p > first) && !less(pv, a[p]);
                uint64_t mk = __ballot(stop);
                if (stop) W.dlist[nd + popc64(mk & lt)] = (uint16_t)p;
                nd += popc64(mk);
            }
            if (lane == 0) W.dlist[nd] = (uint16_t)first;
            WAVE_SYNC();
            int k = 0;
            {
                const int tmax = nu < nd ? nu : nd;
                bool open = true;
                for (int t0 = 0; t0 < tmax && open; t0 += 64) {
                    int t = t0 + lane;
                    bool ok = (t < tmax) && (W.ulist[t] < W.dlist[t]);
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
                if (t < k) { xu = a[W.ulist[t]]; xd = a[W.dlist[t]]; }
                WAVE_SYNC();
                if (t < k) { a[W.ulist[t]] = xd; a[W.dlist[t]] = xu; }
                WAVE_SYNC();
            }
            int cut;
            {
                int cu = (k < nu) ? (int)W.ulist[k] : 0x7FFFFFFF;
                int cd = (k > 0) ? (int)W.dlist[k - 1] : 0x7FFFFFFF;
                cut = cu < cd ? cu : cd;
            }
            WAVE_SYNC();
            if (last - cut > 16) {
                if (lane == 0) { W.stk[3 * sp] = cut; W.stk[3 * sp + 1] = last; W.stk[3 * sp + 2] = depth; }
                ++sp;
            } else if (lane == 0) {
                atomicOr(&W.segmark[cut >> 5], 1u << (cut & 31));
            }
            WAVE_SYNC();
            last = cut;
        }
        if (!whole && lane == 0) atomicOr(&W.segmark[first >> 5], 1u << (first & 31));
        WAVE_SYNC();
    }
    // final insertion sort: lane l takes the segments that start in the 32-position words l, l+64, ...
    for (int w0 = lane; w0 * 32 < n; w0 += 64) {
        uint32_t bits = W.segmark[w0];
        while (bits) {
            const int s0 = w0 * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            int e0 = n;
            if (bits) e0 = w0 * 32 + __builtin_ctz(bits);
            else {
                for (int w1 = w0 + 1; w1 * 32 < n; ++w1) { uint32_t bb = W.segmark[w1]; if (bb) { e0 = w1 * 32 + __builtin_ctz(bb); break; } }
            }
            for (int i = s0 + 1; i < e0; ++i) {
                T v = a[i];
                int j = i - 1;
                while (j >= s0 && less(v, a[j])) { a[j + 1] = a[j]; --j; }
                a[j + 1] = v;
            }
        }
    }
    WAVE_SYNC();
    return true;
}

// ---------------------------------------------------------------------------
// DISTINCT non-negative 32-bit keys: the sorted order is unique, so std::sort's permutation is whatever sorts -- here a stable LSD
// radix sort, 8 bits per pass, by one wave in LDS (about 100 instructions per 64 keys and pass against about 350 per key and
// recursion level of the std::sort forms).  The caller knows that no key repeats (the dataflow kernel's id set).
// k0, k1: n uint32 each; i0, i1: n uint16 each; hist: 256 uint32 -- all LDS.  `ids` may be anywhere.  out(position, key, source index)
// receives the result during the last pass (it must not write into the four buffers).
// ---------------------------------------------------------------------------
template <class OutFn>
__device__ __forceinline__ void wave_radix_distinct(const int32_t* ids, const int n, const int lane, uint32_t* k0, uint32_t* k1, uint16_t* i0, uint16_t* i1,
                                                    uint32_t* hist, OutFn out) {
    const uint64_t lt = lanemask_lt(lane);
    int mx = 1;
    for (int i = lane; i < n; i += 64) { const int32_t v = ids[i]; k0[i] = (uint32_t)v; i0[i] = (uint16_t)i; mx = v > mx ? v : mx; }
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
    const int passes = (32 - __builtin_clz((unsigned)mx) + 7) / 8;
    WAVE_SYNC();
    uint32_t* ks = k0; uint32_t* kd = k1; uint16_t* is = i0; uint16_t* id = i1;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        const bool last = p == passes - 1;
        for (int q = lane; q < 256; q += 64) hist[q] = 0u;
        WAVE_SYNC();
        for (int i = lane; i < n; i += 64) atomicAdd(&hist[(ks[i] >> shift) & 255u], 1u);
        WAVE_SYNC();
        {   // exclusive scan of the 256 counts: four per lane
            const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            const uint32_t sum = c0 + c1 + c2 + c3;
            uint32_t inc = sum;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
            const uint32_t ex = inc - sum;
            WAVE_SYNC();
            hist[4 * lane] = ex; hist[4 * lane + 1] = ex + c0; hist[4 * lane + 2] = ex + c0 + c1; hist[4 * lane + 3] = ex + c0 + c1 + c2;
        }
        WAVE_SYNC();
        for (int b0 = 0; b0 < n; b0 += 64) {
            const int i = b0 + lane;
            const bool act = i < n;
            const uint32_t key = act ? ks[i] : 0u;
            const uint32_t dg = (key >> shift) & 255u;
            uint64_t m = __ballot(act);   // lanes of this turn with my digit
#pragma unroll
            for (int b = 0; b < 8; ++b) { const uint64_t bb = __ballot(act && ((dg >> b) & 1u)); m &= ((dg >> b) & 1u) ? bb : ~bb; }
            const int rank = popc64(m & lt);
            const uint32_t base = act ? hist[dg] : 0u;
            WAVE_SYNC();
            if (act && rank == 0) hist[dg] = base + (uint32_t)popc64(m);
            const uint32_t pos = base + (uint32_t)rank;
            if (act) {
                const uint16_t src = is[i];
                if (last) out((int)pos, key, (int)src);
                else { kd[pos] = key; id[pos] = src; }
            }
            WAVE_SYNC();
        }
        { uint32_t* t = ks; ks = kd; kd = t; uint16_t* u = is; is = id; id = u; }
    }
}

// ---------------------------------------------------------------------------
// Arrays too long for the LDS forms: the partitions of wave_std_sort run on the array where it lies (global memory: a pass is
// latency bound, about 1.5 us per 64 elements) only while a segment is longer than `cap` elements; a segment at or below that is
// copied into LDS, sorted through there with the hybrid form under the loop's remaining depth limit, and copied back.  Segments
// are disjoint and std::sort treats them independently, so the permutation is std::sort's.  No final pass: what the partitions
// leave at 16 elements or fewer is insertion-sorted on the spot by one lane (rare at these lengths), a staged segment comes back
// sorted.  ulist / dlist: n + 2 uint16 each beside the array (n <= 65000); stk: 3 * 48 ints, shared with the LDS sort (WL.stk is not
// used); lrec / WL / tab / tab2: the LDS form's.
// ---------------------------------------------------------------------------
template <class T, class Less, int LREG>
__device__ void wave_std_sort_staged(T* a, int n, Less less, uint16_t* ulist, uint16_t* dlist, int32_t* stk,
                                     T* lrec, const int cap, const WaveSortPtrs WL, uint32_t* tab, uint16_t* tab2, int lane) {
    if (n < 2) return;
    const uint64_t lt = lanemask_lt(lane);
    int depth0 = 0;
    for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
    depth0 *= 2;
    int sp = 0;
    if (lane == 0) { stk[0] = 0; stk[1] = n; stk[2] = depth0; }
    sp = 1;
    WAVE_SYNC();
    while (sp > 0) {
        --sp;
        int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        bool done = false;
        while (last - first > 16) {
            const int len = last - first;
            if (len <= cap) {
                // (the LDS sort's own segment stack sits behind this loop's entries: together they hold no more than the depth limit, 2 log2 n <= 32 of 48)
                const WaveSortPtrs WLi = {WL.ulist, WL.dlist, WL.segmark, stk + 3 * sp};
                for (int i = lane; i < len; i += 64) lrec[i] = a[first + i];
                WAVE_SYNC();
                if (!wave_std_sort_hyb<T, Less, LREG>(lrec, len, less, WLi, tab, tab2, lane, depth)) {
                    // a level-synchronous part met the depth limit: once more from the segment's order, following std::sort into its heap sort
                    WAVE_SYNC();
                    for (int i = lane; i < len; i += 64) lrec[i] = a[first + i];
                    WAVE_SYNC();
                    wave_std_sort<T, Less, 0>(lrec, len, less, WLi, lane, nullptr, depth);
                }
                WAVE_SYNC();
                for (int i = lane; i < len; i += 64) a[first + i] = lrec[i];
                WAVE_SYNC();
                done = true;
                break;
            }
            if (depth == 0) {
                if (lane == 0) gs_heap_sort<T>(a, first, last, less);
                WAVE_SYNC();
                done = true;
                break;
            }
            --depth;
            if (lane == 0) {   // __move_median_to_first(first, first+1, mid, last-1)
                int ia = first + 1, ib = first + len / 2, ic = last - 1;
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
            int nu = 0, nd = 0;
            for (int p0 = first + 1; p0 < last; p0 += 64) {
                int p = p0 + lane;
                bool stop = (p < last) && !less(a[p], pv);
                uint64_t mk = __ballot(stop);
                if (stop) ulist[nu + popc64(mk & lt)] = (uint16_t)p;
                nu += popc64(mk);
            }
            for (int p0 = last - 1; p0 > first; p0 -= 64) {
                int p = p0 - lane;
                bool stop = (p > first) && !less(pv, a[p]);
                uint64_t mk = __ballot(stop);
                if (stop) dlist[nd + popc64(mk & lt)] = (uint16_t)p;
                nd += popc64(mk);
            }
            if (lane == 0) dlist[nd] = (uint16_t)first;   // the pivot itself stops the down-scan
            WAVE_SYNC();
            int k = 0;
            {
                const int tmax = nu < nd ? nu : nd;
                bool open = true;
                for (int t0 = 0; t0 < tmax && open; t0 += 64) {
                    int t = t0 + lane;
                    bool ok = (t < tmax) && (ulist[t] < dlist[t]);
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
                if (t < k) { xu = a[ulist[t]]; xd = a[dlist[t]]; }
                WAVE_SYNC();
                if (t < k) { a[ulist[t]] = xd; a[dlist[t]] = xu; }
                WAVE_SYNC();
            }
            int cut;
            {
                int cu = (k < nu) ? (int)ulist[k] : 0x7FFFFFFF;
                int cd = (k > 0) ? (int)dlist[k - 1] : 0x7FFFFFFF;
                cut = cu < cd ? cu : cd;
            }
            WAVE_SYNC();
            if (last - cut > 16) {
                if (lane == 0) { stk[3 * sp] = cut; stk[3 * sp + 1] = last; stk[3 * sp + 2] = depth; }
                ++sp;
            } else if (lane == 0 && last - cut > 1) {
                gs_insertion_sort<T>(a + cut, last - cut, less);
            }
            WAVE_SYNC();
            last = cut;
        }
        if (!done && lane == 0 && last - first > 1) gs_insertion_sort<T>(a + first, last - first, less);
        WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------
// One array shared by NW waves of a workgroup, level by level (the form of rlap_kernels.hip::block_std_sort), with the
// level-synchronous form for every segment of 64*LREG elements and less: a level's segments are dealt to the waves; a long one is
// partitioned by its wave (stop lists at the segment's own offsets) and its parts are queued for the next level; a short one is
// sorted through by its wave alone (wave_lvl_sort with the loop's remaining depth limit) and marked every 16 elements.  Segments
// are disjoint and std::sort treats them independently, so the permutation is std::sort's whatever the dealing.
// ALL NW waves call this with the same arguments (workgroup barriers inside).  Returns false when a short segment met the
// depth limit (the array is partly sorted: the caller starts over from the original order with a form that follows std::sort
// into its heap sort).  segq: two queues of QCAP segments (first, last, depth); qcnt[3]: the queues' lengths and the failure flag;
// tab / tab2: NW tables of 64*LREG entries.
// ---------------------------------------------------------------------------
template <class T, class Less, int NW, int LREG>
__device__ bool block_std_sort_lvl(T* a, const int n, Less less, uint16_t* ulist, uint16_t* dlist, uint32_t* segmark, int32_t* segq, int32_t* qcnt, const int QCAP,
                                   uint32_t* tab, uint16_t* tab2, const int wave, const int lane) {
    constexpr int NTB = NW * 64;
    const int tid = wave * 64 + lane;
    const uint64_t lt = lanemask_lt(lane);
    for (int q = tid; q < (n + 31) / 32 + 1; q += NTB) segmark[q] = 0u;
    if (tid == 0) {
        int depth0 = 0;
        for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
        qcnt[0] = n > 16 ? 1 : 0; qcnt[1] = 0; qcnt[2] = 0;
        segq[0] = 0; segq[1] = n; segq[2] = 2 * depth0;
    }
    __syncthreads();
    if (n < 2) return true;
    if (n <= 16) {
        if (tid == 0) gs_insertion_sort<T>(a, n, less);
        __syncthreads();
        return true;
    }
    uint32_t* const mytab = tab + wave * 64 * LREG;
    uint16_t* const mytab2 = tab2 + wave * 64 * LREG;
    int cur = 0;
    while (true) {
        const int ncur = qcnt[cur];
        if (ncur == 0) break;
        int32_t* const qin = segq + cur * 3 * QCAP;
        int32_t* const qout = segq + (cur ^ 1) * 3 * QCAP;
        for (int sidx = wave; sidx < ncur; sidx += NW) {
            const int first = qin[3 * sidx], last = qin[3 * sidx + 1];
            int depth = qin[3 * sidx + 2];
            if (last - first <= 64 * LREG) {
                if (!wave_lvl_sort<T, Less, LREG>(a + first, last - first, less, ulist + first, dlist + first, mytab, mytab2, lane, depth)) { if (lane == 0) qcnt[2] = 1; }
                for (int q = first + 16 * lane; q < last; q += 16 * 64) atomicOr(&segmark[q >> 5], 1u << (q & 31));
                WAVE_SYNC();
                continue;
            }
            if (depth == 0) {
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
    const bool ok = qcnt[2] == 0;
    // final insertion sort: thread t takes the segments that start in the 32-position words t, t+NTB, ...
    if (ok) for (int w0 = tid; w0 * 32 < n; w0 += NTB) {
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
    return ok;
}

// ---------------------------------------------------------------------------
// The same restatement for n <= 64 with ONE ELEMENT PER LANE (key and source index in registers):
// partitions exchange elements with ds_bpermute instead of LDS round trips, the two stop lists are
// rank-indexed lane ids in a 2 x 66 int LDS scratch, the segment marks are one 64-bit mask and the
// final insertion sort is a stable rank inside each <=16-element segment.  On return lane p holds
// in `key`/`idx` an arbitrary element; *pos = final position of the element the lane holds, so the caller
// stores out[*pos] = idx.  Returns false (nothing usable) if the depth limit is hit -- the caller falls
// back to the sequential restatement (heap-sort branch).
// ---------------------------------------------------------------------------
template <bool DESC>
__device__ __forceinline__ bool wave_sort64(double& key, int& idx, const int n, const int lane, int32_t* tmp, int* pos) {
    auto less = [](double x, double y) { return DESC ? (x > y) : (x < y); };
    int32_t* tmpu = tmp;
    int32_t* tmpd = tmp + 66;
    int32_t* stk = tmp + 132;   // 3 ints per pending segment, at most 3 pending
    const uint64_t lt = lanemask_lt(lane);
    const uint64_t gt = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    uint64_t segmask = 0ull;
    bool ok_depth = true;
    if (n > 16) {
        int depth0 = 0;
        for (unsigned q = (unsigned)n; q > 1u; q >>= 1) ++depth0;
        depth0 *= 2;
        int sp = 0;
        int first = 0, last = n, depth = depth0;
        while (true) {
            while (last - first > 16) {
                if (depth == 0) { ok_depth = false; break; }
                --depth;
                // __move_median_to_first(first, first+1, mid, last-1)
                const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                const double ka = __shfl(key, ia), kb = __shfl(key, ib), kc = __shfl(key, ic);
                int pick;
                if (less(ka, kb)) {
                    if (less(kb, kc)) pick = ib;
                    else if (less(ka, kc)) pick = ic;
                    else pick = ia;
                } else if (less(ka, kc)) pick = ia;
                else if (less(kb, kc)) pick = ic;
                else pick = ib;
                {
                    const double kf = __shfl(key, first), kp = __shfl(key, pick);
                    const int xf = __shfl(idx, first), xp = __shfl(idx, pick);
                    if (lane == first) { key = kp; idx = xp; }
                    else if (lane == pick) { key = kf; idx = xf; }
                }
                const double pv = __shfl(key, first);
                const bool inr = lane > first && lane < last;
                const bool su = inr && !less(key, pv);
                const bool sd = inr && !less(pv, key);
                const uint64_t mu = __ballot(su), md = __ballot(sd);
                const int nu = popc64(mu), nd = popc64(md);
                const int ru = popc64(mu & lt), rd = popc64(md & gt);
                if (su) tmpu[ru] = lane;
                if (sd) tmpd[rd] = lane;
                if (lane == 0) tmpd[nd] = first;   // the pivot itself stops the down-scan
                WAVE_SYNC();
                const int tmax = nu < nd ? nu : nd;
                const bool okp = (lane < tmax) && (tmpu[lane] < tmpd[lane]);
                const int k = popc64(__ballot(okp));   // a prefix of the pairs
                int partner = lane;
                if (su && ru < k) partner = tmpd[ru];
                else if (sd && rd < k) partner = tmpu[rd];
                int cu = 0x7FFFFFFF, cd = 0x7FFFFFFF;
                if (k < nu) cu = tmpu[k];
                if (k > 0) cd = tmpd[k - 1];
                const int cut = cu < cd ? cu : cd;
                WAVE_SYNC();
                key = __shfl(key, partner);
                idx = __shfl(idx, partner);
                if (last - cut > 16) {
                    if (lane == 0) { stk[3 * sp] = cut; stk[3 * sp + 1] = last; stk[3 * sp + 2] = depth; }
                    ++sp;
                } else {
                    segmask |= 1ull << cut;
                }
                last = cut;
            }
            if (!ok_depth) break;
            segmask |= 1ull << first;
            if (sp == 0) break;
            --sp;
            WAVE_SYNC();
            first = stk[3 * sp]; last = stk[3 * sp + 1]; depth = stk[3 * sp + 2];
            WAVE_SYNC();
        }
    } else {
        segmask = 1ull;
    }
    if (!ok_depth) return false;
    // final insertion sort == stable sort of every marked segment: my rank inside my segment
    const int l2 = lane < n ? lane : 0;
    const uint64_t below = segmask & (lt | (1ull << l2));
    const int s0 = 63 - __builtin_clzll(below | 1ull);
    const uint64_t above = (l2 == 63) ? 0ull : (segmask >> (l2 + 1));
    const int e0 = above ? (l2 + 1 + __builtin_ctzll(above)) : n;
    int cnt = 0;
    for (int q = 0; q < 16; ++q) {
        const int p = s0 + q;
        const double kq = __shfl(key, p < 64 ? p : 63);
        const bool in = p < e0 && p != lane;
        cnt += (in && (less(kq, key) || (!less(key, kq) && p < lane))) ? 1 : 0;
    }
    *pos = s0 + cnt;
    return true;
}

// The same for groups of BC lanes (BC = 32: two candidates per wave sort side by side).  `n` and `want`
// are uniform inside a group; the groups run the introsort loop as one predicated state machine (a group
// that has nothing left idles through the others' partitions).  Scratch: 2*(BC+2)+12 ints per group.
// Returns (per group) false when the depth limit was hit.
template <bool DESC, int BC>
__device__ __forceinline__ bool group_sort(double& key, int& idx, const int n, const bool want, const int lane, int32_t* tmp_wave, int* pos) {
    auto less = [](double x, double y) { return DESC ? (x > y) : (x < y); };
    constexpr int TU = BC + 2;
    const int gl = lane & (BC - 1), gbase = lane & ~(BC - 1);
    int32_t* tmpu = tmp_wave + (lane / BC) * (2 * TU + 12);
    int32_t* tmpd = tmpu + TU;
    int32_t* stk = tmpd + TU;
    const uint64_t gmask = (BC == 64) ? ~0ull : (0xFFFFFFFFull << gbase);
    const uint64_t lt = lanemask_lt(lane) & gmask;
    const uint64_t gt = ((lane == 63) ? 0ull : (~0ull << (lane + 1))) & gmask;
    uint64_t segmask = 0ull;   // group-relative positions where a <=16-element segment starts
    bool fail = false;
    int first = 0, last = want ? n : 0, sp = 0;
    int depth = n > 1 ? 2 * (31 - __clz(n)) : 0;
    bool done = !(want && n > 16);
    if (done) segmask = 1ull;
    while (true) {
        if (!done && last - first <= 16) {   // this segment is left to the final insertion sort
            segmask |= 1ull << first;
            if (sp == 0) done = true;
            else { --sp; first = stk[3 * sp]; last = stk[3 * sp + 1]; depth = stk[3 * sp + 2]; }   // pushed segments are > 16 long
        }
        if (!done && depth == 0) { fail = true; done = true; }
        if (__ballot(!done) == 0ull) break;
        const bool act = !done;
        depth -= act ? 1 : 0;
        // __move_median_to_first(first, first+1, mid, last-1)
        const int ia = act ? first + 1 : 0, ib = act ? first + (last - first) / 2 : 0, ic = act ? last - 1 : 0, fi = act ? first : 0;
        const double ka = __shfl(key, gbase + ia), kb = __shfl(key, gbase + ib), kc = __shfl(key, gbase + ic);
        int pick;
        if (less(ka, kb)) {
            if (less(kb, kc)) pick = ib;
            else if (less(ka, kc)) pick = ic;
            else pick = ia;
        } else if (less(ka, kc)) pick = ia;
        else if (less(kb, kc)) pick = ic;
        else pick = ib;
        {
            const double kf = __shfl(key, gbase + fi), kp = __shfl(key, gbase + pick);
            const int xf = __shfl(idx, gbase + fi), xp = __shfl(idx, gbase + pick);
            if (act) {
                if (gl == first) { key = kp; idx = xp; }
                else if (gl == pick) { key = kf; idx = xf; }
            }
        }
        const double pv = __shfl(key, gbase + fi);
        const bool inr = act && gl > first && gl < last;
        const bool su = inr && !less(key, pv);
        const bool sd = inr && !less(pv, key);
        const uint64_t mu = __ballot(su) & gmask, md = __ballot(sd) & gmask;
        const int nu = popc64(mu), nd = popc64(md);
        const int ru = popc64(mu & lt), rd = popc64(md & gt);
        if (su) tmpu[ru] = gl;
        if (sd) tmpd[rd] = gl;
        if (act && gl == 0) tmpd[nd] = first;   // the pivot itself stops the down-scan
        WAVE_SYNC();
        const int tmax = nu < nd ? nu : nd;
        const bool okp = act && (gl < tmax) && (tmpu[gl] < tmpd[gl]);
        const int k = popc64(__ballot(okp) & gmask);   // a prefix of the pairs
        int partner = gl;
        if (su && ru < k) partner = tmpd[ru];
        else if (sd && rd < k) partner = tmpu[rd];
        int cu = 0x7FFFFFFF, cd = 0x7FFFFFFF;
        if (act && k < nu) cu = tmpu[k];
        if (act && k > 0) cd = tmpd[k - 1];
        const int cut = cu < cd ? cu : cd;
        WAVE_SYNC();
        key = __shfl(key, gbase + partner);
        idx = __shfl(idx, gbase + partner);
        if (act) {
            if (last - cut > 16) {
                if (gl == 0) { stk[3 * sp] = cut; stk[3 * sp + 1] = last; stk[3 * sp + 2] = depth; }
                ++sp;
            } else {
                segmask |= 1ull << cut;
            }
            last = cut;
        }
        WAVE_SYNC();
    }
    // final insertion sort == stable sort of every marked segment: my rank inside my segment
    const int l2 = gl < n ? gl : 0;
    const uint64_t below = segmask & ((2ull << l2) - 1ull);
    const int s0 = 63 - __builtin_clzll(below | 1ull);
    const uint64_t above = (l2 == 63) ? 0ull : (segmask >> (l2 + 1));
    const int e0 = above ? (l2 + 1 + __builtin_ctzll(above)) : n;
    int cnt = 0;
    for (int q = 0; q < 16; ++q) {
        const int p = s0 + q;
        const double kq = __shfl(key, gbase + (p < BC ? p : BC - 1));
        const bool in = p < e0 && p != gl;
        cnt += (in && (less(kq, key) || (!less(key, kq) && p < gl))) ? 1 : 0;
    }
    *pos = s0 + cnt;
    return !fail;
}

// wave_sort64 out of line for the round loop of the 64-slot kernels: the sort's registers are then not part of the round's pressure
struct WS64 { double key; int idx; int pos; int ok; };
template <bool DESC>
__device__ __noinline__ WS64 wave_sort64_call(double key, int idx, const int n, const int lane, int32_t* tmp) {
    ASSUME_LDS(tmp);
    int pos = lane;
    const bool ok = wave_sort64<DESC>(key, idx, n, lane, tmp, &pos);
    WS64 r; r.key = key; r.idx = idx; r.pos = pos; r.ok = ok ? 1 : 0;
    return r;
}

struct SRecLessKey { __device__ bool operator()(const SRec& x, const SRec& y) const { return x.key < y.key; } };
struct SRecGreaterKey { __device__ bool operator()(const SRec& x, const SRec& y) const { return x.key > y.key; } };

}  // namespace rlap
