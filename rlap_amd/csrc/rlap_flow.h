// rlap_flow.h -- the multi-CU ("dataflow") form of the exact elimination for o_v = random, shared by the HIP kernel
// (rlap_flow.hip) and the host mirror (tests/csrc/host_mirror.cc).
//
// Reference behaviour restated (paths relative to /root/reference): rlap/csrc/preconditioner.cc:713-787 (the elimination loop of
// RandomPreconditioner), :616-676 (getColumnLength, compressColumn), :588-613 (the order is known up front: node_id popped from the back).
//
// The sequential loop is a chain through three things: the columns (fill-in), the single uniform stream (:721) and the list
// order inside a column (:409-411 push at the head).  Here every position of the order is taken by one wave, on any compute
// unit, and the chain is kept by data, not by program order:
//   * pend[v] (VRec::key) = live entries (y -> v) in columns of vertices y that come EARLIER in the order and have not committed.
//     v's column is touched only by eliminations of its current neighbours (the twin is rewritten in place :404-406, an entry is
//     pushed :409-411, a merged duplicate's twin dies :655), so once pend[v] == 0 the column is final: one gather, no retry.
//     An eliminating vertex adds 1 for every live entry it creates that points from an earlier to a later vertex, makes all
//     its writes, waits for them, and only then takes its own entries off its neighbours' counters.
//   * the uniforms of position p start at D(p) = sum over earlier positions of max(len - 1, 0) (:729 consumes them in order):
//     a decoupled look-back over published counts (lb[]).
//   * pushes of concurrent eliminations reach a column out of order.  Every appended entry carries the index of the uniform
//     that created it (atag[]): unique, increasing in the sequential order.  The list order matters only where ids repeat
//     (std::sort by id :641-644 is a function of the input permutation only among equal keys ... and, above 16 entries, of the
//     whole permutation once any keys are equal), so a gather that finds equal ids puts the appended entries into tag order
//     (newest first) and sorts again; the columns that survive are put into tag order once, after the elimination.
// Chunks keep the layout of rlap_core.h (header slot + entries) so that the output pass reads them unchanged; the header links
// and VRec::app_chunk are filled in afterwards (flow_finish_vertex), the chunk bases live in a per-vertex directory meanwhile.
#pragma once
#include "rlap_core.h"

namespace rlap {

constexpr int FDIR = 8;                   // directory words per vertex: chunks 0..FDIR-2 inline, word FDIR-1 = slot of the overflow directory
constexpr int FDIR_OVF = 24;              // chunks FDIR-1 .. FDIR-2+FDIR_OVF (covers 2^31 appended entries)
constexpr int FDIR_OVF_SLOTS = (FDIR_OVF * 4 + 15) / 16;   // pool slots an overflow directory takes
constexpr int32_t FD_EMPTY = -1, FD_BUSY = -2, FD_FAIL = -3;
constexpr int32_t FPOS_NONE = 0x7FFFFFFF;  // VRec::pqpos of a vertex that is not eliminated

// look-back word of a position: bit 63 = count published, bit 62 = the value is the INCLUSIVE prefix (else the position's own count)
constexpr unsigned long long LB_VALID = 1ull << 63, LB_PREFIX = 1ull << 62, LB_MASK = (1ull << 62) - 1ull;

// control words (index into FlowArrays::ctrl): next claim, abort status, long-column scratch top (entries), long columns met, committed positions,
// very long columns being worked on (no stall is declared meanwhile)
enum { FC_CLAIM = 0, FC_ABORT = 32, FC_SCR = 64, FC_LONG = 96, FC_PROGRESS = 128, FC_HEAVY = 160, FC_WORDS = 192 };

struct FlowArrays {
    int32_t* cdir;              // [N * FDIR]
    int32_t* atag;              // [slot_cap] uniform index of the draw that created an appended entry
    unsigned long long* lb;     // [Q] look-back words; a graph's positions are preceded by one sentinel (prefix 0)
    const int32_t* qv;          // [Q] vertex of a position, -1 for a sentinel
    const int32_t* qg;          // [Q] graph of a position (sentinels too)
    int32_t* ctrl;              // control words, each on a 128-byte line of its own (FC_*): all waves hammer the first two
    int32_t Q;                  // positions + sentinels
    const uint32_t* qorder;     // claim order (batches: positions of all graphs interleaved, [Qclaim]); nullptr = 0, 1, 2, ... over lb[]
    int32_t Qclaim;             // claims to hand out
};

RLAP_HD int32_t* flow_dir_word(const FlowArrays& F, int32_t v, int c) { return F.cdir + (int64_t)v * FDIR + c; }

// Header links and VRec::app_chunk of one column, from its directory (after the elimination; one thread per vertex).
RLAP_HD void flow_finish_vertex(const Arrays& A, const FlowArrays& F, int32_t v) {
    const int32_t acnt = A.vr[v].app_cnt;
    if (acnt <= 0) { A.vr[v].app_chunk = -1; return; }
    const int ct = chunk_of(acnt - 1);
    int32_t prev = -1;
    const int32_t* ovf = nullptr;
    if (ct >= FDIR - 1) ovf = reinterpret_cast<const int32_t*>(A.e + F.cdir[(int64_t)v * FDIR + FDIR - 1]);
    for (int c = 0; c <= ct; ++c) {
        const int32_t base = c < FDIR - 1 ? F.cdir[(int64_t)v * FDIR + c] : ovf[c - (FDIR - 1)];
        A.e[base].nbr = prev;
        prev = base;
    }
    A.vr[v].app_chunk = prev;
}

}  // namespace rlap
