// sfs_core.h -- per-read ping-pong state machine shared by the HIP kernels
// (one instance per lane) and by the host-side lane emulator used in tests.
//
// Restates PingPong::ping_pong_search (/root/reference/ping_pong.cpp:4-49) as
// a flat loop with exactly ONE rank (LF) site per iteration, so that the 64
// lanes of a wavefront -- each walking its own read, some in the backward
// phase (ping_pong.cpp:15-22), some in the forward phase (:31-37) -- stay
// convergent on the only expensive operation (the 64-byte BWT block fetch).
// Phase changes, SFS emission (:39-41) and the restart at end-1 (:47,
// overlap == -1) are handled by cheap ALU-only transitions before the step.
#pragma once
#include "fmd_layout.h"

// Streaming form of Assembler::assemble (/root/reference/assembler.cpp:34-56).
// ping_pong_search pushes SFS with strictly decreasing qs AND strictly
// decreasing end, so the sort at assembler.cpp:36 is a reversal and the chain
// rule "sfs[j-1].qs + sfs[j-1].l > sfs[j].qs" can be applied as records are
// produced: a new SFS (q,l) joins the open chain iff q + l > (qs of the
// previously produced SFS).  The chain's end is the end of its first-produced
// member (largest qs), exactly sfs[j-1].qs + sfs[j-1].l at assembler.cpp:42,50.
struct SvdssChain {
  int32_t lo;    // smallest qs in the open chain
  int32_t end;   // end (qs + l) of the chain's first-produced member
  int32_t open;  // 0/1
};

struct SvdssLane {
  // current SA interval [lo, hi) of the string being extended (W in the
  // backward phase, revcomp(W) in the forward phase)
  int64_t lo, hi;
  int32_t pos;    // read position of the last consumed symbol
  int32_t begin;  // SFS start (valid in forward phase)
  int32_t len;    // read length
  int32_t dir;    // 0 backward, 1 forward
  int32_t c;      // symbol to prepend in the pending LF step
  int32_t n_sfs;  // records produced so far (after optional assembly)
  int64_t n_ext;  // LF steps == rb3_fmd_extend calls of the reference
  SvdssChain chain;
};

template <class Sym>
SVDSS_HD void svdss_lane_init(SvdssLane& s, const SvdssDevIndex& ix, Sym&& sym, int len) {
  s.len = len;
  s.pos = len - 1;
  s.begin = 0;
  s.dir = 0;
  s.n_sfs = 0;
  s.n_ext = 0;
  s.chain.open = 0;
  s.chain.lo = 0;
  s.chain.end = 0;
  s.c = len > 0 ? sym(len - 1) : 0;
  s.lo = svdss_acc(ix, s.c);       // rb3_fmd_set_intv, ping_pong.cpp:12
  s.hi = svdss_acc(ix, s.c + 1);
}

// Emits one SFS through the optional streaming assembler.
template <class Emit>
SVDSS_HD void svdss_lane_emit(SvdssLane& s, int qs, int l, bool assemble, Emit&& emit) {
  if (!assemble) {
    emit(s.n_sfs++, qs, l);
    return;
  }
  if (s.chain.open) {
    if (qs + l > s.chain.lo) {  // overlaps the previously produced SFS: extend chain
      s.chain.lo = qs;
      return;
    }
    emit(s.n_sfs++, s.chain.lo, s.chain.end - s.chain.lo);
  }
  s.chain.open = 1;
  s.chain.lo = qs;
  s.chain.end = qs + l;
}

template <class Emit>
SVDSS_HD void svdss_lane_flush(SvdssLane& s, bool assemble, Emit&& emit) {
  if (assemble && s.chain.open) {
    emit(s.n_sfs++, s.chain.lo, s.chain.end - s.chain.lo);
    s.chain.open = 0;
  }
}

// Runs the ALU-only transitions until either an LF step is pending (returns
// true; s.c holds the symbol, s.pos already points at it) or the read is
// finished (returns false).
template <class Sym, class Emit>
SVDSS_HD bool svdss_lane_resolve(SvdssLane& s, const SvdssDevIndex& ix, Sym&& sym, bool assemble, Emit&& emit) {
  if (s.len <= 0) return false;
  for (;;) {
    const bool nonempty = s.hi > s.lo;
    if (s.dir == 0) {
      if (nonempty && s.pos > 0) {            // ping_pong.cpp:15-16
        --s.pos;
        s.c = sym(s.pos);                     // :21 ik = ok[P[begin]]
        return true;
      }
      if (s.pos == 0 && nonempty) return false;  // :24 whole prefix matched
      s.begin = s.pos;                        // :28 end = begin
      s.dir = 1;
      s.c = svdss_comp(sym(s.pos));           // :30 set_intv(P[end]) on the revcomp strand
      s.lo = svdss_acc(ix, s.c);
      s.hi = svdss_acc(ix, s.c + 1);
    } else {
      if (nonempty) {                         // :31-32
        ++s.pos;
        s.c = svdss_comp(s.pos < s.len ? sym(s.pos) : 0);  // :36 ok[comp(P[end])], P[l] == 0
        return true;
      }
      svdss_lane_emit(s, s.begin, s.pos - s.begin + 1, assemble, emit);  // :38-41
      if (s.begin == 0) return false;         // :42
      s.pos = s.pos - 1;                      // :47 begin = end + overlap, overlap == -1
      s.dir = 0;
      s.c = sym(s.pos);                       // :12 set_intv(P[begin])
      s.lo = svdss_acc(ix, s.c);
      s.hi = svdss_acc(ix, s.c + 1);
    }
  }
}

// One LF step with the block(s) already fetched: qlo holds block(lo>>7),
// qhi holds block(hi>>7) (may alias qlo).
SVDSS_HD void svdss_lane_step(SvdssLane& s, const SvdssDevIndex& ix, const svdss_u4 qlo[4], const svdss_u4 qhi[4]) {
  const int64_t a = svdss_acc(ix, s.c);
  s.lo = a + svdss_rank_in_block(ix, qlo, s.c, s.lo);
  s.hi = a + svdss_rank_in_block(ix, qhi, s.c, s.hi);
  ++s.n_ext;
}
