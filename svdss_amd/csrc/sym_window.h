// sym_window.h -- per-lane register window over the read being searched.
//
// The reference reads P[begin] / P[end] one byte at a time
// (/root/reference/ping_pong.cpp:12,21,30,36).  On the GPU a byte load per
// step would double the number of memory requests of the search (the L1 is
// thrashed by the BWT block stream), so each lane keeps two adjacent 16-byte
// chunks of its read in VGPRs and prefetches the next chunk in its direction
// of travel together with the BWT block loads of the current step, so the two
// latencies overlap.  The restart at end-1 (ping_pong.cpp:47) always lands in
// the chunk the lane just left, which is why the window is two chunks wide.
#pragma once
#include "fmd_layout.h"

struct SvdssSymWindow {
  svdss_u4 A, B, T;  // chunks ia, ia+1 and the prefetched chunk it
  int64_t ia;
  int64_t it;        // -1: T empty
};

struct SvdssReadView {
  const svdss_u4* chunks;  // concatenated read symbols viewed as 16-byte chunks
  int64_t max_chunk;       // last chunk index that may be loaded
  int64_t off;             // absolute position of the read's first symbol
};

SVDSS_HD svdss_u4 svdss_load_chunk(const SvdssReadView& rv, int64_t ch) {
  ch = ch < 0 ? 0 : (ch > rv.max_chunk ? rv.max_chunk : ch);
  return rv.chunks[ch];
}

SVDSS_HD uint32_t svdss_chunk_dword(const svdss_u4& w, int d) {
  return d == 0 ? w.x : d == 1 ? w.y : d == 2 ? w.z : w.w;
}

SVDSS_HD void svdss_window_reset(SvdssSymWindow& w) {
  w.ia = -4;  // nothing cached
  w.it = -1;
}

// symbol at read position pos; dir is the lane's current direction of travel
SVDSS_HD int svdss_window_sym(SvdssSymWindow& w, const SvdssReadView& rv, int32_t pos, int dir) {
  const int64_t a = rv.off + pos;
  const int64_t ch = a >> 4;
  if (ch != w.ia && ch != w.ia + 1) {
    if (ch == w.it && ch == w.ia - 1) {
      w.B = w.A; w.A = w.T; w.ia = ch;
    } else if (ch == w.it && ch == w.ia + 2) {
      w.A = w.B; w.B = w.T; w.ia = ch - 1;
    } else {
      w.ia = (dir && ch > 0) ? ch - 1 : ch;
      w.A = svdss_load_chunk(rv, w.ia);
      w.B = svdss_load_chunk(rv, w.ia + 1);
    }
    w.it = -1;
  }
  // select by value (a reference select between A and B would force both into scratch memory)
  const int byte = (int)(a & 15);
  const uint32_t va = svdss_chunk_dword(w.A, byte >> 2), vb = svdss_chunk_dword(w.B, byte >> 2);
  const uint32_t v = (ch == w.ia) ? va : vb;
  return (int)((v >> ((byte & 3) * 8)) & 0xffu);
}

// issue the load of the chunk holding read position pos if it is not cached
SVDSS_HD void svdss_window_prefetch(SvdssSymWindow& w, const SvdssReadView& rv, int32_t pos) {
  const int64_t ch = (rv.off + pos) >> 4;
  if (ch != w.ia && ch != w.ia + 1 && ch != w.it) {
    w.T = svdss_load_chunk(rv, ch);
    w.it = ch;
  }
}
