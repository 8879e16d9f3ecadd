// tests/lane_emulator.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the exact per-lane code of the HIP search kernel (sfs_core.h,
// sym_window.h, fmd_layout.h) on the CPU, one "lane" at a time, so the state
// machine, the read window and the streaming assembler can be checked against
// the oracle in this GPU-less container before the kernel goes to an MI355X.
// It is never loaded by the product.
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../svdss_amd/csrc/index_host.h"
#include "../svdss_amd/csrc/sfs_core.h"
#include "../svdss_amd/csrc/sym_window.h"

extern "C" int64_t emu_search(const svdss_index* ix, const uint8_t* reads_padded,
                              const int64_t* offsets, int64_t n_reads, int64_t total_syms,
                              int assemble, int64_t* counts, int32_t* qs, int32_t* len,
                              int64_t cap_total, int64_t* n_ext) {
  SvdssDevIndex v;
  v.blocks = ix->blocks.data();
  v.dollar = ix->dollar.data();
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.pad = 0;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  int64_t total = 0;
  for (int64_t r = 0; r < n_reads; ++r) {
    SvdssLane st;
    SvdssSymWindow win;
    SvdssReadView rv;
    rv.chunks = (const svdss_u4*)reads_padded;
    rv.max_chunk = total_syms > 0 ? ((total_syms + 15) >> 4) - 1 : 0;
    rv.off = offsets[r];
    const int64_t l = offsets[r + 1] - offsets[r];
    std::vector<std::pair<int32_t, int32_t>> recs;
    auto sym = [&](int32_t pos) -> int { return svdss_window_sym(win, rv, pos, st.dir); };
    auto emit = [&](int32_t idx, int32_t q, int32_t ln) {
      if ((int64_t)recs.size() != idx) __builtin_trap();
      recs.emplace_back(q, ln);
    };
    svdss_window_reset(win);
    st.dir = 0;
    svdss_lane_init(st, v, sym, (int32_t)l);
    while (svdss_lane_resolve(st, v, sym, assemble != 0, emit)) {
      const int32_t np = st.dir ? st.pos + 1 : st.pos - 1;
      if (np >= 0 && np < st.len) svdss_window_prefetch(win, rv, np);
      const int64_t blo = st.lo >> SVDSS_BLOCK_SHIFT, bhi = st.hi >> SVDSS_BLOCK_SHIFT;
      svdss_u4 ql[4], qh[4];
      for (int j = 0; j < 4; ++j) ql[j] = v.blocks[4 * blo + j];
      for (int j = 0; j < 4; ++j) qh[j] = v.blocks[4 * bhi + j];
      svdss_lane_step(st, v, ql, qh);
    }
    svdss_lane_flush(st, assemble != 0, emit);
    if (assemble) std::reverse(recs.begin(), recs.end());
    if (total + (int64_t)recs.size() > cap_total) return -1;
    for (auto& rc : recs) { qs[total] = rc.first; len[total] = rc.second; ++total; }
    counts[r] = (int64_t)recs.size();
    n_ext[r] = st.n_ext;
  }
  return total;
}
