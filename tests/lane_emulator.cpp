// tests/lane_emulator.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the exact per-lane code of the HIP search kernel (sfs_core.h,
// sym_window.h, fmd_layout.h) on the CPU, one "lane" at a time, so the state
// machine, the read window and the streaming assembler can be checked against
// the oracle in this GPU-less container before the kernel goes to an MI355X.
// It is never loaded by the product.
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
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
  v.k = 0; v.text = nullptr; v.sa = nullptr; v.table = nullptr;
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

// ---------------------------------------------------------------------------
// v2 state machine (sfs_core2.h): k-mer table + LF + unique-match TEXT mode.
#include "../svdss_amd/csrc/sfs_core2.h"

namespace {
struct U4u { uint8_t b[16]; };

template <class P>
int64_t emu2_run(const svdss_index* ix, const uint8_t* reads_padded, const int64_t* offsets,
                 int64_t n_reads, int64_t total_syms, int assemble, int K, int use_text,
                 int64_t* counts, int32_t* qs, int32_t* len, int64_t cap_total, int64_t* n_ext,
                 int64_t* op_counts) {
  SvdssDevIndex v;
  v.blocks = ix->blocks.data();
  v.dollar = ix->dollar.data();
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = K;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  std::vector<uint8_t> text((size_t)ix->n + 128 + 16, 0);
  memcpy(text.data() + 64, ix->text.data(), (size_t)ix->n);
  v.text = text.data() + 64;
  v.sa = use_text ? (ix->sa64.empty() ? (const void*)ix->sa32.data() : (const void*)ix->sa64.data()) : nullptr;
  std::vector<SvdssTabEntry> table;
  if (K > 0) {
    table.resize((size_t)1 << (2 * K));
    for (uint64_t key = 0; key < table.size(); ++key)
      sv_table_entry<P>(v, (uint32_t)key, K, table[key].lo, table[key].info);
  }
  v.table = K > 0 ? table.data() : nullptr;
  const int64_t max_chunk = ((total_syms + 15) >> 4) - 1;
  int64_t total = 0;
  for (int64_t r = 0; r < n_reads; ++r) {
    uint32_t ring_mem[16];
    memset(ring_mem, 0xee, sizeof ring_mem);
    SvRing g{ring_mem, 1};
    SvLane<P> st;
    const int64_t off = offsets[r];
    const int64_t l = offsets[r + 1] - off;
    std::vector<std::pair<int32_t, int32_t>> recs;
    auto emit = [&](int32_t idx, int32_t q, int32_t ln) {
      if ((int64_t)recs.size() != idx) __builtin_trap();
      recs.emplace_back(q, ln);
    };
    sv_lane_init(st, (int32_t)l);
    for (int64_t guard = 0;; ++guard) {
      if (guard > 8 * l + 1000) { fprintf(stderr, "emu2: no termination read %ld op=%d pos=%d mode=%d lo=%ld hi=%ld wrel=%d\n", (long)r, -1, st.pos, st.mode, (long)st.lo, (long)st.hi, st.wrel); abort(); }
      SvOp o = sv_decide(st, v, g, off, assemble != 0, emit);
      if (op_counts) op_counts[o.op]++;
      if (getenv("EMU_TRACE") && guard < 80)
        fprintf(stderr, "r%ld it%ld op=%d a=%ld pos=%d begin=%d mode=%d lo=%ld hi=%ld wrel=%d ext=%d\n", (long)r,
                (long)guard, o.op, (long)o.a, st.pos, st.begin, st.mode, (long)st.lo, (long)st.hi, st.wrel, st.n_ext);
      if (o.op == SV_OP_DONE) break;
      if (o.op == SV_OP_TEXT_SLOW) { sv_apply_text_slow(st, v.text, reads_padded, off); continue; }
      svdss_u4 A[4], B[4];
      if (o.op == SV_OP_LF) {
        const int64_t blo = (int64_t)st.lo >> SVDSS_BLOCK_SHIFT, bhi = (int64_t)st.hi >> SVDSS_BLOCK_SHIFT;
        for (int j = 0; j < 4; ++j) A[j] = v.blocks[4 * blo + j];
        if (bhi != blo) for (int j = 0; j < 4; ++j) B[j] = v.blocks[4 * bhi + j];
        else memset(B, 0xcd, sizeof B);  // must not be used
        sv_apply_lf(st, v, A, B, bhi == blo);
      } else if (o.op == SV_OP_TABLE) {
        const SvdssTabEntry e = v.table[o.a];
        sv_apply_table(st, v, e.lo, e.info);
      } else if (o.op == SV_OP_SA) {
        sv_apply_sa(st, (int64_t)((const P*)v.sa)[o.a]);
      } else if (o.op == SV_OP_TEXT) {
        memcpy(A, v.text + st.tdelta + st.pos - 64, 64);
        memcpy(B, reads_padded + off + st.pos - 64, 64);
        sv_apply_text(st, A, B);
      } else if (o.op == SV_OP_FILL) {
        int64_t c0 = o.a;
        if (c0 > max_chunk - 3) c0 = max_chunk - 3;
        if (c0 < 0) c0 = 0;
        memcpy(B, reads_padded + 16 * c0, 64);
        sv_ring_fill(g, c0, B);
        st.wrel = (int32_t)(16 * c0 - off);
      }
    }
    sv_flush(st, assemble != 0, emit);
    if (assemble) std::reverse(recs.begin(), recs.end());
    if (total + (int64_t)recs.size() > cap_total) return -1;
    for (auto& rc : recs) { qs[total] = rc.first; len[total] = rc.second; ++total; }
    counts[r] = (int64_t)recs.size();
    n_ext[r] = st.n_ext;
  }
  return total;
}
}  // namespace

extern "C" int64_t emu_search2(const svdss_index* ix, const uint8_t* reads_padded,
                               const int64_t* offsets, int64_t n_reads, int64_t total_syms,
                               int assemble, int K, int use_text, int64_t* counts, int32_t* qs,
                               int32_t* len, int64_t cap_total, int64_t* n_ext, int64_t* op_counts) {
  if (ix->sa64.empty())
    return emu2_run<uint32_t>(ix, reads_padded, offsets, n_reads, total_syms, assemble, K, use_text,
                              counts, qs, len, cap_total, n_ext, op_counts);
  return emu2_run<uint64_t>(ix, reads_padded, offsets, n_reads, total_syms, assemble, K, use_text,
                            counts, qs, len, cap_total, n_ext, op_counts);
}
