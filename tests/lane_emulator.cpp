// tests/lane_emulator.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the exact per-lane code of the HIP search kernel (sfs_core2.h,
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
#include "../svdss_amd/csrc/sym_window.h"

// ---------------------------------------------------------------------------
// v2 state machine (sfs_core2.h): k-mer table + LF + unique-match TEXT mode.
#include "../svdss_amd/csrc/sfs_core2.h"

namespace {
struct U4u { uint8_t b[16]; };

template <class P>
int64_t emu2_run(const svdss_index* ix, const uint8_t* reads_padded, const int64_t* offsets,
                 int64_t n_reads, int64_t total_syms, int assemble, int K, int use_text,
                 int64_t* counts, int32_t* qs, int32_t* len, int64_t cap_total, int64_t* n_ext,
                 int64_t* op_counts, int n_seg, int64_t* seg_stats) {
  SvdssDevIndex v;
  v.blocks = ix->blocks.data();
  v.dollar = ix->dollar.data();
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = K;
  v.bs_after = getenv("SVDSS_BS_AFTER") ? atoi(getenv("SVDSS_BS_AFTER")) : SV_BS_AFTER_DEFAULT;
  v.pad_ = 0;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  std::vector<uint8_t> text((size_t)ix->n + 128 + 16, 0);
  memcpy(text.data() + 64, ix->text.data(), (size_t)ix->n);
  v.text = text.data() + 64;
  v.sa = (use_text & 1) ? (ix->sa64.empty() ? (const void*)ix->sa32.data() : (const void*)ix->sa64.data()) : nullptr;
  const bool use_set = (use_text & 2) != 0;   // bit 1: SET mode (2-4 occurrences followed in the text)
  // bit 2: BS mode (deep intervals finished by binary search of the suffix array); the text positions of the '$', sorted
  std::vector<int64_t> dsorted;
  if ((use_text & 4) && v.sa) {
    for (int64_t i = 0; i < ix->acc[1]; ++i) dsorted.push_back((int64_t)((const P*)v.sa)[i]);
    std::sort(dsorted.begin(), dsorted.end());
  }
  const bool use_bs = !dsorted.empty() && (int)dsorted.size() <= SV_BS_MAX_DOLLAR;
  std::vector<SvdssTabEntry> table;
  if (K > 0) {
    table.resize((size_t)1 << (2 * K));
    for (uint64_t key = 0; key < table.size(); ++key)
      sv_table_entry<P>(v, (uint32_t)key, K, table[key].lo, table[key].info);
  }
  v.table = K > 0 ? table.data() : nullptr;
  const int64_t max_chunk = ((total_syms + 15) >> 4) - 1;
  int64_t total = 0;
  struct Rec { int32_t qs, len, ext; };
  // one lane: the chain that starts with a fresh backward phase at start_pos
  auto run_lane = [&](int64_t off, int64_t l, int start_pos, int stop_lo, bool asm_, std::vector<Rec>& recs,
                      int32_t& ext_total, bool& complete, const std::vector<Rec>* left = nullptr) {
    int32_t nb_cur = 0;
    int64_t set_mem[SV_SET_MAX];
    const SvSet ts{set_mem, 1};
    uint32_t ring_mem[16];
    memset(ring_mem, 0xee, sizeof ring_mem);
    SvRing g{ring_mem, 1};
    SvLane<P> st;
    auto emit = [&](int32_t idx, int32_t q, int32_t ln) {
      if ((int64_t)recs.size() != idx) __builtin_trap();
      recs.push_back({q, ln, st.n_ext - (st.pos - st.begin)});
    };
    sv_lane_init(st, (int32_t)l, start_pos, stop_lo);
    for (int64_t guard = 0;; ++guard) {
      if (guard > 400 * l + 10000) { fprintf(stderr, "emu2: no termination off=%ld l=%ld start=%d stop=%d pos=%d begin=%d mode=%d lo=%ld hi=%ld wrel=%d nsfs=%d\n", (long)off, (long)l, start_pos, stop_lo, st.pos, st.begin, st.mode, (long)st.lo, (long)st.hi, st.wrel, st.n_sfs); abort(); }
      SvOp o = sv_decide(st, v, g, off, asm_, emit, left != nullptr, use_set, use_bs);   // (BS off: sv_decide_flat)
      if (op_counts) op_counts[o.op]++;
      if (o.op == SV_OP_DONE) break;
      if (o.op == SV_OP_TEXT_SLOW) { sv_apply_text_slow(st, v.text, reads_padded, off); continue; }
      if (o.op == SV_OP_BS_TEXT_SLOW) { sv_apply_bs_text_slow(st, v, v.text, reads_padded, off); continue; }
      if (o.op == SV_OP_BS_ORD) {
        const int64_t rp = (int64_t)st.pos + v.k - 1 - st.bs_m;
        sv_apply_bs_ord(st, v, (int)reads_padded[off + rp], (int)v.text[st.tdelta + rp]);
        continue;
      }
      if (o.op == SV_OP_BS_SA) { sv_apply_bs_sa(st, v, (int64_t)((const P*)v.sa)[o.a], dsorted.data(), (int)dsorted.size()); continue; }
      if (o.op == SV_OP_BS_TEXT) {
        svdss_u4 ta[4], rb[4];
        const int cp = st.pos + v.k - st.bs_m;
        memcpy(ta, v.text + st.tdelta + cp - 64, 64);
        memcpy(rb, reads_padded + off + cp - 64, 64);
        sv_apply_bs_text(st, v, ta, rb);
        continue;
      }
      if (o.op == SV_OP_SA_SET) {
        int64_t tp[SV_SET_MAX];
        const int n_occ = (int)(st.hi - st.lo);
        for (int i = 0; i < SV_SET_MAX; ++i) tp[i] = (int64_t)((const P*)v.sa)[(int64_t)st.lo + (i < n_occ ? i : 0)];
        sv_apply_sa_set(st, ts, tp);
        continue;
      }
      if (o.op == SV_OP_SET) {
        svdss_u4 tw[SV_SET_MAX], rb;
        const int alive = (st.mode >> SV_SET_SHIFT) & ((1 << SV_SET_MAX) - 1);
        memset(tw, 0, sizeof tw);
        for (int i = 0; i < SV_SET_MAX; ++i)
          if ((alive >> i) & 1) memcpy(&tw[i], v.text + set_mem[i] + st.pos - SV_SET_WIN, 16);
        memcpy(&rb, reads_padded + off + st.pos - SV_SET_WIN, 16);
        sv_apply_set(st, ts, tw, rb);
        continue;
      }
      if (o.op == SV_OP_PEEK) {   // the neighbour's chain is complete here (segments run left to right)
        int32_t q[SV_PEEK_RECS];
        bool written[SV_PEEK_RECS];
        for (int i = 0; i < SV_PEEK_RECS; ++i) {
          written[i] = (size_t)(nb_cur + i) < left->size();
          q[i] = written[i] ? (*left)[(size_t)(nb_cur + i)].qs : 0;
        }
        sv_apply_peek(st, q, written, nb_cur, 1 << 30);
        continue;
      }
      svdss_u4 A[4], B[4];
      if (o.op == SV_OP_LF) {
        const int64_t blo = (int64_t)st.lo >> SVDSS_BLOCK_SHIFT, bhi = (int64_t)st.hi >> SVDSS_BLOCK_SHIFT;
        for (int j = 0; j < 4; ++j) A[j] = v.blocks[4 * blo + j];
        if (bhi != blo) for (int j = 0; j < 4; ++j) B[j] = v.blocks[4 * bhi + j];
        else memset(B, 0xcd, sizeof B);  // must not be used
        sv_apply_lf(st, v, A, B, bhi == blo);
      } else if (o.op == SV_OP_TABLE) {
        const SvdssTabEntry e = v.table[o.a];
        sv_apply_table(st, v, e.lo, e.info, g, off, use_set && off >= 64, use_bs && off >= 64);
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
    sv_flush(st, asm_, emit);
    ext_total = st.n_ext;
    complete = !(st.mode & SV_M_PARTIAL);
  };
  for (int64_t r = 0; r < n_reads; ++r) {
    const int64_t off = offsets[r];
    const int64_t l = offsets[r + 1] - off;
    std::vector<std::pair<int32_t, int32_t>> recs;
    int64_t ext_read = 0;
    int C = n_seg > 1 ? (int)std::min<int64_t>(n_seg, std::max<int64_t>(1, l >> 8)) : 1;
    bool stitched = false;
    if (C > 1) {
      // segmented: raw chains per segment, stitch, then (optionally) the streaming assembler
      std::vector<std::vector<Rec>> seg((size_t)C);
      std::vector<SvSegInfo> info((size_t)C);
      std::vector<int32_t> seg_lo((size_t)C), tlo((size_t)C), thi((size_t)C);
      for (int j = 0; j < C; ++j) {
        const int32_t sl = (int32_t)(l / C);
        seg_lo[(size_t)j] = j * sl;
        const int start = j == C - 1 ? (int)l - 1 : (j + 1) * sl - 1;
        int32_t et; bool comp;
        run_lane(off, l, start, seg_lo[(size_t)j], false, seg[(size_t)j], et, comp,
                 j > 0 ? &seg[(size_t)(j - 1)] : nullptr);
        info[(size_t)j] = {(int32_t)seg[(size_t)j].size(), 1 << 30, et, comp ? 1 : 0};
      }
      auto get = [&](int sg, int32_t i, int32_t& q, int32_t& e) { q = seg[(size_t)sg][(size_t)i].qs; e = seg[(size_t)sg][(size_t)i].ext; };
      if (sv_stitch(C, info.data(), seg_lo.data(), get, tlo.data(), thi.data(), &ext_read)) {
        stitched = true;
        if (seg_stats) seg_stats[0]++;
        SvLane<P> as;     // only the assembler fields are used
        sv_lane_init(as, 0);
        auto emit2 = [&](int32_t, int32_t q, int32_t ln) { recs.emplace_back(q, ln); };
        for (int j = C - 1; j >= 0; --j)
          for (int32_t i = tlo[(size_t)j]; i < thi[(size_t)j]; ++i)
            sv_emit(as, seg[(size_t)j][(size_t)i].qs, seg[(size_t)j][(size_t)i].len, assemble != 0, emit2);
        sv_flush(as, assemble != 0, emit2);
      } else if (seg_stats) seg_stats[1]++;
    }
    if (!stitched) {
      std::vector<Rec> one;
      int32_t et; bool comp;
      run_lane(off, l, -1, 0, assemble != 0, one, et, comp);
      if (!comp) __builtin_trap();
      for (auto& x : one) recs.emplace_back(x.qs, x.len);
      ext_read = et;
    }
    if (assemble) std::reverse(recs.begin(), recs.end());
    if (total + (int64_t)recs.size() > cap_total) return -1;
    for (auto& rc : recs) { qs[total] = rc.first; len[total] = rc.second; ++total; }
    counts[r] = (int64_t)recs.size();
    n_ext[r] = ext_read;
  }
  return total;
}
}  // namespace

// the k-mer table of order K as the kernel builds it (sv_table_entry), for tests of its fields
extern "C" int emu_table(const svdss_index* ix, int K, uint64_t* out_lo, uint64_t* out_info) {
  SvdssDevIndex v;
  v.blocks = ix->blocks.data();
  v.dollar = ix->dollar.data();
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = K;
  v.bs_after = getenv("SVDSS_BS_AFTER") ? atoi(getenv("SVDSS_BS_AFTER")) : SV_BS_AFTER_DEFAULT;
  v.pad_ = 0;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  v.text = ix->text.data();
  v.sa = ix->sa64.empty() ? (const void*)ix->sa32.data() : (const void*)ix->sa64.data();
  v.table = nullptr;
  const uint64_t n = (uint64_t)1 << (2 * K);
  for (uint64_t key = 0; key < n; ++key) {
    if (ix->sa64.empty()) sv_table_entry<uint32_t>(v, (uint32_t)key, K, out_lo[key], out_info[key]);
    else sv_table_entry<uint64_t>(v, (uint32_t)key, K, out_lo[key], out_info[key]);
  }
  return 0;
}

extern "C" int64_t emu_search2(const svdss_index* ix, const uint8_t* reads_padded,
                               const int64_t* offsets, int64_t n_reads, int64_t total_syms,
                               int assemble, int K, int use_text, int64_t* counts, int32_t* qs,
                               int32_t* len, int64_t cap_total, int64_t* n_ext, int64_t* op_counts,
                               int n_seg, int64_t* seg_stats) {
  if (ix->sa64.empty())
    return emu2_run<uint32_t>(ix, reads_padded, offsets, n_reads, total_syms, assemble, K, use_text,
                              counts, qs, len, cap_total, n_ext, op_counts, n_seg, seg_stats);
  return emu2_run<uint64_t>(ix, reads_padded, offsets, n_reads, total_syms, assemble, K, use_text,
                            counts, qs, len, cap_total, n_ext, op_counts, n_seg, seg_stats);
}
