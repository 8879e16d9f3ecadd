// poa_quad_core.h -- POA consensus with SEVERAL SUB-CLUSTERS PER WAVEFRONT (device code of poa_quad.hip).
//
// Same specification as poa_wave.hip / poa.hip / oracle/svdss_oracle_poa.c, bit for bit (caller.cpp:257-308).
//
// Why: poa_wave.hip gives every sub-cluster a whole wavefront and one DP row per step.  A banded row is 35-80
// columns, so the row's arithmetic is a few dozen vector instructions -- and around them ~100 scalar instructions
// of per-row bookkeeping (band, descriptors, ring slots, branches), three 6-step DPP scans and a scalar round trip
// for the row maximum: ~290 wave-instructions per row all told (profiles/r03j_what_bounds_a_step.txt), 45 % of them
// scalar.  The kernel is bound by instruction issue, not by memory, so the way to make it faster is fewer
// instructions per row.
//
// Here a wavefront is G = 64 / GW groups of GW lanes (GW = 16: one group per DPP row), one sub-cluster per group,
// C consecutive band columns per lane (GW * C >= the widest row).  The groups walk their graphs in lock-step -- row r
// of every group in the same step -- so
//   * all per-row bookkeeping is vector code shared by the G groups (a group's "scalars" are values that happen to
//     be equal in its GW lanes); the scalar unit only counts rows;
//   * the F prefix maxima and the row maximum are 4-step DPP scans / rotations inside a 16-lane DPP row, the
//     per-lane part of a row is C independent cells (instruction-level parallelism for a lone wavefront);
//   * every row is written to an LDS ring of four rows per group (tagged H, E1, E2; cells outside the band and four
//     guard cells on each side hold "no path", so a successor reads its predecessor unchecked at any small band
//     shift); a predecessor further back than the ring comes from a copy in HBM (rows flagged in advance);
//   * direction words go to HBM relative to the row's first column, the C words of a lane side by side.
// Values are the tagged scores of poa_wave.hip's chain rows (TGB + score * 16 + tag), generalised to any number of
// predecessors: a candidate through predecessor slot k carries tag 15 - k (match), 15 - k / 7 - k (gap opened from H /
// extended from E of slot k), so one maximum picks the score and, among equal scores, the source the specification
// tries first; the direction nibbles are the inverted tags.
// Traceback, graph update and topological re-ordering are the algorithms of poa_wave.hip with a group's GW lanes in the
// place of the wavefront's 64 and vector state in the place of scalars.
// What does not fit (a row wider than GW * C, more than 7 predecessors, a predecessor more than 254 rows back, a band
// that loses the sink, a graph beyond its allocation) ends that group with status 3 | reason << 8 exactly like
// poa_wave.hip; the host hands those sub-clusters to poa_wave.hip's rounds.
//
// The file is compiled twice: by hipcc for gfx950 (poa_quad.hip, behind poa_quad_gfx950.h), and by g++ for the CPU wave
// emulator of the tests (behind their own back end), where the cross-lane primitives are meetings of 64 fibres.
// Everything else is the same source.
#pragma once
#include <cstdint>

#include "poa_task.h"

#include "poa_quad_defs.h"

// The cross-lane back end (pq::Grp<GW>, pq::lane_id, PQ_DEV, PQ_SITE, the LDS / memory helpers) comes from the file that
// includes this one: poa_quad_gfx950.h for the GPU, the tests' own emulator back end for the CPU -- the product does not
// know where the tests live.
#ifndef PQ_BACKEND
#error "include a back end (poa_quad_gfx950.h) before poa_quad_core.h"
#endif

namespace pq {

PQ_DEV int imax(int a, int b) { return a > b ? a : b; }
PQ_DEV int imin(int a, int b) { return a < b ? a : b; }
PQ_DEV int imax3(int a, int b, int c) { return imax(imax(a, b), c); }
PQ_DEV int tg_down(int x) { return x <= PQ_TGB / 2 ? PQ_NEG : (x - PQ_TGB) >> 4; }

// LDS the kernel needs for one wavefront (bytes)
template <int GW, int C>
struct Geom {
  static constexpr int G = 64 / GW;
  static constexpr int WSR = GW * C;                 // columns of a row (HBM stride of the row pools, T.ws)
  static constexpr int RW = WSR + 2 * PQ_GD;         // cells of a ring row
  static constexpr int ROW3 = 3 * RW;                // H, E1, E2 of one ring row
  static constexpr int RING_INTS = G * PQ_RING * ROW3;
  static constexpr int NULL_OFF = RING_INTS;         // a row of "no path" (predecessor slots a group does not have)
  static constexpr int CNT_OFF = NULL_OFF + ROW3;    // per group: edge counter
  static constexpr int LDS_INTS = CNT_OFF + 16;             // behind them: the G reads, qcap bytes each
  // bytes of a group's read buffer: the read at byte 4, and the lanes past the band's end read (and ignore) up to WSR
  // symbols behind it
  static constexpr int qcap(int max_len) { return (max_len + WSR + 24 + 15) & ~15; }
  static constexpr size_t lds_bytes(int max_len) { return sizeof(int32_t) * (size_t)LDS_INTS + (size_t)G * (size_t)qcap(max_len); }
};

// direction word: the low nibbles of its four bytes are the INVERTED source codes of H, H', E1, E2 (the tags of the values
// that won, as they are): H 0-6 match through predecessor slot k, 8 E1, 9 E2, 10 F1, 11 F2; H' 0-6, 8, 9; E1 0-6 opened
// from H of slot k, 8-14 extended from E1 of slot k - 8; E2 likewise.  Bit 4: F1 opened from H'(v, j - 1) (else
// extended), bit 5: F2 likewise.  (poa_wave.hip's codes, laid out for two byte permutes instead of a dozen shifts)
// row descriptor dA[r]: bits 0-2 base, 3-6 number of predecessors (0-7), bit 7 the row is read again after it left the
// ring (keep a copy in HBM), bits 8-15 / 16-23 / 24-31 row deltas of predecessor slots 0-2; dB[r]: slots 3-6

template <int GW, int C>
PQ_DEV void poaq_run(const PoaWaveTask* tasks, int n_tasks, const uint8_t* seqs, const int64_t* seq_off, int32_t* ws32,
                     int32_t* cons_len, int32_t* status, unsigned long long* cells, int32_t* lds, int block, int qcap) {
  typedef Geom<GW, C> GE;
  typedef Grp<GW> GR;
  static_assert(C >= 1 && C <= 7, "the read window of a lane is 8 bytes");
  constexpr int G = GE::G, WSR = GE::WSR, RW = GE::RW, GD = PQ_GD;
  const int lane = lane_id();
  const int g = lane / GW, l = lane % GW;
  const int ti = block * G + g;
  const bool has = ti < n_tasks;
  const PoaWaveTask T = tasks[has ? ti : 0];
  const int n = has ? (int)T.n_seqs : 0;
  const int nc = T.nc, ec = T.ec;
  const int opcap = nc + T.max_len + 4;
  // ---- HBM arrays of the sub-cluster (ws_layout of poa_wave.hip with ws = WSR), as 32-bit offsets into ws32
  const uint32_t W0 = (uint32_t)T.ws_off;
  const uint32_t a_out_head = W0, a_in_head = W0 + (uint32_t)nc, a_order = W0 + 2u * nc, a_index = W0 + 3u * nc,
                 a_col = W0 + 4u * nc, a_base = W0 + 5u * nc, a_row_beg = W0 + 6u * nc, a_row_end = W0 + 7u * nc,
                 a_hl = W0 + 8u * nc, a_dB = W0 + 9u * nc, a_row_mpl = W0 + 11u * nc, a_row_mpr = W0 + 12u * nc,
                 a_aln = W0 + 13u * nc, a_scr = W0 + 18u * nc, a_dA = W0 + 19u * nc + 64u, a_keepf = W0 + 20u * nc + 128u;
  const uint32_t E0 = W0 + 21u * nc + 192u;
  const uint32_t a_e_from = E0, a_e_to = E0 + (uint32_t)ec, a_e_w = E0 + 2u * ec, a_e_next_out = E0 + 3u * ec, a_e_next_in = E0 + 4u * ec;
  const uint32_t OP0 = E0 + 5u * ec;
  const uint32_t a_op_node = OP0, a_op_q = OP0 + (uint32_t)opcap, a_path_use = OP0 + 2u * opcap, a_path_aux = OP0 + 3u * opcap;
  const uint32_t P0 = OP0 + 4u * opcap;
  const uint32_t a_gdir = P0, a_gH = P0 + (uint32_t)nc * WSR, a_gE1 = P0 + 2u * nc * WSR, a_gE2 = P0 + 3u * nc * WSR;
#define WS(off) ws32[(uint32_t)(off)]
  // ---- LDS
  const int ring_g = g * (PQ_RING * GE::ROW3);        // the group's ring
  int32_t* const cnt = lds + GE::CNT_OFF + g;         // edge counter
  uint8_t* const qb = (uint8_t*)(lds + GE::LDS_INTS) + (size_t)g * (size_t)qcap;   // the read being aligned
  for (int x = lane; x < GE::CNT_OFF; x += 64) lds[x] = 0;   // ring cells, guards, null row: "no path"
  unsigned long long my_cells = 0;
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = 0;
#define PQ_PROF(k) do { const unsigned long long t_ = prof_clock(); prof[k] += t_ - prof_t; prof_t = t_; } while (0)
  bool alive = n > 0;
  int fail_code = 0;
#define PQ_FAIL(cond, code) do { if (alive && (cond)) { alive = false; fail_code = 3 | ((code) << 8); } } while (0)
#define PQ_GLOOP(var, bound, pred) for (int var = l; wave_any((pred) && var < (bound), PQ_SITE); var += GW) if ((pred) && var < (bound))
  int N = 0, ncols = 0;
  // ------------------------------------------------------------------ graph of the first read
  {
    const int64_t so = alive ? seq_off[T.seq_first] : 0;
    const int L0 = alive ? (int)(seq_off[T.seq_first + 1] - so) : 0;
    PQ_FAIL(L0 + 2 > nc || L0 + 1 > ec, 1);
    const uint8_t* q0 = seqs + so;
    PQ_GLOOP(v, L0 + 2, alive) {
      const int b = v < 2 ? 4 : q0[v - 2];
      WS(a_base + v) = b;
      WS(a_out_head + v) = v == 1 ? -1 : (v == 0 ? 0 : v - 1);
      WS(a_in_head + v) = v == 0 ? -1 : (v == 1 ? L0 : v - 2);
      for (int x = 0; x < 5; ++x) WS(a_aln + 5 * v + x) = -1;
      if (v >= 2) WS(a_aln + 5 * v + b) = v;
      const int idx = v == 0 ? 0 : v == 1 ? L0 + 1 : v - 1;
      WS(a_col + v) = v == 1 ? PQ_COL_SINK : idx;
      WS(a_index + v) = idx;
      WS(a_order + idx) = v;
    }
    PQ_GLOOP(e, L0 + 1, alive) {
      WS(a_e_from + e) = e == 0 ? 0 : e + 1;
      WS(a_e_to + e) = e == L0 ? 1 : e + 2;
      WS(a_e_w + e) = 1;
      WS(a_e_next_out + e) = -1; WS(a_e_next_in + e) = -1;
    }
    N = L0 + 2; ncols = L0 + 1;
    if (l == 0) *cnt = L0 + 1;
    mem_sync(PQ_SITE);
  }
  for (int i = 1; wave_any(alive && i < n, PQ_SITE); ++i) {
    bool act = alive && i < n;
    const int64_t so = act ? seq_off[T.seq_first + i] : 0;
    const int L = act ? (int)(seq_off[T.seq_first + i + 1] - so) : 0;
    const uint8_t* q = seqs + so;
    PQ_FAIL(act && L > T.max_len, 6);
    act = act && alive;
    const int w = 10 + (int)(0.01 * L);
    prof_t = prof_clock();
    // ---------------------------------------------------------- row descriptors
    {
      bool bad = false;
      PQ_GLOOP(r, N, act) {
        const int v = WS(a_order + r);
        uint32_t dA = (uint32_t)WS(a_base + v) & 7u, dB = 0;
        int np = 0;
        for (int e = WS(a_in_head + v); e >= 0; e = WS(a_e_next_in + e)) {
          const int d = r - WS(a_index + WS(a_e_from + e));
          if (d < 1 || d > 254) bad = true;
          if (np < 3) dA |= (uint32_t)(d & 255) << (8 + 8 * np);
          else if (np < 7) dB |= (uint32_t)(d & 255) << (8 * (np - 3));
          ++np;
        }
        if (np > 7) { bad = true; np = 7; }
        dA |= (uint32_t)np << 3;
        WS(a_dA + r) = (int32_t)dA;
        WS(a_dB + r) = (int32_t)dB;
        WS(a_hl + r) = PQ_NEG;
        WS(a_keepf + r) = 0;
      }
      const bool gbad = GR::bits(bad, PQ_SITE) != 0;
      PQ_FAIL(act && gbad, 2);
      act = act && alive;
      mem_sync(PQ_SITE);
      // rows that are read again after they left the ring keep a copy of H / E1 / E2 in HBM
      PQ_GLOOP(r, N - 1, act) {
        const uint32_t dA = (uint32_t)WS(a_dA + r), dB = (uint32_t)WS(a_dB + r);
        const int np = (int)((dA >> 3) & 15u);
        for (int k = 0; k < np; ++k) {
          const int d = (int)((k < 3 ? dA >> (8 + 8 * k) : dB >> (8 * (k - 3))) & 255u);
          if (d >= PQ_RING) WS(a_keepf + (r - d)) = 1;
        }
      }
      mem_sync(PQ_SITE);
    }
    // ------------------------------------------------------------ the read into LDS: 3 * q[j] at byte 4 + j (the bit offset of
    // the symbol's field in the row's score word), q[-1] = N (column 0 has no match score), zeros behind the end
    {
      uint32_t* qw32 = (uint32_t*)qb;
      for (int x = l; wave_any(act && 4 * x < L + 24 && 4 * x < qcap, PQ_SITE); x += GW) {
        if (!(act && 4 * x < L + 24 && 4 * x < qcap)) continue;
        uint32_t wv = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int idx = 4 * x + b - 4;
          const uint32_t sym = idx == -1 ? 4u : (idx >= 0 && idx < L) ? (uint32_t)q[idx] : 0u;
          wv |= (3u * (sym > 4u ? 4u : sym)) << (8 * b);
        }
        qw32[x] = wv;
      }
      lds_sync(PQ_SITE);
    }
    PQ_PROF(0);
    // ------------------------------------------------------------ forward (the sink is order[N-1])
    {
      int pb1 = 0, pl1 = 0, pr1 = 0, pb2 = 0, pl2 = 0, pr2 = 0, pb3 = 0, pl3 = 0, pr3 = 0;   // beg / mpl / mpr of rows r-1..r-3
      int32_t pH[C], pE1[C], pE2[C];   // the row computed last, as it went into the ring
#pragma unroll
      for (int c = 0; c < C; ++c) { pH[c] = 0; pE1[c] = 0; pE2[c] = 0; }
      uint64_t qcur = 0;               // read symbols q[j - 1] of the lane's columns, one byte each
      uint32_t qpre = 0;               // the symbol the group's last lane takes in if the next row is a chain row
      int failed = 0;                  // a row wider than the lanes (reported after the pass: the row loop has no per-group exits)
      unsigned cells_read = 0;
      // F(j) = max_{k<j} (H'(k) + k e) - o - j e: any origin of k and j does, so the lane's own column offsets (constants)
      // stand in for the columns
      int32_t LO1[C], LO2[C];
#pragma unroll
      for (int c = 0; c < C; ++c) { LO1[c] = (l * C + c) * (PQ_E1 * 16); LO2[c] = (l * C + c) * (PQ_E2 * 16); }
      const int nrows = act ? N - 1 : 0;
      const int maxN1 = wave_gmax<GW>(nrows, PQ_SITE);
      // descriptors of GW rows at a time, one per lane of the group, waited for where they are loaded: the row loop itself
      // issues no vector loads on its common path, so it never waits for its own direction-word stores (vmcnt counts loads
      // and stores alike)
      uint32_t ri_blk = 0;
      int blk0 = 0;
      if (act && l < N) ri_blk = (uint32_t)WS(a_dA + l) | (WS(a_keepf + l) ? 0x80u : 0u);
      force_ready(ri_blk);
      uint32_t ri = (uint32_t)GR::from((int)ri_blk, 0, PQ_SITE);
      for (int r = 0; r < maxN1; ++r) {
        const bool fw = r < nrows;
        if (r + 1 - blk0 >= GW) {
          blk0 = r + 1;
          ri_blk = 0;
          if (act && r + 1 + l < N) ri_blk = (uint32_t)WS(a_dA + r + 1 + l) | (WS(a_keepf + r + 1 + l) ? 0x80u : 0u);
          force_ready(ri_blk);
        }
        const uint32_t ri_nx = (uint32_t)GR::from((int)ri_blk, r + 1 - blk0, PQ_SITE);   // (a permute whose latency the row hides)
        const int bv = (int)(ri & 7u), np = (int)((ri >> 3) & 15u);
        const bool keep = ((ri >> 7) & 1u) != 0;
        const int d0 = (int)((ri >> 8) & 255u), d1 = (int)((ri >> 16) & 255u);
        // ---- chain rows (9 in 10): one predecessor, the previous row, whose band starts one column before this one's,
        // nothing kept in HBM.  The previous row is still in the lane's registers.
        int beg = imax(pl1 + 1 - w, 0), end = imin(pr1 + 1 + w, L);
        if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
        const bool chain = (ri & 0xFFF8u) == ((1u << 3) | (1u << 8)) && beg == pb1 + 1 && end - beg < WSR;
        const bool general = r == 0 || wave_any(fw && !chain, PQ_SITE);
        ++prof[4]; if (general) ++prof[5];
        int32_t m[C], e1[C], e2[C];
        int sc[C];
        // match + 1, mismatch - 2, N 0 (x 32) as signed 3-bit fields by read symbol: 110 110 110 110 with the base's own 001
        const uint32_t tab = bv < 4 ? (0xDB6u ^ (7u << (3 * bv))) : 0u;
        int jb;
        if (!general) {
          jb = beg + l * C;
          // the lane's symbols move down one place; the new one is the next lane's first (the last lane's was fetched a row ago)
          {
            const uint32_t nx = (uint32_t)GR::shl1z((int)(uint32_t)qcur, PQ_SITE);
            const uint32_t nb = (l == GW - 1 ? qpre : nx) & 0xffu;
            qcur = (qcur >> 8) | ((uint64_t)nb << (8 * (C - 1)));
          }
          const int32_t nH = GR::shl1z(pH[0], PQ_SITE), n1 = GR::shl1z(pE1[0], PQ_SITE), n2 = GR::shl1z(pE2[0], PQ_SITE);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            sc[c] = sbfe3(tab, (uint32_t)(qcur >> (8 * c)) & 0xffu) * 32;
            const int32_t hB = c + 1 < C ? pH[c + 1] : nH, xa = c + 1 < C ? pE1[c + 1] : n1, xb = c + 1 < C ? pE2[c + 1] : n2;
            m[c] = pH[c] + sc[c];
            e1[c] = imax(hB - (PQ_O1 + PQ_E1) * 16, xa - PQ_E1 * 16);
            e2[c] = imax(hB - (PQ_O2 + PQ_E2) * 16, xb - PQ_E2 * 16);
          }
        } else {
          // ---- band around the row maxima of the predecessors: the first two slots from the registers of the last three rows
          int lo = d0 == 1 ? pl1 : d0 == 2 ? pl2 : pl3, hi = d0 == 1 ? pr1 : d0 == 2 ? pr2 : pr3;
          if (np == 2) { lo = imin(lo, d1 == 1 ? pl1 : d1 == 2 ? pl2 : pl3); hi = imax(hi, d1 == 1 ? pr1 : d1 == 2 ? pr2 : pr3); }
          if (r == 0) { beg = 0; end = w < L ? w : L; }
          else {
            beg = lo + 1 - w; if (beg < 0) beg = 0;
            end = hi + 1 + w; if (end > L) end = L;
            if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
          }
          const int pbeg0 = d0 == 1 ? pb1 : d0 == 2 ? pb2 : pb3, pbeg1 = d1 == 1 ? pb1 : d1 == 2 ? pb2 : pb3;
          // one or two predecessors inside the ring whose stored row (with its guards) covers every column this row reads,
          // nothing to keep in HBM: read unchecked.  Anything else sends the whole wavefront's step the long way.
          const bool rare = fw && (np > 2 || d0 >= PQ_RING || beg - pbeg0 < 1 - GD || beg - pbeg0 > GD || keep || end - beg + 1 > WSR ||
                                   (np == 2 && (d1 >= PQ_RING || beg - pbeg1 < 1 - GD || beg - pbeg1 > GD)));
          const bool any_rare = r == 0 || wave_any(rare, PQ_SITE);
          uint32_t riB = 0;
          if (any_rare && r > 0) {
            if (wave_any(fw && np > 3, PQ_SITE)) { if (fw && np > 3) riB = (uint32_t)WS(a_dB + r); }
            lo = 1 << 30; hi = -1;
            for (int k = 0; wave_any(fw && k < np, PQ_SITE); ++k) {
              const bool on = fw && k < np;
              const int d = (int)((k < 3 ? ri >> (8 + 8 * k) : riB >> (8 * (k - 3))) & 255u);
              int ml = d == 1 ? pl1 : d == 2 ? pl2 : pl3, mr = d == 1 ? pr1 : d == 2 ? pr2 : pr3;
              if (wave_any(on && d >= PQ_RING, PQ_SITE)) {
                if (on && d >= PQ_RING) { ml = WS(a_row_mpl + (r - d)); mr = WS(a_row_mpr + (r - d)); }
              }
              if (on) { lo = imin(lo, ml); hi = imax(hi, mr); }
            }
            if (fw) {
              beg = lo + 1 - w; if (beg < 0) beg = 0;
              end = hi + 1 + w; if (end > L) end = L;
              if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
            }
            vm_drain();   // (see the end of the long way below)
          }
          if (fw && end - beg + 1 > WSR) failed = 1;
          jb = beg + l * C;
          // ---- read symbols q[j - 1] of the lane's columns (bytes 3 + jb .. of the group's copy in LDS)
          {
            const int a = 3 + jb;
            const uint32_t* qw32 = (const uint32_t*)qb + (a >> 2);
            const uint32_t w0 = qw32[0], w1 = qw32[1];
            const int sh = 8 * (a & 3);
            qcur = (((uint64_t)w1 << 32) | w0) >> sh;
            if (C > 5) { const uint32_t w2 = qw32[2]; qcur |= sh ? ((uint64_t)w2 << (64 - sh)) : 0ull; }
          }
#pragma unroll
          for (int c = 0; c < C; ++c) sc[c] = sbfe3(tab, (uint32_t)(qcur >> (8 * c)) & 0xffu) * 32;
          if (!any_rare) {
            {
              const int a0 = fw ? ring_g + ((r - d0) & (PQ_RING - 1)) * GE::ROW3 + (jb - 1 - pbeg0 + GD) : GE::NULL_OFF;
              int32_t hv[C + 1], x1[C], x2[C];
#pragma unroll
              for (int t = 0; t <= C; ++t) hv[t] = lds[a0 + t];
#pragma unroll
              for (int c = 0; c < C; ++c) { x1[c] = lds[a0 + RW + 1 + c]; x2[c] = lds[a0 + 2 * RW + 1 + c]; }
#pragma unroll
              for (int c = 0; c < C; ++c) {
                m[c] = hv[c] + sc[c];
                e1[c] = imax(hv[c + 1] - (PQ_O1 + PQ_E1) * 16, x1[c] - PQ_E1 * 16);
                e2[c] = imax(hv[c + 1] - (PQ_O2 + PQ_E2) * 16, x2[c] - PQ_E2 * 16);
              }
            }
            if (wave_any(fw && np == 2, PQ_SITE)) {
              const int a1 = (fw && np == 2) ? ring_g + ((r - d1) & (PQ_RING - 1)) * GE::ROW3 + (jb - 1 - pbeg1 + GD) : GE::NULL_OFF;
              int32_t hv[C + 1], x1[C], x2[C];
#pragma unroll
              for (int t = 0; t <= C; ++t) hv[t] = lds[a1 + t];
#pragma unroll
              for (int c = 0; c < C; ++c) { x1[c] = lds[a1 + RW + 1 + c]; x2[c] = lds[a1 + 2 * RW + 1 + c]; }
#pragma unroll
              for (int c = 0; c < C; ++c) {
                m[c] = imax(m[c], hv[c] + sc[c] - 1);
                e1[c] = imax3(e1[c], hv[c + 1] - ((PQ_O1 + PQ_E1) * 16 + 1), x1[c] - (PQ_E1 * 16 + 1));
                e2[c] = imax3(e2[c], hv[c + 1] - ((PQ_O2 + PQ_E2) * 16 + 1), x2[c] - (PQ_E2 * 16 + 1));
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < C; ++c) { m[c] = (r == 0 && jb + c == 0) ? (PQ_TGB + 15) : 0; e1[c] = 0; e2[c] = 0; }
            for (int k = 0; wave_any(fw && k < np, PQ_SITE); ++k) {
              const bool on = fw && k < np;
              const int d = (int)((k < 3 ? ri >> (8 + 8 * k) : riB >> (8 * (k - 3))) & 255u);
              const int ur = r - d;
              const int pbeg = d == 1 ? pb1 : d == 2 ? pb2 : pb3;
              const bool near = d < PQ_RING;
              // far predecessors from their HBM copy, near ones with clamped ring indices
              int fb = 0, fe = -1;
              if (wave_any(on && !near, PQ_SITE)) {
                mem_sync(PQ_SITE);   // (the copy was stored by this wavefront, some rows ago)
                if (on && !near) { fb = WS(a_row_beg + ur); fe = WS(a_row_end + ur); }
              }
              const int rb = ring_g + (ur & (PQ_RING - 1)) * GE::ROW3;
              int32_t hv[C + 1], x1[C], x2[C];
#pragma unroll
              for (int t = 0; t <= C; ++t) {
                const int jj = jb - 1 + t;
                int32_t vh = 0, v1 = 0, v2 = 0;
                if (on && near) {
                  int idx = jj - pbeg + GD;
                  idx = idx < 0 ? 0 : (idx > RW - 1 ? RW - 1 : idx);
                  vh = lds[rb + idx]; v1 = lds[rb + RW + idx]; v2 = lds[rb + 2 * RW + idx];
                } else if (on && jj >= fb && jj <= fe) {
                  const uint32_t o = (uint32_t)ur * WSR + (uint32_t)(jj - fb);
                  vh = WS(a_gH + o); v1 = WS(a_gE1 + o); v2 = WS(a_gE2 + o);
                }
                hv[t] = vh;
                if (t >= 1) { x1[t - 1] = v1; x2[t - 1] = v2; }
              }
#pragma unroll
              for (int c = 0; c < C; ++c) {
                const int32_t mk = hv[c] + sc[c] - k;                                   // tag 15 - k
                const int32_t a1 = hv[c + 1] - ((PQ_O1 + PQ_E1) * 16 + k);              // tag 15 - k: opened from H of slot k
                const int32_t b1 = x1[c] - (PQ_E1 * 16 + k);                            // tag 7 - k: extended from E1 of slot k
                const int32_t a2 = hv[c + 1] - ((PQ_O2 + PQ_E2) * 16 + k);
                const int32_t b2 = x2[c] - (PQ_E2 * 16 + k);                            // tag 7 - k (E2 is stored with tag 7 too)
                if (k == 0) { m[c] = mk; e1[c] = imax(a1, b1); e2[c] = imax(a2, b2); }
                else { m[c] = imax(m[c], mk); e1[c] = imax3(e1[c], a1, b1); e2[c] = imax3(e2[c], a2, b2); }
              }
            }
            // (the loads of this path end here: a wait at their first use -- or at the next write of a register one of them
            // was loaded into --, behind the join, would be a wait for the direction-word stores in every row)
#pragma unroll
            for (int c = 0; c < C; ++c) { force_ready_i(m[c]); force_ready_i(e1[c]); force_ready_i(e2[c]); }
            vm_drain();
          }
        }
        // the symbol the group's last lane needs if the next row is a chain row: q[jb + C - 1]
        qpre = (uint32_t)qb[4 + jb + C - 1];
        // ---- H' = max(M, E1, E2); the lane-serial half of the F prefix maxima; the row maximum
        // Tags: match through slot k 15 - k; E1 7; E2 6 as a candidate (stored with 7: its own sources are told apart like
        // E1's); F1 5; F2 4.  The direction nibbles are the inverted tags.
        int32_t hp[C], hq[C], t1[C], t2[C], p1[C], p2[C], e1s[C], e2s[C], nm[C];
        int32_t lmax = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int j = jb + c;
          nm[c] = ~((end - j) >> 31);     // all ones inside the band, 0 behind its end
          e1s[c] = (int32_t)(((uint32_t)e1[c] & ~15u) | 7u);
          e2s[c] = (int32_t)(((uint32_t)e2[c] & ~15u) | 7u);
          hp[c] = imax3(m[c], e1s[c], e2s[c] - 1) & nm[c];
          hq[c] = hp[c] | 15;
          t1[c] = hq[c] + LO1[c]; t2[c] = hq[c] + LO2[c];
          p1[c] = c ? imax(p1[c - 1], t1[c]) : t1[c];
          p2[c] = c ? imax(p2[c - 1], t2[c]) : t2[c];
          lmax = c ? imax(lmax, hq[c]) : hq[c];
        }
        const int32_t s1 = GR::scan_max(p1[C - 1], PQ_SITE), s2 = GR::scan_max(p2[C - 1], PQ_SITE);
        // (a group's first lane has nothing to its left: prefix maximum 0 = "no path"; its "opened here" bits come out set,
        // and are never read: F is no path there)
        const int32_t X1 = GR::shr1z(s1, PQ_SITE), X2 = GR::shr1z(s2, PQ_SITE);
        const int32_t t1_prev = GR::shr1z(t1[C - 1], PQ_SITE), t2_prev = GR::shr1z(t2[C - 1], PQ_SITE);
        // the row maximum of H is the maximum of H', attained where H' attains it (F(j) < max H' -- poa_wave.hip)
        const int32_t wmx = GR::all_max(lmax, PQ_SITE);
        int mpl, mpr;
        if (GW == 64) {
          // one group = the wavefront: first / last lane of each column's ballot, on the scalar unit
          int lcl = C, rcl = -1;
#pragma unroll
          for (int c = C - 1; c >= 0; --c) { if (hq[c] == wmx) { lcl = c; if (rcl < 0) rcl = c; } }
          int fl, ll;
          GR::first_last(rcl >= 0, fl, ll, PQ_SITE);   // (some lane holds the maximum)
          mpl = beg + fl * C + GR::from(lcl, fl, PQ_SITE); mpr = beg + ll * C + GR::from(rcl, ll, PQ_SITE);
        } else {
          int lc = 1 << 20, rc = -1;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            if (hq[c] == wmx) { lc = imin(lc, l * C + c); rc = l * C + c; }
          }
          mpl = GR::all_min(lc, PQ_SITE) + beg; mpr = GR::all_max(rc, PQ_SITE) + beg;
        }
        if (wmx <= PQ_TGB / 2) { mpl = beg; mpr = end; }
        // ---- F, H, direction words, the row into the ring
        const int slot_off = ring_g + (r & (PQ_RING - 1)) * GE::ROW3 + GD + l * C;
        const uint32_t rowo = (uint32_t)r * WSR + (uint32_t)(l * C);
        uint32_t dw[C];
        int32_t hfull[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int32_t x1 = c ? imax(X1, p1[c - 1]) : X1, x2 = c ? imax(X2, p2[c - 1]) : X2;
          const int32_t f1 = x1 - (PQ_O1 * 16 + 10) - LO1[c];      // tag 15 -> 5
          const int32_t f2 = x2 - (PQ_O2 * 16 + 11) - LO2[c];      // tag 15 -> 4
          const int32_t h = imax3(hp[c], f1, f2);
          hfull[c] = h;
          // the tags of H, H', E1, E2 (the winners' sources, inverted) in the low nibbles of the word's four bytes;
          // bit 4 / 5: F1 / F2 opened from H'(v, j - 1)
          const uint32_t o1 = (c ? t1[c - 1] : t1_prev) == x1 ? 0x10u : 0u;
          const uint32_t o2 = (c ? t2[c - 1] : t2_prev) == x2 ? 0x20u : 0u;
          dw[c] = (pack4((uint32_t)h, (uint32_t)hp[c], (uint32_t)e1[c], (uint32_t)e2[c]) & 0x0F0F0F0Fu) | o1 | o2;
          pH[c] = (h | 15) & nm[c]; pE1[c] = e1s[c] & nm[c]; pE2[c] = e2s[c] & nm[c];
        }
        if (fw) {
          int32_t* const dp = &WS(a_gdir + rowo);
          int32_t* const rp = lds + slot_off;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            dp[c] = (int32_t)dw[c];
            rp[c] = pH[c]; rp[RW + c] = pE1[c]; rp[2 * RW + c] = pE2[c];
          }
          if (end == L) {   // the row reaches the read's end: its last cell is a candidate for the sink
#pragma unroll
            for (int c = 0; c < C; ++c) if (jb + c == L) WS(a_hl + r) = tg_down(hfull[c]);
          }
          if (l == 0) { WS(a_row_beg + r) = beg; cells_read += (unsigned)(end - beg + 1); }
        }
        if (general && wave_any(fw && keep, PQ_SITE)) {
          if (fw && keep) {
#pragma unroll
            for (int c = 0; c < C; ++c) { WS(a_gH + rowo + c) = pH[c]; WS(a_gE1 + rowo + c) = pE1[c]; WS(a_gE2 + rowo + c) = pE2[c]; }
            if (l == 0) { WS(a_row_end + r) = end; WS(a_row_mpl + r) = mpl; WS(a_row_mpr + r) = mpr; }
          }
        }
        pb3 = pb2; pl3 = pl2; pr3 = pr2; pb2 = pb1; pl2 = pl1; pr2 = pr1; pb1 = beg; pl1 = mpl; pr1 = mpr;
        ri = ri_nx;
        lds_sync(PQ_SITE);
      }
      my_cells += cells_read;
      PQ_FAIL(act && failed, 3);
      act = act && alive;
    }
    mem_sync(PQ_SITE);   // direction words, row origins and end cells are in HBM
    PQ_PROF(1);
    // --------------------------------------------------------- traceback
    int nops = -1;
    {
      int bu = -1; int32_t bsc = PQ_NEG;
      for (int e = act ? WS(a_in_head + 1) : -1; wave_any(e >= 0, PQ_SITE); e = e >= 0 ? WS(a_e_next_in + e) : -1) {
        if (e >= 0) {
          const int ur = WS(a_index + WS(a_e_from + e));
          const int32_t h = WS(a_hl + ur);
          if (h > bsc) { bsc = h; bu = ur; }
        }
      }
      bool walking = act && bu >= 0 && bsc > PQ_NEG / 2;
      if (walking) nops = 0;
      int tr = bu, tj = L, st = 0;   // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 H'
      int r0 = -(1 << 28), j0 = 0;
      uint32_t W0_ = 0, W1_ = 0, W2_ = 0, W3_ = 0, PA = 0, PB = 0;
      for (;;) {
        bool walk = walking && (tr != 0 || tj > 0);
        if (!wave_any(walk, PQ_SITE)) break;
        if (walk && (nops + tj + 2 > opcap || tj < 0 || tr < 0)) { walking = false; walk = false; nops = -1; }   // cannot happen on a valid path
        if (wave_any(walk && tr == 0, PQ_SITE)) {   // only inserted bases remain
          const bool s0 = walk && tr == 0;
          PQ_GLOOP(t, tj, s0) { WS(a_op_node + nops + t) = -1; WS(a_op_q + nops + t) = tj - 1 - t; }
          if (s0) { nops += tj; tj = 0; walk = false; }
        }
        int k = r0 - tr, d = k - (j0 - tj);
        const bool need = walk && (k < 0 || k >= GW || d < 0 || d > 3);
        if (wave_any(need, PQ_SITE)) {
          if (need) {
            r0 = tr; j0 = tj; k = 0; d = 0;
            const int rr = r0 - l;
            W0_ = W1_ = W2_ = W3_ = 0; PA = 0; PB = 0;
            if (rr >= 0) {
              const int cc = j0 - l - WS(a_row_beg + rr);
              const uint32_t bp = a_gdir + (uint32_t)rr * WSR;
              if ((unsigned)cc < (unsigned)WSR) W0_ = (uint32_t)WS(bp + cc);
              if ((unsigned)(cc + 1) < (unsigned)WSR) W1_ = (uint32_t)WS(bp + cc + 1);
              if ((unsigned)(cc + 2) < (unsigned)WSR) W2_ = (uint32_t)WS(bp + cc + 2);
              if ((unsigned)(cc + 3) < (unsigned)WSR) W3_ = (uint32_t)WS(bp + cc + 3);
              PA = (uint32_t)WS(a_dA + rr); PB = (uint32_t)WS(a_dB + rr);
            }
          }
        }
        const uint32_t wsel = d == 0 ? W0_ : d == 1 ? W1_ : d == 2 ? W2_ : W3_;
        // a run of plain diagonal steps -- a match through predecessor slot 0, which is the row above -- all at once
        const bool diag = (wsel & 15u) == 15u && ((PA >> 8) & 255u) == 1u && r0 - l >= 1;
        const uint64_t gm = GR::bits(diag, PQ_SITE);
        const int kk = k & (GW - 1);
        int run = ctz64(~(gm >> kk));
        if (run > tj) run = tj;
        const uint32_t dwk = (uint32_t)GR::from((int)wsel, kk, PQ_SITE), pak = (uint32_t)GR::from((int)PA, kk, PQ_SITE),
                       pbk = (uint32_t)GR::from((int)PB, kk, PQ_SITE);
        if (walk && st == 0 && run > 0) {
          if (l < run) { WS(a_op_node + nops + l) = tr - l; WS(a_op_q + nops + l) = tj - 1 - l; }
          nops += run; tr -= run; tj -= run;
        } else if (walk) {
          int s = -1;   // predecessor slot to follow
          if (st == 0 || st == 5) {
            const uint32_t dd = ~(st == 0 ? dwk : (dwk >> 8)) & 15u;
            if (dd < 8) {
              if (l == 0) { WS(a_op_node + nops) = tr; WS(a_op_q + nops) = tj - 1; }
              ++nops; --tj; st = 0; s = (int)dd;
            } else st = (int)dd - 7;   // 8 -> E1, 9 -> E2, 10 -> F1, 11 -> F2
          } else if (st == 1 || st == 2) {
            const uint32_t dd = ~(st == 1 ? (dwk >> 16) : (dwk >> 24)) & 15u;
            if (l == 0) { WS(a_op_node + nops) = tr; WS(a_op_q + nops) = -1; }
            ++nops; s = (int)(dd & 7u);
            if (dd < 8) st = 0;
          } else {
            const uint32_t open = st == 3 ? ((dwk >> 4) & 1u) : ((dwk >> 5) & 1u);
            if (l == 0) { WS(a_op_node + nops) = -1; WS(a_op_q + nops) = tj - 1; }
            ++nops;
            if (open) st = 5;
            --tj;
          }
          if (s >= 0) tr -= (int)((s < 3 ? pak >> (8 + 8 * s) : pbk >> (8 * (s - 3))) & 255u);
        }
      }
    }
    PQ_PROF(2);
    PQ_FAIL(act && nops < 0, 4);
    PQ_FAIL(act && nops > 32000, 5);   // (path positions are packed into 15 bits of a scan key)
    act = act && alive;
    mem_sync(PQ_SITE);
    // ------------------------------------------------------- graph update
    const int n_old = N;
    PQ_GLOOP(c, ncols, act) WS(a_scr + c) = 0;
    mem_sync(PQ_SITE);
    int carry_c = 0, carry_n = 0, carry_key = 0;
    for (int p0 = 0; wave_any(act && p0 < nops, PQ_SITE); p0 += GW) {
      const int p = p0 + l;
      const bool valid = act && p < nops;
      int row = -1, j = -1;
      if (valid) { row = WS(a_op_node + nops - 1 - p); j = WS(a_op_q + nops - 1 - p); }
      const bool ali = valid && row >= 0 && j >= 0, ins = valid && row < 0;
      const int v = ali ? WS(a_order + row) : -1;
      const int qb = (ali || ins) ? (int)q[j] : 0;
      int use = -1;
      bool isnew = ins;
      if (ali) {
        if (WS(a_base + v) == qb) use = v;
        else { const int a = WS(a_aln + 5 * v + qb); if (a >= 0) use = a; else isnew = true; }
      }
      const int inc_c = GR::scan_add((ali || ins) ? 1 : 0, PQ_SITE), inc_n = GR::scan_add(isnew ? 1 : 0, PQ_SITE);
      const int cidx = carry_c + inc_c - ((ali || ins) ? 1 : 0);
      const int nrank = carry_n + inc_n - (isnew ? 1 : 0);
      const int key = ali ? (((cidx + 1) << 16) | WS(a_col + v)) : 0;     // (columns and path positions < 65536)
      const int inc_k = GR::scan_max(key, PQ_SITE);
      const int ikey = imax(carry_key, inc_k);
      carry_c += GR::last(inc_c, PQ_SITE);
      carry_n += GR::last(inc_n, PQ_SITE);
      carry_key = imax(carry_key, GR::last(inc_k, PQ_SITE));
      uint32_t aux = 0xFFFFFFFFu;
      if (isnew && n_old + nrank < nc) {   // (capacity is checked after the loop)
        const int nid = n_old + nrank;
        use = nid;
        WS(a_base + nid) = qb;
        WS(a_out_head + nid) = -1; WS(a_in_head + nid) = -1;
        if (ali) {   // a new base at the column of v: joins v's aligned group
          for (int b = 0; b < 5; ++b) {
            const int sib = WS(a_aln + 5 * v + b);
            WS(a_aln + 5 * nid + b) = sib;
            if (sib >= 0) WS(a_aln + 5 * sib + qb) = nid;
          }
          WS(a_aln + 5 * nid + qb) = nid;
          WS(a_col + nid) = WS(a_col + v);
        } else {     // an inserted base: a new column, the t-th after its anchor's
          for (int b = 0; b < 5; ++b) WS(a_aln + 5 * nid + b) = b == qb ? nid : -1;
          WS(a_col + nid) = PQ_COL_NEW;
          const int ac = ikey & 0xffff, t = (cidx + 1) - (ikey >> 16);
          atomic_max(&WS(a_scr + ac), t);
          aux = (uint32_t)ac | ((uint32_t)t << 16);
        }
      }
      if (ali || ins) { WS(a_path_use + cidx) = use; WS(a_path_aux + cidx) = (int32_t)aux; }
    }
    const int PC = carry_c, n_new = n_old + carry_n;
    // the graph outgrew its allocation: this sub-cluster goes to the roomier rounds of poa_wave.hip
    PQ_FAIL(act && (n_new > nc || ncols + carry_n > nc || ncols + carry_n > 65000 || PC > 65000), 5);
    act = act && alive;
    mem_sync(PQ_SITE);
    // one edge per path step; no other lane touches the out-list of u or the in-list of v
    for (int t = l; wave_any(act && t <= PC, PQ_SITE); t += GW) {
      if (!(act && t <= PC)) continue;
      const int u = t == 0 ? 0 : WS(a_path_use + t - 1), v = t == PC ? 1 : WS(a_path_use + t);
      int tail = -1, e = WS(a_out_head + u);
      bool found = false;
      for (; e >= 0; e = WS(a_e_next_out + e)) {
        if (WS(a_e_to + e) == v) { WS(a_e_w + e) += 1; found = true; break; }
        tail = e;
      }
      if (found) continue;
      const int ne = atomic_add(cnt, 1);
      if (ne >= ec) continue;   // out of edge slots: the counter above ec gives the sub-cluster up below
      WS(a_e_from + ne) = u; WS(a_e_to + ne) = v; WS(a_e_w + ne) = 1;
      WS(a_e_next_out + ne) = -1; WS(a_e_next_in + ne) = -1;
      if (tail < 0) WS(a_out_head + u) = ne; else WS(a_e_next_out + tail) = ne;
      int ie = WS(a_in_head + v);
      if (ie < 0) WS(a_in_head + v) = ne;
      else {
        while (WS(a_e_next_in + ie) >= 0) ie = WS(a_e_next_in + ie);
        WS(a_e_next_in + ie) = ne;
      }
    }
    mem_sync(PQ_SITE);
    PQ_FAIL(act && *cnt > ec, 5);
    act = act && alive;
    // column ranks: every column moves right by the number of columns inserted before it
    int carry = 0;
    for (int c0 = 0; wave_any(act && c0 < ncols, PQ_SITE); c0 += GW) {
      const int c = c0 + l;
      const int x = (act && c < ncols) ? WS(a_scr + c) : 0;
      const int inc = GR::scan_add(x, PQ_SITE);
      if (act && c < ncols) WS(a_scr + c) = carry + inc - x;
      carry += GR::last(inc, PQ_SITE);
    }
    mem_sync(PQ_SITE);
    PQ_GLOOP(v, n_new, act) {
      const int cv = WS(a_col + v);
      if (cv < PQ_COL_SINK) WS(a_col + v) = cv + WS(a_scr + cv);
    }
    mem_sync(PQ_SITE);
    PQ_GLOOP(t, PC, act) {
      const uint32_t aux = (uint32_t)WS(a_path_aux + t);
      if (aux != 0xFFFFFFFFu) { const int ac = (int)(aux & 0xffffu); WS(a_col + WS(a_path_use + t)) = ac + WS(a_scr + ac) + (int)(aux >> 16); }
    }
    const int ncols_new = ncols + carry;
    mem_sync(PQ_SITE);
    // counting sort of the nodes by column = a topological order; the sink goes last
    PQ_GLOOP(c, ncols_new, act) WS(a_scr + c) = 0;
    mem_sync(PQ_SITE);
    PQ_GLOOP(v, n_new, act) { if (v != 1) (void)atomic_add(&WS(a_scr + WS(a_col + v)), 1); }
    mem_sync(PQ_SITE);
    carry = 0;
    for (int c0 = 0; wave_any(act && c0 < ncols_new, PQ_SITE); c0 += GW) {
      const int c = c0 + l;
      const int x = (act && c < ncols_new) ? WS(a_scr + c) : 0;
      const int inc = GR::scan_add(x, PQ_SITE);
      if (act && c < ncols_new) WS(a_scr + c) = carry + inc - x;
      carry += GR::last(inc, PQ_SITE);
    }
    mem_sync(PQ_SITE);
    PQ_GLOOP(v, n_new, act) {
      if (v != 1) {
        const int pos = atomic_add(&WS(a_scr + WS(a_col + v)), 1);
        WS(a_order + pos) = v; WS(a_index + v) = pos;
      }
    }
    if (act) {
      if (l == 0) { WS(a_order + n_new - 1) = 1; WS(a_index + 1) = n_new - 1; }
      N = n_new; ncols = ncols_new;
    }
    mem_sync(PQ_SITE);
    PQ_PROF(3);
  }
  prof_out(prof);
  // the heaviest-bundle consensus is poa_bundle_kernel's job: hand over the number of graph rows
  if (has && l == 0) {
    if (n <= 0) { cons_len[ti] = 0; status[ti] = 0; }
    else if (!alive) status[ti] = fail_code;
    else { cons_len[ti] = N; status[ti] = 0; atomic_add64(cells, my_cells); }
  }
#undef WS
#undef PQ_FAIL
#undef PQ_GLOOP
#undef PQ_PROF
}

}  // namespace pq
