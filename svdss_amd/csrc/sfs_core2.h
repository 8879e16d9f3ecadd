// sfs_core2.h -- per-read state machine of the v2 search kernel, shared by the
// HIP kernel (one instance per lane) and tests/lane_emulator.cpp.
//
// Same observable behaviour as PingPong::ping_pong_search
// (/root/reference/ping_pong.cpp:4-49): identical SFS (start, length, order)
// and identical count of rb3_fmd_extend calls -- but the interval-size
// predicate "does this substring occur" is evaluated three ways, whichever is
// cheapest for the state the read is in:
//   TABLE  the first K symbols of a phase (backward :12-22 or forward :30-37)
//          are resolved by ONE lookup in a 4^K table holding, for every K-mer,
//          either its SA interval or the depth at which it stops occurring;
//   LF     one backward rank step on the BWT block layout (fmd_layout.h), as in v1;
//   TEXT   once a backward phase has narrowed to a single occurrence (size 1),
//          "prepend c succeeds" <=> "the text byte before the occurrence is c",
//          so up to 64 read symbols per iteration are compared directly with the
//          reference text (located through the full suffix array) -- no BWT walk.
//   SET    the same with 2-4 occurrences left (reads in low-copy repeats, where LF would spend one
//          iteration per symbol until the copies diverge): their text positions are kept, 16 read symbols
//          per iteration are compared with every surviving copy; the interval size is the number of
//          survivors, which is all the algorithm ever looks at.
// Every iteration of the kernel issues the loads of exactly one of these
// operations per lane, waits once, and applies the result.
#pragma once
#include "fmd_layout.h"

#ifndef SV_COUNT_PASS
#define SV_COUNT_PASS()
#endif

enum { SV_OP_DONE = 0, SV_OP_LF = 1, SV_OP_TABLE = 2, SV_OP_SA = 3, SV_OP_TEXT = 4, SV_OP_FILL = 5,
       SV_OP_TEXT_SLOW = 6, SV_OP_PEEK = 7, SV_OP_SA_SET = 8, SV_OP_SET = 9, SV_OP_BS_SA = 10, SV_OP_BS_TEXT = 11,
       SV_OP_BS_TEXT_SLOW = 12, SV_OP_BS_ORD = 13, SV_N_OPS = 14 };

#define SV_M_DIR 1     // 0 backward (ping_pong.cpp:15-22), 1 forward (:31-37)
#define SV_M_START 2   // at a phase start: no interval yet (before :12 / :30)
#define SV_M_TEXT 4    // unique-occurrence text-compare mode (backward only)
#define SV_M_CHAIN 8   // streaming assembler has an open chain
#define SV_NO_WINDOW (-0x40000000)

// Per-lane window of 64 read symbols staged in LDS (device) or a local array
// (emulator): dword row r lives at base[r * stride]; the byte of absolute read
// buffer position a is byte (a & 3) of row ((a & 63) >> 2).
struct SvRing {
  uint32_t* base;
  int stride;
};

// text index deltas of the occurrences of SET mode: slot i of this lane at base[i * stride]
struct SvSet {
  int64_t* base;
  int stride;
};

template <class P>
struct SvLane {
  P lo, hi;          // LF mode: SA interval [lo,hi) of W (backward) / revcomp(W) (forward)
  int64_t tdelta;    // TEXT mode: text index of read position p is tdelta + p
  int32_t pos;       // read position of the last consumed symbol
  int32_t begin;     // start of the SFS being closed (forward phase)
  int32_t len;
  int32_t mode;
  int32_t c;         // symbol of the pending LF step / start of the pending table lookup
  int32_t wrel;      // ring holds read positions [wrel, wrel+64)
  int32_t n_sfs;
  int32_t n_ext;
  int32_t chain_lo, chain_end;
  int32_t stop_lo;   // segmented search: this lane owns read positions >= stop_lo (0: whole read)
  int32_t n_below;   // SFS produced with start < stop_lo (overrun into the next segment)
  int32_t bs_m;      // BS: symbols of Q matched with the middle row so far
};

// A segment's chain keeps going past its lower boundary until it sees (peek) that the next
// segment's chain started a forward phase at the same position, at most this many SFS, so that
// the stitcher can find the SFS start the two chains share (see sv_stitch).
#define SV_OVERRUN 48
#define SV_M_PARTIAL 16   // the lane stopped after its overrun, not at the start of the read
#define SV_M_PEEK 32      // waiting for the left neighbour's records (segmented search, see sv_apply_peek)
#define SV_PEEK_RECS 4    // neighbour records examined per PEEK operation
#define SV_PEEK_OPS 8     // PEEK operations per SFS before giving up (the overrun goes on)
#define SV_M_SET 64       // 2-4 occurrences followed in the text, 16 symbols per operation (backward only)
#define SV_SET_MAX 4
#define SV_SET_SHIFT 8    // mode bits 8-11: which of the occurrences are still alive
#define SV_SET_WIN 16
#define SV_LFC_SHIFT 12   // mode bits 12-13: LF steps taken on an interval of 2-4 since the phase started
#define SV_LFC_MASK (3 << SV_LFC_SHIFT)
#define SV_SET_AFTER 2    // ... chance matches of a K-mer die within a symbol or two; real copies do not
#define SV_M_SHIFT (1 << 14)   // s.c = read symbols consumed since the table lookup whose SA rows are being fetched
#define SV_M_FEWSET (1 << 15)  // the alive bits already hold the occurrences of a FEW entry that are left: fetch their rows
#define SV_PEEK_VISIBLE 16 // records per segment stored so that a concurrently running neighbour can see them
// BS: a backward phase on a DEEP interval (a K-mer with SV_BS_MIN or more occurrences: reads inside repeat families)
// finished by binary search instead of one rank step per symbol.  The phase ends where the longest suffix of
// P[0..e] that occurs anywhere ends; reverse-complemented (the text holds both strands) that is the longest PREFIX of
// Q = revcomp(P[0..e]) that occurs, i.e. the longest common prefix of Q with its two neighbours in the suffix array
// among the rows that start with Q's first K symbols -- the interval of the reverse-complemented K-mer, one more table
// lookup.  log2(occurrences) rows are looked at (suffix array entry, then 64 symbols per text comparison, from the
// prefix length both current neighbours are known to share with Q), instead of hundreds of dependent block fetches.
// The comparison itself is the TEXT mode's: row M holds text position p', whose mirror image in the other strand
// (2 * middle '$' of its record pair - p') is where P[e] sits, and P[e - x] is compared with the text going left.
// State while it runs: lo / hi = the rows [A, B) not decided yet, begin / c = symbols Q shares with row A - 1 / row B
// (K for a neighbour outside the K-mer's interval), tdelta / bs_m = the middle row's text alignment and the symbols
// matched with it so far; pos stays at the K-mer's first symbol (e = pos + K - 1).
#define SV_M_BS (1 << 16)
#define SV_M_BS_TAB (1 << 17)   // waiting for the interval of the reverse-complemented K-mer
#define SV_M_BS_CMP (1 << 18)   // the middle row's text position is known: compare
#define SV_M_BS_ORD (1 << 19)   // a comparison ended at a mismatch: which of the two symbols is the smaller
#define SV_BS_MIN 8
// ... but not always: most phases on a deep interval are over after a dozen symbols (the ones that follow an SFS: they
// end at the sequencing error that SFS is about), and of the others most narrow fast (old, diverged repeat copies: a few dozen rank steps until four
// occurrences are left and SET / TEXT take over), and a binary search costs ~3 log2(occurrences) memory operations
// whatever the outcome.  The walk starts as before; after SV_BS_PROBE rank steps the shrinkage of the interval says how
// many more it would take to get down to four occurrences at this rate -- Np / N0 = q^p, so ln(Np / 4) / ln(1 / q) --
// and a phase that would need more than bs_after of them (near-identical copies: the interval barely moves) is started
// again from its K-mer by binary search.  Measured on a 3.1 Gb reference with 45 % of its bases in 40 repeat families
// (profiles/r04h_*): copies 1 % apart 431 -> 344 ms per million reads; at 5 % and 15 % the walk is the cheaper way and
// the estimate leaves it alone (a binary search started at once there costs +8 % / +35 %).  tdelta holds N0 and bs_m counts the rank steps while SV_M_BS_OK is set.
// (The estimate decides cost only: either way ends the phase at the same symbol with the same extension count.)
#define SV_BS_PROBE 8
#define SV_BS_AFTER_DEFAULT 256  // (SvdssDevIndex::bs_after; SVDSS_BS_AFTER overrides, 0: binary search at once)
#define SV_M_BS_OK (1 << 20)    // the backward phase began with a table lookup on a deep interval: BS may take it over
#define SV_BS_MAX_DOLLAR 256    // '$' positions kept next to the lanes (128 records); beyond: no BS

struct SvOp {
  int op;
  int64_t a;  // LF: unused; TABLE: key; SA: SA index; TEXT: unused; FILL: first chunk index
};

SVDSS_HD uint32_t sv_ring_row(const SvRing& g, int r) { return g.base[(r & 15) * g.stride]; }

SVDSS_HD int sv_ring_sym(const SvRing& g, int64_t a) {
  return (int)((sv_ring_row(g, (int)((a & 63) >> 2)) >> ((a & 3) * 8)) & 0xffu);
}

SVDSS_HD void sv_ring_fill(const SvRing& g, int64_t first_chunk, const svdss_u4 b[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (int)(((first_chunk + j) & 3) * 4);
    g.base[(r + 0) * g.stride] = b[j].x;
    g.base[(r + 1) * g.stride] = b[j].y;
    g.base[(r + 2) * g.stride] = b[j].z;
    g.base[(r + 3) * g.stride] = b[j].w;
  }
}

// 2-bit key (text order, first symbol in the low bits) of the K symbols at
// absolute positions [a0, a0+K); returns false if any of them is not A/C/G/T.
SVDSS_HD bool sv_ring_kmer(const SvRing& g, int64_t a0, int K, uint32_t& key) {
  const int r0 = (int)((a0 & 63) >> 2);
  const int sh = (int)(a0 & 3) * 8;
  uint32_t row[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) row[i] = sv_ring_row(g, r0 + i);
  uint32_t k = 0, bad = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = (uint32_t)(((((uint64_t)row[i + 1]) << 32) | row[i]) >> sh);
    const uint32_t t = w - 0x01010101u;             // A,C,G,T -> 0..3; '$' -> 0xff; N -> 4
    const int nb = K - 4 * i;                       // bytes of this dword that belong to the K-mer
    const uint32_t m = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
    bad |= t & 0xfcfcfcfcu & m;
    k |= (((t & 0x03030303u) * 0x01041040u) >> 24) << (8 * i);
  }
  key = K >= 16 ? k : (k & ((1u << (2 * K)) - 1u));
  return bad == 0;
}

// key of revcomp(W) from the key of W (K symbols)
SVDSS_HD uint32_t sv_key_revcomp(uint32_t key, int K) {
#if defined(__HIPCC__)
  uint32_t z = __builtin_bitreverse32(key);
#else
  uint32_t z = key;
  z = ((z >> 1) & 0x55555555u) | ((z & 0x55555555u) << 1);
  z = ((z >> 2) & 0x33333333u) | ((z & 0x33333333u) << 2);
  z = ((z >> 4) & 0x0f0f0f0fu) | ((z & 0x0f0f0f0fu) << 4);
  z = __builtin_bswap32(z);
#endif
  z = ((z >> 1) & 0x55555555u) | ((z & 0x55555555u) << 1);
  z >>= (32 - 2 * K);
  return z ^ (K >= 16 ? 0xffffffffu : ((1u << (2 * K)) - 1u));
}

template <class P>
SVDSS_HD void sv_lane_init(SvLane<P>& s, int len, int start_pos = -1, int stop_lo = 0) {
  s.lo = 0; s.hi = 0; s.tdelta = 0;
  s.len = len;
  s.pos = start_pos >= 0 ? start_pos : len - 1;
  s.begin = 0;
  s.stop_lo = stop_lo;
  s.n_below = 0;
  s.bs_m = 0;
  s.mode = SV_M_START;  // backward phase about to start at pos = len-1 (ping_pong.cpp:8-12)
  s.c = 0;
  s.wrel = SV_NO_WINDOW;
  s.n_sfs = 0; s.n_ext = 0;
  s.chain_lo = 0; s.chain_end = 0;
}

template <class P>
SVDSS_HD bool sv_in_window(const SvLane<P>& s, int p) { return p >= s.wrel && p < s.wrel + 64; }

// Streaming form of Assembler::assemble (/root/reference/assembler.cpp:34-56).  ping_pong_search pushes SFS with strictly
// decreasing qs AND strictly decreasing end, so the sort at assembler.cpp:36 is a reversal and the chain rule
// "sfs[j-1].qs + sfs[j-1].l > sfs[j].qs" can be applied as records are produced: a new SFS (q, l) joins the open chain
// iff q + l > (qs of the previously produced SFS).  The chain's end is the end of its first-produced member (largest
// qs), exactly sfs[j-1].qs + sfs[j-1].l at assembler.cpp:42,50.
template <class P, class Emit>
SVDSS_HD void sv_emit(SvLane<P>& s, int qs, int l, bool assemble, Emit&& emit) {
  if (!assemble) { emit(s.n_sfs++, qs, l); return; }
  if (s.mode & SV_M_CHAIN) {
    if (qs + l > s.chain_lo) { s.chain_lo = qs; return; }
    emit(s.n_sfs++, s.chain_lo, s.chain_end - s.chain_lo);
  }
  s.mode |= SV_M_CHAIN;
  s.chain_lo = qs;
  s.chain_end = qs + l;
}

template <class P, class Emit>
SVDSS_HD void sv_flush(SvLane<P>& s, bool assemble, Emit&& emit) {
  if (assemble && (s.mode & SV_M_CHAIN)) {
    emit(s.n_sfs++, s.chain_lo, s.chain_end - s.chain_lo);
    s.mode &= ~SV_M_CHAIN;
  }
}

// sv_decide without its branches (round 6; the kernels that carry no BS code).  A wavefront executes the union of the
// paths its 64 lanes take, and nearly every pass has a lane on every path: as a tree of early returns sv_decide cost ~650
// instructions per pass, two thirds of them exec-mask bookkeeping and the copies that merge (op, a) at every join
// (profiles/r06af_*).  Here every lane computes the predicates of all paths and the results are selected; the branches
// left are the one around the record store of a forward phase's end and the rare single-symbol phase start.  Same decisions,
// same state, same order of the LFC increment and the window checks as sv_decide below -- tests/lane_emulator.cpp runs
// this form whenever BS is off.
template <class P, class Emit>
SVDSS_HD SvOp sv_decide_flat(SvLane<P>& s, const SvdssDevIndex& ix, const SvRing& g, int64_t off, bool assemble, Emit&& emit,
                             bool can_peek, bool use_set) {
  const int K = ix.k;
  const bool has_sa = ix.sa != nullptr;
  int op = SV_OP_DONE;
  int64_t a = 0;
  for (;;) {
    const int mode = s.mode;
    const bool usable = s.len > 0 && !(mode & SV_M_PARTIAL);
    const bool peeking = usable && (mode & SV_M_PEEK);
    const bool live = usable && !(mode & SV_M_PEEK);
    const bool tmode = (mode & (SV_M_SET | SV_M_TEXT)) != 0;
    const bool dir = (mode & SV_M_DIR) != 0;
    const bool nonempty = s.hi > s.lo;
    const bool body = live && !tmode;
    const bool mid = body && !(mode & SV_M_START);          // inside a phase: an interval to look at
    const bool walk = mid && nonempty && (dir || s.pos > 0);
    const bool bend = mid && !nonempty && !dir;             // backward phase over (ping_pong.cpp:28)
    const bool fend = mid && !nonempty && dir;              // forward phase over (:38-47)
    op = peeking ? SV_OP_PEEK : SV_OP_DONE;                 // (mid && nonempty && !dir && pos == 0 stays DONE, :24)
    if (live && tmode) op = (mode & SV_M_SET) ? SV_OP_SET : ((off + s.pos >= 64) ? SV_OP_TEXT : SV_OP_TEXT_SLOW);
    // -- inside a phase, interval not empty: the next extension (:15-22 / :31-37)
    {
      const uint64_t sz = (uint64_t)(s.hi - s.lo);
      const bool few = (mode & SV_M_FEWSET) != 0;
      const bool one = sz == 1 && has_sa;
      const bool small = use_set && sz <= (uint64_t)SV_SET_MAX && has_sa && off >= 64;
      const bool ripe = ((mode & SV_LFC_MASK) >> SV_LFC_SHIFT) >= SV_SET_AFTER;
      const bool bw = walk && !dir;
      const bool w_saset = bw && (few || (!one && small && ripe));
      const bool w_sa = bw && !few && one;
      const bool w_inc = bw && !few && !one && small && !ripe;
      const bool w_lf = walk && !w_saset && !w_sa;
      const int np = dir ? s.pos + 1 : s.pos - 1;
      const bool np_in = np >= s.wrel && np < s.wrel + 64;
      const bool w_fill = w_lf && !np_in && (!dir || np < s.len);
      const bool w_do = w_lf && !w_fill;
      const int sym = sv_ring_sym(g, off + np);
      const int cnew = dir ? svdss_comp(np < s.len ? sym : 0) : sym;
      if (w_inc) s.mode = mode + (1 << SV_LFC_SHIFT);
      if (w_do) { s.pos = np; s.c = cnew; }
      if (w_saset) op = SV_OP_SA_SET;
      if (w_sa) op = SV_OP_SA;
      if (w_fill) op = SV_OP_FILL;
      if (w_do) op = SV_OP_LF;
      if (w_saset || w_sa) a = (int64_t)s.lo;
      if (w_fill) a = (off + np - (dir ? 24 : 40)) >> 4;
    }
    // -- a phase is over
    bool go = body && (mode & SV_M_START);
    if (bend) {
      s.begin = s.pos;
      s.mode = (mode & ~SV_LFC_MASK) | SV_M_DIR | SV_M_START;
      go = true;
    }
    {
      // the SFS [begin, pos] (:38-41) through the streaming assembler (sv_emit), as selects around ONE record store
      const int qs = s.begin, l = s.pos - s.begin + 1;
      const bool chain = (mode & SV_M_CHAIN) != 0;
      const bool joins = assemble && chain && qs + l > s.chain_lo;
      const bool store = fend && (!assemble || (chain && !joins));
      if (store) emit(s.n_sfs, assemble ? s.chain_lo : qs, assemble ? s.chain_end - s.chain_lo : l);
      if (store) ++s.n_sfs;
      const bool open = fend && assemble && !joins;         // a new chain opens with this SFS
      if (fend && assemble) s.chain_lo = qs;
      if (open) s.chain_end = qs + l;
      int m2 = open ? (mode | SV_M_CHAIN) : mode;
      const bool below = fend && qs != 0 && qs < s.stop_lo;
      const bool peek = below && can_peek;                  // ask the neighbour before going on
      const bool count = below && !can_peek;
      const int nb = s.n_below + 1;
      const bool partial = count && nb >= SV_OVERRUN;       // segment finished; the stitcher takes over
      const bool next = fend && qs != 0 && !peek && !partial;   // (qs == 0: :42 -> DONE)
      if (count) s.n_below = nb;
      if (peek) { m2 |= SV_M_PEEK; s.c = 0; op = SV_OP_PEEK; }
      if (partial) m2 |= SV_M_PARTIAL;
      if (next) {
        s.pos = s.pos - 1;                                  // :47
        m2 = (m2 & ~(SV_M_DIR | SV_LFC_MASK)) | SV_M_START;
        go = true;
      }
      if (fend) s.mode = m2;
    }
    // -- a phase starts (backward :12, forward :30): its first K symbols through the table
    bool single = false;
    {
      const bool dir2 = (s.mode & SV_M_DIR) != 0;
      const int st = s.pos;
      const int first = dir2 ? st : st - K + 1;             // lowest read position of the K-mer
      const bool ok = K > 0 && first >= 0 && first + K <= s.len;
      const bool kin = first >= s.wrel && first + K <= s.wrel + 64;   // the K symbols are resident
      uint32_t key = 0;
      const bool good = sv_ring_kmer(g, off + first, K, key);
      const bool tab = go && ok && kin && good;
      const bool kfill = go && ok && !kin;
      single = go && !tab && !kfill;
      if (tab) { op = SV_OP_TABLE; a = (int64_t)(dir2 ? sv_key_revcomp(key, K) : key); }
      if (kfill) { op = SV_OP_FILL; a = (off + first - 20) >> 4; }
    }
    if (!single) break;
    // fewer than K symbols left in this direction, or an N among them: start from the single symbol like the
    // reference does (rb3_fmd_set_intv, :12 / :30), then once more through the body
    {
      const int st = s.pos;
      if (!sv_in_window(s, st)) {
        op = SV_OP_FILL;
        a = (off + st - 24) >> 4;
        break;
      }
      int c = sv_ring_sym(g, off + st);
      if (s.mode & SV_M_DIR) c = svdss_comp(c);
      s.lo = (P)svdss_acc(ix, c);
      s.hi = (P)svdss_acc(ix, c + 1);
      s.mode &= ~(SV_M_START | SV_M_BS_OK);
    }
  }
  SvOp o;
  o.op = op;
  o.a = a;
  return o;
}

// Decide the one memory operation of this iteration.  `off` = absolute buffer
// position of the read's first symbol.  ALU + ring (LDS) reads only.
//
// can_peek: segmented search only -- the lane has a left neighbour.  After every SFS that starts below the
// segment's lower boundary the lane looks (SV_OP_PEEK, like any other memory operation of an iteration)
// whether the neighbour's chain is already known to have started a forward phase at the same `begin`:
// then the two chains are identical from here on and this lane can stop.  A wrong or missing answer only
// lengthens the overrun; sv_stitch verifies everything.
template <class P, class Emit>
SVDSS_HD SvOp sv_decide(SvLane<P>& s, const SvdssDevIndex& ix, const SvRing& g, int64_t off,
                        bool assemble, Emit&& emit, bool can_peek = false, bool use_set = false, bool use_bs = true) {
#ifndef SV_DECIDE_TREE   // (developer build: the tree of early returns everywhere, for an A/B)
  if (!use_bs) return sv_decide_flat(s, ix, g, off, assemble, emit, can_peek, use_set);
#endif
  SvOp o;
  o.op = SV_OP_DONE;
  o.a = 0;
  if (s.len <= 0) return o;
  if (s.mode & SV_M_PARTIAL) return o;
  if (s.mode & SV_M_PEEK) { o.op = SV_OP_PEEK; return o; }
  // One pass of this loop takes a lane from the end of a phase (interval empty) through the start of the next one to
  // its table lookup: the phase-end handling comes first and falls through into the phase start.  (With the phase
  // start on top, as ping_pong.cpp is written, every phase end cost the whole wavefront a second pass over the body.)
  for (;;) {
    SV_COUNT_PASS();
    if (use_bs && (s.mode & SV_M_BS)) {
      const int K = ix.k;
      if (s.mode & SV_M_BS_TAB) {                       // the K-mer at [pos, pos + K) once more, reverse-complemented
        if (s.pos < s.wrel || s.pos + K > s.wrel + 64) {
          o.op = SV_OP_FILL;
          o.a = ((off + s.pos - 20) >> 4);
          return o;
        }
        uint32_t key = 0;
        (void)sv_ring_kmer(g, off + s.pos, K, key);
        o.op = SV_OP_TABLE;
        o.a = sv_key_revcomp(key, K);
        return o;
      }
      if (s.lo < s.hi) {
        if (s.mode & SV_M_BS_ORD) { o.op = SV_OP_BS_ORD; return o; }
        if (s.mode & SV_M_BS_CMP) {
          const int cp = s.pos + K - s.bs_m;            // read positions below cp are still to be compared
          (void)cp;                                     // (off >= 64 for every lane in BS mode: a full window fits)
          o.op = SV_OP_BS_TEXT;
          return o;
        }
        o.op = SV_OP_BS_SA;
        o.a = (int64_t)(s.lo + ((s.hi - s.lo) >> 1));
        return o;
      }
      // every row decided: Q shares lmax symbols with its nearer neighbour and no more with any suffix; the extend that
      // prepends P[e - lmax] empties the interval (ping_pong.cpp:15-22: lmax extends in all, K - 1 of them counted by
      // the table lookup)
      const int lmax = s.begin > s.c ? s.begin : s.c;
      const int e = s.pos + K - 1;
      s.n_ext += lmax - (K - 1);
      s.pos = e - lmax;
      s.lo = 0;
      s.hi = 0;
      s.mode &= ~(SV_M_BS | SV_M_BS_TAB | SV_M_BS_CMP | SV_M_BS_ORD);
    }
    if (s.mode & SV_M_SET) { o.op = SV_OP_SET; return o; }
    if (s.mode & SV_M_TEXT) {
      o.op = (off + s.pos >= 64) ? SV_OP_TEXT : SV_OP_TEXT_SLOW;
      return o;
    }
    if (!(s.mode & SV_M_START)) {
      const bool nonempty = s.hi > s.lo;
      if (!(s.mode & SV_M_DIR)) {
        if (nonempty && s.pos > 0) {                    // ping_pong.cpp:15
          if (s.mode & SV_M_FEWSET) {                   // the survivors of a FEW entry: their text positions, then SET
            o.op = SV_OP_SA_SET;
            o.a = (int64_t)s.lo;
            return o;
          }
          if (s.hi - s.lo == 1 && ix.sa != nullptr) {   // single occurrence: switch to TEXT
            o.op = SV_OP_SA;
            o.a = (int64_t)s.lo;
            return o;
          }
          if (use_set && s.hi - s.lo <= SV_SET_MAX && ix.sa != nullptr && off >= 64) {
            // a few left: if they survived SV_SET_AFTER more symbols they are copies, follow them all in the text
            if (((s.mode & SV_LFC_MASK) >> SV_LFC_SHIFT) >= SV_SET_AFTER) {
              o.op = SV_OP_SA_SET;
              o.a = (int64_t)s.lo;
              return o;
            }
            s.mode += 1 << SV_LFC_SHIFT;
          }
          if (use_bs && (s.mode & SV_M_BS_OK) && s.bs_m >= SV_BS_PROBE) {
            s.mode &= ~SV_M_BS_OK;
            const float n0 = (float)s.tdelta, n8 = (float)(int64_t)(s.hi - s.lo);
            // steps to four occurrences at the rate seen so far: SV_BS_PROBE * log(n8 / 4) / log(n0 / n8)
            const bool slow = n8 >= (float)SV_BS_MIN &&
                              (float)SV_BS_PROBE * svdss_log2f(n8 * 0.25f) > (float)ix.bs_after * svdss_log2f(n0 / n8);
            if (slow) {                                 // back to the K-mer, the rest by binary search
              s.pos += s.bs_m;
              s.n_ext -= s.bs_m;
              s.mode = (s.mode & ~SV_LFC_MASK) | SV_M_BS | SV_M_BS_TAB;
              continue;
            }
          }
          const int np = s.pos - 1;
          if (!sv_in_window(s, np)) {
            o.op = SV_OP_FILL;
            o.a = ((off + np - 40) >> 4);
            return o;
          }
          s.pos = np;
          s.c = sv_ring_sym(g, off + np);               // :21
          o.op = SV_OP_LF;
          return o;
        }
        if (s.pos == 0 && nonempty) return o;           // :24 -> DONE
        s.begin = s.pos;                                // :28
        s.mode = (s.mode & ~SV_LFC_MASK) | SV_M_DIR | SV_M_START;
      } else {
        if (nonempty) {                                 // :31
          const int np = s.pos + 1;
          if (np < s.len && !sv_in_window(s, np)) {
            o.op = SV_OP_FILL;
            o.a = ((off + np - 24) >> 4);
            return o;
          }
          s.pos = np;
          s.c = svdss_comp(np < s.len ? sv_ring_sym(g, off + np) : 0);  // :36, P[l] == 0
          o.op = SV_OP_LF;
          return o;
        }
        sv_emit(s, s.begin, s.pos - s.begin + 1, assemble, emit);  // :38-41
        if (s.begin == 0) return o;                     // :42 -> DONE
        if (s.begin < s.stop_lo) {
          if (can_peek) {                               // ask the neighbour before going on
            s.mode |= SV_M_PEEK;
            s.c = 0;
            o.op = SV_OP_PEEK;
            return o;
          }
          if (++s.n_below >= SV_OVERRUN) {
            s.mode |= SV_M_PARTIAL;                     // segment finished; the stitcher takes over
            return o;
          }
        }
        s.pos = s.pos - 1;                              // :47
        s.mode = (s.mode & ~(SV_M_DIR | SV_LFC_MASK)) | SV_M_START;
      }
    }
    // a phase starts (backward :12, forward :30): its first K symbols through the table
    const int dir = s.mode & SV_M_DIR;
    const int K = ix.k;
    const int st = s.pos;
    const int first = dir ? st : st - K + 1;  // lowest read position of the K-mer
    if (K > 0 && first >= 0 && first + K <= s.len) {
      if (first < s.wrel || first + K > s.wrel + 64) {    // the K symbols must be resident
        o.op = SV_OP_FILL;
        o.a = ((off + first - 20) >> 4);
        return o;
      }
      uint32_t key;
      if (sv_ring_kmer(g, off + first, K, key)) {
        o.op = SV_OP_TABLE;
        o.a = dir ? sv_key_revcomp(key, K) : key;
        return o;
      }
    }
    // fewer than K symbols left in this direction, or an N among them: start
    // from the single symbol like the reference does (rb3_fmd_set_intv, :12 / :30)
    if (!sv_in_window(s, st)) {
      o.op = SV_OP_FILL;
      o.a = ((off + st - 24) >> 4);
      return o;
    }
    int c = sv_ring_sym(g, off + st);
    if (dir) c = svdss_comp(c);
    s.lo = (P)svdss_acc(ix, c);
    s.hi = (P)svdss_acc(ix, c + 1);
    s.mode &= ~(SV_M_START | SV_M_BS_OK);
  }
}

// ---- apply: the result of the iteration's memory operation -----------------

// matches among the first r (< 128) positions of a block: whole words below word r / 32, part of that word
SVDSS_HD int sv_rank128(const uint32_t m[4], int r) {
  const int wi = r >> 5;
  const uint32_t part = (wi == 0 ? m[0] : wi == 1 ? m[1] : wi == 2 ? m[2] : m[3]) & ((1u << (r & 31)) - 1u);
  return svdss_popc(part) + svdss_popc(wi > 0 ? m[0] : 0u) + svdss_popc(wi > 1 ? m[1] : 0u) + svdss_popc(wi > 2 ? m[2] : 0u);
}

template <class P>
SVDSS_HD void sv_apply_lf(SvLane<P>& s, const SvdssDevIndex& ix, const svdss_u4 qa[4],
                          const svdss_u4 qb[4], bool same_block) {
  const int c = s.c;
  const int64_t a = svdss_acc(ix, c);
  if (c >= 1 && c <= 4) {
    const uint32_t code = (uint32_t)(c - 1);
    const uint32_t m0 = (code & 1u) ? 0u : 0xffffffffu;
    const uint32_t m1 = (code & 2u) ? 0u : 0xffffffffu;
    const int rl = (int)(s.lo & (SVDSS_BLOCK_SYMS - 1)), rh = (int)(s.hi & (SVDSS_BLOCK_SYMS - 1));
    uint32_t cl = code == 0 ? qa[0].x : code == 1 ? qa[1].x : code == 2 ? qa[2].x : qa[3].x;
    uint32_t cb = code == 0 ? qb[0].x : code == 1 ? qb[1].x : code == 2 ? qb[2].x : qb[3].x;
    uint32_t ch = same_block ? cl : cb;
    uint32_t ml[4], mh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ml[j] = (qa[j].y ^ m0) & (qa[j].z ^ m1) & ~qa[j].w;
      const uint32_t mb = (qb[j].y ^ m0) & (qb[j].z ^ m1) & ~qb[j].w;
      mh[j] = same_block ? ml[j] : mb;
    }
    const int sl = sv_rank128(ml, rl), sh = sv_rank128(mh, rh);
    s.lo = (P)(a + cl + sl);
    s.hi = (P)(a + ch + sh);
  } else {
    s.lo = (P)(a + svdss_rank_in_block(ix, qa, c, (int64_t)s.lo));
    s.hi = (P)(a + svdss_rank_in_block(ix, same_block ? qa : qb, c, (int64_t)s.hi));
  }
  ++s.n_ext;
  ++s.bs_m;
}

// g / off: the lane's window of read symbols (the extension symbols of a UNIQUE / FEW entry are compared with the
// read symbols next to the K-mer when those are resident); can_set: SET mode may be entered (sv_decide's rule)
template <class P>
SVDSS_HD void sv_apply_table(SvLane<P>& s, const SvdssDevIndex& ix, uint64_t e_lo, uint64_t e_info, const SvRing& g,
                             int64_t off, bool can_set, bool can_bs = false) {
  const int K = ix.k;
  const int dir = s.mode & SV_M_DIR;
  const uint64_t type = e_info >> 62;
  const uint64_t val = e_info & SVDSS_TAB_MASK;
  if (s.mode & SV_M_BS_TAB) {
    // the entry of the reverse-complemented K-mer (as many occurrences as the K-mer itself: MULTI): the rows that start
    // with Q's first K symbols.  Nothing is decided yet; both neighbours are outside the interval (K symbols shared).
    s.mode &= ~SV_M_BS_TAB;
    if (type != SVDSS_TAB_MULTI) {          // (cannot happen with both strands indexed: walk the BWT after all)
      s.mode &= ~SV_M_BS;
      return;
    }
    s.lo = (P)e_lo;
    s.hi = (P)(e_lo + val);
    s.begin = K;
    s.c = K;
    return;
  }
  s.mode &= ~(SV_M_START | SV_M_BS_OK);
  if (type == SVDSS_TAB_EMPTY) {
    const int d = (int)(val & 0xff);      // symbols consumed with a non-empty interval
    // the reference consumed 1 (set_intv) + d extends, the last one emptied the interval
    s.pos = dir ? s.pos + d : s.pos - d;
    s.n_ext += d;
    s.lo = 0;
    s.hi = 0;
    const int df = (int)((val >> 8) & 0xff);
    if (!dir && df > 0) {
      // The forward phase that follows (ping_pong.cpp:28-37) starts at the symbol that emptied the interval and
      // cannot get past the d + 1 symbols that just failed to occur together -- all of them inside this K-mer:
      // its outcome (df leading symbols occur) was computed with the entry, no second lookup.
      s.begin = s.pos;                    // :28
      s.pos = s.begin + df;               // set_intv + df extends, the last one emptied the interval
      s.n_ext += df;
      s.mode = (s.mode & ~SV_LFC_MASK) | SV_M_DIR;
    }
    return;
  }
  s.pos = dir ? s.pos + (K - 1) : s.pos - (K - 1);
  s.n_ext += K - 1;
  if (type == SVDSS_TAB_MULTI) {
    s.lo = (P)e_lo;
    s.hi = (P)(e_lo + val);
    if (can_bs && !dir && val >= (uint64_t)SV_BS_MIN && s.pos > 0) {
      s.bs_m = 0;
      s.tdelta = (int64_t)val;
      s.mode |= ix.bs_after > 0 ? SV_M_BS_OK : (SV_M_BS | SV_M_BS_TAB);
    }
    return;
  }
  // one to four occurrences, each with the SVDSS_TAB_EXT text symbols in front of it: the next extensions
  // (ping_pong.cpp:15-22 backward, :31-37 forward on the reverse-complement key) keep the occurrences whose symbol
  // agrees with the read's -- the interval size the reference looks at is the number that are left
  const bool uniq = type == SVDSS_TAB_UNIQUE;
  const int size = uniq ? 1 : (int)((e_info >> 59) & 7u);
  const uint64_t sa_lo = uniq ? e_lo : (e_lo & SVDSS_TAB_LO_MASK);
  const int pk = s.pos;                   // read position of the K-mer's last consumed symbol
  int steps_max;
  if (!dir) {
    steps_max = pk < SVDSS_TAB_EXT ? pk : SVDSS_TAB_EXT;
    if (pk - steps_max < s.wrel) steps_max = pk - s.wrel;          // (positions below wrel are not resident)
  } else {
    steps_max = s.len - 1 - pk < SVDSS_TAB_EXT ? s.len - 1 - pk : SVDSS_TAB_EXT;
    if (pk + steps_max >= s.wrel + 64) steps_max = s.wrel + 63 - pk;
  }
  if (steps_max < 0) steps_max = 0;
  // All SVDSS_TAB_EXT steps at once: the read's next symbols in the order they are consumed, 3 bits each like the
  // entry's; occurrence j agrees with the first m[j] of them.  Extension after extension keeps the occurrences that
  // agree so far, so the interval empties at step max(m) + 1 -- or not within the steps_max symbols at hand.
  uint32_t rs = 0;
  {
    const int64_t a0 = off + (dir ? pk + 1 : pk - SVDSS_TAB_EXT);   // lowest buffer position of the six
    const int r0 = (int)((a0 & 63) >> 2);
    const int sh = (int)(a0 & 3) * 8;
    const uint32_t w0 = sv_ring_row(g, r0), w1 = sv_ring_row(g, r0 + 1), w2 = sv_ring_row(g, r0 + 2);
    uint64_t w = (uint64_t)(uint32_t)(((((uint64_t)w1) << 32) | w0) >> sh) |
                 ((uint64_t)(uint32_t)(((((uint64_t)w2) << 32) | w1) >> sh) << 32);
    if (!dir) w = __builtin_bswap64(w << 16);                       // nearest symbol first
#pragma unroll
    for (int e = 0; e < SVDSS_TAB_EXT; ++e)
      rs |= (uint32_t)((w >> (8 * e)) & 7u) << (3 * e);             // (positions past steps_max: not looked at)
    if (dir) {
      // complement of all six at once (svdss_comp: 1 <-> 4, 2 <-> 3, 0 and 5 stay): bit 0 flips for 1..4, bit 2 for 1 and 4
      const uint32_t b0 = rs & 0x9249u, b1 = (rs >> 1) & 0x9249u, b2 = (rs >> 2) & 0x9249u;
      const uint32_t t = b2 ^ b0;
      rs ^= (t | b1) | ((t & ~b1) << 2);
    }
  }
  int m[SV_SET_MAX], e = -1;
#pragma unroll
  for (int j = 0; j < SV_SET_MAX; ++j) {
    const uint32_t x = (svdss_tab_ext(e_lo, e_info, j) ^ rs) & 0x3ffffu;
    const uint32_t y = (x | (x >> 1) | (x >> 2)) & 0x9249u;         // bit 3i: symbol i differs
    int mj = y ? (__builtin_ctz(y) * 11) >> 5 : SVDSS_TAB_EXT;
    if (mj > steps_max) mj = steps_max;
    m[j] = j < size ? mj : -1;
    if (m[j] > e) e = m[j];
  }
  const bool emptied = e < steps_max;
  int alive = 0;
#pragma unroll
  for (int j = 0; j < SV_SET_MAX; ++j) if (m[j] >= e) alive |= 1 << j;
  if (emptied) {                          // extension e + 1 emptied the interval
    s.pos = dir ? pk + e + 1 : pk - e - 1;
    s.n_ext += e + 1;
    s.lo = 0;
    s.hi = 0;
    return;
  }
  if (dir || e == 0) {
    // forward and still occurring after the symbols at hand (rare: the string with the error in it goes on
    // matching), or nothing resident to compare with: go on from the K-mer's interval as before
    s.lo = (P)sa_lo;
    s.hi = (P)(sa_lo + (uint64_t)size);
    if (uniq && !dir && pk > 0) {         // straight to TEXT mode: no SA lookup needed
      s.tdelta = (int64_t)(val & SVDSS_TAB_POS_MASK) - pk;
      s.mode |= SV_M_TEXT;
    }
    return;
  }
  // backward, e symbols further, at least one occurrence left
  s.pos = pk - e;
  s.n_ext += e;
  if (s.pos == 0) {                       // ping_pong.cpp:24: prefix matched, size != 0
    s.lo = 0;
    s.hi = 1;
    return;
  }
  if (uniq) {
    s.lo = (P)sa_lo;
    s.hi = (P)(sa_lo + 1);
    s.tdelta = (int64_t)(val & SVDSS_TAB_POS_MASK) - pk;
    s.mode |= SV_M_TEXT;
    return;
  }
  if ((alive & (alive - 1)) == 0) {       // one left: its text position (suffix array row sa_lo + j), then TEXT
    const int j = alive == 1 ? 0 : alive == 2 ? 1 : alive == 4 ? 2 : 3;
    s.lo = (P)(sa_lo + (uint64_t)j);
    s.hi = (P)(sa_lo + (uint64_t)j + 1);
    s.c = e;
    s.mode |= SV_M_SHIFT;
    return;
  }
  s.lo = (P)sa_lo;
  s.hi = (P)(sa_lo + (uint64_t)size);
  if (can_set && ix.sa != nullptr) {      // several left after SVDSS_TAB_EXT more symbols: copies; follow them in the text
    s.c = e;
    s.mode = (s.mode & ~(((1 << SV_SET_MAX) - 1) << SV_SET_SHIFT)) | (alive << SV_SET_SHIFT) | SV_M_SHIFT | SV_M_FEWSET;
    return;
  }
  // no SET mode here: forget the e symbols, walk the BWT from the K-mer's interval
  s.pos = pk;
  s.n_ext -= e;
}

template <class P>
SVDSS_HD void sv_apply_sa(SvLane<P>& s, int64_t text_pos) {
  // (SV_M_SHIFT: the row belongs to the K-mer of the last table lookup, s.c read symbols back)
  s.tdelta = text_pos - s.pos - ((s.mode & SV_M_SHIFT) ? s.c : 0);
  s.mode = (s.mode & ~SV_M_SHIFT) | SV_M_TEXT;
}

// SA_SET: text positions of the hi - lo <= SV_SET_MAX occurrences (suffix array entries lo, lo+1, ...)
template <class P>
SVDSS_HD void sv_apply_sa_set(SvLane<P>& s, const SvSet& ts, const int64_t text_pos[SV_SET_MAX]) {
  const int n = (int)(s.hi - s.lo);
  const int shift = (s.mode & SV_M_SHIFT) ? s.c : 0;   // rows of the K-mer of the last table lookup, s.c symbols back
#pragma unroll
  for (int i = 0; i < SV_SET_MAX; ++i)
    if (i < n) ts.base[i * ts.stride] = text_pos[i] - s.pos - shift;
  if (s.mode & SV_M_FEWSET) s.mode = (s.mode & ~(SV_M_SHIFT | SV_M_FEWSET)) | SV_M_SET;   // alive bits are set already
  else s.mode |= SV_M_SET | (((1 << n) - 1) << SV_SET_SHIFT);
}

// SET: tw[i] = the 16 text bytes of occurrence i for read positions [pos-16, pos), rb = the read's.  Every
// extension keeps the copies whose next text byte equals the read symbol (what rb3_fmd_extend does to the
// interval, ping_pong.cpp:15-22); the phase goes on while at least one is left.
template <class P>
SVDSS_HD void sv_apply_set(SvLane<P>& s, const SvSet& ts, const svdss_u4 tw[SV_SET_MAX], const svdss_u4& rb) {
  const int alive = (s.mode >> SV_SET_SHIFT) & ((1 << SV_SET_MAX) - 1);
  int m[SV_SET_MAX];
  int best = -1;
#pragma unroll
  for (int i = 0; i < SV_SET_MAX; ++i) {
    const uint32_t x3 = tw[i].w ^ rb.w, x2 = tw[i].z ^ rb.z, x1 = tw[i].y ^ rb.y, x0 = tw[i].x ^ rb.x;
    // matching symbols counted down from pos-1 (byte 15 of the window)
    const uint32_t v = x3 ? x3 : x2 ? x2 : x1 ? x1 : x0;         // the highest word that differs (selects, no branches)
    const int top = x3 ? 3 : x2 ? 7 : x1 ? 11 : 15;
    const int k = v ? top - ((31 - __builtin_clz(v | 1u)) >> 3) : SV_SET_WIN;
    m[i] = ((alive >> i) & 1) ? k : -1;
    if (m[i] > best) best = m[i];
  }
  const int avail = s.pos < SV_SET_WIN ? s.pos : SV_SET_WIN;   // symbols left before the read start
  if (best >= avail) {
    // at least one copy agrees with every remaining symbol of this window
    int keep = 0;
#pragma unroll
    for (int i = 0; i < SV_SET_MAX; ++i) if (m[i] >= avail) keep |= 1 << i;
    s.pos -= avail;
    s.n_ext += avail;
    s.mode = (s.mode & ~(((1 << SV_SET_MAX) - 1) << SV_SET_SHIFT)) | (keep << SV_SET_SHIFT);
    if (s.pos == 0) {                              // ping_pong.cpp:24: prefix matched, size != 0
      s.mode &= ~(SV_M_SET | (((1 << SV_SET_MAX) - 1) << SV_SET_SHIFT));
      s.lo = 0;
      s.hi = 1;
    } else if ((keep & (keep - 1)) == 0) {         // one copy left: the 64-symbol TEXT mode takes over
      const int i = keep == 1 ? 0 : keep == 2 ? 1 : keep == 4 ? 2 : 3;
      s.tdelta = ts.base[i * ts.stride];
      s.mode = (s.mode & ~(SV_M_SET | (((1 << SV_SET_MAX) - 1) << SV_SET_SHIFT))) | SV_M_TEXT;
    }
  } else {
    // after `best` agreeing symbols the next one disagrees with every copy: that extend empties the interval
    s.pos -= best + 1;
    s.n_ext += best + 1;
    s.mode &= ~(SV_M_SET | (((1 << SV_SET_MAX) - 1) << SV_SET_SHIFT));
    s.lo = 0;
    s.hi = 0;
  }
}

// PEEK: q[i] / written[i] = SFS start and "already produced" of the neighbour's records nb_cur, nb_cur+1, ...
// (production order = descending start).  Decides whether the neighbour started a forward phase at s.begin.
template <class P>
SVDSS_HD void sv_apply_peek(SvLane<P>& s, const int32_t q[SV_PEEK_RECS], const bool written[SV_PEEK_RECS],
                            int32_t& nb_cur, int32_t nb_cap) {
  bool decided = false, found = false;
  int adv = 0;
#pragma unroll
  for (int i = 0; i < SV_PEEK_RECS; ++i) {
    if (decided) continue;
    if (nb_cur + i >= nb_cap || !written[i]) { decided = true; continue; }
    if (q[i] == s.begin) { decided = true; found = true; continue; }
    if (q[i] < s.begin) { decided = true; continue; }
    ++adv;
  }
  nb_cur += adv;
  if (!decided && ++s.c < SV_PEEK_OPS) return;       // look at the next records
  s.mode &= ~SV_M_PEEK;
  if (found || ++s.n_below >= SV_OVERRUN) {
    s.mode |= SV_M_PARTIAL;                           // segment finished; the stitcher takes over
    return;
  }
  s.pos = s.pos - 1;                                  // ping_pong.cpp:47
  s.mode = (s.mode & ~(SV_M_DIR | SV_LFC_MASK)) | SV_M_START;
}

// TEXT: ta[] = text bytes, rb[] = read bytes, both for read positions
// [pos-64, pos) (byte 0 of ta[0]/rb[0] <-> position pos-64).  The lane consumes
// symbols pos-1, pos-2, ... while they agree (one rb3_fmd_extend each,
// ping_pong.cpp:15-22 with a size-1 interval), stops at the read start.
// number of symbols that agree, counted down from the last of the 64 (byte 63 of the windows)
SVDSS_HD int sv_text_matched(const svdss_u4 ta[4], const svdss_u4 rb[4]) {
  uint32_t x[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[4 * j + 0] = ta[j].x ^ rb[j].x;
    x[4 * j + 1] = ta[j].y ^ rb[j].y;
    x[4 * j + 2] = ta[j].z ^ rb[j].z;
    x[4 * j + 3] = ta[j].w ^ rb[j].w;
  }
  // highest differing byte of the 64: binary selection, upper half first
  int idx = 0;
  uint32_t v8[8], v4[4], v2[2], v1;
  {
    const bool up = (x[8] | x[9] | x[10] | x[11] | x[12] | x[13] | x[14] | x[15]) != 0;
    idx = up ? 8 : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = up ? x[8 + i] : x[i];
  }
  {
    const bool up = (v8[4] | v8[5] | v8[6] | v8[7]) != 0;
    idx += up ? 4 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = up ? v8[4 + i] : v8[i];
  }
  {
    const bool up = (v4[2] | v4[3]) != 0;
    idx += up ? 2 : 0;
    v2[0] = up ? v4[2] : v4[0];
    v2[1] = up ? v4[3] : v4[1];
  }
  {
    const bool up = v2[1] != 0;
    idx += up ? 1 : 0;
    v1 = up ? v2[1] : v2[0];
  }
  // number of matching symbols counted down from pos-1
  int matched;
  if (v1 == 0) matched = 64;
  else matched = 63 - (4 * idx + ((31 - __builtin_clz(v1)) >> 3));
  return matched;
}

template <class P>
SVDSS_HD void sv_apply_text(SvLane<P>& s, const svdss_u4 ta[4], const svdss_u4 rb[4]) {
  const int matched = sv_text_matched(ta, rb);
  const int avail = s.pos < 64 ? s.pos : 64;     // symbols left before the read start
  if (matched >= avail) {
    // every remaining symbol of this window agrees
    s.pos -= avail;
    s.n_ext += avail;
    if (s.pos == 0) {                            // ping_pong.cpp:24: prefix matched, size != 0
      s.mode &= ~SV_M_TEXT;
      s.lo = 0;
      s.hi = 1;
    }
  } else {
    // symbol pos-1-matched disagrees: that extend empties the interval (:15 fails next)
    s.pos -= matched + 1;
    s.n_ext += matched + 1;
    s.mode &= ~SV_M_TEXT;
    s.lo = 0;
    s.hi = 0;
  }
}

// byte-wise variant for the first 64 bytes of the whole read buffer (where a
// 64-byte window ending at pos would start before the buffer)
template <class P>
SVDSS_HD void sv_apply_text_slow(SvLane<P>& s, const uint8_t* text, const uint8_t* reads, int64_t off) {
  while (s.pos > 0) {
    const int np = s.pos - 1;
    ++s.n_ext;
    s.pos = np;
    if (text[s.tdelta + np] != reads[off + np]) {
      s.mode &= ~SV_M_TEXT;
      s.lo = 0;
      s.hi = 0;
      return;
    }
  }
  s.mode &= ~SV_M_TEXT;
  s.lo = 0;
  s.hi = 1;
}


// ---- BS (see the mode bits above) ----

// the middle '$' of the record pair (record $ revcomp $) that holds text position x: dsorted = the text positions of
// all '$', ascending (suffix array rows 0 .. n_dollar - 1, sorted); x -> 2 * mid - x maps a position to the position
// of the complementary base in the other strand
SVDSS_HD int64_t sv_mirror(const int64_t* dsorted, int n_d, int64_t x) {
  int a = 0, b = n_d;                       // first '$' position >= x
  while (a < b) { const int m = (a + b) >> 1; if (dsorted[m] < x) a = m + 1; else b = m; }
  int mid = a & ~1;                         // '$' 2c closes record c, '$' 2c + 1 its reverse complement: x lies in pair a / 2
  if (mid >= n_d) mid = n_d - 2 < 0 ? 0 : n_d - 2;
  return 2 * dsorted[mid] - x;
}

// BS_SA: text position of row M = (lo + hi) / 2
template <class P>
SVDSS_HD void sv_apply_bs_sa(SvLane<P>& s, const SvdssDevIndex& ix, int64_t text_pos, const int64_t* dsorted, int n_d) {
  const int e = s.pos + ix.k - 1;
  s.tdelta = sv_mirror(dsorted, n_d, text_pos) - e;
  s.bs_m = s.begin < s.c ? s.begin : s.c;   // rows between the two neighbours share at least this much with Q
  s.mode |= SV_M_BS_CMP;
}

// what a comparison of the middle row found: `matched` more symbols agree; then either the read's start was reached
// (at_start: every symbol of P[0..e] occurs -- ping_pong.cpp:24) or the read symbol rsym met the text symbol tsym
template <class P>
SVDSS_HD void sv_bs_outcome(SvLane<P>& s, const SvdssDevIndex& ix, int matched, bool at_start, bool more, int rsym, int tsym) {
  const int K = ix.k;
  s.bs_m += matched;
  if (at_start) {
    const int e = s.pos + K - 1;
    s.n_ext += e - (K - 1);                 // e extends in all
    s.pos = 0;
    s.lo = 0;
    s.hi = 1;
    s.mode &= ~(SV_M_BS | SV_M_BS_TAB | SV_M_BS_CMP | SV_M_BS_ORD);
    return;
  }
  if (more) return;                         // the whole window agreed: the next one
  // Q[m] = comp(rsym), the row's symbol there = comp(tsym) (the text is read in the other strand)
  const P M = s.lo + ((s.hi - s.lo) >> 1);
  if (svdss_comp(tsym) < svdss_comp(rsym)) { s.lo = M + 1; s.begin = s.bs_m; }   // row M < Q
  else { s.hi = M; s.c = s.bs_m; }
  s.mode &= ~SV_M_BS_CMP;
}

// BS_TEXT: ta[] / rb[] = text / read bytes for read positions [cp - 64, cp), cp = pos + K - bs_m (the TEXT mode's
// comparison).  A mismatch leaves the lane in SV_M_BS_ORD: the two symbols that differ decide on which side of Q the
// middle row lies (sv_apply_bs_ord).
template <class P>
SVDSS_HD void sv_apply_bs_text(SvLane<P>& s, const SvdssDevIndex& ix, const svdss_u4 ta[4], const svdss_u4 rb[4]) {
  const int matched = sv_text_matched(ta, rb);
  const int cp = s.pos + ix.k - s.bs_m;
  const int avail = cp < 64 ? cp : 64;
  if (matched >= avail) sv_bs_outcome(s, ix, avail, cp - avail == 0, cp - avail > 0, 0, 0);
  else { s.bs_m += matched; s.mode |= SV_M_BS_ORD; }
}

// BS_ORD: rsym / tsym = the read's and the text's symbol at read position e - bs_m (they differ)
template <class P>
SVDSS_HD void sv_apply_bs_ord(SvLane<P>& s, const SvdssDevIndex& ix, int rsym, int tsym) {
  s.mode &= ~SV_M_BS_ORD;
  sv_bs_outcome(s, ix, 0, false, false, rsym, tsym);
}

// the same byte by byte (the first 64 bytes of the whole read buffer)
template <class P>
SVDSS_HD void sv_apply_bs_text_slow(SvLane<P>& s, const SvdssDevIndex& ix, const uint8_t* text, const uint8_t* reads, int64_t off) {
  int cp = s.pos + ix.k - s.bs_m;
  int matched = 0;
  while (cp > 0) {
    const int r = reads[off + cp - 1], t = text[s.tdelta + cp - 1];
    if (r != t) { sv_bs_outcome(s, ix, matched, false, false, r, t); return; }
    ++matched;
    --cp;
  }
  sv_bs_outcome(s, ix, matched, true, false, 0, 0);
}

// ---- k-mer table construction (one entry per key; device kernel and emulator) ----

template <class P>
SVDSS_HD void sv_table_entry(const SvdssDevIndex& ix, uint32_t key, int K, uint64_t& e_lo, uint64_t& e_info) {
  // W[i] = ((key >> 2i) & 3) + 1; symbols are consumed from W[K-1] down to W[0]
  int c = (int)((key >> (2 * (K - 1))) & 3u) + 1;
  int64_t lo = svdss_acc(ix, c), hi = svdss_acc(ix, c + 1);
  int d = 0;  // symbols consumed that left a non-empty interval
  if (hi > lo) {
    d = 1;
    for (int i = K - 2; i >= 0; --i) {
      c = (int)((key >> (2 * i)) & 3u) + 1;
      const int64_t a = svdss_acc(ix, c);
      const int64_t nlo = a + svdss_rank_in_block(ix, ix.blocks + 4 * (lo >> SVDSS_BLOCK_SHIFT), c, lo);
      const int64_t nhi = a + svdss_rank_in_block(ix, ix.blocks + 4 * (hi >> SVDSS_BLOCK_SHIFT), c, hi);
      lo = nlo;
      hi = nhi;
      if (hi <= lo) break;
      ++d;
    }
  }
  if (d == K) {
    const uint64_t size = (uint64_t)(hi - lo);
    e_lo = (uint64_t)lo;
    if (size == 1 && ix.sa != nullptr) {
      e_info = (SVDSS_TAB_UNIQUE << 62) | ((uint64_t)svdss_ext_symbols(ix, lo) << 40) | (uint64_t)((const P*)ix.sa)[lo];
    } else if (size >= 2 && size <= 4 && ix.sa != nullptr) {
      e_info = (SVDSS_TAB_FEW << 62) | (size << 59);
      for (int j = 0; j < (int)size; ++j) {
        const uint64_t x = svdss_ext_symbols(ix, lo + j);
        if (j < 3) e_info |= x << (18 * j); else e_lo |= x << 36;
      }
    } else {
      e_info = (SVDSS_TAB_MULTI << 62) | size;
    }
  } else {
    // W[K-1-d .. K-1] (d + 1 symbols) does not occur: how many of its leading symbols do -- the forward phase the
    // reference starts at W[K-1-d] (interval of revcomp, extended with complemented symbols, ping_pong.cpp:30-37)
    int df = 0;
    if (d >= 1) {
      int cc = svdss_comp((int)((key >> (2 * (K - 1 - d))) & 3u) + 1);
      int64_t flo = svdss_acc(ix, cc), fhi = svdss_acc(ix, cc + 1);
      if (fhi > flo) {
        df = 1;
        for (int i = K - d; i <= K - 1; ++i) {
          cc = svdss_comp((int)((key >> (2 * i)) & 3u) + 1);
          const int64_t a = svdss_acc(ix, cc);
          const int64_t nlo = a + svdss_rank_in_block(ix, ix.blocks + 4 * (flo >> SVDSS_BLOCK_SHIFT), cc, flo);
          const int64_t nhi = a + svdss_rank_in_block(ix, ix.blocks + 4 * (fhi >> SVDSS_BLOCK_SHIFT), cc, fhi);
          flo = nlo;
          fhi = nhi;
          if (fhi <= flo) break;
          ++df;
        }
      }
    }
    if (df > d) df = 0;   // (cannot happen with both strands indexed; no shortcut then)
    e_lo = 0;
    e_info = (SVDSS_TAB_EMPTY << 62) | (uint64_t)d | ((uint64_t)df << 8);
  }
}

// ---- segmented search: stitching the per-segment chains of one read -------------------
//
// A read is cut into C segments; the lane of segment j starts a FRESH backward phase at the
// segment's last position (exactly what ping_pong_search does at l-1, ping_pong.cpp:8-12)
// and runs the unmodified algorithm over the whole read, stopping SV_OVERRUN SFS past its
// lower boundary.  The state of ping_pong_search at the start of a forward phase is fully
// determined by `begin` (ping_pong.cpp:28-30 resets the interval), so as soon as the chain
// coming from the right (the true one) starts a forward phase at a position where the fresh
// chain of the next segment also started one, the two are identical from there on.  The
// stitcher looks for that shared SFS start; if the overrun was too short to contain it the
// read is searched again unsegmented (exactness never depends on the heuristic).
//
// rec[] per segment: {qs, len, ext_at_begin, -} in production order (descending qs), where
// ext_at_begin = extensions counted when the forward phase of that SFS started.
struct SvSegInfo {
  int32_t n_rec;     // records produced (may exceed cap -> overflow)
  int32_t cap;
  int32_t ext_total;
  int32_t complete;  // 1: reached the start of the read, 0: stopped after the overrun
};

// Returns false if the read must be redone unsegmented.  On success take_lo/take_hi give the
// record range of every segment that belongs to the read's chain (empty range = superseded)
// and *n_ext the reference's extension count for the whole read.
template <class GetRec>
SVDSS_HD bool sv_stitch(int n_seg, const SvSegInfo* info, const int32_t* seg_lo, GetRec&& rec_qs_ext,
                        int32_t* take_lo, int32_t* take_hi, int64_t* n_ext) {
  for (int j = 0; j < n_seg; ++j) {
    take_lo[j] = take_hi[j] = 0;
    if (info[j].n_rec > info[j].cap) return false;
  }
  int a = n_seg - 1;           // current segment (the true chain lives in it)
  int32_t lo = 0;              // first record of segment a that belongs to the chain
  int64_t ext = 0;
  int32_t ext_base = 0;        // ext_at_begin of record lo of segment a (0 for the rightmost)
  for (;;) {
    if (info[a].complete || a == 0) {
      if (!info[a].complete) return false;   // segment 0 always runs to the start of the read
      take_lo[a] = lo;
      take_hi[a] = info[a].n_rec;
      ext += info[a].ext_total - ext_base;
      break;
    }
    const int b = a - 1;
    const int32_t x = seg_lo[a];             // positions < x belong to segment b
    int32_t ia = lo, ib = 0;
    int32_t qa, ea, qb, eb;
    bool found = false;
    {
      // skip a's own territory: records are in descending start order, find the first one below x
      int32_t hi_ = info[a].n_rec;
      while (ia < hi_) {
        const int32_t mid = ia + ((hi_ - ia) >> 1);
        rec_qs_ext(a, mid, qa, ea);
        if (qa >= x) ia = mid + 1; else hi_ = mid;
      }
    }
    while (ia < info[a].n_rec && ib < info[b].n_rec) {
      rec_qs_ext(a, ia, qa, ea);
      rec_qs_ext(b, ib, qb, eb);
      if (qa == qb) { found = true; break; }
      if (qa > qb) ++ia; else ++ib;
    }
    if (!found) return false;
    take_lo[a] = lo;
    take_hi[a] = ia;
    ext += ea - ext_base;
    a = b;
    lo = ib;
    ext_base = eb;
  }
  *n_ext = ext;
  return true;
}
