/*
 * oracle/svdss_oracle_call.c -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the two third-party DP computations that `SVDSS call`
 * reaches at /root/reference/caller.cpp:348 (ksw2 `ksw_extd2_sse`, global,
 * dual affine gap, full band) and caller.cpp:456,458 (`rapidfuzz::fuzz::ratio`).
 *
 * PARITY STATUS: **parity unpinned** against the libraries themselves -- ksw2
 * (unpinned HEAD, CMakeLists.txt:116-118) and rapidfuzz-cpp v1.10.4
 * (CMakeLists.txt:131-133) are git-fetched at build time and absent here, and
 * the reference has no tests.  What IS pinned:
 *   - the alignment SCORE (VCF `AS`, caller.cpp:351) is the optimum of a
 *     well-defined recurrence; orc_global_score_general() recomputes it with an
 *     independent O(n m (n+m)) general-gap DP and tests compare the two;
 *   - the ratio is a closed form of the LCS length.
 * The CIGAR among co-optimal alignments follows ksw2's published conventions
 * (left-aligned gaps; H-source priority diagonal > E > F > E2 > F2 by strict
 * comparison; a gap state continues only if strictly better than re-opening)
 * as restated in SURVEY App. B.3 -- [UPSTREAM-UNVERIFIED].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NEG (-0x20000000)

/* gap of length l: min(q + l e, q2 + l e2)  (caller.cpp:333: 16,2,41,1) */
static inline int orc_gap(int l, int q, int e, int q2, int e2) {
  int a = q + l * e, b = q2 + l * e2;
  return a < b ? a : b;
}

/*
 * ksw_extd2_sse(km=0, qlen, query, tlen, target, m, mat, q, e, q2, e2,
 *               w=-1, zdrop=-1, end_bonus=-1, flag=0, &ez)   (caller.cpp:348-349)
 * = global alignment of query against target.  CIGAR ops as ksw2 packs them:
 * len<<4 | op, op 0=M 1=I(consumes query) 2=D(consumes target) (caller.cpp:353-355
 * indexes "MID").  Returns n_cigar; *score = ez.score.
 * dirs (tl*ql bytes) may be NULL (allocated internally).
 */
int64_t orc_ksw_extd2_global(const uint8_t *query, int ql, const uint8_t *target, int tl, int m,
                             const int8_t *mat, int q, int e, int q2, int e2, int32_t *score,
                             uint32_t *cigar, int64_t cigar_cap) {
  if (ql <= 0 || tl <= 0) {
    /* ksw2 returns without touching ez for empty input; caller.cpp then reads score 0, no CIGAR */
    *score = 0;
    return 0;
  }
  /* column arrays indexed by i (target); we sweep j (query) in the outer loop */
  int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tl + 1)); /* H(i, j-1) then H(i, j) */
  int32_t *F = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tl + 1)); /* F(i, j): gap consuming query */
  int32_t *F2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tl + 1));
  uint8_t *dir = (uint8_t *)malloc((size_t)tl * (size_t)ql);
  /* column j = -1 */
  for (int i = 0; i < tl; ++i) {
    H[i] = -orc_gap(i + 1, q, e, q2, e2);
    F[i] = ORC_NEG;
    F2[i] = ORC_NEG;
  }
  for (int j = 0; j < ql; ++j) {
    int32_t hdiag = j == 0 ? 0 : -orc_gap(j, q, e, q2, e2);      /* H(-1, j-1) */
    int32_t hup = -orc_gap(j + 1, q, e, q2, e2);                  /* H(-1, j)   */
    int32_t E = ORC_NEG, E2 = ORC_NEG;                            /* E(-1, j): none */
    for (int i = 0; i < tl; ++i) {
      /* E(i,j) = max(H(i-1,j) - q, E(i-1,j)) - e : gap consuming target (deletion) */
      int32_t Ein = (hup - q > E ? hup - q : E) - e;
      int32_t E2in = (hup - q2 > E2 ? hup - q2 : E2) - e2;
      /* F(i,j) = max(H(i,j-1) - q, F(i,j-1)) - e : gap consuming query (insertion) */
      int32_t Fin = (H[i] - q > F[i] ? H[i] - q : F[i]) - e;
      int32_t F2in = (H[i] - q2 > F2[i] ? H[i] - q2 : F2[i]) - e2;
      int32_t z = hdiag + mat[target[i] * m + query[j]];
      uint8_t d = 0;
      if (Ein > z) { d = 1; z = Ein; }
      if (Fin > z) { d = 2; z = Fin; }
      if (E2in > z) { d = 3; z = E2in; }
      if (F2in > z) { d = 4; z = F2in; }
      /* continuation bits: the gap entering the NEXT cell extends iff strictly better than opening */
      if (Ein > z - q) d |= 0x08;
      if (Fin > z - q) d |= 0x10;
      if (E2in > z - q2) d |= 0x20;
      if (F2in > z - q2) d |= 0x40;
      dir[(size_t)i * ql + j] = d;
      hdiag = H[i];
      H[i] = z;
      hup = z;
      E = Ein; E2 = E2in;
      F[i] = Fin; F2[i] = F2in;
    }
  }
  *score = H[tl - 1];
  /* ksw_backtrack from (tl-1, ql-1) */
  int64_t n = 0;
  int i = tl - 1, j = ql - 1, state = 0;
#define PUSH(op_, len_)                                                        \
  do {                                                                         \
    if (n > 0 && (cigar[n - 1] & 0xf) == (uint32_t)(op_)) cigar[n - 1] += (uint32_t)(len_) << 4; \
    else { if (n >= cigar_cap) { n = -1; goto done; } cigar[n++] = ((uint32_t)(len_) << 4) | (uint32_t)(op_); } \
  } while (0)
  while (i >= 0 && j >= 0) {
    uint8_t tmp = dir[(size_t)i * ql + j];
    if (state == 0) state = tmp & 7;
    else if (!((tmp >> (state + 2)) & 1)) state = 0;
    if (state == 0) state = tmp & 7;
    if (state == 0) { PUSH(0, 1); --i; --j; }
    else if (state == 1 || state == 3) { PUSH(2, 1); --i; }
    else { PUSH(1, 1); --j; }
  }
  if (i >= 0) PUSH(2, i + 1);
  if (j >= 0) PUSH(1, j + 1);
  for (int64_t a = 0; a < n / 2; ++a) { uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
done:
#undef PUSH
  free(H); free(F); free(F2); free(dir);
  return n;
}

/* Independent check of the optimum: global alignment with an arbitrary gap
 * function g(l) = min(q + l e, q2 + l e2), O(n m (n + m)); tiny inputs only. */
int32_t orc_global_score_general(const uint8_t *query, int ql, const uint8_t *target, int tl, int m,
                                 const int8_t *mat, int q, int e, int q2, int e2) {
  int W = ql + 1;
  int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tl + 1) * (size_t)W);
  for (int i = 0; i <= tl; ++i)
    for (int j = 0; j <= ql; ++j) {
      int32_t best;
      if (i == 0 && j == 0) { S[0] = 0; continue; }
      best = ORC_NEG;
      if (i > 0 && j > 0) best = S[(i - 1) * W + j - 1] + mat[target[i - 1] * m + query[j - 1]];
      for (int k = 1; k <= i; ++k) { int32_t v = S[(i - k) * W + j] - orc_gap(k, q, e, q2, e2); if (v > best) best = v; }
      for (int k = 1; k <= j; ++k) { int32_t v = S[i * W + j - k] - orc_gap(k, q, e, q2, e2); if (v > best) best = v; }
      S[i * W + j] = best;
    }
  int32_t r = S[tl * W + ql];
  free(S);
  return r;
}

/* score of a CIGAR under the same model (used to check that the traceback is co-optimal) */
int32_t orc_cigar_score(const uint8_t *query, int ql, const uint8_t *target, int tl, int m, const int8_t *mat,
                        int q, int e, int q2, int e2, const uint32_t *cigar, int64_t n) {
  int i = 0, j = 0;
  int32_t s = 0;
  for (int64_t k = 0; k < n; ++k) {
    int len = (int)(cigar[k] >> 4), op = (int)(cigar[k] & 0xf);
    if (op == 0) { for (int x = 0; x < len; ++x) s += mat[target[i + x] * m + query[j + x]]; i += len; j += len; }
    else if (op == 1) { s -= orc_gap(len, q, e, q2, e2); j += len; }
    else { s -= orc_gap(len, q, e, q2, e2); i += len; }
  }
  return (i == tl && j == ql) ? s : ORC_NEG;
}

/* rapidfuzz::fuzz::ratio(s1, s2), score_cutoff 0 (caller.cpp:456,458):
 * LCS by the plain O(n m) DP, then the library's floating-point operation order. */
int64_t orc_lcs(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb) {
  int64_t *row = (int64_t *)calloc((size_t)lb + 1, sizeof(int64_t));
  for (int64_t i = 0; i < la; ++i) {
    int64_t diag = 0;
    for (int64_t j = 0; j < lb; ++j) {
      int64_t up = row[j + 1];
      int64_t v = a[i] == b[j] ? diag + 1 : (up > row[j] ? up : row[j]);
      diag = up;
      row[j + 1] = v;
    }
  }
  int64_t r = row[lb];
  free(row);
  return r;
}

double orc_fuzz_ratio(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb) {
  int64_t maximum = la + lb;
  int64_t dist = maximum - 2 * orc_lcs(a, la, b, lb);
  double norm_dist = maximum ? (double)dist / (double)maximum : 0.0;
  double norm_sim = 1.0 - norm_dist;
  return norm_sim * 100.0;
}
