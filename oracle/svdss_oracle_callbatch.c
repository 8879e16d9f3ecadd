/*
 * svdss_oracle_callbatch.c -- TEST INFRASTRUCTURE / CPU baseline leg only (never linked or loaded by the product).
 *
 * The call-side DP of Caller::pcall over a list of sub-clusters, on host threads, the way the reference runs it:
 * `#pragma omp parallel for` over the clusters (caller.cpp:319-321), per sub-cluster
 *   run_poa(seqs)                       caller.cpp:257-308  -> orc_poa_consensus      (svdss_oracle_poa.c)
 *   ksw_extd2_sse(consensus, window)    caller.cpp:332-355  -> orc_ksw_extd2_global   (svdss_oracle_call.c)
 * and, once all are done, fuzz::ratio of adjacent consensus sequences as filter_sv_chains meets them on neighbouring
 * alleles (caller.cpp:456-458) -> orc_fuzz_ratio.  bench.py times this on a sample of the sub-clusters of a step
 * (cpu_baseline.call) and compares consensus lengths, alignment scores and ratios with what the HIP kernels returned
 * for the same sub-clusters.  parity unpinned against abPOA / ksw2 / rapidfuzz themselves (see the headers of the two
 * files above).
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int64_t orc_poa_consensus(const uint8_t *seqs, const int64_t *offs, int n, uint8_t *cons, int64_t cap);
int64_t orc_ksw_extd2_global(const uint8_t *query, int ql, const uint8_t *target, int tl, int m, const int8_t *mat, int q,
                             int e, int q2, int e2, int32_t *score, uint32_t *cigar, int64_t cigar_cap);
double orc_fuzz_ratio(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb);

/* seqs / seq_off: all sub-reads, concatenated; cluster_off[n_sub + 1]: sub-reads of sub-cluster k;
 * refs / ref_off[n_sub + 1]: its reference window; mat: 5 x 5.  Outputs (caller-owned): cons_len[n_sub],
 * score[n_sub], n_cigar[n_sub], ratio[n_sub - 1] (adjacent pairs); cons_cap_per = room per consensus in cons_out
 * (may be NULL).  Returns 0, or -1 when a consensus did not fit / an allocation failed. */
int orc_call_batch(const uint8_t *seqs, const int64_t *seq_off, const int64_t *cluster_off, int64_t n_sub,
                   const uint8_t *refs, const int64_t *ref_off, const int8_t *mat, int threads, int64_t *cons_len,
                   int32_t *score, int64_t *n_cigar, double *ratio) {
  if (threads < 1) threads = 1;
  uint8_t **cons = (uint8_t **)calloc((size_t)(n_sub > 0 ? n_sub : 1), sizeof(uint8_t *));
  if (!cons) return -1;
  int bad = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (int64_t k = 0; k < n_sub; ++k) {
    const int64_t s0 = cluster_off[k], s1 = cluster_off[k + 1];
    const int n = (int)(s1 - s0);
    int64_t longest = 0, total = seq_off[s1] - seq_off[s0];
    for (int64_t i = s0; i < s1; ++i)
      if (seq_off[i + 1] - seq_off[i] > longest) longest = seq_off[i + 1] - seq_off[i];
    const int64_t cap = total + 8;
    uint8_t *c = (uint8_t *)malloc((size_t)cap);
    int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    if (!c || !offs) { free(c); free(offs); bad = 1; continue; }
    for (int i = 0; i <= n; ++i) offs[i] = seq_off[s0 + i] - seq_off[s0];
    const int64_t cl = orc_poa_consensus(seqs + seq_off[s0], offs, n, c, cap);
    free(offs);
    if (cl < 0) { free(c); bad = 1; continue; }
    cons[k] = c;
    cons_len[k] = cl;
    const int tl = (int)(ref_off[k + 1] - ref_off[k]);
    const int64_t ccap = cl + tl + 4;
    uint32_t *cg = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)ccap);
    if (!cg) { bad = 1; continue; }
    int32_t sc = 0;
    n_cigar[k] = orc_ksw_extd2_global(c, (int)cl, refs + ref_off[k], tl, 5, mat, 16, 2, 41, 1, &sc, cg, ccap);
    score[k] = sc;
    free(cg);
  }
  if (!bad && ratio) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (int64_t k = 0; k < n_sub - 1; ++k) ratio[k] = orc_fuzz_ratio(cons[k], cons_len[k], cons[k + 1], cons_len[k + 1]);
  }
  for (int64_t k = 0; k < n_sub; ++k) free(cons[k]);
  free(cons);
  return bad ? -1 : 0;
}
