"""VERDICT r4 item 9 (lines per read of the search kernel): how far apart are the K-mer windows of consecutive backward
phases of ping_pong_search (ping_pong.cpp:4-49)?  A scaled model of the headline workload on the CPU: a random reference
with n / 4^K = 1.44 (what GRCh38 with both strands has at K = 16), here K = 10, reads with 0.5 % substitutions; "occurs"
by binary search over the sorted 26-mers of both strands (longer substrings occur iff they hold no error: the read is a
copy).  Prints the distribution of st - st' between consecutive phase starts.  Result: profiles/r05y_table_lookup_locality.txt."""
import numpy as np
rng = np.random.default_rng(1)
K = 10
n1 = int(1.44 * 4**K / 2)
ref = rng.integers(0, 4, size=n1).astype(np.int64)
rc = (3 - ref)[::-1]
text = np.concatenate([ref, [4], rc, [4]])
W = 26
pad = np.concatenate([text, np.full(W, 4)])
codes = np.zeros(len(text), dtype=np.int64)
for j in range(W):
    codes = codes * 5 + pad[j:j + len(text)]
codes.sort()
P5 = [5 ** k for k in range(W + 1)]
L, ERR = 6000, 0.005
def run(read, errpos):
    cum = np.concatenate([[0], np.cumsum(errpos)])
    def occurs(b, e):
        k = e - b + 1
        if k > W:
            return cum[e + 1] - cum[b] == 0
        v = 0
        for c in read[b:e + 1]: v = v * 5 + int(c)
        lo = v * P5[W - k]; hi = (v + 1) * P5[W - k]
        return np.searchsorted(codes, lo) < np.searchsorted(codes, hi)
    starts = []; sfs = []
    begin = L - 1
    while begin >= 0:
        st = begin
        starts.append(st)
        while begin > 0 and occurs(begin, st): begin -= 1
        # loop ended: either begin==0 or P[begin..st] does not occur
        if begin == 0 and occurs(0, st): break
        end = begin
        while occurs(begin, end): end += 1
        sfs.append((begin, end - begin + 1, st))
        if begin == 0: break
        begin = end - 1
    return starts, sfs
sh = {}; n_sfs = 0; n_err = 0; dd = {}
for r in range(40):
    a = int(rng.integers(0, n1 - L))
    P = ref[a:a + L].copy()
    e = rng.random(L) < ERR
    P[e] = (P[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
    starts, sfs = run(P, e.astype(np.int64))
    n_sfs += len(sfs); n_err += int(e.sum())
    for x, y in zip(starts, starts[1:]):
        sh[x - y] = sh.get(x - y, 0) + 1
    for b, l, st in sfs:
        d = st - b          # symbols consumed backward with a non-empty interval
        key = "d<K" if d < K else "d>=K"
        dd[key] = dd.get(key, 0) + 1
tot = sum(sh.values())
print("K", K, "reads 40 x", L, "errors", n_err, "raw SFS", n_sfs, "= %.1f per error" % (n_sfs / n_err), dd)
acc = 0
for s in sorted(sh):
    acc += sh[s]
    if s <= 8 or sh[s] / tot > 0.01: print("shift %3d: %5.1f %%  (cum %5.1f %%)" % (s, 100.0 * sh[s] / tot, 100.0 * acc / tot))
