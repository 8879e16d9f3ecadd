// tests/native/poa_quad_emu.cpp -- TEST INFRASTRUCTURE ONLY.  Never loaded by the product.
//
// Runs the device code of the several-sub-clusters-per-wavefront POA kernel (svdss_amd/csrc/poa_quad_core.h) on the
// CPU wave emulator (wave_emu.h), so that its row loop, traceback and graph update can be held against the oracle in
// this GPU-less container.  Task sizing mirrors poa.hip's first round; the heaviest-bundle step (poa_bundle_kernel on
// the GPU) is a plain loop over the graph the emulated kernel leaves in its workspace.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "poa_quad_backend_emu.h"
#include "../../svdss_amd/csrc/poa_quad_core.h"

namespace {

struct Args {
  const PoaWaveTask* tasks; int n_tasks; const uint8_t* seqs; const int64_t* seq_off; int32_t* ws32; int32_t* cons_len;
  int32_t* status; unsigned long long* cells; int gw, c, max_len;
};

template <int GW, int C>
void body(void* p) {
  const Args* a = (const Args*)p;
  pq::poaq_run<GW, C>(a->tasks, a->n_tasks, a->seqs, a->seq_off, a->ws32, a->cons_len, a->status, a->cells,
                      (int32_t*)wemu::lds_base(), wemu::block_id(), pq::Geom<GW, C>::qcap(a->max_len));
}

template <int GW, int C>
void run_blocks(Args& a) {
  const int G = 64 / GW;
  for (int b = 0; b * G < a.n_tasks; ++b) wemu::run_block(b, pq::Geom<GW, C>::lds_bytes(a.max_len), &body<GW, C>, &a);
}

int64_t ws_ints(int nc, int ec, int max_len, int ws) {
  const int64_t opcap = (int64_t)nc + max_len + 4;
  return 21 * (int64_t)nc + 192 + 5 * (int64_t)ec + 4 * opcap + 4 * (int64_t)nc * ws;
}

}  // namespace

// clusters as svdss_poa_consensus_batch takes them; out_len[c] = consensus length or -1 if the kernel handed the
// sub-cluster back (status != 0, out_status[c]); cons: concatenated, cons_off[c] = start.  Returns 0, or -1 for an
// unsupported (gw, c).
extern "C" int poaq_emu_consensus(const uint8_t* seqs_in, const int64_t* seq_off, const int64_t* cluster_off, int64_t n_clusters, int gw, int c,
                                  int64_t* out_len, int32_t* out_status, uint8_t* cons, int64_t* cons_off, int64_t cons_cap,
                                  unsigned long long* cells) {
  std::vector<PoaWaveTask> tasks;
  std::vector<int64_t> ids;
  int64_t w32 = 0;
  const int64_t total = seq_off[cluster_off[n_clusters]];
  std::vector<uint8_t> seqs((size_t)total + 256, 0);
  memcpy(seqs.data(), seqs_in, (size_t)total);
  for (int64_t k = 0; k < n_clusters; ++k) {
    PoaWaveTask t;
    memset(&t, 0, sizeof t);
    t.seq_first = cluster_off[k];
    t.n_seqs = cluster_off[k + 1] - cluster_off[k];
    int64_t tot = 0, maxl = 0;
    for (int64_t s = t.seq_first; s < t.seq_first + t.n_seqs; ++s) { const int64_t l = seq_off[s + 1] - seq_off[s]; tot += l; maxl = std::max(maxl, l); }
    int64_t nc = std::min<int64_t>(tot + 2, maxl * 150 / 100 + 8 * t.n_seqs + 64);
    nc = std::min<int64_t>(nc, 65000);
    t.nc = (int32_t)nc;
    t.ec = (int32_t)std::min<int64_t>(nc + nc / 4 + t.n_seqs + 64, 100000);
    t.max_len = (int32_t)maxl;
    t.ws = gw * c;
    t.ws_off = w32;
    w32 += ws_ints(t.nc, t.ec, t.max_len, t.ws);
    tasks.push_back(t);
  }
  std::vector<int32_t> ws((size_t)w32 + 64, 0x5A5A5A5A), len((size_t)n_clusters, -7), st((size_t)n_clusters, -1);
  *cells = 0;
  int max_len = 0;
  for (const PoaWaveTask& t : tasks) max_len = std::max(max_len, (int)t.max_len);
  Args a{tasks.data(), (int)tasks.size(), seqs.data(), seq_off, ws.data(), len.data(), st.data(), cells, gw, c, max_len};
  if (n_clusters > 0) {
    if (gw == 16 && c == 3) run_blocks<16, 3>(a);
    else if (gw == 16 && c == 4) run_blocks<16, 4>(a);
    else if (gw == 16 && c == 5) run_blocks<16, 5>(a);
    else if (gw == 16 && c == 7) run_blocks<16, 7>(a);
    else if (gw == 32 && c == 2) run_blocks<32, 2>(a);
    else if (gw == 32 && c == 3) run_blocks<32, 3>(a);
    else if (gw == 64 && c == 1) run_blocks<64, 1>(a);
    else if (gw == 64 && c == 2) run_blocks<64, 2>(a);
    else return -1;
  }
  int64_t o = 0;
  for (int64_t k = 0; k < n_clusters; ++k) {
    cons_off[k] = o;
    out_status[k] = st[(size_t)k];
    out_len[k] = -1;
    if (st[(size_t)k] != 0) continue;
    const PoaWaveTask& t = tasks[(size_t)k];
    if (t.n_seqs <= 0) { out_len[k] = 0; continue; }
    // heaviest bundle over the graph in the workspace (oracle/svdss_oracle_poa.c "cons"; poa_bundle_kernel on the GPU)
    const int32_t* W = ws.data() + t.ws_off;
    const int nc = t.nc, ec = t.ec, N = len[(size_t)k];
    const int32_t *out_head = W, *order = W + 2 * (int64_t)nc, *base = W + 5 * (int64_t)nc;
    const int32_t* E = W + 21 * (int64_t)nc + 192;
    const int32_t *e_to = E + ec, *e_w = E + 2 * (int64_t)ec, *e_next_out = E + 3 * (int64_t)ec;
    std::vector<int64_t> score((size_t)nc, 0);
    std::vector<int> best((size_t)nc, -1);
    for (int r = N - 1; r >= 0; --r) {
      const int v = order[r];
      int bw = -1; int64_t bsc = -1;
      for (int e = out_head[v]; e >= 0; e = e_next_out[e]) {
        const int x = e_to[e];
        if (e_w[e] > bw || (e_w[e] == bw && score[(size_t)x] > bsc)) { bw = e_w[e]; bsc = score[(size_t)x]; best[(size_t)v] = x; }
      }
      score[(size_t)v] = best[(size_t)v] >= 0 ? bw + bsc : 0;
    }
    int64_t n = 0;
    for (int v = best[0]; v >= 0 && v != 1; v = best[(size_t)v]) {
      if (o + n >= cons_cap) return -2;
      cons[o + n++] = (uint8_t)base[v];
    }
    out_len[k] = n;
    o += n;
  }
  return 0;
}
