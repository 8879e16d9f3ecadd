// poa_wave.h -- task descriptor and launcher of the one-wavefront-per-sub-cluster POA kernel (poa_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct PoaWaveTask {
  int64_t seq_first, n_seqs;
  int32_t nc, ec;          // node / edge capacity of the graph
  int32_t max_len;         // longest read of the cluster
  int32_t ws;              // HBM stride of a DP row: power of two >= the widest row
  int32_t rs;              // LDS stride of a ring row: >= the widest row
  int32_t ring;            // DP rows kept in LDS (power of two); two more slots stage rows read back from HBM
  int32_t prio;            // s_setprio level (0-3): the longest chains of a batch decide its duration
  int32_t pad_;
  int64_t ws_off;          // into the int32 workspace (poa_wave_ws_ints of it)
  int64_t cons_off;        // into the byte workspace, nc bytes
};

// columns per lane the kernel is instantiated for
static const int kPoaWaveCols[4] = {1, 2, 3, 5};
static const int kPoaWaveNCols = 4;

size_t poa_wave_lds_bytes(int nc, int max_len, int rs, int ring);
size_t poa_bundle_lds_bytes(int nc);
int64_t poa_wave_ws_ints(int nc, int ec, int max_len, int ws);
// the row-loop kernel followed by the consensus kernel on the same stream
hipError_t poa_wave_launch(int cols, const PoaWaveTask* d_tasks, int n_tasks, size_t lds_bytes, size_t bundle_lds_bytes,
                           const uint8_t* d_seqs, const int64_t* d_seq_off, int32_t* ws32, uint8_t* ws8, int32_t* d_len,
                           int32_t* d_status, unsigned long long* d_cells, hipStream_t stream);
void poa_wave_debug_report();
