// poa_task.h -- task descriptor of the LDS-resident POA kernels (poa_wave.hip, poa_quad.hip); no HIP types: the CPU wave
// emulator of the tests includes it too.
#pragma once
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
