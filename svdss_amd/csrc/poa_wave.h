// poa_wave.h -- task descriptor and launcher of the one-wavefront-per-sub-cluster POA kernel (poa_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "poa_task.h"

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
// the consensus kernel alone (poa_quad.hip's launches end with it): cons_len[task] = graph rows in, consensus length out
hipError_t poa_bundle_launch(const PoaWaveTask* d_tasks, int n_tasks, size_t bundle_lds_bytes, int32_t* ws32, uint8_t* ws8, int32_t* d_len,
                             const int32_t* d_status, hipStream_t stream);
void poa_wave_debug_report();
