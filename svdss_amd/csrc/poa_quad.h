// poa_quad.h -- launcher of the several-sub-clusters-per-wavefront POA kernel (poa_quad.hip, poa_quad_core.h).  Tasks
// are poa_wave.hip's records with ws = group width x columns per lane; consecutive tasks of a launch share a wavefront
// (64 / group width of them).  The heaviest-bundle consensus is poa_wave.hip's poa_bundle_kernel (poa_bundle_launch).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "poa_task.h"

// (group width, columns per lane) the kernel is instantiated for
static const int kPoaQuadVariants[10][2] = {{16, 3}, {16, 4}, {16, 5}, {16, 6}, {16, 7}, {32, 2}, {32, 3}, {32, 4}, {64, 1}, {64, 2}};
static const int kPoaQuadNVariants = 10;

bool poa_quad_supported(int gw, int cols);
// max_len: the longest read of the launch (every group keeps the read being aligned in LDS)
size_t poa_quad_lds_bytes(int gw, int cols, int max_len);
hipError_t poa_quad_launch(int gw, int cols, const PoaWaveTask* d_tasks, int n_tasks, int max_len, const uint8_t* d_seqs, const int64_t* d_seq_off,
                           int32_t* ws32, int32_t* d_len, int32_t* d_status, unsigned long long* d_cells, hipStream_t stream);
void poa_quad_debug_report();
