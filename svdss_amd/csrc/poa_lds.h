// poa_lds.h -- task descriptor and launcher of the LDS-resident POA kernel (poa_lds.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct PoaLdsTask {
  int64_t seq_first, n_seqs;
  int32_t nc, ec;          // node / edge capacity of the LDS graph (< 65000)
  int32_t max_len;         // longest read of the cluster
  int32_t ws, ring;        // DP row stride (power of two >= widest row) and rows kept in LDS (power of two)
  // offsets into the int32 workspace
  int64_t row_off;         // 7 x nc: row_beg, row_end, H at column L, 2 words of predecessor deltas per row, mpl, mpr
  int64_t dp_off;          // H, E1, E2 (int32) and direction words (uint32): 4 x nc x ws
  int64_t aln_off;         // 5 x nc
  int64_t op_off;          // 4 x (nc + max_len + 4): traceback ops (row, column), path nodes, path aux
  int64_t cons_off;        // into the byte workspace, nc bytes
};

size_t poa_lds_bytes(int nc, int ec, int max_len, int ws, int ring);
hipError_t poa_lds_launch(const PoaLdsTask* d_tasks, int n_tasks, size_t lds_bytes, const uint8_t* d_seqs,
                          const int64_t* d_seq_off, int32_t* ws32, uint8_t* ws8, int32_t* d_len, int32_t* d_status,
                          unsigned long long* d_cells);
