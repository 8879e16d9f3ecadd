// deflate_dev.h -- the BGZF deflate kernels for callers inside the library whose bytes are already on the device
// (csrc/bam_device.hip, svdss_bam_smooth_run): nothing but the launches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// bytes a member's slot needs in the strided output (a multiple of 64)
int64_t svdss_deflate_stride(int32_t block_bytes);
// d_in[0, in_bytes) cut into blocks of block_bytes -> member b at d_out + b * stride (d_out zeroed by the caller), its
// length with the 8 footer bytes LEFT TO THE CALLER in d_len[b].  d_in must be readable 16 bytes behind its end.
hipError_t svdss_deflate_enqueue(hipStream_t st, const uint8_t* d_in, int64_t in_bytes, int32_t block_bytes, uint8_t* d_out,
                                 int64_t stride, int32_t* d_len);
// the members back to back: d_off[nb + 1] = their offsets (exclusive sums of d_len), d_dense = the bytes
hipError_t svdss_deflate_compact_enqueue(hipStream_t st, const uint8_t* d_strided, int64_t stride, const int32_t* d_len, int64_t nb,
                                         int64_t* d_off, uint8_t* d_dense);
