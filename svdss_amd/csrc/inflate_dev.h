// inflate_dev.h -- the inflate kernel for callers inside the library whose compressed bytes are already on the device
// (csrc/bam_device.hip): nothing but the launch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/svdss_hip.h"

// block i: deflate stream d_comp[coff, coff + clen) -> isize bytes at d_out + uoff; d_status[i] = 0 or the reason it
// stopped (d_comp must be readable 8 KB past the last stream's end: the kernel keeps 4 KB of input ahead in LDS)
hipError_t svdss_inflate_enqueue(hipStream_t st, const uint8_t* d_comp, const svdss_bgzf_block_t* d_blocks, int64_t n_blocks,
                                 uint8_t* d_out, int32_t* d_status);
