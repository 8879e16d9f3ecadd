// gpu_inflate_hook.h -- connects BamReader (which knows nothing about the library) to svdss_bgzf_inflate.
#pragma once
#include <cstdlib>

#include "../../include/svdss_hip.h"
#include "bam_reader.h"

// BGZF blocks of `bam` are inflated on the GPU (csrc/inflate.hip) when one is present.  SVDSS_GPU_INFLATE: 0 = host
// workers only; 1..99 = that share of the chunks goes to the GPU; 100 = the GPU takes whatever the host workers cannot
// start at once; 101 (default) = every chunk (the host's cores are better spent on the other stages).  Returns whether
// the GPU path is on.
inline bool svdss_enable_gpu_inflate(BamReader& bam, int device = 0, int n_devices = 1) {
  const int pct = getenv("SVDSS_GPU_INFLATE") ? atoi(getenv("SVDSS_GPU_INFLATE")) : 101;
  if (pct <= 0 || svdss_device_count() <= 0) return false;
  BamReader::GpuInflateApi api;
  api.inflate = [](void** obj, int dev, const uint8_t* comp, int64_t comp_bytes, const void* blocks, int64_t n_blocks,
                   void* d_out, uint8_t* host_out, int64_t out_bytes, int64_t* bad) {
    return svdss_bgzf_inflate((svdss_inflate_t**)obj, dev, comp, comp_bytes, (const svdss_bgzf_block_t*)blocks, n_blocks, d_out,
                              host_out, out_bytes, bad);
  };
  api.inflate_free = [](void* obj) { svdss_inflate_free((svdss_inflate_t*)obj); };
  api.device_alloc = svdss_device_alloc;
  api.device_free = svdss_device_free;
  api.host_alloc = svdss_host_alloc;
  api.host_free = svdss_host_free;
  bam.enable_gpu_inflate(api, device, pct, n_devices);
  return true;
}
