// gpu_deflate_hook.h -- connects BgzfWriter (which knows nothing about the library) to svdss_bgzf_deflate.
#pragma once
#include <cstdlib>

#include "../../include/svdss_hip.h"
#include "bam_writer.h"

// The output blocks of `w` are deflated on the GPU (csrc/deflate.hip: literals under dynamic Huffman codes, a
// level-1-class encoder) when one is present; SVDSS_GPU_DEFLATE=0 keeps the host's libdeflate / zlib at
// SVDSS_BAM_LEVEL.  Returns whether the GPU path is on.
inline bool svdss_enable_gpu_deflate(BgzfWriter& w, int device = 0) {
  const char* e = getenv("SVDSS_GPU_DEFLATE");
  if ((e && atoi(e) == 0) || svdss_device_count() <= 0) return false;
  BgzfWriter::GpuDeflateApi api;
  api.deflate = [](void** obj, int dev, const uint8_t* in, int64_t in_bytes, int32_t block_bytes, uint8_t* out,
                   int64_t out_stride, int32_t* out_len) {
    return svdss_bgzf_deflate((svdss_deflate_t**)obj, dev, in, in_bytes, block_bytes, out, out_stride, out_len);
  };
  api.free_ = [](void* obj) { svdss_deflate_free((svdss_deflate_t*)obj); };
  api.host_alloc = svdss_host_alloc;
  api.host_free = svdss_host_free;
  api.device = device;
  w.enable_gpu_deflate(api);
  return true;
}
