// dev_arena.h -- grow-only device buffer handed out by a bump pointer.  The workspaces of a batched call (tens of
// GB of direction bytes / graph pools) are reused by the next call on the same batch object instead of going
// through hipMalloc / hipFree again (hundreds of milliseconds per call at these sizes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

struct DevArena {
  void* p = nullptr;
  size_t cap = 0, used = 0;
  DevArena() = default;
  DevArena(const DevArena&) = delete;
  DevArena& operator=(const DevArena&) = delete;
  ~DevArena() { drop(); }
  void drop() { if (p) (void)hipFree(p); p = nullptr; cap = used = 0; }
  // make room for `bytes` and start handing out from the beginning; only while nothing handed out is still in use.
  // hipSuccess or the hipMalloc error
  hipError_t reserve(size_t bytes) {
    used = 0;
    if (bytes <= cap) return hipSuccess;
    drop();
    const size_t want = bytes + bytes / 8 + 4096;
    const hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    used = at + bytes;
    return (char*)p + at;
  }
  static size_t padded(size_t bytes) { return ((bytes + 255) & ~(size_t)255) + 256; }
};
