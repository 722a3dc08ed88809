// Page-locks a range of ordinary host memory in place for the lifetime of the object.
//
// hipMemcpy between device and PAGEABLE host memory ran anywhere between 4 and 55 GB/s on this platform, depending on
// the history of the pages (tests/gpu_copy_probe.py: the same 80 MiB took 1.6 ms in one round and 19.4 ms in the next),
// and a scene upload is ~150 MB of such copies.  hipHostRegister on the range first costs ~1 ms per 80 MiB and makes
// every copy 55 GB/s; unregistering is free.  Only for blocks that own their pages (host_scene.hpp PageAllocator): a
// registration covers whole pages, and one that shares its end pages with a neighbouring heap block another thread is
// registering or copying from at the same time is not something to lean on.  A failed registration just leaves the copy
// to the pageable path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace rayca {

class PinnedRange {
 public:
  PinnedRange(const void* ptr, size_t bytes) {
    if (!ptr || bytes < (size_t(1) << 20)) return;
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) p_ = const_cast<void*>(ptr);
    else (void)hipGetLastError();   // (not an error of the caller's: clear it)
  }
  ~PinnedRange() { release(); }
  void release() {
    if (p_) (void)hipHostUnregister(p_);
    p_ = nullptr;
  }
  PinnedRange(const PinnedRange&) = delete;
  PinnedRange& operator=(const PinnedRange&) = delete;

 private:
  void* p_ = nullptr;
};

}  // namespace rayca
