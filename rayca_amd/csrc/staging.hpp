// Host -> device copies through page-locked staging blocks.
//
// hipMemcpy from ordinary (pageable) host memory has to page-lock the source range first, and on this platform that
// costs next to nothing for a range the driver has seen before and a great deal for one it has not: the same 80 MiB took
// 1.6 ms in one round of tests/gpu_copy_probe.py and 19.4 ms in the next, and the first scene of a process spent 25-30 ms
// in uploads its later scenes do in 4 (tests/gpu_build_probe.py).  Page-locking the sources in place (hipHostRegister) is
// fast but went wrong.  So the large uploads of a scene build go through blocks that are page-locked
// once per process: the CPU copies a chunk into a block (8-10 GB/s on one thread, on threads that are waiting for the
// GPU anyway), the copy engine takes it from there at ~55 GB/s, two blocks in turn so that the two overlap.  What it costs
// is the same every time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <mutex>
#include <vector>

namespace rayca {

constexpr size_t kStagingBlock = size_t(8) << 20;

// Page-locked blocks, handed out and taken back; they stay with the process (a handful of 8 MiB blocks).
class StagingCache {
 public:
  static void* acquire() {
    {
      std::lock_guard<std::mutex> lock(mu());
      if (!blocks().empty()) {
        void* p = blocks().back();
        blocks().pop_back();
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, kStagingBlock, hipHostMallocPortable) != hipSuccess) {   // (portable: scenes of several devices share the cache)
      (void)hipGetLastError();
      return nullptr;
    }
    return p;
  }
  static void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(mu());
    blocks().push_back(p);
  }
  static void prime(unsigned count) {  // (the runtime warm-up of a process's first scene makes the first few)
    std::vector<void*> got;
    for (unsigned i = 0; i < count; ++i) got.push_back(acquire());
    for (void* p : got) release(p);
  }

 private:
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
  static std::vector<void*>& blocks() {
    static std::vector<void*> b;
    return b;
  }
};

// Two blocks in turn, for the copies one thread puts on one stream.
class StagedCopier {
 public:
  StagedCopier() = default;
  StagedCopier(const StagedCopier&) = delete;
  StagedCopier& operator=(const StagedCopier&) = delete;
  ~StagedCopier() { finish(); }

  // Queues dst[0, bytes) <- src on `stream`.  When it returns, `src` has been read completely (it may be freed or
  // overwritten); the device copy itself completes in stream order.  Small copies and a cache that cannot page-lock any
  // more memory fall back to a plain hipMemcpyAsync + synchronize, which has the same "src is free" property.
  hipError_t copy(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (bytes < (size_t(256) << 10) || !ensure()) {
      hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      return e;
    }
    for (size_t off = 0; off < bytes; off += kStagingBlock) {
      const size_t n = bytes - off < kStagingBlock ? bytes - off : kStagingBlock;
      const int k = turn_;
      turn_ ^= 1;
      hipError_t e = hipSuccess;
      if (busy_[k] && (e = hipEventSynchronize(ev_[k])) != hipSuccess) return e;
      std::memcpy(block_[k], static_cast<const char*>(src) + off, n);
      if ((e = hipMemcpyAsync(static_cast<char*>(dst) + off, block_[k], n, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
      if ((e = hipEventRecord(ev_[k], stream)) != hipSuccess) return e;
      busy_[k] = true;
    }
    return hipSuccess;
  }
  // The other direction: dst[0, bytes) (host) <- src (device), in stream order behind what `stream` already holds.  The copy
  // engine writes a block while the CPU empties the other; when it returns, `dst` is complete.  Like copy() it keeps the
  // runtime from page-locking the library's own heap memory in place ("The fault of round 2").
  hipError_t copy_back(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (bytes < (size_t(256) << 10) || !ensure()) {   // (small: the runtime stages it through its own page-locked buffers)
      hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      return e;
    }
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2; ++k) {   // blocks still being read by an earlier copy() of this object
      if (busy_[k] && (e = hipEventSynchronize(ev_[k])) != hipSuccess) return e;
      busy_[k] = false;
    }
    size_t pending_off = 0, pending_n = 0;
    int pending_k = -1;
    auto drain = [&]() -> hipError_t {   // the chunk queued one step ago: wait for it, copy it out
      if (pending_k < 0) return hipSuccess;
      const hipError_t de = hipEventSynchronize(ev_[pending_k]);
      if (de == hipSuccess) std::memcpy(static_cast<char*>(dst) + pending_off, block_[pending_k], pending_n);
      pending_k = -1;
      return de;
    };
    for (size_t off = 0; off < bytes; off += kStagingBlock) {
      const size_t n = bytes - off < kStagingBlock ? bytes - off : kStagingBlock;
      const int k = turn_;
      turn_ ^= 1;
      if ((e = hipMemcpyAsync(block_[k], static_cast<const char*>(src) + off, n, hipMemcpyDeviceToHost, stream)) != hipSuccess) break;
      if ((e = hipEventRecord(ev_[k], stream)) != hipSuccess) break;
      const hipError_t de = drain();   // (the previous chunk, while this one is in flight)
      pending_off = off; pending_n = n; pending_k = k;
      if (de != hipSuccess) { e = de; break; }
    }
    if (e != hipSuccess) {   // nothing may still be writing a block when it goes back to the cache
      (void)hipStreamSynchronize(stream);
      return e;
    }
    return drain();
  }
  // Waits until the blocks are no longer being read and hands them back.
  void finish() {
    for (int k = 0; k < 2; ++k) {
      if (busy_[k]) (void)hipEventSynchronize(ev_[k]);
      busy_[k] = false;
      if (ev_[k]) (void)hipEventDestroy(ev_[k]);
      ev_[k] = nullptr;
      StagingCache::release(block_[k]);
      block_[k] = nullptr;
    }
  }

 private:
  bool ensure() {
    for (int k = 0; k < 2; ++k) {
      if (!block_[k]) block_[k] = StagingCache::acquire();
      if (!block_[k]) return false;
      if (!ev_[k] && hipEventCreateWithFlags(&ev_[k], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ev_[k] = nullptr;
        return false;
      }
    }
    return true;
  }
  void* block_[2] = {nullptr, nullptr};
  hipEvent_t ev_[2] = {nullptr, nullptr};
  bool busy_[2] = {false, false};
  int turn_ = 0;
};

}  // namespace rayca
