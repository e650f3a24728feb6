// nlsg_pool.h — device memory and streams recycled across engines (host side of the library).
//
// The reference's minimize() is one call (nlsolver.h:2404, 2553, 3188, 3457): behind the drop-in
// header every call creates an engine, solves and destroys it. Measured on MI355X / ROCm 7.2
// (bench.py --workload tts, round 4): the solve of a small problem takes 0.4 ms while the dozen
// hipMalloc / hipFree pairs, the stream and the device query around it took 3.4 ms. So what an
// engine releases stays with the process:
//   * device blocks go to a per-device free list keyed by size (sizes repeat exactly from call to
//     call) and are handed out again by pool_malloc; at most NLSG_POOL_BYTES (default 40 GiB) sit
//     idle per device (enough for configs[2]'s 32 GiB of inverse Hessians: a fresh hipMalloc of that size was seen to take 1.9 s), larger blocks and the overflow are freed for real; a hipMalloc that runs
//     out of memory empties the cache and tries once more;
//   * non-blocking streams are parked and reused;
//   * the gfx950 check of a device is made once.
// An engine synchronises its streams before it releases anything, so a recycled block or stream
// has no work in flight. Contents of a recycled block are whatever the previous owner left: every
// engine initialises what it reads (as it had to with hipMalloc, which promises nothing either).
// nlsg_release_cached() (C-ABI) empties the cache.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace nlsg {

struct DevicePool {
  std::mutex mu;
  struct Block {
    int device;
    size_t bytes;
  };
  std::unordered_map<void *, Block> live;                    // handed out
  std::map<std::pair<int, size_t>, std::vector<void *>> idle;  // (device, bytes) -> parked blocks
  std::unordered_map<int, size_t> idle_bytes;                // per device
  std::unordered_map<int, std::vector<hipStream_t>> streams;  // parked non-blocking streams
  size_t cap = 40ull << 30;
  bool enabled = true;
  bool poison = false;  // NLSG_POOL_POISON=1 (tests): every block handed out is filled with 0xFF bytes
                        // (NaNs / huge integers), so an engine that reads what it never wrote shows
  DevicePool() {
    if (const char *c = std::getenv("NLSG_POOL_POISON")) poison = c[0] == '1';
    if (const char *c = std::getenv("NLSG_POOL_BYTES")) {
      cap = std::strtoull(c, nullptr, 10);
      enabled = cap > 0;
    }
  }
  static size_t rounded(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= (64u << 10)) {  // small blocks: powers of two (states, tickets, records)
      size_t r = 256;
      while (r < bytes) r <<= 1;
      return r;
    }
    return (bytes + 4095) & ~static_cast<size_t>(4095);
  }
  // frees every parked block of `device` (-1: all devices); the caller holds `mu`
  void trim_locked(int device) {
    int cur = 0;
    hipGetDevice(&cur);
    for (auto it = idle.begin(); it != idle.end();) {
      if (device >= 0 && it->first.first != device) {
        ++it;
        continue;
      }
      hipSetDevice(it->first.first);
      for (void *p : it->second) hipFree(p);
      idle_bytes[it->first.first] = 0;
      it = idle.erase(it);
    }
    hipSetDevice(cur);
  }
};
inline DevicePool &device_pool() {
  static DevicePool *p = new DevicePool();  // never destroyed: the HIP runtime may unload first
  return *p;
}

// hipMalloc on the current device, served from the cache when a block of the size is parked
inline hipError_t pool_malloc(void **ptr, size_t bytes) {
  DevicePool &pool = device_pool();
  if (!pool.enabled) return hipMalloc(ptr, bytes);
  int dev = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he != hipSuccess) return he;
  const size_t sz = DevicePool::rounded(bytes);
  std::lock_guard<std::mutex> lock(pool.mu);
  auto it = pool.idle.find({dev, sz});
  if (it != pool.idle.end() && !it->second.empty()) {
    *ptr = it->second.back();
    it->second.pop_back();
    pool.idle_bytes[dev] -= sz;
  } else {
    he = hipMalloc(ptr, sz);
    if (he == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      pool.trim_locked(dev);
      he = hipMalloc(ptr, sz);
    }
    if (he != hipSuccess) return he;
  }
  pool.live[*ptr] = {dev, sz};
  if (pool.poison) {
    he = hipMemset(*ptr, 0xFF, sz);
    if (he == hipSuccess) he = hipDeviceSynchronize();
  }
  return he;
}

// hipFree counterpart; nullptr and pointers the pool did not hand out are passed to hipFree
inline hipError_t pool_free(void *ptr) {
  if (!ptr) return hipSuccess;
  DevicePool &pool = device_pool();
  std::lock_guard<std::mutex> lock(pool.mu);
  auto it = pool.live.find(ptr);
  if (it == pool.live.end()) return hipFree(ptr);
  const DevicePool::Block b = it->second;
  pool.live.erase(it);
  if (!pool.enabled || b.bytes > pool.cap || pool.idle_bytes[b.device] + b.bytes > pool.cap) {
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != b.device) hipSetDevice(b.device);
    const hipError_t he = hipFree(ptr);
    if (cur != b.device) hipSetDevice(cur);
    return he;
  }
  pool.idle[{b.device, b.bytes}].push_back(ptr);
  pool.idle_bytes[b.device] += b.bytes;
  return hipSuccess;
}

// a non-blocking stream on the current device (parked ones first); give it back idle
inline hipError_t pool_stream_get(hipStream_t *out) {
  DevicePool &pool = device_pool();
  int dev = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he != hipSuccess) return he;
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto &v = pool.streams[dev];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
inline void pool_stream_put(int device, hipStream_t s) {
  if (!s) return;
  DevicePool &pool = device_pool();
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto &v = pool.streams[device];
    if (v.size() < 8) {
      v.push_back(s);
      return;
    }
  }
  hipStreamDestroy(s);
}

inline void pool_release_all() {
  DevicePool &pool = device_pool();
  std::lock_guard<std::mutex> lock(pool.mu);
  pool.trim_locked(-1);
  for (auto &kv : pool.streams) {
    for (hipStream_t s : kv.second) hipStreamDestroy(s);
    kv.second.clear();
  }
}
inline size_t pool_idle_bytes() {
  DevicePool &pool = device_pool();
  std::lock_guard<std::mutex> lock(pool.mu);
  size_t t = 0;
  for (auto &kv : pool.idle_bytes) t += kv.second;
  return t;
}

}  // namespace nlsg
#endif  // !__HIPCC_RTC__
