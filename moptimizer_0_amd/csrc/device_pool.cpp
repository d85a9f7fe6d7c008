// Size-class cache in front of hipMalloc / hipFree for the buffers a cost owns.
//
// hipFree synchronises the whole device and both calls go to the driver; a registration pipeline
// builds and drops a cost per frame (measured on a 30 k-point cost: destroy 0.5 -> 0.2 ms,
// construct 0.16 -> 0.1 ms once stream creation is out of the way).  Released blocks of up to 256 MiB are kept per device and size class (eighth-of-an-octave
// classes, <= 12.5 % slack) up to a total of MOPT_DEVICE_POOL_MB (default 1024); larger blocks
// and anything beyond the cap go straight back to the runtime.  Callers release a block only
// after the stream that used it has been synchronised.
#include "cost_state.hpp"

#include <map>
#include <mutex>
#include <unordered_map>

namespace mopt_detail {
namespace {

constexpr size_t kLargestPooled = size_t(256) << 20;

struct Block {
  int device;
  size_t size_class;  // 0: not pooled
};

std::mutex g_mutex;
std::unordered_map<void *, Block> g_live;
std::map<std::pair<int, size_t>, std::vector<void *>> g_free;
std::map<int, size_t> g_cached_bytes;

size_t poolCapBytes() {
  static const size_t cap = size_t(envInt("MOPT_DEVICE_POOL_MB", 1024)) << 20;
  static const bool off = std::getenv("MOPT_DEVICE_POOL_MB") && std::atoi(std::getenv("MOPT_DEVICE_POOL_MB")) == 0;
  return off ? 0 : cap;
}

size_t sizeClass(size_t bytes) {
  if (bytes > kLargestPooled || poolCapBytes() == 0) return 0;
  if (bytes < 4096) return 4096;
  int e = 0;
  while ((size_t(1) << (e + 1)) <= bytes) ++e;  // 2^e <= bytes < 2^(e+1)
  const size_t step = size_t(1) << (e - 3);
  return (bytes + step - 1) / step * step;
}

void trimLocked(int device) {
  for (auto it = g_free.begin(); it != g_free.end();) {
    if (it->first.first == device) {
      for (void *p : it->second) (void)hipFree(p);
      it = g_free.erase(it);
    } else {
      ++it;
    }
  }
  g_cached_bytes[device] = 0;
}

}  // namespace

hipError_t deviceAlloc(void **out, size_t bytes) {
  *out = nullptr;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  const size_t cls = sizeClass(bytes ? bytes : 1);
  std::lock_guard<std::mutex> lock(g_mutex);
  if (cls) {
    auto it = g_free.find({device, cls});
    if (it != g_free.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      g_cached_bytes[device] -= cls;
      g_live[*out] = Block{device, cls};
      return hipSuccess;
    }
  }
  const size_t want = cls ? cls : bytes;
  e = hipMalloc(out, want);
  if (e != hipSuccess && g_cached_bytes[device] > 0) {  // give the cache back and try once more
    (void)hipGetLastError();
    trimLocked(device);
    e = hipMalloc(out, want);
  }
  if (e == hipSuccess) g_live[*out] = Block{device, cls};
  return e;
}

void deviceRelease(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_live.find(p);
  if (it == g_live.end()) {  // not ours (should not happen): hand it to the runtime
    (void)hipFree(p);
    return;
  }
  const Block b = it->second;
  g_live.erase(it);
  if (b.size_class && g_cached_bytes[b.device] + b.size_class <= poolCapBytes()) {
    g_free[{b.device, b.size_class}].push_back(p);
    g_cached_bytes[b.device] += b.size_class;
    return;
  }
  (void)hipFree(p);
}

// The mapped host block a cost's results are published into (one size for every cost): kept per
// device as well — hipHostMalloc + hipHostFree are 0.3 ms of a cost's construction and destruction.
namespace {
std::map<std::pair<int, size_t>, std::vector<void *>> g_free_host;
constexpr size_t kMostHostBlocks = 64;
}  // namespace

hipError_t mappedHostAlloc(void **out, size_t bytes) {
  *out = nullptr;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_free_host.find({device, bytes});
    if (poolCapBytes() != 0 && it != g_free_host.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocCoherent);
}

void mappedHostRelease(int device, void *p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto &list = g_free_host[{device, bytes}];
    if (poolCapBytes() != 0 && list.size() < kMostHostBlocks) {
      list.push_back(p);
      return;
    }
  }
  (void)hipHostFree(p);
}

}  // namespace mopt_detail
