// kuiper_hip_alloc.hpp — the allocator / tensor-transfer / stream layer of the MI355X drop-in, header-only.
//
// Replaces, for kDeviceHIP tensors,
//   base::CUDADeviceAllocator + Factory        kuiper/include/base/alloc.h:45-93, kuiper/source/base/alloc_cu.cpp:7-112
//   DeviceAllocator::memcpy / memset_zero      kuiper/source/base/alloc.cpp:4-60 (the CUDA branches)
//   tensor::Tensor::to_cuda / to_cpu           kuiper/source/tensor/tensor.cpp:104-137
//   kernel::CudaConfig + stream creation       kuiper/include/base/cuda_config.h:6-13, kuiper/source/model/llama3.cpp:117-125
// so that a KuiperLLama build for the MI355X needs no CUDA runtime spelling for memory: tensors are allocated,
// filled, copied and freed through this header and tagged kDeviceHIP; the kernels come from kuiper_hip.h through
// kuiper_hip_adapter.hpp (whose HipConfig is the non-owning view the kernels take).  (tests/cpp/test_ref_binding.cpp and test_ref_layers.cpp run the reference's own Tensor /
// Buffer / op::*Layer classes on it and check that the test-only CUDA stand-in is never called.)
//
// Everything is a template over the reference's own types, so this header includes nothing of the reference and
// binds to it exactly (as kuiper_hip_adapter.hpp does):
//   using HipAllocator = kuiper_hip::HipDeviceAllocator<base::DeviceAllocator, base::DeviceType, base::MemcpyKind,
//                                                       base::DeviceType(3) /* kDeviceHIP */>;
//   auto alloc = kuiper_hip::allocator_instance<HipAllocator>();
//   tensor::Tensor w(base::DataType::kDataTypeFp32, K, M, true, cpu_alloc);  ...  kuiper_hip::to_hip(w, alloc, stream);
//
// Pool policy (alloc_cu.cpp:7-112 keeps freed blocks for reuse; same idea, tighter fit, thread-safe):
//   * requests > 1 MiB take the SMALLEST idle block that fits with < 1 MiB of slack (the reference's rule);
//   * smaller requests take the smallest idle block of at most twice their size (the reference hands out the FIRST
//     idle block that fits, however large);
//   * release() parks the block; when more than 1 GiB of small blocks sit idle on a device they are returned to the
//     driver (the reference's rule, alloc_cu.cpp:73-92); pointers the pool does not know go straight to hipFree;
//   * one pool per device, a mutex around it (the reference's maps are unsynchronised).
#ifndef KUIPER_HIP_ALLOC_HPP
#define KUIPER_HIP_ALLOC_HPP
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace kuiper_hip {

class HipMemoryPool {
 public:
  static HipMemoryPool& instance() {
    static HipMemoryPool p;
    return p;
  }
  void* allocate(size_t bytes) {
    if (!bytes) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu_);
    PerDevice& d = dev_[dev];
    const bool big = bytes > kBig;
    auto& idle = big ? d.idle_big : d.idle_small;
    auto it = idle.lower_bound(bytes);
    if (it != idle.end() && (big ? it->first - bytes < kBig : it->first <= 2 * bytes + 256)) {
      void* p = it->second;
      if (!big) d.idle_small_bytes -= it->first;
      d.busy.emplace(p, it->first);
      idle.erase(it);
      return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      trim_locked(d, dev);  // give the driver back what sits idle, then try once more
      if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    }
    d.busy.emplace(p, bytes);
    return p;
  }
  void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : dev_) {
      PerDevice& d = kv.second;
      auto it = d.busy.find(p);
      if (it == d.busy.end()) continue;
      const size_t bytes = it->second;
      d.busy.erase(it);
      if (bytes > kBig) {
        d.idle_big.emplace(bytes, p);
      } else {
        d.idle_small.emplace(bytes, p);
        d.idle_small_bytes += bytes;
        if (d.idle_small_bytes > kTrimAbove) trim_small_locked(d, kv.first);
      }
      return;
    }
    (void)hipFree(p);  // not ours (alloc_cu.cpp:110-111 does the same)
  }
  // return every idle block of every device to the driver
  void trim() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : dev_) trim_locked(kv.second, kv.first);
  }
  struct Stats {
    size_t busy_blocks = 0, busy_bytes = 0, idle_blocks = 0, idle_bytes = 0;
  };
  Stats stats() {
    std::lock_guard<std::mutex> lk(mu_);
    Stats s;
    for (auto& kv : dev_) {
      for (auto& b : kv.second.busy) {
        ++s.busy_blocks;
        s.busy_bytes += b.second;
      }
      for (auto* m : {&kv.second.idle_big, &kv.second.idle_small})
        for (auto& b : *m) {
          ++s.idle_blocks;
          s.idle_bytes += b.first;
        }
    }
    return s;
  }

 private:
  static constexpr size_t kBig = 1024 * 1024;                // alloc_cu.cpp:11
  static constexpr size_t kTrimAbove = 1024ull * 1024 * 1024;  // alloc_cu.cpp:74
  struct PerDevice {
    std::map<void*, size_t> busy;
    std::multimap<size_t, void*> idle_big, idle_small;
    size_t idle_small_bytes = 0;
  };
  static void free_all(std::multimap<size_t, void*>& m, int dev) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    for (auto& b : m) (void)hipFree(b.second);
    m.clear();
    if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
  }
  static void trim_small_locked(PerDevice& d, int dev) {
    free_all(d.idle_small, dev);
    d.idle_small_bytes = 0;
  }
  static void trim_locked(PerDevice& d, int dev) {
    trim_small_locked(d, dev);
    free_all(d.idle_big, dev);
  }
  std::mutex mu_;
  std::map<int, PerDevice> dev_;
};

// base::DeviceAllocator for kDeviceHIP memory.  AllocBase = base::DeviceAllocator, DeviceTypeT = base::DeviceType,
// MemcpyKindT = base::MemcpyKind (enumerators 0..3 = CPU2CPU, CPU2dev, dev2CPU, dev2dev, alloc.h:7-12), kTag = the
// device tag the tensors carry.
template <class AllocBase, class DeviceTypeT, class MemcpyKindT, DeviceTypeT kTag>
class HipDeviceAllocator : public AllocBase {
 public:
  using memcpy_kind_t = MemcpyKindT;
  using device_type_t = DeviceTypeT;
  static constexpr DeviceTypeT kDeviceTag = kTag;
  HipDeviceAllocator() : AllocBase(kTag) {}
  void* allocate(size_t byte_size) const override { return HipMemoryPool::instance().allocate(byte_size); }
  void release(void* ptr) const override { HipMemoryPool::instance().release(ptr); }
  // alloc.cpp:4-43.  stream = hipStream_t as void*; without a stream the copy is synchronous, with one it is
  // asynchronous on it (pageable host memory: the runtime stages it) unless need_sync.
  void memcpy(const void* src, void* dst, size_t byte_size, MemcpyKindT kind = static_cast<MemcpyKindT>(0),
              void* stream = nullptr, bool need_sync = false) const override {
    if (!src || !dst || !byte_size) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int k = static_cast<int>(kind);
    if (k == 0) {
      std::memcpy(dst, src, byte_size);
      return;
    }
    const hipMemcpyKind hk = k == 1 ? hipMemcpyHostToDevice : (k == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    if (s)
      (void)hipMemcpyAsync(dst, src, byte_size, hk, s);
    else
      (void)hipMemcpy(dst, src, byte_size, hk);
    if (need_sync) (void)hipDeviceSynchronize();
  }
  // alloc.cpp:45-60
  void memset_zero(void* ptr, size_t byte_size, void* stream, bool need_sync = false) override {
    if (!ptr || !byte_size) return;
    if (stream)
      (void)hipMemsetAsync(ptr, 0, byte_size, static_cast<hipStream_t>(stream));
    else
      (void)hipMemset(ptr, 0, byte_size);
    if (need_sync) (void)hipDeviceSynchronize();
  }
};

// CUDADeviceAllocatorFactory::get_instance() (alloc.h:82-93)
template <class Alloc>
std::shared_ptr<Alloc> allocator_instance() {
  static std::shared_ptr<Alloc> inst = std::make_shared<Alloc>();
  return inst;
}

// Tensor::to_cuda (tensor.cpp:104-119): a host tensor becomes a device tensor with the same shape and bytes.
// Tensors that are not on `host_tag` are left alone, like the reference.  Returns false when the tensor is empty
// or the allocation fails.
template <class Tensor, class Alloc>
bool to_hip(Tensor& t, const std::shared_ptr<Alloc>& alloc, typename Alloc::device_type_t host_tag, void* stream = nullptr) {
  if (t.is_empty() || !alloc) return false;
  if (t.device_type() != host_tag) return true;
  Tensor dev(t.data_type(), t.dims(), /*need_alloc=*/true, alloc);
  if (dev.is_empty()) return false;
  alloc->memcpy(t.get_buffer()->ptr(), dev.get_buffer()->ptr(), t.byte_size(), static_cast<typename Alloc::memcpy_kind_t>(1),
                stream);
  t = dev;  // shares the device buffer; the host buffer is released with its last owner
  return true;
}
// Tensor::to_cpu (tensor.cpp:121-137): blocking copy back into a tensor owned by `host_alloc`.
template <class Tensor, class Alloc, class HostAlloc>
bool to_host(Tensor& t, const std::shared_ptr<Alloc>& alloc, const std::shared_ptr<HostAlloc>& host_alloc) {
  if (t.is_empty() || !alloc || !host_alloc) return false;
  if (t.device_type() != Alloc::kDeviceTag) return true;
  Tensor host(t.data_type(), t.dims(), /*need_alloc=*/true, host_alloc);
  if (host.is_empty()) return false;
  alloc->memcpy(t.get_buffer()->ptr(), host.get_buffer()->ptr(), t.byte_size(), static_cast<typename Alloc::memcpy_kind_t>(2));
  t = host;
  return true;
}

// kernel::CudaConfig (cuda_config.h:6-13) + the stream the model creates for itself (llama3.cpp:117-125:
// cudaSetDevice, make_shared<CudaConfig>, cudaStreamCreate): same shape, HIP types.  The reference's kernel
// typedefs take `const kernel::CudaConfig*` and read ->stream; a port changes the member's type, nothing else.
struct HipStreamConfig {
  hipStream_t stream = nullptr;
  HipStreamConfig() = default;
  HipStreamConfig(const HipStreamConfig&) = delete;
  HipStreamConfig& operator=(const HipStreamConfig&) = delete;
  ~HipStreamConfig() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};
inline std::shared_ptr<HipStreamConfig> make_stream_config(int device = 0) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  auto c = std::make_shared<HipStreamConfig>();
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return c;
}

}  // namespace kuiper_hip
#endif  // KUIPER_HIP_ALLOC_HPP
