// Host-side C++ test of include/kuiper_hip_adapter.hpp with a minimal stand-in for tensor::Tensor
// (the reference's headers need glog/armadillo/CUDA, which do not exist on the GPU box): all
// thirteen operator entry points are called through the adapter's get_*_kernel() getters
// (tests/cpp/adapter_cases.hpp).  The same cases run against the reference's REAL tensor::Tensor
// and typedefs in tests/cpp/test_ref_binding.cpp where /root/reference exists.
// Built by __graft_entry__.build(); run by tests/test_cpp_adapter.py.
#include <hip/hip_runtime.h>

#include <vector>

#include "adapter_cases.hpp"
#include "kuiper_hip_alloc.hpp"

// stand-in for base::DeviceAllocator (kuiper/include/base/alloc.h:7-32): the four virtuals include/kuiper_hip_alloc.hpp
// overrides, so that HipDeviceAllocator is exercised where the reference checkout does not exist
enum class StandInMemcpyKind { kCPU2CPU = 0, kCPU2Dev = 1, kDev2CPU = 2, kDev2Dev = 3 };
class StandInAllocBase {
 public:
  explicit StandInAllocBase(int device_type) : device_type_(device_type) {}
  virtual ~StandInAllocBase() = default;
  virtual int device_type() const { return device_type_; }
  virtual void release(void* ptr) const = 0;
  virtual void* allocate(size_t byte_size) const = 0;
  virtual void memcpy(const void*, void*, size_t, StandInMemcpyKind = StandInMemcpyKind::kCPU2CPU, void* = nullptr,
                      bool = false) const {}
  virtual void memset_zero(void*, size_t, void*, bool = false) {}

 private:
  int device_type_;
};
using HipAllocator = kuiper_hip::HipDeviceAllocator<StandInAllocBase, int, StandInMemcpyKind, 3>;
static std::shared_ptr<HipAllocator> hip_alloc() { return kuiper_hip::allocator_instance<HipAllocator>(); }

// stand-in for tensor::Tensor (kuiper/include/tensor/tensor.h:12-95): dims + typed pointer
struct Tensor {
  void* data = nullptr;
  std::vector<int32_t> dims;
  template <class T> const T* ptr() const { return static_cast<const T*>(data); }
  int32_t get_dim(int i) const { return dims.at(i); }
  int32_t dims_size() const { return (int32_t)dims.size(); }
  size_t size() const { size_t n = 1; for (auto d : dims) n *= d; return n; }
};

struct StandIn {
  using Tensor = ::Tensor;
  using Config = kuiper_hip::HipConfig;
  using DeviceType = int;
  template <class T>
  static Tensor dev(const std::vector<T>& h, std::vector<int32_t> dims) {
    Tensor t;
    t.dims = std::move(dims);
    t.data = hip_alloc()->allocate(h.size() * sizeof(T) + 16);  // pooled HIP memory (kuiper_hip_alloc.hpp)
    if (!t.data) std::abort();
    hip_alloc()->memcpy(h.data(), t.data, h.size() * sizeof(T), StandInMemcpyKind::kCPU2Dev);
    return t;
  }
  static Tensor dev_f32(const std::vector<float>& h, std::vector<int32_t> dims) { return dev(h, std::move(dims)); }
  static Tensor dev_i8(const std::vector<int8_t>& h, std::vector<int32_t> dims) { return dev(h, std::move(dims)); }
  static Tensor host_i32(const std::vector<int32_t>& h, std::vector<int32_t> dims) {
    Tensor t;
    t.dims = std::move(dims);
    t.data = std::malloc(h.size() * 4);
    std::memcpy(t.data, h.data(), h.size() * 4);
    return t;
  }
  static Tensor null_f32(int32_t n) {
    Tensor t;
    t.dims = {n};
    return t;
  }
  static std::vector<float> to_host(const Tensor& t) {
    std::vector<float> h(t.size());
    (void)hipDeviceSynchronize();
    hip_alloc()->memcpy(t.data, h.data(), h.size() * 4, StandInMemcpyKind::kDev2CPU);
    return h;
  }
  static DeviceType device() { return 3; /* kDeviceHIP */ }
  static void set_stream(Config& c, void* s) { c.stream = s; }
};

int main() {
  int ndev = kh_device_count();
  if (ndev <= 0) {
    std::printf("SKIP: no HIP device (library loaded, %d)\n", ndev);
    return 77;
  }
  hipStream_t stream;
  if (hipStreamCreate(&stream) != hipSuccess) return 1;
  int rc = adapter_cases::run<StandIn>((void*)stream);
  // the allocator twin: tag, reuse of a released block, zero fill on a stream, stats
  {
    auto al = hip_alloc();
    auto& pool = kuiper_hip::HipMemoryPool::instance();
    void* a = al->allocate(8192);
    al->release(a);
    void* b = al->allocate(6000);
    float h[4] = {1.f, 2.f, 3.f, 4.f}, back[4] = {9.f, 9.f, 9.f, 9.f};
    al->memcpy(h, b, sizeof h, StandInMemcpyKind::kCPU2Dev, (void*)stream);
    al->memset_zero((char*)b + 4, 8, (void*)stream, /*need_sync=*/true);
    al->memcpy(b, back, sizeof back, StandInMemcpyKind::kDev2CPU);
    if (al->device_type() != 3 || b != a || back[0] != 1.f || back[1] != 0.f || back[2] != 0.f || back[3] != 4.f ||
        pool.stats().busy_blocks == 0) {
      std::printf("FAIL HipDeviceAllocator (stand-in base)\n");
      rc = 1;
    }
    al->release(b);
  }
  if (rc == 0) std::printf("OK adapter tests passed (device memory from kuiper_hip_alloc.hpp)\n");
  return rc;
}
