// Host-side C++ test of include/kuiper_hip_adapter.hpp with a minimal stand-in for tensor::Tensor
// (the reference's headers need glog/armadillo/CUDA, which do not exist on the GPU box): all
// thirteen operator entry points are called through the adapter's get_*_kernel() getters
// (tests/cpp/adapter_cases.hpp).  The same cases run against the reference's REAL tensor::Tensor
// and typedefs in tests/cpp/test_ref_binding.cpp where /root/reference exists.
// Built by __graft_entry__.build(); run by tests/test_cpp_adapter.py.
#include <hip/hip_runtime.h>

#include <vector>

#include "adapter_cases.hpp"

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
    if (hipMalloc(&t.data, h.size() * sizeof(T) + 16) != hipSuccess) std::abort();
    (void)hipMemcpy(t.data, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
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
    (void)hipMemcpy(h.data(), t.data, h.size() * 4, hipMemcpyDeviceToHost);
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
  const int rc = adapter_cases::run<StandIn>((void*)stream);
  if (rc == 0) std::printf("OK adapter tests passed\n");
  return rc;
}
