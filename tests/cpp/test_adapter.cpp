// Host-side C++ test of include/kuiper_hip_adapter.hpp: the reference's own op tests
// (test/test_op/test_cu_matmul.cpp:48-106, test_cu_add.cpp:7-75, test_cu_emb.cpp:6-89,
// test_cu_rmsnorm.cpp, test_cu_swiglu.cpp) re-expressed against the adapter's
// kernels_interface.h-shaped functions with a minimal stand-in for tensor::Tensor.
// Built by __graft_entry__.build(); run by tests/test_cpp_adapter_gpu.py on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kuiper_hip_adapter.hpp"

#define REQUIRE(c)                                                        \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);            \
      return 1;                                                           \
    }                                                                     \
  } while (0)

// stand-in for tensor::Tensor (kuiper/include/tensor/tensor.h:12-95): dims + typed pointer
struct Tensor {
  void* data = nullptr;
  std::vector<int32_t> dims;
  bool on_device = true;
  Tensor() = default;
  Tensor(size_t bytes, std::vector<int32_t> d, bool dev = true) : dims(std::move(d)), on_device(dev) {
    if (dev) {
      if (hipMalloc(&data, bytes) != hipSuccess) std::abort();
    } else {
      data = std::malloc(bytes);
    }
  }
  template <class T> const T* ptr() const { return static_cast<const T*>(data); }
  int32_t get_dim(int i) const { return dims.at(i); }
  int32_t dims_size() const { return (int32_t)dims.size(); }
  size_t size() const { size_t n = 1; for (auto d : dims) n *= d; return n; }
};
static Tensor dev_f32(const std::vector<float>& h, std::vector<int32_t> dims) {
  Tensor t(h.size() * 4, std::move(dims));
  (void)hipMemcpy(t.data, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  return t;
}
static std::vector<float> to_host(const Tensor& t) {
  std::vector<float> h(t.size());
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), t.data, h.size() * 4, hipMemcpyDeviceToHost);
  return h;
}
static int h2d(void* dst, const void* src, size_t n, void* stream) {
  return (int)hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)stream);
}

using K = kuiper_hip::Kernels<Tensor>;

int main() {
  int ndev = kh_device_count();
  if (ndev <= 0) {
    std::printf("SKIP: no HIP device (library loaded, %d)\n", ndev);
    return 77;
  }
  hipStream_t stream;
  REQUIRE(hipStreamCreate(&stream) == hipSuccess);
  kuiper_hip::HipConfig cfg{(void*)stream};

  {  // test_matmul_cu.matmul_linear_course: [1,1,-1] x [[1..9]] -> [0,3,6]
    Tensor x = dev_f32({1, 1, -1}, {3});
    Tensor w = dev_f32({1, 2, 3, 4, 5, 6, 7, 8, 9}, {3, 3});
    Tensor y = dev_f32({9, 9, 9}, {3});
    K::matmul(x, w, y, 1.f, &cfg);
    auto h = to_host(y);
    REQUIRE(h[0] == 0.f && h[1] == 3.f && h[2] == 6.f);
  }
  {  // test_add_cu: 2 + 3 = 5 over 4832 elements
    const int n = 4832;
    Tensor a = dev_f32(std::vector<float>(n, 2.f), {n}), b = dev_f32(std::vector<float>(n, 3.f), {n});
    Tensor o = dev_f32(std::vector<float>(n, 0.f), {n});
    K::add(a, b, o, stream);
    for (float v : to_host(o)) REQUIRE(v == 5.f);
  }
  {  // test_emb_cu: table arange(4x512); token 1 -> 512+i
    std::vector<float> tab(4 * 512);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = (float)i;
    Tensor w = dev_f32(tab, {4, 512});
    Tensor toks(sizeof(int32_t), {1}, /*dev=*/false);
    *static_cast<int32_t*>(toks.data) = 1;
    Tensor out = dev_f32(std::vector<float>(512, -1.f), {1, 512});
    int32_t* d_tok = nullptr;
    REQUIRE(hipMalloc((void**)&d_tok, 4) == hipSuccess);
    K::embedding(toks, w, out, 4, stream, d_tok, h2d);
    auto h = to_host(out);
    for (int i = 0; i < 512; ++i) REQUIRE(h[i] == 512.f + i);
  }
  {  // rmsnorm / swiglu vs host formulas, tolerance 1e-5 as in the reference tests
    const int n = 480;
    std::vector<float> x(n), w(n);
    for (int i = 0; i < n; ++i) { x[i] = 0.001f * (i % 97) + 0.1f; w[i] = 0.01f * (i % 13) + 0.5f; }
    double ss = 0; for (float v : x) ss += (double)v * v;
    const double r = 1.0 / std::sqrt(ss / n + 1e-5);
    Tensor xd = dev_f32(x, {n}), wd = dev_f32(w, {n}), od = dev_f32(std::vector<float>(n), {n});
    kuiper_hip::flavor().rms_eps = 1e-5f;
    K::rmsnorm(xd, wd, od, stream);
    auto h = to_host(od);
    for (int i = 0; i < n; ++i) REQUIRE(std::fabs(h[i] - w[i] * r * x[i]) < 1e-5);
    Tensor sd = dev_f32(std::vector<float>(n), {n});
    K::swiglu(xd, wd, sd, stream);
    h = to_host(sd);
    for (int i = 0; i < n; ++i) REQUIRE(std::fabs(h[i] - x[i] / (1 + std::exp(-x[i])) * w[i]) < 1e-5);
  }
  {  // argmax_kernel_cu twin: first maximum
    std::vector<float> l(32000, -1.f);
    l[777] = 4.f; l[31000] = 4.f;
    Tensor ld = dev_f32(l, {32000});
    REQUIRE(K::argmax(ld.ptr<float>(), 32000, stream) == 777);
  }
  {  // error routing: invalid argument reaches the handler instead of aborting silently
    static int seen = 0;
    kuiper_hip::error_handler() = [](int code, const char*) { seen = code; };
    Tensor empty;  // null data
    empty.dims = {4};
    K::add(empty, empty, empty, stream);
    REQUIRE(seen == KH_ERR_INVALID_ARG);
    kuiper_hip::error_handler() = kuiper_hip::default_error_handler;
  }
  std::printf("OK adapter tests passed\n");
  return 0;
}
