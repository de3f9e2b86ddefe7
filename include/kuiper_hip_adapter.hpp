// kuiper_hip_adapter.hpp — header-only C++ glue that exposes libkuiper_hip.so with EXACTLY the
// signatures of KuiperLLama's kernel function pointers
//   (kuiper/source/op/kernels/kernels_interface.h:6-44)
// and getters shaped like kernel::get_*_kernel (kernels_interface.h:48-68,
// kernels_interfaces.cpp:21-132), so that a `kDeviceHIP` branch in those getters is one line per op
// (INTEGRATION.md).
//
// Kernels<Tensor, Config, DeviceType> is a template over the three reference types that appear in
// the typedefs:
//   Tensor      tensor::Tensor            (kuiper/include/tensor/tensor.h:12-95: ptr<T>(), get_dim(),
//                                          size())
//   Config      kernel::CudaConfig        (kuiper/include/base/cuda_config.h:6-13: `.stream`)
//   DeviceType  base::DeviceType          (kuiper/include/base/base.h:35-39, enum class : uint8_t)
// With those three plugged in, every static member below IS a value of the matching typedef:
// tests/cpp/test_ref_binding.cpp includes the reference's own kernels_interface.h and
// static_asserts std::is_same for all eleven, then drives real tensor::Tensor objects through the
// getters on the GPU.  tests/cpp/test_adapter.cpp instantiates the same template with a stand-in
// Tensor so the op tests also run where /root/reference does not exist (the GPU box).
//
// Error behaviour: the reference's kernels are `void` and CHECK-abort on precondition failures.
// The adapter keeps `void` signatures and routes a non-zero C-ABI status to a user-replaceable
// handler (default: print + abort, i.e. the reference's LOG(FATAL) semantics).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "kuiper_hip.h"

namespace kuiper_hip {

using ErrorHandler = void (*)(int code, const char* what);
inline void default_error_handler(int code, const char* what) {
  std::fprintf(stderr, "[kuiper_hip] %s failed: %d (%s)\n", what, code, kh_error_string(code));
  std::abort();  // reference: glog CHECK / LOG(FATAL) (kuiper/include/base/base.h:123-134)
}
inline ErrorHandler& error_handler() {
  static ErrorHandler h = default_error_handler;
  return h;
}
inline void check(int code, const char* what) {
  if (code != KH_OK) error_handler()(code, what);
}

// Runtime replacements for the reference's compile-time switches:
//   LLAMA3_SUPPORT -> {KH_ROPE_HALF, eps 1e-5, theta 5e5}; QWEN2_SUPPORT -> {KH_ROPE_HALF, eps 1e-6,
//   theta 1e6}; neither -> {KH_ROPE_INTERLEAVED, eps 1e-5, theta 1e4}
//   (cpu/rope_kernel.cpp:3-16,43,83; cpu/rmsnorm_kernel.cpp:24-28).
struct Flavor {
  int32_t rope_mode = KH_ROPE_INTERLEAVED;
  float rms_eps = 1e-5f;
  float rope_theta = 10000.0f;
};
// The process default - what the reference fixes at compile time.
inline Flavor& flavor() {
  static Flavor f;
  return f;
}
// Per-stream flavours, so that a Llama and a Qwen model can live in one process (the reference cannot: one
// build, one #ifdef).  Every model object of the reference owns a stream (llama3.cpp:117-125) and passes it to
// every kernel; RMSNorm / RoPE / the sin-cos table look the flavour up by that stream and fall back to the
// process default.  Nothing binds by itself.  The model's stream is private to the reference's model classes, but the
// one-line forward a port writes for kernel::sin_cos_cache_calc_cu - the first kernel
// Model::init issues, on that very stream - sees it: `bind_flavor(stream, flavor())` there pins the default of
// that moment to the model (tests/cpp/test_ref_model.cpp does exactly this): set flavor(), call model.init(), and
// the model keeps that flavour whatever flavor() is changed to for the next model.
namespace detail {
struct BoundFlavor {
  void* stream;
  Flavor f;
};
inline std::vector<BoundFlavor>& bound_flavors() {
  static std::vector<BoundFlavor> v;
  return v;
}
inline std::mutex& bound_flavors_mu() {
  static std::mutex m;
  return m;
}
}  // namespace detail
inline void bind_flavor(void* stream, const Flavor& f) {
  if (!stream) return;  // the null stream always follows the process default
  std::lock_guard<std::mutex> g(detail::bound_flavors_mu());
  for (auto& b : detail::bound_flavors())
    if (b.stream == stream) {
      b.f = f;
      return;
    }
  detail::bound_flavors().push_back({stream, f});
}
inline void unbind_flavor(void* stream) {  // call when the stream is destroyed (handles are reused)
  std::lock_guard<std::mutex> g(detail::bound_flavors_mu());
  auto& v = detail::bound_flavors();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].stream == stream) {
      v[i] = v.back();
      v.pop_back();
      return;
    }
}
inline Flavor flavor_of(void* stream) {
  if (stream) {
    std::lock_guard<std::mutex> g(detail::bound_flavors_mu());
    for (const auto& b : detail::bound_flavors())
      if (b.stream == stream) return b.f;
  }
  return flavor();
}

// CudaConfig twin (kuiper/include/base/cuda_config.h:6-13): only the stream is used.
struct HipConfig {
  void* stream = nullptr;
};

template <class Tensor, class Config = HipConfig, class DeviceType = int>
struct Kernels {
  // the eleven typedefs of kernels_interface.h:6-44, spelled over the template parameters
  using AddKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, void*);
  using MatmulKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, float, const Config*);
  using MatmulKernelQuant = void (*)(const Tensor&, const Tensor&, const Tensor&, int32_t,
                                     const Tensor&, const Config*);
  using EmbeddingKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, int32_t, void*);
  using SwigluKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, void*);
  using MHAKernel = void (*)(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                             const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                             const Tensor&, DeviceType, Config*);
  using RMSNormKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, void*);
  using RoPEKernel = void (*)(int32_t, int32_t, int32_t, const Tensor&, const Tensor&,
                              const Tensor&, const Tensor&, const Tensor&, void*);
  using ScaleKernel = void (*)(float, const Tensor&, void*);
  using SoftmaxInplaceKernel = void (*)(const Tensor&, void*);
  using ScaleSumKernel = void (*)(const Tensor&, const Tensor&, const Tensor&, int, int, int, void*);

  template <class T>
  static T* mut(const Tensor& t) {  // the reference passes outputs as const Tensor& too
    return const_cast<T*>(t.template ptr<T>());
  }
  static void* stream_of(const Config* c) { return c ? (void*)c->stream : nullptr; }

  // AddKernel (kernels_interface.h:6-7)
  static void add(const Tensor& in1, const Tensor& in2, const Tensor& out, void* stream) {
    check(kh_add_f32(in1.template ptr<float>(), in2.template ptr<float>(), mut<float>(out),
                     (int32_t)in1.size(), stream),
          "kh_add_f32");
  }
  // MatmulKernel (kernels_interface.h:9-10): weight [K, M] row-major, input [M]
  static void matmul(const Tensor& input, const Tensor& weight, const Tensor& output, float scale,
                     const Config* config) {
    const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
    check(kh_matmul_f32(input.template ptr<float>(), weight.template ptr<float>(),
                        mut<float>(output), M, K, scale, stream_of(config)),
          "kh_matmul_f32");
  }
  // MatmulKernelQuant (kernels_interface.h:12-14)
  static void matmul_quant8(const Tensor& input, const Tensor& weight, const Tensor& output,
                            int32_t group_size, const Tensor& scale, const Config* config) {
    const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
    check(kh_matmul_q8(input.template ptr<float>(), weight.template ptr<int8_t>(),
                       scale.template ptr<float>(), group_size, mut<float>(output), M, K,
                       stream_of(config)),
          "kh_matmul_q8");
  }
  // EmbeddingKernel (kernels_interface.h:16-17).  `input` is the HOST tensor of token ids the
  // reference hands over (emb_kernel.cu:25-29 uploads it per call); the ids go to the GPU in the
  // kernel arguments (kh_embedding_f32_host), so nothing is staged or owned here.
  static void embedding(const Tensor& input, const Tensor& weight, const Tensor& output,
                        int32_t vocab_size, void* stream) {
    check(kh_embedding_f32_host(input.template ptr<int32_t>(), (int32_t)input.size(),
                                weight.template ptr<float>(), mut<float>(output),
                                weight.get_dim(1), vocab_size, stream),
          "kh_embedding_f32_host");
  }
  // SwigluKernel (kernels_interface.h:19-20)
  static void swiglu(const Tensor& in1, const Tensor& in2, const Tensor& out, void* stream) {
    check(kh_swiglu_f32(in1.template ptr<float>(), in2.template ptr<float>(), mut<float>(out),
                        (int32_t)in1.size(), stream),
          "kh_swiglu_f32");
  }
  // MHAKernel (kernels_interface.h:22-28); device_type is what the reference's mha_kernel uses to
  // pick its inner CPU/CUDA helpers (cpu/mha_kernel.cpp:25-57) - there is only one device here
  static void mha(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len,
                  int32_t kv_dim, int32_t kv_mul, int32_t head_size, const Tensor& mha_out,
                  const Tensor& query, const Tensor& score, const Tensor& key_cache,
                  const Tensor& value_cache, DeviceType /*device_type*/, Config* config) {
    check(kh_mha_f32(nullptr, pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
                     mut<float>(mha_out), query.template ptr<float>(), mut<float>(score),
                     key_cache.template ptr<float>(), value_cache.template ptr<float>(),
                     stream_of(config)),
          "kh_mha_f32");
  }
  // RMSNormKernel (kernels_interface.h:30-31)
  static void rmsnorm(const Tensor& input, const Tensor& weight, const Tensor& output,
                      void* stream) {
    check(kh_rmsnorm_f32(input.template ptr<float>(), weight.template ptr<float>(),
                         mut<float>(output), (int32_t)input.size(), flavor_of(stream).rms_eps, stream),
          "kh_rmsnorm_f32");
  }
  // RoPEKernel (kernels_interface.h:33-36): input_pos is a HOST int32 tensor in the reference
  // (op/rope.cpp:38-39)
  static void rope(int32_t dim, int32_t kv_dim, int32_t head_size, const Tensor& input_q,
                   const Tensor& input_k, const Tensor& input_pos, const Tensor& sin_cache,
                   const Tensor& cos_cache, void* stream) {
    const int32_t pos = *input_pos.template ptr<int32_t>();
    check(kh_rope_f32(dim, kv_dim, head_size, mut<float>(input_q), mut<float>(input_k), nullptr,
                      pos, sin_cache.template ptr<float>(), cos_cache.template ptr<float>(),
                      flavor_of(stream).rope_mode, stream),
          "kh_rope_f32");
  }
  // sin_cos_cache_calc_cu (cuda/rope_kernel.cuh:9-10); Stream = cudaStream_t in the reference.
  // theta is an #ifdef there (cuda/rope_kernel.cu:124-151), the flavour's rope_theta here (the stream's bound
  // flavour, else the process default - see bind_flavor).
  template <class Stream>
  static void sin_cos_cache_calc(int head_size, int max_seq_len, const Tensor& sin_cache,
                                 const Tensor& cos_cache, Stream stream) {
    check(kh_sincos_cache_f32(head_size, max_seq_len, flavor_of((void*)stream).rope_theta, mut<float>(sin_cache),
                              mut<float>(cos_cache), (void*)stream),
          "kh_sincos_cache_f32");
  }
  // argmax_kernel_cu (cuda/argmax_kernel.cuh:4)
  static size_t argmax(const float* logits, size_t size, void* stream) {
    int64_t idx = -1;
    check(kh_argmax_f32_host(logits, (int64_t)size, &idx, stream), "kh_argmax_f32_host");
    return (size_t)idx;
  }
  // CPU-only helpers of the reference (kernels_interface.h:38-44), available on device here
  static void scale(float s, const Tensor& input, void* stream) {
    check(kh_scale_f32(s, mut<float>(input), (int32_t)input.size(), stream), "kh_scale_f32");
  }
  static void softmax_inplace(const Tensor& input, void* stream) {
    check(kh_softmax_f32(mut<float>(input), (int32_t)input.size(), stream), "kh_softmax_f32");
  }
  static void scale_sum(const Tensor& value, const Tensor& scale_t, const Tensor& output, int t,
                        int size, int stride, void* stream) {
    check(kh_scale_sum_f32(value.template ptr<float>(), scale_t.template ptr<float>(),
                           mut<float>(output), t, size, stride, stream),
          "kh_scale_sum_f32");
  }

  // kernel::get_*_kernel(base::DeviceType::kDeviceHIP) (kernels_interfaces.cpp:21-132): what each
  // getter returns for the new enumerator
  static AddKernel get_add_kernel() { return &Kernels::add; }
  static EmbeddingKernel get_emb_kernel() { return &Kernels::embedding; }
  static MatmulKernel get_matmul_kernel() { return &Kernels::matmul; }
  static MatmulKernelQuant get_matmul_kernel_quant8() { return &Kernels::matmul_quant8; }
  static MHAKernel get_mha_kernel() { return &Kernels::mha; }
  static RMSNormKernel get_rmsnorm_kernel() { return &Kernels::rmsnorm; }
  static RoPEKernel get_rope_kernel() { return &Kernels::rope; }
  static ScaleKernel get_scale_kernel() { return &Kernels::scale; }
  static SoftmaxInplaceKernel get_softmax_kernel() { return &Kernels::softmax_inplace; }
  static SwigluKernel get_swiglu_kernel() { return &Kernels::swiglu; }
  static ScaleSumKernel get_scale_sum_kernel() { return &Kernels::scale_sum; }
};

}  // namespace kuiper_hip
