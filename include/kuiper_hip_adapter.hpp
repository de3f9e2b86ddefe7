// kuiper_hip_adapter.hpp — header-only C++ glue that exposes libkuiper_hip.so with the
// SIGNATURES of KuiperLLama's kernel function pointers
//   (kuiper/source/op/kernels/kernels_interface.h:6-44)
// so that a `kDeviceHIP` branch in kuiper/source/op/kernels/kernels_interfaces.cpp:21-132 is a
// one-line-per-op change.
//
// It is a template over the tensor type so it compiles both against the reference's
// tensor::Tensor (kuiper/include/tensor/tensor.h:12-95: ptr<T>(), get_dim(), dims_size(), size())
// and against the tiny stand-in used by this repo's tests — the reference headers pull in
// glog/armadillo/CUDA which do not exist in this build environment.
//
// Error behaviour: the reference's kernels are `void` and CHECK-abort on precondition failures.
// The adapter keeps `void` signatures and routes a non-zero C-ABI status to a user-replaceable
// handler (default: print + abort, i.e. the reference's LOG(FATAL) semantics).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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
//   LLAMA3_SUPPORT -> {KH_ROPE_HALF, eps 1e-5}; QWEN2_SUPPORT -> {KH_ROPE_HALF, eps 1e-6};
//   neither -> {KH_ROPE_INTERLEAVED, eps 1e-5}.
struct Flavor {
  int32_t rope_mode = KH_ROPE_INTERLEAVED;
  float rms_eps = 1e-5f;
};
inline Flavor& flavor() {
  static Flavor f;
  return f;
}

// CudaConfig twin (kuiper/include/base/cuda_config.h:6-13): only the stream is used.
struct HipConfig {
  void* stream = nullptr;
};

template <class Tensor>
struct Kernels {
  // AddKernel (kernels_interface.h:6-7)
  static void add(const Tensor& in1, const Tensor& in2, const Tensor& out, void* stream) {
    check(kh_add_f32(in1.template ptr<float>(), in2.template ptr<float>(),
                     const_cast<float*>(out.template ptr<float>()), (int32_t)in1.size(), stream),
          "kh_add_f32");
  }
  // MatmulKernel (kernels_interface.h:9-10): weight [K, M] row-major, input [M]
  template <class Config>
  static void matmul(const Tensor& input, const Tensor& weight, const Tensor& output, float scale,
                     const Config* config) {
    const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
    check(kh_matmul_f32(input.template ptr<float>(), weight.template ptr<float>(),
                        const_cast<float*>(output.template ptr<float>()), M, K, scale,
                        config ? (void*)config->stream : nullptr),
          "kh_matmul_f32");
  }
  // MatmulKernelQuant (kernels_interface.h:12-14)
  template <class Config>
  static void matmul_quant8(const Tensor& input, const Tensor& weight, const Tensor& output,
                            int32_t group_size, const Tensor& scale, const Config* config) {
    const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
    check(kh_matmul_q8(input.template ptr<float>(), weight.template ptr<int8_t>(),
                       scale.template ptr<float>(), group_size,
                       const_cast<float*>(output.template ptr<float>()), M, K,
                       config ? (void*)config->stream : nullptr),
          "kh_matmul_q8");
  }
  // EmbeddingKernel (kernels_interface.h:16-17).  The reference hands a HOST token tensor and
  // uploads it inside the kernel launcher (cuda/emb_kernel.cu:25-29); `d_tokens` is the caller's
  // device staging buffer for that upload (>= n tokens), which keeps the ABI allocation-free.
  static void embedding(const Tensor& h_tokens, const Tensor& weight, const Tensor& output,
                        int32_t vocab_size, void* stream, int32_t* d_tokens,
                        int (*h2d)(void* dst, const void* src, size_t n, void* stream)) {
    const int32_t n = (int32_t)h_tokens.size();
    check(h2d(d_tokens, h_tokens.template ptr<int32_t>(), sizeof(int32_t) * (size_t)n, stream),
          "token upload");
    check(kh_embedding_f32(d_tokens, n, weight.template ptr<float>(),
                           const_cast<float*>(output.template ptr<float>()), weight.get_dim(1),
                           vocab_size, stream),
          "kh_embedding_f32");
  }
  // SwigluKernel (kernels_interface.h:19-20)
  static void swiglu(const Tensor& in1, const Tensor& in2, const Tensor& out, void* stream) {
    check(kh_swiglu_f32(in1.template ptr<float>(), in2.template ptr<float>(),
                        const_cast<float*>(out.template ptr<float>()), (int32_t)in1.size(), stream),
          "kh_swiglu_f32");
  }
  // MHAKernel (kernels_interface.h:22-28)
  template <class Config>
  static void mha(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len,
                  int32_t kv_dim, int32_t kv_mul, int32_t head_size, const Tensor& mha_out,
                  const Tensor& query, const Tensor& score, const Tensor& key_cache,
                  const Tensor& value_cache, int /*device_type*/, Config* config) {
    check(kh_mha_f32(nullptr, pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
                     const_cast<float*>(mha_out.template ptr<float>()),
                     query.template ptr<float>(), const_cast<float*>(score.template ptr<float>()),
                     key_cache.template ptr<float>(), value_cache.template ptr<float>(),
                     config ? (void*)config->stream : nullptr),
          "kh_mha_f32");
  }
  // RMSNormKernel (kernels_interface.h:30-31)
  static void rmsnorm(const Tensor& input, const Tensor& weight, const Tensor& output,
                      void* stream) {
    check(kh_rmsnorm_f32(input.template ptr<float>(), weight.template ptr<float>(),
                         const_cast<float*>(output.template ptr<float>()), (int32_t)input.size(),
                         flavor().rms_eps, stream),
          "kh_rmsnorm_f32");
  }
  // RoPEKernel (kernels_interface.h:33-36): input_pos is a HOST int32 tensor in the reference
  // (op/rope.cpp:38-39)
  static void rope(int32_t dim, int32_t kv_dim, int32_t head_size, const Tensor& input_q,
                   const Tensor& input_k, const Tensor& input_pos, const Tensor& sin_cache,
                   const Tensor& cos_cache, void* stream) {
    const int32_t pos = *input_pos.template ptr<int32_t>();
    check(kh_rope_f32(dim, kv_dim, head_size, const_cast<float*>(input_q.template ptr<float>()),
                      const_cast<float*>(input_k.template ptr<float>()), nullptr, pos,
                      sin_cache.template ptr<float>(), cos_cache.template ptr<float>(),
                      flavor().rope_mode, stream),
          "kh_rope_f32");
  }
  // sin_cos_cache_calc_cu (cuda/rope_kernel.cuh:9-10); theta is an #ifdef in the reference
  static void sin_cos_cache_calc(int head_size, int max_seq_len, float theta,
                                 const Tensor& sin_cache, const Tensor& cos_cache, void* stream) {
    check(kh_sincos_cache_f32(head_size, max_seq_len, theta,
                              const_cast<float*>(sin_cache.template ptr<float>()),
                              const_cast<float*>(cos_cache.template ptr<float>()), stream),
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
    check(kh_scale_f32(s, const_cast<float*>(input.template ptr<float>()), (int32_t)input.size(),
                       stream),
          "kh_scale_f32");
  }
  static void softmax_inplace(const Tensor& input, void* stream) {
    check(kh_softmax_f32(const_cast<float*>(input.template ptr<float>()), (int32_t)input.size(),
                         stream),
          "kh_softmax_f32");
  }
  static void scale_sum(const Tensor& value, const Tensor& scale_t, const Tensor& output, int t,
                        int size, int stride, void* stream) {
    check(kh_scale_sum_f32(value.template ptr<float>(), scale_t.template ptr<float>(),
                           const_cast<float*>(output.template ptr<float>()), t, size, stride,
                           stream),
          "kh_scale_sum_f32");
  }
};

}  // namespace kuiper_hip
