// test_ref_model.cpp — the reference's OWN model object decoding on the MI355X through the operator boundary.
//
// Linked here, compiled where they lie under $KUIPER_REF (oracle/Makefile target `ref_model`, nothing copied):
//   kuiper/source/model/{model,llama3,raw_model_data}.cpp   Model::read_model_file / init / create_layers / forward /
//                                                            predict (model.cpp:41-263, llama3.cpp:107-745)
//   kuiper/source/sampler/argmax_sampler.cpp, kuiper/source/op/encode.cpp (SpeEncodeLayer)
//   + the op::*Layer, tensor, buffer and allocator sources of `ref_layers`
// on top of tests/cpp/kernels_interfaces_hip.cpp (the kernel::get_*_kernel getters with the HIP branch of
// INTEGRATION.md §1) and libkuiper_hip.so.  This TU supplies the two kernels the reference calls directly instead
// of through a getter - kernel::sin_cos_cache_calc_cu (llama3.cpp:133-140) and kernel::argmax_kernel_cu
// (argmax_sampler.cpp:10) - as one-line forwards to the adapter, exactly what INTEGRATION.md §2 tells a maintainer
// to do, and the CPU twin of the former as a loud error (the CPU kernels are not linked).
//
// The reference's model code is written against kDeviceCUDA (llama3.cpp:117, 425-500: CUDADeviceAllocator,
// Tensor::to_cuda) and cannot be edited from this repo, so in THIS binary its tensors carry that tag, the memory
// behind them is MI355X memory (tests/cpp/ref_stubs/cuda_runtime_api.h forwards the reference's cudaMalloc / cudaMemcpy
// to HIP) and the getters treat kDeviceCUDA as the HIP branch (-DKH_REF_CUDA_TAG_IS_HIP).  test_ref_binding /
// test_ref_layers are the builds where every tensor comes from include/kuiper_hip_alloc.hpp instead.
//
// The same TU builds the Qwen twin: -DKH_REF_MODEL_QWEN2 selects model::Qwen2Model (kuiper/source/model/qwen2.cpp:
// bias wiring :147-167, 307-332) - a second binary, oracle/_ref/test_ref_model_qwen2, because the reference's llama3.h
// and qwen2.h share one include guard.  It is compiled WITHOUT -DQWEN2_SUPPORT (that switch only selects the tiktoken
// tokenizer, the rotate-half RoPE and eps 1e-6 in the reference's own kernels); the RoPE flavour / theta / eps the
// model runs with are runtime state here (kuiper_hip::flavor(), bound to the model's stream at init).
//
// usage: test_ref_model <model.bin> <tokenizer.model> <steps> <prompt ids, comma separated> <expected words | ->
//                       [--quant] [--flavor rope_mode,rms_eps,rope_theta]
// Runs the loop of demo/main.cpp:5-47 (prompt phase: predict with is_prompt, then greedy) and prints the words, the
// tokens/s of the reference's per-op host loop, and OK when the words equal the expected ones.
//   --quant   is_quant_model = true: the int8 reader of llama3.cpp:184-288 (create_param_quant_layers) and
//             MatmulLayer's int8 forward through get_matmul_kernel_quant8
//   --flavor  the reference's compile-time switches as runtime values (e.g. 1,1e-5,500000 = LLAMA3_SUPPORT)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <cstring>

#include "kernels_interface.h"
#include "kuiper_hip_adapter.hpp"
#ifdef KH_REF_MODEL_QWEN2
#include "model/qwen2.h"
using RefModel = model::Qwen2Model;
static const char* const kRefModelName = "Qwen2Model";
#else
#include "model/llama3.h"
using RefModel = model::LLama2Model;
static const char* const kRefModelName = "LLama2Model";
#endif

using HipK = kuiper_hip::Kernels<tensor::Tensor, kernel::CudaConfig, base::DeviceType>;

namespace kernel {
// cuda/rope_kernel.cuh:9-10 and cuda/argmax_kernel.cuh:4: the two direct calls of the model level
void sin_cos_cache_calc_cu(int head_size, int max_seq_len, const tensor::Tensor& sin_cache, const tensor::Tensor& cos_cache,
                           cudaStream_t stream) {
  // Model::init's first kernel, on the model's own (private) stream: pin the flavour current NOW to that stream, so
  // the model keeps it when the process default moves on to the next model (kuiper_hip_adapter.hpp: bind_flavor)
  kuiper_hip::bind_flavor((void*)stream, kuiper_hip::flavor());
  HipK::sin_cos_cache_calc<cudaStream_t>(head_size, max_seq_len, sin_cache, cos_cache, stream);
}
size_t argmax_kernel_cu(const float* input_ptr, size_t size, void* stream) { return HipK::argmax(input_ptr, size, stream); }
// cpu/rope_kernel.h: referenced by the CPU branch of LLama2Model::init only
void sin_cos_cache_calc_cpu(int, int, float*, float*) {
  std::fprintf(stderr, "the CPU backend is not linked into this binary\n");
  std::abort();
}
}  // namespace kernel

static std::vector<int32_t> parse_ids(const char* s) {
  std::vector<int32_t> v;
  const char* p = s;
  while (*p) {
    char* e = nullptr;
    const long x = std::strtol(p, &e, 10);
    if (e == p) break;
    v.push_back((int32_t)x);
    p = *e == ',' ? e + 1 : e;
  }
  return v;
}

// demo/main.cpp:5-47 with the prompt given as token ids
static std::vector<int32_t> generate(const RefModel& model, std::vector<int32_t> tokens, int total_steps) {
  const int32_t prompt_len = (int32_t)tokens.size();
  int32_t pos = 0, next = -1;
  bool is_prompt = true;
  const auto& prompt_embedding = model.embedding(tokens);
  tensor::Tensor pos_tensor = model.get_buffer(model::ModelBufferType::kInputPos);
  std::vector<int32_t> words;
  std::vector<int32_t> cur = tokens;
  while (pos < total_steps) {
    pos_tensor.index<int32_t>(0) = pos;
    if (pos < prompt_len - 1) {
      tensor::Tensor input = model.fill_input(pos_tensor, prompt_embedding, is_prompt);
      model.predict(input, pos_tensor, is_prompt, next);
    } else {
      is_prompt = false;
      cur = std::vector<int32_t>{pos == prompt_len - 1 ? tokens[(size_t)pos] : next};
      const auto& token_embedding = model.embedding(cur);
      tensor::Tensor input = model.fill_input(pos_tensor, token_embedding, is_prompt);
      model.predict(input, pos_tensor, is_prompt, next);
    }
    if (is_prompt) {
      next = tokens.at((size_t)pos + 1);
      words.push_back(next);
    } else {
      words.push_back(next);
    }
    pos += 1;
  }
  return words;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::printf("usage: %s model.bin tokenizer.model steps prompt_ids [expected_words]\n", argv[0]);
    return 2;
  }
  if (kh_device_count() <= 0) {
    std::printf("SKIP: no HIP device; the reference's LLama2Model links against kernels_interfaces_hip.cpp + libkuiper_hip.so "
                "(build-time check passed)\n");
    return 77;
  }
  const int steps = std::atoi(argv[3]);
  const std::vector<int32_t> prompt = parse_ids(argv[4]);
  const std::vector<int32_t> want =
      (argc > 5 && std::strcmp(argv[5], "-") != 0) ? parse_ids(argv[5]) : std::vector<int32_t>();
  bool quant = false;
  for (int i = 6; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--quant")) {
      quant = true;
    } else if (!std::strcmp(argv[i], "--flavor") && i + 1 < argc) {
      int mode = 0;
      float eps = 1e-5f, theta = 10000.f;
      if (std::sscanf(argv[++i], "%d,%f,%f", &mode, &eps, &theta) != 3) {
        std::printf("bad --flavor (rope_mode,rms_eps,rope_theta)\n");
        return 2;
      }
      kuiper_hip::flavor() = kuiper_hip::Flavor{mode, eps, theta};
    } else {
      std::printf("unknown argument %s\n", argv[i]);
      return 2;
    }
  }
  const kuiper_hip::Flavor mine = kuiper_hip::flavor();
  RefModel model(base::TokenizerType::kEncodeSpe, argv[2], argv[1], /*is_quant_model=*/quant);
  const base::Status st = model.init(base::DeviceType::kDeviceCUDA);
  // init() computed the sin / cos table on the model's stream and the forward above bound that stream to `mine`:
  // from here on the process default belongs to the next model.  Set it
  // to something this model must NOT pick up - a second model with another flavour in the same process.
  kuiper_hip::flavor() = kuiper_hip::Flavor{mine.rope_mode == KH_ROPE_HALF ? KH_ROPE_INTERLEAVED : KH_ROPE_HALF,
                                            mine.rms_eps * 100.f, mine.rope_theta * 3.f};
  if (!st) {
    std::printf("FAIL init: %s\n", st.get_err_msg().c_str());
    return 1;
  }
  // the reference's SpeEncodeLayer (encode.cpp:13-57) on this repo's tokenizer: BOS + pieces, and back
  const std::vector<int32_t> enc = model.encode("hello world");
  if (enc.size() < 2 || model.decode(std::vector<int32_t>(enc.begin() + 1, enc.end())) != "hello world") {
    std::printf("FAIL encode / decode round trip through op::SpeEncodeLayer\n");
    return 1;
  }
  (void)generate(model, prompt, steps < 8 ? steps : 8);  // warm (first-touch, lazy kernel loads)
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  const std::vector<int32_t> words = generate(model, prompt, steps);
  (void)hipDeviceSynchronize();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("words:");
  for (int32_t w : words) std::printf(" %d", w);
  std::printf("\nreference %s%s on the HIP kernels (rope mode %d, eps %g, theta %g): %d steps in %.2f ms = %.1f tokens/s "
              "(host loop of llama3.cpp:147-167, one launch per operator, pos on the host)\n",
              kRefModelName, quant ? " [int8]" : "", (int)mine.rope_mode, (double)mine.rms_eps, (double)mine.rope_theta, steps,
              sec * 1e3, steps / sec);
  if (!want.empty()) {
    if (words != want) {
      size_t i = 0;
      while (i < words.size() && i < want.size() && words[i] == want[i]) ++i;
      std::printf("FAIL words differ from the expected ones at step %zu\n", i);
      return 1;
    }
    std::printf("OK %zu words equal the expected ones\n", words.size());
  }
  return 0;
}
