// test_ref_cpu_model.cpp — the reference's OWN CPU backend, end to end: model::LLama2Model / Qwen2Model
// (-DKH_REF_MODEL_QWEN2) with init(kDeviceCPU) over the reference's op::*Layer classes, its CPU kernel getters
// (tests/cpp/kernels_interfaces_cpu.cpp) and its ten CPU kernels (kuiper/source/op/kernels/cpu/*.cpp compiled where
// they lie; Armadillo answered by tests/cpp/ref_stubs/armadillo over numpy's OpenBLAS), running the loop of
// demo/main.cpp:5-47.  No HIP kernel is involved: this binary is the CHECKER the HIP path is compared with
// (tests/test_ref_cpu_backend.py) and bench.py's timed CPU baseline (cpu_baseline.kind = "reference").  The RoPE
// flavour / RMS epsilon are the reference's compile-time switches: three builds (oracle/Makefile `ref_cpu`):
// ref_cpu_model (neither switch: interleaved RoPE, theta 1e4, eps 1e-5), ref_cpu_model_llama3 (-DLLAMA3_SUPPORT on
// the two kernels that read it), ref_cpu_model_qwen2 (-DQWEN2_SUPPORT, Qwen2Model).
//
// usage: ref_cpu_model <model.bin> <tokenizer.model> <steps> <prompt ids, comma separated> [budget_seconds [logits.f32]]
// logits.f32: the model's kForwardOutput after every step (float32 [steps][vocab]) - the reference CPU backend's own logits
// prints "words: ..." (possibly fewer than `steps` when the wall-clock budget runs out; at least the prompt + 2 steps),
// the BLAS in use and "reference CPU backend: N steps in T ms = R tokens/s".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <armadillo>

#include "kernels_interface.h"
#ifdef KH_REF_MODEL_QWEN2
#include "model/qwen2.h"
using RefModel = model::Qwen2Model;
static const char* const kRefModelName = "Qwen2Model";
#else
#include "model/llama3.h"
using RefModel = model::LLama2Model;
static const char* const kRefModelName = "LLama2Model";
#endif

namespace kernel {
// referenced by the CUDA branches of LLama2Model::init / ArgmaxSampler::sample only
void sin_cos_cache_calc_cu(int, int, const tensor::Tensor&, const tensor::Tensor&, cudaStream_t) {
  std::fprintf(stderr, "the CUDA backend is not linked into this binary\n");
  std::abort();
}
size_t argmax_kernel_cu(const float*, size_t, void*) {
  std::fprintf(stderr, "the CUDA backend is not linked into this binary\n");
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

int main(int argc, char** argv) {
  if (argc < 5) {
    std::printf("usage: %s model.bin tokenizer.model steps prompt_ids [budget_seconds]\n", argv[0]);
    return 2;
  }
  const int total_steps = std::atoi(argv[3]);
  const std::vector<int32_t> tokens = parse_ids(argv[4]);
  const double budget = argc > 5 ? std::atof(argv[5]) : 1e30;
  FILE* dump = argc > 6 ? std::fopen(argv[6], "wb") : nullptr;
  RefModel model(base::TokenizerType::kEncodeSpe, argv[2], argv[1], /*is_quant_model=*/false);
  const base::Status st = model.init(base::DeviceType::kDeviceCPU);
  if (!st) {
    std::printf("FAIL init: %s\n", st.get_err_msg().c_str());
    return 1;
  }
  // demo/main.cpp:5-47 with the prompt given as token ids
  const int32_t prompt_len = (int32_t)tokens.size();
  int32_t pos = 0, next = -1;
  bool is_prompt = true;
  const auto& prompt_embedding = model.embedding(tokens);
  tensor::Tensor pos_tensor = model.get_buffer(model::ModelBufferType::kInputPos);
  std::vector<int32_t> words;
  const auto t0 = std::chrono::steady_clock::now();
  double sec = 0;
  while (pos < total_steps) {
    pos_tensor.index<int32_t>(0) = pos;
    if (pos < prompt_len - 1) {
      tensor::Tensor input = model.fill_input(pos_tensor, prompt_embedding, is_prompt);
      model.predict(input, pos_tensor, is_prompt, next);
    } else {
      is_prompt = false;
      std::vector<int32_t> cur{pos == prompt_len - 1 ? tokens[(size_t)pos] : next};
      const auto& token_embedding = model.embedding(cur);
      tensor::Tensor input = model.fill_input(pos_tensor, token_embedding, is_prompt);
      model.predict(input, pos_tensor, is_prompt, next);
    }
    if (dump) {
      const tensor::Tensor lg = model.get_buffer(model::ModelBufferType::kForwardOutput);
      std::fwrite(lg.ptr<float>(), sizeof(float), lg.size(), dump);
    }
    if (is_prompt) next = tokens.at((size_t)pos + 1);
    words.push_back(next);
    pos += 1;
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sec > budget && pos >= prompt_len + 1) break;
  }
  if (dump) std::fclose(dump);
  std::printf("words:");
  for (int32_t w : words) std::printf(" %d", w);
  std::printf("\nblas: %s\n", arma::kh_blas_name());
  std::printf("reference CPU backend (%s, kDeviceCPU): %d steps in %.2f ms = %.3f tokens/s\n", kRefModelName, pos,
              sec * 1e3, pos / sec);
  return 0;
}
