// kuiper_demo — command-line twin of the reference's demo/main.cpp / demo/main_qwen.cpp on top of
// the C-ABI (include/kuiper_hip.h).  The reference hard-codes tokenizer type, quantisation,
// device and prompt in the source; here they are flags.  The prompt is given as
// token ids (--prompt) or as text (--text) with the tokenizer file the reference takes as its second
// argument (main.cpp:56): a SentencePiece-BPE tokenizer.model (--tokenizer: BOS + encode like
// SpeEncodeLayer, stop at eos_id) or a HuggingFace tokenizer.json (--tokenizer-json: the byte-level BPE
// of the LLAMA3_SUPPORT / QWEN2_SUPPORT builds, encode.cpp:59-183 - BOS for Llama-3 and not for Qwen2,
// the reference's space -> "Ġ" replacement unless --hf-spaces, two stop ids); the decoded text is
// printed like main.cpp:43-45.
//
//   kuiper_demo model.bin [--family llama|qwen2] [--quant] [--rope interleaved|half]
//               [--theta 10000] [--eps 1e-5] [--steps 128] [--prompt 1,263] [--stop 2]
//               [--exec graph|fused|unfused] [--max-seq-len N] [--device 0]
//               [--tokenizer tokenizer.model | --tokenizer-json tokenizer.json [--hf-spaces]] [--text "a"]
//               [--exact-prefill] [--fenced-merge]
//
// --exact-prefill = KH_FLAG_PREFILL_EXACT: the prompt phase bit for bit the reference's one-token-per-pass prompt
// phase (demo/main.cpp:20-22); without it prompts of 17+ tokens run as fp32-MFMA GEMMs (tolerance parity, 8-10 x the
// prompt tokens/s).  --fenced-merge = KH_FLAG_ATTN_MERGE_FENCED.  The self-checks of kh_model_create_* are printed.
// Prints the generated ids and "steps/s" like demo/main.cpp:70-72.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kuiper_hip.h"

static void usage() {
  std::fprintf(stderr,
               "usage: kuiper_demo model.bin [--family llama|qwen2] [--quant] [--rope interleaved|half]\n"
               "       [--theta F] [--eps F] [--steps N] [--prompt id,id,...] [--stop id,id] [--exec graph|fused|unfused]\n"
               "       [--max-seq-len N] [--device D] [--tokenizer tokenizer.model | --tokenizer-json tokenizer.json\n"
               "       [--hf-spaces]] [--text \"...\"] [--exact-prefill] [--fenced-merge]\n");
}

int main(int argc, char** argv) {
  if (argc < 2) {
    usage();
    return 2;
  }
  const char* path = argv[1];
  kh_model_opts o{KH_FAMILY_LLAMA, 0, KH_ROPE_INTERLEAVED, 10000.f, 1e-5f, 0, 0, 0};
  int steps = 128, exec = KH_EXEC_GRAPH;
  std::vector<int32_t> stop;  // is_sentence_ending ids (main.cpp:30): eos / <|eot_id|> / ...
  std::vector<int32_t> prompt{1, 263};  // BOS + "a": the reference demo's prompt (main.cpp:64)
  const char* tok_path = nullptr;
  const char* bpe_path = nullptr;
  int bpe_flags = KH_BPE_REF_SPACES;  // the reference's behaviour (encode.cpp:108-111)
  std::string text;
  bool have_text = false;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) {
        usage();
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--family") o.family = std::string(next()) == "qwen2" ? KH_FAMILY_QWEN2 : KH_FAMILY_LLAMA;
    else if (a == "--quant") o.is_quant = 1;
    else if (a == "--rope") o.rope_mode = std::string(next()) == "half" ? KH_ROPE_HALF : KH_ROPE_INTERLEAVED;
    else if (a == "--theta") o.rope_theta = (float)std::atof(next());
    else if (a == "--eps") o.rms_eps = (float)std::atof(next());
    else if (a == "--steps") steps = std::atoi(next());
    else if (a == "--tokenizer") tok_path = next();
    else if (a == "--tokenizer-json") bpe_path = next();
    else if (a == "--hf-spaces") bpe_flags = 0;
    else if (a == "--exact-prefill") o.flags |= KH_FLAG_PREFILL_EXACT;
    else if (a == "--fenced-merge") o.flags |= KH_FLAG_ATTN_MERGE_FENCED;
    else if (a == "--text") {
      text = next();
      have_text = true;
    }
    else if (a == "--max-seq-len") o.max_seq_len = std::atoi(next());
    else if (a == "--device") o.device = std::atoi(next());
    else if (a == "--exec") {
      std::string e = next();
      exec = e == "unfused" ? KH_EXEC_UNFUSED : (e == "fused" ? KH_EXEC_FUSED : KH_EXEC_GRAPH);
    } else if (a == "--prompt" || a == "--stop") {
      std::vector<int32_t>& dst = a == "--prompt" ? prompt : stop;
      dst.clear();
      std::string s = next();
      size_t p = 0;
      while (p < s.size()) {
        size_t q = s.find(',', p);
        if (q == std::string::npos) q = s.size();
        dst.push_back(std::atoi(s.substr(p, q - p).c_str()));
        p = q + 1;
      }
    } else {
      usage();
      return 2;
    }
  }
  kh_spm* tok = nullptr;
  if (tok_path) {
    const int trc = kh_spm_create_from_file(tok_path, &tok);
    if (trc != KH_OK) {
      std::fprintf(stderr, "tokenizer load failed: %d (%s)\n", trc, kh_error_string(trc));
      return 1;
    }
    if (have_text) {  // model.cpp:158-165: BOS on for the Llama family; encode.cpp:37-41
      int32_t n = 0;
      prompt.assign(text.size() * 4 + 8, 0);
      if (kh_spm_encode(tok, text.data(), (int64_t)text.size(), 1, 0, prompt.data(), (int32_t)prompt.size(),
                        &n) != KH_OK) {
        std::fprintf(stderr, "encode failed\n");
        return 1;
      }
      prompt.resize((size_t)n);
    }
    if (stop.empty()) stop.push_back(kh_spm_eos_id(tok));  // is_sentence_ending, encode.cpp:48-51
  }
  kh_bpe* bpe = nullptr;
  if (bpe_path) {
    const int flavor = o.family == KH_FAMILY_QWEN2 ? KH_BPE_QWEN2 : KH_BPE_LLAMA3;
    const int trc = kh_bpe_create_from_file(bpe_path, flavor, &bpe);
    if (trc != KH_OK) {
      std::fprintf(stderr, "tokenizer.json load failed: %d (%s) - expected a byte-level BPE tokenizer.json "
                           "(Llama-3.x / Qwen2.5); SentencePiece models go to --tokenizer\n",
                   trc, kh_error_string(trc));
      return 1;
    }
    if (have_text) {  // model.cpp:158-165: BOS for Llama, none for Qwen
      int32_t n = 0;
      prompt.assign(text.size() * 2 + 8, 0);
      if (kh_bpe_encode(bpe, text.data(), (int64_t)text.size(), flavor == KH_BPE_LLAMA3, 0, bpe_flags,
                        prompt.data(), (int32_t)prompt.size(), &n) != KH_OK) {
        std::fprintf(stderr, "encode failed\n");
        return 1;
      }
      prompt.resize((size_t)n);
    }
    if (stop.empty()) {  // is_sentence_ending, encode.cpp:130-136
      stop.push_back(kh_bpe_stop_id(bpe, 0));
      stop.push_back(kh_bpe_stop_id(bpe, 1));
    }
  }
  if (have_text && !tok && !bpe) {
    std::fprintf(stderr, "--text needs --tokenizer (SentencePiece tokenizer.model) or --tokenizer-json "
                         "(HuggingFace byte-level BPE tokenizer.json)\n");
    return 2;
  }
  kh_model* m = nullptr;
  int rc = kh_model_create_from_file(path, &o, &m);
  if (rc != KH_OK) {
    std::fprintf(stderr, "The model init failed: %d (%s)\n", rc, kh_error_string(rc));
    return 1;
  }
  kh_config c;
  kh_model_get_config(m, &c);
  std::fprintf(stderr, "dim %d hidden %d layers %d heads %d kv_heads %d vocab %d seq_len %d%s\n", c.dim,
               c.hidden_dim, c.layer_num, c.head_num, c.kv_head_num, c.vocab_size, c.seq_len,
               c.is_quant ? " int8" : "");
  std::fprintf(stderr, "weights: %.2f GB uploaded in %.1f ms (%.1f GB/s)\n", c.weight_bytes / 1e9,
               kh_model_get_load_ms(m), c.weight_bytes / 1e6 / (kh_model_get_load_ms(m) + 1e-9));
  // kh_config: 0 n/a, 1 passed, -1 failed -> fallback in use, 2 (merge) fenced form requested
  std::fprintf(stderr, "self-checks: int8 ring kernels %d, attention split merge %d; prompt phase: %s\n", c.ring_selftest,
               c.attn_merge_selftest, (o.flags & KH_FLAG_PREFILL_EXACT) ? "exact (bit-identical to token-by-token)"
                                                                        : "GEMM for 17+ tokens (fp32 round-off)");
  std::vector<int32_t> words((size_t)steps);
  int32_t n = 0;
  float gpu_ms = 0.f;
  std::printf("Generating...\n");
  const auto t0 = std::chrono::steady_clock::now();  // timer excludes init (main.cpp:66)
  rc = kh_model_generate_until(m, prompt.data(), (int32_t)prompt.size(), steps, exec, stop.data(),
                               (int32_t)stop.size(), words.data(), &n, &gpu_ms);
  const auto t1 = std::chrono::steady_clock::now();
  if (rc != KH_OK) {
    std::fprintf(stderr, "generate failed: %d (%s)\n", rc, kh_error_string(rc));
    kh_model_destroy(m);
    return 1;
  }
  if (o.family == KH_FAMILY_QWEN2 && !prompt.empty()) {
    // demo/main_qwen.cpp:12,18 seeds `next` with the first prompt token and pushes it into `words`
    // before the loop (main.cpp starts from next = -1): the Qwen demo's output begins with it
    words.insert(words.begin(), prompt[0]);
    ++n;
  }
  if (tok) {  // main.cpp:43-45: printf("%s ", model.decode(words))
    std::vector<char> buf((size_t)n * 16 + 16);
    int64_t len = 0;
    if (kh_spm_decode(tok, words.data(), n, buf.data(), (int64_t)buf.size(), &len) == KH_OK)
      std::printf("%.*s \n", (int)len, buf.data());
    kh_spm_destroy(tok);
  }
  if (bpe) {
    std::vector<char> buf((size_t)n * 32 + 16);
    int64_t len = 0;
    if (kh_bpe_decode(bpe, words.data(), n, bpe_flags, buf.data(), (int64_t)buf.size(), &len) == KH_OK)
      std::printf("%.*s \n", (int)len, buf.data());
    kh_bpe_destroy(bpe);
  }
  for (int i = 0; i < n; ++i) std::printf("%d ", words[i]);
  const double dur = std::chrono::duration<double>(t1 - t0).count();
  const int steps_done = n - (o.family == KH_FAMILY_QWEN2 && !prompt.empty() ? 1 : 0);
  std::printf("\nsteps/s:%lf\n", (double)steps_done / dur);
  if (o.family == KH_FAMILY_QWEN2) std::printf("\nsteps:%d\n\nduration:%lf\n", steps_done, dur);  // main_qwen.cpp:73-74
  std::fprintf(stderr, "(device time of the step loop: %.3f ms)\n", gpu_ms);
  kh_model_destroy(m);
  return 0;
}
