/*
 * kuiper_oracle.h — CPU restatement of KuiperLLama's CPU (Armadillo/OpenBLAS) decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (kuiperllama_amd/, include/) may
 * include, link or call this.  Allowed callers: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py.
 *
 * Every function names the reference file:line it restates (paths relative to the
 * reference repo root, zjhellofss/KuiperLLama @ 2025-02-19).  This is a *port* ("kind":
 * "port" in bench.py), pinned by (a) the reference tests' integer-exact golden vectors,
 * (b) logits produced by the reference's own Python model (tools/model.py, tools/export.py)
 * — see tests/golden/make_golden.py — and, since round 5, (c) the reference's OWN C++ CPU
 * backend: its ten CPU kernel files (cpu/<op>_kernel.cpp), operator and model classes compiled where they lie
 * (oracle/Makefile `ref_cpu` -> oracle/_ref/ref_cpu_model*; Armadillo, absent here, is
 * answered by tests/cpp/ref_stubs/armadillo over numpy's OpenBLAS): this oracle and that
 * backend agree in every logit of the goldens to 2.5e-7 (tests/test_ref_cpu_backend.py).
 * That backend, not this file, is bench.py's timed baseline ("kind": "reference") and the
 * end-to-end checker of the full-size token parity tests; this file stays the op-level
 * checker and the fp64-accumulate gold.  The int8 path has NO CPU implementation in the reference
 * (kernels_interfaces.cpp:54-61); ko_matmul_q8 restates the CUDA kernel
 * (cuda/matmul_kernel.cu:56-87): parity for int8 is "unpinned" beyond the exporter
 * (tools/export.py:49-73) and the torch dequantised forward.
 */
#ifndef KUIPER_ORACLE_H
#define KUIPER_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RoPE flavour: the reference selects it with #ifdef (cpu/rope_kernel.cpp:3,43,83). */
enum { KO_ROPE_INTERLEAVED = 0, KO_ROPE_HALF = 1 };
/* accumulation mode for dot products */
enum { KO_ACC_F32 = 0, KO_ACC_F64 = 1 };
/* model family: selects the weight layout (llama3.cpp:290-423 vs qwen2.cpp:290-426) */
enum { KO_FAMILY_LLAMA = 0, KO_FAMILY_QWEN2 = 1 };

/* ---------------- op level (one per reference CPU kernel) ---------------- */
/* timing only: route fp32 matmuls through a cblas_sgemv (ILP64) entry point; NULL = off */
void ko_set_sgemv(void* cblas_sgemv64_fn);
void ko_set_threads(int n);
int ko_get_threads(void);

/* cpu/matmul_kernel.cpp:5-41  y[K] = (W[K,M] . x[M]) * scale */
void ko_matmul_f32(const float* x, const float* w, float* y, int M, int K, float scale, int acc);
/* cuda/matmul_kernel.cu:56-87 y[p] = sum_i x[i]*scales[(p*M+i)/group]*float(w8[p*M+i]) */
void ko_matmul_q8(const float* x, const int8_t* w, const float* scales, int group, float* y,
                  int M, int K, int acc);
/* cpu/rmsnorm_kernel.cpp:4-33 */
void ko_rmsnorm_f32(const float* x, const float* w, float* out, int n, float eps);
/* cpu/rope_kernel.cpp:4-16 / 44-56 / 84-96 */
void ko_sincos_cache(int head_size, int max_seq_len, float theta, float* sin_cache,
                     float* cos_cache);
/* cpu/rope_kernel.cpp:18-42 / 58-82 (half) and 98-121 (interleaved) */
void ko_rope_f32(int dim, int kv_dim, int head_size, float* q, float* k, int pos,
                 const float* sin_cache, const float* cos_cache, int mode);
/* cpu/softmax_kernel.cpp:4-15 */
void ko_softmax_f32(float* x, int n);
/* cpu/scale_sum_kernel.cpp:5-22  out += sum_{i<=pos} scale[i]*value[i*stride : +size] */
void ko_scale_sum_f32(const float* value, const float* scale, float* out, int pos, int size,
                      int stride);
/* cpu/scale_kernel.cpp:3-9 */
void ko_scale_f32(float scale, float* x, int n);
/* cpu/mha_kernel.cpp:5-61 */
void ko_mha_f32(int pos, int head_num, int layer_index, int seq_len, int kv_dim, int kv_mul,
                int head_size, float* mha_out, const float* q, float* score,
                const float* kcache, const float* vcache, int acc);
/* cpu/swiglu_kernel.cpp:3-23 (out may alias a) */
void ko_swiglu_f32(const float* a, const float* b, float* out, int n);
/* cpu/add_kernel.cpp:5-19 */
void ko_add_f32(const float* a, const float* b, float* out, int n);
/* cpu/emb_kernel.cpp:4-29 ; returns 0 ok, -1 if a token is out of range */
int ko_embedding_f32(const int32_t* tokens, int n_tokens, const float* w, float* out, int dim,
                     int vocab);
/* argmax_sampler.cpp:7 (std::max_element: first occurrence of the maximum) */
size_t ko_argmax_f32(const float* logits, size_t n);

/* tools/export.py:49-73 quantize_q80 restated (round-half-even like torch.round) */
void ko_quantize_q80(const float* w, size_t n, int group, int8_t* q, float* scales);

/* ---------------- model level ---------------- */
typedef struct ko_config {
  int32_t dim, hidden_dim, layer_num, head_num, kv_head_num, vocab_size, seq_len; /* header */
  int32_t kv_dim, kv_mul, head_size;
  int32_t is_shared_weight; /* header vocab_size > 0 (model.cpp:137-141) */
  int32_t is_quant, group_size;
  int32_t family, rope_mode;
  float rope_theta, rms_eps;
  int32_t cache_len; /* rows of KV cache / sin-cos actually allocated (<= seq_len) */
} ko_config;

typedef struct ko_model ko_model;

/* image = the bytes of a reference .bin file (tools/export.py legacy_export /
 * legacy_export_quant / export_qwen2.py), NOT copied: must outlive the model.
 * cache_len <= 0 -> seq_len.  Returns NULL on malformed image (size mismatch). */
ko_model* ko_model_create(const void* image, size_t nbytes, int family, int is_quant,
                          int rope_mode, float rope_theta, float rms_eps, int cache_len);
void ko_model_destroy(ko_model* m);
const ko_config* ko_model_config(const ko_model* m);
/* total bytes the image must have for this header (for tests) */
size_t ko_model_expected_bytes(const ko_model* m);
/* one token: embedding + LLama2Model::forward (llama3.cpp:147-167) ; logits left in buffer */
int ko_model_forward(ko_model* m, int32_t token, int32_t pos, int acc);
const float* ko_model_logits(const ko_model* m);
float* ko_model_kcache(ko_model* m);
float* ko_model_vcache(ko_model* m);
/* demo/main.cpp:5-47 generate(): feeds prompt one token at a time (no sampling while
 * pos < n_prompt-1), then greedy.  No stop-token check (tokeniser out of scope).
 * Writes the `words` sequence (length == returned count, <= total_steps) to out.
 * family QWEN2 follows demo/main_qwen.cpp:12-18 (words starts with tokens[0]) only in the
 * *returned list*; the forward sequence is identical. */
int ko_model_generate(ko_model* m, const int32_t* prompt, int n_prompt, int total_steps,
                      int32_t* out_words, int acc);
/* the same loop with the stop check of demo/main.cpp:30-32 (the stop token is not appended) */
int ko_model_generate_until(ko_model* m, const int32_t* prompt, int n_prompt, int total_steps,
                            const int32_t* stop, int n_stop, int32_t* out_words, int acc);

#ifdef __cplusplus
}
#endif
#endif
