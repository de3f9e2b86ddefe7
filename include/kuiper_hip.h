/*
 * kuiper_hip.h — C-ABI of libkuiper_hip.so: the MI355X (gfx950) decode path that drops in
 * behind KuiperLLama's kernel-function-pointer interface
 *     kuiper/source/op/kernels/kernels_interface.h:6-68   (typedefs + get_*_kernel getters)
 * and, one level up, behind model::LLama2Model / model::Qwen2Model
 *     kuiper/include/model/model.h:20-56, kuiper/source/model/llama3.cpp:107-167,733-745.
 *
 * Conventions
 *   - every function returns int: 0 = success, >0 = a hipError_t, <0 = KH_ERR_* below.
 *     Nothing aborts (the reference CHECK/LOG(FATAL)s; a C-ABI must not).
 *   - all tensor pointers are DEVICE pointers unless a parameter is named h_* / host.
 *   - tensors are dense row-major fp32 (int8 for quantised weights), exactly the layouts of
 *     the reference (weights [K rows, M cols]; KV cache [layer, seq_len, kv_dim]).
 *   - `stream` is a hipStream_t passed as void* (the reference passes `void* stream` /
 *     CudaConfig::stream the same way, kuiper/include/base/cuda_config.h:6-13).
 *     Launches are asynchronous on that stream, never allocate, never synchronise, and are
 *     hipGraph-capturable — which is why positions/tokens can be given as DEVICE scalars
 *     (the reference reads `pos` on the host: cuda/rope_kernel.cu:157, op/mha.cpp:33).
 *   - callee never takes ownership of caller memory.
 */
#ifndef KUIPER_HIP_H
#define KUIPER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_VERSION 100 /* 0.1.0 */

enum {
  KH_OK = 0,
  KH_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, size mismatch */
  KH_ERR_UNSUPPORTED = -2, /* e.g. int8 + tied classifier (broken in the reference too) */
  KH_ERR_IO = -3,          /* file open/map failure (reference: error::PathNotValid) */
  KH_ERR_FORMAT = -4,      /* malformed .bin (reference: error::ModelParseError) */
  KH_ERR_NO_DEVICE = -5,
  KH_ERR_RANGE = -6,       /* token / position out of range */
  KH_ERR_INTERNAL = -7     /* a C++ exception inside the library (host allocation, thread creation): caught at the
                              boundary, everything the call had built is released; host out-of-memory is reported as
                              hipErrorOutOfMemory (2) */
};
const char* kh_error_string(int code);
int kh_version(void);
/* number of HIP devices visible; <0 on error */
int kh_device_count(void);

enum { KH_ROPE_INTERLEAVED = 0, KH_ROPE_HALF = 1 };
enum { KH_FAMILY_LLAMA = 0, KH_FAMILY_QWEN2 = 1 };

/* ============================ operator level ==========================================
 * One entry point per reference kernel typedef (kernels_interface.h). */

/* AddKernel (kernels_interface.h:6-7; cuda/add_kernel.cu:14-32): out = in1 + in2 */
int kh_add_f32(const float* in1, const float* in2, float* out, int32_t n, void* stream);

/* MatmulKernel (kernels_interface.h:9-10; cuda/matmul_kernel.cu:89-109,
 * cpu/matmul_kernel.cpp:5-41): y[K] = (W[K,M] . x[M]) * scale.
 * (The reference CUDA path ignores `scale`, the CPU path honours it; honoured here.) */
int kh_matmul_f32(const float* x, const float* w, float* y, int32_t M, int32_t K, float scale,
                  void* stream);

/* MatmulKernelQuant (kernels_interface.h:12-14; cuda/matmul_kernel.cu:56-87,111-134):
 * y[p] = sum_i x[i] * scales[(p*M+i)/group_size] * float(w8[p*M+i]) */
int kh_matmul_q8(const float* x, const int8_t* w8, const float* scales, int32_t group_size,
                 float* y, int32_t M, int32_t K, void* stream);

/* EmbeddingKernel (kernels_interface.h:16-17; cuda/emb_kernel.cu:3-48):
 * out[t,:] = W[tokens[t],:].  tokens is a DEVICE int32 array (for callers that already hold the
 * ids on the device; the reference hands over a HOST tensor and uploads it per call,
 * emb_kernel.cu:25-29 - that form is kh_embedding_f32_host below, which the adapter uses).  Rows
 * whose token is outside [0, vocab) are left untouched. */
int kh_embedding_f32(const int32_t* tokens, int32_t n_tokens, const float* w, float* out,
                     int32_t dim, int32_t vocab, void* stream);

/* The same with the token ids in HOST memory, which is what the reference's EmbeddingKernel
 * receives (the input tensor is a CPU tensor, op/embedding.cpp; emb_kernel.cu:25-29 uploads it per
 * call).  The ids travel in the kernel arguments, 64 per launch: no staging buffer, no copy. */
int kh_embedding_f32_host(const int32_t* h_tokens, int32_t n_tokens, const float* w, float* out,
                          int32_t dim, int32_t vocab, void* stream);

/* SwigluKernel (kernels_interface.h:19-20; cuda/swiglu_kernel.cu:4-47):
 * out = a*sigmoid(a) * b ; out may alias a (llama3.cpp:708). */
int kh_swiglu_f32(const float* a, const float* b, float* out, int32_t n, void* stream);

/* RMSNormKernel (kernels_interface.h:30-31; cuda/rmsnorm_kernel.cu:5-78,
 * cpu/rmsnorm_kernel.cpp:4-33): out = w * (x / sqrt(mean(x^2)+eps)); out may alias x.
 * eps is a compile-time #ifdef in the reference (1e-5 / 1e-6 QWEN2). */
int kh_rmsnorm_f32(const float* x, const float* w, float* out, int32_t n, float eps,
                   void* stream);

/* RoPEKernel (kernels_interface.h:33-36; cuda/rope_kernel.cu:5-36,51-82,104-122,153-170):
 * in-place rotation of q[dim] and k[kv_dim] with cached sin/cos row `pos`.
 * mode selects what the reference picks with LLAMA3_SUPPORT/QWEN2_SUPPORT (half) or
 * neither (interleaved).  Position = *d_pos if d_pos != NULL else pos. */
int kh_rope_f32(int32_t dim, int32_t kv_dim, int32_t head_size, float* q, float* k,
                const int32_t* d_pos, int32_t pos, const float* sin_cache,
                const float* cos_cache, int32_t mode, void* stream);

/* sin_cos_cache_calc_cu (cuda/rope_kernel.cuh:9-10; cpu/rope_kernel.cpp:4-16):
 * cache[pos*hs + d] = sin/cos(float(pos) * (1/pow(theta, d/hs))) for d < head_size. */
int kh_sincos_cache_f32(int32_t head_size, int32_t max_seq_len, float theta, float* sin_cache,
                        float* cos_cache, void* stream);

/* MHAKernel (kernels_interface.h:22-28; cuda/mha_kernel.cu:47-130, cpu/mha_kernel.cpp:5-61).
 * Decode attention for one query token over the contiguous KV cache; softmax probabilities
 * are left in score[head, 0..pos] like the reference.  Position = *d_pos or pos.
 * score may be NULL: probabilities are then not materialised and the launch uses the fused
 * decode path's low-latency kernel (per-lane-group online softmax, one memory round trip). */
int kh_mha_f32(const int32_t* d_pos, int32_t pos, int32_t head_num, int32_t layer_index,
               int32_t seq_len, int32_t kv_dim, int32_t kv_mul, int32_t head_size,
               float* mha_out, const float* q, float* score, const float* kcache,
               const float* vcache, void* stream);

/* Multi-token form of MHAKernel for the prompt phase (no reference counterpart: demo/main.cpp:20-22
 * feeds the prompt one token per forward pass, so mha.cpp:19-38 only ever sees one query).  Causal
 * attention of n_tokens consecutive queries at positions pos0 .. pos0+n_tokens-1 against one layer's
 * contiguous cache, whose rows up to pos0+n_tokens-1 are already written; q and mha_out are
 * row-major [n_tokens][head_num*head_size].  Both contractions (q.K^T, P.V) run on
 * v_mfma_f32_16x16x4_f32 with an online softmax; per token the result equals kh_mha_f32 at
 * pos = pos0 + t within the fp32 tolerance.  head_size 48, 64 or 128, else KH_ERR_UNSUPPORTED. */
int kh_mha_prefill_f32(int32_t pos0, int32_t n_tokens, int32_t head_num, int32_t layer_index,
                       int32_t seq_len, int32_t kv_dim, int32_t kv_mul, int32_t head_size,
                       float* mha_out, const float* q, const float* key_cache,
                       const float* value_cache, void* stream);

/* The decode-path attention kernel with its long-context machinery.  Per-head path: the grid
 * carries ceil(seq_len/256) (<= 16) workgroups per head; positions < 256 use one of them, longer
 * contexts split the timesteps and the last workgroup to finish merges the partial
 * (max, sum, o) triples.  GQA models (kv_mul 2/4/7/8) switch, from pos + 1 >= 4096 on
 * (env KH_ATTN_TLONG; 0 = never), to one workgroup per (kv group, time split) that computes the
 * group's kv_mul heads from ONE pass over the K/V rows.  `workspace` =
 * kh_mha_decode_workspace_bytes(...) bytes, 16-byte aligned, ZEROED once before first use (the
 * kernel re-arms it); may be NULL when that is 0. */
int64_t kh_mha_decode_workspace_bytes(int32_t head_num, int32_t head_size, int32_t seq_len);
int kh_mha_decode_f32(const int32_t* d_pos, int32_t pos, int32_t head_num, int32_t layer_index,
                      int32_t seq_len, int32_t kv_dim, int32_t kv_mul, int32_t head_size,
                      float* mha_out, const float* q, const float* kcache, const float* vcache,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* argmax_kernel_cu (cuda/argmax_kernel.cuh:4; argmax_sampler.cpp:5-13): index of the first
 * maximum.  Device-result form (graph-capturable) and host-result form (synchronises the
 * stream, like the reference's D2H copy at argmax_kernel.cu:84). */
int kh_argmax_f32(const float* logits, int64_t n, int32_t* d_out_index, void* stream);
int kh_argmax_f32_host(const float* logits, int64_t n, int64_t* h_out_index, void* stream);

/* CPU-only helpers of the reference (kernels_interface.h:38-46), provided on device so the
 * op set is closed: softmax in place, x *= scale, out += sum_t scale[t]*value[t*stride..] */
int kh_softmax_f32(float* x, int32_t n, void* stream);
int kh_scale_f32(float scale, float* x, int32_t n, void* stream);
int kh_scale_sum_f32(const float* value, const float* scale, float* out, int32_t pos,
                     int32_t size, int32_t stride, void* stream);

/* ============================ model level ==============================================
 * Replaces model::LLama2Model / Qwen2Model for the decode path: .bin image -> HBM arena,
 * per-token forward, greedy generate loop captured in a hipGraph. */
typedef struct kh_model kh_model;

typedef struct kh_model_opts {
  int32_t family;      /* KH_FAMILY_* : weight layout (llama3.cpp:290-423 / qwen2.cpp) */
  int32_t is_quant;    /* int8 group-quantised image (tools/export.py version 3) */
  int32_t rope_mode;   /* KH_ROPE_* */
  float rope_theta;    /* 10000 / 500000 (LLAMA3) / 1000000 (QWEN2) in the reference */
  float rms_eps;       /* 1e-5 / 1e-6 (QWEN2) */
  int32_t max_seq_len; /* rows of KV cache + sin/cos to allocate; 0 = header seq_len */
  int32_t device;      /* HIP device ordinal (reference: cudaSetDevice(0)) */
  int32_t flags;       /* KH_FLAG_* bits, 0 = defaults */
} kh_model_opts;
/* decode attention at positions that need several time splits: merge the split partials inside the
 * attention launch (ticket + last arriver) instead of in the wo kernel that follows (the default) */
#define KH_FLAG_ATTN_MERGE_IN_LAUNCH 1
/* the in-launch merge of time splits (operator entry points, prompt slices, the GQA group path, and everything
 * under KH_FLAG_ATTN_MERGE_IN_LAUNCH) orders its hand-over with agent-scope release / acquire FENCES instead of the
 * default write-through stores + drained vmcnt + sc1 loads (csrc/kh_attn.h: attn_publish_barrier).  Same
 * results; 1-2 us slower per layer at positions that need several splits.  Hook KH_ATTN_FENCED=1 does the same
 * for models created while it is set and for kh_mha_decode_f32 / kh_mha_f32. */
#define KH_FLAG_ATTN_MERGE_FENCED 2
/* The prompt phase of kh_model_generate*.  The reference feeds the prompt one token per forward pass
 * (demo/main.cpp:20-22), so its prompt phase is, bit for bit, its decode path.  Two batched prompt paths exist here:
 *  - default: prompts with >= 16 fed-only tokens run as fp32-MFMA GEMMs (kh_model_prefill_gemm, 45-60 k prompt
 *    tokens/s on Llama-3.2-1B): K/V rows equal to fp32 round-off, i.e. the greedy continuation can differ from the
 *    reference's at near-ties (kh_model_first_sample reports the margin of the first sampled step);
 *  - KH_FLAG_PREFILL_EXACT: every prompt runs on the B-token VALU kernels (kh_model_prefill, 5.9 k prompt
 *    tokens/s): K/V rows and every later logit are bit-identical to the token-by-token prompt phase - what a
 *    drop-in user who needs token identity with the reference's own prompt phase sets.
 * kh_first_sample.prefill_mode says which one the last generate took.  The test hook KH_PREFILL overrides both. */
#define KH_FLAG_PREFILL_EXACT 4

typedef struct kh_config {
  int32_t dim, hidden_dim, layer_num, head_num, kv_head_num, vocab_size, seq_len;
  int32_t kv_dim, kv_mul, head_size, is_shared_weight, is_quant, group_size;
  int32_t family, rope_mode, cache_len;
  float rope_theta, rms_eps;
  int64_t weight_bytes; /* bytes of the weight arena resident in HBM */
  int32_t launches_per_token; /* 5 per layer + classifier + sampler */
  /* Self-checks run once at the end of kh_model_create_* (csrc/kh_model_selftest.hip), about a millisecond:
   *   ring_selftest        the int8 LDS-DMA ring kernels against the register-tile kernels of the same launches on
   *                        this model's own weights: 0 = no ring kernel planned (fp32, geometry, KH_RING=0) or skipped,
   *                        1 = outputs identical, -1 = mismatch: this model runs the register-tile kernels;
   *   attn_merge_selftest  the fence-free in-launch merge of the decode-attention time splits against the fenced
   *                        form: 0 = not applicable (no time splits in this cache) or skipped, 1 = identical,
   *                        -1 = mismatch: this model uses the fenced form, 2 = the fenced form was requested
   *                        (KH_FLAG_ATTN_MERGE_FENCED / KH_ATTN_FENCED=1).
   * Hook KH_SELFTEST=0 skips both, KH_SELFTEST_FAIL="ring,attn" injects a failure (tests). */
  int32_t ring_selftest, attn_merge_selftest;
} kh_config;

/* Model::read_model_file + init (model.cpp:41-123, llama3.cpp:107-145) */
int kh_model_create_from_file(const char* path, const kh_model_opts* opts, kh_model** out);
/* same, from the bytes of a .bin file already in host memory (header included) */
int kh_model_create_from_host_image(const void* h_image, size_t nbytes,
                                    const kh_model_opts* opts, kh_model** out);
/* weights already resident in HBM: h_header = the 7 (fp32) or 8 (int8) header ints,
 * d_weight_data = the file bytes AFTER the header, 16-byte aligned, not owned
 * (the reference's "external buffer", layer.cpp:190-202). */
int kh_model_create_from_device_weights(const int32_t* h_header, const void* d_weight_data,
                                        size_t weight_nbytes, const kh_model_opts* opts,
                                        kh_model** out);
void kh_model_destroy(kh_model* m);
int kh_model_get_config(const kh_model* m, kh_config* out);
void* kh_model_stream(kh_model* m); /* the model's hipStream_t */
/* milliseconds the host-image -> HBM upload took (0 for device-resident weights) */
float kh_model_get_load_ms(const kh_model* m);

enum {
  KH_EXEC_GRAPH = 0,   /* fused kernels, whole step replayed as one hipGraph */
  KH_EXEC_FUSED = 1,   /* fused kernels, eager launches */
  KH_EXEC_UNFUSED = 2  /* one launch per reference kernel, in the reference's order
                          (llama3.cpp:147-167): the literal drop-in sequence */
};

/* Model::predict (llama3.cpp:642-650): embedding of `token` -> forward at `pos` ->
 * argmax unless is_prompt.  *h_next receives the token (or -1 when is_prompt).
 * Synchronises the stream. exec: KH_EXEC_FUSED or KH_EXEC_UNFUSED. */
int kh_model_predict(kh_model* m, int32_t token, int32_t pos, int32_t is_prompt, int32_t exec,
                     int32_t* h_next);
/* copy the last logits (kForwardOutput) to host */
int kh_model_get_logits(kh_model* m, float* h_logits);
/* device pointers of the KV cache [layer, cache_len, kv_dim] (tests) */
int kh_model_get_kv(kh_model* m, float** d_kcache, float** d_vcache);
/* bytes of the KV cache: *reserved = the address range of [layer, cache_len, kv_dim] floats x 2 (the reference's
 * up-front allocation, llama3.cpp:469-472), *committed = HBM actually backing it now.  The range is reserved at
 * creation and memory is mapped in 8-MiB chunks as generate / predict / prefill / kh_model_write_kv first reach
 * rows (hook KH_KV_VMM=0: one plain allocation, committed == reserved).  kh_model_get_kv commits everything.
 * kh_model_destroy releases the memory but keeps the two address ranges in a process-wide list for the next model
 * with caches of the same size (hipMemAddressFree is never called: it crashes inside the runtime after 1000-2000
 * create / destroy cycles, profiles/r6_vmm_destroy_crash.txt). */
int kh_model_kv_bytes(const kh_model* m, int64_t* reserved, int64_t* committed);
/* copy rows [row0, row0+nrows) of one layer's K and V cache to host (tests) */
int kh_model_read_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows, float* h_k,
                     float* h_v);
/* the inverse: overwrite rows [row0, row0+nrows) of one layer's K and V cache from host memory or
 * from memory of the model's device (restoring a saved context; tests and the long-context probes
 * of bench.py place rows at deep positions without decoding up to them).  Rows hold what the
 * forward pass stores: RoPE-rotated keys, raw values. */
int kh_model_write_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows, const float* h_k,
                      const float* h_v);

/* demo/main.cpp:5-47 generate(): prompt fed one token per step without sampling, then
 * greedy decode, `total_steps` forward passes in total; h_words receives the reference's
 * `words` vector.  No stop-token check (see kh_model_generate_until).  *h_elapsed_ms = wall
 * time of the step loop measured with HIP events on the model stream.
 * A generate starts a NEW sequence at position 0 and owns the cache rows [0, max(total_steps, 8)): with
 * KH_EXEC_GRAPH, the first call after model creation (or after a longer run than any before grew the step
 * buffers) launches each freshly captured step graph once from position 0 before the timed loop - rows and words
 * 0 .. 7 - so that no later run pays a graph's first launch; rows beyond that range are never touched. */
int kh_model_generate(kh_model* m, const int32_t* h_prompt, int32_t n_prompt,
                      int32_t total_steps, int32_t exec, int32_t* h_words, int32_t* n_words,
                      float* h_elapsed_ms);
/* The same loop with the reference's stop check (demo/main.cpp:30-32, Model::is_sentence_ending
 * model.cpp:211-214): generation ends at the first SAMPLED token that is in h_stop[0..n_stop)
 * (SentencePiece: eos_id, encode.cpp:48-51; BPE: two stop tokens, encode.cpp:133-139); that token
 * is not appended to `words`, *n_words = the reference's return value min(pos, total_steps).
 * Graph mode checks chunk k's words from pinned memory while chunk k+1 is already queued, so
 * there is no per-token host round trip; at most 2*8 steps run past the stop and are discarded. */
int kh_model_generate_until(kh_model* m, const int32_t* h_prompt, int32_t n_prompt,
                            int32_t total_steps, int32_t exec, const int32_t* h_stop,
                            int32_t n_stop, int32_t* h_words, int32_t* n_words,
                            float* h_elapsed_ms);

/* Prompt prefill (extends the reference, which feeds prompt tokens one forward pass at a time,
 * demo/main.cpp:20-22): forward of tokens[0..n) at positions pos0.., 8 (fp32) or 4 (int8)
 * tokens per pass over the weights, no logits.  Leaves the K/V cache rows pos0..pos0+n-1 BIT-IDENTICAL to n calls of
 * kh_model_predict(.., is_prompt = 1, KH_EXEC_FUSED).  kh_model_generate* use it for the
 * fed-only part of prompts of 3+ tokens (env KH_PREFILL=0 disables).  KH_ERR_UNSUPPORTED for
 * geometries outside the mirrored kernels (head_size <= 32, dim > 4096). */
int kh_model_prefill(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0);

/* The same contract with the contractions as fp32-MFMA GEMMs (csrc/kh_gemm.h): up to 512 prompt
 * tokens share one pass over the weights (a longer prompt is cut into 512-token passes plus a
 * remainder; env KH_PG_CHUNK = 16..512 sets another pass size) (v_mfma_f32_16x16x4_f32, exact fp32 arithmetic, tokens on
 * the MFMA N dimension; int8 weights dequantised per element in registers).  The K/V rows agree
 * with the token-by-token path to fp32 round-off (different summation order), not bit for bit.
 * kh_model_generate* use it for prompts with >= 16 fed-only tokens unless the model was created with
 * KH_FLAG_PREFILL_EXACT - so for such prompts the greedy tokens carry this tolerance too and can differ from the
 * token-by-token prompt phase at near-ties;
 * hook KH_PREFILL = 0 | token | gemv | gemm overrides (gemv = the bit-identical path; any other
 * value makes generate return KH_ERR_INVALID_ARG).  KH_ERR_UNSUPPORTED: head_size <= 32, dim/hidden not a multiple of 16 (fp32) /
 * 64 (int8), int8 group size != 64. */
int kh_model_prefill_gemm(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0);

/* Near-tie report for prompts that went through a prefill.  The token-by-token prompt phase and kh_model_prefill
 * leave bit-identical K/V rows; kh_model_prefill_gemm (the default of kh_model_generate* from 16 fed-only tokens on)
 * leaves them within fp32 round-off, so the greedy continuation is the token-by-token one unless two logits of a
 * sampled step lie closer together than that round-off.  After every kh_model_generate* call whose prompt phase ran
 * as a prefill, the logits of the FIRST sampled step (the one the whole prompt feeds; later steps inherit its choice)
 * are kept on the device; this call returns their two largest entries.  top1_id is the token that step sampled;
 * top1 - top2 is the margin to compare with the tolerance the caller cares about (the parity tests use 4e-5 for
 * fp32, 1e-4 for int8 weights: tests/test_model_gpu.py).  prefill_mode: 1 = kh_model_prefill (bit-identical rows),
 * 2 = kh_model_prefill_gemm.  KH_ERR_UNSUPPORTED when the last generate had no prefill phase (prompts of fewer than
 * 3 tokens, KH_PREFILL=0, unsupported geometry): its tokens are the token-by-token ones by construction. */
typedef struct kh_first_sample {
  int32_t pos;           /* position of the first sampled step = n_prompt - 1 */
  int32_t prefill_mode;  /* 1 = B-token VALU prefill (bit-identical), 2 = MFMA GEMM prefill (fp32 tolerance) */
  int32_t top1_id, top2_id;
  float top1, top2;      /* the two largest logits of that step; ties -> lowest index first, like the sampler */
} kh_first_sample;
int kh_model_first_sample(kh_model* m, kh_first_sample* out);

/* Launch plans, host-only (no device is touched; for tools and the CPU test-suite).
 * kh_plan_decode_shapes: {split, u, grid, wg} of the five GEMV kernels of a decode step (qkv, wo, ffn13, w2,
 * cls) for a geometry - what kh_model_create_* configures (env KH_SHAPE_* overrides included).
 * kh_plan_prefill_shape: {R, NT, ks, token slices, solo, kz, workgroups} of one GEMM of a T-token prefill
 * pass (epi 0 = QKV, 1 = residual GEMM wo / w2, 2 = SwiGLU pair; csrc/kh_model_prefill.hip::pg_shape). */
int kh_plan_decode_shapes(int32_t dim, int32_t hidden_dim, int32_t kv_dim, int32_t vocab_size,
                          int32_t is_quant, int32_t* out20);
/* kh_plan_decode_ring: out4 = {ffn13 ring slots per wave, ffn13 workgroups, cls ring slots, cls workgroups} - which
 * int8 GEMVs of a decode step run on the LDS-DMA ring kernels (csrc/kh_fused_ring.h; 0 slots = the register-tile
 * kernel of kh_plan_decode_shapes) and with how many 256-thread workgroups.  Hook KH_RING=0 turns them off. */
int kh_plan_decode_ring(int32_t dim, int32_t hidden_dim, int32_t vocab_size, int32_t is_quant, int32_t group_size,
                        int32_t* out4);
/* kh_plan_attention: the decode-attention geometry of kh_mha_decode_f32 / the fused step for a cache of seq_len rows:
 * out8 = {time splits per head, splits per KV group (0: no group path), workspace slot stride, first pos + 1 of the
 * group path, path taken at `pos` (0 per-head, 1 group), active splits at `pos`, timesteps per split at `pos`,
 * workgroups that own timesteps at `pos`} (KH_ATTN_TLONG honoured). */
int kh_plan_attention(int32_t head_num, int32_t kv_mul, int32_t head_size, int32_t seq_len, int32_t pos,
                      int32_t* out8);
int kh_plan_prefill_shape(int32_t epi, int32_t T, int32_t rows, int32_t K, int32_t is_quant,
                          int32_t r2_ok, int32_t* out7);

/* Tuning / test hooks.  Every hook the library honours (KH_SHAPE_<QKV|WO|FFN|W2|CLS>, KH_RING, KH_ATTN_WG, KH_ATTN_FENCED,
 * KH_ATTN_TLONG, KH_ATTN_DEFER (0 = never merge time splits in the wo kernel), KH_ATTN_DEFER_MAX (active splits up to
 * which it does), KH_PREFILL, KH_PG_<CHUNK|SHAPE_*|SOLO|KZ|ATTN|ATTN_QT|ROPE_FUSE|DEBUG>,
 * KH_SHAPE_DEBUG) lives in ONE process-wide key -> value table, seeded once from the KH_* variables of
 * the environment when the library is first used and changed afterwards only through kh_debug_set
 * (value NULL = unset).  No launch path reads the environment.  Hooks that shape a model (KH_SHAPE_*,
 * KH_ATTN_*) are read by kh_model_create_*; the others by the call they affect.  Keys must start with "KH_". */
int kh_debug_set(const char* key, const char* value);
const char* kh_debug_get(const char* key); /* NULL when unset */
int64_t kh_debug_list(char* buf, int64_t cap); /* '\n'-separated names; returns bytes needed */

/* Duration (ms, HIP events on the model stream) of the prompt phase alone for n fed-only tokens:
 * KH_PREFILL_TOKEN = the reference's prompt phase, one forward pass per token (demo/main.cpp:20-22)
 * replayed from the decode hipGraph; KH_PREFILL_GEMV = kh_model_prefill's bit-identical B-token
 * kernels; KH_PREFILL_GEMM = the fp32-MFMA GEMM prefill.  Leaves the K/V rows of those positions. */
enum { KH_PREFILL_TOKEN = 0, KH_PREFILL_GEMV = 1, KH_PREFILL_GEMM = 2 };
int kh_model_time_prefill(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0,
                          int32_t mode, float* h_ms);

/* Average duration of ONE kernel class launched back to back (no event between launches, so
 * no event overhead in the figure): the kernel is captured for layers 0..L-1 `reps` times into
 * a hipGraph that is replayed between two HIP events on the model stream (cls / sample: `reps`
 * launches; a graph because a 3-4 us kernel outruns eager host enqueues); consecutive
 * launches read different layers' weights, so nothing is served from cache.  This is the
 * number bench.py's roofline uses and the one rocprofv3's per-kernel average must agree with.
 * Destroys the activation state and KV row `pos` (a later generate/predict resets both). */
int kh_model_profile_kernel(kh_model* m, int32_t kclass, int32_t pos, int32_t reps,
                            float* h_avg_us);

/* Latency of ONE decode step at position `pos` (per-token latency at pos 0/64/127):
 * the 1-step hipGraph replayed `reps` times with the device state reset to `pos` before each,
 * HIP events around each replay; h_us[reps] in microseconds.  KV rows below pos must exist and
 * a graph-mode generate of more than `pos` steps must have run on this model. */
int kh_model_time_step(kh_model* m, int32_t pos, int32_t reps, float* h_us);

/* Per-kernel-class timing of the fused step with HIP events (eager launches, one event
 * between every kernel).  n_steps decode steps starting at position start_pos (KV rows
 * below start_pos must already exist).  Writes avg microseconds per launch for each
 * class into h_avg_us[KH_NUM_KCLASS] and launches-per-step into h_count. */
enum {
  KH_K_QKV = 0, KH_K_ATTN = 1, KH_K_WO = 2, KH_K_FFN13 = 3, KH_K_W2 = 4, KH_K_CLS = 5,
  KH_K_SAMPLE = 6, KH_NUM_KCLASS = 7
};
int kh_model_profile_step(kh_model* m, int32_t start_pos, int32_t n_steps, float* h_avg_us,
                          int32_t* h_count);
const char* kh_kclass_name(int kclass);

/* ---- Tokenizer: SentencePiece BPE (host only) -------------------------------------------------
 * Replaces op::SpeEncodeLayer (kuiper/source/op/encode.cpp:10-57), the reference's wrapper over
 * the external sentencepiece library: Load -> create, Encode (+bos/eos like encode.cpp:37-44),
 * Decode, eos_id (is_sentence_ending, encode.cpp:48-51), GetPieceSize.  BPE model files with the
 * identity character map (Llama-2's tokenizer.model); anything else -> KH_ERR_UNSUPPORTED.
 * encode/decode return KH_ERR_RANGE when the output does not fit and store the needed size. */
typedef struct kh_spm kh_spm;
int kh_spm_create_from_file(const char* tokenizer_model_path, kh_spm** out);
int kh_spm_create_from_memory(const void* model_proto, int64_t nbytes, kh_spm** out);
void kh_spm_destroy(kh_spm* t);
int32_t kh_spm_vocab_size(const kh_spm* t);
int32_t kh_spm_bos_id(const kh_spm* t);
int32_t kh_spm_eos_id(const kh_spm* t);
int32_t kh_spm_unk_id(const kh_spm* t);
int kh_spm_encode(const kh_spm* t, const char* utf8, int64_t len, int32_t add_bos, int32_t add_eos,
                  int32_t* out_ids, int32_t cap, int32_t* n_ids);
int kh_spm_decode(const kh_spm* t, const int32_t* ids, int32_t n, char* out_utf8, int64_t cap,
                  int64_t* out_len);

/* ---- Tokenizer: byte-level BPE over a HuggingFace tokenizer.json (host only) ---------------------
 * Replaces op::BpeEncodeLayer (Llama-3.x) and op::QwenEncodeLayer (Qwen2.5)
 * (kuiper/source/op/encode.cpp:59-183) together with what they link: nlohmann::json, the vendored
 * tiktoken.h (kuiper/include/base/tiktoken.h:17-268), RE2 (the pre-split pattern PAT_STR,
 * encode.cpp:59-60), abseil and the vendored Unicode tables.  `flavor` selects which added_tokens
 * are BOS / EOS / second stop id (encode.cpp:97-103: <|begin_of_text|>, <|end_of_text|>, <|eot_id|>;
 * :170-176: <|im_start|>, <|im_end|>, <|endoftext|>); a name the file lacks is reported as -1.
 * KH_BPE_REF_SPACES reproduces the reference's " " -> "Ġ" replacement before encoding and its
 * inverse after decoding (encode.cpp:108-111, 124-126); without it the text is encoded as it is,
 * which is what HF `tokenizers` produces for the same pattern.  Model::encode adds BOS for Llama and
 * not for Qwen (model.cpp:158-165): that is the caller's add_bos.  Return codes as kh_spm_*. */
typedef struct kh_bpe kh_bpe;
enum { KH_BPE_LLAMA3 = 0, KH_BPE_QWEN2 = 1 };
enum { KH_BPE_REF_SPACES = 1 };
int kh_bpe_create_from_file(const char* tokenizer_json_path, int32_t flavor, kh_bpe** out);
int kh_bpe_create_from_memory(const void* tokenizer_json, int64_t nbytes, int32_t flavor, kh_bpe** out);
void kh_bpe_destroy(kh_bpe* t);
int32_t kh_bpe_vocab_size(const kh_bpe* t); /* |model.vocab| + |added_tokens| (encode.cpp:105) */
int32_t kh_bpe_bos_id(const kh_bpe* t);
int32_t kh_bpe_eos_id(const kh_bpe* t);
int32_t kh_bpe_stop_id(const kh_bpe* t, int32_t which); /* 0, 1: is_sentence_ending (encode.cpp:130-136) */
int kh_bpe_encode(const kh_bpe* t, const char* utf8, int64_t len, int32_t add_bos, int32_t add_eos,
                  int32_t flags, int32_t* out_ids, int32_t cap, int32_t* n_ids);
int kh_bpe_decode(const kh_bpe* t, const int32_t* ids, int32_t n, int32_t flags, char* out_utf8,
                  int64_t cap, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* KUIPER_HIP_H */
