// kh_model_load.hip — model level of the C-ABI, loader side: .bin image -> HBM arena, weight table,
// activation / cache buffers, create / destroy, cache and logits I/O.  Replaces
//   model::Model::read_model_file / generate_model_infos   kuiper/source/model/model.cpp:41-151
//   LLama2Model::create_param_layers / _quant_layers       kuiper/source/model/llama3.cpp:184-423
//   Qwen2Model::create_param_layers (q/k/v bias)           kuiper/source/model/qwen2.cpp:290-426
//   LLama2Model::init_mem                                  llama3.cpp:425-500
// gfx950 only.
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <chrono>
#include <new>
#include <thread>
#include <vector>

#include "kh_model_internal.h"

using namespace khm;

namespace {

// KH_LOAD_DEBUG=1: wall-clock of the phases of a model creation on stderr
struct PhaseClock {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  const bool on = dbg("KH_LOAD_DEBUG") != nullptr;
  void lap(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[kh load] %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

// Host image (typically the mmap of a .bin file: pageable, possibly not yet paged in) -> HBM.
// A plain hipMemcpy from pageable memory is staged by the driver in small pieces.  Here a ring of four pinned
// 32-MiB buffers is filled by a team of host threads - each chunk is cut into one slice per thread, so the page
// faults and the memcpy of ONE chunk run on all of them (a single filler thread moves 6-9 GB/s: rounds 1-4
// uploaded at 14-17 GB/s) - while up to three earlier chunks are in flight on the copy engine.  The main thread
// only sequences: chunk c is handed to the DMA engine when all of its slices are staged, and its buffer goes back
// to the fillers when that transfer has completed.  Matters for the 27 GB fp32 7B image of the 8-replica config.
int upload_threads() {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  // a cgroup CPU quota makes the online count a lie (256 logical CPUs seen, 16 granted on the GPU boxes)
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      quota = atoll(q);
      if (quota > 0 && quota / period < n) n = (long)(quota / period);
    }
    fclose(f);
  }
  n /= 2;
  return n < 1 ? 1 : (n > 8 ? 8 : (int)n);
}
hipError_t upload_chunked(char* d_dst, const char* h_src, size_t n, hipStream_t stream,
                          float* ms_out) {
  constexpr int NB = 4;
  const size_t CH = (size_t)32 << 20;
  const auto t0 = std::chrono::steady_clock::now();
  if (n <= CH) {
    hipError_t e = hipMemcpy(d_dst, h_src, n, hipMemcpyHostToDevice);
    if (ms_out)
      *ms_out = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return e;
  }
  char* pin[NB] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t done[NB] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  PhaseClock pc;
  for (int i = 0; i < NB && e == hipSuccess; ++i) {
    e = hipHostMalloc((void**)&pin[i], CH, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
  }
  pc.lap("pinned staging ring");
  if (e == hipSuccess) {
    const size_t nchunks = (n + CH - 1) / CH;
    const int NT = upload_threads();
    std::vector<std::atomic<int>> filled(nchunks);  // slices of chunk c staged
    for (auto& f : filled) f.store(0, std::memory_order_relaxed);
    std::atomic<long> released{NB};   // chunks [0, released) may be staged: their buffers are free
    std::atomic<bool> abort_{false};
    auto worker = [&](int w) {
      for (size_t c = 0; c < nchunks; ++c) {
        if (abort_.load(std::memory_order_relaxed)) return;
        while ((long)c >= released.load(std::memory_order_acquire)) {
          if (abort_.load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        const size_t off = c * CH, len = off + CH <= n ? CH : n - off;
        // slice w of the chunk, 4-KiB granular so that two threads never fault the same page
        const size_t per = ((len + NT - 1) / NT + 4095) & ~(size_t)4095;
        const size_t s0 = (size_t)w * per, s1 = s0 + per < len ? s0 + per : len;
        if (s0 < len) memcpy(pin[c % NB] + s0, h_src + off + s0, s1 - s0);
        filled[c].fetch_add(1, std::memory_order_release);
      }
    };
    std::vector<std::thread> team;
    try {
      team.reserve((size_t)NT);
      for (int w = 0; w < NT; ++w) team.emplace_back(worker, w);
    } catch (...) {  // a thread could not be started: the chunks would never fill - stop the ones that run
      abort_.store(true);
      e = hipErrorOutOfMemory;
    }
    for (size_t c = 0; c < nchunks && e == hipSuccess; ++c) {
      while (filled[c].load(std::memory_order_acquire) < NT) std::this_thread::yield();
      const size_t off = c * CH, len = off + CH <= n ? CH : n - off;
      e = hipMemcpyAsync(d_dst + off, pin[c % NB], len, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = hipEventRecord(done[c % NB], stream);
      // chunk c + 1 reuses the buffer of chunk c + 1 - NB: hand it back once that transfer is through
      if (c + 1 >= (size_t)NB && c + 1 < nchunks && e == hipSuccess) {
        e = hipEventSynchronize(done[(c + 1) % NB]);
        released.store((long)c + 2, std::memory_order_release);
      }
    }
    if (e != hipSuccess) abort_.store(true);
    released.store((long)nchunks + NB, std::memory_order_release);
    for (auto& t : team) t.join();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    pc.lap("staged upload");
  }
  for (int i = 0; i < NB; ++i) {
    if (done[i]) (void)hipEventDestroy(done[i]);
    if (pin[i]) (void)hipHostFree(pin[i]);
  }
  if (ms_out)
    *ms_out = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return e;
}

// sin / cos table of the RoPE, rows [0, CL): computed on the host with libm exactly as the CPU backend does
// (cpu/rope_kernel.cpp:4-16), so the fp32 table is bit-identical to the CPU reference's down to row 131071.  Every
// entry is independent, so the rows are cut among a few threads (16.8 M libm calls for a 131072-row cache: 0.25 s on
// one core, a third of a model load), and the job runs beside the weight upload (SinCosJob).
struct SinCosJob {
  std::vector<float> s, c;
  std::vector<std::thread> team;
  void start(size_t CL, int head_size, float theta) {
    const size_t n = CL * (size_t)head_size;
    s.resize(n);
    c.resize(n);
    std::vector<float> freq((size_t)head_size);
    for (int d = 0; d < head_size; ++d) freq[d] = 1.0f / powf(theta, (float)d / (float)head_size);
    const int NT = CL >= 4096 ? upload_threads() : 1;
    for (int w = 0; w < NT; ++w)
      team.emplace_back([this, CL, head_size, freq, w, NT] {
        const size_t p0 = CL * (size_t)w / NT, p1 = CL * (size_t)(w + 1) / NT;
        for (size_t p = p0; p < p1; ++p)
          for (int d = 0; d < head_size; ++d) {
            const float val = (float)p * freq[d];
            s[p * head_size + d] = sinf(val);
            c[p * head_size + d] = cosf(val);
          }
      });
  }
  void join() {
    for (auto& t : team) t.join();
    team.clear();
  }
  ~SinCosJob() { join(); }
};

// ---- weight table ------------------------------------------------------------------------
// Byte offsets are relative to the weight data (= file bytes after the header), mirroring
// kuiperllama_amd/binfmt.py::layout.
int build_weight_table(kh_model* m) {
  const kh_config& c = m->cfg;
  const int L = c.layer_num, dim = c.dim, kvd = c.kv_dim, hid = c.hidden_dim, V = c.vocab_size;
  m->layers.assign((size_t)L, LayerW{});
  char* base = m->arena;
  size_t off = 0;
  if (!c.is_quant) {
    const bool bias = c.family == KH_FAMILY_QWEN2;
    auto takef = [&](size_t n) {
      const float* p = (const float*)(base + off);
      off += n * sizeof(float);
      return p;
    };
    m->tok_emb = takef((size_t)V * dim);
    for (int l = 0; l < L; ++l) m->layers[l].att_norm = takef((size_t)dim);
    for (int l = 0; l < L; ++l) {
      m->layers[l].wq.w = takef((size_t)dim * dim);
      if (bias) m->layers[l].wq.bias = takef((size_t)dim);
    }
    for (int l = 0; l < L; ++l) {
      m->layers[l].wk.w = takef((size_t)kvd * dim);
      if (bias) m->layers[l].wk.bias = takef((size_t)kvd);
    }
    for (int l = 0; l < L; ++l) {
      m->layers[l].wv.w = takef((size_t)kvd * dim);
      if (bias) m->layers[l].wv.bias = takef((size_t)kvd);
    }
    for (int l = 0; l < L; ++l) m->layers[l].wo.w = takef((size_t)dim * dim);
    for (int l = 0; l < L; ++l) m->layers[l].ffn_norm = takef((size_t)dim);
    for (int l = 0; l < L; ++l) m->layers[l].w1.w = takef((size_t)hid * dim);
    for (int l = 0; l < L; ++l) m->layers[l].w2.w = takef((size_t)dim * hid);
    for (int l = 0; l < L; ++l) m->layers[l].w3.w = takef((size_t)hid * dim);
    m->final_norm = takef((size_t)dim);
    (void)takef((size_t)c.seq_len * c.head_size);  // freqs_cos + freqs_sin: skipped (:367-368)
    if (c.is_shared_weight) {
      m->cls.w = m->tok_emb;  // llama3.cpp:372-375
    } else {
      m->cls.w = takef((size_t)V * dim);
    }
  } else {
    const size_t gs = (size_t)c.group_size;
    auto takeq = [&](KhLin& Lw, size_t K, size_t M) {
      const size_t n = K * M;
      Lw.w = base + off;
      Lw.scales = (const float*)(base + off + n);  // layer.cpp:209-215
      off += n + (n / gs) * sizeof(float);
    };
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wq, dim, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wk, kvd, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wv, kvd, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wo, dim, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w1, hid, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w2, dim, hid);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w3, hid, dim);
    takeq(m->cls, V, dim);
    const float* fp = (const float*)(base + off);
    m->tok_emb = fp;
    fp += (size_t)V * dim;
    for (int l = 0; l < L; ++l) m->layers[l].att_norm = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    for (int l = 0; l < L; ++l) m->layers[l].ffn_norm = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    m->final_norm = fp;
    fp += dim;
    off = (size_t)((const char*)fp - base);
  }
  if (off > m->arena_bytes) return KH_ERR_FORMAT;
  return KH_OK;
}

int parse_header(const int32_t* h, const kh_model_opts* o, kh_config* c) {
  // model.cpp:57-71, 125-151
  memset(c, 0, sizeof(*c));
  c->dim = h[0];
  c->hidden_dim = h[1];
  c->layer_num = h[2];
  c->head_num = h[3];
  c->kv_head_num = h[4];
  c->is_shared_weight = h[5] > 0;
  c->vocab_size = h[5] < 0 ? -h[5] : h[5];
  c->seq_len = h[6];
  c->is_quant = o->is_quant ? 1 : 0;
  c->group_size = o->is_quant ? h[7] : 0;
  if (c->dim <= 0 || c->hidden_dim <= 0 || c->layer_num <= 0 || c->head_num <= 0 ||
      c->kv_head_num <= 0 || c->vocab_size <= 0 || c->seq_len <= 0)
    return KH_ERR_FORMAT;
  if (c->dim % c->head_num || c->head_num % c->kv_head_num) return KH_ERR_FORMAT;
  c->kv_dim = (c->dim * c->kv_head_num) / c->head_num;
  c->kv_mul = c->head_num / c->kv_head_num;
  c->head_size = c->dim / c->head_num;
  c->family = o->family;
  c->rope_mode = o->rope_mode;
  c->rope_theta = o->rope_theta;
  c->rms_eps = o->rms_eps;
  c->cache_len = (o->max_seq_len > 0 && o->max_seq_len < c->seq_len) ? o->max_seq_len : c->seq_len;
  if (c->is_quant) {
    if (c->group_size <= 0) return KH_ERR_FORMAT;
    // the reference wires an int8 classifier onto fp32 embedding bytes when the classifier
    // is tied (llama3.cpp:259-262) and has no Qwen2 int8 bias layout: refuse both
    if (c->is_shared_weight || c->family == KH_FAMILY_QWEN2) return KH_ERR_UNSUPPORTED;
  }
  if (o->rope_mode != KH_ROPE_HALF && o->rope_mode != KH_ROPE_INTERLEAVED) return KH_ERR_INVALID_ARG;
  if (o->family != KH_FAMILY_LLAMA && o->family != KH_FAMILY_QWEN2) return KH_ERR_INVALID_ARG;
  if (!(o->rope_theta > 0.f) || !(o->rms_eps > 0.f)) return KH_ERR_INVALID_ARG;
  // vector-path preconditions of the fused kernels
  const int a = c->is_quant ? 16 : 4;
  if (c->dim % a || c->hidden_dim % a || c->head_size % 4 || (c->head_size & 1) ||
      c->head_size > 256 || c->kv_dim % 4 || (c->dim & 1) || (c->kv_dim & 1))
    return KH_ERR_UNSUPPORTED;
  if (c->is_quant) {
    const int gs = ilog2_exact(c->group_size);
    if (gs < 4 || c->dim % c->group_size || c->hidden_dim % c->group_size) return KH_ERR_UNSUPPORTED;
  }
  return KH_OK;
}

size_t expected_weight_bytes(const kh_config& c) {
  const size_t L = c.layer_num, dim = c.dim, kvd = c.kv_dim, hid = c.hidden_dim, V = c.vocab_size;
  const size_t lin = L * (2 * dim * dim + 2 * kvd * dim + 3 * hid * dim);
  if (!c.is_quant) {
    size_t n = V * dim + 2 * L * dim + lin + dim + (size_t)c.seq_len * c.head_size;
    if (c.family == KH_FAMILY_QWEN2) n += L * (dim + 2 * kvd);
    if (!c.is_shared_weight) n += V * dim;
    return n * sizeof(float);
  }
  const size_t q = lin + V * dim;
  return q + (q / (size_t)c.group_size) * sizeof(float) + (V * dim + 2 * L * dim + dim) * sizeof(float);
}

// ---- KV cache: addresses reserved, HBM mapped on demand -------------------------------------------------------------
// The reference allocates [layer_num, seq_len, kv_dim] floats for K and for V up front (llama3.cpp:469-472): 8.6 GB
// for Llama-3.2-1B's 131072-row context, whatever the sequence length - and on some boxes of the pool that one
// hipMalloc takes 0.3-0.5 s, more than the weight upload (profiles/r5_load_probe.txt box B).  Here the same contiguous
// address range is RESERVED (hipMemAddressReserve: no memory, no time) and physical memory is mapped in chunks of
// 8 MiB as generate / predict / prefill / kh_model_write_kv first reach rows (hipMemCreate + hipMemMap +
// hipMemSetAccess on the host, before the launches that touch the rows are enqueued; the new memory is zeroed on the
// model's stream).  Kernel addressing (model.cpp:226-243's slices) and captured graphs are unaffected: the base
// pointers never change.  A 128-step run of Llama-3.2-1B commits 256 MiB of cache instead of 8.6 GB.
// Hook KH_KV_VMM=0 (or a runtime without the virtual-memory API): one plain allocation, as before.
static int kv_ensure_impl(kh_model* m, int row0, int rows, int layer);
static void kv_release(kh_model* m);
// Address ranges outlive the model that reserved them: kv_release hands a range (every chunk unmapped) to this
// process-wide list and the next model whose caches reserve the same number of bytes takes it over; the list is never
// emptied.  hipMemAddressFree is NOT called: on this runtime (ROCm 7.2) it dereferences a null pointer once in a few
// thousand create / destroy cycles of a process that keeps mapping and unmapping chunks (tools/stress_destroy.py under
// tools/dbg/segv_bt.c: the second of kv_release's two calls, inside libamdhip64; seen once in five runs of the GPU
// suite, `profiles/r6_vmm_destroy_crash.txt`; the plain allocation survives 28 000 cycles).  A reservation is address
// space only - no HBM, no page tables until a chunk is mapped - so the cost of keeping it is a few dozen distinct
// sizes of a 47-bit space in the worst process this library has seen (its own test suite).
// Hook KH_KV_VA_POOL=0: free the ranges as rounds 6's first version did (the stress tool uses it to show the crash).
struct VaPool {
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> ranges;
};
static VaPool& va_pool() {
  static VaPool* p = new VaPool();  // never destroyed: models may be released from static destructors
  return *p;
}
static void* va_take(size_t bytes, size_t align) {
  {
    VaPool& vp = va_pool();
    std::lock_guard<std::mutex> g(vp.mu);
    for (size_t i = vp.ranges.size(); i-- > 0;)  // the most recently released range of that size first
      if (vp.ranges[i].second == bytes) {
        void* p = vp.ranges[i].first;
        vp.ranges.erase(vp.ranges.begin() + (long)i);
        return p;
      }
  }
  void* p = nullptr;
  if (hipMemAddressReserve(&p, bytes, align, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
static void va_give(void* p, size_t bytes) {
  if (!p) return;
  if (dbg_off("KH_KV_VA_POOL")) {
    (void)hipMemAddressFree(p, bytes);
    return;
  }
  VaPool& vp = va_pool();
  std::lock_guard<std::mutex> g(vp.mu);
  vp.ranges.emplace_back(p, bytes);
}
static int kv_allocate(kh_model* m) {
  const kh_config& c = m->cfg;
  const size_t total = (size_t)c.layer_num * (size_t)c.cache_len * c.kv_dim * sizeof(float);
  kh_model::KvVmm& kv = m->kv;
  if (!dbg_off("KH_KV_VMM")) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = m->opts.device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0) {
      size_t chunk = (size_t)8 << 20;  // ONE size for every mapping of every model in the process (see kv_ensure)
      chunk = (chunk + gran - 1) / gran * gran;
      const size_t reserved = (total + chunk - 1) / chunk * chunk;
      void *pk = va_take(reserved, chunk), *pv = nullptr;
      if (pk) {
        if ((pv = va_take(reserved, chunk)) != nullptr) {
          kv.on = true;
          kv.chunk = chunk;
          kv.reserved = reserved;
          kv.have[0].assign(reserved / chunk, 0);
          kv.have[1].assign(reserved / chunk, 0);
          m->kcache = (float*)pk;
          m->vcache = (float*)pv;
          // trial: the first row of layer 0 (every sequence needs it).  A process in which this fails - e.g. one whose
          // other users of the API mapped other sizes, see kv_ensure - takes the plain allocation below instead of
          // failing at its first generate
          if (kv_ensure_impl(m, 0, 1, 0) == KH_OK) return KH_OK;
          kv_release(m);
          kv = kh_model::KvVmm();
        } else {
          va_give(pk, reserved);
        }
      }
    }
    (void)hipGetLastError();  // no virtual-memory API on this runtime / device: plain allocation
  }
  int rc;
  if ((rc = dalloc(&m->kcache, total / sizeof(float))) != KH_OK) return rc;
  return dalloc(&m->vcache, total / sizeof(float));
}
static void kv_release(kh_model* m) {
  kh_model::KvVmm& kv = m->kv;
  if (!kv.on) {
    if (m->kcache) (void)hipFree(m->kcache);
    if (m->vcache) (void)hipFree(m->vcache);
  } else {
    for (const auto& r : kv.runs) {
      (void)hipMemUnmap(r.va, r.len);
      (void)hipMemRelease(r.h);
    }
    kv.runs.clear();
    if (m->kcache) va_give(m->kcache, kv.reserved);
    if (m->vcache) va_give(m->vcache, kv.reserved);
    kv.on = false;
  }
  m->kcache = m->vcache = nullptr;
}
// rows [row0, rows) of one layer (layer >= 0) or of every layer
static int kv_ensure_impl(kh_model* m, int row0, int rows, int layer) {
  kh_model::KvVmm& kv = m->kv;
  if (!kv.on || rows <= 0) return KH_OK;
  const kh_config& c = m->cfg;
  if (rows > c.cache_len) rows = c.cache_len;
  if (rows <= kv.rows_all || row0 >= rows) return KH_OK;
  if (row0 < 0) row0 = 0;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = m->opts.device;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t row_b = (size_t)c.kv_dim * sizeof(float), layer_b = (size_t)c.cache_len * row_b;
  const int l0 = layer >= 0 ? layer : 0, l1 = layer >= 0 ? layer + 1 : c.layer_num;
  for (int which = 0; which < 2; ++which) {
    char* base = (char*)(which ? m->vcache : m->kcache);
    std::vector<uint8_t>& have = kv.have[which];
    for (int l = l0; l < l1; ++l) {
      // bytes [b0, b1) of the reservation must be mapped
      const size_t b0 = (size_t)l * layer_b + (size_t)row0 * row_b, b1 = (size_t)l * layer_b + (size_t)rows * row_b;
      size_t ch = b0 / kv.chunk;
      const size_t ch_end = (b1 + kv.chunk - 1) / kv.chunk;
      while (ch < ch_end) {
        if (have[ch]) {
          ++ch;
          continue;
        }
        size_t e = ch;
        while (e < ch_end && !have[e]) ++e;  // a run of unmapped chunks: zeroed with one memset
        // ONE handle per chunk, every handle the same size: on this runtime (ROCm 7.2) hipMemSetAccess returns
        // "invalid argument" for a mapping that is smaller than one made earlier in the same process
        // (tools/mb_vmm2.hip, profiles/r6_mb_vmm.txt); equal-sized mappings - what PyTorch's expandable segments
        // use - are reliable, and a chunk costs 13 us to create + map + enable
        for (size_t k = ch; k < e; ++k) {
          kh_model::KvVmm::Run r;
          r.va = base + k * kv.chunk;
          r.len = kv.chunk;
          const char* what = "hipMemCreate";
          hipError_t err = hipMemCreate(&r.h, r.len, &prop, 0);
          if (err == hipSuccess) {
            what = "hipMemMap";
            err = hipMemMap(r.va, r.len, 0, r.h, 0);
            if (err == hipSuccess) {
              what = "hipMemSetAccess";
              err = hipMemSetAccess(r.va, r.len, &acc, 1);
              if (err != hipSuccess) (void)hipMemUnmap(r.va, r.len);
            }
            if (err != hipSuccess) (void)hipMemRelease(r.h);
          }
          if (err != hipSuccess) {
            fprintf(stderr, "[kh] KV cache: %s of %zu MiB at chunk %zu of the %s cache (layer %d) failed: %s\n", what,
                    r.len >> 20, k, which ? "V" : "K", l, hipGetErrorString(err));
            (void)hipGetLastError();
            return (int)err;
          }
          kv.runs.push_back(r);
          have[k] = 1;
          kv.mapped += r.len;
        }
        KH_CHECK_HIP(hipMemsetAsync(base + ch * kv.chunk, 0, (e - ch) * kv.chunk, m->stream));
        ch = e;
      }
    }
  }
  if (layer < 0 && row0 == 0) kv.rows_all = rows;
  return KH_OK;
}

int finish_create(kh_model* m, SinCosJob* sincos = nullptr) {
  const kh_config& c = m->cfg;
  int rc;
  PhaseClock pc;
  if ((rc = build_weight_table(m)) != KH_OK) return rc;
  m->gshift = c.is_quant ? ilog2_exact(c.group_size) : 0;
  const size_t CL = (size_t)c.cache_len;
#define KH_ALLOC(ptr, n) \
  if ((rc = dalloc(&(ptr), (n))) != KH_OK) return rc
  KH_ALLOC(m->x, (size_t)c.dim);
  KH_ALLOC(m->rms, (size_t)c.dim);
  KH_ALLOC(m->q, (size_t)c.dim);
  KH_ALLOC(m->att, (size_t)c.dim);
  KH_ALLOC(m->w2o, (size_t)c.dim);
  KH_ALLOC(m->h1, (size_t)c.hidden_dim);
  KH_ALLOC(m->h3, (size_t)c.hidden_dim);
  KH_ALLOC(m->logits, (size_t)c.vocab_size);
  KH_ALLOC(m->score, (size_t)c.head_num * CL);
  pc.lap("weight table + small buffers");
  if ((rc = kv_allocate(m)) != KH_OK) return rc;
  KH_ALLOC(m->sin_cache, CL * c.head_size);
  KH_ALLOC(m->cos_cache, CL * c.head_size);
  KH_ALLOC(m->d_pos, 1);
  KH_ALLOC(m->d_token, 1);
  KH_ALLOC(m->d_next, 1);
  // launch geometry
  {
    kh_model::Shape sh[5];
    plan_decode_shapes(c.is_quant != 0, c.dim, c.hidden_dim, c.kv_dim, c.vocab_size, sh);
    m->sh_qkv = sh[0]; m->sh_wo = sh[1]; m->sh_ffn = sh[2]; m->sh_w2 = sh[3]; m->sh_cls = sh[4];
  }
  plan_ring(c.is_quant != 0, c.dim, c.hidden_dim, c.vocab_size, c.group_size, &m->ring);
  m->nparts = m->ring.cls_r ? m->ring.cls_grid : m->sh_cls.grid;
  if (dbg("KH_SHAPE_DEBUG")) {
    if (m->ring.ffn_r || m->ring.cls_r)
      fprintf(stderr, "[kh] LDS-DMA ring: ffn13 R %d grid %d | cls R %d grid %d (256 threads)\n", m->ring.ffn_r,
              m->ring.ffn_grid, m->ring.cls_r, m->ring.cls_grid);
    const struct { const char* n; const kh_model::Shape* s; } all[] = {
        {"qkv", &m->sh_qkv}, {"wo", &m->sh_wo}, {"ffn13", &m->sh_ffn}, {"w2", &m->sh_w2}, {"cls", &m->sh_cls}};
    for (const auto& e : all)
      fprintf(stderr, "[kh] shape %-5s split %d u %d grid %d wg %d\n", e.n, e.s->split, e.s->u, e.s->grid, e.s->wg);
  }
  // attention: 8 waves per (head, split) shorten each lane's timestep loop
  m->attn_wg = KH_WG_MAX;
  if (const char* e = dbg("KH_ATTN_WG"))
    if (atoi(e) == 256 || atoi(e) == 512) m->attn_wg = atoi(e);
  {
    const AttnPlan ap = attn_plan(c.head_num, c.kv_mul, c.head_size, c.cache_len, m->attn_wg, attn_tlong_hook());
    m->attn_ns = ap.ns;
    m->attn_ts_shift = ap.ts_shift;
    m->attn_ns_g = ap.ns_g;
    m->attn_ws_stride = ap.stride;
    m->attn_t_long = ap.t_long;
  }
  // step variant 1 - time splits merged by k_wo_comb instead of a last arriver (kh_attn.h, kh_fused.h) -
  // needs wo's in-register staging (dim <= 16 floats per thread) and heads * 16 factor slots per pass
  {
    const char* e = dbg("KH_ATTN_FENCED");
    m->attn_fenced = (m->opts.flags & KH_FLAG_ATTN_MERGE_FENCED) != 0 || (e && e[0] == '1');
  }
  m->attn_defer = m->attn_ns > 1 && !(m->opts.flags & KH_FLAG_ATTN_MERGE_IN_LAUNCH) && !dbg_off("KH_ATTN_DEFER") &&
                  comb_supported(c.dim, c.head_num, c.head_size, m->sh_wo.wg);
  {
    // mirrors k_wo_comb's OVERLAP (kh_fused.h): does the combine travel beside wo's first weight tile?
    const int mv = c.dim <= 2 * 4 * m->sh_wo.wg ? 2 : 4;
    const bool overlap = !(m->sh_wo.u >= 8 || (c.is_quant && m->sh_wo.u >= 4 && mv >= 4));
    m->attn_defer_max = overlap ? KH_ATTN_MAX_NS : 4;
    if (const char* e = dbg("KH_ATTN_DEFER_MAX")) m->attn_defer_max = atoi(e);
  }
  if (const size_t wsb = attn_ws_bytes(c.head_num, c.head_size, m->attn_ws_stride)) {
    KH_CHECK_HIP(hipMalloc(&m->attn_ws, wsb));
    KH_CHECK_HIP(hipMemsetAsync(m->attn_ws, 0, wsb, m->stream));
  }
  KH_ALLOC(m->part_val, (size_t)m->nparts);
  KH_ALLOC(m->part_idx, (size_t)m->nparts);
#undef KH_ALLOC
  pc.lap("caches, tables, plans, workspace");
  if (!m->kv.on) {  // a mapped-on-demand cache zeroes what it maps (kv_ensure)
    KH_CHECK_HIP(hipMemsetAsync(m->kcache, 0, sizeof(float) * c.layer_num * CL * c.kv_dim, m->stream));
    KH_CHECK_HIP(hipMemsetAsync(m->vcache, 0, sizeof(float) * c.layer_num * CL * c.kv_dim, m->stream));
  }
  KH_CHECK_HIP(hipMemsetAsync(m->d_pos, 0, sizeof(int32_t), m->stream));
  KH_CHECK_HIP(hipMemsetAsync(m->d_token, 0, sizeof(int32_t), m->stream));
  // sin/cos table: computed on the host with libm exactly as the CPU backend does
  // (cpu/rope_kernel.cpp:4-16) so the fp32 table is bit-identical to the CPU reference's,
  // then uploaded once.  (kh_sincos_cache_f32 is the on-device twin of sin_cos_cache_calc_cu.)
  {
    SinCosJob local;
    SinCosJob* job = sincos;  // started beside the weight upload by the callers that upload; here otherwise
    if (!job) {
      local.start(CL, c.head_size, c.rope_theta);
      job = &local;
    }
    job->join();
    pc.lap("wait for the RoPE table threads");
    const size_t n = CL * c.head_size;
    KH_CHECK_HIP(hipMemcpy(m->sin_cache, job->s.data(), n * sizeof(float), hipMemcpyHostToDevice));
    KH_CHECK_HIP(hipMemcpy(m->cos_cache, job->c.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  pc.lap("RoPE table upload");
  if ((rc = configure_step_kernels(m)) != KH_OK) return rc;
  KH_CHECK_HIP(hipEventCreate(&m->ev0));
  KH_CHECK_HIP(hipEventCreate(&m->ev1));
  if ((rc = ensure_seq_cap(m, 256)) != KH_OK) return rc;
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  pc.lap("kernel attributes, events, sync");
  return KH_OK;
}

int new_model(const int32_t* h_header, const kh_model_opts* opts, kh_model** out) {
  if (!h_header || !opts || !out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return KH_ERR_NO_DEVICE;
  }
  if (opts->device < 0 || opts->device >= ndev) return KH_ERR_INVALID_ARG;
  kh_config cfg;
  int rc = parse_header(h_header, opts, &cfg);
  if (rc != KH_OK) return rc;
  KH_CHECK_HIP(hipSetDevice(opts->device));
  kh_model* m = new (std::nothrow) kh_model();
  if (!m) return (int)hipErrorOutOfMemory;
  m->cfg = cfg;
  m->opts = *opts;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete m;
    return (int)e;
  }
  *out = m;
  return KH_OK;
}

}  // namespace

namespace khm {
int kv_ensure(kh_model* m, int rows, int layer) { return kv_ensure_impl(m, 0, rows, layer); }
}  // namespace khm

// =============================================================================================
extern "C" void kh_model_destroy(kh_model* m) {
  if (!m) return;
  if (m->unmap_thread.joinable()) m->unmap_thread.join();  // kh_model_create_from_file's munmap helper
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  destroy_step_graphs(m);
  if (m->ev0) (void)hipEventDestroy(m->ev0);
  if (m->ev1) (void)hipEventDestroy(m->ev1);
  for (void* q : {(void*)m->pf_x, (void*)m->pf_q, (void*)m->pf_att, (void*)m->pf_h, m->pf_ws,
                  (void*)m->pg_x, (void*)m->pg_xn, (void*)m->pg_q, (void*)m->pg_att, (void*)m->pg_h, (void*)m->pg_part,
                  m->pg_ws})
    if (q) (void)hipFree(q);
  for (auto e : m->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  if (m->h_words_pin) (void)hipHostFree(m->h_words_pin);
  if (m->h_forced_pin) (void)hipHostFree(m->h_forced_pin);
  if (m->first_logits) (void)hipFree(m->first_logits);
  void* bufs[] = {m->x,      m->rms,    m->q,         m->att,       m->h1,       m->h3,
                  m->w2o,    m->logits, m->score,     m->sin_cache,
                  m->cos_cache, m->part_val, m->part_idx, m->d_pos, m->d_token,  m->d_next,
                  m->d_forced, m->d_words, m->attn_ws};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  kv_release(m);
  if (m->owns_arena && m->arena) (void)hipFree(m->arena);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

static int create_from_device_weights_impl(const int32_t* h_header, const void* d_weight_data, size_t weight_nbytes,
                                           const kh_model_opts* opts, kh_model** out) {
  kh_model* m = nullptr;
  int rc = new_model(h_header, opts, &m);
  if (rc != KH_OK) return rc;
  *out = m;  // see create_from_host_image_impl
  if (weight_nbytes < expected_weight_bytes(m->cfg)) {
    kh_model_destroy(m);
    *out = nullptr;
    return KH_ERR_FORMAT;
  }
  m->arena = (char*)const_cast<void*>(d_weight_data);
  m->owns_arena = false;
  m->arena_bytes = weight_nbytes;
  m->cfg.weight_bytes = (int64_t)expected_weight_bytes(m->cfg);
  rc = finish_create(m);
  if (rc == KH_OK) rc = run_selftests(m);  // the weights are resident: ring kernels / split merge against their fallbacks
  if (rc != KH_OK) {
    kh_model_destroy(m);
    m = nullptr;
  }
  *out = m;
  return rc;
}
extern "C" int kh_model_create_from_device_weights(const int32_t* h_header,
                                                   const void* d_weight_data,
                                                   size_t weight_nbytes,
                                                   const kh_model_opts* opts, kh_model** out) {
  if (!d_weight_data || !kh_aligned16(d_weight_data)) return KH_ERR_INVALID_ARG;
  if (!out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  const int rc =
      kh_api_guard([&] { return create_from_device_weights_impl(h_header, d_weight_data, weight_nbytes, opts, out); });
  if (rc != KH_OK && *out) {  // only after an exception
    kh_model_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

// *out holds the model from the moment it exists, so that the boundary below can release it when an exception
// (host allocation, thread creation) cuts the construction short
static int create_from_host_image_impl(const void* h_image, size_t nbytes, const kh_model_opts* opts,
                                       kh_model** out) {
  const size_t hdr = opts->is_quant ? 32 : 28;
  if (nbytes < hdr) return KH_ERR_FORMAT;
  int32_t header[8] = {0};
  memcpy(header, h_image, hdr);
  kh_model* m = nullptr;
  int rc = new_model(header, opts, &m);
  if (rc != KH_OK) return rc;
  *out = m;
  const size_t need = expected_weight_bytes(m->cfg);
  if (nbytes - hdr < need) {
    kh_model_destroy(m);
    *out = nullptr;
    return KH_ERR_FORMAT;
  }
  SinCosJob sincos;  // the RoPE table is computed on host threads while the weights travel
  sincos.start((size_t)m->cfg.cache_len, m->cfg.head_size, m->cfg.rope_theta);
  PhaseClock pc;
  hipError_t e = hipMalloc((void**)&m->arena, need);
  pc.lap("arena hipMalloc");
  if (e != hipSuccess) {
    sincos.join();
    kh_model_destroy(m);
    *out = nullptr;
    return (int)e;
  }
  m->owns_arena = true;
  m->arena_bytes = need;
  m->cfg.weight_bytes = (int64_t)need;
  // Weights go up once, in file order, into one arena (the reference cudaMallocs and copies every tensor
  // separately: tensor.cpp:104-119) - on a helper thread, while this thread builds everything that does not read
  // them: the weight table (addresses only), the KV cache and activation buffers (hipMalloc of 8.6 GB of cache for
  // a 131072-row Llama-3.2-1B context takes 0.1-0.4 s on some boxes: as long as the upload itself), launch plans,
  // the RoPE table's upload.  Both sides enqueue on the model's stream; nothing here launches a kernel.
  hipError_t eu = hipSuccess;
  const int dev = m->opts.device;
  std::thread uploader([&] {
    eu = hipSetDevice(dev);
    if (eu == hipSuccess) eu = upload_chunked(m->arena, (const char*)h_image + hdr, need, m->stream, &m->load_ms);
  });
  KhJoinOnExit join_uploader{uploader};
  rc = finish_create(m, &sincos);
  uploader.join();
  if (rc == KH_OK && eu != hipSuccess) rc = (int)eu;
  if (rc == KH_OK) {
    const hipError_t es = hipStreamSynchronize(m->stream);
    if (es != hipSuccess) rc = (int)es;
  }
  pc.lap("upload || buffers, joined");
  if (rc == KH_OK) rc = run_selftests(m);
  pc.lap("self-tests");
  if (rc != KH_OK) {
    kh_model_destroy(m);
    m = nullptr;
  }
  *out = m;
  return rc;
}
extern "C" int kh_model_create_from_host_image(const void* h_image, size_t nbytes,
                                               const kh_model_opts* opts, kh_model** out) {
  if (!h_image || !opts || !out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  const int rc = kh_api_guard([&] { return create_from_host_image_impl(h_image, nbytes, opts, out); });
  if (rc != KH_OK && *out) {  // only after an exception: the failure paths of the body release the model themselves
    kh_model_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

extern "C" int kh_model_create_from_file(const char* path, const kh_model_opts* opts,
                                         kh_model** out) {
  if (!path || !opts || !out) return KH_ERR_INVALID_ARG;
  // model.cpp:41-123: open + fstat + mmap(PROT_READ, MAP_PRIVATE)
  const int fd = open(path, O_RDONLY);
  if (fd == -1) return KH_ERR_IO;
  struct stat st;
  if (fstat(fd, &st) == -1 || st.st_size <= 0) {
    close(fd);
    return KH_ERR_IO;
  }
  PhaseClock pc;
  void* data = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (data == MAP_FAILED || data == nullptr) {
    close(fd);
    return KH_ERR_IO;
  }
  // one sequential pass over the whole file: read ahead aggressively, drop behind
  (void)madvise(data, (size_t)st.st_size, MADV_SEQUENTIAL);
  (void)madvise(data, (size_t)st.st_size, MADV_WILLNEED);
  pc.lap("open + mmap + madvise");
  const int rc = kh_model_create_from_host_image(data, (size_t)st.st_size, opts, out);
  pc.lap("create_from_host_image");
  // Tearing down the page tables of a multi-GB mapping whose every page was touched costs tens of milliseconds
  // (1.2 M PTEs for the 4.98 GB Llama-3.2-1B image: 60-80 ms measured) and nobody waits for it: a helper thread
  // unmaps and closes while the caller already decodes.  The thread belongs to the model and kh_model_destroy joins
  // it, so no code of this library runs behind the caller's back once every model is destroyed (dlclose-safe).
  const size_t len = (size_t)st.st_size;
  bool handed = false;
  if (rc == KH_OK && out && *out) {
    try {
      (*out)->unmap_thread = std::thread([data, len, fd] {
        munmap(data, len);
        close(fd);
      });
      handed = true;
    } catch (...) {  // no helper thread to be had: unmap here
    }
  }
  if (!handed) {
    munmap(data, len);
    close(fd);
  }
  pc.lap("munmap handed to a helper thread");
  return rc;
}

extern "C" int kh_model_get_config(const kh_model* m, kh_config* out) {
  if (!m || !out) return KH_ERR_INVALID_ARG;
  *out = m->cfg;
  out->launches_per_token = 5 * m->cfg.layer_num + 2;
  return KH_OK;
}
extern "C" float kh_model_get_load_ms(const kh_model* m) { return m ? m->load_ms : -1.f; }
extern "C" void* kh_model_stream(kh_model* m) { return m ? (void*)m->stream : nullptr; }

extern "C" int kh_model_get_logits(kh_model* m, float* h_logits) {
  if (!m || !h_logits) return KH_ERR_INVALID_ARG;
  KH_CHECK_HIP(hipMemcpyAsync(h_logits, m->logits, sizeof(float) * m->cfg.vocab_size,
                              hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}
extern "C" int kh_model_first_sample(kh_model* m, kh_first_sample* out) {
  if (!m || !out) return KH_ERR_INVALID_ARG;
  if (m->first_pos < 0 || !m->first_logits) return KH_ERR_UNSUPPORTED;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  const int V = m->cfg.vocab_size;
  std::vector<float> h((size_t)V);
  KH_CHECK_HIP(hipMemcpyAsync(h.data(), m->first_logits, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  // two largest, ties -> lowest index first (the sampler's rule, argmax_sampler.cpp:7)
  int i1 = 0, i2 = -1;
  for (int i = 1; i < V; ++i) {
    if (h[i] > h[i1]) {
      i2 = i1;
      i1 = i;
    } else if (i2 < 0 || h[i] > h[i2]) {
      i2 = i;
    }
  }
  out->pos = m->first_pos;
  out->prefill_mode = m->first_mode;
  out->top1_id = i1;
  out->top2_id = i2;
  out->top1 = h[i1];
  out->top2 = i2 >= 0 ? h[i2] : h[i1];
  return KH_OK;
}
extern "C" int kh_model_get_kv(kh_model* m, float** d_kcache, float** d_vcache) {
  if (!m || !d_kcache || !d_vcache) return KH_ERR_INVALID_ARG;
  // raw pointers leave the library: every row must be backed (a mapped-on-demand cache commits all of it here)
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  const int rc = kv_ensure(m, m->cfg.cache_len);
  if (rc != KH_OK) return rc;
  *d_kcache = m->kcache;
  *d_vcache = m->vcache;
  return KH_OK;
}

extern "C" int kh_model_kv_bytes(const kh_model* m, int64_t* reserved, int64_t* committed) {
  if (!m || !reserved || !committed) return KH_ERR_INVALID_ARG;
  const int64_t total = 2 * (int64_t)m->cfg.layer_num * m->cfg.cache_len * m->cfg.kv_dim * (int64_t)sizeof(float);
  *reserved = total;
  *committed = m->kv.on ? (int64_t)m->kv.mapped : total;
  return KH_OK;
}

extern "C" int kh_model_read_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows,
                                float* h_k, float* h_v) {
  if (!m || !h_k || !h_v || layer < 0 || layer >= m->cfg.layer_num || row0 < 0 || nrows <= 0 ||
      (int64_t)row0 + nrows > m->cfg.cache_len)
    return KH_ERR_INVALID_ARG;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  {
    const int rc = kv_ensure_impl(m, row0, row0 + nrows, layer);  // rows nobody reached yet read as zeros
    if (rc != KH_OK) return rc;
  }
  const size_t off = ((size_t)layer * m->cfg.cache_len + row0) * m->cfg.kv_dim;
  const size_t nb = (size_t)nrows * m->cfg.kv_dim * sizeof(float);
  KH_CHECK_HIP(hipMemcpyAsync(h_k, m->kcache + off, nb, hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipMemcpyAsync(h_v, m->vcache + off, nb, hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

extern "C" int kh_model_write_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows,
                                 const float* h_k, const float* h_v) {
  if (!m || !h_k || !h_v || layer < 0 || layer >= m->cfg.layer_num || row0 < 0 || nrows <= 0 ||
      (int64_t)row0 + nrows > m->cfg.cache_len)
    return KH_ERR_INVALID_ARG;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  {
    const int rc = kv_ensure_impl(m, row0, row0 + nrows, layer);  // only the chunks these rows live in
    if (rc != KH_OK) return rc;
  }
  const size_t off = ((size_t)layer * m->cfg.cache_len + row0) * m->cfg.kv_dim;
  const size_t nb = (size_t)nrows * m->cfg.kv_dim * sizeof(float);
  // hipMemcpyDefault: the source may be host memory or memory of this device (unified addressing)
  KH_CHECK_HIP(hipMemcpyAsync(m->kcache + off, h_k, nb, hipMemcpyDefault, m->stream));
  KH_CHECK_HIP(hipMemcpyAsync(m->vcache + off, h_v, nb, hipMemcpyDefault, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}
