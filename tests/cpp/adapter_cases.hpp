// adapter_cases.hpp — the thirteen operator checks driven through
// kuiper_hip::Kernels<Tensor, Config, DeviceType>'s get_*_kernel() getters, written once over a
// small traits type so that the SAME cases run with
//   * a stand-in Tensor                      (tests/cpp/test_adapter.cpp, travels to the GPU box), and
//   * the reference's real tensor::Tensor    (tests/cpp/test_ref_binding.cpp, built only where
//                                             /root/reference exists).
// The integer-exact cases restate the reference's own op tests (test/test_op/test_cu_matmul.cpp:48-106,
// test_cu_add.cpp:7-75, test_cu_emb.cpp:6-89, test_load.cpp:49-108 style); the others compare with
// the arithmetic of the CPU backend (cpu/*.cpp) evaluated in double right here.
//
// Traits contract:
//   using Tensor / Config / DeviceType;
//   static Tensor dev_f32(const std::vector<float>&, std::vector<int32_t> dims);
//   static Tensor dev_i8(const std::vector<int8_t>&, std::vector<int32_t> dims);
//   static Tensor host_i32(const std::vector<int32_t>&, std::vector<int32_t> dims);
//   static Tensor null_f32(int32_t n);                    // dims set, no storage
//   static std::vector<float> to_host(const Tensor&);     // synchronises
//   static DeviceType device();
//   static void set_stream(Config&, void* stream);
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kuiper_hip_adapter.hpp"

#define KH_REQUIRE(c)                                                     \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);            \
      return 1;                                                           \
    }                                                                     \
  } while (0)

namespace adapter_cases {

struct Lcg {  // deterministic values in [-1, 1)
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  float next() {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f;
  }
};

template <class Tr>
int run(void* stream) {
  using Tensor = typename Tr::Tensor;
  using K = kuiper_hip::Kernels<Tensor, typename Tr::Config, typename Tr::DeviceType>;
  typename Tr::Config cfg;
  Tr::set_stream(cfg, stream);
  int covered = 0;

  {  // 1. MatmulKernel — test_matmul_cu.matmul_linear_course: [1,1,-1] x [[1..9]] -> [0,3,6]
    Tensor x = Tr::dev_f32({1, 1, -1}, {3});
    Tensor w = Tr::dev_f32({1, 2, 3, 4, 5, 6, 7, 8, 9}, {3, 3});
    Tensor y = Tr::dev_f32({9, 9, 9}, {3});
    K::get_matmul_kernel()(x, w, y, 1.f, &cfg);
    auto h = Tr::to_host(y);
    KH_REQUIRE(h[0] == 0.f && h[1] == 3.f && h[2] == 6.f);
    // and a vector-path shape against a double dot product
    const int Kr = 48, M = 512;
    Lcg g(7);
    std::vector<float> xv(M), wv((size_t)Kr * M);
    for (auto& v : xv) v = g.next();
    for (auto& v : wv) v = g.next();
    Tensor x2 = Tr::dev_f32(xv, {M}), w2 = Tr::dev_f32(wv, {Kr, M});
    Tensor y2 = Tr::dev_f32(std::vector<float>(Kr, 0.f), {Kr});
    K::get_matmul_kernel()(x2, w2, y2, 0.5f, &cfg);
    h = Tr::to_host(y2);
    for (int r = 0; r < Kr; ++r) {
      double acc = 0;
      for (int i = 0; i < M; ++i) acc += (double)xv[i] * wv[(size_t)r * M + i];
      KH_REQUIRE(std::fabs(h[r] - 0.5 * acc) < 2e-5);
    }
    ++covered;
  }
  {  // 2. AddKernel — test_add_cu: 2 + 3 = 5 over 4832 elements
    const int n = 4832;
    Tensor a = Tr::dev_f32(std::vector<float>(n, 2.f), {n}), b = Tr::dev_f32(std::vector<float>(n, 3.f), {n});
    Tensor o = Tr::dev_f32(std::vector<float>(n, 0.f), {n});
    K::get_add_kernel()(a, b, o, stream);
    for (float v : Tr::to_host(o)) KH_REQUIRE(v == 5.f);
    ++covered;
  }
  {  // 3. EmbeddingKernel — test_emb_cu: table arange(4x512); host token tensor [1, 3, 99]
    std::vector<float> tab(4 * 512);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = (float)i;
    Tensor w = Tr::dev_f32(tab, {4, 512});
    Tensor toks = Tr::host_i32({1, 3, 99}, {3});
    Tensor out = Tr::dev_f32(std::vector<float>(3 * 512, -1.f), {3, 512});
    K::get_emb_kernel()(toks, w, out, 4, stream);
    auto h = Tr::to_host(out);
    for (int i = 0; i < 512; ++i) {
      KH_REQUIRE(h[i] == 512.f + i);
      KH_REQUIRE(h[512 + i] == 3 * 512.f + i);
      KH_REQUIRE(h[1024 + i] == -1.f);  // out-of-vocabulary id: row left untouched (emb_kernel.cu:10-12)
    }
    ++covered;
  }
  const int n = 480;
  std::vector<float> xs(n), ws(n);
  for (int i = 0; i < n; ++i) {
    xs[i] = 0.001f * (i % 97) + 0.1f;
    ws[i] = 0.01f * (i % 13) + 0.5f;
  }
  {  // 4./5. RMSNormKernel, SwigluKernel vs the CPU formulas, tolerance 1e-5 as in the reference tests
    double ss = 0;
    for (float v : xs) ss += (double)v * v;
    const double r = 1.0 / std::sqrt(ss / n + 1e-5);
    Tensor xd = Tr::dev_f32(xs, {n}), wd = Tr::dev_f32(ws, {n}), od = Tr::dev_f32(std::vector<float>(n), {n});
    kuiper_hip::flavor().rms_eps = 1e-5f;
    K::get_rmsnorm_kernel()(xd, wd, od, stream);
    auto h = Tr::to_host(od);
    for (int i = 0; i < n; ++i) KH_REQUIRE(std::fabs(h[i] - ws[i] * r * xs[i]) < 1e-5);
    ++covered;
    Tensor sd = Tr::dev_f32(std::vector<float>(n), {n});
    K::get_swiglu_kernel()(xd, wd, sd, stream);
    h = Tr::to_host(sd);
    for (int i = 0; i < n; ++i) KH_REQUIRE(std::fabs(h[i] - xs[i] / (1 + std::exp(-xs[i])) * ws[i]) < 1e-5);
    ++covered;
  }
  {  // 6. argmax_kernel_cu twin: first maximum
    std::vector<float> l(32000, -1.f);
    l[777] = 4.f;
    l[31000] = 4.f;
    Tensor ld = Tr::dev_f32(l, {32000});
    KH_REQUIRE(K::argmax(ld.template ptr<float>(), 32000, stream) == 777);
    ++covered;
  }
  {  // 7. MatmulKernelQuant: y[p] = sum_i x[i] * scale[(p*M+i)/g] * w8[p*M+i]  (matmul_kernel.cu:56-87)
    const int Kr = 10, M = 256, g = 64;
    Lcg r(11);
    std::vector<float> xv(M), sc((size_t)Kr * M / g);
    std::vector<int8_t> w8((size_t)Kr * M);
    for (auto& v : xv) v = r.next();
    for (auto& v : sc) v = 0.01f + 0.005f * (r.next() + 1.f);
    for (auto& v : w8) v = (int8_t)std::lrint(127.f * r.next());
    Tensor x = Tr::dev_f32(xv, {M}), w = Tr::dev_i8(w8, {Kr, M}), s = Tr::dev_f32(sc, {(int32_t)sc.size()});
    Tensor y = Tr::dev_f32(std::vector<float>(Kr, 0.f), {Kr});
    K::get_matmul_kernel_quant8()(x, w, y, g, s, &cfg);
    auto h = Tr::to_host(y);
    for (int p = 0; p < Kr; ++p) {
      double acc = 0, nx = 0, nw = 0;
      for (int i = 0; i < M; ++i) {
        const double wd = (double)sc[((size_t)p * M + i) / g] * w8[(size_t)p * M + i];
        acc += xv[i] * wd;
        nx += (double)xv[i] * xv[i];
        nw += wd * wd;
      }
      KH_REQUIRE(std::fabs(h[p] - acc) <= 1e-5 * std::sqrt(nx) * std::sqrt(nw) + 1e-7);
    }
    ++covered;
  }
  const int hs = 64, heads = 4, kv_mul = 2, kv_heads = heads / kv_mul, dim = hs * heads, kv_dim = hs * kv_heads;
  const int seq = 32;
  std::vector<float> sinc, cosc;
  {  // 8. sin_cos_cache_calc_cu: cache[pos*hs+d] = sin/cos(float(pos) * 1/pow(theta, d/hs))  (rope_kernel.cpp:4-16)
    kuiper_hip::flavor().rope_theta = 10000.0f;
    Tensor s = Tr::dev_f32(std::vector<float>((size_t)seq * hs), {seq, hs});
    Tensor c = Tr::dev_f32(std::vector<float>((size_t)seq * hs), {seq, hs});
    K::sin_cos_cache_calc(hs, seq, s, c, stream);
    sinc = Tr::to_host(s);
    cosc = Tr::to_host(c);
    for (int p = 0; p < seq; ++p)
      for (int d = 0; d < hs; ++d) {
        const float freq = 1.0f / std::pow(10000.0f, (float)d / (float)hs);
        const float val = (float)p * freq;
        KH_REQUIRE(std::fabs(sinc[p * hs + d] - std::sin(val)) < 5e-7);
        KH_REQUIRE(std::fabs(cosc[p * hs + d] - std::cos(val)) < 5e-7);
      }
    ++covered;
  }
  Lcg rq(23);
  std::vector<float> q(dim), kk(kv_dim);
  for (auto& v : q) v = rq.next();
  for (auto& v : kk) v = rq.next();
  for (int mode = 0; mode < 2; ++mode) {  // 9. RoPEKernel, both flavours (rope_kernel.cpp:18-42, 98-121)
    const int pos = 3;
    kuiper_hip::flavor().rope_mode = mode ? KH_ROPE_HALF : KH_ROPE_INTERLEAVED;
    Tensor qd = Tr::dev_f32(q, {dim}), kd = Tr::dev_f32(kk, {kv_dim});
    Tensor sd = Tr::dev_f32(sinc, {seq, hs}), cd = Tr::dev_f32(cosc, {seq, hs});
    Tensor pd = Tr::host_i32({pos}, {1});
    K::get_rope_kernel()(dim, kv_dim, hs, qd, kd, pd, sd, cd, stream);
    auto hq = Tr::to_host(qd), hk = Tr::to_host(kd);
    auto expect = [&](const std::vector<float>& v, int len, std::vector<float>& out) {
      out = v;
      if (mode == 0) {
        for (int i = 0; i < len; i += 2) {
          const int hd = i % hs;
          const float fci = sinc[pos * hs + hd], fcr = cosc[pos * hs + hd];
          out[i] = v[i] * fcr - v[i + 1] * fci;
          out[i + 1] = v[i] * fci + v[i + 1] * fcr;
        }
      } else {
        for (int h0 = 0; h0 < len; h0 += hs)
          for (int j = 0; j < hs / 2; ++j) {
            const float fci = sinc[pos * hs + 2 * j], fcr = cosc[pos * hs + 2 * j];
            const float v0 = v[h0 + j], v1 = v[h0 + j + hs / 2];
            out[h0 + j] = v0 * fcr - v1 * fci;
            out[h0 + j + hs / 2] = v0 * fci + v1 * fcr;
          }
      }
    };
    std::vector<float> eq, ek;
    expect(q, dim, eq);
    expect(kk, kv_dim, ek);
    for (int i = 0; i < dim; ++i) KH_REQUIRE(std::fabs(hq[i] - eq[i]) < 1e-6);
    for (int i = 0; i < kv_dim; ++i) KH_REQUIRE(std::fabs(hk[i] - ek[i]) < 1e-6);
  }
  kuiper_hip::flavor().rope_mode = KH_ROPE_INTERLEAVED;
  ++covered;
  {  // 10. MHAKernel (mha_kernel.cpp:5-61): layer 1 of a 2-layer cache, pos 5, GQA kv_mul 2
    const int layers = 2, layer = 1, pos = 5;
    std::vector<float> kc((size_t)layers * seq * kv_dim), vc(kc.size());
    for (auto& v : kc) v = rq.next();
    for (auto& v : vc) v = rq.next();
    Tensor qd = Tr::dev_f32(q, {dim});
    Tensor kd = Tr::dev_f32(kc, {layers, seq, kv_dim}), vd = Tr::dev_f32(vc, {layers, seq, kv_dim});
    Tensor sc = Tr::dev_f32(std::vector<float>((size_t)heads * seq, 0.f), {heads, seq});
    Tensor od = Tr::dev_f32(std::vector<float>(dim, 0.f), {dim});
    K::get_mha_kernel()(pos, heads, layer, seq, kv_dim, kv_mul, hs, od, qd, sc, kd, vd, Tr::device(), &cfg);
    auto ho = Tr::to_host(od), hsc = Tr::to_host(sc);
    const size_t loff = (size_t)layer * seq * kv_dim;
    for (int h = 0; h < heads; ++h) {
      const size_t hoff = (size_t)(h / kv_mul) * hs;
      std::vector<double> p(pos + 1);
      double mx = -1e30, sum = 0;
      for (int t = 0; t <= pos; ++t) {
        double d = 0;
        for (int i = 0; i < hs; ++i) d += (double)q[h * hs + i] * kc[loff + (size_t)t * kv_dim + hoff + i];
        p[t] = d / std::sqrt((double)hs);
        mx = std::max(mx, p[t]);
      }
      for (auto& v : p) {
        v = std::exp(v - mx);
        sum += v;
      }
      for (int t = 0; t <= pos; ++t) {
        p[t] /= sum;
        KH_REQUIRE(std::fabs(hsc[(size_t)h * seq + t] - p[t]) < 1e-5);
      }
      for (int i = 0; i < hs; ++i) {
        double o = 0;
        for (int t = 0; t <= pos; ++t) o += p[t] * vc[loff + (size_t)t * kv_dim + hoff + i];
        KH_REQUIRE(std::fabs(ho[h * hs + i] - o) < 1e-5);
      }
    }
    ++covered;
  }
  {  // 11.-13. ScaleKernel, SoftmaxInplaceKernel, ScaleSumKernel (scale/softmax/scale_sum_kernel.cpp)
    Tensor xd = Tr::dev_f32(xs, {n});
    K::get_scale_kernel()(0.25f, xd, stream);
    auto h = Tr::to_host(xd);
    for (int i = 0; i < n; ++i) KH_REQUIRE(h[i] == 0.25f * xs[i]);
    ++covered;
    Tensor sd = Tr::dev_f32(xs, {n});
    K::get_softmax_kernel()(sd, stream);
    h = Tr::to_host(sd);
    double mx = -1e30, sum = 0;
    for (float v : xs) mx = std::max(mx, (double)v);
    for (float v : xs) sum += std::exp(v - mx);
    for (int i = 0; i < n; ++i) KH_REQUIRE(std::fabs(h[i] - std::exp(xs[i] - mx) / sum) < 1e-6);
    ++covered;
    const int t = 6, size = 64, stride = 96;
    std::vector<float> val((size_t)(t + 1) * stride), sc(t + 1);
    Lcg r(5);
    for (auto& v : val) v = r.next();
    for (auto& v : sc) v = 0.1f + 0.05f * (r.next() + 1.f);
    Tensor vd = Tr::dev_f32(val, {(int32_t)val.size()}), scd = Tr::dev_f32(sc, {t + 1});
    Tensor od = Tr::dev_f32(std::vector<float>(size, 0.f), {size});
    K::get_scale_sum_kernel()(vd, scd, od, t, size, stride, stream);
    h = Tr::to_host(od);
    for (int i = 0; i < size; ++i) {
      double o = 0;
      for (int j = 0; j <= t; ++j) o += (double)sc[j] * val[(size_t)j * stride + i];
      KH_REQUIRE(std::fabs(h[i] - o) < 1e-5);
    }
    ++covered;
  }
  {  // error routing: an invalid argument reaches the handler instead of aborting silently
    static int seen = 0;
    kuiper_hip::error_handler() = [](int code, const char*) { seen = code; };
    Tensor empty = Tr::null_f32(4);
    K::get_add_kernel()(empty, empty, empty, stream);
    KH_REQUIRE(seen == KH_ERR_INVALID_ARG);
    kuiper_hip::error_handler() = kuiper_hip::default_error_handler;
  }
  std::printf("OK %d/13 operator entry points exercised through the adapter getters\n", covered);
  return covered == 13 ? 0 : 1;
}

}  // namespace adapter_cases
