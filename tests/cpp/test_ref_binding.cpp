// test_ref_binding.cpp — the drop-in boundary bound to the reference's REAL types.
//
// This TU includes KuiperLLama's own headers
//     kuiper/source/op/kernels/kernels_interface.h   (the eleven kernel typedefs + getters, :6-68)
//     kuiper/include/tensor/tensor.h, base/base.h, base/cuda_config.h
// (found under $KUIPER_REF = /root/reference; third-party includes resolved by the test-only
// stand-ins in tests/cpp/ref_stubs/) and
//   1. static_asserts that every member of kuiper_hip::Kernels<tensor::Tensor, kernel::CudaConfig,
//      base::DeviceType> has EXACTLY the type of the matching reference typedef, and that the
//      adapter's getters return those typedefs - i.e. `return HipK::get_add_kernel();` compiles
//      inside kernel::get_add_kernel (kernels_interfaces.cpp:21-132) with no cast or lambda;
//   2. links the reference's own tensor / buffer / allocator sources (compiled where they lie by
//      oracle/Makefile, never copied) and drives real tensor::Tensor objects - device memory from
//      include/kuiper_hip_alloc.hpp's HipDeviceAllocator, tagged kDeviceHIP - through all thirteen
//      operator entry points on the GPU (tests/cpp/adapter_cases.hpp);
//   3. checks the allocator / Tensor::to_cuda / to_cpu / CudaConfig twins of kuiper_hip_alloc.hpp and that
//      no memory call reached the test-only CUDA stand-in.
// Built into oracle/_ref/test_ref_binding (reference-derived binaries live there and stay out of
// git); /root/reference does not exist on the GPU box, the prebuilt binary travels.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <vector>

#include "kernels_interface.h"  // the reference's, via -I$KUIPER_REF/kuiper/source/op/kernels

#include "adapter_cases.hpp"
#include "kuiper_hip_alloc.hpp"

using HipK = kuiper_hip::Kernels<tensor::Tensor, kernel::CudaConfig, base::DeviceType>;

// 1a. the adapter's spelled-out typedefs ARE the reference's (kernels_interface.h:6-44)
static_assert(std::is_same_v<HipK::AddKernel, kernel::AddKernel>);
static_assert(std::is_same_v<HipK::MatmulKernel, kernel::MatmulKernel>);
static_assert(std::is_same_v<HipK::MatmulKernelQuant, kernel::MatmulKernelQuant>);
static_assert(std::is_same_v<HipK::EmbeddingKernel, kernel::EmbeddingKernel>);
static_assert(std::is_same_v<HipK::SwigluKernel, kernel::SwigluKernel>);
static_assert(std::is_same_v<HipK::MHAKernel, kernel::MHAKernel>);
static_assert(std::is_same_v<HipK::RMSNormKernel, kernel::RMSNormKernel>);
static_assert(std::is_same_v<HipK::RoPEKernel, kernel::RoPEKernel>);
static_assert(std::is_same_v<HipK::ScaleKernel, kernel::ScaleKernel>);
static_assert(std::is_same_v<HipK::SoftmaxInplaceKernel, kernel::SoftmaxInplaceKernel>);
static_assert(std::is_same_v<HipK::ScaleSumKernel, kernel::ScaleSumKernel>);
// 1b. the functions themselves convert to them with no adaptor
static_assert(std::is_convertible_v<decltype(&HipK::add), kernel::AddKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::matmul), kernel::MatmulKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::matmul_quant8), kernel::MatmulKernelQuant>);
static_assert(std::is_convertible_v<decltype(&HipK::embedding), kernel::EmbeddingKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::swiglu), kernel::SwigluKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::mha), kernel::MHAKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::rmsnorm), kernel::RMSNormKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::rope), kernel::RoPEKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::scale), kernel::ScaleKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::softmax_inplace), kernel::SoftmaxInplaceKernel>);
static_assert(std::is_convertible_v<decltype(&HipK::scale_sum), kernel::ScaleSumKernel>);
// 1c. sin_cos_cache_calc_cu / argmax_kernel_cu shapes (cuda/rope_kernel.cuh:9-10, argmax_kernel.cuh:4)
using SinCosFn = void (*)(int, int, const tensor::Tensor&, const tensor::Tensor&, cudaStream_t);
using ArgmaxFn = size_t (*)(const float*, size_t, void*);
static_assert(std::is_convertible_v<decltype(&HipK::sin_cos_cache_calc<cudaStream_t>), SinCosFn>);
static_assert(std::is_convertible_v<decltype(&HipK::argmax), ArgmaxFn>);

// 1d. the getters as a maintainer would write them: a kDeviceHIP branch in each kernel::get_*_kernel.
// kDeviceHIP is the enumerator INTEGRATION.md adds to base::DeviceType (value 3).
namespace hip_branch {
constexpr base::DeviceType kDeviceHIP = static_cast<base::DeviceType>(3);
kernel::AddKernel get_add_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_add_kernel() : nullptr; }
kernel::EmbeddingKernel get_emb_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_emb_kernel() : nullptr; }
kernel::MatmulKernel get_matmul_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_matmul_kernel() : nullptr; }
kernel::MatmulKernelQuant get_matmul_kernel_quant8(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_matmul_kernel_quant8() : nullptr; }
kernel::MHAKernel get_mha_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_mha_kernel() : nullptr; }
kernel::RMSNormKernel get_rmsnorm_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_rmsnorm_kernel() : nullptr; }
kernel::RoPEKernel get_rope_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_rope_kernel() : nullptr; }
kernel::ScaleKernel get_scale_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_scale_kernel() : nullptr; }
kernel::SoftmaxInplaceKernel get_softmax_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_softmax_kernel() : nullptr; }
kernel::SwigluKernel get_swiglu_kernel(base::DeviceType d, void* = nullptr) { return d == kDeviceHIP ? HipK::get_swiglu_kernel() : nullptr; }
kernel::ScaleSumKernel get_scale_sum_kernel(base::DeviceType d) { return d == kDeviceHIP ? HipK::get_scale_sum_kernel() : nullptr; }
}  // namespace hip_branch

// 2. real tensor::Tensor objects: host tensors from the reference's CPUDeviceAllocator, device tensors from
// include/kuiper_hip_alloc.hpp's HipDeviceAllocator (a base::DeviceAllocator tagged kDeviceHIP)
using HipAllocator = kuiper_hip::HipDeviceAllocator<base::DeviceAllocator, base::DeviceType, base::MemcpyKind,
                                                    hip_branch::kDeviceHIP>;
static std::shared_ptr<HipAllocator> hip_alloc() { return kuiper_hip::allocator_instance<HipAllocator>(); }
struct RefTensors {
  using Tensor = tensor::Tensor;
  using Config = kernel::CudaConfig;
  using DeviceType = base::DeviceType;
  static Tensor make(base::DataType dt, const std::vector<int32_t>& dims, bool on_device) {
    std::shared_ptr<base::DeviceAllocator> alloc;
    if (on_device) alloc = hip_alloc();
    else alloc = base::CPUDeviceAllocatorFactory::get_instance();
    return Tensor(dt, dims, /*need_alloc=*/true, alloc);
  }
  static Tensor dev_f32(const std::vector<float>& h, std::vector<int32_t> dims) {
    Tensor t = make(base::DataType::kDataTypeFp32, dims, true);
    hip_alloc()->memcpy(h.data(), t.ptr<float>(), h.size() * 4, base::MemcpyKind::kMemcpyCPU2CUDA);
    return t;
  }
  static Tensor dev_i8(const std::vector<int8_t>& h, std::vector<int32_t> dims) {
    Tensor t = make(base::DataType::kDataTypeInt8, dims, true);
    hip_alloc()->memcpy(h.data(), t.ptr<int8_t>(), h.size(), base::MemcpyKind::kMemcpyCPU2CUDA);
    return t;
  }
  static Tensor host_i32(const std::vector<int32_t>& h, std::vector<int32_t> dims) {
    Tensor t = make(base::DataType::kDataTypeInt32, dims, false);
    for (size_t i = 0; i < h.size(); ++i) t.index<int32_t>((int64_t)i) = h[i];
    return t;
  }
  static Tensor null_f32(int32_t n) { return Tensor(base::DataType::kDataTypeFp32, n); }  // no buffer
  static std::vector<float> to_host(const Tensor& t) {
    (void)hipDeviceSynchronize();
    Tensor c = t;  // shares the buffer
    if (!kuiper_hip::to_host(c, hip_alloc(), base::CPUDeviceAllocatorFactory::get_instance())) return {};
    return std::vector<float>(c.ptr<float>(), c.ptr<float>() + c.size());
  }
  static DeviceType device() { return hip_branch::kDeviceHIP; }
  static void set_stream(Config& c, void* s) { c.stream = (cudaStream_t)s; }
};

int main() {
  // every getter of the new branch hands back the adapter function
  if (hip_branch::get_add_kernel(hip_branch::kDeviceHIP) != &HipK::add ||
      hip_branch::get_mha_kernel(hip_branch::kDeviceHIP) != &HipK::mha ||
      hip_branch::get_matmul_kernel_quant8(hip_branch::kDeviceHIP) != &HipK::matmul_quant8 ||
      hip_branch::get_emb_kernel(base::DeviceType::kDeviceCPU) != nullptr) {
    std::printf("FAIL getter wiring\n");
    return 1;
  }
  int ndev = kh_device_count();
  if (ndev <= 0) {
    std::printf("SKIP: no HIP device; the reference typedefs bind (static_asserts passed at build time)\n");
    return 77;
  }
  // ---- the allocator twin on its own (alloc.h:14-93, alloc_cu.cpp:7-112, tensor.cpp:104-137) ----
  {
    auto al = hip_alloc();
    auto& pool = kuiper_hip::HipMemoryPool::instance();
    if (al->base::DeviceAllocator::device_type() != hip_branch::kDeviceHIP) { std::printf("FAIL allocator tag\n"); return 1; }
    void* a = al->allocate(4096);
    void* big = al->allocate(3u << 20);
    if (!a || !big) { std::printf("FAIL allocate\n"); return 1; }
    al->release(a);
    al->release(big);
    void* a2 = al->allocate(3000);          // a small request takes the parked block of at most twice its size
    void* a3 = al->allocate(100);           // ... but not a block 40x too large
    void* big2 = al->allocate((3u << 20) - 4096);  // a big one the parked block with < 1 MiB of slack
    void* big3 = al->allocate(1500000);     // 1.5 MB of slack: a new block
    if (a2 != a || a3 == a || big2 != big || big3 == big) { std::printf("FAIL pool reuse policy\n"); return 1; }
    al->release(a2); al->release(a3); al->release(big2); al->release(big3);
    if (pool.stats().busy_blocks != 0) { std::printf("FAIL pool accounting\n"); return 1; }
    // Tensor::to_cuda / to_cpu twins on a real tensor::Tensor, asynchronous on a stream of the config twin
    auto cfg = kuiper_hip::make_stream_config(0);
    if (!cfg || !cfg->stream) { std::printf("FAIL make_stream_config\n"); return 1; }
    tensor::Tensor t(base::DataType::kDataTypeFp32, 1000, true, base::CPUDeviceAllocatorFactory::get_instance());
    for (int i = 0; i < 1000; ++i) t.index<float>(i) = 0.5f * (float)i;
    if (!kuiper_hip::to_hip(t, al, base::DeviceType::kDeviceCPU, cfg->stream) || t.device_type() != hip_branch::kDeviceHIP) {
      std::printf("FAIL to_hip\n");
      return 1;
    }
    (void)hipStreamSynchronize(cfg->stream);
    al->memset_zero(t.ptr<float>() + 10, 40, cfg->stream, /*need_sync=*/true);
    if (!kuiper_hip::to_host(t, al, base::CPUDeviceAllocatorFactory::get_instance()) ||
        t.device_type() != base::DeviceType::kDeviceCPU) {
      std::printf("FAIL to_host\n");
      return 1;
    }
    for (int i = 0; i < 1000; ++i)
      if (t.index<float>(i) != ((i >= 10 && i < 20) ? 0.f : 0.5f * (float)i)) { std::printf("FAIL round trip at %d\n", i); return 1; }
    // base::Buffer::copy_from device <- host through the virtual memcpy (buffer.cpp:60-85)
    tensor::Tensor d(base::DataType::kDataTypeFp32, 1000, true, al);
    d.get_buffer()->copy_from(t.get_buffer().get());
    tensor::Tensor back = d;
    if (!kuiper_hip::to_host(back, al, base::CPUDeviceAllocatorFactory::get_instance()) || back.index<float>(999) != 499.5f) {
      std::printf("FAIL Buffer::copy_from through HipDeviceAllocator::memcpy\n");
      return 1;
    }
  }
  hipStream_t stream;
  if (hipStreamCreate(&stream) != hipSuccess) return 1;
  int rc;
  {
    rc = adapter_cases::run<RefTensors>((void*)stream);  // Config's destructor destroys the stream
  }
  if (rc == 0 && refstub::mem_calls() != 0) {
    std::printf("FAIL %d memory calls went through the CUDA stand-in\n", refstub::mem_calls());
    rc = 1;
  }
  if (rc == 0)
    std::printf("OK reference tensor::Tensor (tagged kDeviceHIP, from HipDeviceAllocator) + kernels_interface.h typedefs bound to "
                "libkuiper_hip.so; allocator / to_hip / to_host / HipStreamConfig twins checked; 0 memory calls into the CUDA stand-in\n");
  return rc;
}
