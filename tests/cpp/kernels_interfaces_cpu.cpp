// kernels_interfaces_cpu.cpp — the CPU branches of the reference's kernel::get_*_kernel getters
// (kuiper/source/op/kernels/kernels_interfaces.cpp:21-132), for oracle/_ref/ref_cpu_model*: the reference's OWN CPU
// kernels (kuiper/source/op/kernels/cpu/*.cpp, compiled where they lie over tests/cpp/ref_stubs/armadillo) under the
// reference's OWN op::*Layer and model classes = the reference CPU backend as a checker and a timed baseline.  The
// reference's own getters file cannot be compiled here: it also names the CUDA kernels.  Test infrastructure.
#include "kernels_interface.h"  // the reference's: -I$KUIPER_REF/kuiper/source/op/kernels

#include <glog/logging.h>

#include "cpu/add_kernel.h"
#include "cpu/emb_kernel.h"
#include "cpu/matmul_kernel.h"
#include "cpu/mha_kernel.h"
#include "cpu/rmsnorm_kernel.h"
#include "cpu/rope_kernel.h"
#include "cpu/scale_kernel.h"
#include "cpu/scale_sum_kernel.h"
#include "cpu/softmax_kernel.h"
#include "cpu/swiglu_kernel.h"

namespace kernel {
#define KH_CPU_GETTER(TYPE, NAME, ARGS, FN)                                            \
  TYPE NAME ARGS {                                                                     \
    if (device_type == base::DeviceType::kDeviceCPU) return FN;                        \
    LOG(FATAL) << #NAME ": only the CPU backend is linked into this binary";           \
    return nullptr;                                                                    \
  }
KH_CPU_GETTER(AddKernel, get_add_kernel, (base::DeviceType device_type), add_kernel_cpu)
KH_CPU_GETTER(EmbeddingKernel, get_emb_kernel, (base::DeviceType device_type), emb_kernel_normal)
KH_CPU_GETTER(MatmulKernel, get_matmul_kernel, (base::DeviceType device_type), matmul_kernel_cpu)
KH_CPU_GETTER(MHAKernel, get_mha_kernel, (base::DeviceType device_type), mha_kernel)
KH_CPU_GETTER(RoPEKernel, get_rope_kernel, (base::DeviceType device_type), rope_kernel_cpu)
KH_CPU_GETTER(ScaleKernel, get_scale_kernel, (base::DeviceType device_type), scale_inplace_cpu)
KH_CPU_GETTER(SoftmaxInplaceKernel, get_softmax_kernel, (base::DeviceType device_type), softmax_inplace_cpu)
KH_CPU_GETTER(SwigluKernel, get_swiglu_kernel, (base::DeviceType device_type, void* /*stream*/), swiglu_kernel_cpu)
KH_CPU_GETTER(RMSNormKernel, get_rmsnorm_kernel, (base::DeviceType device_type), rmsnorm_kernel_cpu)
KH_CPU_GETTER(ScaleSumKernel, get_scale_sum_kernel, (base::DeviceType device_type), scale_sum_kernel_cpu)
#undef KH_CPU_GETTER
// the reference has no CPU int8 matmul (kernels_interfaces.cpp:54-61)
MatmulKernelQuant get_matmul_kernel_quant8(base::DeviceType) {
  LOG(FATAL) << "the reference has no CPU int8 matmul";
  return nullptr;
}
}  // namespace kernel
