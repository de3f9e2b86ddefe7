// TEST-ONLY stand-in for the CUDA runtime headers the reference's host sources include, so that the reference
// translation units that cannot be edited from this repo (tensor.cpp, alloc.cpp, alloc_cu.cpp, cuda_config.h, the
// op/*.cpp layers) COMPILE.  The calls are forwarded to the HIP runtime.  In test_ref_binding / test_ref_layers
// nothing allocates, copies or clears memory through them: device tensors come from include/kuiper_hip_alloc.hpp
// (HipDeviceAllocator, tagged kDeviceHIP), every memory call below counts itself in refstub::mem_calls() and those
// two tests require the counter to stay 0 (only the stream destructor of kernel::CudaConfig, a reference type the
// kernel typedefs name, passes through here).  test_ref_model[_qwen2] is different: the reference's model code
// stamps kDeviceCUDA and takes its memory from its own CUDADeviceAllocator (llama3.cpp:117-125, 425-500), i.e.
// through the forwards below - MI355X memory under the reference's own tag; the counter is not checked there.
// Not part of the product (which has no CUDA spelling anywhere).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstring>  // the real header brings it in; base/alloc.cpp calls std::memcpy / std::memset
typedef ihipStream_t CUstream_st;  // alloc.cpp:14 spells the stream struct
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
constexpr hipError_t cudaSuccess = hipSuccess;
enum cudaMemcpyKind {
  cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2,
  cudaMemcpyDeviceToDevice = 3
};
inline hipMemcpyKind refstub_kind(cudaMemcpyKind k) { return (hipMemcpyKind)(int)k; }
namespace refstub {
inline int& mem_calls() {  // one counter for the whole program (inline function, static local)
  static int n = 0;
  return n;
}
}  // namespace refstub
inline cudaError_t cudaMalloc(void** p, size_t n) { ++refstub::mem_calls(); return hipMalloc(p, n); }
inline cudaError_t cudaFree(void* p) { ++refstub::mem_calls(); return hipFree(p); }
inline cudaError_t cudaGetDevice(int* d) { return hipGetDevice(d); }
inline cudaError_t cudaSetDevice(int d) { return hipSetDevice(d); }
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind k) { ++refstub::mem_calls(); return hipMemcpy(d, s, n, refstub_kind(k)); }
inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind k, cudaStream_t st = nullptr) { ++refstub::mem_calls(); return hipMemcpyAsync(d, s, n, refstub_kind(k), st); }
inline cudaError_t cudaMemset(void* p, int v, size_t n) { ++refstub::mem_calls(); return hipMemset(p, v, n); }
inline cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t st = nullptr) { ++refstub::mem_calls(); return hipMemsetAsync(p, v, n, st); }
inline cudaError_t cudaDeviceSynchronize() { return hipDeviceSynchronize(); }
inline cudaError_t cudaStreamDestroy(cudaStream_t s) { return hipStreamDestroy(s); }
inline cudaError_t cudaStreamCreate(cudaStream_t* s) { return hipStreamCreate(s); }
inline cudaError_t cudaGetLastError() { return hipGetLastError(); }
