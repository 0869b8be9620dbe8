// TEST INFRASTRUCTURE ONLY: a host stand-in for the few HIP runtime calls gpumd_amd/host uses, so that the C++ host
// (run.in / model.xyz parsing, the run loop, the multi-rank driver, the output files) can be exercised on the CPU test
// tier against the kernel-logic emulator library (tests/emu/libnepmi_emu.so), whose "device pointers" are host
// pointers.  Never part of the product: gpumd_amd/host/Makefile builds gpumd-mi against the real HIP runtime.
#pragma once
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct hipDeviceProp_t {
  char name[64] = "kernel-logic emulator";
  char gcnArchName[64] = "host";
  int multiProcessorCount = 0;
  size_t totalGlobalMem = 0;
};
inline const char* hipGetErrorString(hipError_t) { return "emulated HIP call failed"; }
inline hipError_t hipMalloc(void** p, size_t n)
{
  *p = std::calloc(1, n ? n : 1);
  return *p ? hipSuccess : 1;
}
inline hipError_t hipFree(void* p)
{
  std::free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n)
{
  std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n)
{
  *n = 1;
  return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
  *p = hipDeviceProp_t();
  return hipSuccess;
}
