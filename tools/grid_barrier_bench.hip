// Micro-benchmark: cost of one grid-wide barrier among co-resident workgroups on MI355X (8 XCDs, private L2s).
// Decides the form of the persistent small-system step kernel (engine.hip: nepmi_tersoff_steps).
//   A  one counter: atomicAdd (agent scope) + poll
//   B  arrival words, one per workgroup, gathered by workgroup 0, which publishes the generation
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_bench grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void barrier_a(unsigned* ctr, unsigned nblk, unsigned& gen)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    ++gen;
    __atomic_thread_fence(__ATOMIC_RELEASE); // (HIP: agent scope by default for __threadfence-like use below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * nblk;
    while (ld_agent(ctr) < target)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ void barrier_b(unsigned* arrive, unsigned* go, unsigned nblk, unsigned& gen)
{
  __syncthreads();
  ++gen;
  if (blockIdx.x == 0) {
    // every thread of workgroup 0 waits for its share of the arrival words
    for (unsigned w = threadIdx.x + 1; w < nblk; w += blockDim.x)
      while (ld_agent(arrive + w) < gen)
        __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
      __hip_atomic_store(go, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(arrive + blockIdx.x, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (ld_agent(go) < gen)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(256) bench(unsigned* ctr, unsigned* arrive, unsigned* go, double* data, int nbar, int check)
{
  unsigned gen = 0;
  const unsigned nblk = gridDim.x;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)nblk * blockDim.x;
  for (int it = 0; it < nbar; ++it) {
    // some traffic that has to cross XCDs: write own element, after the barrier read the element of a far thread
    data[gid] = (double)(it + 1);
    if (MODE == 0)
      barrier_a(ctr, nblk, gen);
    else
      barrier_b(arrive, go, nblk, gen);
    const size_t other = (gid + nth / 2 + 7) % nth;
    const double v = __builtin_nontemporal_load(data + other);
    if (check && v != (double)(it + 1))
      atomicAdd(ctr + 1, 1u);
    if (MODE == 0)
      barrier_a(ctr, nblk, gen);
    else
      barrier_b(arrive, go, nblk, gen);
  }
}

int main()
{
  unsigned *ctr, *arrive, *go;
  double* data;
  CK(hipMalloc(&ctr, 64));
  CK(hipMalloc(&arrive, 4096 * 4));
  CK(hipMalloc(&go, 64));
  CK(hipMalloc(&data, 4096 * 256 * 8));
  const int grids[] = {64, 216, 432, 864};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int g : grids) {
      const int nbar = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctr, 0, 64));
        CK(hipMemset(arrive, 0, 4096 * 4));
        CK(hipMemset(go, 0, 64));
        int check = 1;
        int nb = nbar;
        void* args[] = {&ctr, &arrive, &go, &data, &nb, &check};
        CK(hipEventRecord(e0, 0));
        if (mode == 0)
          CK(hipLaunchCooperativeKernel((const void*)bench<0>, dim3(g), dim3(256), args, 0, 0));
        else
          CK(hipLaunchCooperativeKernel((const void*)bench<1>, dim3(g), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2];
        CK(hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost));
        if (rep == 1)
          printf("mode %c grid %4d: %.3f us per barrier (%d barriers, stale reads %u)\n", mode ? 'B' : 'A', g, 1000.0 * ms / (2.0 * nbar), 2 * nbar, h[1]);
      }
    }
  return 0;
}
