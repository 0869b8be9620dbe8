// Issue rate of v_pk_fma_f32 vs v_fma_f32 vs v_pk_mul/add on gfx950 (one wavefront-instruction = 64 lanes).
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate_mb && ./pk_rate_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
  float s0 = threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
  f2 p0 = {s0, s1}, p1 = {s2, s3}, p2 = {s4, s5}, p3 = {s6, s7}, p4 = {s1, s0}, p5 = {s3, s2}, p6 = {s5, s4}, p7 = {s7, s6};
  const f2 A = {a, a}, B = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { // 8 scalar fma
      s0 = fmaf(s0, a, b); s1 = fmaf(s1, a, b); s2 = fmaf(s2, a, b); s3 = fmaf(s3, a, b);
      s4 = fmaf(s4, a, b); s5 = fmaf(s5, a, b); s6 = fmaf(s6, a, b); s7 = fmaf(s7, a, b);
    } else if (MODE == 1) { // 8 packed fma (16 flops-pairs)
      p0 = __builtin_elementwise_fma(p0, A, B); p1 = __builtin_elementwise_fma(p1, A, B);
      p2 = __builtin_elementwise_fma(p2, A, B); p3 = __builtin_elementwise_fma(p3, A, B);
      p4 = __builtin_elementwise_fma(p4, A, B); p5 = __builtin_elementwise_fma(p5, A, B);
      p6 = __builtin_elementwise_fma(p6, A, B); p7 = __builtin_elementwise_fma(p7, A, B);
    } else if (MODE == 2) { // 8 packed mul
      p0 = p0 * A; p1 = p1 * A; p2 = p2 * A; p3 = p3 * A; p4 = p4 * A; p5 = p5 * A; p6 = p6 * A; p7 = p7 * A;
    } else if (MODE == 3) { // 8 v_rsq
      s0 = __frsqrt_rn(s0); s1 = __frsqrt_rn(s1); s2 = __frsqrt_rn(s2); s3 = __frsqrt_rn(s3);
      s4 = __frsqrt_rn(s4); s5 = __frsqrt_rn(s5); s6 = __frsqrt_rn(s6); s7 = __frsqrt_rn(s7);
    }
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
double run(float* d, int blocks, int iters)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  const int blocks = 256 * 8, iters = 20000; // 8 workgroups per CU: 8 waves per SIMD
  float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
  const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_rsq_f32"};
  double ms[4] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters)};
  for (int m = 0; m < 4; ++m) {
    const double winst = (double)blocks * 4 * iters * 8; // wavefront-instructions
    const double per_simd_per_s = winst / (ms[m] * 1e-3) / 1024.0;
    printf("%-14s %8.3f ms  %.3e wave-instr/s/SIMD  (cycles per instr at 2.4 GHz: %.2f)\n", names[m], ms[m], per_simd_per_s,
           2.4e9 / per_simd_per_s);
  }
  return 0;
}
