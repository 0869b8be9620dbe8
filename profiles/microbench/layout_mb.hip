// micro-benchmark: [row][N] vs chunked row layouts, one lane per atom, R rows per atom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, int WR>
__global__ void __launch_bounds__(64) k(float* a, float* out, int64_t N, int R)
{
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f;
  for (int d = 0; d < R; ++d) {
    int64_t idx;
    if (MODE == 0) idx = (int64_t)d * N + i;
    else if (MODE == 1) idx = (i >> 10) * R * 1024 + (int64_t)d * 1024 + (i & 1023);
    else idx = (i >> 6) * R * 64 + (int64_t)d * 64 + (i & 63);
    if (WR) a[idx] = (float)d; else acc += a[idx];
  }
  if (!WR) out[i] = acc;
}
template <int MODE, int WR>
float run(float* a, float* out, int64_t N, int R)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, WR>), dim3((N + 63) / 64), dim3(64), 0, 0, a, out, N, R);
  hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<MODE, WR>), dim3((N + 63) / 64), dim3(64), 0, 0, a, out, N, R);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main()
{
  const int64_t N = 1024000;
  for (int R : {16, 42, 66, 128}) {
    float *a, *out; hipMalloc(&a, sizeof(float) * N * R); hipMalloc(&out, sizeof(float) * N);
    hipMemset(a, 0, sizeof(float) * N * R);
    float r0 = run<0, 0>(a, out, N, R), r1 = run<1, 0>(a, out, N, R), r2 = run<2, 0>(a, out, N, R);
    float w0 = run<0, 1>(a, out, N, R), w1 = run<1, 1>(a, out, N, R), w2 = run<2, 1>(a, out, N, R);
    double gb = (double)N * R * 4 / 1e9;
    printf("R=%3d  read ms [row][N] %.3f  chunk1024 %.3f  chunk64 %.3f | write %.3f %.3f %.3f   (%.2f GB -> ideal %.3f ms @8TB/s)\n",
           R, r0, r1, r2, w0, w1, w2, gb, gb / 8000 * 1000);
    hipFree(a); hipFree(out);
  }
  return 0;
}
