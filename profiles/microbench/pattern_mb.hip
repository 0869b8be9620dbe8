// micro-benchmark: cost of wave64 dword access patterns over a [row][N] array (N*R accesses each).
//   P0 contiguous            tile t: lane l -> column 64 t + l, rows 0..R-1
//   P1 half-wave stride-2    unit u (32 columns 64 (u>>1) + 2 j + (u&1)): lanes 0-31 row 2s, lanes 32-63 row 2s+1
//   P2 two interleaved runs  tile t: lane l -> column 1024 (t>>4) + 512 (l&1) + 32 (t&15) + (l>>1)
//   P3 full-wave stride-2    tile t: lane l -> column 128 (t>>1) + 2 l + (t&1)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int P, int WR>
__global__ void __launch_bounds__(64) k(float* a, float* out, int64_t N, int R)
{
  const int l = threadIdx.x;
  const int64_t t = blockIdx.x;
  int64_t x;
  if (P == 0) x = t * 64 + l;
  else if (P == 1) x = 64 * (t >> 1) + 2 * (l & 31) + (t & 1);
  else if (P == 2) x = 1024 * (t >> 4) + 512 * (l & 1) + 32 * (t & 15) + (l >> 1);
  else x = 128 * (t >> 1) + 2 * l + (t & 1);
  if (x >= N) return;
  float acc = 0.f;
  if (P == 1) {
    for (int s = 0; s < R / 2; ++s) {
      const int64_t idx = (int64_t)(2 * s + (l >> 5)) * N + x;
      if (WR) a[idx] = (float)s; else acc += a[idx];
    }
  } else {
    for (int d = 0; d < R; ++d) {
      const int64_t idx = (int64_t)d * N + x;
      if (WR) a[idx] = (float)d; else acc += a[idx];
    }
  }
  if (!WR) out[t * 64 + l] = acc;
}
template <int P, int WR>
float run(float* a, float* out, int64_t N, int R)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int64_t tiles = P == 1 ? N / 32 : N / 64;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<P, WR>), dim3(tiles), dim3(64), 0, 0, a, out, N, R);
  hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<P, WR>), dim3(tiles), dim3(64), 0, 0, a, out, N, R);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main()
{
  const int64_t N = 1024000;
  for (int R : {40, 64}) {
    float *a, *out; hipMalloc(&a, sizeof(float) * N * R); hipMalloc(&out, sizeof(float) * N * 2);
    hipMemset(a, 0, sizeof(float) * N * R);
    printf("R=%d read  P0 %.3f  P1 %.3f  P2 %.3f  P3 %.3f ms\n", R, run<0, 0>(a, out, N, R), run<1, 0>(a, out, N, R), run<2, 0>(a, out, N, R), run<3, 0>(a, out, N, R));
    printf("R=%d write P0 %.3f  P1 %.3f  P2 %.3f  P3 %.3f ms\n", R, run<0, 1>(a, out, N, R), run<1, 1>(a, out, N, R), run<2, 1>(a, out, N, R), run<3, 1>(a, out, N, R));
    hipFree(a); hipFree(out);
  }
  return 0;
}
