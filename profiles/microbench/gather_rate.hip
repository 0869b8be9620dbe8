// Throughput of per-lane random 16-byte reads on gfx950: global (L1/L2-resident table rows, as the force assembly gathers the
// neighbours' radial-table rows) against LDS (ds_read_b128 at random 16-byte slots, as the window records are read).
// One number per mode: CU cycles per wavefront-instruction with every CU busy (4 workgroups of 256 threads per CU).
// hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate_mb && ./gather_rate_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct alignas(16) R4 { float x, y, z, w; };

// MODE 0: one 16-B global gather per iteration, rows of 16 B;  1: two 16-B gathers of one 32-B row;  2: one 4-B gather;
// MODE 3: one ds_read_b128 at a random slot of a 24 KB LDS array;  4: two ds_read_b128 of one 32-B row (48 KB);
// MODE 5: coalesced 16-B global loads (lane l reads row base + l)
template <int MODE>
__global__ void __launch_bounds__(256) k(const R4* __restrict__ tab, float* out, unsigned rows_mask, unsigned region_rows, int iters)
{
  __shared__ R4 lds[3072];
  for (int i = threadIdx.x; i < 3072; i += 256)
    lds[i] = R4{(float)i, 1.0f, 2.0f, 3.0f};
  __syncthreads();
  // every workgroup works in its own region of the table (a brick's window), lanes pick random rows inside it
  const R4* base = tab + (size_t)((blockIdx.x * 2654435761u) & rows_mask & ~(region_rows - 1));
  unsigned s = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned r = (s >> 9) & (region_rows - 1);
    if (MODE == 0) {
      const R4 v = base[r];
      a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    } else if (MODE == 1) {
      const R4 v = base[r & ~1u], w = base[(r & ~1u) + 1];
      a0 += v.x + w.x; a1 += v.y + w.y; a2 += v.z + w.z; a3 += v.w + w.w;
    } else if (MODE == 2) {
      a0 += reinterpret_cast<const float*>(base)[r * 4];
    } else if (MODE == 3) {
      const R4 v = lds[r % 1536];
      a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    } else if (MODE == 4) {
      const unsigned q = (r % 1536) * 2;
      const R4 v = lds[q], w = lds[q + 1];
      a0 += v.x + w.x; a1 += v.y + w.y; a2 += v.z + w.z; a3 += v.w + w.w;
    } else {
      const R4 v = base[((r & ~63u) + (threadIdx.x & 63)) & (region_rows - 1)];
      a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int MODE>
double run(const R4* tab, float* out, unsigned rows_mask, unsigned region_rows, int blocks, int iters)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, out, rows_mask, region_rows, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, out, rows_mask, region_rows, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  const int blocks = 256 * 4, iters = 4096;
  const size_t rows = (size_t)1 << 22; // 64 MB of 16-B rows
  R4* tab; float* out;
  hipMalloc(&tab, rows * sizeof(R4)); hipMemset(tab, 0, rows * sizeof(R4));
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const char* names[6] = {"global 16 B gather", "global 2 x 16 B (32-B row)", "global 4 B gather", "LDS 16 B random", "LDS 2 x 16 B (32-B row)",
                          "global 16 B coalesced"};
  // region = the rows one workgroup draws from: 4096 rows = 64 KB (a brick window's table rows), 65536 rows = 1 MB
  for (unsigned region : {4096u, 65536u}) {
    printf("region %u rows (%u KB) per workgroup, table 64 MB\n", region, region * 16 / 1024);
    double ms[6] = {run<0>(tab, out, rows - 1, region, blocks, iters), run<1>(tab, out, rows - 1, region, blocks, iters),
                    run<2>(tab, out, rows - 1, region, blocks, iters), run<3>(tab, out, rows - 1, region, blocks, iters),
                    run<4>(tab, out, rows - 1, region, blocks, iters), run<5>(tab, out, rows - 1, region, blocks, iters)};
    for (int m = 0; m < 6; ++m) {
      const double winst_per_cu = (double)blocks / 256.0 * 4 * iters * ((m == 1 || m == 4) ? 2 : 1); // wave-instructions per CU
      const double cyc = ms[m] * 1e-3 * 2.4e9 / winst_per_cu;
      printf("  %-28s %8.3f ms   %.1f CU cycles per wave-instruction (2.4 GHz)   %.2f TB/s useful\n", names[m], ms[m], cyc,
             (double)blocks * 256 * iters * ((m == 1 || m == 4) ? 32 : (m == 2 ? 4 : 16)) / (ms[m] * 1e-3) / 1e12);
    }
  }
  return 0;
}
