// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
//
// The reference's own Langevin kernels (src/integrate/langevin_utilities.cuh: initialize_curand_states, gpu_langevin,
// gpu_find_momentum, gpu_correct_momentum), compiled for gfx950 from the header where it lies, behind three C entry
// points on device pointers.  tests/test_langevin.py (GPU tier) runs integrate_nvt_lan_half's kernel sequence
// (ensemble_lan.cu:96-127) through them and through libnepmi's nepmi_lan_half_step with the same seed and expects
// the same velocities bit for bit.  Built by `make -C oracle _ref` into oracle/_ref/liblangevin_ref.so.
#include <hiprand/hiprand_kernel.h> // the reference pulls it in through ensemble_lan.cuh
#include "integrate/langevin_utilities.cuh"

extern "C" int ref_lan_state_bytes() { return (int)sizeof(gpurandState); }

extern "C" int ref_lan_init(void* states, int N, int seed)
{
  initialize_curand_states<<<(N - 1) / 128 + 1, 128>>>((gpurandState*)states, N, seed);
  return (int)gpuDeviceSynchronize();
}

// Ensemble_LAN::integrate_nvt_lan_half with c1, c2 handed in (ensemble_lan.cu:101-126)
extern "C" int ref_lan_half(void* states, int N, double c1, double c2, const double* mass, double* v)
{
  gpu_langevin<<<(N - 1) / 128 + 1, 128>>>((gpurandState*)states, N, c1, c2, mass, v, v + N, v + 2 * N);
  gpu_find_momentum<<<4, 1024>>>(N, mass, v, v + N, v + 2 * N);
  gpu_correct_momentum<<<(N - 1) / 128 + 1, 128>>>(N, v, v + N, v + 2 * N);
  return (int)gpuDeviceSynchronize();
}
