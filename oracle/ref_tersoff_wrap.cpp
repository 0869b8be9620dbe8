// TEST INFRASTRUCTURE ONLY.
// The reference's own Tersoff-1989 device code compiled for the HOST (nothing is copied into the repository): the
// Makefile cuts, at build time and into the git-ignored _ref directory,
//   * struct Tersoff1989_Parameters out of src/force/tersoff1989.cuh,
//   * the device helpers and the kernels find_force_tersoff_step1 / _step2 out of src/force/tersoff1989.cu:157-505,
//   * gpu_find_force_many_body (double) out of src/force/potential.cu:35-134,
// because the files as a whole need the CUDA tool chain (GPU_Vector, kernel launches).  The CUDA qualifiers are defined
// away and a kernel "launch" is a host loop over threadIdx.x.  Box and apply_mic come from src/model/box.cuh as it lies.
// Pins oracle/tersoff_oracle.c (tests/test_tersoff.py): same neighbour list in, energies / forces / virials out.
#define __global__
#define __device__
#define __host__
#define __restrict__
#include <cmath>
#include <vector>

namespace {
struct Idx {
  int x;
};
Idx blockIdx{0}, blockDim{0}, threadIdx{0};
} // namespace

#include "model/box.cuh"
#include "tersoff_params_extract.inc"
#include "tersoff_kernels_extract.inc"
#include "many_body_extract.inc"

extern "C" {

// par: three parameter sets (type 0-0, type 1-1, mixed), 16 doubles each in the order
//   a b lambda mu beta n c d h r1 r2 c2 d2 one_plus_c2overd2 pi_factor minus_half_over_n
// NN[n], NL[slot * n + atom] (the local list, tersoff1989.cu:141-149), position SoA; pe / force / virial are added to.
void nepref_tersoff(
  int n, const double* h18, const int* pbc, int orthogonal, const double* par, const int* NN, const int* NL,
  const int* type, const double* pos, double* pe, double* force, double* virial)
{
  Box box;
  box.pbc_x = pbc[0];
  box.pbc_y = pbc[1];
  box.pbc_z = pbc[2];
  for (int k = 0; k < 18; ++k) {
    box.cpu_h[k] = h18[k];
    box.float_h[k] = (float)h18[k];
  }
  box.is_orthogonal = orthogonal != 0;
  Tersoff1989_Parameters t[3];
  for (int s = 0; s < 3; ++s) {
    const double* p = par + 16 * s;
    t[s].a = p[0];
    t[s].b = p[1];
    t[s].lambda = p[2];
    t[s].mu = p[3];
    t[s].beta = p[4];
    t[s].n = p[5];
    t[s].c = p[6];
    t[s].d = p[7];
    t[s].h = p[8];
    t[s].r1 = p[9];
    t[s].r2 = p[10];
    t[s].c2 = p[11];
    t[s].d2 = p[12];
    t[s].one_plus_c2overd2 = p[13];
    t[s].pi_factor = p[14];
    t[s].minus_half_over_n = p[15];
  }
  int max_nn = 0;
  for (int i = 0; i < n; ++i)
    max_nn = NN[i] > max_nn ? NN[i] : max_nn;
  const size_t sz = (size_t)(max_nn > 0 ? max_nn : 1) * n;
  std::vector<double> b(sz), bp(sz), f12x(sz), f12y(sz), f12z(sz);
  const double *x = pos, *y = pos + n, *z = pos + 2 * (size_t)n;
  blockIdx.x = 0;
  blockDim.x = 0;
  for (threadIdx.x = 0; threadIdx.x < n; ++threadIdx.x)
    find_force_tersoff_step1(n, 0, n, box, t[0], t[1], t[2], NN, NL, type, x, y, z, b.data(), bp.data());
  for (threadIdx.x = 0; threadIdx.x < n; ++threadIdx.x)
    find_force_tersoff_step2(n, 0, n, box, t[0], t[1], t[2], NN, NL, type, b.data(), bp.data(), x, y, z, pe, f12x.data(),
                             f12y.data(), f12z.data());
  for (threadIdx.x = 0; threadIdx.x < n; ++threadIdx.x)
    gpu_find_force_many_body(n, 0, n, box, NN, NL, f12x.data(), f12y.data(), f12z.data(), x, y, z, force, force + n,
                             force + 2 * (size_t)n, virial);
}
}
