// TEST INFRASTRUCTURE ONLY -- not part of the product.
//
// Thin extern "C" shim around the reference's own vendored CPU implementation
// (NEP_CPU: /root/reference/tools/Miscellaneous/for_coding/for_perioidc_table/nep.{h,cpp}).
// The reference sources are compiled where they lie (see oracle/Makefile, target _ref);
// nothing from them is copied into this repository.  The resulting
// oracle/_ref/libnepcpu_ref.so is used only by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg.
//
// API of the wrapped class: NEP3::compute(type, box[9], position[3N], potential[N],
// force[3N], virial[9N]) -- nep.h:92-115.  box order ax,bx,cx,ay,by,cy,az,bz,cz (same as
// GPUMD's Box::cpu_h[0..8]); virial order xx,xy,xz,yx,yy,yz,zx,zy,zz (nep.h:104-106), which
// this shim re-orders to GPUMD's plane order xx,yy,zz,xy,xz,yz,yx,zx,zy (force.cu:568-571).

#include "nep.h"

#include <cstring>
#include <vector>

extern "C" {

void* nepref_create(const char* nep_txt)
{
  return new NEP3(std::string(nep_txt));
}

void nepref_destroy(void* h)
{
  delete static_cast<NEP3*>(h);
}

int nepref_info(void* h, double* rc_radial, double* rc_angular, int* dim, int* num_types)
{
  NEP3* nep = static_cast<NEP3*>(h);
  *rc_radial = nep->paramb.rc_radial;
  *rc_angular = nep->paramb.rc_angular;
  *dim = nep->annmb.dim;
  *num_types = nep->paramb.num_types;
  return 0;
}

// pe[N], force[3N] SoA, virial[9N] in GPUMD plane order.
int nepref_compute(
  void* h, int n, const int* type, const double* box9, const double* pos, double* pe, double* force,
  double* virial)
{
  NEP3* nep = static_cast<NEP3*>(h);
  std::vector<int> t(type, type + n);
  std::vector<double> b(box9, box9 + 9);
  std::vector<double> p(pos, pos + 3 * (size_t)n);
  std::vector<double> e(n), f(3 * (size_t)n), v(9 * (size_t)n);
  nep->compute(t, b, p, e, f, v);
  std::memcpy(pe, e.data(), sizeof(double) * n);
  std::memcpy(force, f.data(), sizeof(double) * 3 * n);
  // NEP_CPU: xx xy xz yx yy yz zx zy zz  ->  GPUMD: xx yy zz xy xz yz yx zx zy
  static const int map_gpumd_from_cpu[9] = {0, 4, 8, 1, 2, 5, 3, 6, 7};
  for (int k = 0; k < 9; ++k) {
    std::memcpy(
      virial + (size_t)k * n, v.data() + (size_t)map_gpumd_from_cpu[k] * n, sizeof(double) * n);
  }
  return 0;
}

// descriptor[dim*N] ordered d0[N], d1[N], ... (already multiplied by q_scaler)
int nepref_descriptor(
  void* h, int n, const int* type, const double* box9, const double* pos, double* descriptor)
{
  NEP3* nep = static_cast<NEP3*>(h);
  std::vector<int> t(type, type + n);
  std::vector<double> b(box9, box9 + 9);
  std::vector<double> p(pos, pos + 3 * (size_t)n);
  std::vector<double> d((size_t)n * nep->annmb.dim);
  nep->find_descriptor(t, b, p, d);
  std::memcpy(descriptor, d.data(), sizeof(double) * d.size());
  return 0;
}

// Neighbour lists of the LAST compute()/descriptor call.  which: 0 radial, 1 angular.
// nn[N]; nl[ld*N] column-major (slot*N + atom) like the reference; returns max count or -1.
int nepref_neighbors(void* h, int which, int n, int* nn, int* nl, int ld)
{
  NEP3* nep = static_cast<NEP3*>(h);
  const std::vector<int>& NN = which == 0 ? nep->NN_radial : nep->NN_angular;
  const std::vector<int>& NL = which == 0 ? nep->NL_radial : nep->NL_angular;
  if ((int)NN.size() < n)
    return -1;
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    nn[i] = NN[i];
    if (NN[i] > mx)
      mx = NN[i];
    for (int s = 0; s < NN[i] && s < ld; ++s)
      nl[(size_t)s * n + i] = NL[(size_t)s * n + i];
  }
  return mx;
}

} // extern "C"
