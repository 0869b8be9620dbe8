// TEST INFRASTRUCTURE ONLY.
// The reference's own thermostat host code compiled in place (nothing is copied into the repository):
//   * resamplekin / gasdev / gamdev of src/integrate/svr_utilities.cuh (Bussi's stochastic velocity rescaling),
//     included from where it lies; its only include, utilities/gpu_macro.cuh, is satisfied by an empty stub;
//   * the static function nhc() of src/integrate/ensemble_nhc.cu (Nose-Hoover chain, Suzuki-Yoshida 7 x 4): the
//     Makefile cuts that one function out of the .cu file into _ref/nhc_extract.inc at build time (the file itself
//     needs the CUDA tool chain), which is included below.
// The wrappers only forward; they pin oracle/nep_oracle.c's nepo_bdp_factor / nepo_nhc (tests/test_oracle_golden.py).
#include <cmath>
#include <random>

#include "integrate/svr_utilities.cuh"
#include "nhc_extract.inc"

extern "C" {

// out[k] = sqrt(resamplekin(ek_k, sigma, ndeg, taut) / ek_k) with ek_k = T[k] ndeg k_B / 2: the velocity scale factors of
// Ensemble_BDP::integrate_nvt_bdp_2 (ensemble_bdp.cu:88-103) for a sequence of instantaneous temperatures, one
// std::mt19937 seeded with `seed`.  gasdev keeps a function-local cache: call this once per process.
void nepref_bdp_factors(unsigned seed, int count, int n_atoms, const double* T_now, double T_target, double t_coup, double* out)
{
  const double K_B = 8.617343e-5;
  std::mt19937 rng(seed);
  const int ndeg = 3 * n_atoms;
  for (int k = 0; k < count; ++k) {
    const double ek = T_now[k] * ndeg * K_B * 0.5;
    const double sigma = ndeg * K_B * T_target * 0.5;
    out[k] = std::sqrt(resamplekin(ek, sigma, ndeg, t_coup, rng) / ek);
  }
}

// one call of nhc() (ensemble_nhc.cu:102-164) on caller-held chain arrays; returns the velocity scale factor
double nepref_nhc(int M, double* pos_eta, double* vel_eta, double* mas_eta, double Ek2, double kT, double dN, double dt2)
{
  return nhc(M, pos_eta, vel_eta, mas_eta, Ek2, kT, dN, dt2);
}
}
