// TEST INFRASTRUCTURE ONLY.
// The reference's own device functions for the angular descriptor -- accumulate_s, find_q and
// accumulate_f12 of src/utilities/nep_utilities.cuh (all invariants, including the extra 4-body rows
// 112/123/233/134 that NEP_CPU does not carry) -- compiled for the HOST straight from the header where
// it lies under $(REFERENCE): the CUDA qualifiers are defined away, nothing is copied.  The wrappers
// below only forward; they pin oracle/nep_oracle.c's restatement of these rows (tests/test_oracle_golden.py).
#define __device__
#define __host__
#define __forceinline__ inline
#define __constant__ static const
#include <cmath>
using std::abs;
#include "utilities/nep_utilities.cuh"

extern "C" {

int nepref_num_abc(void) { return NUM_OF_ABC; }

// s[NUM_OF_ABC] += fn * basis(r12 / d12), rows of L <= L_max (accumulate_s, :1725-1756)
void nepref_accumulate_s(int L_max, float d12, float x12, float y12, float z12, float fn, float* s)
{
  accumulate_s(L_max, d12, x12, y12, z12, fn, s);
}

// q[L_index * n_max_angular_plus_1 + n] for every invariant row (find_q, :1819-1947)
void nepref_find_q(
  int L_max, int has_222, int has_1111, int has_112, int has_123, int has_233, int has_134, int n_max_angular_plus_1,
  int n, const float* s, float* q)
{
  find_q(L_max, has_222, has_1111, has_112, has_123, has_233, has_134, n_max_angular_plus_1, n, s, q);
}

// f12[3] += dU_i/dr_ij of radial order n (accumulate_f12, :1523-1672); sum_fxyz is [n][NUM_OF_ABC]
void nepref_accumulate_f12(
  int L_max, int has_222, int has_1111, int has_112, int has_123, int has_233, int has_134, int num_L, int n,
  int n_max_angular_plus_1, float d12, const float* r12, float fn, float fnp, const float* Fp, const float* sum_fxyz,
  float* f12)
{
  accumulate_f12(
    L_max, has_222, has_1111, has_112, has_123, has_233, has_134, num_L, n, n_max_angular_plus_1, d12, r12, fn, fnp, Fp,
    sum_fxyz, f12);
}
}
