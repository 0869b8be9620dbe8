/* TEST INFRASTRUCTURE ONLY -- not part of the product, never linked into libnepmi.so.
 *
 * CPU restatement (plain C) of the reference's NEP force path:
 *   neighbour construction  src/force/neighbor.cu:85-215, src/force/nep.cu:436-486,
 *                           src/force/nep_small_box.cuh:56-132, src/force/nep.cu:1295-1354
 *   descriptor + ANN        src/force/nep.cu:488-659, src/utilities/nep_utilities.cuh:169-194,
 *                           :409-431, :572-623, :1674-1817
 *   radial force            src/force/nep.cu:661-772
 *   angular partial force   src/force/nep.cu:774-861, nep_utilities.cuh:625-718, :1327-1521
 *   many-body accumulate    src/force/potential.cu:170-297
 *   ZBL                     src/force/nep.cu:863-975, nep_utilities.cuh:433-508
 *   pbc wrap / zero         src/force/force.cu:314-333, :424-459
 *   velocity-Verlet, thermo src/integrate/ensemble.cu:176-214, :434-633
 *
 * Pinned against (tests/test_oracle_golden.py):
 *   - the reference's CUDA-path known answer examples/gpumd_static/dump.xyz,
 *   - examples/nep_prediction/{energy,force,virial}_train.out,
 *   - tests_pytest/fixtures/golden/bulk_bazro3.npz,
 *   - the reference's own vendored NEP_CPU compiled in place (oracle/_ref).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef NEP_ORACLE_H
#define NEP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nepo_model nepo_model;

typedef struct {
  int version;      /* 3, 4 or 5 */
  int num_types;
  int zbl_enabled;
  int zbl_flexible;
  double rc_radial_max;
  double rc_angular_max;
  int n_max_radial, n_max_angular;
  int basis_size_radial, basis_size_angular;
  int L_max, has_222, has_1111, num_L;
  int dim, num_neurons;
  int MN_radial, MN_angular; /* enlarged by 1.25 like nep.cu:234-235 */
  int num_para;              /* ANN + descriptor parameters (without q_scaler) */
} nepo_info;

nepo_model* nepo_model_load(const char* path, char* err, int errlen);
void nepo_model_free(nepo_model* m);
void nepo_model_info(const nepo_model* m, nepo_info* out);
const char* nepo_model_symbol(const nepo_model* m, int t);
double nepo_model_param(const nepo_model* m, int idx);
/* temperature-dependent NEP (nep4[_zbl]_temperature, nep.cu:125-130): the ANN has one more input, the temperature
 * that Force::compute passes to NEP::compute(temperature, ...) (force.cu:516-525; nep.cu:1483 q[dim-1] = temperature). */
int nepo_model_is_temperature(const nepo_model* m);
void nepo_model_set_temperature(nepo_model* m, double temperature);

/* Neighbour lists of one configuration.
 * path: -1 = choose like NEP::compute (small box iff a periodic thickness <= 2.5*(rc+1)),
 *        0 = force the large-box (cell list + MIC) path, 1 = force the small-box path.
 * which: 0 = skin list (rc_radial_max + 1 A; large-box path only), 1 = radial, 2 = angular.
 * Lists are returned sorted ascending by neighbour index (neighbor.cuh:112-136); for the
 * small-box path an entry may repeat with a different periodic image.
 * nl is column-major nl[slot*n + atom] with ld slots; returns max count, or <0 on error. */
/* the angular rows alone (FP64): find_q and accumulate_f12 of nep_utilities.cuh for one radial order n;
 * has[6] = has_q_222, _1111, _112, _123, _233, _134; s / sums_n = the 24 sums of that order */
void nepo_rows_find_q(int L_max, const int has[6], int nA1, int n, const double* s, double* q);
void nepo_rows_accumulate_f12(
  int L_max, const int has[6], int n, int nA1, double d12, const double* r12, double fn, double fnp, const double* Fp,
  const double* sums_n, double* f12);

typedef struct nepo_lists nepo_lists;
nepo_lists* nepo_lists_build(
  const nepo_model* m, int n, const int* type, const double h[9], const int pbc[3],
  const double* pos, int path);
int nepo_lists_path(const nepo_lists* l);
int nepo_lists_get(const nepo_lists* l, int which, int* nn, int* nl, int ld);
void nepo_lists_free(nepo_lists* l);

/* Force::compute + NEP::compute for one configuration (positions are NOT wrapped here).
 * precision 32: float arithmetic like the reference GPU kernels; 64: the same formulas in
 * double (comparable with NEP_CPU).  pos/force SoA [x..|y..|z..]; virial 9 planes in GPUMD
 * order xx,yy,zz,xy,xz,yz,yx,zx,zy.  Outputs are ASSIGNED (zero + accumulate).
 * q_out (dim*n, plane per component, scaled by q_scaler) and fp_out (dim*n) may be NULL. */
int nepo_compute(
  const nepo_model* m, int precision, int path, int n, const int* type, const double h[9],
  const int pbc[3], const double* pos, double* pe, double* force, double* virial, double* q_out,
  double* fp_out);

/* gpu_apply_pbc (force.cu:424-459): wrap into the cell, in place. */
void nepo_apply_pbc(int n, const double h[9], const int pbc[3], double* pos);

/* gpu_velocity_verlet (ensemble.cu:176-214). dt in natural units (fs/10.18051). */
void nepo_velocity_verlet(
  int is_step1, int n, double dt, const double* mass, const double* force, double* pos,
  double* vel);

/* gpu_find_thermo_instant_temperature (ensemble.cu:434-633): thermo[8] =
 * T, U, sxx, syy, szz, sxy, sxz, syz (stress = (virial + m v v)/V, eV/A^3). */
/* Ensemble_NHC (ensemble_nhc.cu:30-49, 102-164): st = eta[4] | p_eta[4] | Q[4] */
void nepo_nhc_init(int n, double temperature, double t_coup, double dt, double* st);
double nepo_nhc(double* st, double ek2, double kT, double dN, double dt2_particle);

/* Ensemble_BDP (ensemble_bdp.cu:71-104, svr_utilities.cuh:28-122): state = MT19937 + gasdev cache in an
 * opaque buffer of nepo_bdp_sizeof() bytes */
int nepo_bdp_sizeof(void);
void nepo_bdp_seed(void* g, unsigned int seed);
double nepo_bdp_factor(void* g, int n, double T_now, double T_target, double t_coup);

void nepo_thermo(
  int n, double volume, const double* mass, const double* pe, const double* vel,
  const double* virial, double* thermo8);

/* NVE loop as Run::perform_a_run drives it (run.cu:250-318): vv1, wrap, force, vv2, thermo.
 * thermo_out[(nsteps)*8]; returns number of neighbour rebuilds the reference's skin policy
 * (neighbor.cu:741-800) would have performed. */
int nepo_run_nve(
  const nepo_model* m, int precision, int n, const int* type, const double h[9],
  const int pbc[3], const double* mass, double dt, int nsteps, double* pos, double* vel,
  double* pe, double* force, double* virial, double* thermo_out);

#ifdef __cplusplus
}
#endif
#endif
