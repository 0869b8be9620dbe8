/* nepmi.h -- C ABI of libnepmi.so, the MI355X (gfx950) NEP force engine + velocity-Verlet path.
 *
 * This is the drop-in boundary for GPUMD's per-step force path.  The reference has no FFI: the
 * boundary there is the C++ virtual interface Potential::compute / Force::compute operating on
 * device arrays (GPU_Vector<T>::data()).  Each entry point below names the reference interface
 * it replaces (paths relative to the reference tree, file:line).  INTEGRATION.md shows the
 * `class NEP_MI : public Potential` adaptor a GPUMD maintainer would add on top of these calls.
 *
 * Conventions (identical to the reference's, src/model/atom.cuh:32-42, src/force/force.cu:568-571):
 *   - every `double*` / `int*` named pos/vel/force/pe/virial/type/mass is a DEVICE pointer on the
 *     current HIP device, owned by the caller, SoA in the caller's atom order:
 *       pos/vel/force  [x0..xN-1 | y0..yN-1 | z0..zN-1]          (3N doubles)
 *       virial         9 planes xx,yy,zz,xy,xz,yz,yx,zx,zy        (9N doubles)
 *       pe             N doubles;  type N ints;  mass N doubles
 *   - units eV, Angstrom, amu; time in GPUMD's natural unit (fs / 10.18051).
 *   - box h[9] is the HOST array Box::cpu_h[0..8] = ax,bx,cx,ay,by,cy,az,bz,cz
 *     (lattice vectors are the columns; src/model/read_xyz.cu:208-216); pbc[3] host ints.
 *   - all calls are asynchronous on the engine's stream unless stated; single host thread per
 *     engine (like the reference), several engines (one per GPU / process) may coexist.
 *   - return value 0 on success, negative nepmi_status on error; nepmi_last_error() gives the text.
 *     (The reference exits the process on error, src/utilities/error.cuh:23-63; the adaptor in
 *     INTEGRATION.md restores that behaviour with one macro.)
 *   - no torch / STL / HIP types in any signature: plain pointers and sizes only.  The stream is
 *     passed as void* (a hipStream_t); NULL means the default stream.
 */
#ifndef NEPMI_H
#define NEPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEPMI_VERSION 200

typedef enum {
  NEPMI_OK = 0,
  NEPMI_ERR_IO = -1,          /* cannot open / short file */
  NEPMI_ERR_FORMAT = -2,      /* nep.txt violates the format of src/force/nep.cu:100-377 */
  NEPMI_ERR_UNSUPPORTED = -3, /* valid model/box outside this engine's envelope */
  NEPMI_ERR_ARG = -4,
  NEPMI_ERR_HIP = -5,         /* HIP runtime error or no gfx950 device */
  NEPMI_ERR_CAPACITY = -6,    /* a neighbour list exceeded its MN capacity (the reference corrupts
                                 memory silently here, src/force/nep.cu:234-235,1014-1034) */
  NEPMI_ERR_SMALL_BOX = -7    /* a box the small-box branch (a periodic thickness <= 2.5*(rc+skin), reference:
                                 nep_small_box.cuh) cannot take: more than 20000 atoms, or thicker than 10 rc in
                                 another direction (the reference refuses that one too, nep.cu:1316-1324) */
} nepmi_status;

typedef struct nepmi_model nepmi_model;   /* parsed nep.txt (host)            */
typedef struct nepmi_engine nepmi_engine; /* device state of one potential    */

/* Mirrors what NEP::NEP prints / keeps in ParaMB + ANN (src/force/nep.cuh, nep.cu:100-395). */
typedef struct {
  int version;     /* 3, 4, 5 */
  int num_types;
  int zbl_enabled, zbl_flexible;
  double zbl_rc_inner, zbl_rc_outer;
  double rc_radial, rc_angular; /* maxima over types */
  int MN_radial, MN_angular;    /* already enlarged by 1.25 (nep.cu:234-235) */
  int n_max_radial, n_max_angular;
  int basis_size_radial, basis_size_angular;
  int L_max, has_q_222, has_q_1111, num_L;
  int dim, num_neurons;
  int num_para; /* ANN + descriptor parameters, without q_scaler */
  int has_q_112, has_q_123, has_q_233, has_q_134; /* optional extra 4-body rows (nep.cu:275-310) */
  int model_type; /* nep.cuh:48: 0 = potential, 3 = temperature-dependent (nep4[_zbl]_temperature); `dim` counts the
                   * descriptor components, the reference's annmb.dim is dim + 1 for model_type 3 (nep.cu:321-325) */
} nepmi_info;

const char* nepmi_last_error(void);
int nepmi_version(void);

/* ---- model: replaces NEP::NEP(const char* file_potential, int num_atoms) parsing half,
 *      src/force/nep.cu:100-377 + update_potential :402-434.  Host only, no GPU needed. ---- */
nepmi_model* nepmi_model_load(const char* nep_txt_path);
void nepmi_model_free(nepmi_model* m);
int nepmi_model_info(const nepmi_model* m, nepmi_info* out);
const char* nepmi_model_symbol(const nepmi_model* m, int type); /* element symbol of a type */

/* ---- engine: replaces the allocation half of NEP::NEP (nep.cu:379-391) + Neighbor
 *      (src/force/neighbor.cu:802-833).  n_atoms is fixed for the life of the engine. ---- */
nepmi_engine* nepmi_engine_create(const nepmi_model* m, int64_t n_atoms, void* hip_stream);
void nepmi_engine_destroy(nepmi_engine* e);

/* ---- Force::compute, src/force/force.cu:771-855 (gpu_apply_pbc :424-459,
 *      initialize_properties :314-333, then potentials[0]->compute): wraps pos in place, zeroes
 *      pe/force/virial, adds the NEP contribution. ---- */
int nepmi_force_compute(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, double* pos,
  double* pe, double* force, double* virial);

/* ---- Potential::compute == NEP::compute, src/force/nep.cu:1356-1389 (large-box branch
 *      :996-1137): ADDS to pe/force/virial, expects wrapped positions, applies the Verlet-skin
 *      policy of Neighbor::find_neighbor_global (neighbor.cu:741-800) internally. ---- */
int nepmi_potential_compute(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* pos, double* pe, double* force, double* virial);

/* ---- the same call on the LOCAL system of a spatial domain decomposition (replaces the per-GPU
 *      ranges N1..N5 of NEP_MULTIGPU::compute, src/force/nep_multigpu.cuh:42-50, :1416-1803):
 *      n (<= the engine's capacity) owned + ghost atoms; level[i] (DEVICE, n signed chars) = 2 owned
 *      (forces, virial, energy are produced), 1 inner ghost (descriptors + partial forces are
 *      recomputed redundantly, like the reference's inner ring), 0 outer ghost (position only).
 *      level == NULL: all owned.  Directions with pbc = 0 are open: ghosts carry their periodic
 *      image shift explicitly.  Levels are sampled when the Verlet list is rebuilt; call
 *      nepmi_engine_invalidate after changing the composition of the local system. ---- */
int nepmi_potential_compute_levels(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* pos, const signed char* level, double* pe, double* force, double* virial);
int nepmi_engine_invalidate(nepmi_engine* e);
/* on != 0: the caller runs the skin policy itself (a domain-decomposed host votes on the 0.5 A
 * displacement criterion across ranks and calls nepmi_engine_invalidate): the force calls then skip
 * the per-step flag read-back and are enqueued without any host round trip.  List-capacity overflow
 * is reported at the next rebuild or nepmi_engine_stats call instead of immediately. */
int nepmi_engine_set_external_skin(nepmi_engine* e, int on);
/* Unwrapped coordinates (Atom::unwrapped_position; gpu_update_unwrapped_position,
 * src/integrate/integrate.cu:312-372): a caller-owned device array [3N] that every first
 * half-step (nepmi_vv_step1 and the fused nepmi_run_* loops) adds its un-wrapped drift
 * (new - old position, before the periodic wrap) to.  NULL (the default) switches it off. */
int nepmi_engine_set_unwrapped(nepmi_engine* e, double* d_unwrapped);
/* The same call in two halves, so that a domain-decomposed host can overlap its ghost-position
 * exchange (the RCCL send/recv that replaces NEP_MULTIGPU's staged copies) with compute:
 *   _begin: the OWNED (level 2) entries of pos are final; ghost entries may still be in flight and
 *           are not read.  Enqueues the skin check + gather of the owned atoms and the radial pass
 *           of the interior bricks (those whose 8x8x8-cell window holds no ghost).  Returns 1 if
 *           work was started, 0 if not (no valid list yet, geometry changed, LDS-window pass not
 *           applicable): _end then does everything.
 *   _end:   every entry of pos is final (the caller has made the engine's stream wait for its
 *           exchange).  Ghosts are gathered and checked, then the boundary bricks and the rest of
 *           the force path run; if the skin check asks for a rebuild the interior work is redone
 *           on the new list.  Results are identical to nepmi_potential_compute_levels. */
int nepmi_potential_compute_levels_begin(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* pos, const signed char* level, double* pe, double* force, double* virial);
int nepmi_potential_compute_levels_end(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* pos, const signed char* level, double* pe, double* force, double* virial);

/* gpu_apply_pbc (force.cu:424-459) and initialize_properties (force.cu:314-333) on their own. */
int nepmi_apply_pbc(nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, double* pos);
int nepmi_zero_properties(nepmi_engine* e, int64_t n, double* pe, double* force, double* virial);
/* gpu_average_properties (force.cu:461-480): pe, force, virial /= denominator -- the last step of
 * Force::compute in the "average" mode of several NEP potentials (force.cu:533-562). */
int nepmi_average_properties(
  nepmi_engine* e, int64_t n, double denominator, double* pe, double* force, double* virial);

/* ---- Ensemble::velocity_verlet, src/integrate/ensemble.cu:176-214,348-397
 *      (Ensemble_NVE::compute1/compute2, ensemble_nve.cu:31-95).  dt in natural units. ---- */
int nepmi_vv_step1(
  nepmi_engine* e, int64_t n, double dt, const double* mass, const double* force, double* pos,
  double* vel);
int nepmi_vv_step2(
  nepmi_engine* e, int64_t n, double dt, const double* mass, const double* force, double* vel);

/* ---- Ensemble::find_thermo, ensemble.cu:434-673: thermo8 (DEVICE, 8 doubles) =
 *      T, U, sxx, syy, szz, sxy, sxz, syz   (stress = (sum virial + sum m v v) / volume). ---- */
int nepmi_find_thermo(
  nepmi_engine* e, int64_t n, double volume, const double* mass, const double* pe,
  const double* vel, const double* virial, double* thermo8);

/* ---- fused NVE loop == Run::perform_a_run restricted to `ensemble nve`
 *      (src/main_gpumd/run.cu:250-318): nsteps x { vv1, Force::compute, vv2, find_thermo }.
 *      thermo_host (HOST, 8 doubles per recorded step) receives find_thermo's output every
 *      `thermo_every` steps (0 = never); returns after the last step has finished.  The caller's
 *      device arrays hold the final state in the caller's atom order. ---- */
int nepmi_run_nve(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host);

/* ---- Berendsen thermostat: gpu_berendsen_temperature (src/integrate/ensemble_ber.cu:70-86),
 *      v *= sqrt(1 + coupling (T_target / T - 1)) with T = thermo8[0] (DEVICE, from find_thermo) and
 *      coupling = 1 / T_coup; and the whole `ensemble nvt_ber T1 T2 T_coup` loop
 *      (Ensemble_BER::compute1/compute2, ensemble_ber.cu:180-235; linear target ramp T1 -> T2,
 *      integrate.cu:341-344).  thermo_host records find_thermo's output (before the rescale). ---- */
int nepmi_berendsen_scale(
  nepmi_engine* e, int64_t n, double temperature, double coupling, const double* thermo8, double* vel);
int nepmi_run_nvt_ber(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos,
  double* vel, double* pe, double* force, double* virial, int64_t thermo_every, double* thermo_host);

/* ---- Nose-Hoover chain thermostat: Ensemble_NHC (src/integrate/ensemble_nhc.cu:30-49 constructor,
 *      :102-164 nhc(), :166-232 integrate_nvt_nhc_1/2), `ensemble nvt_nhc T1 T2 T_coup`.
 *      chain_state: NEPMI_NHC_STATE_SIZE doubles of DEVICE memory owned by the caller
 *      (pos_nhc1[4] | vel_nhc1[4] | mas_nhc1[4] | last scale factor).  The reference integrates the
 *      chain on the host after copying the temperature back; here nepmi_nhc_half_step advances it on
 *      the device from thermo8[0] (DEVICE, find_thermo) and rescales the velocities -- no host round
 *      trip.  A step of the ensemble is: find_thermo, half_step, vv_step1, force, vv_step2,
 *      find_thermo, half_step.  nepmi_run_nvt_nhc is the whole loop (T ramps T1 -> T2,
 *      integrate.cu:341-344); thermo_host records the second find_thermo of each recorded step. ---- */
#define NEPMI_NHC_STATE_SIZE 13
int nepmi_nhc_init(nepmi_engine* e, int64_t n, double temperature, double t_coup, double dt, double* chain_state);
int nepmi_nhc_half_step(
  nepmi_engine* e, int64_t n, double temperature, double dt, const double* thermo8, double* chain_state,
  double* vel);
int nepmi_run_nvt_nhc(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos,
  double* vel, double* pe, double* force, double* virial, int64_t thermo_every, double* thermo_host);
/* The chain of nepmi_run_nvt_nhc is fresh at an engine's first call and after this call (a new `run` keyword,
 * integrate.cu:85-92); otherwise it continues, so that a host can run in segments between its output steps. */
int nepmi_engine_reset_thermostat(nepmi_engine* e);

/* ---- Bussi-Donadio-Parrinello stochastic velocity rescaling: Ensemble_BDP
 *      (src/integrate/ensemble_bdp.cu:71-104) with resamplekin / gasdev / gamdev of
 *      src/integrate/svr_utilities.cuh:28-122, `ensemble nvt_bdp T1 T2 T_coup`.  The noise comes from a
 *      host std::mt19937 read through uniform_real_distribution<double>(0,1), as in the reference (which
 *      seeds it from the clock; here the seed is explicit, default 12345678 = the reference's DEBUG seed).
 *      nepmi_bdp_scale: T = thermo8[0] (DEVICE) is read back, the new kinetic energy is drawn, the
 *      velocities are rescaled.  A step of the ensemble: vv_step1, force, vv_step2, find_thermo, bdp_scale. ---- */
int nepmi_bdp_seed(nepmi_engine* e, uint64_t seed);
int nepmi_bdp_scale(nepmi_engine* e, int64_t n, double temperature, double t_coup, const double* thermo8, double* vel);
int nepmi_run_nvt_bdp(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos,
  double* vel, double* pe, double* force, double* virial, int64_t thermo_every, double* thermo_host);

/* ---- Langevin thermostat: Ensemble_LAN (src/integrate/ensemble_lan.cu:30-41, :96-127, :206-262; kernels of
 *      src/integrate/langevin_utilities.cuh), `ensemble nvt_lan T1 T2 T_coup`.  One XORWOW generator state per atom
 *      (hiprand_init(seed, n, 0), n = the caller's atom index, as in the reference, which takes seed = rand()):
 *      nepmi_lan_seed sets the seed and makes the next half-step initialise the states (default 12345678).
 *      nepmi_lan_half_step = integrate_nvt_lan_half: v <- c1 v + c2 sqrt(1/m) xi with c1 = exp(-1/(2 T_coup)),
 *      c2 = sqrt((1 - c1^2) k_B T), three normal draws per atom, then the centre-of-mass velocity is removed (the four
 *      sums in gpu_find_momentum's order: with the same seed the velocities equal the reference kernels' bit for bit).
 *      A step of the ensemble: lan_half_step, vv_step1, force, vv_step2, lan_half_step, find_thermo.
 *      nepmi_run_nvt_lan is that loop, device-resident (the states stay in the caller's atom order; bit-identical to the
 *      sequence above). ---- */
int nepmi_lan_seed(nepmi_engine* e, int seed);
int nepmi_lan_half_step(nepmi_engine* e, int64_t n, double temperature, double t_coup, const double* mass, double* vel);
int nepmi_run_nvt_lan(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos,
  double* vel, double* pe, double* force, double* virial, int64_t thermo_every, double* thermo_host);

/* ---- BAOAB Langevin integrator: Ensemble_BAO (src/integrate/ensemble_bao.cu:30-41, :87-117, :224-250, :340-373,
 *      :419-514), `ensemble nvt_bao T1 T2 T_coup`.  A step: B (half kick), A (half drift), O (the Langevin kernels of
 *      nvt_lan over a whole step, c1 = exp(-1 / T_coup)), A, force, B, find_thermo; the generators are those of
 *      nepmi_lan_seed.  As in the reference the noise amplitude is fixed when the ensemble is set up (c2 from T1,
 *      ensemble_bao.cu:36): t2 is accepted and has no effect. ---- */
int nepmi_run_nvt_bao(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type,
  const double* mass, double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos,
  double* vel, double* pe, double* force, double* virial, int64_t thermo_every, double* thermo_host);

/* ---- multi-GPU: spatial domain decomposition, one process per GPU ----
 *      Replaces NEP_MULTIGPU (src/force/nep_multigpu.cuh:42-50 ranges, nep_multigpu.cu:1416-1803 compute) and
 *      Force::parse_potential's choice of it for `potential <file> [x|y|z]` (src/force/force.cu:122-160).  Every
 *      rank owns the atoms of its sub-box of a Cartesian process grid (grid[0] * grid[1] * grid[2] ranks over the
 *      fractional coordinates of the global cell) and their integrator state; ghost positions travel once per step
 *      (shell 2 (rc + skin), descriptors of the inner ring recomputed, forces for owned atoms only, no reverse
 *      communication -- the reference's semantics); migration and ghost lists are rebuilt only when some atom of
 *      some rank moved more than skin/2 (neighbor.cu:741-800, voted globally).  DESIGN.md section 7.
 *
 *      The transport is a table of function pointers.  device_buffers = 1: buffers are DEVICE memory and the calls
 *      enqueue on `stream` (RCCL over xGMI: nepmi_transport_rccl) -- the step then runs without host round trips,
 *      the skin vote is reduced on the device; device_buffers = 0: buffers are HOST memory, calls block
 *      (nepmi_transport_tcp, or the caller's own functions, e.g. over MPI): the driver stages through the host. ---- */
typedef struct {
  void* buf;
  int64_t bytes;
  int peer;
} nepmi_msg;
typedef struct {
  void* ctx;
  int rank, nranks;
  int device_buffers;
  /* all sends and receives of one call may progress concurrently (ncclGroupStart/End semantics); several messages
     to / from the same peer match in order */
  int (*exchange)(void* ctx, int nsend, const nepmi_msg* sends, int nrecv, const nepmi_msg* recvs, void* stream);
  /* in place; dtype 0 = f64, 1 = i32, 2 = i64; op 0 = sum, 1 = max; every rank must end with identical bits.
     A transport that sets bit 1 of device_buffers (value 3) accepts dtype | NEPMI_DT_DEFER: the result is first read by work
     enqueued AFTER the next exchange() call on the same stream, so the reduction may be posted inside that call's group --
     the skin vote of a step then costs no collective of its own (nepmi_transport_rccl with NEPMI_RCCL_FUSE_VOTE=1). */
  int (*allreduce)(void* ctx, void* buf, int64_t count, int dtype, int op, void* stream);
  void (*destroy)(void* ctx);
} nepmi_transport;

#define NEPMI_DT_DEFER 0x100
#define NEPMI_RCCL_ID_BYTES 128
/* RCCL: rank 0 creates the id (ncclGetUniqueId) and hands it to the others by any means (a file, MPI, torch's
 * store); then every rank builds its communicator (ncclCommInitRank) on the current HIP device. */
int nepmi_transport_rccl_id(char id[NEPMI_RCCL_ID_BYTES]);
int nepmi_transport_rccl(const char id[NEPMI_RCCL_ID_BYTES], int rank, int nranks, nepmi_transport* out);
/* TCP sockets on one node (rank 0 listens on master_addr:port, the MASTER_ADDR / MASTER_PORT of a torchrun-style
 * launch); host buffers. */
int nepmi_transport_tcp(const char* master_addr, int port, int rank, int nranks, nepmi_transport* out);
/* What an RCCL transport has moved so far (bench.py prints it next to an N-GPU line, so that a run explains itself): the size and
 * rank the COMMUNICATOR reports (ncclCommCount / ncclCommUserRank), grouped exchanges, their point-to-point messages and bytes,
 * reductions, and the mean duration of the exchanges that were timed.  time_every > 0: from now on every time_every-th exchange
 * is bracketed by two HIP events on the stream it is enqueued on (0: no events -- the default); reset != 0 clears the counters
 * after reading them.  out may be NULL (only set time_every / reset).  Not an RCCL transport: NEPMI_ERR_ARG. */
typedef struct {
  int64_t comm_nranks, comm_rank;
  int64_t exchanges, messages, bytes_sent, bytes_received, allreduces;
  int64_t timed_exchanges;
  double us_per_timed_exchange;
} nepmi_rccl_stats;
int nepmi_transport_rccl_stats(const nepmi_transport* t, int time_every, int reset, nepmi_rccl_stats* out);
void nepmi_transport_destroy(nepmi_transport* t);

typedef struct nepmi_dist nepmi_dist;
/* h, pbc: the GLOBAL cell.  The transport is used (not owned) until nepmi_dist_destroy. */
nepmi_dist* nepmi_dist_create(
  const nepmi_model* m, const nepmi_transport* t, const double h[9], const int pbc[3], const int grid[3],
  void* hip_stream);
void nepmi_dist_destroy(nepmi_dist* d);
/* The atoms this rank contributes (DEVICE arrays as in the conventions above, n may be 0; any positions: they are
 * migrated to their owners).  ids: global atom numbers, or NULL for a running number over the ranks. */
int nepmi_dist_setup(
  nepmi_dist* d, int64_t n, const int* type, const double* mass, const double* pos, const double* vel,
  const int64_t* ids);
/* Force::compute on the decomposed system (the initial force of Run::perform_a_run). */
int nepmi_dist_compute(nepmi_dist* d);
/* The run loop for ensemble 0 = nve, 1 = nvt_ber, 2 = nvt_nhc, 3 = nvt_bdp, 4 = nvt_lan, 5 = nvt_bao (see nepmi_run_*); thermo_host (HOST,
 * 8 doubles per record, may be NULL) receives the GLOBAL T, U and stresses on every rank. */
int nepmi_dist_run(
  nepmi_dist* d, int ensemble, double dt, int64_t nsteps, double t1, double t2, double t_coup,
  int64_t thermo_every, double* thermo_host);
int nepmi_dist_thermo(nepmi_dist* d, double thermo8_host[8]);
int nepmi_dist_bdp_seed(nepmi_dist* d, uint64_t seed);
/* Langevin thermostats of a decomposed run (ensembles 4 and 5 of nepmi_dist_run = `ensemble nvt_lan` / `nvt_bao`): the seed of the
 * per-atom generators, the same value on every rank (the reference seeds them with rand(), ensemble_lan.cu:39).  The state of the
 * atom with global id s is hiprand_init(seed, s, 0) exactly as in the single-domain nepmi_run_nvt_lan, so the noise of an atom does
 * not depend on the decomposition.  A rank holds the states of the atoms it OWNS only (48 bytes each); they travel with the atom's
 * record when it migrates (memory and work per rank are O(N_local), no collective beyond the momentum sums).  The ids of
 * nepmi_dist_setup only seed the generators and label the gathered output -- nothing on the device is indexed by them during a run.
 * nepmi_dist_setup checks their range (0 .. n_total - 1; the check is all-reduced, every rank returns -4 together); two atoms with
 * the same id would draw the same noise and land on the same row of nepmi_dist_gather_global: ids must be unique over the ranks. */
int nepmi_dist_lan_seed(nepmi_dist* d, int seed);
/* on: the radial pass of the interior bricks (no ghost in their 8x8x8-cell window) is enqueued on the compute stream while the
 * skin vote and the ghost positions travel on a communication stream; off (default since round 3): the plain
 * exchange-then-compute order.  Both orders give bit-identical results.  Two launches of the radial pass have two ramps and two
 * tails of one brick's latency each (~50 us): measured in process, ranks sharing one GPU, the split costs 6 % of a 1 M-atom step
 * per rank (profiles/r3u_*: 2 ranks weak +10.7 % off / +16.8 % on; 8 ranks strong 2.2 / 2.4) -- more than the exchange it hides
 * is expected to take on xGMI; turn it on where the exchange is slow (host transports over a network).
 * With reverse-mode ghosts the same switch also splits the force assembly (scatter form): the bricks whose window holds a ghost
 * and the ghosts' fold first, then the ghosts' partial forces travel on the communication stream while the interior bricks and
 * the owned atoms' fold run; bit-identical to the plain order.  Default off -- a rule, not a measurement of this hardware: the
 * only multi-rank device runs so far share ONE GPU, where a second launch per kernel costs more than the exchange it hides;
 * `bench.py --overlap 1` turns both splits on for an A/B on a node with one GPU per rank. */
int nepmi_dist_set_overlap(nepmi_dist* d, int on);
/* What the ghost atoms are for; call between nepmi_dist_create and nepmi_dist_setup (the shell width shapes the local box).
 *   0 forward: shell 2 (rc + skin), the reference's ranges (src/force/nep_multigpu.cuh:42-50) -- descriptors of the inner ring
 *     rc + skin recomputed on every rank that sees them, forces for owned atoms only, ONE exchange per step (positions);
 *   1 reverse: shell rc + skin, descriptors / network / partial forces for owned atoms only; the force assembly also runs on the
 *     ghosts and leaves on each the pair halves this rank's owned atoms contribute, which travel back to the owner and are
 *     added there: a SECOND exchange per step (3 doubles per ghost), no redundant descriptor work, 1.4 instead of 1.9 local
 *     atoms per owned one for a 1 M-atom PbTe cell on 2 x 2 x 2 ranks -- the form for strong scaling.  NEP models only;
 *  -1 (default) the counted rule: reverse when the forward shell would leave less than 70 % of the local atoms owned at
 *     uniform density (volume of the sub-box / volume of its padded box), or when the sub-box is thinner than the forward
 *     shell.  Per-atom virials of nepmi_dist_gather_* are completed by one extra reverse exchange when asked for.
 * Environment read by the library (experiments and A/B runs; an embedding application that must not be steered from outside calls
 * nepmi_dist_set_ghost_mode / nepmi_dist_set_overlap explicitly, which take precedence): NEPMI_DIST_GHOSTS=forward|reverse replaces
 * the counted rule of mode -1 when the context is created; NEPMI_RCCL_FUSE_VOTE=1 makes the RCCL transport carry the skin vote
 * inside the ghost exchange's group (NEPMI_DT_DEFER); NEPMI_DIST_TRACE prints the stages of a re-decomposition to stderr.
 * (The guard band of the scatter-form force assembly is narrowed for tests through nepmi_engine_set_scatter_guard, not through the
 * environment.) */
int nepmi_dist_set_ghost_mode(nepmi_dist* d, int mode);
typedef struct {
  int64_t n_owned, n_local, n_total; /* atoms owned by this rank, owned + ghosts, in the whole system */
  int64_t num_decompositions, num_steps;
  int64_t num_overlapped; /* steps whose interior radial pass was enqueued before the ghost exchange completed */
  double decompose_ms;    /* wall time spent in the (re-)decompositions so far: migration, ghost stages, list rebuild */
  int64_t reverse_ghosts; /* 1: reverse-mode ghosts (nepmi_dist_set_ghost_mode), 0: forward */
  int64_t num_range_handovers; /* times the scatter-form force assembly was left for the gather form because a force reached its
                                  guard band (nepmi_engine_set_force_form): 0 or 1 per engine generation */
} nepmi_dist_info;
int nepmi_dist_get_info(nepmi_dist* d, nepmi_dist_info* out);
/* sizeof(nepmi_dist_info) of the library: a caller compiled against an older header (a shorter struct) can tell before
 * nepmi_dist_get_info writes past its buffer; counters added after round 3 have getters of their own instead of new fields. */
int nepmi_dist_info_bytes(void);
/* steps whose interior bricks' force assembly ran while the ghosts' partial forces travelled (reverse-mode ghosts with
 * nepmi_dist_set_overlap(1)): the boundary bricks and the ghosts' fold first, the reverse exchange on the communication stream */
int64_t nepmi_dist_num_overlapped_reverse(nepmi_dist* d);
/* The owned atoms of this rank (n_owned entries per plane, global coordinates) into the caller's DEVICE arrays;
 * any pointer may be NULL.  With reverse-mode ghosts (nepmi_dist_set_ghost_mode -- also when the counted rule picked them)
 * the call is COLLECTIVE whatever the arguments: every rank has to make it, the virial halves computed on other ranks'
 * ghosts come home through one more reverse exchange (once per force evaluation) whether or not this rank passes `virial`.
 * Per-atom virials are the reference's attribution (a virial-only pass of the gather form after scatter-form steps).
 * ids (nepmi_dist_setup) have to be unique; nepmi_dist_gather_global additionally needs them to be 0 .. n_total - 1.  The
 * Langevin generator of an atom is created from (seed, id) and travels with the atom: no array is indexed by an id. */
int nepmi_dist_gather_owned(
  nepmi_dist* d, int64_t* ids, double* pos, double* vel, double* force, double* pe, double* virial);
/* Every atom of the system on rank `root`, ordered by global id (which must be 0 .. n_total-1, the default): DEVICE
 * arrays with n_total entries per plane on the root (any may be NULL), ignored on the other ranks.  Collective. */
int nepmi_dist_gather_global(
  nepmi_dist* d, int root, double* pos, double* vel, double* force, double* pe, double* virial);
/* The next nepmi_dist_run starts a fresh Nose-Hoover chain (a new `run` keyword; the chain otherwise continues
 * across the calls, so that a host can run in segments between its output steps). */
int nepmi_dist_reset_thermostat(nepmi_dist* d);
/* The engine of the local (owned + ghost) system, e.g. for nepmi_engine_stats / nepmi_engine_set_timing. */
nepmi_engine* nepmi_dist_engine(nepmi_dist* d);

/* ---- diagnostics / parity hooks ---- */

/* Per-step radial (which = 0) / angular (which = 1) neighbour lists of the LAST compute, in the
 * caller's atom indices, ascending, column-major nl[slot*n + atom] with ld slots: the contents
 * of NEP_Data::NN_radial/NL_radial/NN_angular/NL_angular (nep.cuh) after
 * find_neighbor_list_large_box (nep.cu:436-486).  which = 2: the Verlet-skin list
 * (Neighbor::NN/NL, neighbor.cu:85-162).  nn and nl are DEVICE pointers.  Returns the largest
 * count, or a negative status. */
int nepmi_neighbors_export(nepmi_engine* e, int which, int* nn, int* nl, int64_t ld);

/* Descriptor q (scaled by q_scaler) and Fp = dU/dq * q_scaler of the last compute; DEVICE float
 * arrays [dim][n] in caller order (NEP_Data::Fp layout, nep.cu:655-657).  Either may be NULL. */
int nepmi_descriptors_export(nepmi_engine* e, float* q, float* fp);

typedef struct {
  int64_t num_compute;   /* calls of potential_compute               */
  int64_t num_rebuild;   /* Verlet-list rebuilds (neighbor.cu:741-800) */
  int max_nn_skin, max_nn_radial, max_nn_angular; /* what neighbor.out reports, nep.cu:1014-1034 */
  double mean_nn_radial, mean_nn_angular;         /* measured means of the last compute */
  double ms_force_last;  /* HIP-event time of the last force evaluation (all force kernels)   */
  double ms_kernel[8];   /* last launch of: 0 gather/skin-check, 1 radial descriptor, 2 angular
                            descriptor, 3 ANN, 4 angular partial force, 5 force assembly,
                            6 velocity-Verlet, 7 list rebuild (whole) */
  double ms_kernel_sum[8]; /* same slots: sum over all launches since timing was (re)enabled */
  int64_t launches[8];     /* ... and their number (slot 7: rebuilds)                       */
  int radial_tiles;        /* LDS-window kernels in the last force call: 0 no (gather kernels), 2 yes, 3 yes on the static
                            * window layout (one lane per atom: Verlet entries kept as LDS slots, see below) */
  int64_t discarded_steps; /* steps of the fused run loops that were enqueued speculatively and then re-run after a list
                              rebuild: their launches returned at once; launches[] counts them, so a mean kernel time is
                              ms_kernel_sum[k] / (launches[k] - discarded_steps) for the per-step kernels (slots 1..5) */
} nepmi_stats;
/* Synchronises the stream.  with_lists != 0 also recounts the per-step list lengths. */
int nepmi_engine_stats(nepmi_engine* e, int with_lists, nepmi_stats* out);
/* Which kernel forms the LAST force evaluation ran, as a short text (e.g. "shape=PbTe-A window=static lanes=1 radial=win2
 * ann=fused_fp32 angular_force=lane_pairs force=lds_rows"): the counted rules of the engine made visible (bench.py prints it
 * as config.kernel_forms).  Returns the length written (without the terminator) or a negative status. */
int nepmi_engine_describe(nepmi_engine* e, char* buf, int len);
/* HIP-event timing on the engine's stream.  on = 1: every kernel and region (two event records per launch: the
 * kernels no longer run back to back, about 5 % slower steps); on = 2: the force-assembly kernel only; on = 16 + k: only the
 * kernel of slot k of nepmi_stats::ms_kernel_sum (what bench.py keeps inside its timed region for the roofline figure: the
 * slot of the step's longest kernel); 0: off. */
int nepmi_engine_set_timing(nepmi_engine* e, int on);
/* Form of the force assembly (find_force_radial + gpu_find_force_many_body: nep.cu:661-772, potential.cu:170-297).
 *   gather  : every lane evaluates both halves of its pairs, f12 - f21, the partner's half from rows gathered from the
 *             partner -- per-atom virials in the reference's attribution (W_i = sum_j r_ij (x) f21);
 *   scatter : every lane evaluates its own half only and adds the reaction to the partner's slot of a fixed-point accumulator
 *             over the brick's LDS window (the reference's small-box formulation, nep_small_box.cuh:473-478, made local and
 *             deterministic); forces, energies and the TOTAL virial are the same to f32 rounding, the per-atom virial planes
 *             hold the own-half form until a virial-only pass of the gather form replaces them -- the engine runs that pass
 *             itself whenever per-atom virials leave it.  Static window layout, one lane per atom, one or two types.
 * mode -1 (default): the fused run loops (nepmi_run_*, nepmi_dist_*) take the scatter form where it applies, the per-call
 * entry points (nepmi_potential_compute, nepmi_force_compute) the gather form; 0: gather everywhere; 1: scatter wherever it
 * applies (per-call evaluations then add the virial-only pass).  Range: the fixed-point sums hold +-512 eV/A net per atom (they
 * are modular, so only the net has to fit).  A pair half beyond 64 eV/A or a net force component beyond 128 eV/A returns the engine
 * to the gather form for the rest of its life, and the evaluation that met it does not stand: a per-call evaluation (one-call or
 * begin/end form) is repeated in the gather form before it returns, a step of a single-domain run loop freezes like a skin trip and
 * is re-run.  In a decomposed run (nepmi_dist_*) the flag travels with the skin vote (one reduction of three words per step), every
 * rank leaves the scatter form at the same step -- at most twelve steps later (the host looks at the voted words every fourth
 * step, two looks in flight) -- and the steps in between stand: they are exact while every value stays inside the sums, and a pair
 * half or a net component beyond 256 eV/A met in that window is an ERROR on every rank (NEPMI_ERR_STATE at the next look), never a
 * silent wrap.  nepmi_dist_compute checks the flag before it returns (one 4-byte reduction) and repeats the evaluation in the
 * gather form on every rank.  Not seen by either guard: a net beyond 768 eV/A made of a dozen or more aligned pair halves that
 * each stay under 64 eV/A. */
int nepmi_engine_set_force_form(nepmi_engine* e, int mode);
/* What the per-call entry points (nepmi_potential_compute, nepmi_force_compute, the _levels forms) owe the caller in the virial
 * planes.  mode 0 (default): per-atom virials in the reference's attribution, W_i = sum_j r_ij (x) f_21 (potential.cu:203-296) --
 * what compute_hac / compute_hnemd / dump_xyz ... virial read; the gather form of the force assembly provides it.  mode 1: only
 * the TOTAL has to be right (Ensemble::find_thermo, dump_thermo: ensemble.cu:434-633 sums the planes): the per-call evaluations
 * then follow the run loops' rule and take the scatter form where it applies (systems of more than 512 bricks), whose planes hold the own-half
 * attribution -- the same sum, the same forces and energies to f32 rounding, a third less time per call at a million atoms
 * (bench.py: pbte_per_call_dropin / _totals).  A host sets 1 while no consumer of per-atom virials is active. */
int nepmi_engine_set_virial_mode(nepmi_engine* e, int mode);
/* Experiment and test switches of an engine, by name (one entry point instead of a setter per switch: these select between
 * kernel forms that give the same results, or narrow a guard for a test; production uses the defaults).  Returns NEPMI_ERR_ARG
 * for an unknown name or a value outside the switch's range.
 *
 *   "generic": Force the run-time-shaped (generic) kernel instantiation instead of a model-shape-specialised
 *       one; used by the parity tests to cover both code paths with one model.
 *   "tiles": LDS-window kernels: 0 = none (plain gather kernel for the radial pass, pair records for the force
 *       assembly); anything else (the default) = the radial pass and the force assembly both work from the LDS position
 *       window.  The choice is a rule, never a timing: the same input always runs the same kernels.  The window
 *       kernels are dropped automatically when a periodic direction has fewer than 8 cells, a brick's window does not
 *       fit LDS or an atom sits far outside the box along an open direction.  Both give identical lists and forces to
 *       f32 rounding.
 *   "win_lanes": Lanes per atom of the LDS-window kernels: 0 (default) = by the number of bricks (4 up to 256 bricks, 2 up to 512,
 *       else 1: small systems are bound by the latency of one workgroup); 1, 2, 4 pin it.
 *   "win_max_atoms": Capacity of a brick's LDS window in atoms: 0 (default) = the rule (6,656 for shapes with type-pure list streams,
 *       5,000 for many-type and run-time shapes: beyond it the gather kernels serve the model); a value pins it (at most 6,656).
 *   "scatter_guard / scatter_guard_hard": Test hook: the guard band of the scatter form per pair half in eV/A (default and maximum 64; the net-force guard is twice the
 *       value), so that the hand-over can be exercised with ordinary forces; hard_factor: the hard limit of decomposed runs as a
 *       multiple of the band (<= 0: the default 4; the limit never exceeds 256 eV/A).
 *   "scatter_guard_delay": test hook: the NEXT "scatter_guard" takes effect at the value-th force assembly after it (to trip the band
 *       at a chosen step of a run loop).
 *   "radial_mask": The per-step radial list of the scatter-form steps of the run loops (find_neighbor_list_large_box, nep.cu:436-486, is what it
 *       replaces): value = 1: one inside bit per candidate of the packed Verlet words, which the force assembly walks with the bits as
 *       weights -- no compacted list is written (a conditional 2-byte store per pair and its bookkeeping: a third of the radial
 *       pass's time); 0 (default): the compacted list on every step.  Same pairs, same per-pair arithmetic: identical trajectories
 *       bit for bit (tests/test_gpu_parity.py).  Measured on PbTe 1 M atoms the radial pass gains 0.07 ms and the force assembly,
 *       which then evaluates the 24 % of the candidates outside the cutoff in lockstep, loses as much (profiles/r4q_ab_mask.txt);
 *       carbon gains 2 %.  One or two atom types; the compacted list is rebuilt on demand when per-atom virials leave the engine.
 *   "angular_fused": Angular descriptor, per-atom ANN and partial angular forces (the angular half of find_descriptor, nep.cu:549-640;
 *       apply_ann_one_layer, nep_utilities.cuh:169-194; find_partial_force_angular, nep.cu:774-861; find_force_ZBL, nep.cu:863-975) in
 *       ONE kernel with two lanes per atom: value = 1 (default) wherever the descriptor + ANN fusion applies (compiled shapes, at most 4
 *       types, fewer than 9 angular channels) -- the sums s_{n,lm} stay in the registers across the ANN and become the adjoint table in
 *       place, where the separate kernels evaluate them twice; 0: the separate kernels.  Same results up to the summation order of
 *       the ANN's dot products (tests/test_gpu_parity.py).
 *   "brick_force": ... and the scatter-form force assembly (find_force_radial, nep.cu:661-772; gpu_find_force_many_body, potential.cu:170-297) in
 *       the SAME kernel, one 512-thread workgroup per brick behind the radial pass: value = 1 where the fused angular kernel and the
 *       scatter form both apply, on shapes with two register-resident atom types, in single-domain engines; the partial forces and
 *       the per-atom radial table then never reach HBM (the virial-only pass of the gather form runs the separate angular kernel
 *       first when per-atom virials leave the engine).  0 (default): the separate kernels -- measured faster on MI355X (PbTe 1 M atoms:
 *       0.44 + 0.25 ms against 0.95 ms; gpumd_amd/csrc/nep_brick.h says why).  Same results to FP32 rounding.
 *   "win_static": Static window layout of the one-lane window kernels (default on): between two list rebuilds the LDS slot of every window
 *       atom is fixed, so the rebuild tabulates the windows and stores the Verlet entries as LDS slots, four to an 8-byte word
 *       (two-type models: list B as two type-pure streams); value = 0 keeps the per-launch scan of the window cells and the
 *       (window cell, rank) codes.  Same lists bit for bit, sums differ by their order only.  Forces a list rebuild.
 *   "stepwise_loops": Test hook: value = 1 makes nepmi_run_nvt_lan / nepmi_run_nvt_bao run as the plain sequence of the per-call steps on the caller's
 *       arrays (what they were before they became device-resident loops); the resident forms reproduce it bit for bit.
 *   "mfma": How the per-atom ANN runs.  value = 1 (default): inside the angular-descriptor kernel where the shape allows it (one
 *       lane per atom, at most 4 types: the descriptor never leaves the registers), else the matrix-core
 *       (v_mfma_f32_32x32x2_f32) ANN kernel; value = 2: the matrix-core kernel wherever it applies; value = 0: the per-atom ANN
 *       kernel, which is also taken automatically for models with more than 4 types, more than 128 neurons or more than
 *       128 descriptor + radial-table rows.  The three differ by f32 summation order only.  Forces a list rebuild (the
 *       work order of the descriptor columns follows the mode).
 *   "angular_recompute": Angular s_{n,lm} sums between the angular descriptor and angular force kernels: value 0 = stored
 *       ((n_a+1)*24 floats per atom through HBM), 1 = rebuilt in the force kernel from the compact pair
 *       records, -1 (default) = rebuilt when the model has few angular neighbours (MN_angular <= 16).
 *       Both give bit-identical results.
 *   "radial_sync": the per-step radial list of the scatter-form steps as WAVE-SYNCHRONOUS words (default 1): the accepted LDS
 *       slots wait in a short queue in the lane's registers and every lane of a wavefront stores one 8-byte word of four at the
 *       same time (a lane with fewer than four pads with the sentinel slot) -- whole 512-byte rows instead of 2-byte stores at
 *       per-lane rows (gpumd_amd/csrc/nep_window.h: SyncFifo); 0: the slot-major compact list.  Same pairs, integer sums:
 *       identical forces bit for bit (tests/test_gpu_parity.py).  One or two atom types. */
int nepmi_engine_set_option(nepmi_engine* e, const char* name, double value);
/* Temperature-dependent NEP (nep4[_zbl]_temperature): the `temperature` argument of
 * NEP::compute(const float temperature, Box&, ...) (src/force/nep.cuh:126, nep.cu:1813-1856; Force::compute passes
 * its own `temperature`, advanced by delta_T before every compute of a run, force.cu:803).  It stays in force for
 * every later compute / nepmi_run_nve call of this engine; the default is 0 K.  The thermostatted run loops
 * (nepmi_run_nvt_*, nepmi_dist_run) set it themselves like Run::parse_run + Force::compute do: step s of a run from
 * t1 to t2 sees t1 + (s + 2) (t2 - t1) / nsteps (the initial force call of the run has already advanced it once).
 * The extra ANN input q[dim] = temperature * q_scaler[dim] (nep.cu:1483-1486) is folded into the hidden-layer bias, so the kernels are the ones of a plain model.  For a model
 * of any other type the call is accepted and has no effect (Potential::compute(temperature, ...) falls back to the
 * plain overload, potential.cuh:46-56). */
int nepmi_engine_set_temperature(nepmi_engine* e, double temperature);

#ifdef __cplusplus
}
#endif
#endif /* NEPMI_H */
