// Engine: owns the device state of one NEP potential instance and sequences the kernels.
// Host logic only; templated on the backend so that tests/emu can drive the same sequencing
// with the host-loop backend.  The product (engine.hip) instantiates it with HipBackend only.
//
// Replaces (reference, src/force): NEP::NEP allocation half (nep.cu:379-391), Neighbor
// (neighbor.cu:741-833), NEP::compute_large_box (nep.cu:996-1137).
#pragma once
#include "nep_bodies.h"
#include "nep_md.h"
#include "nep_model.h"
#include "nep_window.h"
#include "nep_highl.h"
#include "tersoff_bodies.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <random>
#include <vector>

#ifndef NEPMI_TERSOFF_SEAM
#define NEPMI_TERSOFF_SEAM 1 // A/B switch: 0 = the Tersoff force assembly always as its own launch
#endif
#ifndef NEPMI_WIN2_DEFAULT
#define NEPMI_WIN2_DEFAULT 1 // A/B switch (profiles/ab_variants.sh): 0 = the scanned window layout unless asked for
#endif

// NEPMI_SHAPE_DISPATCH(fn, generic, (args)): fn<S>(args) for the compiled shape this engine selected (shape_), or for the
// run-time shape (generic = 1) / nothing (0).  A JIT core (capi_jit.h: the same sources compiled once more for ONE further
// shape, -DNEPMI_JIT_SHAPE=n_r,k_r,n_a,k_a,n_L,types -DNEPMI_JIT_CORE) knows that shape and the run-time shape only.
#define NEPMI_UNPAREN(...) __VA_ARGS__
#if defined(NEPMI_JIT_CORE)
// (a core serves models of its own shape only -- nepmi_model_load hands it nothing else -- so it does not carry the run-time-shape
// kernels, the slowest part of the library to compile; nepmi_engine_set_generic is refused there)
#define NEPMI_SHAPE_DISPATCH(FN, GENERIC, ARGS)                                   \
  switch (shape_) {                                                               \
    case 6: FN<S_JIT>(NEPMI_UNPAREN ARGS); break;                                 \
    default: if (GENERIC) throw EngineError{-4, "a JIT core carries the kernels of its own shape only"}; break; \
  }
#else
#define NEPMI_SHAPE_DISPATCH(FN, GENERIC, ARGS)                                   \
  switch (shape_) {                                                               \
    case 1: FN<S_PbTeA>(NEPMI_UNPAREN ARGS); break;                               \
    case 2: FN<S_PbTeB>(NEPMI_UNPAREN ARGS); break;                               \
    case 3: FN<S_C2022>(NEPMI_UNPAREN ARGS); break;                               \
    case 4: FN<S_UNEP>(NEPMI_UNPAREN ARGS); break;                                \
    case 5: FN<S_BZO>(NEPMI_UNPAREN ARGS); break;                                 \
    case 7: FN<S_COV1>(NEPMI_UNPAREN ARGS); break;                                \
    case 8: FN<S_COV2>(NEPMI_UNPAREN ARGS); break;                                \
    case 9: FN<S_COV3>(NEPMI_UNPAREN ARGS); break;                                \
    default: if (GENERIC) FN<ShapeGeneric>(NEPMI_UNPAREN ARGS); break;            \
  }
#endif

namespace nepmi {

// Does a compiled kernel shape serve this model?  l_max_3body < 4 and the optional 4-body rows live in the run-time shape only.
inline bool shape_is_compilable(const NepModel& m)
{
  return m.kind == 0 && m.L_max == 4 && !m.has_q_112 && !m.has_q_123 && !m.has_q_233 && !m.has_q_134;
}
template <class S>
inline bool model_matches_shape(const NepModel& m)
{
  if (!S::fixed)
    return true;
  if (!shape_is_compilable(m))
    return false;
  return S::NR == m.n_max_radial && S::KR == m.basis_size_radial && S::NA == m.n_max_angular &&
         S::KA == m.basis_size_angular && S::NL == m.num_L && (S::TS == 0 || S::TS == m.num_types);
}
// the shapes every build of the library carries (EngineT::select_shape numbers them 1..5; 0: none of them)
inline int builtin_shape_of(const NepModel& m)
{
  if (model_matches_shape<Shape<6, 6, 6, 6, 5, 2>>(m)) return 1;
  if (model_matches_shape<Shape<4, 8, 4, 8, 5, 2>>(m)) return 2;
  if (model_matches_shape<Shape<10, 10, 8, 8, 6, 1>>(m)) return 3;
  if (model_matches_shape<Shape<4, 8, 4, 8, 6, 0>>(m)) return 4;
  if (model_matches_shape<Shape<8, 8, 6, 8, 5, 0>>(m)) return 5;
  if (model_matches_shape<Shape<8, 12, 8, 12, 6, 2>>(m)) return 9; // (two types: the cover with type-pure lists, before the any-types one)
  if (model_matches_shape<Shape<8, 12, 8, 12, 6, 0>>(m)) return 7;
  if (model_matches_shape<Shape<12, 16, 10, 12, 6, 0>>(m)) return 8;
  return 0;
}
// a model whose compiled shape keeps type-pure list segments (Shape::TS > 0)
inline bool served_by_type_pure_shape(const NepModel& m)
{
#if defined(NEPMI_JIT_CORE)
  using SJ = Shape<NEPMI_JIT_SHAPE>;
  return SJ::TS > 0 && model_matches_shape<SJ>(m);
#else
  const int bs = builtin_shape_of(m);
  return (bs >= 1 && bs <= 3) || bs == 9;
#endif
}
// The COVER shapes (7, 8: any number of types, all six invariant rows): a model of a shape nobody compiled kernels for is
// zero-padded into the smallest one that holds it (nep_model.h: embed_model) instead of falling to the run-time-shape kernels.
// n_r, k_r, n_a, k_a of the cover, or false.
inline bool cover_shape_for(const NepModel& m, int out[4])
{
  static const int covers[2][4] = {{8, 12, 8, 12}, {12, 16, 10, 12}};
  if (m.kind != 0 || m.L_max < 1 || m.L_max > 4 || m.has_q_112 || m.has_q_123 || m.has_q_233 || m.has_q_134)
    return false;
  for (const auto& c : covers)
    if (m.n_max_radial <= c[0] && m.basis_size_radial <= c[1] && m.n_max_angular <= c[2] && m.basis_size_angular <= c[3]) {
      for (int i = 0; i < 4; ++i)
        out[i] = c[i];
      return true;
    }
  return false;
}


enum KernelSlot {
  kSlotGather = 0,
  kSlotRadial = 1,
  kSlotAngular = 2, // angular descriptor
  kSlotAnn = 3,
  kSlotAngForce = 4,
  kSlotForce = 5,
  kSlotVV = 6,
  kSlotThermo = 7,
  kSlotRebuild = 8,
  kSlotMisc = 9,
  kNumSlots = 10
};
enum RegionSlot { kRegionRebuild = 0, kRegionForce = 1, kNumRegions = 2 };

struct EngineError {
  int code;
  std::string msg;
};

inline void box_from_h9(const double h9[9], const int pbc[3], BoxD& box)
{
  // Box::get_inverse / get_volume / get_area / set_is_orthogonal, src/model/box.cu:23-117
  double* h = box.h;
  for (int k = 0; k < 9; ++k)
    h[k] = h9[k];
  h[9] = h[4] * h[8] - h[5] * h[7];
  h[10] = h[2] * h[7] - h[1] * h[8];
  h[11] = h[1] * h[5] - h[2] * h[4];
  h[12] = h[5] * h[6] - h[3] * h[8];
  h[13] = h[0] * h[8] - h[2] * h[6];
  h[14] = h[2] * h[3] - h[0] * h[5];
  h[15] = h[3] * h[7] - h[4] * h[6];
  h[16] = h[1] * h[6] - h[0] * h[7];
  h[17] = h[0] * h[4] - h[1] * h[3];
  const double det = h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) +
                     h[2] * (h[3] * h[7] - h[4] * h[6]);
  for (int k = 9; k < 18; ++k)
    h[k] /= det;
  for (int k = 0; k < 18; ++k)
    box.hf[k] = (float)h[k];
  for (int d = 0; d < 3; ++d)
    box.pbc[d] = pbc[d] ? 1 : 0;
  box.ortho = h[1] == 0 && h[2] == 0 && h[3] == 0 && h[5] == 0 && h[6] == 0 && h[7] == 0;
  box.volume = std::fabs(det);
  auto cross_norm = [](const double* a, const double* b) {
    const double s1 = a[1] * b[2] - a[2] * b[1];
    const double s2 = a[2] * b[0] - a[0] * b[2];
    const double s3 = a[0] * b[1] - a[1] * b[0];
    return std::sqrt(s1 * s1 + s2 * s2 + s3 * s3);
  };
  const double a[3] = {h[0], h[3], h[6]}, bb[3] = {h[1], h[4], h[7]}, c[3] = {h[2], h[5], h[8]};
  box.thickness[0] = box.volume / cross_norm(bb, c);
  box.thickness[1] = box.volume / cross_norm(c, a);
  box.thickness[2] = box.volume / cross_norm(a, bb);
  // a displacement shorter than this has all fractional components below 1/2: the fractional minimum
  // image leaves it where it is
  double tmin = 1.0e30;
  for (int d = 0; d < 3; ++d)
    if (box.pbc[d] && box.thickness[d] < tmin)
      tmin = box.thickness[d];
  box.rfree2 = tmin < 1.0e29 ? (float)(0.49 * tmin * 0.49 * tmin) : 3.0e38f;
}

template <class B>
class EngineT
{
public:
  static constexpr double kSkin = 1.0; // neighbor.cuh:212
  // workgroup of the angular descriptor kernel: the coefficient table is staged once per workgroup; measured on PbTe 1M
  // 64: 0.160, 128: 0.162, 256: 0.150, 512: 0.137 ms (1024 would halve the register budget: 1.0 ms); the angular force
  // kernel does not care (64)
  static constexpr int kAngDescBlock = 512;
  static constexpr int kAngFusedBlock = 256; // descriptor + ANN: ~150 registers per lane

  EngineT(const NepModel& model, int64_t n_atoms, B backend) : model_(model), be_(backend), cap_(n_atoms), N_(n_atoms)
  {
    std::memset(&b_, 0, sizeof(b_));
    b_.scatter_limit = 64.0f;   // nep_scatter.h: kScatterFlagLimit
    b_.fold_guard = 1 << 29;    // kFoldGuard
    b_.scatter_hard = 0.0f;     // set_flagged_steps_stand
    b_.fold_hard = 0;
    std::memset(&md_, 0, sizeof(md_));
    std::memset(&box_, 0, sizeof(box_));
    try {
      upload_model();
      allocate();
    } catch (...) { // no destructor runs for a throwing constructor
      for (void* p : allocs_)
        be_.free(p);
      throw;
    }
  }

  ~EngineT()
  {
    for (void* p : allocs_)
      be_.free(p);
  }

  B& backend() { return be_; }
  const BoxD& box() const { return box_; }
  double* thermo_scratch() { return thermo_scratch_; }
  void check_overflow_public(const int* flags) { check_overflow(flags); }
  const NepModel& model() const { return model_; }
  const Bufs& bufs() const { return b_; }
  int64_t num_atoms() const { return N_; }
  int64_t num_compute = 0, num_rebuild = 0, num_discarded = 0;
  int64_t num_range_handovers = 0; // times the scatter form was left for the gather form (guard band of its fixed-point sums)
  // kPhaseBoundaryRadial / kPhaseAfterRadial: kPhaseBoundary in two parts, so that the boundary bricks' radial pass can be
  // enqueued on the communication stream right behind the ghost unpack (force_kernels_on) and run beside the tail of the
  // interior launch instead of after it
  enum Phase { kPhaseAll = 0, kPhaseInterior = 1, kPhaseBoundary = 2, kPhaseRecords = 3, kPhaseBoundaryRadial = 4,
               kPhaseAfterRadial = 5, kPhaseRadialOnly = 6 };

  // Potential::compute (adds to pe/force/virial; positions already wrapped)
  void potential_compute(
    const double h9[9], const int pbc[3], int64_t n, const int* type, const double* pos, double* pe,
    double* force, double* virial)
  {
    potential_compute_levels(h9, pbc, n, type, pos, nullptr, pe, force, virial);
  }

  // The same on a LOCAL system of a domain decomposition: n <= capacity atoms, of which only those
  // with level 2 (owned) receive forces; level 1 (inner ghosts) get descriptors and partial
  // forces, level 0 (outer ghosts) only lend their positions.  Mirrors the N1..N5 ranges of
  // NEP_MULTIGPU (src/force/nep_multigpu.cuh:42-50).  level == nullptr: every atom is owned.
  void potential_compute_levels(
    const double h9[9], const int pbc[3], int64_t n, const int* type, const double* pos,
    const signed char* level, double* pe, double* force, double* virial)
  {
    if (!prepare_lists(h9, pbc, n, type, pos, level)) { // small box: NEP::compute -> compute_small_box
      small_box_compute(type, pos);
      scatter_add(pe, force, virial);
      ++num_compute;
      return;
    }
    force_kernels(kPhaseAll);
    redo_outside_scatter_range();
    scatter_add(pe, force, virial);
    ++num_compute;
  }

  // A force evaluation in the scatter form met a value beyond the guard band of its fixed-point sums (nep_scatter.h): the gather
  // form, which has no such limit, evaluates the same positions again and takes over for the rest of the run.  One read of the
  // flag word per evaluation in that form (systems of >= kScatterMinBricks bricks).
  void redo_outside_scatter_range()
  {
    if (!last_scatter_form_ || scatter_disabled_)
      return;
    int flag = 0;
    be_.d2h(&flag, b_.flags + kFlagRange, sizeof(int));
    if (!flag)
      return;
    scatter_disabled_ = true;
    ++num_range_handovers;
    be_.memset(b_.flags + kFlagRange, 0, sizeof(int));
    force_kernels(kPhaseAll);
  }

  // Neighbor::find_neighbor_global (neighbor.cu:741-800): gather the caller's positions into internal order,
  // run the skin check, rebuild the Verlet lists when needed.  Returns false when the box takes the small-box
  // branch (no lists kept).
  bool prepare_lists(
    const double h9[9], const int pbc[3], int64_t n, const int* type, const double* pos, const signed char* level)
  {
    split_pending_ = false;
    if (n < 1 || n > cap_)
      throw EngineError{-4, "number of atoms exceeds the engine's capacity"};
    BoxD box;
    box_from_h9(h9, pbc, box);
    bool need_rebuild = !have_list_ || n != N_;
    N_ = n;
    b_.N = n;
    b_.level = level;
    if (model_.kind == 1 && is_small_box(box))
      throw EngineError{-7, "Tersoff-1989: box thickness <= 2.5 (rc + skin) in a periodic direction is not supported"};
    if (is_small_box(box)) { // NEP::compute -> compute_small_box (nep.cu:1356-1389)
      box_ = box;
      have_list_ = false;
      last_small_ = true;
      return false;
    }
    last_small_ = false;
    if (have_list_ && !same_box(box))
      need_rebuild = true;
    box_ = box;
    if (!need_rebuild) {
      be_.memset(b_.flags + kFlagMoved, 0, sizeof(int));
      CheckGatherBody cg{box_, b_, pos, 0};
      be_.template launch<128>(kSlotGather, N_, cg);
      if (!external_skin_) { // the one host round trip of a per-call force evaluation (neighbor.cu:752 does the same)
        int flags[kNumFlags];
        be_.d2h(flags, b_.flags, sizeof(flags));
        check_overflow(flags);
        if (flags[kFlagMoved])
          need_rebuild = true;
      }
    }
    if (need_rebuild)
      rebuild(type, pos);
    return true;
  }

  // Potential::compute adds to the caller's arrays: caller[perm[k]] += internal[k]
  void scatter_add(double* pe, double* force, double* virial)
  {
    be_.template launch<256>(kSlotMisc, N_, ScatterAddBody{b_, pe, force, virial});
  }

  // Split form for a domain-decomposed host that overlaps its ghost exchange with compute:
  //   begin: positions of the OWNED atoms are final, ghost entries of `pos` may still be in flight
  //          (they are not read): skin check + gather of the owned atoms and the radial pass of the
  //          interior bricks (no ghost in the brick's window) are enqueued.  Returns false when
  //          nothing could be started (no valid list / geometry changed / no LDS-window pass);
  //   end:   all of `pos` is final: ghosts are gathered and checked, then the boundary bricks and
  //          the rest of the force path run.  If the skin check asks for a rebuild the interior
  //          work is discarded and everything is redone on the new list.
  bool potential_compute_levels_begin(
    const double h9[9], const int pbc[3], int64_t n, const int* type, const double* pos,
    const signed char* level, double* pe, double* force, double* virial)
  {
    (void)type;
    (void)pe;
    (void)force;
    (void)virial;
    split_pending_ = false;
    if (n < 1 || n > cap_)
      throw EngineError{-4, "number of atoms exceeds the engine's capacity"};
    BoxD box;
    box_from_h9(h9, pbc, box);
    if (!have_list_ || n != N_ || model_.kind != 0 || !tile_ok_ || is_small_box(box) || !same_box(box) ||
        level != b_.level)
      return false;
    be_.memset(b_.flags + kFlagMoved, 0, sizeof(int));
    CheckGatherBody cg{box_, b_, pos, 1};
    be_.template launch<128>(kSlotGather, N_, cg);
    force_kernels(kPhaseInterior);
    split_pending_ = true;
    return true;
  }

  void potential_compute_levels_end(
    const double h9[9], const int pbc[3], int64_t n, const int* type, const double* pos,
    const signed char* level, double* pe, double* force, double* virial)
  {
    if (!split_pending_) {
      potential_compute_levels(h9, pbc, n, type, pos, level, pe, force, virial);
      return;
    }
    split_pending_ = false;
    if (n != N_)
      throw EngineError{-4, "compute_levels_end: atom count differs from compute_levels_begin"};
    CheckGatherBody cg{box_, b_, pos, 2};
    be_.template launch<128>(kSlotGather, N_, cg);
    bool moved = false;
    if (!external_skin_) {
      int flags[kNumFlags];
      be_.d2h(flags, b_.flags, sizeof(flags));
      check_overflow(flags);
      moved = flags[kFlagMoved] != 0;
    }
    if (moved) {
      rebuild(type, pos);
      force_kernels(kPhaseAll);
    } else {
      force_kernels(kPhaseBoundary);
    }
    redo_outside_scatter_range(); // (the split form can take the scatter form too: set_force_form(1))
    scatter_add(pe, force, virial);
    ++num_compute;
  }

  bool same_box(const BoxD& box) const
  {
    for (int k = 0; k < 9; ++k)
      if (box.h[k] != box_.h[k])
        return false;
    for (int d = 0; d < 3; ++d)
      if (box.pbc[d] != box_.pbc[d])
        return false;
    return true;
  }

  void apply_pbc(const double h9[9], const int pbc[3], int64_t n, double* pos)
  {
    BoxD box;
    box_from_h9(h9, pbc, box);
    ApplyPbcBody body{box, n, pos};
    be_.template launch<256>(kSlotMisc, n, body);
  }

  void zero_properties(int64_t n, double* pe, double* force, double* virial)
  {
    ZeroPropsBody body{n, pe, force, virial};
    be_.template launch<256>(kSlotMisc, n, body);
  }

  void average_properties(int64_t n, double denominator, double* pe, double* force, double* virial)
  {
    AveragePropsBody body{n, denominator, pe, force, virial};
    be_.template launch<256>(kSlotMisc, n, body);
  }

  void velocity_verlet(
    bool step1, int64_t n, double dt, const double* mass, const double* force, double* pos, double* vel,
    const BoxD* wrap_box)
  {
    VelocityVerletBody body;
    std::memset(&body, 0, sizeof(body));
    body.N = n;
    body.dt = dt;
    body.is_step1 = step1 ? 1 : 0;
    body.fuse_wrap = wrap_box ? 1 : 0;
    if (wrap_box)
      body.box = *wrap_box;
    body.mass = mass;
    body.force = force;
    body.pos = pos;
    body.vel = vel;
    body.unwrapped = step1 ? unwrapped_ : nullptr;
    be_.template launch<256>(kSlotVV, n, body);
  }

  void find_thermo(
    int64_t n, double volume, const double* mass, const double* pe, const double* vel,
    const double* virial, double* thermo8)
  {
    be_.thermo(kSlotThermo, n, volume, mass, pe, vel, virial, thermo8, thermo_scratch_);
  }

  void berendsen(int64_t n, double temperature, double coupling, const double* thermo8, double* vel)
  {
    if (coupling > 1.0e-5) { // ensemble_ber.cu:223
      BerendsenBody body{n, temperature, coupling, thermo8, vel};
      be_.template launch<256>(kSlotMisc, n, body);
    }
  }

  // Ensemble_NHC (ensemble_nhc.cu): chain state in caller-owned device memory (kNhcStateSize doubles)
  void nhc_init(int64_t n, double temperature, double t_coup, double dt, double* state)
  {
    be_.template launch<64>(kSlotMisc, 1, NhcInitBody{n, temperature, t_coup, dt, state});
  }
  // one thermostat half-step: chain update from thermo8[0], then v *= factor (both on the device)
  void nhc_half_step(int64_t n, double temperature, double dt, const double* thermo8, double* state, double* vel)
  {
    be_.template launch<64>(kSlotMisc, 1, NhcChainBody{n, temperature, 0.5 * dt, thermo8, state});
    be_.template launch<256>(kSlotMisc, n, ScaleVelocityBody{n, state + 3 * kNhcLinks, vel});
  }

  // ---- Langevin thermostat (Ensemble_LAN, ensemble_lan.cu): one generator state per atom, in the CALLER's atom order
  //      (the reference initialises state n with hiprand_init(seed, n, 0)); seed = rand() there (ensemble_lan.cu:39) ----
  void lan_seed(int seed)
  {
    lan_seed_ = seed;
    lan_fresh_ = true;
  }
  void lan_prepare(int64_t n)
  {
    if (n < 1 || n > cap_)
      throw EngineError{-4, "number of atoms exceeds the engine's capacity"};
    if (!lan_states_) {
      lan_states_ = dalloc<char>(be_.lan_state_bytes() * (size_t)cap_);
      lan_sums_ = dalloc<double>(4);
    }
    if (lan_fresh_) {
      be_.lan_init(lan_states_, n, lan_seed_);
      lan_fresh_ = false;
    }
  }
  // integrate_nvt_lan_half (ensemble_lan.cu:96-127): v <- c1 v + c2 sqrt(1/m) xi, then the centre-of-mass velocity is removed
  void lan_half_step(int64_t n, double temperature, double t_coup, const double* mass, double* vel)
  {
    lan_prepare(n);
    const double c1 = std::exp(-0.5 / t_coup);
    const double c2 = std::sqrt((1.0 - c1 * c1) * kBoltzmann * temperature);
    be_.lan_half(lan_states_, n, c1, c2, mass, vel, lan_sums_);
  }
  // the O step of the BAOAB integrator (Ensemble_BAO::integrate_nvt_lan, ensemble_bao.cu:30-41, :87-117): the same
  // kernels with c1 = exp(-1 / T_coup) over a whole step
  void bao_o_step(int64_t n, double temperature, double t_coup, const double* mass, double* vel)
  {
    const double keep = t_coup;
    lan_prepare(n);
    const double c1 = std::exp(-1.0 / keep);
    const double c2 = std::sqrt((1.0 - c1 * c1) * kBoltzmann * temperature);
    be_.lan_half(lan_states_, n, c1, c2, mass, vel, lan_sums_);
  }
  void half_drift(int64_t n, double dt, double* pos, const double* vel) // operator A
  {
    be_.template launch<256>(kSlotVV, n, HalfDriftBody{n, dt, pos, vel, unwrapped_});
  }

  // ---- Bussi-Donadio-Parrinello stochastic velocity rescaling (Ensemble_BDP, ensemble_bdp.cu:71-104;
  //      resamplekin & co., svr_utilities.cuh:28-122, after Bussi's reference code).  Like the
  //      reference the noise is drawn on the host from std::mt19937 through
  //      uniform_real_distribution<double>(0, 1), so a run is reproducible from its seed; this is the one
  //      thermostat with a host round trip per step (the kinetic energy is needed to draw the factor). ----
  void bdp_seed(uint64_t seed)
  {
    bdp_rng_ = std::mt19937((std::mt19937::result_type)seed);
    bdp_iset_ = 0;
    bdp_gset_ = 0.0;
  }
  double bdp_uniform()
  {
    std::uniform_real_distribution<double> rand1(0, 1);
    return rand1(bdp_rng_);
  }
  double bdp_gauss() // polar Box-Muller with one cached deviate (gasdev)
  {
    if (bdp_iset_) {
      bdp_iset_ = 0;
      return bdp_gset_;
    }
    double v1, v2, rsq;
    do {
      v1 = 2.0 * bdp_uniform() - 1.0;
      v2 = 2.0 * bdp_uniform() - 1.0;
      rsq = v1 * v1 + v2 * v2;
    } while (rsq >= 1.0 || rsq == 0.0);
    const double fac = std::sqrt(-2.0 * std::log(rsq) / rsq);
    bdp_gset_ = v1 * fac;
    bdp_iset_ = 1;
    return v2 * fac;
  }
  double bdp_gamma(int ia) // gamma deviate of integer order (gamdev)
  {
    double x;
    if (ia < 6) {
      x = 1.0;
      for (int j = 1; j <= ia; ++j)
        x *= bdp_uniform();
      return -std::log(x);
    }
    double e, y;
    do {
      do {
        double v1, v2;
        do {
          v1 = bdp_uniform();
          v2 = 2.0 * bdp_uniform() - 1.0;
        } while (v1 * v1 + v2 * v2 > 1.0);
        y = v2 / v1;
        const double am = ia - 1;
        const double sq = std::sqrt(2.0 * am + 1.0);
        x = sq * y + am;
        if (x > 0.0)
          e = (1.0 + y * y) * std::exp(am * std::log(x / am) - sq * y);
      } while (x <= 0.0);
    } while (bdp_uniform() > e);
    return x;
  }
  double bdp_sum_noises(int nn) // sum of nn squared gaussian deviates
  {
    if (nn == 0)
      return 0.0;
    if (nn == 1) {
      const double rr = bdp_gauss();
      return rr * rr;
    }
    if (nn % 2 == 0)
      return 2.0 * bdp_gamma(nn / 2);
    const double rr = bdp_gauss();
    return 2.0 * bdp_gamma((nn - 1) / 2) + rr * rr;
  }
  // new kinetic energy drawn from the canonical distribution's relaxation kernel (resamplekin)
  double bdp_resample(double kk, double sigma, int ndeg, double taut)
  {
    const double factor = taut > 0.1 ? std::exp(-1.0 / taut) : 0.0;
    const double rr = bdp_gauss();
    return kk + (1.0 - factor) * (sigma * (bdp_sum_noises(ndeg - 1) + rr * rr) / ndeg - kk) +
           2.0 * rr * std::sqrt(kk * sigma / ndeg * (1.0 - factor) * factor);
  }
  // integrate_nvt_bdp_2 after the velocity update: T = thermo8[0] (device) -> host, draw, rescale
  double bdp_factor(int64_t n_total, double T, double temperature, double t_coup)
  {
    const int ndeg = 3 * (int)n_total;
    const double ek = T * ndeg * kBoltzmann * 0.5;
    const double sigma = ndeg * kBoltzmann * temperature * 0.5;
    return std::sqrt(bdp_resample(ek, sigma, ndeg, t_coup) / ek);
  }
  double bdp_scale(int64_t n, double temperature, double t_coup, const double* thermo8, double* vel)
  {
    double T = 0.0;
    be_.d2h(&T, thermo8, sizeof(double));
    const double factor = bdp_factor(n, T, temperature, t_coup);
    be_.template launch<256>(kSlotMisc, n, ScaleVelocityConstBody{n, factor, vel});
    return factor;
  }

  // ---------------------------------------------------------------------------------------------
  // Fused run loops == Run::perform_a_run (src/main_gpumd/run.cu:250-318) for `ensemble nve`,
  // `nvt_ber` (ensemble_ber.cu:195-235), `nvt_nhc` (ensemble_nhc.cu:166-232) and `nvt_bdp`
  // (ensemble_bdp.cu:71-104); target temperature ramp of Integrate::compute2 (integrate.cu:341-344).
  //
  // While the loop runs the state lives in internal order (nep_md.h) and the steps are enqueued
  // speculatively: the skin check is a device-side word that freezes the state when it fires; the host looks
  // at it only every kPollEvery steps (and at thermo records), rebuilds the lists and resumes from the frozen
  // step.  The caller's arrays are read at entry and written at exit.
  // ---------------------------------------------------------------------------------------------
  enum Ensemble { kNve = 0, kBer = 1, kNhc = 2, kBdp = 3, kLan = 4, kBao = 5 };
  // (r4: 768 = 3 workgroups x 256 CUs.  r6, with the radial pass's lists as wave-synchronous words / rows, which only the scatter form reads:
  // 729 bricks 0.282 ms per step against 0.312 in the gather form, 512 bricks 0.230 against 0.239 with two lanes per atom
  // (profiles/r6l_size_rule.txt) -- every system the rule gives one lane per atom, i.e. more than 512 bricks)
  static constexpr int64_t kScatterMinBricks = 513;
  static constexpr int kPollEvery = 4; // steps between two snapshots of the device flags
  static constexpr int kPollDepth = 2; // snapshots in flight: the host runs 8-12 steps ahead of the device
  static constexpr int kPollEveryCalm = 16, kPollCalmSteps = 256; // ... and every 16 steps once the lists have stood for 256 (run_md)

  void run_md(
    int ens, const double h9[9], const int pbc[3], int64_t n, const int* type, const double* mass, double dt,
    int64_t nsteps, double t1, double t2, double tcoup, double* pos, double* vel, double* pe, double* force,
    double* virial, int64_t thermo_every, double* thermo_host)
  {
    BoxD box;
    box_from_h9(h9, pbc, box);
    if (n < 1 || n > cap_)
      throw EngineError{-4, "number of atoms exceeds the engine's capacity"};
    struct LoopCtx { // the steps of this loop need forces, energies and the total virial only
      EngineT& e;
      bool was;
      explicit LoopCtx(EngineT& e_) : e(e_), was(e_.loop_ctx_) { e.loop_ctx_ = true; }
      ~LoopCtx() { e.loop_ctx_ = was; }
    } loop_ctx(*this);
    if ((is_small_box(box) && model_.kind == 0) || (stepwise_loops_ && (ens == kLan || ens == kBao))) {
      // (set_stepwise_loops: the Langevin ensembles as the plain sequence of the per-call entry points on the caller's
      // arrays -- what the resident forms are checked against, bit for bit)
      run_md_small_box(ens, h9, pbc, n, type, mass, dt, nsteps, t1, t2, tcoup, pos, vel, pe, force, virial, thermo_every,
                       thermo_host);
      return;
    }
    if (nsteps <= 0)
      return;
    // entry: lists valid for the caller's positions, state imported
    prepare_lists(h9, pbc, n, type, pos, nullptr);
    resident_alloc();
    resident_import(vel, mass, pe, force, virial);
    if (ens == kNhc) {
      if (!nhc_dev_)
        nhc_dev_ = dalloc<double>(kNhcStateSize);
      if (nhc_fresh_) // a fresh chain per `run` (integrate.cu:85-92); it continues across the calls of one run
        nhc_init(n, t1, tcoup, dt, nhc_dev_);
      nhc_fresh_ = false;
    }
    if (!factor_dev_)
      factor_dev_ = dalloc<double>(1);
    if (ens == kLan || ens == kBao)
      lan_prepare(n);
    // Ensemble_LAN (ensemble_lan.cu:96-127, :206-262): one thermostat half-step on the internal velocities.  The generator
    // states stay in the caller's atom order (state perm[k] for internal atom k) and the momentum sums are formed in the
    // caller's order through the inverse permutation: the same numbers as the stepwise nepmi_lan_half_step, bit for bit.
    auto lan_half = [&](double target, bool whole_step = false) { // whole_step: the O of BAOAB, c1 = exp(-1 / T_coup)
      const double c1 = std::exp((whole_step ? -1.0 : -0.5) / tcoup);
      const double c2 = std::sqrt((1.0 - c1 * c1) * kBoltzmann * target);
      be_.lan_kick_resident(lan_states_, N_, c1, c2, b_.mi, b_.vi, b_.perm, b_.lvl, nullptr, b_.flags);
      be_.lan_momentum_resident(N_, b_.mi, b_.vi, b_.invp, b_.lvl, lan_sums_, b_.flags);
      be_.lan_momentum_fix_resident(N_, lan_sums_, b_.vi, b_.lvl, b_.flags);
    };
    const int* frozen = b_.flags + kFlagMoved;
    be_.memset(b_.flags + kFlagMoved, 0, sizeof(int));
    const int64_t compute0 = num_compute;
    auto tag_of = [](int64_t step) { return (int)(step % 1000000000) + 1; };
    auto target_of = [&](int64_t step) { return t1 + (t2 - t1) * ((double)step / (double)nsteps); };
    auto thermo_now = [&]() {
      be_.thermo(kSlotThermo, N_, box.volume, b_.mi, b_.fo, b_.vi, b_.fo + (int64_t)kOutW * N_, thermo_dev_, thermo_scratch_);
    };
    auto nhc_half = [&](double target) { // one thermostat half-step on the internal velocities
      thermo_now();
      be_.template launch<64>(kSlotMisc, 1, NhcChainBody{N_, target, 0.5 * dt, thermo_dev_, nhc_dev_, frozen});
      be_.template launch<256>(kSlotVV, N_, ResidentScaleBody{b_, nhc_dev_ + 3 * kNhcLinks, 1.0});
    };
    struct Snapshot {
      int ring;
    };
    std::vector<Snapshot> pending;
    int ring_next = 0;
    // a trip of the skin check at (0-based) step m, found at a sync or in a snapshot: the device state is frozen
    // right after that step's first half-step; rebuild the lists on it
    auto handle_trip = [&](int moved_tag, int64_t enqueued_through) -> int64_t {
      num_discarded += enqueued_through - ((int64_t)moved_tag - 1) + 1; // steps m .. enqueued_through ran as no-ops
      be_.sync();
      pending.clear();
      resident_export(pos, vel, nullptr, nullptr, nullptr);
      rebuild(type, pos);
      resident_import(vel, mass, nullptr, nullptr, nullptr);
      return (int64_t)moved_tag - 1;
    };
    int flags[kNumFlags];
    auto sync_and_check = [&]() -> int { // returns the trip tag (0: none)
      be_.sync();
      pending.clear();
      be_.d2h(flags, b_.flags, sizeof(flags));
      check_overflow(flags);
      return flags[kFlagMoved];
    };

    // rows of the thermo records of this call, on the device until the loop ends
    double* thermo_rows = nullptr;
    const int64_t nrec = (thermo_every > 0 && thermo_host) ? nsteps / thermo_every : 0;
    if (nrec > 0) {
      if (thermo_rows_cap_ < nrec) {
        dfree(thermo_rows_);
        thermo_rows_ = nullptr; // (an allocation that throws must not leave the freed pointer behind)
        thermo_rows_cap_ = 0;
        thermo_rows_ = dalloc<double>(8 * (size_t)nrec);
        thermo_rows_cap_ = nrec;
      }
      thermo_rows = thermo_rows_;
    }
    // A look costs a 32-byte copy and an event on the stream (7.6 us: 1.9 us per step of config 2's 28); a late look costs the steps
    // enqueued behind a trip, which run as no-ops.  While the lists have been standing for kPollCalmSteps the looks are taken every
    // kPollEveryCalm steps (r6x: Si 13,824 atoms 4.86e8 -> 5.22e8 atom-steps/s; PbTe 16,000 atoms, a rebuild every ~40 steps, loses
    // 5 % when it always looks that rarely -- it never gets there).  NEPMI_POLL_EVERY fixes the interval (A/B switch).
    static const int poll_fixed = std::getenv("NEPMI_POLL_EVERY") ? std::max(1, std::atoi(std::getenv("NEPMI_POLL_EVERY"))) : 0;
    int64_t calm_since = -calm_steps_; // the step the lists were last rebuilt at (negative: in an earlier call; prepare_lists above resets the count when it rebuilds)
    int64_t step = 0;
    // Temperature-dependent NEP under a thermostat: Force::temperature starts at t1 (run.cu:679-681) and EVERY
    // Force::compute of the run adds delta_T = (t2 - t1) / nsteps first (force.cu:803) -- the initial one included
    // (run.cu:232-241 calls the same overload), so the compute of step s sees t1 + (s + 2) delta_T.  NVE keeps what
    // set_temperature said.
    const bool temp_ramp = model_.temperature_model && ens != kNve && t1 != t2;
    if (model_.temperature_model && ens != kNve && temperature_ != t1)
      set_temperature(t1);
    bool resume_after_vv1 = false; // the pre-force phase of `step` has already run (replay after a rebuild)
    bool kick2_pending = false;    // NVE: the second half-kick of step - 1 rides on this step's first pass
    while (step < nsteps) {
      const double target = target_of(step);
      if (!resume_after_vv1) {
        if (ens == kNhc)
          nhc_half(target); // integrate_nvt_nhc_1: thermostat half-step before the first velocity-Verlet half
        if (ens == kLan)
          lan_half(target); // Ensemble_LAN::compute1
        if (ens == kBao) {
          // Ensemble_BAO::compute1 (ensemble_bao.cu:419-446): B A O A; the noise amplitude keeps the temperature the
          // ensemble was set up with (T1: its c2 is fixed in the constructor, :36)
          be_.template launch<256>(kSlotVV, N_, ResidentBaoBody{box_, b_, dt, 1, 0, tag_of(step)});
          lan_half(t1, true);
          be_.template launch<256>(kSlotVV, N_, ResidentBaoBody{box_, b_, dt, 2, 0, tag_of(step)});
        } else if (tersoff_deferred_) {
          be_.template launch<64>(kSlotVV, N_, TersoffSeamBody{TersoffAssembleBody{b_, tb_}, ResidentStepBody{box_, b_, dt, 1, 1, tag_of(step)}});
          tersoff_deferred_ = false;
        } else {
          be_.template launch<256>(kSlotVV, N_, ResidentStepBody{box_, b_, dt, kick2_pending ? 1 : 0, 1, tag_of(step)});
        }
      }
      resume_after_vv1 = false;
      kick2_pending = false;
      if (temp_ramp)
        set_temperature(t1 + (t2 - t1) * ((double)(step + 2) / (double)nsteps));
      const bool record = thermo_every > 0 && (step + 1) % thermo_every == 0;
      const bool last = step + 1 == nsteps;
      step_outputs_ = record || last; // per-atom energies and virials: read at thermo records and at the exit only
      b_.trip_tag = tag_of(step);     // (scatter form: a force beyond its fixed-point guard band freezes the loop at this step)
      tersoff_defer_ = NEPMI_TERSOFF_SEAM && model_.kind == 1 && ens == kNve && !record && !last;
      force_kernels(kPhaseAll, frozen);
      tersoff_defer_ = false;
      b_.trip_tag = 0;
      // A thermo record does not stop the pipeline: find_thermo's eight numbers go to row `rec` of a device buffer (a 64-byte copy on the
      // stream) and all rows come to the host when the loop ends -- the reference reduces thermo on EVERY step (ensemble_nve.cu:59-95),
      // so a record must not cost a host round trip.  (Measured, r6v: thermo every step 1.287 ms with or without the per-record
      // synchronisation -- the speculative enqueue already hid it; what a record step costs is its work: energies and virials written,
      // the second half-kick as a pass of its own, the reduction.)
      bool need_sync = last;
      if (ens == kNve && !record && !last) {
        kick2_pending = true; // fused into the next step's pass over the atoms
      } else {
        be_.template launch<256>(kSlotVV, N_, ResidentStepBody{box_, b_, dt, 1, 0, 0});
        if (ens == kBer) {
          thermo_now();
          if (1.0 / tcoup > 1.0e-5) { // ensemble_ber.cu:223
            be_.template launch<64>(kSlotMisc, 1, BerendsenFactorBody{b_.flags, target, 1.0 / tcoup, thermo_dev_, factor_dev_});
            be_.template launch<256>(kSlotVV, N_, ResidentScaleBody{b_, factor_dev_, 1.0});
          }
        } else if (ens == kNhc) {
          nhc_half(target);
        } else if (ens == kBdp) {
          thermo_now();
          need_sync = true; // the noise is drawn on the host from the kinetic energy, as in the reference
        } else if (ens == kLan) {
          lan_half(target); // Ensemble_LAN::compute2: the second half-step of the thermostat precedes find_thermo
          if (record)
            thermo_now();
        } else if (record) {
          thermo_now();
        }
      }
      if (record && thermo_rows)
        be_.d2d(thermo_rows + 8 * ((step + 1) / thermo_every - 1), thermo_dev_, 8 * sizeof(double)); // (a frozen step's row is written again by its replay)
      int trip = 0;
      if (need_sync) {
        trip = sync_and_check();
        if (!trip) {
          if (ens == kBdp) {
            double T = 0.0;
            be_.d2h(&T, thermo_dev_, sizeof(double));
            be_.template launch<256>(kSlotVV, N_, ResidentScaleBody{b_, nullptr, bdp_factor(N_, T, target, tcoup)});
          }
        }
      } else if ((step + 1) % (poll_fixed ? poll_fixed : (step - calm_since >= kPollCalmSteps ? kPollEveryCalm : kPollEvery)) == 0) {
        be_.poll_record(ring_next, b_.flags);
        pending.push_back(Snapshot{ring_next});
        ring_next = (ring_next + 1) % 8;
        if ((int)pending.size() > kPollDepth) {
          int snap[8];
          be_.poll_wait(pending.front().ring, snap);
          pending.erase(pending.begin());
          if (snap[kFlagMoved] || snap[kFlagOverflow]) // (a capacity bit: the full look throws)
            trip = sync_and_check();
        }
      }
      if (trip) {
        step = handle_trip(trip, step);
        calm_since = step;
        resume_after_vv1 = true;
        tersoff_deferred_ = false; // (the frozen step's pass had taken the assembly of the step before it; what was enqueued since ran as no-ops)
        continue;
      }
      ++step;
    }
    num_compute = compute0 + nsteps;
    calm_steps_ = nsteps - calm_since;
    if (virial)
      exact_virials(); // per-atom virials leave the engine
    resident_export(pos, vel, pe, force, virial);
    be_.sync();
    if (nrec > 0)
      be_.d2h(thermo_host, thermo_rows, sizeof(double) * 8 * (size_t)nrec);
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
  }

  // internal-order integrator state (allocated at the first fused run)
  void resident_alloc()
  {
    if (!b_.vi) {
      b_.vi = dalloc<double>(3 * cap_);
      b_.mi = dalloc<double>(cap_);
      b_.invp = dalloc<int>(cap_);
    }
    if (unwrapped_ && !ui_alloc_)
      ui_alloc_ = dalloc<double>(3 * cap_);
    b_.ui = unwrapped_ ? ui_alloc_ : nullptr;
  }
  void resident_import(const double* vel, const double* mass, const double* pe, const double* force, const double* virial)
  {
    be_.template launch<256>(kSlotMisc, N_, ImportStateBody{b_, vel, mass, pe, force, virial, unwrapped_});
  }
  void resident_export(double* pos, double* vel, double* pe, double* force, double* virial, int owned_only = 0)
  {
    be_.template launch<256>(kSlotMisc, N_, ExportStateBody{b_, pos, vel, pe, force, virial, unwrapped_, owned_only});
  }

  // The small-box branch has no Verlet lists and no internal order (every call rebuilds all image pairs): its run
  // loop is the plain sequence of the per-call entry points on the caller's arrays.
  void run_md_small_box(
    int ens, const double h9[9], const int pbc[3], int64_t n, const int* type, const double* mass, double dt,
    int64_t nsteps, double t1, double t2, double tcoup, double* pos, double* vel, double* pe, double* force,
    double* virial, int64_t thermo_every, double* thermo_host)
  {
    BoxD box;
    box_from_h9(h9, pbc, box);
    if (ens == kNhc) {
      if (!nhc_dev_)
        nhc_dev_ = dalloc<double>(kNhcStateSize);
      if (nhc_fresh_)
        nhc_init(n, t1, tcoup, dt, nhc_dev_);
      nhc_fresh_ = false;
    }
    int64_t rec = 0;
    for (int64_t step = 0; step < nsteps; ++step) {
      const double target = t1 + (t2 - t1) * ((double)step / (double)nsteps);
      if (ens == kNhc) {
        find_thermo(n, box.volume, mass, pe, vel, virial, thermo_dev_);
        nhc_half_step(n, target, dt, thermo_dev_, nhc_dev_, vel);
      } else if (ens == kLan) { // Ensemble_LAN::compute1, ensemble_lan.cu:206-218
        lan_half_step(n, target, tcoup, mass, vel);
      }
      if (ens == kBao) {
        // Ensemble_BAO::compute1 (ensemble_bao.cu:419-446): B A O A; the noise amplitude keeps the temperature the
        // ensemble was constructed with (its c2 is set in the constructor, :36, and never updated: T1 of the run)
        velocity_verlet(false, n, dt, mass, force, pos, vel, nullptr);
        half_drift(n, dt, pos, vel);
        bao_o_step(n, t1, tcoup, mass, vel);
        half_drift(n, dt, pos, vel);
        apply_pbc(h9, pbc, n, pos); // Force::compute wraps (force.cu:787-795)
      } else {
        velocity_verlet(true, n, dt, mass, force, pos, vel, &box);
      }
      zero_properties(n, pe, force, virial);
      if (model_.temperature_model && ens != kNve) // as in run_md
        set_temperature(t1 + (t2 - t1) * ((double)(step + 2) / (double)nsteps));
      potential_compute(h9, pbc, n, type, pos, pe, force, virial);
      velocity_verlet(false, n, dt, mass, force, pos, vel, nullptr);
      const bool record = thermo_every > 0 && (step + 1) % thermo_every == 0;
      if (ens == kLan) // Ensemble_LAN::compute2 (:241-262): second half-step of the thermostat, then find_thermo
        lan_half_step(n, target, tcoup, mass, vel);
      if ((ens != kNve && ens != kLan && ens != kBao) || record)
        find_thermo(n, box.volume, mass, pe, vel, virial, thermo_dev_);
      if (ens == kBer)
        berendsen(n, target, 1.0 / tcoup, thermo_dev_, vel);
      else if (ens == kNhc)
        nhc_half_step(n, target, dt, thermo_dev_, nhc_dev_, vel);
      else if (ens == kBdp)
        bdp_scale(n, target, tcoup, thermo_dev_, vel);
      if (record) {
        be_.d2h(thermo_host + 8 * rec, thermo_dev_, 8 * sizeof(double));
        ++rec;
      }
    }
    be_.sync();
    int flags[kNumFlags];
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
  }

  void export_lists(int which, int* nn, int* nl, int64_t ld, int* max_out)
  {
    if (!have_list_ && !last_small_)
      throw EngineError{-4, "no force evaluation has been performed yet"};
    if (model_.kind == 1 && which != 2) {
      be_.template launch<64>(kSlotMisc, N_, TersoffExportBody{b_, tb_, nn, nl, ld});
    } else {
      // the per-step radial/angular lists are read off the pair records; in tile mode 2 the force
      // path does not write them, so rerun the (idempotent) radial pass with record writing on
      if (which != 2 && !records_valid_ && have_list_)
        force_kernels(kPhaseRecords);
      ExportListsBody body{b_, which, nn, nl, ld};
      be_.template launch<64>(kSlotMisc, N_, body);
    }
    int flags[kNumFlags];
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
    // the caller computes maxima from nn; engine-side maxima of the Verlet lists:
    *max_out = which == 2 ? flags[kFlagMaxSkin] : -1;
  }

  void export_descriptors(float* q, float* fp)
  {
    if (!have_list_ && !last_small_)
      throw EngineError{-4, "no force evaluation has been performed yet"};
    if (model_.kind != 0)
      throw EngineError{-4, "descriptors exist for NEP models only"};
    if (!last_small_ && last_ang_fused_ && last_ang_window_) {
      NEPMI_SHAPE_DISPATCH(launch_angular_fused_window, 0, (1))
    } else if (!last_small_ && last_ang_fused_) {
      // one kernel from the sums to the partial forces: run it once more with the descriptor and Fp written out
      NEPMI_SHAPE_DISPATCH(launch_angular_fused, 0, (1))
    } else if (q && !last_small_ && fuse_ann_active()) {
      // the fused descriptor + ANN kernel keeps the angular descriptor in registers: write it out now (the compact
      // angular records of the last evaluation are still in place)
      NEPMI_SHAPE_DISPATCH(launch_angular_desc, 0, ())
    }
    // (a zero-padded model: the caller's arrays hold the FILE's components, dmap says where each lives in the padded descriptor)
    if (model_.embedded() && !dmap_dev_)
      dmap_dev_ = upload(model_.dmap);
    ExportDescBody body{b_, model_.embedded() ? model_.file_dim : model_.dim, q, fp, model_.embedded() ? dmap_dev_ : nullptr};
    be_.template launch<64>(kSlotMisc, N_, body);
  }

  // per-step list statistics of the last compute (host reduction over nn arrays)
  void list_stats(int& max_skin, int& max_rad, int& max_ang, double& mean_rad, double& mean_ang)
  {
    if (model_.kind == 0 && have_list_ && !last_small_ && !ccode_valid_) { // (padded rows count more than the atom's neighbours)
      be_.frozen = nullptr;
      NEPMI_SHAPE_DISPATCH(compact_lists_shape, 1, ())
    }
    std::vector<int> nr(N_), na(N_);
    be_.d2h(nr.data(), b_.nn_rad, sizeof(int) * N_);
    be_.d2h(na.data(), b_.nn_angstep, sizeof(int) * N_);
    int flags[kNumFlags];
    be_.d2h(flags, b_.flags, sizeof(flags));
    max_skin = flags[kFlagMaxSkin];
    max_rad = max_ang = 0;
    double sr = 0, sa = 0;
    for (int64_t i = 0; i < N_; ++i) {
      if (nr[i] > max_rad) max_rad = nr[i];
      if (na[i] > max_ang) max_ang = na[i];
      sr += nr[i];
      sa += na[i];
    }
    mean_rad = sr / (double)N_;
    mean_ang = sa / (double)N_;
  }

private:
  template <class T>
  T* dalloc(size_t count)
  {
    void* p = be_.alloc(sizeof(T) * (count ? count : 1));
    allocs_.push_back(p);
    return (T*)p;
  }

  template <class T>
  const T* upload(const std::vector<T>& v)
  {
    T* p = dalloc<T>(v.size());
    if (!v.empty())
      be_.h2d(p, v.data(), sizeof(T) * v.size());
    return p;
  }

  void upload_model()
  {
    const NepModel& m = model_;
    md_.T = m.num_types;
    md_.NR = m.n_max_radial;
    md_.KR = m.basis_size_radial;
    md_.NA = m.n_max_angular;
    md_.KA = m.basis_size_angular;
    md_.has222 = m.has_q_222;
    md_.has1111 = m.has_q_1111;
    md_.numL = m.num_L;
    md_.Lmax = m.L_max;
    md_.extra = (m.has_q_112 ? 1 : 0) | (m.has_q_123 ? 2 : 0) | (m.has_q_233 ? 4 : 0) | (m.has_q_134 ? 8 : 0);
    md_.dim = m.dim;
    md_.nneu = m.num_neurons;
    md_.version = m.version;
    md_.zbl_enabled = m.zbl_enabled;
    md_.zbl_flexible = m.zbl_flexible;
    md_.zbl_rc_inner = (float)m.zbl_rc_inner;
    md_.zbl_rc_outer = (float)m.zbl_rc_outer;
    md_.b1 = m.b1;
    md_.rc_r_max = (float)m.rc_radial_max;
    md_.rc_a_max = (float)m.rc_angular_max;
    md_.rcinv_r = 1.0f / md_.rc_r_max;
    md_.rcinv_a = 1.0f / md_.rc_a_max;
    md_.uniform_rc = 1;
    for (int t = 0; t < m.num_types; ++t)
      if (m.rc_radial_f[t] != m.rc_radial_f[0] || m.rc_angular_f[t] != m.rc_angular_f[0])
        md_.uniform_rc = 0;
    md_.c_rad = upload(m.c_rad);
    md_.c_ang = upload(m.c_ang);
    md_.w0 = upload(m.w0);
    md_.b0 = upload(m.b0);
    md_.w1 = upload(m.w1);
    md_.b1t = upload(m.b1t);
    md_.qscale = upload(m.q_scaler);
    md_.rc_r = upload(m.rc_radial_f);
    md_.rc_a = upload(m.rc_angular_f);
    md_.zbl_para = upload(m.zbl_para_f);
    md_.zbl_rco = m.zbl_typewise ? upload(m.zbl_rc_outer_pair) : nullptr;
    md_.atomic_number = upload(m.atomic_numbers);
    md_.ctab_img[0] = md_.ctab_img[1] = md_.cang_img = nullptr;
    if (m.kind != 0) // (Tersoff: no descriptor tables)
      return;
    // the radial coefficient table in the padded LDS layouts of the many-type window kernels (ctab_stage_padded)
    for (int v = 0; v < 2; ++v) {
      const int raw = (m.n_max_radial + 1) * (m.basis_size_radial + 1), blk = ctab_block(m.n_max_radial, m.basis_size_radial, v != 0);
      const int npair = m.num_types * m.num_types;
      std::vector<float> img(((size_t)npair * blk + 3) / 4 * 4, 0.0f);
      if (m.c_rad.size() < (size_t)npair * raw)
        continue;
      for (int pr = 0; pr < npair; ++pr)
        for (int e = 0; e < raw; ++e)
          img[(size_t)pr * blk + e] = m.c_rad[(size_t)pr * raw + e];
      md_.ctab_img[v] = upload(img);
    }
    {
      md_.cang_img = nullptr; // (cang_stride / cang_floats read the shape fields set above)
      const int per = (md_.NA + 1) * (md_.KA + 1), stride = cang_stride(md_);
      std::vector<float> img((size_t)cang_floats(md_), 0.0f);
      for (int pr = 0; pr < md_.T * md_.T && m.c_ang.size() >= (size_t)md_.T * md_.T * per; ++pr)
        for (int r = 0; r < per; ++r)
          img[(size_t)pr * stride + r] = m.c_ang[(size_t)pr * per + r];
      if (m.c_ang.size() >= (size_t)md_.T * md_.T * per)
        md_.cang_img = upload(img);
    }
  }

  void allocate()
  {
    const NepModel& m = model_;
    const int64_t N = cap_;
    if (N < 1 || N > kMaxAtomsPerEngine)
      throw EngineError{-4, "number of atoms per engine must be in [1, 2^25]"};
    if (m.num_types > 127)
      throw EngineError{-3, "more than 127 types"};
    b_.N = N;
    // Neighbor::initialize, neighbor.cu:824-833
    const double rcs = m.rc_radial_max + kSkin;
    b_.MN_acomp = m.MN_angular;
    // (the padded rows of a wavefront exceed its longest true list by a few when one lane fills its queue early: simulated worst
    // cases 4-5 rows at 55-70 % acceptance, more for long sparse lists -- a quarter of the capacity on top of the fixed pad)
    b_.MN_arows = b_.MN_acomp + kAngRowPad + b_.MN_acomp / 4;
    b_.MN_skin = (int)(m.MN_radial * rcs * rcs * rcs / (m.rc_radial_max * m.rc_radial_max * m.rc_radial_max));
    const double ras = m.rc_angular_max + kSkin;
    // List A (the part of the Verlet list inside rc_a + skin) is this engine's own structure: its capacity
    // must never bind before the reference's MN limits do, so the density-scaled estimate gets 50 % headroom
    // (a thermalised crystal does put 21 atoms inside 5 A of a PbTe atom against a scaled estimate of 19.5).
    b_.MN_ang = (int)(1.5 * m.MN_angular * ras * ras * ras / (m.rc_angular_max * m.rc_angular_max * m.rc_angular_max)) + 4;
    if (b_.MN_ang > b_.MN_skin)
      b_.MN_ang = b_.MN_skin;
    if (b_.MN_ang > 65000)
      throw EngineError{-3, "angular Verlet list capacity above 65000 slots is not supported"};
    b_.rc_skin_sq = (float)(rcs * rcs);
    b_.rc_askin_sq = (float)(ras * ras);
    b_.cid = dalloc<int>(N);
    b_.kcell = dalloc<int>(N);
    b_.perm = dalloc<int>(N);
    b_.posq = dalloc<PosQ>(N);
    b_.x0s = dalloc<double>(3 * N);
    b_.nn_skin = dalloc<int>(N);
    b_.nl_skin = dalloc<int>((size_t)b_.MN_skin * N);
    b_.code_skin = dalloc<unsigned short>((size_t)b_.MN_skin * N);
    b_.nn_ang = dalloc<int>(N);
    b_.nl_ang = dalloc<int>((size_t)b_.MN_ang * N);
    b_.code_ang = dalloc<unsigned short>((size_t)b_.MN_ang * N);
    b_.rev_ang = dalloc<unsigned short>((size_t)b_.MN_ang * N);
    b_.nn_rad = dalloc<int>(N);
    b_.nn_angstep = dalloc<int>(N);
    b_.nn_angtrue = dalloc<int>(N);
    if (m.kind == 0) {
      b_.rstash = dalloc<F4>((size_t)(b_.MN_ang + b_.MN_skin) * N);
      b_.acomp = dalloc<F4>((size_t)b_.MN_arows * N);
      b_.amap = dalloc<unsigned short>((size_t)b_.MN_ang * N);
      b_.f12 = dalloc<F4>((size_t)b_.MN_arows * N);
      b_.q = dalloc<float>((size_t)m.dim * N);
      b_.fp = dalloc<float>((size_t)m.dim * N);
      b_.sbuf = dalloc<float>((size_t)(m.n_max_angular + 1) * kNumHarm * N);
      b_.shi = m.L_max > 4 ? dalloc<float>((size_t)(m.n_max_angular + 1) * kHighSums * N) : nullptr;
      b_.KRP = ((m.basis_size_radial + 1) + 3) / 4 * 4;
      b_.atab = dalloc<float>((size_t)N * m.num_types * b_.KRP);
      const AnnMfmaShape as = ann_mfma_shape(m.num_types, m.dim, m.num_neurons, b_.KRP);
      b_.ann_img = as.ok ? dalloc<float>(as.img_floats * m.num_types) : nullptr;
      be_.ann_prepare(md_, b_);
      b_.pe_i = dalloc<float>(N);
      b_.MN_rad = m.MN_radial;
      b_.ccode = dalloc<unsigned short>((size_t)b_.MN_rad * N);
      b_.nn_t0 = dalloc<int>(N);
      b_.prec = dalloc<WinRec>(N);
      b_.aidx = dalloc<unsigned short>((size_t)b_.MN_acomp * N);
      b_.amask = dalloc<unsigned>(4 * (size_t)N);
      // static window layout (Bufs::wtab comes with the cell arrays): list words of four LDS slots
      b_.MN_wchunks = (b_.MN_ang + 3) / 4 + 2 * ((b_.MN_skin + 3) / 4) + 4;
      b_.wcode = dalloc<unsigned short>((size_t)b_.MN_wchunks * 4 * N);
      b_.wseg = dalloc<int>(N);
      b_.FPR = (m.n_max_radial + 1 + 3) / 4 * 4;
      // models served by a shape WITHOUT type-pure lists (more than two types, the cover shapes, the run-time shape): the radial
      // Fp row per atom, what their force assembly contracts from (ForceWinBody<..., FPJ>, the many-type LDS scatter)
      b_.fpr = (m.num_types > 4 || !served_by_type_pure_shape(m)) ? dalloc<float>((size_t)N * b_.FPR) : nullptr;
      b_.MN_cw = (b_.MN_rad + 3) / 4 + 1;
      b_.cword = dalloc<unsigned short>((size_t)2 * b_.MN_cw * 4 * N);
      // scatter-form force assembly (nep_scatter.h; device backends only)
      b_.aslot = B::kHasScatter ? dalloc<unsigned short>((size_t)b_.MN_arows * N) : nullptr;
      b_.compact_all = b_.aslot ? 1 : 0;
      // mask form of the per-step radial list (Bufs::rmaskA / rmaskB / tmaskA)
      b_.MAW = (4 * ((b_.MN_ang + 3) / 4) + 31) / 32 + 1;
      b_.MBW = (8 * ((b_.MN_skin + 3) / 4) + 31) / 32 + 1;
      b_.rmaskA = B::kHasScatter ? dalloc<unsigned>((size_t)b_.MAW * N) : nullptr;
      b_.rmaskB = B::kHasScatter ? dalloc<unsigned>((size_t)b_.MBW * N) : nullptr;
      b_.tmaskA = B::kHasScatter ? dalloc<unsigned>((size_t)b_.MAW * N) : nullptr;
      b_.MA2 = 2 * ((b_.MN_ang + 3) / 4) + 2;
      const bool two = B::kHasScatter && m.num_types == 2;
      b_.acode2 = two ? dalloc<unsigned short>((size_t)b_.MA2 * N * 4) : nullptr;
      b_.aorig2 = two ? dalloc<unsigned>((size_t)b_.MA2 * N) : nullptr;
      b_.aseg2 = two ? dalloc<int>(N) : nullptr;
      b_.use_rmask = 0;
      b_.use_csync = 0;
    } else { // Tersoff-1989: Tersoff1989::Tersoff1989 allocations (tersoff1989.cu:141-149)
      tb_.rec = dalloc<D4>((size_t)b_.MN_ang * N);
      tb_.bb = dalloc<double>((size_t)b_.MN_ang * N);
      tb_.bp = dalloc<double>((size_t)b_.MN_ang * N);
      tb_.f12 = dalloc<D4>((size_t)b_.MN_ang * N);
      tb_.pe_d = dalloc<double>(N);
      tb_.mask = dalloc<unsigned long long>(N);
      for (int k = 0; k < 3; ++k) {
        const TersoffSet& s = m.ters[k];
        tp_.p[k] = TersoffSetD{s.a, s.b, s.lambda, s.mu, s.beta, s.n, s.c, s.d, s.h, s.r1, s.r2, s.c2, s.d2,
                               s.one_plus_c2overd2, s.pi_factor, s.minus_half_over_n};
      }
      tp_.rc_sq = (float)(m.rc_radial_max * m.rc_radial_max);
    }
    b_.fo = dalloc<double>((size_t)kOutPlanes * N);
    b_.zbl = dalloc<float>(m.zbl_enabled ? (size_t)10 * N : 1);
    b_.lvl = dalloc<signed char>(N);
    b_.angf = dalloc<signed char>(N);
    b_.tperm = dalloc<int>(N);
    b_.tpos = dalloc<int>(N);
    b_.tcount = dalloc<int>(((size_t)(N >> kTypeChunkShift) + 2) * m.num_types + 2);
    b_.flags = dalloc<int>(kNumFlags);
    be_.memset(b_.flags, 0, sizeof(int) * kNumFlags);
    thermo_scratch_ = dalloc<double>(8 * 1024);
    thermo_dev_ = dalloc<double>(8);
    select_shape();
  }

  void check_overflow(const int* flags)
  {
    if (flags[kFlagRange] && !scatter_disabled_) {
      // nep_scatter.h: a pair half beyond the guard band of the fixed-point accumulators (the sums are still exact: the band
      // sits a factor eight below their range).  The gather form has no such limit: it takes over for the rest of the run.
      scatter_disabled_ = true;
      ++num_range_handovers;
      be_.memset(b_.flags + kFlagRange, 0, sizeof(int));
    }
    if (flags[kFlagOverflow] & kOverflowRangeHard)
      throw EngineError{-4, "a force beyond 256 eV/A reached the fixed-point sums of the scatter-form force assembly before the gather "
                            "form had taken over (decomposed run: a flagged step stands): run with nepmi_engine_set_force_form(e, 0)"};
    if (flags[kFlagOverflow] & 8)
      throw EngineError{-4, "non-finite atom coordinates (the simulation has blown up, or the position array is not initialised)"};
    if (flags[kFlagOverflow]) {
      char msg[256];
      std::snprintf(
        msg, sizeof msg,
        "neighbour list capacity exceeded (flags=%d; Verlet max %d of %d, angular Verlet max %d of %d, "
        "angular capacity %d): increase MN in the cutoff line of nep.txt",
        flags[kFlagOverflow], flags[kFlagMaxSkin], b_.MN_skin, flags[kFlagMaxAng], b_.MN_ang, b_.MN_acomp);
      throw EngineError{-6, msg};
    }
  }

  // get_expanded_box (nep.cu:1295-1354): small iff a periodic thickness <= 2.5 (rc + skin)
  bool is_small_box(const BoxD& box) const
  {
    const double lim = 2.5 * (model_.rc_radial_max + kSkin);
    for (int d = 0; d < 3; ++d)
      if (box.pbc[d] && box.thickness[d] <= lim)
        return true;
    return false;
  }

  void small_box_compute(const int* type, const double* pos)
  {
    const double rc = model_.rc_radial_max;
    // the reference refuses a box that is thin in one periodic direction and very thick in another
    // (get_expanded_box, nep.cu:1316-1324): "The box has a thickness < 2.5 radial cutoffs in a periodic direction
    // and is too large in another"
    for (int d = 0; d < 3; ++d)
      if (box_.thickness[d] > 10.0 * rc)
        throw EngineError{-7, "small-box branch: a periodic thickness <= 2.5 (rc+1) while another direction is thicker than "
                              "10 rc (the reference refuses this box too, nep.cu:1316-1324)"};
    if (N_ > 20000)
      throw EngineError{-7, "small-box branch (a periodic thickness <= 2.5 (rc+1)) is limited to 20000 atoms"};
    if (!b_.sh_ang)
      b_.sh_ang = dalloc<int>((size_t)b_.MN_ang * cap_);
    SmallBoxPairsBody sb;
    std::memset(&sb, 0, sizeof(sb));
    sb.box = box_;
    double eh[18];
    for (int d = 0; d < 3; ++d)
      sb.nc[d] = box_.pbc[d] ? (int)std::ceil(2.0 * rc / box_.thickness[d]) : 1;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        eh[3 * r + c] = box_.h[3 * r + c] * sb.nc[c];
    {
      BoxD tmp;
      const int pbc3[3] = {box_.pbc[0], box_.pbc[1], box_.pbc[2]};
      box_from_h9(eh, pbc3, tmp);
      for (int k = 0; k < 18; ++k)
        sb.E[k] = (float)tmp.h[k];
    }
    for (int d = 0; d < 3; ++d)
      if (sb.nc[d] > 100)
        throw EngineError{-3, "box far too thin for the radial cutoff"};
    sb.m = md_;
    sb.b = b_;
    sb.pos = pos;
    sb.type = type;
    be_.memset(b_.flags + kFlagMaxSkin, 0, 2 * sizeof(int));
    be_.begin_region(kRegionForce);
    be_.template launch<64>(kSlotRadial, N_, sb);
    be_.template launch<64>(kSlotMisc, N_, ReverseSlotsSmallBody{b_});
    small_force_kernels();
    be_.end_region(kRegionForce);
    int flags[kNumFlags];
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
  }

  template <class S>
  void small_force_kernels_shape()
  {
    be_.template launch<64>(kSlotRadial, N_, RadialFromRecordsBody<S>{md_, b_});
    launch_angular_desc<S>();
    be_.template launch_ann<S>(kSlotAnn, N_, md_, b_, false); // identity work order, no type groups
    launch_angular_force<S>();
    be_.template launch<64>(kSlotForce, N_, ForceAssembleBody<S>{md_, b_});
  }

  void small_force_kernels()
  {
    NEPMI_SHAPE_DISPATCH(small_force_kernels_shape, 1, ())
  }

  // Neighbor::find_neighbor (neighbor.cu:303-365) + find_cell_list (:164-215)
  void rebuild(const int* type, const double* pos)
  {
    const NepModel& m = model_;
    calm_steps_ = 0;
    const double rc_list = m.rc_radial_max + kSkin;
    double rc_cell = 0.5 * rc_list;
    int nb[3];
    auto bin_grid = [&]() {
      for (int d = 0; d < 3; ++d) {
        if (box_.pbc[d]) {
          if (box_.thickness[d] <= 2.5 * rc_list) { // NEP::compute -> small box (nep.cu:1305-1314)
            char msg[200];
            std::snprintf(
              msg, sizeof msg,
              "box thickness %.3f A in periodic direction %d is <= 2.5*(rc+skin) = %.3f A: small-box path "
              "is not implemented on the device",
              box_.thickness[d], d, 2.5 * rc_list);
            throw EngineError{-7, msg};
          }
          nb[d] = (int)std::floor(box_.thickness[d] / rc_cell);
        } else {
          // The reference uses ONE bin in a non-periodic direction (box.cu:80-91), i.e. an O(N^2)
          // sweep along it; the neighbour SETS do not depend on the binning, so bin it as well
          // (edge bins absorb atoms outside the box).  Needed for the ghost-padded local boxes.
          nb[d] = (int)std::floor(box_.thickness[d] / rc_cell);
          if (nb[d] < 1)
            nb[d] = 1;
        }
      }
    };
    // cells per direction -> Bufs; returns the number of cells padded to whole bricks
    auto set_grid = [&]() -> int64_t {
      b_.nbx = nb[0];
      b_.nby = nb[1];
      b_.nbz = nb[2];
      b_.gbx = (nb[0] + kBrick - 1) / kBrick;
      b_.gby = (nb[1] + kBrick - 1) / kBrick;
      b_.gbz = (nb[2] + kBrick - 1) / kBrick;
      b_.rc_inv_cell = 1.0 / rc_cell;
      return (int64_t)b_.gbx * b_.gby * b_.gbz * (kBrick * kBrick * kBrick);
    };
    bin_grid();
    int64_t ncell = set_grid(); // the finest grid: the largest cell arrays this box needs
    if (ncell > ((int64_t)1 << 30))
      throw EngineError{-3, "too many cells"};
    if (ncell > ncell_cap_) {
      // grow-only (cell arrays are small next to the lists)
      b_.cell_count = dalloc<int>(ncell + 1);
      b_.cell_fill = dalloc<int>(ncell);
      b_.cell_ghost = dalloc<int>(ncell);
      brick_live_buf_ = dalloc<int>(ncell / 64 + 2);
      b_.brick_flag = dalloc<int>(ncell / 64 + 2);
      b_.brick_order = dalloc<int>(ncell / 64 + 1);
      b_.wtab = model_.kind == 0 ? dalloc<int>((size_t)(ncell / 64 + 1) * 1024) : nullptr;
      scan_scratch_ = dalloc<int>(ncell / 1024 + 1024);
      ncell_cap_ = ncell;
    }
    // Cell size.  The reference's edge (rc + skin) / 2 is the smallest that works (neighbour SETS do not depend on
    // it); the window kernels give every 4x4x4-cell brick 256 atom slots per pass, so a brick that holds 170-220
    // atoms leaves a fifth of the lanes idle.  Among the grids with 0..6 fewer cells along the first direction
    // (the others follow the same edge) take the coarsest whose fullest brick still fits one pass with a margin;
    // counted, not timed: the same input always gives the same grid.  Redone when the box or the atom count changes.
    if (model_.kind == 0 && use_tiles_ && tile_mode_ != 0 && nb[0] >= 8 && nb[1] >= 8 && nb[2] >= 8) {
      bool same = grid_n_ == N_;
      for (int k = 0; k < 9; ++k)
        same = same && grid_h_[k] == box_.h[k];
      if (!same) {
        const double edge0 = rc_cell;
        const int nb0 = nb[0];
        // (when no coarser grid fits -- dense long-cutoff models, whose bricks take several passes anyway -- at least the even grid
        // with the same number of cells: cells of exactly (rc + skin) / 2 leave the box's remainder to the last cell of every
        // direction, up to twice as wide, and the window that holds eight of those per direction sets the LDS budget of every brick:
        // C_2024_NEP4 in diamond, 35 cells per direction: 8,000 window slots against 6,100)
        double best = box_.thickness[0] / nb0 * (1.0 - 1.0e-12);
        if (best < edge0)
          best = edge0;
        for (int k = 1; k <= 6 && nb0 - k >= 8; ++k) {
          rc_cell = box_.thickness[0] / (nb0 - k) * (1.0 - 1.0e-12);
          if (rc_cell < edge0)
            continue;
          bin_grid();
          const int64_t nc = set_grid();
          be_.memset(b_.cell_count, 0, sizeof(int) * (nc + 1));
          be_.memset(b_.flags + kFlagMaxBrick, 0, sizeof(int));
          be_.template launch<256>(kSlotMisc, N_, BinAtomsBody{box_, b_, pos});
          be_.template launch<64>(kSlotMisc, nc / 64, BrickMaxBody{b_});
          int fullest = 0;
          be_.sync();
          be_.d2h(&fullest, b_.flags + kFlagMaxBrick, sizeof(int));
          if (fullest <= kBrickFill)
            best = rc_cell;
        }
        grid_edge_ = best;
        grid_n_ = N_;
        for (int k = 0; k < 9; ++k)
          grid_h_[k] = box_.h[k];
      }
      rc_cell = grid_edge_;
      bin_grid();
      ncell = set_grid();
    }
    {
      // fixed-point frame of the window kernels: +-R covers the 8x8x8-cell window seen from its centre (the widened
      // edge cells of an open direction and half a cell of outliers included) plus the drift between two rebuilds
      double R = 0.0, lmax = 0.0;
      for (int c = 0; c < 3; ++c) {
        double r = 0.0;
        for (int d = 0; d < 3; ++d)
          r += 6.5 * std::fabs(box_.h[3 * c + d]) / nb[d];
        R = r > R ? r : R;
      }
      R += 2.0;
      WinGeom& g = b_.wg;
      g.inv_unit = 1073741824.0 / R;
      g.unit = (float)(R / 1073741824.0);
      g.unit2 = g.unit * g.unit;
      for (int d = 0; d < 3; ++d) {
        const double len = std::sqrt(box_.h[d] * box_.h[d] + box_.h[3 + d] * box_.h[3 + d] + box_.h[6 + d] * box_.h[6 + d]);
        if (box_.pbc[d] && len > lmax)
          lmax = len;
        g.cell_frac[d] = 1.0 / (box_.thickness[d] * b_.rc_inv_cell);
        for (int c = 0; c < 3; ++c) {
          g.cv[3 * c + d] = (int)std::llround(box_.h[3 * c + d] * g.cell_frac[d] * g.inv_unit);
          g.sv[3 * c + d] = (int)std::llround(box_.h[3 * c + d] * (1.0 - nb[d] * g.cell_frac[d]) * g.inv_unit);
        }
      }
      // The reference forms r12 in FP32 with an FP32 minimum image: across a periodic face that carries rounding of
      // the order of ulp(box length).  A list decision closer to a cutoff than this band is retaken exactly.
      g.band = (float)(1.0e-4 + 4.0 * model_.rc_radial_max * lmax * 1.2e-7);
    }
    be_.memset(b_.cell_count, 0, sizeof(int) * (ncell + 1));
    be_.memset(b_.cell_fill, 0, sizeof(int) * ncell);
    be_.memset(b_.cell_ghost, 0, sizeof(int) * ncell);
    be_.memset(b_.flags + kFlagMaxSkin, 0, 2 * sizeof(int));
    be_.memset(b_.flags + kFlagMoved, 0, sizeof(int)); // the rebuild this flag asked for is happening
    be_.memset(b_.flags + kFlagOutlier, 0, sizeof(int));
    be_.begin_region(kRegionRebuild);
    be_.template launch<256>(kSlotMisc, N_, BinAtomsBody{box_, b_, pos});
    be_.exclusive_scan(b_.cell_count, ncell + 1, scan_scratch_);
    be_.template launch<256>(kSlotMisc, N_, FillCellsBody{b_});
    be_.template launch<256>(kSlotMisc, ncell, SortCellsBody{b_});
    be_.template launch<256>(kSlotMisc, N_, GatherSortedBody{box_, b_, pos, type});
    if (fuse_ann_active()) {
      be_.template launch<256>(kSlotMisc, N_, IdentityOrderBody{b_}); // no type groups: the ANN runs per atom, in place
    } else {
      const int64_t nkeys = ((N_ >> kTypeChunkShift) + 1) * model_.num_types;
      be_.memset(b_.tcount, 0, sizeof(int) * (nkeys + 1));
      be_.template launch<64>(kSlotMisc, nkeys, TypeCountBody{b_, model_.num_types});
      be_.exclusive_scan(b_.tcount, nkeys + 1, scan_scratch_);
      be_.template launch<256>(kSlotMisc, N_, TypeFillBody{b_, model_.num_types});
    }
    // the bricks' statistics first (they need the cells only): they say whether the Verlet lists can be built from LDS windows
    be_.memset(b_.flags + kFlagMaxWindow, 0, 3 * sizeof(int));
    num_bricks_ = (int64_t)b_.gbx * b_.gby * b_.gbz;
    b_.brick_live = (b_.level && brick_live_buf_) ? brick_live_buf_ : nullptr;
    if (b_.level) {
      be_.memset(brick_live_buf_, 0, sizeof(int) * (size_t)(num_bricks_ + 1));
      be_.template launch<256>(kSlotMisc, N_, MarkGhostCellsBody{b_});
    }
    be_.template launch<64>(kSlotMisc, num_bricks_, TileStatsBody{box_, b_});
    be_.memset(b_.brick_flag + num_bricks_, 0, sizeof(int));
    be_.exclusive_scan(b_.brick_flag, num_bricks_ + 1, scan_scratch_);
    be_.template launch<64>(kSlotMisc, num_bricks_, BrickOrderBody{b_, num_bricks_});
    int flags[kNumFlags];
    bool build_in_windows = false;
    if (NEPMI_BUILD_WIN && use_tiles_ && model_.kind == 0 && b_.prec) {
      be_.d2h(flags, b_.flags, sizeof(flags));
      build_in_windows = flags[kFlagMaxCell] <= 127 && flags[kFlagMaxWindow] <= win_max_atoms() && !flags[kFlagOutlier];
      for (int d = 0; d < 3; ++d)
        if (box_.pbc[d] && nb[d] < 8)
          build_in_windows = false;
      if (build_in_windows) {
        WinLayout lay = win_;
        lay.wmax = (flags[kFlagMaxWindow] + 63) / 64 * 64;
        lay.compact = 0;
        be_.launch_win(kSlotMisc, num_bricks_, BuildListsWinBody{WinStage{box_, b_, lay}});
      }
    }
    if (!build_in_windows)
      be_.template launch<128>(kSlotMisc, N_, BuildListsBody{box_, b_});
    // The reverse slots of list A (where does i sit in j's list: what the GATHER form needs to find the partner's partial force)
    // are built when a kernel that reads them is about to run (ensure_reverse_slots): the run loops' radial pass in its
    // wave-synchronous form does not, so a rebuild inside a scatter-form loop never pays for them (PbTe 0.32 ms, UNEP-v1 1.5 ms,
    // carbon 3.6 ms per million atoms and rebuild).  Tersoff reads them on every step; single-domain engines only.
    rev_valid_ = false;
    if (model_.kind == 1 || b_.level) // (decomposed runs: eagerly -- their boundary bricks' radial pass may run on another stream)
      ensure_reverse_slots();
    be_.end_region(kRegionRebuild);
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
    num_boundary_bricks_ = flags[kFlagNumBoundary];
    max_ang_rebuild_ = flags[kFlagMaxAng];
    // LDS-window radial pass: unique window cells (>= 8 cells per periodic direction), 7-bit rank
    // in cell, window fits the LDS budget
    tile_ok_ = use_tiles_ && model_.kind == 0 && flags[kFlagMaxCell] <= 127 && flags[kFlagMaxWindow] <= win_max_atoms() &&
               !flags[kFlagOutlier];
    for (int d = 0; d < 3; ++d)
      if (box_.pbc[d] && nb[d] < 8)
        tile_ok_ = false;
    if (std::getenv("NEPMI_DEBUG_REBUILD"))
      std::fprintf(stderr, "nepmi rebuild: cells %d %d %d max_cell %d max_window %d (cap %d) max_brick %d outlier %d tiles %d -> tile_ok %d\n", nb[0], nb[1], nb[2],
                   flags[kFlagMaxCell], flags[kFlagMaxWindow], win_max_atoms(), flags[kFlagMaxBrick], flags[kFlagOutlier], (int)use_tiles_, (int)tile_ok_);
    win_.wmax = (flags[kFlagMaxWindow] + 63) / 64 * 64;
    // static window layout of the one-lane window kernels: Verlet entries as LDS slots, four to a word
    win2_ok_ = tile_ok_ && use_win2_ && b_.wtab != nullptr && win_.wmax < 65535;
    if (win2_ok_) {
      b_.wsent = win_.wmax;
      be_.template launch<128>(kSlotMisc, N_, PackCodesBody{b_, shape_ts() == 2 ? 2 : 1});
      be_.d2h(flags, b_.flags, sizeof(flags));
      check_overflow(flags);
    }
    if (win2_ok_ && b_.aslot) {
      // scatter-form force assembly: one halo row per brick (the window sums its workgroup leaves for ForceFoldBody, 16 bytes
      // per window slot) and, per atom, the table of the windows that hold it (nep_scatter.h: FoldMapBody)
      const size_t need = (size_t)num_bricks_ * (size_t)win_.wmax * 4;
      if (need > halo_cap_) {
        dfree(halo_);
        halo_ = nullptr;
        halo_ = dalloc<int>(need + need / 8);
        halo_cap_ = need + need / 8;
      }
      if (!fmap_)
        fmap_ = dalloc<unsigned>((size_t)fold_rows_ * cap_);
      fold_ok_ = num_bricks_ < ((int64_t)1 << 19) && win_.wmax < (1 << 13);
      if (fold_ok_) {
        int most = be_.build_fold_map(N_, box_, b_, win_.wmax, fold_rows_, fmap_);
        if (most > fold_rows_) { // (partly filled bricks next to a periodic face: a cell can lie in up to 27 windows)
          fold_rows_ = most;
          dfree(fmap_);
          fmap_ = nullptr;
          fmap_ = dalloc<unsigned>((size_t)fold_rows_ * cap_);
          most = be_.build_fold_map(N_, box_, b_, win_.wmax, fold_rows_, fmap_);
        }
      }
    }
    if (reverse_ghosts_ && b_.level) {
      // the rows a ghost would have written in forward mode are read by its neighbours' (and its own) force assembly: zero
      // = no contribution; owned atoms rewrite theirs every step
      be_.memset(b_.atab, 0, sizeof(float) * (size_t)cap_ * model_.num_types * b_.KRP);
      be_.memset(b_.f12, 0, sizeof(F4) * (size_t)b_.MN_arows * cap_);
      if (b_.fpr)
        be_.memset(b_.fpr, 0, sizeof(float) * (size_t)cap_ * b_.FPR);
    }
    have_list_ = true;
    ++num_rebuild;
  }

  // Angular force: with >= 7 radial channels the per-atom table G (24 floats per channel) holds the
  // kernel at one wavefront per SIMD; two lanes per atom (channels split between them) bring it to two.
  // With fewer channels the one-lane form already runs two wavefronts and the split only adds work.
  template <class S>
  void launch_angular_desc(bool fuse = false)
  {
    // the descriptor kernel carries only the sums (no P/Q): it drops to one wavefront per SIMD from 9 channels on
    if (S::fixed && S::NA + 1 >= 9)
      be_.template launch_lds_pairs<kAngDescBlock>(kSlotAngular, N_, AngularDescBody<S>{md_, b_, recompute_s(), 0});
    else if (fuse && S::fixed)
      be_.template launch_lds<kAngFusedBlock>(kSlotAngular, N_, AngularDescBody<S>{md_, b_, recompute_s(), 1});
    else
      be_.template launch_lds<kAngDescBlock>(kSlotAngular, N_, AngularDescBody<S>{md_, b_, recompute_s(), 0});
    if (!S::fixed && model_.L_max > 4) // rows l = 5..l_max_3body on top of the 24-sum kernel (nep_highl.h)
      be_.template launch<64>(kSlotAngular, N_, HighLDescBody{md_, b_});
  }

  // Descriptor + ANN in one kernel (AngularDescBody::fuse_ann): the one-lane descriptor form, few types (the
  // weight image of all types sits in LDS), default ANN mode.  Decided per engine, so that the lists' work order
  // (identity instead of type groups) can follow it.
  bool fuse_ann_active() const
  {
    if (model_.kind != 0 || shape_ == 0 || ann_mode_ != 1 || model_.n_max_angular + 1 >= 9 || model_.num_types > 4)
      return false;
    const int dp = (model_.dim + 3) / 4 * 4;
    const size_t floats = (size_t)model_.num_types * (model_.num_neurons * dp + 40 + 2 * model_.num_neurons) +
                          (size_t)model_.num_types * model_.num_types *
                            ((model_.n_max_radial + 1) * (model_.basis_size_radial + 1) +
                             (model_.n_max_angular + 1) * (model_.basis_size_angular + 1) + 1);
    return floats * sizeof(float) <= 60 * 1024;
  }

  // Angular descriptor + ANN + partial angular forces in one lane-pair kernel (nep_fused.h: the sums never leave the
  // registers, no second descriptor evaluation in the force kernel): wherever the descriptor + ANN fusion applies, on a
  // device backend.  A counted rule; set_angular_fused(0) keeps the two kernels.
  bool ang_fused_active() const
  {
    if (!B::kHasFusedAngular || !ang_fused_ || model_.kind != 0 || shape_ == 0 || ann_mode_ != 1 || model_.num_types > 4)
      return false;
    if ((model_.n_max_angular + 2) / 2 > 5) // more than five channels (120 sums) per lane: the register file does not hold them
      return false;
    // the LDS image of nep_fused.h (fused_lds_layout): both weight half-rows of every neuron and type, the two coefficient tables
    const int nrh = (model_.n_max_radial + 2) / 2, nloc = (model_.n_max_angular + 2) / 2;
    const int dph = (nrh + model_.num_L * nloc + 3) / 4 * 4;
    const size_t T = (size_t)model_.num_types;
    const size_t floats = T * ((size_t)model_.num_neurons * 2 * dph + 32 + 2 * model_.num_neurons) + 2 * dph +
                          T * T * ((model_.n_max_radial + 1) * (model_.basis_size_radial + 1) +
                                   (model_.n_max_angular + 1) * (model_.basis_size_angular + 1) + 1) + 8;
    return floats * sizeof(float) <= 64 * 1024;
  }
  // The same kernel for MANY-TYPE models (more than four types: the weight image of all types does not fit the LDS): atoms in the
  // type-sorted work order, a window of four types resident (engine.hip: nepmi_fused_window_kernel).  Where the force assembly
  // contracts from the atoms' radial Fp rows (skip_atab: the fused kernel writes them, not the radial table).
  bool ang_fused_window_active() const
  {
    if (!B::kHasFusedWindow || !B::kHasFusedAngular || !ang_fused_ || model_.kind != 0 || shape_ == 0 || ann_mode_ != 1 || model_.num_types <= 4 || !b_.fpr)
      return false;
    if ((model_.n_max_angular + 2) / 2 > 5)
      return false;
    const int nrh = (model_.n_max_radial + 2) / 2, nloc = (model_.n_max_angular + 2) / 2;
    const int dph = (nrh + model_.num_L * nloc + 3) / 4 * 4;
    const size_t T = (size_t)model_.num_types, tw = 4;
    const size_t floats = tw * ((size_t)model_.num_neurons * 2 * dph + 32 + 2 * model_.num_neurons) + 2 * dph +
                          tw * T * ((model_.n_max_angular + 1) * (model_.basis_size_angular + 1) + 1) + 16;
    return floats * sizeof(float) <= 80 * 1024; // two workgroups per CU
  }
  template <class S>
  void launch_angular_fused_window(int export_qfp = 0)
  {
    if constexpr (B::kHasFusedWindow && B::kHasFusedAngular && S::fixed && S::TS == 0) {
      const size_t need = be_.template fused_image_floats<S>(md_);
      if (!fused_img_ || fused_img_floats_ < need) {
        dfree(fused_img_);
        fused_img_ = dalloc<float>(need);
        fused_img_floats_ = need;
        fused_img_stale_ = true;
      }
      be_.template launch_angular_fused_window<S>(kSlotAngular, N_, md_, b_, export_qfp, fused_img_, fused_img_stale_);
      fused_img_stale_ = false;
    }
  }
  template <class S>
  void launch_angular_fused(int export_qfp = 0)
  {
    if constexpr (B::kHasFusedAngular && S::fixed) {
      const size_t need = be_.template fused_image_floats<S>(md_);
      if (!fused_img_ || fused_img_floats_ < need) { // (once per engine: the shape does not change)
        dfree(fused_img_);
        fused_img_ = dalloc<float>(need);
        fused_img_floats_ = need;
        fused_img_stale_ = true;
      }
      be_.template launch_angular_fused<S>(kSlotAngular, N_, md_, b_, export_qfp, fused_img_, fused_img_stale_);
      fused_img_stale_ = false;
    }
  }

  // One force kernel per brick behind the radial pass (nep_brick.h): the fused angular kernel and the scatter-form force
  // assembly in one launch -- where both apply, on shapes with two register-resident types, in single-domain engines (no ghost
  // levels), with the compact radial list.  Opt-in (set_brick_force(1)): measured SLOWER than the two kernels, see nep_brick.h.
  template <class S>
  bool brick_force_wanted(const WinStage& ws2) const
  {
    if constexpr (!(B::kHasBrickForce && S::fixed && S::TS == 2)) {
      return false;
    } else {
      if (!brick_force_ || !ang_fused_active() || b_.level || b_.use_rmask || b_.use_csync || !scatter_wanted<S>(ws2))
        return false;
      return be_.template brick_lds_bytes<S>(md_, win_.wmax) <= B::kMaxLdsBytes;
    }
  }
  template <class S>
  void launch_brick_force(const WinStage& ws2, const int* frozen)
  {
    if constexpr (B::kHasBrickForce && S::fixed && S::TS == 2) {
      if (fused_img_stale_ || !fused_img_) { // the image is written by the fused angular kernel's launcher: once, before anything
        launch_angular_fused<S>(0);        // (a whole launch of that kernel, once per engine / temperature change)
      }
      be_.template launch_brick_force<S>(kSlotAngular, kSlotForce, num_bricks_, N_, ws2, md_, halo_, fmap_, fold_rows_, step_outputs_,
                                         fused_img_, frozen);
    }
  }

  template <class S>
  void launch_angular_force()
  {
    // The coefficient table c_ang is staged in LDS once per workgroup: T^2 (n_a+1)(k_a+1) floats -- 46 KB for the 16-type
    // UNEP-v1.  With 64-thread workgroups that is three WAVEFRONTS per CU (r3a: 2.06 ms at 1 M atoms); large tables take
    // 512-thread workgroups so that the table is shared by eight wavefronts.
    const bool big_table = (size_t)cang_floats(md_) * sizeof(float) > 16 * 1024;
#ifndef NEPMI_AF_PAIRS_BIG
#define NEPMI_AF_PAIRS_BIG 0 // A/B switch: 1 = many-type models with a large table (UNEP-v1: 5 channels) run the partial forces with lane pairs in 512-thread workgroups
#endif
    if (S::fixed && S::NA + 1 >= 7)
      be_.template launch_lds_pairs<64>(kSlotAngForce, N_, AngularForceBody<S>{md_, b_, recompute_s()});
    else if (NEPMI_AF_PAIRS_BIG && S::fixed && big_table)
      be_.template launch_lds_pairs<512>(kSlotAngForce, N_, AngularForceBody<S>{md_, b_, recompute_s()});
    else if (big_table)
      be_.template launch_lds<512>(kSlotAngForce, N_, AngularForceBody<S>{md_, b_, recompute_s()});
    else
      be_.template launch_lds<64>(kSlotAngForce, N_, AngularForceBody<S>{md_, b_, recompute_s()});
    if (!S::fixed && model_.L_max > 4)
      be_.template launch<64>(kSlotAngForce, N_, HighLForceBody{md_, b_});
  }

  // ---- shape dispatch ----
#if defined(NEPMI_JIT_SHAPE)
  using S_JIT = Shape<NEPMI_JIT_SHAPE>;     // the shape this core was compiled for
#endif
  using S_PbTeA = Shape<6, 6, 6, 6, 5, 2>;   // examples/nep_train/nep.txt
  using S_PbTeB = Shape<4, 8, 4, 8, 5, 2>;   // tests/gpumd/dump_observer/PbTe_species/PbTe.txt
  using S_C2022 = Shape<10, 10, 8, 8, 6, 1>; // potentials/nep/C_2022_NEP4.txt
  using S_UNEP = Shape<4, 8, 4, 8, 6, 0>;    // potentials/nep/Song-2024-UNEP-v1-...txt (16 types)
  using S_BZO = Shape<8, 8, 6, 8, 5, 0>;     // tests_pytest/fixtures/models/nep_BaZrO3.txt (3 types)
  using S_COV1 = Shape<8, 12, 8, 12, 6, 0>;   // cover shapes: models of other shapes are zero-padded into them (cover_shape_for)
  using S_COV2 = Shape<12, 16, 10, 12, 6, 0>;
  using S_COV3 = Shape<8, 12, 8, 12, 6, 2>;   // ... two-type models: the first cover with type-pure list segments (two-type window kernels, 2-type scatter)

  template <class S>
  bool shape_matches() const
  {
    return model_matches_shape<S>(model_);
  }

  void select_shape()
  {
#if defined(NEPMI_JIT_CORE)
    shape_ = shape_matches<S_JIT>() ? 6 : 0;
#else
    if (shape_matches<S_PbTeA>()) shape_ = 1;
    else if (shape_matches<S_PbTeB>()) shape_ = 2;
    else if (shape_matches<S_C2022>()) shape_ = 3;
    else if (shape_matches<S_UNEP>()) shape_ = 4;
    else if (shape_matches<S_BZO>()) shape_ = 5;
    else if (shape_matches<S_COV3>()) shape_ = 9;
    else if (shape_matches<S_COV1>()) shape_ = 7;
    else if (shape_matches<S_COV2>()) shape_ = 8;
    else shape_ = 0;
#endif
    if (force_generic_)
      shape_ = 0;
  }

public:
  void invalidate() { have_list_ = false; }
  // 0: no LDS-window kernels (gather kernels + pair records); anything else (default): the radial pass and the
  // force assembly both work from the LDS position window.  A counted rule, never a timing: the same input
  // always runs the same kernels.
  void set_tile_mode(int mode)
  {
    use_tiles_ = mode != 0;
    tile_mode_ = mode;
    have_list_ = false;
  }
  // Lanes per atom of the window kernels: one when the bricks fill the chip (more than ~1.5 workgroups per CU); below that a
  // window kernel takes as long as ONE workgroup does, and 2 or 4 lanes share each atom's list (counted rule:
  // the same input always runs the same kernels).  set_win_lanes(1 | 2 | 4) pins it, 0 = this rule.
  int win_lanes() const
  {
    if (!B::kSplitLanes)
      return 1;
    if (win_lanes_ > 0)
      return win_lanes_;
    // (profiles/r4al_lanes_sweep.txt: 512 bricks 0.259 ms with two lanes / 0.273 with one; 640 bricks 0.348 / 0.333)
    return num_bricks_ <= 256 ? 4 : (num_bricks_ <= 512 ? 2 : 1);
  }
  void set_win_lanes(int lanes) { win_lanes_ = (lanes == 1 || lanes == 2 || lanes == 4) ? lanes : 0; }
  // capacity of a brick's LDS window in atoms (win_max_atoms() below); 0 = the rule
  void set_win_max_atoms(int v)
  {
    win_max_atoms_ = v > 0 ? (v < kWinMaxAtoms ? v : kWinMaxAtoms) : 0;
    have_list_ = false;
  }
  int tile_mode_in_use() const { return tile_ok_ ? ((win2_ok_ && win_lanes() == 1) ? 3 : 2) : 0; }
  bool tiles_active() const { return tile_ok_; }
  // 0: per-atom ANN kernel; 1 (default): descriptor + ANN fused where the shape allows it, else the matrix-core
  // kernel; 2: the matrix-core kernel wherever it applies (no fusion)
  void set_use_mfma(int mode)
  {
    ann_mode_ = mode < 0 ? 1 : (mode > 2 ? 2 : mode);
    be_.set_mfma(ann_mode_ != 0);
    have_list_ = false; // the work order of q / fp follows the mode
  }
  // The caller runs the skin policy itself (a domain-decomposed host votes on it globally and calls
  // invalidate()): no per-step flag read-back, the force path is enqueued without a host round trip.
  // List-capacity overflow is then reported at the next rebuild or stats() call.
  void set_external_skin(bool on) { external_skin_ = on; }
  // Reverse-mode ghosts of a decomposed run (Bufs::lvl_desc / lvl_force): descriptors, ANN and partial forces for owned atoms
  // only; the force assembly also runs on the ghosts, whose own rows stay zero, and leaves on each the halves its owned
  // neighbours contribute -- DistT sends them to the owners.  NEP models only.
  void set_reverse_ghosts(bool on)
  {
    if (on && model_.kind != 0)
      throw EngineError{-4, "reverse-mode ghosts: NEP models only"};
    reverse_ghosts_ = on;
    b_.lvl_desc = on ? 2 : 1;
    b_.lvl_force = on ? 1 : 2;
    invalidate();
  }
  bool reverse_ghosts() const { return reverse_ghosts_; }
  void reset_thermostat() { nhc_fresh_ = true; }
  // caller-owned [3N] array that every first-half-step drift is also added to (atom.unwrapped_position)
  void set_unwrapped(double* u) { unwrapped_ = u; }
  void check_flags_now()
  {
    int flags[kNumFlags];
    be_.d2h(flags, b_.flags, sizeof(flags));
    check_overflow(flags);
  }
  // angular s sums: -1 auto (recompute in the force kernel when the model has few angular
  // neighbours, MN_angular <= 16), 0 always through sbuf, 1 always recompute
  void set_angular_recompute(int mode) { recompute_mode_ = mode; }
  // NEP::compute(temperature, ...) of a temperature-dependent model (nep.cu:1483-1486: q[dim] = temperature *
  // q_scaler[dim] is one more ANN input).  That input is the same for every atom, so its product with the last column
  // of w0 is a constant per (type, neuron): it is folded into the hidden-layer bias the ANN kernels read,
  //   tanh(sum_d w0[j][d] q[d] + w0[j][dim] qT - b0[j]) = tanh(sum_d w0[j][d] q[d] - (b0[j] - w0[j][dim] qT)),
  // and every kernel runs as for a plain model (the two forms differ by one f32 rounding of the neuron input).
  void set_temperature(double temperature)
  {
    const NepModel& m = model_;
    if (!m.temperature_model || m.kind != 0 || temperature == temperature_) // (the uploaded bias is the one of 0 K)
      return;
    temperature_ = temperature;
    const float qT = (float)temperature * m.q_scaler_temp;
    b0_eff_.resize(m.b0.size());
    for (size_t k = 0; k < m.b0.size(); ++k)
      b0_eff_[k] = m.b0[k] - m.w0_temp[k] * qT;
    be_.h2d(const_cast<float*>(md_.b0), b0_eff_.data(), sizeof(float) * b0_eff_.size());
    be_.ann_prepare(md_, b_); // the matrix-core weight image carries the bias as well
    fused_img_stale_ = true;  // ... and so does the fused angular kernel's LDS image
  }
  double temperature() const { return temperature_; }
  bool temperature_model() const { return model_.temperature_model; }
  int recompute_s() const
  {
    return recompute_mode_ < 0 ? (model_.MN_angular <= 16 ? 1 : 0) : (recompute_mode_ ? 1 : 0);
  }
  // A decomposed run replaces its engine when the local system outgrows it (DistT::decompose): everything that is not
  // derived from the atom count moves over -- option switches, the temperature of a temperature-dependent model, the BDP
  // generator (the noise sequence continues), the thermostat chain flag and the counters.
  void adopt_from(const EngineT& o)
  {
    use_tiles_ = o.use_tiles_;
    tile_mode_ = o.tile_mode_;
    recompute_mode_ = o.recompute_mode_;
    win_lanes_ = o.win_lanes_;
    win_max_atoms_ = o.win_max_atoms_;
    use_win2_ = o.use_win2_;
    external_skin_ = o.external_skin_;
    force_form_ = o.force_form_;
    use_rmask_ = o.use_rmask_;
    use_csync_ = o.use_csync_;
    ang_fused_ = o.ang_fused_;
    brick_force_ = o.brick_force_;
    loop_ctx_ = o.loop_ctx_;
    scatter_disabled_ = o.scatter_disabled_;
    b_.scatter_limit = o.b_.scatter_limit;
    b_.fold_guard = o.b_.fold_guard;
    b_.scatter_hard = o.b_.scatter_hard;
    b_.fold_hard = o.b_.fold_hard;
    hard_factor_ = o.hard_factor_;
    hard_asked_ = o.hard_asked_;
    guard_delay_ = o.guard_delay_;
    guard_delayed_ = o.guard_delayed_;
    if (reverse_ghosts_ != o.reverse_ghosts_)
      set_reverse_ghosts(o.reverse_ghosts_);
    unwrapped_ = o.unwrapped_;
    nhc_fresh_ = o.nhc_fresh_;
    bdp_rng_ = o.bdp_rng_;
    bdp_iset_ = o.bdp_iset_;
    bdp_gset_ = o.bdp_gset_;
    lan_seed_ = o.lan_seed_;
    num_compute = o.num_compute;
    num_rebuild = o.num_rebuild;
    num_discarded = o.num_discarded;
    num_range_handovers = o.num_range_handovers;
    be_.adopt_options(o.be_);
    if (ann_mode_ != o.ann_mode_)
      set_use_mfma(o.ann_mode_);
    if (force_generic_ != o.force_generic_)
      set_force_generic(o.force_generic_);
    if (o.temperature_ != temperature_)
      set_temperature(o.temperature_);
    have_list_ = false;
  }
  void set_force_generic(bool on)
  {
#if defined(NEPMI_JIT_CORE)
    if (on)
      throw EngineError{-4, "nepmi_engine_set_generic: a JIT core carries the kernels of its own shape only (NEPMI_JIT=0 loads the "
                            "model into the library itself)"};
#endif
    force_generic_ = on;
    select_shape();
    have_list_ = false; // the packed list words follow the shape (two type-pure streams for two-type shapes)
  }
  // types with register-resident sums of the selected shape (Shape::TS)
  int shape_ts() const
  {
#if defined(NEPMI_JIT_SHAPE)
    if (shape_ == 6)
      return S_JIT::TS;
#endif
    return (shape_ == 1 || shape_ == 2 || shape_ == 9) ? 2 : (shape_ == 3 ? 1 : 0);
  }
  // 1 (default): the one-lane window kernels run on the static window layout (RadialWin2Body); 0: the scanned layout
  void set_win2(bool on)
  {
    use_win2_ = on;
    have_list_ = false;
  }
  int shape_id() const { return shape_; }

private:
#ifndef NEPMI_RMASK
#define NEPMI_RMASK 1 // A/B switch (profiles/ab_variants.sh): 0 = the compact list on every step
#endif
#ifndef NEPMI_CSYNC
#define NEPMI_CSYNC 1 // A/B switch: 0 = never the wave-synchronous words
#endif
  template <class S>
  void force_kernels_shape(int phase, const int* frozen)
  {
#ifndef NEPMI_AMASK
#define NEPMI_AMASK 1 // A/B switch (profiles/ab_variants.sh): 0 = amap entries instead of the membership mask
#endif
    // membership mask instead of amap: the one-lane window kernels, list A within the mask's 128 bits (counted at the rebuild)
    b_.use_amask = (NEPMI_AMASK && tile_ok_ && win_lanes() == 1 && max_ang_rebuild_ <= 128) ? 1 : 0;
    // test hook (option "scatter_guard_delay"): a narrowed guard band that starts to apply at the n-th force assembly from now
    if (guard_delay_ > 0 && (phase == kPhaseAll || phase == kPhaseBoundary || phase == kPhaseAfterRadial) && --guard_delay_ == 0)
      set_scatter_guard(guard_delayed_, -1.0);
    const WinStage ws{box_, b_, win_};
    if (phase == kPhaseRecords) { // diagnostics: materialise the pair records of the current positions
      ensure_reverse_slots();
      be_.template launch<64>(kSlotMisc, N_, RadialDescBody<S>{box_, md_, b_, 1});
      records_valid_ = true;
      return;
    }
    const int lanes = win_lanes();
    const bool win2 = win2_ok_ && lanes == 1;
    last_rows_form_ = false;
    last_fpj_form_ = false;
    // Mask form of the per-step radial list: the run loops' scatter-form steps of shapes with type-pure streams.  Decided
    // before the radial pass (which writes either the masks or the compact list); a per-call evaluation in the forced scatter
    // form keeps the compact list (its virial-only pass of the gather form follows at once).
    if (phase != kPhaseRadialOnly) {
      WinLayout layq = win_;
      layq.compact = 1;
      const WinStage wsq{box_, b_, layq};
      b_.use_rmask = (NEPMI_RMASK && use_rmask_ && S::TS > 0 && win2 && b_.rmaskB && loop_ctx_ && max_ang_rebuild_ <= 96 &&
                      (S::TS == 1 || b_.acode2) && scatter_wanted<S>(wsq))
                       ? 1
                       : 0;
      last_mask_form_ = b_.use_rmask != 0;
      // Wave-synchronous words (nep_window.h: SyncFifo): under the same conditions, whenever the mask form was not asked for
      b_.use_csync = (NEPMI_CSYNC && use_csync_ && !b_.use_rmask && win2 && b_.cword && loop_ctx_ && b_.MN_cw < 256 &&
                      scatter_wanted<S>(wsq))
                       ? 1
                       : 0;
      last_sync_form_ = b_.use_csync != 0;
    } else {
      b_.use_rmask = 0; // (before the kernel bodies below copy Bufs)
      b_.use_csync = 0;
    }
    WinLayout lay2 = win_;
    lay2.compact = 1;
    const WinStage ws2{box_, b_, lay2};
    if (!(win2 && b_.use_csync && S::fixed)) // (every other form of the radial pass stores the reverse slot of each angular pair)
      ensure_reverse_slots();
    B& rbe = radial_side_ ? *radial_side_ : be_; // (force_kernels_on: the boundary bricks on the communication stream)
    auto radial = [&](int64_t nb, int first) {
      if (win2 && b_.use_csync)
        rbe.launch_win2(kSlotRadial, nb, RadialWin2Body<S, 1>{ws2, md_, first, frozen});
      else if (win2)
        rbe.launch_win2(kSlotRadial, nb, RadialWin2Body<S>{ws2, md_, first, frozen});
      else if (lanes == 4)
        rbe.launch_win_split(kSlotRadial, nb, RadialWinSplitBody<S, 4>{ws, md_, first, frozen});
      else if (lanes == 2)
        rbe.launch_win_split(kSlotRadial, nb, RadialWinSplitBody<S, 2>{ws, md_, first, frozen});
      else
        rbe.launch_win(kSlotRadial, nb, RadialWinBody<S>{ws, md_, first, frozen});
    };
    if (phase == kPhaseRadialOnly) { // exact_virials: the compact list of the current positions (the masks were written instead)
      b_.use_rmask = 0;
      b_.use_csync = 0;
      if (tile_ok_)
        radial(num_bricks_, -1);
      ccode_valid_ = true;
      return;
    }
    if (phase == kPhaseInterior) { // radial pass of the bricks whose window holds no ghost
      radial(num_bricks_ - num_boundary_bricks_, 0);
      return;
    }
    if (phase == kPhaseBoundaryRadial) {
      radial(num_boundary_bricks_, (int)(num_bricks_ - num_boundary_bricks_));
      return;
    }
    records_valid_ = !tile_ok_;
    be_.begin_region(kRegionForce);
    if (phase == kPhaseBoundary)
      radial(num_boundary_bricks_, (int)(num_bricks_ - num_boundary_bricks_));
    else if (phase == kPhaseAfterRadial)
      ; // (both parts of the radial pass are enqueued already)
    else if (tile_ok_)
      radial(num_bricks_, -1);
    else
      be_.template launch<64>(kSlotRadial, N_, RadialDescBody<S>{box_, md_, b_, 1});
    ccode_valid_ = b_.use_rmask == 0 && b_.use_csync == 0;
    b_.skip_atab = (win2 && fpj_wanted<S>(ws2)) ? 1 : 0; // the FPJ force assembly needs no radial table from the ANN kernel
    last_ang_fused_ = false;
    last_brick_ = false;
    brick_pending_ = false;
    if (win2 && brick_force_wanted<S>(ws2)) {
      launch_brick_force<S>(ws2, frozen);
      last_ang_fused_ = last_brick_ = true;
      brick_pending_ = true;
      last_scatter_form_ = true;
      outputs_stale_ = !step_outputs_;
      virial_local_ = true; // the virial planes hold the own-half form: exact_virials() before they leave the engine
      if (force_form_ == 1 && !loop_ctx_) { // a per-call evaluation in the forced scatter form returns per-atom virials
        materialise_for_gather<S>();
        gather_assembly<S>(ws, ws2, lanes, win2, frozen, 1);
      }
      be_.end_region(kRegionForce);
      return;
    }
    angular_kernels<S>();
    last_scatter_form_ = false;
    outputs_stale_ = false;
    if (win2 && scatter_form<S>(ws2, frozen)) {
      virial_local_ = true; // the virial planes hold the own-half form: exact_virials() before they leave the engine
      if (force_form_ == 1 && !loop_ctx_) // a per-call evaluation in the forced scatter form returns per-atom virials
        gather_assembly<S>(ws, ws2, lanes, win2, frozen, 1);
    } else {
      virial_local_ = false;
      gather_assembly<S>(ws, ws2, lanes, win2, frozen, 0);
    }
    be_.end_region(kRegionForce);
  }

  void ensure_reverse_slots()
  {
    if (rev_valid_)
      return;
    const int* saved = be_.frozen; // (never skipped: a later gather-form step would read what was not written)
    be_.frozen = nullptr;
    be_.template launch<128>(kSlotMisc, N_, ReverseSlotsBody{box_, b_});
    be_.frozen = saved;
    rev_valid_ = true;
  }
  // angular descriptor, ANN and partial angular forces of the current angular records: one kernel or three
  template <class S>
  void angular_kernels()
  {
    last_ang_window_ = false;
    last_ang_fused_ = false;
    if (ang_fused_active()) {
      launch_angular_fused<S>();
      last_ang_fused_ = true;
    } else if (ang_fused_window_active() && b_.skip_atab) {
      launch_angular_fused_window<S>();
      last_ang_fused_ = last_ang_window_ = true;
    } else if (fuse_ann_active()) {
      launch_angular_desc<S>(true);
    } else {
      launch_angular_desc<S>();
      be_.template launch_ann<S>(kSlotAnn, N_, md_, b_, true);
    }
    if (!last_ang_fused_)
      launch_angular_force<S>();
  }

  // After a per-brick force kernel the partial forces and the radial table exist in registers only: the gather form's
  // virial-only pass (exact_virials) reads them from the arrays the separate angular kernel writes
  template <class S>
  void materialise_for_gather()
  {
    if (!brick_pending_)
      return;
    launch_angular_fused<S>(0);
    brick_pending_ = false;
  }

  // The gather form of the force assembly (ForceWinBody / ForceAssembleBody); wonly: only the nine virial planes are written
  template <class S>
  void gather_assembly(const WinStage& ws, const WinStage& ws2, int lanes, bool win2, const int* frozen, int wonly)
  {
    if (wonly)
      virial_local_ = false;
    if (!tile_ok_)
      be_.template launch<64>(kSlotForce, N_, ForceAssembleBody<S>{md_, b_});
    else if (wonly && win2 && fpj_wanted<S>(ws2))
      be_.launch_win2(kSlotMisc, num_bricks_, ForceWinBody<S, 1, false, false, true>{ws2, md_, frozen, 1});
    else if (wonly && win2)
      be_.launch_win2(kSlotMisc, num_bricks_, ForceWinBody<S, 1, NEPMI_CW != 0>{ws2, md_, frozen, 1});
    else if (win2 && rows_form<S>(ws2, frozen))
      ; // (launched by rows_form)
    else if (win2 && fpj_form<S>(ws2, frozen))
      ; // (launched by fpj_form)
    else if (win2)
      be_.launch_win2(kSlotForce, num_bricks_, ForceWinBody<S, 1, NEPMI_CW != 0>{ws2, md_, frozen});
    else if (lanes == 4)
      be_.launch_win_split(kSlotForce, num_bricks_, ForceWinBody<S, 4>{ws, md_, frozen});
    else if (lanes == 2)
      be_.launch_win_split(kSlotForce, num_bricks_, ForceWinBody<S, 2>{ws, md_, frozen});
    else
      be_.launch_win(kSlotForce, num_bricks_, ForceWinBody<S>{ws, md_, frozen});
  }

  // Force assembly as an LDS-local scatter of the own pair halves (nep_scatter.h): the static window layout with one lane per
  // atom, shapes with register-resident per-type rows, a device backend; in the run loops (or wherever set_force_form(1)
  // asks for it) and until a pair half has left the fixed-point guard band (flags[kFlagRange]).  A counted rule.
  template <class S>
  bool scatter_wanted(const WinStage& ws2) const
  {
    if (!B::kHasScatter || !b_.aslot || !halo_ || !fmap_ || !fold_ok_ || scatter_disabled_ || force_form_ == 0)
      return false;
    if (force_form_ < 0 && !loop_ctx_)
      return false;
    // Fewer bricks than workgroup places on the chip (three per CU): both forms then take one brick's latency, and the scatter
    // form has a second launch (the fold) on top -- 128,000 PbTe atoms (512 bricks): 4.7e8 atom-steps/s against 4.8e8 for the
    // gather form, 250,000 atoms (1,000 bricks): 6.07e8 against 5.9e8 (profiles/r4p_size_sweep.txt).  A counted rule.
    // (With reverse-mode ghosts the scatter form is kept at every size: there the ghosts have no own pair halves and skip the
    // assembly altogether, where the gather form runs it on them in full -- 8 ranks sharing a GPU, 1 M atoms: 3.30 against
    // 3.53 ms per step, profiles/r4l_strong8_g1_ov0.json / r4z_inproc_strong8_g1.json.)
    if (force_form_ < 0 && num_bricks_ < kScatterMinBricks && !reverse_ghosts_)
      return false;
    if (S::TS > 0)
      return 24 * (size_t)(win_.wmax + 4) <= B::kMaxLdsBytes;
    // many types / run-time shapes: the own half from the atom's radial Fp row and the coefficient table in LDS -- where the
    // FPJ gather form applies (it supplies the virial-only pass; neither needs the per-atom radial table from the ANN kernel)
    return fpj_wanted<S>(ws2) &&
           25 * (size_t)win_.wmax + 4 * (size_t)model_.num_types * model_.num_types * ctab_block(md_.NR, md_.KR, true) <= B::kMaxLdsBytes;
  }
  // Window capacity in atoms.  Shapes with type-pure list streams (one or two types: their window kernels keep nothing but the
  // window in LDS) take windows up to kWinMaxAtoms = 6,656 slots -- C_2024_NEP4 in diamond, 6,100-6,400 slots, 512,000 atoms on
  // its JIT core: 6.57 -> 5.00 ms/step (r6c / r6za); the many-type and run-time-shape kernels (coefficient tables in LDS next to the
  // window, one workgroup per CU at that size) lose against their gather kernels there (same model zero-padded into the
  // any-types cover: 13.8 -> 18.7 ms/step, run-time shape 60 -> 102; with 1,024-thread workgroups as well, r6f: 15.5 and 68 -- the radial
  // pass then wins, 6.9 against 7.5 ms, the window form of their force assembly loses more) and keep the former 5,000.  A counted rule; option
  // "win_max_atoms" / NEPMI_WIN_MAX_ATOMS pin it (A/B switch, tests).
  static constexpr int kWinMaxAtomsManyType = 5000;
  int win_max_atoms() const
  {
    static const int env = std::getenv("NEPMI_WIN_MAX_ATOMS") ? std::atoi(std::getenv("NEPMI_WIN_MAX_ATOMS")) : 0;
    if (win_max_atoms_ > 0)
      return win_max_atoms_;
    if (env > 0)
      return env;
    return shape_ts() > 0 ? kWinMaxAtoms : kWinMaxAtomsManyType;
  }
  // how this step's radial pass left the pairs inside the cutoff (nep_scatter.h: MODE)
  int list_mode() const { return b_.use_csync ? 2 : (b_.use_rmask ? 1 : 0); }
  template <class S>
  bool scatter_form(const WinStage& ws2, const int* frozen)
  {
    if (!scatter_wanted<S>(ws2))
      return false;
    if (assembly_part_ == 1 && b_.level && num_boundary_bricks_ > 0 && num_boundary_bricks_ < num_bricks_) {
      // a decomposed run with reverse-mode ghosts: the bricks whose window holds a ghost first, then the ghosts' fold -- their
      // partial forces can travel while force_assembly_rest() runs the interior bricks and the owned atoms' fold
      be_.template launch_force_scatter<S>(kSlotForce, num_boundary_bricks_, (int)(num_bricks_ - num_boundary_bricks_), N_, ws2, md_,
                                           halo_, fmap_, fold_rows_, step_outputs_, list_mode(), 0, 1, frozen);
      assembly_pending_ = true;
      pending_outputs_ = step_outputs_;
    } else {
      be_.template launch_force_scatter<S>(kSlotForce, num_bricks_, -1, N_, ws2, md_, halo_, fmap_, fold_rows_, step_outputs_,
                                           list_mode(), 0, 2, frozen);
    }
    if (!step_outputs_)
      outputs_stale_ = true;
    last_scatter_form_ = true;
    return true;
  }
  template <class S>
  void assembly_rest_shape(const int* frozen)
  {
    WinLayout lay2 = win_;
    lay2.compact = 1;
    const WinStage ws2{box_, b_, lay2};
    be_.template launch_force_scatter<S>(kSlotForce, num_bricks_ - num_boundary_bricks_, 0, N_, ws2, md_, halo_, fmap_, fold_rows_,
                                         pending_outputs_, list_mode(), 2, 2, frozen);
  }

public:
  // -1 (default): the run loops take the scatter form where it applies, the per-call entry points the gather form (their
  // per-atom virials are the reference's); 0: gather everywhere; 1: scatter everywhere it applies (per-call evaluations
  // then add one virial-only pass of the gather form)
  void set_force_form(int mode) { force_form_ = mode < 0 ? -1 : (mode > 1 ? 1 : mode); }
  // Decomposed runs with reverse-mode ghosts: 1 = the NEXT force evaluation stops after the boundary bricks' force assembly and
  // the ghosts' fold (scatter form; else it runs whole as always); assembly_pending() then says that force_assembly_rest() has
  // the interior bricks and the owned atoms' fold still to do.  Asked for evaluation by evaluation, like set_step_outputs.
  void set_assembly_part(int part) { assembly_part_ = part; }
  bool assembly_pending() const { return assembly_pending_; }
  void force_assembly_rest(const int* frozen)
  {
    if (!assembly_pending_)
      return;
    assembly_pending_ = false;
    be_.frozen = frozen;
    NEPMI_SHAPE_DISPATCH(assembly_rest_shape, 1, (frozen))
    be_.frozen = nullptr;
  }
  // 1: scatter-form steps of the run loops keep the per-step radial list as inside bits over the packed Verlet words
  // (Bufs::rmaskB) instead of compacting it; 0 (default): the compact list on every step.  Identical results, bit for bit.
  void set_radial_mask(bool on) { use_rmask_ = on; }
  // 1 (default): scatter-form steps of the run loops keep the per-step radial list as wave-synchronous words (SyncFifo); 0: slot-major compact list
  void set_radial_sync(bool on) { use_csync_ = on; }
  // Guard band of the scatter form (nepmi_engine_set_scatter_guard): a pair half beyond `limit` eV/A (default 64; a net force
  // component beyond twice that) hands the force assembly over to the gather form.  Tests lower it so that ordinary forces trip it.
  // limit < 0 / hard_factor < 0: that one stays as it is.  The hard factor is kept as ASKED and clamped against the band in
  // force (never beyond 256 eV/A), so the two can be set in either order.
  void set_scatter_guard(double limit, double hard_factor = 4.0)
  {
    if (limit < 0.0)
      limit = b_.scatter_limit;
    if (hard_factor >= 0.0)
      hard_asked_ = hard_factor;
    if (!(limit > 0.0) || limit > 64.0)
      limit = 64.0;
    hard_factor = hard_asked_;
    if (!(hard_factor >= 1.0) || hard_factor * limit > 256.0)
      hard_factor = 256.0 / limit < 4.0 ? 256.0 / limit : 4.0;
    b_.scatter_limit = (float)limit;
    b_.fold_guard = (int)(2.0 * limit * 4194304.0);
    hard_factor_ = hard_factor;
    if (b_.scatter_hard > 0.0f)
      set_flagged_steps_stand(true);
  }
  double scatter_guard() const { return b_.scatter_limit; }
  // test hook: `limit` becomes the guard band at the n-th force assembly from now (n <= 0: at once) -- to trip the band at a
  // chosen step of a run loop (tests/test_dist_inproc.py: on a step the host looks at)
  void set_scatter_guard_delayed(double limit, int n)
  {
    if (n <= 0) {
      set_scatter_guard(limit, -1.0);
      return;
    }
    guard_delayed_ = limit;
    guard_delay_ = n;
  }
  void set_guard_delay(int n) { guard_delay_next_ = n > 0 ? n : 0; }
  int take_guard_delay()
  {
    const int n = guard_delay_next_;
    guard_delay_next_ = 0;
    return n;
  }
  // Decomposed runs: a flagged step stands (every rank leaves the scatter form together, a few steps later, when the flag has
  // travelled with the skin vote), so a value beyond four times the guard band met meanwhile is an error instead of a wrap
  void set_flagged_steps_stand(bool on)
  {
    b_.scatter_hard = on ? (float)(hard_factor_ * b_.scatter_limit) : 0.0f;
    // (the net-force guard is twice the band: the hard limit on the net never sits below the hand-over it backs up)
    b_.fold_hard = on ? (int)((hard_factor_ > 2.0 ? hard_factor_ : 2.0) * b_.scatter_limit * 4194304.0) : 0;
  }
  bool scatter_enabled() const { return !scatter_disabled_; }
  // a flagged evaluation that is being repeated in the gather form does not stand: its hard-limit bit goes with it
  void clear_range_hard()
  {
    int ov = 0;
    be_.d2h(&ov, b_.flags + kFlagOverflow, sizeof(int));
    if (ov & kOverflowRangeHard) {
      ov &= ~kOverflowRangeHard;
      be_.h2d(b_.flags + kFlagOverflow, &ov, sizeof(int));
    }
  }
  void disable_scatter()
  {
    if (!scatter_disabled_)
      ++num_range_handovers;
    scatter_disabled_ = true;
    be_.memset(b_.flags + kFlagRange, 0, sizeof(int));
  }
  bool last_scatter_form() const { return last_scatter_form_; }
  // 1 (default): angular descriptor, ANN and partial angular forces in one kernel where ang_fused_active() allows; 0: separately
  void set_angular_fused(bool on) { ang_fused_ = on; }
  // 1: ... and the scatter-form force assembly in the same kernel, one workgroup per brick (nep_brick.h); 0 (default)
  void set_brick_force(bool on) { brick_force_ = on; }
  static constexpr bool has_brick_force() { return B::kHasBrickForce; }
  // the callers whose steps need forces, energies and the TOTAL virial only (run loops; the decomposed driver)
  void set_loop_context(bool on) { loop_ctx_ = on; }
  // Run loops: does the NEXT force evaluation have to leave per-atom energies and virials (a thermo record, a thermostat that
  // reads the sums, the last step)?  The scatter form skips their arithmetic and their 80 bytes of stores per atom otherwise.
  // Reset to true after every evaluation: a caller has to ask for the saving step by step.
  void set_step_outputs(bool on) { step_outputs_ = on; }
  bool loop_context() const { return loop_ctx_; }
  // Per-atom virials in the reference's attribution (W_i = sum_j r_ij (x) f21): after a scatter-form step the virial planes
  // hold the own-half form, whose sum is the same; one virial-only pass of the gather form on the step's data replaces them.
  void exact_virials()
  {
    if (!virial_local_ || model_.kind != 0 || last_small_)
      return;
    be_.frozen = nullptr;
    NEPMI_SHAPE_DISPATCH(exact_virials_shape, 1, ())
  }
  bool virial_local() const { return virial_local_; }

private:
  // The last radial pass wrote the masks / the wave-synchronous words: the compact list the gather form (and the list statistics)
  // read, on the same positions -- and, where the angular records were padded rows, the partial forces once more at the compact slots
  template <class S>
  void compact_lists_shape()
  {
    const bool padded_rows = b_.use_csync != 0;
    force_kernels_shape<S>(kPhaseRadialOnly, nullptr);
    if (padded_rows)
      angular_kernels<S>();
  }
  template <class S>
  void exact_virials_shape()
  {
    const WinStage ws{box_, b_, win_};
    WinLayout lay2 = win_;
    lay2.compact = 1;
    const WinStage ws2{box_, b_, lay2};
    const int lanes = win_lanes();
    if (!ccode_valid_)
      compact_lists_shape<S>();
    materialise_for_gather<S>(); // (after a per-brick force kernel: the partial forces and the radial table, once more, to HBM)
    gather_assembly<S>(ws, ws2, lanes, win2_ok_ && lanes == 1, nullptr, 1);
  }

  // Force assembly with the neighbours' table rows in LDS (ForceWinBody<..., ROWS>): shapes with register-resident sums whose
  // records + rows fit the LDS of a CU; a counted rule.  The emulator's host loop runs the same body with one lane per atom.
#ifndef NEPMI_FW_ROWS
#define NEPMI_FW_ROWS 0 // A/B switch (profiles/ab_variants.sh); r3d: carbon 0.79 (gather) vs 1.10 (rows, 4 lanes) vs 0.95 ms (2 lanes): off.  PbTe 1 M: records + rows
                        // (64 B per window atom) exceed the 160 KB of a CU, the rule below keeps the gathered form
#endif
  template <class S>
  bool rows_form(const WinStage& ws2, const int* frozen)
  {
    if (!(S::TS > 0) || !NEPMI_FW_ROWS || NEPMI_CW || !use_rows_)
      return false;
#ifndef NEPMI_FW_ROWS_LANES
#define NEPMI_FW_ROWS_LANES 4
#endif
    constexpr int LR = B::kSplitLanes ? NEPMI_FW_ROWS_LANES : 1;
    const ForceWinBody<S, LR, false, true> body{ws2, md_, frozen};
    if ((size_t)body.lds_bytes() > B::kMaxLdsBytes)
      return false;
    be_.launch_win2_split(kSlotForce, num_bricks_, body);
    last_rows_form_ = true;
    return true;
  }

  // Force assembly of many-type models with the neighbour's half contracted from its radial Fp row (ForceWinBody<..., FPJ>):
  // one-wide shapes, more than four types (the per-atom ANN kernel writes Bufs::fpr), window + coefficient table within 80 KB
  // (two workgroups per CU).  A counted rule.
#ifndef NEPMI_FW_FPJ
#define NEPMI_FW_FPJ 1 // A/B switch (profiles/ab_variants.sh)
#endif
  template <class S>
  bool fpj_wanted(const WinStage& ws2) const
  {
    if (S::TS > 0 || !NEPMI_FW_FPJ || NEPMI_CW || !b_.fpr || (fuse_ann_active() && !ang_fused_active()))
      return false; // (fpr is written by the per-atom ANN kernel, the matrix-core kernel and the fused angular kernel)
    const ForceWinBody<S, 1, false, false, true> body{ws2, md_, nullptr};
    return body.lds_bytes() <= 80 * 1024;
  }
  template <class S>
  bool fpj_form(const WinStage& ws2, const int* frozen)
  {
    if (!fpj_wanted<S>(ws2))
      return false;
    const ForceWinBody<S, 1, false, false, true> body{ws2, md_, frozen};
    be_.launch_win2(kSlotForce, num_bricks_, body);
    last_fpj_form_ = true;
    return true;
  }

public:
  void set_rows(bool on) { use_rows_ = on; }
  void set_stepwise_loops(bool on) { stepwise_loops_ = on; }
  // the kernel forms of the last force evaluation (the counted rules above, in words)
  std::string describe() const
  {
    static const char* shapes[] = {"generic", "PbTe-A(6,6,6,6,5;2)", "PbTe-B(4,8,4,8,5;2)", "C-2022(10,10,8,8,6;1)", "UNEP(4,8,4,8,6;T)",
                                   "BaZrO3(8,8,6,8,5;T)"};
    std::string s;
    if (model_.kind == 1)
      return "potential=tersoff1989 kernels=bond_order+force(fp64)";
    if (shape_ == 6) {
      char jb[96];
      std::snprintf(jb, sizeof jb, "shape=jit(%d,%d,%d,%d,%d;%s)", model_.n_max_radial, model_.basis_size_radial, model_.n_max_angular,
                    model_.basis_size_angular, model_.num_L, model_.num_types <= 2 ? std::to_string(model_.num_types).c_str() : "T");
      s += jb;
    } else {
      if (shape_ == 7 || shape_ == 8 || shape_ == 9) {
        s += shape_ == 7 ? "shape=cover(8,12,8,12,6;T)" : (shape_ == 8 ? "shape=cover(12,16,10,12,6;T)" : "shape=cover(8,12,8,12,6;2)");
        if (model_.embedded()) {
          char eb[128];
          std::snprintf(eb, sizeof eb, "<-zero_padded_model(%d,%d,%d,%d,%d)", model_.file_n_max_radial, model_.file_basis_size_radial,
                        model_.file_n_max_angular, model_.file_basis_size_angular, model_.file_num_L);
          s += eb;
        }
      } else {
        s += std::string("shape=") + shapes[shape_ >= 0 && shape_ <= 5 ? shape_ : 0];
      }
    }
    if (last_small_)
      return s + " path=small_box(all image pairs)";
    const int lanes = win_lanes();
    const bool win2 = win2_ok_ && lanes == 1;
    s += tile_ok_ ? (win2 ? " window=lds_static" : " window=lds_scanned") : " window=none(gather kernels)";
    s += " lanes_per_atom=" + std::to_string(tile_ok_ ? lanes : 1);
    if (last_brick_)
      s += " force=one_kernel_per_brick(descriptor+ann+partial_forces+lds_scatter_of_own_halves,lane_pairs)";
    else if (last_ang_fused_)
      s += last_ang_window_ ? " angular=descriptor+ann+partial_forces_in_one_kernel(lane_pairs,sums_in_registers,type_sorted,window_of_4_types)"
                            : " angular=descriptor+ann+partial_forces_in_one_kernel(lane_pairs,sums_in_registers)";
    else if (fuse_ann_active())
      s += " ann=fused_with_angular_descriptor(packed_fp32,no_mfma)";
    else if (ann_mode_ != 0 && b_.ann_img && (model_.num_types <= 4 || b_.skip_atab))
      s += model_.num_types > 4 ? " ann=mfma_f32_32x32x2(one_type_per_workgroup)" : " ann=mfma_f32_32x32x2";
    else
      s += " ann=per_atom";
    if (!last_ang_fused_) {
      s += (shape_ != 0 && model_.n_max_angular + 1 >= 7) ? " angular_force=lane_pairs" : " angular_force=one_lane";
      s += recompute_s() ? " angular_sums=recomputed" : " angular_sums=stored";
    }
    s += (last_scatter_form_ && last_mask_form_)   ? " radial_list=inside_bits_over_the_verlet_words"
         : (last_scatter_form_ && last_sync_form_) ? " radial_list=wave_synchronous_words"
                                                   : " radial_list=compacted";
    s += last_scatter_form_ ? " force_assembly=lds_scatter_of_own_halves(fixed_point)+fold" :
         last_rows_form_ ? " force_assembly=table_rows_in_lds"
                         : (last_fpj_form_ ? " force_assembly=neighbour_half_from_fp_rows" : " force_assembly=table_rows_gathered");
    if (tile_ok_)
      s += " bricks=" + std::to_string(num_bricks_) + " window_slots=" + std::to_string(win_.wmax);
    return s;
  }
  // frozen != nullptr: a speculatively enqueued step of a fused run loop -- every kernel of the force path looks at
  // that device word first and returns when a list rebuild is pending
  void force_kernels(int phase, const int* frozen = nullptr)
  {
    be_.frozen = frozen;
    force_kernels_dispatch(phase, frozen);
    be_.frozen = nullptr;
    step_outputs_ = true; // (set_step_outputs: asked for evaluation by evaluation)
    assembly_part_ = 0;
  }
  // kPhaseBoundaryRadial on another stream of the same device (a backend from B::make_side_stream)
  void force_kernels_on(B& side, int phase, const int* frozen)
  {
    if (phase != kPhaseBoundaryRadial || !tile_ok_)
      throw EngineError{-4, "force_kernels_on: the boundary bricks' radial pass of the window kernels only"};
    radial_side_ = &side;
    side.frozen = frozen;
    force_kernels_dispatch(phase, frozen);
    side.frozen = nullptr;
    radial_side_ = nullptr;
  }

private:
  void force_kernels_dispatch(int phase, const int* frozen)
  {
    if (model_.kind == 1) { // Tersoff1989::compute, tersoff1989.cu:508-586
      be_.begin_region(kRegionForce);
      // members of the local list in LDS; kTersoffLanes lanes per atom while the system leaves CUs short of wavefronts (a
      // counted rule: 13,824 atoms 36 -> 16 us, 884,736 atoms 0.55 -> 0.63 ms with four lanes)
      if (N_ < 200000)
        be_.template launch_lds_parts<kTersoffBlock, kTersoffLanes>(kSlotRadial, N_, TersoffPartialBody{box_, tp_, b_, tb_, kTersoffLanes});
      else
        be_.template launch_lds<kTersoffBlock>(kSlotRadial, N_, TersoffPartialBody{box_, tp_, b_, tb_});
      // (a run loop's forces-only NVE step: the assembly rides in the next pass over the atoms, TersoffSeamBody)
      if (tersoff_defer_)
        tersoff_deferred_ = true;
      else
        be_.template launch<64>(kSlotForce, N_, TersoffAssembleBody{b_, tb_});
      be_.end_region(kRegionForce);
      return;
    }
    NEPMI_SHAPE_DISPATCH(force_kernels_shape, 1, (phase, frozen))
  }

  NepModel model_;
  B be_;
  int64_t cap_; // allocation size (atoms)
  int64_t N_;   // atoms of the current (local) system, <= cap_; stride of every [slot][atom] array
  ModelD md_{};
  Bufs b_;
  BoxD box_;
  WinLayout win_{0, 0};
  bool win2_ok_ = false, use_win2_ = NEPMI_WIN2_DEFAULT != 0;
  bool use_rows_ = true; // force assembly with the table rows in LDS where they fit
  bool last_rows_form_ = false, last_fpj_form_ = false, last_scatter_form_ = false;
  int force_form_ = -1;          // set_force_form
  bool loop_ctx_ = false;        // set_loop_context
  bool virial_local_ = false;    // the virial planes of the last force evaluation hold the own-half form (exact_virials)
  double hard_factor_ = 4.0;     // set_scatter_guard: hard limit of runs whose flagged steps stand = factor x guard band
  bool ang_fused_ = true;        // set_angular_fused
  bool brick_force_ = false;     // set_brick_force (off: measured slower, see nep_brick.h)
  bool last_brick_ = false;      // the last force evaluation ran the per-brick force kernel
  bool brick_pending_ = false;   // ... and its partial forces / radial table have not been written to HBM since (materialise_for_gather)
  float* fused_img_ = nullptr;   // LDS image of the fused angular kernel (nep_fused.h), built at its first launch
  size_t fused_img_floats_ = 0;
  bool fused_img_stale_ = true;
  bool last_ang_fused_ = false;
  bool use_rmask_ = false;       // set_radial_mask (off: on PbTe 1 M atoms the radial pass gains what the force assembly's lockstep
                                 // walk over all candidates loses -- profiles/r4q_ab_mask.txt)
  bool last_mask_form_ = false;
  bool last_ang_window_ = false; // the last fused angular launch was the many-type (type window) form
  bool last_sync_form_ = false;
  int guard_delay_ = 0;          // set_scatter_guard_delayed
  int guard_delay_next_ = 0;
  bool tersoff_defer_ = false;    // this force evaluation leaves the Tersoff assembly to the next pass over the atoms (run loop, NVE)
  bool tersoff_deferred_ = false; // ... and that assembly is still due
  int64_t calm_steps_ = 0;        // run-loop steps since the last list rebuild, across calls (run_md: how often the host looks at the flags)
  int* brick_live_buf_ = nullptr; // Bufs::brick_live of decomposed runs
  double* thermo_rows_ = nullptr; // thermo records of a run call (device), copied to the host when the loop ends
  int64_t thermo_rows_cap_ = 0;
  bool rev_valid_ = false;        // Bufs::rev_ang holds the reverse slots of the current Verlet lists (ensure_reverse_slots)
  const int* dmap_dev_ = nullptr; // NepModel::dmap on the device (export_descriptors of a zero-padded model)
  double hard_asked_ = 4.0;      // set_scatter_guard: the hard factor as asked for
  double guard_delayed_ = 64.0;
  bool use_csync_ = true;
  bool ccode_valid_ = true;      // the compact radial list of the last force evaluation exists (else: the masks, Bufs::rmaskB)
  bool step_outputs_ = true;     // set_step_outputs
  int assembly_part_ = 0;        // set_assembly_part
  bool assembly_pending_ = false, pending_outputs_ = true;
  bool outputs_stale_ = false;   // the last force evaluation left the energy / virial planes as they were
  bool scatter_disabled_ = false; // a pair half left the fixed-point guard band of the scatter form: gather form from then on
  int* halo_ = nullptr;          // [bricks][wmax][4] window sums of the scatter form (fixed point)
  size_t halo_cap_ = 0;
  unsigned* fmap_ = nullptr;     // [fold_rows_][cap_] the windows that hold each atom (brick << 13 | slot)
  int fold_rows_ = 8;
  bool fold_ok_ = false;
  bool stepwise_loops_ = false; // test hook: nvt_lan / nvt_bao as the stepwise sequence // static window layout (Bufs::wtab / wcode) in use / allowed
  double* ui_alloc_ = nullptr;
  double* factor_dev_ = nullptr;
  bool tile_ok_ = false, use_tiles_ = true;
  int max_ang_rebuild_ = 1 << 30; // longest list A of the last rebuild
  int tile_mode_ = -1;           // -1 auto, 0 none, 1 radial window only, 2 radial + force windows
  bool records_valid_ = false; // rstash holds the pair records of the last evaluated positions
  int recompute_mode_ = -1;
  char* lan_states_ = nullptr; // Langevin generator states (backend-defined size per atom)
  double* lan_sums_ = nullptr;
  int lan_seed_ = 12345678;
  bool lan_fresh_ = true;
  double temperature_ = 0.0;  // set_temperature (temperature-dependent models)
  std::vector<float> b0_eff_; // hidden-layer bias with the temperature input folded in
  int ann_mode_ = 1;
  int win_lanes_ = 0;
  int win_max_atoms_ = 0; // set_win_max_atoms (0: the rule)
  bool external_skin_ = false;
#ifndef NEPMI_BRICK_FILL
#define NEPMI_BRICK_FILL 256 // A/B switch (profiles/ab_variants.sh); r3l: 253 -> 256 lets PbTe 1 M atoms take the 64^3 grid (4,096 full bricks
                             // instead of 4,913 of which 817 partly filled): force assembly 0.524 -> 0.490 ms, step 1.650 -> 1.596; 264 and 280 pick the same grid
#endif
  static constexpr int kBrickFill = NEPMI_BRICK_FILL; // of the 256 atom slots of a window-kernel pass (atoms drift between rebuilds;
                                         // a brick that does overflow just takes a second pass)
  double grid_edge_ = 0.0, grid_h_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // cell edge chosen for this box and atom count
  int64_t grid_n_ = -1;
  double* unwrapped_ = nullptr;
  bool reverse_ghosts_ = false;  // see set_reverse_ghosts
  B* radial_side_ = nullptr;     // force_kernels_on
  bool split_pending_ = false;   // compute_levels_begin ran, compute_levels_end has not yet
  int64_t num_boundary_bricks_ = 0;
  int64_t num_bricks_ = 0;
  TersoffBufs tb_{};
  TersoffParamsD tp_{};
  bool have_list_ = false;
  bool last_small_ = false;
  bool force_generic_ = false;
  int shape_ = 0;
  int64_t ncell_cap_ = 0;
  int* scan_scratch_ = nullptr;
  double* thermo_scratch_ = nullptr;
  double* thermo_dev_ = nullptr;
  double* nhc_dev_ = nullptr;
  bool nhc_fresh_ = true;
  std::mt19937 bdp_rng_{12345678u};
  int bdp_iset_ = 0;
  double bdp_gset_ = 0.0;
  std::vector<void*> allocs_;
  void dfree(void* p) // a buffer that is being replaced (halo_, fmap_): back to the device now, not at the engine's end
  {
    if (!p)
      return;
    for (size_t i = 0; i < allocs_.size(); ++i)
      if (allocs_[i] == p) {
        allocs_.erase(allocs_.begin() + i);
        break;
      }
    be_.sync(); // (a launch still reading it may be in flight)
    be_.free(p);
  }
};

} // namespace nepmi
