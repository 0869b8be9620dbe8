// Integrator bodies of the fused run loops: while nepmi_run_* (or the domain-decomposed driver) owns the step,
// positions (posq), velocities, masses and the force-path outputs live in the engine's INTERNAL (brick) order,
// so every pass of a step is a coalesced stream and nothing is permuted between list rebuilds.  The caller's
// arrays are read once at entry (ImportStateBody) and written once at exit / at list rebuilds (ExportStateBody).
//
// Replaces, per step: gpu_velocity_verlet x2 (src/integrate/ensemble.cu:176-214), gpu_apply_pbc
// (src/force/force.cu:424-459), gpu_check_atom_distance (src/force/neighbor.cu:646-684) and the
// initialize_properties memsets (force.cu:314-333; the force path assigns its outputs here).
//
// Speculative enqueue: flags[kFlagMoved] is the device-side "a list rebuild is pending" word.  The kernel that
// moves the atoms (ResidentStepBody with do_vv1) sets it to its step tag when an atom has drifted more than
// skin/2; every later kernel of the run loop finds it non-zero and returns at once, so the state stays frozen
// right after that first half-step until the host -- which keeps enqueueing steps and looks at the flag only
// every few steps -- rebuilds the lists and resumes from exactly that point.  No host round trip per step.
#pragma once
#include "nep_bodies.h"

namespace nepmi {

struct ImportStateBody {
  Bufs b;
  const double* vel;   // caller order [3][N]
  const double* mass;  // [N]
  const double* pe;    // [N]
  const double* force; // [3][N]
  const double* virial; // [9][N]
  const double* unwrapped; // [3][N] or nullptr
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N, i = b.perm[k];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      b.vi[d * N + k] = vel[d * N + i];
    b.mi[k] = mass[i];
    if (pe)
      b.fo[k] = pe[i];
    if (force) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        b.fo[(kOutF + d) * N + k] = force[d * N + i];
    }
    if (virial) {
#pragma unroll
      for (int d = 0; d < 9; ++d)
        b.fo[(kOutW + d) * N + k] = virial[d * N + i];
    }
    if (unwrapped && b.ui) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        b.ui[d * N + k] = unwrapped[d * N + i];
    }
    if (b.invp)
      b.invp[i] = (int)k;
  }
};

struct ExportStateBody {
  Bufs b;
  double* pos;    // caller order; any of them may be nullptr
  double* vel;
  double* pe;
  double* force;
  double* virial;
  double* unwrapped;
  int owned_only; // domain decomposition: ghosts carry no state
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N, i = b.perm[k];
    if (owned_only && b.lvl[k] < 2)
      return;
    if (pos) {
      const PosQ p = b.posq[k];
      pos[i] = p.x;
      pos[N + i] = p.y;
      pos[2 * N + i] = p.z;
    }
    if (vel) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        vel[d * N + i] = b.vi[d * N + k];
    }
    if (pe)
      pe[i] = b.fo[k];
    if (force) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        force[d * N + i] = b.fo[(kOutF + d) * N + k];
    }
    if (virial) {
#pragma unroll
      for (int d = 0; d < 9; ++d)
        virial[d * N + i] = b.fo[(kOutW + d) * N + k];
    }
    if (unwrapped && b.ui) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        unwrapped[d * N + i] = b.ui[d * N + k];
    }
  }
};

// Per-call entry points (Potential::compute adds to the caller's arrays): caller[perm[k]] += internal[k]
struct ScatterAddBody {
  Bufs b;
  double* pe;
  double* force;
  double* virial;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < 2)
      return;
    const int64_t i = b.perm[k];
    pe[i] += b.fo[k];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      force[d * N + i] += b.fo[(kOutF + d) * N + k];
#pragma unroll
    for (int d = 0; d < 9; ++d)
      virial[d * N + i] += b.fo[(kOutW + d) * N + k];
  }
};

// One pass over the owned atoms in internal order:
//   do_kick2: second half-kick of the step just evaluated (Ensemble_NVE::compute2)
//   do_vv1:   first half-kick + drift of the next step (compute1), gpu_apply_pbc, the skin check and the
//             lattice-jump bookkeeping of posq
// Both together are the seam between two NVE steps.  The two half-kicks stay two separately rounded additions,
// so the trajectory is bit-identical to the unfused sequence of the per-call entry points.
struct ResidentStepBody {
  BoxD box;
  Bufs b;
  double dt;
  int do_kick2, do_vv1;
  int step_tag; // > 0; written to flags[kFlagMoved] when the skin check fires
  NEPMI_HD bool frozen_now() const
  {
    const int mv = b.flags[kFlagMoved];
    return mv != 0 && mv != step_tag; // a rebuild is pending since an earlier step
  }
  NEPMI_HD void operator()(int64_t k) const
  {
    if (frozen_now())
      return;
    if (b.lvl[k] < 2)
      return;
    const int64_t N = b.N;
    const double f[3] = {b.fo[(kOutF + 0) * N + k], b.fo[(kOutF + 1) * N + k], b.fo[(kOutF + 2) * N + k]};
    step(k, f);
  }
  // the pass itself, with the atom's force handed in (TersoffSeamBody assembles it in the same kernel)
  NEPMI_HD void step(int64_t k, const double* f) const
  {
#pragma clang fp contract(off)
    const int64_t N = b.N;
    const double half = dt * 0.5;
    const double minv = 1.0 / b.mi[k];
    double v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double a = f[d] * minv;
      const double kick = a * half;
      v[d] = b.vi[d * N + k];
      if (do_kick2)
        v[d] = v[d] + kick;
      if (do_vv1)
        v[d] = v[d] + kick;
      b.vi[d * N + k] = v[d];
    }
    if (!do_vv1)
      return;
    PosQ p = b.posq[k];
    double r[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double drift = v[d] * dt;
      const double old = r[d];
      r[d] = old + drift;
      if (b.ui)
        b.ui[d * N + k] += r[d] - old; // new - old of the un-wrapped drift, like the reference
    }
    wrap_position(box, r[0], r[1], r[2]);
    float dx = (float)(r[0] - b.x0s[k]);
    float dy = (float)(r[1] - b.x0s[N + k]);
    float dz = (float)(r[2] - b.x0s[2 * N + k]);
    int n0, n1, n2;
    mic_f_img(box, dx, dy, dz, n0, n1, n2);
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!((double)d2 <= 0.25))
      NEPMI_ATOMIC_MAX(&b.flags[kFlagMoved], step_tag);
    p.x = r[0];
    p.y = r[1];
    p.z = r[2];
    p.pad = pack_img(n0, n1, n2);
    b.posq[k] = p;
    if (b.prec)
      b.prec[k] = make_prec(box, b, k, p);
  }
};

// BAOAB (Ensemble_BAO::compute1, ensemble_bao.cu:419-446) on the internal-order state: phase 1 = B (half kick; with do_kick2 the
// closing B of the previous step first) + A (half drift), then the O step (Langevin kernels), phase 2 = A (half drift) + what
// Force::compute does to the positions (wrap), the skin check and the lattice-jump / fixed-point bookkeeping of posq.
struct ResidentBaoBody {
  BoxD box;
  Bufs b;
  double dt;
  int phase, do_kick2;
  int step_tag; // phase 2: written to flags[kFlagMoved] when the skin check fires
  NEPMI_HD void operator()(int64_t k) const
  {
#pragma clang fp contract(off)
    const int mv = b.flags[kFlagMoved];
    if (mv != 0 && !(phase == 2 && mv == step_tag))
      return; // a rebuild is pending: frozen
    if (b.lvl[k] < 2)
      return;
    const int64_t N = b.N;
    const double half = dt * 0.5;
    PosQ p = b.posq[k];
    double r[3] = {p.x, p.y, p.z};
    double v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      v[d] = b.vi[d * N + k];
    if (phase == 1) {
      const double minv = 1.0 / b.mi[k];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double a = b.fo[(kOutF + d) * N + k] * minv;
        const double kick = a * half;
        if (do_kick2)
          v[d] = v[d] + kick;
        v[d] = v[d] + kick;
        b.vi[d * N + k] = v[d];
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double old = r[d];
      r[d] = half_drift_1(old, v[d], half);
      if (b.ui)
        b.ui[d * N + k] += r[d] - old;
    }
    if (phase == 1) {
      p.x = r[0];
      p.y = r[1];
      p.z = r[2];
      b.posq[k] = p;
      return;
    }
    wrap_position(box, r[0], r[1], r[2]);
    float dx = (float)(r[0] - b.x0s[k]);
    float dy = (float)(r[1] - b.x0s[N + k]);
    float dz = (float)(r[2] - b.x0s[2 * N + k]);
    int n0, n1, n2;
    mic_f_img(box, dx, dy, dz, n0, n1, n2);
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!((double)d2 <= 0.25))
      NEPMI_ATOMIC_MAX(&b.flags[kFlagMoved], step_tag);
    p.x = r[0];
    p.y = r[1];
    p.z = r[2];
    p.pad = pack_img(n0, n1, n2);
    b.posq[k] = p;
    if (b.prec)
      b.prec[k] = make_prec(box, b, k, p);
  }
};

// v *= factor for the owned atoms (internal order); factor read from device memory (Berendsen / NHC) or a constant
struct ResidentScaleBody {
  Bufs b;
  const double* factor_dev; // or nullptr
  double factor;
  NEPMI_HD void operator()(int64_t k) const
  {
    if (b.flags[kFlagMoved] != 0 || b.lvl[k] < 2)
      return;
    const double f = factor_dev ? *factor_dev : factor;
    const int64_t N = b.N;
    b.vi[k] *= f;
    b.vi[N + k] *= f;
    b.vi[2 * N + k] *= f;
  }
};

// gpu_berendsen_temperature's factor (ensemble_ber.cu:70-86) as one device scalar
struct BerendsenFactorBody {
  const int* flags;
  double temperature, coupling;
  const double* thermo;
  double* out;
  NEPMI_HD void operator()(int64_t i) const
  {
    if (i != 0 || flags[kFlagMoved] != 0)
      return;
    *out = sqrt(1.0 + coupling * (temperature / thermo[0] - 1.0));
  }
};

} // namespace nepmi
