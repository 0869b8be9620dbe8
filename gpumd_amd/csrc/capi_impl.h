// C ABI (include/nepmi.h) on top of EngineT<NEPMI_BACKEND>.  Included exactly once by
// engine.hip (NEPMI_BACKEND = HipBackend, the product) and by tests/emu/emu.cpp (host loops, test
// infrastructure only).  The including file defines:
//   using NepmiBackend = ...;   NepmiBackend nepmi_make_backend(void* stream);
#pragma once
#include "../../include/nepmi.h"
#include "engine_impl.h"

#include <exception>
#include <string>

// Every handle starts with the function table of the library that created it (capi_dispatch.inc: a model whose shape is not
// among this library's compiled ones is served by a JIT core, capi_jit.h, and calls on its handles are forwarded there);
// nullptr in builds without the dispatch layer (tests/emu).
struct nepmi_api;
#if defined(NEPMI_CAPI_DISPATCH)
extern "C" const nepmi_api nepmi_self_api;
#define NEPMI_SELF_API (&nepmi_self_api)
#else
#define NEPMI_SELF_API nullptr
#endif

struct nepmi_model {
  const nepmi_api* api;
  nepmi::NepModel m;
};

struct nepmi_engine {
  const nepmi_api* api;
  nepmi::EngineT<NepmiBackend>* e;
};

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}

template <class F>
int guarded(F&& f)
{
  try {
    f();
    return NEPMI_OK;
  } catch (const nepmi::EngineError& e) {
    return fail(e.code, e.msg);
  } catch (const std::exception& e) {
    return fail(NEPMI_ERR_HIP, e.what());
  }
}

} // namespace

extern "C" {

const char* nepmi_last_error(void) { return g_last_error.c_str(); }
int nepmi_version(void) { return NEPMI_VERSION; }

nepmi_model* nepmi_model_load(const char* path)
{
  if (!path) {
    fail(NEPMI_ERR_ARG, "null path");
    return nullptr;
  }
  nepmi_model* m = new nepmi_model();
  m->api = NEPMI_SELF_API;
  bool unsupported = false;
  const std::string err = nepmi::load_nep_model(path, m->m, &unsupported);
  if (!err.empty()) {
    g_last_error = err;
    delete m;
    return nullptr;
  }
#if !defined(NEPMI_JIT_CORE)
  // A model whose shape has no compiled kernels here (and no JIT core took it: the public nepmi_model_load asks for one
  // first) is zero-padded into the smallest COVER shape that holds it (nep_model.h: embed_model) -- compiled kernels at the
  // price of the padded work -- instead of the run-time-shape kernels.  NEPMI_JIT=0 or NEPMI_COVER=0: the run-time-shape
  // kernels; NEPMI_FORCE_COVER=1 (measurements): pad even a model of a compiled shape.
  {
    const char* jit = std::getenv("NEPMI_JIT");
    const char* cov = std::getenv("NEPMI_COVER");
    const char* force = std::getenv("NEPMI_FORCE_COVER");
    const bool forced = force && force[0] == '1';
    const bool allowed = !(jit && jit[0] == '0') && !(cov && cov[0] == '0');
    int c[4];
    const int builtin = nepmi::builtin_shape_of(m->m);
    if (m->m.kind == 0 && (forced || (allowed && builtin == 0)) && builtin != 7 && builtin != 8 && builtin != 9 && nepmi::cover_shape_for(m->m, c)) {
      const int f[5] = {m->m.n_max_radial, m->m.basis_size_radial, m->m.n_max_angular, m->m.basis_size_angular, m->m.num_L};
      if (!nepmi::embed_model(m->m, c[0], c[1], c[2], c[3]))
        std::fprintf(stderr, "nepmi: the run-time-shape kernels serve this model, several times slower than compiled ones\n");
      else if (!forced)
        std::fprintf(stderr, "nepmi: no kernels compiled for this model's shape (n_max %d %d, basis_size %d %d, %d invariant rows): it is served, "
                             "zero-padded, by the kernels of the compiled shape (n_max %d %d, basis_size %d %d, 6 rows)\n",
                     f[0], f[2], f[1], f[3], f[4], c[0], c[2], c[1], c[3]);
    } else if (m->m.kind == 0 && builtin == 0 && !(jit && jit[0] == '0')) {
      std::fprintf(stderr, "nepmi: the run-time-shape kernels serve this model (no compiled shape holds it), several times slower than compiled ones\n");
    }
  }
#endif
  return m;
}

void nepmi_model_free(nepmi_model* m) { delete m; }

int nepmi_model_info(const nepmi_model* mm, nepmi_info* o)
{
  if (!mm || !o)
    return fail(NEPMI_ERR_ARG, "null argument");
  const nepmi::NepModel& m = mm->m;
  o->version = m.version;
  o->num_types = m.num_types;
  o->zbl_enabled = m.zbl_enabled;
  o->zbl_flexible = m.zbl_flexible;
  o->zbl_rc_inner = m.zbl_rc_inner;
  o->zbl_rc_outer = m.zbl_rc_outer;
  o->rc_radial = m.rc_radial_max;
  o->rc_angular = m.rc_angular_max;
  o->MN_radial = m.MN_radial;
  o->MN_angular = m.MN_angular;
  o->n_max_radial = m.embedded() ? m.file_n_max_radial : m.n_max_radial;
  o->n_max_angular = m.embedded() ? m.file_n_max_angular : m.n_max_angular;
  o->basis_size_radial = m.embedded() ? m.file_basis_size_radial : m.basis_size_radial;
  o->basis_size_angular = m.embedded() ? m.file_basis_size_angular : m.basis_size_angular;
  o->L_max = m.embedded() ? m.file_L_max : m.L_max;
  o->has_q_222 = m.embedded() ? m.file_has_q_222 : m.has_q_222;
  o->has_q_1111 = m.embedded() ? m.file_has_q_1111 : m.has_q_1111;
  o->num_L = m.embedded() ? m.file_num_L : m.num_L;
  o->dim = m.embedded() ? m.file_dim : m.dim;
  o->num_neurons = m.num_neurons;
  o->num_para = m.num_para;
  o->has_q_112 = m.has_q_112;
  o->has_q_123 = m.has_q_123;
  o->has_q_233 = m.has_q_233;
  o->has_q_134 = m.has_q_134;
  o->model_type = m.temperature_model ? 3 : 0;
  return NEPMI_OK;
}

const char* nepmi_model_symbol(const nepmi_model* m, int type)
{
  if (!m || type < 0 || type >= m->m.num_types)
    return "";
  return m->m.symbols[type].c_str();
}

nepmi_engine* nepmi_engine_create(const nepmi_model* m, int64_t n_atoms, void* stream)
{
  if (!m) {
    fail(NEPMI_ERR_ARG, "null model");
    return nullptr;
  }
  nepmi_engine* e = new nepmi_engine();
  e->api = NEPMI_SELF_API;
  e->e = nullptr;
  const int st = guarded([&] { e->e = new nepmi::EngineT<NepmiBackend>(m->m, n_atoms, nepmi_make_backend(stream)); });
  if (st != NEPMI_OK) {
    delete e;
    return nullptr;
  }
  return e;
}

void nepmi_engine_destroy(nepmi_engine* e)
{
  if (e) {
    delete e->e;
    delete e;
  }
}

int nepmi_force_compute(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, double* pos, double* pe,
  double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] {
    e->e->apply_pbc(h, pbc, n, pos);
    e->e->zero_properties(n, pe, force, virial);
    e->e->potential_compute(h, pbc, n, type, pos, pe, force, virial);
  });
}

int nepmi_potential_compute(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* pos,
  double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->potential_compute(h, pbc, n, type, pos, pe, force, virial); });
}

int nepmi_potential_compute_levels(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* pos,
  const signed char* level, double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->potential_compute_levels(h, pbc, n, type, pos, level, pe, force, virial); });
}

int nepmi_potential_compute_levels_begin(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* pos,
  const signed char* level, double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  bool started = false;
  const int st = guarded([&] { started = e->e->potential_compute_levels_begin(h, pbc, n, type, pos, level, pe, force, virial); });
  return st != NEPMI_OK ? st : (started ? 1 : 0);
}

int nepmi_potential_compute_levels_end(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* pos,
  const signed char* level, double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->potential_compute_levels_end(h, pbc, n, type, pos, level, pe, force, virial); });
}

int nepmi_engine_invalidate(nepmi_engine* e)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->invalidate();
  return NEPMI_OK;
}

int nepmi_apply_pbc(nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, double* pos)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->apply_pbc(h, pbc, n, pos); });
}

int nepmi_zero_properties(nepmi_engine* e, int64_t n, double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->zero_properties(n, pe, force, virial); });
}

int nepmi_average_properties(
  nepmi_engine* e, int64_t n, double denominator, double* pe, double* force, double* virial)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (!(denominator > 0.0))
    return fail(NEPMI_ERR_ARG, "average_properties: denominator should be positive");
  return guarded([&] { e->e->average_properties(n, denominator, pe, force, virial); });
}

int nepmi_vv_step1(
  nepmi_engine* e, int64_t n, double dt, const double* mass, const double* force, double* pos, double* vel)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->velocity_verlet(true, n, dt, mass, force, pos, vel, nullptr); });
}

int nepmi_vv_step2(nepmi_engine* e, int64_t n, double dt, const double* mass, const double* force, double* vel)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->velocity_verlet(false, n, dt, mass, force, nullptr, vel, nullptr); });
}

int nepmi_find_thermo(
  nepmi_engine* e, int64_t n, double volume, const double* mass, const double* pe, const double* vel,
  const double* virial, double* thermo8)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->find_thermo(n, volume, mass, pe, vel, virial, thermo8); });
}

int nepmi_run_nve(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double* pos, double* vel, double* pe, double* force, double* virial,
  int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] {
    e->e->run_md(e->e->kNve, h, pbc, n, type, mass, dt, nsteps, 0.0, 0.0, 1.0, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_berendsen_scale(
  nepmi_engine* e, int64_t n, double temperature, double coupling, const double* thermo8, double* vel)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->berendsen(n, temperature, coupling, thermo8, vel); });
}

int nepmi_run_nvt_ber(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] {
    e->e->run_md(e->e->kBer, h, pbc, n, type, mass, dt, nsteps, t1, t2, t_coup, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_nhc_init(nepmi_engine* e, int64_t n, double temperature, double t_coup, double dt, double* chain_state)
{
  if (!e || !chain_state)
    return fail(NEPMI_ERR_ARG, "bad argument");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] { e->e->nhc_init(n, temperature, t_coup, dt, chain_state); });
}

int nepmi_nhc_half_step(
  nepmi_engine* e, int64_t n, double temperature, double dt, const double* thermo8, double* chain_state,
  double* vel)
{
  if (!e || !chain_state || !thermo8 || !vel)
    return fail(NEPMI_ERR_ARG, "bad argument");
  return guarded([&] { e->e->nhc_half_step(n, temperature, dt, thermo8, chain_state, vel); });
}

int nepmi_run_nvt_nhc(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] {
    e->e->run_md(e->e->kNhc, h, pbc, n, type, mass, dt, nsteps, t1, t2, t_coup, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_engine_reset_thermostat(nepmi_engine* e)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->reset_thermostat();
  return NEPMI_OK;
}

int nepmi_bdp_seed(nepmi_engine* e, uint64_t seed)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->bdp_seed(seed);
  return NEPMI_OK;
}

int nepmi_bdp_scale(nepmi_engine* e, int64_t n, double temperature, double t_coup, const double* thermo8, double* vel)
{
  if (!e || !thermo8 || !vel)
    return fail(NEPMI_ERR_ARG, "bad argument");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] { e->e->bdp_scale(n, temperature, t_coup, thermo8, vel); });
}

int nepmi_run_nvt_bdp(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] {
    e->e->run_md(e->e->kBdp, h, pbc, n, type, mass, dt, nsteps, t1, t2, t_coup, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_lan_seed(nepmi_engine* e, int seed)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->lan_seed(seed);
  return NEPMI_OK;
}

int nepmi_lan_half_step(nepmi_engine* e, int64_t n, double temperature, double t_coup, const double* mass, double* vel)
{
  if (!e || !mass || !vel)
    return fail(NEPMI_ERR_ARG, "bad argument");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] { e->e->lan_half_step(n, temperature, t_coup, mass, vel); });
}

int nepmi_run_nvt_lan(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] {
    e->e->run_md(e->e->kLan, h, pbc, n, type, mass, dt, nsteps, t1, t2, t_coup, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_run_nvt_bao(
  nepmi_engine* e, const double h[9], const int pbc[3], int64_t n, const int* type, const double* mass,
  double dt, int64_t nsteps, double t1, double t2, double t_coup, double* pos, double* vel, double* pe,
  double* force, double* virial, int64_t thermo_every, double* thermo_host)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] {
    e->e->run_md(e->e->kBao, h, pbc, n, type, mass, dt, nsteps, t1, t2, t_coup, pos, vel, pe, force, virial, thermo_every,
                 thermo_host);
  });
}

int nepmi_neighbors_export(nepmi_engine* e, int which, int* nn, int* nl, int64_t ld)
{
  if (!e || which < 0 || which > 2 || !nn || !nl)
    return fail(NEPMI_ERR_ARG, "bad argument");
  int mx = -1;
  const int st = guarded([&] { e->e->export_lists(which, nn, nl, ld, &mx); });
  if (st != NEPMI_OK)
    return st;
  int ms, mr, ma;
  double ar, aa;
  const int st2 = guarded([&] { e->e->list_stats(ms, mr, ma, ar, aa); });
  if (st2 != NEPMI_OK)
    return st2;
  return which == 0 ? mr : which == 1 ? ma : ms;
}

int nepmi_descriptors_export(nepmi_engine* e, float* q, float* fp)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->export_descriptors(q, fp); });
}

int nepmi_engine_stats(nepmi_engine* e, int with_lists, nepmi_stats* out)
{
  if (!e || !out)
    return fail(NEPMI_ERR_ARG, "null argument");
  return guarded([&] {
    auto& eng = *e->e;
    eng.backend().sync();
    if (eng.num_compute > 0)
      eng.check_flags_now(); // overflow flags of the force calls since the last read-back
    std::memset(out, 0, sizeof(*out));
    out->num_compute = eng.num_compute;
    out->num_rebuild = eng.num_rebuild;
    if (with_lists && eng.num_compute > 0)
      eng.list_stats(out->max_nn_skin, out->max_nn_radial, out->max_nn_angular, out->mean_nn_radial, out->mean_nn_angular);
    out->ms_force_last = eng.backend().region_ms(nepmi::kRegionForce);
    const int slots[7] = {nepmi::kSlotGather, nepmi::kSlotRadial, nepmi::kSlotAngular, nepmi::kSlotAnn,
                          nepmi::kSlotAngForce, nepmi::kSlotForce, nepmi::kSlotVV};
    for (int k = 0; k < 7; ++k) {
      out->ms_kernel[k] = eng.backend().slot_ms(slots[k]);
      out->ms_kernel_sum[k] = eng.backend().slot_sum(slots[k]);
      out->launches[k] = eng.backend().slot_count(slots[k]);
    }
    out->radial_tiles = eng.tile_mode_in_use();
    out->discarded_steps = eng.num_discarded;
    out->ms_kernel_sum[7] = eng.backend().region_sum(nepmi::kRegionRebuild);
    out->launches[7] = eng.backend().region_count(nepmi::kRegionRebuild);
    out->ms_kernel[7] = eng.backend().region_ms(nepmi::kRegionRebuild);
  });
}

int nepmi_engine_describe(nepmi_engine* e, char* buf, int len)
{
  if (!e || !buf || len < 1)
    return fail(NEPMI_ERR_ARG, "bad argument");
  const std::string s = e->e->describe();
  const int n = (int)s.size() < len - 1 ? (int)s.size() : len - 1;
  std::memcpy(buf, s.data(), (size_t)n);
  buf[n] = 0;
  return n;
}

int nepmi_engine_set_timing(nepmi_engine* e, int on)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->backend().set_timing(on < 0 ? 0 : ((on >= 16 && on < 32) ? on : (on > 2 ? 1 : on)));
  e->e->num_discarded = 0; // counted over the same window as the kernel sums
  return NEPMI_OK;
}

int nepmi_engine_set_force_form(nepmi_engine* e, int mode)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (mode < -1 || mode > 1)
    return fail(NEPMI_ERR_ARG, "force form: -1 (by caller), 0 (gather), 1 (scatter)");
  e->e->set_force_form(mode);
  return NEPMI_OK;
}

int nepmi_engine_set_virial_mode(nepmi_engine* e, int mode)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  if (mode != 0 && mode != 1)
    return fail(NEPMI_ERR_ARG, "virial mode: 0 (per-atom, the reference's attribution), 1 (totals)");
  e->e->set_loop_context(mode == 1);
  return NEPMI_OK;
}

int nepmi_engine_set_temperature(nepmi_engine* e, double temperature)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  return guarded([&] { e->e->set_temperature(temperature); });
}

int nepmi_engine_set_external_skin(nepmi_engine* e, int on)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->set_external_skin(on != 0);
  return NEPMI_OK;
}

int nepmi_engine_set_unwrapped(nepmi_engine* e, double* d_unwrapped)
{
  if (!e)
    return fail(NEPMI_ERR_ARG, "null engine");
  e->e->set_unwrapped(d_unwrapped);
  return NEPMI_OK;
}

int nepmi_engine_set_option(nepmi_engine* e, const char* name, double value)
{
  if (!e || !name)
    return fail(NEPMI_ERR_ARG, "null engine or option name");
  const std::string n(name);
  const int iv = (int)value;
  auto& eng = *e->e;
  if (n == "generic")
    return guarded([&] { eng.set_force_generic(iv != 0); });
  if (n == "tiles")
    eng.set_tile_mode(iv < 0 ? -1 : iv > 2 ? 2 : iv);
  else if (n == "win_lanes")
    eng.set_win_lanes(iv);
  else if (n == "win_max_atoms")
    eng.set_win_max_atoms(iv);
  else if (n == "scatter_guard")
    eng.set_scatter_guard_delayed(value, eng.take_guard_delay());
  else if (n == "scatter_guard_hard")
    eng.set_scatter_guard(-1.0, value);
  else if (n == "scatter_guard_delay") // the NEXT "scatter_guard" applies from the value-th force assembly after it on
    eng.set_guard_delay(iv);
  else if (n == "radial_mask")
    eng.set_radial_mask(iv != 0);
  else if (n == "radial_sync")
    eng.set_radial_sync(iv != 0);
  else if (n == "angular_fused")
    eng.set_angular_fused(iv != 0);
  else if (n == "brick_force") {
    if (iv != 0 && !eng.has_brick_force())
      return fail(NEPMI_ERR_ARG, "option 'brick_force': the per-brick force kernel is not part of this build (make BRICK=1)");
    eng.set_brick_force(iv != 0);
  }
  else if (n == "win_static")
    eng.set_win2(iv != 0);
  else if (n == "stepwise_loops")
    eng.set_stepwise_loops(iv != 0);
  else if (n == "mfma")
    eng.set_use_mfma(iv);
  else if (n == "angular_recompute")
    eng.set_angular_recompute(iv);
  else
    return fail(NEPMI_ERR_ARG, "unknown engine option '" + n + "'");
  return NEPMI_OK;
}

} // extern "C"
