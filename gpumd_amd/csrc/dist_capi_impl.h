// C ABI of the domain-decomposed driver (include/nepmi.h, nepmi_dist_*) on top of DistT<NepmiBackend>.
// Included after capi_impl.h by engine.hip (the product) and by tests/emu/emu.cpp (host loops, test only).
#pragma once
#include "capi_impl.h"
#include "dist_impl.h"

struct nepmi_dist {
  const nepmi_api* api; // (capi_impl.h: the library that created the handle)
  nepmi::DistT<NepmiBackend>* d;
  nepmi_engine view; // non-owning handle of the local system's engine (nepmi_dist_engine)
};

extern "C" {

void nepmi_transport_destroy(nepmi_transport* t)
{
  if (t && t->destroy && t->ctx)
    t->destroy(t->ctx);
  if (t)
    t->ctx = nullptr;
}

nepmi_dist* nepmi_dist_create(
  const nepmi_model* m, const nepmi_transport* t, const double h[9], const int pbc[3], const int grid[3], void* stream)
{
  if (!m || !t || !h || !pbc || !grid) {
    fail(NEPMI_ERR_ARG, "null argument");
    return nullptr;
  }
  nepmi_dist* d = new nepmi_dist();
  d->api = NEPMI_SELF_API;
  d->d = nullptr;
  d->view.api = NEPMI_SELF_API;
  d->view.e = nullptr;
  const int st = guarded([&] { d->d = new nepmi::DistT<NepmiBackend>(m->m, *t, h, pbc, grid, nepmi_make_backend(stream)); });
  if (st != NEPMI_OK) {
    delete d;
    return nullptr;
  }
  return d;
}

void nepmi_dist_destroy(nepmi_dist* d)
{
  if (d) {
    delete d->d;
    delete d;
  }
}

int nepmi_dist_setup(
  nepmi_dist* d, int64_t n, const int* type, const double* mass, const double* pos, const double* vel, const int64_t* ids)
{
  if (!d || n < 0)
    return fail(NEPMI_ERR_ARG, "bad argument");
  return guarded([&] { d->d->setup(n, type, mass, pos, vel, ids); });
}

int nepmi_dist_compute(nepmi_dist* d)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  if (!d->d->engine())
    return fail(NEPMI_ERR_ARG, "nepmi_dist_setup has not been called");
  return guarded([&] { d->d->compute(); });
}

int nepmi_dist_run(
  nepmi_dist* d, int ensemble, double dt, int64_t nsteps, double t1, double t2, double t_coup, int64_t thermo_every,
  double* thermo_host)
{
  if (!d || ensemble < 0 || ensemble > 5 || nsteps < 0) // 0..3 as nepmi_run_*; 4: nvt_lan, 5: nvt_bao
    return fail(NEPMI_ERR_ARG, "bad argument");
  if (!d->d->engine())
    return fail(NEPMI_ERR_ARG, "nepmi_dist_setup has not been called");
  if (thermo_every > 0 && !thermo_host)
    return fail(NEPMI_ERR_ARG, "thermo_every > 0 needs a thermo_host buffer");
  if (ensemble != 0 && t_coup < 1.0)
    return fail(NEPMI_ERR_ARG, "Temperature coupling should >= 1.");
  return guarded([&] { d->d->run(ensemble, dt, nsteps, t1, t2, t_coup, thermo_every, thermo_host); });
}

int nepmi_dist_thermo(nepmi_dist* d, double thermo8_host[8])
{
  if (!d || !thermo8_host)
    return fail(NEPMI_ERR_ARG, "null argument");
  if (!d->d->engine())
    return fail(NEPMI_ERR_ARG, "nepmi_dist_setup has not been called");
  return guarded([&] { d->d->thermo(thermo8_host); });
}

int nepmi_dist_bdp_seed(nepmi_dist* d, uint64_t seed)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  d->d->bdp_seed(seed);
  return NEPMI_OK;
}

int nepmi_dist_lan_seed(nepmi_dist* d, int seed)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  d->d->lan_seed(seed);
  return NEPMI_OK;
}

int nepmi_dist_set_overlap(nepmi_dist* d, int on)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  d->d->set_overlap(on != 0);
  return NEPMI_OK;
}

int nepmi_dist_set_ghost_mode(nepmi_dist* d, int mode)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  return guarded([&] { d->d->set_ghost_mode(mode); });
}

int nepmi_dist_get_info(nepmi_dist* d, nepmi_dist_info* out)
{
  if (!d || !out)
    return fail(NEPMI_ERR_ARG, "null argument");
  out->n_owned = d->d->num_owned();
  out->n_local = d->d->num_local();
  out->n_total = d->d->num_total();
  out->num_decompositions = d->d->num_decompositions;
  out->num_steps = d->d->num_steps;
  out->num_overlapped = d->d->num_overlapped;
  out->decompose_ms = d->d->decompose_ms;
  out->reverse_ghosts = d->d->reverse_ghosts() ? 1 : 0;
  out->num_range_handovers = d->d->num_range_handovers();
  return NEPMI_OK;
}

int nepmi_dist_info_bytes(void) { return (int)sizeof(nepmi_dist_info); }

int64_t nepmi_dist_num_overlapped_reverse(nepmi_dist* d) { return d ? d->d->num_overlapped_reverse : 0; }

int nepmi_dist_gather_owned(nepmi_dist* d, int64_t* ids, double* pos, double* vel, double* force, double* pe, double* virial)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  if (!d->d->engine())
    return fail(NEPMI_ERR_ARG, "nepmi_dist_setup has not been called");
  return guarded([&] { d->d->gather_owned(ids, pos, vel, force, pe, virial); });
}

int nepmi_dist_gather_global(nepmi_dist* d, int root, double* pos, double* vel, double* force, double* pe, double* virial)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  if (!d->d->engine())
    return fail(NEPMI_ERR_ARG, "nepmi_dist_setup has not been called");
  return guarded([&] { d->d->gather_global(root, pos, vel, force, pe, virial); });
}

int nepmi_dist_reset_thermostat(nepmi_dist* d)
{
  if (!d)
    return fail(NEPMI_ERR_ARG, "null handle");
  d->d->reset_thermostat();
  return NEPMI_OK;
}

nepmi_engine* nepmi_dist_engine(nepmi_dist* d)
{
  if (!d || !d->d->engine())
    return nullptr;
  d->view.e = d->d->engine();
  return &d->view;
}

} // extern "C"
