// Spatial domain decomposition of the NEP MD step over the GPUs of one node: the multi-GPU host of libnepmi.
//
// Replaces NEP_MULTIGPU (src/force/nep_multigpu.cu:1416-1803, chosen by Force::parse_potential, force.cu:139-160):
// the reference keeps the whole system on GPU 0, cuts it into slabs along ONE direction every step, copies slab +
// halo positions to every GPU, rebuilds each GPU's neighbour lists from scratch every step and gathers the forces
// back through blocking peer copies; the integrator runs on GPU 0.  Here:
//   * one process per GPU, a Cartesian process grid over the fractional coordinates of the global cell; every rank
//     OWNS the atoms inside its sub-box and their integrator state for the whole run;
//   * ghost shell 2 (rc + skin) with the reference's semantics (nep_multigpu.cuh:42-50, ranges N1..N5): positions
//     only travel (forward communication), descriptors of the inner ring rc + skin are recomputed redundantly
//     (level 1), forces are produced for owned atoms only (level 2), no reverse communication -- or, when the sub-boxes are
//     small against that shell (set_ghost_mode: a counted rule), shell rc + skin with descriptors for owned atoms only and a
//     reverse exchange of the pair halves the force assembly leaves on the ghosts;
//   * per step ONE grouped ghost-position exchange with the (up to 26) neighbours of the process grid -- faces, edges
//     and corners as separate direct messages, nothing forwarded: one pack launch, one transport call, one unpack
//     launch (on an 8-GPU node every peer of a 2x2x2 grid is a direct xGMI link) -- and one all-reduce of the skin
//     flag; in reverse mode one more grouped exchange of the ghosts' partial forces;  every ensemble of the fused run
//     loops (NVE, Berendsen, Nose-Hoover chain, Bussi-Donadio-Parrinello, Langevin, BAOAB) on top of one all-reduce
//     of the eight thermodynamic sums;
//   * migration + ghost-list rebuild only when some atom of some rank has moved more than skin/2 since the last
//     decomposition (the criterion of Neighbor::find_neighbor_global, neighbor.cu:741-800, made global).
//
// The transport is a small table of function pointers (include/nepmi.h: nepmi_transport): RCCL over xGMI on device
// buffers, everything enqueued on HIP streams with the skin flag reduced on the device (no host round trip per
// step), or a host transport (TCP sockets; a caller's own callbacks) whose payloads are staged through pinned
// memory -- the same driver code, used by the CPU test tier and by ranks that share one GPU.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include "../../include/nepmi.h"
#include "dist_bodies.h"
#include "engine_impl.h"

#include <algorithm>
#include <memory>

namespace nepmi {

using TransportMsg = nepmi_msg;        // include/nepmi.h
using TransportView = nepmi_transport;
enum { kDtF64 = 0, kDtI32 = 1, kDtI64 = 2 };
enum { kOpSum = 0, kOpMax = 1 };

template <class B>
class DistT
{
public:
  using Engine = EngineT<B>;
  static constexpr double kSkin = Engine::kSkin;

  DistT(const NepModel& model, const TransportView& tr, const double h9[9], const int pbc[3], const int grid[3], B backend)
    : model_(model), tr_(tr), be_(backend)
  {
    if (grid[0] * grid[1] * grid[2] != tr.nranks)
      throw EngineError{-4, "process grid does not match the number of ranks"};
    BoxD gb;
    box_from_h9(h9, pbc, gb);
    DomainGeom& g = geom_;
    for (int k = 0; k < 9; ++k) {
      g.H[k] = gb.h[k];
      g.G[k] = gb.h[9 + k];
      hg_[k] = h9[k];
    }
    volume_ = gb.volume;
    const int r = tr.rank;
    g.coords[0] = r % grid[0];
    g.coords[1] = (r / grid[0]) % grid[1];
    g.coords[2] = r / (grid[0] * grid[1]);
    for (int d = 0; d < 3; ++d) {
      g.grid[d] = grid[d];
      g.pbc[d] = pbc[d] ? 1 : 0;
      g.decomposed[d] = grid[d] > 1;
      g.lo[d] = (double)g.coords[d] / grid[d];
      g.hi[d] = (double)(g.coords[d] + 1) / grid[d];
      thick_[d] = gb.thickness[d];
      pbc_loc_[d] = g.decomposed[d] ? 0 : g.pbc[d];
    }
    if (const char* env = std::getenv("NEPMI_DIST_GHOSTS")) // experiments: "forward" / "reverse" instead of the counted rule
      ghost_mode_ = env[0] == 'r' ? 1 : (env[0] == 'f' ? 0 : -1);
    configure_shell();
    flag_dev_ = (int*)be_.alloc(sizeof(int) * 4);
    sums_dev_ = (double*)be_.alloc(sizeof(double) * 8);
    thermo_dev_ = (double*)be_.alloc(sizeof(double) * 8);
    nhc_dev_ = (double*)be_.alloc(sizeof(double) * kNhcStateSize);
    factor_dev_ = (double*)be_.alloc(sizeof(double));
  }

  ~DistT()
  {
    for (void* p : {(void*)flag_dev_, (void*)sums_dev_, (void*)thermo_dev_, (void*)nhc_dev_, (void*)factor_dev_})
      be_.free(p);
    free_state(cur_);
    free_state(nxt_);
    free_plan();
    for (void* p : scratch_)
      be_.free(p);
    for (auto& kv : pool_size_)
      be_.free(kv.first);
  }

  // ---- setup: the atoms this rank starts with (DEVICE arrays, any position: they are migrated to their owners) ----
  void setup(int64_t n, const int* type, const double* mass, const double* pos, const double* vel, const int64_t* ids)
  {
    // global ids default to a running number over the ranks
    std::vector<int64_t> all((size_t)tr_.nranks, 0);
    all[tr_.rank] = n;
    host_allreduce(all.data(), tr_.nranks, kDtI64, kOpSum);
    int64_t base = 0;
    n_total_ = 0;
    for (int r = 0; r < tr_.nranks; ++r) {
      if (r < tr_.rank)
        base += all[r];
      n_total_ += all[r];
    }
    // caller-supplied ids label the gathered output and seed the Langevin generators: 0 .. n_total - 1 (checked here on every
    // rank together; uniqueness is the caller's to keep -- nothing on the device is indexed by an id during a run)
    int64_t bad_ids = 0;
    if (ids && n > 0) {
      std::vector<int64_t> hid((size_t)n);
      be_.d2h(hid.data(), ids, sizeof(int64_t) * n);
      for (int64_t i = 0; i < n; ++i)
        if (hid[i] < 0 || hid[i] >= n_total_)
          bad_ids = 1;
    }
    host_allreduce(&bad_ids, 1, kDtI64, kOpSum);
    if (bad_ids)
      throw EngineError{-4, "nepmi_dist_setup: atom ids must be 0 .. n_total - 1"};
    alloc_state(cur_, n > 0 ? n : 1);
    cur_.n = n;
    cur_.n_own = n;
    if (n > 0) {
      // global -> local coordinates
      std::vector<double> hx(3 * (size_t)n);
      be_.d2h(hx.data(), pos, sizeof(double) * 3 * n);
      for (int d = 0; d < 3; ++d)
        for (int64_t i = 0; i < n; ++i)
          hx[d * n + i] -= geom_.origin[d];
      be_.h2d(cur_.x, hx.data(), sizeof(double) * 3 * n);
      copy_dev(cur_.v, vel, sizeof(double) * 3 * n);
      copy_dev(cur_.m, mass, sizeof(double) * n);
      copy_dev(cur_.t, type, sizeof(int) * n);
      if (ids) {
        copy_dev(cur_.id, ids, sizeof(int64_t) * n);
      } else {
        std::vector<int64_t> hid((size_t)n);
        for (int64_t i = 0; i < n; ++i)
          hid[i] = base + i;
        be_.h2d(cur_.id, hid.data(), sizeof(int64_t) * n);
      }
    }
    resident_ = false;
    have_force_ = false;
    lan_fresh_ = true; // new atoms: their generator states are created from (seed, global id) when a Langevin run starts
    decompose();
  }

  // Ghost shell and what the ghosts are for.
  //   forward (0): shell 2 (rc + skin), the reference's ranges (nep_multigpu.cuh:42-50): the inner ring's descriptors are
  //     recomputed, forces only on owned atoms, one exchange per step (positions);
  //   reverse (1): shell rc + skin, descriptors / ANN / partial forces for owned atoms only; the force assembly runs on the
  //     ghosts as well and yields there the pair halves -f21 that this rank's owned atoms contribute (a ghost's own rows are
  //     zero), which travel back to the owners: a second exchange per step (3 doubles per ghost), no redundant descriptor
  //     work -- the strong-scaling form (SURVEY.md:541-546);
  //   -1: the counted rule -- reverse when the forward shell would leave less than 70 % of the local atoms owned (uniform
  //     density: the volume ratio of the sub-box and its padded box), or when the sub-box is too thin for the forward shell.
  void set_ghost_mode(int mode)
  {
    if (cur_.cap != 0)
      throw EngineError{-4, "nepmi_dist_set_ghost_mode: call it before nepmi_dist_setup"};
    if (mode < -1 || mode > 1)
      throw EngineError{-4, "ghost mode: -1 (counted rule), 0 (forward), 1 (reverse)"};
    ghost_mode_ = mode;
    configure_shell();
  }
  bool reverse_ghosts() const { return reverse_; }

  int64_t num_owned() const { return cur_.n_own; }
  int64_t num_local() const { return cur_.n; }
  int64_t num_total() const { return n_total_; }
  int64_t num_decompositions = 0, num_steps = 0;
  Engine* engine() { return eng_.get(); }

  // ---- Force::compute on the decomposed system (the initial force of Run::perform_a_run) ----
  void compute()
  {
    halo_exchange();
    eng_->force_kernels(Engine::kPhaseAll);
    force_reverse();
    // The scatter form's guard band (nep_scatter.h): one word, reduced over the ranks -- whichever forms they ran -- so that
    // all of them repeat the evaluation in the gather form, or none (a per-call evaluation never stands flagged).
    {
      int* word = eng_->bufs().flags + kFlagRange;
      device_allreduce(word, 1, kDtI32, kOpMax);
      int w = 0;
      be_.d2h(&w, word, sizeof(int));
      if (w) {
        eng_->disable_scatter();
        eng_->clear_range_hard();
        halo_exchange(); // (the ghosts' positions are in place; the exchange keeps the step's collectives paired on every rank)
        eng_->force_kernels(Engine::kPhaseAll);
        force_reverse();
      }
    }
    ++eng_->num_compute;
    have_force_ = true;
  }
  int64_t num_range_handovers() const { return eng_ ? eng_->num_range_handovers : 0; } // (nepmi_dist_info)

  // ---- the run loop (Run::perform_a_run, run.cu:250-318) for the ensembles of EngineT::run_md ----
  void run(int ens, double dt, int64_t nsteps, double t1, double t2, double tcoup, int64_t thermo_every, double* thermo_host)
  {
    if (!have_force_)
      compute();
    Engine* e = eng_.get(); // re-fetched after every decompose(): a grown local system replaces the engine
    if (ens == Engine::kNhc && nhc_fresh_) { // a fresh chain per `run` (integrate.cu:85-92), kept across the calls of one
      be_.template launch<64>(kSlotMisc, 1, NhcInitBody{n_total_, t1, tcoup, dt, nhc_dev_});
      nhc_fresh_ = false;
    }
    const bool spec = tr_.device_buffers != 0 && ens != Engine::kBdp; // speculative enqueue needs a device-side vote
    auto tag_of = [](int64_t step) { return (int)(step % 1000000000) + 1; };
    auto target_of = [&](int64_t step) { return t1 + (t2 - t1) * ((double)step / (double)nsteps); };
    auto nhc_half = [&](double target) {
      thermo_global();
      be_.template launch<64>(kSlotMisc, 1, NhcChainBody{n_total_, target, 0.5 * dt, thermo_dev_, nhc_dev_, frozen()});
      be_.template launch<256>(kSlotVV, e->num_atoms(), ResidentScaleBody{e->bufs(), nhc_dev_ + 3 * kNhcLinks, 1.0});
    };
    // Ensemble_LAN (ensemble_lan.cu:96-127, :206-262) on the decomposed system.  The generator state of an atom is per-atom
    // state like its velocity (ensemble_lan.cu:96-127): it is created from (seed, GLOBAL id), lives on the rank that owns the
    // atom (State::rng, local order) and migrates with it at a re-decomposition, so it only ever advances on its owner -- an
    // atom's noise is the single-domain run's whatever the decomposition, and a rank holds and touches the states of its own
    // atoms only.  The four momentum sums are all-reduced before the centre-of-mass velocity is removed.
    if (ens == Engine::kLan || ens == Engine::kBao)
      lan_prepare();
    auto lan_half = [&](double target, bool whole_step = false) { // whole_step: the O of BAOAB
      const double c1 = std::exp((whole_step ? -1.0 : -0.5) / tcoup);
      const double c2 = std::sqrt((1.0 - c1 * c1) * kBoltzmann * target);
      const Bufs& bb = e->bufs();
      be_.lan_kick_resident(cur_.rng, e->num_atoms(), c1, c2, bb.mi, bb.vi, bb.perm, bb.lvl, nullptr, bb.flags);
      be_.lan_momentum_resident(e->num_atoms(), bb.mi, bb.vi, nullptr, bb.lvl, lan_sums_, bb.flags);
      device_allreduce(lan_sums_, 4, kDtF64, kOpSum);
      be_.lan_momentum_fix_resident(e->num_atoms(), lan_sums_, bb.vi, bb.lvl, bb.flags);
    };
    // temperature-dependent NEP: as in EngineT::run_md (every rank sets the same value)
    const bool temp_ramp = e->temperature_model() && ens != Engine::kNve && t1 != t2;
    if (e->temperature_model() && ens != Engine::kNve && e->temperature() != t1)
      e->set_temperature(t1);
    std::vector<int> pending;
    int ring_next = 0;
    int64_t step = 0;
    bool resume_after_vv1 = false, kick2_pending = false;
    while (step < nsteps) {
      const double target = target_of(step);
      if (temp_ramp)
        e->set_temperature(t1 + (t2 - t1) * ((double)(step + 2) / (double)nsteps));
      if (!resume_after_vv1) {
        if (ens == Engine::kNhc)
          nhc_half(target);
        if (ens == Engine::kLan)
          lan_half(target);
        if (ens == Engine::kBao) { // B A O A (Ensemble_BAO::compute1); noise amplitude of T1
          be_.template launch<256>(kSlotVV, e->num_atoms(), ResidentBaoBody{e->box(), e->bufs(), dt, 1, 0, tag_of(step)});
          lan_half(t1, true);
          be_.template launch<256>(kSlotVV, e->num_atoms(), ResidentBaoBody{e->box(), e->bufs(), dt, 2, 0, tag_of(step)});
        } else {
          be_.template launch<256>(kSlotVV, e->num_atoms(),
                                   ResidentStepBody{e->box(), e->bufs(), dt, kick2_pending ? 1 : 0, 1, tag_of(step)});
        }
      }
      resume_after_vv1 = false;
      kick2_pending = false;
      // The vote (every rank's skin flag -> the same word on all ranks) and the ghost positions travel on the
      // communication stream; meanwhile the radial pass of the interior bricks -- those whose window holds no ghost --
      // runs on the compute stream.  Host transports block: the interior pass simply runs first, which proves its
      // independence of this step's ghosts (tests/test_dist.py compares both orders bit for bit).
      const bool record = thermo_every > 0 && (step + 1) % thermo_every == 0;
      const bool last = step + 1 == nsteps;
      // per-atom energies and virials are read at thermo records and at the exit only (EngineT::set_step_outputs)
      auto force_phase = [&](int phase, const int* frz) {
        e->set_step_outputs(record || last);
        e->force_kernels(phase, frz);
      };
      const bool split = overlap_ && e->tiles_active() && tr_.nranks > 1;
      B& comm = (split && tr_.device_buffers) ? comm_backend() : be_;
      if (&comm != &be_)
        be_.fork_to(comm);
      if (split) {
        force_phase(Engine::kPhaseInterior, nullptr);
        ++num_overlapped;
      }
      int trip = vote(spec, comm);
      if (!trip)
        halo_exchange_on(comm);
      // the boundary bricks' radial pass right behind the unpack on the communication stream: it runs beside the tail of the
      // interior launch instead of after it (two launches in a row cost two ramps and two tails of one brick's latency)
      const bool side_radial = split && &comm != &be_ && !trip;
      if (side_radial)
        e->force_kernels_on(comm, Engine::kPhaseBoundaryRadial, frozen());
      if (&comm != &be_)
        be_.join_from(comm);
      // The host's look at the voted words (every kPollEvery-th step, kPollDepth looks in flight) is taken HERE, behind the vote
      // and in front of the force phase: all three words of the snapshot are then the reduced ones -- the same on every rank.
      // (Taken behind the force phase, the range word could also hold what THIS rank's scatter kernel raised during the step,
      // one look before the others see it: one rank alone would have left the loop's collective sequence.)
      const bool look = !trip && spec && !(record || last || ens == Engine::kBdp) && (step + 1) % Engine::kPollEvery == 0;
      if (look) {
        be_.poll_record(ring_next, e->bufs().flags);
        pending.push_back(ring_next);
        ring_next = (ring_next + 1) % 8;
      }
      if (!trip) {
        // Reverse-mode ghosts with the overlap on: the boundary bricks' force assembly and the ghosts' fold first, then the
        // ghosts' partial forces travel on the communication stream while the interior bricks (and the owned atoms' fold) run
        // on the compute stream; the returned parts are added after both (bit-identical to the plain order: the same integer
        // sums, the same additions in the same order).  Host transports block: the exchange simply runs first.
        const bool rsplit = overlap_ && reverse_ && tr_.nranks > 1;
        if (rsplit)
          e->set_assembly_part(1);
        force_phase(split ? (side_radial ? Engine::kPhaseAfterRadial : Engine::kPhaseBoundary) : Engine::kPhaseAll, frozen());
        if (rsplit && e->assembly_pending()) {
          B& c2 = tr_.device_buffers ? comm_backend() : be_;
          if (&c2 != &be_)
            be_.fork_to(c2);
          virial_folded_ = false;
          reverse_send(c2, kOutF, 3, frozen());
          e->force_assembly_rest(frozen());
          if (&c2 != &be_)
            be_.join_from(c2);
          reverse_add(kOutF, 3, frozen());
          ++num_overlapped_reverse;
        } else {
          force_reverse(frozen());
        }
        bool need_sync = record || last;
        if (ens == Engine::kNve && !record && !last) {
          kick2_pending = true;
        } else {
          be_.template launch<256>(kSlotVV, e->num_atoms(), ResidentStepBody{e->box(), e->bufs(), dt, 1, 0, 0});
          if (ens == Engine::kBer) {
            thermo_global();
            if (1.0 / tcoup > 1.0e-5) {
              be_.template launch<64>(kSlotMisc, 1,
                                      BerendsenFactorBody{e->bufs().flags, target, 1.0 / tcoup, thermo_dev_, factor_dev_});
              be_.template launch<256>(kSlotVV, e->num_atoms(), ResidentScaleBody{e->bufs(), factor_dev_, 1.0});
            }
          } else if (ens == Engine::kNhc) {
            nhc_half(target);
          } else if (ens == Engine::kBdp) {
            thermo_global();
            need_sync = true;
          } else if (ens == Engine::kLan) {
            lan_half(target);
            if (record)
              thermo_global();
          } else if (record) {
            thermo_global();
          }
        }
        if (need_sync) {
          trip = sync_flags();
          pending.clear();
          if (!trip) {
            if (ens == Engine::kBdp) {
              double T = 0.0;
              be_.d2h(&T, thermo_dev_, sizeof(double));
              be_.template launch<256>(kSlotVV, e->num_atoms(),
                                       ResidentScaleBody{e->bufs(), nullptr, e->bdp_factor(n_total_, T, target, tcoup)});
            }
            if (record && thermo_host)
              be_.d2h(thermo_host + 8 * ((step + 1) / thermo_every - 1), thermo_dev_, 8 * sizeof(double));
          }
        } else if (look) {
          if ((int)pending.size() > Engine::kPollDepth) {
            int snap[8];
            be_.poll_wait(pending.front(), snap);
            pending.erase(pending.begin());
            if (snap[kFlagMoved])
              trip = sync_flags();
            else if (snap[kFlagRange] && e->scatter_enabled()) {
              // the scatter form's guard band, voted: every rank reads the same snapshot of the same step and leaves the form
              // here; the flagged steps stand (a value beyond the hard limit met meanwhile raised kOverflowRangeHard, which the
              // vote carries to every rank as well)
              trip = sync_flags();
            }
          }
        }
      }
      if (trip) {
        // frozen right after the first half-step of step `trip - 1` on every rank: re-decompose and resume there
        be_.sync();
        pending.clear();
        e->num_discarded += step - ((int64_t)trip - 1) + 1;
        step = (int64_t)trip - 1;
        decompose();
        e = eng_.get();
        resume_after_vv1 = true;
        kick2_pending = false;
        continue;
      }
      ++step;
    }
    num_steps += nsteps;
    // the last step's kernels may have raised a capacity bit after its vote: reduce the words once more, so that the
    // final check throws on every rank or on none
    device_allreduce(e->bufs().flags + kFlagMoved, 3, kDtI32, kOpMax);
    be_.sync();
    e->check_flags_now();
  }

  // T, U and the six stress components of the whole system -> thermo8 (HOST)
  void thermo(double* thermo8_host)
  {
    thermo_global();
    be_.d2h(thermo8_host, thermo_dev_, 8 * sizeof(double));
  }

  // owned atoms -> the caller's DEVICE arrays (n_own entries per plane, global coordinates, local order)
  void gather_owned(int64_t* ids, double* pos, double* vel, double* force, double* pe, double* virial)
  {
    Engine& e = *eng_;
    // Per-atom virials leave the engine: the reference's attribution (a virial-only pass of the gather form after scatter-form
    // steps) and, with reverse ghosts, the halves computed on other ranks' ghosts.  Both on EVERY call, whether or not this
    // rank asked for the virials: the fold is a collective, and a caller that passes NULL on some ranks must not leave the
    // others waiting in it (as gather_global does).
    e.exact_virials();
    fold_virial();
    (void)virial;
    be_.template launch<256>(kSlotMisc, e.num_atoms(),
                             GatherOwnedBody{e.bufs(), geom_, cur_.n_own, cur_.id, ids, pos, vel, force, pe, virial});
    be_.sync();
  }

  // Every atom of the system to rank `root`, in the order of the global ids 0 .. n_total-1 (DEVICE arrays with
  // n_total entries per plane on the root, ignored elsewhere): what dump_xyz / dump_restart of a multi-GPU run write.
  void gather_global(int root, double* pos, double* vel, double* force, double* pe, double* virial)
  {
    Engine& e = *eng_;
    e.exact_virials();
    fold_virial(); // collective: every rank, whether or not the root asked for the virials
    const int P = tr_.nranks, me = tr_.rank;
    const int64_t no = cur_.n_own;
    std::vector<int64_t> cnt((size_t)P, 0);
    cnt[me] = no;
    host_allreduce(cnt.data(), P, kDtI64, kOpSum);
    double* mine = (double*)be_.alloc(sizeof(double) * 20 * (no > 0 ? no : 1));
    if (no > 0)
      be_.template launch<256>(kSlotMisc, e.num_atoms(), PackGlobalBody{e.bufs(), geom_, no, cur_.id, mine});
    be_.sync();
    int* bad = iscratch(11, 1);
    int zero = 0;
    be_.h2d(bad, &zero, sizeof(int));
    if (me != root) {
      TransportMsg s{mine, (int64_t)sizeof(double) * 20 * no, root};
      if (no > 0)
        exchange(1, &s, 0, nullptr);
    } else {
      for (int r = 0; r < P; ++r) {
        const int64_t c = cnt[r];
        if (c == 0)
          continue;
        double* buf = mine;
        if (r != me) {
          buf = (double*)be_.alloc(sizeof(double) * 20 * c);
          TransportMsg m{buf, (int64_t)sizeof(double) * 20 * c, r};
          exchange(0, nullptr, 1, &m);
        }
        be_.template launch<256>(kSlotMisc, c, ScatterGlobalBody{c, n_total_, buf, pos, vel, force, pe, virial, bad});
        be_.sync();
        if (r != me)
          be_.free(buf);
      }
      int hb = 0;
      be_.d2h(&hb, bad, sizeof(int));
      if (hb)
        throw EngineError{-4, "gather_global: atom ids must be 0 .. n_total - 1"};
    }
    be_.free(mine);
  }
  void reset_thermostat() { nhc_fresh_ = true; }

  void bdp_seed(uint64_t seed) { seed_ = seed; if (eng_) eng_->bdp_seed(seed); }
  // Langevin thermostat: the seed of the per-atom generators (the same on every rank; state s = atom with global id s)
  void lan_seed(int seed)
  {
    lan_seed_ = seed;
    lan_fresh_ = true;
  }
  void lan_prepare()
  {
    const size_t sb = be_.lan_state_bytes();
    if (!lan_on_ || !cur_.rng) {
      lan_on_ = true; // from now on the states are part of the per-atom state (alloc_state, decompose)
      if (!cur_.rng)
        cur_.rng = (char*)be_.alloc(sb * (size_t)cur_.cap);
      lan_fresh_ = true;
    }
    if (!lan_sums_)
      lan_sums_ = (double*)be_.alloc(sizeof(double) * 4);
    if (lan_fresh_) { // state of the atom with global id g = the single-domain run's state g (hiprand_init(seed, g, 0))
      if (cur_.n_own > 0)
        be_.lan_init_ids(cur_.rng, cur_.n_own, cur_.id, lan_seed_);
      lan_fresh_ = false;
    }
  }
  void set_overlap(bool on) { overlap_ = on; }
  int64_t num_overlapped = 0; // steps whose interior radial pass ran before / while the ghosts travelled
  int64_t num_overlapped_reverse = 0; // steps whose interior force assembly ran while the ghosts' partial forces travelled
  double decompose_ms = 0.0;  // wall time of all (re-)decompositions (migration, ghost stages, list rebuild), synchronised

private:
  struct State { // owned + ghost atoms in local order, stride = n (DEVICE)
    int64_t cap = 0, n = 0, n_own = 0;
    double *x = nullptr, *v = nullptr, *m = nullptr;
    int* t = nullptr;
    int64_t* id = nullptr;
    signed char* lvl = nullptr;
    char* rng = nullptr; // Langevin generator states of the owned atoms, local order (allocated once a Langevin ensemble runs)
  };
  // The halo of the current decomposition: who gets which owned atoms, where the ghosts sit (dist_bodies.h: direct halo).
  struct Peer {
    int rank = 0;
    int64_t cnt_send = 0, cnt_recv = 0, send_off = 0, recv_off = 0; // entries; blocks of the buffers
  };
  struct Link { // everything that travels between this rank and ONE other rank is one message each way
    int rank = 0;
    int64_t send_off = 0, cnt_send = 0, recv_off = 0, cnt_recv = 0;
  };
  struct Plan {
    PeerTable pt;
    // Peers ordered by (rank, grid offset ascending, z slowest): the send entries of all peers that are the same rank (a
    // direction with two ranks and periodic images; any small grid) are contiguous and travel as ONE message.  What comes back
    // from that rank is the concatenation of ITS blocks for us in ITS ascending offset order -- our offsets negated, i.e. our
    // peers of that rank in DESCENDING order: the receive blocks of a rank are laid out that way (Peer::recv_off).
    std::vector<Peer> peers;
    std::vector<Link> links; // one per distinct peer rank, ascending
    int64_t n_send = 0, n_recv = 0, n_src = 0;
    int* send_idx = nullptr;            // [n_send] local indices (owned atoms), peer-major
    int* send_int = nullptr;            // [n_send] the same as internal indices of the engine
    unsigned char* send_peer = nullptr; // [n_send] peer of every entry
    int* recv_int = nullptr;            // [n_recv] internal indices of the ghosts (peer-major, ascending peers)
    double* sendbuf = nullptr;          // [n_send][bw]
    double* recvbuf = nullptr;          // [n_recv][bw]
    int *src_int = nullptr, *src_start = nullptr, *src_entry = nullptr; // reverse path: the entries of every shell atom
    int* src_idx = nullptr;             // [n_src] local indices of the shell atoms
  };

  const int* frozen() const { return eng_->bufs().flags + kFlagMoved; }
  B& comm_backend()
  {
    if (!side_ready_) {
      side_ = be_.make_side_stream();
      side_ready_ = true;
    }
    return side_;
  }

  void copy_dev(void* dst, const void* src, size_t bytes) { be_.d2d(dst, src, bytes); }

  void alloc_state(State& s, int64_t cap)
  {
    free_state(s);
    s.cap = cap;
    s.x = (double*)be_.alloc(sizeof(double) * 3 * cap);
    s.v = (double*)be_.alloc(sizeof(double) * 3 * cap);
    s.m = (double*)be_.alloc(sizeof(double) * cap);
    s.t = (int*)be_.alloc(sizeof(int) * cap);
    s.id = (int64_t*)be_.alloc(sizeof(int64_t) * cap);
    s.lvl = (signed char*)be_.alloc(cap);
    s.rng = lan_on_ ? (char*)be_.alloc(be_.lan_state_bytes() * (size_t)cap) : nullptr;
  }
  // Transient device buffers of the (re-)decomposition -- migration payloads, per-stage index lists and send/receive
  // buffers, the 8-byte bounce words of the count exchange -- come from a small pool: hipMalloc / hipFree cost 0.1-1 ms
  // each (hipFree synchronises the device) and a re-decomposition asked for about thirty of them.  A freed block is
  // handed out again when it is large enough and not more than twice the request; everything returns to HIP in ~DistT.
  void* palloc(size_t bytes)
  {
    bytes = bytes ? bytes : 1;
    auto it = pool_free_.lower_bound(bytes);
    if (it != pool_free_.end() && it->first <= 2 * bytes + 4096) {
      void* p = it->second;
      pool_free_.erase(it);
      return p;
    }
    const size_t cap = bytes + bytes / 4 + 256; // headroom: the next decomposition asks for slightly different sizes
    void* p = be_.alloc(cap);
    pool_size_[p] = cap;
    return p;
  }
  void pfree(void* p)
  {
    if (p)
      pool_free_.emplace(pool_size_.at(p), p);
  }
  std::multimap<size_t, void*> pool_free_;
  std::unordered_map<void*, size_t> pool_size_;

  void free_state(State& s)
  {
    for (void* p : {(void*)s.x, (void*)s.v, (void*)s.m, (void*)s.t, (void*)s.id, (void*)s.lvl, (void*)s.rng})
      if (p)
        be_.free(p);
    s = State();
  }
  void free_plan()
  {
    Plan& h = plan_;
    for (void* p : {(void*)h.send_idx, (void*)h.send_int, (void*)h.send_peer, (void*)h.recv_int, (void*)h.sendbuf, (void*)h.recvbuf,
                    (void*)h.src_int, (void*)h.src_start, (void*)h.src_entry, (void*)h.src_idx})
      pfree(p);
    h = Plan();
  }
  int* iscratch(int which, int64_t count) // grow-only int scratch arrays
  {
    if ((int)iscr_.size() <= which) {
      iscr_.resize(which + 1, nullptr);
      iscr_cap_.resize(which + 1, 0);
    }
    if (iscr_cap_[which] < count) {
      iscr_[which] = (int*)be_.alloc(sizeof(int) * (count + 1024));
      scratch_.push_back(iscr_[which]);
      iscr_cap_[which] = count + 1024;
    }
    return iscr_[which];
  }

  int neighbor(int d, int step) const
  {
    int c[3] = {geom_.coords[0], geom_.coords[1], geom_.coords[2]};
    c[d] = (c[d] + step + geom_.grid[d]) % geom_.grid[d];
    return c[0] + geom_.grid[0] * (c[1] + geom_.grid[1] * c[2]);
  }

  // ---- transport helpers: device buffers go straight to the transport, host transports are staged ----
  void host_allreduce(void* buf, int64_t count, int dtype, int op)
  {
    if (tr_.nranks == 1)
      return;
    if (tr_.device_buffers) { // a host buffer through a device transport: bounce through device memory
      const size_t bytes = (size_t)count * (dtype == kDtI32 ? 4 : 8);
      void* d = palloc(bytes);
      be_.h2d(d, buf, bytes);
      if (tr_.allreduce(tr_.ctx, d, count, dtype, op, be_.stream_handle()) != 0)
        throw EngineError{-5, "transport all-reduce failed"};
      be_.d2h(buf, d, bytes);
      pfree(d);
    } else if (tr_.allreduce(tr_.ctx, buf, count, dtype, op, nullptr) != 0) {
      throw EngineError{-5, "transport all-reduce failed"};
    }
  }
  void device_allreduce(void* dbuf, int64_t count, int dtype, int op) { device_allreduce_on(be_, dbuf, count, dtype, op); }
  void device_allreduce_on(B& on, void* dbuf, int64_t count, int dtype, int op)
  {
    if (tr_.nranks == 1)
      return;
    if (tr_.device_buffers) {
      if (tr_.allreduce(tr_.ctx, dbuf, count, dtype, op, on.stream_handle()) != 0)
        throw EngineError{-5, "transport all-reduce failed"};
    } else {
      char tmp[64];
      dtype &= 0xFF;
      const size_t bytes = (size_t)count * (dtype == kDtI32 ? 4 : 8);
      be_.d2h(tmp, dbuf, bytes); // synchronises the stream
      if (tr_.allreduce(tr_.ctx, tmp, count, dtype, op, nullptr) != 0)
        throw EngineError{-5, "transport all-reduce failed"};
      be_.h2d(dbuf, tmp, bytes);
    }
  }
  // grouped exchange of DEVICE buffers
  void exchange(int ns, TransportMsg* sends, int nr, TransportMsg* recvs) { exchange_on(be_, ns, sends, nr, recvs); }
  void exchange_on(B& on, int ns, TransportMsg* sends, int nr, TransportMsg* recvs)
  {
    if (tr_.nranks == 1)
      return;
    if (tr_.device_buffers) {
      if (tr_.exchange(tr_.ctx, ns, sends, nr, recvs, on.stream_handle()) != 0)
        throw EngineError{-5, "transport exchange failed"};
      return;
    }
    std::vector<std::vector<char>> hs(ns), hr(nr);
    std::vector<TransportMsg> s2(ns), r2(nr);
    for (int k = 0; k < ns; ++k) {
      hs[k].resize((size_t)sends[k].bytes);
      if (sends[k].bytes)
        be_.d2h(hs[k].data(), sends[k].buf, (size_t)sends[k].bytes);
      s2[k] = TransportMsg{hs[k].data(), sends[k].bytes, sends[k].peer};
    }
    for (int k = 0; k < nr; ++k) {
      hr[k].resize((size_t)recvs[k].bytes);
      r2[k] = TransportMsg{hr[k].data(), recvs[k].bytes, recvs[k].peer};
    }
    if (tr_.exchange(tr_.ctx, ns, s2.data(), nr, r2.data(), nullptr) != 0)
      throw EngineError{-5, "transport exchange failed"};
    for (int k = 0; k < nr; ++k)
      if (recvs[k].bytes)
        be_.h2d(recvs[k].buf, hr[k].data(), (size_t)recvs[k].bytes);
  }

  // the global skin vote after a first half-step: returns the trip tag when it can be known now (host transports
  // and non-speculative ensembles), 0 otherwise (device transports: the kernels look at the reduced word themselves)
  int vote(bool spec, B& on)
  {
    // [moved, overflow] are adjacent: the capacity bits travel with the vote, so that every rank sees a capacity error at
    // the same synchronisation point and they all report it (a rank that threw alone would leave the others waiting in
    // their next collective)
    static_assert(kFlagOverflow == kFlagMoved + 1 && kFlagRange == kFlagMoved + 2, "the vote reduces three adjacent flag words");
    int* word = eng_->bufs().flags + kFlagMoved;
    // a transport that can (nepmi.h: NEPMI_DT_DEFER) posts the reduction inside the group of the ghost exchange that follows on
    // the same stream: the word is first read by the kernels behind that exchange
    const bool defer = spec && (tr_.device_buffers & 2) != 0;
    device_allreduce_on(on, word, 3, kDtI32 | (defer ? NEPMI_DT_DEFER : 0), kOpMax);
    if (spec)
      return 0;
    int w = 0;
    be_.d2h(&w, word, sizeof(int));
    return w;
  }
  // A full look at the flags, taken by every rank at the same step (thermo records, the last step, a voted skin or range
  // trip).  The three voted words are reduced once more first: the step's force kernels may have raised a capacity bit, the
  // hard-limit bit or the range word on ONE rank after its vote, and a rank that threw or left the scatter form alone would
  // leave the others waiting in their next collective.
  int sync_flags()
  {
    device_allreduce(eng_->bufs().flags + kFlagMoved, 3, kDtI32, kOpMax);
    be_.sync();
    int flags[kNumFlags];
    be_.d2h(flags, eng_->bufs().flags, sizeof(flags));
    eng_->check_overflow_public(flags);
    return flags[kFlagMoved];
  }

  // eight raw sums over the owned atoms -> all-reduce -> thermo_dev_
  void thermo_global()
  {
    Engine& e = *eng_;
    const Bufs& b = e.bufs();
    be_.thermo(kSlotThermo, e.num_atoms(), volume_, b.mi, b.fo, b.vi, b.fo + (int64_t)kOutW * e.num_atoms(), sums_dev_,
               e.thermo_scratch(), b.lvl, 1, 0, (reverse_ && !virial_folded_) ? 1 : 2);
    device_allreduce(sums_dev_, 8, kDtF64, kOpSum);
    be_.template launch<64>(kSlotMisc, 1, ThermoNormBody{sums_dev_, (double)n_total_, volume_, thermo_dev_});
  }

  // ---- per-step forward communication of the ghost positions: one pack launch, one grouped exchange, one unpack launch ----
  void halo_exchange() { halo_exchange_on(be_); }
  void halo_exchange_on(B& on)
  {
    Engine& e = *eng_;
    Plan& h = plan_;
    if (h.n_send > 0)
      on.template launch<256>(kSlotMisc, h.n_send, HaloPackPeersBody{e.bufs(), h.send_int, h.send_peer, h.pt, h.sendbuf});
    peer_exchange(on, 3, false);
    if (h.n_recv > 0)
      on.template launch<256>(kSlotMisc, h.n_recv, HaloUnpackPeersBody{e.box(), e.bufs(), h.recv_int, h.recvbuf});
  }
  // the grouped exchange of the plan: forward = send entries out, ghosts in; backward = the ghosts' planes out, the entries'
  // in.  `w` doubles per entry.  One message per distinct peer rank each way (Plan::links).
  void peer_exchange(B& on, int w, bool backward)
  {
    Plan& h = plan_;
    TransportMsg sm[kMaxPeers], rm[kMaxPeers];
    int ns = 0, nr = 0;
    const int64_t wb = (int64_t)sizeof(double) * w;
    for (const Link& k : h.links) {
      const int64_t cs = backward ? k.cnt_recv : k.cnt_send, cr = backward ? k.cnt_send : k.cnt_recv;
      double* sb = backward ? h.recvbuf + (int64_t)w * k.recv_off : h.sendbuf + (int64_t)w * k.send_off;
      double* rb = backward ? h.sendbuf + (int64_t)w * k.send_off : h.recvbuf + (int64_t)w * k.recv_off;
      if (cs)
        sm[ns++] = TransportMsg{sb, wb * cs, k.rank};
      if (cr)
        rm[nr++] = TransportMsg{rb, wb * cr, k.rank};
    }
    exchange_on(on, ns, sm, nr, rm);
  }

  // ---- reverse mode: per-step reverse communication of what the force assembly left on the ghosts ----
  // The planes of every ghost go back to the rank that owns the atom (one grouped exchange); there one work-item per shell
  // atom adds what its images collected, in ascending peer order.
  void reverse_exchange(int first, int planes, const int* frz)
  {
    reverse_send(be_, first, planes, frz);
    reverse_add(first, planes, frz);
  }
  void reverse_send(B& on, int first, int planes, const int* frz)
  {
    Engine& e = *eng_;
    Plan& h = plan_;
    if (h.peers.empty())
      return;
    if (h.n_recv > 0)
      on.template launch<256>(kSlotMisc, h.n_recv, GhostPackPeersBody{e.bufs(), h.recv_int, first, planes, h.recvbuf, frz});
    peer_exchange(on, planes, true);
  }
  void reverse_add(int first, int planes, const int* frz)
  {
    Engine& e = *eng_;
    Plan& h = plan_;
    if (h.peers.empty() || h.n_src == 0)
      return;
    be_.template launch<256>(kSlotMisc, h.n_src,
                             GhostAddPeersBody{e.bufs(), h.src_int, h.src_start, h.src_entry, first, planes, h.sendbuf, frz});
  }
  void force_reverse(const int* frz = nullptr) // frz: as the force kernels of this step got it
  {
    virial_folded_ = false;
    if (reverse_)
      reverse_exchange(kOutF, 3, frz);
  }
  // per-atom virials for output: the halves computed on other ranks' ghosts come home once per force evaluation (the run
  // loop only needs the global sum, which thermo_global takes over owned atoms AND ghosts until then)
  void fold_virial()
  {
    if (!reverse_ || virial_folded_ || !have_force_)
      return;
    reverse_exchange(kOutW, 9, nullptr);
    virial_folded_ = true;
  }

  // shell width, local box and origin for the chosen ghost mode (constructor; set_ghost_mode before setup)
  void configure_shell()
  {
    DomainGeom& g = geom_;
    const double rc = model_.rc_radial_max;
    double owned_frac = 1.0;
    bool forward_fits = true;
    for (int d = 0; d < 3; ++d)
      if (g.decomposed[d]) {
        const double w = (2.0 * rc + 2.0 * kSkin) / thick_[d];
        owned_frac *= (1.0 / g.grid[d]) / (1.0 / g.grid[d] + 2.0 * w);
        forward_fits = forward_fits && w <= 1.0 / g.grid[d];
      }
    reverse_ = ghost_mode_ == 1 || (ghost_mode_ < 0 && model_.kind == 0 && (owned_frac < 0.7 || !forward_fits));
    if (reverse_ && model_.kind != 0)
      throw EngineError{-4, "reverse-mode ghosts: NEP models only"};
    double ext[3] = {1, 1, 1}, org[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) {
      g.ifrac[d] = (rc + kSkin) / thick_[d];
      g.wfrac[d] = reverse_ ? g.ifrac[d] : 2.0 * g.ifrac[d];
      if (g.decomposed[d]) {
        if (g.wfrac[d] > 1.0 / g.grid[d])
          throw EngineError{-3, reverse_ ? "domain thinner than the ghost shell rc + skin in a decomposed direction"
                                         : "domain thinner than the ghost shell 2 (rc + skin) in a decomposed direction"};
        ext[d] = (g.hi[d] - g.lo[d]) + 2.0 * g.wfrac[d];
        org[d] = g.lo[d] - g.wfrac[d];
      }
    }
    // local box handed to the engine: the ghost-padded sub-box, open in the decomposed directions
    for (int c = 0; c < 3; ++c) {
      g.origin[c] = 0.0;
      for (int d = 0; d < 3; ++d) {
        h_loc_[3 * c + d] = g.H[3 * c + d] * ext[d];
        g.origin[c] += g.H[3 * c + d] * org[d];
      }
    }
  }

  // ---- migration + ghost construction + list rebuild ----
  void decompose()
  {
    const auto t_begin = std::chrono::steady_clock::now();
    decompose_body();
    be_.sync();
    decompose_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }

  // NEPMI_DIST_TRACE=1: wall time of the phases of every (re-)decomposition on stderr (synchronised: a diagnostic, not the product path)
  void decompose_body()
  {
    const bool trace = std::getenv("NEPMI_DIST_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    std::string tline;
    auto mark = [&](const char* what) {
      if (!trace)
        return;
      be_.sync();
      const auto now = std::chrono::steady_clock::now();
      char buf[64];
      std::snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      tline += buf;
      t_last = now;
    };
    const int me = tr_.rank, P = tr_.nranks;
    if (resident_) {
      // owned state back to local order (positions in local coordinates, velocities)
      Engine& e = *eng_;
      e.resident_export(cur_.x, cur_.v, nullptr, nullptr, nullptr, 1);
      be_.sync();
    }
    const int64_t n_old = cur_.n, n_own = cur_.n_own;
    // 1. owners
    int* dest = iscratch(0, n_own + 1);
    int* stay = iscratch(1, n_own + 1);
    int* scan = iscratch(2, n_old + 2);
    int* sidx = iscratch(3, n_old + 2);
    int* sscr = iscratch(4, n_old / 512 + 2048);
    int64_t n_stay = 0;
    std::vector<int> h_dest, h_leave;
    if (n_own > 0) {
      be_.template launch<256>(kSlotMisc, n_own, OwnerBody{geom_, n_old, n_own, me, cur_.x, dest, stay});
      n_stay = compact(stay, n_own, scan, sidx, sscr);
    }
    const int64_t n_leave = n_own - n_stay;
    int* lidx = iscratch(5, n_leave + 1);
    if (n_leave > 0) {
      int* leave = iscratch(6, n_own + 1);
      be_.template launch<256>(kSlotMisc, n_own, InvertFlagBody{stay, leave});
      compact(leave, n_own, scan, lidx, sscr);
      h_leave.resize((size_t)n_leave);
      be_.d2h(h_leave.data(), lidx, sizeof(int) * n_leave);
      // destinations of the leaving atoms only (the whole array is 4 bytes per owned atom through pageable memory)
      int* dleave = iscratch(12, n_leave + 1);
      be_.template launch<256>(kSlotMisc, n_leave, MapIndexBody{dest, lidx, 0, dleave});
      h_dest.resize((size_t)n_leave);
      be_.d2h(h_dest.data(), dleave, sizeof(int) * n_leave);
    }
    mark("owners");
    // 2. who sends how many to whom (all ranks learn the whole matrix)
    std::vector<int64_t> mat((size_t)P * P, 0);
    std::vector<std::vector<int>> by_dest(P);
    for (int64_t q = 0; q < n_leave; ++q)
      by_dest[h_dest[q]].push_back(h_leave[q]);
    for (int r = 0; r < P; ++r)
      mat[(size_t)me * P + r] = (int64_t)by_dest[r].size();
    host_allreduce(mat.data(), (int64_t)P * P, kDtI64, kOpSum);
    int64_t n_arrive = 0;
    for (int r = 0; r < P; ++r)
      n_arrive += mat[(size_t)r * P + me];
    // 3. migration payloads
    std::vector<double*> sbuf, rbuf;
    std::vector<TransportMsg> sends, recvs;
    std::vector<int64_t> rcount;
    for (int r = 0; r < P; ++r) {
      const int64_t c = (int64_t)by_dest[r].size();
      if (r != me && c > 0) {
        int* di = iscratch(7, c + 1);
        be_.h2d(di, by_dest[r].data(), sizeof(int) * c);
        double* buf = (double*)palloc(sizeof(double) * 9 * c);
        be_.template launch<256>(kSlotMisc, c, PackStateBody{geom_, n_old, c, di, cur_.x, cur_.v, cur_.m, cur_.t, cur_.id, buf});
        be_.sync();
        sbuf.push_back(buf);
        sends.push_back(TransportMsg{buf, (int64_t)sizeof(double) * 9 * c, r});
      }
      const int64_t a = mat[(size_t)r * P + me];
      if (r != me && a > 0) {
        double* buf = (double*)palloc(sizeof(double) * 9 * a);
        rbuf.push_back(buf);
        rcount.push_back(a);
        recvs.push_back(TransportMsg{buf, (int64_t)sizeof(double) * 9 * a, r});
      }
    }
    exchange((int)sends.size(), sends.data(), (int)recvs.size(), recvs.data());
    // the Langevin generator states of the migrating atoms: a second message per (source, destination) in the same order
    const size_t sb = lan_on_ ? be_.lan_state_bytes() : 0;
    const int rw = (int)(sb / 8);
    std::vector<char*> rs_buf, rr_buf;
    if (lan_on_) {
      std::vector<TransportMsg> rsends, rrecvs;
      for (int r = 0; r < P; ++r) {
        const int64_t c = (int64_t)by_dest[r].size();
        if (r != me && c > 0) {
          int* di = iscratch(7, c + 1);
          be_.h2d(di, by_dest[r].data(), sizeof(int) * c);
          char* buf = (char*)palloc(sb * (size_t)c);
          be_.template launch<256>(kSlotMisc, c, GatherRecordsBody{(const unsigned long long*)cur_.rng, di, rw, (unsigned long long*)buf});
          be_.sync();
          rs_buf.push_back(buf);
          rsends.push_back(TransportMsg{buf, (int64_t)(sb * (size_t)c), r});
        }
        const int64_t a = mat[(size_t)r * P + me];
        if (r != me && a > 0) {
          char* buf = (char*)palloc(sb * (size_t)a);
          rr_buf.push_back(buf);
          rrecvs.push_back(TransportMsg{buf, (int64_t)(sb * (size_t)a), r});
        }
      }
      exchange((int)rsends.size(), rsends.data(), (int)rrecvs.size(), rrecvs.data());
    }
    mark("migrate");
    // 4. new owned set: stayers in their order, then the arrivals by source rank
    const int64_t n_own_new = n_stay + n_arrive;
    // capacity of the local system: the ghost shell's share of the padded sub-box at uniform density, with 30 % headroom
    double grow = 1.3;
    for (int d = 0; d < 3; ++d)
      if (geom_.decomposed[d])
        grow *= 1.0 + 2.0 * geom_.wfrac[d] * geom_.grid[d];
    const int64_t cap_need = (int64_t)((double)(n_own_new > 1024 ? n_own_new : 1024) * grow) + 4096;
    State& N = nxt_;
    if (N.cap < cap_need)
      alloc_state(N, cap_need);
    if (lan_on_ && !N.rng) // (allocated before the first Langevin run)
      N.rng = (char*)be_.alloc(be_.lan_state_bytes() * (size_t)N.cap);
    // the gather uses stride n_new, which is only known after the ghost stages: build owned arrays with a
    // provisional stride = N.cap and repack at the end
    const int64_t S = N.cap;
    if (n_stay > 0)
      be_.template launch<256>(kSlotMisc, n_stay,
                               GatherStateBody{n_old, S, n_stay, sidx, cur_.x, cur_.v, cur_.m, cur_.t, cur_.id, N.x, N.v, N.m,
                                               N.t, N.id});
    int64_t off = n_stay;
    for (size_t k = 0; k < rbuf.size(); ++k) {
      be_.template launch<256>(kSlotMisc, rcount[k], UnpackStateBody{geom_, S, off, rcount[k], rbuf[k], N.x, N.v, N.m, N.t, N.id});
      off += rcount[k];
    }
    if (lan_on_) { // stayers in their order, then the arrivals by source rank: the order of the other per-atom arrays
      if (n_stay > 0)
        be_.template launch<256>(kSlotMisc, n_stay,
                                 GatherRecordsBody{(const unsigned long long*)cur_.rng, sidx, rw, (unsigned long long*)N.rng});
      int64_t o2 = n_stay;
      for (size_t k = 0; k < rr_buf.size(); ++k) {
        copy_dev(N.rng + sb * (size_t)o2, rr_buf[k], sb * (size_t)rcount[k]);
        o2 += rcount[k];
      }
    }
    be_.sync();
    for (double* p : sbuf) pfree(p);
    for (double* p : rbuf) pfree(p);
    for (char* p : rs_buf) pfree(p);
    for (char* p : rr_buf) pfree(p);
    mark("gather");
    // 5. ghosts: the owned atoms in the shell next to a face go to the neighbour through it -- faces, edges and corners of the
    //    process grid as separate direct messages (dist_bodies.h: direct halo)
    free_plan();
    Plan& h = plan_;
    {
      const DomainGeom& g = geom_;
      h.pt.n = 0;
      struct Cand {
        int rank, o[3];
      };
      std::vector<Cand> cands;
      for (int oz = -1; oz <= 1; ++oz)
        for (int oy = -1; oy <= 1; ++oy)
          for (int ox = -1; ox <= 1; ++ox) {
            const int o[3] = {ox, oy, oz};
            if (ox == 0 && oy == 0 && oz == 0)
              continue;
            bool ok = true;
            int c[3];
            for (int d = 0; d < 3; ++d) {
              c[d] = g.coords[d] + o[d];
              if (o[d] != 0 && !g.decomposed[d])
                ok = false;
              if (c[d] < 0 || c[d] >= g.grid[d]) {
                if (!g.pbc[d])
                  ok = false;
                c[d] = (c[d] + g.grid[d]) % g.grid[d];
              }
            }
            if (!ok)
              continue;
            Cand cd;
            cd.rank = c[0] + g.grid[0] * (c[1] + g.grid[1] * c[2]);
            for (int d = 0; d < 3; ++d)
              cd.o[d] = o[d];
            cands.push_back(cd);
          }
      // by rank, the enumeration order (ascending offsets) within a rank
      std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b2) { return a.rank < b2.rank; });
      for (const Cand& cd : cands) {
        const int i = h.pt.n++;
        Peer pe;
        pe.rank = cd.rank;
        h.peers.push_back(pe);
        for (int d = 0; d < 3; ++d)
          h.pt.off[i][d] = cd.o[d];
        // receiver-local coordinates: + (sender origin - receiver origin); across a periodic face the lattice-vector
        // image and the jump of the origin cancel, so the shift is the same for edge ranks
        for (int cc = 0; cc < 3; ++cc) {
          h.pt.shift[i][cc] = 0.0;
          for (int d = 0; d < 3; ++d)
            h.pt.shift[i][cc] -= g.H[3 * cc + d] * (double)cd.o[d] / g.grid[d];
        }
      }
    }
    const int NP = (int)h.peers.size();
    int64_t n_loc = n_own_new;
    if (NP > 0) {
      // shell atoms L (stable compaction of the owned atoms with a non-empty peer mask), then ONE scan numbers all entries
      unsigned* mask = (unsigned*)iscratch(13, n_own_new + 1);
      int* any = iscratch(1, n_own_new + 1);
      int* gscan = iscratch(2, n_own_new + 2);
      int* L = iscratch(8, n_own_new + 1);
      int* gscr = iscratch(4, (int64_t)(NP + 1) * (n_own_new + 1) / 512 + 2048);
      int64_t m = 0;
      if (n_own_new > 0) {
        be_.template launch<256>(kSlotMisc, n_own_new, PeerMaskBody{geom_, h.pt, S, N.x, mask, any});
        m = compact(any, n_own_new, gscan, L, gscr);
      }
      const int64_t nflag = (int64_t)NP * (m + 1) + 1;
      int* pscan = iscratch(14, nflag + 1);
      be_.template launch<256>(kSlotMisc, nflag, PeerFlagBody{mask, L, m, NP, pscan});
      int* pflag = nullptr; // (the fill reads the mask again: the flags are not kept)
      (void)pflag;
      be_.exclusive_scan(pscan, nflag, gscr);
      int* offs = iscratch(15, NP + 2);
      be_.template launch<64>(kSlotMisc, NP + 1, PeerOffsetsBody{pscan, m, NP, offs});
      std::vector<int> hoff((size_t)NP + 1, 0);
      be_.sync();
      be_.d2h(hoff.data(), offs, sizeof(int) * (NP + 1));
      h.n_send = hoff[NP];
      for (int i = 0; i < NP; ++i) {
        h.peers[i].send_off = hoff[i];
        h.peers[i].cnt_send = hoff[i + 1] - hoff[i];
      }
      h.send_idx = (int*)palloc(sizeof(int) * (h.n_send + 1));
      h.send_int = (int*)palloc(sizeof(int) * (h.n_send + 1));
      h.send_peer = (unsigned char*)palloc((size_t)h.n_send + 8);
      if (h.n_send > 0)
        be_.template launch<256>(kSlotMisc, nflag - 1, PeerFillBody{mask, L, m, NP, pscan, h.send_idx, h.send_peer});
      // the reverse path's table: every shell atom's entries in ascending peer order
      h.n_src = m;
      h.src_idx = (int*)palloc(sizeof(int) * (m + 1));
      h.src_int = (int*)palloc(sizeof(int) * (m + 1));
      h.src_start = (int*)palloc(sizeof(int) * (m + 2));
      h.src_entry = (int*)palloc(sizeof(int) * (h.n_send + 1));
      if (m > 0) {
        copy_dev(h.src_idx, L, sizeof(int) * m);
        be_.template launch<256>(kSlotMisc, m + 1, PeerCountBody{mask, L, m, h.src_start});
        be_.exclusive_scan(h.src_start, m + 1, gscr);
        be_.template launch<256>(kSlotMisc, m, PeerCsrBody{mask, L, m, NP, pscan, h.src_start, h.src_entry});
      }
      // counts: one 8-byte message per peer; between a pair of ranks the sender's ascending offsets are the receiver's
      // descending ones (Plan::peers), so the receives of a rank are posted in descending order of our peer index
      {
        std::vector<int64_t> cs((size_t)NP), cr((size_t)NP, 0);
        for (int i = 0; i < NP; ++i)
          cs[i] = h.peers[i].cnt_send;
        int64_t* d = (int64_t*)palloc(sizeof(int64_t) * 2 * NP);
        be_.h2d(d, cs.data(), sizeof(int64_t) * NP);
        TransportMsg sm[kMaxPeers], rm[kMaxPeers];
        std::vector<int> rorder; // our peers in the order their blocks arrive: rank ascending, offsets descending within a rank
        for (int i = 0; i < NP;) {
          int j = i;
          while (j < NP && h.peers[j].rank == h.peers[i].rank)
            ++j;
          for (int q = j - 1; q >= i; --q)
            rorder.push_back(q);
          i = j;
        }
        for (int i = 0; i < NP; ++i)
          sm[i] = TransportMsg{d + i, 8, h.peers[i].rank};
        for (int i = 0; i < NP; ++i)
          rm[i] = TransportMsg{d + NP + rorder[i], 8, h.peers[rorder[i]].rank};
        exchange(NP, sm, NP, rm);
        be_.sync();
        be_.d2h(cr.data(), d + NP, sizeof(int64_t) * NP);
        pfree(d);
        h.n_recv = 0;
        for (int i = 0; i < NP; ++i) { // receive blocks in arrival order
          Peer& pe = h.peers[rorder[i]];
          pe.cnt_recv = cr[rorder[i]];
          pe.recv_off = h.n_recv;
          h.n_recv += pe.cnt_recv;
        }
        h.links.clear();
        for (int i = 0; i < NP;) {
          int j = i;
          Link k;
          k.rank = h.peers[i].rank;
          k.send_off = h.peers[i].send_off;
          k.recv_off = h.peers[i].recv_off;
          while (j < NP && h.peers[j].rank == k.rank) {
            k.cnt_send += h.peers[j].cnt_send;
            k.cnt_recv += h.peers[j].cnt_recv;
            k.recv_off = h.peers[j].recv_off < k.recv_off ? h.peers[j].recv_off : k.recv_off;
            ++j;
          }
          h.links.push_back(k);
          i = j;
        }
      }
      if (n_loc + h.n_recv > N.cap)
        throw EngineError{-6, "domain decomposition: more ghost atoms than the density-based capacity of the local system "
                              "(strongly non-uniform density across the sub-boxes)"};
      const int bw = reverse_ ? 9 : 4; // doubles per entry: [x y z type] here; reverse mode: up to nine virial planes
      h.sendbuf = (double*)palloc(sizeof(double) * bw * (h.n_send + 1));
      h.recvbuf = (double*)palloc(sizeof(double) * bw * (h.n_recv + 1));
      h.recv_int = (int*)palloc(sizeof(int) * (h.n_recv + 1));
      if (h.n_send > 0)
        be_.template launch<256>(kSlotMisc, h.n_send,
                                 PackGhostPeersBody{S, h.n_send, h.send_idx, h.send_peer, N.x, N.t, h.pt, h.sendbuf});
      be_.sync();
      peer_exchange(be_, 4, false);
      if (h.n_recv > 0)
        be_.template launch<256>(kSlotMisc, h.n_recv, UnpackGhostPeersBody{S, n_loc, h.n_recv, h.recvbuf, N.x, N.t});
      n_loc += h.n_recv;
      be_.sync();
    }
    mark("ghosts");
    // 6. repack to stride n_loc, levels
    State& C = cur_;
    if (C.cap < n_loc + 1)
      alloc_state(C, n_loc + n_loc / 4 + 1024);
    repack(N, S, C, n_loc, n_own_new);
    C.n = n_loc;
    C.n_own = n_own_new;
    if (n_loc > 0)
      be_.template launch<256>(kSlotMisc, n_loc, LevelBody{geom_, n_loc, n_own_new, C.x, C.lvl, reverse_ ? 1 : 0});
    mark("repack");
    // 7. engine: lists on the local system, internal index lists of the halo, integrator state
    if (!eng_ || n_loc > eng_cap_) {
      eng_cap_ = n_loc + n_loc / 7 + 1024;
      std::unique_ptr<Engine> grown(new Engine(model_, eng_cap_, be_));
      grown->set_external_skin(true); // the global vote is the skin policy
      grown->set_loop_context(true);  // every force evaluation here feeds the integrator and the global sums: the force
                                      // assembly may take its scatter form (per-atom virials: exact_virials at the gathers)
      grown->set_flagged_steps_stand(true); // (the scatter form's guard band: hand-over by vote, hard limit as an error)
      grown->set_reverse_ghosts(reverse_);
      grown->bdp_seed(seed_);
      if (eng_) // a grown local system: the switches, the temperature, the noise sequence and the counters move over
        grown->adopt_from(*eng_);
      eng_ = std::move(grown);
    }
    Engine& e = *eng_;
    e.invalidate();
    {
      // a list-capacity error of ONE rank's rebuild must end the run on every rank
      int64_t failed = 0;
      std::string what;
      try {
        e.prepare_lists(h_loc_, pbc_loc_, n_loc, C.t, C.x, C.lvl);
      } catch (const EngineError& ex) {
        failed = 1;
        what = ex.msg;
      }
      int64_t any = failed;
      host_allreduce(&any, 1, kDtI64, kOpMax);
      if (any)
        throw EngineError{-6, failed ? what : std::string("another rank exceeded a neighbour list capacity at the rebuild")};
    }
    mark("lists");
    e.resident_alloc();
    e.resident_import(C.v, C.m, nullptr, nullptr, nullptr);
    int* inv = iscratch(10, n_loc + 1);
    be_.template launch<256>(kSlotMisc, n_loc, InversePermBody{e.bufs().perm, inv});
    if (h.n_send > 0)
      be_.template launch<256>(kSlotMisc, h.n_send, MapIndexBody{inv, h.send_idx, 0, h.send_int});
    if (h.n_recv > 0)
      be_.template launch<256>(kSlotMisc, h.n_recv, MapIndexBody{inv, nullptr, n_own_new, h.recv_int});
    if (h.n_src > 0)
      be_.template launch<256>(kSlotMisc, h.n_src, MapIndexBody{inv, h.src_idx, 0, h.src_int});
    be_.sync();
    mark("maps");
    if (trace)
      std::fprintf(stderr, "[nepmi dist rank %d] decomposition %lld (ms):%s\n", me, (long long)num_decompositions, tline.c_str());
    resident_ = true;
    ++num_decompositions;
  }

  // stable compaction of the indices with flag set; returns their number
  int64_t compact(int* flag, int64_t n, int* scan, int* idx_out, int* scr)
  {
    copy_dev(scan, flag, sizeof(int) * n);
    be_.memset(scan + n, 0, sizeof(int)); // (stream-ordered: no host round trip for a zero)
    be_.exclusive_scan(scan, n + 1, scr);
    be_.template launch<256>(kSlotMisc, n, CompactBody{flag, scan, idx_out});
    int total = 0;
    be_.sync();
    be_.d2h(&total, scan + n, sizeof(int));
    return total;
  }

  // [3][S] staging arrays -> [3][n] arrays of the current state
  void repack(const State& from, int64_t S, State& to, int64_t n, int64_t n_own)
  {
    for (int d = 0; d < 3; ++d) {
      if (n)
        copy_dev(to.x + d * n, from.x + d * S, sizeof(double) * n);
      if (n_own)
        copy_dev(to.v + d * n, from.v + d * S, sizeof(double) * n_own);
    }
    if (n_own) {
      copy_dev(to.m, from.m, sizeof(double) * n_own);
      copy_dev(to.id, from.id, sizeof(int64_t) * n_own);
      if (lan_on_)
        copy_dev(to.rng, from.rng, be_.lan_state_bytes() * (size_t)n_own);
    }
    if (n)
      copy_dev(to.t, from.t, sizeof(int) * n);
  }

  NepModel model_;
  TransportView tr_;
  B be_;
  DomainGeom geom_;
  double hg_[9], h_loc_[9], volume_ = 0.0;
  int pbc_loc_[3];
  int64_t n_total_ = 0;
  State cur_, nxt_;
  Plan plan_;
  std::unique_ptr<Engine> eng_;
  int64_t eng_cap_ = 0;
  bool resident_ = false, have_force_ = false;
  bool nhc_fresh_ = true;
  bool overlap_ = false; // interior bricks' radial pass while the ghost positions travel (set_overlap; off: see nepmi.h)
  int ghost_mode_ = -1;  // set_ghost_mode
  bool reverse_ = false, virial_folded_ = false;
  double thick_[3] = {0, 0, 0};
  B side_;               // the backend on the communication stream (device transports)
  bool side_ready_ = false;
  uint64_t seed_ = 12345678u;
  bool lan_on_ = false;              // a Langevin ensemble has run: State::rng is part of the per-atom state
  double* lan_sums_ = nullptr;
  int lan_seed_ = 12345678;
  bool lan_fresh_ = true;
  int* flag_dev_ = nullptr;
  double *sums_dev_ = nullptr, *thermo_dev_ = nullptr, *nhc_dev_ = nullptr, *factor_dev_ = nullptr;
  std::vector<int*> iscr_;
  std::vector<int64_t> iscr_cap_;
  std::vector<void*> scratch_;
};

} // namespace nepmi
