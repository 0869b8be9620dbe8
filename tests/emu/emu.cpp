// TEST INFRASTRUCTURE ONLY -- never part of, linked into, or loaded by the product.
//
// Kernel-logic emulator: the per-atom kernel bodies of gpumd_amd/csrc/nep_bodies.h and the engine
// sequencing of engine_impl.h, compiled by g++ and driven by plain host loops, exported through
// the same C ABI (include/nepmi.h) as tests/emu/libnepmi_emu.so.  Its only purpose is to let the
// CPU test tier (`pytest -m "not gpu"`, this container has no GPU) check the kernel *logic*
// against the oracle before the code ever reaches an MI355X.  "Device pointers" are host pointers.
// The product library (gpumd_amd/lib/libnepmi.so) contains none of this and the gpumd_amd package
// refuses to run without a gfx950 device.
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <vector>

#include "../../gpumd_amd/csrc/nep_bodies.h"
#include "../../gpumd_amd/csrc/nep_md.h"
#include "../../gpumd_amd/csrc/nep_window.h"

namespace nepmi {

struct HostLoopBackend {
  // zero-filled by default; NEPMI_POISON=<byte> fills fresh blocks with that byte instead (as the HIP backend does on request), so
  // that body logic which reads what nobody wrote shows up on the CPU tier as well
  void* alloc(size_t bytes)
  {
    static const int poison = std::getenv("NEPMI_POISON") ? std::atoi(std::getenv("NEPMI_POISON")) : -1;
    if (poison < 0)
      return std::calloc(1, bytes ? bytes : 1);
    void* p = std::malloc(bytes ? bytes : 1);
    std::memset(p, poison & 0xFF, bytes ? bytes : 1);
    return p;
  }
  void free(void* p) { std::free(p); }
  void memset(void* p, int v, size_t bytes) { std::memset(p, v, bytes); }
  void h2d(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
  void d2h(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
  void d2d(void* dst, const void* src, size_t bytes) { std::memmove(dst, src, bytes); }
  void* stream_handle() { return nullptr; }
  HostLoopBackend make_side_stream() { return *this; }
  void fork_to(HostLoopBackend&) {}
  void join_from(HostLoopBackend&) {}
  void sync() {}
  void begin_region(int) {}
  void end_region(int) {}
  double region_ms(int) { return 0.0; }
  double slot_ms(int) { return 0.0; }
  double slot_sum(int) { return 0.0; }
  int64_t slot_count(int) { return 0; }
  double region_sum(int) { return 0.0; }
  int64_t region_count(int) { return 0; }
  void set_timing(int) {}

  const int* frozen = nullptr; // see HipBackend::frozen
  bool is_frozen() const { return frozen && *frozen != 0; }
  template <int BLOCK, class Body>
  void launch(int, int64_t n, const Body& body)
  {
    if (is_frozen())
      return;
    for (int64_t i = 0; i < n; ++i)
      body(i);
  }
  // flag snapshots of the speculative run loops: the host loops have already finished when they are taken
  int snapshots[8][8];
  void poll_record(int ring, const int* flags) { std::memcpy(snapshots[ring], flags, sizeof(int) * 8); }
  void poll_wait(int ring, int* out8) { std::memcpy(out8, snapshots[ring], sizeof(int) * 8); }

  // the product's MFMA ANN kernel is device-only; the emulator runs the per-atom body
  template <class S>
  void launch_ann(int slot, int64_t n, const ModelD& m, const Bufs& b, bool)
  {
    launch<64>(slot, n, AnnBody<S>{m, b});
  }
  void set_mfma(bool) {}
  void adopt_options(const HostLoopBackend&) {}
  void probe_start() {}
  double probe_stop_ms() { return 0.0; } // no clock on the CPU tier: the engine keeps its default variant
  void ann_prepare(const ModelD&, const Bufs&) {}

  // one "workgroup" per brick, phases run back to back (the LDS-window kernels of nep_window.h)
  static constexpr bool kSplitLanes = false; // host loops run one lane per atom
  template <class Body>
  void launch_win_split(int, int64_t, const Body&)
  {
    std::abort(); // never selected: the engine asks kSplitLanes first
  }
  template <class Body>
  void launch_win(int, int64_t nbricks, const Body& body)
  {
    if (body.skip())
      return;
    std::vector<double> raw((size_t)body.lds_bytes() / 8 + 8);
    char* lds = reinterpret_cast<char*>(raw.data());
    for (int64_t wg = 0; wg < nbricks; ++wg) {
      const int64_t brick = body.map_brick(wg);
      body.stage_cells(brick, lds, 0, 1);
      int* woff = reinterpret_cast<int*>(lds);
      int run = 0;
      for (int c = 0; c < 512; ++c) {
        const int v = woff[c];
        woff[c] = run;
        run += v;
      }
      woff[512] = run;
      body.stage_copy(brick, lds, 0, 1);
      int64_t a0, a1;
      body.brick_range(brick, a0, a1);
      for (int64_t k = a0; k < a1; ++k)
        body.compute(brick, k, lds);
    }
  }

  // the same on the static window layout (Bufs::wtab): one staging pass, then the atoms of the brick
  template <class Body>
  void launch_win2(int, int64_t nbricks, const Body& body)
  {
    if (body.skip())
      return;
    std::vector<double> raw((size_t)body.lds_bytes() / 8 + 8);
    char* lds = reinterpret_cast<char*>(raw.data());
    for (int64_t wg = 0; wg < nbricks; ++wg) {
      const int64_t brick = body.map_brick(wg);
      body.stage(brick, lds, 0, 1);
      int64_t a0, a1;
      body.brick_range(brick, a0, a1);
      for (int64_t k = a0; k < a1; ++k)
        body.compute(brick, k, lds);
    }
  }

  // ... with the second staging phase of the bodies that keep the neighbours' table rows in "LDS"; the host loop runs one lane
  template <class Body>
  void launch_win2_split(int, int64_t nbricks, const Body& body)
  {
    if (body.skip())
      return;
    std::vector<double> raw((size_t)body.lds_bytes() / 8 + 8);
    char* lds = reinterpret_cast<char*>(raw.data());
    for (int64_t wg = 0; wg < nbricks; ++wg) {
      const int64_t brick = body.map_brick(wg);
      body.stage(brick, lds, 0, 1);
      body.stage_rows(brick, lds, 0, 1);
      int64_t a0, a1;
      body.brick_range(brick, a0, a1);
      for (int64_t k = a0; k < a1; ++k)
        body.compute(brick, k, lds, 0);
    }
  }
  static constexpr size_t kMaxLdsBytes = 160 * 1024;
  // the scatter form of the force assembly (gpumd_amd/csrc/nep_scatter.h) is device code only: never selected here
  static constexpr bool kHasScatter = false;
  static constexpr bool kHasFusedAngular = false; // gpumd_amd/csrc/nep_fused.h: device code only
  static constexpr bool kHasFusedWindow = false;
  static constexpr bool kHasBrickForce = false; // gpumd_amd/csrc/nep_brick.h: device code only
  template <class S>
  size_t brick_lds_bytes(const ModelD&, int) const { return 0; }
  template <class S>
  void launch_brick_force(int, int, int64_t, int64_t, const WinStage&, const ModelD&, int*, const unsigned*, int, bool, const float*, const int*)
  {
    std::abort();
  }
  template <class S>
  size_t fused_image_floats(const ModelD&) const { return 0; }
  template <class S>
  void launch_angular_fused_window(int, int64_t, const ModelD&, const Bufs&, int, float*, bool)
  {
  }
  template <class S>
  void launch_angular_fused(int, int64_t, const ModelD&, const Bufs&, int, float*, bool)
  {
    std::abort();
  }
  template <class S>
  void launch_force_scatter(int, int64_t, int, int64_t, const WinStage&, const ModelD&, int*, const unsigned*, int, bool, bool, int, int, const int*)
  {
    std::abort();
  }
  int build_fold_map(int64_t, const BoxD&, const Bufs&, int, int, unsigned*) { return 0; }

  // bodies with a workgroup-staged table: the "LDS" is an ordinary host buffer here
  template <int BLOCK, class Body>
  void launch_lds(int, int64_t n, const Body& body)
  {
    if (is_frozen())
      return;
    std::vector<float> lds((size_t)body.lds_floats() + 1);
    body.lds_stage(lds.data(), 0, 1);
    for (int64_t i = 0; i < n; ++i)
      body.run(i, (const float*)lds.data());
  }

  // P lanes per atom on the device; the host loop runs the one-lane form of the same body
  template <int BLOCK, int P, class Body>
  void launch_lds_parts(int, int64_t n, const Body& body)
  {
    if (is_frozen())
      return;
    std::vector<float> lds((size_t)body.lds_floats() + 1);
    for (int64_t i = 0; i < n; ++i)
      body.template run_parts<1>(i, 0, (const float*)lds.data());
  }

  // the device runs these bodies with two lanes per atom; the host loop runs the one-lane form
  template <int BLOCK, class Body>
  void launch_lds_pairs(int slot, int64_t n, const Body& body)
  {
    launch_lds<BLOCK>(slot, n, body);
  }

  void exclusive_scan(int* data, int64_t n, int*)
  {
    int run = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int v = data[i];
      data[i] = run;
      run += v;
    }
  }

  // Langevin thermostat, host stand-in: the same update with a different normal generator (std::mt19937_64 per atom
  // seeded from (seed, n)); the reference's XORWOW stream is pinned on the GPU tier only
  size_t lan_state_bytes() const { return sizeof(std::mt19937_64); }
  void lan_init(void* states, int64_t n, int seed)
  {
    std::mt19937_64* st = (std::mt19937_64*)states;
    for (int64_t i = 0; i < n; ++i)
      new (&st[i]) std::mt19937_64((uint64_t)seed * 1000003ull + (uint64_t)i);
  }
  // resident forms of the thermostat (HipBackend::lan_kick_resident & co.): the same update on the internal-order arrays
  void lan_kick_resident(void* states, int64_t n, double c1, double c2, const double* mi, double* vi, const int* perm,
                         const signed char* lvl, const int64_t* ids, const int* flags)
  {
    if (flags[kFlagMoved] != 0)
      return;
    std::mt19937_64* st = (std::mt19937_64*)states;
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int64_t k = 0; k < n; ++k) {
      if (lvl[k] < 2)
        continue;
      const int64_t s = ids ? ids[perm[k]] : (int64_t)perm[k];
      const double c2m = c2 * std::sqrt(1.0 / mi[k]);
      for (int d = 0; d < 3; ++d) {
        nd.reset(); // one fresh pair per draw: the stream of an atom does not depend on who draws it
        vi[d * n + k] = c1 * vi[d * n + k] + c2m * nd(st[s]);
      }
    }
  }
  void lan_init_ids(void* states, int64_t n, const int64_t* ids, int seed) // state q = the generator of global id ids[q]
  {
    std::mt19937_64* st = (std::mt19937_64*)states;
    for (int64_t q = 0; q < n; ++q)
      new (&st[q]) std::mt19937_64((uint64_t)seed * 1000003ull + (uint64_t)ids[q]);
  }
  void lan_momentum_fix_resident(int64_t n, const double* sums4, double* vi, const signed char* lvl, const int* flags)
  {
    if (flags[kFlagMoved] != 0)
      return;
    const double inverse_of_total_mass = 1.0 / sums4[3];
    for (int64_t k = 0; k < n; ++k)
      if (lvl[k] >= 2)
        for (int d = 0; d < 3; ++d)
          vi[d * n + k] -= sums4[d] * inverse_of_total_mass;
  }
  void lan_momentum_resident(int64_t n, const double* mi, const double* vi, const int* invp, const signed char* lvl,
                             double* sums4, const int* flags)
  {
    if (flags[kFlagMoved] != 0)
      return;
    double s[4] = {0, 0, 0, 0};
    for (int64_t q = 0; q < n; ++q) {
      const int64_t k = invp ? invp[q] : q;
      if (!invp && lvl[k] < 2)
        continue;
      for (int d = 0; d < 3; ++d)
        s[d] += mi[k] * vi[d * n + k];
      s[3] += mi[k];
    }
    for (int d = 0; d < 4; ++d)
      sums4[d] = s[d];
  }
  void lan_half(void* states, int64_t n, double c1, double c2, const double* mass, double* vel, double* sums4)
  {
    std::mt19937_64* st = (std::mt19937_64*)states;
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) {
      const double c2m = c2 * std::sqrt(1.0 / mass[i]);
      for (int d = 0; d < 3; ++d) {
        nd.reset(); // (as in the resident form: an atom's draws come from its own generator only)
        vel[d * n + i] = c1 * vel[d * n + i] + c2m * nd(st[i]);
      }
    }
    double s[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
      for (int d = 0; d < 3; ++d)
        s[d] += mass[i] * vel[d * n + i];
      s[3] += mass[i];
    }
    for (int d = 0; d < 4; ++d)
      sums4[d] = s[d];
    const double inverse_of_total_mass = 1.0 / s[3]; // (as the reference's gpu_correct_momentum and the resident form)
    for (int64_t i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d)
        vel[d * n + i] -= s[d] * inverse_of_total_mass;
  }

  void thermo(
    int, int64_t n, double volume, const double* mass, const double* pe, const double* vel,
    const double* virial, double* th, double*, const signed char* lvl = nullptr, int raw = 0, int64_t n_norm = 0,
    int virial_lvl = 2)
  {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
    for (int64_t i = 0; i < n; ++i) {
      if (lvl && lvl[i] < 2) {
        if (lvl[i] >= virial_lvl)
          for (int q = 0; q < 6; ++q)
            s[2 + q] += virial[(int64_t)q * n + i];
        continue;
      }
      const double m = mass[i];
      s[0] += (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]) * m;
      s[1] += pe[i];
      s[2] += virial[0 * n + i] + vx[i] * vx[i] * m;
      s[3] += virial[1 * n + i] + vy[i] * vy[i] * m;
      s[4] += virial[2 * n + i] + vz[i] * vz[i] * m;
      s[5] += virial[3 * n + i] + vx[i] * vy[i] * m;
      s[6] += virial[4 * n + i] + vx[i] * vz[i] * m;
      s[7] += virial[5 * n + i] + vy[i] * vz[i] * m;
    }
    if (raw) {
      for (int k = 0; k < 8; ++k)
        th[k] = s[k];
      return;
    }
    th[0] = s[0] / (3.0 * (double)(n_norm > 0 ? n_norm : n) * 8.617343e-5);
    th[1] = s[1];
    for (int k = 2; k < 8; ++k)
      th[k] = s[k] / volume;
  }
};

} // namespace nepmi

using NepmiBackend = nepmi::HostLoopBackend;
static NepmiBackend nepmi_make_backend(void*) { return NepmiBackend(); }

#include "../../gpumd_amd/csrc/dist_capi_impl.h"

// the emulator has no devices: the RCCL entry points exist (same C ABI) and refuse
extern "C" int nepmi_transport_rccl_id(char*) { return fail(NEPMI_ERR_HIP, "no RCCL in the kernel-logic emulator"); }
extern "C" int nepmi_transport_rccl(const char*, int, int, nepmi_transport*)
{
  return fail(NEPMI_ERR_HIP, "no RCCL in the kernel-logic emulator");
}
extern "C" int nepmi_transport_rccl_stats(const nepmi_transport*, int, int, nepmi_rccl_stats*)
{
  return fail(NEPMI_ERR_ARG, "not an RCCL transport");
}
