// Tersoff-1989 (BASELINE config 2) on the shared Verlet-list machinery, FP64 like the reference
// (src/force/tersoff1989.cu:157-586, src/force/potential.cu:35-134).
//
// The engine's list A (rc + skin, static reverse slots) is the Verlet list.  Two kernels per call:
//   TersoffPartialBody   per atom: local-list membership (FLOAT geometry, as
//                        gpu_find_local_neighbor_from_global, neighbor.cu:699-737), bond order b and
//                        b' (step 1, tersoff1989.cu:337-405), partial forces dU_i/dr_ij and U_i
//                        (step 2, :408-505); pair geometry (double + double MIC) is computed once
//                        and kept in a per-slot record.
//   TersoffAssembleBody  F_i = sum_j f12 - f21, virial, through the reverse slot (the reference
//                        searches j's list linearly, potential.cu:86-92); output planes in internal order.
#pragma once
#include "nep_bodies.h"
#include "nep_md.h"

namespace nepmi {

struct TersoffSetD {
  double a, b, lambda, mu, beta, n, c, d, h, r1, r2;
  double c2, d2, one_plus_c2overd2, pi_factor, minus_half_over_n;
};

struct TersoffParamsD {
  TersoffSetD p[3]; // type 0-0, type 1-1, mixed
  float rc_sq;      // float(rc * rc), the local-list test
};

struct alignas(16) D4 {
  double x, y, z;
  long long w; // 1 = member of the local list this step
};

// apply_mic (double), src/model/box.cuh:39-82
NEPMI_HD void mic_d(const BoxD& box, double& x, double& y, double& z)
{
  const double* h = box.h;
  if (box.ortho) {
    if (box.pbc[0]) { if (x < -h[0] * 0.5) x += h[0]; else if (x > h[0] * 0.5) x -= h[0]; }
    if (box.pbc[1]) { if (y < -h[4] * 0.5) y += h[4]; else if (y > h[4] * 0.5) y -= h[4]; }
    if (box.pbc[2]) { if (z < -h[8] * 0.5) z += h[8]; else if (z > h[8] * 0.5) z -= h[8]; }
  } else {
    double sx = h[9] * x + h[10] * y + h[11] * z;
    double sy = h[12] * x + h[13] * y + h[14] * z;
    double sz = h[15] * x + h[16] * y + h[17] * z;
    if (box.pbc[0]) sx -= nearbyint(sx);
    if (box.pbc[1]) sy -= nearbyint(sy);
    if (box.pbc[2]) sz -= nearbyint(sz);
    x = h[0] * sx + h[1] * sy + h[2] * sz;
    y = h[3] * sx + h[4] * sy + h[5] * sz;
    z = h[6] * sx + h[7] * sy + h[8] * sz;
  }
}

NEPMI_HD const TersoffSetD& ters_pair(const TersoffParamsD& t, int t1, int t2)
{
  return (t1 == 0 && t2 == 0) ? t.p[0] : ((t1 == 1 && t2 == 1) ? t.p[1] : t.p[2]);
}

NEPMI_HD void ters_fc(const TersoffSetD& p, double d, double& fc, double& fcp)
{
  if (d < p.r1) {
    fc = 1.0;
    fcp = 0.0;
  } else if (d < p.r2) {
    fc = cos(p.pi_factor * (d - p.r1)) * 0.5 + 0.5;
    fcp = -sin(p.pi_factor * (d - p.r1)) * p.pi_factor * 0.5;
  } else {
    fc = 0.0;
    fcp = 0.0;
  }
}

struct TersoffBufs {
  D4* rec;      // [MN_ang][N] pair geometry + membership
  double* bb;   // [MN_ang][N] bond order
  double* bp;   // [MN_ang][N] its derivative
  D4* f12;      // [MN_ang][N] partial forces
  double* pe_d; // [N]
  unsigned long long* mask; // [N] membership bits of the Verlet slots (lists of at most 64 entries), written by the partial kernel
};

// Members of the local list of an atom kept in LDS by the fast path of TersoffPartialBody: x, y, z, type, b, b', u as doubles,
// element e of member m of the a-th atom of the workgroup at [(m * 7 + e) * atoms_per_block + a]
constexpr int kTersoffLocalMax = 8;
constexpr int kTersoffBlock = 64;
constexpr int kTersoffLanes = 4; // lanes per atom on the device (the emulator's host loop runs one)
constexpr int kTersoffLdsDoubles = kTersoffLocalMax * 7 * kTersoffBlock;
// The several-lanes form (run_shared) keeps eleven numbers per member -- x, y, z, type, b, b', u and the member's own d, fc, fc', fA,
// which every bond of the atom needs and one lane evaluates -- plus a staging area where the lane that tested a Verlet entry leaves
// the record of a member it found (its position was in that lane's registers: no second gather):
//   member table  [(m * 11 + e) * A + a],   stage  [((lane * kTersoffLocalMax + q) * 4 + e) * A + a],   A = atoms per workgroup
constexpr int kTersoffMemberWords = 11;
constexpr int tersoff_shared_doubles(int lanes)
{
  return (kTersoffLocalMax * kTersoffMemberWords + lanes * kTersoffLocalMax * 4) * (kTersoffBlock / lanes);
}

#if defined(__HIP_DEVICE_COMPILE__)
// lanes of one wavefront hand data to each other through LDS: order the writes before the reads
#define NEPMI_WAVE_LDS_SYNC()                                  \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)
#else
#define NEPMI_WAVE_LDS_SYNC() \
  do {                        \
  } while (0)
#endif

struct TersoffPartialBody {
  BoxD box;
  TersoffParamsD tp;
  Bufs b;
  TersoffBufs tb;
  int lanes = 1; // lanes per atom of the launch (sizes the LDS: run_shared's layout for more than one)
  static constexpr int kMinWavesPerEu = 1;
  NEPMI_HD int lds_floats() const // (the emulator's host loop runs the one-lane form whatever `lanes` says: never less than that needs)
  {
    const int shared = lanes > 1 ? tersoff_shared_doubles(lanes) : 0;
    return 2 * (shared > kTersoffLdsDoubles ? shared : kTersoffLdsDoubles);
  }
  template <class LP>
  NEPMI_HD void lds_stage(LP, int, int) const {}
  template <class LP>
  NEPMI_HD void run(int64_t k, LP lds_f) const { run_parts<1>(k, 0, lds_f); }

  // Fast path.  13,824 silicon atoms are 54 atoms per CU: the kernel's run time is the latency of ONE wavefront, and the
  // plain form below walks an atom's bonded neighbours through global memory (record written, read back 2 (n - 1) times, bond
  // order written and read back: some forty dependent round trips per atom).  With at most kTersoffLocalMax members of the
  // local list (silicon: 4) their geometry and bond orders live in LDS, P adjacent lanes share the atom -- lane `part` tests
  // the Verlet entries part, part + P, ... and owns the bonds part, part + P, ... -- and every group of loads is requested
  // before its first use.  Same arithmetic per bond, the energy summed in bond order by one lane: bit-identical results.
  template <int P, class LP>
  NEPMI_HD void run_parts(int64_t k, int part, LP lds_f) const
  {
    if constexpr (P > 1) {
      run_shared<P>(k, part, lds_f);
      return;
    }
    const int64_t N = b.N;
    if (b.lvl[k] < 1)
      return;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    const TersoffSetD& s1 = tp.p[t1];
    const int nn = b.nn_ang[k];
    // pass A: membership of every Verlet entry (float geometry)
    unsigned long long inr = 0ull;
    constexpr int RB = 8 / P > 0 ? 8 / P : 1; // entries per lane and round: their index loads, then their gathers, together
    for (int s0 = 0; s0 < nn && s0 < 64; s0 += RB * P) {
      int jj[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int s = s0 + u * P + part;
        jj[u] = b.nl_ang[(int64_t)(s < nn ? s : nn - 1) * N + k];
      }
      PosQ pp[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u)
        pp[u] = b.posq[jj[u]];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int s = s0 + u * P + part;
        float xf, yf, zf;
        const float d2f = pair_geometry(box, p1, pp[u], xf, yf, zf);
        if (s < nn && s < 64 && d2f < tp.rc_sq)
          inr |= 1ull << s;
      }
    }
    if (P > 1) { // every lane of the atom gets all the bits
      unsigned lo = (unsigned)inr, hi = (unsigned)(inr >> 32);
#pragma unroll
      for (int msk = 1; msk < P; msk <<= 1) {
        lo |= (unsigned)NEPMI_SHFL_XOR((int)lo, msk);
        hi |= (unsigned)NEPMI_SHFL_XOR((int)hi, msk);
      }
      inr = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    const int cnt = __builtin_popcountll(inr);
    if (nn > 64 || cnt > kTersoffLocalMax) { // long lists: the plain form, one lane
      if (part == 0)
        (*this)(k);
      return;
    }
    constexpr int APB = kTersoffBlock / P;
    double* L = const_cast<double*>(reinterpret_cast<const double*>(&lds_f[0])) + (int)(k % APB);
    auto at = [&](int m, int e) -> double& { return L[(m * 7 + e) * APB]; };
    if (part == 0) {
      tb.mask[k] = inr;
      b.nn_rad[k] = cnt;
      b.nn_angstep[k] = cnt;
    }
    // Verlet slot of member m = position of the m-th set bit
    int ms[kTersoffLocalMax];
    {
      unsigned long long rest = inr;
#pragma unroll
      for (int m = 0; m < kTersoffLocalMax; ++m) {
        ms[m] = rest ? (int)__builtin_ctzll(rest) : 0;
        rest &= rest - 1ull;
      }
    }
    // pass B: the members' records (FP64 difference + FP64 minimum image); lane `part` takes members part, part + P, ...
    {
      constexpr int MP = (kTersoffLocalMax + P - 1) / P;
      PosQ mp[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u) {
        const int m = u * P + part;
        int slot = 0;
#pragma unroll
        for (int q = 0; q < kTersoffLocalMax; ++q)
          slot = q == m ? ms[q] : slot;
        if (m < cnt)
          mp[u] = b.posq[b.nl_ang[(int64_t)slot * N + k]];
      }
#pragma unroll
      for (int u = 0; u < MP; ++u) {
        const int m = u * P + part;
        if (m < cnt) {
          int slot = 0;
#pragma unroll
          for (int q = 0; q < kTersoffLocalMax; ++q)
            slot = q == m ? ms[q] : slot;
          D4 r;
          r.x = mp[u].x - p1.x;
          r.y = mp[u].y - p1.y;
          r.z = mp[u].z - p1.z;
          mic_d(box, r.x, r.y, r.z);
          r.w = 1 | ((long long)mp[u].type << 8);
          at(m, 0) = r.x;
          at(m, 1) = r.y;
          at(m, 2) = r.z;
          at(m, 3) = (double)mp[u].type;
          tb.rec[(int64_t)slot * N + k] = r;
        }
      }
      // (non-members: the force kernel and the list export read the membership bits, not their records)
    }
    NEPMI_WAVE_LDS_SYNC();
    // step 1: bond order of the bonds this lane owns
    for (int i1 = part; i1 < cnt; i1 += P) {
      const double x12 = at(i1, 0), y12 = at(i1, 1), z12 = at(i1, 2);
      const double d12 = sqrt(x12 * x12 + y12 * y12 + z12 * z12);
      double zeta = 0.0;
      for (int i2 = 0; i2 < cnt; ++i2) {
        if (i2 == i1)
          continue;
        const double x13 = at(i2, 0), y13 = at(i2, 1), z13 = at(i2, 2);
        const int t3 = (int)at(i2, 3);
        const double d13 = sqrt(x13 * x13 + y13 * y13 + z13 * z13);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) / (d12 * d13);
        double fc13, fcp13;
        ters_fc(ters_pair(tp, t1, t3), d13, fc13, fcp13);
        const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
        zeta += fc13 * (s1.one_plus_c2overd2 - s1.c2 / tmp);
      }
      const double bzn = pow(s1.beta * zeta, s1.n);
      const double b12 = pow(1.0 + bzn, s1.minus_half_over_n);
      if (zeta < 1.0e-16) { // avoid division by 0
        at(i1, 4) = 1.0;
        at(i1, 5) = 0.0;
      } else {
        at(i1, 4) = b12;
        at(i1, 5) = -b12 * bzn * 0.5 / ((1.0 + bzn) * zeta);
      }
    }
    NEPMI_WAVE_LDS_SYNC();
    // step 2: partial forces and energy of the bonds this lane owns
    for (int i1 = part; i1 < cnt; i1 += P) {
      int slot = 0;
#pragma unroll
      for (int q = 0; q < kTersoffLocalMax; ++q)
        slot = q == i1 ? ms[q] : slot;
      const double x12 = at(i1, 0), y12 = at(i1, 1), z12 = at(i1, 2);
      const int t2 = (int)at(i1, 3);
      const TersoffSetD& p12 = ters_pair(tp, t1, t2);
      const double d12 = sqrt(x12 * x12 + y12 * y12 + z12 * z12);
      const double d12inv = 1.0 / d12;
      double fc12, fcp12;
      ters_fc(p12, d12, fc12, fcp12);
      const double fa12 = p12.b * exp(-p12.mu * d12), fap12 = -p12.mu * fa12;
      const double fr12 = p12.a * exp(-p12.lambda * d12), frp12 = -p12.lambda * fr12;
      const double b12 = at(i1, 4), bp12 = at(i1, 5);
      const double factor3 = (fcp12 * (fr12 - b12 * fa12) + fc12 * (frp12 - b12 * fap12)) * d12inv;
      double fx = x12 * factor3 * 0.5, fy = y12 * factor3 * 0.5, fz = z12 * factor3 * 0.5;
      at(i1, 6) = fc12 * (fr12 - b12 * fa12) * 0.5;
      for (int i2 = 0; i2 < cnt; ++i2) {
        if (i2 == i1)
          continue;
        const double x13 = at(i2, 0), y13 = at(i2, 1), z13 = at(i2, 2);
        const int t3 = (int)at(i2, 3);
        const TersoffSetD& p13 = ters_pair(tp, t1, t3);
        const double d13 = sqrt(x13 * x13 + y13 * y13 + z13 * z13);
        double fc13, fcp13;
        ters_fc(p13, d13, fc13, fcp13);
        const double fa13 = p13.b * exp(-p13.mu * d13);
        const double bp13 = at(i2, 5);
        const double od = 1.0 / (d12 * d13);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) * od;
        const double c_over = c123 * d12inv * d12inv;
        const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
        const double g123 = s1.one_plus_c2overd2 - s1.c2 / tmp;
        const double gp123 = 2.0 * s1.c2 * (c123 - s1.h) / (tmp * tmp);
        const double ta = (-bp12 * fc12 * fa12 * fc13 - bp13 * fc13 * fa13 * fc12) * gp123;
        const double tbb = -bp13 * fc13 * fa13 * fcp12 * g123 * d12inv;
        fx += (x12 * tbb + ta * (x13 * od - x12 * c_over)) * 0.5;
        fy += (y12 * tbb + ta * (y13 * od - y12 * c_over)) * 0.5;
        fz += (z12 * tbb + ta * (z13 * od - z12 * c_over)) * 0.5;
      }
      D4 out;
      out.x = fx;
      out.y = fy;
      out.z = fz;
      out.w = 0;
      tb.f12[(int64_t)slot * N + k] = out;
    }
    NEPMI_WAVE_LDS_SYNC();
    if (part == 0) { // the bonds' energies in bond order: the sum of the one-lane form, bit for bit
      double u = 0.0;
      for (int m = 0; m < cnt; ++m)
        u += at(m, 6);
      tb.pe_d[k] = u;
    }
  }

  // Several lanes per atom (P > 1), two changes against the form above, same arithmetic per bond:
  //  * the lane that TESTS a Verlet entry has the neighbour's position in its registers; when the entry is a member it forms the FP64
  //    record at once and leaves it in its staging rows, and once all lanes know all membership bits it moves the record to row
  //    m = (number of members in front of it) of the member table -- the form above gathered the members' positions a second time
  //    (index -> position: two dependent round trips of a kernel that is one wavefront's latency long);
  //  * a member's distance, cutoff function and attractive term are needed by every bond of the atom (as the "13" terms of steps 1
  //    and 2): its owner lane evaluates them once (the same expressions on the same operands: the same bits) and the others read
  //    them -- three exponentials and three square roots per lane less (silicon: four bonds on four lanes).
  template <int P, class LP>
  NEPMI_HD void run_shared(int64_t k, int part, LP lds_f) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < 1)
      return;
    constexpr int APB = kTersoffBlock / P, E = kTersoffMemberWords;
    double* L = const_cast<double*>(reinterpret_cast<const double*>(&lds_f[0])) + (int)(k % APB);
    double* ST = L + kTersoffLocalMax * E * APB;
    auto at = [&](int m, int e) -> double& { return L[(m * E + e) * APB]; };
    auto st = [&](int q, int e) -> double& { return ST[((part * kTersoffLocalMax + q) * 4 + e) * APB]; };
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    const TersoffSetD& s1 = tp.p[t1];
    const int nn = b.nn_ang[k];
    // pass A: membership of every Verlet entry (float geometry); a member's record goes to the global rows and to the stage
    unsigned long long inr = 0ull;
    int nk = 0; // members this lane has found
    constexpr int RB = 8 / P > 0 ? 8 / P : 1;
    for (int s0 = 0; s0 < nn && s0 < 64; s0 += RB * P) {
      int jj[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int s = s0 + u * P + part;
        jj[u] = b.nl_ang[(int64_t)(s < nn ? s : nn - 1) * N + k];
      }
      PosQ pp[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u)
        pp[u] = b.posq[jj[u]];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int s = s0 + u * P + part;
        float xf, yf, zf;
        const float d2f = pair_geometry(box, p1, pp[u], xf, yf, zf);
        if (s < nn && s < 64 && d2f < tp.rc_sq) {
          inr |= 1ull << s;
          if (nk < kTersoffLocalMax) {
            D4 r;
            r.x = pp[u].x - p1.x;
            r.y = pp[u].y - p1.y;
            r.z = pp[u].z - p1.z;
            mic_d(box, r.x, r.y, r.z);
            r.w = 1 | ((long long)pp[u].type << 8);
            tb.rec[(int64_t)s * N + k] = r;
            st(nk, 0) = r.x;
            st(nk, 1) = r.y;
            st(nk, 2) = r.z;
            st(nk, 3) = (double)(pp[u].type + 256 * s);
          }
          ++nk;
        }
      }
    }
    {
      unsigned lo = (unsigned)inr, hi = (unsigned)(inr >> 32);
#pragma unroll
      for (int msk = 1; msk < P; msk <<= 1) {
        lo |= (unsigned)NEPMI_SHFL_XOR((int)lo, msk);
        hi |= (unsigned)NEPMI_SHFL_XOR((int)hi, msk);
      }
      inr = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    const int cnt = __builtin_popcountll(inr);
    if (nn > 64 || cnt > kTersoffLocalMax) { // long lists: the plain form, one lane (it writes every record again)
      if (part == 0)
        (*this)(k);
      return;
    }
    if (part == 0) {
      tb.mask[k] = inr;
      b.nn_rad[k] = cnt;
      b.nn_angstep[k] = cnt;
    }
    // the staged records to their rows of the member table (cnt <= kTersoffLocalMax: every lane's members were staged)
    for (int q = 0; q < nk; ++q) {
      const int code = (int)st(q, 3);
      const int s = code >> 8;
      const int m = __builtin_popcountll(inr & ((1ull << s) - 1ull));
      at(m, 0) = st(q, 0);
      at(m, 1) = st(q, 1);
      at(m, 2) = st(q, 2);
      at(m, 3) = (double)(code & 255);
    }
    NEPMI_WAVE_LDS_SYNC();
    // the members' own terms, by their owner lanes
    for (int i1 = part; i1 < cnt; i1 += P) {
      const double x12 = at(i1, 0), y12 = at(i1, 1), z12 = at(i1, 2);
      const TersoffSetD& p12 = ters_pair(tp, t1, (int)at(i1, 3));
      const double d12 = sqrt(x12 * x12 + y12 * y12 + z12 * z12);
      double fc12, fcp12;
      ters_fc(p12, d12, fc12, fcp12);
      at(i1, 7) = d12;
      at(i1, 8) = fc12;
      at(i1, 9) = fcp12;
      at(i1, 10) = p12.b * exp(-p12.mu * d12);
    }
    NEPMI_WAVE_LDS_SYNC();
    // step 1: bond order of the bonds this lane owns
    for (int i1 = part; i1 < cnt; i1 += P) {
      const double x12 = at(i1, 0), y12 = at(i1, 1), z12 = at(i1, 2);
      const double d12 = at(i1, 7);
      double zeta = 0.0;
      for (int i2 = 0; i2 < cnt; ++i2) {
        if (i2 == i1)
          continue;
        const double x13 = at(i2, 0), y13 = at(i2, 1), z13 = at(i2, 2);
        const double d13 = at(i2, 7), fc13 = at(i2, 8);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) / (d12 * d13);
        const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
        zeta += fc13 * (s1.one_plus_c2overd2 - s1.c2 / tmp);
      }
      const double bzn = pow(s1.beta * zeta, s1.n);
      const double b12 = pow(1.0 + bzn, s1.minus_half_over_n);
      if (zeta < 1.0e-16) { // avoid division by 0
        at(i1, 4) = 1.0;
        at(i1, 5) = 0.0;
      } else {
        at(i1, 4) = b12;
        at(i1, 5) = -b12 * bzn * 0.5 / ((1.0 + bzn) * zeta);
      }
    }
    NEPMI_WAVE_LDS_SYNC();
    // step 2: partial forces and energy of the bonds this lane owns
    for (int i1 = part; i1 < cnt; i1 += P) {
      int slot = 0;
      {
        unsigned long long rest = inr;
        for (int q = 0; q < i1; ++q)
          rest &= rest - 1ull;
        slot = (int)__builtin_ctzll(rest);
      }
      const double x12 = at(i1, 0), y12 = at(i1, 1), z12 = at(i1, 2);
      const int t2 = (int)at(i1, 3);
      const TersoffSetD& p12 = ters_pair(tp, t1, t2);
      const double d12 = at(i1, 7), fc12 = at(i1, 8), fcp12 = at(i1, 9), fa12 = at(i1, 10);
      const double d12inv = 1.0 / d12;
      const double fap12 = -p12.mu * fa12;
      const double fr12 = p12.a * exp(-p12.lambda * d12), frp12 = -p12.lambda * fr12;
      const double b12 = at(i1, 4), bp12 = at(i1, 5);
      const double factor3 = (fcp12 * (fr12 - b12 * fa12) + fc12 * (frp12 - b12 * fap12)) * d12inv;
      double fx = x12 * factor3 * 0.5, fy = y12 * factor3 * 0.5, fz = z12 * factor3 * 0.5;
      at(i1, 6) = fc12 * (fr12 - b12 * fa12) * 0.5;
      for (int i2 = 0; i2 < cnt; ++i2) {
        if (i2 == i1)
          continue;
        const double x13 = at(i2, 0), y13 = at(i2, 1), z13 = at(i2, 2);
        const double d13 = at(i2, 7), fc13 = at(i2, 8), fa13 = at(i2, 10);
        const double bp13 = at(i2, 5);
        const double od = 1.0 / (d12 * d13);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) * od;
        const double c_over = c123 * d12inv * d12inv;
        const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
        const double g123 = s1.one_plus_c2overd2 - s1.c2 / tmp;
        const double gp123 = 2.0 * s1.c2 * (c123 - s1.h) / (tmp * tmp);
        const double ta = (-bp12 * fc12 * fa12 * fc13 - bp13 * fc13 * fa13 * fc12) * gp123;
        const double tbb = -bp13 * fc13 * fa13 * fcp12 * g123 * d12inv;
        fx += (x12 * tbb + ta * (x13 * od - x12 * c_over)) * 0.5;
        fy += (y12 * tbb + ta * (y13 * od - y12 * c_over)) * 0.5;
        fz += (z12 * tbb + ta * (z13 * od - z12 * c_over)) * 0.5;
      }
      D4 out;
      out.x = fx;
      out.y = fy;
      out.z = fz;
      out.w = 0;
      tb.f12[(int64_t)slot * N + k] = out;
    }
    NEPMI_WAVE_LDS_SYNC();
    if (part == 0) { // the bonds' energies in bond order: the sum of the one-lane form, bit for bit
      double u = 0.0;
      for (int m = 0; m < cnt; ++m)
        u += at(m, 6);
      tb.pe_d[k] = u;
    }
  }

  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < 1)
      return;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    const TersoffSetD& s1 = tp.p[t1];
    const int nn = b.nn_ang[k];
    int cnt = 0;
    // the Verlet slots inside the cutoff, as bits: the nested loops below visit only those (a Verlet list of ~20
    // entries holds 4 bonded neighbours in silicon; skipping the others one dependent load at a time was what this
    // kernel's 0.2 ms consisted of).  Lists beyond 64 entries keep every slot marked and test the record.
    unsigned long long inr = 0ull;
    const bool masked = nn <= 64;
    // geometry + membership of every Verlet entry
    for (int s = 0; s < nn; ++s) {
      const int j = b.nl_ang[(int64_t)s * N + k];
      const PosQ p2 = b.posq[j];
      float xf, yf, zf;
      const float d2f = pair_geometry(box, p1, p2, xf, yf, zf); // float test of the local list
      D4 r;
      r.x = p2.x - p1.x;
      r.y = p2.y - p1.y;
      r.z = p2.z - p1.z;
      mic_d(box, r.x, r.y, r.z);
      r.w = d2f < tp.rc_sq ? (1 | ((long long)p2.type << 8)) : 0;
      cnt += (int)(r.w & 1);
      if (masked && (r.w & 1))
        inr |= 1ull << s;
      tb.rec[(int64_t)s * N + k] = r;
    }
    // next slot >= s to visit (nn when there is none)
    auto next_slot = [&](int s) -> int {
      if (!masked)
        return s;
      const unsigned long long rest = s < 64 ? inr >> s : 0ull;
      return rest ? s + (int)__builtin_ctzll(rest) : nn;
    };
    b.nn_rad[k] = cnt;
    b.nn_angstep[k] = cnt;
    tb.mask[k] = inr; // (meaningful for lists of at most 64 entries: the readers check nn)
    // step 1: bond order
    for (int i1 = next_slot(0); i1 < nn; i1 = next_slot(i1 + 1)) {
      const D4 r12 = tb.rec[(int64_t)i1 * N + k];
      if (!(r12.w & 1))
        continue;
      const double d12 = sqrt(r12.x * r12.x + r12.y * r12.y + r12.z * r12.z);
      double zeta = 0.0;
      for (int i2 = next_slot(0); i2 < nn; i2 = next_slot(i2 + 1)) {
        if (i2 == i1)
          continue;
        const D4 r13 = tb.rec[(int64_t)i2 * N + k];
        if (!(r13.w & 1))
          continue;
        const int t3 = (int)(r13.w >> 8);
        const double d13 = sqrt(r13.x * r13.x + r13.y * r13.y + r13.z * r13.z);
        const double c123 = (r12.x * r13.x + r12.y * r13.y + r12.z * r13.z) / (d12 * d13);
        double fc13, fcp13;
        ters_fc(ters_pair(tp, t1, t3), d13, fc13, fcp13);
        const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
        zeta += fc13 * (s1.one_plus_c2overd2 - s1.c2 / tmp);
      }
      const double bzn = pow(s1.beta * zeta, s1.n);
      const double b12 = pow(1.0 + bzn, s1.minus_half_over_n);
      if (zeta < 1.0e-16) { // avoid division by 0
        tb.bb[(int64_t)i1 * N + k] = 1.0;
        tb.bp[(int64_t)i1 * N + k] = 0.0;
      } else {
        tb.bb[(int64_t)i1 * N + k] = b12;
        tb.bp[(int64_t)i1 * N + k] = -b12 * bzn * 0.5 / ((1.0 + bzn) * zeta);
      }
    }
    // step 2: partial forces and energy
    double u = 0.0;
    for (int i1 = next_slot(0); i1 < nn; i1 = next_slot(i1 + 1)) {
      const D4 r12 = tb.rec[(int64_t)i1 * N + k];
      D4 out;
      out.x = out.y = out.z = 0.0;
      out.w = 0;
      if (r12.w & 1) {
        const int t2 = (int)(r12.w >> 8);
        const TersoffSetD& p12 = ters_pair(tp, t1, t2);
        const double d12 = sqrt(r12.x * r12.x + r12.y * r12.y + r12.z * r12.z);
        const double d12inv = 1.0 / d12;
        double fc12, fcp12;
        ters_fc(p12, d12, fc12, fcp12);
        const double fa12 = p12.b * exp(-p12.mu * d12), fap12 = -p12.mu * fa12;
        const double fr12 = p12.a * exp(-p12.lambda * d12), frp12 = -p12.lambda * fr12;
        const double b12 = tb.bb[(int64_t)i1 * N + k], bp12 = tb.bp[(int64_t)i1 * N + k];
        const double factor3 = (fcp12 * (fr12 - b12 * fa12) + fc12 * (frp12 - b12 * fap12)) * d12inv;
        double fx = r12.x * factor3 * 0.5, fy = r12.y * factor3 * 0.5, fz = r12.z * factor3 * 0.5;
        u += fc12 * (fr12 - b12 * fa12) * 0.5;
        for (int i2 = next_slot(0); i2 < nn; i2 = next_slot(i2 + 1)) {
          if (i2 == i1)
            continue;
          const D4 r13 = tb.rec[(int64_t)i2 * N + k];
          if (!(r13.w & 1))
            continue;
          const int t3 = (int)(r13.w >> 8);
          const TersoffSetD& p13 = ters_pair(tp, t1, t3);
          const double d13 = sqrt(r13.x * r13.x + r13.y * r13.y + r13.z * r13.z);
          double fc13, fcp13;
          ters_fc(p13, d13, fc13, fcp13);
          const double fa13 = p13.b * exp(-p13.mu * d13);
          const double bp13 = tb.bp[(int64_t)i2 * N + k];
          const double od = 1.0 / (d12 * d13);
          const double c123 = (r12.x * r13.x + r12.y * r13.y + r12.z * r13.z) * od;
          const double c_over = c123 * d12inv * d12inv;
          const double tmp = s1.d2 + (c123 - s1.h) * (c123 - s1.h);
          const double g123 = s1.one_plus_c2overd2 - s1.c2 / tmp;
          const double gp123 = 2.0 * s1.c2 * (c123 - s1.h) / (tmp * tmp);
          const double ta = (-bp12 * fc12 * fa12 * fc13 - bp13 * fc13 * fa13 * fc12) * gp123;
          const double tbb = -bp13 * fc13 * fa13 * fcp12 * g123 * d12inv;
          fx += (r12.x * tbb + ta * (r13.x * od - r12.x * c_over)) * 0.5;
          fy += (r12.y * tbb + ta * (r13.y * od - r12.y * c_over)) * 0.5;
          fz += (r12.z * tbb + ta * (r13.z * od - r12.z * c_over)) * 0.5;
        }
        out.x = fx;
        out.y = fy;
        out.z = fz;
      }
      tb.f12[(int64_t)i1 * N + k] = out;
    }
    tb.pe_d[k] = u;
  }
};

struct TersoffAssembleBody {
  Bufs b;
  TersoffBufs tb;
  NEPMI_HD void operator()(int64_t k) const
  {
    if (b.lvl[k] < 2)
      return;
    double F[3];
    assemble<true>(k, F);
  }
  // OUT: energy and virial planes written too (else the force planes only; the force is also handed back)
  template <bool OUT>
  NEPMI_HD void assemble(int64_t k, double* F) const
  {
    const int64_t N = b.N;
    F[0] = F[1] = F[2] = 0.0;
    double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nn = b.nn_ang[k];
    if (nn <= 64) {
      // the members straight from the membership bits, four at a time with their loads in flight together, in slot order
      // (the same order of additions as the scan below)
      unsigned long long rest = tb.mask[k];
      while (rest) {
        constexpr int MB = 4;
        int sl[MB];
        bool on[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
          on[u] = rest != 0ull;
          sl[u] = on[u] ? (int)__builtin_ctzll(rest) : 0;
          rest &= rest - 1ull;
        }
        D4 r[MB], fa[MB], fb[MB];
        int jj[MB], rs[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
          r[u] = tb.rec[(int64_t)sl[u] * N + k];
          jj[u] = b.nl_ang[(int64_t)sl[u] * N + k];
          rs[u] = b.rev_ang[(int64_t)sl[u] * N + k];
          fa[u] = tb.f12[(int64_t)sl[u] * N + k];
        }
#pragma unroll
        for (int u = 0; u < MB; ++u)
          fb[u] = tb.f12[(int64_t)rs[u] * N + jj[u]];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
          if (!on[u])
            continue;
          F[0] += fa[u].x - fb[u].x;
          F[1] += fa[u].y - fb[u].y;
          F[2] += fa[u].z - fb[u].z;
          W[0] += r[u].x * fb[u].x;
          W[1] += r[u].y * fb[u].y;
          W[2] += r[u].z * fb[u].z;
          W[3] += r[u].x * fb[u].y;
          W[4] += r[u].x * fb[u].z;
          W[5] += r[u].y * fb[u].z;
          W[6] += r[u].y * fb[u].x;
          W[7] += r[u].z * fb[u].x;
          W[8] += r[u].z * fb[u].y;
        }
      }
    }
    for (int s = 0; s < (nn <= 64 ? 0 : nn); ++s) {
      const D4 r = tb.rec[(int64_t)s * N + k];
      if (!(r.w & 1))
        continue;
      const int j = b.nl_ang[(int64_t)s * N + k];
      const int rs = b.rev_ang[(int64_t)s * N + k];
      const D4 f12 = tb.f12[(int64_t)s * N + k];
      const D4 f21 = tb.f12[(int64_t)rs * N + j];
      F[0] += f12.x - f21.x;
      F[1] += f12.y - f21.y;
      F[2] += f12.z - f21.z;
      W[0] += r.x * f21.x;
      W[1] += r.y * f21.y;
      W[2] += r.z * f21.z;
      W[3] += r.x * f21.y;
      W[4] += r.x * f21.z;
      W[5] += r.y * f21.z;
      W[6] += r.y * f21.x;
      W[7] += r.z * f21.x;
      W[8] += r.z * f21.y;
    }
    double* __restrict__ fo = b.fo + k;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      fo[(int64_t)(kOutF + d) * N] = F[d];
    if (OUT) {
      fo[0] = tb.pe_d[k];
#pragma unroll
      for (int d = 0; d < 9; ++d)
        fo[(int64_t)(kOutW + d) * N] = W[d];
    }
  }
};

// The seam between two NVE steps of a Tersoff run loop as ONE pass over the atoms: the force assembly of the step just
// evaluated (many-body accumulate, potential.cu:35-134: partial forces and reverse slots only -- it reads no positions, so the
// drift below cannot disturb another lane's assembly), the second half-kick of that step and the first half of the next.
// Config 2 (13,824 atoms) is bound by the number of launches (three kernels of 4-13 us per step): one launch and its
// ramp less.  Steps that record thermo data or end the run keep the separate kernels (energies and virials are written).
// Bit-identical to the separate kernels (the same additions in the same order).
struct TersoffSeamBody {
  TersoffAssembleBody as;
  ResidentStepBody rs;
  NEPMI_HD void operator()(int64_t k) const
  {
    if (rs.frozen_now())
      return;
    if (as.b.lvl[k] < 2)
      return;
    double F[3];
    as.template assemble<false>(k, F);
    rs.step(k, F);
  }
};

// the local list of the last call, caller indices, ascending (for bit-exact list checks)
struct TersoffExportBody {
  Bufs b;
  TersoffBufs tb;
  int* nn_out;
  int* nl_out;
  int64_t ld;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t i = b.perm[k];
    int cnt = 0;
    const bool by_mask = b.nn_ang[k] <= 64;
    const unsigned long long mk = tb.mask[k];
    for (int s = 0; s < b.nn_ang[k]; ++s) {
      if (by_mask ? !((mk >> s) & 1ull) : !(tb.rec[(int64_t)s * N + k].w & 1))
        continue;
      const int jc = b.perm[b.nl_ang[(int64_t)s * N + k]];
      if (cnt < ld) {
        int p = cnt - 1;
        while (p >= 0 && nl_out[(int64_t)p * N + i] > jc) {
          nl_out[(int64_t)(p + 1) * N + i] = nl_out[(int64_t)p * N + i];
          --p;
        }
        nl_out[(int64_t)(p + 1) * N + i] = jc;
      }
      ++cnt;
    }
    nn_out[i] = cnt;
  }
};

} // namespace nepmi
