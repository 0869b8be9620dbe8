// l_max_3body = 5..8: the part of the angular descriptor and of its force that lies beyond the 24 sums of l <= 4.
//
// The reference carries 80 sums per radial channel (NUM_OF_ABC, src/utilities/nep_utilities.cuh:18) and walks them with
// run-time loops over coefficient tables (accumulate_s_one<L> :1674-1723, find_q_one<L> :1758-1770, calculate_s_one<L> /
// accumulate_f12_one<L> :1327-1434).  Blocks l >= 5 enter the descriptor only through the 3-body rows
//     q_l = C3B[l,0] s_{l,0}^2 + 2 sum_m C3B[l,m] (s_{l,m,Re}^2 + s_{l,m,Im}^2),
// every 4-/5-body row is built from l <= 4.  So a model with l_max > 4 is the generic shape of this engine (24 sums,
// rows l = 1..4 and the optional rows, untouched) PLUS the two kernels below, which add the rows l = 5..l_max to q and
// their adjoint contribution to the per-pair partial forces f12:
//     HighLDescBody   per channel n: s_{n,lm} = sum_j g_n(r_ij) b_lm(rhat_ij) (registers; stored to Bufs::shi), rows q_l
//     HighLForceBody  per pair: P_lm = sum_n G_{n,lm} g_n, Q_lm = sum_n G_{n,lm} g_n' with G = dU_i/ds = (2|4) C3B Fp s,
//                     contracted with b_lm and grad b_lm exactly like the l <= 4 part (AngularForceBody), added to f12
// The basis functions are b_{l,0} = Z_{l,0}(z), b_{l,m} = Z_{l,m}(z) (Re, Im)(x+iy)^m with the integer polynomials
// Z_{l,m} = d^m P_l / dz^m (scaled to coprime coefficients) -- tables generated from that definition by
// tools/gen_highl_tables.py, constants folded at compile time (every loop below has constant bounds).
// A rarely used model family: one lane per atom, no tuning beyond keeping the sums in registers.
#pragma once
#include "nep_bodies.h"
#include "nep_highl_tables.h"

namespace nepmi {

constexpr int kHighSums = 56; // sums 24..79: l = 5 (11), 6 (13), 7 (15), 8 (17)
NEPMI_HD constexpr int highl_offset(int L) { return L * L - 25; } // first sum of block l within the 56

struct HighLGeom {
  float zp[9];    // z^k
  float re[9];    // Re (x+iy)^m
  float im[9];    // Im (x+iy)^m
};

NEPMI_HD void highl_geom(float x, float y, float z, HighLGeom& g)
{
  g.zp[0] = 1.0f;
  g.re[0] = 1.0f;
  g.im[0] = 0.0f;
#pragma unroll
  for (int k = 1; k <= 8; ++k) {
    g.zp[k] = g.zp[k - 1] * z;
    g.re[k] = x * g.re[k - 1] - y * g.im[k - 1];
    g.im[k] = x * g.im[k - 1] + y * g.re[k - 1];
  }
}

// Z_{L,m}(z) and its derivative (compile-time L, m: the zero coefficients vanish from the code)
template <int L>
NEPMI_HD void highl_zpoly(int m, const HighLGeom& g, float& zf, float& dzf)
{
  const float Z[4][9][9] = NEPMI_HIGHL_Z_INIT;
  zf = 0.0f;
  dzf = 0.0f;
#pragma unroll
  for (int k = 0; k <= 8; ++k) {
    const float c = Z[L - 5][m][k];
    if (k <= L - m && c != 0.0f) {
      zf += c * g.zp[k];
      if (k > 0)
        dzf += c * (float)k * g.zp[k - 1];
    }
  }
}

// s[block L] += gn * b_{L,m}(u)
template <int L>
NEPMI_HD void highl_accumulate(const HighLGeom& g, float gn, float* s)
{
  constexpr int off = highl_offset(L);
#pragma unroll
  for (int m = 0; m <= L; ++m) {
    float zf, dzf;
    highl_zpoly<L>(m, g, zf, dzf);
    zf *= gn;
    if (m == 0) {
      s[off] += zf;
    } else {
      s[off + 2 * m - 1] += zf * g.re[m];
      s[off + 2 * m] += zf * g.im[m];
    }
  }
}

template <int L>
NEPMI_HD float highl_row(const float* s)
{
  const float C3B[kHighSums] = NEPMI_HIGHL_C3B_INIT;
  constexpr int off = highl_offset(L);
  float acc = 0.0f;
#pragma unroll
  for (int h = 1; h < 2 * L + 1; ++h)
    acc += C3B[off + h] * s[off + h] * s[off + h];
  return 2.0f * acc + C3B[off] * s[off] * s[off];
}

// angular part of find_descriptor for l = 5..l_max (nep.cu:549-640 with L_max > 4), after AngularDescBody
struct HighLDescBody {
  ModelD m;
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_desc)
      return;
    const int64_t gk = b.tpos[k];
    const int NR = m.NR, NA = m.NA, KA = m.KA, Lmax = m.Lmax;
    const int t1 = b.posq[k].type;
    const float rc1 = m.rc_a[t1];
    const int na = b.nn_angstep[k];
    const F4* __restrict__ acomp = b.acomp + k;
    for (int n = 0; n <= NA; ++n) {
      float s[kHighSums];
#pragma unroll
      for (int h = 0; h < kHighSums; ++h)
        s[h] = 0.0f;
      for (int a = 0; a < na; ++a) {
        const F4 e = acomp[(int64_t)a * N];
        const int t2 = (int)((unsigned)e.w >> kIdxBits);
        float d, dinv;
        dist_and_inv(dot3f(e.x, e.x, e.y, e.y, e.z, e.z), d, dinv);
        const float rc = m.uniform_rc ? m.rc_a_max : (rc1 + m.rc_a[t2]) * 0.5f;
        const float rcinv = fast_rcp(rc);
        float fc;
        cutoff_fc(rcinv, d, fc);
        float fn[20];
        basis_fn_rt(KA, rcinv, d, fc, fn);
        const float* c = m.c_ang + ((size_t)(t1 * m.T + t2) * (NA + 1) + n) * (KA + 1);
        float gn = 0.0f;
        for (int kk = 0; kk <= KA; ++kk)
          gn = fmaf(fn[kk], c[kk], gn);
        HighLGeom g;
        highl_geom(e.x * dinv, e.y * dinv, e.z * dinv, g);
        highl_accumulate<5>(g, gn, s);
        if (Lmax >= 6)
          highl_accumulate<6>(g, gn, s);
        if (Lmax >= 7)
          highl_accumulate<7>(g, gn, s);
        if (Lmax >= 8)
          highl_accumulate<8>(g, gn, s);
      }
      auto put = [&](int L, float q) {
        const int d = (NR + 1) + (L - 1) * (NA + 1) + n;
        b.q[(int64_t)d * N + gk] = q * m.qscale[d];
      };
      put(5, highl_row<5>(s));
      if (Lmax >= 6)
        put(6, highl_row<6>(s));
      if (Lmax >= 7)
        put(7, highl_row<7>(s));
      if (Lmax >= 8)
        put(8, highl_row<8>(s));
      const int nh = (Lmax + 1) * (Lmax + 1) - 25;
#pragma unroll
      for (int h = 0; h < kHighSums; ++h)
        if (h < nh)
          b.shi[(int64_t)(n * kHighSums + h) * N + k] = s[h];
    }
  }
};

// contribution of block L to (w, v) of one pair: w = sum Q b, v = sum P grad b (harmonics_contract's convention)
template <int L>
NEPMI_HD void highl_force_block(
  const ModelD& m, const Bufs& b, int64_t k, int64_t gk, const HighLGeom& g, const float* gn, const float* gpn,
  float& w, float& vx, float& vy, float& vz)
{
  const float C3B[kHighSums] = NEPMI_HIGHL_C3B_INIT;
  constexpr int off = highl_offset(L);
  const int64_t N = b.N;
  float P[2 * L + 1], Q[2 * L + 1];
#pragma unroll
  for (int h = 0; h < 2 * L + 1; ++h)
    P[h] = Q[h] = 0.0f;
  for (int n = 0; n <= m.NA; ++n) {
    const float F = b.fp[(int64_t)((m.NR + 1) + (L - 1) * (m.NA + 1) + n) * N + gk];
    const float gF = gn[n] * F, gpF = gpn[n] * F;
#pragma unroll
    for (int h = 0; h < 2 * L + 1; ++h) {
      const float G = (h == 0 ? 2.0f : 4.0f) * C3B[off + h] * b.shi[(int64_t)(n * kHighSums + off + h) * N + k];
      P[h] = fmaf(G, gF, P[h]);
      Q[h] = fmaf(G, gpF, Q[h]);
    }
  }
#pragma unroll
  for (int mm = 0; mm <= L; ++mm) {
    float zf, dzf;
    highl_zpoly<L>(mm, g, zf, dzf);
    if (mm == 0) {
      w = fmaf(Q[0], zf, w);
      vz = fmaf(P[0], dzf, vz);
    } else {
      const float pr = P[2 * mm - 1], pi = P[2 * mm];
      w = fmaf(zf, Q[2 * mm - 1] * g.re[mm] + Q[2 * mm] * g.im[mm], w);
      vz = fmaf(dzf, pr * g.re[mm] + pi * g.im[mm], vz);
      // d/dx (Re, Im)_m = m (Re, Im)_{m-1};  d/dy (Re, Im)_m = m (-Im, Re)_{m-1}
      const float zm = zf * (float)mm;
      vx = fmaf(zm, pr * g.re[mm - 1] + pi * g.im[mm - 1], vx);
      vy = fmaf(zm, pi * g.re[mm - 1] - pr * g.im[mm - 1], vy);
    }
  }
}

// find_partial_force_angular for l = 5..l_max (accumulate_f12_one<L>, nep_utilities.cuh:1342-1434), after
// AngularForceBody: f12[slot] += ...
struct HighLForceBody {
  ModelD m;
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_desc || (b.level && !b.angf[k]))
      return;
    const int64_t gk = b.tpos[k];
    const int NA = m.NA, KA = m.KA, Lmax = m.Lmax;
    const int t1 = b.posq[k].type;
    const float rc1 = m.rc_a[t1];
    const int na = b.nn_angstep[k];
    const F4* __restrict__ acomp = b.acomp + k;
    F4* __restrict__ f12 = b.f12 + k;
    for (int a = 0; a < na; ++a) {
      const F4 e = acomp[(int64_t)a * N];
      const int t2 = (int)((unsigned)e.w >> kIdxBits);
      float d, dinv;
      dist_and_inv(dot3f(e.x, e.x, e.y, e.y, e.z, e.z), d, dinv);
      const float rc = m.uniform_rc ? m.rc_a_max : (rc1 + m.rc_a[t2]) * 0.5f;
      const float rcinv = fast_rcp(rc);
      float fc, fcp;
      cutoff_fc_fcp(rcinv, d, fc, fcp);
      float fn[20], fnp[20];
      basis_fn_fnp_rt(KA, rcinv, d, fc, fcp, fn, fnp);
      float gn[20], gpn[20];
      for (int n = 0; n <= NA; ++n) {
        const float* c = m.c_ang + ((size_t)(t1 * m.T + t2) * (NA + 1) + n) * (KA + 1);
        float g0 = 0.0f, g1 = 0.0f;
        for (int kk = 0; kk <= KA; ++kk) {
          g0 = fmaf(fn[kk], c[kk], g0);
          g1 = fmaf(fnp[kk], c[kk], g1);
        }
        gn[n] = g0;
        gpn[n] = g1;
      }
      const float ux = e.x * dinv, uy = e.y * dinv, uz = e.z * dinv;
      HighLGeom g;
      highl_geom(ux, uy, uz, g);
      float w = 0.0f, vx = 0.0f, vy = 0.0f, vz = 0.0f;
      highl_force_block<5>(m, b, k, gk, g, gn, gpn, w, vx, vy, vz);
      if (Lmax >= 6)
        highl_force_block<6>(m, b, k, gk, g, gn, gpn, w, vx, vy, vz);
      if (Lmax >= 7)
        highl_force_block<7>(m, b, k, gk, g, gn, gpn, w, vx, vy, vz);
      if (Lmax >= 8)
        highl_force_block<8>(m, b, k, gk, g, gn, gpn, w, vx, vy, vz);
      const float udv = ux * vx + uy * vy + uz * vz;
      F4 out = f12[(int64_t)a * N];
      out.x += ux * w + (vx - ux * udv) * dinv;
      out.y += uy * w + (vy - uy * udv) * dinv;
      out.z += uz * w + (vz - uz * udv) * dinv;
      f12[(int64_t)a * N] = out;
    }
  }
};

} // namespace nepmi
