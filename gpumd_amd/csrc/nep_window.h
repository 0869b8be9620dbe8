// LDS-window kernels of the NEP force path: the two passes that walk the whole Verlet list.
//
// One 256-thread workgroup per brick (4x4x4 cells, ~240 atoms, one lane per atom).  The workgroup first
// stages every atom of the brick's 8x8x8-cell window (~1,700 atoms) in LDS as a 16-byte record
//     { int32 x, y, z : position relative to the window centre in fixed point,  word : index | type << 25 }
// with the periodic image already chosen (each atom is placed next to ITS cell of the rebuild-time grid, and
// the lattice-vector jumps the stored position has made since the rebuild -- PosQ::pad -- are undone).  A pair
// vector is then one ds_read_b128, three integer subtractions and three conversions: no minimum image, no
// FP64, no branches in the inner loops.  The fixed-point grid (window extent / 2^30, ~1e-7 A) is finer than the
// FP32 rounding of the reference's r12 = float(double(x_j) - double(x_i)), so the geometry is as accurate as
// the reference's; it is not bit-identical to it, which matters only for LIST DECISIONS: a candidate whose
// squared distance lies within a narrow band of a cutoff is decided again with the reference's exact arithmetic
// (pair_geometry on the FP64 positions, float minimum image included), so the per-step radial and angular lists
// stay bit-exact (tests/parity_cases.py compares them with the oracle's).
//
//   RadialWinBody  find_neighbor_list_large_box (nep.cu:436-486) + radial half of find_descriptor
//                  (nep.cu:488-547): walks list A then list B, accumulates the radial basis sums, emits the
//                  compact angular records (acomp / amap / aidx) and the compact radial list (ccode: LDS slot of
//                  every pair inside the cutoff)
//   ForceWinBody   find_force_radial (nep.cu:661-772) + gpu_find_force_many_body (potential.cu:170-297):
//                  angular part from the compact records (f12 - f21 through the static reverse slot), radial
//                  part over ccode -- only pairs inside the cutoff, branch-free -- with the neighbour's table
//                  row gathered by index; writes the 13 output planes in internal order (coalesced)
#pragma once
#include "nep_bodies.h"

namespace nepmi {

constexpr int kWinThreads = 256;
constexpr int kWinCells = 512;
constexpr int kWinMaxAtoms = 5000; // window capacity (LDS budget); larger windows take the gather path

struct alignas(16) WinRec {
  int x, y, z;
  int w; // internal index | type << kIdxBits
};

#if defined(__HIP_DEVICE_COMPILE__)
#define NEPMI_LDS(T) __attribute__((address_space(3))) T
#else
#define NEPMI_LDS(T) T
#endif

// LDS layout (bytes): int woff[513] | int wstart[512] | WinRec rec[wmax]
struct WinLayout {
  int wmax;
  NEPMI_HD int off_woff() const { return 0; }
  NEPMI_HD int off_wstart() const { return 2064; } // 513 ints, padded to 16 B
  NEPMI_HD int off_rec() const { return 2064 + 2048; }
  NEPMI_HD int bytes() const { return off_rec() + 16 * wmax; }
};

// geometry of the fixed-point window, set at every list rebuild
struct WinGeom {
  double inv_unit;     // grid points per Angstrom
  double cell_frac[3]; // fractional width of a cell along each lattice direction
  float unit;          // Angstrom per grid point
  float unit2;         // unit^2
  float band;          // |d^2 - rc^2| below this (A^2): the list decision is retaken exactly
};

// Staging, shared by the two passes (identical window contents and slot order in both).
struct WinStage {
  BoxD box;
  Bufs b;
  WinLayout lay;
  WinGeom g;

  NEPMI_HD void brick_coords(int64_t brick, int& bx, int& by, int& bz) const
  {
    bx = (int)(brick % b.gbx);
    by = (int)((brick / b.gbx) % b.gby);
    bz = (int)(brick / ((int64_t)b.gbx * b.gby));
  }

  // phase 1 (all threads): count and first atom of each of the 512 window cells
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const
  {
    NEPMI_LDS(int)* woff = (NEPMI_LDS(int)*)(lds + lay.off_woff());
    NEPMI_LDS(int)* wstart = (NEPMI_LDS(int)*)(lds + lay.off_wstart());
    int bx, by, bz;
    brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += nth) {
      int cx = 4 * bx - 2 + (wc & 7), cy = 4 * by - 2 + ((wc >> 3) & 7), cz = 4 * bz - 2 + (wc >> 6);
      bool ok = true;
      if (box.pbc[0]) cx = ((cx % b.nbx) + b.nbx) % b.nbx; else ok = ok && cx >= 0 && cx < b.nbx;
      if (box.pbc[1]) cy = ((cy % b.nby) + b.nby) % b.nby; else ok = ok && cy >= 0 && cy < b.nby;
      if (box.pbc[2]) cz = ((cz % b.nbz) + b.nbz) % b.nbz; else ok = ok && cz >= 0 && cz < b.nbz;
      int cnt = 0, st = 0;
      if (ok) {
        const int c = cell_index(b, cx, cy, cz);
        st = b.cell_count[c];
        cnt = b.cell_count[c + 1] - st;
      }
      woff[wc] = cnt; // turned into the exclusive prefix by the backend's scan (woff[512] = total)
      wstart[wc] = st;
    }
    if (tid == 0)
      woff[kWinCells] = 0;
  }

  // fixed-point position of an atom placed next to the point with fractional coordinates (fx, fy, fz)
  // (a cell centre for window atoms, the window centre for the brick's own atoms), relative to the
  // window centre c0
  NEPMI_HD void place(const PosQ& p, const double* c0, double fx, double fy, double fz, int& qx, int& qy, int& qz) const
  {
    const double* h = box.h;
    double sx = h[9] * p.x + h[10] * p.y + h[11] * p.z;
    double sy = h[12] * p.x + h[13] * p.y + h[14] * p.z;
    double sz = h[15] * p.x + h[16] * p.y + h[17] * p.z;
    // undo the lattice-vector jumps since the rebuild, then take the image nearest to the reference point
    sx -= (double)img_of(p.pad, 0);
    sy -= (double)img_of(p.pad, 1);
    sz -= (double)img_of(p.pad, 2);
    if (box.pbc[0]) sx += nearbyint(fx - sx);
    if (box.pbc[1]) sy += nearbyint(fy - sy);
    if (box.pbc[2]) sz += nearbyint(fz - sz);
    sx -= c0[0];
    sy -= c0[1];
    sz -= c0[2];
    const double rx = h[0] * sx + h[1] * sy + h[2] * sz;
    const double ry = h[3] * sx + h[4] * sy + h[5] * sz;
    const double rz = h[6] * sx + h[7] * sy + h[8] * sz;
    qx = (int)nearbyint(rx * g.inv_unit);
    qy = (int)nearbyint(ry * g.inv_unit);
    qz = (int)nearbyint(rz * g.inv_unit);
  }

  NEPMI_HD void window_centre(int64_t brick, double* c0) const
  {
    int bx, by, bz;
    brick_coords(brick, bx, by, bz);
    c0[0] = (double)(4 * bx + 2) * g.cell_frac[0];
    c0[1] = (double)(4 * by + 2) * g.cell_frac[1];
    c0[2] = (double)(4 * bz + 2) * g.cell_frac[2];
  }

  // phase 3 (all threads, after the scan of woff): copy the window atoms
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const
  {
    NEPMI_LDS(const int)* woff = (NEPMI_LDS(const int)*)(lds + lay.off_woff());
    NEPMI_LDS(const int)* wstart = (NEPMI_LDS(const int)*)(lds + lay.off_wstart());
    NEPMI_LDS(WinRec)* rec = (NEPMI_LDS(WinRec)*)(lds + lay.off_rec());
    int bx, by, bz;
    brick_coords(brick, bx, by, bz);
    double c0[3];
    window_centre(brick, c0);
    const int W = woff[kWinCells] < lay.wmax ? woff[kWinCells] : lay.wmax;
    for (int w = tid; w < W; w += nth) {
      int lo = 0, hi = kWinCells - 1; // largest wc with woff[wc] <= w
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (woff[mid] <= w) lo = mid; else hi = mid - 1;
      }
      const int j = wstart[lo] + (w - woff[lo]);
      const PosQ p = b.posq[j];
      // centre of the (unwrapped) window cell in fractional coordinates
      const double fx = ((double)(4 * bx - 2 + (lo & 7)) + 0.5) * g.cell_frac[0];
      const double fy = ((double)(4 * by - 2 + ((lo >> 3) & 7)) + 0.5) * g.cell_frac[1];
      const double fz = ((double)(4 * bz - 2 + (lo >> 6)) + 0.5) * g.cell_frac[2];
      WinRec r;
      place(p, c0, fx, fy, fz, r.x, r.y, r.z);
      r.w = (int)((unsigned)j | ((unsigned)p.type << kIdxBits));
      rec[w] = r;
    }
  }

  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const
  {
    a0 = b.cell_count[brick * 64];
    a1 = b.cell_count[brick * 64 + 64];
  }

  // the brick's own atom k in the same fixed-point frame
  NEPMI_HD void place_own(int64_t brick, const PosQ& p, int& qx, int& qy, int& qz) const
  {
    double c0[3];
    window_centre(brick, c0);
    place(p, c0, c0[0], c0[1], c0[2], qx, qy, qz);
  }
};

constexpr int kWinG = 4; // candidates whose LDS look-ups and arithmetic are interleaved

template <class S>
struct RadialWinBody {
  WinStage st;
  ModelD m;
  int first;          // workgroup w runs brick_order[first + w] (first < 0: brick w)
  const int* frozen;  // fused run loops: a non-zero value means "a list rebuild is pending": do nothing
  static constexpr int kMinWavesPerEu = 1;

  NEPMI_HD int lds_bytes() const { return st.lay.bytes(); }
  NEPMI_HD int64_t map_brick(int64_t w) const { return first < 0 ? w : (int64_t)st.b.brick_order[first + w]; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const { st.stage_copy(brick, lds, tid, nth); }
  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const { st.brick_range(brick, a0, a1); }

  template <class LC>
  NEPMI_HD void compute(int64_t brick, int64_t k, LC lds) const
  {
    const Bufs& b = st.b;
    const int64_t N = b.N;
    if (b.lvl[k] < 1) { // outer ghost: lends its position only
      b.nn_rad[k] = 0;
      b.nn_angstep[k] = 0;
      return;
    }
    NEPMI_LDS(const int)* woff = (NEPMI_LDS(const int)*)(lds + st.lay.off_woff());
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(brick, p1, ox, oy, oz);
    const float rc1 = m.rc_r[t1], rca1 = m.rc_a[t1];
    const float unit = st.g.unit, unit2 = st.g.unit2, band = st.g.band;
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    float Ssum[TSM][S::KRM + 1];
    float q[S::NRM + 1];
#pragma unroll
    for (int t = 0; t < TSM; ++t)
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk)
        Ssum[t][kk] = 0.0f;
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n)
      q[n] = 0.0f;

    const int na = b.nn_ang[k], nbn = b.nn_skin[k];
    int cnt = 0, ca = 0;
    F4* __restrict__ acomp = b.acomp + k;
    unsigned short* __restrict__ amap = b.amap + k;
    unsigned short* __restrict__ aidx = b.aidx + k;
    unsigned short* __restrict__ ccode = b.ccode + k;

    // one candidate: window slot -> pair vector, list decisions, accumulation
    auto candidate = [&](const unsigned code, const int idx, const bool live, auto in_list_a) {
      constexpr bool LIST_A = decltype(in_list_a)::value;
      const int slot = woff[code >> 7] + (int)(code & 127u);
      const WinRec r = wrec[slot];
      const float fx = (float)(r.x - ox), fy = (float)(r.y - oy), fz = (float)(r.z - oz);
      const float d2 = dot3f(fx, fx, fy, fy, fz, fz) * unit2;
      const int t2 = (int)((unsigned)r.w >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
      const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[t2]) * 0.5f;
      bool inside = d2 < rc * rc;
      bool ang = LIST_A && d2 < rca * rca;
      // within the band of a cutoff the decision is retaken with the reference's arithmetic (rare)
      if (live && (fabsf(d2 - rc * rc) < band || (LIST_A && fabsf(d2 - rca * rca) < band))) {
        float ex, ey, ez;
        const float d2e = pair_geometry(st.box, p1, b.posq[(unsigned)r.w & (unsigned)kIdxMask], ex, ey, ez);
        inside = d2e < rc * rc;
        ang = LIST_A && d2e < rca * rca;
      }
      inside = inside && live;
      ang = ang && live;
      if (LIST_A && live) {
        unsigned short cs = kNoSlot;
        if (ang) {
          if (ca < b.MN_acomp) {
            F4 e;
            e.x = fx * unit;
            e.y = fy * unit;
            e.z = fz * unit;
            e.w = r.w;
            acomp[(int64_t)ca * N] = e;
            aidx[(int64_t)ca * N] = (unsigned short)idx;
            cs = (unsigned short)ca;
          }
          ++ca;
        }
        amap[(int64_t)idx * N] = cs;
      }
      if (inside) {
        if (cnt < b.MN_rad)
          ccode[(int64_t)cnt * N] = (unsigned short)slot;
        ++cnt;
      }
      if (S::TS > 0) {
        // branch-free accumulation: entries outside the cutoff run the same arithmetic with weight 0 (the
        // envelope is evaluated at min(d, rc) to stay finite)
        float d, dinv;
        dist_and_inv(d2, d, dinv);
        const float rcinv = m.uniform_rc ? m.rcinv_r : fast_rcp(rc);
        const float dc = inside ? d : rc;
        float fc;
        cutoff_fc(rcinv, dc, fc);
        float fn[S::KRM + 1];
        basis_fn<S::KRM>(rcinv, dc, fc, fn);
#pragma unroll
        for (int t = 0; t < TSM; ++t) {
          const float w = (inside && (TSM == 1 || t2 == t)) ? 1.0f : 0.0f;
#pragma unroll
          for (int kk = 0; kk <= S::KRM; ++kk)
            Ssum[t][kk] = fmaf(w, fn[kk], Ssum[t][kk]);
        }
      } else if (inside) {
        float d, dinv;
        dist_and_inv(d2, d, dinv);
        const float rcinv = fast_rcp(rc);
        float fc;
        cutoff_fc(rcinv, d, fc);
        float fn[S::KRM + 1];
        if (S::fixed)
          basis_fn<S::KRM>(rcinv, d, fc, fn);
        else
          basis_fn_rt(KR, rcinv, d, fc, fn);
        const float* c = m.c_rad + (size_t)(t1 * m.T + t2) * (NR + 1) * (KR + 1);
        for (int n = 0; n <= NR; ++n) {
          float gsum = 0.0f;
          for (int kk = 0; kk <= KR; ++kk)
            gsum += fn[kk] * c[n * (KR + 1) + kk];
          q[n] += gsum;
        }
      }
    };

    // walk a list in chunks of kWinG: the codes of the next chunk are requested before this chunk's stores
    auto walk = [&](const unsigned short* __restrict__ codes, const int nn, auto in_list_a) {
      unsigned cur[kWinG], nxt[kWinG];
#pragma unroll
      for (int u = 0; u < kWinG; ++u)
        cur[u] = nn > 0 ? codes[(int64_t)(u < nn ? u : nn - 1) * N] : 0u;
      for (int s0 = 0; s0 < nn; s0 += kWinG) {
#pragma unroll
        for (int u = 0; u < kWinG; ++u) {
          const int idx = s0 + kWinG + u;
          nxt[u] = codes[(int64_t)(idx < nn ? idx : nn - 1) * N];
        }
#pragma unroll
        for (int u = 0; u < kWinG; ++u)
          candidate(cur[u], s0 + u, s0 + u < nn, in_list_a);
#pragma unroll
        for (int u = 0; u < kWinG; ++u)
          cur[u] = nxt[u];
      }
    };
    walk(b.code_ang + k, na, std::true_type{});
    walk(b.code_skin + k, nbn, std::false_type{});

    if (ca > b.MN_acomp || cnt > b.MN_rad) {
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 4);
      ca = ca > b.MN_acomp ? b.MN_acomp : ca;
    }
    b.nn_rad[k] = cnt;
    b.nn_angstep[k] = ca;

    if (S::TS > 0) {
      // q[n] = sum_t2 sum_k c[t1][t2][n][k] S[t2][k]; type loop is wave-uniform => scalar loads
      for (int tu = 0; tu < m.T; ++tu) {
        if (!NEPMI_WAVE_ANY(t1 == tu))
          continue;
        float qq[S::NRM + 1];
#pragma unroll
        for (int n = 0; n <= S::NRM; ++n)
          qq[n] = 0.0f;
#pragma unroll
        for (int t2 = 0; t2 < TSM; ++t2) {
          cfloat_ptr c = as_const(m.c_rad) + (size_t)(tu * m.T + t2) * (S::NRM + 1) * (S::KRM + 1);
#pragma unroll
          for (int n = 0; n <= S::NRM; ++n)
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              qq[n] = fmaf(c[n * (S::KRM + 1) + kk], Ssum[t2][kk], qq[n]);
        }
        if (t1 == tu) {
#pragma unroll
          for (int n = 0; n <= S::NRM; ++n)
            q[n] = qq[n];
        }
      }
    }
    const int64_t gk = b.tpos[k];
    for (int n = 0; n <= NR; ++n)
      b.q[(int64_t)n * N + gk] = q[n] * m.qscale[n];
  }
};

template <class S>
struct ForceWinBody {
  WinStage st;
  ModelD m;
  const int* frozen;
  static constexpr int kMinWavesPerEu = 4; // <= 128 VGPRs: four 256-thread workgroups per CU

  NEPMI_HD int lds_bytes() const { return st.lay.bytes(); }
  NEPMI_HD int64_t map_brick(int64_t w) const { return w; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const { st.stage_copy(brick, lds, tid, nth); }
  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const { st.brick_range(brick, a0, a1); }

  template <class LC>
  NEPMI_HD void compute(int64_t brick, int64_t k, LC lds) const
  {
    const Bufs& b = st.b;
    const int64_t N = b.N;
    if (b.lvl[k] < 2) // forces only for owned atoms
      return;
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(brick, p1, ox, oy, oz);
    const float rc1 = m.rc_r[t1];
    const float unit = st.g.unit, unit2 = st.g.unit2;
    const int KRP = b.KRP;
    const int arow = m.T * KRP;
    const float* __restrict__ atab = b.atab;

    // ---- angular part: f12 - f21 of this step's angular pairs (compact records) ----
    float F[3] = {0, 0, 0};
    float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // xx yy zz xy xz yz yx zx zy
    {
      const int nang = b.nn_angstep[k];
      const F4* __restrict__ acomp = b.acomp + k;
      const F4* __restrict__ f12o = b.f12 + k;
      const unsigned short* __restrict__ aidx = b.aidx + k;
      const unsigned short* __restrict__ rev = b.rev_ang + k;
      for (int a = 0; a < nang; ++a) {
        const F4 e = acomp[(int64_t)a * N];
        const F4 fa = f12o[(int64_t)a * N];
        const int idx = aidx[(int64_t)a * N];
        const int rs = rev[(int64_t)idx * N];
        const int j = (int)((unsigned)e.w & (unsigned)kIdxMask);
        const unsigned short ap = rs != (int)kNoSlot ? b.amap[(int64_t)rs * N + j] : kNoSlot;
        // j has no compact slot for this pair only if its per-step angular list overflowed (MN_angular):
        // that is reported through the overflow flag; never index with kNoSlot
        F4 fb;
        fb.x = fb.y = fb.z = 0.0f;
        fb.w = 0;
        if (ap != kNoSlot)
          fb = b.f12[(int64_t)ap * N + j];
        F[0] += fa.x - fb.x;
        F[1] += fa.y - fb.y;
        F[2] += fa.z - fb.z;
        Wa[0] += e.x * fb.x;
        Wa[1] += e.y * fb.y;
        Wa[2] += e.z * fb.z;
        Wa[3] += e.x * fb.y;
        Wa[4] += e.x * fb.z;
        Wa[5] += e.y * fb.z;
        Wa[6] += e.y * fb.x;
        Wa[7] += e.z * fb.x;
        Wa[8] += e.z * fb.y;
      }
    }

    // ---- radial part over the compact list: every entry is a pair inside the cutoff ----
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    float Aown[TSM][S::KRM + 1];
    if (S::TS > 0) {
#pragma unroll
      for (int t = 0; t < TSM; ++t)
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
          Aown[t][kk] = atab[(size_t)k * arow + t * KRP + kk];
    }
    // accumulated in grid units (x, y, z of a pair are integers times `unit`): Fr = unit * sum, W = unit^2 * sum
    float Fr[3] = {0, 0, 0};
    float W[6] = {0, 0, 0, 0, 0, 0}; // symmetric: xx yy zz xy xz yz
    const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
    const unsigned short* __restrict__ ccode = b.ccode + k;
    constexpr int G = 2;
    unsigned cur[G], nxt[G];
#pragma unroll
    for (int u = 0; u < G; ++u)
      cur[u] = nrad > 0 ? ccode[(int64_t)(u < nrad ? u : nrad - 1) * N] : 0u;
    for (int s0 = 0; s0 < nrad; s0 += G) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int idx = s0 + G + u;
        nxt[u] = ccode[(int64_t)(idx < nrad ? idx : nrad - 1) * N];
      }
      WinRec rr[G];
      float Aj[G][S::KRM + 1];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        rr[u] = wrec[cur[u]];
        const int j = (int)((unsigned)rr[u].w & (unsigned)kIdxMask);
        const float* row = atab + (size_t)j * arow + t1 * KRP;
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk) {
          if (!S::fixed && kk > KR)
            break;
          Aj[u][kk] = row[kk];
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const bool live = s0 + u < nrad;
        const WinRec r = rr[u];
        const float fx = (float)(r.x - ox), fy = (float)(r.y - oy), fz = (float)(r.z - oz);
        const float d2 = dot3f(fx, fx, fy, fy, fz, fz) * unit2;
        const int t2 = (int)((unsigned)r.w >> kIdxBits);
        float d, dinv;
        dist_and_inv(d2, d, dinv);
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
        const float rcinv = m.uniform_rc ? m.rcinv_r : fast_rcp(rc);
        // a pair the exact test admitted can sit a rounding above rc here: the envelope is clamped there
        const float dc = d < rc ? d : rc;
        float fc, fcp;
        cutoff_fc_fcp(rcinv, dc, fc, fcp);
        float fn[S::KRM + 1], fnp[S::KRM + 1];
        if (S::fixed)
          basis_fn_fnp<S::KRM>(rcinv, dc, fc, fcp, fn, fnp);
        else
          basis_fn_fnp_rt(KR, rcinv, dc, fc, fcp, fn, fnp);
        float s12 = 0.0f, s21 = 0.0f;
        if (S::TS > 0) {
#pragma unroll
          for (int t = 0; t < TSM; ++t) {
            float a = 0.0f;
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              a = fmaf(fnp[kk], Aown[t][kk], a);
            if (TSM == 1 || t2 == t)
              s12 = a;
          }
        } else {
          const float* Ai = atab + (size_t)k * arow + t2 * KRP;
          for (int kk = 0; kk <= KR; ++kk)
            s12 = fmaf(fnp[kk], Ai[kk], s12);
        }
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk) {
          if (!S::fixed && kk > KR)
            break;
          s21 = fmaf(fnp[kk], Aj[u][kk], s21);
        }
        const float wgt = live ? dinv : 0.0f;
        const float fs = (s12 + s21) * wgt; // f12 - f21 = fs * r12
        const float bb = s21 * wgt;         // f21 = -bb * r12
        Fr[0] = fmaf(fs, fx, Fr[0]);
        Fr[1] = fmaf(fs, fy, Fr[1]);
        Fr[2] = fmaf(fs, fz, Fr[2]);
        const float bx = bb * fx, by = bb * fy, bz = bb * fz;
        W[0] = fmaf(-fx, bx, W[0]);
        W[1] = fmaf(-fy, by, W[1]);
        W[2] = fmaf(-fz, bz, W[2]);
        W[3] = fmaf(-fx, by, W[3]);
        W[4] = fmaf(-fx, bz, W[4]);
        W[5] = fmaf(-fy, bz, W[5]);
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
        cur[u] = nxt[u];
    }

    // ---- outputs, internal order ----
    double E = (double)b.pe_i[k];
    double Fd[3], Wd[9];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      Fd[d] = (double)(F[d] + Fr[d] * unit);
    float Wr[6];
#pragma unroll
    for (int d = 0; d < 6; ++d)
      Wr[d] = W[d] * unit2;
    Wd[0] = (double)(Wr[0] + Wa[0]);
    Wd[1] = (double)(Wr[1] + Wa[1]);
    Wd[2] = (double)(Wr[2] + Wa[2]);
    Wd[3] = (double)(Wr[3] + Wa[3]);
    Wd[4] = (double)(Wr[4] + Wa[4]);
    Wd[5] = (double)(Wr[5] + Wa[5]);
    Wd[6] = (double)(Wr[3] + Wa[6]);
    Wd[7] = (double)(Wr[4] + Wa[7]);
    Wd[8] = (double)(Wr[5] + Wa[8]);
    if (m.zbl_enabled) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        Fd[d] += (double)b.zbl[(int64_t)d * N + k];
#pragma unroll
      for (int d = 0; d < 6; ++d)
        Wd[d] += (double)b.zbl[(int64_t)(3 + d) * N + k];
      Wd[6] += (double)b.zbl[(int64_t)(3 + 3) * N + k];
      Wd[7] += (double)b.zbl[(int64_t)(3 + 4) * N + k];
      Wd[8] += (double)b.zbl[(int64_t)(3 + 5) * N + k];
      E += (double)b.zbl[(int64_t)9 * N + k];
    }
    double* __restrict__ fo = b.fo + k;
    fo[0] = E;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      fo[(int64_t)(kOutF + d) * N] = Fd[d];
#pragma unroll
    for (int d = 0; d < 9; ++d)
      fo[(int64_t)(kOutW + d) * N] = Wd[d];
  }
};

} // namespace nepmi
