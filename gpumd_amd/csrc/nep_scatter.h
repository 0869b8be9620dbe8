// Force assembly as an LDS-local scatter (device only: gfx950; no emulator twin -- tests/emu keeps the gather form).
//
// Replaces find_force_radial (nep.cu:661-772) + gpu_find_force_many_body (potential.cu:170-297) in the fused run loops.
// The gather form (nep_window.h: ForceWinBody) evaluates BOTH halves of every ordered pair on lane i,
//     F_i = sum_j (f12 - f21),   f12 from lane i's own rows,   f21 from rows GATHERED from neighbour j,
// and those per-lane gathers (two 16-byte loads of j's radial-table row per pair, the membership mask and the partial
// force of the partner per angular pair) are what binds it: 190 + 28 of 298 vector-memory instructions per wavefront, texture
// addresser 75 % busy (profiles/r3b_pmc_ta1.csv, r3z_pmc_sq1.csv).  Here lane i evaluates only its OWN half g = f12 of a pair
// (own table row, own partial angular forces: coalesced or register-resident operands) and
//     adds  +g to its own sum,   -g to the PARTNER's slot of an accumulator over the brick's 8x8x8-cell window in LDS
// -- every neighbour of a brick's atom is in that window by construction (it is where its position came from) -- which is
// the formulation of the reference's small-box kernels (nep_small_box.cuh:473-478: atomicAdd of -f12 to the neighbour), made
// LOCAL (LDS atomics, no global atomic) and DETERMINISTIC: the accumulators are 32-bit fixed point (2^-22 eV/A), integer adds
// commute, so the sums do not depend on the order the lanes arrive in and a second call is bit-identical; +g and -g are the
// same integer, so the total force is zero to the last bit.  The workgroup then writes its window accumulator to its own row
// of the halo buffer (plain coalesced 16-byte stores, {fx, fy, fz, 0} per window slot), and ForceFoldBody adds, for every atom,
// the entries of the (normally eight) windows its cell lies in -- tabulated per atom at the list rebuild (FoldMapBody), a fixed
// order, no atomics anywhere outside the LDS.
//
// LDS per workgroup: positions as {x, y} (8 B) + z (4 B) planes -- the index | type word of the 16-byte records of the other
// window kernels is not needed here (type-pure list segments; nothing is gathered by index) -- + three accumulator planes:
// 24 B per window atom, 49 KB for the 2,048-slot windows of PbTe 1 M atoms: three workgroups per CU.
//
// Range: a pair half beyond +-64 eV/A (|s12| or a partial angular force component) sets flags[kFlagRange]; the engine then
// returns to the gather form for good at its next look at the flags.  The accumulators wrap modulo 2^32 (two's complement),
// so only the NET sum of a window has to stay inside +-512 eV/A -- eight aligned pair halves of a size that has already
// tripped the flag.
//
// Per-atom virials: the own half gives W'_i = -sum_j r_ij (x) g_ij, whose SUM over the atoms is the reference's total
// (sum_i sum_j r_ij (x) f21 re-indexed) but whose per-atom attribution is not; the run loops need the total only
// (find_thermo) and the engine re-runs the gather form for the virial planes when per-atom virials leave the engine.
#pragma once
#include "nep_window.h"

namespace nepmi {

constexpr float kScatterScale = 4194304.0f;             // 2^22 fixed-point units per eV/A
constexpr double kScatterInvScale = 1.0 / 4194304.0;
constexpr float kScatterFlagLimit = 64.0f;              // eV/A per pair half: beyond it the engine leaves this form
constexpr unsigned kFoldNone = 0xFFFFFFFFu;             // unused entry of the fold map
constexpr int kFoldSlotBits = 13;                       // fold map entry = brick << 13 | slot (windows hold <= 5,000 atoms)

struct alignas(8) I2 {
  int x, y;
};
struct alignas(16) I4 {
  int x, y, z, w;
};

struct ScatterLayout {
  int wmax; // a multiple of 64; slot wmax = the sentinel of the other window kernels (never addressed here)
  __device__ __host__ int off_xy() const { return 0; }
  __device__ __host__ int off_z() const { return 8 * wmax; }
  __device__ __host__ int off_acc() const { return 12 * wmax; }
  __device__ __host__ int bytes() const { return 24 * wmax; }
};

template <class S>
struct ForceScatterBody {
  WinStage st; // lay.compact == 1 (static window layout)
  ModelD m;
  const int* frozen;
  I4* halo; // [brick][wmax] {fx, fy, fz, 0} in fixed point
};

__device__ __forceinline__ void lds_add(NEPMI_LDS(int)* p, int v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int to_fixed(float v) { return (int)__builtin_rintf(v); }

// one lane = one atom of the brick
template <class S>
__device__ __forceinline__ void force_scatter_atom(const ForceScatterBody<S>& B, const int64_t brick, const int64_t k,
                                                   NEPMI_LDS(char)* lds, const ScatterLayout lay)
{
  static_assert(S::TS > 0, "type-pure list segments (one or two types with register-resident rows)");
  const Bufs& b = B.st.b;
  const ModelD& m = B.m;
  const int64_t N = b.N;
  const int lv = b.lvl[k];
  double* __restrict__ fo = b.fo + k;
  if (lv < b.lvl_desc) {
    if (lv >= b.lvl_force) { // a reverse-mode ghost: it only collects what its owned neighbours scatter; no own terms
      fo[0] = 0.0;
#pragma unroll
      for (int d = 0; d < 9; ++d)
        fo[(int64_t)(kOutW + d) * N] = 0.0;
    }
    return;
  }
  NEPMI_LDS(const I2)* wxy = (NEPMI_LDS(const I2)*)(lds + lay.off_xy());
  NEPMI_LDS(const int)* wz = (NEPMI_LDS(const int)*)(lds + lay.off_z());
  NEPMI_LDS(int)* acc = (NEPMI_LDS(int)*)(lds + lay.off_acc());
  const int W = lay.wmax;
  // own record in the window frame and own LDS slot (the brick's cells are the 4x4x4 in the middle of the window)
  const int l = b.kcell[k] & 63;
  const int wc_own = ((l & 3) + 2) + 8 * (((l >> 2) & 3) + 2) + 64 * ((l >> 4) + 2);
  int ox, oy, oz;
  B.st.cell_offset(0, 0, 0, (l & 3) + 2, ((l >> 2) & 3) + 2, (l >> 4) + 2, ox, oy, oz);
  const WinRec pr = b.prec[k];
  ox += pr.x;
  oy += pr.y;
  oz += pr.z;
  const int t1 = (int)((unsigned)pr.w >> kIdxBits);
  const int* tab = b.wtab + (brick * 512 + wc_own) * 2;
  const int own_slot = (tab[1] & 0xFFFF) + (int)(k - tab[0]);
  const float rc1 = m.rc_r[t1];
  const float unit = b.wg.unit, unit2 = b.wg.unit2;
  const float qs = unit * kScatterScale; // grid-unit force coefficient -> fixed-point force
  const int KRP = b.KRP;
  const float* __restrict__ atab = b.atab + (size_t)k * (m.T * KRP);
  constexpr int TSM = S::TS;

  int Fi[3] = {0, 0, 0};  // own half, fixed point (the same integers the partners receive with the other sign)
  float big = 0.0f;       // largest |pair-half coefficient| (eV/A) met
  const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
  const unsigned short* __restrict__ ccode = b.ccode + k;
  f2 W2[6] = {bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f)}; // -sum r (x) g, grid units^2, xx yy zz xy xz yz

  // type-pure segments of the compact list (front: neighbours of type 0, back: of type 1): the own row of the segment's
  // type stays in registers, the pair cutoff is a constant of the segment
  const int n0 = b.nn_t0[k] < nrad ? b.nn_t0[k] : nrad;
#pragma unroll
  for (int t = 0; t < TSM; ++t) {
    float Aown[S::KRM + 1];
#pragma unroll
    for (int kk = 0; kk <= S::KRM; ++kk)
      Aown[kk] = atab[t * KRP + kk];
    const int count = t == 0 ? n0 : nrad - n0;
    const int row0 = t == 0 ? 0 : b.MN_rad - 1, step = t == 0 ? 1 : -1;
    const float rcp = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t]) * 0.5f;
    const float rip = m.uniform_rc ? m.rcinv_r : fast_rcp(rcp);
    auto load2 = [&](int s0, unsigned& c0, unsigned& c1) __attribute__((always_inline)) {
      const int i0 = s0 < count ? s0 : count - 1, i1 = s0 + 1 < count ? s0 + 1 : count - 1;
      c0 = ccode[(int64_t)(row0 + i0 * step) * N];
      c1 = ccode[(int64_t)(row0 + i1 * step) * N];
    };
    // list entries two chunks ahead of the arithmetic (2-byte coalesced loads: the only global latency of this loop)
    unsigned a0 = 0, a1 = 0, n0c = 0, n1c = 0, m0 = 0, m1 = 0;
    if (count > 0) {
      load2(0, a0, a1);
      load2(2, n0c, n1c);
    }
    for (int s0 = 0; s0 < count; s0 += 2) {
      if (s0 + 4 < count)
        load2(s0 + 4, m0, m1);
      // two pairs side by side (packed FP32)
      const bool live1 = s0 + 1 < count;
      const I2 p0 = wxy[a0], p1 = wxy[a1];
      const int z0 = wz[a0], z1 = wz[a1];
      const f2 fx = mk2((float)(p0.x - ox), (float)(p1.x - ox));
      const f2 fy = mk2((float)(p0.y - oy), (float)(p1.y - oy));
      const f2 fz = mk2((float)(z0 - oz), (float)(z1 - oz));
      const f2 d2 = vfma(fz, fz, vfma(fy, fy, fx * fx)) * unit2;
      float d0, d1, i0, i1;
      dist_and_inv(d2.x, d0, i0);
      dist_and_inv(d2.y, d1, i1);
      const f2 dc = mk2(d0 < rcp ? d0 : rcp, d1 < rcp ? d1 : rcp); // (a pair the exact test admitted can sit a rounding above rc)
      const f2 rcinv = bc2(rip);
      f2 fc, fcp;
      cutoff_fc_fcp_v(rcinv, dc, fc, fcp);
      f2 fnp[S::KRM + 1];
      basis_fnp_v<S::KRM>(rcinv, dc, fc, fcp, fnp);
      f2 s12 = bc2(0.0f);
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk)
        s12 = vfma(fnp[kk], bc2(Aown[kk]), s12);
      big = fmaxf(big, fmaxf(fabsf(s12.x), live1 ? fabsf(s12.y) : 0.0f));
      const f2 g = s12 * mk2(i0, live1 ? i1 : 0.0f); // own half of the pair force = g * r12 (r12 in grid units here)
      const f2 gx = g * fx, gy = g * fy, gz = g * fz;
      W2[0] = vfma(-fx, gx, W2[0]);
      W2[1] = vfma(-fy, gy, W2[1]);
      W2[2] = vfma(-fz, gz, W2[2]);
      W2[3] = vfma(-fx, gy, W2[3]);
      W2[4] = vfma(-fx, gz, W2[4]);
      W2[5] = vfma(-fy, gz, W2[5]);
      const int ax = to_fixed(gx.x * qs), ay = to_fixed(gy.x * qs), az = to_fixed(gz.x * qs);
      const int bx = to_fixed(gx.y * qs), by = to_fixed(gy.y * qs), bz = to_fixed(gz.y * qs);
      Fi[0] += ax + bx;
      Fi[1] += ay + by;
      Fi[2] += az + bz;
      lds_add(acc + a0, -ax);
      lds_add(acc + W + a0, -ay);
      lds_add(acc + 2 * W + a0, -az);
      if (live1) {
        lds_add(acc + a1, -bx);
        lds_add(acc + W + a1, -by);
        lds_add(acc + 2 * W + a1, -bz);
      }
      a0 = n0c;
      a1 = n1c;
      n0c = m0;
      n1c = m1;
    }
  }

  // ---- angular part: own partial forces f12 of this step's angular pairs (AngularForceBody wrote them) ----
  float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // -sum r12 (x) f12: xx yy zz xy xz yz yx zx zy
  if (!b.level || b.angf[k]) { // (an inner-ring ghost nobody asked for partial forces has none)
    const int nang = b.nn_angstep[k];
    const F4* __restrict__ acomp = b.acomp + k;
    const F4* __restrict__ f12o = b.f12 + k;
    const unsigned short* __restrict__ aslot = b.aslot + k;
    constexpr int C = 4;
    for (int a0 = 0; a0 < nang; a0 += C) {
      F4 e[C], fa[C];
      int sl[C];
#pragma unroll
      for (int u = 0; u < C; ++u) {
        const int aa = a0 + u < nang ? a0 + u : a0;
        e[u] = acomp[(int64_t)aa * N];
        fa[u] = f12o[(int64_t)aa * N];
        sl[u] = aslot[(int64_t)aa * N];
      }
#pragma unroll
      for (int u = 0; u < C; ++u) {
        if (a0 + u < nang) {
          big = fmaxf(big, fmaxf(fabsf(fa[u].x), fmaxf(fabsf(fa[u].y), fabsf(fa[u].z))));
          const int ax = to_fixed(fa[u].x * kScatterScale), ay = to_fixed(fa[u].y * kScatterScale),
                    az = to_fixed(fa[u].z * kScatterScale);
          Fi[0] += ax;
          Fi[1] += ay;
          Fi[2] += az;
          lds_add(acc + sl[u], -ax);
          lds_add(acc + W + sl[u], -ay);
          lds_add(acc + 2 * W + sl[u], -az);
          Wa[0] -= e[u].x * fa[u].x;
          Wa[1] -= e[u].y * fa[u].y;
          Wa[2] -= e[u].z * fa[u].z;
          Wa[3] -= e[u].x * fa[u].y;
          Wa[4] -= e[u].x * fa[u].z;
          Wa[5] -= e[u].y * fa[u].z;
          Wa[6] -= e[u].y * fa[u].x;
          Wa[7] -= e[u].z * fa[u].x;
          Wa[8] -= e[u].z * fa[u].y;
        }
      }
    }
  }
  lds_add(acc + own_slot, Fi[0]);
  lds_add(acc + W + own_slot, Fi[1]);
  lds_add(acc + 2 * W + own_slot, Fi[2]);
  if (big >= kScatterFlagLimit)
    atomicOr(&b.flags[kFlagRange], 1);

  // ---- outputs of this kernel, internal order: energy and the local-form virial (the force comes from ForceFoldBody) ----
  if (lv < b.lvl_force)
    return; // (a forward-mode ring ghost: its halves are delivered, its own outputs are nobody's)
  double E = lv >= 2 ? (double)b.pe_i[k] : 0.0;
  float Wr[6];
#pragma unroll
  for (int d = 0; d < 6; ++d)
    Wr[d] = (W2[d].x + W2[d].y) * unit2;
  double Wd[9];
  Wd[0] = (double)(Wr[0] + Wa[0]);
  Wd[1] = (double)(Wr[1] + Wa[1]);
  Wd[2] = (double)(Wr[2] + Wa[2]);
  Wd[3] = (double)(Wr[3] + Wa[3]);
  Wd[4] = (double)(Wr[4] + Wa[4]);
  Wd[5] = (double)(Wr[5] + Wa[5]);
  Wd[6] = (double)(Wr[3] + Wa[6]);
  Wd[7] = (double)(Wr[4] + Wa[7]);
  Wd[8] = (double)(Wr[5] + Wa[8]);
  if (m.zbl_enabled && lv >= 2) {
#pragma unroll
    for (int d = 0; d < 6; ++d)
      Wd[d] += (double)b.zbl[(int64_t)(3 + d) * N + k];
    Wd[6] += (double)b.zbl[(int64_t)(3 + 3) * N + k];
    Wd[7] += (double)b.zbl[(int64_t)(3 + 4) * N + k];
    Wd[8] += (double)b.zbl[(int64_t)(3 + 5) * N + k];
    E += (double)b.zbl[(int64_t)9 * N + k];
  }
  fo[0] = E;
#pragma unroll
  for (int d = 0; d < 9; ++d)
    fo[(int64_t)(kOutW + d) * N] = Wd[d];
}

#ifndef NEPMI_FS_WAVES
#define NEPMI_FS_WAVES 3
#endif
template <class S>
__global__ void __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(NEPMI_FS_WAVES)))
nepmi_force_scatter_kernel(const ForceScatterBody<S> body, const int64_t nbricks)
{
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.frozen && *body.frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t brick = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (brick >= nbricks)
    return;
  const int tid = (int)threadIdx.x;
  const ScatterLayout lay{body.st.lay.wmax};
  const Bufs& b = body.st.b;
  {
    // staging: the window cells' fixed-point records from Bufs::prec with the cell's offset from the window centre added
    // (WinStage::stage_direct without the index | type word), accumulators cleared
    NEPMI_LDS(I2)* wxy = (NEPMI_LDS(I2)*)(lds + lay.off_xy());
    NEPMI_LDS(int)* wz = (NEPMI_LDS(int)*)(lds + lay.off_z());
    const int* tab = b.wtab + brick * 1024;
    int bx, by, bz;
    body.st.brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += kWinThreads) {
      const int j0 = tab[2 * wc], pk = tab[2 * wc + 1];
      const int w0 = pk & 0xFFFF;
      int cnt = pk >> 16;
      if (w0 + cnt > lay.wmax)
        cnt = lay.wmax > w0 ? lay.wmax - w0 : 0;
      if (cnt == 0)
        continue;
      int qx, qy, qz;
      body.st.cell_offset(bx, by, bz, wc & 7, (wc >> 3) & 7, wc >> 6, qx, qy, qz);
      for (int a = 0; a < cnt; a += 4) {
        WinRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          r[u] = b.prec[j0 + (a + u < cnt ? a + u : cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (a + u < cnt) {
            wxy[w0 + a + u] = I2{r[u].x + qx, r[u].y + qy};
            wz[w0 + a + u] = r[u].z + qz;
          }
      }
    }
    NEPMI_LDS(U4)* a4 = (NEPMI_LDS(U4)*)(lds + lay.off_acc());
    const int n4 = 3 * lay.wmax / 4;
    const U4 zero{0u, 0u, 0u, 0u};
    for (int i = tid; i < n4; i += kWinThreads)
      a4[i] = zero;
  }
  __syncthreads();
  int64_t a0, a1;
  body.st.brick_range(brick, a0, a1);
  for (int64_t k = a0 + tid; k < a1; k += kWinThreads)
    force_scatter_atom<S>(body, brick, k, lds, lay);
  __syncthreads();
  {
    // the window sums, one 16-byte row per slot: what ForceFoldBody gathers
    NEPMI_LDS(const int)* acc = (NEPMI_LDS(const int)*)(lds + lay.off_acc());
    I4* __restrict__ out = body.halo + (size_t)brick * lay.wmax;
    const int W = lay.wmax;
    for (int i = tid; i < W; i += kWinThreads)
      out[i] = I4{acc[i], acc[W + i], acc[2 * W + i], 0};
  }
}

// Which windows hold atom k, and where: entry r of the fold map = brick << 13 | slot, or kFoldNone.  A cell lies in the
// window of brick q along one direction when its distance from q's first window cell (4 q - 2), taken modulo the number of
// cells in a periodic direction, is 0..7; q is the own brick or one of the two bricks either way (a partly filled last brick
// next to a periodic face puts a cell two bricks from a window that holds it).  Run at every list rebuild.
struct FoldMapBody {
  BoxD box;
  Bufs b;
  int wmax, rows;
  unsigned* fmap; // [rows][N]
  int* max_rows;  // device word: the largest number of windows any atom lies in
  __device__ void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int c = b.kcell[k];
    const int brick = c >> 6, l = c & 63;
    const int bx = brick % b.gbx, by = (brick / b.gbx) % b.gby, bz = brick / (b.gbx * b.gby);
    const int cc[3] = {4 * bx + (l & 3), 4 * by + ((l >> 2) & 3), 4 * bz + (l >> 4)};
    const int bb[3] = {bx, by, bz};
    const int nb[3] = {b.nbx, b.nby, b.nbz};
    const int gb[3] = {b.gbx, b.gby, b.gbz};
    int cand[3][5], wpos[3][5], nc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      nc[d] = 0;
      for (int o = -2; o <= 2; ++o) {
        int q = bb[d] + o;
        if (box.pbc[d])
          q = ((q % gb[d]) + gb[d]) % gb[d];
        else if (q < 0 || q >= gb[d])
          continue;
        bool seen = false;
        for (int i = 0; i < nc[d]; ++i)
          seen = seen || cand[d][i] == q;
        if (seen)
          continue;
        int w = cc[d] - (4 * q - 2);
        if (box.pbc[d])
          w = ((w % nb[d]) + nb[d]) % nb[d];
        if (w < 0 || w > 7)
          continue;
        cand[d][nc[d]] = q;
        wpos[d][nc[d]] = w;
        ++nc[d];
      }
    }
    int n = 0;
    for (int iz = 0; iz < nc[2]; ++iz)
      for (int iy = 0; iy < nc[1]; ++iy)
        for (int ix = 0; ix < nc[0]; ++ix) {
          const int64_t q = cand[0][ix] + (int64_t)b.gbx * (cand[1][iy] + (int64_t)b.gby * cand[2][iz]);
          const int wc = wpos[0][ix] + 8 * wpos[1][iy] + 64 * wpos[2][iz];
          const int* tab = b.wtab + (q * 512 + wc) * 2;
          const int r = (int)(k - tab[0]);
          if (r < 0 || r >= (tab[1] >> 16))
            continue;
          const int slot = (tab[1] & 0xFFFF) + r;
          if (slot >= wmax)
            continue;
          if (n < rows)
            fmap[(int64_t)n * N + k] = ((unsigned)q << kFoldSlotBits) | (unsigned)slot;
          ++n;
        }
    for (int r = n; r < rows; ++r)
      fmap[(int64_t)r * N + k] = kFoldNone;
    atomicMax(max_rows, n);
  }
};

// F_k = sum over the windows that hold k of their accumulator entry for k (+ the ZBL pair force of an owned atom)
struct ForceFoldBody {
  Bufs b;
  ModelD m;
  int wmax, rows;
  const unsigned* fmap;
  const I4* halo;
  __device__ void operator()(int64_t k) const
  {
    const int lv = b.lvl[k];
    if (lv < b.lvl_force)
      return;
    const int64_t N = b.N;
    int s0 = 0, s1 = 0, s2 = 0; // (modular: the net of a window is what has to fit)
    constexpr int G = 4;
    for (int r0 = 0; r0 < rows; r0 += G) {
      unsigned e[G];
#pragma unroll
      for (int u = 0; u < G; ++u)
        e[u] = r0 + u < rows ? fmap[(int64_t)(r0 + u) * N + k] : kFoldNone;
      I4 h[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        h[u] = I4{0, 0, 0, 0};
        if (e[u] != kFoldNone)
          h[u] = halo[(size_t)(e[u] >> kFoldSlotBits) * wmax + (e[u] & ((1u << kFoldSlotBits) - 1u))];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        s0 += h[u].x;
        s1 += h[u].y;
        s2 += h[u].z;
      }
    }
    double F[3] = {(double)s0 * kScatterInvScale, (double)s1 * kScatterInvScale, (double)s2 * kScatterInvScale};
    if (m.zbl_enabled && lv >= 2) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        F[d] += (double)b.zbl[(int64_t)d * N + k];
    }
    double* __restrict__ fo = b.fo + k;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      fo[(int64_t)(kOutF + d) * N] = F[d];
  }
};

} // namespace nepmi
