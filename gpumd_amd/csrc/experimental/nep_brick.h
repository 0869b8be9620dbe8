// One force kernel per brick behind the radial pass (device only: gfx950): angular descriptor + ANN + partial angular forces
// (nep_fused.h) AND the scatter-form force assembly (nep_scatter.h) in one launch.
//
// Replaces find_descriptor's angular half + apply_ann_one_layer (nep.cu:549-659), find_partial_force_angular (nep.cu:774-861),
// find_force_ZBL (nep.cu:863-975), find_force_radial (nep.cu:661-772) and gpu_find_force_many_body (potential.cu:170-297).
// The separate kernels hand three arrays through HBM that exist only because the step is cut there: the partial forces f12
// (16 B per angular pair, written by one kernel and read by the next), the per-atom radial table (32 B per neighbour type) and,
// on output steps, the pair records a second time.  Every term of the scatter form is a function of the OWN atom's Fp and of
// positions in the brick's window, so one workgroup can go from the sums to the window accumulators without leaving the chip:
//
//   stage   the window's positions (12-byte rows) and the zeroed fixed-point accumulators in LDS, the model image (nep_fused.h)
//   per atom, two adjacent lanes (lane `part` owns the radial channels n = part mod 2):
//     sums s_{n,lm} over the compact angular records -> descriptor half -> ANN -> Fp, energy             (registers)
//     radial-table rows of both neighbour types, whole on both lanes (two half sums, one DPP add each)    (registers)
//     adjoint G in place of the sums; per angular pair f12 -> +f12 to the own sum, -f12 to the partner's LDS slot
//     own pair halves of the compact radial list: lane `part` walks the segment of neighbour type `part` with that type's row
//     own sum -> own LDS slot
//   halo rows out (ForceFoldBody adds every atom's rows, as for the scatter kernel)
//
// 512 threads per brick, two wavefronts per SIMD (the angular part's register table decides that): one workgroup per CU.
// Shapes with two register-resident types (the PbTe shapes); single-domain engines (no ghost levels); compact radial list.
//
// MEASURED (round 5, PbTe 1,024,000 atoms, same box; profiles/r5h_*): this kernel 0.953-0.962 ms against 0.441 + 0.247 ms for
// the fused angular kernel and the scatter kernel it replaces -- the step 1.45 ms instead of 1.15.  Ablations
// (NEPMI_BRK_ABL): without the radial walk 0.64 ms (the angular part alone takes 0.44 in its own kernel: +0.2 ms for the
// structure), the walk itself 0.32 ms (0.10-0.14 in the scatter kernel), the halo rows 0.03.  What it shows: the angular part's
// 230-register table holds the workgroup to two wavefronts per SIMD, i.e. ONE 512-thread workgroup per CU -- nothing overlaps
// its staging, its barrier tails and its halo rows (16 bricks per CU in sequence), and the radial walk, an LDS-latency loop that
// the scatter kernel runs at three workgroups per CU, gets a third of the wavefronts.  The HBM round trips this kernel removes
// (f12, the radial table: ~0.3 GB per step) are worth less than that.  Hence OFF by default (nepmi_engine_set_brick_force(e, 1)
// turns it on; tests/test_gpu_parity.py keeps it correct): the step stays cut where the register budget changes.
#pragma once
#include "nep_fused.h"
#include "nep_scatter.h"

namespace nepmi {

#ifndef NEPMI_BRK_ABL
#define NEPMI_BRK_ABL 0 // ablation builds (timings only): 1 no radial walk, 2 no halo rows
#endif

struct BrickLayout {
  int wmax, img_floats;
  __device__ __host__ int rows() const { return wmax + 4; }
  __device__ __host__ int off_pos() const { return 0; }
  __device__ __host__ int off_acc() const { return 12 * rows(); }
  __device__ __host__ int off_img() const { return (24 * rows() + 15) / 16 * 16; }
  __device__ __host__ int bytes() const { return off_img() + 4 * img_floats; }
};

template <class S>
struct BrickForceBody {
  ForceScatterBody<S> sc; // window stage, model, frozen word, halo rows, brick order
  AngularFusedBody<S> ang; // model, buffers, LDS image (img != nullptr)
};

// the partial force of an angular pair goes straight into the accumulators: +f12 to the own sum (registers), -f12 to the
// partner's LDS slot; lane 0 of the pair does it (both lanes hold the same f12 after the DPP adds)
struct BrickAngularSink {
  NEPMI_LDS(char)* wacc;
  const unsigned short* aslot; // + k
  int64_t N;
  int nang;
  bool out;
  int* Fi;
  float* Wa;
  float* big;
  unsigned next_slot;
  __device__ __forceinline__ void operator()(int a, int part, const F4& f, const F4& e)
  {
    const unsigned sl = next_slot;
    if (a + 1 < nang)
      next_slot = aslot[(int64_t)(a + 1) * N]; // in flight while the next pair is evaluated
    if (part != 0)
      return;
    *big = fmaxf(*big, fmaxf(fabsf(f.x), fmaxf(fabsf(f.y), fabsf(f.z))));
    const int ax = to_fixed(f.x * kScatterScale), ay = to_fixed(f.y * kScatterScale), az = to_fixed(f.z * kScatterScale);
    Fi[0] += ax;
    Fi[1] += ay;
    Fi[2] += az;
    NEPMI_LDS(int)* rj = (NEPMI_LDS(int)*)(wacc + row12(sl));
    lds_sub(rj, ax);
    lds_sub(rj + 1, ay);
    lds_sub(rj + 2, az);
    if (out) {
      Wa[0] -= e.x * f.x;
      Wa[1] -= e.y * f.y;
      Wa[2] -= e.z * f.z;
      Wa[3] -= e.x * f.y;
      Wa[4] -= e.x * f.z;
      Wa[5] -= e.y * f.z;
      Wa[6] -= e.y * f.x;
      Wa[7] -= e.z * f.x;
      Wa[8] -= e.z * f.y;
    }
  }
};

template <class S, bool OUT>
__device__ __forceinline__ void brick_force_atom(const BrickForceBody<S>& B, const int64_t brick, const int64_t k, const int part,
                                                 NEPMI_LDS(char)* lds, const BrickLayout lay)
{
  static_assert(S::TS == 2, "two register-resident neighbour types: one list segment per lane");
  using F = FusedShape<S>;
  constexpr int NLOC = F::NLOC, DPH = F::DPH, K = S::KRM;
  constexpr int KRPC = (S::KR + 1 + 3) / 4 * 4;
  const Bufs& b = B.sc.st.b;
  const ModelD& m = B.sc.m;
  const int64_t N = b.N;
  NEPMI_LDS(const char)* wpos = (NEPMI_LDS(const char)*)(lds + lay.off_pos());
  NEPMI_LDS(char)* wacc = (NEPMI_LDS(char)*)(lds + lay.off_acc());
  lds_cfloat_ptr img = (lds_cfloat_ptr)(lds + lay.off_img());

  const int t1 = b.posq[k].type;
  // ---- angular part: sums -> descriptor -> ANN -> Fp; rows of the radial table; adjoint; pairs -> accumulators ----
  // (only what the radial part needs of the ANN's result stays live across the pair loop: this lane's radial Fp components)
  float s[NLOC * kNumHarm], E, Fpr[F::NRH];
  {
    float Fp[DPH];
    B.ang.descriptor_and_ann(k, part, img, t1, s, Fp, E);
#pragma unroll
    for (int i = 0; i < F::NRH; ++i)
      Fpr[i] = Fp[i];
    B.ang.adjoint_in_place(part, Fp, s);
  }

  int Fi[3] = {0, 0, 0};
  float big = 0.0f;
  float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    const int nang = b.nn_angstep[k];
    BrickAngularSink sink{wacc, b.aslot + k, N, nang, OUT, Fi, Wa, &big, nang > 0 ? (unsigned)b.aslot[k] : 0u};
    const AngularForceBody<S> af{m, b, 1};
    af.template pairs_from_G<2>(k, part, img, t1, s, sink);
  }

  // ---- radial part: own halves of this lane's segment (force_scatter_atom's pair arithmetic) ----
  // ---- own record in the window frame, own LDS slot ----
  const int l = b.kcell[k] & 63;
  const int wc_own = ((l & 3) + 2) + 8 * (((l >> 2) & 3) + 2) + 64 * ((l >> 4) + 2);
  int ox, oy, oz;
  B.sc.st.cell_offset(0, 0, 0, (l & 3) + 2, ((l >> 2) & 3) + 2, (l >> 4) + 2, ox, oy, oz);
  const WinRec pr = b.prec[k];
  ox += pr.x;
  oy += pr.y;
  oz += pr.z;
  const int* tab = b.wtab + (brick * 512 + wc_own) * 2;
  const int own_slot = (tab[1] & 0xFFFF) + (int)(k - tab[0]);
  const float rc1 = m.rc_r[t1];
  const float unit = b.wg.unit;
  const float qs = unit * kScatterScale;

  // the first entries of this lane's segment of the compact radial list, requested now (their latency passes behind the
  // angular part): segment `part` = the neighbours of type `part` (front of ccode: type 0, back: type 1)
  const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
  const int n0 = b.nn_t0[k] < nrad ? b.nn_t0[k] : nrad;
  const int count = part == 0 ? n0 : nrad - n0;
  const int64_t stride = part == 0 ? N : -N;
  const unsigned short* __restrict__ q = b.ccode + k + (part == 0 ? (int64_t)0 : (int64_t)(b.MN_rad - 1) * N);
  auto load2 = [&](const unsigned short* at, unsigned& c0, unsigned& c1) __attribute__((always_inline)) {
    c0 = at[0];
    c1 = at[stride];
  };
  const int npairs = count >> 1;
  unsigned a0 = 0, a1 = 0, n0c = 0, n1c = 0, tail = 0;
  if (npairs > 0)
    load2(q, a0, a1);
  if (npairs > 1)
    load2(q + 2 * stride, n0c, n1c);
  if (count & 1)
    tail = q[(int64_t)(count - 1) * stride];
  q += 4 * stride;

  float row[KRPC]; // the radial-table row of this lane's neighbour type
  {
    float r0[KRPC], r1[KRPC];
    B.ang.radial_row(part, img, t1, 0, Fpr, r0);
    B.ang.radial_row(part, img, t1, 1, Fpr, r1);
#pragma unroll
    for (int kk = 0; kk < KRPC; ++kk)
      row[kk] = part == 0 ? r0[kk] : r1[kk];
  }
  f2 W2[6] = {bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f)};
  {
    f2 A[K + 1], Bk[K + 1], SA, rc2, ri2;
    {
      float sa = 0.0f;
#pragma unroll
      for (int kk = 0; kk <= K; ++kk) {
        A[kk] = bc2(row[kk]);
        Bk[kk] = bc2((float)kk * row[kk]);
        sa += row[kk];
      }
      SA = bc2(sa);
      const float rcp = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[part]) * 0.5f;
      rc2 = bc2(rcp);
      ri2 = bc2(m.uniform_rc ? m.rcinv_r : fast_rcp(rcp));
    }
    auto two_pairs = [&](const unsigned s0, const unsigned s1, const float w1) __attribute__((always_inline)) {
      const unsigned o0 = row12(s0), o1 = row12(s1);
      const I3 p0 = *(NEPMI_LDS(const I3)*)(wpos + o0);
      const I3 p1 = *(NEPMI_LDS(const I3)*)(wpos + o1);
      const f2 fx = mk2((float)(p0.x - ox), (float)(p1.x - ox));
      const f2 fy = mk2((float)(p0.y - oy), (float)(p1.y - oy));
      const f2 fz = mk2((float)(p0.z - oz), (float)(p1.z - oz));
      const f2 d2 = vfma(fz, fz, vfma(fy, fy, fx * fx)) * b.wg.unit2;
      float d0, d1, i0, i1;
      dist_and_inv(d2.x, d0, i0);
      dist_and_inv(d2.y, d1, i1);
      const f2 dc = mk2(d0 < rc2.x ? d0 : rc2.x, d1 < rc2.y ? d1 : rc2.y);
      f2 fc, fcp;
      cutoff_fc_fcp_v(ri2, dc, fc, fcp);
      const f2 dr = dc * ri2 - 1.0f;
      const f2 x = vfma(dr * 2.0f, dr, bc2(-1.0f));
      const f2 x2 = x * 2.0f;
      f2 tm2 = bc2(1.0f), tm1 = x;
      f2 u0 = bc2(1.0f), u1 = x2;
      f2 ST = vfma(x, A[1], SA + A[0]);
      f2 SU = Bk[1];
#pragma unroll
      for (int kk = 2; kk <= K; ++kk) {
        const f2 tk = vfma(x2, tm1, -tm2);
        tm2 = tm1;
        tm1 = tk;
        ST = vfma(tk, A[kk], ST);
        SU = vfma(u1, Bk[kk], SU);
        if (kk < K) {
          const f2 u2 = vfma(x2, u1, -u0);
          u0 = u1;
          u1 = u2;
        }
      }
      const f2 s12 = vfma(dr * ri2 * 2.0f * fc, SU, fcp * 0.5f * ST);
      big = fmaxf(big, fmaxf(fabsf(s12.x), fabsf(s12.y) * w1));
      const f2 g = s12 * mk2(i0 * qs, i1 * (qs * w1));
      const f2 gx = g * fx, gy = g * fy, gz = g * fz;
      if (OUT) {
        W2[0] = vfma(-fx, gx, W2[0]);
        W2[1] = vfma(-fy, gy, W2[1]);
        W2[2] = vfma(-fz, gz, W2[2]);
        W2[3] = vfma(-fx, gy, W2[3]);
        W2[4] = vfma(-fx, gz, W2[4]);
        W2[5] = vfma(-fy, gz, W2[5]);
      }
      const int ax = to_fixed(gx.x), ay = to_fixed(gy.x), az = to_fixed(gz.x);
      const int bx = to_fixed(gx.y), by = to_fixed(gy.y), bz = to_fixed(gz.y);
      Fi[0] += ax + bx;
      Fi[1] += ay + by;
      Fi[2] += az + bz;
      NEPMI_LDS(int)* r0 = (NEPMI_LDS(int)*)(wacc + o0);
      NEPMI_LDS(int)* r1 = (NEPMI_LDS(int)*)(wacc + o1);
      lds_sub(r0, ax);
      lds_sub(r0 + 1, ay);
      lds_sub(r0 + 2, az);
      lds_sub(r1, bx); // (the repeated entry of an odd end subtracts zero)
      lds_sub(r1 + 1, by);
      lds_sub(r1 + 2, bz);
    };
    for (int pr2 = 0; pr2 < (NEPMI_BRK_ABL == 1 ? 0 : npairs); pr2 += 2) {
      const unsigned x0 = a0, x1 = a1;
      if (pr2 + 2 < npairs)
        load2(q, a0, a1);
      two_pairs(x0, x1, 1.0f);
      if (pr2 + 1 < npairs) {
        const unsigned y0 = n0c, y1 = n1c;
        if (pr2 + 3 < npairs)
          load2(q + 2 * stride, n0c, n1c);
        two_pairs(y0, y1, 1.0f);
      }
      q += 4 * stride;
    }
    if (NEPMI_BRK_ABL != 1 && (count & 1))
      two_pairs(tail, tail, 0.0f);
  }
  {
    NEPMI_LDS(int)* ro = (NEPMI_LDS(int)*)(wacc + row12((unsigned)own_slot)); // (both lanes their part: integer adds commute)
    lds_add(ro, Fi[0]);
    lds_add(ro + 1, Fi[1]);
    lds_add(ro + 2, Fi[2]);
  }
  if (big >= b.scatter_limit) {
    scatter_range_trip(b);
    if (b.scatter_hard > 0.0f && big >= b.scatter_hard)
      scatter_range_hard(b);
  }
  if (part == 0)
    b.pe_i[k] = E; // (exact_virials / exports after the step)
  if (!OUT)
    return;
  // ---- energy and the own-half virial, internal order (the force comes from ForceFoldBody) ----
  float Wr[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    float v = (W2[d].x + W2[d].y) * (unit * (1.0f / kScatterScale));
    v += NEPMI_PAIR_XCHG(v); // the two segments
    Wr[d] = v;
  }
  if (part != 0)
    return;
  double Ed = (double)E;
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
  if (m.zbl_enabled) {
#pragma unroll
    for (int d = 0; d < 6; ++d)
      Wd[d] += (double)b.zbl[(int64_t)(3 + d) * N + k];
    Wd[6] += (double)b.zbl[(int64_t)(3 + 3) * N + k];
    Wd[7] += (double)b.zbl[(int64_t)(3 + 4) * N + k];
    Wd[8] += (double)b.zbl[(int64_t)(3 + 5) * N + k];
    Ed += (double)b.zbl[(int64_t)9 * N + k];
  }
  double* __restrict__ fo = b.fo + k;
  fo[0] = Ed;
#pragma unroll
  for (int d = 0; d < 9; ++d)
    fo[(int64_t)(kOutW + d) * N] = Wd[d];
}

constexpr int kBrickThreads = 2 * kWinThreads; // two lanes per atom

template <class S, bool OUT>
__global__ void __launch_bounds__(kBrickThreads) __attribute__((amdgpu_waves_per_eu(2)))
nepmi_brick_force_kernel(const BrickForceBody<S> body, const int64_t nbricks)
{
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.sc.frozen && *body.sc.frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wgi = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wgi >= nbricks)
    return;
  const int64_t brick = body.sc.first < 0 ? wgi : (int64_t)body.sc.st.b.brick_order[body.sc.first + wgi];
  const int tid = (int)threadIdx.x;
  const Bufs& b = body.sc.st.b;
  const BrickLayout lay{body.sc.st.lay.wmax, fused_lds_layout<S>(body.sc.m).total};
  {
    // staging: window positions (WinStage::stage_direct without the index | type word), cleared accumulators, the model image
    NEPMI_LDS(I3)* wp = (NEPMI_LDS(I3)*)(lds + lay.off_pos());
    const int* tab = b.wtab + brick * 1024;
    int bx, by, bz;
    body.sc.st.brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += kBrickThreads) {
      const int j0 = tab[2 * wc], pk = tab[2 * wc + 1];
      const int w0 = pk & 0xFFFF;
      int cnt = pk >> 16;
      if (w0 + cnt > lay.wmax)
        cnt = lay.wmax > w0 ? lay.wmax - w0 : 0;
      if (cnt == 0)
        continue;
      int qx, qy, qz;
      body.sc.st.cell_offset(bx, by, bz, wc & 7, (wc >> 3) & 7, wc >> 6, qx, qy, qz);
      for (int a = 0; a < cnt; a += 4) {
        WinRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          r[u] = b.prec[j0 + (a + u < cnt ? a + u : cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (a + u < cnt)
            wp[w0 + a + u] = I3{r[u].x + qx, r[u].y + qy, r[u].z + qz};
      }
    }
    NEPMI_LDS(U4)* a4 = (NEPMI_LDS(U4)*)(lds + lay.off_acc());
    const int n4 = 3 * lay.rows() / 4;
    if (tid == 0)
      wp[lay.wmax] = I3{0x38000000, 0x38000000, 0x38000000}; // the sentinel slot
    const U4 zero{0u, 0u, 0u, 0u};
    for (int i = tid; i < n4; i += kBrickThreads)
      a4[i] = zero;
    const F4f* __restrict__ src = reinterpret_cast<const F4f*>(body.ang.img);
    NEPMI_LDS(F4f)* d4 = (NEPMI_LDS(F4f)*)(lds + lay.off_img());
    for (int i = tid; i < (lay.img_floats + 3) / 4; i += kBrickThreads)
      d4[i] = src[i];
  }
  __syncthreads();
  int64_t a0, a1;
  body.sc.st.brick_range(brick, a0, a1);
  for (int64_t k = a0 + (tid >> 1); k < a1; k += kWinThreads)
    brick_force_atom<S, OUT>(body, brick, k, tid & 1, lds, lay);
  __syncthreads();
  {
    NEPMI_LDS(const I3)* acc = (NEPMI_LDS(const I3)*)(lds + lay.off_acc());
    I4* __restrict__ out = body.sc.halo + (size_t)brick * lay.wmax;
    for (int i = tid; i < (NEPMI_BRK_ABL == 2 ? 0 : lay.wmax); i += kBrickThreads) {
      const I3 v = acc[i];
      out[i] = I4{v.x, v.y, v.z, 0};
    }
  }
}

} // namespace nepmi
