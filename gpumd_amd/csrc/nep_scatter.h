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
// LDS per workgroup: positions as 12-byte rows {x, y, z} -- the index | type word of the 16-byte records of the other window
// kernels is not needed here (type-pure list segments; nothing is gathered by index) -- + accumulators as 12-byte rows:
// 24 B per window atom, 49 KB for the 2,048-slot windows of PbTe 1 M atoms: three workgroups per CU.
//
// Range: a pair half beyond +-64 eV/A (|s12| or a partial angular force component) or a net force component beyond 128 eV/A
// sets flags[kFlagRange]; the engine then returns to the gather form for good -- a single-domain evaluation or loop step is
// re-run at once, a decomposed run hands over within a dozen steps (the flag travels with the skin vote) and treats a value
// beyond 256 eV/A met before that as an error (kOverflowRangeHard).  The accumulators wrap modulo 2^32 (two's complement), so
// only the NET sum of a window has to stay inside +-512 eV/A.
//
// Per-atom virials: the own half gives W'_i = -sum_j r_ij (x) g_ij, whose SUM over the atoms is the reference's total
// (sum_i sum_j r_ij (x) f21 re-indexed) but whose per-atom attribution is not; the run loops need the total only
// (find_thermo) and the engine re-runs the gather form for the virial planes when per-atom virials leave the engine.
#pragma once
#include <climits>
#include "nep_window.h"

#ifndef NEPMI_FS_MT_VEC
#define NEPMI_FS_MT_VEC 1 // many-type form: coefficient blocks padded to 4 (odd) floats and read with 16-byte ds_reads (r4j: UNEP-v1 1 M atoms, force assembly 1.02 -> 0.78 ms); 0 = element-wise, odd stride
#endif
#ifndef NEPMI_FS_ABL
#define NEPMI_FS_ABL 0 // ablation builds (profiles/ab_variants.sh; timings only, results are wrong): 1 no LDS atomics in the pair loop,
                       // 2 no pair loop, 3 no staging of the positions, 4 no angular part, 5 no halo rows written
#endif

namespace nepmi {

constexpr float kScatterScale = 4194304.0f;             // 2^22 fixed-point units per eV/A
constexpr double kScatterInvScale = 1.0 / 4194304.0;
constexpr float kScatterFlagLimit = 64.0f;              // eV/A per pair half: beyond it the engine leaves this form
constexpr int kFoldGuard = 1 << 29;                     // ... and per component of an atom's net force: 128 eV/A in fixed point
constexpr float kScatterHardLimit = 256.0f;             // decomposed runs: a pair half beyond this is an error, not a hand-over
constexpr int kFoldHard = 1 << 30;                      // ... and a net force component beyond 256 eV/A (the sums wrap at 512)
constexpr unsigned kFoldNone = 0xFFFFFFFFu;             // unused entry of the fold map
constexpr int kFoldSlotBits = 13;                       // fold map entry = brick << 13 | slot (windows hold <= 5,000 atoms)

struct alignas(8) I2 {
  int x, y;
};
struct alignas(16) I4 {
  int x, y, z, w;
};

struct I3 { // position of a window atom, fixed point, relative to the window centre
  int x, y, z;
};
struct ScatterLayout {
  int wmax; // a multiple of 64; row wmax = the sentinel slot the padded list words point at (far beyond every cutoff)
  // positions and accumulators as 12-byte rows [slot]{x, y, z}: one address (12 slot) serves the three words of either
  // (immediate offsets 0, 4, 8); 3 is coprime with the number of banks, so random slots spread over all of them
  __device__ __host__ int rows() const { return wmax + 4; } // (a multiple of four rows: 16-byte aligned planes)
  __device__ __host__ int off_pos() const { return 0; }
  __device__ __host__ int off_acc() const { return 12 * rows(); }
  __device__ __host__ int bytes() const { return 24 * rows(); }
};

template <class S>
struct ForceScatterBody {
  WinStage st; // lay.compact == 1 (static window layout)
  ModelD m;
  const int* frozen;
  I4* halo;  // [brick][wmax] {fx, fy, fz, 0} in fixed point
  int first; // workgroup w runs brick brick_order[first + w] (first < 0: brick w): the boundary bricks of a decomposed run first,
             // so that the ghosts' partial forces can travel while the interior bricks run (DistT, reverse-mode ghosts)
};

// a value beyond the guard band: the engine leaves this form (check_overflow); inside a run loop the step is frozen and re-run
__device__ __forceinline__ void scatter_range_trip(const Bufs& b)
{
  atomicOr(&b.flags[kFlagRange], 1);
  if (b.trip_tag)
    atomicCAS(&b.flags[kFlagMoved], 0, b.trip_tag);
}

// the value may have been beyond what the sums hold: where a flagged step stands (Bufs::scatter_hard != 0) that is an error
__device__ __forceinline__ void scatter_range_hard(const Bufs& b)
{
  atomicOr(&b.flags[kFlagOverflow], kOverflowRangeHard);
}

__device__ __forceinline__ void lds_add(NEPMI_LDS(int)* p, int v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_sub(NEPMI_LDS(int)* p, int v) // ds_sub_u32: the reaction needs no negation
{
  __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// float -> fixed point, round to nearest (ties towards +inf: floor(v + 0.5)) in one instruction; the value is added to one
// atom and subtracted from the other as the SAME integer, so the rounding never unbalances a pair
__device__ __forceinline__ unsigned row12(unsigned slot) // byte offset of a 12-byte row: two full-rate shift-adds (a 32-bit
{                                                         // v_mul_lo_u32 runs at a quarter of the rate)
  unsigned r;
  asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(slot));
  return r << 2;
}
__device__ __forceinline__ int to_fixed(float v)
{
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// one lane = one atom of the brick.  OUT: the step's energies and (own-half) virials are wanted -- a thermo record, the last
// step of a run loop, a per-call evaluation; the other steps of a run loop need forces only (the reference computes and
// stores all thirteen per-atom outputs every step, potential.cu:170-297; find_thermo reads them at the dump_thermo interval
// only): no virial arithmetic, no pair-vector loads in the angular part, no 80 bytes of stores per atom.
// MODE: how this step's radial pass left the list of pairs inside the cutoff -- 0 the slot-major compact list (Bufs::ccode),
// 1 inside bits over the packed Verlet words (Bufs::rmaskA / rmaskB), 2 wave-synchronous words (SyncFifo, Bufs::cword)
template <class S, bool OUT, int MODE>
__device__ __forceinline__ void force_scatter_atom(const ForceScatterBody<S>& B, const int64_t brick, const int64_t k,
                                                   NEPMI_LDS(char)* lds, const ScatterLayout lay)
{
  static_assert(S::TS > 0, "type-pure list segments (one or two types with register-resident rows)");
  constexpr bool MASK = MODE != 0; // places of weight zero exist (and touch the LDS neither for a position nor for the reaction)
  const Bufs& b = B.st.b;
  const ModelD& m = B.m;
  const int64_t N = b.N;
  const int lv = b.lvl[k];
  double* __restrict__ fo = b.fo + k;
  if (lv < b.lvl_desc) {
    if (OUT && lv >= b.lvl_force) { // a reverse-mode ghost: it only collects what its owned neighbours scatter; no own terms
      fo[0] = 0.0;
#pragma unroll
      for (int d = 0; d < 9; ++d)
        fo[(int64_t)(kOutW + d) * N] = 0.0;
    }
    return;
  }
  NEPMI_LDS(const char)* wpos = (NEPMI_LDS(const char)*)(lds + lay.off_pos());
  NEPMI_LDS(char)* wacc = (NEPMI_LDS(char)*)(lds + lay.off_acc());
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
  const float unit = b.wg.unit;
  const float qs = unit * kScatterScale; // grid-unit force coefficient -> fixed-point force
  const int KRP = b.KRP;
  const float* __restrict__ atab = b.atab + (size_t)k * (m.T * KRP);
  constexpr int TSM = S::TS;
  constexpr int K = S::KRM;

  int Fi[3] = {0, 0, 0};  // own half, fixed point (the same integers the partners receive with the other sign)
  float big = 0.0f;       // largest |pair-half coefficient| (eV/A) met
  const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
  // -sum r (x) g in units of (grid unit)^2 * qs (the fixed-point scale rides along; taken out at the end): xx yy zz xy xz yz
  f2 W2[6] = {bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f)};

  // operands of the first chunk of the angular part, requested now: their latency passes behind the radial loop
  constexpr int kAngChunk = 4;
  const bool ang_on = NEPMI_FS_ABL != 4 && (!b.level || b.angf[k]); // (an inner-ring ghost nobody asked for partial forces has none)
  const int nang = ang_on ? b.nn_angstep[k] : 0;
  const F4* __restrict__ acomp = b.acomp + k;
  const F4* __restrict__ f12o = b.f12 + k;
  const unsigned short* __restrict__ aslot = b.aslot + k;
  F4 fa_first[kAngChunk];
  int sl_first[kAngChunk];
#pragma unroll
  for (int u = 0; u < kAngChunk; ++u) {
    fa_first[u] = F4{0.0f, 0.0f, 0.0f, 0};
    sl_first[u] = 0;
    if (u < nang) {
      fa_first[u] = f12o[(int64_t)u * N];
      sl_first[u] = aslot[(int64_t)u * N];
    }
  }
  // Own half of a pair: s12 = sum_k A_k f_k'(r) with f_k' = (k U_{k-1}(x)) dx/dr fc/2 + (T_k(x) + 1) fc'/2
  // (find_fn_and_fnp, nep_utilities.cuh:590-623), contracted as  fc'/2 (SA + sum_k A_k T_k) + (dx/dr fc/2) sum_k (k A_k) U_{k-1}:
  // the Chebyshev recurrences feed two running sums instead of seven separate basis derivatives.  The own rows of BOTH
  // neighbour types stay in registers as f2 pairs (two pairs are evaluated side by side).
  struct Rows { // the own row of one neighbour type: A_k, k A_k, sum_k A_k as f2 pairs, the pair cutoff of that type
    f2 A[K + 1], B[K + 1], SA, rc, ri;
  };
  auto load_rows = [&](const int t, Rows& r) __attribute__((always_inline)) {
    float sa = 0.0f;
#pragma unroll
    for (int kk = 0; kk <= K; ++kk) {
      const float a = atab[t * KRP + kk];
      r.A[kk] = bc2(a);
      r.B[kk] = bc2((float)kk * a);
      sa += a;
    }
    r.SA = bc2(sa);
    const float rcp = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t]) * 0.5f;
    r.rc = bc2(rcp);
    r.ri = bc2(m.uniform_rc ? m.rcinv_r : fast_rcp(rcp));
  };
  // two pairs side by side (packed FP32): LDS slots a0 / a1, weights w0 / w1 (0: the place adds nothing), own rows Ax / Bx / SAx
  // and pair cutoffs rc2 / ri2 per half
  auto two_pairs = [&](const unsigned a0, const unsigned a1, const float w0, const float w1, const f2* Ax, const f2* Bx, const f2 SAx,
                       const f2 rc2, const f2 ri2) __attribute__((always_inline)) {
    const unsigned o0 = row12(a0), o1 = row12(a1);
    I3 p0, p1;
    if (MASK) {
      // a place of weight zero (a candidate outside the cutoff) touches the LDS neither for its position nor for the
      // reaction: it is evaluated at the sentinel's distance, where the envelope and its derivative vanish
      p0 = I3{ox + 0x38000000, oy, oz};
      p1 = p0;
      if (w0 != 0.0f)
        p0 = *(NEPMI_LDS(const I3)*)(wpos + o0);
      if (w1 != 0.0f)
        p1 = *(NEPMI_LDS(const I3)*)(wpos + o1);
    } else {
      p0 = *(NEPMI_LDS(const I3)*)(wpos + o0);
      p1 = *(NEPMI_LDS(const I3)*)(wpos + o1);
    }
    const f2 fx = mk2((float)(p0.x - ox), (float)(p1.x - ox));
    const f2 fy = mk2((float)(p0.y - oy), (float)(p1.y - oy));
    const f2 fz = mk2((float)(p0.z - oz), (float)(p1.z - oz));
    const f2 d2 = vfma(fz, fz, vfma(fy, fy, fx * fx)) * b.wg.unit2;
    float d0, d1, i0, i1;
    dist_and_inv(d2.x, d0, i0);
    dist_and_inv(d2.y, d1, i1);
    const f2 dc = mk2(d0 < rc2.x ? d0 : rc2.x, d1 < rc2.y ? d1 : rc2.y); // (a pair the exact test admitted can sit a rounding above rc)
    f2 fc, fcp;
    cutoff_fc_fcp_v(ri2, dc, fc, fcp);
    const f2 dr = dc * ri2 - 1.0f;
    const f2 x = vfma(dr * 2.0f, dr, bc2(-1.0f));
    const f2 x2 = x * 2.0f;
    f2 tm2 = bc2(1.0f), tm1 = x;
    f2 u0 = bc2(1.0f), u1 = x2; // U_0, U_1
    f2 ST = vfma(x, Ax[1], SAx + Ax[0]); // SA + A_0 T_0 + A_1 T_1
    f2 SU = Bx[1];                       // B_1 U_0
#pragma unroll
    for (int kk = 2; kk <= K; ++kk) {
      const f2 tk = vfma(x2, tm1, -tm2);
      tm2 = tm1;
      tm1 = tk;
      ST = vfma(tk, Ax[kk], ST);
      SU = vfma(u1, Bx[kk], SU); // k A_k U_{k-1}
      if (kk < K) {
        const f2 u2 = vfma(x2, u1, -u0);
        u0 = u1;
        u1 = u2;
      }
    }
    const f2 s12 = vfma(dr * ri2 * 2.0f * fc, SU, fcp * 0.5f * ST); // (dx/dr fc / 2 = 2 dr fc / rc)
    big = fmaxf(big, fmaxf(fabsf(s12.x) * w0, fabsf(s12.y) * w1));
    // own half of the pair force = g r12; here already in fixed-point units per grid unit of r12
    const f2 g = s12 * mk2(i0 * (qs * w0), i1 * (qs * w1));
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
    if (NEPMI_FS_ABL == 1)
      return;
    if (!MASK || w0 != 0.0f) {
      lds_sub(r0, ax);
      lds_sub(r0 + 1, ay);
      lds_sub(r0 + 2, az);
    }
    if (!MASK || w1 != 0.0f) {
      lds_sub(r1, bx); // (compact form: the repeated entry of an odd end subtracts zero)
      lds_sub(r1 + 1, by);
      lds_sub(r1 + 2, bz);
    }
  };

  if constexpr (MODE == 2) {
    // Wave-synchronous words: stream t = rows [t MN_cw, t MN_cw + rows_t) of Bufs::cword, rows_t the same on every lane of the
    // wavefront that wrote them (RadialWin2Body<S, 1>, the same atoms on the same lanes as here); a place holding the
    // sentinel slot is padding
    const int rw = b.nn_t0[k];
    const unsigned sent = (unsigned)b.wsent;
    auto wgt = [&](unsigned slot) __attribute__((always_inline)) -> float { return slot != sent ? 1.0f : 0.0f; };
#pragma unroll
    for (int t = 0; t < TSM; ++t) {
      const int rows = (rw >> (8 * t)) & 255;
      Rows R;
      load_rows(t, R);
      const U2w* __restrict__ wq = reinterpret_cast<const U2w*>(b.cword) + k + (int64_t)t * b.MN_cw * N;
      // requested two rows ahead of the arithmetic, unconditionally (the last rows are requested again: a conditional
      // load would put a full wait in front of every pair, see v3 / v4 in DESIGN section 5)
      const int last = rows > 0 ? rows - 1 : 0;
      U2w cur = wq[0], nxt = wq[(int64_t)(1 < last ? 1 : last) * N];
      for (int r = 0; r < (NEPMI_FS_ABL == 2 ? 0 : rows); ++r) {
        const U2w c = cur;
        cur = nxt;
        nxt = wq[(int64_t)(r + 2 < last ? r + 2 : last) * N];
        const unsigned s0 = c.lo & 0xFFFFu, s1 = c.lo >> 16, s2 = c.hi & 0xFFFFu, s3 = c.hi >> 16;
        two_pairs(s0, s1, wgt(s0), wgt(s1), R.A, R.B, R.SA, R.rc, R.ri);
        two_pairs(s2, s3, wgt(s2), wgt(s3), R.A, R.B, R.SA, R.rc, R.ri);
      }
    }
  } else if constexpr (!MASK) {
    // type-pure segments of the compact list (Bufs::ccode; front: neighbours of type 0, back: of type 1)
    const int n0 = b.nn_t0[k] < nrad ? b.nn_t0[k] : nrad;
#pragma unroll
    for (int t = 0; t < TSM; ++t) {
      const int count = t == 0 ? n0 : nrad - n0;
      Rows R;
      load_rows(t, R);
      // the segment's entries: row r of ccode at r N; the front segment walks rows 0, 1, ..., the back one MN_rad-1, MN_rad-2, ...
      const int64_t stride = t == 0 ? N : -N;
      const unsigned short* __restrict__ q = b.ccode + k + (t == 0 ? (int64_t)0 : (int64_t)(b.MN_rad - 1) * N);
      // The list entries (2-byte coalesced loads: the only global latency of this loop) are requested two chunks ahead of the
      // arithmetic, whole pairs unconditionally; the entry of an odd end is requested before the loop.
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
      // unrolled by two: the entries of chunk p + 2 are requested into the registers chunk p has just released -- no register
      // rotation, so nothing waits for a load before two chunks of arithmetic have passed
      for (int pr2 = 0; pr2 < (NEPMI_FS_ABL == 2 ? 0 : npairs); pr2 += 2) {
        const unsigned x0 = a0, x1 = a1;
        if (pr2 + 2 < npairs)
          load2(q, a0, a1);
        two_pairs(x0, x1, 1.0f, 1.0f, R.A, R.B, R.SA, R.rc, R.ri);
        if (pr2 + 1 < npairs) {
          const unsigned y0 = n0c, y1 = n1c;
          if (pr2 + 3 < npairs)
            load2(q + 2 * stride, n0c, n1c);
          two_pairs(y0, y1, 1.0f, 1.0f, R.A, R.B, R.SA, R.rc, R.ri);
        }
        q += 4 * stride;
      }
      if (count & 1)
        two_pairs(tail, tail, 1.0f, 0.0f, R.A, R.B, R.SA, R.rc, R.ri);
    }
  } else {
    // Mask form: the packed Verlet words (Bufs::wcode: four LDS slots per 8 bytes, segments padded with the sentinel slot) with
    // this step's inside bits as weights -- no compact list was written (Bufs::rmaskA / rmaskB).  One type-pure stream of words
    // per neighbour type: the entries of list A of that type (two-type shapes: Bufs::acode2, tabulated at the rebuild with
    // their list-A indices), then the rows of list B of that type.
    const int seg = b.wseg[k];
    const int wa = seg & 255, wb = (seg >> 8) & 255;
    const U2w* __restrict__ words = reinterpret_cast<const U2w*>(b.wcode) + k;
    const U2w* __restrict__ wbp = words + (int64_t)wa * N;
    auto wgt = [](unsigned bits, int i) __attribute__((always_inline)) -> float { return (float)((bits >> i) & 1u); };
    unsigned mA[3] = {0u, 0u, 0u}; // the inside bits of list A (lists longer than 96 entries: the rule keeps the compact form)
#pragma unroll
    for (int w = 0; w < 3; ++w)
      if (w < b.MAW)
        mA[w] = b.rmaskA[(int64_t)w * N + k];
    auto bitA = [&](unsigned idx) __attribute__((always_inline)) -> float {
      const unsigned w = idx >> 5;
      const unsigned m3 = w == 0 ? mA[0] : (w == 1 ? mA[1] : (w == 2 ? mA[2] : 0u));
      return (float)((m3 >> (idx & 31u)) & 1u);
    };
#pragma unroll
    for (int t = 0; t < TSM; ++t) {
      Rows R;
      load_rows(t, R);
      // list A entries of type t
      if (TSM == 2) {
        const int sg = b.aseg2[k];
        const int wat = t == 0 ? (sg & 255) : ((sg >> 8) & 255), base = t == 0 ? 0 : (sg & 255);
        const U2w* __restrict__ aw = reinterpret_cast<const U2w*>(b.acode2) + k + (int64_t)base * N;
        const unsigned* __restrict__ ao = b.aorig2 + k + (int64_t)base * N;
        U2w cur = {0u, 0u}, nxt = {0u, 0u};
        unsigned ic = 0u, in = 0u;
        if (wat > 0) {
          cur = aw[0];
          ic = ao[0];
        }
        for (int w = 0; w < (NEPMI_FS_ABL == 2 ? 0 : wat); ++w) {
          if (w + 1 < wat) {
            nxt = aw[(int64_t)(w + 1) * N];
            in = ao[(int64_t)(w + 1) * N];
          }
          two_pairs(cur.lo & 0xFFFFu, cur.lo >> 16, bitA(ic & 255u), bitA((ic >> 8) & 255u), R.A, R.B, R.SA, R.rc, R.ri);
          two_pairs(cur.hi & 0xFFFFu, cur.hi >> 16, bitA((ic >> 16) & 255u), bitA(ic >> 24), R.A, R.B, R.SA, R.rc, R.ri);
          cur = nxt;
          ic = in;
        }
      } else {
        U2w cur = {0u, 0u}, nxt = {0u, 0u};
        if (wa > 0)
          cur = words[0];
        for (int w = 0; w < (NEPMI_FS_ABL == 2 ? 0 : wa); ++w) {
          if (w + 1 < wa)
            nxt = words[(int64_t)(w + 1) * N];
          two_pairs(cur.lo & 0xFFFFu, cur.lo >> 16, bitA(4u * w), bitA(4u * w + 1u), R.A, R.B, R.SA, R.rc, R.ri);
          two_pairs(cur.hi & 0xFFFFu, cur.hi >> 16, bitA(4u * w + 2u), bitA(4u * w + 3u), R.A, R.B, R.SA, R.rc, R.ri);
          cur = nxt;
        }
      }
      // list B rows of type t: two-type shapes keep them as word pairs (row 2p: type 0, row 2p + 1: type 1)
      {
        const int64_t rstep = (int64_t)TSM * N;
        const U2w* __restrict__ bw = wbp + (int64_t)t * N;
        U2w cur = {0u, 0u}, nxt = {0u, 0u};
        if (wb > 0)
          cur = bw[0];
        unsigned mw = 0u;
        for (int p = 0; p < (NEPMI_FS_ABL == 2 ? 0 : wb); ++p) {
          if (p + 1 < wb)
            nxt = bw[(int64_t)(p + 1) * rstep];
          unsigned bits;
          if (TSM == 2) {
            if ((p & 3) == 0)
              mw = b.rmaskB[(int64_t)(p >> 2) * N + k];
            bits = mw >> (8 * (p & 3) + 4 * t);
          } else {
            if ((p & 7) == 0)
              mw = b.rmaskB[(int64_t)(p >> 3) * N + k];
            bits = mw >> (4 * (p & 7));
          }
          two_pairs(cur.lo & 0xFFFFu, cur.lo >> 16, wgt(bits, 0), wgt(bits, 1), R.A, R.B, R.SA, R.rc, R.ri);
          two_pairs(cur.hi & 0xFFFFu, cur.hi >> 16, wgt(bits, 2), wgt(bits, 3), R.A, R.B, R.SA, R.rc, R.ri);
          cur = nxt;
        }
      }
    }
  }

  // ---- angular part: own partial forces f12 of this step's angular pairs (AngularForceBody wrote them); the first chunk's
  //      operands were requested before the radial loop ----
  float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // -sum r12 (x) f12: xx yy zz xy xz yz yx zx zy
  if (ang_on) {
    for (int c0 = 0; c0 < nang; c0 += kAngChunk) {
      F4 e[kAngChunk], fa[kAngChunk];
      int sl[kAngChunk];
#pragma unroll
      for (int u = 0; u < kAngChunk; ++u) {
        if (c0 == 0) {
          fa[u] = fa_first[u];
          sl[u] = sl_first[u];
        } else {
          const int aa = c0 + u < nang ? c0 + u : c0;
          fa[u] = f12o[(int64_t)aa * N];
          sl[u] = aslot[(int64_t)aa * N];
        }
        if (OUT) {
          const int aa = c0 + u < nang ? c0 + u : c0;
          e[u] = acomp[(int64_t)aa * N];
        }
      }
#pragma unroll
      for (int u = 0; u < kAngChunk; ++u) {
        if (c0 + u < nang && sl[u] != b.wsent) { // (the sentinel slot: a padding row of the wave-synchronous records, no partial force)
          big = fmaxf(big, fmaxf(fabsf(fa[u].x), fmaxf(fabsf(fa[u].y), fabsf(fa[u].z))));
          const int ax = to_fixed(fa[u].x * kScatterScale), ay = to_fixed(fa[u].y * kScatterScale),
                    az = to_fixed(fa[u].z * kScatterScale);
          Fi[0] += ax;
          Fi[1] += ay;
          Fi[2] += az;
          NEPMI_LDS(int)* rj = (NEPMI_LDS(int)*)(wacc + row12((unsigned)sl[u]));
          lds_sub(rj, ax);
          lds_sub(rj + 1, ay);
          lds_sub(rj + 2, az);
          if (OUT) {
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
  }
  {
    NEPMI_LDS(int)* ro = (NEPMI_LDS(int)*)(wacc + row12((unsigned)own_slot));
    lds_add(ro, Fi[0]);
    lds_add(ro + 1, Fi[1]);
    lds_add(ro + 2, Fi[2]);
  }
  if (NEPMI_FS_ABL == 0 && big >= b.scatter_limit) {
    scatter_range_trip(b);
    if (b.scatter_hard > 0.0f && big >= b.scatter_hard)
      scatter_range_hard(b);
  }

  // ---- outputs of this kernel, internal order: energy and the local-form virial (the force comes from ForceFoldBody) ----
  if (!OUT || lv < b.lvl_force)
    return; // (a forward-mode ring ghost: its halves are delivered, its own outputs are nobody's)
  double E = lv >= 2 ? (double)b.pe_i[k] : 0.0;
  float Wr[6];
#pragma unroll
  for (int d = 0; d < 6; ++d)
    Wr[d] = (W2[d].x + W2[d].y) * (unit * (1.0f / kScatterScale)); // (grid unit)^2 qs -> eV
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
template <class S, bool OUT, int MODE, int NT = kWinThreads>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT > kWinThreads ? (NT + 255) / 256 : NEPMI_FS_WAVES)))
nepmi_force_scatter_kernel(const ForceScatterBody<S> body, const int64_t nbricks)
{
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.frozen && *body.frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wgi = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wgi >= nbricks)
    return;
  const int64_t brick = body.first < 0 ? wgi : (int64_t)body.st.b.brick_order[body.first + wgi];
  const int tid = (int)threadIdx.x;
  const ScatterLayout lay{body.st.lay.wmax};
  const Bufs& b = body.st.b;
  if (b.brick_live && !b.brick_live[brick])
    return; // (outer ghost ring of a decomposed run: no atom of the brick has descriptors; FoldMapBody leaves the brick out)
  {
    // staging: the window cells' fixed-point records from Bufs::prec with the cell's offset from the window centre added
    // (WinStage::stage_direct without the index | type word), accumulators cleared
    NEPMI_LDS(I3)* wp = (NEPMI_LDS(I3)*)(lds + lay.off_pos());
    const int* tab = b.wtab + brick * 1024;
    int bx, by, bz;
    body.st.brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < (NEPMI_FS_ABL == 3 ? 0 : kWinCells); wc += NT) {
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
          if (a + u < cnt)
            wp[w0 + a + u] = I3{r[u].x + qx, r[u].y + qy, r[u].z + qz};
      }
    }
    NEPMI_LDS(U4)* a4 = (NEPMI_LDS(U4)*)(lds + lay.off_acc());
    const int n4 = 3 * lay.rows() / 4;
    if (tid == 0)
      wp[lay.wmax] = I3{0x38000000, 0x38000000, 0x38000000}; // the sentinel slot: 0.875 R away along every axis
    const U4 zero{0u, 0u, 0u, 0u};
    for (int i = tid; i < n4; i += NT)
      a4[i] = zero;
  }
  __syncthreads();
  int64_t a0, a1;
  body.st.brick_range(brick, a0, a1);
  for (int64_t k = a0 + tid; k < a1; k += NT)
    force_scatter_atom<S, OUT, MODE>(body, brick, k, lds, lay);
  __syncthreads();
  {
    // the window sums, one 16-byte row per slot: what ForceFoldBody gathers
    NEPMI_LDS(const I3)* acc = (NEPMI_LDS(const I3)*)(lds + lay.off_acc());
    I4* __restrict__ out = body.halo + (size_t)brick * lay.wmax;
    for (int i = tid; i < (NEPMI_FS_ABL == 5 ? 0 : lay.wmax); i += NT) {
      const I3 v = acc[i];
      out[i] = I4{v.x, v.y, v.z, 0};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Many types (UNEP-v1: 16) and run-time shapes: the pair's coefficient block c[t_i][t_j] differs from lane to lane, so no
// type-pure segments and no register-resident rows.  The own half is contracted per pair from the coefficient table in LDS,
//   s12 = sum_n Fp_i[n] sum_k c[t_i][t_j][n][k] f_k'(r)
// (Fp_i: the atom's own radial Fp row, Bufs::fpr, in registers) -- half of what the gather form's FPJ variant does per pair
// (no c[t_j][t_i] block, no gather of the neighbour's Fp row).  The table (46 KB for UNEP-v1) + 24 B per window atom + a byte
// of type fill most of a CU's LDS: ONE workgroup of 256 L threads per CU, L = 4 adjacent lanes per atom (lane `sub` takes
// every L-th chunk of two pairs and every L-th angular pair; the scatter is atomic anyway, the own sums and the virial are
// added across the lanes with shuffles): 16 wavefronts per CU.
// ---------------------------------------------------------------------------------------------------------------------
struct ScatterLayoutMT {
  int wmax, ctab_floats;
  __device__ __host__ int off_pos() const { return 0; }
  __device__ __host__ int off_acc() const { return 12 * wmax; }
  __device__ __host__ int off_type() const { return 24 * wmax; }
  __device__ __host__ int off_ctab() const { return 25 * wmax; } // wmax is a multiple of 64
  __device__ __host__ int bytes() const { return 25 * wmax + 4 * ctab_floats; }
};

template <class S, bool OUT, int L, int MODE = 0>
__device__ __forceinline__ void force_scatter_atom_mt(const ForceScatterBody<S>& B, const int64_t brick, const int64_t k, const int sub,
                                                      NEPMI_LDS(char)* lds, const ScatterLayoutMT lay)
{
  const Bufs& b = B.st.b;
  const ModelD& m = B.m;
  const int64_t N = b.N;
  const int lv = b.lvl[k];
  double* __restrict__ fo = b.fo + k;
  if (lv < b.lvl_desc) {
    if (OUT && sub == 0 && lv >= b.lvl_force) {
      fo[0] = 0.0;
#pragma unroll
      for (int d = 0; d < 9; ++d)
        fo[(int64_t)(kOutW + d) * N] = 0.0;
    }
    return;
  }
  NEPMI_LDS(const char)* wpos = (NEPMI_LDS(const char)*)(lds + lay.off_pos());
  NEPMI_LDS(char)* wacc = (NEPMI_LDS(char)*)(lds + lay.off_acc());
  NEPMI_LDS(const unsigned char)* wtyp = (NEPMI_LDS(const unsigned char)*)(lds + lay.off_type());
  NEPMI_LDS(const float)* ctl = (NEPMI_LDS(const float)*)(lds + lay.off_ctab());
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
  const float unit = b.wg.unit;
  const float qs = unit * kScatterScale;
  const int NR = S::fixed ? S::NR : m.NR;
  const int KR = S::fixed ? S::KR : m.KR;
  const int cblk = ctab_block(NR, KR, NEPMI_FS_MT_VEC != 0);
  float Fpi[S::NRM + 1]; // the own radial Fp row
#pragma unroll
  for (int n = 0; n <= S::NRM; ++n)
    Fpi[n] = (S::fixed || n <= NR) ? b.fpr[(size_t)k * b.FPR + n] : 0.0f;

  int Fi[3] = {0, 0, 0};
  float big = 0.0f;
  const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
  float W[6] = {0, 0, 0, 0, 0, 0}; // -sum r (x) g in (grid unit)^2 qs: xx yy zz xy xz yz

  // operands of this lane's first angular pair, requested now
  const bool ang_on = NEPMI_FS_ABL != 4 && (!b.level || b.angf[k]);
  const int nang = ang_on ? b.nn_angstep[k] : 0;
  const F4* __restrict__ acomp = b.acomp + k;
  const F4* __restrict__ f12o = b.f12 + k;
  const unsigned short* __restrict__ aslot = b.aslot + k;

  auto one_pair = [&](const unsigned a0) __attribute__((always_inline)) {
    const unsigned o0 = row12(a0);
    const I3 p0 = *(NEPMI_LDS(const I3)*)(wpos + o0);
    const int t2 = wtyp[a0];
    const float fx = (float)(p0.x - ox), fy = (float)(p0.y - oy), fz = (float)(p0.z - oz);
    const float d2 = dot3f(fx, fx, fy, fy, fz, fz) * b.wg.unit2;
    float d, dinv;
    dist_and_inv(d2, d, dinv);
    const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
    const float rcinv = m.uniform_rc ? m.rcinv_r : fast_rcp(rc);
    const float dc = d < rc ? d : rc;
    float fc, fcp;
    cutoff_fc_fcp(rcinv, dc, fc, fcp);
    float fn[S::KRM + 1], fnp[S::KRM + 1];
    if (S::fixed)
      basis_fn_fnp<S::KRM>(rcinv, dc, fc, fcp, fn, fnp);
    else
      basis_fn_fnp_rt(KR, rcinv, dc, fc, fcp, fn, fnp);
    float g12[S::NRM + 1];
    if (NEPMI_FS_ABL == 6) { // (ablation: no table contraction)
#pragma unroll
      for (int n = 0; n <= S::NRM; ++n)
        g12[n] = fnp[n % (S::KRM + 1)];
    } else {
      ctab_contract<S, NEPMI_FS_MT_VEC != 0>(ctl + (t1 * m.T + t2) * cblk, NR, KR, fnp, g12);
    }
    float s12 = 0.0f;
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n) {
      if (!S::fixed && n > NR)
        break;
      s12 = fmaf(Fpi[n], g12[n], s12);
    }
    big = fmaxf(big, fabsf(s12));
    const float g = s12 * dinv * qs;
    const float gx = g * fx, gy = g * fy, gz = g * fz;
    if (OUT) {
      W[0] = fmaf(-fx, gx, W[0]);
      W[1] = fmaf(-fy, gy, W[1]);
      W[2] = fmaf(-fz, gz, W[2]);
      W[3] = fmaf(-fx, gy, W[3]);
      W[4] = fmaf(-fx, gz, W[4]);
      W[5] = fmaf(-fy, gz, W[5]);
    }
    const int ax = to_fixed(gx), ay = to_fixed(gy), az = to_fixed(gz);
    Fi[0] += ax;
    Fi[1] += ay;
    Fi[2] += az;
    NEPMI_LDS(int)* r0 = (NEPMI_LDS(int)*)(wacc + o0);
    if (NEPMI_FS_ABL == 1)
      return;
    lds_sub(r0, ax);
    lds_sub(r0 + 1, ay);
    lds_sub(r0 + 2, az);
  };
  if constexpr (MODE == 2) {
    // Wave-synchronous words (nep_window.h: SyncFifo; one stream): row r of Bufs::cword holds four entries of this atom's list --
    // one for each of its L = 4 lanes (the lanes of an atom read the same 8 bytes: one request); the sentinel slot is padding
    static_assert(L == 4, "one entry of every word per lane");
    const int rows = b.nn_t0[k] & 255;
    const unsigned sent = (unsigned)b.wsent;
    const U2w* __restrict__ wq = reinterpret_cast<const U2w*>(b.cword) + k;
    const int last = rows > 0 ? rows - 1 : 0;
    U2w cur = wq[0], nxt = wq[(int64_t)(1 < last ? 1 : last) * N];
    for (int r = 0; r < (NEPMI_FS_ABL == 2 ? 0 : rows); ++r) {
      const U2w c = cur;
      cur = nxt;
      nxt = wq[(int64_t)(r + 2 < last ? r + 2 : last) * N];
      const unsigned half = (sub & 2) ? c.hi : c.lo;
      const unsigned slot = (sub & 1) ? (half >> 16) : (half & 0xFFFFu);
      if (slot != sent)
        one_pair(slot);
    }
  } else {
    // this lane's entries: sub, sub + L, ...; the next one requested while the current one is evaluated
    const unsigned short* __restrict__ q = b.ccode + k + (int64_t)sub * N;
    const int64_t stride = (int64_t)L * N;
    unsigned cur = 0, nxt = 0;
    if (sub < nrad)
      cur = q[0];
    for (int s0 = sub; s0 < (NEPMI_FS_ABL == 2 ? 0 : nrad); s0 += L) {
      q += stride;
      if (s0 + L < nrad)
        nxt = q[0];
      one_pair(cur);
      cur = nxt;
    }
  }
  float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = sub; a < nang; a += L) {
    const F4 fa = f12o[(int64_t)a * N];
    const int sl = aslot[(int64_t)a * N];
    if (sl == b.wsent)
      continue; // a padding row of the wave-synchronous records
    big = fmaxf(big, fmaxf(fabsf(fa.x), fmaxf(fabsf(fa.y), fabsf(fa.z))));
    const int ax = to_fixed(fa.x * kScatterScale), ay = to_fixed(fa.y * kScatterScale), az = to_fixed(fa.z * kScatterScale);
    Fi[0] += ax;
    Fi[1] += ay;
    Fi[2] += az;
    NEPMI_LDS(int)* rj = (NEPMI_LDS(int)*)(wacc + row12((unsigned)sl));
    lds_sub(rj, ax);
    lds_sub(rj + 1, ay);
    lds_sub(rj + 2, az);
    if (OUT) {
      const F4 e = acomp[(int64_t)a * N];
      Wa[0] -= e.x * fa.x;
      Wa[1] -= e.y * fa.y;
      Wa[2] -= e.z * fa.z;
      Wa[3] -= e.x * fa.y;
      Wa[4] -= e.x * fa.z;
      Wa[5] -= e.y * fa.z;
      Wa[6] -= e.y * fa.x;
      Wa[7] -= e.z * fa.x;
      Wa[8] -= e.z * fa.y;
    }
  }
  {
    NEPMI_LDS(int)* ro = (NEPMI_LDS(int)*)(wacc + row12((unsigned)own_slot)); // (every lane its part: integer adds commute)
    lds_add(ro, Fi[0]);
    lds_add(ro + 1, Fi[1]);
    lds_add(ro + 2, Fi[2]);
  }
  if (NEPMI_FS_ABL == 0 && big >= b.scatter_limit) {
    scatter_range_trip(b);
    if (b.scatter_hard > 0.0f && big >= b.scatter_hard)
      scatter_range_hard(b);
  }
  if (!OUT)
    return;
#pragma unroll
  for (int msk = 1; msk < L; msk <<= 1) { // the L lanes of an atom are adjacent: a fixed tree
#pragma unroll
    for (int d = 0; d < 6; ++d)
      W[d] += NEPMI_SHFL_XOR(W[d], msk);
#pragma unroll
    for (int d = 0; d < 9; ++d)
      Wa[d] += NEPMI_SHFL_XOR(Wa[d], msk);
  }
  if (sub != 0 || lv < b.lvl_force)
    return;
  double E = lv >= 2 ? (double)b.pe_i[k] : 0.0;
  float Wr[6];
#pragma unroll
  for (int d = 0; d < 6; ++d)
    Wr[d] = W[d] * (unit * (1.0f / kScatterScale));
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

template <class S, bool OUT, int L, int MODE>
__global__ void __launch_bounds__(kWinThreads * L) nepmi_force_scatter_mt_kernel(const ForceScatterBody<S> body, const int64_t nbricks)
{
  constexpr int NT = kWinThreads * L;
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.frozen && *body.frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wgi = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wgi >= nbricks)
    return;
  const int64_t brick = body.first < 0 ? wgi : (int64_t)body.st.b.brick_order[body.first + wgi];
  const int tid = (int)threadIdx.x;
  const Bufs& b = body.st.b;
  const ModelD& m = body.m;
  if (b.brick_live && !b.brick_live[brick])
    return; // (outer ghost ring: no atom of the brick has descriptors; FoldMapBody leaves the brick out)
  const ScatterLayoutMT lay{body.st.lay.wmax, m.T * m.T * ctab_block(m.NR, m.KR, NEPMI_FS_MT_VEC != 0)};
  {
    NEPMI_LDS(I3)* wp = (NEPMI_LDS(I3)*)(lds + lay.off_pos());
    NEPMI_LDS(unsigned char)* wt = (NEPMI_LDS(unsigned char)*)(lds + lay.off_type());
    const int* tab = b.wtab + brick * 1024;
    int bx, by, bz;
    body.st.brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += NT) {
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
            wp[w0 + a + u] = I3{r[u].x + qx, r[u].y + qy, r[u].z + qz};
            wt[w0 + a + u] = (unsigned char)((unsigned)r[u].w >> kIdxBits);
          }
      }
    }
    NEPMI_LDS(U4)* a4 = (NEPMI_LDS(U4)*)(lds + lay.off_acc());
    const int n4 = 3 * lay.wmax / 4;
    const U4 zero{0u, 0u, 0u, 0u};
    for (int i = tid; i < n4; i += NT)
      a4[i] = zero;
    ctab_stage_padded(m, lds + lay.off_ctab(), tid, NT, NEPMI_FS_MT_VEC != 0);
  }
  __syncthreads();
  int64_t a0, a1;
  body.st.brick_range(brick, a0, a1);
  const int sub = tid % L;
  for (int64_t k = a0 + tid / L; k < a1; k += kWinThreads)
    force_scatter_atom_mt<S, OUT, L, MODE>(body, brick, k, sub, lds, lay);
  __syncthreads();
  {
    NEPMI_LDS(const I3)* acc = (NEPMI_LDS(const I3)*)(lds + lay.off_acc());
    I4* __restrict__ out = body.halo + (size_t)brick * lay.wmax;
    for (int i = tid; i < (NEPMI_FS_ABL == 5 ? 0 : lay.wmax); i += NT) {
      const I3 v = acc[i];
      out[i] = I4{v.x, v.y, v.z, 0};
    }
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
          if (b.brick_live && !b.brick_live[q])
            continue; // (a brick of the outer ghost ring: the scatter kernels skip it, its rows do not exist)
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
  int lv_lo, lv_hi; // only atoms with level in [lv_lo, lv_hi] (the ghosts first when their forces travel during the interior bricks)
  __device__ void operator()(int64_t k) const
  {
    const int lv = b.lvl[k];
    if (lv < b.lvl_force || lv < lv_lo || lv > lv_hi)
      return;
    const int64_t N = b.N;
    int s0 = 0, s1 = 0, s2 = 0; // (modular: the net of a window is what has to fit)
    unsigned g0 = 0u, g1 = 0u, g2 = 0u; // sum of |row| / 16 per component (up to 27 rows of < 2^31: no overflow): the shadow below
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
        g0 += (unsigned)(h[u].x < 0 ? -h[u].x : h[u].x) >> 4;
        g1 += (unsigned)(h[u].y < 0 ? -h[u].y : h[u].y) >> 4;
        g2 += (unsigned)(h[u].z < 0 ? -h[u].z : h[u].z) >> 4;
      }
    }
    // the sums are modular: the NET force of an atom has to fit.  A quarter of the range (128 eV/A) is the guard band of the
    // total, half of it (256 eV/A) the hard limit of runs whose flagged steps stand; the pair halves have their own (64 / 256).
    // A net beyond 768 eV/A made of rows that each fit would alias into the band: the shadow sums g = sum |row| / 16 bound every
    // partial sum of the fold, so while g stays below 2^31 / 16 the modular sum above IS the sum; beyond that the step is
    // flagged like a value outside the band (the rows of ordinary forces add up to a few dozen eV/A).  What is still not seen: a
    // wrap INSIDE one brick's accumulator -- eight or more aligned pair halves of nearly 64 eV/A each into one slot.
    const int a0 = s0 < 0 ? -s0 : s0, a1 = s1 < 0 ? -s1 : s1, a2 = s2 < 0 ? -s2 : s2;
    const bool shadow = g0 >= (1u << 27) || g1 >= (1u << 27) || g2 >= (1u << 27);
    if (shadow || a0 >= b.fold_guard || a1 >= b.fold_guard || a2 >= b.fold_guard || s0 == INT_MIN || s1 == INT_MIN || s2 == INT_MIN) {
      scatter_range_trip(b);
      if (b.fold_hard > 0 && (shadow || a0 >= b.fold_hard || a1 >= b.fold_hard || a2 >= b.fold_hard || s0 == INT_MIN || s1 == INT_MIN || s2 == INT_MIN))
        scatter_range_hard(b);
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
