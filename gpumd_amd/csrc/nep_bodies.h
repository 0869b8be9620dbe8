// Per-atom kernel bodies of the NEP force path (one work-item = one atom unless stated).
//
// Each body is a small functor: the HIP backend launches it as a gfx950 kernel
// (backend_hip.h: nepmi_kernel<BLOCK, Body>), the test-only emulator (tests/emu) runs the same
// functor in a host loop.  Bodies only see raw device pointers held in `Bufs`.
//
// Data-flow per force evaluation (internal, cell-sorted atom order):
//   CheckGather   caller x,y,z  -> posq (FP64 pos + type, 32 B/atom), skin-violation flag
//   RadialDesc    skin list + posq -> per-step radial pair stash (r12 + packed j/t2, 16 B/pair),
//                 radial descriptor components q[0..NR]
//   AngularDesc   angular skin list + posq -> angular pair stash, s_{n,lm} sums, q[NR+1..dim)
//   Ann           q -> U_i, Fp = dU/dq, and the per-atom radial force table A_i[t2][k]
//   AngularForce  s, Fp -> adjoint G; partial forces f12 = dU_i/dr_ij per angular pair (+ ZBL)
//   ForceAssemble radial pair forces from A_i/A_j, angular f12 - f21 via reverse slots,
//                 per-atom virial; scatter-adds into the caller's FP64 arrays.
// Reference kernels replaced: find_neighbor_list_large_box, find_descriptor, find_force_radial,
// find_partial_force_angular, gpu_find_force_many_body, find_force_ZBL (src/force/nep.cu:436-975,
// src/force/potential.cu:170-297).
#pragma once
#include <type_traits>

#include "nep_dev.h"

namespace nepmi {

struct alignas(16) F4 {
  float x, y, z;
  int w; // packed integer payload (never interpreted as a float)
};

// Fixed-point position record of the LDS-window kernels (nep_window.h): x, y, z in grid units of WinGeom::unit,
// relative to the corner of the atom's cell of the rebuild-time grid (global array Bufs::prec) or, once staged,
// relative to the centre of a brick's window (LDS); w = internal index | type << kIdxBits.
struct alignas(16) U4 { // four membership words of Bufs::amask
  unsigned x, y, z, w;
};

struct alignas(16) WinRec {
  int x, y, z;
  int w;
};

// geometry of the fixed-point frame, set at every list rebuild
struct WinGeom {
  double inv_unit;     // grid points per Angstrom
  double cell_frac[3]; // fractional width of a cell along each lattice direction
  int cv[9];           // the three cell edge vectors in grid units: cv[3 * c + d] = component c of edge d
  int sv[9];           // lattice vector minus (cells per direction) cell edges: what a periodic wrap adds on top
  float unit;          // Angstrom per grid point
  float unit2;         // unit^2
  float band;          // |d^2 - rc^2| below this (A^2): the list decision is retaken exactly
};

#if defined(__HIP_DEVICE_COMPILE__)
#define NEPMI_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
#define NEPMI_ATOMIC_OR(ptr, v) atomicOr((ptr), (v))
#define NEPMI_ATOMIC_MAX(ptr, v) atomicMax((ptr), (v))
#else
NEPMI_HD int host_fetch_add(int* p, int v) { int o = *p; *p = o + v; return o; }
#define NEPMI_ATOMIC_ADD(ptr, v) host_fetch_add((ptr), (v))
#define NEPMI_ATOMIC_OR(ptr, v) (*(ptr) |= (v))
#define NEPMI_ATOMIC_MAX(ptr, v) (*(ptr) = (*(ptr) > (v) ? *(ptr) : (v)))
#endif

// compile-time model shape; -1 = take the value from ModelD at run time (generic fallback).
// TS > 0: number of atom types handled with register-resident per-type accumulators.
template <int NR_, int KR_, int NA_, int KA_, int NL_, int TS_>
struct Shape {
  static constexpr int NR = NR_, KR = KR_, NA = NA_, KA = KA_, NL = NL_, TS = TS_;
  static constexpr bool fixed = NR_ >= 0;
  static constexpr int NRM = NR_ >= 0 ? NR_ : 19;
  static constexpr int KRM = KR_ >= 0 ? KR_ : 19;
  static constexpr int NAM = NA_ >= 0 ? NA_ : 19;
  static constexpr int KAM = KA_ >= 0 ? KA_ : 19;
  static constexpr int NLM = NL_ >= 0 ? NL_ : 14; // L = 1..8, 222, 1111 and the four extra 4-body rows
  static constexpr int kRows = NR_ >= 0 ? 6 : 14;  // invariant rows a kernel of this shape can meet
  static constexpr int DIMM = (NRM + 1) + (NAM + 1) * NLM;
};
using ShapeGeneric = Shape<-1, -1, -1, -1, -1, 0>;

enum FlagSlot {
  kFlagMoved = 0, kFlagOverflow = 1,
  kFlagRange = 2,   // the scatter-form force assembly met a pair half beyond its fixed-point guard band (nep_scatter.h); right
                    // behind the two words of the skin vote: a decomposed run reduces the three together (dist_impl.h: vote)
  kFlagMaxSkin = 3, kFlagMaxAng = 4,
  kFlagMaxWindow = 5, kFlagMaxBrick = 6, kFlagMaxCell = 7, kFlagNumBoundary = 8,
  kFlagOutlier = 9, // an atom sits more than half a cell outside the box along an open direction
  kFlagFoldRows = 10, // scratch word of FoldMapBody: the most windows an atom lies in
  kNumFlags = 12
};
// bits of flags[kFlagOverflow]: 1, 2, 4 list capacities; 8 non-finite coordinates; 16 a force beyond the HARD limit of the
// scatter form's fixed-point sums in a run whose flagged steps stand (decomposed runs)
constexpr int kOverflowRangeHard = 16;

constexpr int kAngRowPad = 8;   // rows beyond MN_acomp (see Bufs::MN_arows)
constexpr int kNullRecord = -1; // F4::w of a padding row of the angular records (nep_window.h: wave-synchronous rows): every walk skips it
struct Bufs {
  int64_t N;
  // cell list
  int nbx, nby, nbz;    // cells per direction (Box::get_num_bins)
  int gbx, gby, gbz;    // bricks (4x4x4 cells) per direction = ceil(nb / 4)
  double rc_inv_cell;
  float rc_skin_sq;     // float(double (rc_r_max+skin)^2), neighbor.cu:363
  float rc_askin_sq;    // (rc_a_max+skin)^2
  int* cell_count;      // [ncell+1] -> exclusive scan in place = cell_start
  int* cell_fill;       // [ncell]
  int* cell_ghost;      // [ncell] 1: the cell holds an atom that is not owned (level < 2)
  int* brick_live;      // [nbricks] decomposed runs (else nullptr): 1 = the brick holds an atom of level >= 1 -- the window kernels have
                        // nothing to do in the bricks of the outer ghost ring (their atoms only lend positions) and skip them unstaged
  int* brick_flag;      // [nbricks+1] 1: a ghost sits in the brick's 8x8x8-cell window; scanned in place
  int* brick_order;     // [nbricks] interior bricks first (ascending), then boundary bricks
  int* cid;             // [N] caller order
  int* perm;            // [N] internal k -> caller index
  PosQ* posq;           // [N]
  double* x0s;          // [3][N] internal order, positions at the last rebuild
  // Verlet lists (rebuilt when an atom moved > skin/2), internal indices, [slot][atom]:
  //   list A: candidates with d < rc_a + skin (angular AND radial), with reverse slots
  //   list B: the remaining radial candidates, rc_a + skin <= d < rc_r + skin
  int MN_skin, MN_ang, MN_acomp;
  int MN_arows; // rows of acomp / f12 / aslot that exist: MN_acomp + kAngRowPad (the wave-synchronous form of the angular records pads)
  int* nn_ang;   int* nl_ang;  unsigned short* rev_ang; // A: [MN_ang][N]
  int* nn_skin;  int* nl_skin;                          // B: [MN_skin][N]
  unsigned short* code_ang;  // A: window code (window cell << 7 | rank in cell) of each entry
  unsigned short* code_skin; // B: the same; used by the LDS-window radial pass
  // per step
  int* nn_rad;   F4* rstash;   // pair records (r12, j | t2 << 25 or -1) at rows [A slots | MN_ang + B slots]
  int* nn_angstep; F4* acomp;  // compacted angular pair records [MN_acomp][N]
  int* nn_angtrue;             // [N] angular neighbours behind the padded rows (wave-synchronous records: nn_angstep counts the rows)
  unsigned short* amap;        // [MN_ang][N]: A slot -> compact angular slot of this step, 0xFFFF = none
  F4* f12;                     // [MN_acomp][N] partial forces dU_i/dr_ij at compact slots
  float* q;    // [dim][N]
  float* fp;   // [dim][N]
  float* sbuf; // [(NA+1)*24][N]
  float* shi;  // [(NA+1)*56][N] sums of l = 5..8 (models with l_max_3body > 4 only, nep_highl.h), else nullptr
  float* atab; // [N][T*KRP]
  int KRP;
  float* ann_img; // [T][AnnMfmaShape::img_floats] MFMA-operand-order weight images, or null
  float* pe_i; // [N]
  float* zbl;  // [10][N] (fx fy fz, vxx vyy vzz vxy vxz vyz, pe) when zbl enabled
  int* flags;  // [kNumFlags]
  const signed char* level; // caller order, or nullptr: 2 owned, 1 inner ghost, 0 outer ghost
  signed char* lvl;         // [N] the same in internal order (sampled at list rebuild)
  // Which levels do what.  Default (forward-only ghosts, shell 2 (rc + skin)): descriptors, ANN and partial angular forces for
  // level >= 1, compact radial list and force assembly for level 2.  Reverse mode (shell rc + skin, every ghost level 1): the
  // former for owned atoms only, the latter for ghosts too -- a ghost's table rows, Fp rows and f12 rows stay zero (cleared at
  // the rebuild), so its force assembly yields exactly the halves -f21 its OWNED neighbours contribute, which the decomposed
  // driver adds to the owner's force (dist_impl.h: reverse exchange).
  signed char lvl_desc = 1, lvl_force = 2;
  signed char* angf;        // [N] 1 = this atom's partial angular forces f12 are needed: it is owned, or an inner-ring ghost with an
                            // owned atom in its list A (decided at the rebuild; nobody reads the f12 of the other ghosts)
  int* tperm;               // [N] atoms ordered by (chunk of 1024, type): the ANN kernel's work order
  int* tpos;                // [N] inverse of tperm: q and fp are stored in work order, [dim][tpos]
  int* tcount;              // [(nchunks * T) + 1] histogram / offsets of that order
  int* sh_ang;              // small-box path only: packed periodic-image shift of each list-A entry
  // Outputs of the force path in INTERNAL order, assigned (not accumulated): 13 planes of N doubles,
  // pe | fx fy fz | virial xx yy zz xy xz yz yx zx zy.  The per-call entry points add them to the caller's
  // arrays through perm (ScatterAddBody); the fused run loops integrate on them directly.
  double* fo;
  // per-step compact list of the LDS-window path: window slots of the pairs inside the radial cutoff, written by
  // the radial pass and walked by the force assembly (no out-of-cutoff candidates there), and the reverse
  // slot (in the neighbour's list A) of every compact angular slot
  int MN_rad;
  unsigned short* ccode; // [MN_rad][N]; two-type shapes: neighbours of type 0 from row 0 up, of type 1 from row MN_rad-1 down
  int* nn_t0;            // [N] entries at the front of ccode (all of them unless the shape has two types)
  unsigned short* aidx;  // [MN_acomp][N] reverse slot (rev_ang) of every compact angular slot's pair
  unsigned* amask;       // [N][4] membership bits of list A this step (bit i: entry i is an angular neighbour and has a compact
                         // slot = the number of set bits below it): what amap holds, in 16 bytes per atom that stay in L2,
                         // written by the one-lane window radial pass instead of one amap store per list-A entry
  int use_amask;         // 1: amask is valid this step (window kernels, one lane per atom, list A <= 128 entries)
  // integrator state in internal order while a fused run loop owns the step (positions live in posq)
  double* vi; // [3][N]
  double* mi; // [N]
  double* ui; // [3][N] unwrapped positions, or nullptr
  int* invp;  // [N] caller index -> internal index (inverse of perm), written with the state import; or nullptr
  // fixed-point records of the current positions (written wherever posq is written) and the cell of every atom
  WinRec* prec; // [N]
  int* kcell;   // [N] cell index (brick-major numbering) at the last rebuild
  WinGeom wg;
  // Static window layout (one-lane window kernels): which atom sits at which LDS slot of a brick's window is fixed between two
  // list rebuilds (cells keep their members), so it is tabulated once per rebuild instead of being scanned by every launch:
  //   wtab[(brick * 512 + wc) * 2 + {0, 1}] = first atom of window cell wc, LDS slot of its first atom | atoms << 16
  // and the Verlet entries are kept as LDS slots, four to an 8-byte word:
  //   wcode[(chunk * N + k) * 4 + u]: list A (its own order), then list B -- for two-type shapes first the neighbours of
  //   type 0 and of type 1 as word pairs (row 2p: four of type 0, row 2p + 1: four of type 1) -- every segment padded to
  //   whole words with the sentinel slot `wsent` (a record far outside every cutoff); wseg[k] = words of A | words (word
  //   pairs) of B << 8
  int* wtab;
  unsigned short* wcode;
  int* wseg;
  int wsent;      // sentinel slot = WinLayout::wmax (one record beyond the fullest window)
  int MN_wchunks; // rows of wcode
  // per step, static layout: the compact radial list as words of four LDS slots -- rows [0, MN_cw): neighbours of type 0 (or
  // all), rows [MN_cw, 2 MN_cw): neighbours of type 1 (two-type shapes); nn_t0 / nn_rad count entries as before
  unsigned short* cword;
  int MN_cw;
  // radial part of Fp per atom, contiguous ([N][FPR], FPR = n_r + 1 rounded up to a multiple of 4): what the force assembly of
  // many-type models gathers from the neighbour (ForceWinBody<..., FPJ>); written by the per-atom ANN kernel; else nullptr
  float* fpr;
  int FPR;
  int skip_atab; // 1: this step's force assembly contracts both halves of a pair from Fp rows (FPJ form): the ANN kernel need not
                 // form the radial table (T k_r' floats per atom)
  // scatter-form force assembly (nep_scatter.h, device only): LDS window slot of the partner of every compact angular slot,
  // written by the static-layout radial pass next to aidx; nullptr on backends without that form
  unsigned short* aslot; // [MN_acomp][N]
  // Mask form of the per-step radial list (scatter-form steps of the run loops, shapes with type-pure streams): instead of
  // compacting the pairs inside the cutoff into ccode -- one scattered 2-byte store per pair, a third of the radial pass's
  // time (profiles/r4m_ab_radial_stores.txt) -- the radial pass sets one bit per candidate of the packed Verlet words
  // (Bufs::wcode) and the force assembly walks those words with the bits as weights.
  //   rmaskA[w >> 3][k] bit 4 (w & 7) + u : candidate u of list-A word w is inside the radial cutoff
  //   rmaskB[...]                          : two-type shapes: word pair p, bit 8 (p & 3) + 4 t + u of word p >> 2 (t: type stream);
  //                                          one type: word w of list B, bit 4 (w & 7) + u of word w >> 3
  //   tmaskA (written at the rebuild)      : two-type shapes: bit s = entry s of list A is of type 1
  unsigned* rmaskA;
  unsigned* rmaskB;
  unsigned* tmaskA;
  int MAW, MBW;  // words per atom of rmaskA / tmaskA and of rmaskB
  // Two-type shapes, written at the rebuild for the mask-form force assembly: list A once more as two type-pure runs of words
  // (rows [0, wa0): entries of type 0, rows [wa0, wa0 + wa1): of type 1, padded with the sentinel slot), with the list-A index
  // of every entry (one byte each, 255 = padding) to find its bit in rmaskA; aseg2[k] = wa0 | wa1 << 8
  unsigned short* acode2; // [MA2][N][4]
  unsigned* aorig2;       // [MA2][N]
  int* aseg2;
  int MA2;
  int use_rmask; // 1: this step's radial pass writes the masks instead of ccode
  int use_csync; // 1: this step's radial pass writes the list as wave-synchronous words into cword (nep_window.h: SyncFifo);
                 //    nn_t0[k] = words of stream 0 | words of stream 1 << 8
  float scatter_limit; // guard band of the scatter-form assembly per pair half, eV/A (nep_scatter.h: kScatterFlagLimit)
  int fold_guard;      // ... and per component of an atom's net force, fixed point (kFoldGuard)
  // Decomposed runs (a flagged step stands until every rank has seen the vote): a pair half / net component beyond these is
  // an ERROR (flags[kFlagOverflow] |= kOverflowRangeHard) instead of a silent wrap of the 32-bit sums; 0: no hard limit (the
  // single-domain callers re-run a flagged evaluation in the gather form at once)
  float scatter_hard;
  int fold_hard;
  int trip_tag;  // != 0: a speculatively enqueued step of a single-domain run loop -- a force beyond the fixed-point guard band of the
                 // scatter-form assembly freezes the loop at this step like a skin trip (flags[kFlagMoved] = tag), and the host
                 // re-runs the step in the gather form
  int compact_all;       // 1: every atom with level >= 1 writes its compact radial list (the scatter form walks the lists of the
                         // atoms that have descriptors, the gather form those of the atoms that receive forces)
};

// planes of Bufs::fo
constexpr int kOutPe = 0, kOutF = 1, kOutW = 4, kOutPlanes = 13;

constexpr unsigned short kNoSlot = 0xFFFF;

// ------------------------------------------------------------------------------------------------
// streaming bodies on the caller's arrays
// ------------------------------------------------------------------------------------------------

// gpu_apply_pbc, force.cu:424-459
struct ApplyPbcBody {
  BoxD box;
  int64_t N;
  double* pos;
  NEPMI_HD void operator()(int64_t i) const
  {
    double x = pos[i], y = pos[N + i], z = pos[2 * N + i];
    wrap_position(box, x, y, z);
    pos[i] = x;
    pos[N + i] = y;
    pos[2 * N + i] = z;
  }
};

// initialize_properties, force.cu:314-333
struct ZeroPropsBody {
  int64_t N;
  double *pe, *force, *virial;
  NEPMI_HD void operator()(int64_t i) const
  {
    pe[i] = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      force[d * N + i] = 0.0;
#pragma unroll
    for (int d = 0; d < 9; ++d)
      virial[d * N + i] = 0.0;
  }
};

// gpu_average_properties, force.cu:461-480 (a division, as there: bit-identical averages)
struct AveragePropsBody {
  int64_t N;
  double denominator;
  double *pe, *force, *virial;
  NEPMI_HD void operator()(int64_t i) const
  {
    pe[i] /= denominator;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      force[d * N + i] /= denominator;
#pragma unroll
    for (int d = 0; d < 9; ++d)
      virial[d * N + i] /= denominator;
  }
};

// gpu_velocity_verlet, ensemble.cu:176-214 (+ optional fused wrap for step 1 of the fused loop)
struct VelocityVerletBody {
  int64_t N;
  double dt;
  int is_step1;
  int fuse_wrap;
  BoxD box;
  const double* mass;
  const double* force;
  double* pos;
  double* vel;
  double* unwrapped; // [3N] or nullptr: gpu_update_unwrapped_position, integrate.cu:312-372
  NEPMI_HD void operator()(int64_t i) const
  {
#pragma clang fp contract(off) // v + (a * half): two roundings, as the oracle computes it
    const double half = dt * 0.5;
    const double minv = 1.0 / mass[i];
    double v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double a = force[d * N + i] * minv;
      const double kick = a * half;
      v[d] = vel[d * N + i] + kick;
      vel[d * N + i] = v[d];
    }
    if (is_step1) {
      double r[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double drift = v[d] * dt;
        const double old = pos[d * N + i];
        r[d] = old + drift;
        if (unwrapped)
          unwrapped[d * N + i] += r[d] - old; // new - old of the un-wrapped drift, like the reference
      }
      if (fuse_wrap)
        wrap_position(box, r[0], r[1], r[2]);
#pragma unroll
      for (int d = 0; d < 3; ++d)
        pos[d * N + i] = r[d];
    }
  }
};

// one coordinate of gpu_operator_A (shared by the stepwise and the resident form: the same expression, the same rounding)
NEPMI_HD double half_drift_1(const double old, const double v, const double half) { return old + v * half; }

// gpu_operator_A of the BAOAB Langevin integrator (ensemble_bao.cu:224-250): half a drift
struct HalfDriftBody {
  int64_t N;
  double dt;
  double* pos;
  const double* vel;
  double* unwrapped; // as in VelocityVerletBody
  NEPMI_HD void operator()(int64_t i) const
  {
    const double half = dt * 0.5;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double old = pos[d * N + i];
      const double r = half_drift_1(old, vel[d * N + i], half);
      if (unwrapped)
        unwrapped[d * N + i] += r - old;
      pos[d * N + i] = r;
    }
  }
};

// Step n's second half-kick, step n+1's first half-kick + drift + wrap, and initialize_properties
// for step n+1 in ONE pass over the atoms (run_nve between thermo records): the three kernels it
// replaces read the same force and touch the same arrays.  The two half-kicks stay two separate
// rounded additions, so the trajectory is bit-identical to the unfused sequence.
struct VerletSeamBody {
  int64_t N;
  double dt;
  BoxD box;
  const double* mass;
  double* force;
  double* pos;
  double* vel;
  double* pe;
  double* virial;
  double* unwrapped; // [3N] or nullptr
  NEPMI_HD void operator()(int64_t i) const
  {
#pragma clang fp contract(off)
    const double half = dt * 0.5;
    const double minv = 1.0 / mass[i];
    double r[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double a = force[d * N + i] * minv;
      const double kick = a * half;
      const double v2 = vel[d * N + i] + kick; // gpu_velocity_verlet, second call of step n
      const double v1 = v2 + kick;             // first call of step n + 1 (same force)
      vel[d * N + i] = v1;
      const double drift = v1 * dt;
      const double old = pos[d * N + i];
      r[d] = old + drift;
      if (unwrapped)
        unwrapped[d * N + i] += r[d] - old;
      force[d * N + i] = 0.0;
    }
    wrap_position(box, r[0], r[1], r[2]);
#pragma unroll
    for (int d = 0; d < 3; ++d)
      pos[d * N + i] = r[d];
    pe[i] = 0.0;
#pragma unroll
    for (int d = 0; d < 9; ++d)
      virial[d * N + i] = 0.0;
  }
};

// gpu_berendsen_temperature, src/integrate/ensemble_ber.cu:70-86
struct BerendsenBody {
  int64_t N;
  double temperature, coupling; // target T, 1 / T_coup
  const double* thermo;         // thermo[0] = instantaneous T (find_thermo)
  double* vel;
  NEPMI_HD void operator()(int64_t i) const
  {
    const double factor = sqrt(1.0 + coupling * (temperature / thermo[0] - 1.0));
    vel[i] *= factor;
    vel[N + i] *= factor;
    vel[2 * N + i] *= factor;
  }
};

// Nose-Hoover chain (Ensemble_NHC, src/integrate/ensemble_nhc.cu:102-232): the reference copies the
// temperature to the host and integrates the 4-link chain there; here the chain lives in device
// memory (state = pos[4] | vel[4] | mas[4] | factor) and one work-item advances it, so a thermostat
// half-step is two launches and no host round trip.  Suzuki-Yoshida 7 x 4 (Tuckerman's weights).
constexpr int kNhcLinks = 4;
constexpr int kNhcStateSize = 3 * kNhcLinks + 1;
constexpr double kBoltzmann = 8.617343e-5; // src/utilities/common.cuh:22

struct NhcInitBody { // Ensemble_NHC::Ensemble_NHC, ensemble_nhc.cu:30-49
  int64_t N;
  double temperature, t_coup, dt;
  double* st;
  NEPMI_HD void operator()(int64_t i) const
  {
    if (i != 0)
      return;
    const double tau = dt * t_coup, kT = kBoltzmann * temperature;
    for (int m = 0; m < kNhcLinks; ++m) {
      st[m] = 0.0;
      st[kNhcLinks + m] = (m & 1) ? -1.0 : 1.0;
      st[2 * kNhcLinks + m] = kT * tau * tau;
    }
    st[2 * kNhcLinks] *= 3.0 * (double)N;
    st[3 * kNhcLinks] = 1.0;
  }
};

struct NhcChainBody { // nhc(), ensemble_nhc.cu:102-164, with Ek2 = T * 3N * k_B (ensemble_nhc.cu:189-191)
  int64_t N;
  double temperature, dt2_particle;
  const double* thermo; // thermo[0] = instantaneous T (find_thermo)
  double* st;
  const int* frozen = nullptr; // fused run loops: non-zero = a list rebuild is pending, the chain must not advance
  NEPMI_HD void operator()(int64_t i) const
  {
    if (i != 0 || (frozen && *frozen != 0))
      return;
    constexpr int M = kNhcLinks;
    double pos[M], vel[M], mas[M];
    for (int m = 0; m < M; ++m) {
      pos[m] = st[m];
      vel[m] = st[M + m];
      mas[m] = st[2 * M + m];
    }
    const double kT = kBoltzmann * temperature, dN = 3.0 * (double)N;
    double Ek2 = thermo[0] * dN * kBoltzmann;
    const double w[7] = {0.784513610477560, 0.235573213359357, -1.17767998417887, 1.31518632068391,
                         -1.17767998417887, 0.235573213359357, 0.784513610477560};
    const int n_respa = 4;
    double factor = 1.0;
    for (int n1 = 0; n1 < 7; ++n1) {
      const double dt2 = dt2_particle * w[n1] / n_respa, dt4 = dt2 * 0.5, dt8 = dt4 * 0.5;
      for (int n2 = 0; n2 < n_respa; ++n2) {
        double G = vel[M - 2] * vel[M - 2] / mas[M - 2] - kT;
        vel[M - 1] += dt4 * G;
        for (int m = M - 2; m >= 0; --m) {
          const double tmp = exp(-dt8 * vel[m + 1] / mas[m + 1]);
          G = m == 0 ? Ek2 - dN * kT : vel[m - 1] * vel[m - 1] / mas[m - 1] - kT;
          vel[m] = tmp * (tmp * vel[m] + dt4 * G);
        }
        for (int m = M - 1; m >= 0; --m)
          pos[m] += dt2 * vel[m] / mas[m];
        const double fl = exp(-dt2 * vel[0] / mas[0]);
        Ek2 *= fl * fl;
        factor *= fl;
        for (int m = 0; m < M - 1; ++m) {
          const double tmp = exp(-dt8 * vel[m + 1] / mas[m + 1]);
          G = m == 0 ? Ek2 - dN * kT : vel[m - 1] * vel[m - 1] / mas[m - 1] - kT;
          vel[m] = tmp * (tmp * vel[m] + dt4 * G);
        }
        G = vel[M - 2] * vel[M - 2] / mas[M - 2] - kT;
        vel[M - 1] += dt4 * G;
      }
    }
    for (int m = 0; m < M; ++m) {
      st[m] = pos[m];
      st[M + m] = vel[m];
    }
    st[3 * M] = factor;
  }
};

struct ScaleVelocityBody { // scale_velocity_global, ensemble.cu (gpu_scale_velocity)
  int64_t N;
  const double* factor;
  double* vel;
  NEPMI_HD void operator()(int64_t i) const
  {
    const double f = *factor;
    vel[i] *= f;
    vel[N + i] *= f;
    vel[2 * N + i] *= f;
  }
};

struct ScaleVelocityConstBody { // scale_velocity_global with a host-computed factor (BDP thermostat)
  int64_t N;
  double factor;
  double* vel;
  NEPMI_HD void operator()(int64_t i) const
  {
    vel[i] *= factor;
    vel[N + i] *= factor;
    vel[2 * N + i] *= factor;
  }
};

// ------------------------------------------------------------------------------------------------
// neighbour rebuild (find_cell_list + gpu_find_neighbor_ON1, neighbor.cu:42-215)
// ------------------------------------------------------------------------------------------------

// Cells are numbered brick-major: 4x4x4-cell bricks in x-fastest order, cells z,y,x inside a brick.
// Atoms are stored in cell order, so 64 consecutive atoms (one wavefront) sit in a compact ~4x4x1
// cell slab and their 5x5x5-cell neighbourhoods overlap almost completely: neighbour gathers
// (posq, A-table, f12) of a wavefront hit the same few KB, and a contiguous range of workgroups
// (what one XCD runs, see backend_hip) stays inside a few-MB working set of its private L2.
constexpr int kBrick = 4;
NEPMI_HD int cell_index(const Bufs& b, int cx, int cy, int cz)
{
  const int brick = ((cz >> 2) * b.gby + (cy >> 2)) * b.gbx + (cx >> 2);
  return (brick << 6) | ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
}
NEPMI_HD void cell_coords(const Bufs& b, int c, int& cx, int& cy, int& cz)
{
  const int brick = c >> 6, l = c & 63;
  const int bx = brick % b.gbx, by = (brick / b.gbx) % b.gby, bz = brick / (b.gbx * b.gby);
  cx = (bx << 2) | (l & 3);
  cy = (by << 2) | ((l >> 2) & 3);
  cz = (bz << 2) | (l >> 4);
}

// PosQ::pad: how many lattice vectors the stored (wrapped) position has jumped since the list rebuild, two bits
// per direction (two's complement: -1, 0, 1).  The LDS-window kernels undo the jump when they place an atom
// relative to its cell of the rebuild-time grid; everything that applies a minimum image per pair ignores it.
NEPMI_HD int pack_img(int n0, int n1, int n2) { return (n0 & 3) | ((n1 & 3) << 2) | ((n2 & 3) << 4); }
NEPMI_HD int img_of(int pad, int d) { return (((pad >> (2 * d)) & 3) ^ 2) - 2; }

// apply_mic (float) that also reports the lattice-vector multiples it removed
NEPMI_HD void mic_f_img(const BoxD& box, float& x, float& y, float& z, int& n0, int& n1, int& n2)
{
  const float* H = box.hf;
  n0 = n1 = n2 = 0;
  if (box.ortho) {
    if (box.pbc[0]) {
      const float L = H[0], hl = L * 0.5f;
      if (x < -hl) { x += L; n0 = -1; } else if (x > hl) { x -= L; n0 = 1; }
    }
    if (box.pbc[1]) {
      const float L = H[4], hl = L * 0.5f;
      if (y < -hl) { y += L; n1 = -1; } else if (y > hl) { y -= L; n1 = 1; }
    }
    if (box.pbc[2]) {
      const float L = H[8], hl = L * 0.5f;
      if (z < -hl) { z += L; n2 = -1; } else if (z > hl) { z -= L; n2 = 1; }
    }
  } else {
    float sx = dot3f(H[9], x, H[10], y, H[11], z);
    float sy = dot3f(H[12], x, H[13], y, H[14], z);
    float sz = dot3f(H[15], x, H[16], y, H[17], z);
    if (box.pbc[0]) { const float r = nearbyintf(sx); sx -= r; n0 = (int)r; }
    if (box.pbc[1]) { const float r = nearbyintf(sy); sy -= r; n1 = (int)r; }
    if (box.pbc[2]) { const float r = nearbyintf(sz); sz -= r; n2 = (int)r; }
    x = dot3f(H[0], sx, H[1], sy, H[2], sz);
    y = dot3f(H[3], sx, H[4], sy, H[5], sz);
    z = dot3f(H[6], sx, H[7], sy, H[8], sz);
  }
}

// Fixed-point record of atom k from its current (wrapped) position: the lattice-vector jumps since the rebuild are
// undone, the periodic image next to the atom's cell of the rebuild-time grid is taken (this also puts the atoms of
// the thin top layer that find_cell_id folds into cell 0 next to that cell), and the offset from the cell's corner
// is rounded to the grid.  FP64 once per atom and step; the window kernels then only add integers.
NEPMI_HD WinRec make_prec(const BoxD& box, const Bufs& b, int64_t k, const PosQ& p)
{
  const double* h = box.h;
  int cx, cy, cz;
  cell_coords(b, b.kcell[k], cx, cy, cz);
  double sx = h[9] * p.x + h[10] * p.y + h[11] * p.z;
  double sy = h[12] * p.x + h[13] * p.y + h[14] * p.z;
  double sz = h[15] * p.x + h[16] * p.y + h[17] * p.z;
  sx -= (double)img_of(p.pad, 0);
  sy -= (double)img_of(p.pad, 1);
  sz -= (double)img_of(p.pad, 2);
  const double fx = (double)cx * b.wg.cell_frac[0], fy = (double)cy * b.wg.cell_frac[1], fz = (double)cz * b.wg.cell_frac[2];
  if (box.pbc[0]) sx += nearbyint(fx + 0.5 * b.wg.cell_frac[0] - sx);
  if (box.pbc[1]) sy += nearbyint(fy + 0.5 * b.wg.cell_frac[1] - sy);
  if (box.pbc[2]) sz += nearbyint(fz + 0.5 * b.wg.cell_frac[2] - sz);
  sx -= fx;
  sy -= fy;
  sz -= fz;
  WinRec r;
  r.x = (int)nearbyint((h[0] * sx + h[1] * sy + h[2] * sz) * b.wg.inv_unit);
  r.y = (int)nearbyint((h[3] * sx + h[4] * sy + h[5] * sz) * b.wg.inv_unit);
  r.z = (int)nearbyint((h[6] * sx + h[7] * sy + h[8] * sz) * b.wg.inv_unit);
  r.w = (int)((unsigned)k | ((unsigned)p.type << kIdxBits));
  return r;
}

struct BinAtomsBody {
  BoxD box;
  Bufs b;
  const double* pos; // caller order
  NEPMI_HD void operator()(int64_t i) const
  {
    int cx, cy, cz;
    const double x = pos[i], y = pos[b.N + i], z = pos[2 * b.N + i];
    cell_of(box, x, y, z, b.rc_inv_cell, b.nbx, b.nby, b.nbz, cx, cy, cz);
    if (!(x - x == 0.0 && y - y == 0.0 && z - z == 0.0)) // NaN or infinity
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 8);
    if (!(box.pbc[0] && box.pbc[1] && box.pbc[2])) {
      // open directions: the edge cells absorb atoms outside the box; the fixed-point windows of the LDS-window
      // kernels only reach half a cell beyond it
      const double* h = box.h;
      const double f[3] = {(h[9] * x + h[10] * y + h[11] * z) * box.thickness[0] * b.rc_inv_cell,
                           (h[12] * x + h[13] * y + h[14] * z) * box.thickness[1] * b.rc_inv_cell,
                           (h[15] * x + h[16] * y + h[17] * z) * box.thickness[2] * b.rc_inv_cell};
      for (int d = 0; d < 3; ++d)
        if (!box.pbc[d] && (f[d] < -0.5 || f[d] > box.thickness[d] * b.rc_inv_cell + 0.5))
          b.flags[kFlagOutlier] = 1; // benign race: all writers store 1
    }
    const int c = cell_index(b, cx, cy, cz);
    b.cid[i] = c;
    NEPMI_ATOMIC_ADD(&b.cell_count[c], 1);
  }
};

// atoms of the fullest brick, from the per-cell counts BinAtomsBody left (before the scan); used to size the cells
struct BrickMaxBody {
  Bufs b;
  NEPMI_HD void operator()(int64_t brick) const
  {
    int n = 0;
    for (int c = 0; c < 64; ++c)
      n += b.cell_count[brick * 64 + c];
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxBrick], n);
  }
};

struct FillCellsBody {
  Bufs b; // cell_count already scanned (cell_start)
  NEPMI_HD void operator()(int64_t i) const
  {
    const int c = b.cid[i];
    const int slot = NEPMI_ATOMIC_ADD(&b.cell_fill[c], 1);
    b.perm[b.cell_count[c] + slot] = (int)i;
  }
};

// one work-item per cell: ascending caller index inside a cell => deterministic internal order
struct SortCellsBody {
  Bufs b;
  NEPMI_HD void operator()(int64_t c) const
  {
    const int lo = b.cell_count[c], hi = b.cell_count[c + 1];
    for (int a = lo + 1; a < hi; ++a) {
      const int v = b.perm[a];
      int p = a - 1;
      while (p >= lo && b.perm[p] > v) {
        b.perm[p + 1] = b.perm[p];
        --p;
      }
      b.perm[p + 1] = v;
    }
  }
};

// internal k <- caller perm[k]: packed position/type and the rebuild snapshot x0
struct GatherSortedBody {
  BoxD box;
  Bufs b;
  const double* pos;
  const int* type;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t i = b.perm[k];
    PosQ p;
    p.x = pos[i];
    p.y = pos[b.N + i];
    p.z = pos[2 * b.N + i];
    p.type = type[i];
    p.pad = 0;
    b.posq[k] = p;
    b.x0s[k] = p.x;
    b.x0s[b.N + k] = p.y;
    b.x0s[2 * b.N + k] = p.z;
    b.lvl[k] = b.level ? b.level[i] : (signed char)2;
    b.kcell[k] = b.cid[i];
    if (b.prec)
      b.prec[k] = make_prec(box, b, k, p);
  }
};

// gpu_find_neighbor_ON1 (neighbor.cu:85-162) in internal indices: because atoms are stored in
// cell order, the members of cell c are simply the index range [cell_start[c], cell_start[c+1]).
// The Verlet list (d < rc_r + skin) is written as two lists: A = also within rc_a + skin (the only
// pairs that can become angular neighbours before the next rebuild), B = the rest.
struct BuildListsBody {
  BoxD box;
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const PosQ p1 = b.posq[k];
    const int c = b.kcell[k];
    int cx, cy, cz;
    cell_coords(b, c, cx, cy, cz);
    // periodic directions wrap (>= 5 bins guaranteed), non-periodic ones stop at the box edge
    const int lx = b.nbx > 1 ? 2 : 0, ly = b.nby > 1 ? 2 : 0, lz = b.nbz > 1 ? 2 : 0;
    int cnta = 0, cntb = 0;
    bool near_owned = false; // an owned atom within rc_a + skin: only then can an owned atom ask for this atom's f12
    for (int kz = -lz; kz <= lz; ++kz) {
      int z2 = cz + kz;
      if (box.pbc[2]) { if (z2 < 0) z2 += b.nbz; else if (z2 >= b.nbz) z2 -= b.nbz; }
      else if (z2 < 0 || z2 >= b.nbz) continue;
      for (int ky = -ly; ky <= ly; ++ky) {
        int y2 = cy + ky;
        if (box.pbc[1]) { if (y2 < 0) y2 += b.nby; else if (y2 >= b.nby) y2 -= b.nby; }
        else if (y2 < 0 || y2 >= b.nby) continue;
        for (int kx = -lx; kx <= lx; ++kx) {
          int x2 = cx + kx;
          if (box.pbc[0]) { if (x2 < 0) x2 += b.nbx; else if (x2 >= b.nbx) x2 -= b.nbx; }
          else if (x2 < 0 || x2 >= b.nbx) continue;
          const int c2 = cell_index(b, x2, y2, z2);
          const int lo = b.cell_count[c2], hi = b.cell_count[c2 + 1];
          // window cell of this neighbour cell, relative to the atom's brick (8x8x8 cells)
          const int wc = (((cz & 3) + 2 + kz) << 6) | (((cy & 3) + 2 + ky) << 3) | ((cx & 3) + 2 + kx);
          for (int j = lo; j < hi; ++j) {
            if (j == k)
              continue;
            const PosQ p2 = b.posq[j];
            float x, y, z;
            const float d2 = pair_geometry(box, p1, p2, x, y, z);
            if (d2 < b.rc_skin_sq) {
              const unsigned short code = (unsigned short)((wc << 7) | ((j - lo) & 127));
              if (d2 < b.rc_askin_sq) {
                if (cnta < b.MN_ang) {
                  b.nl_ang[(int64_t)cnta * N + k] = j;
                  b.code_ang[(int64_t)cnta * N + k] = code;
                }
                ++cnta;
                near_owned = near_owned || b.lvl[j] >= 2;
              } else {
                if (cntb < b.MN_skin) {
                  b.nl_skin[(int64_t)cntb * N + k] = j;
                  b.code_skin[(int64_t)cntb * N + k] = code;
                }
                ++cntb;
              }
            }
          }
        }
      }
    }
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxSkin], cnta + cntb);
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxAng], cnta);
    if (cnta > b.MN_ang || cnta + cntb > b.MN_skin) {
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 1);
      cnta = cnta > b.MN_ang ? b.MN_ang : cnta;
      cntb = cntb > b.MN_skin ? b.MN_skin : cntb;
    }
    b.nn_ang[k] = cnta;
    b.nn_skin[k] = cntb;
    b.angf[k] = (b.lvl[k] >= 2 || (b.lvl[k] >= b.lvl_desc && near_owned)) ? 1 : 0;
  }
};

// Work order of the ANN kernel: atoms grouped by type inside chunks of 1024 consecutive atoms, so
// that a wavefront holds (almost always) one type and runs the network once, with that type's
// weights as scalar operands -- instead of once per type present (16x for the 16-metal UNEP model).
constexpr int kTypeChunkShift = 10;

// Shape of the matrix-core ANN kernel (engine.hip: nepmi_ann_mfma) for a model: MT 32-neuron row
// tiles, DT 32-row output tiles (descriptor rows + T*KRP radial-table rows), KS k-pairs over the
// descriptor.  img_floats = one type's weight image  [KS][MT][64] | [16 MT][DT][64] | b0[32 MT] | w1[32 MT].
struct AnnMfmaShape {
  int MT, DT, KS;
  size_t img_floats;
  bool ok;
};
inline AnnMfmaShape ann_mfma_shape(int T, int dim, int nneu, int KRP)
{
  AnnMfmaShape a;
  a.MT = (nneu + 31) / 32;
  // more than four types (UNEP-v1: 16): output rows = Fp only -- the radial-table rows (T KRP per atom) are what the force
  // assembly of such models does not read (it contracts from the atom's radial Fp row, Bufs::fpr); one workgroup then serves ONE
  // type over several chunks (engine.hip: nepmi_ann_mfma<.., BYTYPE>), operand buffer of 24 k-pairs
  const bool many = T > 4;
  a.DT = ((many ? dim : dim + T * KRP) + 31) / 32;
  a.KS = (dim + 1) / 2;
  a.img_floats = (size_t)(a.KS * a.MT + a.MT * 16 * a.DT) * 64 + 2 * a.MT * 32;
  a.ok = a.MT <= 4 && a.DT <= 4 && a.KS <= (many ? 24 : 40) && a.img_floats * sizeof(float) <= 144 * 1024;
  return a;
}
struct IdentityOrderBody { // q / fp columns in internal atom order (fused descriptor + ANN kernel)
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    b.tperm[k] = (int)k;
    b.tpos[k] = (int)k;
  }
};
struct TypeCountBody { // one work-item per (chunk, type): no atomics, deterministic
  Bufs b;
  int T;
  NEPMI_HD void operator()(int64_t key) const
  {
    const int64_t chunk = key / T;
    const int t = (int)(key - chunk * T);
    const int64_t k0 = chunk << kTypeChunkShift;
    const int64_t k1 = k0 + ((int64_t)1 << kTypeChunkShift) < b.N ? k0 + ((int64_t)1 << kTypeChunkShift) : b.N;
    int cnt = 0;
    for (int64_t k = k0; k < k1; ++k)
      cnt += (b.posq[k].type == t) ? 1 : 0;
    b.tcount[key] = cnt;
  }
};
struct TypeFillBody {
  Bufs b;
  int T;
  // rank of k among the same-type atoms of its chunk that precede it: the work order inside a type
  // group is ascending in k (deterministic, and neighbouring lanes touch neighbouring rows of q/Fp).
  // The scan index is wave-uniform, so the type reads are broadcasts.
  NEPMI_HD void operator()(int64_t k) const
  {
    const int t = b.posq[k].type;
    const int64_t k0 = (k >> kTypeChunkShift) << kTypeChunkShift;
    int rank = 0;
    for (int64_t kp = k0; kp < k; ++kp)
      rank += (b.posq[kp].type == t) ? 1 : 0;
    const int pos = b.tcount[(k >> kTypeChunkShift) * T + t] + rank;
    b.tperm[pos] = (int)k;
    b.tpos[k] = pos;
  }
};

// rev_ang[s][k] = slot of k in j's list A (the pair test is exactly symmetric).
// j's list is sorted by its window codes (BuildListsBody sweeps the cells in the order of the code's window-cell field, and a
// cell's atoms by index): the code k has in j's frame follows from the two cells, and a binary search over j's codes finds the
// slot in log2(n) two-byte probes -- the widening search from the mirrored slot that it replaces took ~15 four-byte probes per
// entry, each a cache line of its own (carbon, 77 entries per list: 6.1 ms per million atoms).
struct ReverseSlotsBody {
  BoxD box;
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int nn = b.nn_ang[k];
    if (nn == 0)
      return;
    const int ck = b.kcell[k];
    int kx, ky, kz;
    cell_coords(b, ck, kx, ky, kz);
    const int rank = (int)k - b.cell_count[ck];
    for (int s = 0; s < nn; ++s) {
      const int j = b.nl_ang[(int64_t)s * N + k];
      const int nj = b.nn_ang[j];
      int r = kNoSlot;
      if (rank < 128) {
        int jx, jy, jz;
        cell_coords(b, b.kcell[j], jx, jy, jz);
        int dx = kx - jx, dy = ky - jy, dz = kz - jz; // within +-2 cells, up to a periodic wrap
        if (box.pbc[0]) dx += dx > 2 ? -b.nbx : (dx < -2 ? b.nbx : 0);
        if (box.pbc[1]) dy += dy > 2 ? -b.nby : (dy < -2 ? b.nby : 0);
        if (box.pbc[2]) dz += dz > 2 ? -b.nbz : (dz < -2 ? b.nbz : 0);
        const int wc = (((jz & 3) + 2 + dz) << 6) | (((jy & 3) + 2 + dy) << 3) | ((jx & 3) + 2 + dx);
        const unsigned target = (unsigned)((wc << 7) | rank);
        int lo = 0, hi = (nj < b.MN_ang ? nj : b.MN_ang) - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((unsigned)b.code_ang[(int64_t)mid * N + j] < target)
            lo = mid + 1;
          else
            hi = mid;
        }
        if (lo == hi && b.nl_ang[(int64_t)lo * N + j] == (int)k)
          r = lo;
      }
      if (r == kNoSlot) {
        // (cells of more than 127 atoms, whose ranks wrap in the code: the search from the mirrored slot, widening)
        const int g = nn > 1 ? (nj - 1) - (s * (nj - 1)) / (nn - 1) : 0;
        for (int w = 0; w < nj && r == kNoSlot; ++w) {
          const int lo = g - w, hi = g + w;
          if (lo >= 0 && lo < nj && b.nl_ang[(int64_t)lo * N + j] == (int)k)
            r = lo;
          else if (w > 0 && hi < nj && hi >= 0 && b.nl_ang[(int64_t)hi * N + j] == (int)k)
            r = hi;
        }
      }
      if (r == kNoSlot)
        NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 2);
      b.rev_ang[(int64_t)s * N + k] = (unsigned short)r;
    }
  }
};

// per brick: atoms in the brick, atoms in its 8x8x8-cell window, largest cell -- decides at rebuild
// whether the LDS-window radial pass can be used for this list generation
struct TileStatsBody {
  BoxD box;
  Bufs b;
  NEPMI_HD void operator()(int64_t brick) const
  {
    const int bx = (int)(brick % b.gbx), by = (int)((brick / b.gbx) % b.gby), bz = (int)(brick / ((int64_t)b.gbx * b.gby));
    const int nbr = b.cell_count[brick * 64 + 64] - b.cell_count[brick * 64];
    int win = 0, mxc = 0, ghost = 0;
    for (int wz = 0; wz < 8; ++wz)
      for (int wy = 0; wy < 8; ++wy)
        for (int wx = 0; wx < 8; ++wx) {
          int cx = 4 * bx - 2 + wx, cy = 4 * by - 2 + wy, cz = 4 * bz - 2 + wz;
          bool ok = true;
          if (box.pbc[0]) cx = ((cx % b.nbx) + b.nbx) % b.nbx; else ok = ok && cx >= 0 && cx < b.nbx;
          if (box.pbc[1]) cy = ((cy % b.nby) + b.nby) % b.nby; else ok = ok && cy >= 0 && cy < b.nby;
          if (box.pbc[2]) cz = ((cz % b.nbz) + b.nbz) % b.nbz; else ok = ok && cz >= 0 && cz < b.nbz;
          if (!ok) { // a window cell beyond an open face of the box holds nothing (the running offset stays valid)
            if (b.wtab) {
              int* t = b.wtab + ((int64_t)brick * 512 + ((wz << 6) | (wy << 3) | wx)) * 2;
              t[0] = 0;
              t[1] = win & 0xFFFF;
            }
            continue;
          }
          const int c = cell_index(b, cx, cy, cz);
          const int cnt = b.cell_count[c + 1] - b.cell_count[c];
          if (b.wtab) {
            int* t = b.wtab + ((int64_t)brick * 512 + ((wz << 6) | (wy << 3) | wx)) * 2;
            t[0] = b.cell_count[c];
            t[1] = (win & 0xFFFF) | (cnt << 16);
          }
          win += cnt;
          mxc = cnt > mxc ? cnt : mxc;
          ghost |= b.cell_ghost[c];
        }
    b.brick_flag[brick] = ghost;
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxWindow], win);
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxBrick], nbr);
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxCell], mxc);
  }
};

// Domain decomposition: cells that hold ghosts, and (after the in-place scan of brick_flag) the
// brick order "interior first": a brick is interior when no ghost sits in its window, i.e. its
// radial pass can run while the ghost positions of this step are still in flight.
struct MarkGhostCellsBody {
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    if (b.lvl[k] < 2)
      b.cell_ghost[b.kcell[k]] = 1; // benign race: all writers store 1
    if (b.lvl[k] >= 1 && b.brick_live)
      b.brick_live[b.kcell[k] >> 6] = 1;
  }
};
struct BrickOrderBody {
  Bufs b;
  int64_t nbricks;
  NEPMI_HD void operator()(int64_t brick) const
  {
    const int before = b.brick_flag[brick]; // boundary bricks with a smaller index
    const int nbound = b.brick_flag[nbricks];
    const bool boundary = b.brick_flag[brick + 1] != before;
    if (boundary)
      b.brick_order[(nbricks - nbound) + before] = (int)brick;
    else
      b.brick_order[brick - before] = (int)brick;
    if (brick == 0)
      b.flags[kFlagNumBoundary] = nbound;
  }
};

// Verlet entries as static LDS slots, packed four to a word (Bufs::wcode): one work-item per atom, at the rebuild, after
// TileStatsBody has tabulated the windows.  parts = 2: list B split by the neighbour's type (two-type shapes).
struct PackCodesBody {
  Bufs b;
  int parts;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t brick = b.kcell[k] >> 6;
    const int* tab = b.wtab + brick * 1024;
    unsigned short* out = b.wcode + k * 4;
    const int64_t row = N * 4; // halfwords per row of words
    int word = 0, fill = 0;
    unsigned short cur[4];
    auto put = [&](unsigned short slot) {
      cur[fill++] = slot;
      if (fill == 4) {
        if (word < b.MN_wchunks)
          for (int u = 0; u < 4; ++u)
            out[(int64_t)word * row + u] = cur[u];
        ++word;
        fill = 0;
      }
    };
    auto flush = [&]() { // pad the segment to a whole word
      while (fill != 0)
        put((unsigned short)b.wsent);
    };
    auto slot_of = [&](unsigned code) { return (unsigned short)((tab[(code >> 7) * 2 + 1] & 0xFFFF) + (code & 127u)); };
    const int na = b.nn_ang[k], nb = b.nn_skin[k];
    for (int s = 0; s < na; ++s)
      put(slot_of(b.code_ang[(int64_t)s * N + k]));
    flush();
    const int wa = word;
    if (b.tmaskA && parts == 2) { // the types of list A's entries (they do not change between rebuilds)
      for (int w = 0; w < b.MAW; ++w) {
        unsigned bits = 0u;
        for (int s = 32 * w; s < na && s < 32 * w + 32; ++s)
          bits |= (b.posq[b.nl_ang[(int64_t)s * N + k]].type != 0 ? 1u : 0u) << (s & 31);
        b.tmaskA[(int64_t)w * N + k] = bits;
      }
    }
    if (b.acode2 && parts == 2) { // list A as two type-pure runs of words for the mask-form force assembly
      int row2 = 0, wa0 = 0;
      for (int t = 0; t < 2; ++t) {
        unsigned short sl[4];
        unsigned ix = 0u;
        int f2 = 0;
        auto emit = [&]() {
          if (row2 < b.MA2) {
            for (int u = 0; u < 4; ++u)
              b.acode2[((int64_t)row2 * N + k) * 4 + u] = sl[u];
            b.aorig2[(int64_t)row2 * N + k] = ix;
          }
          ++row2;
          f2 = 0;
          ix = 0u;
        };
        for (int s2 = 0; s2 < na; ++s2) {
          if ((b.posq[b.nl_ang[(int64_t)s2 * N + k]].type != 0) != (t != 0))
            continue;
          sl[f2] = slot_of(b.code_ang[(int64_t)s2 * N + k]);
          ix |= (unsigned)(s2 < 255 ? s2 : 255) << (8 * f2);
          if (++f2 == 4)
            emit();
        }
        if (f2 != 0) {
          for (; f2 < 4; ++f2) {
            sl[f2] = (unsigned short)b.wsent;
            ix |= 255u << (8 * f2);
          }
          emit();
        }
        if (t == 0)
          wa0 = row2;
      }
      if (row2 > b.MA2 || na > 255)
        NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 1);
      b.aseg2[k] = wa0 | ((row2 - wa0) << 8);
    }
    int wb = 0;
    if (parts == 2) {
      // two type-pure streams, word by word side by side: row wa + 2p = four neighbours of type 0, row wa + 2p + 1 = four of
      // type 1; the shorter stream is padded with sentinel words
      int s0 = 0, s1 = 0; // cursors over list B: next entry of type 0 / of type 1
      // the types of the entries, looked up once (bit s: entry s is not of type 0); lists longer than the mask use the look-up
      unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull; // (four scalars: no dynamically indexed register array)
      const bool masked = nb <= 256;
      if (masked)
        for (int s = 0; s < nb; ++s) {
          const unsigned long long bit = (b.posq[b.nl_skin[(int64_t)s * N + k]].type != 0 ? 1ull : 0ull) << (s & 63);
          const int w = s >> 6;
          m0 |= w == 0 ? bit : 0ull;
          m1 |= w == 1 ? bit : 0ull;
          m2 |= w == 2 ? bit : 0ull;
          m3 |= w == 3 ? bit : 0ull;
        }
      auto other = [&](int s) -> bool {
        const int w = s >> 6;
        const unsigned long long m = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : m3));
        return ((m >> (s & 63)) & 1ull) != 0ull;
      };
      auto next_of = [&](int& s, int want) -> int {
        if (masked) {
          while (s < nb && other(s) != (want != 0))
            ++s;
        } else {
          while (s < nb && (b.posq[b.nl_skin[(int64_t)s * N + k]].type != 0) != (want != 0))
            ++s;
        }
        return s < nb ? s++ : -1;
      };
      for (;;) {
        int e0[4], e1[4];
        for (int u = 0; u < 4; ++u)
          e0[u] = next_of(s0, 0);
        for (int u = 0; u < 4; ++u)
          e1[u] = next_of(s1, 1);
        if (e0[0] < 0 && e1[0] < 0)
          break;
        for (int u = 0; u < 4; ++u)
          put(e0[u] >= 0 ? slot_of(b.code_skin[(int64_t)e0[u] * N + k]) : (unsigned short)b.wsent);
        for (int u = 0; u < 4; ++u)
          put(e1[u] >= 0 ? slot_of(b.code_skin[(int64_t)e1[u] * N + k]) : (unsigned short)b.wsent);
        ++wb;
      }
    } else {
      for (int s = 0; s < nb; ++s)
        put(slot_of(b.code_skin[(int64_t)s * N + k]));
      flush();
      wb = word - wa;
    }
    if (word > b.MN_wchunks || wa > 255 || wb > 255)
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 1);
    b.wseg[k] = wa | (wb << 8);
  }
};

// gpu_check_atom_distance (neighbor.cu:646-684) fused with the per-step gather of the caller's
// positions into internal order.  which: 0 all atoms, 1 owned only (level 2), 2 ghosts only.
struct CheckGatherBody {
  BoxD box;
  Bufs b;
  const double* pos;
  int which;
  NEPMI_HD void operator()(int64_t k) const
  {
#pragma clang fp contract(off) // d2 with separate roundings, as the oracle's skin check
    const int64_t N = b.N;
    if (which != 0 && (b.lvl[k] >= 2) != (which == 1))
      return;
    const int64_t i = b.perm[k];
    const double x = pos[i], y = pos[N + i], z = pos[2 * N + i];
    float dx = (float)(x - b.x0s[k]);
    float dy = (float)(y - b.x0s[N + k]);
    float dz = (float)(z - b.x0s[2 * N + k]);
    int n0, n1, n2;
    mic_f_img(box, dx, dy, dz, n0, n1, n2);
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!((double)d2 <= 0.25)) // skin^2/4, skin = 1 A (neighbor.cuh:212); also true for NaN
      NEPMI_ATOMIC_OR(&b.flags[kFlagMoved], 1);
    PosQ p = b.posq[k];
    p.x = x;
    p.y = y;
    p.z = z;
    p.pad = pack_img(n0, n1, n2);
    b.posq[k] = p;
    if (b.prec)
      b.prec[k] = make_prec(box, b, k, p);
  }
};

// ------------------------------------------------------------------------------------------------
// force path
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// small-box path (NEP::compute -> compute_small_box when a periodic thickness <= 2.5 (rc + 1),
// src/force/nep.cu:1295-1389, src/force/nep_small_box.cuh:37-132): every atom pair under every
// periodic image of the box replicated nc times, rebuilt at every call (no Verlet skin).  It emits
// the same pair records / lists as the large-box radial pass, so the angular kernels, the ANN and
// the force assembly are shared.  Internal order = caller order here.
// ------------------------------------------------------------------------------------------------

NEPMI_HD int pack_shift(int a, int b, int c) { return (a + 128) | ((b + 128) << 8) | ((c + 128) << 16); }

struct SmallBoxPairsBody {
  BoxD box;      // the cell
  float E[18];   // expanded cell (columns scaled by nc) and its inverse, float (nep.cu:1320-1350)
  int nc[3];
  ModelD m;
  Bufs b;
  const double* pos; // caller order
  const int* type;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const float* H = box.hf;
    // find_neighbor_list_small_box declares `float x1 = g_x[n1]` (nep_small_box.cuh:70-75)
    const float x1 = (float)pos[k], y1 = (float)pos[N + k], z1 = (float)pos[2 * N + k];
    const int t1 = type[k];
    PosQ p;
    p.x = pos[k];
    p.y = pos[N + k];
    p.z = pos[2 * N + k];
    p.type = t1;
    p.pad = 0;
    b.posq[k] = p;
    b.perm[k] = (int)k;
    b.tperm[k] = (int)k;
    b.tpos[k] = (int)k;
    b.lvl[k] = 2;
    b.angf[k] = 1;
    int cnta = 0, cntb = 0;
    for (int64_t j = 0; j < N; ++j) {
      const int t2 = type[j];
      const float rcr = (m.rc_r[t1] + m.rc_r[t2]) * 0.5f;
      const float rca = (m.rc_a[t1] + m.rc_a[t2]) * 0.5f;
      const double xj = pos[j], yj = pos[N + j], zj = pos[2 * N + j];
      for (int ia = 0; ia < nc[0]; ++ia)
        for (int ib = 0; ib < nc[1]; ++ib)
          for (int ic = 0; ic < nc[2]; ++ic) {
            if (ia == 0 && ib == 0 && ic == 0 && j == k)
              continue;
            const float d0 = dot3f(H[0], (float)ia, H[1], (float)ib, H[2], (float)ic);
            const float d1 = dot3f(H[3], (float)ia, H[4], (float)ib, H[5], (float)ic);
            const float d2v = dot3f(H[6], (float)ia, H[7], (float)ib, H[8], (float)ic);
            float x = (float)(xj + (double)d0 - (double)x1);
            float y = (float)(yj + (double)d1 - (double)y1);
            float z = (float)(zj + (double)d2v - (double)z1);
            // apply_mic_small_box (nep_small_box.cuh:37-54): nearest image of the expanded cell
            float sx = dot3f(E[9], x, E[10], y, E[11], z);
            float sy = dot3f(E[12], x, E[13], y, E[14], z);
            float sz = dot3f(E[15], x, E[16], y, E[17], z);
            int sh0 = ia, sh1 = ib, sh2 = ic;
            if (box.pbc[0]) { const float r = nearbyintf(sx); sx -= r; sh0 -= (int)r * nc[0]; }
            if (box.pbc[1]) { const float r = nearbyintf(sy); sy -= r; sh1 -= (int)r * nc[1]; }
            if (box.pbc[2]) { const float r = nearbyintf(sz); sz -= r; sh2 -= (int)r * nc[2]; }
            x = dot3f(E[0], sx, E[1], sy, E[2], sz);
            y = dot3f(E[3], sx, E[4], sy, E[5], sz);
            z = dot3f(E[6], sx, E[7], sy, E[8], sz);
            const float dd = dot3f(x, x, y, y, z, z);
            if (dd >= rcr * rcr)
              continue;
            F4 e;
            e.x = x;
            e.y = y;
            e.z = z;
            e.w = (int)((unsigned)j | ((unsigned)t2 << kIdxBits));
            if (dd < rca * rca) {
              if (cnta < b.MN_ang && cnta < b.MN_acomp) {
                b.nl_ang[(int64_t)cnta * N + k] = (int)j;
                b.sh_ang[(int64_t)cnta * N + k] = pack_shift(sh0, sh1, sh2);
                b.rstash[(int64_t)cnta * N + k] = e;
                b.acomp[(int64_t)cnta * N + k] = e;
                b.amap[(int64_t)cnta * N + k] = (unsigned short)cnta;
              }
              ++cnta;
            } else {
              if (cntb < b.MN_skin) {
                b.nl_skin[(int64_t)cntb * N + k] = (int)j;
                b.rstash[(int64_t)(b.MN_ang + cntb) * N + k] = e;
              }
              ++cntb;
            }
          }
    }
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxSkin], cnta + cntb);
    NEPMI_ATOMIC_MAX(&b.flags[kFlagMaxAng], cnta);
    if (cnta > b.MN_ang || cnta > b.MN_acomp || cntb > b.MN_skin) {
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 1); // list capacity (bit 8 is reserved for non-finite coordinates)
      cnta = cnta > b.MN_ang ? b.MN_ang : cnta;
      cnta = cnta > b.MN_acomp ? b.MN_acomp : cnta;
      cntb = cntb > b.MN_skin ? b.MN_skin : cntb;
    }
    b.nn_ang[k] = cnta;
    b.nn_skin[k] = cntb;
    b.nn_rad[k] = cnta + cntb;
    b.nn_angstep[k] = cnta;
  }
};

// reverse slot of (k -> j, shift s) is the entry (j -> k, shift -s) of j's list A
struct ReverseSlotsSmallBody {
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int nn = b.nn_ang[k];
    for (int s = 0; s < nn; ++s) {
      const int j = b.nl_ang[(int64_t)s * N + k];
      const int sh = b.sh_ang[(int64_t)s * N + k];
      const int a = (sh & 255) - 128, bb = ((sh >> 8) & 255) - 128, c = ((sh >> 16) & 255) - 128;
      const int want = pack_shift(-a, -bb, -c);
      const int nj = b.nn_ang[j];
      int r = kNoSlot;
      for (int s2 = 0; s2 < nj; ++s2)
        if (b.nl_ang[(int64_t)s2 * N + j] == (int)k && b.sh_ang[(int64_t)s2 * N + j] == want) {
          r = s2;
          break;
        }
      if (r == kNoSlot)
        NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 2);
      b.rev_ang[(int64_t)s * N + k] = (unsigned short)r;
    }
  }
};

constexpr int kGather = 4; // neighbour entries whose gathers are issued together
constexpr int kGatherR = 6; // the same for the radial pass: its chunk carries fewer live values (measured: 6 beats 2 and 4 there,
                            // 2 per lane beats 1 and 3 in the force assembly)

// c_ang staged in LDS: [T*T pairs][stride] with an odd stride so that lanes of different type
// pairs land on different banks (lanes of the same pair read one address: broadcast).
NEPMI_HD int cang_stride(const ModelD& m) { return ((m.NA + 1) * (m.KA + 1)) | 1; }
NEPMI_HD int cang_floats(const ModelD& m) { return m.T * m.T * cang_stride(m); }
NEPMI_HD void cang_stage(const ModelD& m, float* dst, int tid, int nth)
{
  const int per = (m.NA + 1) * (m.KA + 1), stride = cang_stride(m);
  if (m.cang_img) { // the engine's image of this layout (upload_model): a plain copy, no division per element
    const int n = cang_floats(m);
    for (int i = tid; i < n; i += nth)
      dst[i] = m.cang_img[i];
    return;
  }
  for (int idx = tid; idx < m.T * m.T * per; idx += nth) {
    const int pair = idx / per, r = idx - pair * per;
    dst[pair * stride + r] = m.c_ang[idx];
  }
}

// find_neighbor_list_large_box (nep.cu:436-486: the per-step radial AND angular membership test) +
// radial part of find_descriptor (nep.cu:488-547).  Walks list A then list B; every candidate's
// pair record goes to rstash at its Verlet row (coalesced), angular members are additionally
// compacted into acomp (what the angular kernels iterate) and registered in amap.
template <class S>
struct RadialDescBody {
  BoxD box;
  ModelD m;
  Bufs b;
  int write_records; // 0: the force pass recomputes the pair geometry from its own LDS window

  // candidate source of the default path: neighbour indices from the Verlet lists, positions
  // gathered from global memory (L2)
  struct GlobalFetch {
    const int* nlA;
    const int* nlB;
    const PosQ* posq;
    int64_t N;
    struct Tok {
      int jj[kGatherR];
    };
    // global loads of a chunk's neighbour indices; issued one chunk ahead, i.e. BEFORE the stores
    // of the chunk being processed (vmcnt retires in order: a load issued behind stores cannot be
    // waited for without draining those stores)
    NEPMI_HD void prefetch(int s0, int nn, int na, Tok& t) const
    {
      // entries past the end re-read the last valid slot (no branches around the loads)
#pragma unroll
      for (int u = 0; u < kGatherR; ++u) {
        const int idx = s0 + u < nn ? s0 + u : nn - 1;
        t.jj[u] = idx < na ? nlA[(int64_t)idx * N] : nlB[(int64_t)(idx - na) * N];
      }
    }
    NEPMI_HD void resolve(const Tok& t, int* jj, PosQ* pp) const
    {
#pragma unroll
      for (int u = 0; u < kGatherR; ++u) {
        jj[u] = t.jj[u];
        pp[u] = posq[jj[u]];
      }
    }
  };

  NEPMI_HD void operator()(int64_t k) const
  {
    run_with(k, GlobalFetch{b.nl_ang + k, b.nl_skin + k, b.posq, b.N});
  }

  template <class Fetch>
  NEPMI_HD void run_with(int64_t k, const Fetch& fetch) const
  {
    run_parts<1>(k, 0, fetch);
  }

  // PARTS lanes (1, or 2 adjacent lanes of a wavefront) share atom k: lane `part` walks the chunks
  // part, part + PARTS, ... of the list; the compact angular slots are handed out through a
  // prefix over the lane pair per chunk round, which keeps the sequential slot order, and the
  // partial sums are combined at the end (NEPMI_PAIR_XCHG = exchange with the partner lane).
  template <int PARTS, class Fetch>
  NEPMI_HD void run_parts(int64_t k, int part, const Fetch& fetch) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < 1) { // outer ghost: lends its position only
      if (part == 0) {
        b.nn_rad[k] = 0;
        b.nn_angstep[k] = 0;
      }
      return;
    }
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    const float rc1 = m.rc_r[t1], rca1 = m.rc_a[t1];
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

    const int na = b.nn_ang[k], nb = b.nn_skin[k];
    const int nn = na + nb;
    int cnt = 0, ca = 0;
    // The lists are walked in chunks of kGatherR entries: all index loads of a chunk, then all
    // position gathers, then the arithmetic -- kGatherR independent gathers in flight per lane.
    F4* __restrict__ rstash = b.rstash + k;
    F4* __restrict__ acomp = b.acomp + k;
    unsigned short* __restrict__ amap = b.amap + k;
    constexpr int kStride = PARTS * kGatherR;
    const int nrounds = (nn + kStride - 1) / kStride; // the same for every lane that shares the atom
    typename Fetch::Tok cur, nxt;
    if (nn > 0)
      fetch.prefetch(part * kGatherR < nn ? part * kGatherR : 0, nn, na, cur);
    for (int r = 0; r < nrounds; ++r) {
      const int s0 = r * kStride + part * kGatherR;
      int jj[kGatherR];
      PosQ pp[kGatherR];
      nxt = cur;
      if (s0 + kStride < nn)
        fetch.prefetch(s0 + kStride, nn, na, nxt); // next chunk's loads go out before this chunk's stores
      fetch.resolve(cur, jj, pp);
      cur = nxt;
      float xs[kGatherR], ys[kGatherR], zs[kGatherR], d2s[kGatherR];
      int t2s[kGatherR];
      bool ins[kGatherR], angs[kGatherR];
      int mine = 0; // angular members among this lane's entries of the round
#pragma unroll
      for (int u = 0; u < kGatherR; ++u) {
        const int idx = s0 + u;
        const PosQ p2 = pp[u];
        d2s[u] = pair_geometry(box, p1, p2, xs[u], ys[u], zs[u]);
        t2s[u] = p2.type;
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2s[u]]) * 0.5f;
        const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[t2s[u]]) * 0.5f;
        ins[u] = idx < nn && d2s[u] < rc * rc;
        angs[u] = idx < na && d2s[u] < rca * rca;
        mine += angs[u] ? 1 : 0;
      }
      int slot_next = ca;
      if (PARTS > 1 && r * kStride < na) { // pair-uniform: a chunk of this round touches list A
        const int other = NEPMI_PAIR_XCHG(mine);
        slot_next = ca + (part == 0 ? 0 : other);
        ca += mine + other;
      } else {
        ca += mine;
      }
#pragma unroll
      for (int u = 0; u < kGatherR; ++u) {
        const int idx = s0 + u;
        if (idx >= nn)
          continue;
        const int t2 = t2s[u];
        const float d2 = d2s[u];
        const bool inside = ins[u];
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
        // The pair record goes to the Verlet row itself (not a compacted slot): every lane of
        // the wavefront stores to the same row -> one contiguous 1 KiB store per slot.
        F4 e;
        e.x = xs[u];
        e.y = ys[u];
        e.z = zs[u];
        e.w = inside ? (int)((unsigned)jj[u] | ((unsigned)t2 << kIdxBits)) : -1;
        const int row = idx < na ? idx : b.MN_ang + (idx - na);
        if (write_records)
          rstash[(int64_t)row * N] = e;
        if (idx < na) {
          unsigned short slot = kNoSlot;
          if (angs[u]) {
            if (slot_next < b.MN_acomp) {
              acomp[(int64_t)slot_next * N] = e;
              slot = (unsigned short)slot_next;
            }
            ++slot_next;
          }
          amap[(int64_t)idx * N] = slot;
        }
        if (S::TS > 0) {
          // branch-free: entries outside the cutoff run the same arithmetic with weight 0, so the
          // kGatherR candidates of a chunk are independent instruction streams the scheduler can
          // interleave (the envelope fc is evaluated at min(d, rc) to stay finite)
          cnt += inside ? 1 : 0;
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
        } else {
          if (!inside)
            continue;
          ++cnt;
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
            float g = 0.0f;
            for (int kk = 0; kk <= KR; ++kk)
              g += fn[kk] * c[n * (KR + 1) + kk];
            q[n] += g;
          }
        }
      }
    }
    if (PARTS > 1) { // totals on both lanes of the pair
      cnt += NEPMI_PAIR_XCHG(cnt);
      if (S::TS > 0) {
#pragma unroll
        for (int t = 0; t < TSM; ++t)
#pragma unroll
          for (int kk = 0; kk <= S::KRM; ++kk)
            Ssum[t][kk] += NEPMI_PAIR_XCHG(Ssum[t][kk]);
      } else {
#pragma unroll
        for (int n = 0; n <= S::NRM; ++n)
          q[n] += NEPMI_PAIR_XCHG(q[n]);
      }
    }
    if (ca > b.MN_acomp) {
      if (part == 0)
        NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 4);
      ca = b.MN_acomp;
    }
    if (part == 0) {
      b.nn_rad[k] = cnt;
      b.nn_angstep[k] = ca;
    }

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
    if (part == 0)
      for (int n = 0; n <= NR; ++n)
        b.q[(int64_t)n * N + b.tpos[k]] = q[n] * m.qscale[n];
  }
};


// radial part of find_descriptor from pair records that already exist (small-box path)
template <class S>
struct RadialFromRecordsBody {
  ModelD m;
  Bufs b;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const int t1 = b.posq[k].type;
    const float rc1 = m.rc_r[t1];
    float q[S::NRM + 1];
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n)
      q[n] = 0.0f;
    const int na = b.nn_ang[k], nn = na + b.nn_skin[k];
    for (int idx = 0; idx < nn; ++idx) {
      const int row = idx < na ? idx : b.MN_ang + (idx - na);
      const F4 e = b.rstash[(int64_t)row * N + k];
      const int t2 = (int)((unsigned)e.w >> kIdxBits);
      const float d = sqrtf(dot3f(e.x, e.x, e.y, e.y, e.z, e.z));
      const float rc = (rc1 + m.rc_r[t2]) * 0.5f;
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
        float g = 0.0f;
        for (int kk = 0; kk <= KR; ++kk)
          g += fn[kk] * c[n * (KR + 1) + kk];
        q[n] += g;
      }
    }
    for (int n = 0; n <= NR; ++n)
      b.q[(int64_t)n * N + b.tpos[k]] = q[n] * m.qscale[n];
  }
};

// s_{n,lm}(i) = sum_j g_n(r_ij) b_lm(rhat_ij) over the compacted angular pair records of atom k
// (find_descriptor, nep.cu:588-610 + accumulate_s).  Shared by the angular descriptor kernel and by
// the angular force kernel's recompute path, so that both see bit-identical sums.
template <class S, int PARTS, class LP>
NEPMI_HD void angular_s_sums_scalar(const ModelD& m, const Bufs& b, int64_t k, int t1, LP cang, int part, float* s)
{
  // PARTS lanes share the atom: lane `part` owns the radial channels n = part, part + PARTS, ... and
  // keeps them at local rows i = 0, 1, ... of s (PARTS == 1: all channels, row i = n)
  constexpr int NLOC = (S::NAM + PARTS) / PARTS;
  const int64_t N = b.N;
  const int NA = S::fixed ? S::NA : m.NA;
  const int KA = S::fixed ? S::KA : m.KA;
  const float rc1 = m.rc_a[t1];
  const int cstride = cang_stride(m);
#pragma unroll
  for (int a = 0; a < NLOC * kNumHarm; ++a)
    s[a] = 0.0f;
  const int na = b.nn_angstep[k];
  const F4* __restrict__ acomp = b.acomp + k;
  F4 e_next;
  if (na > 0)
    e_next = acomp[0];
  for (int a = 0; a < na; ++a) {
    const F4 e = e_next;
    if (a + 1 < na)
      e_next = acomp[(int64_t)(a + 1) * N]; // in flight while this record is processed
    if (e.w == kNullRecord)
      continue; // a padding row (wave-synchronous records)
    const int t2 = (int)((unsigned)e.w >> kIdxBits);
    const float x = e.x, y = e.y, z = e.z;
    float d, dinv;
    dist_and_inv(dot3f(x, x, y, y, z, z), d, dinv);
    const float rc = m.uniform_rc ? m.rc_a_max : (rc1 + m.rc_a[t2]) * 0.5f;
    const float rcinv = fast_rcp(rc);
    float fc;
    cutoff_fc(rcinv, d, fc);
    float fn[S::KAM + 1];
    if (S::fixed)
      basis_fn<S::KAM>(rcinv, d, fc, fn);
    else
      basis_fn_rt(KA, rcinv, d, fc, fn);
    float bh[kNumHarm];
    harmonics(x * dinv, y * dinv, z * dinv, bh);
    LP c = cang + (t1 * m.T + t2) * cstride;
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const int n = part + PARTS * i;
      if (n > NA)
        break;
      float g = 0.0f;
#pragma unroll
      for (int kk = 0; kk <= S::KAM; ++kk) {
        if (!S::fixed && kk > KA)
          break;
        g = fmaf(fn[kk], c[n * (KA + 1) + kk], g);
      }
#pragma unroll
      for (int h = 0; h < kNumHarm; ++h)
        s[i * kNumHarm + h] = fmaf(g, bh[h], s[i * kNumHarm + h]);
    }
  }
}

// Fixed shapes: the same sums held as 12 register pairs per channel (harmonics_pairs): one v_pk_fma_f32 per pair
// and channel instead of two v_fma_f32.  Every sum sees the same operations in the same order as in the scalar
// form, so the two are bit-identical.
template <class S, int PARTS, class LP>
NEPMI_HD void angular_s_sums(const ModelD& m, const Bufs& b, int64_t k, int t1, LP cang, int part, float* s)
{
  if constexpr (!S::fixed) {
    angular_s_sums_scalar<S, PARTS>(m, b, k, t1, cang, part, s);
  } else {
    constexpr int NLOC = (S::NAM + PARTS) / PARTS;
    const int64_t N = b.N;
    const float rc1 = m.rc_a[t1];
    const int cstride = cang_stride(m);
    f2 s2[NLOC * kHarmPairs];
#pragma unroll
    for (int a = 0; a < NLOC * kHarmPairs; ++a)
      s2[a] = bc2(0.0f);
    const int na = b.nn_angstep[k];
    const F4* __restrict__ acomp = b.acomp + k;
    F4 e_next;
    if (na > 0)
      e_next = acomp[0];
    for (int a = 0; a < na; ++a) {
      const F4 e = e_next;
      if (a + 1 < na)
        e_next = acomp[(int64_t)(a + 1) * N]; // in flight while this record is processed
      if (e.w == kNullRecord)
        continue; // a padding row (wave-synchronous records)
      const int t2 = (int)((unsigned)e.w >> kIdxBits);
      const float x = e.x, y = e.y, z = e.z;
      float d, dinv;
      dist_and_inv(dot3f(x, x, y, y, z, z), d, dinv);
      const float rc = m.uniform_rc ? m.rc_a_max : (rc1 + m.rc_a[t2]) * 0.5f;
      const float rcinv = fast_rcp(rc);
      float fc;
      cutoff_fc(rcinv, d, fc);
      float fn[S::KAM + 1];
      basis_fn<S::KAM>(rcinv, d, fc, fn);
      f2 bh[kHarmPairs];
      harmonics_pairs(x * dinv, y * dinv, z * dinv, bh);
      LP c = cang + (t1 * m.T + t2) * cstride;
#pragma unroll
      for (int i = 0; i < NLOC; ++i) {
        const int n = part + PARTS * i;
        if (n > S::NA)
          break;
        float g = 0.0f;
#pragma unroll
        for (int kk = 0; kk <= S::KAM; ++kk)
          g = fmaf(fn[kk], c[n * (S::KA + 1) + kk], g);
        const f2 g2 = bc2(g);
#pragma unroll
        for (int q = 0; q < kHarmPairs; ++q)
          s2[i * kHarmPairs + q] = vfma(g2, bh[q], s2[i * kHarmPairs + q]);
      }
    }
    const int hp[kNumHarm] = NEPMI_HARM_PAIR_INIT;
#pragma unroll
    for (int i = 0; i < NLOC; ++i)
#pragma unroll
      for (int q = 0; q < kHarmPairs; ++q) {
        s[i * kNumHarm + hp[2 * q]] = s2[i * kHarmPairs + q].x;
        s[i * kNumHarm + hp[2 * q + 1]] = s2[i * kHarmPairs + q].y;
      }
  }
}

// angular part of find_descriptor (nep.cu:549-640) on the compacted angular pair records:
// no gathers, no geometry, every lane of the wavefront has real work in every iteration.
// LDS image of the per-atom ANN for the fused descriptor + ANN kernel (AngularDescBody::fuse_ann), behind c_ang:
//   W[t][neuron][DP] (DP = dim rounded up to 4, zero padded) | b0[t][neuron] | w1[t][neuron] | c_rad[t1 t2][n][k]
// The type stride of W is padded to 8 mod 32 words, so that lanes of two types read different banks.
struct AnnLdsLayout {
  int DP, wstride, off_w, off_b0, off_w1, off_c, total;
};
NEPMI_HD AnnLdsLayout ann_lds_layout(const ModelD& m)
{
  AnnLdsLayout a;
  a.DP = (m.dim + 3) / 4 * 4;
  a.wstride = m.nneu * a.DP;
  a.wstride += (8 - (a.wstride & 31) + 32) & 31;
  a.off_w = (cang_floats(m) + 3) / 4 * 4;
  a.off_b0 = a.off_w + m.T * a.wstride;
  a.off_w1 = a.off_b0 + m.T * m.nneu;
  a.off_c = a.off_w1 + m.T * m.nneu;
  a.total = a.off_c + m.T * m.T * (m.NR + 1) * (m.KR + 1);
  return a;
}
NEPMI_HD float ann_tanh(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // 1 - 2 / (exp(2x) + 1), branch-free (v_exp_f32 + v_rcp_f32), as in the matrix-core kernel
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
#else
  return tanhf(x);
#endif
}

template <class S>
struct AngularDescBody {
  ModelD m;
  Bufs b;
  int recompute_s; // the force kernel rebuilds s from the records: do not write sbuf
  int fuse_ann;    // one-lane form only: run the per-atom ANN right here (q never leaves the registers): writes
                   // pe_i, fp and the radial force table instead of q
  static constexpr bool kUsesLds = true;
  static constexpr int kMinWavesPerEu = 1, kMinWavesPerEuPairs = 1;
  NEPMI_HD int lds_floats() const { return fuse_ann ? ann_lds_layout(m).total : cang_floats(m); }
  NEPMI_HD void lds_stage(float* dst, int tid, int nth) const
  {
    cang_stage(m, dst, tid, nth);
    if (!fuse_ann)
      return;
    const AnnLdsLayout a = ann_lds_layout(m);
    for (int idx = tid; idx < m.T * m.nneu * a.DP; idx += nth) {
      const int t = idx / (m.nneu * a.DP), r = idx - t * (m.nneu * a.DP);
      const int j = r / a.DP, d = r - j * a.DP;
      dst[a.off_w + t * a.wstride + j * a.DP + d] = d < m.dim ? m.w0[((size_t)t * m.nneu + j) * m.dim + d] : 0.0f;
    }
    for (int idx = tid; idx < m.T * m.nneu; idx += nth) {
      dst[a.off_b0 + idx] = m.b0[idx];
      dst[a.off_w1 + idx] = m.w1[idx];
    }
    for (int idx = tid; idx < m.T * m.T * (m.NR + 1) * (m.KR + 1); idx += nth)
      dst[a.off_c + idx] = m.c_rad[idx];
  }

  template <class LP>
  NEPMI_HD void run(int64_t k, LP cang) const
  {
    run_parts<1>(k, 0, cang);
  }

  // PARTS lanes share the atom, each owning the radial channels n = part, part + PARTS, ...: the sums
  // and the invariants of a channel never leave its lane, so there is nothing to combine.
  template <int PARTS, class LP>
  NEPMI_HD void run_parts(int64_t k, int part, LP cang) const
  {
    constexpr int NLOC = (S::NAM + PARTS) / PARTS;
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_desc)
      return;
    const int64_t gk = b.tpos[k]; // q / fp column of this atom (work order)
    const int NR = S::fixed ? S::NR : m.NR;
    const int NA = S::fixed ? S::NA : m.NA;
    const int t1 = b.posq[k].type;
    constexpr int kQ = (S::fixed && PARTS == 1) ? (S::DIMM + 3) / 4 * 4 : 1;
    float qfull[kQ]; // fuse_ann: the scaled descriptor of this atom
    float s[NLOC * kNumHarm];
    angular_s_sums<S, PARTS>(m, b, k, t1, cang, part, s);

#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const int n = part + PARTS * i;
      if (n > NA)
        break;
      if (!recompute_s) {
#pragma unroll
        for (int h = 0; h < kNumHarm; ++h)
          b.sbuf[(int64_t)(n * kNumHarm + h) * N + k] = s[i * kNumHarm + h];
      }
      float qn[S::kRows];
#pragma unroll
      for (int L = 0; L < S::kRows; ++L)
        qn[L] = 0.0f;
      invariants<!S::fixed>(m, &s[i * kNumHarm], qn, 1);
#pragma unroll
      for (int L = 0; L < S::kRows; ++L) { // constant trip count: qn stays in registers
        if (L < m.numL) {
          const int d = (NR + 1) + L * (NA + 1) + n;
          if (S::fixed && PARTS == 1 && fuse_ann)
            qfull[(S::fixed ? d : 0)] = qn[L] * m.qscale[d];
          else
            b.q[(int64_t)d * N + gk] = qn[L] * m.qscale[d];
        }
      }
    }
    if constexpr (S::fixed && PARTS == 1) {
      if (fuse_ann)
        ann_tail(k, gk, t1, cang, qfull);
    }
  }

  // apply_ann_one_layer (nep_utilities.cuh:169-194) + the radial force table, for the atom of this lane, with the
  // weights of its type read from LDS (two types in a wavefront read different banks).  Per neuron: one pass over
  // its 16-byte weight groups feeds both the forward dot product and the backward axpy (packed FP32).
  template <class LP>
  NEPMI_HD void ann_tail(int64_t k, int64_t gk, int t1, LP lds, float* q) const
  {
    constexpr int DP = (S::DIMM + 3) / 4 * 4;
    const int64_t N = b.N;
    const AnnLdsLayout a = ann_lds_layout(m);
    const int NR = S::NR, KR = S::KR;
#pragma unroll
    for (int n = 0; n <= S::NR; ++n)
      q[n] = b.q[(int64_t)n * N + gk]; // radial part, written (scaled) by the radial pass
#pragma unroll
    for (int d = S::DIMM; d < DP; ++d)
      q[d] = 0.0f;
    f2 g2[DP / 2];
#pragma unroll
    for (int i = 0; i < DP / 2; ++i)
      g2[i] = bc2(0.0f);
    float e = 0.0f;
    LP W = lds + a.off_w + t1 * a.wstride;
    LP B0 = lds + a.off_b0 + t1 * m.nneu;
    LP W1 = lds + a.off_w1 + t1 * m.nneu;
    for (int j = 0; j < m.nneu; ++j) {
      LP w = W + j * DP;
      f2 w2[DP / 2];
#pragma unroll
      for (int i = 0; i < DP / 2; ++i)
        w2[i] = mk2(w[2 * i], w[2 * i + 1]);
      f2 acc = bc2(0.0f);
#pragma unroll
      for (int i = 0; i < DP / 2; ++i)
        acc = vfma(w2[i], mk2(q[2 * i], q[2 * i + 1]), acc);
      const float h = ann_tanh(acc.x + acc.y - B0[j]);
      const float wj = W1[j];
      e = fmaf(wj, h, e);
      const f2 coef = bc2(wj * (1.0f - h * h));
#pragma unroll
      for (int i = 0; i < DP / 2; ++i)
        g2[i] = vfma(coef, w2[i], g2[i]);
    }
    b.pe_i[k] = e - (m.b1 + m.b1t[t1]);
    float Fp[S::DIMM];
#pragma unroll
    for (int d = 0; d < S::DIMM; ++d) {
      Fp[d] = ((d & 1) ? g2[d >> 1].y : g2[d >> 1].x) * m.qscale[d];
      b.fp[(int64_t)d * N + gk] = Fp[d];
    }
    // radial force table A[t2][k] = sum_n Fp[n] c[t1][t2][n][k] (AnnBody)
    const int KRP = b.KRP;
    for (int t2 = 0; t2 < m.T; ++t2) {
      LP c = lds + a.off_c + (t1 * m.T + t2) * (NR + 1) * (KR + 1);
#pragma unroll
      for (int kk = 0; kk <= S::KR; ++kk) {
        float v = 0.0f;
#pragma unroll
        for (int n = 0; n <= S::NR; ++n)
          v = fmaf(Fp[n], c[n * (KR + 1) + kk], v);
        b.atab[(size_t)k * (m.T * KRP) + t2 * KRP + kk] = v;
      }
    }
  }
};

// apply_ann_one_layer (nep_utilities.cuh:169-194, NEP5 :285-310) + the per-atom radial force
// table  A_i[t2][k] = sum_n Fp_i[n] c[t1][t2][n][k]  (the n-contraction of find_force_radial,
// nep.cu:699-754, done once per atom instead of once per pair).
template <class S>
struct AnnBody {
  ModelD m;
  Bufs b;
  NEPMI_HD void operator()(int64_t g) const
  {
    const int64_t N = b.N;
    const int64_t k = b.tperm[g]; // type-grouped work order
    if (b.lvl[k] < b.lvl_desc)
      return;
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const int dim = S::fixed ? S::DIMM : m.dim;
    const int nneu = m.nneu;
    const int t1 = b.posq[k].type;
    float q[S::DIMM], Fp[S::DIMM];
#pragma unroll
    for (int d = 0; d < S::DIMM; ++d) {
      if (!S::fixed && d >= dim)
        break;
      q[d] = b.q[(int64_t)d * N + g];
      Fp[d] = 0.0f;
    }
    float E = 0.0f;
    for (int tu = 0; tu < m.T; ++tu) {
      if (!NEPMI_WAVE_ANY(t1 == tu))
        continue;
      cfloat_ptr w0 = as_const(m.w0) + (size_t)tu * nneu * dim;
      cfloat_ptr b0 = as_const(m.b0) + (size_t)tu * nneu;
      cfloat_ptr w1 = as_const(m.w1) + (size_t)tu * nneu;
      cfloat_ptr qs = as_const(m.qscale);
      float g[S::DIMM];
#pragma unroll
      for (int d = 0; d < S::DIMM; ++d)
        g[d] = 0.0f;
      float e = 0.0f;
      for (int j = 0; j < nneu; ++j) {
        cfloat_ptr w = w0 + (size_t)j * dim;
        float a = 0.0f;
#pragma unroll
        for (int d = 0; d < S::DIMM; ++d) {
          if (!S::fixed && d >= dim)
            break;
          a = fmaf(w[d], q[d], a);
        }
        const float h = tanhf(a - b0[j]);
        const float wj = w1[j];
        e = fmaf(wj, h, e);
        const float coef = wj * (1.0f - h * h);
#pragma unroll
        for (int d = 0; d < S::DIMM; ++d) {
          if (!S::fixed && d >= dim)
            break;
          g[d] = fmaf(coef, w[d], g[d]);
        }
      }
      e -= m.b1 + as_const(m.b1t)[tu];
      if (t1 == tu) {
        E = e;
#pragma unroll
        for (int d = 0; d < S::DIMM; ++d) {
          if (!S::fixed && d >= dim)
            break;
          Fp[d] = g[d] * qs[d];
        }
      }
      // radial force table for atoms of this type
      const int KRP = b.KRP;
      for (int t2 = 0; t2 < (b.skip_atab ? 0 : m.T); ++t2) {
        cfloat_ptr c = as_const(m.c_rad) + (size_t)(tu * m.T + t2) * (NR + 1) * (KR + 1);
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk) {
          if (!S::fixed && kk > KR)
            break;
          float a = 0.0f;
#pragma unroll
          for (int n = 0; n <= S::NRM; ++n) {
            if (!S::fixed && n > NR)
              break;
            a = fmaf(g[n] * qs[n], c[n * (KR + 1) + kk], a);
          }
          if (t1 == tu)
            b.atab[(size_t)k * (m.T * KRP) + t2 * KRP + kk] = a;
        }
      }
    }
    b.pe_i[k] = E;
#pragma unroll
    for (int d = 0; d < S::DIMM; ++d) {
      if (!S::fixed && d >= dim)
        break;
      b.fp[(int64_t)d * N + g] = Fp[d];
    }
    if (b.fpr) { // the radial rows again, atom-major (gathered by the neighbours' force assembly)
#pragma unroll
      for (int n = 0; n <= S::NRM; ++n) {
        if (!S::fixed && n > NR)
          break;
        b.fpr[(size_t)k * b.FPR + n] = Fp[n];
      }
    }
  }
};

// find_partial_force_angular (nep.cu:774-861) in adjoint form + find_force_ZBL (nep.cu:863-975).
//   f12 = dU_i/dr_ij = rhat * sum_abc Q_abc b_abc + (1/d)(I - rhat rhat^T) sum_abc P_abc grad b_abc,
//   P_abc = sum_n G[n][abc] g_n(d),  Q_abc = sum_n G[n][abc] g_n'(d),  G = dU_i/ds (invariants_adjoint)
template <class S>
struct AngularForceBody {
  ModelD m;
  Bufs b;
  int recompute_s; // few angular neighbours: rebuilding s (a second walk over <= MN_a records) is
                   // cheaper than the (n_a+1)*24 floats per atom written and read back through HBM
  static constexpr bool kUsesLds = true;
#ifndef NEPMI_AF_WAVES
#define NEPMI_AF_WAVES 2
#endif
  // one-lane form: the per-atom table G and the P/Q sums sit a few registers above 256; held to two wavefronts per SIMD
  static constexpr int kMinWavesPerEu = (S::fixed && S::NA + 1 < 7) ? NEPMI_AF_WAVES : 1;
  // lane-pair form: half the table per lane; two wavefronts per SIMD is what it exists for
#ifndef NEPMI_AF_WAVES_PAIRS
#define NEPMI_AF_WAVES_PAIRS 2 // A/B switch (profiles/ab_variants.sh): 1 = the whole register file for one wavefront per SIMD
#endif
  static constexpr int kMinWavesPerEuPairs = S::fixed ? NEPMI_AF_WAVES_PAIRS : 1;
  NEPMI_HD int lds_floats() const { return cang_floats(m); }
  NEPMI_HD void lds_stage(float* dst, int tid, int nth) const { cang_stage(m, dst, tid, nth); }

  template <class LP>
  NEPMI_HD void run(int64_t k, LP cang) const
  {
    run_parts<1>(k, 0, cang);
  }

  // PARTS lanes (2 = an adjacent lane pair) share the atom: lane `part` holds the rows of G of the radial
  // channels n = part, part + PARTS, ... only (half the registers: two wavefronts per SIMD instead of
  // one), forms its share of P and Q, contracts it with the harmonics -- the contraction is linear, so
  // the partial (w, v) are simply added across the pair -- and lane 0 writes f12.  Same instruction
  // stream on both lanes: no divergence.
  template <int PARTS, class LP>
  NEPMI_HD void run_parts(int64_t k, int part, LP cang) const
  {
    constexpr int NLOC = (S::NAM + PARTS) / PARTS;
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_desc || (b.level && !b.angf[k])) // outer ghosts; inner-ring ghosts whose f12 no owned atom will read
      return;
    const int64_t gk = b.tpos[k]; // q / fp column of this atom (work order)
    const int NR = S::fixed ? S::NR : m.NR;
    const int NA = S::fixed ? S::NA : m.NA;
    const int t1 = b.posq[k].type;

    float G[NLOC * kNumHarm];
    if (recompute_s)
      angular_s_sums<S, PARTS>(m, b, k, t1, cang, part, G);
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const int n = part + PARTS * i;
      if (n > NA)
        break;
      float fpn[S::kRows];
#pragma unroll
      for (int L = 0; L < S::kRows; ++L)
        fpn[L] = L < m.numL ? b.fp[(int64_t)((NR + 1) + L * (NA + 1) + n) * N + gk] : 0.0f;
      if (!recompute_s) {
#pragma unroll
        for (int h = 0; h < kNumHarm; ++h)
          G[i * kNumHarm + h] = b.sbuf[(int64_t)(n * kNumHarm + h) * N + k];
      }
      invariants_adjoint<!S::fixed>(m, fpn, 1, &G[i * kNumHarm]);
    }
    pairs_from_G<PARTS>(k, part, cang, t1, G, F12Store{b.f12 + k, b.N});
  }

  // where the partial force of pair `a` goes: the compact f12 rows (the separate force-assembly kernels read them) -- or, in the
  // per-brick kernel of nep_fused.h, straight into the LDS accumulators of the scatter form
  struct F12Store {
    F4* f12;
    int64_t N;
    NEPMI_HD void operator()(int a, int part, const F4& out, const F4& /*record*/) const
    {
      if (part == 0)
        f12[(int64_t)a * N] = out;
    }
  };

  // The pair loop: partial forces f12 of this step's angular pairs from the atom's adjoint table G (this lane's channels,
  // harmonic order), + ZBL.  Also the tail of the fused descriptor + ANN + force kernel (nep_fused.h), which arrives here
  // with G built from sums that never left the registers.
  template <int PARTS, class LP, class Sink>
  NEPMI_HD void pairs_from_G(int64_t k, int part, LP cang, int t1, const float* G, Sink&& sink) const
  {
    constexpr int NLOC = (S::NAM + PARTS) / PARTS;
    const int64_t N = b.N;
    const int NA = S::fixed ? S::NA : m.NA;
    const int KA = S::fixed ? S::KA : m.KA;
    const float rc1 = m.rc_a[t1];
    const int cstride = cang_stride(m);
    // fixed shapes: the table as register pairs (harmonics_pairs order), P and Q as pairs as well
    constexpr int NG2 = S::fixed ? NLOC * kHarmPairs : 1;
    f2 G2[NG2];
    if constexpr (S::fixed) {
      const int hp[kNumHarm] = NEPMI_HARM_PAIR_INIT;
#pragma unroll
      for (int i = 0; i < NLOC; ++i)
#pragma unroll
        for (int q = 0; q < kHarmPairs; ++q)
          G2[i * kHarmPairs + q] = mk2(G[i * kNumHarm + hp[2 * q]], G[i * kNumHarm + hp[2 * q + 1]]);
    }

    float zf[3] = {0, 0, 0}, zv[6] = {0, 0, 0, 0, 0, 0}, zpe = 0.0f;
    float pzi = 0.0f;
    int zi = 0;
    if (m.zbl_enabled) {
      zi = m.atomic_number[t1];
      pzi = powf((float)zi, 0.23f);
    }

    const int na = b.nn_angstep[k];
    const F4* __restrict__ acomp = b.acomp + k;
    F4 e_next;
    if (na > 0)
      e_next = acomp[0];
    for (int a = 0; a < na; ++a) {
      const F4 e = e_next;
      if (a + 1 < na)
        e_next = acomp[(int64_t)(a + 1) * N];
      if (e.w == kNullRecord)
        continue; // a padding row (wave-synchronous records): no partial force is written for it, its aslot is the sentinel
      const int t2 = (int)((unsigned)e.w >> kIdxBits);
      const float x = e.x, y = e.y, z = e.z;
      float d, dinv;
      dist_and_inv(dot3f(x, x, y, y, z, z), d, dinv);
      const float rc = m.uniform_rc ? m.rc_a_max : (rc1 + m.rc_a[t2]) * 0.5f;
      const float rcinv = fast_rcp(rc);
      float fc, fcp;
      cutoff_fc_fcp(rcinv, d, fc, fcp);
      float fn[S::KAM + 1], fnp[S::KAM + 1];
      if (S::fixed)
        basis_fn_fnp<S::KAM>(rcinv, d, fc, fcp, fn, fnp);
      else
        basis_fn_fnp_rt(KA, rcinv, d, fc, fcp, fn, fnp);
      LP c = cang + (t1 * m.T + t2) * cstride;
      const float ux = x * dinv, uy = y * dinv, uz = z * dinv;
      float w, vx, vy, vz;
      if constexpr (S::fixed) {
        // (g_n, g_n') side by side: one packed fma per basis function and channel; then P += G g, Q += G g' on the
        // register pairs of G
        f2 ffp[S::KAM + 1];
#pragma unroll
        for (int kk = 0; kk <= S::KAM; ++kk)
          ffp[kk] = mk2(fn[kk], fnp[kk]);
        f2 P2[kHarmPairs], Q2[kHarmPairs];
#pragma unroll
        for (int q = 0; q < kHarmPairs; ++q)
          P2[q] = Q2[q] = bc2(0.0f);
#pragma unroll
        for (int i = 0; i < NLOC; ++i) {
          const int n = part + PARTS * i;
          if (n > S::NA)
            break;
          f2 ggp = bc2(0.0f);
#pragma unroll
          for (int kk = 0; kk <= S::KAM; ++kk)
            ggp = vfma(ffp[kk], bc2(c[n * (S::KA + 1) + kk]), ggp);
          const f2 g2 = bc2(ggp.x), gp2 = bc2(ggp.y);
#pragma unroll
          for (int q = 0; q < kHarmPairs; ++q) {
            P2[q] = vfma(G2[i * kHarmPairs + q], g2, P2[q]);
            Q2[q] = vfma(G2[i * kHarmPairs + q], gp2, Q2[q]);
          }
        }
        harmonics_contract_pairs(ux, uy, uz, P2, Q2, w, vx, vy, vz);
      } else {
        float P[kNumHarm], Q[kNumHarm];
#pragma unroll
        for (int h = 0; h < kNumHarm; ++h)
          P[h] = Q[h] = 0.0f;
#pragma unroll
        for (int i = 0; i < NLOC; ++i) {
          const int n = part + PARTS * i;
          if (n > NA)
            break;
          float g = 0.0f, gp = 0.0f;
#pragma unroll
          for (int kk = 0; kk <= S::KAM; ++kk) {
            if (!S::fixed && kk > KA)
              break;
            const float cc = c[n * (KA + 1) + kk];
            g = fmaf(fn[kk], cc, g);
            gp = fmaf(fnp[kk], cc, gp);
          }
#pragma unroll
          for (int h = 0; h < kNumHarm; ++h) {
            P[h] = fmaf(G[i * kNumHarm + h], g, P[h]);
            Q[h] = fmaf(G[i * kNumHarm + h], gp, Q[h]);
          }
        }
        harmonics_contract(ux, uy, uz, P, Q, w, vx, vy, vz);
      }
      if (PARTS > 1) {
        w += NEPMI_PAIR_XCHG(w);
        vx += NEPMI_PAIR_XCHG(vx);
        vy += NEPMI_PAIR_XCHG(vy);
        vz += NEPMI_PAIR_XCHG(vz);
      }
      const float udv = ux * vx + uy * vy + uz * vz;
      F4 out;
      out.x = ux * w + (vx - ux * udv) * dinv;
      out.y = uy * w + (vy - uy * udv) * dinv;
      out.z = uz * w + (vz - uz * udv) * dinv;
      out.w = 0;
      sink(a, part, out, e);

      // ZBL is per pair and independent of the channel split: the lanes take alternate neighbours.  A pair beyond
      // the outer cutoff contributes exact zeros (fc = 0): it is skipped, and with it four exponentials, a sine, a
      // cosine and a power -- for the alloy models that is nearly every angular pair.
      float zbl_r2 = 0.0f;
      const float* zbl_p10 = nullptr;
      if (m.zbl_enabled) {
        if (m.zbl_flexible) {
          const int ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
          const int zidx = ta * m.T - (ta * (ta - 1)) / 2 + (tb - ta);
          zbl_p10 = m.zbl_para + 10 * zidx;
          zbl_r2 = zbl_p10[1];
        } else if (m.zbl_rco) { // type-wise outer cutoff, inner cutoff 0 (nep.cu:935-941)
          zbl_r2 = m.zbl_rco[t1 * m.T + t2];
        } else {
          zbl_r2 = m.zbl_rc_outer;
        }
      }
      if (m.zbl_enabled && d < zbl_r2 && (PARTS == 1 || (a % PARTS) == part)) {
        const int zj = m.atomic_number[t2];
        const float a_inv = (pzi + powf((float)zj, 0.23f)) * 2.134563f;
        const float zizj = 14.399645f * (float)zi * (float)zj;
        float f, fp;
        if (m.zbl_flexible)
          zbl_pair(zbl_p10, zizj, a_inv, 0.0f, 0.0f, d, dinv, f, fp);
        else if (m.zbl_rco)
          zbl_pair(nullptr, zizj, a_inv, 0.0f, zbl_r2, d, dinv, f, fp);
        else
          zbl_pair(nullptr, zizj, a_inv, m.zbl_rc_inner, m.zbl_rc_outer, d, dinv, f, fp);
        const float f2 = fp * dinv * 0.5f;
        const float fx = x * f2, fy = y * f2, fz = z * f2; // f12; f21 = -f12
        zf[0] += fx + fx;
        zf[1] += fy + fy;
        zf[2] += fz + fz;
        zv[0] -= x * fx;
        zv[1] -= y * fy;
        zv[2] -= z * fz;
        zv[3] -= x * fy;
        zv[4] -= x * fz;
        zv[5] -= y * fz;
        zpe += f * 0.5f;
      }
    }
    if (m.zbl_enabled && PARTS > 1) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        zf[d] += NEPMI_PAIR_XCHG(zf[d]);
#pragma unroll
      for (int d = 0; d < 6; ++d)
        zv[d] += NEPMI_PAIR_XCHG(zv[d]);
      zpe += NEPMI_PAIR_XCHG(zpe);
    }
    if (m.zbl_enabled && part == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        b.zbl[(int64_t)d * N + k] = zf[d];
#pragma unroll
      for (int d = 0; d < 6; ++d)
        b.zbl[(int64_t)(3 + d) * N + k] = zv[d];
      b.zbl[(int64_t)9 * N + k] = zpe;
    }
  }
};

// find_force_radial (nep.cu:661-772) + gpu_find_force_many_body (potential.cu:170-297); the 13 output planes
// are written in internal order (Bufs::fo).  Gather form: used when the LDS-window kernels do not apply
// (fewer than 8 cells in a periodic direction, oversized windows) and by the small-box branch.
// One walk over the pair records (list A rows, then list B rows): every record inside rc_r gives
// the radial pair force from the two per-atom tables A_i, A_j; records of list A that are angular
// members this step (amap) add f12 - f21, with f21 found through the static reverse slot.
template <class S>
struct ForceAssembleBody {
  ModelD m;
  Bufs b;
  // pair records straight from the radial pass (rows [A slots | MN_ang + B slots] of rstash)
  struct RecordSource {
    const F4* rstash; // + k
    int64_t N;
    int MN_ang;
    struct State {
    };
    NEPMI_HD void begin(int, int, int, int, State&) const {}
    template <int G>
    NEPMI_HD void load(int s0, int, int nn, int na, State&, F4* ee) const
    {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int idx = s0 + u < nn ? s0 + u : nn - 1;
        const int row = idx < na ? idx : MN_ang + (idx - na);
        ee[u] = rstash[(int64_t)row * N];
      }
    }
  };

  NEPMI_HD void operator()(int64_t k) const
  {
    run_with(k, RecordSource{b.rstash + k, b.N, b.MN_ang});
  }

  // src delivers, kGather at a time, the pair records of atom k in list order (A slots, then B
  // slots): r12 and (j | t2 << kIdxBits), or -1 when the pair is outside the radial cutoff
  template <class Src>
  NEPMI_HD void run_with(int64_t k, const Src& src) const
  {
    run_parts<1, kGather>(k, 0, src);
  }

  // PARTS lanes share atom k (see RadialDescBody::run_parts): lane `part` takes every PARTS-th
  // chunk of the pair list; forces and virials are linear in the pairs, so the partial sums are
  // simply added across the lane pair at the end.
  // G = pair entries per lane and chunk (their table gathers are issued together)
  template <int PARTS, int G, class Src>
  NEPMI_HD void run_parts(int64_t k, int part, const Src& src) const
  {
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_force) // forces only for owned atoms (reverse mode: the neighbour halves on the ghosts too)
      return;
    const int KR = S::fixed ? S::KR : m.KR;
    const int t1 = b.posq[k].type;
    const float rc1 = m.rc_r[t1];
    const int KRP = b.KRP;
    const int arow = m.T * KRP;
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    float Aown[TSM][S::KRM + 1];
    if (S::TS > 0) {
#pragma unroll
      for (int t = 0; t < TSM; ++t)
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
          Aown[t][kk] = b.atab[(size_t)k * arow + t * KRP + kk];
    }
    float F[3] = {0, 0, 0};
    float W[6] = {0, 0, 0, 0, 0, 0};  // radial part: symmetric (xx yy zz xy xz yz)
    float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // angular part: xx yy zz xy xz yz yx zx zy

    const int na = b.nn_ang[k], nbn = b.nn_skin[k];
    const int nn = na + nbn;
    const float* __restrict__ atab = b.atab;
    const unsigned short* __restrict__ amap = b.amap;
    const unsigned short* __restrict__ rev = b.rev_ang + k;
    const F4* __restrict__ f12 = b.f12;
    typename Src::State st;
    constexpr int kStride = PARTS * G;
    if (nn > 0)
      src.begin(part * G, kStride, nn, na, st);
    // one chunk of the walk; WITH_ANG = the chunk can still contain list-A entries (s0 < na): only
    // those carry the f12 - f21 code, the long list-B tail runs a branch-free radial-only body
    auto chunk = [&](const int s0, auto with_ang) {
      constexpr bool WITH_ANG = decltype(with_ang)::value;
      F4 ee[G];
      float Aj[G][S::KRM + 1];
      src.template load<G>(s0, kStride, nn, na, st, ee);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = ee[u].w == -1 ? (int)k : (int)((unsigned)ee[u].w & (unsigned)kIdxMask);
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
        const int idx = s0 + u;
        // branch-free radial part: invalid records (past the end / outside rc) run with weight 0
        const F4 e = ee[u];
        const bool valid = idx < nn && e.w != -1;
        const unsigned wbits = (unsigned)e.w;
        const int j = (int)(wbits & (unsigned)kIdxMask);
        const int t2 = valid ? (int)(wbits >> kIdxBits) : t1;
        const float x = e.x, y = e.y, z = e.z;
        float d, dinv;
        dist_and_inv(dot3f(x, x, y, y, z, z), d, dinv);
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
        const float rcinv = m.uniform_rc ? m.rcinv_r : fast_rcp(rc);
        const float dc = valid ? d : 0.5f * rc;
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
        const float wgt = valid ? dinv : 0.0f;
        const float fs = (s12 + s21) * wgt; // f12 - f21 = fs * r12
        const float bb = s21 * wgt;         // f21 = -bb * r12
        F[0] = fmaf(fs, x, F[0]);
        F[1] = fmaf(fs, y, F[1]);
        F[2] = fmaf(fs, z, F[2]);
        const float bx = bb * x, by = bb * y, bz = bb * z;
        W[0] -= x * bx;
        W[1] -= y * by;
        W[2] -= z * bz;
        W[3] -= x * by;
        W[4] -= x * bz;
        W[5] -= y * bz;

        if (WITH_ANG && valid && idx < na) {
          const unsigned short a = amap[(int64_t)idx * N + k];
          if (a != kNoSlot) {
            const int rs = rev[(int64_t)idx * N];
            const unsigned short ap = rs != (int)kNoSlot ? amap[(int64_t)rs * N + j] : kNoSlot;
            const F4 fa = f12[(int64_t)a * N + k];
            // j has no compact slot for this pair only if its per-step angular list overflowed
            // (MN_angular): that is reported through the overflow flag; never index with kNoSlot
            F4 fb;
            fb.x = fb.y = fb.z = 0.0f;
            fb.w = 0;
            if (ap != kNoSlot)
              fb = f12[(int64_t)ap * N + j];
            F[0] += fa.x - fb.x;
            F[1] += fa.y - fb.y;
            F[2] += fa.z - fb.z;
            Wa[0] += x * fb.x;
            Wa[1] += y * fb.y;
            Wa[2] += z * fb.z;
            Wa[3] += x * fb.y;
            Wa[4] += x * fb.z;
            Wa[5] += y * fb.z;
            Wa[6] += y * fb.x;
            Wa[7] += z * fb.x;
            Wa[8] += z * fb.y;
          }
        }
      }
    };
    int s0 = part * G;
    for (; s0 < na; s0 += kStride)
      chunk(s0, std::true_type{});
    for (; s0 < nn; s0 += kStride)
      chunk(s0, std::false_type{});

    if (PARTS > 1) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        F[d] += NEPMI_PAIR_XCHG(F[d]);
#pragma unroll
      for (int d = 0; d < 6; ++d)
        W[d] += NEPMI_PAIR_XCHG(W[d]);
#pragma unroll
      for (int d = 0; d < 9; ++d)
        Wa[d] += NEPMI_PAIR_XCHG(Wa[d]);
      if (part != 0)
        return;
    }
    double E = b.lvl[k] >= 2 ? (double)b.pe_i[k] : 0.0;
    double Fd[3] = {(double)F[0], (double)F[1], (double)F[2]};
    double Wd[9];
    Wd[0] = (double)(W[0] + Wa[0]);
    Wd[1] = (double)(W[1] + Wa[1]);
    Wd[2] = (double)(W[2] + Wa[2]);
    Wd[3] = (double)(W[3] + Wa[3]);
    Wd[4] = (double)(W[4] + Wa[4]);
    Wd[5] = (double)(W[5] + Wa[5]);
    Wd[6] = (double)(W[3] + Wa[6]);
    Wd[7] = (double)(W[4] + Wa[7]);
    Wd[8] = (double)(W[5] + Wa[8]);
    if (m.zbl_enabled && b.lvl[k] >= 2) { // (a reverse-mode ghost: its pair potential is its owner's business)
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

// ------------------------------------------------------------------------------------------------
// diagnostics
// ------------------------------------------------------------------------------------------------

// which: 0 radial (per step), 1 angular (per step), 2 Verlet skin list.  Caller indices, ascending.
struct ExportListsBody {
  Bufs b;
  int which;
  int* nn_out;
  int* nl_out;
  int64_t ld;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t i = b.perm[k];
    const int na = b.nn_ang[k], nb = b.nn_skin[k];
    int cnt = 0;
    const int total = which == 1 ? b.nn_angstep[k] : na + nb;
    for (int s = 0; s < total; ++s) {
      int j;
      if (which == 1) {
        j = (int)((unsigned)b.acomp[(int64_t)s * N + k].w & (unsigned)kIdxMask);
      } else {
        const int row = s < na ? s : b.MN_ang + (s - na);
        if (which == 0) {
          const int w = b.rstash[(int64_t)row * N + k].w;
          if (w == -1)
            continue;
          j = (int)((unsigned)w & (unsigned)kIdxMask);
        } else {
          j = s < na ? b.nl_ang[(int64_t)s * N + k] : b.nl_skin[(int64_t)(s - na) * N + k];
        }
      }
      const int jc = b.perm[j];
      if (cnt < ld) {
        // insertion into the ascending column
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

// q / Fp planes to caller order
struct ExportDescBody {
  Bufs b;
  int dim;
  float* q_out;
  float* fp_out;
  const int* dmap; // component d of the caller's arrays = component dmap[d] of the engine's (nullptr: the same)
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t i = b.perm[k];
    for (int d = 0; d < dim; ++d) {
      const int de = dmap ? dmap[d] : d;
      if (q_out)
        q_out[(int64_t)d * N + i] = b.q[(int64_t)de * N + b.tpos[k]];
      if (fp_out)
        fp_out[(int64_t)d * N + i] = b.fp[(int64_t)de * N + b.tpos[k]];
    }
  }
};

} // namespace nepmi
