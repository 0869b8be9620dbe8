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
#ifndef NEPMI_BIGWIN_RADIAL
#define NEPMI_BIGWIN_RADIAL 1024
#endif
#ifndef NEPMI_BIGWIN_SCATTER
#define NEPMI_BIGWIN_SCATTER 768
#endif
constexpr int kWinThreadsBig = NEPMI_BIGWIN_RADIAL;         // ... of the one-lane window kernels on windows beyond kBigWindowLds (engine.hip: nepmi_win2_kernel)
constexpr int kWinThreadsBigScatter = NEPMI_BIGWIN_SCATTER; // ... of the scatter kernel there (its 137 registers allow three wavefronts per SIMD)
constexpr size_t kBigWindowLds = 80 * 1024; // bytes of LDS per workgroup from which only one workgroup fits a CU
#ifndef NEPMI_MIDWIN
#define NEPMI_MIDWIN 1 // A/B switch: 512-thread workgroups for the radial pass of many-type models whose LDS use leaves room for two per CU
#endif
constexpr int kWinThreadsMid = 512;
constexpr size_t kMidWindowLds = 54 * 1024; // ... from which at most two fit
constexpr int kWinCells = 512;
constexpr int kWinMaxAtoms = 6656; // window capacity (LDS budget: 104 KB of records, 156 KB of positions + accumulators in the scatter form); larger windows take the gather path
                                   // (r6: 5000 -> 6656 puts C_2024_NEP4 in diamond, 6,100-6,400 slots, on the window kernels: 6.57 -> 5.01 ms/step at 512,000 atoms)

#if defined(__HIP_DEVICE_COMPILE__)
#define NEPMI_LDS(T) __attribute__((address_space(3))) T
#else
#define NEPMI_LDS(T) T
#endif

// LDS layout (bytes): int woff[513] | int wstart[512] | WinRec rec[wmax]
// compact = 1 (static window layout, Bufs::wtab): WinRec rec[wmax + 1] only -- rec[wmax] is the sentinel record
struct WinLayout {
  int wmax;
  int compact;
  NEPMI_HD int off_woff() const { return 0; }
  NEPMI_HD int off_wstart() const { return 2064; } // 513 ints, padded to 16 B
  NEPMI_HD int off_rec() const { return compact ? 0 : 2064 + 2048; }
  NEPMI_HD int bytes() const { return off_rec() + 16 * (wmax + (compact ? 1 : 0)); }
};

#ifndef NEPMI_CT_VEC_FORCE
#define NEPMI_CT_VEC_FORCE 0 // 1: the force assembly (FPJ form) reads its coefficient blocks with 16-byte ds_reads (block stride 52 floats for
                             // UNEP-v1: 53 KB, which no longer leaves room for two workgroups per CU next to its 34.8 KB window -- the
                             // counted rule then falls back to the gathered table rows, 1.54 -> 2.91 ms: profiles/r3z1_*); 0: element-wise, odd stride
#endif
// Radial coefficient table of many-type shapes in LDS (static layout): one block of (n_r+1)(k_r+1) floats per ordered type
// pair, padded to a multiple of four so that a lane reads its pair's block with 16-byte ds_reads (12 instead of 45 for UNEP-v1)
struct alignas(16) F4f {
  float x, y, z, w;
};
// Block stride in floats.  Lanes read the SAME element of DIFFERENT blocks at the same time, so the stride decides the bank
// conflicts: element-wise reads want an odd stride (every block starts on another bank); 16-byte reads want a multiple of four
// floats whose quarter is odd (16 lanes x 4 banks tile the 64 banks).  A stride of 48 floats -- the plain round-up of UNEP-v1's
// 45 -- puts every block on one of four bank offsets: r3g measured the radial pass at 2.15 ms against 1.13 ms with stride 45.
NEPMI_HD int ctab_block(int NR, int KR, bool vec)
{
  const int raw = (NR + 1) * (KR + 1);
  if (!vec)
    return raw | 1;
  const int b4 = (raw + 3) / 4;
  return 4 * (b4 | 1);
}
template <class LC>
NEPMI_HD void ctab_stage_padded(const ModelD& m, LC dst_bytes, int tid, int nth, bool vec)
{
  NEPMI_LDS(float)* ct = (NEPMI_LDS(float)*)dst_bytes;
  const int raw = (m.NR + 1) * (m.KR + 1), blk = ctab_block(m.NR, m.KR, vec), npair = m.T * m.T;
  if (const float* img = m.ctab_img[vec ? 1 : 0]) { // the image of the engine (upload_model): a plain copy
    const int n = npair * blk, n4 = n >> 2;
    const F4f* __restrict__ src4 = reinterpret_cast<const F4f*>(img);
    NEPMI_LDS(F4f)* d4 = (NEPMI_LDS(F4f)*)dst_bytes;
    for (int i = tid; i < n4; i += nth)
      d4[i] = src4[i];
    for (int i = 4 * n4 + tid; i < n; i += nth)
      ct[i] = img[i];
    return;
  }
  for (int i = tid; i < npair * blk; i += nth) {
    const int pr = i / blk, e = i - pr * blk;
    ct[i] = e < raw ? m.c_rad[pr * raw + e] : 0.0f;
  }
}
// g[n] = sum_k c[n][k] f[k] from one padded block
// VEC: the block is read with 16-byte ds_reads into registers first; else element by element as it is used (measured on UNEP-v1,
// r3g: the radial pass, which runs this once per candidate next to its bookkeeping, takes 2.15 ms with the wide reads against
// 1.13 ms without; the force assembly, which runs it twice per pair, keeps them)
template <class S, bool VEC, class LP>
NEPMI_HD void ctab_contract(LP blk_ptr, int NR, int KR, const float* f, float* g)
{
  if (S::fixed && VEC) {
    constexpr int RAW = (S::NRM + 1) * (S::KRM + 1), BLK = (RAW + 3) / 4 * 4;
    float v[BLK];
    NEPMI_LDS(const F4f)* p4 = (NEPMI_LDS(const F4f)*)blk_ptr;
#pragma unroll
    for (int i = 0; i < BLK / 4; ++i) {
      const F4f t = p4[i];
      v[4 * i] = t.x;
      v[4 * i + 1] = t.y;
      v[4 * i + 2] = t.z;
      v[4 * i + 3] = t.w;
    }
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n) {
      float gs = 0.0f;
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk)
        gs = fmaf(v[n * (S::KRM + 1) + kk], f[kk], gs);
      g[n] = gs;
    }
  } else {
    for (int n = 0; n <= NR; ++n) {
      float gs = 0.0f;
      for (int kk = 0; kk <= KR; ++kk)
        gs = fmaf(blk_ptr[n * (KR + 1) + kk], f[kk], gs);
      g[n] = gs;
    }
  }
}

// Staging, shared by the two passes (identical window contents and slot order in both): the records of the window
// cells' atoms are copied from Bufs::prec (fixed point, relative to the corner of the atom's own cell) with the
// integer offset of that cell from the window centre added -- no FP64, no search.
struct WinStage {
  BoxD box;
  Bufs b;
  WinLayout lay;

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

  // offset of window cell (wx, wy, wz) of brick (bx, by, bz) from the window centre (the corner shared by window
  // cells 3 and 4), grid units.  The cells do not tile the box exactly (the last one is wider, find_cell_id folds
  // the remainder into cell 0): a cell reached through a periodic wrap sits one lattice vector away, which is the
  // uniform cell spacing plus WinGeom::sv.
  NEPMI_HD void cell_offset(int bx, int by, int bz, int wx, int wy, int wz, int& qx, int& qy, int& qz) const
  {
    const int* cv = b.wg.cv;
    const int* sv = b.wg.sv;
    const int ux = wx - 4, uy = wy - 4, uz = wz - 4;
    const int cx = 4 * bx - 2 + wx, cy = 4 * by - 2 + wy, cz = 4 * bz - 2 + wz;
    const int sx = box.pbc[0] ? (cx < 0 ? -1 : (cx >= b.nbx ? 1 : 0)) : 0;
    const int sy = box.pbc[1] ? (cy < 0 ? -1 : (cy >= b.nby ? 1 : 0)) : 0;
    const int sz = box.pbc[2] ? (cz < 0 ? -1 : (cz >= b.nbz ? 1 : 0)) : 0;
    qx = ux * cv[0] + uy * cv[1] + uz * cv[2] + sx * sv[0] + sy * sv[1] + sz * sv[2];
    qy = ux * cv[3] + uy * cv[4] + uz * cv[5] + sx * sv[3] + sy * sv[4] + sz * sv[5];
    qz = ux * cv[6] + uy * cv[7] + uz * cv[8] + sx * sv[6] + sy * sv[7] + sz * sv[8];
  }

  // phase 3 (all threads, after the scan of woff): copy the window atoms, one cell per thread and turn
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const
  {
    NEPMI_LDS(const int)* woff = (NEPMI_LDS(const int)*)(lds + lay.off_woff());
    NEPMI_LDS(const int)* wstart = (NEPMI_LDS(const int)*)(lds + lay.off_wstart());
    NEPMI_LDS(WinRec)* rec = (NEPMI_LDS(WinRec)*)(lds + lay.off_rec());
    int bx, by, bz;
    brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += nth) {
      const int w0 = woff[wc];
      int cnt = woff[wc + 1] - w0;
      if (w0 + cnt > lay.wmax)
        cnt = lay.wmax - w0; // never beyond the LDS capacity (the engine sizes wmax from the fullest window)
      const int j0 = wstart[wc];
      int qx, qy, qz;
      cell_offset(bx, by, bz, wc & 7, (wc >> 3) & 7, wc >> 6, qx, qy, qz);
      for (int a = 0; a < cnt; a += 4) { // four records per round trip (a cell holds 3-4 atoms on average)
        WinRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          r[u] = b.prec[j0 + (a + u < cnt ? a + u : cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          r[u].x += qx;
          r[u].y += qy;
          r[u].z += qz;
          if (a + u < cnt)
            rec[w0 + a + u] = r[u];
        }
      }
    }
  }

  // Static layout: the window cells' first atoms and LDS offsets come from the table of the last rebuild (Bufs::wtab), so
  // staging is one pass without a scan: thread t copies window cells t, t + nth, ...; rec[wmax] = the sentinel record the
  // padded list words point at (0.875 R away along every axis: beyond every cutoff, within the 32-bit difference).
  template <class LC>
  NEPMI_HD void stage_direct(int64_t brick, LC lds, int tid, int nth) const
  {
    NEPMI_LDS(WinRec)* rec = (NEPMI_LDS(WinRec)*)(lds + lay.off_rec());
    const int* tab = b.wtab + brick * 1024;
    int bx, by, bz;
    brick_coords(brick, bx, by, bz);
    for (int wc = tid; wc < kWinCells; wc += nth) {
      const int j0 = tab[2 * wc], pk = tab[2 * wc + 1];
      const int w0 = pk & 0xFFFF;
      int cnt = pk >> 16;
      if (w0 + cnt > lay.wmax)
        cnt = lay.wmax > w0 ? lay.wmax - w0 : 0;
      if (cnt == 0)
        continue;
      int qx, qy, qz;
      cell_offset(bx, by, bz, wc & 7, (wc >> 3) & 7, wc >> 6, qx, qy, qz);
      for (int a = 0; a < cnt; a += 4) {
        WinRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          r[u] = b.prec[j0 + (a + u < cnt ? a + u : cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          r[u].x += qx;
          r[u].y += qy;
          r[u].z += qz;
          if (a + u < cnt)
            rec[w0 + a + u] = r[u];
        }
      }
    }
    if (tid == 0) {
      WinRec sr;
      sr.x = sr.y = sr.z = 0x38000000;
      sr.w = 0;
      rec[lay.wmax] = sr;
    }
  }

  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const
  {
    a0 = b.cell_count[brick * 64];
    a1 = b.cell_count[brick * 64 + 64];
  }

  // the brick's own atom k in the same frame: its cell is one of the 4x4x4 cells in the middle of the window
  NEPMI_HD void place_own(int64_t k, int& qx, int& qy, int& qz) const
  {
    const int l = b.kcell[k] & 63;
    cell_offset(0, 0, 0, (l & 3) + 2, ((l >> 2) & 3) + 2, (l >> 4) + 2, qx, qy, qz); // never through a wrap
    const WinRec r = b.prec[k];
    qx += r.x;
    qy += r.y;
    qz += r.z;
  }
};

// gpu_find_neighbor_ON1 (neighbor.cu:85-162) as a window kernel: the Verlet lists A (rc_a + skin) and B (up to rc_r + skin) of the
// brick's atoms from the fixed-point records of its 8x8x8-cell window in LDS -- the same 5x5x5-cell sweep, order, codes and list
// decisions as BuildListsBody (a decision closer to a cutoff than the band is retaken with the reference's arithmetic), but the
// ~490 candidate positions per atom are LDS reads instead of 32-byte gathers (BuildListsBody is bound by the address unit: one
// cache line per lane and candidate).  Scanned window layout (cell counts -> scan -> records): the window table of the static
// layout is built FROM these lists' codes, later in the rebuild.
struct BuildListsWinBody {
  WinStage st;
  static constexpr int kMinWavesPerEu = 1;
  NEPMI_HD int lds_bytes() const { return st.lay.bytes(); }
  NEPMI_HD int64_t map_brick(int64_t w) const { return w; }
  NEPMI_HD bool skip() const { return false; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const { st.stage_copy(brick, lds, tid, nth); }
  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const { st.brick_range(brick, a0, a1); }

  template <class LC>
  NEPMI_HD void compute(int64_t, int64_t k, LC lds) const
  {
    const Bufs& b = st.b;
    const BoxD& box = st.box;
    const int64_t N = b.N;
    NEPMI_LDS(const int)* woff = (NEPMI_LDS(const int)*)(lds + st.lay.off_woff());
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const PosQ p1 = b.posq[k];
    int ox, oy, oz;
    st.place_own(k, ox, oy, oz);
    const int c = b.kcell[k];
    int cx, cy, cz;
    cell_coords(b, c, cx, cy, cz);
    const int l = c & 63;
    const int wx0 = (l & 3) + 2, wy0 = ((l >> 2) & 3) + 2, wz0 = (l >> 4) + 2; // own cell in the window
    const float unit2 = b.wg.unit2, band = b.wg.band;
    const int lx = b.nbx > 1 ? 2 : 0, ly = b.nby > 1 ? 2 : 0, lz = b.nbz > 1 ? 2 : 0;
    int cnta = 0, cntb = 0;
    bool near_owned = false;
    // a row of the sweep (fixed kz, ky; kx = -lx .. lx) is ONE range of window slots: consecutive window cells hold consecutive
    // slots.  One loop per row instead of one per cell: a wavefront runs the longest of its lanes' loops, and the longest of 64
    // rows of ~20 atoms is relatively much shorter than the longest of 64 cells of ~4.
    int xlo = -lx, xhi = lx;
    if (!box.pbc[0]) { // cells beyond an open face hold nothing: the row stops at the box
      xlo = cx + xlo < 0 ? -cx : xlo;
      xhi = cx + xhi >= b.nbx ? b.nbx - 1 - cx : xhi;
    }
    for (int kz = -lz; kz <= lz; ++kz) {
      if (!box.pbc[2] && (cz + kz < 0 || cz + kz >= b.nbz))
        continue;
      for (int ky = -ly; ky <= ly; ++ky) {
        if (!box.pbc[1] && (cy + ky < 0 || cy + ky >= b.nby))
          continue;
        int wc = ((wz0 + kz) << 6) | ((wy0 + ky) << 3) | (wx0 + xlo);
        const int wc_end = wc + (xhi - xlo) + 1;
        const int s_end = woff[wc_end];
        int cell_lo = woff[wc], cell_hi = woff[wc + 1];
        // 32 slots at a time: first the decisions as two bit masks (no memory traffic but LDS reads), then one trip per ACCEPTED
        // slot.  A store instruction costs the address unit the same whether one lane or all take part, and some lane of the
        // wavefront accepts in nearly every trip of a loop over all slots (one candidate in five is a neighbour): storing inside
        // that loop issued ~1,400 store instructions per wavefront, this form issues ~400.
        for (int base = cell_lo; base < s_end; base += 32) {
          const int nch = s_end - base < 32 ? s_end - base : 32;
          unsigned ms = 0u, ma = 0u;
          for (int i2 = 0; i2 < nch; ++i2) {
            const WinRec r = wrec[base + i2];
            const int j = (int)((unsigned)r.w & (unsigned)kIdxMask);
            const float fx = (float)(r.x - ox), fy = (float)(r.y - oy), fz = (float)(r.z - oz);
            const float d2 = dot3f(fx, fx, fy, fy, fz, fz) * unit2;
            const float es = d2 - b.rc_skin_sq, ea = d2 - b.rc_askin_sq;
            bool in_skin = es < 0.0f, in_a = ea < 0.0f;
            if (fminf(fabsf(es), fabsf(ea)) < band && j != (int)k) { // the reference's arithmetic decides (rare)
              float x, y, z;
              const float d2e = pair_geometry(box, p1, b.posq[j], x, y, z);
              in_skin = d2e < b.rc_skin_sq;
              in_a = d2e < b.rc_askin_sq;
            }
            in_skin = in_skin && j != (int)k;
            ms |= (in_skin ? 1u : 0u) << i2;
            ma |= ((in_skin && in_a) ? 1u : 0u) << i2;
          }
          while (ms != 0u) {
            const int i2 = __builtin_ctz(ms);
            ms &= ms - 1u;
            const int s = base + i2;
            while (s >= cell_hi) { // the cell of this slot (slots ascend: the cursor only moves forward; empty cells are stepped over)
              ++wc;
              cell_lo = cell_hi;
              cell_hi = woff[wc + 1];
            }
            const int j = (int)((unsigned)wrec[s].w & (unsigned)kIdxMask);
            const unsigned short code = (unsigned short)((wc << 7) | ((s - cell_lo) & 127));
            if ((ma >> i2) & 1u) {
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

constexpr int kWinG = 4; // candidates whose LDS look-ups and arithmetic are interleaved

// build-time switches for A/B measurements (profiles/ab_variants.sh); the defaults are the product
#ifndef NEPMI_RW_PACK
#define NEPMI_RW_PACK 1
#endif
#ifndef NEPMI_FW_PACK
#define NEPMI_FW_PACK 1
#endif
#ifndef NEPMI_FW_WAVES
#define NEPMI_FW_WAVES 4
#endif
#ifndef NEPMI_RW_WAVES
#define NEPMI_RW_WAVES 1
#endif
#ifndef NEPMI_RW_PIPE
#define NEPMI_RW_PIPE 0 // 1: the window records of the next chunk are read while this chunk is processed (measured: no gain)
#endif

template <class S>
struct RadialWinBody {
  WinStage st;
  ModelD m;
  int first;          // workgroup w runs brick_order[first + w] (first < 0: brick w)
  const int* frozen;  // fused run loops: a non-zero value means "a list rebuild is pending": do nothing
  static constexpr int kMinWavesPerEu = NEPMI_RW_WAVES;

  // Shapes without register-resident per-type sums (many types: UNEP-v1 has 16) contract the radial coefficients
  // c[t1][t2][n][k] per pair, with (t1, t2) different from lane to lane: 45 loads per pair that hit up to 64 different
  // cache lines each.  The whole table (T^2 (n_r+1)(k_r+1) floats, 46 KB for UNEP-v1) is therefore staged in LDS behind
  // the window when two workgroups per CU still fit (<= 80 KB together); the loads become LDS reads.
  NEPMI_HD int ctab_floats() const { return S::TS > 0 ? 0 : m.T * m.T * (m.NR + 1) * (m.KR + 1); }
  NEPMI_HD int ctab_offset() const { return (st.lay.bytes() + 15) / 16 * 16; }
#ifndef NEPMI_RW_CTAB
#define NEPMI_RW_CTAB 1 // A/B switch (profiles/ab_variants.sh)
#endif
  NEPMI_HD bool ctab_on() const { return NEPMI_RW_CTAB && S::TS == 0 && ctab_offset() + 4 * ctab_floats() <= 80 * 1024; }
  NEPMI_HD int lds_bytes() const { return ctab_on() ? ctab_offset() + 4 * ctab_floats() : st.lay.bytes(); }
  NEPMI_HD int64_t map_brick(int64_t w) const { return first < 0 ? w : (int64_t)st.b.brick_order[first + w]; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const
  {
    st.stage_copy(brick, lds, tid, nth);
    if (ctab_on()) {
      NEPMI_LDS(float)* ct = (NEPMI_LDS(float)*)(lds + ctab_offset());
      const int nf = ctab_floats();
      for (int i = tid; i < nf; i += nth)
        ct[i] = m.c_rad[i];
    }
  }
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
    const bool ctab = ctab_on();
    NEPMI_LDS(const float)* ctab_lds = (NEPMI_LDS(const float)*)(lds + ctab_offset());
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(k, ox, oy, oz);
    const float rc1 = m.rc_r[t1], rca1 = m.rc_a[t1];
    const float unit = st.b.wg.unit, unit2 = st.b.wg.unit2, band = st.b.wg.band;
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
    const bool owned = b.lvl[k] >= b.lvl_force; // ("owned" = its force assembly runs and reads the compact list)
    unsigned am_cur = 0u; // membership bits of the current 32 entries of list A (Bufs::amask)
    int cnt = 0, cnt1 = 0, ca = 0; // cnt: entries at the front of ccode, cnt1: at its back (type-1 neighbours)
    F4* __restrict__ acomp = b.acomp + k;
    unsigned short* __restrict__ amap = b.amap + k;
    unsigned short* __restrict__ aidx = b.aidx + k;
    unsigned short* __restrict__ ccode = b.ccode + k;

    // list decisions and bookkeeping of one candidate (scalar): LDS slot -> integer pair vector, d^2, cutoff tests
    // (retaken exactly inside the band), compact angular record, compact radial entry
    struct Cand {
      float fx, fy, fz, d2;
      int t2;
      bool inside;
    };
    auto decide = [&](const int slot, const WinRec r, const int idx, const bool live, auto in_list_a)
                    __attribute__((always_inline)) -> Cand {
      constexpr bool LIST_A = decltype(in_list_a)::value;
      Cand c;
      c.fx = (float)(r.x - ox);
      c.fy = (float)(r.y - oy);
      c.fz = (float)(r.z - oz);
      c.d2 = dot3f(c.fx, c.fx, c.fy, c.fy, c.fz, c.fz) * unit2;
      c.t2 = (int)((unsigned)r.w >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[c.t2]) * 0.5f;
      const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[c.t2]) * 0.5f;
      const float er = c.d2 - rc * rc, ea = c.d2 - rca * rca;
      bool inside = er < 0.0f;
      bool ang = LIST_A && ea < 0.0f;
      // within the band of a cutoff the decision is retaken with the reference's arithmetic (rare)
      const float near = LIST_A ? fminf(fabsf(er), fabsf(ea)) : fabsf(er);
      if (live && near < band) {
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
            e.x = c.fx * unit;
            e.y = c.fy * unit;
            e.z = c.fz * unit;
            e.w = r.w;
            acomp[(int64_t)ca * N] = e;
            aidx[(int64_t)ca * N] = b.rev_ang[(int64_t)idx * N + k]; // reverse slot of this pair in j's list A
            cs = (unsigned short)ca;
            am_cur |= 1u << (idx & 31); // membership bit (Bufs::amask); the walk stores the word every 32 entries
          }
          ++ca;
        }
        if (!b.use_amask)
          amap[(int64_t)idx * N] = cs;
      }
      if (inside) {
        // two-type shapes: the compact list is partitioned by the neighbour's type (front / back), so that the force
        // assembly walks type-pure segments with one row of its own table in registers
        int pos;
        if (S::TS == 2 && c.t2 == 1) {
          pos = b.MN_rad - 1 - cnt1;
          ++cnt1;
        } else {
          pos = cnt;
          ++cnt;
        }
        if (owned && cnt + cnt1 <= b.MN_rad) // the compact list is read by the force assembly: owned atoms only
          ccode[(int64_t)pos * N] = (unsigned short)slot;
      }
      c.inside = inside;
      return c;
    };

    // accumulation of one candidate, one-wide: shapes without register-resident per-type sums (many types,
    // run-time shape) contract the coefficients per pair
    auto accumulate1 = [&](const Cand& c) __attribute__((always_inline)) {
      if (S::TS > 0) { // one-wide form of accumulate2 (kept for A/B measurements)
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[c.t2]) * 0.5f;
        float d, dinv;
        dist_and_inv(c.d2, d, dinv);
        const float rcinv = m.uniform_rc ? m.rcinv_r : fast_rcp(rc);
        const float dc = d < rc ? d : rc;
        float fc;
        cutoff_fc(rcinv, dc, fc);
        float fn[S::KRM + 1];
        basis_fn<S::KRM>(rcinv, dc, fc, fn);
#pragma unroll
        for (int t = 0; t < TSM; ++t) {
          const float w = (c.inside && (TSM == 1 || c.t2 == t)) ? 1.0f : 0.0f;
#pragma unroll
          for (int kk = 0; kk <= S::KRM; ++kk)
            Ssum[t][kk] = fmaf(w, fn[kk], Ssum[t][kk]);
        }
        return;
      }
      if (!c.inside)
        return;
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[c.t2]) * 0.5f;
      float d, dinv;
      dist_and_inv(c.d2, d, dinv);
      const float rcinv = fast_rcp(rc);
      const float dc = d < rc ? d : rc;
      float fc;
      cutoff_fc(rcinv, dc, fc);
      float fn[S::KRM + 1];
      if (S::fixed)
        basis_fn<S::KRM>(rcinv, dc, fc, fn);
      else
        basis_fn_rt(KR, rcinv, dc, fc, fn);
      const int coff = (t1 * m.T + c.t2) * (NR + 1) * (KR + 1);
      if (ctab) {
        NEPMI_LDS(const float)* cc = ctab_lds + coff;
        for (int n = 0; n <= NR; ++n) {
          float gsum = 0.0f;
          for (int kk = 0; kk <= KR; ++kk)
            gsum += fn[kk] * cc[n * (KR + 1) + kk];
          q[n] += gsum;
        }
      } else {
        const float* cc = m.c_rad + coff;
        for (int n = 0; n <= NR; ++n) {
          float gsum = 0.0f;
          for (int kk = 0; kk <= KR; ++kk)
            gsum += fn[kk] * cc[n * (KR + 1) + kk];
          q[n] += gsum;
        }
      }
    };
    // accumulation of two candidates side by side (f2: packed FP32), branch-free: entries outside the cutoff run
    // the same arithmetic with weight 0 (the envelope is evaluated at min(d, rc) and stays finite)
    f2 Ssum2[TSM][S::KRM + 1];
#pragma unroll
    for (int t = 0; t < TSM; ++t)
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk)
        Ssum2[t][kk] = bc2(0.0f);
    auto accumulate2 = [&](const Cand& c0, const Cand& c1) __attribute__((always_inline)) {
      float rc0, rc1v, ri0, ri1;
      if (m.uniform_rc) {
        rc0 = rc1v = m.rc_r_max;
        ri0 = ri1 = m.rcinv_r;
      } else {
        rc0 = (rc1 + m.rc_r[c0.t2]) * 0.5f;
        rc1v = (rc1 + m.rc_r[c1.t2]) * 0.5f;
        ri0 = fast_rcp(rc0);
        ri1 = fast_rcp(rc1v);
      }
      float d0, d1, i0, i1;
      dist_and_inv(c0.d2, d0, i0);
      dist_and_inv(c1.d2, d1, i1);
      const f2 dc = mk2(d0 < rc0 ? d0 : rc0, d1 < rc1v ? d1 : rc1v);
      const f2 rcinv = mk2(ri0, ri1);
      f2 fc;
      cutoff_fc_v(rcinv, dc, fc);
      f2 fn[S::KRM + 1];
      basis_fn_v<S::KRM>(rcinv, dc, fc, fn);
#pragma unroll
      for (int t = 0; t < TSM; ++t) {
        const f2 w = mk2((c0.inside && (TSM == 1 || c0.t2 == t)) ? 1.0f : 0.0f,
                         (c1.inside && (TSM == 1 || c1.t2 == t)) ? 1.0f : 0.0f);
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
          Ssum2[t][kk] = vfma(w, fn[kk], Ssum2[t][kk]);
      }
    };

    // walk a list in chunks of kWinG: the codes of chunk c + 2 are on their way from the list while chunk c is
    // decided and accumulated; the window records of a chunk (slot offset, then the record: two dependent LDS reads)
    // are issued together, ahead of the branchy bookkeeping -- one LDS latency per chunk instead of two per candidate.
    // NEPMI_RW_PIPE = 1 additionally reads chunk c + 1's records before chunk c is processed (16 more registers).
    auto walk = [&](const unsigned short* __restrict__ codes, const int nn, auto in_list_a) __attribute__((always_inline)) {
      auto load_codes = [&](int s0, unsigned* cc) {
#pragma unroll
        for (int u = 0; u < kWinG; ++u) {
          const int idx = s0 + u;
          cc[u] = nn > 0 ? codes[(int64_t)(idx < nn ? idx : nn - 1) * N] : 0u;
        }
      };
      auto load_recs = [&](const unsigned* cc, int* sl, WinRec* rr) {
#pragma unroll
        for (int u = 0; u < kWinG; ++u)
          sl[u] = woff[cc[u] >> 7] + (int)(cc[u] & 127u);
#pragma unroll
        for (int u = 0; u < kWinG; ++u)
          rr[u] = wrec[sl[u]];
      };
      unsigned c1[kWinG], c2[kWinG];
      int sl0[kWinG], sl1[kWinG];
      WinRec r0[kWinG], r1[kWinG];
      load_codes(0, c1);
      load_recs(c1, sl0, r0);
      load_codes(kWinG, c1);
      for (int s0 = 0; s0 < nn; s0 += kWinG) {
        if (decltype(in_list_a)::value && b.use_amask && (s0 & 31) == 0 && s0 > 0) { // a word of the mask is complete
          b.amask[4 * k + (s0 >> 5) - 1] = am_cur;
          am_cur = 0u;
        }
        load_codes(s0 + 2 * kWinG, c2);
        if (NEPMI_RW_PIPE)
          load_recs(c1, sl1, r1);
        Cand c[kWinG];
#pragma unroll
        for (int u = 0; u < kWinG; ++u)
          c[u] = decide(sl0[u], r0[u], s0 + u, s0 + u < nn, in_list_a);
        if (S::TS > 0 && NEPMI_RW_PACK) {
#pragma unroll
          for (int u = 0; u < kWinG; u += 2)
            accumulate2(c[u], c[u + 1]);
        } else {
#pragma unroll
          for (int u = 0; u < kWinG; ++u)
            accumulate1(c[u]);
        }
        if (!NEPMI_RW_PIPE)
          load_recs(c1, sl1, r1);
#pragma unroll
        for (int u = 0; u < kWinG; ++u) {
          sl0[u] = sl1[u];
          r0[u] = r1[u];
          c1[u] = c2[u];
        }
      }
    };
    walk(b.code_ang + k, na, std::true_type{});
    if (b.use_amask) { // the last (partial) word and zeros behind it; na <= 128 is what use_amask guarantees
      const int wlast = na > 0 ? (na - 1) >> 5 : 0;
      for (int w = 0; w < 4; ++w)
        if (w >= wlast)
          b.amask[4 * k + w] = w == wlast ? am_cur : 0u;
    }
    walk(b.code_skin + k, nbn, std::false_type{});
    if (S::TS > 0 && NEPMI_RW_PACK) {
#pragma unroll
      for (int t = 0; t < TSM; ++t)
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
          Ssum[t][kk] = Ssum2[t][kk].x + Ssum2[t][kk].y;
    }

    if (ca > b.MN_acomp || cnt + cnt1 > b.MN_rad) {
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 4);
      ca = ca > b.MN_acomp ? b.MN_acomp : ca;
    }
    b.nn_rad[k] = cnt + cnt1;
    b.nn_t0[k] = cnt;
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

// ---------------------------------------------------------------------------------------------------------------------
// RadialWin2Body: the one-lane radial pass on the STATIC window layout (Bufs::wtab / wcode / wseg, written once per list
// rebuild).  Same results as RadialWinBody -- the same lists bit for bit, the same sums up to their order -- with less
// work per candidate:
//   * a Verlet entry is the LDS slot itself (no cell-offset look-up: one ds_read_b128 per candidate, no scan and one
//     barrier instead of three in the staging, 4 KB less LDS per workgroup);
//   * four entries arrive as one 8-byte word (a quarter of the list-load instructions);
//   * list B of a two-type model is stored as two type-pure streams, walked side by side: the two halves of a packed
//     FP32 value carry one neighbour of type 0 and one of type 1, so the basis sums of both types are ONE packed
//     accumulator row (7 v_pk_fma per two candidates instead of 14, no per-type weights), and the compact radial list
//     grows at its front and at its back without a select;
//   * segments are padded with a sentinel slot whose record lies beyond every cutoff: no "live" predicate in list B;
//   * the exact retake of a decision inside the band of a cutoff is one test per word, not one per candidate.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef NEPMI_RW2_WAVES
#define NEPMI_RW2_WAVES 1
#endif
#ifndef NEPMI_CW
#define NEPMI_CW 0 // 1: the compact radial list of the static layout is written / read as words of four slots (Bufs::cword) and
                   // the force assembly walks them software-pipelined (win_force_words).  Measured on PbTe 1 M atoms
                   // (profiles/r3cd_ab_window_variants.txt, r3e): radial pass 0.364 -> 0.359 ms; force assembly 0.517 (2-byte list,
                   // no pipeline) vs 0.508 (words, pipelined, 3 waves) vs 1.02 (4 waves: the two stages spill); carbon 0.79 ->
                   // 0.98.  Off.
#endif
#ifndef NEPMI_BUILD_WIN
#define NEPMI_BUILD_WIN 1 // the Verlet lists of a rebuild from LDS windows (BuildListsWinBody) where the window kernels apply; 0: BuildListsBody
#endif
#ifndef NEPMI_RW2_ABL
#define NEPMI_RW2_ABL 0 // ablation builds (timings only): 1 = no compact-list stores, 2 = no stores and no counters, 3 = no angular record stores
#endif
#ifndef NEPMI_RW2_HALF
#define NEPMI_RW2_HALF 1 // 1: a word pair is processed as two halves of 2 + 2 candidates (fewer live registers), 0: 4 + 4 at once
#endif
struct alignas(8) U2w { // four 16-bit LDS slots
  unsigned lo, hi;
};

// The compact radial list as WAVE-SYNCHRONOUS words (RadialWin2Body<S, 1> writes them, the scatter-form force assembly walks
// them).  The slot-major compact list (Bufs::ccode) costs the radial pass a third of its time although it is 130 bytes per
// atom: the lanes of a wavefront accept different candidates, so their list cursors drift apart (like the square root of the
// number of candidates), every 2-byte store instruction touches half a dozen partially written lines, and a store
// instruction occupies the address unit the same with one lane as with 64 (profiles/r4m_ab_radial_stores.txt,
// r4ad_ab_radial_push.txt: 0.63 GB of HBM-side write traffic for 0.27 GB of list).  Here an accepted slot waits in a queue of
// eight 16-bit entries in the lane's registers; when any lane of the wavefront holds more than four after a word of
// candidates, EVERY lane stores its four oldest entries as one 8-byte word -- row r of the stream, r the same on all lanes,
// 512 contiguous bytes per wavefront -- a lane with fewer than four fills the word with the sentinel slot and starts again from
// an empty queue.  The consumer is one lane per atom in lockstep too: it walks max-over-the-wavefront entries whatever the
// layout, and the padded stream has exactly that many rows (simulated and measured: rows = ceil(longest list / 4) for the
// two-type streams of ~11 Verlet words, + 4 % for 22), so the padding costs it nothing.  A fifth of the store instructions
// of the 2-byte list, every one of them whole lines.  The ORDER of the entries inside an atom's list is not the Verlet order
// any more where a lane padded; nothing downstream depends on it (integer accumulators; nepmi_neighbors_export reads the
// pair records).
struct SyncFifo {
  unsigned f0, f1, f2, f3; // entries 0 (oldest) .. 7 (newest): two per register; the p newest are waiting
  int p, rows;
  NEPMI_HD void init()
  {
    f0 = f1 = f2 = f3 = 0u;
    p = 0;
    rows = 0;
  }
  static NEPMI_HD unsigned funnel16(unsigned hi, unsigned lo) // (hi:lo) >> 16
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, 16u);
#else
    return (lo >> 16) | (hi << 16);
#endif
  }
  NEPMI_HD void push(unsigned slot) // the queue moves down by one entry, the new one enters at the top: four v_alignbit_b32
  {
    f0 = funnel16(f1, f0);
    f1 = funnel16(f2, f1);
    f2 = funnel16(f3, f2);
    f3 = funnel16(slot, f3);
    ++p;
  }
  // the oldest waiting entry alone (none waiting: the sentinel)
  NEPMI_HD unsigned pop_one(unsigned sent)
  {
    if (p <= 0)
      return sent;
    const int s = 8 - p; // place of the oldest waiting entry
    const unsigned r01 = (s & 2) ? f1 : f0, r23 = (s & 2) ? f3 : f2;
    const unsigned r = (s & 4) ? r23 : r01;
    --p;
    return (s & 1) ? (r >> 16) : (r & 0xFFFFu);
  }
  // the four oldest waiting entries as one word (fewer than four: padded with the sentinel, and the queue is empty afterwards)
  NEPMI_HD unsigned long long pop_word(unsigned long long sent64)
  {
    const int s = 8 - p; // place of the oldest waiting entry
    const unsigned long long lo01 = ((unsigned long long)f1 << 32) | f0, hi23 = ((unsigned long long)f3 << 32) | f2;
    const unsigned long long lo = (s & 4) ? hi23 : lo01, hi = (s & 4) ? 0ull : hi23;
    const int sh = 16 * (s & 3);
    unsigned long long w = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    if (p < 4) {
      const unsigned long long keep = (1ull << (16 * p)) - 1ull; // (p = 0: nothing is kept -- s = 8 read entries that have left)
      w = (w & keep) | (sent64 & ~keep);
      p = 0;
    } else {
      p -= 4;
    }
    return w;
  }
};

template <class S, int SYNC = 0>
struct RadialWin2Body {
  WinStage st;
  ModelD m;
  int first;         // workgroup w runs brick_order[first + w] (first < 0: brick w)
  const int* frozen;
  static constexpr int kMinWavesPerEu = NEPMI_RW2_WAVES;
  static constexpr bool kBigWindows = S::TS > 0; // 1,024-thread workgroups on windows beyond kBigWindowLds (engine.hip: launch_win2)
  static constexpr bool kMidWindows = NEPMI_MIDWIN != 0 && S::TS == 0 && S::fixed && SYNC != 0;

  NEPMI_HD int ctab_floats() const { return S::TS > 0 ? 0 : m.T * m.T * ctab_block(m.NR, m.KR, false); }
  NEPMI_HD int ctab_offset() const { return (st.lay.bytes() + 15) / 16 * 16; }
  NEPMI_HD bool ctab_on() const { return NEPMI_RW_CTAB && S::TS == 0 && ctab_offset() + 4 * ctab_floats() <= 80 * 1024; }
  NEPMI_HD int lds_bytes() const { return ctab_on() ? ctab_offset() + 4 * ctab_floats() : st.lay.bytes(); }
  NEPMI_HD int64_t map_brick(int64_t w) const { return first < 0 ? w : (int64_t)st.b.brick_order[first + w]; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage(int64_t brick, LC lds, int tid, int nth) const
  {
    if (st.b.brick_live && !st.b.brick_live[brick])
      return; // a brick of the outer ghost ring: compute() returns for every one of its atoms before it looks at the window
    st.stage_direct(brick, lds, tid, nth);
    if (ctab_on())
      ctab_stage_padded(m, lds + ctab_offset(), tid, nth, false);
  }
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
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const bool ctab = ctab_on();
    NEPMI_LDS(const float)* ctab_lds = (NEPMI_LDS(const float)*)(lds + ctab_offset());
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(k, ox, oy, oz);
    const float rc1 = m.rc_r[t1], rca1 = m.rc_a[t1];
    const float unit = st.b.wg.unit, unit2 = st.b.wg.unit2, band = st.b.wg.band;
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    constexpr bool ZIP = S::TS == 2; // list B as two type-pure streams side by side
    float q[S::NRM + 1];
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n)
      q[n] = 0.0f;

    const int na = b.nn_ang[k];
    const bool owned = b.lvl[k] >= (b.compact_all ? 1 : b.lvl_force); // ("owned" = its force assembly runs and reads the compact list)
    const int seg = b.wseg[k];
    const int wa = seg & 255, wb = (seg >> 8) & 255; // words of list A; words (ZIP: word pairs) of list B
    unsigned am[4] = {0u, 0u, 0u, 0u}; // membership bits of list A (Bufs::amask)
    int cnt = 0, cnt1 = 0, ca = 0;     // cnt: entries at the front of ccode, cnt1: at its back (type-1 neighbours)
    F4* __restrict__ acomp = b.acomp + k;
    unsigned short* __restrict__ amap = b.amap + k;
    unsigned short* __restrict__ aidx = b.aidx + k;
    unsigned short* __restrict__ ccode = b.ccode + k;
    const U2w* __restrict__ words = reinterpret_cast<const U2w*>(b.wcode) + k;

    struct Cand {
      float fx, fy, fz, d2;
      int rw, slot;
      bool inside, ang;
    };
    // geometry and list decisions from the fixed-point record; *nearband: the decision has to be retaken exactly
    auto judge = [&](int slot, bool list_a, bool& nearband) __attribute__((always_inline)) -> Cand {
      Cand c;
      const WinRec r = wrec[slot];
      c.slot = slot;
      c.rw = r.w;
      c.fx = (float)(r.x - ox);
      c.fy = (float)(r.y - oy);
      c.fz = (float)(r.z - oz);
      c.d2 = dot3f(c.fx, c.fx, c.fy, c.fy, c.fz, c.fz) * unit2;
      const int t2 = (int)((unsigned)r.w >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
      const float er = c.d2 - rc * rc;
      c.inside = er < 0.0f;
      float near = fabsf(er);
      c.ang = false;
      if (list_a) {
        const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[t2]) * 0.5f;
        const float ea = c.d2 - rca * rca;
        c.ang = ea < 0.0f;
        near = fminf(near, fabsf(ea));
      }
      nearband = nearband || near < band;
      return c;
    };
    // the reference's arithmetic decides (pair_geometry: FP64 difference, float minimum image); rare
    auto retake = [&](Cand& c, bool list_a) __attribute__((always_inline)) {
      const int t2 = (int)((unsigned)c.rw >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
      const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[t2]) * 0.5f;
      const float er = c.d2 - rc * rc, ea = c.d2 - rca * rca;
      const float near = list_a ? fminf(fabsf(er), fabsf(ea)) : fabsf(er);
      if (near < band && c.slot != b.wsent) {
        float ex, ey, ez;
        const float d2e = pair_geometry(st.box, p1, b.posq[(unsigned)c.rw & (unsigned)kIdxMask], ex, ey, ez);
        c.inside = d2e < rc * rc;
        c.ang = list_a && d2e < rca * rca;
      }
    };
    // The compact radial list.  NEPMI_CW: words of four slots, staged in two registers per stream and stored as 8 bytes when
    // full -- a quarter of the store instructions (the window kernels are bound by the number of vector-memory
    // instructions); the front stream (neighbours of type 0, or all) fills rows 0, 1, ... of
    // Bufs::cword, the back stream (type 1) rows MN_cw, MN_cw + 1, ...; the last word of a stream is padded with the
    // sentinel slot, which the force assembly evaluates to zero.
    U2w* __restrict__ cword = reinterpret_cast<U2w*>(b.cword) + k;
    unsigned long long accf = 0ull, accb = 0ull;
    unsigned mcur = 0u; // the mask word being filled (Bufs::rmaskA / rmaskB)
    // one-wide accumulation with the coefficients contracted per pair (many types / run-time shape)
    auto accumulate1 = [&](const Cand& c) __attribute__((always_inline)) {
      if (!c.inside)
        return;
      const int t2 = (int)((unsigned)c.rw >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
      float d, dinv;
      dist_and_inv(c.d2, d, dinv);
      const float rcinv = fast_rcp(rc);
      const float dc = d < rc ? d : rc;
      float fc;
      cutoff_fc(rcinv, dc, fc);
      float fn[S::KRM + 1];
      if (S::fixed)
        basis_fn<S::KRM>(rcinv, dc, fc, fn);
      else
        basis_fn_rt(KR, rcinv, dc, fc, fn);
      if (ctab) {
        float g[S::NRM + 1];
        ctab_contract<S, false>(ctab_lds + (t1 * m.T + t2) * ctab_block(NR, KR, false), NR, KR, fn, g);
#pragma unroll
        for (int n = 0; n <= S::NRM; ++n) {
          if (!S::fixed && n > NR)
            break;
          q[n] += g[n];
        }
      } else {
        const float* cc = m.c_rad + (t1 * m.T + t2) * (NR + 1) * (KR + 1);
        for (int n = 0; n <= NR; ++n) {
          float gsum = 0.0f;
          for (int kk = 0; kk <= KR; ++kk)
            gsum += fn[kk] * cc[n * (KR + 1) + kk];
          q[n] += gsum;
        }
      }
    };
    // The same for the NC candidates of a word AT ONCE, without a branch (compiled many-type shapes with the table in LDS): a
    // candidate outside the cutoff is evaluated at d = rc with weight zero and adds exact zeros, in the same order as the
    // one-by-one form -- the sums are the same bit for bit.  Why: accumulate1 sits behind `if (inside)`, so the chains of the four
    // candidates of a word (record -> distance -> 9 basis functions -> 45 table reads and multiply-adds) run one after the
    // other.  Side by side the four chains are independent and interleave -- measured SLOWER (profiles/r6g_ab_wide.txt: UNEP-v1
    // 1 M atoms, radial pass 1.196 ms against 0.983): the quarter of the candidates that lies in the skin is then contracted too,
    // and the kernel's time follows the number of table reads (the LDS pipe is active 47 % of the kernel, two thirds of that
    // bank conflicts: 64 lanes gather from 256 random blocks), not their latency.  Off (NEPMI_RW2_WIDE).
    auto accumulate_wide = [&](const Cand* c, const int NC) __attribute__((always_inline)) {
      constexpr int W = 4;
      float fn[W][S::KRM + 1];
      NEPMI_LDS(const float)* blk[W];
#pragma unroll
      for (int u = 0; u < W; ++u) {
        if (u >= NC)
          break;
        const int t2 = (int)((unsigned)c[u].rw >> kIdxBits);
        const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[t2]) * 0.5f;
        float d, dinv;
        dist_and_inv(c[u].d2, d, dinv);
        const float rcinv = fast_rcp(rc);
        const float dc = d < rc ? d : rc;
        float fc;
        cutoff_fc(rcinv, dc, fc);
        fc *= c[u].inside ? 1.0f : 0.0f;
        basis_fn<S::KRM>(rcinv, dc, fc, fn[u]);
        blk[u] = ctab_lds + (t1 * m.T + t2) * ctab_block(NR, KR, false);
      }
      float g[W][S::NRM + 1];
#pragma unroll
      for (int n = 0; n <= S::NRM; ++n) {
#pragma unroll
        for (int u = 0; u < W; ++u)
          g[u][n] = 0.0f;
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
#pragma unroll
          for (int u = 0; u < W; ++u)
            if (u < NC)
              g[u][n] = fmaf(blk[u][n * (S::KRM + 1) + kk], fn[u][kk], g[u][n]);
      }
#pragma unroll
      for (int u = 0; u < W; ++u) {
        if (u >= NC)
          break;
#pragma unroll
        for (int n = 0; n <= S::NRM; ++n)
          q[n] += g[u][n];
      }
    };
#ifndef NEPMI_RW2_WIDE
#define NEPMI_RW2_WIDE 0
#endif
    const bool wide = NEPMI_RW2_WIDE && S::fixed && S::TS == 0 && ctab;
    // SYNC: the compact list as WAVE-SYNCHRONOUS words (see SyncFifo above): the accepted slots wait in the lane's queue and
    // every lane of the wavefront stores one 8-byte word of four at the same time
    SyncFifo q0, q1;
    q0.init();
    q1.init();
    U2w* __restrict__ sync0 = cword;
    U2w* __restrict__ sync1 = cword + (int64_t)b.MN_cw * N;
    const unsigned long long sent64 = 0x0001000100010001ull * (unsigned long long)(unsigned)b.wsent;
    // (Measured and dropped, r6c: many-type shapes evaluating a neighbour when its word LEAVES the queue -- basis functions and the
    // c[t1][t2] contraction once per row of the padded list, 74 instead of 96 times per atom for UNEP-v1: 1.02 ms against 0.99.
    // Neither the compact-list stores nor the number of contractions bind that kernel; two workgroups per CU -- the 46 KB
    // coefficient table next to the window -- leave the walk's dependent chain of LDS reads uncovered.)
    auto sync_emit = [&](SyncFifo& f, U2w* __restrict__& at) __attribute__((always_inline)) {
      const unsigned long long w = f.pop_word(sent64);
      if (f.rows < b.MN_cw) {
        U2w v;
        v.lo = (unsigned)w;
        v.hi = (unsigned)(w >> 32);
        *at = v;
        at += N;
      }
      ++f.rows;
    };
    // after a word of candidates: no lane may enter the next one with more than four entries waiting
    auto sync_check = [&]() __attribute__((always_inline)) {
      if (SYNC) {
        if (NEPMI_WAVE_ANY(q0.p > 4))
          sync_emit(q0, sync0);
        if (S::TS == 2 && NEPMI_WAVE_ANY(q1.p > 4))
          sync_emit(q1, sync1);
      }
    };
    // The angular records of list A the same way (compiled shapes): the accepted slots wait in a third queue, and ROW r of
    // acomp / aslot is stored by every lane of the wavefront at once -- the record rebuilt from the LDS window at that moment
    // (16 contiguous bytes per lane: 1 KB per wavefront and row), a lane with nothing waiting stores a null record
    // (kNullRecord / the sentinel slot: every walk skips it).  No aidx, amap or amask: the gather form, which finds the
    // partner's partial force through them, re-runs this pass in the compact layout when it is needed (ccode_valid_).
    // Rows: max-over-the-wavefront(angular neighbours), occasionally one or two more (a lane that filled its queue early
    // forces a row the others pad): the row arrays have kAngRowPad rows beyond MN_acomp.
    constexpr bool ASYNC = SYNC != 0 && S::fixed;
    SyncFifo qa;
    qa.init();
    int arows = 0;
    auto ang_emit = [&]() __attribute__((always_inline)) {
      const unsigned slot = qa.pop_one((unsigned)b.wsent);
      if (arows < b.MN_arows) {
        F4 e;
        e.x = e.y = e.z = 0.0f;
        e.w = kNullRecord;
        if (slot != (unsigned)b.wsent) {
          const WinRec r = wrec[slot];
          e.x = (float)(r.x - ox) * unit;
          e.y = (float)(r.y - oy) * unit;
          e.z = (float)(r.z - oz) * unit;
          e.w = r.w;
        }
        acomp[(int64_t)arows * N] = e;
        b.aslot[(int64_t)arows * N + k] = (unsigned short)slot;
      }
      ++arows;
    };
    auto push_front = [&](const Cand& c, const int bit) __attribute__((always_inline)) {
      if (SYNC) {
        if (c.inside) {
          q0.push((unsigned)c.slot);
          ++cnt;
        }
        return;
      }
      if (b.use_rmask) {
        mcur |= c.inside ? (1u << bit) : 0u;
        cnt += c.inside ? 1 : 0;
        return;
      }
      if (c.inside) {
        if (NEPMI_CW) {
          const int f = cnt & 3;
          accf |= (unsigned long long)(unsigned)c.slot << (16 * f);
          if (f == 3) {
            if (owned && (cnt >> 2) < b.MN_cw) {
              U2w v;
              v.lo = (unsigned)accf;
              v.hi = (unsigned)(accf >> 32);
              cword[(int64_t)(cnt >> 2) * N] = v;
            }
            accf = 0ull;
          }
        } else if (NEPMI_RW2_ABL == 0 && owned && cnt + cnt1 < b.MN_rad) {
          ccode[(int64_t)cnt * N] = (unsigned short)c.slot;
        }
        if (NEPMI_RW2_ABL < 2)
          ++cnt;
      }
    };
    auto push_back = [&](const Cand& c, const int bit) __attribute__((always_inline)) {
      if (SYNC) {
        if (c.inside) {
          q1.push((unsigned)c.slot);
          ++cnt1;
        }
        return;
      }
      if (b.use_rmask) {
        mcur |= c.inside ? (1u << bit) : 0u;
        cnt1 += c.inside ? 1 : 0;
        return;
      }
      if (c.inside) {
        if (NEPMI_CW) {
          const int f = cnt1 & 3;
          accb |= (unsigned long long)(unsigned)c.slot << (16 * f);
          if (f == 3) {
            if (owned && (cnt1 >> 2) < b.MN_cw) {
              U2w v;
              v.lo = (unsigned)accb;
              v.hi = (unsigned)(accb >> 32);
              cword[(int64_t)(b.MN_cw + (cnt1 >> 2)) * N] = v;
            }
            accb = 0ull;
          }
        } else if (NEPMI_RW2_ABL == 0 && owned && cnt + cnt1 < b.MN_rad) {
          ccode[(int64_t)(b.MN_rad - 1 - cnt1) * N] = (unsigned short)c.slot;
        }
        if (NEPMI_RW2_ABL < 2)
          ++cnt1;
      }
    };
    // the last, partial word of a stream: the free places take the sentinel slot
    auto flush_words = [&]() __attribute__((always_inline)) {
      if (!NEPMI_CW || !owned)
        return;
      const unsigned long long sent = 0x0001000100010001ull * (unsigned long long)(unsigned)b.wsent;
      if ((cnt & 3) != 0 && (cnt >> 2) < b.MN_cw) {
        const unsigned long long a = accf | (sent << (16 * (cnt & 3)));
        U2w v;
        v.lo = (unsigned)a;
        v.hi = (unsigned)(a >> 32);
        cword[(int64_t)(cnt >> 2) * N] = v;
      }
      if ((cnt1 & 3) != 0 && (cnt1 >> 2) < b.MN_cw) {
        const unsigned long long a = accb | (sent << (16 * (cnt1 & 3)));
        U2w v;
        v.lo = (unsigned)a;
        v.hi = (unsigned)(a >> 32);
        cword[(int64_t)(b.MN_cw + (cnt1 >> 2)) * N] = v;
      }
    };
    auto rc_of = [&](int rw, float& rc, float& ri) __attribute__((always_inline)) {
      if (m.uniform_rc) {
        rc = m.rc_r_max;
        ri = m.rcinv_r;
      } else {
        rc = (rc1 + m.rc_r[(unsigned)rw >> kIdxBits]) * 0.5f;
        ri = fast_rcp(rc);
      }
    };
    // packed basis functions of two candidates, already weighted by "inside" (the envelope is evaluated at min(d, rc))
    auto basis2 = [&](const Cand& c0, const Cand& c1, f2* fn) __attribute__((always_inline)) {
      float rc0, rc1v, ri0, ri1;
      rc_of(c0.rw, rc0, ri0);
      rc_of(c1.rw, rc1v, ri1);
      float d0, d1, i0, i1;
      dist_and_inv(c0.d2, d0, i0);
      dist_and_inv(c1.d2, d1, i1);
      const f2 dc = mk2(d0 < rc0 ? d0 : rc0, d1 < rc1v ? d1 : rc1v);
      const f2 rcinv = mk2(ri0, ri1);
      f2 fc;
      cutoff_fc_v(rcinv, dc, fc);
      fc = fc * mk2(c0.inside ? 1.0f : 0.0f, c1.inside ? 1.0f : 0.0f);
      basis_fn_v<S::KRM>(rcinv, dc, fc, fn);
    };

    // The packed sums: ZIP: SS[k] = {sum over type-0 neighbours, sum over type-1 neighbours}; one type: the two halves
    // are added at the end.  List A of a two-type model (mixed types, word order = list order) uses its own two rows.
    f2 SS[S::KRM + 1];
#pragma unroll
    for (int kk = 0; kk <= S::KRM; ++kk)
      SS[kk] = bc2(0.0f);

    auto load_word = [&](int w, int wend) __attribute__((always_inline)) -> U2w {
      U2w v = {0u, 0u};
      if (wend > 0)
        v = words[(int64_t)(w < wend ? w : wend - 1) * N];
      return v;
    };

    // ---- list A: angular membership + radial sums; words 0 .. wa-1 (two candidates at a time: register budget) ----
    {
      U2w w1 = load_word(0, wa), w2 = load_word(1, wa);
      for (int w = 0; w < wa; ++w) {
        const U2w cur = w1;
        w1 = w2;
        w2 = load_word(w + 2, wa);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned pr = hh == 0 ? cur.lo : cur.hi;
          bool nearband = false;
          Cand c[2];
          c[0] = judge((int)(pr & 0xFFFFu), true, nearband);
          c[1] = judge((int)(pr >> 16), true, nearband);
          if (nearband) {
            retake(c[0], true);
            retake(c[1], true);
          }
          if (ASYNC) { // (a half brings at most two: no lane enters it with more than six waiting)
            while (NEPMI_WAVE_ANY(qa.p > 6))
              ang_emit();
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int idx = 4 * w + 2 * hh + u;
            const bool live = idx < na; // (a sentinel is never inside a cutoff; `live` guards the amap rows)
            unsigned short cs = kNoSlot;
            if (ASYNC) {
              if (c[u].ang) {
                qa.push((unsigned)c[u].slot);
                ++ca;
              }
            } else if (c[u].ang) {
              if (ca < b.MN_acomp) {
                F4 e;
                e.x = c[u].fx * unit;
                e.y = c[u].fy * unit;
                e.z = c[u].fz * unit;
                e.w = c[u].rw;
                if (NEPMI_RW2_ABL != 3) {
                  acomp[(int64_t)ca * N] = e;
                  aidx[(int64_t)ca * N] = b.rev_ang[(int64_t)idx * N + k]; // reverse slot of this pair in j's list A
                  if (b.aslot)
                    b.aslot[(int64_t)ca * N + k] = (unsigned short)c[u].slot; // the partner's place in this brick's window
                }
                cs = (unsigned short)ca;
                const unsigned bit = 1u << (idx & 31);
                const int aw = idx >> 5;
                am[0] |= aw == 0 ? bit : 0u;
                am[1] |= aw == 1 ? bit : 0u;
                am[2] |= aw == 2 ? bit : 0u;
                am[3] |= aw == 3 ? bit : 0u;
              }
              ++ca;
            }
            if (!ASYNC && !b.use_amask && live)
              amap[(int64_t)idx * N] = cs;
            if (S::TS == 2 && ((unsigned)c[u].rw >> kIdxBits) == 1u)
              push_back(c[u], 4 * (w & 7) + 2 * hh + u);
            else
              push_front(c[u], 4 * (w & 7) + 2 * hh + u);
          }
          if (S::TS > 0 && !ZIP) { // one type: the two candidates side by side
            f2 fn[S::KRM + 1];
            basis2(c[0], c[1], fn);
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              SS[kk] = SS[kk] + fn[kk];
          } else if (ZIP) {
            // two types, mixed order: each candidate in both halves, the half of its type carries the weight (list A is a
            // quarter of the candidates; separate per-type rows would cost 28 registers over the whole kernel)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              f2 fn[S::KRM + 1];
              basis2(c[u], c[u], fn);
              const bool one = ((unsigned)c[u].rw >> kIdxBits) == 1u;
              const f2 wt = mk2(one ? 0.0f : 1.0f, one ? 1.0f : 0.0f);
#pragma unroll
              for (int kk = 0; kk <= S::KRM; ++kk)
                SS[kk] = vfma(wt, fn[kk], SS[kk]);
            }
          } else if (wide) {
            accumulate_wide(c, 2);
          } else {
            accumulate1(c[0]);
            accumulate1(c[1]);
          }
        }
        if (!SYNC && b.use_rmask && ((w & 7) == 7 || w == wa - 1)) {
          if ((w >> 3) < b.MAW)
            b.rmaskA[(int64_t)(w >> 3) * N + k] = mcur;
          mcur = 0u;
        }
        sync_check();
      }
    }
    if (ASYNC) { // what still waits: again every lane of the wavefront at once
      while (NEPMI_WAVE_ANY(qa.p > 0))
        ang_emit();
    } else if (b.use_amask) {
      U4 mv;
      mv.x = am[0];
      mv.y = am[1];
      mv.z = am[2];
      mv.w = am[3];
      reinterpret_cast<U4*>(b.amask)[k] = mv;
    }

    // ---- list B: radial only ----
    if (ZIP) {
      // word pair p: row wa + 2p holds four neighbours of type 0, row wa + 2p + 1 four of type 1
      const int rows = 2 * wb;
      U2w a1 = load_word(wa, wa + rows), b1 = load_word(wa + 1, wa + rows);
      U2w a2 = load_word(wa + 2, wa + rows), b2 = load_word(wa + 3, wa + rows);
      for (int p = 0; p < wb; ++p) {
        const U2w ca0 = a1, cb0 = b1;
        a1 = a2;
        b1 = b2;
        a2 = load_word(wa + 2 * p + 4, wa + rows);
        b2 = load_word(wa + 2 * p + 5, wa + rows);
        constexpr int HN = NEPMI_RW2_HALF ? 2 : 4;
#pragma unroll
        for (int hh = 0; hh < 4 / HN; ++hh) {
          const unsigned xa = HN == 4 ? ca0.lo : (hh == 0 ? ca0.lo : ca0.hi), xb = HN == 4 ? ca0.hi : 0u;
          const unsigned ya = HN == 4 ? cb0.lo : (hh == 0 ? cb0.lo : cb0.hi), yb = HN == 4 ? cb0.hi : 0u;
          bool nearband = false;
          Cand x[HN], y[HN];
          x[0] = judge((int)(xa & 0xFFFFu), false, nearband);
          x[1] = judge((int)(xa >> 16), false, nearband);
          y[0] = judge((int)(ya & 0xFFFFu), false, nearband);
          y[1] = judge((int)(ya >> 16), false, nearband);
          if (HN == 4) {
            x[HN - 2] = judge((int)(xb & 0xFFFFu), false, nearband);
            x[HN - 1] = judge((int)(xb >> 16), false, nearband);
            y[HN - 2] = judge((int)(yb & 0xFFFFu), false, nearband);
            y[HN - 1] = judge((int)(yb >> 16), false, nearband);
          }
          if (nearband) {
#pragma unroll
            for (int u = 0; u < HN; ++u) {
              retake(x[u], false);
              retake(y[u], false);
            }
          }
#pragma unroll
          for (int u = 0; u < HN; ++u) {
            push_front(x[u], 8 * (p & 3) + HN * hh + u);
            push_back(y[u], 8 * (p & 3) + 4 + HN * hh + u);
            f2 fn[S::KRM + 1];
            basis2(x[u], y[u], fn);
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              SS[kk] = SS[kk] + fn[kk];
          }
        }
        if (!SYNC && b.use_rmask && ((p & 3) == 3 || p == wb - 1)) {
          if ((p >> 2) < b.MBW)
            b.rmaskB[(int64_t)(p >> 2) * N + k] = mcur;
          mcur = 0u;
        }
        sync_check();
      }
    } else {
      U2w w1 = load_word(wa, wa + wb), w2 = load_word(wa + 1, wa + wb);
      for (int w = 0; w < wb; ++w) {
        const U2w cur = w1;
        w1 = w2;
        w2 = load_word(wa + w + 2, wa + wb);
        bool nearband = false;
        Cand c[4];
        c[0] = judge((int)(cur.lo & 0xFFFFu), false, nearband);
        c[1] = judge((int)(cur.lo >> 16), false, nearband);
        c[2] = judge((int)(cur.hi & 0xFFFFu), false, nearband);
        c[3] = judge((int)(cur.hi >> 16), false, nearband);
        if (nearband) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            retake(c[u], false);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          push_front(c[u], 4 * (w & 7) + u);
        if (!SYNC && b.use_rmask && ((w & 7) == 7 || w == wb - 1)) {
          if ((w >> 3) < b.MBW)
            b.rmaskB[(int64_t)(w >> 3) * N + k] = mcur;
          mcur = 0u;
        }
        sync_check();
        if (S::TS > 0) {
#pragma unroll
          for (int u = 0; u < 4; u += 2) {
            f2 fn[S::KRM + 1];
            basis2(c[u], c[u + 1], fn);
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              SS[kk] = SS[kk] + fn[kk];
          }
        } else if (wide) {
          accumulate_wide(c, 4);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            accumulate1(c[u]);
        }
      }
    }

    flush_words();
    if (SYNC) { // what still waits leaves as padded words: again every lane of the wavefront at once
      while (NEPMI_WAVE_ANY(q0.p > 0))
        sync_emit(q0, sync0);
      while (S::TS == 2 && NEPMI_WAVE_ANY(q1.p > 0))
        sync_emit(q1, sync1);
    }
    if (ca > b.MN_acomp || (ASYNC && arows > b.MN_arows) || (SYNC ? (q0.rows > b.MN_cw || q1.rows > b.MN_cw) : cnt + cnt1 > b.MN_rad)) {
      NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 4);
      ca = ca > b.MN_acomp ? b.MN_acomp : ca;
    }
    if (ASYNC) {
      b.nn_angtrue[k] = ca;
      ca = arows < b.MN_arows ? arows : b.MN_arows; // rows of this atom's records, padding included (the same on the whole wavefront)
    }
    b.nn_rad[k] = cnt + cnt1;
    if (SYNC) {
      const int r0 = q0.rows < b.MN_cw ? q0.rows : b.MN_cw, r1 = q1.rows < b.MN_cw ? q1.rows : b.MN_cw;
      b.nn_t0[k] = r0 | (r1 << 8); // words per stream (the same on every lane of the wavefront)
    } else {
      b.nn_t0[k] = cnt;
    }
    b.nn_angstep[k] = ca;
    if (S::TS > 0) {
      float Ssum[TSM][S::KRM + 1];
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk) {
        if (ZIP) {
          Ssum[0][kk] = SS[kk].x;
          Ssum[TSM - 1][kk] = SS[kk].y;
        } else {
          Ssum[0][kk] = SS[kk].x + SS[kk].y;
        }
      }
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

// The radial pass with L = 2 or 4 adjacent lanes per atom (workgroups of 256 L threads): for systems with too few
// bricks to fill the chip, where the kernel's run time is the latency of ONE workgroup walking ~86 candidates per lane.
// A round covers 2 L candidates of the atom's list, two per lane (side by side in the packed accumulation).  The
// compact lists keep exactly the order and density of the one-lane form -- a candidate's slot is the running count
// plus the number of admitted candidates before it in the round, found from the lanes' flag bits (shuffles) -- so the
// downstream kernels and the results are the same; the basis sums are added across the lanes at the end.
#if defined(__HIP_DEVICE_COMPILE__)
#define NEPMI_SHFL_XOR(v, mask) ((mask) == 1 ? nepmi::quad_xor<1>(v) : nepmi::quad_xor<2>(v)) // (mask: 1 or 2 -- L <= 4 adjacent lanes)
#else
#define NEPMI_SHFL_XOR(v, mask) (v) // host loops run one lane per atom: the split bodies are never selected there
#endif

template <class S, int L>
struct RadialWinSplitBody {
  WinStage st;
  ModelD m;
  int first;
  const int* frozen;
  static constexpr int kMinWavesPerEu = 1;
  static constexpr int kLanes = L;

  NEPMI_HD int lds_bytes() const { return st.lay.bytes(); }
  template <class LC>
  NEPMI_HD void stage_lists(int64_t, LC, int, int) const {}
  NEPMI_HD int64_t map_brick(int64_t w) const { return first < 0 ? w : (int64_t)st.b.brick_order[first + w]; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const { st.stage_copy(brick, lds, tid, nth); }
  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const { st.brick_range(brick, a0, a1); }

  template <class LC>
  NEPMI_HD void compute(int64_t brick, int64_t k, LC lds, int sub) const
  {
    const Bufs& b = st.b;
    const int64_t N = b.N;
    if (b.lvl[k] < 1) { // outer ghost: lends its position only
      if (sub == 0) {
        b.nn_rad[k] = 0;
        b.nn_angstep[k] = 0;
      }
      return;
    }
    NEPMI_LDS(const int)* woff = (NEPMI_LDS(const int)*)(lds + st.lay.off_woff());
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const int NR = S::fixed ? S::NR : m.NR;
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(k, ox, oy, oz);
    const float rc1 = m.rc_r[t1], rca1 = m.rc_a[t1];
    const float unit = st.b.wg.unit, unit2 = st.b.wg.unit2, band = st.b.wg.band;
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    f2 Ssum2[TSM][S::KRM + 1];
    float q[S::NRM + 1];
#pragma unroll
    for (int t = 0; t < TSM; ++t)
#pragma unroll
      for (int kk = 0; kk <= S::KRM; ++kk)
        Ssum2[t][kk] = bc2(0.0f);
#pragma unroll
    for (int n = 0; n <= S::NRM; ++n)
      q[n] = 0.0f;

    const int na = b.nn_ang[k], nbn = b.nn_skin[k];
    int cnt = 0, cnt1 = 0, ca = 0; // the same on every lane of the atom
    F4* __restrict__ acomp = b.acomp + k;
    unsigned short* __restrict__ amap = b.amap + k;
    unsigned short* __restrict__ aidx = b.aidx + k;
    unsigned short* __restrict__ ccode = b.ccode + k;

    struct Cand {
      float fx, fy, fz, d2;
      int t2, slot, rw;
      bool inside, ang;
    };
    // the list decisions of one candidate (RadialWinBody::decide without the bookkeeping)
    auto judge = [&](const unsigned code, const bool live, const bool list_a) -> Cand {
      Cand c;
      c.slot = woff[code >> 7] + (int)(code & 127u);
      const WinRec r = wrec[c.slot];
      c.rw = r.w;
      c.fx = (float)(r.x - ox);
      c.fy = (float)(r.y - oy);
      c.fz = (float)(r.z - oz);
      c.d2 = dot3f(c.fx, c.fx, c.fy, c.fy, c.fz, c.fz) * unit2;
      c.t2 = (int)((unsigned)r.w >> kIdxBits);
      const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[c.t2]) * 0.5f;
      const float rca = m.uniform_rc ? m.rc_a_max : (rca1 + m.rc_a[c.t2]) * 0.5f;
      const float er = c.d2 - rc * rc, ea = c.d2 - rca * rca;
      bool inside = er < 0.0f;
      bool ang = list_a && ea < 0.0f;
      const float near = list_a ? fminf(fabsf(er), fabsf(ea)) : fabsf(er);
      if (live && near < band) { // within the band of a cutoff: the reference's arithmetic decides (rare)
        float ex, ey, ez;
        const float d2e = pair_geometry(st.box, p1, b.posq[(unsigned)r.w & (unsigned)kIdxMask], ex, ey, ez);
        inside = d2e < rc * rc;
        ang = list_a && d2e < rca * rca;
      }
      c.inside = inside && live;
      c.ang = ang && live;
      return c;
    };
    // bit s of the result: `flag` of lane s of this atom
    auto gather_bits = [&](bool flag) -> unsigned {
      unsigned w = (flag ? 1u : 0u) << sub;
#pragma unroll
      for (int msk = 1; msk < L; msk <<= 1)
        w |= (unsigned)NEPMI_SHFL_XOR((int)w, msk);
      return w;
    };
    const unsigned below = (1u << sub) - 1u;

    auto walk = [&](const unsigned short* __restrict__ codes, const int nn, const bool list_a) {
      auto load_code = [&](int idx) -> unsigned { return nn > 0 ? codes[(int64_t)(idx < nn ? idx : nn - 1) * N] : 0u; };
      unsigned nxt0 = load_code(sub), nxt1 = load_code(L + sub);
      for (int s0 = 0; s0 < nn; s0 += 2 * L) {
        const int i0 = s0 + sub, i1 = s0 + L + sub;
        const bool live0 = i0 < nn, live1 = i1 < nn;
        const unsigned code0 = nxt0, code1 = nxt1;
        nxt0 = load_code(i0 + 2 * L); // the next round's codes travel while this round is decided
        nxt1 = load_code(i1 + 2 * L);
        const Cand c0 = judge(code0, live0, list_a), c1 = judge(code1, live1, list_a);
        // slots in candidate order: first halves of all lanes, then second halves
        const bool b0 = S::TS == 2 && c0.t2 == 1, b1 = S::TS == 2 && c1.t2 == 1;
        const unsigned f0 = gather_bits(c0.inside && !b0), f1 = gather_bits(c1.inside && !b1);
        const unsigned k0 = S::TS == 2 ? gather_bits(c0.inside && b0) : 0u, k1 = S::TS == 2 ? gather_bits(c1.inside && b1) : 0u;
        const int nf = __builtin_popcount(f0) + __builtin_popcount(f1);
        const int nk = __builtin_popcount(k0) + __builtin_popcount(k1);
        const bool room = cnt + cnt1 + nf + nk <= b.MN_rad;
        if (c0.inside && room) {
          const int pos = b0 ? b.MN_rad - 1 - (cnt1 + __builtin_popcount(k0 & below))
                             : cnt + __builtin_popcount(f0 & below);
          ccode[(int64_t)pos * N] = (unsigned short)c0.slot;
        }
        if (c1.inside && room) {
          const int pos = b1 ? b.MN_rad - 1 - (cnt1 + __builtin_popcount(k0) + __builtin_popcount(k1 & below))
                             : cnt + __builtin_popcount(f0) + __builtin_popcount(f1 & below);
          ccode[(int64_t)pos * N] = (unsigned short)c1.slot;
        }
        cnt += nf;
        cnt1 += nk;
        if (list_a) {
          const unsigned a0 = gather_bits(c0.ang), a1 = gather_bits(c1.ang);
          const int s_a0 = ca + __builtin_popcount(a0 & below);
          const int s_a1 = ca + __builtin_popcount(a0) + __builtin_popcount(a1 & below);
          auto commit = [&](const Cand& c, int idx, bool live, int sa) {
            if (!live)
              return;
            unsigned short cs = kNoSlot;
            if (c.ang && sa < b.MN_acomp) {
              F4 e;
              e.x = c.fx * unit;
              e.y = c.fy * unit;
              e.z = c.fz * unit;
              e.w = c.rw;
              acomp[(int64_t)sa * N] = e;
              aidx[(int64_t)sa * N] = b.rev_ang[(int64_t)idx * N + k];
              cs = (unsigned short)sa;
            }
            amap[(int64_t)idx * N] = cs;
          };
          commit(c0, i0, live0, s_a0);
          commit(c1, i1, live1, s_a1);
          ca += __builtin_popcount(a0) + __builtin_popcount(a1);
        }
        // accumulation (RadialWinBody::accumulate2 / accumulate1)
        if (S::TS > 0) {
          float rc0, rc1v, ri0, ri1;
          if (m.uniform_rc) {
            rc0 = rc1v = m.rc_r_max;
            ri0 = ri1 = m.rcinv_r;
          } else {
            rc0 = (rc1 + m.rc_r[c0.t2]) * 0.5f;
            rc1v = (rc1 + m.rc_r[c1.t2]) * 0.5f;
            ri0 = fast_rcp(rc0);
            ri1 = fast_rcp(rc1v);
          }
          float d0, d1, iv0, iv1;
          dist_and_inv(c0.d2, d0, iv0);
          dist_and_inv(c1.d2, d1, iv1);
          const f2 dc = mk2(d0 < rc0 ? d0 : rc0, d1 < rc1v ? d1 : rc1v);
          const f2 rcinv = mk2(ri0, ri1);
          f2 fc;
          cutoff_fc_v(rcinv, dc, fc);
          f2 fn[S::KRM + 1];
          basis_fn_v<S::KRM>(rcinv, dc, fc, fn);
#pragma unroll
          for (int t = 0; t < TSM; ++t) {
            const f2 w = mk2((c0.inside && (TSM == 1 || c0.t2 == t)) ? 1.0f : 0.0f,
                             (c1.inside && (TSM == 1 || c1.t2 == t)) ? 1.0f : 0.0f);
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk)
              Ssum2[t][kk] = vfma(w, fn[kk], Ssum2[t][kk]);
          }
        } else {
          auto acc1 = [&](const Cand& c) {
            if (!c.inside)
              return;
            const float rc = m.uniform_rc ? m.rc_r_max : (rc1 + m.rc_r[c.t2]) * 0.5f;
            float d, dinv;
            dist_and_inv(c.d2, d, dinv);
            const float rcinv = fast_rcp(rc);
            const float dc = d < rc ? d : rc;
            float fc;
            cutoff_fc(rcinv, dc, fc);
            float fn[S::KRM + 1];
            if (S::fixed)
              basis_fn<S::KRM>(rcinv, dc, fc, fn);
            else
              basis_fn_rt(KR, rcinv, dc, fc, fn);
            const float* cc = m.c_rad + (size_t)(t1 * m.T + c.t2) * (NR + 1) * (KR + 1);
            for (int n = 0; n <= NR; ++n) {
              float gsum = 0.0f;
              for (int kk = 0; kk <= KR; ++kk)
                gsum += fn[kk] * cc[n * (KR + 1) + kk];
              q[n] += gsum;
            }
          };
          acc1(c0);
          acc1(c1);
        }
      }
    };
    walk(b.code_ang + k, na, true);
    walk(b.code_skin + k, nbn, false);

    if (sub == 0) {
      if (ca > b.MN_acomp || cnt + cnt1 > b.MN_rad)
        NEPMI_ATOMIC_OR(&b.flags[kFlagOverflow], 4);
      b.nn_rad[k] = cnt + cnt1;
      b.nn_t0[k] = cnt;
      b.nn_angstep[k] = ca > b.MN_acomp ? b.MN_acomp : ca;
    }
    if (S::TS > 0) {
      float Ssum[TSM][S::KRM + 1];
#pragma unroll
      for (int t = 0; t < TSM; ++t)
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk) {
          float v = Ssum2[t][kk].x + Ssum2[t][kk].y;
#pragma unroll
          for (int msk = 1; msk < L; msk <<= 1)
            v += NEPMI_SHFL_XOR(v, msk);
          Ssum[t][kk] = v;
        }
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
    } else {
#pragma unroll
      for (int n = 0; n <= S::NRM; ++n)
#pragma unroll
        for (int msk = 1; msk < L; msk <<= 1)
          q[n] += NEPMI_SHFL_XOR(q[n], msk);
    }
    if (sub == 0) {
      const int64_t gk = b.tpos[k];
      for (int n = 0; n <= NR; ++n)
        b.q[(int64_t)n * N + gk] = q[n] * m.qscale[n];
    }
  }
};

// Angular part of the force assembly: F_i += f12 - f21, W_i += r12 (x) f21 over this step's angular pairs (compact
// records), f21 found through the reverse slot the radial pass stored next to the record (arev) and j's slot map.
// The chain arev -> amap[.][j] -> f12[.][j] is two dependent gathers per pair: four pairs are kept in flight.
template <int LANES>
NEPMI_HD void win_force_angular(const Bufs& b, int64_t k, int part, float* F, float* Wa)
{
  const int64_t N = b.N;
  const int nang = b.nn_angstep[k];
  const F4* __restrict__ acomp = b.acomp + k;
  const F4* __restrict__ f12o = b.f12 + k;
  const unsigned short* __restrict__ arev = b.aidx + k;
  constexpr int C = 4;
  for (int a0 = part; a0 < nang; a0 += C * LANES) {
    F4 e[C], fa[C], fb[C];
    int rs[C];
    unsigned short ap[C];
#pragma unroll
    for (int u = 0; u < C; ++u) {
      const int a = a0 + u * LANES;
      const int aa = a < nang ? a : part; // a valid slot (a0 = part < nang); its contribution is dropped below
      e[u] = acomp[(int64_t)aa * N];
      fa[u] = f12o[(int64_t)aa * N];
      rs[u] = arev[(int64_t)aa * N];
    }
    if (b.use_amask) {
      // the partner's compact slot = rank of the reverse entry among the set bits of the partner's membership mask
      U4 mj[C];
#pragma unroll
      for (int u = 0; u < C; ++u)
        mj[u] = reinterpret_cast<const U4*>(b.amask)[(int)((unsigned)e[u].w & (unsigned)kIdxMask)];
#pragma unroll
      for (int u = 0; u < C; ++u) {
        const int w = rs[u] >> 5;
        const unsigned bit = 1u << (rs[u] & 31);
        const unsigned word = w == 0 ? mj[u].x : (w == 1 ? mj[u].y : (w == 2 ? mj[u].z : mj[u].w));
        const int base = (w > 0 ? __builtin_popcount(mj[u].x) : 0) + (w > 1 ? __builtin_popcount(mj[u].y) : 0) +
                         (w > 2 ? __builtin_popcount(mj[u].z) : 0);
        const bool member = rs[u] != (int)kNoSlot && (word & bit) != 0u;
        ap[u] = member ? (unsigned short)(base + __builtin_popcount(word & (bit - 1u))) : kNoSlot;
      }
    } else {
#pragma unroll
      for (int u = 0; u < C; ++u) {
        const int j = (int)((unsigned)e[u].w & (unsigned)kIdxMask);
        ap[u] = rs[u] != (int)kNoSlot ? b.amap[(int64_t)rs[u] * N + j] : kNoSlot;
      }
    }
#pragma unroll
    for (int u = 0; u < C; ++u) {
      // j has no compact slot for this pair only if its per-step angular list overflowed (MN_angular): that is
      // reported through the overflow flag; never index with kNoSlot
      const int j = (int)((unsigned)e[u].w & (unsigned)kIdxMask);
      fb[u].x = fb[u].y = fb[u].z = 0.0f;
      fb[u].w = 0;
      if (ap[u] != kNoSlot)
        fb[u] = b.f12[(int64_t)ap[u] * N + j];
    }
#pragma unroll
    for (int u = 0; u < C; ++u) {
      const float w = a0 + u * LANES < nang ? 1.0f : 0.0f;
      const float gx = w * fb[u].x, gy = w * fb[u].y, gz = w * fb[u].z;
      F[0] += w * fa[u].x - gx;
      F[1] += w * fa[u].y - gy;
      F[2] += w * fa[u].z - gz;
      Wa[0] += e[u].x * gx;
      Wa[1] += e[u].y * gy;
      Wa[2] += e[u].z * gz;
      Wa[3] += e[u].x * gy;
      Wa[4] += e[u].x * gz;
      Wa[5] += e[u].y * gz;
      Wa[6] += e[u].y * gx;
      Wa[7] += e[u].z * gx;
      Wa[8] += e[u].z * gy;
    }
  }
}

// Radial pair forces of one type-pure segment of the compact list, two pairs side by side in f2 values (packed
// FP32): envelope, Chebyshev derivatives, the two table contractions and the force / virial sums.  `lanes` lanes
// share the atom, lane `part` takes every lanes-th chunk of two entries; entry i of the segment sits at row
// row0 + i * row_step of ccode.  rows(slot, Aj): the neighbours' table rows for the own type (gathered from L2 or
// read from LDS).  Sums are in grid units.
template <class S, class LC, class Rows>
NEPMI_HD void win_force_segment(
  const ModelD& m, const unsigned short* __restrict__ ccode, int64_t N, LC wrec, const Rows& rows, const float* Aown,
  int count, int row0, int row_step, int part, int lanes, int ox, int oy, int oz, float rc1, float unit2, f2* Fr2, f2* W2)
{
  auto load_codes = [&](int s0, unsigned& c0, unsigned& c1) {
    const int i0 = s0 < count ? s0 : count - 1, i1 = s0 + 1 < count ? s0 + 1 : count - 1;
    c0 = ccode[(int64_t)(row0 + i0 * row_step) * N];
    c1 = ccode[(int64_t)(row0 + i1 * row_step) * N];
  };
  const int stride = 2 * lanes;
  unsigned cur0 = 0, cur1 = 0, nxt0 = 0, nxt1 = 0;
  if (2 * part < count)
    load_codes(2 * part, cur0, cur1);
  for (int s0 = 2 * part; s0 < count; s0 += stride) {
    if (s0 + stride < count)
      load_codes(s0 + stride, nxt0, nxt1); // in flight while this chunk is processed
    const WinRec r0 = wrec[cur0], r1 = wrec[cur1];
    f2 Aj[S::KRM + 1];
    rows(cur0, cur1, r0, r1, Aj);
    const f2 fx = mk2((float)(r0.x - ox), (float)(r1.x - ox));
    const f2 fy = mk2((float)(r0.y - oy), (float)(r1.y - oy));
    const f2 fz = mk2((float)(r0.z - oz), (float)(r1.z - oz));
    const f2 d2 = vfma(fz, fz, vfma(fy, fy, fx * fx)) * unit2;
    float d0, d1, i0, i1;
    dist_and_inv(d2.x, d0, i0);
    dist_and_inv(d2.y, d1, i1);
    float rc0, rc1v, ri0, ri1;
    if (m.uniform_rc) {
      rc0 = rc1v = m.rc_r_max;
      ri0 = ri1 = m.rcinv_r;
    } else {
      rc0 = (rc1 + m.rc_r[(unsigned)r0.w >> kIdxBits]) * 0.5f;
      rc1v = (rc1 + m.rc_r[(unsigned)r1.w >> kIdxBits]) * 0.5f;
      ri0 = fast_rcp(rc0);
      ri1 = fast_rcp(rc1v);
    }
    // a pair the exact test admitted can sit a rounding above rc here: the envelope is clamped there
    const f2 dc = mk2(d0 < rc0 ? d0 : rc0, d1 < rc1v ? d1 : rc1v);
    const f2 rcinv = mk2(ri0, ri1);
    f2 fc, fcp;
    cutoff_fc_fcp_v(rcinv, dc, fc, fcp);
    f2 fnp[S::KRM + 1];
    basis_fnp_v<S::KRM>(rcinv, dc, fc, fcp, fnp);
    f2 s12 = bc2(0.0f), s21 = bc2(0.0f);
#pragma unroll
    for (int kk = 0; kk <= S::KRM; ++kk) {
      s12 = vfma(fnp[kk], bc2(Aown[kk]), s12);
      s21 = vfma(fnp[kk], Aj[kk], s21);
    }
    const f2 wgt = mk2(i0, s0 + 1 < count ? i1 : 0.0f); // the entry past the end re-read the last slot
    const f2 fs = (s12 + s21) * wgt; // f12 - f21 = fs * r12
    const f2 bb = s21 * wgt;         // f21 = -bb * r12
    Fr2[0] = vfma(fs, fx, Fr2[0]);
    Fr2[1] = vfma(fs, fy, Fr2[1]);
    Fr2[2] = vfma(fs, fz, Fr2[2]);
    const f2 bx = bb * fx, by = bb * fy, bz = bb * fz;
    W2[0] = vfma(-fx, bx, W2[0]);
    W2[1] = vfma(-fy, by, W2[1]);
    W2[2] = vfma(-fz, bz, W2[2]);
    W2[3] = vfma(-fx, by, W2[3]);
    W2[4] = vfma(-fx, bz, W2[4]);
    W2[5] = vfma(-fy, bz, W2[5]);
    cur0 = nxt0;
    cur1 = nxt1;
  }
}

// The same over a stream of compact-list WORDS (Bufs::cword: four LDS slots per 8 bytes, the last word padded with the
// sentinel slot, whose record lies beyond the cutoff: the clamped envelope and its derivative vanish there, so a padded
// place adds exactly zero and the loop needs no "live" weights).  One lane per atom; `nwords` words at rows row0, row0 + 1, ...
//
// Software-pipelined: a wavefront's pair loop is a chain  list word -> window record (LDS) -> neighbour's table row (a gather
// that misses L1) -> arithmetic, and with four wavefronts per SIMD nothing hides a global round trip per evaluation (r3c/r3d:
// the loop took the same time with the rows gathered from L2, read from LDS, or the list packed -- it was waiting for whichever
// global load came last).  Here the list words are requested NEPMI_FW_AHEAD words (two evaluations each) ahead and the records
// + rows of evaluation e + 1 are requested before evaluation e is computed.
#ifndef NEPMI_FW_AHEAD
#define NEPMI_FW_AHEAD 3
#endif
template <class S, class LC, class Rows>
NEPMI_HD void win_force_words(
  const ModelD& m, const U2w* __restrict__ cword, int64_t N, LC wrec, const Rows& rows, const float* Aown, int nwords, int row0,
  int ox, int oy, int oz, float rc1, float unit2, f2* Fr2, f2* W2)
{
  if (nwords <= 0)
    return;
  auto load_word = [&](int w) -> U2w { return cword[(int64_t)(row0 + (w < nwords ? w : nwords - 1)) * N]; };
  struct Stage { // operands of one packed evaluation of two pairs: the two slots and the neighbours' table rows (the window
    unsigned pr; // records are read again from LDS when the evaluation starts: cheaper than eight registers per stage)
    f2 Aj[S::KRM + 1];
  };
  auto fetch = [&](unsigned pr, Stage& st) __attribute__((always_inline)) {
    const unsigned c0 = pr & 0xFFFFu, c1 = pr >> 16;
    st.pr = pr;
    const WinRec r0 = wrec[c0], r1 = wrec[c1];
    rows(c0, c1, r0, r1, st.Aj);
  };
  auto evaluate = [&](const Stage& st) __attribute__((always_inline)) {
    const WinRec r0 = wrec[st.pr & 0xFFFFu], r1 = wrec[st.pr >> 16];
    const f2 fx = mk2((float)(r0.x - ox), (float)(r1.x - ox));
    const f2 fy = mk2((float)(r0.y - oy), (float)(r1.y - oy));
    const f2 fz = mk2((float)(r0.z - oz), (float)(r1.z - oz));
    const f2 d2 = vfma(fz, fz, vfma(fy, fy, fx * fx)) * unit2;
    float d0, d1, i0, i1;
    dist_and_inv(d2.x, d0, i0);
    dist_and_inv(d2.y, d1, i1);
    float rc0, rc1v, ri0, ri1;
    if (m.uniform_rc) {
      rc0 = rc1v = m.rc_r_max;
      ri0 = ri1 = m.rcinv_r;
    } else {
      rc0 = (rc1 + m.rc_r[(unsigned)r0.w >> kIdxBits]) * 0.5f;
      rc1v = (rc1 + m.rc_r[(unsigned)r1.w >> kIdxBits]) * 0.5f;
      ri0 = fast_rcp(rc0);
      ri1 = fast_rcp(rc1v);
    }
    const f2 dc = mk2(d0 < rc0 ? d0 : rc0, d1 < rc1v ? d1 : rc1v);
    const f2 rcinv = mk2(ri0, ri1);
    f2 fc, fcp;
    cutoff_fc_fcp_v(rcinv, dc, fc, fcp);
    f2 fnp[S::KRM + 1];
    basis_fnp_v<S::KRM>(rcinv, dc, fc, fcp, fnp);
    f2 s12 = bc2(0.0f), s21 = bc2(0.0f);
#pragma unroll
    for (int kk = 0; kk <= S::KRM; ++kk) {
      s12 = vfma(fnp[kk], bc2(Aown[kk]), s12);
      s21 = vfma(fnp[kk], st.Aj[kk], s21);
    }
    const f2 wgt = mk2(i0, i1);
    const f2 fs = (s12 + s21) * wgt; // f12 - f21 = fs * r12
    const f2 bb = s21 * wgt;         // f21 = -bb * r12
    Fr2[0] = vfma(fs, fx, Fr2[0]);
    Fr2[1] = vfma(fs, fy, Fr2[1]);
    Fr2[2] = vfma(fs, fz, Fr2[2]);
    const f2 bx = bb * fx, by = bb * fy, bz = bb * fz;
    W2[0] = vfma(-fx, bx, W2[0]);
    W2[1] = vfma(-fy, by, W2[1]);
    W2[2] = vfma(-fz, bz, W2[2]);
    W2[3] = vfma(-fx, by, W2[3]);
    W2[4] = vfma(-fx, bz, W2[4]);
    W2[5] = vfma(-fy, bz, W2[5]);
  };
  constexpr int AH = NEPMI_FW_AHEAD;
  U2w q[AH]; // words w + 1 .. w + AH (clamped to the last word: a repeated word is fetched, never evaluated)
  U2w cur = load_word(0);
#pragma unroll
  for (int a = 0; a < AH; ++a)
    q[a] = load_word(1 + a);
  Stage A, Bs;
  fetch(cur.lo, A);
  for (int w = 0; w < nwords; ++w) {
    fetch(cur.hi, Bs); // second evaluation of this word: in flight while the first is computed
    evaluate(A);
    const U2w nxt = q[0];
#pragma unroll
    for (int a = 0; a + 1 < AH; ++a)
      q[a] = q[a + 1];
    q[AH - 1] = load_word(w + 1 + AH);
    fetch(nxt.lo, A); // first evaluation of the next word (of the same word again past the end: unused)
    evaluate(Bs);
    cur = nxt;
  }
}

// L = 1: one lane per atom.  L = 2, 4 (small systems, see RadialWinSplitBody): L adjacent lanes share the atom, lane
// `sub` takes every L-th chunk of two entries of the compact radial list and every L-th group of angular records; the
// sums are linear in the pairs and are added across the lanes at the end, lane 0 writes.
// ROWS (static layout, shapes with register-resident sums): the radial-table rows of every window atom are staged in LDS
// behind the records (T KRP floats per atom) and the pair loop reads the neighbour's row with ds_read instead of gathering it
// from L2.  The gathers are what binds the plain form: a per-lane random 16-byte global load costs the CU ~46-115 cycles per
// wavefront instruction against ~13-16 for a random ds_read_b128 (profiles/r3b_gather_rate.txt), and the texture-address unit
// is busy for 75 % of the kernel (profiles/r3b_pmc_ta1.csv).  records + rows fill the LDS of a CU (PbTe: 80 B x ~1,700 window
// atoms), so the workgroup is 1024 threads, L = 4 lanes per atom, one workgroup per CU (the same 16 wavefronts).
// FPJ (static layout, many-type shapes in the one-wide form): the neighbour's half of a pair force is contracted on the fly,
//   s21 = sum_n Fp_j[n] sum_k c[t_j][t_i][n][k] f'_k(r),
// from the neighbour's radial Fp row (Bufs::fpr: (n_r+1 -> multiple of 4) floats per atom, 32 MB for a million atoms: the
// gathers stay in L2 / Infinity Cache) and the coefficient table staged in LDS behind the window -- instead of gathering row
// t_i of the neighbour's radial table (T k_r' floats per atom: 768 B for the 16-type UNEP-v1, 0.8 GB per million atoms, every
// row read only ~4 times per step: each gather a line from HBM; r3a: 2.54 ms of a 7.2 ms step).
template <class S, int L = 1, bool CW = false, bool ROWS = false, bool FPJ = false> // CW: compact list as words (Bufs::cword, L = 1)
struct ForceWinBody {
  WinStage st;
  ModelD m;
  const int* frozen;
  int wonly = 0; // 1: only the nine virial planes are written (the per-atom virials of the reference's attribution after a
                 // step whose forces came from the scatter form, nep_scatter.h)
  NEPMI_HD int rows_offset() const { return (st.lay.bytes() + 15) / 16 * 16; }
#ifndef NEPMI_FW_ROWPAD
#define NEPMI_FW_ROWPAD 4 // floats of padding per LDS row: a 64-byte stride puts every row on one of four bank groups
#endif
  NEPMI_HD int row_stride() const { return m.T * st.b.KRP + ((m.T * st.b.KRP) % 16 == 0 ? NEPMI_FW_ROWPAD : 0); }
  NEPMI_HD int rows_bytes() const { return ROWS ? 4 * row_stride() * (st.lay.wmax + 1) : 0; }
  // second staging phase (after a barrier: the records are in place): the table row of the atom at every window slot
  template <class LC>
  NEPMI_HD void stage_rows(int64_t brick, LC lds, int tid, int nth) const
  {
    if (!ROWS)
      return;
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    NEPMI_LDS(F4)* rows = (NEPMI_LDS(F4)*)(lds + rows_offset());
    const int q4 = m.T * st.b.KRP / 4; // 16-byte groups per row (KRP is a multiple of 4)
    const int last = st.b.wtab[(brick * 512 + 511) * 2 + 1]; // atoms in this window = offset + count of the last window cell
    int total = (last & 0xFFFF) + (last >> 16);
    total = total < st.lay.wmax ? total : st.lay.wmax;
    const F4* src = reinterpret_cast<const F4*>(st.b.atab);
    const int s4 = row_stride() / 4;
    for (int i = tid; i < total * q4; i += nth) {
      const int slot = i / q4, g = i - slot * q4;
      const int j = (int)((unsigned)wrec[slot].w & (unsigned)kIdxMask);
      rows[slot * s4 + g] = src[(size_t)j * q4 + g];
    }
  }
  // 4: <= 128 VGPRs, four 256-thread workgroups per CU.  Shapes with register-resident table rows of 9 or more
  // coefficients (carbon: 11) spill 60-90 bytes per lane there; three wavefronts (<= 168 VGPRs) keep them in registers
  // (carbon 1 M atoms: 1.09 -> 0.97 ms)
  // (FPJ: window + coefficient table leave room for two workgroups per CU = two wavefronts per SIMD anyway)
  static constexpr int kMinWavesPerEu = L != 1 ? 1 : (FPJ ? 2 : ((S::TS > 0 && S::KR >= 8) ? 3 : NEPMI_FW_WAVES));
  static constexpr bool kBigWindows = false;
  static constexpr bool kMidWindows = NEPMI_MIDWIN != 0 && L == 1 && S::TS > 0; // (the gather form on windows of dense long-cutoff models: 512 threads)
  static constexpr int kLanes = L;

  NEPMI_HD int ctab_floats() const { return FPJ ? m.T * m.T * ctab_block(m.NR, m.KR, NEPMI_CT_VEC_FORCE != 0) : 0; }
  NEPMI_HD int lds_bytes() const { return ROWS ? rows_offset() + rows_bytes() : (FPJ ? rows_offset() + 4 * ctab_floats() : st.lay.bytes()); }
  template <class LC>
  NEPMI_HD void stage_lists(int64_t, LC, int, int) const {}
  NEPMI_HD int64_t map_brick(int64_t w) const { return w; }
  NEPMI_HD bool skip() const { return frozen && *frozen != 0; }
  template <class LC>
  NEPMI_HD void stage_cells(int64_t brick, LC lds, int tid, int nth) const { st.stage_cells(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage_copy(int64_t brick, LC lds, int tid, int nth) const { st.stage_copy(brick, lds, tid, nth); }
  template <class LC>
  NEPMI_HD void stage(int64_t brick, LC lds, int tid, int nth) const // static layout
  {
    st.stage_direct(brick, lds, tid, nth);
    if (FPJ)
      ctab_stage_padded(m, lds + rows_offset(), tid, nth, NEPMI_CT_VEC_FORCE != 0);
  }
  NEPMI_HD void brick_range(int64_t brick, int64_t& a0, int64_t& a1) const { st.brick_range(brick, a0, a1); }

  template <class LC>
  NEPMI_HD void compute(int64_t brick, int64_t k, LC lds, int sub = 0) const
  {
    const Bufs& b = st.b;
    const int64_t N = b.N;
    if (b.lvl[k] < b.lvl_force) // forces only for owned atoms (reverse mode: the neighbour halves on the ghosts too)
      return;
    NEPMI_LDS(const WinRec)* wrec = (NEPMI_LDS(const WinRec)*)(lds + st.lay.off_rec());
    const int KR = S::fixed ? S::KR : m.KR;
    const PosQ p1 = b.posq[k];
    const int t1 = p1.type;
    int ox, oy, oz;
    st.place_own(k, ox, oy, oz);
    const float rc1 = m.rc_r[t1];
    const float unit = st.b.wg.unit, unit2 = st.b.wg.unit2;
    const int KRP = b.KRP;
    const int arow = m.T * KRP;
    const float* __restrict__ atab = b.atab;

    // ---- angular part: f12 - f21 of this step's angular pairs (compact records) ----
    // (CW: after the radial loop, whose software pipeline needs the registers of these twelve sums)
    float F[3] = {0, 0, 0};
    float Wa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // xx yy zz xy xz yz yx zx zy
#ifndef NEPMI_FW_ANG_LAST
#define NEPMI_FW_ANG_LAST 1 // 1: the angular part runs after the radial loop (its twelve sums are not live across the loop)
#endif
    if (!CW && !NEPMI_FW_ANG_LAST)
      win_force_angular<L>(b, k, sub, F, Wa);

    // ---- radial part over the compact list: every entry is a pair inside the cutoff ----
    constexpr int TSM = S::TS > 0 ? S::TS : 1;
    // accumulated in grid units (x, y, z of a pair are integers times `unit`): Fr = unit * sum, W = unit^2 * sum
    float Fr[3] = {0, 0, 0};
    float W[6] = {0, 0, 0, 0, 0, 0}; // symmetric: xx yy zz xy xz yz
    const int nrad = b.nn_rad[k] < b.MN_rad ? b.nn_rad[k] : b.MN_rad;
    const unsigned short* __restrict__ ccode = b.ccode + k;
    if (S::TS > 0 && NEPMI_FW_PACK) {
      // type-pure segments of the compact list (front: neighbours of type 0, back: of type 1), packed arithmetic
      f2 Fr2[3] = {bc2(0.0f), bc2(0.0f), bc2(0.0f)};
      f2 W2[6] = {bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f), bc2(0.0f)};
      NEPMI_LDS(const float)* lrows = (NEPMI_LDS(const float)*)(lds + rows_offset()) + t1 * KRP;
      const int lstride = row_stride();
      auto rows = [&](unsigned c0, unsigned c1, const WinRec& r0, const WinRec& r1, f2* Aj) {
        if (ROWS) { // the neighbours' rows from the LDS copy, by window slot
          NEPMI_LDS(const float)* row0 = lrows + c0 * lstride;
          NEPMI_LDS(const float)* row1 = lrows + c1 * lstride;
#pragma unroll
          for (int kk = 0; kk <= S::KRM; ++kk)
            Aj[kk] = mk2(row0[kk], row1[kk]);
        } else {
          const float* row0 = atab + (size_t)((unsigned)r0.w & (unsigned)kIdxMask) * arow + t1 * KRP;
          const float* row1 = atab + (size_t)((unsigned)r1.w & (unsigned)kIdxMask) * arow + t1 * KRP;
#pragma unroll
          for (int kk = 0; kk <= S::KRM; ++kk)
            Aj[kk] = mk2(row0[kk], row1[kk]);
        }
      };
      const int n0 = b.nn_t0[k] < nrad ? b.nn_t0[k] : nrad;
#pragma unroll
      for (int t = 0; t < TSM; ++t) {
        float Aown[S::KRM + 1];
#pragma unroll
        for (int kk = 0; kk <= S::KRM; ++kk)
          Aown[kk] = atab[(size_t)k * arow + t * KRP + kk];
        if (CW) {
          const int ne = t == 0 ? n0 : nrad - n0;
          int nw = (ne + 3) >> 2;
          nw = nw < b.MN_cw ? nw : b.MN_cw;
          win_force_words<S>(m, reinterpret_cast<const U2w*>(b.cword) + k, N, wrec, rows, Aown, nw, t == 0 ? 0 : b.MN_cw, ox, oy,
                             oz, rc1, unit2, Fr2, W2);
        } else {
          win_force_segment<S>(m, ccode, N, wrec, rows, Aown, t == 0 ? n0 : nrad - n0, t == 0 ? 0 : b.MN_rad - 1,
                               t == 0 ? 1 : -1, sub, L, ox, oy, oz, rc1, unit2, Fr2, W2);
        }
      }
#pragma unroll
      for (int d = 0; d < 3; ++d)
        Fr[d] = Fr2[d].x + Fr2[d].y;
#pragma unroll
      for (int d = 0; d < 6; ++d)
        W[d] = W2[d].x + W2[d].y;
    } else {
      // one-wide form: the neighbour-type row of the own table is gathered per pair (many types / run-time shape)
      constexpr int G = 2;
      const int NRr = S::fixed ? S::NR : m.NR;
      NEPMI_LDS(const float)* ctl = (NEPMI_LDS(const float)*)(lds + rows_offset());
      const int cblk = ctab_block(NRr, KR, NEPMI_CT_VEC_FORCE != 0);
      float Fpi[S::NRM + 1]; // FPJ: the own radial Fp row (the own half of a pair is contracted on the fly as well)
#pragma unroll
      for (int n = 0; n <= S::NRM; ++n)
        Fpi[n] = (FPJ && (S::fixed || n <= NRr)) ? b.fpr[(size_t)k * b.FPR + n] : 0.0f;
      auto chunk = [&](const unsigned* cur, const int s0) __attribute__((always_inline)) {
        WinRec rr[G];
        float Aj[G][FPJ ? S::NRM + 1 : S::KRM + 1]; // FPJ: the neighbour's radial Fp row, else row t1 of its radial table
#pragma unroll
        for (int u = 0; u < G; ++u) {
          rr[u] = wrec[cur[u]];
          const int j = (int)((unsigned)rr[u].w & (unsigned)kIdxMask);
          if (FPJ) {
            const float* row = b.fpr + (size_t)j * b.FPR;
#pragma unroll
            for (int n = 0; n <= S::NRM; ++n) {
              if (!S::fixed && n > NRr)
                break;
              Aj[u][n] = row[n];
            }
          } else {
            const float* row = atab + (size_t)j * arow + t1 * KRP;
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk) {
              if (!S::fixed && kk > KR)
                break;
              Aj[u][kk] = row[kk];
            }
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
          const float dc = d < rc ? d : rc;
          float fc, fcp;
          cutoff_fc_fcp(rcinv, dc, fc, fcp);
          float fn[S::KRM + 1], fnp[S::KRM + 1];
          if (S::fixed)
            basis_fn_fnp<S::KRM>(rcinv, dc, fc, fcp, fn, fnp);
          else
            basis_fn_fnp_rt(KR, rcinv, dc, fc, fcp, fn, fnp);
          float s12 = 0.0f, s21 = 0.0f;
          if (FPJ) {
            // s12 = sum_n Fp_i[n] sum_k c[t1][t2][n][k] f'_k,  s21 = sum_n Fp_j[n] sum_k c[t2][t1][n][k] f'_k
            float g12[S::NRM + 1], g21[S::NRM + 1];
            ctab_contract<S, NEPMI_CT_VEC_FORCE != 0>(ctl + (t1 * m.T + t2) * cblk, NRr, KR, fnp, g12);
            ctab_contract<S, NEPMI_CT_VEC_FORCE != 0>(ctl + (t2 * m.T + t1) * cblk, NRr, KR, fnp, g21);
#pragma unroll
            for (int n = 0; n <= S::NRM; ++n) {
              if (!S::fixed && n > NRr)
                break;
              s12 = fmaf(Fpi[n], g12[n], s12);
              s21 = fmaf(Aj[u][n], g21[n], s21);
            }
          } else {
            const float* Ai = atab + (size_t)k * arow + t2 * KRP;
            for (int kk = 0; kk <= KR; ++kk)
              s12 = fmaf(fnp[kk], Ai[kk], s12);
          }
          if (!FPJ) {
#pragma unroll
            for (int kk = 0; kk <= S::KRM; ++kk) {
              if (!S::fixed && kk > KR)
                break;
              s21 = fmaf(fnp[kk], Aj[u][kk], s21);
            }
          }
          const float wgt = live ? dinv : 0.0f;
          const float fs = (s12 + s21) * wgt;
          const float bb = s21 * wgt;
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
      };
      if (CW) {
        // the compact list as words of four slots (Bufs::cword); a sentinel place past the end carries weight zero
        const U2w* __restrict__ cw = reinterpret_cast<const U2w*>(b.cword) + k;
        int nw = (nrad + 3) >> 2;
        nw = nw < b.MN_cw ? nw : b.MN_cw;
        auto load_word = [&](int w) -> U2w {
          U2w v = {0u, 0u};
          if (nw > 0)
            v = cw[(int64_t)(w < nw ? w : nw - 1) * N];
          return v;
        };
        U2w w1 = load_word(0);
        for (int w = 0; w < nw; ++w) {
          const U2w cu = w1;
          w1 = load_word(w + 1);
#pragma unroll 1
          for (int hh = 0; hh < 2; ++hh) {
            const unsigned pr = hh == 0 ? cu.lo : cu.hi;
            const unsigned c2[G] = {pr & 0xFFFFu, pr >> 16};
            chunk(c2, 4 * w + 2 * hh);
          }
        }
      } else {
        unsigned cur[G], nxt[G];
#pragma unroll
        for (int u = 0; u < G; ++u)
          cur[u] = nrad > 0 ? ccode[(int64_t)(G * sub + u < nrad ? G * sub + u : nrad - 1) * N] : 0u;
        for (int s0 = G * sub; s0 < nrad; s0 += G * L) {
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int idx = s0 + G * L + u;
            nxt[u] = ccode[(int64_t)(idx < nrad ? idx : nrad - 1) * N];
          }
          chunk(cur, s0);
#pragma unroll
          for (int u = 0; u < G; ++u)
            cur[u] = nxt[u];
        }
      }
    }

    if (CW || NEPMI_FW_ANG_LAST)
      win_force_angular<L>(b, k, sub, F, Wa);
    if (L > 1) {
#pragma unroll
      for (int msk = 1; msk < L; msk <<= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          F[d] += NEPMI_SHFL_XOR(F[d], msk);
          Fr[d] += NEPMI_SHFL_XOR(Fr[d], msk);
        }
#pragma unroll
        for (int d = 0; d < 9; ++d)
          Wa[d] += NEPMI_SHFL_XOR(Wa[d], msk);
#pragma unroll
        for (int d = 0; d < 6; ++d)
          W[d] += NEPMI_SHFL_XOR(W[d], msk);
      }
      if (sub != 0)
        return;
    }
    // ---- outputs, internal order ----
    double E = b.lvl[k] >= 2 ? (double)b.pe_i[k] : 0.0;
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
    if (!wonly) {
      fo[0] = E;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        fo[(int64_t)(kOutF + d) * N] = Fd[d];
    }
#pragma unroll
    for (int d = 0; d < 9; ++d)
      fo[(int64_t)(kOutW + d) * N] = Wd[d];
  }
};

} // namespace nepmi
