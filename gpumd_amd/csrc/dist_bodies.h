// Kernel bodies of the domain-decomposed driver (dist_impl.h): ownership, ghost selection, halo pack / unpack.
// One work-item per atom or per message entry; same functor style as nep_bodies.h (the test-only emulator runs them
// in host loops).
//
// Replaces the per-step staging of NEP_MULTIGPU::compute (src/force/nep_multigpu.cu:1553-1802: GPU 0 scatters
// slab + halo positions to every GPU and gathers the forces back through blocking peer copies) by persistent
// ownership: every rank keeps its atoms and their integrator state, only ghost POSITIONS travel.
#pragma once
#include "nep_bodies.h"

namespace nepmi {

// geometry of this rank's sub-box in the global cell
struct DomainGeom {
  double H[9], G[9];   // global cell (columns a, b, c) and its inverse
  double origin[3];    // local coordinates = global - origin (the ghost-padded local box starts at 0 in fractional terms)
  double lo[3], hi[3]; // fractional bounds of the sub-box
  double wfrac[3];     // ghost shell 2 (rc + skin) in fractional units
  double ifrac[3];     // inner ring rc + skin (descriptors are recomputed for ghosts inside it)
  int grid[3], coords[3], pbc[3];
  int decomposed[3];   // grid[d] > 1
};

NEPMI_HD void frac_of(const DomainGeom& g, double x, double y, double z, double* s)
{
  s[0] = g.G[0] * x + g.G[1] * y + g.G[2] * z;
  s[1] = g.G[3] * x + g.G[4] * y + g.G[5] * z;
  s[2] = g.G[6] * x + g.G[7] * y + g.G[8] * z;
}

// Owner of every owned atom after a drift: the position (local coordinates, stride n) is wrapped into the global
// cell in place (as a lattice-vector shift, so it stays continuous for the integrator) and the rank whose sub-box
// holds it is written to dest; stay[i] = 1 when that is this rank.
struct OwnerBody {
  DomainGeom g;
  int64_t n;      // stride of x
  int64_t n_own;
  int me;
  double* x;      // [3][n] local coordinates
  int* dest;
  int* stay;
  NEPMI_HD void operator()(int64_t i) const
  {
    double X = x[i] + g.origin[0], Y = x[n + i] + g.origin[1], Z = x[2 * n + i] + g.origin[2];
    double s[3];
    frac_of(g, X, Y, Z, s);
    int c[3];
    for (int d = 0; d < 3; ++d) {
      if (g.pbc[d]) {
        const double f = floor(s[d]);
        if (f != 0.0) {
          s[d] -= f;
          X -= f * g.H[d];
          Y -= f * g.H[3 + d];
          Z -= f * g.H[6 + d];
        }
      }
      int cd = (int)floor(s[d] * g.grid[d]);
      cd = cd < 0 ? 0 : (cd >= g.grid[d] ? g.grid[d] - 1 : cd);
      c[d] = cd;
    }
    x[i] = X - g.origin[0];
    x[n + i] = Y - g.origin[1];
    x[2 * n + i] = Z - g.origin[2];
    const int r = c[0] + g.grid[0] * (c[1] + g.grid[1] * c[2]);
    dest[i] = r;
    stay[i] = r == me ? 1 : 0;
  }
};

// stable compaction: idx_out[scan[i]] = i where flag[i] (scan = exclusive prefix of flag)
struct CompactBody {
  const int* flag;
  const int* scan;
  int* idx_out;
  NEPMI_HD void operator()(int64_t i) const
  {
    if (flag[i])
      idx_out[scan[i]] = (int)i;
  }
};
struct InvertFlagBody {
  const int* in;
  int* out;
  NEPMI_HD void operator()(int64_t i) const { out[i] = in[i] ? 0 : 1; }
};

// owned state of the listed atoms as 9 doubles per atom (global position, velocity, mass, type, id): migration
// payload, [9][cnt] with stride cnt
struct PackStateBody {
  DomainGeom g;
  int64_t n;   // stride of x / v
  int64_t cnt;
  const int* idx;
  const double* x;
  const double* v;
  const double* mass;
  const int* type;
  const int64_t* id;
  double* out;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t i = idx[q];
    for (int d = 0; d < 3; ++d) {
      out[d * cnt + q] = x[d * n + i] + g.origin[d];
      out[(3 + d) * cnt + q] = v[d * n + i];
    }
    out[6 * cnt + q] = mass[i];
    out[7 * cnt + q] = (double)type[i];
    out[8 * cnt + q] = (double)id[i];
  }
};
// new owned arrays: stayers (stable order) followed by arrivals; stride n_new
struct GatherStateBody {
  int64_t n_old, n_new, n_stay;
  const int* stay_idx;
  const double* x_old;
  const double* v_old;
  const double* m_old;
  const int* t_old;
  const int64_t* id_old;
  double* x;
  double* v;
  double* m;
  int* t;
  int64_t* id;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t i = stay_idx[q];
    for (int d = 0; d < 3; ++d) {
      x[d * n_new + q] = x_old[d * n_old + i];
      v[d * n_new + q] = v_old[d * n_old + i];
    }
    m[q] = m_old[i];
    t[q] = t_old[i];
    id[q] = id_old[i];
  }
};
struct UnpackStateBody {
  DomainGeom g;
  int64_t n_new, off, cnt;
  const double* in; // [9][cnt]
  double* x;
  double* v;
  double* m;
  int* t;
  int64_t* id;
  NEPMI_HD void operator()(int64_t q) const
  {
    for (int d = 0; d < 3; ++d) {
      x[d * n_new + off + q] = in[d * cnt + q] - g.origin[d];
      v[d * n_new + off + q] = in[(3 + d) * cnt + q];
    }
    m[off + q] = in[6 * cnt + q];
    t[off + q] = (int)in[7 * cnt + q];
    id[off + q] = (int64_t)in[8 * cnt + q];
  }
};

// Ghost selection of one stage (decomposed direction d): flag the local atoms (owned + ghosts of earlier stages,
// so that edges and corners are forwarded) inside the shell next to the lower / upper face.
struct GhostFlagBody {
  DomainGeom g;
  int64_t n; // stride
  int d;
  int upper; // 0: atoms to send to the lower neighbour, 1: to the upper neighbour
  const double* x;
  int* flag;
  NEPMI_HD void operator()(int64_t i) const
  {
    double s[3];
    frac_of(g, x[i] + g.origin[0], x[n + i] + g.origin[1], x[2 * n + i] + g.origin[2], s);
    flag[i] = upper ? (s[d] >= g.hi[d] - g.wfrac[d] ? 1 : 0) : (s[d] < g.lo[d] + g.wfrac[d] ? 1 : 0);
  }
};

// decomposition-time message of a stage: position (receiver-local coordinates) + type of the listed atoms, [4][cnt]
struct PackGhostBody {
  int64_t n, cnt;
  const int* idx;
  const double* x;
  const int* type;
  double shift[3]; // periodic image + (sender origin - receiver origin)
  double* out;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t i = idx[q];
    for (int d = 0; d < 3; ++d)
      out[d * cnt + q] = x[d * n + i] + shift[d];
    out[3 * cnt + q] = (double)type[i];
  }
};
struct UnpackGhostBody {
  int64_t n, off, cnt;
  const double* in;
  double* x;
  int* type;
  NEPMI_HD void operator()(int64_t q) const
  {
    for (int d = 0; d < 3; ++d)
      x[d * n + off + q] = in[d * cnt + q];
    type[off + q] = (int)in[3 * cnt + q];
  }
};

// levels: 2 owned, 1 ghost inside the inner ring (rc + skin from the sub-box), 0 outer ghost
struct LevelBody {
  DomainGeom g;
  int64_t n, n_own;
  const double* x;
  signed char* level;
  int all_inner; // reverse-mode ghosts: the shell IS the inner ring
  NEPMI_HD void operator()(int64_t i) const
  {
    if (i < n_own) {
      level[i] = 2;
      return;
    }
    if (all_inner) {
      level[i] = 1;
      return;
    }
    double s[3];
    frac_of(g, x[i] + g.origin[0], x[n + i] + g.origin[1], x[2 * n + i] + g.origin[2], s);
    signed char l = 1;
    for (int d = 0; d < 3; ++d)
      if (g.decomposed[d]) {
        const double a = g.lo[d] - s[d], b = s[d] - g.hi[d];
        const double out = a > b ? a : b;
        if (out > g.ifrac[d])
          l = 0;
      }
    level[i] = l;
  }
};

// owned[id[q]] = 1 for the owned atoms q of this rank (Langevin generator states are indexed by global id)
// Langevin generator states travel with their atoms: records of `words` 8-byte words each.
//   out[q] = in[idx[q]] (idx == nullptr: in[q]); one work-item per record
struct GatherRecordsBody {
  const unsigned long long* in;
  const int* idx;
  int words;
  unsigned long long* out;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t i = idx ? (int64_t)idx[q] : q;
    for (int w = 0; w < words; ++w)
      out[q * words + w] = in[i * words + w];
  }
};

struct InversePermBody {
  const int* perm;
  int* inv;
  NEPMI_HD void operator()(int64_t k) const { inv[perm[k]] = (int)k; }
};
struct MapIndexBody { // out[q] = inv[base + in[q]] (in == nullptr: identity)
  const int* inv;
  const int* in;
  int64_t base;
  int* out;
  NEPMI_HD void operator()(int64_t q) const { out[q] = inv[base + (in ? in[q] : (int)q)]; }
};

// ---------------------------------------------------------------------------------------------------------------------
// Direct halo: every rank exchanges with its (up to 26) neighbours of the process grid in ONE grouped exchange -- faces,
// edges and corners as separate messages, no forwarding through intermediate ranks (every peer of a 2x2x2
// grid on an 8-GPU node is a direct xGMI link).  Peer p has the grid offset off[p] in {-1, 0, +1}^3 (0 along directions
// that are not decomposed); an owned atom goes to p when it lies in the shell next to every face off[p] points through.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxPeers = 26;
struct PeerTable {
  int n;
  int off[kMaxPeers][3];
  double shift[kMaxPeers][3]; // added to a position sent to peer p: the receiver's local coordinates
};

// decomposition: bit p of mask[i] = owned atom i goes to peer p; any[i] = mask != 0
struct PeerMaskBody {
  DomainGeom g;
  PeerTable pt;
  int64_t n; // stride of x
  const double* x;
  unsigned* mask;
  int* any;
  NEPMI_HD void operator()(int64_t i) const
  {
    double s[3];
    frac_of(g, x[i] + g.origin[0], x[n + i] + g.origin[1], x[2 * n + i] + g.origin[2], s);
    bool lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
      lo[d] = s[d] < g.lo[d] + g.wfrac[d];
      hi[d] = s[d] >= g.hi[d] - g.wfrac[d];
    }
    unsigned m = 0u;
    for (int p = 0; p < pt.n; ++p) {
      bool ok = true;
      for (int d = 0; d < 3; ++d)
        ok = ok && (pt.off[p][d] == 0 || (pt.off[p][d] < 0 ? lo[d] : hi[d]));
      m |= ok ? (1u << p) : 0u;
    }
    mask[i] = m;
    any[i] = m != 0u ? 1 : 0;
  }
};
// flag[p (m + 1) + j] = shell atom j (= owned atom L[j]) goes to peer p; the slot j = m of every block and the last word stay 0:
// ONE exclusive scan of the whole array then numbers the send entries peer-major (entry = scan value) and its value at the
// start of block p is the offset of peer p's message
struct PeerFlagBody {
  const unsigned* mask;
  const int* L;
  int64_t m;
  int npeers;
  int* flag;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t p = q / (m + 1), j = q - p * (m + 1);
    flag[q] = (p < npeers && j < m && ((mask[L[j]] >> p) & 1u)) ? 1 : 0;
  }
};
struct PeerFillBody { // send_idx / send_peer of every entry; `scan` = the exclusive scan of PeerFlagBody's array
  const unsigned* mask;
  const int* L;
  int64_t m;
  int npeers;
  const int* scan;
  int* send_idx;
  unsigned char* send_peer;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t p = q / (m + 1), j = q - p * (m + 1);
    if (j < m && ((mask[L[j]] >> p) & 1u)) {
      send_idx[scan[q]] = L[j];
      send_peer[scan[q]] = (unsigned char)p;
    }
  }
};
struct PeerOffsetsBody { // out[p] = first entry of peer p, out[npeers] = number of entries
  const int* scan;
  int64_t m;
  int npeers;
  int* out;
  NEPMI_HD void operator()(int64_t p) const { out[p] = scan[p * (m + 1)]; }
};
// reverse path: the entries of every shell atom in ascending peer order (what GhostAddPeersBody adds, in that order)
struct PeerCountBody { // cnt[j] = number of peers of shell atom j (cnt[m] = 0)
  const unsigned* mask;
  const int* L;
  int64_t m;
  int* cnt;
  NEPMI_HD void operator()(int64_t j) const
  {
    unsigned v = j < m ? mask[L[j]] : 0u;
    int c = 0;
    for (; v; v &= v - 1u)
      ++c;
    cnt[j] = c;
  }
};
struct PeerCsrBody {
  const unsigned* mask;
  const int* L;
  int64_t m;
  int npeers;
  const int* scan;  // entry numbers (PeerFlagBody's array, scanned)
  const int* start; // exclusive scan of PeerCountBody's counts
  int* entry;
  NEPMI_HD void operator()(int64_t j) const
  {
    const unsigned v = mask[L[j]];
    int pos = start[j];
    for (int p = 0; p < npeers; ++p)
      if ((v >> p) & 1u)
        entry[pos++] = scan[(int64_t)p * (m + 1) + j];
  }
};
// decomposition-time payload: [x y z type] of every send entry, entry-major (a peer's message is a contiguous range)
struct PackGhostPeersBody {
  int64_t n, cnt; // stride of x, number of entries
  const int* idx;
  const unsigned char* peer;
  const double* x;
  const int* type;
  PeerTable pt;
  double* out; // [cnt][4]
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t i = idx[q];
    const int p = peer[q];
    for (int d = 0; d < 3; ++d)
      out[4 * q + d] = x[d * n + i] + pt.shift[p][d];
    out[4 * q + 3] = (double)type[i];
  }
};
struct UnpackGhostPeersBody {
  int64_t n, off, cnt;
  const double* in; // [cnt][4]
  double* x;
  int* type;
  NEPMI_HD void operator()(int64_t q) const
  {
    for (int d = 0; d < 3; ++d)
      x[d * n + off + q] = in[4 * q + d];
    type[off + q] = (int)in[4 * q + 3];
  }
};
// per step, forward: positions of the send entries (internal indices) + their peer's shift -> [cnt][3]
struct HaloPackPeersBody {
  Bufs b;
  const int* idx;
  const unsigned char* peer;
  PeerTable pt;
  double* out;
  NEPMI_HD void operator()(int64_t q) const
  {
    const PosQ p = b.posq[idx[q]];
    const int r = peer[q];
    out[3 * q] = p.x + pt.shift[r][0];
    out[3 * q + 1] = p.y + pt.shift[r][1];
    out[3 * q + 2] = p.z + pt.shift[r][2];
  }
};
// received ghost positions [cnt][3] -> posq (internal order), with the lattice-jump bookkeeping and the fixed-point record
// of CheckGatherBody
struct HaloUnpackPeersBody {
  BoxD box;
  Bufs b;
  const int* idx; // internal indices of the ghosts
  const double* in;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t N = b.N;
    const int k = idx[q];
    const double x = in[3 * q], y = in[3 * q + 1], z = in[3 * q + 2];
    float dx = (float)(x - b.x0s[k]);
    float dy = (float)(y - b.x0s[N + k]);
    float dz = (float)(z - b.x0s[2 * N + k]);
    int n0, n1, n2;
    mic_f_img(box, dx, dy, dz, n0, n1, n2);
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
// per step, reverse (reverse-mode ghosts): planes [first, first + planes) of Bufs::fo of every ghost -> [cnt][planes], back to
// the rank that owns it ...
struct GhostPackPeersBody {
  Bufs b;
  const int* idx; // internal indices of the ghosts
  int first, planes;
  double* out;
  const int* frozen; // the run loops' "rebuild pending" word (the force kernels of this step did not run), or nullptr
  NEPMI_HD void operator()(int64_t q) const
  {
    if (frozen && *frozen != 0)
      return;
    const int64_t N = b.N;
    const int k = idx[q];
    for (int p = 0; p < planes; ++p)
      out[(int64_t)planes * q + p] = b.fo[(int64_t)(first + p) * N + k];
  }
};
// ... where one work-item per shell atom adds what its images collected, in ascending peer order (a fixed order: the sum does
// not depend on which message arrived first)
struct GhostAddPeersBody {
  Bufs b;
  const int* src_int;   // internal index of shell atom j
  const int* src_start; // its entries: src_entry[src_start[j] .. src_start[j + 1])
  const int* src_entry;
  int first, planes;
  const double* in; // [entries][planes]
  const int* frozen;
  NEPMI_HD void operator()(int64_t j) const
  {
    if (frozen && *frozen != 0)
      return;
    const int64_t N = b.N;
    const int k = src_int[j];
    for (int t = src_start[j]; t < src_start[j + 1]; ++t) {
      const int64_t e = src_entry[t];
      for (int p = 0; p < planes; ++p)
        b.fo[(int64_t)(first + p) * N + k] += in[(int64_t)planes * e + p];
    }
  }
};

// Per-step halo of one stage: positions of the listed atoms (internal indices) + the message's shift -> send buffer
// [3][cnt] per message; both messages of the stage in one launch (entries of the second message follow the first).
struct HaloPackBody {
  Bufs b;
  const int* idx; // internal indices, cnt0 + cnt1 entries
  int64_t cnt0, cnt1;
  double shift0[3], shift1[3];
  double* out0;
  double* out1;
  NEPMI_HD void operator()(int64_t q) const
  {
    const PosQ p = b.posq[idx[q]];
    if (q < cnt0) {
      out0[q] = p.x + shift0[0];
      out0[cnt0 + q] = p.y + shift0[1];
      out0[2 * cnt0 + q] = p.z + shift0[2];
    } else {
      const int64_t r = q - cnt0;
      out1[r] = p.x + shift1[0];
      out1[cnt1 + r] = p.y + shift1[1];
      out1[2 * cnt1 + r] = p.z + shift1[2];
    }
  }
};
// Reverse-mode ghosts: planes [first, first + planes) of Bufs::fo (forces: kOutF, 3; virials: kOutW, 9) of the ghosts one
// stage received -> [planes][cnt] per message, back to the ranks the ghosts came from ...
struct GhostForcePackBody {
  Bufs b;
  const int* idx; // internal indices of the received ghosts, cnt0 + cnt1 entries
  int64_t cnt0, cnt1;
  int first, planes;
  double* out0;
  double* out1;
  const int* frozen; // the run loops' "rebuild pending" word (the force kernels of this step did not run), or nullptr
  NEPMI_HD void operator()(int64_t q) const
  {
    if (frozen && *frozen != 0)
      return;
    const int64_t N = b.N;
    const int k = idx[q];
    const bool lower = q < cnt0;
    const int64_t r = lower ? q : q - cnt0, cnt = lower ? cnt0 : cnt1;
    double* out = lower ? out0 : out1;
    for (int p = 0; p < planes; ++p)
      out[p * cnt + r] = b.fo[(int64_t)(first + p) * N + k];
  }
};
// ... where they are added to the atoms that were sent (owned atoms, or ghosts of an earlier stage that travel on in the next
// reverse stage).  The two messages of a stage in one launch when no atom sits in both send lists (a sub-box at least two
// shells wide: DistT::reverse_exchange), else one launch per message.
struct GhostForceAddBody {
  Bufs b;
  const int* idx; // internal indices of the atoms sent: cnt0 of message 0, then cnt1 of message 1
  int64_t cnt0, cnt1;
  int first, planes;
  const double* in0;
  const double* in1;
  const int* frozen;
  NEPMI_HD void operator()(int64_t q) const
  {
    if (frozen && *frozen != 0)
      return;
    const int64_t N = b.N;
    const int k = idx[q];
    const bool lower = q < cnt0;
    const int64_t r = lower ? q : q - cnt0, cnt = lower ? cnt0 : cnt1;
    const double* in = lower ? in0 : in1;
    for (int p = 0; p < planes; ++p)
      b.fo[(int64_t)(first + p) * N + k] += in[p * cnt + r];
  }
};
// received ghost positions -> posq (internal order), with the lattice-jump bookkeeping and the fixed-point record of
// CheckGatherBody (a ghost of a direction that is periodic in the local box wraps with its owner)
struct HaloUnpackBody {
  BoxD box;
  Bufs b;
  const int* idx; // internal indices of the ghosts, cnt0 + cnt1 entries
  int64_t cnt0, cnt1;
  const double* in0;
  const double* in1;
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t N = b.N;
    const int k = idx[q];
    double x, y, z;
    if (q < cnt0) {
      x = in0[q];
      y = in0[cnt0 + q];
      z = in0[2 * cnt0 + q];
    } else {
      const int64_t r = q - cnt0;
      x = in1[r];
      y = in1[cnt1 + r];
      z = in1[2 * cnt1 + r];
    }
    float dx = (float)(x - b.x0s[k]);
    float dy = (float)(y - b.x0s[N + k]);
    float dz = (float)(z - b.x0s[2 * N + k]);
    int n0, n1, n2;
    mic_f_img(box, dx, dy, dz, n0, n1, n2);
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

// thermo8 from the all-reduced raw sums (find_thermo, ensemble.cu:434-673)
struct ThermoNormBody {
  const double* sums;
  double n_total, volume;
  double* thermo8;
  NEPMI_HD void operator()(int64_t i) const
  {
    if (i != 0)
      return;
    thermo8[0] = sums[0] / (3.0 * n_total * 8.617343e-5);
    thermo8[1] = sums[1];
    for (int q = 2; q < 8; ++q)
      thermo8[q] = sums[q] / volume;
  }
};

// owned atoms of the engine's internal arrays -> packed output in local order (ids ascending is the caller's job)
struct GatherOwnedBody {
  Bufs b;
  DomainGeom g;
  int64_t n_own;
  const int64_t* id_local; // [n_loc] local order
  int64_t* ids;            // outputs, n_own entries each (stride n_own); any may be nullptr
  double* pos;
  double* vel;
  double* force;
  double* pe;
  double* virial;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t i = b.perm[k];
    if (i >= n_own)
      return;
    if (ids)
      ids[i] = id_local[i];
    const PosQ p = b.posq[k];
    if (pos) {
      pos[i] = p.x + g.origin[0];
      pos[n_own + i] = p.y + g.origin[1];
      pos[2 * n_own + i] = p.z + g.origin[2];
    }
    if (vel)
      for (int d = 0; d < 3; ++d)
        vel[d * n_own + i] = b.vi[d * N + k];
    if (force)
      for (int d = 0; d < 3; ++d)
        force[d * n_own + i] = b.fo[(kOutF + d) * N + k];
    if (pe)
      pe[i] = b.fo[k];
    if (virial)
      for (int d = 0; d < 9; ++d)
        virial[d * n_own + i] = b.fo[(kOutW + d) * N + k];
  }
};

// owned atoms as 20 doubles per atom (id, global position, velocity, force, pe, 9 virial planes): [20][n_own]
struct PackGlobalBody {
  Bufs b;
  DomainGeom g;
  int64_t n_own;
  const int64_t* id_local;
  double* out;
  NEPMI_HD void operator()(int64_t k) const
  {
    const int64_t N = b.N;
    const int64_t i = b.perm[k];
    if (i >= n_own)
      return;
    const PosQ p = b.posq[k];
    out[i] = (double)id_local[i];
    out[1 * n_own + i] = p.x + g.origin[0];
    out[2 * n_own + i] = p.y + g.origin[1];
    out[3 * n_own + i] = p.z + g.origin[2];
    for (int d = 0; d < 3; ++d) {
      out[(4 + d) * n_own + i] = b.vi[d * N + k];
      out[(7 + d) * n_own + i] = b.fo[(kOutF + d) * N + k];
    }
    out[10 * n_own + i] = b.fo[k];
    for (int d = 0; d < 9; ++d)
      out[(11 + d) * n_own + i] = b.fo[(kOutW + d) * N + k];
  }
};
// one rank's payload -> the global arrays (stride n_total), by atom id
struct ScatterGlobalBody {
  int64_t cnt, n_total;
  const double* in; // [20][cnt]
  double* pos;
  double* vel;
  double* force;
  double* pe;
  double* virial;
  int* bad; // set when an id is out of range
  NEPMI_HD void operator()(int64_t q) const
  {
    const int64_t id = (int64_t)in[q];
    if (id < 0 || id >= n_total) {
      *bad = 1;
      return;
    }
    for (int d = 0; d < 3; ++d) {
      if (pos) pos[d * n_total + id] = in[(1 + d) * cnt + q];
      if (vel) vel[d * n_total + id] = in[(4 + d) * cnt + q];
      if (force) force[d * n_total + id] = in[(7 + d) * cnt + q];
    }
    if (pe) pe[id] = in[10 * cnt + q];
    if (virial)
      for (int d = 0; d < 9; ++d)
        virial[d * n_total + id] = in[(11 + d) * cnt + q];
  }
};

} // namespace nepmi
