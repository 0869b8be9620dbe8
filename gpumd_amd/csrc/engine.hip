// libnepmi.so -- gfx950 (MI355X) build of the NEP force engine: HIP backend + C ABI.
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared engine.hip nep_model.cpp -o libnepmi.so
//
// All force-path kernels are the per-atom bodies of nep_bodies.h launched through
// nepmi_kernel<BLOCK, Body>; the block-cooperative kernels (prefix scan of the cell histogram,
// thermo reduction) live here.  One engine = one HIP stream; no host<->device traffic on the hot
// path other than the 32-byte flag read-back per force call (the reference's 4-byte skin-check
// D2H, neighbor.cu:752, extended with overflow flags).
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h> // XORWOW states of the Langevin thermostat (device API only)

#include <cstring>
#include <stdexcept>
#include <string>

#include "engine_impl.h"
#include "nep_scatter.h" // device-only force assembly (LDS scatter of the own pair halves)
#include "nep_fused.h"   // device-only: angular descriptor + ANN + partial angular forces in one kernel
#ifndef NEPMI_WITH_BRICK
#define NEPMI_WITH_BRICK 0 // 1 (make BRICK=1): also the one-force-kernel-per-brick experiment (experimental/nep_brick.h: built, parity-tested,
                           // measured slower than the two kernels it replaces -- not part of the default library or of the JIT cores)
#endif
#if NEPMI_WITH_BRICK
#include "experimental/nep_brick.h" // device-only: ... and the scatter-form force assembly behind them, one kernel per brick
#endif

namespace nepmi {

#define NEPMI_HIP_CHECK(expr)                                                                       \
  do {                                                                                              \
    hipError_t err__ = (expr);                                                                      \
    if (err__ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(err__) + " at " +     \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");         \
  } while (0)

// One work-item per atom.  Workgroup -> tile mapping is XCD-aware: the dispatcher places workgroup
// b on XCD b % 8 (MI355X_MICROARCH.md, observed, used for speed only), so XCD x is given the
// contiguous tile range [x * gridDim/8, (x+1) * gridDim/8).  With the brick-major atom order this
// keeps each XCD's private 4 MiB L2 on one compact slab of the crystal instead of the whole box.
// gridDim.x is always a multiple of 8 (surplus tiles exit).
// `frozen` (may be null): device word of the fused run loops; non-zero = a list rebuild is pending and the
// force path of this (speculatively enqueued) step must not run.
template <int BLOCK, class Body>
__global__ void __launch_bounds__(BLOCK) nepmi_kernel(const Body body, const int64_t n, const int* frozen)
{
  if (frozen && *frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i = (int64_t)tile * BLOCK + threadIdx.x;
  if (i < n)
    body(i);
}

// Same mapping, for bodies that stage a read-only table (descriptor coefficients) in LDS first.
template <int BLOCK, class Body>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEu)))
nepmi_kernel_lds(const Body body, const int64_t n, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_lds[];
  if (frozen && *frozen != 0)
    return;
  body.lds_stage(nepmi_lds, (int)threadIdx.x, BLOCK);
  __syncthreads();
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i = (int64_t)tile * BLOCK + threadIdx.x;
  if (i < n)
    body.run(i, (lds_cfloat_ptr)nepmi_lds);
}

// ---- per-atom ANN on the matrix cores ---------------------------------------------------------
// The ANN of one type is two dense contractions over a batch of atoms (nep.cu:521-577 calls
// apply_ann_one_layer per atom):  H = W0 Q  (neurons x dim  times  dim x atoms)  and
// dE/dq = W0^T C  (dim x neurons  times  neurons x atoms,  C = w1 (1 - tanh^2)).
// Work order: one 256-thread workgroup per 1024-atom chunk; inside a chunk the atoms are grouped by
// type (tperm/tcount), so every wave tile is type-pure.  Per type the workgroup stages the weights
// in LDS already in v_mfma_f32_32x32x2_f32 operand order (A: lane l holds A[i = l & 31][k = l >> 5]),
// then each wave runs 64 atoms as two 32-column tiles:
//   forward   acc[mt] += A(W0 rows 32mt.., k-pair s) x B(q[2s + (l >> 5)][atom l & 31])
//   backward  the forward accumulator layout (row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = l & 31)
//             is exactly a B operand for the k-pair (neuron n, n + 4), so C is consumed in place.
// The backward weight matrix carries extra output rows  sum_n qs[n] c[t1][t2][n][k] W0[neuron][n],
// which makes the radial force table A_i[t2][k] (see AnnBody) fall out of the same MFMA chain.
// f32 MFMA is an exact k-ordered fmaf chain, so this differs from AnnBody by summation order only.
typedef float nepmi_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kAnnSplit = 2;               // workgroups per 1024-atom chunk
constexpr int kAnnStride = 4 * kAnnSplit;  // wave tiles per pass over a chunk

// tanh(x) = 1 - 2 / (exp(2x) + 1), branch-free (v_exp_f32 + v_rcp_f32); absolute error ~1e-7, the
// rounding level of the f32 hidden activations themselves.  Saturates correctly at +-inf.
__device__ __forceinline__ float tanh_fast(float x)
{
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// register budget: two accumulator sets (16 MT + 16 DT), the C operand (16 MT, partly dead), the
// q operand buffer (QB k-pairs) and ~60 working registers
constexpr int ann_mfma_waves(int MT, int DT, int QB)
{
  return 16 * (MT + DT) + 60 + 12 * MT + QB <= 128 ? 4 : 16 * (MT + DT) + 60 + 12 * MT + QB <= 168 ? 3 : 2;
}

// One-off: write type blockIdx.x's weight image (layout: AnnMfmaShape) in MFMA operand order.
__global__ void __launch_bounds__(256) nepmi_ann_pack(const ModelD m, const Bufs b, const int MT, const int DT)
{
  const int dim = m.dim, nneu = m.nneu, T = m.T, NR = m.NR, KR = m.KR, KRP = b.KRP;
  const int KS = (dim + 1) >> 1;
  const int tu = blockIdx.x, tid = threadIdx.x;
  const size_t img_floats = (size_t)(KS * MT + MT * 16 * DT) * 64 + 2 * MT * 32;
  float* Wf = b.ann_img + tu * img_floats; // [KS][MT][64]
  float* Wb = Wf + KS * MT * 64;           // [MT*16][DT][64]
  float* B0 = Wb + MT * 16 * DT * 64;      // [MT*32]
  float* W1 = B0 + MT * 32;                // [MT*32]
  const float* w0 = m.w0 + (size_t)tu * nneu * dim;
  for (int idx = tid; idx < KS * MT * 64; idx += 256) {
    const int l = idx & 63, mt = (idx >> 6) % MT, s = (idx >> 6) / MT;
    const int neuron = mt * 32 + (l & 31), kk = 2 * s + (l >> 5);
    Wf[idx] = (neuron < nneu && kk < dim) ? w0[neuron * dim + kk] : 0.0f;
  }
  for (int idx = tid; idx < MT * 16 * DT * 64; idx += 256) {
    const int l = idx & 63, dt = (idx >> 6) % DT, step = (idx >> 6) / DT;
    const int r = step & 15, mt = step >> 4;
    const int neuron = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int d = dt * 32 + (l & 31);
    float v = 0.0f;
    if (neuron < nneu) {
      if (d < dim) {
        v = w0[neuron * dim + d] * m.qscale[d];
      } else {
        const int xr = d - dim, t2 = xr / KRP, kk = xr - t2 * KRP;
        if (t2 < T && kk <= KR) {
          const float* c = m.c_rad + (size_t)(tu * T + t2) * (NR + 1) * (KR + 1);
          for (int n = 0; n <= NR; ++n)
            v = fmaf(m.qscale[n] * w0[neuron * dim + n], c[n * (KR + 1) + kk], v);
        }
      }
    }
    Wb[idx] = v;
  }
  for (int idx = tid; idx < MT * 32; idx += 256) {
    B0[idx] = idx < nneu ? m.b0[(size_t)tu * nneu + idx] : 0.0f;
    W1[idx] = idx < nneu ? m.w1[(size_t)tu * nneu + idx] : 0.0f;
  }
}

// BYTYPE (models with more than four types, e.g. UNEP-v1's 16): a 1024-atom chunk then holds ~64 atoms of a type -- one wave
// tile -- and a workgroup that staged one type's image (39 KB) for a single tile would spend its time staging.  Instead one
// workgroup serves ONE type over kAnnGroup consecutive chunks: the image is staged once, the four waves take that type's tiles
// chunk after chunk.  Grid: ceil(nchunks / kAnnGroup) x T workgroups.
constexpr int kAnnGroup = 16;
template <int MT, int DT, int QB, bool BYTYPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ann_mfma_waves(MT, DT, QB))))
nepmi_ann_mfma(const ModelD m, const Bufs b, const int64_t nchunks, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_ann_lds[];
  if (frozen && *frozen != 0)
    return;
  const int dim = m.dim, T = m.T, KRP = b.KRP;
  const int KS = (dim + 1) >> 1;
  const int img_floats = (KS * MT + MT * 16 * DT) * 64 + 2 * MT * 32;
  const float* Wf = nepmi_ann_lds;          // [KS][MT][64]
  const float* Wb = Wf + KS * MT * 64;      // [MT*16][DT][64]
  const float* B0 = Wb + MT * 16 * DT * 64; // [MT*32]
  const float* W1 = B0 + MT * 32;           // [MT*32]
  // kAnnSplit workgroups share one chunk (more, shorter workgroups: less tail at 3 waves/SIMD)
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t c_first = BYTYPE ? (wg / T) * kAnnGroup : wg / kAnnSplit;
  if (c_first >= nchunks)
    return;
  const int64_t c_last = BYTYPE ? (c_first + kAnnGroup < nchunks ? c_first + kAnnGroup : nchunks) : c_first + 1;
  const int64_t N = b.N;
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, col = lane & 31;
  // tile slot of this wave within the chunk.  BYTYPE: a (chunk, type) segment is about ONE tile (1024 atoms / 16 types) and a
  // second, nearly empty tile half of the time, so the type's segments of the group's chunks are walked as ONE flat range:
  // flat index f -> work index seg_lo[s] + f - seg_pre[s] (two small tables in LDS behind the image); the four waves take
  // every fourth tile of that range
  const int wave = BYTYPE ? (tid >> 6) : (tid >> 6) + 4 * (int)(wg % kAnnSplit);
  constexpr int kStride = BYTYPE ? 4 : kAnnStride;
  int* seg_lo = reinterpret_cast<int*>(nepmi_ann_lds + img_floats); // [kAnnGroup]
  int* seg_pre = seg_lo + kAnnGroup;                                // [kAnnGroup + 1] exclusive prefix of the segment lengths
  auto WI = [&](const int f) __attribute__((always_inline)) -> int { // flat index -> work index (column of q / fp, entry of tperm)
    if (!BYTYPE)
      return f;
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < kAnnGroup; ++i)
      sidx += (f >= seg_pre[i]) ? 1 : 0;
    return seg_lo[sidx] + (f - seg_pre[sidx]);
  };
  const int t_first = BYTYPE ? (int)(wg % T) : 0, t_last = BYTYPE ? t_first + 1 : T;
  for (int tu = t_first; tu < t_last; ++tu) {
    if (!BYTYPE && b.tcount[c_first * T + tu] == b.tcount[c_first * T + tu + 1])
      continue;
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(b.ann_img + (size_t)tu * img_floats);
      float4* dst = reinterpret_cast<float4*>(nepmi_ann_lds);
      for (int idx = tid; idx < img_floats / 4; idx += 256)
        dst[idx] = src[idx];
    }
    if (BYTYPE && tid == 0) {
      int run = 0;
      for (int i = 0; i < kAnnGroup; ++i) {
        const int64_t c = c_first + i;
        const int l0 = c < c_last ? b.tcount[c * T + tu] : 0, l1 = c < c_last ? b.tcount[c * T + tu + 1] : 0;
        seg_lo[i] = l0;
        seg_pre[i] = run;
        run += l1 - l0;
      }
      seg_pre[kAnnGroup] = run;
    }
    __syncthreads();
   for (int64_t chunk = c_first; chunk < (BYTYPE ? c_first + 1 : c_last); ++chunk) {
    const int lo = BYTYPE ? 0 : b.tcount[chunk * T + tu], end = BYTYPE ? seg_pre[kAnnGroup] : b.tcount[chunk * T + tu + 1];
    if (lo == end)
      continue;
    const float ebias = m.b1 + m.b1t[tu];
    const nepmi_f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const int nrows = T > 4 ? dim : dim + T * KRP; // (more than four types: no radial-table rows, the force assembly contracts from Bufs::fpr)
    // This wave's tiles: lo + 64 (wave + 4 i).  The loop runs over units = (tile, 32-column half);
    // the q rows of unit u+1 are requested right after unit u's forward MFMAs have consumed the
    // operand registers -- i.e. before unit u's stores, so that (vmcnt being in-order) waiting for
    // them never waits for a store, and the round trip overlaps the backward MFMAs.
    const int ntiles = (end - lo + 63) >> 6;
    if (wave >= ntiles)
      continue;
    const int nunits = 2 * ((ntiles - wave + kStride - 1) / kStride);
    int tile = lo + wave * 64;
    int k_cur, act_cur, k_nxt = 0, act_nxt = 0;
    {
      const int gl = tile + lane;
      k_cur = b.tperm[WI(gl < end ? gl : lo)];
      act_cur = (gl < end && b.lvl[k_cur] >= b.lvl_desc) ? 1 : 0;
    }
    // q and fp are stored in work order: column g of the [dim][N] arrays is work item g
    int gc = min(tile + col, end - 1);
    float bq[QB];
#pragma unroll
    for (int s = 0; s < QB; ++s)
      bq[s] = b.q[(int64_t)min(2 * s + hi, dim - 1) * N + WI(gc)];
    float e_own = 0.0f;
#pragma unroll 1
    for (int u = 0; u < nunits; ++u) {
      const int nt = u & 1;
      // keep the weight reads and the row arithmetic inside the loop (registers, not LICM)
      int hi_o = hi;
      asm volatile("" : "+v"(hi_o) : : "memory");
      if (nt == 0) {
        const int gl = tile + 64 * kStride + lane;
        const bool more = tile + 64 * kStride < end;
        k_nxt = b.tperm[WI((more && gl < end) ? gl : lo)];
        act_nxt = (more && gl < end && b.lvl[k_nxt] >= b.lvl_desc) ? 1 : 0;
      }
      const int actc = __shfl(act_cur, nt * 32 + col);
      nepmi_f32x16 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = zero16;
#pragma unroll
      for (int s = 0; s < QB; ++s) {
        if (s < KS) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wf[(s * MT + mt) * 64 + lane], bq[s], acc[mt], 0, 0, 0);
        }
      }
      const int kc = __shfl(k_cur, nt * 32 + col);
      const int gc_n = min(nt == 0 ? tile + 32 + col : tile + 64 * kStride + col, end - 1);
      const int gcw_n = WI(gc_n);
      if (u + 1 < nunits) {
#pragma unroll
        for (int s = 0; s < QB; ++s)
          bq[s] = b.q[(int64_t)min(2 * s + hi_o, dim - 1) * N + gcw_n];
      }
      float e_part = 0.0f;
      float cf[MT][16]; // C = w1 (1 - tanh^2): the backward B operand, same lane layout as acc
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int neuron = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi_o;
          const float h = tanh_fast(acc[mt][r] - B0[neuron]);
          const float wj = W1[neuron];
          e_part = fmaf(wj, h, e_part);
          cf[mt][r] = wj * (1.0f - h * h);
        }
      const float e_tot = e_part + __shfl_xor(e_part, 32);
      if (hi == nt)
        e_own = e_tot;
      nepmi_f32x16 out[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        out[dt] = zero16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bc = cf[mt][r];
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              Wb[((mt * 16 + r) * DT + dt) * 64 + lane], bc, out[dt], 0, 0, 0);
        }
      if (actc) {
        // rows in register order: d = 32 dt + {0,1,2,3, 8,.., 27} + 4 hi, i.e. steps of +1,+1,+1,+5;
        // rows [dim, dim + T KRP) are the radial-table rows, laid out exactly like atab's row
        int d = 4 * hi_o;
        float* pf = b.fp + WI(gc) + (int64_t)d * N;
        float* pa = b.atab + (size_t)kc * (T * KRP) - dim;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = out[dt][r];
            if (d < dim) {
              *pf = v;
              if (b.fpr && d <= m.NR) // the radial rows again, atom-major (what AnnBody writes for the many-type force assembly)
                b.fpr[(size_t)kc * b.FPR + d] = v;
            } else if (d < nrows)
              pa[d] = v;
            const int step = (r & 3) == 3 ? 5 : 1;
            d += step;
            pf += (int64_t)step * N;
          }
      }
      if (nt == 1) {
        if (act_cur)
          b.pe_i[k_cur] = e_own - ebias;
        k_cur = k_nxt;
        act_cur = act_nxt;
        tile += 64 * kStride;
      }
      gc = gc_n;
    }
   }
  }
}

// Two adjacent lanes per atom (Body::run_parts<2>): for bodies whose per-atom register table is what
// limits them to one wavefront per SIMD (angular force).
template <int BLOCK, class Body>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEuPairs)))
nepmi_kernel_lds_pairs(const Body body, const int64_t n, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_lds_pairs[];
  if (frozen && *frozen != 0)
    return;
  body.lds_stage(nepmi_lds_pairs, (int)threadIdx.x, BLOCK);
  __syncthreads();
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i = ((int64_t)tile * BLOCK + threadIdx.x) >> 1;
  if (i < n)
    body.template run_parts<2>(i, (int)(threadIdx.x & 1u), (lds_cfloat_ptr)nepmi_lds_pairs);
}

// The fused angular kernel of MANY-TYPE models (nep_fused.h: AngularFusedBody with a type window): 256 threads = 128 atoms, two lanes
// each, taken in the type-sorted work order of the matrix-core ANN (Bufs::tperm: sorted by type inside every 1024-atom chunk, and a
// workgroup lies inside one chunk), so a workgroup's atoms span a few consecutive types.  It stages those types' slices of the image
// -- kFusedWindowTypes at a time; a range that is longer (rare types) takes another pass -- and every lane pair whose type is
// resident runs the whole chain descriptor -> ANN -> adjoint -> partial forces.
// MEASURED SLOWER than the three kernels it replaces (profiles/r6i_ab_fused_window.txt: UNEP-v1 1 M atoms 2.19 ms against 0.51 + 0.32 +
// 0.97): in the type-sorted order the 16-byte pair records of a wavefront's atoms lie in 32 different 64-byte sectors (35 pairs,
// read twice: 4.5 GB through the L2 per step against 1.1 GB in brick order), and brick order would need every type's weights at
// once.  Compiled only with -DNEPMI_WITH_FUSED_WINDOW=1 (make FUSED_WINDOW=1); the default step keeps descriptor / matrix-core ANN /
// partial forces as three launches.
#ifndef NEPMI_WITH_FUSED_WINDOW
#define NEPMI_WITH_FUSED_WINDOW 0
#endif
constexpr int kFusedWindowTypes = 4;
#if NEPMI_WITH_FUSED_WINDOW
template <class Body>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEuPairs)))
nepmi_fused_window_kernel(const Body body, const int64_t n, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_lds_window[];
  if (frozen && *frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i0 = wg * 128;
  if (i0 >= n)
    return;
  const int tid = (int)threadIdx.x;
  const int64_t i = i0 + (tid >> 1);
  const bool live = i < n;
  const int64_t k = body.b.tperm[live ? i : i0];
  const int t1 = body.b.posq[k].type;
  const int64_t ilast = i0 + 127 < n ? i0 + 127 : n - 1;
  const int t_lo = body.b.posq[body.b.tperm[i0]].type, t_hi = body.b.posq[body.b.tperm[ilast]].type;
  for (int tb = t_lo; tb <= t_hi; tb += kFusedWindowTypes) {
    const int nt = body.m.T - tb < kFusedWindowTypes ? body.m.T - tb : kFusedWindowTypes;
    __syncthreads(); // (the lanes of the pass before are done with the window)
    body.stage_window(nepmi_lds_window, tb, nt, tid, 256);
    __syncthreads();
    if (live && t1 >= tb && t1 < tb + nt)
      body.run_window(k, tid & 1, (lds_cfloat_ptr)nepmi_lds_window, tb);
  }
}
#endif

// The same with the workgroup's atoms handed to the lane pairs IN THE ORDER OF THEIR NUMBER OF ANGULAR NEIGHBOURS (Body::sort_key).
// The angular loops run in lockstep: a wavefront walks the longest list among its 32 atoms (PbTe: 6.1 neighbours on average, ~10
// for the longest of 32), and the lanes of the shorter lists wait -- 40 % of the lane time of those loops.  A counting sort of the
// 128 keys in LDS (a histogram, one short serial prefix, one atomic per atom: about a microsecond of a ~30 us workgroup) gives the
// first wavefront the longest lists, the last one the shortest: the four wavefronts walk ~10 + 7 + 6 + 5 rows instead of 4 x ~10.
// Even workgroups sort descending, odd ones ascending, so that the wavefronts that share a SIMD (wave i of either workgroup on a
// CU) are one heavy and one light.  Which lane pair serves which atom has no influence on the atom's results (every output is
// indexed by the atom), so the results are the unsorted kernel's bit for bit whatever order equal keys end up in; the records
// of the 128 atoms lie in one 2 KB span per row either way.
template <int BLOCK, class Body>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEuPairs)))
nepmi_kernel_lds_pairs_sorted(const Body body, const int64_t n, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_lds_pairs_s[];
  __shared__ int hist[64];
  __shared__ unsigned short perm[BLOCK / 2];
  if (frozen && *frozen != 0)
    return;
  const int tid = (int)threadIdx.x;
  body.lds_stage(nepmi_lds_pairs_s, tid, BLOCK);
  if (tid < 64)
    hist[tid] = 0;
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t base = (int64_t)tile * (BLOCK / 2);
  const int j = tid >> 1;
  const bool live = base + j < n;
  int key = 0;
  if (live) {
    key = body.sort_key(base + j);
    key = key > 63 ? 63 : key;
    if (tile & 1u)
      key = 63 - key; // (odd workgroups: ascending)
  }
  __syncthreads();
  if (live && (tid & 1) == 0)
    atomicAdd(&hist[key], 1);
  __syncthreads();
  if (tid == 0) { // first position of every key, longest lists first
    int run = 0;
    for (int c = 63; c >= 0; --c) {
      const int h = hist[c];
      hist[c] = run;
      run += h;
    }
  }
  __syncthreads();
  if (live && (tid & 1) == 0)
    perm[atomicAdd(&hist[key], 1)] = (unsigned short)j;
  __syncthreads();
  if (live) // (the live atoms fill perm[0 .. number of live atoms): the same count as the live lane pairs)
    body.template run_parts<2>(base + perm[j], tid & 1, (lds_cfloat_ptr)nepmi_lds_pairs_s);
}

// One-off: a body's LDS image written to global memory by the body's own staging code (nep_fused.h)
template <class Body>
__global__ void __launch_bounds__(256) nepmi_fused_image(const Body body, float* img)
{
  body.lds_stage(img, (int)threadIdx.x, 256);
}

// P adjacent lanes per atom (Body::run_parts<P>): TersoffPartialBody -- 54 atoms per CU need more than one wavefront per CU
template <int BLOCK, int P, class Body>
__global__ void __launch_bounds__(BLOCK) nepmi_kernel_lds_parts(const Body body, const int64_t n, const int* frozen)
{
  extern __shared__ __attribute__((aligned(16))) float nepmi_lds_parts[];
  if (frozen && *frozen != 0)
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i = ((int64_t)tile * BLOCK + threadIdx.x) / P;
  if (i < n)
    body.template run_parts<P>(i, (int)(threadIdx.x % P), (lds_cfloat_ptr)nepmi_lds_parts);
}

// ---- exclusive scan of int32, in place: 3 kernels (block scan, scan of block sums, add) ----
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_scan(int v, int* total);

// One 256-thread workgroup per brick (RadialWinBody, ForceWinBody; nep_window.h): stage the brick's
// 8x8x8-cell window in LDS (cell counts -> block scan -> fixed-point records), then one lane per atom of the
// brick.  The 16-byte records keep the window at ~30 KB, so four workgroups share a CU (16 wavefronts).
// XCD-aware brick order as in nepmi_kernel.
template <class Body>
__global__ void __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEu)))
nepmi_win_kernel(const Body body, const int64_t nbricks)
{
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.skip())
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wg >= nbricks)
    return; // the whole workgroup leaves before the first barrier
  const int64_t brick = body.map_brick(wg);
  const int tid = (int)threadIdx.x;
  body.stage_cells(brick, lds, tid, kWinThreads);
  __syncthreads();
  {
    static_assert(kWinThreads * 2 == kWinCells, "two window cells per thread");
    NEPMI_LDS(int)* woff = (NEPMI_LDS(int)*)lds;
    const int v0 = woff[2 * tid], v1 = woff[2 * tid + 1];
    int total;
    const int ex = block_exclusive_scan<kWinThreads>(v0 + v1, &total);
    woff[2 * tid] = ex;
    woff[2 * tid + 1] = ex + v0;
    if (tid == 0)
      woff[kWinCells] = total;
  }
  __syncthreads();
  body.stage_copy(brick, lds, tid, kWinThreads);
  __syncthreads();
  int64_t a0, a1;
  body.brick_range(brick, a0, a1);
  for (int64_t k = a0 + tid; k < a1; k += kWinThreads)
    body.compute(brick, k, lds);
}

// The window kernels on the static window layout (Bufs::wtab, RadialWin2Body / ForceWinBody::stage): the records are copied
// straight to their slots -- no cell-count scan, one barrier.
// NT = 512 (kWinThreadsBig): windows beyond half a CU's LDS (dense long-cutoff models, one workgroup per CU whatever its size, bricks
// of several passes): eight wavefronts per CU instead of four walk the brick's atoms.  The atoms keep their places in the wavefronts
// (64 consecutive atoms from the brick's first), so the wave-synchronous rows of one kernel are read in step by the other.
template <class Body, int NT = kWinThreads>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEu)))
nepmi_win2_kernel(const Body body, const int64_t nbricks)
{
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.skip())
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wg >= nbricks)
    return;
  const int64_t brick = body.map_brick(wg);
  const int tid = (int)threadIdx.x;
  body.stage(brick, lds, tid, NT);
  __syncthreads();
  int64_t a0, a1;
  body.brick_range(brick, a0, a1);
  for (int64_t k = a0 + tid; k < a1; k += NT)
    body.compute(brick, k, lds);
}

// Static layout with Body::kLanes lanes per atom and a second staging phase (ForceWinBody<..., ROWS>: the table rows of the
// window atoms behind the records): 256 kLanes threads, one workgroup per CU when the rows fill the LDS.
template <class Body>
__global__ void __launch_bounds__(kWinThreads * Body::kLanes) nepmi_win2_kernel_split(const Body body, const int64_t nbricks)
{
  constexpr int NT = kWinThreads * Body::kLanes;
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.skip())
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wg >= nbricks)
    return;
  const int64_t brick = body.map_brick(wg);
  const int tid = (int)threadIdx.x;
  body.stage(brick, lds, tid, NT);
  __syncthreads();
  body.stage_rows(brick, lds, tid, NT);
  __syncthreads();
  int64_t a0, a1;
  body.brick_range(brick, a0, a1);
  const int sub = tid % Body::kLanes;
  for (int64_t k = a0 + tid / Body::kLanes; k < a1; k += kWinThreads)
    body.compute(brick, k, lds, sub);
}

// The same with Body::kLanes = 2 or 4 adjacent lanes per atom (256 kLanes threads): systems with too few bricks to
// fill the chip, where a window kernel's run time is the latency of one workgroup (RadialWinSplitBody).
template <class Body>
__global__ void __launch_bounds__(kWinThreads * Body::kLanes) nepmi_win_kernel_split(const Body body, const int64_t nbricks)
{
  constexpr int NT = kWinThreads * Body::kLanes;
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.skip())
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wg >= nbricks)
    return;
  const int64_t brick = body.map_brick(wg);
  const int tid = (int)threadIdx.x;
  body.stage_cells(brick, lds, tid, NT);
  __syncthreads();
  {
    NEPMI_LDS(int)* woff = (NEPMI_LDS(int)*)lds;
    const bool scans = tid < kWinCells / 2;
    const int v0 = scans ? woff[2 * tid] : 0, v1 = scans ? woff[2 * tid + 1] : 0;
    int total;
    const int ex = block_exclusive_scan<NT>(v0 + v1, &total);
    if (scans) {
      woff[2 * tid] = ex;
      woff[2 * tid + 1] = ex + v0;
    }
    if (tid == 0)
      woff[kWinCells] = total;
  }
  __syncthreads();
  body.stage_copy(brick, lds, tid, NT);
  body.stage_lists(brick, lds, tid, NT);
  __syncthreads();
  int64_t a0, a1;
  body.brick_range(brick, a0, a1);
  const int sub = tid % Body::kLanes;
  for (int64_t k = a0 + tid / Body::kLanes; k < a1; k += kWinThreads)
    body.compute(brick, k, lds, sub);
}

// Few bricks: Body::kLanes workgroups of 256 threads per brick instead of one workgroup of 256 kLanes threads.  Each stages the
// window (the staging of nepmi_win_kernel) and takes 256 / kLanes atoms of the brick, kLanes lanes per atom: a 16,000-atom
// system has 64 bricks -- 64 workgroups of 16 wavefronts leave three quarters of the CUs idle, 256 workgroups of 4 do not, and a
// wavefront that has its SIMD to itself walks its list faster.
template <class Body>
__global__ void __launch_bounds__(kWinThreads) nepmi_win_kernel_parts(const Body body, const int64_t nbricks)
{
  constexpr int P = Body::kLanes;
  extern __shared__ __attribute__((aligned(16))) char nepmi_win_lds[];
  NEPMI_LDS(char)* lds = (NEPMI_LDS(char)*)nepmi_win_lds;
  if (body.skip())
    return;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t wg = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3); // the parts of a brick sit on one XCD
  const int64_t bw = wg / P;
  const int part = (int)(wg - bw * P);
  if (bw >= nbricks)
    return;
  const int64_t brick = body.map_brick(bw);
  const int tid = (int)threadIdx.x;
  body.stage_cells(brick, lds, tid, kWinThreads);
  __syncthreads();
  {
    NEPMI_LDS(int)* woff = (NEPMI_LDS(int)*)lds;
    const int v0 = woff[2 * tid], v1 = woff[2 * tid + 1];
    int total;
    const int ex = block_exclusive_scan<kWinThreads>(v0 + v1, &total);
    woff[2 * tid] = ex;
    woff[2 * tid + 1] = ex + v0;
    if (tid == 0)
      woff[kWinCells] = total;
  }
  __syncthreads();
  body.stage_copy(brick, lds, tid, kWinThreads);
  body.stage_lists(brick, lds, tid, kWinThreads);
  __syncthreads();
  int64_t a0, a1;
  body.brick_range(brick, a0, a1);
  const int sub = tid % P;
  for (int64_t k = a0 + part * (kWinThreads / P) + tid / P; k < a1; k += kWinThreads)
    body.compute(brick, k, lds, sub);
}

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off)
      v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_scan(int v, int* total)
{
  __shared__ int wave_sums[BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(v, lane);
  if (lane == 63)
    wave_sums[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) {
    const int s = wave_sums[w];
    if (w < wid)
      base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(kScanBlock) nepmi_scan_tiles(int* data, int64_t n, int* tile_sums)
{
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    sum += v[k];
  }
  int total;
  int run = block_exclusive_scan<kScanBlock>(sum, &total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n)
      data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0)
    tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) nepmi_scan_sums(int* sums, int64_t m)
{
  __shared__ int carry;
  if (threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for (int64_t start = 0; start < m; start += 1024) {
    const int64_t i = start + threadIdx.x;
    const int v = i < m ? sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan<1024>(v, &total);
    const int c = carry;
    if (i < m)
      sums[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0)
      carry = c + total;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kScanBlock) nepmi_scan_add(int* data, int64_t n, const int* tile_sums)
{
  const int add = tile_sums[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n)
      data[base + k] += add;
}

// ---- Langevin thermostat (Ensemble_LAN, src/integrate/ensemble_lan.cu:30-41, :96-127; kernels of
//      src/integrate/langevin_utilities.cuh:24-124).  One XORWOW state per atom, hiprand_init(seed, n, 0, ...) like the
//      reference (its gpurand_* are hiprand_* in the HIP build), v <- c1 v + c2 sqrt(1/m) xi with three
//      hiprand_normal_double draws per atom, then the centre-of-mass velocity is removed.  The four sums (m vx, m vy,
//      m vz, m) are formed by four 1024-thread blocks in the same order as gpu_find_momentum, so the corrected
//      velocities equal the reference's bit for bit (tests/test_langevin.py pins them on its kernels). ----

// The arithmetic of one kick / one momentum term / one correction as device functions shared by the stepwise kernels (pinned bit
// for bit on the reference's own kernels, tests/test_langevin.py) and the resident ones: the same expressions reach the compiler,
// so both forms take the same fused multiply-adds and round alike.
__device__ __forceinline__ void nepmi_lan_kick3(hiprandState& state, const double c1, const double c2, const double mass, double& vx,
                                                double& vy, double& vz)
{
  const double c2m = c2 * sqrt(1.0 / mass);
  vx = c1 * vx + c2m * hiprand_normal_double(&state);
  vy = c1 * vy + c2m * hiprand_normal_double(&state);
  vz = c1 * vz + c2m * hiprand_normal_double(&state);
}
__device__ __forceinline__ double nepmi_momentum_term(const double acc, const double m, const double v) { return acc + m * v; }
__device__ __forceinline__ double nepmi_momentum_fixed(const double v, const double sum, const double inverse_of_total_mass)
{
  return v - sum * inverse_of_total_mass;
}

__global__ void nepmi_lan_init(hiprandState* state, const int64_t N, const int seed)
{
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N)
    hiprand_init(seed, n, 0, &state[n]);
}

__global__ void nepmi_lan_kick(
  hiprandState* g_state, const int64_t N, const double c1, const double c2, const double* __restrict__ g_mass, double* g_v)
{
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    hiprandState state = g_state[n];
    double vx = g_v[n], vy = g_v[N + n], vz = g_v[2 * N + n];
    nepmi_lan_kick3(state, c1, c2, g_mass[n], vx, vy, vz);
    g_v[n] = vx;
    g_v[N + n] = vy;
    g_v[2 * N + n] = vz;
    g_state[n] = state;
  }
}

// block b < 3: sum_n m v_b, block 3: sum_n m; thread t takes atoms t, t + 1024, ... then a binary tree over the threads
__global__ void __launch_bounds__(1024) nepmi_momentum_sum(
  const int64_t N, const double* __restrict__ g_mass, const double* __restrict__ g_v, double* __restrict__ sums4)
{
  __shared__ double s_sum[1024];
  const int tid = threadIdx.x, bid = blockIdx.x;
  double acc = 0.0;
  if (bid < 3) { // the product and the sum contract to one fma, as in the reference's build of gpu_find_momentum
    const double* __restrict__ v = g_v + (int64_t)bid * N;
    for (int64_t n = tid; n < N; n += 1024)
      acc = nepmi_momentum_term(acc, g_mass[n], v[n]);
  } else {
    for (int64_t n = tid; n < N; n += 1024)
      acc += g_mass[n];
  }
  s_sum[tid] = acc;
  __syncthreads();
  for (int offset = 512; offset > 0; offset >>= 1) {
    if (tid < offset)
      s_sum[tid] += s_sum[tid + offset];
    __syncthreads();
  }
  if (tid == 0)
    sums4[bid] = s_sum[0];
}

__global__ void nepmi_momentum_fix(const int64_t N, const double* __restrict__ sums4, double* g_v)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const double inverse_of_total_mass = 1.0 / sums4[3];
    g_v[i] = nepmi_momentum_fixed(g_v[i], sums4[0], inverse_of_total_mass);
    g_v[N + i] = nepmi_momentum_fixed(g_v[N + i], sums4[1], inverse_of_total_mass);
    g_v[2 * N + i] = nepmi_momentum_fixed(g_v[2 * N + i], sums4[2], inverse_of_total_mass);
  }
}

// ---- the same thermostat inside the device-resident run loops (and the decomposed driver): velocities and masses in the
//      engine's INTERNAL order (stride n), generator states where the reference keeps them -- state s belongs to the atom with
//      caller index s = perm[k] (single domain: the caller's atom; decomposed: the rank's local atom, whose state was created
//      from its global id and migrates with it).  `flags`: the frozen word of the speculative loops. ----
__global__ void nepmi_lan_kick_resident(
  hiprandState* g_state, const int64_t n, const double c1, const double c2, const double* __restrict__ mi, double* vi,
  const int* __restrict__ perm, const signed char* __restrict__ lvl, const int64_t* __restrict__ ids, const int* flags)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n || flags[nepmi::kFlagMoved] != 0 || lvl[k] < 2)
    return;
  const int64_t s = ids ? ids[perm[k]] : (int64_t)perm[k];
  hiprandState state = g_state[s];
  double vx = vi[k], vy = vi[n + k], vz = vi[2 * n + k];
  nepmi_lan_kick3(state, c1, c2, mi[k], vx, vy, vz);
  vi[k] = vx;
  vi[n + k] = vy;
  vi[2 * n + k] = vz;
  g_state[s] = state;
}

// decomposed runs: state q of a rank belongs to its owned atom q (local order) and is the single-domain run's state of that
// atom's GLOBAL id (subsequence id of the XORWOW generator); it migrates with the atom (dist_impl.h)
__global__ void nepmi_lan_init_ids(hiprandState* state, const int64_t n, const int64_t* __restrict__ ids, const int seed)
{
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
    hiprand_init(seed, ids[q], 0, &state[q]);
}

// invp != nullptr (single domain): the four sums in CALLER order exactly as nepmi_momentum_sum forms them (thread t takes
// the atoms t, t + 1024, ... of the caller's numbering), so the corrected velocities stay bit-identical to the stepwise form;
// invp == nullptr (decomposed): the owned atoms in internal order (a fixed order per rank; the ranks' sums are all-reduced)
__global__ void __launch_bounds__(1024) nepmi_momentum_sum_resident(
  const int64_t n, const double* __restrict__ mi, const double* __restrict__ vi, const int* __restrict__ invp,
  const signed char* __restrict__ lvl, double* __restrict__ sums4, const int* flags)
{
  __shared__ double s_sum[1024];
  if (flags[nepmi::kFlagMoved] != 0)
    return;
  const int tid = threadIdx.x, bid = blockIdx.x;
  double acc = 0.0;
  for (int64_t q = tid; q < n; q += 1024) {
    const int64_t k = invp ? invp[q] : q;
    if (!invp && lvl[k] < 2)
      continue;
    if (bid < 3)
      acc = nepmi_momentum_term(acc, mi[k], vi[(int64_t)bid * n + k]);
    else
      acc += mi[k];
  }
  s_sum[tid] = acc;
  __syncthreads();
  for (int offset = 512; offset > 0; offset >>= 1) {
    if (tid < offset)
      s_sum[tid] += s_sum[tid + offset];
    __syncthreads();
  }
  if (tid == 0)
    sums4[bid] = s_sum[0];
}

__global__ void nepmi_momentum_fix_resident(const int64_t n, const double* __restrict__ sums4, double* vi,
                                            const signed char* __restrict__ lvl, const int* flags)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n || flags[nepmi::kFlagMoved] != 0 || lvl[k] < 2)
    return;
  const double inverse_of_total_mass = 1.0 / sums4[3];
  vi[k] = nepmi_momentum_fixed(vi[k], sums4[0], inverse_of_total_mass);
  vi[n + k] = nepmi_momentum_fixed(vi[n + k], sums4[1], inverse_of_total_mass);
  vi[2 * n + k] = nepmi_momentum_fixed(vi[2 * n + k], sums4[2], inverse_of_total_mass);
}

// ---- Ensemble::find_thermo (ensemble.cu:434-673): 8 sums in one pass over the atoms ----
constexpr int kThermoBlock = 256;
constexpr int kThermoMaxBlocks = 1024;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(kThermoBlock) nepmi_thermo_partial(
  int64_t n, const double* __restrict__ mass, const double* __restrict__ pe, const double* __restrict__ vel,
  const double* __restrict__ virial, const signed char* __restrict__ lvl, double* __restrict__ partial, int virial_lvl)
{
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kThermoBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThermoBlock) {
    if (lvl && lvl[i] < 2) { // domain decomposition: ghosts carry no state ...
      if (lvl[i] >= virial_lvl) // ... but, with reverse-mode ghosts, the virial of the pair halves computed on them
        for (int q = 0; q < 6; ++q)
          s[2 + q] += virial[(int64_t)q * n + i];
      continue;
    }
    const double m = mass[i];
    const double vx = vel[i], vy = vel[n + i], vz = vel[2 * n + i];
    s[0] += (vx * vx + vy * vy + vz * vz) * m;
    s[1] += pe[i];
    s[2] += virial[i] + vx * vx * m;
    s[3] += virial[n + i] + vy * vy * m;
    s[4] += virial[2 * n + i] + vz * vz * m;
    s[5] += virial[3 * n + i] + vx * vy * m;
    s[6] += virial[4 * n + i] + vx * vz * m;
    s[7] += virial[5 * n + i] + vy * vz * m;
  }
  __shared__ double red[kThermoBlock / 64][8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double w = wave_sum(s[q]);
    if (lane == 0)
      red[wid][q] = w;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kThermoBlock / 64; ++w)
      t += red[w][threadIdx.x];
    partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = t;
  }
}

// raw != 0: the eight sums as they are (a decomposed run all-reduces them before normalising)
__global__ void __launch_bounds__(64) nepmi_thermo_final(
  int nblocks, int64_t n, double volume, const double* __restrict__ partial, double* __restrict__ thermo8, int raw)
{
  // 8 quantities x 8 lanes each; fixed order => deterministic
  const int q = threadIdx.x >> 3, sub = threadIdx.x & 7;
  double t = 0.0;
  for (int b = sub; b < nblocks; b += 8)
    t += partial[(int64_t)b * 8 + q];
  t += __shfl_down(t, 4, 8);
  t += __shfl_down(t, 2, 8);
  t += __shfl_down(t, 1, 8);
  if (sub == 0) {
    if (raw)
      thermo8[q] = t;
    else if (q == 0)
      thermo8[0] = t / (3.0 * (double)n * 8.617343e-5); // K_B, src/utilities/common.cuh:22
    else if (q == 1)
      thermo8[1] = t;
    else
      thermo8[q] = t / volume;
  }
}

// ------------------------------------------------------------------------------------------------

struct EventTimer {
  static constexpr int kPool = 128;
  hipEvent_t start[kPool], stop[kPool];
  int used = 0;
  bool created = false;
  double sum_ms = 0.0, last_ms = 0.0;
  int64_t count = 0;
};

struct HipTiming {
  EventTimer slot[16];
  EventTimer reg[4];
};

struct HipBackend {
  hipStream_t stream = nullptr;
  HipTiming* timing = nullptr; // shared by copies of the backend
  int* pinned = nullptr;       // 64-byte pinned staging for flag read-back
  bool timing_on = false; // any timing
  int timing_mode = 0;    // 1: every kernel and region; 2: the force-assembly kernel only (two events per step)
  int timing_slot = kSlotForce; // mode 2: the one slot that carries events (set_timing(16 + slot) chooses it)
  bool timed(int slot) const { return timing_mode == 1 ? slot != kSlotMisc : (timing_mode == 2 && slot == timing_slot); }
  hipEvent_t probe_ev[2] = {nullptr, nullptr};
  bool mfma_on = true; // per-atom ANN on the matrix cores when the model shape allows it
  // device word the launches of the force path look at first (null: always run); set by the engine around the force
  // path of a speculatively enqueued step
  const int* frozen = nullptr;
  // flag snapshots of the speculative run loops: pinned slots + events
  static constexpr int kPollRing = 8;
  int* poll_pinned = nullptr; // [kPollRing][8]
  hipEvent_t poll_ev[kPollRing] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void poll_record(int ring, const int* dev_flags)
  {
    if (!poll_pinned) {
      void* p = nullptr;
      NEPMI_HIP_CHECK(hipHostMalloc(&p, sizeof(int) * 8 * kPollRing, hipHostMallocDefault));
      poll_pinned = (int*)p;
      for (int k = 0; k < kPollRing; ++k)
        NEPMI_HIP_CHECK(hipEventCreateWithFlags(&poll_ev[k], hipEventDisableTiming));
    }
    NEPMI_HIP_CHECK(hipMemcpyAsync(poll_pinned + 8 * ring, dev_flags, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
    NEPMI_HIP_CHECK(hipEventRecord(poll_ev[ring], stream));
  }
  void poll_wait(int ring, int* out8)
  {
    NEPMI_HIP_CHECK(hipEventSynchronize(poll_ev[ring]));
    std::memcpy(out8, poll_pinned + 8 * ring, sizeof(int) * 8);
  }

  void* alloc(size_t bytes)
  {
    void* p = nullptr;
    NEPMI_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
    // NEPMI_POISON=<byte>: debugging aid -- fresh device memory is filled with that byte (e.g. 255: floats are NaN), so that a
    // kernel which reads what nobody wrote fails every time instead of depending on what the block held before
    static const int poison = std::getenv("NEPMI_POISON") ? std::atoi(std::getenv("NEPMI_POISON")) : -1;
    if (poison >= 0 && bytes) {
      NEPMI_HIP_CHECK(hipMemset(p, poison & 0xFF, bytes));
      NEPMI_HIP_CHECK(hipDeviceSynchronize());
    }
    return p;
  }
  void free(void* p) { (void)hipFree(p); }
  void memset(void* p, int v, size_t bytes) { NEPMI_HIP_CHECK(hipMemsetAsync(p, v, bytes, stream)); }
  void h2d(void* dst, const void* src, size_t bytes)
  {
    NEPMI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    NEPMI_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void d2h(void* dst, const void* src, size_t bytes)
  {
    if (bytes <= 64 && pinned) {
      NEPMI_HIP_CHECK(hipMemcpyAsync(pinned, src, bytes, hipMemcpyDeviceToHost, stream));
      NEPMI_HIP_CHECK(hipStreamSynchronize(stream));
      std::memcpy(dst, pinned, bytes);
    } else {
      NEPMI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
      NEPMI_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  void d2d(void* dst, const void* src, size_t bytes)
  {
    NEPMI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
  }
  void* stream_handle() { return (void*)stream; }
  // a second stream for communication that overlaps compute (domain decomposition): same device, own events
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  HipBackend make_side_stream()
  {
    HipBackend s = *this;
    hipStream_t st = nullptr;
    NEPMI_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    s.stream = st;
    s.timing_on = false;
    s.timing_mode = 0;
    s.frozen = nullptr;
    s.fork_ev = s.join_ev = nullptr;
    return s;
  }
  void fork_to(HipBackend& side) // side continues after everything enqueued here so far
  {
    if (!fork_ev)
      NEPMI_HIP_CHECK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    NEPMI_HIP_CHECK(hipEventRecord(fork_ev, stream));
    NEPMI_HIP_CHECK(hipStreamWaitEvent(side.stream, fork_ev, 0));
  }
  void join_from(HipBackend& side) // this stream continues after everything enqueued on side so far
  {
    if (!join_ev)
      NEPMI_HIP_CHECK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    NEPMI_HIP_CHECK(hipEventRecord(join_ev, side.stream));
    NEPMI_HIP_CHECK(hipStreamWaitEvent(stream, join_ev, 0));
  }
  void sync() { NEPMI_HIP_CHECK(hipStreamSynchronize(stream)); }

  // ---- HIP-event timing on the engine's own stream (bench.py's roofline leg) ----
  void timer_drain(EventTimer& t)
  {
    if (t.used == 0)
      return;
    NEPMI_HIP_CHECK(hipEventSynchronize(t.stop[t.used - 1]));
    for (int k = 0; k < t.used; ++k) {
      float ms = 0.0f;
      NEPMI_HIP_CHECK(hipEventElapsedTime(&ms, t.start[k], t.stop[k]));
      t.sum_ms += ms;
      t.last_ms = ms;
      ++t.count;
    }
    t.used = 0;
  }
  void timer_start(EventTimer& t)
  {
    if (!t.created) {
      for (int k = 0; k < EventTimer::kPool; ++k) {
        NEPMI_HIP_CHECK(hipEventCreate(&t.start[k]));
        NEPMI_HIP_CHECK(hipEventCreate(&t.stop[k]));
      }
      t.created = true;
    }
    if (t.used == EventTimer::kPool)
      timer_drain(t);
    NEPMI_HIP_CHECK(hipEventRecord(t.start[t.used], stream));
  }
  void timer_stop(EventTimer& t)
  {
    NEPMI_HIP_CHECK(hipEventRecord(t.stop[t.used], stream));
    ++t.used;
  }
  void set_timing(int mode)
  {
    for (int k = 0; k < 16; ++k) {
      timer_drain(timing->slot[k]);
      timing->slot[k].sum_ms = 0.0;
      timing->slot[k].count = 0;
    }
    for (int k = 0; k < 4; ++k) {
      timer_drain(timing->reg[k]);
      timing->reg[k].sum_ms = 0.0;
      timing->reg[k].count = 0;
    }
    timing_slot = mode >= 16 ? mode - 16 : kSlotForce;
    timing_mode = mode >= 16 ? 2 : mode;
    timing_on = mode != 0;
  }
  // stand-alone stopwatch on the engine's stream (independent of set_timing): used once per engine
  // to choose between equivalent kernel variants
  void probe_start()
  {
    if (!probe_ev[0]) {
      NEPMI_HIP_CHECK(hipEventCreate(&probe_ev[0]));
      NEPMI_HIP_CHECK(hipEventCreate(&probe_ev[1]));
    }
    NEPMI_HIP_CHECK(hipEventRecord(probe_ev[0], stream));
  }
  double probe_stop_ms()
  {
    NEPMI_HIP_CHECK(hipEventRecord(probe_ev[1], stream));
    NEPMI_HIP_CHECK(hipEventSynchronize(probe_ev[1]));
    float ms = 0.0f;
    NEPMI_HIP_CHECK(hipEventElapsedTime(&ms, probe_ev[0], probe_ev[1]));
    return (double)ms;
  }
  void begin_region(int r)
  {
    if (timing_mode == 1)
      timer_start(timing->reg[r]);
  }
  void end_region(int r)
  {
    if (timing_mode == 1)
      timer_stop(timing->reg[r]);
  }
  double region_ms(int r) { timer_drain(timing->reg[r]); return timing->reg[r].last_ms; }
  double region_sum(int r) { timer_drain(timing->reg[r]); return timing->reg[r].sum_ms; }
  int64_t region_count(int r) { timer_drain(timing->reg[r]); return timing->reg[r].count; }
  double slot_ms(int s) { timer_drain(timing->slot[s]); return timing->slot[s].last_ms; }
  double slot_sum(int s) { timer_drain(timing->slot[s]); return timing->slot[s].sum_ms; }
  int64_t slot_count(int s) { timer_drain(timing->slot[s]); return timing->slot[s].count; }

  template <int BLOCK, class Body>
  void launch(int slot, int64_t n, const Body& body)
  {
    if (n <= 0)
      return;
    const int64_t grid = ((n + BLOCK - 1) / BLOCK + 7) / 8 * 8;
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_kernel<BLOCK, Body>), dim3((unsigned)grid), dim3(BLOCK), 0, stream, body, n, frozen);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  // ANN launch: the MFMA kernel when the shape fits (few types, <= 128 neurons, <= 128 output rows
  // incl. the radial-table rows), else the per-atom AnnBody.
  template <int MT, int QB, bool BYTYPE = false>
  void launch_ann_mfma(int DT, size_t lds_bytes, int64_t grid, const ModelD& m, const Bufs& b, int64_t nchunks)
  {
#define NEPMI_ANN_CASE(D)                                                                           \
  case D:                                                                                           \
    if (lds_bytes > 64 * 1024)                                                                      \
      NEPMI_HIP_CHECK(hipFuncSetAttribute(                                                          \
        reinterpret_cast<const void*>(&nepmi_ann_mfma<MT, D, QB, BYTYPE>), hipFuncAttributeMaxDynamicSharedMemorySize, \
        (int)lds_bytes));                                                                           \
    hipLaunchKernelGGL((nepmi_ann_mfma<MT, D, QB, BYTYPE>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, m, b, nchunks, frozen); \
    break;
    switch (DT) {
      NEPMI_ANN_CASE(1)
      NEPMI_ANN_CASE(2)
      NEPMI_ANN_CASE(3)
      NEPMI_ANN_CASE(4)
    }
#undef NEPMI_ANN_CASE
  }

  void ann_prepare(const ModelD& m, const Bufs& b)
  {
    if (!b.ann_img)
      return;
    const AnnMfmaShape a = ann_mfma_shape(m.T, m.dim, m.nneu, b.KRP);
    hipLaunchKernelGGL(nepmi_ann_pack, dim3((unsigned)m.T), dim3(256), 0, stream, m, b, a.MT, a.DT);
    NEPMI_HIP_CHECK(hipGetLastError());
  }

  template <class S>
  void launch_ann(int slot, int64_t n, const ModelD& m, const Bufs& b, bool grouped)
  {
    // (more than four types: the image carries no radial-table rows -- the matrix-core kernel serves the steps whose force
    // assembly contracts from Bufs::fpr, the per-atom kernel the others)
    if (!mfma_on || !b.ann_img || !grouped || (m.T > 4 && !b.skip_atab)) {
      launch<64>(slot, n, AnnBody<S>{m, b});
      return;
    }
    if (n <= 0)
      return;
    const AnnMfmaShape a = ann_mfma_shape(m.T, m.dim, m.nneu, b.KRP);
    const bool bytype = m.T > 4;
    const size_t lds_bytes = a.img_floats * sizeof(float) + (bytype ? (2 * kAnnGroup + 2) * sizeof(int) : 0);
    const int64_t nchunks = (n >> kTypeChunkShift) + 1;
    const int64_t grid = bytype ? ((nchunks + kAnnGroup - 1) / kAnnGroup * m.T + 7) / 8 * 8 : (nchunks * kAnnSplit + 7) / 8 * 8;
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    if (bytype) {
      switch (a.MT) {
        case 1: launch_ann_mfma<1, 24, true>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 2: launch_ann_mfma<2, 24, true>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 3: launch_ann_mfma<3, 24, true>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        default: launch_ann_mfma<4, 24, true>(a.DT, lds_bytes, grid, m, b, nchunks); break;
      }
    } else if (a.KS <= 24) {
      switch (a.MT) {
        case 1: launch_ann_mfma<1, 24>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 2: launch_ann_mfma<2, 24>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 3: launch_ann_mfma<3, 24>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        default: launch_ann_mfma<4, 24>(a.DT, lds_bytes, grid, m, b, nchunks); break;
      }
    } else if (a.KS <= 34 && a.MT == 4) {
      // C_2022_NEP4 (dim 65: 33 k-pairs, 100 neurons): the operand buffer of 40 k-pairs pushed the kernel ten registers past
      // the 256 of two wavefronts per SIMD (r3a: scratch 44 bytes per lane)
      launch_ann_mfma<4, 34>(a.DT, lds_bytes, grid, m, b, nchunks);
    } else {
      switch (a.MT) {
        case 1: launch_ann_mfma<1, 40>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 2: launch_ann_mfma<2, 40>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        case 3: launch_ann_mfma<3, 40>(a.DT, lds_bytes, grid, m, b, nchunks); break;
        default: launch_ann_mfma<4, 40>(a.DT, lds_bytes, grid, m, b, nchunks); break;
      }
    }
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }
  void set_mfma(bool on) { mfma_on = on; }
  void adopt_options(const HipBackend& o) // a replacement engine keeps the switches of the one it replaces
  {
    timing_on = o.timing_on;
    timing_mode = o.timing_mode;
    mfma_on = o.mfma_on;
  }

  template <class Body>
  void launch_win(int slot, int64_t nbricks, const Body& body)
  {
    if (nbricks <= 0)
      return;
    const int64_t grid = (nbricks + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_bytes() + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&nepmi_win_kernel<Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_win_kernel<Body>), dim3((unsigned)grid), dim3(kWinThreads), lds_bytes, stream, body, nbricks);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  template <class Body>
  void launch_win2(int slot, int64_t nbricks, const Body& body)
  {
    if (nbricks <= 0)
      return;
    const int64_t grid = (nbricks + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_bytes() + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&nepmi_win2_kernel<Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    if constexpr (Body::kBigWindows) {
      if (lds_bytes > kBigWindowLds) {
        NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_win2_kernel<Body, kWinThreadsBig>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((nepmi_win2_kernel<Body, kWinThreadsBig>), dim3((unsigned)grid), dim3(kWinThreadsBig), lds_bytes, stream, body, nbricks);
        NEPMI_HIP_CHECK(hipGetLastError());
        if (t)
          timer_stop(timing->slot[slot]);
        return;
      }
    }
    if constexpr (Body::kMidWindows) {
      // two workgroups per CU (many-type models: the window and the coefficient table, 79 KB for UNEP-v1): 512 threads each
      if (lds_bytes > kMidWindowLds) {
        NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_win2_kernel<Body, kWinThreadsMid>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((nepmi_win2_kernel<Body, kWinThreadsMid>), dim3((unsigned)grid), dim3(kWinThreadsMid), lds_bytes, stream, body, nbricks);
        NEPMI_HIP_CHECK(hipGetLastError());
        if (t)
          timer_stop(timing->slot[slot]);
        return;
      }
    }
    hipLaunchKernelGGL((nepmi_win2_kernel<Body>), dim3((unsigned)grid), dim3(kWinThreads), lds_bytes, stream, body, nbricks);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  template <class Body>
  void launch_win2_split(int slot, int64_t nbricks, const Body& body)
  {
    if (nbricks <= 0)
      return;
    const int64_t grid = (nbricks + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_bytes() + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&nepmi_win2_kernel_split<Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_win2_kernel_split<Body>), dim3((unsigned)grid), dim3(kWinThreads * Body::kLanes), lds_bytes,
                       stream, body, nbricks);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }
  static constexpr size_t kMaxLdsBytes = 160 * 1024; // per workgroup (one per CU)

  // Force assembly in the scatter form (nep_scatter.h): one workgroup per brick scatters the own pair halves into its LDS
  // window accumulator and writes it to its row of hacc; ForceFoldBody then adds every atom's entries.  One timing bracket
  // around both launches: together they are the force assembly.
  static constexpr bool kHasScatter = true;
  static constexpr bool kHasFusedAngular = true; // nep_fused.h
  // nb bricks from brick_order[first ...] (first < 0: all bricks in their own order); then the fold of the atoms with level in
  // [fold_lo, fold_hi] (fold_lo > fold_hi: no fold)
  template <class S>
  void launch_force_scatter(int slot, int64_t nb, int first, int64_t natoms, const WinStage& ws2, const ModelD& md, int* halo,
                            const unsigned* fmap, int fold_rows, bool outputs, int mode, int fold_lo, int fold_hi, const int* frz)
  {
    const ForceScatterBody<S> body{ws2, md, frz, reinterpret_cast<I4*>(halo), first};
    const int64_t grid = (nb + 7) / 8 * 8;
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    if (nb > 0) {
      if constexpr (S::TS > 0) {
        const ScatterLayout lay{ws2.lay.wmax};
        const size_t lds_bytes = ((size_t)lay.bytes() + 15) / 16 * 16;
#define NEPMI_FS_LAUNCH(OUTV, MODEV)                                                                                             \
  do {                                                                                                                           \
    if (lds_bytes > kBigWindowLds) { /* one workgroup per CU whatever its size: eight wavefronts (nepmi_win2_kernel) */           \
      NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_force_scatter_kernel<S, OUTV, MODEV, kWinThreadsBigScatter>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                         \
      hipLaunchKernelGGL((nepmi_force_scatter_kernel<S, OUTV, MODEV, kWinThreadsBigScatter>), dim3((unsigned)grid), dim3(kWinThreadsBigScatter), lds_bytes, \
                         stream, body, nb);                                                                                     \
      break;                                                                                                                     \
    }                                                                                                                            \
    if (lds_bytes > 64 * 1024)                                                                                                   \
      NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_force_scatter_kernel<S, OUTV, MODEV>),           \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                         \
    hipLaunchKernelGGL((nepmi_force_scatter_kernel<S, OUTV, MODEV>), dim3((unsigned)grid), dim3(kWinThreads), lds_bytes, stream, \
                       body, nb);                                                                                               \
  } while (0)
        if (outputs && mode == 2)
          NEPMI_FS_LAUNCH(true, 2);
        else if (outputs && mode == 1)
          NEPMI_FS_LAUNCH(true, 1);
        else if (outputs)
          NEPMI_FS_LAUNCH(true, 0);
        else if (mode == 2)
          NEPMI_FS_LAUNCH(false, 2);
        else if (mode == 1)
          NEPMI_FS_LAUNCH(false, 1);
        else
          NEPMI_FS_LAUNCH(false, 0);
#undef NEPMI_FS_LAUNCH
      } else {
        // many types / run-time shapes: nepmi_force_scatter_mt_kernel, four lanes per atom, one workgroup per CU
#ifndef NEPMI_FS_MT_LANES
#define NEPMI_FS_MT_LANES 4
#endif
        constexpr int L = NEPMI_FS_MT_LANES;
        const ScatterLayoutMT lay{ws2.lay.wmax, md.T * md.T * ctab_block(md.NR, md.KR, NEPMI_FS_MT_VEC != 0)};
        const size_t lds_bytes = ((size_t)lay.bytes() + 15) / 16 * 16;
#define NEPMI_FSMT_LAUNCH(OUTV, MODEV)                                                                                                  \
  do {                                                                                                                                 \
    if (lds_bytes > 64 * 1024)                                                                                                         \
      NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_force_scatter_mt_kernel<S, OUTV, L, MODEV>),           \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                               \
    hipLaunchKernelGGL((nepmi_force_scatter_mt_kernel<S, OUTV, L, MODEV>), dim3((unsigned)grid), dim3(kWinThreads * L), lds_bytes,      \
                       stream, body, nb);                                                                                             \
  } while (0)
        if (outputs && mode == 2)
          NEPMI_FSMT_LAUNCH(true, 2);
        else if (outputs)
          NEPMI_FSMT_LAUNCH(true, 0);
        else if (mode == 2)
          NEPMI_FSMT_LAUNCH(false, 2);
        else
          NEPMI_FSMT_LAUNCH(false, 0);
#undef NEPMI_FSMT_LAUNCH
      }
      NEPMI_HIP_CHECK(hipGetLastError());
    }
    if (fold_lo <= fold_hi && natoms > 0) {
      const ForceFoldBody fold{ws2.b, md, ws2.lay.wmax, fold_rows, fmap, reinterpret_cast<const I4*>(halo), fold_lo, fold_hi};
      const int64_t fgrid = ((natoms + 255) / 256 + 7) / 8 * 8;
      hipLaunchKernelGGL((nepmi_kernel<256, ForceFoldBody>), dim3((unsigned)fgrid), dim3(256), 0, stream, fold, natoms, frz);
      NEPMI_HIP_CHECK(hipGetLastError());
    }
    if (t)
      timer_stop(timing->slot[slot]);
  }
  // the windows that hold each atom (FoldMapBody), once per list rebuild; returns the largest number of windows met
  int build_fold_map(int64_t natoms, const BoxD& box, const Bufs& b, int wmax, int rows, unsigned* fmap)
  {
    int* word = b.flags + kFlagFoldRows;
    NEPMI_HIP_CHECK(hipMemsetAsync(word, 0, sizeof(int), stream));
    const FoldMapBody body{box, b, wmax, rows, fmap, word};
    const int64_t grid = ((natoms + 255) / 256 + 7) / 8 * 8;
    hipLaunchKernelGGL((nepmi_kernel<256, FoldMapBody>), dim3((unsigned)grid), dim3(256), 0, stream, body, natoms, nullptr);
    NEPMI_HIP_CHECK(hipGetLastError());
    int most = 0;
    d2h(&most, word, sizeof(int));
    return most;
  }

  static constexpr bool kSplitLanes = true; // window kernels with several lanes per atom exist on this backend
  template <class Body>
  void launch_win_split(int slot, int64_t nbricks, const Body& body)
  {
    if (nbricks <= 0)
      return;
    const int64_t grid = (nbricks + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_bytes() + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&nepmi_win_kernel_split<Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    // kLanes workgroups of 256 threads per brick (nepmi_win_kernel_parts) while they fit two to a CU: a counted rule
    // (profiles/r3x: PbTe 16,000 atoms 0.183 -> 0.155 ms/step, 31,250 atoms 0.166 -> 0.162, 54,000 atoms 0.178 -> 0.193)
    if (win_parts && nbricks * Body::kLanes <= 512) {
      const int64_t pgrid = (nbricks * Body::kLanes + 7) / 8 * 8;
      if (lds_bytes > 64 * 1024)
        NEPMI_HIP_CHECK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&nepmi_win_kernel_parts<Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
          (int)lds_bytes));
      hipLaunchKernelGGL((nepmi_win_kernel_parts<Body>), dim3((unsigned)pgrid), dim3(kWinThreads), lds_bytes, stream, body,
                         nbricks);
    } else {
      hipLaunchKernelGGL((nepmi_win_kernel_split<Body>), dim3((unsigned)grid), dim3(kWinThreads * Body::kLanes), lds_bytes,
                         stream, body, nbricks);
    }
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }
  // A/B switch of launch_win_split (NEPMI_WIN_PARTS=0: one workgroup of 256 kLanes threads per brick)
  bool win_parts = std::getenv("NEPMI_WIN_PARTS") == nullptr || std::getenv("NEPMI_WIN_PARTS")[0] != '0';

  template <int BLOCK, class Body>
  void launch_lds(int slot, int64_t n, const Body& body)
  {
    if (n <= 0)
      return;
    const int64_t grid = ((n + BLOCK - 1) / BLOCK + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_floats() * sizeof(float) + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&nepmi_kernel_lds<BLOCK, Body>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_kernel_lds<BLOCK, Body>), dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, body, n, frozen);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  template <int BLOCK, int P, class Body>
  void launch_lds_parts(int slot, int64_t n, const Body& body)
  {
    if (n <= 0)
      return;
    const int64_t grid = ((P * n + BLOCK - 1) / BLOCK + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_floats() * sizeof(float) + 15) / 16 * 16;
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_kernel_lds_parts<BLOCK, P, Body>), dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, body, n,
                       frozen);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  template <int BLOCK, class Body, bool SORTED = false>
  void launch_lds_pairs(int slot, int64_t n, const Body& body)
  {
    if (n <= 0)
      return;
    const int64_t grid = ((2 * n + BLOCK - 1) / BLOCK + 7) / 8 * 8;
    const size_t lds_bytes = ((size_t)body.lds_floats() * sizeof(float) + 15) / 16 * 16;
    void (*kern)(const Body, const int64_t, const int*) = nullptr;
    if constexpr (SORTED)
      kern = &nepmi_kernel_lds_pairs_sorted<BLOCK, Body>;
    else
      kern = &nepmi_kernel_lds_pairs<BLOCK, Body>;
    if (lds_bytes > 60 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, body, n, frozen);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
  }

  // angular descriptor + ANN + partial angular forces, two lanes per atom (nep_fused.h).  The kernel's LDS image lives in
  // global memory (`img`, owned by the engine; written here when `build` is set) and is copied by every workgroup.
  template <class S>
  size_t fused_image_floats(const ModelD& md) const
  {
    return (size_t)fused_lds_layout<S>(md).total;
  }
  template <class S>
  void launch_angular_fused(int slot, int64_t n, const ModelD& md, const Bufs& b, int export_qfp, float* img, bool build)
  {
    AngularFusedBody<S> body{md, b, export_qfp, nullptr};
    if (build) {
      hipLaunchKernelGGL((nepmi_fused_image<AngularFusedBody<S>>), dim3(1), dim3(256), 0, stream, body, img);
      NEPMI_HIP_CHECK(hipGetLastError());
    }
    body.img = img;
#ifndef NEPMI_AFU_BLOCK
#define NEPMI_AFU_BLOCK 256 // A/B switch: threads per workgroup (half as many atoms)
#endif
#ifndef NEPMI_AFU_SORT
#define NEPMI_AFU_SORT 0 // 1: nepmi_kernel_lds_pairs_sorted.  Measured (profiles/r6q_ab_sort.txt, same box): PbTe 0.458 against 0.450 ms, carbon 2.326
                         // against 2.290 -- the lockstep waiting it removes is not what the kernel's time is made of (vector issue of the heaviest
                         // wavefront of a workgroup, which the sort makes no lighter).  Off.
#endif
    launch_lds_pairs<NEPMI_AFU_BLOCK, AngularFusedBody<S>, NEPMI_AFU_SORT != 0>(slot, n, body);
  }

  // ... for many-type models: type-sorted work order, a window of kFusedWindowTypes types in LDS (nepmi_fused_window_kernel)
  template <class S>
  size_t fused_window_lds_bytes(const ModelD& md) const
  {
    return (size_t)fused_lds_layout<S>(md, kFusedWindowTypes).total * sizeof(float);
  }
  static constexpr bool kHasFusedWindow = NEPMI_WITH_FUSED_WINDOW != 0;
  template <class S>
  void launch_angular_fused_window(int slot, int64_t n, const ModelD& md, const Bufs& b, int export_qfp, float* img, bool build)
  {
#if NEPMI_WITH_FUSED_WINDOW
    if (n <= 0)
      return;
    AngularFusedBody<S> body{md, b, export_qfp, nullptr};
    if (build) { // the image of ALL types (tw = 0), once
      hipLaunchKernelGGL((nepmi_fused_image<AngularFusedBody<S>>), dim3(1), dim3(256), 0, stream, body, img);
      NEPMI_HIP_CHECK(hipGetLastError());
    }
    body.img = img;
    body.tw = kFusedWindowTypes;
    const int64_t grid = ((n + 127) / 128 + 7) / 8 * 8;
    const size_t lds_bytes = (fused_window_lds_bytes<S>(md) + 15) / 16 * 16;
    if (lds_bytes > 64 * 1024)
      NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_fused_window_kernel<AngularFusedBody<S>>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const bool t = timed(slot);
    if (t)
      timer_start(timing->slot[slot]);
    hipLaunchKernelGGL((nepmi_fused_window_kernel<AngularFusedBody<S>>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, body, n, frozen);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (t)
      timer_stop(timing->slot[slot]);
#else
    (void)slot; (void)n; (void)md; (void)b; (void)export_qfp; (void)img; (void)build;
#endif
  }

  // One force kernel per brick behind the radial pass (nep_brick.h) + the fold.  Two timing brackets: the brick kernel in the
  // angular slot (it replaces the angular kernel and the scatter kernel), the fold in the force-assembly slot.
#if NEPMI_WITH_BRICK
  static constexpr bool kHasBrickForce = true;
  template <class S>
  size_t brick_lds_bytes(const ModelD& md, int wmax) const
  {
    return (size_t)BrickLayout{wmax, fused_lds_layout<S>(md).total}.bytes();
  }
  template <class S>
  void launch_brick_force(int slot_brick, int slot_fold, int64_t nb, int64_t natoms, const WinStage& ws2, const ModelD& md, int* halo,
                          const unsigned* fmap, int fold_rows, bool outputs, const float* img, const int* frz)
  {
    if constexpr (S::fixed && S::TS == 2) {
      const BrickForceBody<S> body{ForceScatterBody<S>{ws2, md, frz, reinterpret_cast<I4*>(halo), -1},
                                   AngularFusedBody<S>{md, ws2.b, 0, img}};
      const int64_t grid = (nb + 7) / 8 * 8;
      const size_t lds_bytes = (brick_lds_bytes<S>(md, ws2.lay.wmax) + 15) / 16 * 16;
      bool t = timed(slot_brick);
      if (t)
        timer_start(timing->slot[slot_brick]);
      if (nb > 0) {
        if (outputs) {
          if (lds_bytes > 64 * 1024)
            NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_brick_force_kernel<S, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
          hipLaunchKernelGGL((nepmi_brick_force_kernel<S, true>), dim3((unsigned)grid), dim3(kBrickThreads), lds_bytes, stream, body, nb);
        } else {
          if (lds_bytes > 64 * 1024)
            NEPMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nepmi_brick_force_kernel<S, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
          hipLaunchKernelGGL((nepmi_brick_force_kernel<S, false>), dim3((unsigned)grid), dim3(kBrickThreads), lds_bytes, stream, body, nb);
        }
        NEPMI_HIP_CHECK(hipGetLastError());
      }
      if (t)
        timer_stop(timing->slot[slot_brick]);
      t = timed(slot_fold);
      if (t)
        timer_start(timing->slot[slot_fold]);
      if (natoms > 0) {
        const ForceFoldBody fold{ws2.b, md, ws2.lay.wmax, fold_rows, fmap, reinterpret_cast<const I4*>(halo), 0, 2};
        const int64_t fgrid = ((natoms + 255) / 256 + 7) / 8 * 8;
        hipLaunchKernelGGL((nepmi_kernel<256, ForceFoldBody>), dim3((unsigned)fgrid), dim3(256), 0, stream, fold, natoms, frz);
        NEPMI_HIP_CHECK(hipGetLastError());
      }
      if (t)
        timer_stop(timing->slot[slot_fold]);
    }
  }
#else
  static constexpr bool kHasBrickForce = false;
  template <class S>
  size_t brick_lds_bytes(const ModelD&, int) const { return 0; }
  template <class S>
  void launch_brick_force(int, int, int64_t, int64_t, const WinStage&, const ModelD&, int*, const unsigned*, int, bool, const float*, const int*)
  {
  }
#endif


  void exclusive_scan(int* data, int64_t n, int* scratch)
  {
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(nepmi_scan_tiles, dim3((unsigned)tiles), dim3(kScanBlock), 0, stream, data, n, scratch);
    NEPMI_HIP_CHECK(hipGetLastError());
    if (tiles > 1) {
      hipLaunchKernelGGL(nepmi_scan_sums, dim3(1), dim3(1024), 0, stream, scratch, tiles);
      hipLaunchKernelGGL(nepmi_scan_add, dim3((unsigned)tiles), dim3(kScanBlock), 0, stream, data, n, scratch);
      NEPMI_HIP_CHECK(hipGetLastError());
    }
  }

  void thermo(
    int slot, int64_t n, double volume, const double* mass, const double* pe, const double* vel,
    const double* virial, double* thermo8, double* scratch, const signed char* lvl = nullptr, int raw = 0,
    int64_t n_norm = 0, int virial_lvl = 2)
  {
    int64_t nb = (n + kThermoBlock - 1) / kThermoBlock;
    if (nb > kThermoMaxBlocks)
      nb = kThermoMaxBlocks;
    hipLaunchKernelGGL(
      nepmi_thermo_partial, dim3((unsigned)nb), dim3(kThermoBlock), 0, stream, n, mass, pe, vel, virial, lvl, scratch, virial_lvl);
    hipLaunchKernelGGL(nepmi_thermo_final, dim3(1), dim3(64), 0, stream, (int)nb, n_norm > 0 ? n_norm : n, volume, scratch,
                       thermo8, raw);
    NEPMI_HIP_CHECK(hipGetLastError());
    (void)slot;
  }

  // Langevin thermostat (kernels above): per-atom generator states, one half-step
  size_t lan_state_bytes() const { return sizeof(hiprandState); }
  void lan_init(void* states, int64_t n, int seed)
  {
    hipLaunchKernelGGL(nepmi_lan_init, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, (hiprandState*)states, n, seed);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
  // resident forms (see the kernels above); the velocity correction itself is ResidentMomentumFixBody
  void lan_kick_resident(void* states, int64_t n, double c1, double c2, const double* mi, double* vi, const int* perm,
                         const signed char* lvl, const int64_t* ids, const int* flags)
  {
    hipLaunchKernelGGL(nepmi_lan_kick_resident, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, (hiprandState*)states, n,
                       c1, c2, mi, vi, perm, lvl, ids, flags);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
  void lan_init_ids(void* states, int64_t n, const int64_t* ids, int seed)
  {
    hipLaunchKernelGGL(nepmi_lan_init_ids, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, (hiprandState*)states, n, ids,
                       seed);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
  void lan_momentum_resident(int64_t n, const double* mi, const double* vi, const int* invp, const signed char* lvl,
                             double* sums4, const int* flags)
  {
    hipLaunchKernelGGL(nepmi_momentum_sum_resident, dim3(4), dim3(1024), 0, stream, n, mi, vi, invp, lvl, sums4, flags);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
  void lan_momentum_fix_resident(int64_t n, const double* sums4, double* vi, const signed char* lvl, const int* flags)
  {
    hipLaunchKernelGGL(nepmi_momentum_fix_resident, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, n, sums4, vi, lvl, flags);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
  void lan_half(void* states, int64_t n, double c1, double c2, const double* mass, double* vel, double* sums4)
  {
    const unsigned grid = (unsigned)((n + 127) / 128);
    hipLaunchKernelGGL(nepmi_lan_kick, dim3(grid), dim3(128), 0, stream, (hiprandState*)states, n, c1, c2, mass, vel);
    hipLaunchKernelGGL(nepmi_momentum_sum, dim3(4), dim3(1024), 0, stream, n, mass, vel, sums4);
    hipLaunchKernelGGL(nepmi_momentum_fix, dim3(grid), dim3(128), 0, stream, n, sums4, vel);
    NEPMI_HIP_CHECK(hipGetLastError());
  }
};

} // namespace nepmi

using NepmiBackend = nepmi::HipBackend;

static NepmiBackend nepmi_make_backend(void* stream)
{
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
    throw nepmi::EngineError{-5, "no HIP device visible: libnepmi.so needs an MI355X (gfx950); there is no CPU path"};
  int dev = 0;
  NEPMI_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  NEPMI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    throw nepmi::EngineError{-5, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only"};
  NepmiBackend b;
  b.stream = (hipStream_t)stream;
  b.timing = new nepmi::HipTiming();
  void* p = nullptr;
  NEPMI_HIP_CHECK(hipHostMalloc(&p, 64, hipHostMallocDefault));
  b.pinned = (int*)p;
  return b;
}

// ---- C ABI.  The implementations below (capi_impl.h, dist_capi_impl.h, the RCCL transport) are compiled as nepmi_*__impl;
//      the public names are the per-handle trampolines at the end of this file (capi_dispatch.inc, generated from nepmi.h). ----
#define NEPMI_CAPI_DISPATCH
#define NEPMI_CAPI_RENAME
#include "capi_dispatch.inc"
#undef NEPMI_CAPI_RENAME
#include "dist_capi_impl.h"

// ---- RCCL transport (device buffers over xGMI): nepmi_transport_rccl ---------------------------------------
// librccl.so is opened on first use, so that a single-GPU host without RCCL can still load libnepmi.so.
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: every call goes through the table below

namespace {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;       // (optional: statistics only)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  bool ok = false;
};

const RcclApi& rccl_api()
{
  static const RcclApi api = [] {
    RcclApi a;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
      h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
      h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
      return a;
#define NEPMI_RCCL_SYM(name) a.name = reinterpret_cast<decltype(a.name)>(dlsym(h, "nccl" #name))
    NEPMI_RCCL_SYM(GetUniqueId);
    NEPMI_RCCL_SYM(CommInitRank);
    NEPMI_RCCL_SYM(CommDestroy);
    NEPMI_RCCL_SYM(GroupStart);
    NEPMI_RCCL_SYM(GroupEnd);
    NEPMI_RCCL_SYM(Send);
    NEPMI_RCCL_SYM(Recv);
    NEPMI_RCCL_SYM(AllReduce);
    NEPMI_RCCL_SYM(CommCount);
    NEPMI_RCCL_SYM(CommUserRank);
#undef NEPMI_RCCL_SYM
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.AllReduce;
    return a;
  }();
  return api;
}

struct RcclCtx {
  ncclComm_t comm;
  // what travelled (nepmi_transport_rccl_stats): grouped exchanges, their messages and bytes, reductions; and, for every
  // `time_every`-th exchange, a pair of HIP events on the exchange's own stream around the ncclGroup
  int64_t n_exchange = 0, n_msgs = 0, bytes_sent = 0, bytes_recv = 0, n_allreduce = 0;
  int time_every = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed; // recorded pairs, drained by the stats call
  double timed_us_sum = 0.0;
  int64_t timed_count = 0;
  // NEPMI_RCCL_FUSE_VOTE=1: a reduction passed with NEPMI_DT_DEFER waits for the next exchange on its stream and is posted inside
  // that call's ncclGroup (off by default: a collective and point-to-point operations in one group has only ever run here on a
  // single rank)
  bool fuse = false, pending = false;
  void *p_buf = nullptr, *p_stream = nullptr;
  int64_t p_count = 0;
  int p_dtype = 0, p_op = 0;
};

ncclResult_t rccl_reduce_now(RcclCtx* c, void* buf, int64_t count, int dtype, int op, void* stream)
{
  const ncclDataType_t dt = dtype == 0 ? ncclFloat64 : (dtype == 1 ? ncclInt32 : ncclInt64);
  const ncclRedOp_t ro = op == 0 ? ncclSum : ncclMax;
  return rccl_api().AllReduce(buf, buf, (size_t)count, dt, ro, c->comm, (hipStream_t)stream);
}

int rccl_exchange(void* vctx, int ns, const nepmi_msg* sends, int nr, const nepmi_msg* recvs, void* stream)
{
  RcclCtx* c = (RcclCtx*)vctx;
  const RcclApi& R = rccl_api();
  bool ok = true;
  if (c->pending && c->p_stream != stream) { // a deferred reduction of another stream: on its own, first
    ok = rccl_reduce_now(c, c->p_buf, c->p_count, c->p_dtype, c->p_op, c->p_stream) == ncclSuccess;
    c->pending = false;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (c->time_every > 0 && c->n_exchange % c->time_every == 0 && c->timed.size() < 4096 &&
      hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess)
    (void)hipEventRecord(ev0, (hipStream_t)stream);
  ++c->n_exchange;
  c->n_msgs += ns + nr;
  for (int k = 0; k < ns; ++k)
    c->bytes_sent += sends[k].bytes;
  for (int k = 0; k < nr; ++k)
    c->bytes_recv += recvs[k].bytes;
  ok = (R.GroupStart() == ncclSuccess) && ok;
  if (c->pending && ok) { // the skin vote rides in this group: one collective for the vote and the ghost positions
    ok = rccl_reduce_now(c, c->p_buf, c->p_count, c->p_dtype, c->p_op, stream) == ncclSuccess;
    c->pending = false;
  }
  for (int k = 0; k < ns && ok; ++k)
    ok = R.Send(sends[k].buf, (size_t)sends[k].bytes, ncclChar, sends[k].peer, c->comm, (hipStream_t)stream) == ncclSuccess;
  for (int k = 0; k < nr && ok; ++k)
    ok = R.Recv(recvs[k].buf, (size_t)recvs[k].bytes, ncclChar, recvs[k].peer, c->comm, (hipStream_t)stream) == ncclSuccess;
  ok = (R.GroupEnd() == ncclSuccess) && ok;
  if (ev0 && ev1) {
    (void)hipEventRecord(ev1, (hipStream_t)stream);
    c->timed.emplace_back(ev0, ev1);
  } else if (ev0) { // the second event could not be created: this exchange is not timed
    (void)hipEventDestroy(ev0);
  }
  return ok ? 0 : -1;
}

int rccl_allreduce(void* vctx, void* buf, int64_t count, int dtype, int op, void* stream)
{
  RcclCtx* c = (RcclCtx*)vctx;
  bool ok = true;
  ++c->n_allreduce;
  if (c->pending) { // reductions stay in order
    ok = rccl_reduce_now(c, c->p_buf, c->p_count, c->p_dtype, c->p_op, c->p_stream) == ncclSuccess;
    c->pending = false;
  }
  if ((dtype & NEPMI_DT_DEFER) && c->fuse) {
    c->pending = true;
    c->p_buf = buf;
    c->p_count = count;
    c->p_dtype = dtype & 0xFF;
    c->p_op = op;
    c->p_stream = stream;
    return ok ? 0 : -1;
  }
  return (rccl_reduce_now(c, buf, count, dtype & 0xFF, op, stream) == ncclSuccess && ok) ? 0 : -1;
}

void rccl_drain_timed(RcclCtx* c)
{
  for (auto& pr : c->timed) {
    float ms = 0.0f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      c->timed_us_sum += 1.0e3 * (double)ms;
      ++c->timed_count;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->timed.clear();
}

void rccl_destroy(void* vctx)
{
  RcclCtx* c = (RcclCtx*)vctx;
  if (c) {
    rccl_drain_timed(c);
    rccl_api().CommDestroy(c->comm);
    delete c;
  }
}

} // namespace

extern "C" int nepmi_transport_rccl_id(char id[NEPMI_RCCL_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) <= NEPMI_RCCL_ID_BYTES, "ncclUniqueId does not fit the id buffer");
  if (!rccl_api().ok)
    return fail(NEPMI_ERR_HIP, "librccl.so could not be opened: the RCCL transport is not available on this host");
  ncclUniqueId u;
  if (rccl_api().GetUniqueId(&u) != ncclSuccess)
    return fail(NEPMI_ERR_HIP, "ncclGetUniqueId failed");
  std::memset(id, 0, NEPMI_RCCL_ID_BYTES);
  std::memcpy(id, &u, sizeof u);
  return NEPMI_OK;
}

extern "C" int nepmi_transport_rccl(const char id[NEPMI_RCCL_ID_BYTES], int rank, int nranks, nepmi_transport* out)
{
  if (!id || !out || rank < 0 || rank >= nranks)
    return fail(NEPMI_ERR_ARG, "bad argument");
  if (!rccl_api().ok)
    return fail(NEPMI_ERR_HIP, "librccl.so could not be opened: the RCCL transport is not available on this host");
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  RcclCtx* c = new RcclCtx();
  if (rccl_api().CommInitRank(&c->comm, nranks, u, rank) != ncclSuccess) {
    delete c;
    return fail(NEPMI_ERR_HIP, "ncclCommInitRank failed (one process per GPU: two ranks cannot share a device)");
  }
  out->ctx = c;
  out->rank = rank;
  out->nranks = nranks;
  {
    const char* fv = std::getenv("NEPMI_RCCL_FUSE_VOTE");
    ((RcclCtx*)out->ctx)->fuse = fv && fv[0] == '1';
  }
  out->device_buffers = ((RcclCtx*)out->ctx)->fuse ? 3 : 1;
  out->exchange = rccl_exchange;
  out->allreduce = rccl_allreduce;
  out->destroy = rccl_destroy;
  return NEPMI_OK;
}

extern "C" int nepmi_transport_rccl_stats(const nepmi_transport* t, int time_every, int reset, nepmi_rccl_stats* out)
{
  if (!t || !t->ctx || t->exchange != rccl_exchange)
    return fail(NEPMI_ERR_ARG, "not an RCCL transport");
  RcclCtx* c = (RcclCtx*)t->ctx;
  rccl_drain_timed(c);
  if (out) {
    int cnt = -1, ur = -1;
    if (rccl_api().CommCount)
      (void)rccl_api().CommCount(c->comm, &cnt);
    if (rccl_api().CommUserRank)
      (void)rccl_api().CommUserRank(c->comm, &ur);
    out->comm_nranks = cnt;
    out->comm_rank = ur;
    out->exchanges = c->n_exchange;
    out->messages = c->n_msgs;
    out->bytes_sent = c->bytes_sent;
    out->bytes_received = c->bytes_recv;
    out->allreduces = c->n_allreduce;
    out->timed_exchanges = c->timed_count;
    out->us_per_timed_exchange = c->timed_count > 0 ? c->timed_us_sum / (double)c->timed_count : 0.0;
  }
  if (reset) {
    c->n_exchange = c->n_msgs = c->bytes_sent = c->bytes_recv = c->n_allreduce = 0;
    c->timed_us_sum = 0.0;
    c->timed_count = 0;
  }
  c->time_every = time_every > 0 ? time_every : 0;
  return NEPMI_OK;
}

// ---- which library serves a model (capi_jit.h); host code only ----
#if !defined(__HIP_DEVICE_COMPILE__)
#include "capi_jit.h"

static const nepmi_api* nepmi_jit_core_for_model_file(const char* path); // (defined behind the table: it checks the core's table size)

static void nepmi_adopt_error(const nepmi_api* core); // (defined behind the table: it reads the core's own last error)

#define NEPMI_CAPI_TRAMPOLINES
#include "capi_dispatch.inc"
#undef NEPMI_CAPI_TRAMPOLINES

static void nepmi_adopt_error(const nepmi_api* core)
{
  if (core && core->nepmi_last_error)
    g_last_error = core->nepmi_last_error();
}

// nullptr: this library (a compiled shape, a shape only the run-time-shape kernels handle, a Tersoff file, a file that does not
// parse -- the caller's own nepmi_model_load reports that -- or NEPMI_JIT=0); else the JIT core compiled for the model's shape
static const nepmi_api* nepmi_jit_core_for_model_file(const char* path)
{
#if defined(NEPMI_JIT_CORE)
  (void)path;
  return nullptr; // a core serves the shape it was compiled for (and nothing is compiled from inside a core)
#else
  const char* mode = std::getenv("NEPMI_JIT");
  const char* force_cover = std::getenv("NEPMI_FORCE_COVER");
  if (!path || (mode && mode[0] == '0') || (force_cover && force_cover[0] == '1'))
    return nullptr;
  nepmi::NepModel m;
  bool unsupported = false;
  if (!nepmi::load_nep_model(path, m, &unsupported).empty())
    return nullptr;
  if (m.kind != 0 || nepmi::builtin_shape_of(m) != 0 || !nepmi::shape_is_compilable(m))
    return nullptr;
  const nepmi::jit::ShapeKey key{m.n_max_radial, m.basis_size_radial, m.n_max_angular, m.basis_size_angular, m.num_L,
                                 m.num_types <= 2 ? m.num_types : 0};
  return nepmi::jit::core_for(key, (uint64_t)sizeof(nepmi_api));
#endif
}

// what this library was built from (capi_jit.h: load_core checks a core against the library that loads it)
extern "C" nepmi::jit::CoreAbi nepmi_core_abi(void) { return nepmi::jit::CoreAbi{nepmi::jit::baked_hash(), (uint64_t)sizeof(nepmi_api)}; }
#endif // !__HIP_DEVICE_COMPILE__
