// Angular descriptor + per-atom ANN + partial angular forces in ONE kernel (device only: gfx950; tests/emu keeps the separate
// kernels, whose results these equal up to the summation order of the ANN's dot products).
//
// Replaces, in one launch: the angular half of find_descriptor (nep.cu:549-640), apply_ann_one_layer
// (nep_utilities.cuh:169-194), find_partial_force_angular (nep.cu:774-861) and find_force_ZBL (nep.cu:863-975) -- the
// reference's own descriptor + ANN fusion (nep.cu:488-659) taken one kernel further.  The separate kernels
// (AngularDescBody::fuse_ann, AngularForceBody<.., recompute_s>) evaluate the sums s_{n,lm} TWICE, because the adjoint table
// G = dU/ds needs Fp, which only exists after the ANN, i.e. in the next launch: a second walk over the pair records, 145 of the
// 479 instructions per pair of the force kernel.  Here the sums stay in the registers across the ANN and become G in place.
//
// Two adjacent lanes per atom (the lane-pair form of the angular kernels): lane `part` owns the radial channels
// n = part, part + 2, ... -- its 24 sums per channel, its rows of G -- and, for the ANN, the descriptor components that belong
// to those channels (d-split): q_half = {q_rad[n], q_ang[L][n] : n = part mod 2}.  Per neuron each lane forms its half of the
// dot product from its half of the weight row (LDS image below), the halves meet in one v_add_f32_dpp, both lanes apply tanh,
// and each accumulates Fp for ITS components only: neither q nor Fp is ever exchanged, and what each lane ends up with is
// exactly the Fp rows its own channels' adjoint needs.  The radial force table A[t2][k] = sum_n Fp[n] c[t1][t2][n][k]
// (AnnBody) is two half sums joined the same way.
//
// LDS per workgroup (floats):  c_ang [T T][stride] | Wh [T][neuron][2][DPH] | b0 [T][neuron] | w1 [T][neuron] |
//                              c_rad [T T][(n_r+1)(k_r+1)] | qscale_half [2][DPH]
// DPH = components per lane, zero padded to a multiple of four (PbTe: 4 radial + 5 x 4 angular = 24): a weight half-row is
// DPH / 4 ds_read_b128.
#pragma once
#include "nep_window.h" // F4f

namespace nepmi {

template <class S>
struct FusedShape {
  static_assert(S::fixed, "compiled shapes only");
  static constexpr int NRH = (S::NR + 2) / 2;  // radial components per lane (lane 0 has the extra one of an odd count)
  static constexpr int NLOC = (S::NA + 2) / 2; // angular channels per lane
  static constexpr int DPH = (NRH + S::NL * NLOC + 3) / 4 * 4;
  // descriptor component of local index i on lane `part`, or -1 (padding)
  NEPMI_HD static int component(int part, int i)
  {
    if (i < NRH) {
      const int n = 2 * i + part;
      return n <= S::NR ? n : -1;
    }
    const int ii = i - NRH, L = ii / NLOC, c = ii - L * NLOC, n = 2 * c + part;
    if (L >= S::NL || n > S::NA)
      return -1;
    return (S::NR + 1) + L * (S::NA + 1) + n;
  }
};

struct FusedLdsLayout {
  int wstride, off_w, off_b0, off_w1, off_c, off_qs, total;
};
// TW = 0: every type of the model (models with at most four types).  TW > 0: a WINDOW of TW consecutive types -- the form for
// many-type models (UNEP-v1: 16 types, 246 KB of weight half-rows): the atoms are taken in the type-sorted work order of the
// matrix-core ANN (Bufs::tperm), a workgroup's 128 atoms span two or three types, and only those types' slices of the image
// are staged: c_ang rows [tb, tb + TW) x T, the weight half-rows, biases and output weights of those types, the scalers.  No
// radial table (such models' force assembly contracts from Bufs::fpr).
template <class S>
NEPMI_HD FusedLdsLayout fused_lds_layout(const ModelD& m, int TW = 0)
{
  using F = FusedShape<S>;
  const int tw = TW > 0 ? TW : m.T;
  FusedLdsLayout a;
  a.wstride = m.nneu * 2 * F::DPH;
  a.wstride += (8 - (a.wstride & 31) + 32) & 31; // type stride 8 mod 32 words: lanes of two types read different banks
  a.off_w = (tw * m.T * cang_stride(m) + 3) / 4 * 4;
  a.off_b0 = a.off_w + tw * a.wstride;
  a.off_w1 = a.off_b0 + tw * m.nneu;
  a.off_c = a.off_w1 + tw * m.nneu;
  a.off_qs = a.off_c + (TW > 0 ? 0 : m.T * m.T * (m.NR + 1) * (m.KR + 1));
  a.off_qs = (a.off_qs + 3) / 4 * 4;
  a.total = (a.off_qs + 2 * F::DPH + 3) / 4 * 4;
  return a;
}

template <class S>
struct AngularFusedBody {
  ModelD m;
  Bufs b;
  int export_qfp; // parity hooks (nepmi_descriptors_export): also write the angular descriptor and Fp, which otherwise never
                  // leave the registers
  const float* img; // the LDS image below, built once in global memory (backend: nepmi_fused_image), or nullptr = build it here.
                    // Building it costs a few integer divisions per element -- per workgroup of 128 atoms that was as many
                    // instructions as the ANN itself; the copy is one 16-byte load and store per four elements.
  int tw = 0;       // type slots of the LDS window (0: all types; see fused_lds_layout).  The global image always has every type.
  static constexpr bool kUsesLds = true;
#ifndef NEPMI_AFU_WAVES
#define NEPMI_AFU_WAVES 2
#endif
#ifndef NEPMI_AFU_ABL
#define NEPMI_AFU_ABL 0 // ablation builds (timings only, results are wrong): 1 no ANN loop, 2 no pair loop, 3 no sums, 4 no atab
#endif
#ifndef NEPMI_AFU_NJ
#define NEPMI_AFU_NJ 1 // neurons per trip of the ANN loop (A/B switch; r5: 1 -> 0.500 ms, 2 -> 0.531, 3 -> 0.538 on PbTe 1 M atoms)
#endif
  static constexpr int kMinWavesPerEu = 1, kMinWavesPerEuPairs = NEPMI_AFU_WAVES;
  using F = FusedShape<S>;

  NEPMI_HD int lds_floats() const { return fused_lds_layout<S>(m, tw).total; }
  // the slices of types [tb, tb + nt) of the global image -> the LDS window (layout with `tw` slots)
  NEPMI_HD void stage_window(float* dst, int tb, int nt, int tid, int nth) const
  {
    const FusedLdsLayout G = fused_lds_layout<S>(m, 0), A = fused_lds_layout<S>(m, tw);
    const int cs = cang_stride(m);
    {
      const float* src = img + (size_t)tb * m.T * cs;
      for (int i = tid; i < nt * m.T * cs; i += nth)
        dst[i] = src[i];
    }
    {
      const F4f* __restrict__ src = reinterpret_cast<const F4f*>(img + G.off_w + (size_t)tb * G.wstride); // (16-byte aligned: off_w and
      F4f* d4 = reinterpret_cast<F4f*>(dst + A.off_w);                                                     //  wstride are multiples of 4)
      for (int i = tid; i < nt * A.wstride / 4; i += nth)
        d4[i] = src[i];
    }
    for (int i = tid; i < nt * m.nneu; i += nth) {
      dst[A.off_b0 + i] = img[G.off_b0 + tb * m.nneu + i];
      dst[A.off_w1 + i] = img[G.off_w1 + tb * m.nneu + i];
    }
    for (int i = tid; i < 2 * F::DPH; i += nth)
      dst[A.off_qs + i] = img[G.off_qs + i];
  }
  NEPMI_HD void lds_stage(float* dst, int tid, int nth) const
  {
    if (img) {
      const int n4 = (fused_lds_layout<S>(m).total + 3) / 4;
      const F4f* __restrict__ src = reinterpret_cast<const F4f*>(img);
      F4f* d4 = reinterpret_cast<F4f*>(dst);
      for (int i = tid; i < n4; i += nth)
        d4[i] = src[i];
      return;
    }
    cang_stage(m, dst, tid, nth);
    const FusedLdsLayout a = fused_lds_layout<S>(m);
    const int per_t = m.nneu * 2 * F::DPH;
    for (int idx = tid; idx < m.T * per_t; idx += nth) {
      const int t = idx / per_t, r = idx - t * per_t;
      const int j = r / (2 * F::DPH), pi = r - j * (2 * F::DPH);
      const int p = pi / F::DPH, i = pi - p * F::DPH;
      const int d = F::component(p, i);
      dst[a.off_w + t * a.wstride + r] = d >= 0 ? m.w0[((size_t)t * m.nneu + j) * m.dim + d] : 0.0f;
    }
    for (int idx = tid; idx < m.T * m.nneu; idx += nth) {
      dst[a.off_b0 + idx] = m.b0[idx];
      dst[a.off_w1 + idx] = m.w1[idx];
    }
    for (int idx = tid; idx < m.T * m.T * (m.NR + 1) * (m.KR + 1); idx += nth)
      dst[a.off_c + idx] = m.c_rad[idx];
    for (int idx = tid; idx < 2 * F::DPH; idx += nth) {
      const int d = F::component(idx / F::DPH, idx % F::DPH);
      dst[a.off_qs + idx] = d >= 0 ? m.qscale[d] : 0.0f;
    }
  }

  // what the sorted launch orders a workgroup's atoms by: the atom's angular neighbours this step (0: no work at all)
  NEPMI_HD int sort_key(int64_t k) const
  {
    if (b.lvl[k] < b.lvl_desc)
      return 0;
    return (b.use_csync && b.nn_angtrue) ? b.nn_angtrue[k] : b.nn_angstep[k]; // (padded rows: the neighbours behind them)
  }
  template <int PARTS, class LP>
  NEPMI_HD void run_parts(int64_t k, int part, LP lds) const
  {
    static_assert(PARTS == 2, "lane pairs");
    run_window(k, part, lds, 0);
  }
  // tb: the first type of the LDS window (0 when every type is resident)
  template <class LP>
  NEPMI_HD void run_window(int64_t k, int part, LP lds, int tb) const
  {
    constexpr int NLOC = F::NLOC, DPH = F::DPH;
    if (b.lvl[k] < b.lvl_desc)
      return;
    const int t1 = b.posq[k].type;
    LP cang = lds - tb * m.T * cang_stride(m); // (block (t1, t2) of the window sits where block (t1 - tb, t2) of a full table would)
    float s[NLOC * kNumHarm], Fp[DPH], e;
    descriptor_and_ann(k, part, lds, cang, t1, t1 - tb, s, Fp, e);
    if (part == 0)
      b.pe_i[k] = e;
    // many-type form of the force assembly (shapes without type-pure lists): the atom's radial Fp row, atom-major -- each lane
    // its own components (what AnnBody writes for it)
    if (b.fpr) {
#pragma unroll
      for (int i = 0; i < F::NRH; ++i) {
        const int n = 2 * i + part;
        if (n <= S::NR)
          b.fpr[(size_t)k * b.FPR + n] = Fp[i];
      }
    }
    // ---- radial force table A[t2][k] = sum_n Fp[n] c[t1][t2][n][k]: two half sums; lane t2 mod 2 stores row t2 ----
    if (!b.skip_atab && tw == 0 && NEPMI_AFU_ABL != 4) {
      constexpr int KRPC = (S::KR + 1 + 3) / 4 * 4; // = Bufs::KRP: rows of whole 16-byte groups
      const int KRP = b.KRP;
      for (int t2 = 0; t2 < m.T; ++t2) {
        float row[KRPC];
        radial_row(part, lds, t1, t2, Fp, row);
        if ((t2 & 1) == part) {
          F4f* __restrict__ out = reinterpret_cast<F4f*>(b.atab + (size_t)k * (m.T * KRP) + t2 * KRP);
#pragma unroll
          for (int g = 0; g < KRPC / 4; ++g)
            out[g] = F4f{row[4 * g], row[4 * g + 1], row[4 * g + 2], row[4 * g + 3]};
        }
      }
    }
    // ---- adjoint table in place of the sums, then the pair loop of AngularForceBody ----
    if (b.level && !b.angf[k]) // an inner-ring ghost whose partial forces no owned atom will read
      return;
    adjoint_in_place(part, Fp, s);
    const AngularForceBody<S> af{m, b, 1};
    if (NEPMI_AFU_ABL != 2) {
      af.template pairs_from_G<2>(k, part, cang, t1, s, typename AngularForceBody<S>::F12Store{b.f12 + k, b.N});
    } else if (s[3] + s[NLOC * kNumHarm - 1] == 12345.0f) {
      b.pe_i[k] = s[5] + s[NLOC * kNumHarm - 2];
    }
  }

  // row t2 of the radial force table, whole on both lanes (kk beyond k_r: zero)
  template <class LP>
  NEPMI_HD void radial_row(int part, LP lds, int t1, int t2, const float* Fp, float* row) const
  {
    constexpr int NRH = F::NRH;
    constexpr int KRPC = (S::KR + 1 + 3) / 4 * 4;
    const FusedLdsLayout a = fused_lds_layout<S>(m);
    LP c = lds + a.off_c + (t1 * m.T + t2) * (S::NR + 1) * (S::KR + 1);
#pragma unroll
    for (int kk = 0; kk < KRPC; ++kk) {
      float v = 0.0f;
      if (kk <= S::KR) {
#pragma unroll
        for (int i = 0; i < NRH; ++i) {
          const int n = 2 * i + part;
          if (n <= S::NR)
            v = fmaf(Fp[i], c[n * (S::KR + 1) + kk], v);
        }
        v += NEPMI_PAIR_XCHG(v);
      }
      row[kk] = v;
    }
  }

  // G = dU/ds of this lane's channels in place of their sums (invariants_adjoint with this lane's Fp rows)
  NEPMI_HD void adjoint_in_place(int part, const float* Fp, float* s) const
  {
    constexpr int NRH = F::NRH, NLOC = F::NLOC;
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const int n = part + 2 * i;
      if (n > S::NA)
        break;
      float fpn[S::kRows];
#pragma unroll
      for (int L = 0; L < S::kRows; ++L)
        fpn[L] = L < S::NL ? Fp[NRH + (L < S::NL ? L : 0) * NLOC + i] : 0.0f;
      invariants_adjoint<false>(m, fpn, 1, &s[i * kNumHarm]);
    }
  }

  // sums of this lane's channels -> its half of the descriptor -> ANN: s (kept for the adjoint), Fp of this lane's components,
  // e = the atom's energy (both lanes)
  template <class LP>
  NEPMI_HD void descriptor_and_ann(int64_t k, int part, LP lds, LP cang, int t1, int tl, float* s, float* Fp, float& e_out) const
  {
    constexpr int NRH = F::NRH, NLOC = F::NLOC, DPH = F::DPH;
    const int64_t N = b.N;
    const FusedLdsLayout a = fused_lds_layout<S>(m, tw);
    const int64_t gk = b.tpos[k];
    LP QS = lds + a.off_qs + part * DPH;

    // ---- sums of this lane's channels (angular_s_sums: what AngularDescBody runs) ----
    if (NEPMI_AFU_ABL != 3) {
      angular_s_sums<S, 2>(m, b, k, t1, cang, part, s);
    } else {
#pragma unroll
      for (int i = 0; i < NLOC * kNumHarm; ++i)
        s[i] = (float)(k & 15) * 0.01f + 0.001f * i;
    }

    // ---- this lane's half of the scaled descriptor ----
    float ql[DPH];
#pragma unroll
    for (int i = 0; i < DPH; ++i)
      ql[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NRH; ++i) {
      const int n = 2 * i + part;
      if (n <= S::NR)
        ql[i] = b.q[(int64_t)n * N + gk]; // radial part, written (scaled) by the radial pass
    }
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const int n = part + 2 * i;
      if (n > S::NA)
        break;
      float qn[S::kRows];
#pragma unroll
      for (int L = 0; L < S::kRows; ++L)
        qn[L] = 0.0f;
      invariants<false>(m, &s[i * kNumHarm], qn, 1);
#pragma unroll
      for (int L = 0; L < S::NL; ++L) {
        ql[NRH + L * NLOC + i] = qn[L] * QS[NRH + L * NLOC + i];
        if (export_qfp)
          b.q[(int64_t)((S::NR + 1) + L * (S::NA + 1) + n) * N + gk] = ql[NRH + L * NLOC + i];
      }
    }

    // ---- ANN: per neuron, half a dot product per lane; Fp of this lane's components only ----
    f2 g2[DPH / 2];
#pragma unroll
    for (int i = 0; i < DPH / 2; ++i)
      g2[i] = bc2(0.0f);
    float e = 0.0f;
    {
      LP W = lds + a.off_w + tl * a.wstride + part * DPH; // (tl: the type's slot in the LDS window)
      LP B0 = lds + a.off_b0 + tl * m.nneu;
      LP W1 = lds + a.off_w1 + tl * m.nneu;
      // a half-row of up to 24 weights stays in the registers between the forward dot product and the backward axpy; longer
      // ones (carbon: 36) are read from LDS twice instead -- the sums and the two descriptor halves already fill the file.
      // NJ neurons per trip: their LDS reads go out together and their dot-product / tanh chains (one dependent chain each:
      // 12 packed fmas, a DPP add, v_exp, v_rcp) interleave -- at two wavefronts per SIMD nothing else hides those latencies.
      constexpr bool kKeepRow = DPH <= 24;
      constexpr int NJ = kKeepRow ? NEPMI_AFU_NJ : 1;
      auto neurons = [&](const int j0, const int nj) __attribute__((always_inline)) {
        f2 w2[NJ][kKeepRow ? DPH / 2 : 1];
        f2 acc0[NJ], acc1[NJ];
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
          if (u >= nj)
            break;
          LP w = W + (j0 + u) * (2 * DPH);
          acc0[u] = bc2(0.0f);
          acc1[u] = bc2(0.0f);
#pragma unroll
          for (int i = 0; i < DPH / 2; ++i) {
            const f2 wv = mk2(w[2 * i], w[2 * i + 1]);
            if (kKeepRow)
              w2[u][i] = wv;
            if (i & 1)
              acc1[u] = vfma(wv, mk2(ql[2 * i], ql[2 * i + 1]), acc1[u]);
            else
              acc0[u] = vfma(wv, mk2(ql[2 * i], ql[2 * i + 1]), acc0[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
          if (u >= nj)
            break;
          LP w = W + (j0 + u) * (2 * DPH);
          const f2 acc = acc0[u] + acc1[u];
          float dot = acc.x + acc.y;
          dot += NEPMI_PAIR_XCHG(dot); // (a + b and b + a: both lanes hold the same bits)
          const float h = ann_tanh(dot - B0[j0 + u]);
          const float wj = W1[j0 + u];
          e = fmaf(wj, h, e);
          const f2 coef = bc2(wj * (1.0f - h * h));
#pragma unroll
          for (int i = 0; i < DPH / 2; ++i)
            g2[i] = vfma(coef, kKeepRow ? w2[u][i] : mk2(w[2 * i], w[2 * i + 1]), g2[i]);
        }
      };
      int j = 0;
      for (; j + NJ <= (NEPMI_AFU_ABL == 1 ? 0 : m.nneu); j += NJ)
        neurons(j, NJ);
      if (NJ > 1 && j < m.nneu)
        neurons(j, m.nneu - j);
    }
    e_out = e - (m.b1 + m.b1t[t1]);
#pragma unroll
    for (int i = 0; i < DPH; ++i)
      Fp[i] = ((i & 1) ? g2[i >> 1].y : g2[i >> 1].x) * QS[i];
    if (export_qfp) {
#pragma unroll
      for (int i = 0; i < DPH; ++i) {
        const int d = F::component(part, i);
        if (d >= 0)
          b.fp[(int64_t)d * N + gk] = Fp[i];
      }
    }
  }
};

} // namespace nepmi
