// Device-side data structures and NEP math (inline, single precision unless noted).
//
// Everything here is `NEPMI_HD` so that the per-atom kernel bodies (nep_bodies.h) can also be
// compiled by g++ into the test-only logic emulator under tests/emu (this container has no GPU).
// The product library libnepmi.so contains only the gfx950 device code paths.
//
// Arithmetic that decides neighbour membership (r12, MIC, d^2 < rc^2) is written as explicit fma
// chains in a fixed order, the same chains the oracle documents (oracle/nep_oracle_core.inc:21-27),
// so that neighbour indices are bit-exact; everything else is free to be contracted.
#pragma once
#include <cmath>
#include <cstdint>

#include "nep_invariants_extra.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NEPMI_HD __host__ __device__ __forceinline__
#else
#define NEPMI_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define NEPMI_WAVE_ANY(pred) (__any((int)(pred)))
#define NEPMI_PAIR_XCHG(v) (nepmi::quad_xor<1>(v)) // value held by the partner lane (lanes 2i, 2i+1)
#else
#define NEPMI_WAVE_ANY(pred) (pred)
#define NEPMI_PAIR_XCHG(v) (v) // host loops run one lane per atom: never reached with PARTS > 1
#endif

namespace nepmi {

#if defined(__HIP_DEVICE_COMPILE__)
// Value held by lane (l ^ MSK), MSK = 1 or 2: the lanes that share an atom are adjacent, so the exchange stays inside a quad
// and is a DPP operand modifier (quad_perm) of the consuming VALU instruction -- `v_add_f32_dpp` -- instead of the
// ds_bpermute_b32 + s_waitcnt lgkmcnt(0) that __shfl_xor compiles to (an LDS-pipe round trip that also drains every LDS read
// in flight).  Same values, same sums.
template <int MSK>
__device__ __forceinline__ int quad_xor(int v)
{
  static_assert(MSK == 1 || MSK == 2, "quad-local exchange");
  return __builtin_amdgcn_update_dpp(0, v, MSK == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);
}
template <int MSK>
__device__ __forceinline__ float quad_xor(float v)
{
  return __int_as_float(quad_xor<MSK>(__float_as_int(v)));
}
template <int MSK>
__device__ __forceinline__ unsigned quad_xor(unsigned v)
{
  return (unsigned)quad_xor<MSK>((int)v);
}
#endif

// Read-only model tables (weights, descriptor coefficients) are read through the CONSTANT address
// space on the device: loads whose address is wave-uniform then become scalar (s_load) loads
// feeding SGPR operands instead of 64 identical vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float* cfloat_ptr;
NEPMI_HD cfloat_ptr as_const(const float* p) { return (cfloat_ptr)p; }
// tables staged in LDS by the workgroup are read through an LDS-typed pointer (ds_read, not flat)
typedef const __attribute__((address_space(3))) float* lds_cfloat_ptr;
#else
typedef const float* cfloat_ptr;
NEPMI_HD cfloat_ptr as_const(const float* p) { return p; }
typedef const float* lds_cfloat_ptr;
#endif

constexpr int kNumHarm = 24;       // (L_max+1)^2 - 1 for L_max = 4
constexpr int kIdxBits = 25;       // neighbour index bits in a packed pair word
constexpr int kIdxMask = (1 << kIdxBits) - 1;
constexpr int64_t kMaxAtomsPerEngine = (int64_t)1 << kIdxBits;

// numerical constants of the NEP descriptor definition (nep_utilities.cuh:18-46): squared
// normalisation of the real spherical harmonics and the 4-/5-body coupling constants.
#define NEPMI_C3B_INIT                                                                              \
  {0.238732414637843f, 0.119366207318922f, 0.119366207318922f, 0.099471839432435f,                 \
   0.596831036594608f, 0.596831036594608f, 0.149207759148652f, 0.149207759148652f,                 \
   0.139260575205408f, 0.104445431404056f, 0.104445431404056f, 1.044454314040563f,                 \
   1.044454314040563f, 0.174075719006761f, 0.174075719006761f, 0.011190581936149f,                 \
   0.223811638722978f, 0.223811638722978f, 0.111905819361489f, 0.111905819361489f,                 \
   1.566681471060845f, 1.566681471060845f, 0.195835183882606f, 0.195835183882606f}
#define NEPMI_C4B_0 (-0.007499480826664f)
#define NEPMI_C4B_1 (-0.134990654879954f)
#define NEPMI_C4B_2 (0.067495327439977f)
#define NEPMI_C4B_3 (0.404971964639861f)
#define NEPMI_C4B_4 (-0.809943929279723f)
#define NEPMI_C5B_0 (0.026596810706114f)
#define NEPMI_C5B_1 (0.053193621412227f)
#define NEPMI_C5B_2 (0.026596810706114f)

#define NEPMI_PI 3.1415927f
#define NEPMI_HALF_PI 1.5707963f

// One atom in the engine's internal (cell-sorted) order: FP64 position + type, 32 B.
struct alignas(16) PosQ {
  double x, y, z;
  int type;
  int pad;
};

// Box, the subset of src/model/box.cuh the path needs: h[0..8] cell (columns a,b,c), h[9..17]
// inverse; float copy for the FP32 kernels (Box::float_h... refreshed by set_is_orthogonal,
// box.cu:111-117).
struct BoxD {
  double h[18];
  float hf[18];
  double thickness[3];
  double volume;
  int pbc[3];
  int ortho;
  float rfree2; // (0.49 min periodic thickness)^2: shorter displacements need no periodic image
};

struct ModelD {
  int T, NR, KR, NA, KA;
  int has222, has1111, numL, dim, nneu, version;
  int extra; // bit 0..3: has_q_112, _123, _233, _134 (generic shape only)
  int Lmax;  // l_max_3body, 1..4 (anything but 4: generic shape only)
  int zbl_enabled, zbl_flexible;
  float zbl_rc_inner, zbl_rc_outer;
  float b1;
  float rc_r_max, rc_a_max;
  float rcinv_r, rcinv_a; // 1 / rc_*_max (used when uniform_rc)
  int uniform_rc; // every type has the same (rc_radial, rc_angular): pair cutoffs are constants
  const float* c_rad;  // [T*T][NR+1][KR+1]
  const float* c_ang;  // [T*T][NA+1][KA+1]
  const float* w0;     // [T][nneu][dim]
  const float* b0;     // [T][nneu]
  const float* w1;     // [T][nneu]
  const float* b1t;    // [T]
  const float* qscale; // [dim]
  const float* rc_r;   // [T]
  const float* rc_a;   // [T]
  const float* zbl_para;   // [T(T+1)/2][10]
  const float* zbl_rco;    // [T*T] type-wise outer cutoff of the universal ZBL (inner cutoff 0 then), or nullptr
  const int* atomic_number; // [T]
  // c_rad once more in the two padded LDS layouts of the window kernels of many-type shapes (nep_window.h: ctab_block; [0]
  // element-wise reads, odd block stride; [1] 16-byte reads), so that a workgroup stages its copy with 16-byte loads instead
  // of an integer division per element; nullptr: built by the workgroup
  const float* ctab_img[2];
  const float* cang_img; // c_ang in the LDS layout of the angular kernels (nep_bodies.h: cang_stage), or nullptr
};

// ---- geometry -------------------------------------------------------------------------------

// a*b + c*d + e*f as fma(e,f, fma(c,d, a*b))
NEPMI_HD float dot3f(float a, float b, float c, float d, float e, float f)
{
  return fmaf(e, f, fmaf(c, d, a * b));
}

// apply_mic (float), src/model/box.cuh:84-129
NEPMI_HD void mic_f(const BoxD& box, float& x, float& y, float& z)
{
  const float* H = box.hf;
  if (box.ortho) {
    if (box.pbc[0]) {
      const float L = H[0], hl = L * 0.5f;
      if (x < -hl) x += L; else if (x > hl) x -= L;
    }
    if (box.pbc[1]) {
      const float L = H[4], hl = L * 0.5f;
      if (y < -hl) y += L; else if (y > hl) y -= L;
    }
    if (box.pbc[2]) {
      const float L = H[8], hl = L * 0.5f;
      if (z < -hl) z += L; else if (z > hl) z -= L;
    }
  } else {
    float sx = dot3f(H[9], x, H[10], y, H[11], z);
    float sy = dot3f(H[12], x, H[13], y, H[14], z);
    float sz = dot3f(H[15], x, H[16], y, H[17], z);
    if (box.pbc[0]) sx -= nearbyintf(sx);
    if (box.pbc[1]) sy -= nearbyintf(sy);
    if (box.pbc[2]) sz -= nearbyintf(sz);
    x = dot3f(H[0], sx, H[1], sy, H[2], sz);
    y = dot3f(H[3], sx, H[4], sy, H[5], sz);
    z = dot3f(H[6], sx, H[7], sy, H[8], sz);
  }
}

// r12 exactly as the reference kernels form it: double subtraction, round to float, float MIC
// (nep.cu:467-472); returns d^2 as the fixed fma chain.
NEPMI_HD float pair_geometry(const BoxD& box, const PosQ& a, const PosQ& b, float& x, float& y, float& z)
{
  x = (float)(b.x - a.x);
  y = (float)(b.y - a.y);
  z = (float)(b.z - a.z);
  mic_f(box, x, y, z);
  return dot3f(x, x, y, y, z, z);
}

// The same pair vector for kernels that do not take list decisions (force assembly): in a triclinic
// box the float minimum image is a fractional round trip H (H^-1 r - n) per pair; when no lane of the
// wavefront has a displacement long enough to need an image (every brick away from the box faces) it
// is skipped.  r12 then differs from pair_geometry's by the round trip's rounding (~1e-7 relative).
NEPMI_HD float pair_geometry_fast(const BoxD& box, const PosQ& a, const PosQ& b, float& x, float& y, float& z)
{
  x = (float)(b.x - a.x);
  y = (float)(b.y - a.y);
  z = (float)(b.z - a.z);
  if (box.ortho) {
    mic_f(box, x, y, z);
    return dot3f(x, x, y, y, z, z);
  }
  float d2 = dot3f(x, x, y, y, z, z);
  if (NEPMI_WAVE_ANY(d2 >= box.rfree2)) {
    mic_f(box, x, y, z);
    d2 = dot3f(x, x, y, y, z, z);
  }
  return d2;
}

// gpu_apply_pbc, force.cu:424-459 (double; explicit non-fused ops so the CPU oracle, the device
// and the emulator give identical bits).
NEPMI_HD void wrap_position(const BoxD& box, double& x, double& y, double& z)
{
#pragma clang fp contract(off) // separate roundings, like the oracle (and gcc -ffp-contract=off)
  const double* h = box.h;
  double sx = (h[9] * x + h[10] * y) + h[11] * z;
  double sy = (h[12] * x + h[13] * y) + h[14] * z;
  double sz = (h[15] * x + h[16] * y) + h[17] * z;
  if (box.pbc[0]) { if (sx < 0.0) sx += 1.0; else if (sx > 1.0) sx -= 1.0; }
  if (box.pbc[1]) { if (sy < 0.0) sy += 1.0; else if (sy > 1.0) sy -= 1.0; }
  if (box.pbc[2]) { if (sz < 0.0) sz += 1.0; else if (sz > 1.0) sz -= 1.0; }
  x = (h[0] * sx + h[1] * sy) + h[2] * sz;
  y = (h[3] * sx + h[4] * sy) + h[5] * sz;
  z = (h[6] * sx + h[7] * sy) + h[8] * sz;
}

// find_cell_id, neighbor.cuh:76-110
NEPMI_HD void cell_of(
  const BoxD& box, double x, double y, double z, double rc_inv, int nbx, int nby, int nbz,
  int& cx, int& cy, int& cz)
{
  const double* h = box.h;
  const double sx = h[9] * x + h[10] * y + h[11] * z;
  const double sy = h[12] * x + h[13] * y + h[14] * z;
  const double sz = h[15] * x + h[16] * y + h[17] * z;
  // the products are clamped before the conversion: a run that has blown up (NaN / astronomically large
  // coordinates) must end in an error report, not in an out-of-range cell or an endless wrap loop
  const double fx = sx * box.thickness[0] * rc_inv, fy = sy * box.thickness[1] * rc_inv,
               fz = sz * box.thickness[2] * rc_inv;
  const double lim = 1.0e9;
  cx = (int)floor(fx > -lim ? (fx < lim ? fx : lim) : -lim); // NaN compares false -> -lim
  cy = (int)floor(fy > -lim ? (fy < lim ? fy : lim) : -lim);
  cz = (int)floor(fz > -lim ? (fz < lim ? fz : lim) : -lim);
  if (box.pbc[0]) { cx %= nbx; if (cx < 0) cx += nbx; }
  else { cx = cx < 0 ? 0 : (cx >= nbx ? nbx - 1 : cx); }
  if (box.pbc[1]) { cy %= nby; if (cy < 0) cy += nby; }
  else { cy = cy < 0 ? 0 : (cy >= nby ? nby - 1 : cy); }
  if (box.pbc[2]) { cz %= nbz; if (cz < 0) cz += nbz; }
  else { cz = cz < 0 ? 0 : (cz >= nbz ? nbz - 1 : cz); }
}

// ---- two-wide FP32 values -------------------------------------------------------------------------
// gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two FP32 operations per lane) at the rate of their
// one-wide forms, and the per-pair arithmetic of the list-walking kernels is what bounds them (VALU issue, see
// DESIGN.md): those kernels run two pairs of a lane side by side in one f2.  clang's ext_vector_type maps onto the
// packed instructions (scalar operands are broadcast through op_sel); g++ (the test-only emulator) gets a plain
// struct with the same interface.
#if defined(__clang__)
typedef float f2 __attribute__((ext_vector_type(2)));
NEPMI_HD f2 mk2(float a, float b)
{
  f2 r;
  r.x = a;
  r.y = b;
  return r;
}
NEPMI_HD f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2 {
  float x, y;
};
NEPMI_HD f2 mk2(float a, float b) { return f2{a, b}; }
NEPMI_HD f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
NEPMI_HD f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
NEPMI_HD f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
NEPMI_HD f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
NEPMI_HD f2 operator+(f2 a, float b) { return f2{a.x + b, a.y + b}; }
NEPMI_HD f2 operator-(f2 a, float b) { return f2{a.x - b, a.y - b}; }
NEPMI_HD f2 operator*(f2 a, float b) { return f2{a.x * b, a.y * b}; }
NEPMI_HD f2 operator*(float a, f2 b) { return f2{a * b.x, a * b.y}; }
NEPMI_HD f2 vfma(f2 a, f2 b, f2 c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#endif
NEPMI_HD f2 bc2(float a) { return mk2(a, a); }
NEPMI_HD float vfma(float a, float b, float c) { return fmaf(a, b, c); }
template <class T>
NEPMI_HD T vbc(float a);
template <>
NEPMI_HD float vbc<float>(float a) { return a; }
template <>
NEPMI_HD f2 vbc<f2>(float a) { return bc2(a); }

// ---- radial functions -----------------------------------------------------------------------

// cos(pi t) and sin(pi t) for t in [0, 1]: with y = t - 1/2, cos(pi t) = -sin(pi y) and
// sin(pi t) = cos(pi y); Taylor polynomials in y (|y| <= 1/2) to y^13 / y^14, max error 2e-7.
// The cutoff envelope only ever needs this range, so the general-argument cosf/sinf (range
// reduction, ~40 instructions each) is not needed.
NEPMI_HD void cospi_sinpi_unit(float t, float& c, float& s)
{
  const float y = t - 0.5f, y2 = y * y;
  float sp = 4.663028058e-04f;
  sp = fmaf(sp, y2, -7.370430946e-03f);
  sp = fmaf(sp, y2, 8.214588661e-02f);
  sp = fmaf(sp, y2, -5.992645293e-01f);
  sp = fmaf(sp, y2, 2.550164040e+00f);
  sp = fmaf(sp, y2, -5.167712780e+00f);
  sp = fmaf(sp, y2, 3.141592654e+00f);
  float cp = -1.046381049e-04f;
  cp = fmaf(cp, y2, 1.929574309e-03f);
  cp = fmaf(cp, y2, -2.580689139e-02f);
  cp = fmaf(cp, y2, 2.353306304e-01f);
  cp = fmaf(cp, y2, -1.335262769e+00f);
  cp = fmaf(cp, y2, 4.058712126e+00f);
  cp = fmaf(cp, y2, -4.934802201e+00f);
  c = -(sp * y);
  s = fmaf(cp, y2, 1.0f);
}

// 1 / x by v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~9 instructions per pair)
NEPMI_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// d = sqrt(d2) and 1/d from one reciprocal-square-root (v_rsq_f32, 1 ulp)
NEPMI_HD void dist_and_inv(float d2, float& d, float& dinv)
{
#if defined(__HIP_DEVICE_COMPILE__)
  dinv = __frsqrt_rn(d2);
#else
  dinv = 1.0f / sqrtf(d2);
#endif
  d = d2 * dinv;
}

// find_fc / find_fc_and_fcp, nep_utilities.cuh:409-431: fc = (1 + cos(pi d/rc)) / 2 for d < rc
NEPMI_HD void cutoff_fc(float rcinv, float d, float& fc)
{
  float c, s;
  cospi_sinpi_unit(d * rcinv, c, s);
  fc = fmaf(0.5f, c, 0.5f);
}
NEPMI_HD void cutoff_fc_fcp(float rcinv, float d, float& fc, float& fcp)
{
  float c, s;
  cospi_sinpi_unit(d * rcinv, c, s);
  fc = fmaf(0.5f, c, 0.5f);
  fcp = -NEPMI_HALF_PI * s * rcinv;
}

// find_fn, nep_utilities.cuh:572-588: f_k = (T_k(x)+1)/2 * fc, x = 2 (d/rc - 1)^2 - 1
template <int K>
NEPMI_HD void basis_fn(float rcinv, float d, float fc, float* fn)
{
  const float dr = d * rcinv - 1.0f;
  const float x = 2.0f * dr * dr - 1.0f;
  const float hfc = 0.5f * fc;
  fn[0] = fc;
  if (K >= 1)
    fn[1] = (x + 1.0f) * hfc;
  float tm2 = 1.0f, tm1 = x;
#pragma unroll
  for (int k = 2; k <= K; ++k) {
    const float t = 2.0f * x * tm1 - tm2;
    tm2 = tm1;
    tm1 = t;
    fn[k] = (t + 1.0f) * hfc;
  }
}

// find_fn_and_fnp, nep_utilities.cuh:590-623: values and d/dr via U_{k-1}
template <int K>
NEPMI_HD void basis_fn_fnp(float rcinv, float d, float fc, float fcp, float* fn, float* fnp)
{
  const float dr = d * rcinv - 1.0f;
  const float x = 2.0f * dr * dr - 1.0f;
  const float dxdr = 4.0f * dr * rcinv; // dx/dr
  const float hfc = 0.5f * fc, hfcp = 0.5f * fcp;
  fn[0] = fc;
  fnp[0] = fcp;
  if (K >= 1) {
    fn[1] = (x + 1.0f) * hfc;
    fnp[1] = dxdr * hfc + (x + 1.0f) * hfcp;
  }
  float tm2 = 1.0f, tm1 = x;
  float u0 = 1.0f, u1 = 2.0f * x; // U_0, U_1
#pragma unroll
  for (int k = 2; k <= K; ++k) {
    const float t = 2.0f * x * tm1 - tm2;
    tm2 = tm1;
    tm1 = t;
    // dT_k/dx = k U_{k-1}
    fnp[k] = ((float)k * u1) * dxdr * hfc + (t + 1.0f) * hfcp;
    fn[k] = (t + 1.0f) * hfc;
    const float u2 = 2.0f * x * u1 - u0;
    u0 = u1;
    u1 = u2;
  }
}

// The same envelope and basis for a value type T = float or f2 (two pairs side by side); rcinv is a T as well
// (per-pair cutoffs).  Arithmetic identical to the scalar functions above, element by element.
template <class T>
NEPMI_HD void cospi_sinpi_unit_v(T t, T& c, T& s)
{
  const T y = t - 0.5f, y2 = y * y;
  T sp = vbc<T>(4.663028058e-04f);
  sp = vfma(sp, y2, vbc<T>(-7.370430946e-03f));
  sp = vfma(sp, y2, vbc<T>(8.214588661e-02f));
  sp = vfma(sp, y2, vbc<T>(-5.992645293e-01f));
  sp = vfma(sp, y2, vbc<T>(2.550164040e+00f));
  sp = vfma(sp, y2, vbc<T>(-5.167712780e+00f));
  sp = vfma(sp, y2, vbc<T>(3.141592654e+00f));
  T cp = vbc<T>(-1.046381049e-04f);
  cp = vfma(cp, y2, vbc<T>(1.929574309e-03f));
  cp = vfma(cp, y2, vbc<T>(-2.580689139e-02f));
  cp = vfma(cp, y2, vbc<T>(2.353306304e-01f));
  cp = vfma(cp, y2, vbc<T>(-1.335262769e+00f));
  cp = vfma(cp, y2, vbc<T>(4.058712126e+00f));
  cp = vfma(cp, y2, vbc<T>(-4.934802201e+00f));
  c = -(sp * y);
  s = vfma(cp, y2, vbc<T>(1.0f));
}
// fc only (radial descriptor pass): the sine polynomial of cos(pi t)
template <class T>
NEPMI_HD void cutoff_fc_v(T rcinv, T d, T& fc)
{
  const T y = d * rcinv - 0.5f, y2 = y * y;
  T sp = vbc<T>(4.663028058e-04f);
  sp = vfma(sp, y2, vbc<T>(-7.370430946e-03f));
  sp = vfma(sp, y2, vbc<T>(8.214588661e-02f));
  sp = vfma(sp, y2, vbc<T>(-5.992645293e-01f));
  sp = vfma(sp, y2, vbc<T>(2.550164040e+00f));
  sp = vfma(sp, y2, vbc<T>(-5.167712780e+00f));
  sp = vfma(sp, y2, vbc<T>(3.141592654e+00f));
  fc = vfma(sp * y, vbc<T>(-0.5f), vbc<T>(0.5f));
}
template <class T>
NEPMI_HD void cutoff_fc_fcp_v(T rcinv, T d, T& fc, T& fcp)
{
  T c, s;
  cospi_sinpi_unit_v(d * rcinv, c, s);
  fc = vfma(c, vbc<T>(0.5f), vbc<T>(0.5f));
  fcp = s * rcinv * (-NEPMI_HALF_PI);
}
template <int K, class T>
NEPMI_HD void basis_fn_v(T rcinv, T d, T fc, T* fn)
{
  const T dr = d * rcinv - 1.0f;
  const T x = vfma(dr * 2.0f, dr, vbc<T>(-1.0f));
  const T hfc = fc * 0.5f;
  fn[0] = fc;
  if (K >= 1)
    fn[1] = vfma(x, hfc, hfc);
  const T x2 = x * 2.0f;
  T tm2 = vbc<T>(1.0f), tm1 = x;
#pragma unroll
  for (int k = 2; k <= K; ++k) {
    const T t = vfma(x2, tm1, -tm2);
    tm2 = tm1;
    tm1 = t;
    fn[k] = vfma(t, hfc, hfc);
  }
}
// derivatives only (the force assembly never needs the values)
template <int K, class T>
NEPMI_HD void basis_fnp_v(T rcinv, T d, T fc, T fcp, T* fnp)
{
  const T dr = d * rcinv - 1.0f;
  const T x = vfma(dr * 2.0f, dr, vbc<T>(-1.0f));
  const T hfc = fc * 0.5f, hfcp = fcp * 0.5f;
  const T a = dr * rcinv * 4.0f * hfc; // dx/dr * fc / 2
  fnp[0] = fcp;
  if (K >= 1)
    fnp[1] = vfma(x, hfcp, a + hfcp);
  const T x2 = x * 2.0f;
  T tm2 = vbc<T>(1.0f), tm1 = x;
  T u0 = vbc<T>(1.0f), u1 = x2; // U_0, U_1
#pragma unroll
  for (int k = 2; k <= K; ++k) {
    const T t = vfma(x2, tm1, -tm2);
    tm2 = tm1;
    tm1 = t;
    // dT_k/dx = k U_{k-1}
    fnp[k] = vfma(u1 * (float)k, a, vfma(t, hfcp, hfcp));
    const T u2 = vfma(x2, u1, -u0);
    u0 = u1;
    u1 = u2;
  }
}

// Runtime-K variants (generic fallback path).
NEPMI_HD void basis_fn_rt(int K, float rcinv, float d, float fc, float* fn)
{
  const float dr = d * rcinv - 1.0f;
  const float x = 2.0f * dr * dr - 1.0f;
  const float hfc = 0.5f * fc;
  fn[0] = fc;
  if (K >= 1)
    fn[1] = (x + 1.0f) * hfc;
  float tm2 = 1.0f, tm1 = x;
  for (int k = 2; k <= K; ++k) {
    const float t = 2.0f * x * tm1 - tm2;
    tm2 = tm1;
    tm1 = t;
    fn[k] = (t + 1.0f) * hfc;
  }
}
NEPMI_HD void basis_fn_fnp_rt(int K, float rcinv, float d, float fc, float fcp, float* fn, float* fnp)
{
  const float dr = d * rcinv - 1.0f;
  const float x = 2.0f * dr * dr - 1.0f;
  const float dxdr = 4.0f * dr * rcinv;
  const float hfc = 0.5f * fc, hfcp = 0.5f * fcp;
  fn[0] = fc;
  fnp[0] = fcp;
  if (K >= 1) {
    fn[1] = (x + 1.0f) * hfc;
    fnp[1] = dxdr * hfc + (x + 1.0f) * hfcp;
  }
  float tm2 = 1.0f, tm1 = x;
  float u0 = 1.0f, u1 = 2.0f * x;
  for (int k = 2; k <= K; ++k) {
    const float t = 2.0f * x * tm1 - tm2;
    tm2 = tm1;
    tm1 = t;
    fnp[k] = ((float)k * u1) * dxdr * hfc + (t + 1.0f) * hfcp;
    fn[k] = (t + 1.0f) * hfc;
    const float u2 = 2.0f * x * u1 - u0;
    u0 = u1;
    u1 = u2;
  }
}

// ---- angular basis ----------------------------------------------------------------------------
//
// b_abc(x,y,z) for a unit vector: the 24 unnormalised real harmonics of L = 1..4 in the order of
// accumulate_s (nep_utilities.cuh:1674-1756): block L starts at L^2-1; entry 0 is the m = 0
// polynomial in z, then (Re, Im) pairs of z-polynomial * (x+iy)^m for m = 1..L.
NEPMI_HD void harmonics(float x, float y, float z, float* b)
{
  const float z2 = z * z;
  const float x2y2 = x * x - y * y, xy2 = 2.0f * x * y;          // (x+iy)^2
  const float x3 = x * x2y2 - y * xy2, y3 = x * xy2 + y * x2y2;  // (x+iy)^3
  const float x4 = x * x3 - y * y3, y4 = x * y3 + y * x3;        // (x+iy)^4
  b[0] = z;
  b[1] = x;
  b[2] = y;
  b[3] = 3.0f * z2 - 1.0f;
  b[4] = z * x;
  b[5] = z * y;
  b[6] = x2y2;
  b[7] = xy2;
  const float p31 = 5.0f * z2 - 1.0f;
  b[8] = (5.0f * z2 - 3.0f) * z;
  b[9] = p31 * x;
  b[10] = p31 * y;
  b[11] = z * x2y2;
  b[12] = z * xy2;
  b[13] = x3;
  b[14] = y3;
  const float p41 = (7.0f * z2 - 3.0f) * z, p42 = 7.0f * z2 - 1.0f;
  b[15] = (35.0f * z2 - 30.0f) * z2 + 3.0f;
  b[16] = p41 * x;
  b[17] = p41 * y;
  b[18] = p42 * x2y2;
  b[19] = p42 * xy2;
  b[20] = z * x3;
  b[21] = z * y3;
  b[22] = x4;
  b[23] = y4;
}

// ---- the same 24 harmonics as 12 register pairs (packed-FP32 angular kernels) --------------------------------
// Pair q holds harmonics kHarmPair[2q], kHarmPair[2q+1]: the m = 0 polynomials two by two (L = 1,2 and L = 3,4),
// then the ten (Re, Im) pairs z-polynomial * (x+iy)^m.  A v_pk_fma_f32 then updates two sums at once, and the
// gradient contraction factors through the complex powers C_m = (x+iy)^m and their rotations i C_m.
constexpr int kHarmPairs = 12;
#define NEPMI_HARM_PAIR_INIT {0, 3, 8, 15, 1, 2, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23}

NEPMI_HD void harmonics_pairs(float x, float y, float z, f2* B)
{
  const float z2 = z * z;
  const float x2y2 = x * x - y * y, xy2 = 2.0f * x * y;
  const float x3 = x * x2y2 - y * xy2, y3 = x * xy2 + y * x2y2;
  const float x4 = x * x3 - y * y3, y4 = x * y3 + y * x3;
  const float p31 = 5.0f * z2 - 1.0f;
  const float p41 = (7.0f * z2 - 3.0f) * z, p42 = 7.0f * z2 - 1.0f;
  const f2 C1 = mk2(x, y), C2 = mk2(x2y2, xy2), C3 = mk2(x3, y3);
  B[0] = mk2(z, 3.0f * z2 - 1.0f);
  B[1] = mk2((5.0f * z2 - 3.0f) * z, (35.0f * z2 - 30.0f) * z2 + 3.0f);
  B[2] = C1;
  B[3] = C1 * z;
  B[4] = C2;
  B[5] = C1 * p31;
  B[6] = C2 * z;
  B[7] = C3;
  B[8] = C1 * p41;
  B[9] = C2 * p42;
  B[10] = C3 * z;
  B[11] = mk2(x4, y4);
}

// harmonics_contract on the paired sums: w = sum Q b, v = sum P grad b.  With U_m = the P pairs that multiply
// m C_{m-1} in d/dx (and m i C_{m-1} in d/dy), the (x, y) gradient is three packed products per direction.
NEPMI_HD void harmonics_contract_pairs(
  float x, float y, float z, const f2* P, const f2* Q, float& w, float& vx, float& vy, float& vz)
{
  const float z2 = z * z;
  const float x2y2 = x * x - y * y, xy2 = 2.0f * x * y;
  const float x3 = x * x2y2 - y * xy2, y3 = x * xy2 + y * x2y2;
  const float x4 = x * x3 - y * y3, y4 = x * y3 + y * x3;
  const float p31 = 5.0f * z2 - 1.0f;
  const float p41 = (7.0f * z2 - 3.0f) * z, p42 = 7.0f * z2 - 1.0f;
  const f2 C1 = mk2(x, y), C2 = mk2(x2y2, xy2), C3 = mk2(x3, y3);
  // values
  f2 ws = Q[0] * mk2(z, 3.0f * z2 - 1.0f);
  ws = vfma(Q[1], mk2((5.0f * z2 - 3.0f) * z, (35.0f * z2 - 30.0f) * z2 + 3.0f), ws);
  ws = vfma(Q[2], C1, ws);
  ws = vfma(Q[3] * z, C1, ws);
  ws = vfma(Q[4], C2, ws);
  ws = vfma(Q[5] * p31, C1, ws);
  ws = vfma(Q[6] * z, C2, ws);
  ws = vfma(Q[7], C3, ws);
  ws = vfma(Q[8] * p41, C1, ws);
  ws = vfma(Q[9] * p42, C2, ws);
  ws = vfma(Q[10] * z, C3, ws);
  ws = vfma(Q[11], mk2(x4, y4), ws);
  w = ws.x + ws.y;
  // d/dz
  f2 zs = P[0] * mk2(1.0f, 6.0f * z);
  zs = vfma(P[1], mk2(15.0f * z2 - 3.0f, (140.0f * z2 - 60.0f) * z), zs);
  const f2 Z1 = vfma(P[8], bc2(21.0f * z2 - 3.0f), vfma(P[5], bc2(10.0f * z), P[3]));
  const f2 Z2 = vfma(P[9], bc2(14.0f * z), P[6]);
  zs = vfma(Z1, C1, zs);
  zs = vfma(Z2, C2, zs);
  zs = vfma(P[10], C3, zs);
  vz = zs.x + zs.y;
  // d/dx, d/dy of the m = 1 pairs: (d/dx Re, d/dy Im) = the z-polynomial itself
  const f2 v1 = vfma(P[8], bc2(p41), vfma(P[5], bc2(p31), vfma(P[3], bc2(z), P[2])));
  // m >= 2: d/dx (Re, Im)_m = m (Re, Im)_{m-1}, d/dy (Re, Im)_m = m (-Im, Re)_{m-1}
  const f2 U1 = vfma(P[9], bc2(p42), vfma(P[6], bc2(z), P[4])) * 2.0f;
  const f2 U2 = vfma(P[10], bc2(z), P[7]) * 3.0f;
  const f2 U3 = P[11] * 4.0f;
  const f2 R1 = mk2(-y, x), R2 = mk2(-xy2, x2y2), R3 = mk2(-y3, x3);
  const f2 xs = vfma(U3, C3, vfma(U2, C2, U1 * C1));
  const f2 ys = vfma(U3, R3, vfma(U2, R2, U1 * R1));
  vx = v1.x + (xs.x + xs.y);
  vy = v1.y + (ys.x + ys.y);
}

// w = sum_abc Q[abc] b_abc ,  v = sum_abc P[abc] grad_u b_abc   (grad over unconstrained x,y,z)
NEPMI_HD void harmonics_contract(
  float x, float y, float z, const float* P, const float* Q, float& w, float& vx, float& vy, float& vz)
{
  const float z2 = z * z;
  const float x2y2 = x * x - y * y, xy2 = 2.0f * x * y;
  const float x3 = x * x2y2 - y * xy2, y3 = x * xy2 + y * x2y2;
  const float x4 = x * x3 - y * y3, y4 = x * y3 + y * x3;
  const float p31 = 5.0f * z2 - 1.0f;
  const float p41 = (7.0f * z2 - 3.0f) * z, p42 = 7.0f * z2 - 1.0f;
  // values
  w = Q[0] * z + Q[1] * x + Q[2] * y + Q[3] * (3.0f * z2 - 1.0f) + Q[4] * (z * x) + Q[5] * (z * y) +
      Q[6] * x2y2 + Q[7] * xy2 + Q[8] * ((5.0f * z2 - 3.0f) * z) + Q[9] * (p31 * x) + Q[10] * (p31 * y) +
      Q[11] * (z * x2y2) + Q[12] * (z * xy2) + Q[13] * x3 + Q[14] * y3 +
      Q[15] * ((35.0f * z2 - 30.0f) * z2 + 3.0f) + Q[16] * (p41 * x) + Q[17] * (p41 * y) +
      Q[18] * (p42 * x2y2) + Q[19] * (p42 * xy2) + Q[20] * (z * x3) + Q[21] * (z * y3) + Q[22] * x4 +
      Q[23] * y4;
  // d/dx
  // (x+iy)^m derivative: d/dx (Re,Im)_m = m (Re,Im)_{m-1};  d/dy (Re,Im)_m = m (-Im, Re)_{m-1}
  vx = P[1] + P[4] * z + P[6] * (2.0f * x) + P[7] * (2.0f * y) + P[9] * p31 + P[11] * (z * 2.0f * x) +
       P[12] * (z * 2.0f * y) + P[13] * (3.0f * x2y2) + P[14] * (3.0f * xy2) + P[16] * p41 +
       P[18] * (p42 * 2.0f * x) + P[19] * (p42 * 2.0f * y) + P[20] * (z * 3.0f * x2y2) +
       P[21] * (z * 3.0f * xy2) + P[22] * (4.0f * x3) + P[23] * (4.0f * y3);
  vy = P[2] + P[5] * z - P[6] * (2.0f * y) + P[7] * (2.0f * x) + P[10] * p31 - P[11] * (z * 2.0f * y) +
       P[12] * (z * 2.0f * x) - P[13] * (3.0f * xy2) + P[14] * (3.0f * x2y2) + P[17] * p41 -
       P[18] * (p42 * 2.0f * y) + P[19] * (p42 * 2.0f * x) - P[20] * (z * 3.0f * xy2) +
       P[21] * (z * 3.0f * x2y2) - P[22] * (4.0f * y3) + P[23] * (4.0f * x3);
  vz = P[0] + P[3] * (6.0f * z) + P[4] * x + P[5] * y + P[8] * (15.0f * z2 - 3.0f) +
       P[9] * (10.0f * z * x) + P[10] * (10.0f * z * y) + P[11] * x2y2 + P[12] * xy2 +
       P[15] * ((140.0f * z2 - 60.0f) * z) + P[16] * ((21.0f * z2 - 3.0f) * x) +
       P[17] * ((21.0f * z2 - 3.0f) * y) + P[18] * (14.0f * z * x2y2) + P[19] * (14.0f * z * xy2) +
       P[20] * x3 + P[21] * y3;
}

// 3-/4-/5-body invariants of one radial order from its 24 sums (find_q, nep_utilities.cuh:
// 1758-1770, 1859-1872).  q is strided by `stride` (= n_max_angular + 1).
template <int NT>
NEPMI_HD float cubic_value(const CubicTerm (&t)[NT], const float* s)
{
  float v = 0.0f;
#pragma unroll
  for (int a = 0; a < NT; ++a)
    v += t[a].w * s[t[a].i] * s[t[a].j] * s[t[a].k];
  return v;
}

// g[abc] += F d(row)/ds[abc]
template <int NT>
NEPMI_HD void cubic_gradient(const CubicTerm (&t)[NT], float F, const float* s, float* g)
{
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const float w = F * t[a].w;
    g[t[a].i] += w * s[t[a].j] * s[t[a].k];
    g[t[a].j] += w * s[t[a].i] * s[t[a].k];
    g[t[a].k] += w * s[t[a].i] * s[t[a].j];
  }
}

// The general form of find_q (:1819-1947) for the generic shape: l_max_3body = 1..4 (sums of higher L do not exist for
// the model: taken as zero), the 222 / 1111 rows right behind the 3-body rows, then the optional rows 112 / 123 / 233 /
// 134; up to 10 rows.
NEPMI_HD void invariants_generic(const ModelD& m, const float* s_in, float* q, int stride)
{
  const float C3B[kNumHarm] = NEPMI_C3B_INIT;
  const int nh = (m.Lmax + 1) * (m.Lmax + 1) - 1;
  float s[kNumHarm];
#pragma unroll
  for (int k = 0; k < kNumHarm; ++k)
    s[k] = k < nh ? s_in[k] : 0.0f;
#pragma unroll
  for (int L = 1; L <= 4; ++L) {
    const int st = L * L - 1;
    float acc = 0.0f;
#pragma unroll
    for (int k = 1; k < 2 * L + 1; ++k)
      acc += C3B[st + k] * s[st + k] * s[st + k];
    if (L <= m.Lmax)
      q[(L - 1) * stride] = 2.0f * acc + C3B[st] * s[st] * s[st];
  }
  int row = m.Lmax;
  if (m.has222)
    q[(row++) * stride] = NEPMI_C4B_0 * s[3] * s[3] * s[3] + NEPMI_C4B_1 * s[3] * (s[4] * s[4] + s[5] * s[5]) +
                          NEPMI_C4B_2 * s[3] * (s[6] * s[6] + s[7] * s[7]) +
                          NEPMI_C4B_3 * s[6] * (s[5] * s[5] - s[4] * s[4]) + NEPMI_C4B_4 * s[4] * s[5] * s[7];
  if (m.has1111) {
    const float s0 = s[0] * s[0], s12 = s[1] * s[1] + s[2] * s[2];
    q[(row++) * stride] = NEPMI_C5B_0 * s0 * s0 + NEPMI_C5B_1 * s0 * s12 + NEPMI_C5B_2 * s12 * s12;
  }
  const CubicTerm t112[] = NEPMI_Q112_TERMS;
  const CubicTerm t123[] = NEPMI_Q123_TERMS;
  const CubicTerm t233[] = NEPMI_Q233_TERMS;
  const CubicTerm t134[] = NEPMI_Q134_TERMS;
  if (m.extra & 1) q[(row++) * stride] = cubic_value(t112, s);
  if (m.extra & 2) q[(row++) * stride] = cubic_value(t123, s);
  if (m.extra & 4) q[(row++) * stride] = cubic_value(t233, s);
  if (m.extra & 8) q[(row++) * stride] = cubic_value(t134, s);
}

// Adjoint of invariants_generic: G[abc] = sum_rows Fp_row dq_row/ds_abc, in place of s (zero for L > l_max_3body).
NEPMI_HD void invariants_adjoint_generic(const ModelD& m, const float* fp, int stride, float* s)
{
  const float C3B[kNumHarm] = NEPMI_C3B_INIT;
  const int nh = (m.Lmax + 1) * (m.Lmax + 1) - 1;
  float g[kNumHarm];
#pragma unroll
  for (int k = 0; k < kNumHarm; ++k) {
    g[k] = 0.0f;
    if (k >= nh)
      s[k] = 0.0f;
  }
  int row = m.Lmax;
  if (m.has222) {
    const float F = fp[(row++) * stride];
    const float s0 = s[3], s1 = s[4], s2 = s[5], s3 = s[6], s4 = s[7];
    g[3] += F * (3.0f * NEPMI_C4B_0 * s0 * s0 + NEPMI_C4B_1 * (s1 * s1 + s2 * s2) + NEPMI_C4B_2 * (s3 * s3 + s4 * s4));
    g[4] += F * (2.0f * NEPMI_C4B_1 * s0 * s1 - 2.0f * NEPMI_C4B_3 * s3 * s1 + NEPMI_C4B_4 * s2 * s4);
    g[5] += F * (2.0f * NEPMI_C4B_1 * s0 * s2 + 2.0f * NEPMI_C4B_3 * s3 * s2 + NEPMI_C4B_4 * s1 * s4);
    g[6] += F * (2.0f * NEPMI_C4B_2 * s0 * s3 + NEPMI_C4B_3 * (s2 * s2 - s1 * s1));
    g[7] += F * (2.0f * NEPMI_C4B_2 * s0 * s4 + NEPMI_C4B_4 * s1 * s2);
  }
  if (m.has1111) {
    const float F = fp[(row++) * stride];
    const float s0 = s[0], s1 = s[1], s2 = s[2];
    const float s12 = s1 * s1 + s2 * s2;
    g[0] += F * (4.0f * NEPMI_C5B_0 * s0 * s0 * s0 + 2.0f * NEPMI_C5B_1 * s0 * s12);
    g[1] += F * (2.0f * NEPMI_C5B_1 * s0 * s0 * s1 + 4.0f * NEPMI_C5B_2 * s12 * s1);
    g[2] += F * (2.0f * NEPMI_C5B_1 * s0 * s0 * s2 + 4.0f * NEPMI_C5B_2 * s12 * s2);
  }
  const CubicTerm t112[] = NEPMI_Q112_TERMS;
  const CubicTerm t123[] = NEPMI_Q123_TERMS;
  const CubicTerm t233[] = NEPMI_Q233_TERMS;
  const CubicTerm t134[] = NEPMI_Q134_TERMS;
  if (m.extra & 1) cubic_gradient(t112, fp[(row++) * stride], s, g);
  if (m.extra & 2) cubic_gradient(t123, fp[(row++) * stride], s, g);
  if (m.extra & 4) cubic_gradient(t233, fp[(row++) * stride], s, g);
  if (m.extra & 8) cubic_gradient(t134, fp[(row++) * stride], s, g);
#pragma unroll
  for (int L = 1; L <= 4; ++L) {
    const int st = L * L - 1;
    const float F = L <= m.Lmax ? fp[(L - 1) * stride] : 0.0f;
    s[st] = 2.0f * C3B[st] * F * s[st];
#pragma unroll
    for (int k = 1; k < 2 * L + 1; ++k)
      s[st + k] = 4.0f * C3B[st + k] * F * s[st + k];
  }
#pragma unroll
  for (int k = 0; k < kNumHarm; ++k)
    s[k] = k < nh ? s[k] + g[k] : 0.0f;
}

// The shapes of the shipped models (l_max 4, rows 222 / 1111 only) keep the form below with constant row indices;
// EXTRAS (the generic shape) takes the general form above.
template <bool EXTRAS = false>
NEPMI_HD void invariants(const ModelD& m, const float* s, float* q, int stride)
{
  if (EXTRAS) {
    invariants_generic(m, s, q, stride);
    return;
  }
  const float C3B[kNumHarm] = NEPMI_C3B_INIT;
#pragma unroll
  for (int L = 1; L <= 4; ++L) {
    const int st = L * L - 1;
    float acc = 0.0f;
#pragma unroll
    for (int k = 1; k < 2 * L + 1; ++k)
      acc += C3B[st + k] * s[st + k] * s[st + k];
    q[(L - 1) * stride] = 2.0f * acc + C3B[st] * s[st] * s[st];
  }
  // the 4-body row sits at index 4, the 5-body row behind it (index 4 when there is no 4-body row);
  // written with constant indices so that q stays in registers
  const float q4b = NEPMI_C4B_0 * s[3] * s[3] * s[3] + NEPMI_C4B_1 * s[3] * (s[4] * s[4] + s[5] * s[5]) +
                    NEPMI_C4B_2 * s[3] * (s[6] * s[6] + s[7] * s[7]) +
                    NEPMI_C4B_3 * s[6] * (s[5] * s[5] - s[4] * s[4]) + NEPMI_C4B_4 * s[4] * s[5] * s[7];
  const float s0 = s[0] * s[0], s12 = s[1] * s[1] + s[2] * s[2];
  const float q5b = NEPMI_C5B_0 * s0 * s0 + NEPMI_C5B_1 * s0 * s12 + NEPMI_C5B_2 * s12 * s12;
  if (m.has222) {
    q[4 * stride] = q4b;
    if (m.has1111)
      q[5 * stride] = q5b;
  } else if (m.has1111) {
    q[4 * stride] = q5b;
  }
}

// Adjoint of `invariants`: G[abc] = sum_L Fp_L dq_L/ds_abc (+ 4-/5-body), in place of s.
// fp is strided like q.
template <bool EXTRAS = false>
NEPMI_HD void invariants_adjoint(const ModelD& m, const float* fp, int stride, float* s /* in: s, out: G */)
{
  const float C3B[kNumHarm] = NEPMI_C3B_INIT;
  if (EXTRAS) {
    invariants_adjoint_generic(m, fp, stride, s);
    return;
  }
  float g4[5] = {0, 0, 0, 0, 0}, g5[3] = {0, 0, 0};
  if (m.has222) {
    const float F = fp[4 * stride];
    const float s0 = s[3], s1 = s[4], s2 = s[5], s3 = s[6], s4 = s[7];
    g4[0] = F * (3.0f * NEPMI_C4B_0 * s0 * s0 + NEPMI_C4B_1 * (s1 * s1 + s2 * s2) + NEPMI_C4B_2 * (s3 * s3 + s4 * s4));
    g4[1] = F * (2.0f * NEPMI_C4B_1 * s0 * s1 - 2.0f * NEPMI_C4B_3 * s3 * s1 + NEPMI_C4B_4 * s2 * s4);
    g4[2] = F * (2.0f * NEPMI_C4B_1 * s0 * s2 + 2.0f * NEPMI_C4B_3 * s3 * s2 + NEPMI_C4B_4 * s1 * s4);
    g4[3] = F * (2.0f * NEPMI_C4B_2 * s0 * s3 + NEPMI_C4B_3 * (s2 * s2 - s1 * s1));
    g4[4] = F * (2.0f * NEPMI_C4B_2 * s0 * s4 + NEPMI_C4B_4 * s1 * s2);
  }
  if (m.has1111) {
    const float F = m.has222 ? fp[5 * stride] : fp[4 * stride]; // constant indices: fp stays in registers
    const float s0 = s[0], s1 = s[1], s2 = s[2];
    const float s12 = s1 * s1 + s2 * s2;
    g5[0] = F * (4.0f * NEPMI_C5B_0 * s0 * s0 * s0 + 2.0f * NEPMI_C5B_1 * s0 * s12);
    g5[1] = F * (2.0f * NEPMI_C5B_1 * s0 * s0 * s1 + 4.0f * NEPMI_C5B_2 * s12 * s1);
    g5[2] = F * (2.0f * NEPMI_C5B_1 * s0 * s0 * s2 + 4.0f * NEPMI_C5B_2 * s12 * s2);
  }
#pragma unroll
  for (int L = 1; L <= 4; ++L) {
    const int st = L * L - 1;
    const float F = fp[(L - 1) * stride];
    s[st] = 2.0f * C3B[st] * F * s[st];
#pragma unroll
    for (int k = 1; k < 2 * L + 1; ++k)
      s[st + k] = 4.0f * C3B[st + k] * F * s[st + k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    s[3 + k] += g4[k];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    s[k] += g5[k];
}

// find_f_and_fp_zbl, nep_utilities.cuh:433-508.  para10 == nullptr: universal ZBL.
NEPMI_HD void zbl_pair(
  const float* para10, float zizj, float a_inv, float rc_in, float rc_out, float d, float dinv, float& f,
  float& fp)
{
  const float U[8] = {0.18175f, 3.1998f, 0.50986f, 0.94229f, 0.28022f, 0.4029f, 0.02817f, 0.20162f};
  const float x = d * a_inv;
  float phi = 0.0f, phip = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = para10 ? para10[2 + 2 * k] : U[2 * k];
    const float b = para10 ? para10[3 + 2 * k] : U[2 * k + 1];
    const float t = a * expf(-b * x);
    phi += t;
    phip -= b * t;
  }
  phi *= zizj;
  phip *= zizj * a_inv;
  phip = phip * dinv - phi * dinv * dinv;
  phi *= dinv;
  const float r1 = para10 ? para10[0] : rc_in;
  const float r2 = para10 ? para10[1] : rc_out;
  float fc, fcp;
  if (d < r1) {
    fc = 1.0f;
    fcp = 0.0f;
  } else if (d < r2) {
    const float pf = NEPMI_PI / (r2 - r1);
    fc = cosf(pf * (d - r1)) * 0.5f + 0.5f;
    fcp = -sinf(pf * (d - r1)) * pf * 0.5f;
  } else {
    fc = 0.0f;
    fcp = 0.0f;
  }
  fp = phip * fc + phi * fcp;
  f = phi * fc;
}

} // namespace nepmi
