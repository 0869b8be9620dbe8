/* TEST INFRASTRUCTURE ONLY -- see nep_oracle.h.  Plain-C restatement of the reference's NEP
 * force path, built by oracle/Makefile into oracle/libnep_oracle.so.  Never linked into or
 * loaded by the product (libnepmi.so / gpumd_amd). */
#define _GNU_SOURCE
#include "nep_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NEPO_LMAX 8
/* The per-atom sweeps are shared among the host's cores (OpenMP) from this size on: a team of hundreds of threads costs more
 * than the work of a small cell (the parity cases of a few thousand atoms run thousands of sweeps).  The results do not
 * depend on the thread count: every atom's arithmetic is that of the serial sweep. */
#define NEPO_OMP_MIN_ATOMS 30000
#define NEPO_NABC 80 /* (L_max+1)^2 - 1 for L_max = 8 (NUM_OF_ABC, nep_utilities.cuh:18) */
#define NEPO_MAX_BASIS 20
#define NEPO_MAX_TYPES 94

/* Normalisation constants of the angular invariants (nep_utilities.cuh:18-46): numerical
 * constants of the NEP model definition (C3B[lm] = N_lm^2 of the real harmonics, e.g.
 * 3/(4 pi), 3/(8 pi), 5/(16 pi), 15/(8 pi), 15/(32 pi), ...; C4B/C5B the Clebsch-Gordan
 * contractions for (222) and (1111)). */
/* sums 0..23 (l <= 4) as they stand in the reference's table; 24..79 (l = 5..8) and Z for every l from their definition
 * (addition theorem), generated in exact arithmetic by gpumd_amd/csrc/tools/gen_highl_tables.py; the generated l <= 4
 * values agree with the 24 below to 1e-14 (NEPO_C3B_LOW_CHECK, tests/test_high_l.py) and the whole set is pinned
 * against NEP_CPU by the tests */
#include "nep_highl_tables.inc"
static const double NEPO_C3B[NEPO_NABC] = {
  0.238732414637843, 0.119366207318922, 0.119366207318922, 0.099471839432435, 0.596831036594608,
  0.596831036594608, 0.149207759148652, 0.149207759148652, 0.139260575205408, 0.104445431404056,
  0.104445431404056, 1.044454314040563, 1.044454314040563, 0.174075719006761, 0.174075719006761,
  0.011190581936149, 0.223811638722978, 0.223811638722978, 0.111905819361489, 0.111905819361489,
  1.566681471060845, 1.566681471060845, 0.195835183882606, 0.195835183882606,
  NEPO_C3B_HIGH_LIST};
static const double NEPO_C4B[5] = {
  -0.007499480826664, -0.134990654879954, 0.067495327439977, 0.404971964639861, -0.809943929279723};
#include "nep_invariants_extra.inc"
static const double NEPO_C5B[3] = {0.026596810706114, 0.053193621412227, 0.026596810706114};

/* NEPO_Z[L][n1][n2]: coefficient of z^n2 in the factor that multiplies Re/Im (x+iy)^n1 (nep_utilities.cuh:87-141): generated above */

static const char* NEPO_ELEMENTS[NEPO_MAX_TYPES] = {
  "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",
  "Cl", "Ar", "K",  "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge",
  "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",  "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd",
  "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd",
  "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au", "Hg",
  "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu"};

struct nepo_model {
  int version;
  int zbl_enabled, zbl_flexible;
  int zbl_typewise;
  double zbl_typewise_factor;
  double zbl_rc_inner, zbl_rc_outer;
  int num_types;
  char symbols[NEPO_MAX_TYPES][4];
  int atomic_numbers[NEPO_MAX_TYPES];
  double rc_radial[NEPO_MAX_TYPES], rc_angular[NEPO_MAX_TYPES];
  double rc_radial_max, rc_angular_max;
  int MN_radial, MN_angular;
  int n_max_radial, n_max_angular, basis_size_radial, basis_size_angular;
  int L_max, has_222, has_1111, num_L, dim, num_neurons;
  int has_112, has_123, has_233, has_134; /* extra 4-body rows, nep.cu:275-310 */
  int temperature_model; /* nep4[_zbl]_temperature (nep.cu:125-130, model_type 3): the last ANN input is the temperature */
  double temperature;    /* what Force::compute hands to NEP::compute(temperature, ...) (force.cu:516-525) */
  int num_para_ann, num_para, num_c_radial;
  int off_w0[NEPO_MAX_TYPES], off_b0[NEPO_MAX_TYPES], off_w1[NEPO_MAX_TYPES], off_b1;
  double* params; /* num_para + dim (q_scaler at the end) */
  double zbl_para[550];
};

typedef struct {
  double h[18];
  float hf[18];
  int pbc[3];
  int is_orthogonal;
  double thickness[3];
  double volume;
} nepo_box;

/* Box::get_inverse, box.cu:52-78 */
static void nepo_invert_box(double* h)
{
  h[9] = h[4] * h[8] - h[5] * h[7];
  h[10] = h[2] * h[7] - h[1] * h[8];
  h[11] = h[1] * h[5] - h[2] * h[4];
  h[12] = h[5] * h[6] - h[3] * h[8];
  h[13] = h[0] * h[8] - h[2] * h[6];
  h[14] = h[2] * h[3] - h[0] * h[5];
  h[15] = h[3] * h[7] - h[4] * h[6];
  h[16] = h[1] * h[6] - h[0] * h[7];
  h[17] = h[0] * h[4] - h[1] * h[3];
  double det = h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) +
               h[2] * (h[3] * h[7] - h[4] * h[6]);
  for (int k = 9; k < 18; ++k)
    h[k] /= det;
}

static double nepo_cross_norm(const double* a, const double* b)
{
  double s1 = a[1] * b[2] - a[2] * b[1];
  double s2 = a[2] * b[0] - a[0] * b[2];
  double s3 = a[0] * b[1] - a[1] * b[0];
  return sqrt(s1 * s1 + s2 * s2 + s3 * s3);
}

/* Box::get_volume/get_area/get_num_bins/set_is_orthogonal, box.cu:23-117 */
static void nepo_box_init(nepo_box* b, const double h9[9], const int pbc[3])
{
  memcpy(b->h, h9, sizeof(double) * 9);
  nepo_invert_box(b->h);
  for (int k = 0; k < 18; ++k)
    b->hf[k] = (float)b->h[k];
  for (int d = 0; d < 3; ++d)
    b->pbc[d] = pbc[d];
  const double* h = b->h;
  b->is_orthogonal = h[1] == 0 && h[2] == 0 && h[3] == 0 && h[5] == 0 && h[6] == 0 && h[7] == 0;
  b->volume = fabs(
    h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) +
    h[2] * (h[3] * h[7] - h[4] * h[6]));
  double a[3] = {h[0], h[3], h[6]}, bb[3] = {h[1], h[4], h[7]}, c[3] = {h[2], h[5], h[8]};
  b->thickness[0] = b->volume / nepo_cross_norm(bb, c);
  b->thickness[1] = b->volume / nepo_cross_norm(c, a);
  b->thickness[2] = b->volume / nepo_cross_norm(a, bb);
}

/* ---- nep.txt parser: NEP::NEP, nep.cu:100-395; nep3 header and shared ANN as in the vendored
 * NEP_CPU (nep.cpp:2568-2872). ------------------------------------------------------------ */

static int nepo_tokens(char* line, char** tok, int maxtok)
{
  int n = 0;
  char* save = NULL;
  for (char* t = strtok_r(line, " \t\r\n", &save); t && n < maxtok; t = strtok_r(NULL, " \t\r\n", &save))
    tok[n++] = t;
  return n;
}

#define NEPO_FAIL(...)                          \
  do {                                          \
    if (err)                                    \
      snprintf(err, errlen, __VA_ARGS__);       \
    if (fp)                                     \
      fclose(fp);                               \
    nepo_model_free(m);                         \
    return NULL;                                \
  } while (0)

nepo_model* nepo_model_load(const char* path, char* err, int errlen)
{
  nepo_model* m = (nepo_model*)calloc(1, sizeof(nepo_model));
  FILE* fp = fopen(path, "r");
  char line[4096];
  char* tok[256];
  if (!fp)
    NEPO_FAIL("Failed to open %s", path);

  if (!fgets(line, sizeof line, fp))
    NEPO_FAIL("empty file");
  int nt = nepo_tokens(line, tok, 256);
  if (nt < 3)
    NEPO_FAIL("The first line of nep.txt should have at least 3 items.");
  if (!strcmp(tok[0], "nep3")) { m->version = 3; }
  else if (!strcmp(tok[0], "nep3_zbl")) { m->version = 3; m->zbl_enabled = 1; }
  else if (!strcmp(tok[0], "nep4")) { m->version = 4; }
  else if (!strcmp(tok[0], "nep4_zbl")) { m->version = 4; m->zbl_enabled = 1; }
  else if (!strcmp(tok[0], "nep5")) { m->version = 5; }
  else if (!strcmp(tok[0], "nep5_zbl")) { m->version = 5; m->zbl_enabled = 1; }
  else if (!strcmp(tok[0], "nep4_temperature")) { m->version = 4; m->temperature_model = 1; }
  else if (!strcmp(tok[0], "nep4_zbl_temperature")) { m->version = 4; m->zbl_enabled = 1; m->temperature_model = 1; }
  else NEPO_FAIL("%s is an unsupported NEP model.", tok[0]);
  m->num_types = atoi(tok[1]);
  if (m->num_types < 1 || m->num_types > NEPO_MAX_TYPES || nt != 2 + m->num_types)
    NEPO_FAIL("The first line of nep.txt should have %d atom symbols.", m->num_types);
  for (int t = 0; t < m->num_types; ++t) {
    strncpy(m->symbols[t], tok[2 + t], 3);
    m->atomic_numbers[t] = 0;
    for (int e = 0; e < NEPO_MAX_TYPES; ++e)
      if (!strcmp(tok[2 + t], NEPO_ELEMENTS[e]))
        m->atomic_numbers[t] = e + 1;
  }

  if (m->zbl_enabled) {
    if (!fgets(line, sizeof line, fp))
      NEPO_FAIL("missing zbl line");
    nt = nepo_tokens(line, tok, 256);
    if (nt != 3 && nt != 4)
      NEPO_FAIL("This line should be zbl rc_inner rc_outer [zbl_factor].");
    m->zbl_rc_inner = atof(tok[1]);
    m->zbl_rc_outer = atof(tok[2]);
    if (m->zbl_rc_inner == 0 && m->zbl_rc_outer == 0)
      m->zbl_flexible = 1;
    else if (nt == 4) { /* universal ZBL with a type-wise outer cutoff, nep.cu:183-186 */
      m->zbl_typewise = 1;
      m->zbl_typewise_factor = atof(tok[3]);
    }
  }

  if (!fgets(line, sizeof line, fp))
    NEPO_FAIL("missing cutoff line");
  nt = nepo_tokens(line, tok, 256);
  if (nt != 5 && nt != m->num_types * 2 + 3)
    NEPO_FAIL("cutoff should have 4 or num_types * 2 + 2 parameters.");
  if (nt == 5) {
    for (int t = 0; t < m->num_types; ++t) {
      m->rc_radial[t] = atof(tok[1]);
      m->rc_angular[t] = atof(tok[2]);
    }
  } else {
    for (int t = 0; t < m->num_types; ++t) {
      m->rc_radial[t] = atof(tok[1 + 2 * t]);
      m->rc_angular[t] = atof(tok[2 + 2 * t]);
    }
  }
  for (int t = 0; t < m->num_types; ++t) {
    if (m->rc_radial[t] > m->rc_radial_max) m->rc_radial_max = m->rc_radial[t];
    if (m->rc_angular[t] > m->rc_angular_max) m->rc_angular_max = m->rc_angular[t];
  }
  m->MN_radial = (int)ceil(atoi(tok[nt - 2]) * 1.25);
  m->MN_angular = (int)ceil(atoi(tok[nt - 1]) * 1.25);

  if (!fgets(line, sizeof line, fp) || nepo_tokens(line, tok, 256) != 3)
    NEPO_FAIL("This line should be n_max n_max_radial n_max_angular.");
  m->n_max_radial = atoi(tok[1]);
  m->n_max_angular = atoi(tok[2]);
  if (!fgets(line, sizeof line, fp) || nepo_tokens(line, tok, 256) != 3)
    NEPO_FAIL("This line should be basis_size basis_size_radial basis_size_angular.");
  m->basis_size_radial = atoi(tok[1]);
  m->basis_size_angular = atoi(tok[2]);
  if (m->basis_size_radial + 1 > NEPO_MAX_BASIS || m->basis_size_angular + 1 > NEPO_MAX_BASIS)
    NEPO_FAIL("basis size too large");
  if (!fgets(line, sizeof line, fp) || (nt = nepo_tokens(line, tok, 256)) < 4)
    NEPO_FAIL("This line should be l_max l_max_3body has_q_222 has_q_1111.");
  m->L_max = atoi(tok[1]);
  m->has_222 = atoi(tok[2]);
  m->has_1111 = atoi(tok[3]);
  m->has_112 = nt >= 5 ? atoi(tok[4]) : 0; /* optional trailing flags, nep.cu:275-286 */
  m->has_123 = nt >= 6 ? atoi(tok[5]) : 0;
  m->has_233 = nt >= 7 ? atoi(tok[6]) : 0;
  m->has_134 = nt >= 8 ? atoi(tok[7]) : 0;
  if (m->L_max < 1 || m->L_max > NEPO_LMAX)
    NEPO_FAIL("l_max must be 1..%d in the oracle", NEPO_LMAX);
  if ((m->has_222 && m->L_max < 2) || (m->has_112 && m->L_max < 2) || ((m->has_123 || m->has_233) && m->L_max < 3) ||
      (m->has_134 && m->L_max < 4))
    NEPO_FAIL("a 4-body row needs sums of a higher l than l_max_3body provides");
  m->num_L = m->L_max + (m->has_222 ? 1 : 0) + (m->has_1111 ? 1 : 0) + (m->has_112 ? 1 : 0) + (m->has_123 ? 1 : 0) +
             (m->has_233 ? 1 : 0) + (m->has_134 ? 1 : 0);
  if (!fgets(line, sizeof line, fp) || nepo_tokens(line, tok, 256) != 3)
    NEPO_FAIL("This line should be ANN num_neurons 0.");
  m->num_neurons = atoi(tok[1]);
  m->dim = (m->n_max_radial + 1) + (m->n_max_angular + 1) * m->num_L;
  if (m->temperature_model)
    m->dim += 1; /* nep.cu:322-325 */

  const int T = m->num_types;
  const int per_type = (m->dim + 2) * m->num_neurons;
  if (m->version == 3)
    m->num_para_ann = per_type + 1;
  else if (m->version == 4)
    m->num_para_ann = per_type * T + 1;
  else
    m->num_para_ann = (per_type + 1) * T + 1;
  const int num_desc = T * T * ((m->n_max_radial + 1) * (m->basis_size_radial + 1) +
                                 (m->n_max_angular + 1) * (m->basis_size_angular + 1));
  m->num_para = m->num_para_ann + num_desc;
  m->num_c_radial = T * T * (m->n_max_radial + 1) * (m->basis_size_radial + 1);
  m->params = (double*)malloc(sizeof(double) * (m->num_para + m->dim));
  for (int k = 0; k < m->num_para + m->dim; ++k) {
    if (!fgets(line, sizeof line, fp) || nepo_tokens(line, tok, 256) < 1)
      NEPO_FAIL("nep.txt ends after %d of %d parameters", k, m->num_para + m->dim);
    m->params[k] = atof(tok[0]);
  }
  /* NEP::update_potential, nep.cu:402-434; nep3 aliases one block for all types */
  int off = 0;
  for (int t = 0; t < T; ++t) {
    if (t > 0 && m->version == 3)
      off = 0;
    m->off_w0[t] = off; off += m->num_neurons * m->dim;
    m->off_b0[t] = off; off += m->num_neurons;
    m->off_w1[t] = off; off += m->num_neurons;
    if (m->version == 5)
      off += 1;
  }
  m->off_b1 = off;
  if (off + 1 != m->num_para_ann)
    NEPO_FAIL("internal: ANN layout mismatch");
  if (m->zbl_flexible) {
    int nz = 10 * (T * (T + 1)) / 2;
    for (int k = 0; k < nz; ++k) {
      if (!fgets(line, sizeof line, fp) || nepo_tokens(line, tok, 256) < 1)
        NEPO_FAIL("missing flexible-ZBL parameters");
      m->zbl_para[k] = atof(tok[0]);
    }
  }
  fclose(fp);
  return m;
}

void nepo_model_free(nepo_model* m)
{
  if (!m)
    return;
  free(m->params);
  free(m);
}

void nepo_model_info(const nepo_model* m, nepo_info* o)
{
  o->version = m->version;
  o->num_types = m->num_types;
  o->zbl_enabled = m->zbl_enabled;
  o->zbl_flexible = m->zbl_flexible;
  o->rc_radial_max = m->rc_radial_max;
  o->rc_angular_max = m->rc_angular_max;
  o->n_max_radial = m->n_max_radial;
  o->n_max_angular = m->n_max_angular;
  o->basis_size_radial = m->basis_size_radial;
  o->basis_size_angular = m->basis_size_angular;
  o->L_max = m->L_max;
  o->has_222 = m->has_222;
  o->has_1111 = m->has_1111;
  o->num_L = m->num_L;
  o->dim = m->dim;
  o->num_neurons = m->num_neurons;
  o->MN_radial = m->MN_radial;
  o->MN_angular = m->MN_angular;
  o->num_para = m->num_para;
}

const char* nepo_model_symbol(const nepo_model* m, int t) { return m->symbols[t]; }
int nepo_model_is_temperature(const nepo_model* m) { return m->temperature_model; }
void nepo_model_set_temperature(nepo_model* m, double temperature) { m->temperature = temperature; }
double nepo_model_param(const nepo_model* m, int idx) { return m->params[idx]; }

/* ---- the two precision instantiations ---------------------------------------------------- */

/* covalent radii (Angstrom) by atomic number: the constant table of src/utilities/nep_utilities.cuh:143-154 */
static const float nepo_covalent_radius[94] = {
  0.426667f, 0.613333f, 1.6f, 1.25333f, 1.02667f, 1.0f, 0.946667f, 0.84f,
  0.853333f, 0.893333f, 1.86667f, 1.66667f, 1.50667f, 1.38667f, 1.46667f, 1.36f,
  1.32f, 1.28f, 2.34667f, 2.05333f, 1.77333f, 1.62667f, 1.61333f, 1.46667f,
  1.42667f, 1.38667f, 1.33333f, 1.32f, 1.34667f, 1.45333f, 1.49333f, 1.45333f,
  1.53333f, 1.46667f, 1.52f, 1.56f, 2.52f, 2.22667f, 1.96f, 1.85333f,
  1.76f, 1.65333f, 1.53333f, 1.50667f, 1.50667f, 1.44f, 1.53333f, 1.64f,
  1.70667f, 1.68f, 1.68f, 1.64f, 1.76f, 1.74667f, 2.78667f, 2.34667f,
  2.16f, 1.96f, 2.10667f, 2.09333f, 2.08f, 2.06667f, 2.01333f, 2.02667f,
  2.01333f, 2.0f, 1.98667f, 1.98667f, 1.97333f, 2.04f, 1.94667f, 1.82667f,
  1.74667f, 1.64f, 1.57333f, 1.54667f, 1.48f, 1.49333f, 1.50667f, 1.76f,
  1.73333f, 1.73333f, 1.81333f, 1.74667f, 1.84f, 1.89333f, 2.68f, 2.41333f,
  2.22667f, 2.10667f, 2.02667f, 2.04f, 2.05333f, 2.06667f};

#define REAL float
#define SFX(name) name##_f32
#define NEPO_PI_F 3.1415927f
#define NEPO_HALF_PI_F 1.5707963f
#define NEPO_LIT(x) ((float)(x##f))
#include "nep_oracle_core.inc"
#undef REAL
#undef SFX
#undef NEPO_PI_F
#undef NEPO_HALF_PI_F
#undef NEPO_LIT

#define REAL double
#define SFX(name) name##_f64
#define NEPO_PI_F 3.14159265358979323846
#define NEPO_HALF_PI_F 1.57079632679489661923
#define NEPO_LIT(x) ((double)(x))
#include "nep_oracle_core.inc"
#undef REAL
#undef SFX
#undef NEPO_PI_F
#undef NEPO_HALF_PI_F
#undef NEPO_LIT

/* ---- public API -------------------------------------------------------------------------- */

/* The angular rows on their own (FP64), so that tests/test_oracle_golden.py can hold them against the reference's
 * find_q / accumulate_f12 (oracle/ref_utils_wrap.cpp).  has[6] = has_q_222, _1111, _112, _123, _233, _134. */
static void nepo_rows_model(nepo_model* m, int L_max, const int has[6])
{
  memset(m, 0, sizeof(*m));
  m->L_max = L_max;
  m->has_222 = has[0]; m->has_1111 = has[1]; m->has_112 = has[2];
  m->has_123 = has[3]; m->has_233 = has[4]; m->has_134 = has[5];
  m->num_L = L_max;
  for (int k = 0; k < 6; ++k)
    m->num_L += has[k] ? 1 : 0;
}

void nepo_rows_find_q(int L_max, const int has[6], int nA1, int n, const double* s, double* q)
{
  nepo_model m;
  nepo_rows_model(&m, L_max, has);
  find_q_f64(&m, nA1, n, s, q);
}

void nepo_rows_accumulate_f12(
  int L_max, const int has[6], int n, int nA1, double d12, const double* r12, double fn, double fnp, const double* Fp,
  const double* sums_n, double* f12)
{
  nepo_model m;
  nepo_rows_model(&m, L_max, has);
  accumulate_f12_f64(&m, n, nA1, d12, r12, fn, fnp, Fp, sums_n, f12);
}


struct nepo_lists {
  int n;
  int path;
  plist_f32 skin, rad, ang;
};

nepo_lists* nepo_lists_build(
  const nepo_model* m, int n, const int* type, const double h[9], const int pbc[3],
  const double* pos, int path)
{
  nepo_box box;
  nepo_box_init(&box, h, pbc);
  nepo_lists* l = (nepo_lists*)calloc(1, sizeof(nepo_lists));
  l->n = n;
  double* pe = (double*)malloc(sizeof(double) * n);
  double* f = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  double* v = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  int st = compute_f32(m, &box, path, n, type, pos, pe, f, v, NULL, NULL, &l->rad, &l->ang, &l->skin, &l->path);
  free(pe); free(f); free(v);
  if (st == -2) {
    free(l);
    return NULL;
  }
  return l;
}

int nepo_lists_path(const nepo_lists* l) { return l->path; }

int nepo_lists_get(const nepo_lists* l, int which, int* nn, int* nl, int ld)
{
  const plist_f32* p = which == 0 ? &l->skin : which == 1 ? &l->rad : &l->ang;
  if (!p->nn)
    return -1;
  int mx = 0;
  for (int i = 0; i < l->n; ++i) {
    if (nn)
      nn[i] = p->nn[i];
    if (p->nn[i] > mx)
      mx = p->nn[i];
    if (nl)
      for (int s = 0; s < p->nn[i] && s < ld; ++s)
        nl[(size_t)s * l->n + i] = p->j[(size_t)i * p->cap + s];
  }
  return mx;
}

void nepo_lists_free(nepo_lists* l)
{
  if (!l)
    return;
  if (l->skin.nn) plist_free_f32(&l->skin);
  if (l->rad.nn) plist_free_f32(&l->rad);
  if (l->ang.nn) plist_free_f32(&l->ang);
  free(l);
}

int nepo_compute(
  const nepo_model* m, int precision, int path, int n, const int* type, const double h[9],
  const int pbc[3], const double* pos, double* pe, double* force, double* virial, double* q_out,
  double* fp_out)
{
  nepo_box box;
  nepo_box_init(&box, h, pbc);
  if (precision == 32)
    return compute_f32(m, &box, path, n, type, pos, pe, force, virial, q_out, fp_out, NULL, NULL, NULL, NULL);
  return compute_f64(m, &box, path, n, type, pos, pe, force, virial, q_out, fp_out, NULL, NULL, NULL, NULL);
}

/* gpu_apply_pbc, force.cu:424-459 */
void nepo_apply_pbc(int n, const double h9[9], const int pbc[3], double* pos)
{
  nepo_box box;
  nepo_box_init(&box, h9, pbc);
  const double* h = box.h;
  for (int i = 0; i < n; ++i) {
    double x = pos[i], y = pos[n + i], z = pos[2 * (size_t)n + i];
    double s[3] = {
      h[9] * x + h[10] * y + h[11] * z, h[12] * x + h[13] * y + h[14] * z,
      h[15] * x + h[16] * y + h[17] * z};
    for (int d = 0; d < 3; ++d)
      if (pbc[d]) {
        if (s[d] < 0.0) s[d] += 1.0;
        else if (s[d] > 1.0) s[d] -= 1.0;
      }
    pos[i] = h[0] * s[0] + h[1] * s[1] + h[2] * s[2];
    pos[n + i] = h[3] * s[0] + h[4] * s[1] + h[5] * s[2];
    pos[2 * (size_t)n + i] = h[6] * s[0] + h[7] * s[1] + h[8] * s[2];
  }
}

/* gpu_velocity_verlet, ensemble.cu:176-214 */
void nepo_velocity_verlet(
  int is_step1, int n, double dt, const double* mass, const double* force, double* pos, double* vel)
{
  const double half = dt * 0.5;
  for (int i = 0; i < n; ++i) {
    const double minv = 1.0 / mass[i];
    for (int d = 0; d < 3; ++d) {
      double v = vel[(size_t)d * n + i];
      v += force[(size_t)d * n + i] * minv * half;
      vel[(size_t)d * n + i] = v;
      if (is_step1)
        pos[(size_t)d * n + i] += v * dt;
    }
  }
}

/* gpu_find_thermo_instant_temperature, ensemble.cu:434-633.  The reference reduces with 1024
 * strided partial sums and a binary tree; the same order is used here so that the f64 result
 * is reproduced exactly. */
static double nepo_tree_sum(int n, const double* a, const double* b, const double* c, const double* m, int mode)
{
  double part[1024];
  for (int t = 0; t < 1024; ++t) {
    double acc = 0.0;
    for (int i = t; i < n; i += 1024) {
      switch (mode) {
        case 0: acc += (a[i] * a[i] + b[i] * b[i] + c[i] * c[i]) * m[i]; break; /* m v^2 */
        case 1: acc += a[i]; break;                                             /* plain */
        default: acc += a[i] + b[i] * c[i] * m[i]; break;                       /* W + m v v */
      }
    }
    part[t] = acc;
  }
  for (int off = 512; off > 0; off >>= 1)
    for (int t = 0; t < off; ++t)
      part[t] += part[t + off];
  return part[0];
}

void nepo_thermo(
  int n, double volume, const double* mass, const double* pe, const double* vel,
  const double* virial, double* th)
{
  const double kB = 8.617343e-5;
  const double *vx = vel, *vy = vel + n, *vz = vel + 2 * (size_t)n;
  th[0] = nepo_tree_sum(n, vx, vy, vz, mass, 0) / (3.0 * n * kB);
  th[1] = nepo_tree_sum(n, pe, NULL, NULL, NULL, 1);
  th[2] = nepo_tree_sum(n, virial + 0 * (size_t)n, vx, vx, mass, 2) / volume;
  th[3] = nepo_tree_sum(n, virial + 1 * (size_t)n, vy, vy, mass, 2) / volume;
  th[4] = nepo_tree_sum(n, virial + 2 * (size_t)n, vz, vz, mass, 2) / volume;
  th[5] = nepo_tree_sum(n, virial + 3 * (size_t)n, vx, vy, mass, 2) / volume;
  th[6] = nepo_tree_sum(n, virial + 4 * (size_t)n, vx, vz, mass, 2) / volume;
  th[7] = nepo_tree_sum(n, virial + 5 * (size_t)n, vy, vz, mass, 2) / volume;
}

/* Nose-Hoover chain of 4 links: Ensemble_NHC constructor (ensemble_nhc.cu:30-49) and the host-side
 * chain integrator nhc() (ensemble_nhc.cu:102-164): Suzuki-Yoshida weights (7) x 4 sub-steps.
 * st = eta[4] | p_eta[4] | Q[4].  Returns the velocity scale factor. */
void nepo_nhc_init(int n, double temperature, double t_coup, double dt, double* st)
{
  const double kB = 8.617343e-5;
  const double tau = dt * t_coup;
  for (int m = 0; m < 4; ++m) {
    st[m] = 0.0;
    st[4 + m] = (m % 2 == 0) ? 1.0 : -1.0;
    st[8 + m] = kB * temperature * tau * tau;
  }
  st[8] *= 3.0 * n;
}

static double nhc_drive(const double* p, const double* Q, int m, double ek2, double dN, double kT)
{
  /* generalised force on link m: link 0 is driven by the particles' kinetic energy */
  return m == 0 ? ek2 - dN * kT : p[m - 1] * p[m - 1] / Q[m - 1] - kT;
}

double nepo_nhc(double* st, double ek2, double kT, double dN, double dt2_particle)
{
  static const double w[7] = {0.784513610477560, 0.235573213359357, -1.17767998417887, 1.31518632068391,
                              -1.17767998417887, 0.235573213359357, 0.784513610477560};
  double *eta = st, *p = st + 4, *Q = st + 8;
  double factor = 1.0;
  for (int iw = 0; iw < 7; ++iw) {
    const double h2 = dt2_particle * w[iw] / 4.0, h4 = 0.5 * h2, h8 = 0.5 * h4;
    for (int sub = 0; sub < 4; ++sub) {
      p[3] += h4 * nhc_drive(p, Q, 3, ek2, dN, kT);
      for (int m = 2; m >= 0; --m) {
        const double a = exp(-h8 * p[m + 1] / Q[m + 1]);
        p[m] = a * (a * p[m] + h4 * nhc_drive(p, Q, m, ek2, dN, kT));
      }
      for (int m = 3; m >= 0; --m)
        eta[m] += h2 * p[m] / Q[m];
      const double s = exp(-h2 * p[0] / Q[0]);
      ek2 *= s * s;
      factor *= s;
      for (int m = 0; m < 3; ++m) {
        const double a = exp(-h8 * p[m + 1] / Q[m + 1]);
        p[m] = a * (a * p[m] + h4 * nhc_drive(p, Q, m, ek2, dN, kT));
      }
      p[3] += h4 * nhc_drive(p, Q, 3, ek2, dN, kT);
    }
  }
  return factor;
}

/* Bussi-Donadio-Parrinello thermostat (ensemble_bdp.cu:71-104) with the noise generators of
 * svr_utilities.cuh:28-122.  The reference draws uniforms with std::mt19937 +
 * std::uniform_real_distribution<double>(0, 1); both are restated here from their definitions
 * (MT19937 of Matsumoto & Nishimura; libstdc++'s generate_canonical<double, 53>: two 32-bit words,
 * sum = w0 + w1 * 2^32 in double, / 2^64) so that a seeded run can be checked draw for draw. */
typedef struct {
  unsigned int mt[624];
  int idx;
  int iset;
  double gset;
} nepo_bdp;

void nepo_bdp_seed(void* state, unsigned int seed)
{
  nepo_bdp* g = (nepo_bdp*)state;
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (unsigned int)i;
  g->idx = 624;
  g->iset = 0;
  g->gset = 0.0;
}

static unsigned int bdp_u32(nepo_bdp* g)
{
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const unsigned int y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  unsigned int y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

static double bdp_uniform(nepo_bdp* g)
{
  double sum = (double)bdp_u32(g);
  sum += (double)bdp_u32(g) * 4294967296.0;
  double r = sum / 18446744073709551616.0;
  if (r >= 1.0)
    r = nextafter(1.0, 0.0);
  return r;
}

static double bdp_gauss(nepo_bdp* g)
{
  if (g->iset) {
    g->iset = 0;
    return g->gset;
  }
  double v1, v2, rsq;
  do {
    v1 = 2.0 * bdp_uniform(g) - 1.0;
    v2 = 2.0 * bdp_uniform(g) - 1.0;
    rsq = v1 * v1 + v2 * v2;
  } while (rsq >= 1.0 || rsq == 0.0);
  const double fac = sqrt(-2.0 * log(rsq) / rsq);
  g->gset = v1 * fac;
  g->iset = 1;
  return v2 * fac;
}

static double bdp_gamma(nepo_bdp* g, int ia)
{
  if (ia < 6) {
    double x = 1.0;
    for (int j = 1; j <= ia; ++j)
      x *= bdp_uniform(g);
    return -log(x);
  }
  for (;;) {
    double x, y, v1, v2;
    const double am = ia - 1, sq = sqrt(2.0 * am + 1.0);
    do {
      do {
        v1 = bdp_uniform(g);
        v2 = 2.0 * bdp_uniform(g) - 1.0;
      } while (v1 * v1 + v2 * v2 > 1.0);
      y = v2 / v1;
      x = sq * y + am;
    } while (x <= 0.0);
    const double e = (1.0 + y * y) * exp(am * log(x / am) - sq * y);
    if (!(bdp_uniform(g) > e))
      return x;
  }
}

static double bdp_sum_noises(nepo_bdp* g, int nn)
{
  if (nn == 0)
    return 0.0;
  if (nn == 1) {
    const double rr = bdp_gauss(g);
    return rr * rr;
  }
  if (nn % 2 == 0)
    return 2.0 * bdp_gamma(g, nn / 2);
  const double rr = bdp_gauss(g);
  return 2.0 * bdp_gamma(g, (nn - 1) / 2) + rr * rr;
}

/* velocity scale factor of one BDP step from the instantaneous temperature (ensemble_bdp.cu:94-101) */
double nepo_bdp_factor(void* state, int n, double T_now, double T_target, double t_coup)
{
  nepo_bdp* g = (nepo_bdp*)state;
  const double kB = 8.617343e-5;
  const int ndeg = 3 * n;
  const double kk = T_now * ndeg * kB * 0.5, sigma = ndeg * kB * T_target * 0.5;
  const double f = t_coup > 0.1 ? exp(-1.0 / t_coup) : 0.0;
  const double rr = bdp_gauss(g);
  const double knew = kk + (1.0 - f) * (sigma * (bdp_sum_noises(g, ndeg - 1) + rr * rr) / ndeg - kk) +
                      2.0 * rr * sqrt(kk * sigma / ndeg * (1.0 - f) * f);
  return sqrt(knew / kk);
}

int nepo_bdp_sizeof(void) { return (int)sizeof(nepo_bdp); }

/* Run::perform_a_run, run.cu:250-318 restricted to ensemble nve (ensemble_nve.cu:31-95). */
int nepo_run_nve(
  const nepo_model* m, int precision, int n, const int* type, const double h[9],
  const int pbc[3], const double* mass, double dt, int nsteps, double* pos, double* vel,
  double* pe, double* force, double* virial, double* thermo_out)
{
  nepo_box box;
  nepo_box_init(&box, h, pbc);
  double* x0 = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  int rebuilds = 0;
  /* initial force (run.cu:220-232) */
  nepo_apply_pbc(n, h, pbc, pos);
  memcpy(x0, pos, sizeof(double) * 3 * (size_t)n);
  rebuilds++;
  int st = nepo_compute(m, precision, -1, n, type, h, pbc, pos, pe, force, virial, NULL, NULL);
  if (st)
    { free(x0); return st; }
  for (int step = 0; step < nsteps; ++step) {
    nepo_velocity_verlet(1, n, dt, mass, force, pos, vel);
    nepo_apply_pbc(n, h, pbc, pos);
    /* skin policy, neighbor.cu:646-684, 741-800: rebuild when an atom moved > skin/2 */
    int moved = 0;
    for (int i = 0; i < n && !moved; ++i) {
      float dx = (float)(pos[i] - x0[i]);
      float dy = (float)(pos[n + i] - x0[n + i]);
      float dz = (float)(pos[2 * (size_t)n + i] - x0[2 * (size_t)n + i]);
      int sh[3];
      apply_mic_f32(&box, &dx, &dy, &dz, sh);
      if ((double)(dx * dx + dy * dy + dz * dz) > 0.25)
        moved = 1;
    }
    if (moved) {
      memcpy(x0, pos, sizeof(double) * 3 * (size_t)n);
      rebuilds++;
    }
    st = nepo_compute(m, precision, -1, n, type, h, pbc, pos, pe, force, virial, NULL, NULL);
    if (st)
      break;
    nepo_velocity_verlet(0, n, dt, mass, force, pos, vel);
    nepo_thermo(n, box.volume, mass, pe, vel, virial, thermo_out + 8 * (size_t)step);
  }
  free(x0);
  return st ? st : rebuilds;
}
