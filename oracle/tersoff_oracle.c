/* TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product.
 *
 * CPU restatement (plain C, FP64) of the reference's Tersoff-1989 potential (BASELINE config 2):
 *   parser            src/force/tersoff1989.cu:30-149
 *   neighbour lists   Verlet list rc + skin (neighbor.cu:85-162) then the local list
 *                     gpu_find_local_neighbor_from_global (neighbor.cu:699-737; FLOAT geometry)
 *   step 1 (b, b')    tersoff1989.cu:337-405
 *   step 2 (f12, U)   tersoff1989.cu:408-505
 *   force + virial    gpu_find_force_many_body (double), src/force/potential.cu:35-134
 *
 * PARITY UNPINNED at the force level: the reference tree holds no force-level golden vector for
 * Tersoff-1989 (only phonon frequencies / graphene thermo indirectly).  The oracle is
 * validated by finite differences of its own energy, Newton's third law and the virial identity
 * (tests/test_tersoff.py).  Built by oracle/Makefile into oracle/libtersoff_oracle.so.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double a, b, lambda, mu, beta, n, c, d, h, r1, r2;
  double c2, d2, one_plus_c2overd2, pi_factor, minus_half_over_n;
} ters_par;

typedef struct {
  int num_types;
  ters_par p[3]; /* 0: type 0-0, 1: type 1-1, 2: mixed */
  double rc;
  char symbols[2][8];
} tersoff_model;

#define TERS_PI 3.14159265358979323846

static void finish(ters_par* t)
{
  t->c2 = t->c * t->c;
  t->d2 = t->d * t->d;
  t->one_plus_c2overd2 = 1.0 + t->c2 / t->d2;
  t->pi_factor = TERS_PI / (t->r2 - t->r1);
  t->minus_half_over_n = -0.5 / t->n;
}

tersoff_model* terso_load(const char* path)
{
  FILE* f = fopen(path, "r");
  if (!f)
    return NULL;
  tersoff_model* m = (tersoff_model*)calloc(1, sizeof(tersoff_model));
  char name[64];
  if (fscanf(f, "%63s%d", name, &m->num_types) != 2 || strcmp(name, "tersoff_1989") != 0 || m->num_types < 1 ||
      m->num_types > 2) {
    fclose(f);
    free(m);
    return NULL;
  }
  for (int t = 0; t < m->num_types; ++t)
    if (fscanf(f, "%7s", m->symbols[t]) != 1) { fclose(f); free(m); return NULL; }
  for (int t = 0; t < m->num_types; ++t) {
    ters_par* p = &m->p[t];
    if (fscanf(f, "%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf", &p->a, &p->b, &p->lambda, &p->mu, &p->beta, &p->n, &p->c, &p->d,
               &p->h, &p->r1, &p->r2) != 11) { fclose(f); free(m); return NULL; }
    finish(p);
  }
  m->rc = m->p[0].r2;
  if (m->num_types == 2) {
    double chi;
    if (fscanf(f, "%lf", &chi) != 1) { fclose(f); free(m); return NULL; }
    ters_par* q = &m->p[2];
    memset(q, 0, sizeof(*q));
    q->a = sqrt(m->p[0].a * m->p[1].a);
    q->b = sqrt(m->p[0].b * m->p[1].b) * chi;
    q->lambda = 0.5 * (m->p[0].lambda + m->p[1].lambda);
    q->mu = 0.5 * (m->p[0].mu + m->p[1].mu);
    q->r1 = sqrt(m->p[0].r1 * m->p[1].r1);
    q->r2 = sqrt(m->p[0].r2 * m->p[1].r2);
    q->pi_factor = TERS_PI / (q->r2 - q->r1);
    m->rc = m->p[0].r2 > m->p[1].r2 ? m->p[0].r2 : m->p[1].r2;
  }
  fclose(f);
  return m;
}

void terso_free(tersoff_model* m) { free(m); }
/* the three parameter sets (type 0-0, type 1-1, mixed), 16 doubles each in the order of ters_par: what the pin against
 * the reference's own kernels (oracle/ref_tersoff_wrap.cpp) feeds them */
void terso_params(const tersoff_model* m, double out[48])
{
  for (int s = 0; s < 3; ++s) {
    const ters_par* p = &m->p[s];
    const double v[16] = {p->a, p->b, p->lambda, p->mu, p->beta, p->n, p->c, p->d, p->h, p->r1, p->r2,
                          p->c2, p->d2, p->one_plus_c2overd2, p->pi_factor, p->minus_half_over_n};
    memcpy(out + 16 * s, v, sizeof v);
  }
}
double terso_rc(const tersoff_model* m) { return m->rc; }

static const ters_par* pair_par(const tersoff_model* m, int t1, int t2)
{
  if (t1 == 0 && t2 == 0) return &m->p[0];
  if (t1 == 1 && t2 == 1) return &m->p[1];
  return &m->p[2];
}

typedef struct {
  double h[18];
  float hf[18];
  int pbc[3];
  int ortho;
} tbox;

static void box_init(tbox* b, const double h9[9], const int pbc[3])
{
  double* h = b->h;
  memcpy(h, h9, 9 * sizeof(double));
  h[9] = h[4] * h[8] - h[5] * h[7];
  h[10] = h[2] * h[7] - h[1] * h[8];
  h[11] = h[1] * h[5] - h[2] * h[4];
  h[12] = h[5] * h[6] - h[3] * h[8];
  h[13] = h[0] * h[8] - h[2] * h[6];
  h[14] = h[2] * h[3] - h[0] * h[5];
  h[15] = h[3] * h[7] - h[4] * h[6];
  h[16] = h[1] * h[6] - h[0] * h[7];
  h[17] = h[0] * h[4] - h[1] * h[3];
  double det = h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
  for (int k = 9; k < 18; ++k) h[k] /= det;
  for (int k = 0; k < 18; ++k) b->hf[k] = (float)h[k];
  for (int d = 0; d < 3; ++d) b->pbc[d] = pbc[d];
  b->ortho = h[1] == 0 && h[2] == 0 && h[3] == 0 && h[5] == 0 && h[6] == 0 && h[7] == 0;
}

/* apply_mic double, box.cuh:39-82 */
static void mic_d(const tbox* b, double* x, double* y, double* z)
{
  const double* h = b->h;
  if (b->ortho) {
    if (b->pbc[0]) { if (*x < -h[0] * 0.5) *x += h[0]; else if (*x > h[0] * 0.5) *x -= h[0]; }
    if (b->pbc[1]) { if (*y < -h[4] * 0.5) *y += h[4]; else if (*y > h[4] * 0.5) *y -= h[4]; }
    if (b->pbc[2]) { if (*z < -h[8] * 0.5) *z += h[8]; else if (*z > h[8] * 0.5) *z -= h[8]; }
  } else {
    double sx = h[9] * *x + h[10] * *y + h[11] * *z;
    double sy = h[12] * *x + h[13] * *y + h[14] * *z;
    double sz = h[15] * *x + h[16] * *y + h[17] * *z;
    if (b->pbc[0]) sx -= nearbyint(sx);
    if (b->pbc[1]) sy -= nearbyint(sy);
    if (b->pbc[2]) sz -= nearbyint(sz);
    *x = h[0] * sx + h[1] * sy + h[2] * sz;
    *y = h[3] * sx + h[4] * sy + h[5] * sz;
    *z = h[6] * sx + h[7] * sy + h[8] * sz;
  }
}

/* apply_mic float, box.cuh:84-129, with the same fma chains as the NEP oracle */
static void mic_f(const tbox* b, float* x, float* y, float* z)
{
  const float* H = b->hf;
  if (b->ortho) {
    if (b->pbc[0]) { if (*x < -H[0] * 0.5f) *x += H[0]; else if (*x > H[0] * 0.5f) *x -= H[0]; }
    if (b->pbc[1]) { if (*y < -H[4] * 0.5f) *y += H[4]; else if (*y > H[4] * 0.5f) *y -= H[4]; }
    if (b->pbc[2]) { if (*z < -H[8] * 0.5f) *z += H[8]; else if (*z > H[8] * 0.5f) *z -= H[8]; }
  } else {
    float sx = fmaf(H[11], *z, fmaf(H[10], *y, H[9] * *x));
    float sy = fmaf(H[14], *z, fmaf(H[13], *y, H[12] * *x));
    float sz = fmaf(H[17], *z, fmaf(H[16], *y, H[15] * *x));
    if (b->pbc[0]) sx -= nearbyintf(sx);
    if (b->pbc[1]) sy -= nearbyintf(sy);
    if (b->pbc[2]) sz -= nearbyintf(sz);
    *x = fmaf(H[2], sz, fmaf(H[1], sy, H[0] * sx));
    *y = fmaf(H[5], sz, fmaf(H[4], sy, H[3] * sx));
    *z = fmaf(H[8], sz, fmaf(H[7], sy, H[6] * sx));
  }
}

static void fc_fcp(const ters_par* p, double d, double* fc, double* fcp)
{
  if (d < p->r1) { *fc = 1.0; *fcp = 0.0; }
  else if (d < p->r2) {
    *fc = cos(p->pi_factor * (d - p->r1)) * 0.5 + 0.5;
    *fcp = -sin(p->pi_factor * (d - p->r1)) * p->pi_factor * 0.5;
  } else { *fc = 0.0; *fcp = 0.0; }
}

static int g_full_scan = 0;
/* tests: 1 = always take the defining O(N^2) candidate scan */
void terso_full_scan(int on) { g_full_scan = on; }

/* Force::compute for one configuration (positions not wrapped here).  Outputs assigned.
 * nn_out/nl_out (may be NULL): the local neighbour list, column-major nl[slot*n+atom], ld slots,
 * ascending neighbour index.  Returns max neighbour count or < 0. */
int terso_compute(
  const tersoff_model* m, int n, const int* type, const double h9[9], const int pbc[3], const double* pos,
  double* pe, double* force, double* virial, int* nn_out, int* nl_out, int ld)
{
  tbox box;
  box_init(&box, h9, pbc);
  const double *X = pos, *Y = pos + n, *Z = pos + 2 * (size_t)n;
  const int cap = 64;
  int* nn = (int*)calloc(n, sizeof(int));
  int* nl = (int*)malloc(sizeof(int) * (size_t)n * cap);
  const float rc2 = (float)(m->rc * m->rc);
  int mx = 0;
  /* Candidate search: the O(N^2) scan below is what defines the list.  For fully periodic boxes with at least 3
   * cells of thickness >= 1.02 rc per direction the candidates of atom i are first narrowed to the 27 cells around
   * it and sorted ascending (a superset of the neighbours, so the list is the same) -- only so that the checker
   * and bench.py's CPU baseline stay O(N). */
  int nc[3] = {0, 0, 0};
  int use_cells = !g_full_scan && box.pbc[0] && box.pbc[1] && box.pbc[2] && n > 512;
  for (int d = 0; d < 3 && use_cells; ++d) {
    const double* r = box.h + 9 + 3 * d;
    const double thick = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    nc[d] = (int)floor(thick / (m->rc * 1.02));
    if (nc[d] < 3) use_cells = 0;
  }
  int *cell_of = NULL, *cell_start = NULL, *cell_atoms = NULL;
  if (use_cells) {
    const int ncell = nc[0] * nc[1] * nc[2];
    cell_of = (int*)malloc(sizeof(int) * (size_t)n);
    cell_start = (int*)calloc((size_t)ncell + 1, sizeof(int));
    cell_atoms = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
      int c[3];
      for (int d = 0; d < 3; ++d) {
        const double* r = box.h + 9 + 3 * d;
        double sf = r[0] * X[i] + r[1] * Y[i] + r[2] * Z[i];
        sf -= floor(sf);
        c[d] = (int)(sf * nc[d]);
        if (c[d] >= nc[d]) c[d] = nc[d] - 1;
      }
      cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
      cell_start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) cell_start[c + 1] += cell_start[c];
    int* fill = (int*)calloc((size_t)ncell, sizeof(int));
    for (int i = 0; i < n; ++i) cell_atoms[cell_start[cell_of[i]] + fill[cell_of[i]]++] = i; /* ascending within a cell */
    free(fill);
  }
  int* cand = (int*)malloc(sizeof(int) * (size_t)(use_cells ? n : 1));
  for (int i = 0; i < n; ++i) {
    int ncand = n;
    if (use_cells) {
      ncand = 0;
      const int c0 = cell_of[i] % nc[0], c1 = (cell_of[i] / nc[0]) % nc[1], c2 = cell_of[i] / (nc[0] * nc[1]);
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int c = (((c2 + dz + nc[2]) % nc[2]) * nc[1] + (c1 + dy + nc[1]) % nc[1]) * nc[0] + (c0 + dx + nc[0]) % nc[0];
            for (int k = cell_start[c]; k < cell_start[c + 1]; ++k) cand[ncand++] = cell_atoms[k];
          }
      for (int a = 1; a < ncand; ++a) { /* insertion sort: ~100 candidates */
        const int v = cand[a];
        int q = a - 1;
        while (q >= 0 && cand[q] > v) { cand[q + 1] = cand[q]; --q; }
        cand[q + 1] = v;
      }
    }
    for (int jj = 0; jj < ncand; ++jj) { /* ascending j == the sorted order of gpu_sort_neighbor_list */
      const int j = use_cells ? cand[jj] : jj;
      if (j == i) continue;
      float x = (float)(X[j] - X[i]), y = (float)(Y[j] - Y[i]), z = (float)(Z[j] - Z[i]);
      mic_f(&box, &x, &y, &z);
      const float d2 = fmaf(z, z, fmaf(y, y, x * x));
      if (d2 >= rc2) continue;
      if (nn[i] >= cap) { free(nn); free(nl); free(cand); free(cell_of); free(cell_start); free(cell_atoms); return -1; }
      nl[(size_t)i * cap + nn[i]++] = j;
    }
    if (nn[i] > mx) mx = nn[i];
  }
  free(cand); free(cell_of); free(cell_start); free(cell_atoms);
  double* bb = (double*)malloc(sizeof(double) * (size_t)n * cap);
  double* bp = (double*)malloc(sizeof(double) * (size_t)n * cap);
  double* f12 = (double*)malloc(sizeof(double) * 3 * (size_t)n * cap);
  for (int i = 0; i < n; ++i) pe[i] = 0.0;
  for (size_t k = 0; k < 3 * (size_t)n; ++k) force[k] = 0.0;
  for (size_t k = 0; k < 9 * (size_t)n; ++k) virial[k] = 0.0;
  /* step 1 */
  for (int n1 = 0; n1 < n; ++n1) {
    const int t1 = type[n1];
    const ters_par* p1 = &m->p[t1];
    for (int i1 = 0; i1 < nn[n1]; ++i1) {
      const int n2 = nl[(size_t)n1 * cap + i1];
      double x12 = X[n2] - X[n1], y12 = Y[n2] - Y[n1], z12 = Z[n2] - Z[n1];
      mic_d(&box, &x12, &y12, &z12);
      const double d12 = sqrt(x12 * x12 + y12 * y12 + z12 * z12);
      double zeta = 0.0;
      for (int i2 = 0; i2 < nn[n1]; ++i2) {
        const int n3 = nl[(size_t)n1 * cap + i2];
        if (n3 == n2) continue;
        double x13 = X[n3] - X[n1], y13 = Y[n3] - Y[n1], z13 = Z[n3] - Z[n1];
        mic_d(&box, &x13, &y13, &z13);
        const double d13 = sqrt(x13 * x13 + y13 * y13 + z13 * z13);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) / (d12 * d13);
        double fc13, fcp13;
        fc_fcp(pair_par(m, t1, type[n3]), d13, &fc13, &fcp13);
        const double tmp = p1->d2 + (c123 - p1->h) * (c123 - p1->h);
        zeta += fc13 * (p1->one_plus_c2overd2 - p1->c2 / tmp);
      }
      const double bzn = pow(p1->beta * zeta, p1->n);
      const double b12 = pow(1.0 + bzn, p1->minus_half_over_n);
      if (zeta < 1.0e-16) { bb[(size_t)n1 * cap + i1] = 1.0; bp[(size_t)n1 * cap + i1] = 0.0; }
      else { bb[(size_t)n1 * cap + i1] = b12; bp[(size_t)n1 * cap + i1] = -b12 * bzn * 0.5 / ((1.0 + bzn) * zeta); }
    }
  }
  /* step 2 */
  for (int n1 = 0; n1 < n; ++n1) {
    const int t1 = type[n1];
    const ters_par* p1 = &m->p[t1];
    double u = 0.0;
    for (int i1 = 0; i1 < nn[n1]; ++i1) {
      const size_t idx = (size_t)n1 * cap + i1;
      const int n2 = nl[idx];
      const ters_par* p12 = pair_par(m, t1, type[n2]);
      double x12 = X[n2] - X[n1], y12 = Y[n2] - Y[n1], z12 = Z[n2] - Z[n1];
      mic_d(&box, &x12, &y12, &z12);
      const double d12 = sqrt(x12 * x12 + y12 * y12 + z12 * z12), d12inv = 1.0 / d12;
      double fc12, fcp12;
      fc_fcp(p12, d12, &fc12, &fcp12);
      const double fa12 = p12->b * exp(-p12->mu * d12), fap12 = -p12->mu * fa12;
      const double fr12 = p12->a * exp(-p12->lambda * d12), frp12 = -p12->lambda * fr12;
      const double b12 = bb[idx], bp12 = bp[idx];
      const double factor3 = (fcp12 * (fr12 - b12 * fa12) + fc12 * (frp12 - b12 * fap12)) * d12inv;
      double fx = x12 * factor3 * 0.5, fy = y12 * factor3 * 0.5, fz = z12 * factor3 * 0.5;
      u += fc12 * (fr12 - b12 * fa12) * 0.5;
      for (int i2 = 0; i2 < nn[n1]; ++i2) {
        const size_t idx2 = (size_t)n1 * cap + i2;
        const int n3 = nl[idx2];
        if (n3 == n2) continue;
        const ters_par* p13 = pair_par(m, t1, type[n3]);
        double x13 = X[n3] - X[n1], y13 = Y[n3] - Y[n1], z13 = Z[n3] - Z[n1];
        mic_d(&box, &x13, &y13, &z13);
        const double d13 = sqrt(x13 * x13 + y13 * y13 + z13 * z13);
        double fc13, fcp13;
        fc_fcp(p13, d13, &fc13, &fcp13);
        const double fa13 = p13->b * exp(-p13->mu * d13);
        const double bp13 = bp[idx2];
        const double od = 1.0 / (d12 * d13);
        const double c123 = (x12 * x13 + y12 * y13 + z12 * z13) * od;
        const double c_over = c123 * d12inv * d12inv;
        const double tmp = p1->d2 + (c123 - p1->h) * (c123 - p1->h);
        const double g123 = p1->one_plus_c2overd2 - p1->c2 / tmp;
        const double gp123 = 2.0 * p1->c2 * (c123 - p1->h) / (tmp * tmp);
        const double ta = (-bp12 * fc12 * fa12 * fc13 - bp13 * fc13 * fa13 * fc12) * gp123;
        const double tb = -bp13 * fc13 * fa13 * fcp12 * g123 * d12inv;
        fx += (x12 * tb + ta * (x13 * od - x12 * c_over)) * 0.5;
        fy += (y12 * tb + ta * (y13 * od - y12 * c_over)) * 0.5;
        fz += (z12 * tb + ta * (z13 * od - z12 * c_over)) * 0.5;
      }
      f12[3 * idx] = fx; f12[3 * idx + 1] = fy; f12[3 * idx + 2] = fz;
    }
    pe[n1] += u;
  }
  /* many-body accumulate (double), potential.cu:35-134 */
  for (int n1 = 0; n1 < n; ++n1) {
    double sf[3] = {0, 0, 0}, sv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i1 = 0; i1 < nn[n1]; ++i1) {
      const size_t idx = (size_t)n1 * cap + i1;
      const int n2 = nl[idx];
      double x12 = X[n2] - X[n1], y12 = Y[n2] - Y[n1], z12 = Z[n2] - Z[n1];
      mic_d(&box, &x12, &y12, &z12);
      int off = 0;
      for (int k = 0; k < nn[n2]; ++k)
        if (nl[(size_t)n2 * cap + k] == n1) { off = k; break; }
      const double* a = f12 + 3 * idx;
      const double* c = f12 + 3 * ((size_t)n2 * cap + off);
      sf[0] += a[0] - c[0]; sf[1] += a[1] - c[1]; sf[2] += a[2] - c[2];
      sv[0] += x12 * c[0]; sv[1] += y12 * c[1]; sv[2] += z12 * c[2];
      sv[3] += x12 * c[1]; sv[4] += x12 * c[2]; sv[5] += y12 * c[2];
      sv[6] += y12 * c[0]; sv[7] += z12 * c[0]; sv[8] += z12 * c[1];
    }
    for (int d = 0; d < 3; ++d) force[(size_t)d * n + n1] += sf[d];
    for (int d = 0; d < 9; ++d) virial[(size_t)d * n + n1] += sv[d];
  }
  if (nn_out)
    for (int i = 0; i < n; ++i) {
      nn_out[i] = nn[i];
      if (nl_out)
        for (int s = 0; s < nn[i] && s < ld; ++s) nl_out[(size_t)s * n + i] = nl[(size_t)i * cap + s];
    }
  free(nn); free(nl); free(bb); free(bp); free(f12);
  return mx;
}
