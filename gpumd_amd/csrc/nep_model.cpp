// nep.txt parser.  Format: src/force/nep.cu:100-377 of the reference; nep3 header and shared ANN
// block as accepted by its vendored NEP_CPU (nep.cpp:2568-2872).
#include "nep_model.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace nepmi {

namespace {

const char* const kElements[kMaxTypes] = {
  "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",
  "Cl", "Ar", "K",  "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge",
  "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",  "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd",
  "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd",
  "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au", "Hg",
  "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu"};

std::vector<std::string> next_tokens(std::istream& in)
{
  std::string line;
  std::vector<std::string> t;
  if (!std::getline(in, line))
    return t;
  std::istringstream ss(line);
  std::string w;
  while (ss >> w)
    t.push_back(w);
  return t;
}

bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }

} // namespace

// Tersoff1989::Tersoff1989, src/force/tersoff1989.cu:30-149
static std::string load_tersoff(std::istream& in, const std::vector<std::string>& head, NepModel& m)
{
  m.kind = 1;
  m.num_types = std::atoi(head[1].c_str());
  if (m.num_types < 1 || m.num_types > 2 || (int)head.size() != 2 + m.num_types)
    return "tersoff_1989 supports 1 or 2 element(s), listed on the first line.";
  m.symbols.assign(head.begin() + 2, head.end());
  auto finish = [](TersoffSet& t) {
    t.c2 = t.c * t.c;
    t.d2 = t.d * t.d;
    t.one_plus_c2overd2 = 1.0 + t.c2 / t.d2;
    t.pi_factor = 3.14159265358979323846 / (t.r2 - t.r1); // PI, src/utilities/common.cuh
    t.minus_half_over_n = -0.5 / t.n;
  };
  for (int t = 0; t < m.num_types; ++t) {
    auto tok = next_tokens(in);
    if (tok.size() != 11)
      return "Reading error for Tersoff-1989 potential.";
    TersoffSet& s = m.ters[t];
    double* f[11] = {&s.a, &s.b, &s.lambda, &s.mu, &s.beta, &s.n, &s.c, &s.d, &s.h, &s.r1, &s.r2};
    for (int k = 0; k < 11; ++k)
      *f[k] = std::atof(tok[k].c_str());
    finish(s);
  }
  double rc = m.ters[0].r2;
  if (m.num_types == 2) {
    auto tok = next_tokens(in);
    if (tok.size() != 1)
      return "Reading error for Tersoff-1989 potential.";
    const double chi = std::atof(tok[0].c_str());
    TersoffSet& q = m.ters[2];
    q.a = std::sqrt(m.ters[0].a * m.ters[1].a);
    q.b = std::sqrt(m.ters[0].b * m.ters[1].b) * chi;
    q.lambda = 0.5 * (m.ters[0].lambda + m.ters[1].lambda);
    q.mu = 0.5 * (m.ters[0].mu + m.ters[1].mu);
    q.r1 = std::sqrt(m.ters[0].r1 * m.ters[1].r1);
    q.r2 = std::sqrt(m.ters[0].r2 * m.ters[1].r2);
    q.pi_factor = 3.14159265358979323846 / (q.r2 - q.r1);
    rc = std::max(m.ters[0].r2, m.ters[1].r2);
  }
  // the shared Verlet-list machinery sees one cutoff; 50 neighbours like the reference (:141-149)
  m.rc_radial_max = m.rc_angular_max = rc;
  m.rc_radial.assign(m.num_types, rc);
  m.rc_angular.assign(m.num_types, rc);
  m.rc_radial_f.assign(m.num_types, (float)rc);
  m.rc_angular_f.assign(m.num_types, (float)rc);
  m.MN_radial = m.MN_angular = 50;
  m.atomic_numbers.assign(m.num_types, 0);
  m.b1t.assign(m.num_types, 0.0f);
  return "";
}

std::string load_nep_model(const std::string& path, NepModel& m, bool* unsupported)
{
  if (unsupported)
    *unsupported = false;
  auto unsup = [&](const std::string& msg) {
    if (unsupported)
      *unsupported = true;
    return msg;
  };
  std::ifstream in(path);
  if (!in)
    return "Failed to open " + path + ".";

  auto tok = next_tokens(in);
  if (tok.size() < 3)
    return "The first line of nep.txt should have at least 3 items.";
  if (tok[0] == "tersoff_1989")
    return load_tersoff(in, tok, m);
  const std::string& head = tok[0];
  // header: nep{3,4,5}[_zbl] and nep4[_zbl]_temperature; every other suffix (charge, dipole, polarizability)
  // is a different model_type in the reference (nep.cu:113-143) and outside this engine.
  if (head == "nep3" || head == "nep3_zbl")
    m.version = 3;
  else if (head == "nep4" || head == "nep4_zbl")
    m.version = 4;
  else if (head == "nep4_temperature" || head == "nep4_zbl_temperature") { // nep.cu:125-130
    m.version = 4;
    m.temperature_model = true;
  }
  else if (head == "nep5" || head == "nep5_zbl")
    m.version = 5;
  else if (starts_with(head, "nep"))
    return unsup(head + " is a NEP variant outside this engine (only potential models nep3/4/5[_zbl] and nep4[_zbl]_temperature).");
  else
    return head + " is an unsupported NEP model.";
  m.zbl_enabled = head.find("_zbl") != std::string::npos;
  m.num_types = std::atoi(tok[1].c_str());
  if (m.num_types < 1 || m.num_types > kMaxTypes || (int)tok.size() != 2 + m.num_types)
    return "The first line of nep.txt should have " + std::to_string(m.num_types) + " atom symbols.";
  const int T = m.num_types;
  m.symbols.assign(tok.begin() + 2, tok.end());
  m.atomic_numbers.assign(T, 0);
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < kMaxTypes; ++e)
      if (m.symbols[t] == kElements[e])
        m.atomic_numbers[t] = e + 1;

  if (m.zbl_enabled) {
    tok = next_tokens(in);
    if (tok.size() != 3 && tok.size() != 4)
      return "This line should be zbl rc_inner rc_outer [zbl_factor].";
    m.zbl_rc_inner = std::atof(tok[1].c_str());
    m.zbl_rc_outer = std::atof(tok[2].c_str());
    if (m.zbl_rc_inner == 0 && m.zbl_rc_outer == 0)
      m.zbl_flexible = true;
    else if (tok.size() == 4) { // universal ZBL with a type-wise outer cutoff (nep.cu:183-186, :935-941)
      m.zbl_typewise = true;
      m.zbl_typewise_factor = std::atof(tok[3].c_str());
      // covalent radii (Angstrom) by atomic number, the constant table of src/utilities/nep_utilities.cuh:143-154
      static const float kCovalentRadius[94] = {
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
      m.zbl_rc_outer_pair.assign((size_t)T * T, (float)m.zbl_rc_outer);
      for (int t1 = 0; t1 < T; ++t1)
        for (int t2 = 0; t2 < T; ++t2) {
          const int z1 = m.atomic_numbers[t1], z2 = m.atomic_numbers[t2];
          if (z1 < 1 || z1 > 94 || z2 < 1 || z2 > 94)
            return unsup("type-wise ZBL cutoff: element beyond Z = 94");
          const float rc = (kCovalentRadius[z1 - 1] + kCovalentRadius[z2 - 1]) * (float)m.zbl_typewise_factor;
          m.zbl_rc_outer_pair[(size_t)t1 * T + t2] = rc < (float)m.zbl_rc_outer ? rc : (float)m.zbl_rc_outer;
        }
    }
  }

  tok = next_tokens(in);
  if (tok.empty() || tok[0] != "cutoff" || ((int)tok.size() != 5 && (int)tok.size() != 2 * T + 3))
    return "cutoff should have 4 or num_types * 2 + 2 parameters.";
  m.rc_radial.assign(T, 0.0);
  m.rc_angular.assign(T, 0.0);
  for (int t = 0; t < T; ++t) {
    const bool per_type = tok.size() != 5;
    m.rc_radial[t] = std::atof(tok[per_type ? 1 + 2 * t : 1].c_str());
    m.rc_angular[t] = std::atof(tok[per_type ? 2 + 2 * t : 2].c_str());
    m.rc_radial_max = std::max(m.rc_radial_max, m.rc_radial[t]);
    m.rc_angular_max = std::max(m.rc_angular_max, m.rc_angular[t]);
    if (m.rc_angular[t] > m.rc_radial[t])
      return "angular cutoff should not be larger than radial cutoff.";
  }
  m.MN_radial = (int)std::ceil(std::atoi(tok[tok.size() - 2].c_str()) * 1.25);
  m.MN_angular = (int)std::ceil(std::atoi(tok[tok.size() - 1].c_str()) * 1.25);

  tok = next_tokens(in);
  if (tok.size() != 3 || tok[0] != "n_max")
    return "This line should be n_max n_max_radial n_max_angular.";
  m.n_max_radial = std::atoi(tok[1].c_str());
  m.n_max_angular = std::atoi(tok[2].c_str());
  tok = next_tokens(in);
  if (tok.size() != 3 || tok[0] != "basis_size")
    return "This line should be basis_size basis_size_radial basis_size_angular.";
  m.basis_size_radial = std::atoi(tok[1].c_str());
  m.basis_size_angular = std::atoi(tok[2].c_str());
  tok = next_tokens(in);
  if (tok.size() < 4 || tok[0] != "l_max")
    return "This line should be l_max l_max_3body has_q_222 has_q_1111 [...].";
  m.L_max = std::atoi(tok[1].c_str());
  m.has_q_222 = std::atoi(tok[2].c_str()) != 0;
  m.has_q_1111 = std::atoi(tok[3].c_str()) != 0;
  m.has_q_112 = tok.size() >= 5 && std::atoi(tok[4].c_str()) != 0;
  m.has_q_123 = tok.size() >= 6 && std::atoi(tok[5].c_str()) != 0;
  m.has_q_233 = tok.size() >= 7 && std::atoi(tok[6].c_str()) != 0;
  m.has_q_134 = tok.size() >= 8 && std::atoi(tok[7].c_str()) != 0;
  if (m.L_max < 1 || m.L_max > 8) // NUM_OF_ABC = 80 sums, nep_utilities.cuh:18
    return "l_max_3body should be 1..8.";
  // a row built from sums the model does not have would be identically zero (and inconsistent with its force)
  if ((m.has_q_222 && m.L_max < 2) || (m.has_q_112 && m.L_max < 2) || ((m.has_q_123 || m.has_q_233) && m.L_max < 3) ||
      (m.has_q_134 && m.L_max < 4))
    return unsup("a 4-body row needs sums of a higher l than l_max_3body provides.");
  m.num_L = m.L_max + m.has_q_222 + m.has_q_1111 + m.has_q_112 + m.has_q_123 + m.has_q_233 + m.has_q_134;
  tok = next_tokens(in);
  if (tok.size() != 3 || tok[0] != "ANN")
    return "This line should be ANN num_neurons 0.";
  m.num_neurons = std::atoi(tok[1].c_str());
  if (m.n_max_radial < 0 || m.n_max_radial > 19 || m.n_max_angular < 0 || m.n_max_angular > 19 ||
      m.basis_size_radial < 0 || m.basis_size_radial > 19 || m.basis_size_angular < 0 ||
      m.basis_size_angular > 19 || m.num_neurons < 1 || m.num_neurons > 200)
    return "n_max / basis_size / num_neurons out of range.";
  const int nR1 = m.n_max_radial + 1, nA1 = m.n_max_angular + 1;
  const int kR1 = m.basis_size_radial + 1, kA1 = m.basis_size_angular + 1;
  m.dim = nR1 + nA1 * m.num_L;
  const int dim_file = m.dim + (m.temperature_model ? 1 : 0); // annmb.dim of the reference (nep.cu:321-325)

  const int per_type = (dim_file + 2) * m.num_neurons;
  if (m.version == 3)
    m.num_para_ann = per_type + 1;
  else if (m.version == 4)
    m.num_para_ann = per_type * T + 1;
  else
    m.num_para_ann = (per_type + 1) * T + 1;
  m.num_c_radial = T * T * nR1 * kR1;
  m.num_para = m.num_para_ann + m.num_c_radial + T * T * nA1 * kA1;

  m.params.resize((size_t)m.num_para + dim_file);
  for (size_t k = 0; k < m.params.size(); ++k) {
    tok = next_tokens(in);
    if (tok.empty())
      return "nep.txt ends after " + std::to_string(k) + " of " + std::to_string(m.params.size()) + " parameters.";
    m.params[k] = std::atof(tok[0].c_str());
  }
  if (m.zbl_flexible) {
    m.zbl_para.resize(10 * (size_t)(T * (T + 1) / 2));
    for (size_t k = 0; k < m.zbl_para.size(); ++k) {
      tok = next_tokens(in);
      if (tok.empty())
        return "missing flexible-ZBL parameters.";
      m.zbl_para[k] = std::atof(tok[0].c_str());
    }
  }

  // ---- single-precision tables.  The reference stores float(parsed value) (nep.cu:353-357). ----
  const int nn = m.num_neurons, dim = m.dim;
  m.w0.resize((size_t)T * nn * dim);
  m.b0.resize((size_t)T * nn);
  m.w1.resize((size_t)T * nn);
  m.b1t.assign(T, 0.0f);
  m.w0_temp.assign(m.temperature_model ? (size_t)T * nn : 0, 0.0f);
  size_t off = 0;
  for (int t = 0; t < T; ++t) {
    if (t > 0 && m.version == 3)
      off = 0; // one ANN block shared by every type
    for (int j = 0; j < nn; ++j) {
      for (int d = 0; d < dim; ++d)
        m.w0[((size_t)t * nn + j) * dim + d] = (float)m.params[off + (size_t)j * dim_file + d];
      if (m.temperature_model)
        m.w0_temp[(size_t)t * nn + j] = (float)m.params[off + (size_t)j * dim_file + dim];
    }
    off += (size_t)nn * dim_file;
    for (int k = 0; k < nn; ++k)
      m.b0[(size_t)t * nn + k] = (float)m.params[off + k];
    off += nn;
    for (int k = 0; k < nn; ++k)
      m.w1[(size_t)t * nn + k] = (float)m.params[off + k];
    off += nn;
    if (m.version == 5)
      m.b1t[t] = (float)m.params[off++];
  }
  m.b1 = (float)m.params[off++];
  if ((int)off != m.num_para_ann)
    return "internal error: ANN layout mismatch.";

  // descriptor coefficients: file index (n*(K+1)+k)*T*T + t1*T + t2, radial block then angular.
  m.c_rad.resize((size_t)T * T * nR1 * kR1);
  m.c_ang.resize((size_t)T * T * nA1 * kA1);
  const double* c = m.params.data() + m.num_para_ann;
  for (int t1 = 0; t1 < T; ++t1)
    for (int t2 = 0; t2 < T; ++t2) {
      const int pair = t1 * T + t2;
      for (int n = 0; n < nR1; ++n)
        for (int k = 0; k < kR1; ++k)
          m.c_rad[((size_t)pair * nR1 + n) * kR1 + k] = (float)c[(size_t)(n * kR1 + k) * T * T + pair];
      for (int n = 0; n < nA1; ++n)
        for (int k = 0; k < kA1; ++k)
          m.c_ang[((size_t)pair * nA1 + n) * kA1 + k] =
            (float)c[m.num_c_radial + (size_t)(n * kA1 + k) * T * T + pair];
    }
  m.q_scaler.resize(dim);
  for (int d = 0; d < dim; ++d)
    m.q_scaler[d] = (float)m.params[(size_t)m.num_para + d];
  if (m.temperature_model)
    m.q_scaler_temp = (float)m.params[(size_t)m.num_para + dim];
  m.rc_radial_f.resize(T);
  m.rc_angular_f.resize(T);
  for (int t = 0; t < T; ++t) {
    m.rc_radial_f[t] = (float)m.rc_radial[t];
    m.rc_angular_f[t] = (float)m.rc_angular[t];
  }
  m.zbl_para_f.resize(m.zbl_para.size());
  for (size_t k = 0; k < m.zbl_para.size(); ++k)
    m.zbl_para_f[k] = (float)m.zbl_para[k];
  return "";
}

bool embed_model(NepModel& m, int NR, int KR, int NA, int KA)
{
  if (m.kind != 0 || m.embedded() || m.L_max < 1 || m.L_max > 4 || m.has_q_112 || m.has_q_123 || m.has_q_233 || m.has_q_134)
    return false;
  if (m.n_max_radial > NR || m.basis_size_radial > KR || m.n_max_angular > NA || m.basis_size_angular > KA)
    return false;
  const int T = m.num_types, nn = m.num_neurons;
  const int nR0 = m.n_max_radial + 1, nA0 = m.n_max_angular + 1, kR0 = m.basis_size_radial + 1, kA0 = m.basis_size_angular + 1;
  const int nR1 = NR + 1, nA1 = NA + 1, kR1 = KR + 1, kA1 = KA + 1;
  const int rows1 = 6; // L = 1, 2, 3, 4, then 222, then 1111
  const int dim0 = m.dim, dim1 = nR1 + nA1 * rows1;
  // the file's row r (0 .. num_L - 1: L = 1 .. l_max, then 222, then 1111, as far as the model has them) -> padded row
  std::vector<int> row_of(m.num_L);
  {
    int r = 0;
    for (int L = 1; L <= m.L_max; ++L)
      row_of[r++] = L - 1;
    if (m.has_q_222)
      row_of[r++] = 4;
    if (m.has_q_1111)
      row_of[r++] = 5;
  }
  std::vector<int> dmap(dim0);
  for (int n = 0; n < nR0; ++n)
    dmap[n] = n;
  for (int r = 0; r < m.num_L; ++r)
    for (int n = 0; n < nA0; ++n)
      dmap[nR0 + r * nA0 + n] = nR1 + row_of[r] * nA1 + n;
  // ANN input weights and the descriptor scaler on the padded components: zero
  std::vector<float> w0((size_t)T * nn * dim1, 0.0f), qs(dim1, 0.0f);
  for (int t = 0; t < T; ++t)
    for (int j = 0; j < nn; ++j)
      for (int d = 0; d < dim0; ++d)
        w0[((size_t)t * nn + j) * dim1 + dmap[d]] = m.w0[((size_t)t * nn + j) * dim0 + d];
  for (int d = 0; d < dim0; ++d)
    qs[dmap[d]] = m.q_scaler[d];
  std::vector<float> cr((size_t)T * T * nR1 * kR1, 0.0f), ca((size_t)T * T * nA1 * kA1, 0.0f);
  for (int pair = 0; pair < T * T; ++pair) {
    for (int n = 0; n < nR0; ++n)
      for (int k = 0; k < kR0; ++k)
        cr[((size_t)pair * nR1 + n) * kR1 + k] = m.c_rad[((size_t)pair * nR0 + n) * kR0 + k];
    for (int n = 0; n < nA0; ++n)
      for (int k = 0; k < kA0; ++k)
        ca[((size_t)pair * nA1 + n) * kA1 + k] = m.c_ang[((size_t)pair * nA0 + n) * kA0 + k];
  }
  m.file_n_max_radial = m.n_max_radial;
  m.file_n_max_angular = m.n_max_angular;
  m.file_basis_size_radial = m.basis_size_radial;
  m.file_basis_size_angular = m.basis_size_angular;
  m.file_L_max = m.L_max;
  m.file_has_q_222 = m.has_q_222;
  m.file_has_q_1111 = m.has_q_1111;
  m.file_num_L = m.num_L;
  m.file_dim = m.dim;
  m.dmap.swap(dmap);
  m.w0.swap(w0);
  m.q_scaler.swap(qs);
  m.c_rad.swap(cr);
  m.c_ang.swap(ca);
  m.n_max_radial = NR;
  m.n_max_angular = NA;
  m.basis_size_radial = KR;
  m.basis_size_angular = KA;
  m.L_max = 4;
  m.has_q_222 = 1;
  m.has_q_1111 = 1;
  m.num_L = rows1;
  m.dim = dim1;
  return true;
}

} // namespace nepmi
