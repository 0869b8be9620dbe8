#include "model.h"

#include <cstdint>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace gmi {

void input_error(const std::string& msg)
{
  std::printf("Input Error:\n    %s\n", msg.c_str());
  std::fflush(stdout);
  std::exit(1);
}

void hip_check(hipError_t e, const char* what)
{
  if (e != hipSuccess) {
    std::printf("HIP Error: %s: %s\n", what, hipGetErrorString(e));
    std::exit(1);
  }
}

static const std::map<std::string, double> kMassTable{
#include "mass_table.inc"
};

void Box::get_inverse()
{
  double* h = cpu_h;
  h[9] = h[4] * h[8] - h[5] * h[7];
  h[10] = h[2] * h[7] - h[1] * h[8];
  h[11] = h[1] * h[5] - h[2] * h[4];
  h[12] = h[5] * h[6] - h[3] * h[8];
  h[13] = h[0] * h[8] - h[2] * h[6];
  h[14] = h[2] * h[3] - h[0] * h[5];
  h[15] = h[3] * h[7] - h[4] * h[6];
  h[16] = h[1] * h[6] - h[0] * h[7];
  h[17] = h[0] * h[4] - h[1] * h[3];
  const double det = h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) +
                     h[2] * (h[3] * h[7] - h[4] * h[6]);
  for (int n = 9; n < 18; ++n)
    h[n] /= det;
}

double Box::get_volume() const
{
  const double* h = cpu_h;
  return std::fabs(h[0] * (h[4] * h[8] - h[5] * h[7]) + h[1] * (h[5] * h[6] - h[3] * h[8]) +
                   h[2] * (h[3] * h[7] - h[4] * h[6]));
}

void Atom::allocate_gpu()
{
  const size_t N = number_of_atoms;
  type.resize(N);
  type.copy_from_host(cpu_type.data());
  mass.resize(N);
  mass.copy_from_host(cpu_mass.data());
  position_per_atom.resize(3 * N);
  position_per_atom.copy_from_host(cpu_position_per_atom.data());
  velocity_per_atom.resize(3 * N);
  velocity_per_atom.copy_from_host(cpu_velocity_per_atom.data());
  force_per_atom.resize(3 * N);
  force_per_atom.fill_zero();
  potential_per_atom.resize(N);
  potential_per_atom.fill_zero();
  virial_per_atom.resize(9 * N);
  virial_per_atom.fill_zero();
}

std::vector<std::string> get_tokens(const std::string& line)
{
  std::istringstream ss(line);
  std::vector<std::string> t;
  std::string w;
  while (ss >> w)
    t.push_back(w);
  return t;
}

static std::string lower(std::string s)
{
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::tolower(c); });
  return s;
}

// value of key="..." or key=word in the comment line (case-insensitive key)
static bool find_key(const std::string& line, const std::string& key, std::string& value)
{
  const std::string low = lower(line);
  size_t p = low.find(key + "=");
  while (p != std::string::npos && p > 0 && !std::isspace((unsigned char)low[p - 1]))
    p = low.find(key + "=", p + 1);
  if (p == std::string::npos)
    return false;
  size_t v = p + key.size() + 1;
  if (v < line.size() && line[v] == '"') {
    const size_t e = line.find('"', v + 1);
    value = line.substr(v + 1, e == std::string::npos ? std::string::npos : e - v - 1);
  } else {
    size_t e = v;
    while (e < line.size() && !std::isspace((unsigned char)line[e]))
      ++e;
    value = line.substr(v, e - v);
  }
  return true;
}

void Group::find_size_and_contents(int N, int k)
{
  cpu_size.assign(number, 0);
  cpu_size_sum.assign(number, 0);
  cpu_contents.resize(N);
  if (number == 1)
    std::printf("There is only one group of atoms in grouping method %d.\n", k);
  else
    std::printf("There are %d groups of atoms in grouping method %d.\n", number, k);
  for (int n = 0; n < N; ++n)
    cpu_size[cpu_label[n]]++;
  for (int m = 0; m < number; ++m)
    std::printf("    %d atoms in group %d.\n", cpu_size[m], m);
  for (int m = 1; m < number; ++m)
    cpu_size_sum[m] = cpu_size_sum[m - 1] + cpu_size[m - 1];
  std::vector<int> fill(cpu_size_sum);
  for (int n = 0; n < N; ++n)
    cpu_contents[fill[cpu_label[n]]++] = n;
}

bool read_xyz(
  const std::string& path, const std::vector<std::string>& elements, Box& box, Atom& atom, std::vector<Group>& groups)
{
  std::ifstream in(path);
  if (!in)
    input_error("Failed to open " + path + ".");
  std::string line;
  std::getline(in, line);
  auto tok = get_tokens(line);
  if (tok.size() != 1)
    input_error("The first line for the xyz file should have one value.");
  const int N = std::atoi(tok[0].c_str());
  if (N < 2)
    input_error("Number of atoms should >= 2.");
  std::printf("Number of atoms is %d.\n", N);

  std::getline(in, line);
  std::string v;
  box.pbc_x = box.pbc_y = box.pbc_z = 1;
  if (find_key(line, "pbc", v)) {
    auto p = get_tokens(lower(v));
    if (p.size() != 3)
      input_error("pbc should have 3 entries.");
    box.pbc_x = p[0] == "t";
    box.pbc_y = p[1] == "t";
    box.pbc_z = p[2] == "t";
  }
  if (!find_key(line, "lattice", v))
    input_error("'lattice' is missing in the second line of the model file.");
  {
    auto l = get_tokens(v);
    if (l.size() != 9)
      input_error("lattice should have 9 numbers.");
    const int transpose_index[9] = {0, 3, 6, 1, 4, 7, 2, 5, 8}; // read_xyz.cu:208-216
    for (int m = 0; m < 9; ++m)
      box.cpu_h[transpose_index[m]] = std::atof(l[m].c_str());
    box.get_inverse();
  }
  if (!find_key(line, "properties", v))
    input_error("'properties' is missing in the second line of the model file.");
  // name:type:count triples -> column offsets
  std::vector<std::string> p;
  {
    std::string w;
    std::istringstream ss(v);
    while (std::getline(ss, w, ':'))
      p.push_back(w);
  }
  int off = 0, off_species = -1, off_pos = -1, off_mass = -1, off_vel = -1, off_charge = -1, off_group = -1;
  int num_grouping_methods = 0;
  for (size_t k = 0; k + 3 <= p.size(); k += 3) {
    const std::string name = lower(p[k]);
    const int cnt = std::atoi(p[k + 2].c_str());
    if (name == "species") off_species = off;
    else if (name == "pos") off_pos = off;
    else if (name == "mass") off_mass = off;
    else if (name == "vel") off_vel = off;
    else if (name == "charge") off_charge = off;
    else if (name == "group") { off_group = off; num_grouping_methods = cnt; }
    off += cnt;
  }
  if (off_species < 0 || off_pos < 0)
    input_error("'species' or 'pos' is missing in properties.");
  std::printf(off_vel < 0 ? "Do not specify initial velocities here.\n" : "Specify initial velocities here.\n");
  groups.clear();
  groups.resize(off_group < 0 ? 0 : num_grouping_methods);
  if (groups.empty())
    std::printf("Have no grouping method.\n");
  else
    std::printf("Have %d grouping method(s).\n", (int)groups.size());
  for (auto& g : groups) {
    g.cpu_label.resize(N);
    g.number = 0;
  }
  atom.has_charge = off_charge >= 0;
  atom.cpu_charge.assign(N, 0.0f);

  atom.number_of_atoms = N;
  atom.cpu_atom_symbol.resize(N);
  atom.cpu_type.resize(N);
  atom.cpu_mass.resize(N);
  atom.cpu_position_per_atom.resize(3 * (size_t)N);
  atom.cpu_velocity_per_atom.assign(3 * (size_t)N, 0.0);
  for (int n = 0; n < N; ++n) {
    if (!std::getline(in, line))
      input_error("model.xyz ends early.");
    tok = get_tokens(line);
    if ((int)tok.size() < off)
      input_error("number of columns does not match properties.");
    atom.cpu_atom_symbol[n] = tok[off_species];
    int t = -1;
    for (size_t e = 0; e < elements.size(); ++e)
      if (elements[e] == tok[off_species])
        t = (int)e;
    if (t < 0)
      input_error("There is atom in model.xyz that is not allowed in the used potential.");
    atom.cpu_type[n] = t;
    for (int d = 0; d < 3; ++d)
      atom.cpu_position_per_atom[n + (size_t)N * d] = std::atof(tok[off_pos + d].c_str());
    if (off_mass >= 0) {
      atom.cpu_mass[n] = std::atof(tok[off_mass].c_str());
    } else {
      auto it = kMassTable.find(tok[off_species]);
      if (it == kMassTable.end())
        input_error("Unknown element " + tok[off_species] + ".");
      atom.cpu_mass[n] = it->second;
    }
    if (off_vel >= 0)
      for (int d = 0; d < 3; ++d) // A/fs -> natural units (read_xyz.cu:380-386)
        atom.cpu_velocity_per_atom[n + (size_t)N * d] = std::atof(tok[off_vel + d].c_str()) * TIME_UNIT_CONVERSION;
    if (off_charge >= 0)
      atom.cpu_charge[n] = (float)std::atof(tok[off_charge].c_str());
    for (size_t m = 0; m < groups.size(); ++m) { // read_xyz.cu:389-398
      const int label = std::atoi(tok[off_group + m].c_str());
      if (label < 0 || label >= N)
        input_error("Group label should >= 0 and < N.");
      groups[m].cpu_label[n] = label;
      if (label + 1 > groups[m].number)
        groups[m].number = label + 1;
    }
  }
  for (size_t m = 0; m < groups.size(); ++m)
    groups[m].find_size_and_contents(N, (int)m);
  return off_vel >= 0;
}

void replicate(const int r[3], Box& box, Atom& atom, std::vector<Group>& groups)
{
  const int N0 = atom.number_of_atoms;
  const int N = N0 * r[0] * r[1] * r[2];
  std::vector<std::string> sym(N);
  std::vector<int> type(N);
  std::vector<double> mass(N), pos(3 * (size_t)N), vel(3 * (size_t)N, 0.0);
  std::vector<float> charge(N);
  std::vector<std::vector<int>> label(groups.size(), std::vector<int>(N));
  const double* h = box.cpu_h;
  int m = 0;
  for (int i = 0; i < r[0]; ++i)
    for (int j = 0; j < r[1]; ++j)
      for (int k = 0; k < r[2]; ++k)
        for (int n = 0; n < N0; ++n, ++m) {
          sym[m] = atom.cpu_atom_symbol[n];
          type[m] = atom.cpu_type[n];
          mass[m] = atom.cpu_mass[n];
          charge[m] = atom.cpu_charge[n];
          for (size_t g = 0; g < groups.size(); ++g)
            label[g][m] = groups[g].cpu_label[n];
          const double d[3] = {h[0] * i + h[1] * j + h[2] * k, h[3] * i + h[4] * j + h[5] * k,
                               h[6] * i + h[7] * j + h[8] * k};
          for (int c = 0; c < 3; ++c) {
            pos[m + (size_t)N * c] = atom.cpu_position_per_atom[n + (size_t)N0 * c] + d[c];
            vel[m + (size_t)N * c] = atom.cpu_velocity_per_atom[n + (size_t)N0 * c];
          }
        }
  for (int c = 0; c < 3; ++c) {
    box.cpu_h[0 + c] *= r[c];
    box.cpu_h[3 + c] *= r[c];
    box.cpu_h[6 + c] *= r[c];
  }
  box.get_inverse();
  atom.number_of_atoms = N;
  atom.cpu_atom_symbol.swap(sym);
  atom.cpu_type.swap(type);
  atom.cpu_mass.swap(mass);
  atom.cpu_charge.swap(charge);
  for (size_t g = 0; g < groups.size(); ++g) { // labels repeat with the cell: sizes scale, numbers stay
    groups[g].cpu_label.swap(label[g]);
    groups[g].find_size_and_contents(N, (int)g);
  }
  atom.cpu_position_per_atom.swap(pos);
  atom.cpu_velocity_per_atom.swap(vel);
  std::printf("Replicated the box: %d x %d x %d, %d atoms.\n", r[0], r[1], r[2], N);
}

// ---- Velocity (src/main_gpumd/velocity.cu:40-271) ----
static void zero_linear_momentum(int N, const double* m, double* vx, double* vy, double* vz)
{
  double p[3] = {0, 0, 0}, M = 0;
  for (int i = 0; i < N; ++i) {
    M += m[i];
    p[0] += m[i] * vx[i];
    p[1] += m[i] * vy[i];
    p[2] += m[i] * vz[i];
  }
  for (int i = 0; i < N; ++i) {
    vx[i] -= p[0] / M;
    vy[i] -= p[1] / M;
    vz[i] -= p[2] / M;
  }
}

static void zero_angular_momentum(int N, const double* m, const double* x, const double* y, const double* z,
                                  double* vx, double* vy, double* vz)
{
  double r0[3] = {0, 0, 0}, M = 0;
  for (int i = 0; i < N; ++i) {
    M += m[i];
    r0[0] += m[i] * x[i];
    r0[1] += m[i] * y[i];
    r0[2] += m[i] * z[i];
  }
  for (double& c : r0)
    c /= M;
  double L[3] = {0, 0, 0}, I[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < N; ++i) {
    const double dx = x[i] - r0[0], dy = y[i] - r0[1], dz = z[i] - r0[2];
    L[0] += m[i] * (dy * vz[i] - dz * vy[i]);
    L[1] += m[i] * (dz * vx[i] - dx * vz[i]);
    L[2] += m[i] * (dx * vy[i] - dy * vx[i]);
    I[0][0] += m[i] * (dy * dy + dz * dz);
    I[1][1] += m[i] * (dx * dx + dz * dz);
    I[2][2] += m[i] * (dx * dx + dy * dy);
    I[0][1] -= m[i] * dx * dy;
    I[1][2] -= m[i] * dy * dz;
    I[0][2] -= m[i] * dx * dz;
  }
  I[1][0] = I[0][1];
  I[2][1] = I[1][2];
  I[2][0] = I[0][2];
  const double det = I[0][0] * (I[1][1] * I[2][2] - I[1][2] * I[2][1]) -
                     I[0][1] * (I[1][0] * I[2][2] - I[1][2] * I[2][0]) +
                     I[0][2] * (I[1][0] * I[2][1] - I[1][1] * I[2][0]);
  if (det > -1.0e-10 && det < 1.0e-10)
    return;
  double inv[3][3];
  inv[0][0] = (I[1][1] * I[2][2] - I[1][2] * I[2][1]) / det;
  inv[0][1] = (I[0][2] * I[2][1] - I[0][1] * I[2][2]) / det;
  inv[0][2] = (I[0][1] * I[1][2] - I[0][2] * I[1][1]) / det;
  inv[1][0] = (I[1][2] * I[2][0] - I[1][0] * I[2][2]) / det;
  inv[1][1] = (I[0][0] * I[2][2] - I[0][2] * I[2][0]) / det;
  inv[1][2] = (I[0][2] * I[1][0] - I[0][0] * I[1][2]) / det;
  inv[2][0] = (I[1][0] * I[2][1] - I[1][1] * I[2][0]) / det;
  inv[2][1] = (I[0][1] * I[2][0] - I[0][0] * I[2][1]) / det;
  inv[2][2] = (I[0][0] * I[1][1] - I[0][1] * I[1][0]) / det;
  double w[3];
  for (int a = 0; a < 3; ++a)
    w[a] = inv[a][0] * L[0] + inv[a][1] * L[1] + inv[a][2] * L[2];
  for (int i = 0; i < N; ++i) {
    const double dx = x[i] - r0[0], dy = y[i] - r0[1], dz = z[i] - r0[2];
    vx[i] -= w[1] * dz - w[2] * dy;
    vy[i] -= w[2] * dx - w[0] * dz;
    vz[i] -= w[0] * dy - w[1] * dx;
  }
}

// glibc's rand() / srand() (the TYPE_3 additive feedback generator of random_r.c: r[i] = r[i-3] + r[i-31], seeded by
// the Lehmer generator 16807 mod 2^31-1, first 310 outputs discarded), restated here so that the velocity stream of
// Velocity::initialize (velocity.cu:40-75) is reproduced independently of whatever else in the process -- the HIP
// runtime, RCCL -- calls the C library's rand().  Checked against the C library in tests/test_host_cli.py.
struct GlibcRand {
  std::vector<uint32_t> ring; // the last 34 values
  size_t pos = 0;
  explicit GlibcRand(unsigned seed = 1) { reseed(seed); }
  void reseed(unsigned seed)
  {
    if (seed == 0)
      seed = 1;
    std::vector<int32_t> s(34);
    s[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
      const int64_t hi = s[i - 1] / 127773, lo = s[i - 1] % 127773;
      int64_t w = 16807 * lo - 2836 * hi;
      if (w < 0)
        w += 2147483647;
      s[i] = (int32_t)w;
    }
    ring.assign(34, 0);
    for (int i = 0; i < 31; ++i)
      ring[i] = (uint32_t)s[i];
    for (int i = 31; i < 34; ++i)
      ring[i] = ring[i - 31];
    pos = 34;
    for (int i = 34; i < 344; ++i)
      next_raw();
  }
  uint32_t next_raw()
  {
    const uint32_t v = ring[(pos - 31) % 34] + ring[(pos - 3) % 34];
    ring[pos % 34] = v;
    ++pos;
    return v;
  }
  int next() { return (int)(next_raw() >> 1); }
};
static GlibcRand g_rand; // the process-wide stream of rand(), default seed 1 like the C library's

int host_rand() { return g_rand.next(); } // the next draw of the process-wide stream (what rand() returns in the reference)

int host_rand_for_tests(unsigned seed, bool reseed)
{
  if (reseed)
    g_rand.reseed(seed);
  return g_rand.next();
}

void correct_velocity(int N, const std::vector<double>& mass, const std::vector<double>& pos, std::vector<double>& vel,
                      const int* contents, int count)
{
  if (!contents) {
    zero_linear_momentum(N, mass.data(), vel.data(), vel.data() + N, vel.data() + 2 * (size_t)N);
    zero_angular_momentum(N, mass.data(), pos.data(), pos.data() + N, pos.data() + 2 * (size_t)N, vel.data(), vel.data() + N,
                          vel.data() + 2 * (size_t)N);
    return;
  }
  // one group: the same on the group's own copies (velocity.cu:284-305)
  std::vector<double> m(count), x(3 * (size_t)count), v(3 * (size_t)count);
  for (int k = 0; k < count; ++k) {
    const int n = contents[k];
    m[k] = mass[n];
    for (int d = 0; d < 3; ++d) {
      x[k + (size_t)d * count] = pos[n + (size_t)d * N];
      v[k + (size_t)d * count] = vel[n + (size_t)d * N];
    }
  }
  zero_linear_momentum(count, m.data(), v.data(), v.data() + count, v.data() + 2 * (size_t)count);
  zero_angular_momentum(count, m.data(), x.data(), x.data() + count, x.data() + 2 * (size_t)count, v.data(), v.data() + count,
                        v.data() + 2 * (size_t)count);
  for (int k = 0; k < count; ++k)
    for (int d = 0; d < 3; ++d)
      vel[contents[k] + (size_t)d * N] = v[k + (size_t)d * count];
}

void initialize_velocity(double temperature, bool use_seed, int seed, Atom& atom)
{
  const int N = atom.number_of_atoms;
  double* vx = atom.cpu_velocity_per_atom.data();
  double* vy = vx + N;
  double* vz = vy + N;
  constexpr double kRandMax = 2147483647.0; // RAND_MAX of glibc
  if (use_seed) {
    const unsigned int s = (unsigned int)seed;
    for (int n = 0; n < N; ++n) {
      g_rand.reseed(s + n * 3);
      vx[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
      g_rand.reseed(s + n * 3 + 1);
      vy[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
      g_rand.reseed(s + n * 3 + 2);
      vz[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
    }
  } else {
    for (int n = 0; n < N; ++n) {
      vx[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
      vy[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
      vz[n] = -1.0 + (g_rand.next() * 2.0) / kRandMax;
    }
  }
  const double* x = atom.cpu_position_per_atom.data();
  zero_linear_momentum(N, atom.cpu_mass.data(), vx, vy, vz);
  zero_angular_momentum(N, atom.cpu_mass.data(), x, x + N, x + 2 * (size_t)N, vx, vy, vz);
  double t = 0.0;
  for (int n = 0; n < N; ++n)
    t += atom.cpu_mass[n] * (vx[n] * vx[n] + vy[n] * vy[n] + vz[n] * vz[n]);
  t /= 3.0 * K_B * N;
  const double factor = std::sqrt(temperature / t);
  for (size_t k = 0; k < 3 * (size_t)N; ++k)
    vx[k] *= factor;
}

} // namespace gmi
