// gpumd-mi host: model I/O kept from GPUMD (src/model/{atom,box,read_xyz}.cu*, src/main_gpumd/
// replicate.cu, velocity.cu): same files, same conventions, re-written in plain C++17 over the
// HIP runtime.  Device arrays keep GPUMD's SoA layout so that they can be handed to the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace gmi {

constexpr double K_B = 8.617343e-5;                     // src/utilities/common.cuh:22
constexpr double PRESSURE_UNIT_CONVERSION = 1.602177e+2; // natural -> GPa
constexpr double TIME_UNIT_CONVERSION = 1.018051e+1;     // natural -> fs

[[noreturn]] void input_error(const std::string& msg);  // PRINT_INPUT_ERROR: message + exit(1)
void hip_check(hipError_t e, const char* what);

// GPU_Vector<T> (src/utilities/gpu_vector.cuh): RAII device buffer
template <class T>
class GPU_Vector
{
public:
  GPU_Vector() = default;
  explicit GPU_Vector(size_t n) { resize(n); }
  GPU_Vector(const GPU_Vector&) = delete;
  GPU_Vector& operator=(const GPU_Vector&) = delete;
  ~GPU_Vector() { if (data_) (void)hipFree(data_); }
  void resize(size_t n)
  {
    if (data_) (void)hipFree(data_);
    data_ = nullptr;
    size_ = n;
    if (n) hip_check(hipMalloc((void**)&data_, n * sizeof(T)), "hipMalloc");
  }
  void fill_zero() { if (size_) hip_check(hipMemset(data_, 0, size_ * sizeof(T)), "hipMemset"); }
  void copy_from_host(const T* h, size_t n = 0)
  {
    if (n == 0) n = size_;
    if (n) hip_check(hipMemcpy(data_, h, n * sizeof(T), hipMemcpyHostToDevice), "H2D");
  }
  void copy_to_host(T* h, size_t n = 0) const
  {
    if (n == 0) n = size_;
    if (n) hip_check(hipMemcpy(h, data_, n * sizeof(T), hipMemcpyDeviceToHost), "D2H");
  }
  size_t size() const { return size_; }
  T* data() { return data_; }
  const T* data() const { return data_; }

private:
  T* data_ = nullptr;
  size_t size_ = 0;
};

// Box (src/model/box.cuh): cpu_h[0..8] = ax,bx,cx,ay,by,cy,az,bz,cz; [9..17] inverse
struct Box {
  int pbc_x = 1, pbc_y = 1, pbc_z = 1;
  double cpu_h[18] = {0};
  void get_inverse();
  double get_volume() const;
};

// Group (src/model/group.cuh, group.cu:25-72): one grouping method of model.xyz's group:I:k columns
struct Group {
  int number = 0;                // number of groups in this grouping method (largest label + 1)
  std::vector<int> cpu_label;    // [N] group label of every atom
  std::vector<int> cpu_size;     // [number] atoms per group
  std::vector<int> cpu_size_sum; // [number] exclusive prefix of cpu_size
  std::vector<int> cpu_contents; // [N] atom indices ordered by group, ascending inside a group
  void find_size_and_contents(int N, int k);
};

// Atom (src/model/atom.cuh:32-42)
struct Atom {
  int number_of_atoms = 0;
  std::vector<std::string> cpu_atom_symbol;
  std::vector<int> cpu_type;
  std::vector<double> cpu_mass;
  std::vector<float> cpu_charge; // zeros unless model.xyz has a charge column
  bool has_charge = false;
  std::vector<double> cpu_position_per_atom; // [x..|y..|z..]
  std::vector<double> cpu_velocity_per_atom;
  GPU_Vector<int> type;
  GPU_Vector<double> mass, position_per_atom, velocity_per_atom, force_per_atom, potential_per_atom, virial_per_atom;
  GPU_Vector<double> unwrapped_position; // empty until a dump asks for it (dump_xyz.cu:60-63)
  void allocate_gpu(); // allocate_memory_gpu, read_xyz.cu:532-557
};

// tokenizer shared by run.in and model.xyz (src/utilities/read_file.cu)
std::vector<std::string> get_tokens(const std::string& line);

// initialize_position (read_xyz.cu:482-530): model.xyz with the potential's element list
// returns has_velocity_in_xyz
bool read_xyz(
  const std::string& path, const std::vector<std::string>& elements, Box& box, Atom& atom, std::vector<Group>& groups);
// Replicate (src/main_gpumd/replicate.cu:50-71): supercell, atom order i,j,k outer, basis inner
void replicate(const int n[3], Box& box, Atom& atom, std::vector<Group>& groups);
// Velocity::initialize (velocity.cu:312-347): glibc rand() stream, momentum corrections, rescale
void initialize_velocity(double temperature, bool use_seed, int seed, Atom& atom);
int host_rand(); // rand() of the restated glibc stream (seeds of the Langevin generators, ensemble_lan.cu:39)
int host_rand_for_tests(unsigned seed, bool reseed); // the restated glibc rand() stream (gpumd-mi --rand-check)
// Velocity::correct_velocity (velocity.cu:210-271): zero the linear and angular momentum of the listed atoms
// (all atoms when contents == nullptr); host arrays, SoA with stride N
void correct_velocity(int N, const std::vector<double>& mass, const std::vector<double>& pos, std::vector<double>& vel,
                      const int* contents, int count);
void write_xyz_frame(FILE* f, const Box& box, const Atom& atom, const char* extra_props);

} // namespace gmi
