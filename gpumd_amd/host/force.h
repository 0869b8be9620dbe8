// Force / Potential: the plugin surface of src/force/{force,potential}.cuh kept as is; the NEP
// implementation behind it is libnepmi.so through its C ABI (include/nepmi.h).
#pragma once
#include "model.h"

#include <memory>

extern "C" {
#include "../../include/nepmi.h"
}

namespace gmi {

// src/force/potential.cuh:21-43
class Potential
{
public:
  int N1 = 0, N2 = 0;
  double rc = 0.0;
  virtual ~Potential() = default;
  virtual void compute(
    Box& box, const GPU_Vector<int>& type, const GPU_Vector<double>& position,
    GPU_Vector<double>& potential, GPU_Vector<double>& force, GPU_Vector<double>& virial) = 0;
};

// class NEP : public Potential (src/force/nep.cuh) on top of the C ABI
class NEP_MI : public Potential
{
public:
  NEP_MI(const char* file_potential, int num_atoms);
  ~NEP_MI() override;
  void compute(
    Box& box, const GPU_Vector<int>& type, const GPU_Vector<double>& position,
    GPU_Vector<double>& potential, GPU_Vector<double>& force, GPU_Vector<double>& virial) override;
  nepmi_engine* engine() { return engine_; }
  void write_neighbor_out() const; // neighbor.out, nep.cu:1014-1034
  int nep_model_type = 0; // potential.cuh:33-34: 3 = temperature-dependent NEP (gets Force::temperature per call)

private:
  nepmi_model* model_ = nullptr;
  nepmi_engine* engine_ = nullptr;
};

// src/force/force.cuh:25-52
class Force
{
public:
  // `potential <file>`: factory keyed on the first token of the file (force.cu:75-218)
  void parse_potential(const std::vector<std::string>& param, const Box& box, int number_of_atoms, bool create = true);
  // wrap, zero, dispatch (force.cu:771-855)
  void compute(
    Box& box, GPU_Vector<double>& position, GPU_Vector<int>& type, GPU_Vector<double>& potential,
    GPU_Vector<double>& force, GPU_Vector<double>& virial);
  std::vector<std::unique_ptr<Potential>> potentials;
  nepmi_engine* engine() const; // the main (first) potential's engine
  // temperature-dependent NEP: Run::parse_run sets temperature = T1 and delta_T = (T2 - T1) / steps (run.cu:679-681); every
  // compute of a run -- the initial one (run.cu:232-241) and one per step -- adds delta_T first (force.cu:803)
  double temperature = 0.0, delta_T = 0.0;
  bool has_temperature_model() const;
  void advance_temperature() { temperature += delta_T; }
  // several `potential` lines (NEP only): "observe" = the first one drives the run, the others are only
  // evaluated by dump_observer; "average" = the run uses their mean (force.cu:514-565, force.cuh:81)
  void set_multiple_potentials_mode(const std::string& mode) { multiple_potentials_mode_ = mode; }
  const std::string& multiple_potentials_mode() const { return multiple_potentials_mode_; }

private:
  std::string multiple_potentials_mode_ = "observe";
  std::vector<std::string> atom_types_; // check_types, force.cu:55-73
  bool has_non_nep_ = false;
  int num_potentials_ = 0;
};

// element list of a potential file (read_xyz.cu:427-480 reads it before model.xyz)
std::vector<std::string> potential_elements(const std::string& file_potential);

} // namespace gmi
