#include "force.h"

#include <cstring>
#include <fstream>

namespace gmi {

static void die_on(int status, const char* where)
{
  if (status < 0) {
    std::printf("%s: %s\n", where, nepmi_last_error());
    std::exit(1);
  }
}

NEP_MI::NEP_MI(const char* file_potential, int num_atoms)
{
  model_ = nepmi_model_load(file_potential);
  if (!model_) {
    std::printf("NEP: %s\n", nepmi_last_error());
    std::exit(1);
  }
  nepmi_info info;
  nepmi_model_info(model_, &info);
  rc = info.rc_radial;
  nep_model_type = info.model_type;
  N1 = 0;
  N2 = num_atoms;
  if (info.version == 0) { // Tersoff-1989
    std::printf("Use Tersoff-1989 (%d-element) potential with element(s):", info.num_types);
    for (int t = 0; t < info.num_types; ++t)
      std::printf(" %s", nepmi_model_symbol(model_, t));
    std::printf("\n    cutoff = %g A.\n", info.rc_radial);
    engine_ = nepmi_engine_create(model_, num_atoms, nullptr);
    if (!engine_) {
      std::printf("Tersoff: %s\n", nepmi_last_error());
      std::exit(1);
    }
    return;
  }
  std::printf("Use the NEP%d potential with %d atom type%s.\n", info.version, info.num_types, info.num_types > 1 ? "s" : "");
  for (int t = 0; t < info.num_types; ++t)
    std::printf("    type %d (%s).\n", t, nepmi_model_symbol(model_, t));
  if (info.zbl_enabled)
    std::printf("    has %s ZBL.\n", info.zbl_flexible ? "flexible" : "universal");
  std::printf("    radial cutoff = %g A.\n    angular cutoff = %g A.\n", info.rc_radial, info.rc_angular);
  std::printf("    enlarged MN_radial = %d.\n    enlarged MN_angular = %d.\n", info.MN_radial, info.MN_angular);
  std::printf("    n_max_radial = %d.\n    n_max_angular = %d.\n", info.n_max_radial, info.n_max_angular);
  std::printf("    basis_size_radial = %d.\n    basis_size_angular = %d.\n", info.basis_size_radial, info.basis_size_angular);
  std::printf("    l_max_3body = %d.\n    l_max_4body = %d.\n    l_max_5body = %d.\n", info.L_max, info.has_q_222 ? 2 : 0,
              info.has_q_1111 ? 1 : 0);
  if (info.has_q_112 || info.has_q_123 || info.has_q_233 || info.has_q_134)
    std::printf("    has_q_112 = %d.\n    has_q_123 = %d.\n    has_q_233 = %d.\n    has_q_134 = %d.\n", info.has_q_112,
                info.has_q_123, info.has_q_233, info.has_q_134);
  std::printf("    ANN = %d-%d-1.\n", info.dim, info.num_neurons);
  engine_ = nepmi_engine_create(model_, num_atoms, nullptr);
  if (!engine_) {
    std::printf("NEP: %s\n", nepmi_last_error());
    std::exit(1);
  }
}

NEP_MI::~NEP_MI()
{
  nepmi_engine_destroy(engine_);
  nepmi_model_free(model_);
}

void NEP_MI::compute(
  Box& box, const GPU_Vector<int>& type, const GPU_Vector<double>& position, GPU_Vector<double>& potential,
  GPU_Vector<double>& force, GPU_Vector<double>& virial)
{
  const int pbc[3] = {box.pbc_x, box.pbc_y, box.pbc_z};
  die_on(
    nepmi_potential_compute(
      engine_, box.cpu_h, pbc, (int64_t)type.size(), type.data(), position.data(), potential.data(), force.data(),
      virial.data()),
    "NEP::compute");
}

void NEP_MI::write_neighbor_out() const
{
  nepmi_stats st;
  if (nepmi_engine_stats(engine_, 1, &st) < 0)
    return;
  FILE* fid = std::fopen("neighbor.out", "a");
  if (!fid)
    return;
  std::fprintf(fid, "Neighbor info at step %lld: radial(max=%d,actual=%d), angular(max=%d,actual=%d)\n",
               (long long)st.num_compute, 0, st.max_nn_radial, 0, st.max_nn_angular);
  std::fclose(fid);
}

std::vector<std::string> potential_elements(const std::string& file_potential)
{
  std::ifstream in(file_potential);
  if (!in)
    input_error("Failed to open " + file_potential + ".");
  std::string line;
  std::getline(in, line);
  auto tok = get_tokens(line);
  if (tok.size() < 3)
    input_error("The first line of the potential file should have at least 3 items.");
  const int n = std::atoi(tok[1].c_str());
  if ((int)tok.size() != 2 + n)
    input_error("The first line of the potential file should have " + std::to_string(n) + " atom symbols.");
  return std::vector<std::string>(tok.begin() + 2, tok.end());
}

void Force::parse_potential(const std::vector<std::string>& param, const Box&, int number_of_atoms, bool create)
{
  if (param.size() != 2 && param.size() != 3)
    input_error("potential should have 1 or 2 parameters.");
  std::ifstream in(param[1]);
  if (!in)
    input_error("Failed to open " + param[1] + ".");
  std::string name;
  in >> name;
  if (name.rfind("nep", 0) != 0 && name != "tersoff_1989") // both live in libnepmi.so
    input_error("illegal potential model: " + name + " (this host carries NEP only; see DESIGN.md section 8).");
  // check_types (force.cu:55-73): every further potential must list the same species in the same order
  const std::vector<std::string> types = potential_elements(param[1]);
  if (num_potentials_ == 0)
    atom_types_ = types;
  else if (types != atom_types_)
    input_error("The atomic species and/or the order of the species are not consistent between the multiple "
                "potentials.");
  if (create) // --check-input stops short of the device
    potentials.emplace_back(new NEP_MI(param[1].c_str(), number_of_atoms));
  ++num_potentials_;
  has_non_nep_ = has_non_nep_ || name.rfind("nep", 0) != 0;
  if (num_potentials_ > 1 && has_non_nep_) // force.cu:213-217
    input_error("Multiple potentials may only be used with NEP potentials.");
}

nepmi_engine* Force::engine() const
{
  auto* p = dynamic_cast<NEP_MI*>(potentials.empty() ? nullptr : potentials[0].get());
  return p ? p->engine() : nullptr;
}

bool Force::has_temperature_model() const
{
  for (auto& p : potentials)
    if (auto* q = dynamic_cast<NEP_MI*>(p.get()))
      if (q->nep_model_type == 3)
        return true;
  return false;
}

void Force::compute(
  Box& box, GPU_Vector<double>& position, GPU_Vector<int>& type, GPU_Vector<double>& potential,
  GPU_Vector<double>& force, GPU_Vector<double>& virial)
{
  if (potentials.empty())
    input_error("No potential is defined.");
  const int pbc[3] = {box.pbc_x, box.pbc_y, box.pbc_z};
  const int64_t n = (int64_t)type.size();
  nepmi_engine* e = engine();
  die_on(nepmi_apply_pbc(e, box.cpu_h, pbc, n, position.data()), "gpu_apply_pbc");
  die_on(nepmi_zero_properties(e, n, potential.data(), force.data(), virial.data()), "initialize_properties");
  // potentials[i]->compute(temperature, ...) for temperature-dependent NEP models (force.cu:516-562)
  for (auto& p : potentials)
    if (auto* q = dynamic_cast<NEP_MI*>(p.get()))
      if (q->nep_model_type == 3)
        die_on(nepmi_engine_set_temperature(q->engine(), temperature), "set_temperature");
  if (multiple_potentials_mode_ == "observe") { // the main potential only
    potentials[0]->compute(box, type, position, potential, force, virial);
  } else if (multiple_potentials_mode_ == "average") { // every compute adds; then one division
    for (auto& p : potentials)
      p->compute(box, type, position, potential, force, virial);
    die_on(nepmi_average_properties(e, n, (double)potentials.size(), potential.data(), force.data(), virial.data()),
           "gpu_average_properties");
  } else {
    input_error("Invalid mode for multiple potentials.");
  }
}

} // namespace gmi
