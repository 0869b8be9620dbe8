#include <cerrno>
#include "run.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <random>
#include <cstring>
#include <fstream>

namespace gmi {

// is_valid_real (src/utilities/read_file.cu): the whole token must be one floating-point number
static bool is_valid_real(const std::string& tok, double* result)
{
  if (tok.empty())
    return false;
  char* end = nullptr;
  errno = 0;
  const double v = std::strtod(tok.c_str(), &end);
  if (errno != 0 || end == tok.c_str() || *end != 0)
    return false;
  *result = v;
  return true;
}

// is_valid_int (src/utilities/read_file.cu:26-41): the whole token must be one integer (any strtol base)
static bool is_valid_int(const std::string& tok, int* result)
{
  if (tok.empty())
    return false;
  char* end = nullptr;
  errno = 0;
  const long v = std::strtol(tok.c_str(), &end, 0);
  if (errno != 0 || end == tok.c_str() || *end != 0)
    return false;
  *result = (int)v;
  return true;
}

static std::vector<std::string> strip(const std::string& line)
{
  // tokens up to a '#' comment (run.cu:188-200)
  std::vector<std::string> t = get_tokens(line), out;
  for (auto& w : t) {
    if (!w.empty() && w[0] == '#')
      break;
    out.push_back(w);
  }
  return out;
}

static void die_on(int status, const char* where)
{
  if (status < 0) {
    std::printf("%s: %s\n", where, nepmi_last_error());
    std::exit(1);
  }
}

Run::~Run()
{
  if (nhc_state_)
    (void)hipFree(nhc_state_);
  if (dist_)
    nepmi_dist_destroy(dist_);
  if (dist_model_)
    nepmi_model_free(dist_model_);
  if (have_rccl_)
    nepmi_transport_destroy(&rccl_);
  if (have_boot_)
    nepmi_transport_destroy(&boot_);
}

Run::Run(bool check_only, const Parallel& par) : check_only_(check_only), par_(par)
{
  // initialize_position (src/model/read_xyz.cu:427-530): run.in is scanned first for the
  // potential file (its element list fixes the type indices), then model.xyz is read.
  std::ifstream in("run.in");
  if (!in)
    input_error("Cannot open run.in.");
  std::string line;
  while (std::getline(in, line)) {
    auto t = strip(line);
    if (!t.empty() && t[0] == "potential") {
      if (t.size() < 2)
        input_error("potential should have 1 or 2 parameters.");
      potential_file = t[1];
      break;
    }
  }
  if (potential_file.empty())
    input_error("There is no 'potential' keyword before run.");
  elements = potential_elements(potential_file);
  has_velocity_in_xyz = read_xyz("model.xyz", elements, box, atom, groups);
  if (!has_velocity_in_xyz)
    initialize_velocity(initial_temperature, false, 0, atom); // default 300 K (run.cu:155-160)
  thermo.resize(check_only_ ? 0 : 12);
}

void Run::execute_run_in()
{
  std::ifstream in("run.in");
  std::string line;
  while (std::getline(in, line)) {
    auto t = strip(line);
    if (!t.empty())
      parse_one_keyword(t);
  }
}

void Run::parse_one_keyword(const std::vector<std::string>& p)
{
  const std::string& k = p[0];
  if (k == "replicate") {
    if (p.size() != 4)
      input_error("replicate should have 3 parameters.");
    if (gpu_allocated)
      input_error("replicate should be put before potential.");
    const int n[3] = {std::atoi(p[1].c_str()), std::atoi(p[2].c_str()), std::atoi(p[3].c_str())};
    if (n[0] < 1 || n[1] < 1 || n[2] < 1)
      input_error("replicate numbers should be >= 1.");
    // like Replicate() (replicate.cu:51-72) the velocities of the cell are repeated with it: no new draw from the
    // rand() stream here, so a later `velocity` keyword sees the stream where the reference's does
    replicate(n, box, atom, groups);
  } else if (k == "potential") {
    if (check_only_) {
      std::printf("potential %s (elements:", p[1].c_str());
      for (auto& e : elements)
        std::printf(" %s", e.c_str());
      std::printf(")\n");
      force.parse_potential(p, box, atom.number_of_atoms, false);
      return;
    }
    if (p.size() == 3 && p[2] != "x" && p[2] != "y" && p[2] != "z") // force.cu:122-123: the partition direction
      input_error("The partition direction of a multi-GPU potential should be x, y or z.");
    if (!gpu_allocated) {
      atom.allocate_gpu();
      gpu_allocated = true;
    }
    if (par_.world > 1) {
      setup_dist(p);
      return;
    }
    if (p.size() == 3)
      std::printf("    (one GPU: the partition direction %s is not used)\n", p[2].c_str());
    force.parse_potential(std::vector<std::string>(p.begin(), p.begin() + 2), box, atom.number_of_atoms);
  } else if (k == "velocity") {
    if (p.size() != 2 && p.size() != 4)
      input_error("velocity should have 1 or 2 parameters.");
    initial_temperature = std::atof(p[1].c_str());
    if (initial_temperature <= 0.0)
      input_error("initial temperature should be a positive number.");
    const bool use_seed = p.size() == 4;
    const int seed = use_seed ? std::atoi(p[3].c_str()) : 0;
    if (!has_velocity_in_xyz) {
      initialize_velocity(initial_temperature, use_seed, seed, atom);
      std::printf("Initialized velocities with input T = %g K.\n", initial_temperature);
    }
    if (gpu_allocated)
      atom.velocity_per_atom.copy_from_host(atom.cpu_velocity_per_atom.data());
  } else if (k == "correct_velocity") { // Run::parse_correct_velocity, run.cu:610-647
    std::printf("Correct linear and angular momenta.\n");
    if (p.size() != 2 && p.size() != 3)
      input_error("correct_velocity should have 1 or 2 parameters.");
    if (!is_valid_int(p[1], &correct_interval_))
      input_error("velocity correction interval should be an integer.");
    if (correct_interval_ < 10)
      input_error("velocity correction interval should >= 10.");
    std::printf("    every %d steps.\n", correct_interval_);
    correct_group_method_ = -1;
    if (p.size() == 3) {
      if (!is_valid_int(p[2], &correct_group_method_))
        input_error("velocity correction group method should be an integer.");
      if (correct_group_method_ < 0)
        input_error("grouping method should >= 0.");
      if (correct_group_method_ >= (int)groups.size())
        input_error("grouping method should < maximum number of grouping methods.");
      std::printf("    for individual groups in group method %d.\n", correct_group_method_);
    } else {
      std::printf("    for the whole system.\n");
    }
  } else if (k == "ensemble") {
    if (p.size() < 2)
      input_error("ensemble should have at least 1 parameter.");
    if (p[1] == "nve") {
      std::printf("Use NVE ensemble for this run.\n");
    } else if (p[1] == "nvt_ber" || p[1] == "nvt_nhc" || p[1] == "nvt_bdp" || p[1] == "nvt_lan" || p[1] == "nvt_bao") { // Integrate::parse_ensemble, integrate.cu:424-437, 569-600
      if (p.size() != 5)
        input_error("ensemble " + p[1] + " should have 3 parameters.");
      temperature1 = std::atof(p[2].c_str());
      temperature2 = std::atof(p[3].c_str());
      temperature_coupling = std::atof(p[4].c_str());
      if (temperature1 <= 0.0 || temperature2 <= 0.0)
        input_error("Temperature should > 0.");
      if (temperature_coupling < 1.0)
        input_error("Temperature coupling should >= 1.");
      std::printf("Use NVT ensemble for this run.\n    choose the %s method.\n    initial temperature is %g K.\n"
                  "    final temperature is %g K.\n    tau_T is %g time_step.\n",
                  p[1] == "nvt_ber" ? "Berendsen" : p[1] == "nvt_nhc" ? "Nose-Hoover chain" : p[1] == "nvt_lan" ? "Langevin" : p[1] == "nvt_bao" ? "BAOAB Langevin" : "Bussi-Donadio-Parrinello",
                  temperature1, temperature2, temperature_coupling);
    } else {
      input_error("ensemble " + p[1] + " is not available in gpumd-mi yet (nve, nvt_ber, nvt_nhc, nvt_bdp, nvt_lan, nvt_bao; DESIGN.md section 8).");
    }
    ensemble = p[1];
  } else if (k == "time_step") {
    if (p.size() != 2)
      input_error("time_step should have 1 parameter.");
    time_step = std::atof(p[1].c_str());
    std::printf("Time step for this run is %g fs.\n", time_step);
    time_step /= TIME_UNIT_CONVERSION;
  } else if (k == "dump_thermo") {
    if (p.size() != 2)
      input_error("dump_thermo should have 1 parameter.");
    dump_thermo_interval = std::atoi(p[1].c_str());
    if (dump_thermo_interval <= 0)
      input_error("thermo dump interval should > 0.");
    std::printf("Dump thermo every %d steps.\n", dump_thermo_interval);
  } else if (k == "dump_restart") {
    if (p.size() != 2)
      input_error("dump_restart should have 1 parameter.");
    dump_restart_interval = std::atoi(p[1].c_str());
  } else if (k == "dump_position") { // run.cu:396-431: the keywords dump_xyz replaced name their successor
    input_error("dump_position has been removed. Use dump_xyz <interval> <filename> instead.");
  } else if (k == "dump_velocity") {
    input_error("dump_velocity has been removed. Use dump_xyz <interval> <filename> velocity instead.");
  } else if (k == "dump_force") {
    input_error("dump_force has been removed. Use dump_xyz <interval> <filename> force instead.");
  } else if (k == "dump_exyz") {
    input_error("dump_exyz has been removed. Use dump_xyz <interval> <filename> velocity force potential instead.");
  } else if (k == "dump_xyz") { // Dump_XYZ::parse, dump_xyz.cu:70-155
    std::printf("Dump extended XYZ.\n");
    if (p.size() < 3)
      input_error("dump_xyz should have at least 2 parameters.");
    int scratch;
    if (p.size() >= 4 && is_valid_int(p[2], &scratch) && is_valid_int(p[3], &scratch))
      input_error("dump_xyz no longer takes <grouping_method> <group_id> as its first two parameters. Use dump_xyz "
                  "<interval> <filename> [group <grouping_method> <group_id>] instead.");
    DumpXyz d;
    if (!is_valid_int(p[1], &d.interval))
      input_error("dump interval should be an integer.");
    if (d.interval <= 0)
      input_error("dump interval should > 0.");
    std::printf("    every %d steps.\n", d.interval);
    std::printf("    into file %s.\n", p[2].c_str());
    d.filename = p[2];
    if (d.filename.back() == '*') {
      d.separated = true;
      d.filename.pop_back();
    }
    bool group_seen = false, precision_seen = false;
    auto quantity = [&](bool& flag, const std::string& name, const char* what) { // set_quantity, parse_utilities.cu:83-95
      if (flag)
        input_error("Quantity '" + name + "' is specified more than once in dump_xyz.");
      flag = true;
      std::printf("    has %s.\n", what);
    };
    for (size_t m = 3; m < p.size(); ++m) {
      if (p[m] == "group") { // parse_group, parse_utilities.cu:27-64
        if (group_seen)
          input_error("Option 'group' is specified more than once in dump_xyz.");
        int probe;
        if (m + 2 >= p.size() || !is_valid_int(p[m + 1], &probe) || !is_valid_int(p[m + 2], &probe))
          input_error("Option 'group' should be followed by a grouping method and a group ID. The quantity that "
                      "writes group labels as a column is now called 'group_labels'.");
        d.grouping_method = std::atoi(p[m + 1].c_str());
        d.group_id = std::atoi(p[m + 2].c_str());
        if (d.grouping_method < 0)
          input_error("Grouping method should >= 0.");
        if (d.grouping_method >= (int)groups.size())
          input_error("Grouping method should < number of grouping methods.");
        if (d.group_id >= groups[d.grouping_method].number)
          input_error("Group ID should < number of groups.");
        if (d.group_id < 0)
          input_error("group ID should >= 0.");
        std::printf("    grouping method is %d and group ID is %d.\n", d.grouping_method, d.group_id);
        group_seen = true;
        m += 2;
      } else if (p[m] == "precision") { // parse_precision, parse_utilities.cu:66-81
        if (precision_seen)
          input_error("Option 'precision' is specified more than once in dump_xyz.");
        if (m + 1 >= p.size())
          input_error("Not enough arguments for option 'precision'.");
        if (p[m + 1] == "single") {
          d.precision = 1;
          std::printf("    with single precision.\n");
        } else if (p[m + 1] == "double") {
          d.precision = 2;
          std::printf("    with double precision.\n");
        } else {
          input_error("Invalid precision.");
        }
        precision_seen = true;
        ++m;
      } else if (p[m] == "velocity") quantity(d.has_velocity, p[m], "velocity");
      else if (p[m] == "force") quantity(d.has_force, p[m], "force");
      else if (p[m] == "potential") quantity(d.has_potential, p[m], "potential");
      else if (p[m] == "unwrapped_position") quantity(d.has_unwrapped_position, p[m], "unwrapped position");
      else if (p[m] == "mass") quantity(d.has_mass, p[m], "mass");
      else if (p[m] == "charge") quantity(d.has_charge, p[m], "charge specified in model.xyz");
      else if (p[m] == "virial") quantity(d.has_virial, p[m], "virial");
      else if (p[m] == "bec") input_error("Cannot output BEC for a non-NEP-charge model.");
      else if (p[m] == "group_labels") {
        if (groups.empty())
          input_error("Cannot output group labels without a grouping method defined in model.xyz.");
        quantity(d.has_group_labels, p[m], "group labels");
      } else input_error("Unrecognized argument in dump_xyz.");
    }
    if (d.grouping_method < 0)
      std::printf("    for the whole system.\n");
    dump_xyzs.push_back(d);
  } else if (k == "dump_observer") { // Dump_Observer::parse, dump_observer.cu:82-139
    std::printf("Dump observer.\n");
    if (p.size() != 6)
      input_error("dump_observer should have 5 parameters.");
    DumpObserver o;
    o.mode = p[1];
    if (o.mode != "observe" && o.mode != "average")
      input_error("observer mode should be 'observe' or 'average'");
    if (!is_valid_int(p[2], &o.interval_thermo))
      input_error("dump interval thermo should be an integer.");
    if (o.interval_thermo <= 0)
      input_error("dump interval thermo should > 0.");
    if (!is_valid_int(p[3], &o.interval_exyz))
      input_error("dump interval exyz should be an integer.");
    if (o.interval_exyz <= 0)
      input_error("dump interval exyz should > 0.");
    std::printf("    .out every %d steps.\n    .exyz every %d steps.\n", o.interval_thermo, o.interval_exyz);
    if (!is_valid_int(p[4], &o.has_velocity))
      input_error("has_velocity should be an integer.");
    std::printf(o.has_velocity ? "    with velocity data.\n" : "    without velocity data.\n");
    if (!is_valid_int(p[5], &o.has_force))
      input_error("has_force should be an integer.");
    std::printf(o.has_force ? "    with force data.\n" : "    without force data.\n");
    if (o.mode == "observe")
      std::printf("    evaluate all potentials, dumping .out every %d and .exyz every %d steps.\n", o.interval_thermo,
                  o.interval_exyz);
    else
      std::printf("    use the average potential in the molecular dynamics run, and dump .out every %d and .exyz every "
                  "%d steps.\n", o.interval_thermo, o.interval_exyz);
    o.active = true;
    observer = o;
  } else if (k == "active") { // Active::parse, active.cu:117-169
    std::printf("Active learning.\n");
    if (p.size() != 6)
      input_error("active should have 5 parameters.");
    ActiveLearning a;
    if (!is_valid_int(p[1], &a.interval))
      input_error("check interval should be an integer.");
    if (a.interval <= 0)
      input_error("check interval should > 0.");
    std::printf("    check uncertainty every %d steps.\n", a.interval);
    if (!is_valid_int(p[2], &a.has_velocity))
      input_error("has_velocity should be an integer.");
    std::printf(a.has_velocity ? "    with velocity data.\n" : "    without velocity data.\n");
    if (!is_valid_int(p[3], &a.has_force))
      input_error("has_force should be an integer.");
    std::printf(a.has_force ? "    with force data.\n" : "    without force data.\n");
    if (!is_valid_int(p[4], &a.has_uncertainty))
      input_error("has_uncertainty should be an integer.");
    std::printf(a.has_uncertainty ? "    with per-atom uncertainty data.\n" : "    without per-atom uncertainty data.\n");
    if (!is_valid_real(p[5], &a.threshold))
      input_error("threshold should be a real number.\n");
    std::printf("    will check if uncertainties exceed %f every %d iterations.\n", a.threshold, a.interval);
    a.active = true;
    active_ = a;
  } else if (k == "run") {
    if (p.size() != 2)
      input_error("run should have 1 parameter.");
    number_of_steps = std::atoi(p[1].c_str());
    std::printf("Run %d steps.\n", number_of_steps);
    if (check_only_) {
      std::printf("(--check-input: %d atoms, dt = %g fs, no GPU work)\n", atom.number_of_atoms, time_step * TIME_UNIT_CONVERSION);
    } else {
      perform_a_run();
    }
    // properties do not propagate to the next run (measure.cu:88)
    dump_thermo_interval = 0;
    dump_restart_interval = 0;
    dump_xyzs.clear();
    observer.active = false;
    active_.active = false;
  } else {
    input_error("'" + k + "' is invalid keyword (or outside the path gpumd-mi covers).");
  }
}

void Run::find_thermo()
{
  die_on(nepmi_find_thermo(
           force.engine(), atom.number_of_atoms, box.get_volume(), atom.mass.data(), atom.potential_per_atom.data(),
           atom.velocity_per_atom.data(), atom.virial_per_atom.data(), thermo.data()),
         "find_thermo");
}

// Dump_Thermo (src/measure/dump_thermo.cu:56-132)
void Run::dump_thermo(int step)
{
  if (dump_thermo_interval <= 0 || (step + 1) % dump_thermo_interval != 0)
    return;
  static FILE* fid = nullptr;
  static int header_for = -1;
  if (!fid)
    fid = std::fopen("thermo.out", "a");
  if (header_for != number_of_steps + dump_thermo_interval * 1000003) {
    header_for = number_of_steps + dump_thermo_interval * 1000003;
  }
  double t[8];
  thermo.copy_to_host(t, 8);
  const double ke = 1.5 * atom.number_of_atoms * K_B * t[0];
  std::fprintf(fid, "%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e", t[0], ke, t[1],
               t[2] * PRESSURE_UNIT_CONVERSION, t[3] * PRESSURE_UNIT_CONVERSION, t[4] * PRESSURE_UNIT_CONVERSION,
               t[7] * PRESSURE_UNIT_CONVERSION, t[6] * PRESSURE_UNIT_CONVERSION, t[5] * PRESSURE_UNIT_CONVERSION);
  const double* h = box.cpu_h;
  std::fprintf(fid, "%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e\n", h[0], h[3], h[6], h[1], h[4],
               h[7], h[2], h[5], h[8]);
  std::fflush(fid);
}

// Dump_XYZ::output_line2 + process (src/measure/dump_xyz.cu:196-440)
void Run::dump_xyz(DumpXyz& d, int step)
{
  if ((step + 1) % d.interval != 0)
    return;
  if (d.separated) // one frame per file, named by the step it belongs to
    d.fid = std::fopen((d.filename + std::to_string(step + 1)).c_str(), "w");
  else if (!d.fid)
    d.fid = std::fopen(d.filename.c_str(), "a");
  if (!d.fid)
    input_error("Cannot open " + d.filename + ".");
  const int N = atom.number_of_atoms;
  const char* fmt = d.precision == 1 ? " %.9g" : " %.17g";
  std::vector<double> pos(3 * (size_t)N), vel, frc, pe, unw, vir(9 * (size_t)N);
  atom.position_per_atom.copy_to_host(pos.data());
  atom.virial_per_atom.copy_to_host(vir.data());
  if (d.has_velocity) { vel.resize(3 * (size_t)N); atom.velocity_per_atom.copy_to_host(vel.data()); }
  if (d.has_force) { frc.resize(3 * (size_t)N); atom.force_per_atom.copy_to_host(frc.data()); }
  if (d.has_potential) { pe.resize(N); atom.potential_per_atom.copy_to_host(pe.data()); }
  if (d.has_unwrapped_position) { unw.resize(3 * (size_t)N); atom.unwrapped_position.copy_to_host(unw.data()); }
  double t[8];
  thermo.copy_to_host(t, 8);
  double tv[6] = {0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 6; ++c)
    for (int n = 0; n < N; ++n)
      tv[c] += vir[(size_t)c * N + n];
  auto tensor = [&](const char* name, const double* v) {
    std::fprintf(d.fid, " %s=\"", name);
    for (int k = 0; k < 9; ++k)
      std::fprintf(d.fid, k == 0 ? fmt + 1 : fmt, v[k]);
    std::fprintf(d.fid, "\"");
  };
  // the header always describes the whole system, also when only one group's atoms follow
  const int num_dump = d.grouping_method >= 0 ? groups[d.grouping_method].cpu_size[d.group_id] : N;
  const int* contents =
    d.grouping_method >= 0
      ? groups[d.grouping_method].cpu_contents.data() + groups[d.grouping_method].cpu_size_sum[d.group_id]
      : nullptr;
  std::fprintf(d.fid, "%d\n", num_dump);
  std::fprintf(d.fid, "Time=%.8f", global_time * TIME_UNIT_CONVERSION);
  std::fprintf(d.fid, " pbc=\"%c %c %c\"", box.pbc_x ? 'T' : 'F', box.pbc_y ? 'T' : 'F', box.pbc_z ? 'T' : 'F');
  const double* h = box.cpu_h;
  const double lattice[9] = {h[0], h[3], h[6], h[1], h[4], h[7], h[2], h[5], h[8]};
  tensor("Lattice", lattice);
  std::fprintf(d.fid, " energy=");
  std::fprintf(d.fid, fmt + 1, t[1]);
  const double virial[9] = {tv[0], tv[3], tv[4], tv[3], tv[1], tv[5], tv[4], tv[5], tv[2]};
  tensor("virial", virial);
  const double stress[9] = {t[2], t[5], t[6], t[5], t[3], t[7], t[6], t[7], t[4]};
  tensor("stress", stress);
  std::fprintf(d.fid, " Properties=species:S:1:pos:R:3");
  if (d.has_mass) std::fprintf(d.fid, ":mass:R:1");
  if (d.has_charge) std::fprintf(d.fid, ":charge:R:1");
  if (d.has_velocity) std::fprintf(d.fid, ":vel:R:3");
  if (d.has_force) std::fprintf(d.fid, ":forces:R:3");
  if (d.has_potential) std::fprintf(d.fid, ":energy_atom:R:1");
  if (d.has_unwrapped_position) std::fprintf(d.fid, ":unwrapped_position:R:3");
  if (d.has_virial) std::fprintf(d.fid, ":virial:R:9");
  if (d.has_group_labels) std::fprintf(d.fid, ":group:I:%d", (int)groups.size());
  std::fprintf(d.fid, "\n");
  const int vidx[9] = {0, 3, 4, 6, 1, 5, 7, 8, 2}; // dump_xyz.cu: xx xy xz yx yy yz zx zy zz
  for (int k = 0; k < num_dump; ++k) {
    const int n = contents ? contents[k] : k;
    std::fprintf(d.fid, "%s", atom.cpu_atom_symbol[n].c_str());
    for (int c = 0; c < 3; ++c) std::fprintf(d.fid, fmt, pos[n + (size_t)N * c]);
    if (d.has_mass) std::fprintf(d.fid, fmt, atom.cpu_mass[n]);
    if (d.has_charge) std::fprintf(d.fid, fmt, atom.cpu_charge[n]);
    if (d.has_velocity)
      for (int c = 0; c < 3; ++c) std::fprintf(d.fid, fmt, vel[n + (size_t)N * c] / TIME_UNIT_CONVERSION);
    if (d.has_force)
      for (int c = 0; c < 3; ++c) std::fprintf(d.fid, fmt, frc[n + (size_t)N * c]);
    if (d.has_potential) std::fprintf(d.fid, fmt, pe[n]);
    if (d.has_unwrapped_position)
      for (int c = 0; c < 3; ++c) std::fprintf(d.fid, fmt, unw[n + (size_t)N * c]);
    if (d.has_virial)
      for (int c = 0; c < 9; ++c) std::fprintf(d.fid, fmt, vir[n + (size_t)N * vidx[c]]);
    if (d.has_group_labels)
      for (const auto& g : groups) std::fprintf(d.fid, " %d", g.cpu_label[n]);
    std::fprintf(d.fid, "\n");
  }
  if (d.separated) {
    std::fclose(d.fid);
    d.fid = nullptr;
  } else {
    std::fflush(d.fid);
  }
}

// Dump_Observer::preprocess (dump_observer.cu:141-166): the mode reaches Force before the first force call
void Run::dump_observer_open()
{
  if (!observer.active)
    return;
  force.set_multiple_potentials_mode(observer.mode);
  const int files = observer.mode == "observe" ? (int)force.potentials.size() : 1;
  for (int i = 0; i < files; ++i) {
    const std::string number = files == 1 ? "" : std::to_string(i);
    observer.exyz_files.push_back(std::fopen(("observer" + number + ".xyz").c_str(), "a"));
    observer.thermo_files.push_back(std::fopen(("observer" + number + ".out").c_str(), "a"));
    if (!observer.exyz_files.back() || !observer.thermo_files.back())
      input_error("Cannot open the observer files.");
  }
}

// Dump_Observer::process (dump_observer.cu:168-262)
void Run::dump_observer_process(int step)
{
  if (!observer.active)
    return;
  if ((step + 1) % observer.interval_thermo != 0 && (step + 1) % observer.interval_exyz != 0)
    return;
  if (observer.mode == "observe") {
    // every potential on the current (already wrapped) coordinates; the main one last, so that the arrays
    // the integrator continues with hold its values again
    const int64_t N = atom.number_of_atoms;
    for (int k = (int)force.potentials.size() - 1; k >= 0; --k) {
      if (nepmi_zero_properties(force.engine(), N, atom.potential_per_atom.data(), atom.force_per_atom.data(),
                                atom.virial_per_atom.data()) != NEPMI_OK)
        input_error(nepmi_last_error());
      force.potentials[k]->compute(box, atom.type, atom.position_per_atom, atom.potential_per_atom, atom.force_per_atom,
                                   atom.virial_per_atom);
      find_thermo();
      dump_observer_write(step, k);
    }
  } else { // average: what the run itself computed
    dump_observer_write(step, 0);
  }
}

// write_exyz + write_thermo (dump_observer.cu:264-442): fixed %.8f columns, unlike dump_xyz
void Run::dump_observer_write(int step, int file_index)
{
  const int N = atom.number_of_atoms;
  double t[8];
  thermo.copy_to_host(t, 8);
  const double* h = box.cpu_h;
  if ((step + 1) % observer.interval_exyz == 0) {
    FILE* fid = observer.exyz_files[file_index];
    std::vector<double> pos(3 * (size_t)N), vel, frc, vir(9 * (size_t)N);
    atom.position_per_atom.copy_to_host(pos.data());
    atom.virial_per_atom.copy_to_host(vir.data());
    if (observer.has_velocity) { vel.resize(3 * (size_t)N); atom.velocity_per_atom.copy_to_host(vel.data()); }
    if (observer.has_force) { frc.resize(3 * (size_t)N); atom.force_per_atom.copy_to_host(frc.data()); }
    double tv[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < 6; ++c)
      for (int n = 0; n < N; ++n)
        tv[c] += vir[(size_t)c * N + n];
    std::fprintf(fid, "%d\n", N);
    std::fprintf(fid, "Time=%.8f", global_time * TIME_UNIT_CONVERSION);
    std::fprintf(fid, " pbc=\"%c %c %c\"", box.pbc_x ? 'T' : 'F', box.pbc_y ? 'T' : 'F', box.pbc_z ? 'T' : 'F');
    std::fprintf(fid, " Lattice=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", h[0], h[3], h[6], h[1], h[4], h[7], h[2],
                 h[5], h[8]);
    std::fprintf(fid, " energy=%.8f", t[1]);
    std::fprintf(fid, " virial=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", tv[0], tv[3], tv[4], tv[3], tv[1], tv[5],
                 tv[4], tv[5], tv[2]);
    std::fprintf(fid, " stress=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", t[2], t[5], t[6], t[5], t[3], t[7], t[6],
                 t[7], t[4]);
    std::fprintf(fid, " Properties=species:S:1:pos:R:3");
    if (observer.has_velocity) std::fprintf(fid, ":vel:R:3");
    if (observer.has_force) std::fprintf(fid, ":forces:R:3");
    std::fprintf(fid, "\n");
    for (int n = 0; n < N; ++n) {
      std::fprintf(fid, "%s", atom.cpu_atom_symbol[n].c_str());
      for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", pos[n + (size_t)N * c]);
      if (observer.has_velocity)
        for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", vel[n + (size_t)N * c] / TIME_UNIT_CONVERSION);
      if (observer.has_force)
        for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", frc[n + (size_t)N * c]);
      std::fprintf(fid, "\n");
    }
    std::fflush(fid);
  }
  if ((step + 1) % observer.interval_thermo == 0) {
    FILE* fid = observer.thermo_files[file_index];
    const double ke = 1.5 * N * K_B * t[0];
    std::fprintf(fid, "%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e", t[0], ke, t[1],
                 t[2] * PRESSURE_UNIT_CONVERSION, t[3] * PRESSURE_UNIT_CONVERSION, t[4] * PRESSURE_UNIT_CONVERSION,
                 t[7] * PRESSURE_UNIT_CONVERSION, t[6] * PRESSURE_UNIT_CONVERSION, t[5] * PRESSURE_UNIT_CONVERSION);
    std::fprintf(fid, "%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e%20.10e\n", h[0], h[3], h[6], h[1], h[4],
                 h[7], h[2], h[5], h[8]);
    std::fflush(fid);
  }
}

// Active::preprocess (active.cu:171-199): always "observe" -- the run follows the first potential, the others are only
// evaluated at the check steps
void Run::active_open()
{
  if (!active_.active)
    return;
  force.set_multiple_potentials_mode("observe");
  active_.exyz_file = std::fopen("active.xyz", "a");
  active_.out_file = std::fopen("active.out", "a");
  if (!active_.exyz_file || !active_.out_file)
    input_error("Cannot open the active learning files.");
}

// Active::process (active.cu:201-289): every potential on the current coordinates, the main one last; per atom and
// Cartesian direction the mean and the mean square of the M forces (each term divided by M before it is added, as the
// reference's compute_mean does), uncertainty_i = sqrt(sum_d (<f_d^2> - <f_d>^2)), sigma_f = max_i; time and sigma_f go to
// active.out, the structure to active.xyz when sigma_f exceeds the threshold (write_exyz / output_line2, active.cu:305-445)
void Run::active_process(int step)
{
  if (!active_.active || (step + 1) % active_.interval != 0)
    return;
  const int N = atom.number_of_atoms;
  const int M = (int)force.potentials.size();
  std::vector<double> mean(3 * (size_t)N, 0.0), mean_sq(3 * (size_t)N, 0.0), f(3 * (size_t)N), unc((size_t)N);
  for (int k = M - 1; k >= 0; --k) {
    if (nepmi_zero_properties(force.engine(), N, atom.potential_per_atom.data(), atom.force_per_atom.data(),
                              atom.virial_per_atom.data()) != NEPMI_OK)
      input_error(nepmi_last_error());
    force.potentials[k]->compute(box, atom.type, atom.position_per_atom, atom.potential_per_atom, atom.force_per_atom,
                                 atom.virial_per_atom);
    atom.force_per_atom.copy_to_host(f.data());
    for (size_t i = 0; i < 3 * (size_t)N; ++i) {
      mean[i] += f[i] / M;
      mean_sq[i] += f[i] * f[i] / M;
    }
  }
  double uncertainty = -1.0;
  for (int n = 0; n < N; ++n) {
    double var = 0.0;
    for (int d = 0; d < 3; ++d) {
      const size_t i = (size_t)d * N + n;
      var += mean_sq[i] - mean[i] * mean[i];
    }
    unc[n] = std::sqrt(var);
    if (uncertainty < unc[n])
      uncertainty = unc[n];
  }
  std::fprintf(active_.out_file, "%20.10e%20.10e\n", global_time * TIME_UNIT_CONVERSION, uncertainty);
  std::fflush(active_.out_file);
  if (!(uncertainty > active_.threshold))
    return;
  // the arrays hold the main potential's values again (it was evaluated last)
  find_thermo();
  double t[8];
  thermo.copy_to_host(t, 8);
  const double* h = box.cpu_h;
  FILE* fid = active_.exyz_file;
  std::vector<double> pos(3 * (size_t)N), vel, vir(9 * (size_t)N);
  atom.position_per_atom.copy_to_host(pos.data());
  atom.virial_per_atom.copy_to_host(vir.data());
  if (active_.has_velocity) { vel.resize(3 * (size_t)N); atom.velocity_per_atom.copy_to_host(vel.data()); }
  double tv[6] = {0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 6; ++c)
    for (int n = 0; n < N; ++n)
      tv[c] += vir[(size_t)c * N + n];
  std::fprintf(fid, "%d\n", N);
  std::fprintf(fid, "Time=%.8f", global_time * TIME_UNIT_CONVERSION);
  std::fprintf(fid, " pbc=\"%c %c %c\"", box.pbc_x ? 'T' : 'F', box.pbc_y ? 'T' : 'F', box.pbc_z ? 'T' : 'F');
  std::fprintf(fid, " uncertainty=%.8f", uncertainty);
  std::fprintf(fid, " Lattice=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", h[0], h[3], h[6], h[1], h[4], h[7], h[2], h[5],
               h[8]);
  std::fprintf(fid, " energy=%.8f", t[1]);
  std::fprintf(fid, " virial=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", tv[0], tv[3], tv[4], tv[3], tv[1], tv[5], tv[4],
               tv[5], tv[2]);
  std::fprintf(fid, " stress=\"%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\"", t[2], t[5], t[6], t[5], t[3], t[7], t[6], t[7],
               t[4]);
  std::fprintf(fid, " Properties=species:S:1:pos:R:3");
  if (active_.has_velocity) std::fprintf(fid, ":vel:R:3");
  if (active_.has_force) std::fprintf(fid, ":forces:R:3");
  if (active_.has_uncertainty) std::fprintf(fid, ":uncertainty:R:1");
  std::fprintf(fid, "\n");
  for (int n = 0; n < N; ++n) {
    std::fprintf(fid, "%s", atom.cpu_atom_symbol[n].c_str());
    for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", pos[n + (size_t)N * c]);
    if (active_.has_velocity)
      for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", vel[n + (size_t)N * c] * (1.0 / TIME_UNIT_CONVERSION));
    if (active_.has_force)
      for (int c = 0; c < 3; ++c) std::fprintf(fid, " %.8f", f[n + (size_t)N * c]); // (the main potential's: evaluated last)
    if (active_.has_uncertainty) std::fprintf(fid, " %.8f", unc[n]);
    std::fprintf(fid, "\n");
  }
  std::fflush(fid);
}

void Run::active_close()
{
  if (active_.exyz_file) std::fclose(active_.exyz_file);
  if (active_.out_file) std::fclose(active_.out_file);
  active_.exyz_file = active_.out_file = nullptr;
}

void Run::dump_observer_close()
{
  for (FILE* f : observer.exyz_files) std::fclose(f);
  for (FILE* f : observer.thermo_files) std::fclose(f);
  observer.exyz_files.clear();
  observer.thermo_files.clear();
}

// Dump_Restart (src/measure/dump_restart.cu:66-136), full precision instead of %g
void Run::dump_restart(int step)
{
  if (dump_restart_interval <= 0 || (step + 1) % dump_restart_interval != 0)
    return;
  const int N = atom.number_of_atoms;
  std::vector<double> pos(3 * (size_t)N), vel(3 * (size_t)N);
  atom.position_per_atom.copy_to_host(pos.data());
  atom.velocity_per_atom.copy_to_host(vel.data());
  FILE* fid = std::fopen("restart.xyz", "w");
  const double* h = box.cpu_h;
  std::fprintf(fid, "%d\n", N);
  std::fprintf(fid, "pbc=\"%c %c %c\" Lattice=\"%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\" "
                    "Properties=species:S:1:pos:R:3:mass:R:1:vel:R:3",
               box.pbc_x ? 'T' : 'F', box.pbc_y ? 'T' : 'F', box.pbc_z ? 'T' : 'F', h[0], h[3], h[6], h[1], h[4], h[7], h[2],
               h[5], h[8]);
  if (!groups.empty()) // dump_restart.cu:111-115
    std::fprintf(fid, ":group:I:%d", (int)groups.size());
  std::fprintf(fid, "\n");
  for (int n = 0; n < N; ++n) {
    std::fprintf(fid, "%s %.17g %.17g %.17g %.17g %.17g %.17g %.17g", atom.cpu_atom_symbol[n].c_str(), pos[n],
                 pos[n + (size_t)N], pos[n + 2 * (size_t)N], atom.cpu_mass[n], vel[n] / TIME_UNIT_CONVERSION,
                 vel[n + (size_t)N] / TIME_UNIT_CONVERSION, vel[n + 2 * (size_t)N] / TIME_UNIT_CONVERSION);
    for (const auto& g : groups)
      std::fprintf(fid, " %d", g.cpu_label[n]);
    std::fprintf(fid, "\n");
  }
  std::fclose(fid);
}

// Velocity::correct_velocity(step, group, atom), velocity.cu:273-310 (host side, like the reference)
void Run::correct_velocity_now()
{
  const int N = atom.number_of_atoms;
  atom.position_per_atom.copy_to_host(atom.cpu_position_per_atom.data());
  atom.velocity_per_atom.copy_to_host(atom.cpu_velocity_per_atom.data());
  if (correct_group_method_ < 0) {
    correct_velocity(N, atom.cpu_mass, atom.cpu_position_per_atom, atom.cpu_velocity_per_atom, nullptr, 0);
  } else {
    const Group& g = groups[correct_group_method_];
    for (int k = 0; k < g.number; ++k)
      correct_velocity(N, atom.cpu_mass, atom.cpu_position_per_atom, atom.cpu_velocity_per_atom,
                       g.cpu_contents.data() + g.cpu_size_sum[k], g.cpu_size[k]);
  }
  atom.velocity_per_atom.copy_from_host(atom.cpu_velocity_per_atom.data());
}

// `steps` steps of the current ensemble through the fused run loops of libnepmi (state in internal order, steps
// enqueued speculatively); the thermodynamic sums of the last step end up in `thermo`
void Run::run_segment(int steps, double t_a, double t_b)
{
  const int N = atom.number_of_atoms;
  const int pbc[3] = {box.pbc_x, box.pbc_y, box.pbc_z};
  nepmi_engine* e = force.engine();
  const bool ramped_temperature_model = force.has_temperature_model() && force.delta_T != 0.0;
  if (ensemble == "nvt_bao" &&
      ((force.potentials.size() > 1 && force.multiple_potentials_mode() == "average") || ramped_temperature_model))
    input_error("ensemble nvt_bao is not available with averaged potentials or a ramped temperature-dependent NEP.");
  if ((force.potentials.size() > 1 && force.multiple_potentials_mode() == "average") || ramped_temperature_model) {
    // the run follows the MEAN of several potentials (force.cu:533-562), or a temperature-dependent NEP sees a new
    // temperature every step (force.cu:803): every step goes through Force::compute
    if (ensemble == "nvt_nhc" && !nhc_state_) {
      hip_check(hipMalloc((void**)&nhc_state_, sizeof(double) * NEPMI_NHC_STATE_SIZE), "hipMalloc");
      die_on(nepmi_nhc_init(e, N, t_a, temperature_coupling, time_step, nhc_state_), "nhc_init");
    }
    for (int s = 0; s < steps; ++s) {
      const double target = t_a + (t_b - t_a) * (double(s) / steps);
      if (ensemble == "nvt_nhc") { // integrate_nvt_nhc_1, ensemble_nhc.cu:166-197
        find_thermo();
        die_on(nepmi_nhc_half_step(e, N, target, time_step, thermo.data(), nhc_state_, atom.velocity_per_atom.data()), "nhc");
      } else if (ensemble == "nvt_lan") { // Ensemble_LAN::compute1
        die_on(nepmi_lan_half_step(e, N, target, temperature_coupling, atom.mass.data(), atom.velocity_per_atom.data()), "lan");
      }
      die_on(nepmi_vv_step1(e, N, time_step, atom.mass.data(), atom.force_per_atom.data(), atom.position_per_atom.data(),
                            atom.velocity_per_atom.data()),
             "vv_step1");
      force.advance_temperature(); // temperature += delta_T (force.cu:803)
      force.compute(box, atom.position_per_atom, atom.type, atom.potential_per_atom, atom.force_per_atom, atom.virial_per_atom);
      die_on(nepmi_vv_step2(e, N, time_step, atom.mass.data(), atom.force_per_atom.data(), atom.velocity_per_atom.data()),
             "vv_step2");
      if (ensemble == "nvt_lan") // Ensemble_LAN::compute2: the thermostat's second half-step precedes find_thermo
        die_on(nepmi_lan_half_step(e, N, target, temperature_coupling, atom.mass.data(), atom.velocity_per_atom.data()), "lan");
      if (ensemble != "nve" || s + 1 == steps)
        find_thermo();
      if (ensemble == "nvt_ber")
        die_on(nepmi_berendsen_scale(e, N, target, 1.0 / temperature_coupling, thermo.data(), atom.velocity_per_atom.data()), "ber");
      else if (ensemble == "nvt_bdp")
        die_on(nepmi_bdp_scale(e, N, target, temperature_coupling, thermo.data(), atom.velocity_per_atom.data()), "bdp");
      else if (ensemble == "nvt_nhc")
        die_on(nepmi_nhc_half_step(e, N, target, time_step, thermo.data(), nhc_state_, atom.velocity_per_atom.data()), "nhc");
    }
    return;
  }
  double th[8];
  int st;
  double *x = atom.position_per_atom.data(), *v = atom.velocity_per_atom.data(), *pe = atom.potential_per_atom.data(),
         *f = atom.force_per_atom.data(), *w = atom.virial_per_atom.data();
  if (ensemble == "nve")
    st = nepmi_run_nve(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, x, v, pe, f, w, steps, th);
  else if (ensemble == "nvt_ber")
    st = nepmi_run_nvt_ber(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, t_a, t_b,
                           temperature_coupling, x, v, pe, f, w, steps, th);
  else if (ensemble == "nvt_nhc")
    st = nepmi_run_nvt_nhc(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, t_a, t_b,
                           temperature_coupling, x, v, pe, f, w, steps, th);
  else if (ensemble == "nvt_lan")
    st = nepmi_run_nvt_lan(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, t_a, t_b,
                           temperature_coupling, x, v, pe, f, w, steps, th);
  else if (ensemble == "nvt_bao") // the noise amplitude stays that of T1 (ensemble_bao.cu:36), whatever the segment
    st = nepmi_run_nvt_bao(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, temperature1,
                           temperature2, temperature_coupling, x, v, pe, f, w, steps, th);
  else
    st = nepmi_run_nvt_bdp(e, box.cpu_h, pbc, N, atom.type.data(), atom.mass.data(), time_step, steps, t_a, t_b,
                           temperature_coupling, x, v, pe, f, w, steps, th);
  die_on(st, "run");
  thermo.copy_from_host(th, 8);
}

// Run::perform_a_run (run.cu:211-341).  The steps between two outputs are one call of a fused run loop; the loop
// body below is what the reference does at an output step (measure.process) plus correct_velocity.
void Run::perform_a_run()
{
  if (par_.world > 1) {
    perform_a_run_dist();
    return;
  }
  if (!gpu_allocated || force.potentials.empty())
    input_error("No potential is defined before run.");
  const int N = atom.number_of_atoms;
  nepmi_engine* e = force.engine();
  // Dump_XYZ's constructor (dump_xyz.cu:60-63): the first dump that asks for unwrapped positions starts the
  // array from the coordinates as they are (not yet wrapped); from then on every first half-step adds its
  // drift to it (integrate.cu:347-372), in this run and the following ones
  bool want_unwrapped = false;
  for (const auto& d : dump_xyzs)
    want_unwrapped = want_unwrapped || d.has_unwrapped_position;
  if (want_unwrapped && atom.unwrapped_position.size() == 0) {
    atom.unwrapped_position.resize(3 * (size_t)N);
    hip_check(hipMemcpy(atom.unwrapped_position.data(), atom.position_per_atom.data(), sizeof(double) * 3 * (size_t)N,
                        hipMemcpyDeviceToDevice), "D2D");
    die_on(nepmi_engine_set_unwrapped(e, atom.unwrapped_position.data()), "set_unwrapped");
  }
  if (dump_thermo_interval > 0) {
    FILE* fid = std::fopen("thermo.out", "a");
    std::fprintf(fid, "# dump_thermo %d\n# format_version 1\n# num_atoms %d\n# dt_output %.10e fs\n", dump_thermo_interval, N,
                 time_step * dump_thermo_interval * TIME_UNIT_CONVERSION);
    std::fprintf(fid, "# columns T KE PE sxx syy szz syz sxz sxy ax ay az bx by bz cx cy cz\n");
    std::fclose(fid);
  }
  dump_observer_open(); // measure.initialize precedes the first force call (run.cu:215)
  active_open();
  // target temperature of a temperature-dependent NEP (Run::parse_run, run.cu:679-681)
  // (integrate.temperature1/2 keep the values of the last ensemble that had them; an NVE-only input leaves them
  // uninitialised in the reference -- here they start at 300 K)
  force.temperature = temperature1;
  force.delta_T = (temperature2 - temperature1) / number_of_steps;
  // initial force (run.cu:220-241): the same Force::compute overload as in the loop, so it advances the temperature too
  force.advance_temperature();
  force.compute(box, atom.position_per_atom, atom.type, atom.potential_per_atom, atom.force_per_atom, atom.virial_per_atom);
  hip_check(hipDeviceSynchronize(), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  die_on(nepmi_engine_reset_thermostat(e), "reset_thermostat"); // Ensemble_NHC: a fresh chain per run (integrate.cu:85-92)
  if (nhc_state_) {
    (void)hipFree(nhc_state_);
    nhc_state_ = nullptr;
  }
  if (ensemble == "nvt_lan" || ensemble == "nvt_bao") { // Ensemble_LAN / Ensemble_BAO constructors (ensemble_lan.cu:39, ensemble_bao.cu:39): seeded with rand()
    const int seed = host_rand();
    die_on(nepmi_lan_seed(e, seed), "lan_seed");
    std::printf("    Langevin generator seed = %d.\n", seed);
  }
  if (ensemble == "nvt_bdp") { // Ensemble_BDP::initialize_rng (ensemble_bdp.cu:32-39): seeded from the clock
    // GPUMD_MI_DEBUG=1 is the run-time counterpart of the reference's -DDEBUG build: the fixed seed 12345678
    const uint64_t seed = std::getenv("GPUMD_MI_DEBUG") ? 12345678u
                                                        : (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
    die_on(nepmi_bdp_seed(e, seed), "bdp_seed");
    std::printf("    BDP noise seed = %llu.\n", (unsigned long long)((std::mt19937::result_type)seed));
  }
  auto temperature_at = [&](int step) { return temperature1 + (temperature2 - temperature1) * (double(step) / number_of_steps); };
  // the next step (1-based count of finished steps) at which something is written
  auto next_output = [&](int done) {
    int nxt = number_of_steps;
    auto upto = [&](int interval) {
      if (interval > 0) {
        const int k = (done / interval + 1) * interval;
        nxt = k < nxt ? k : nxt;
      }
    };
    upto(dump_thermo_interval);
    upto(dump_restart_interval);
    for (const auto& d : dump_xyzs)
      upto(d.interval);
    if (observer.active) {
      upto(observer.interval_thermo);
      upto(observer.interval_exyz);
    }
    if (active_.active)
      upto(active_.interval);
    if (number_of_steps >= 10)
      upto(number_of_steps / 10); // progress lines
    if (correct_interval_ > 0) { // the correction precedes the steps 0, k, 2k, ...: a segment ends right before them
      const int k = (done / correct_interval_ + 1) * correct_interval_;
      nxt = k < nxt ? k : nxt;
    }
    return nxt;
  };
  int done = 0;
  while (done < number_of_steps) {
    if (correct_interval_ > 0 && done % correct_interval_ == 0)
      correct_velocity_now(); // velocity.correct_velocity(step, ...) at the top of step `done` (run.cu:252)
    const int upto = next_output(done);
    run_segment(upto - done, temperature_at(done), temperature_at(upto));
    global_time += time_step * (upto - done);
    done = upto;
    const int step = done - 1; // the reference's 0-based step index of the step just finished
    // measure.process
    dump_thermo(step);
    for (auto& d : dump_xyzs)
      dump_xyz(d, step);
    dump_restart(step);
    dump_observer_process(step);
    active_process(step);
    if (number_of_steps >= 10 && (step + 1) % (number_of_steps / 10) == 0)
      std::printf("    %d steps completed.\n", step + 1);
  }
  hip_check(hipDeviceSynchronize(), "sync");
  dump_observer_close();
  active_close();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("Time used for this run = %g second.\n", sec);
  std::printf("Speed of this run = %g atom*step/second.\n", (double)N * number_of_steps / sec); // run.cu:325-326
  if (auto* p = dynamic_cast<NEP_MI*>(force.potentials[0].get())) {
    p->write_neighbor_out();
    // which kernel forms the engine's counted rules chose, and how often the lists were rebuilt (not in the reference's log)
    char forms[512];
    nepmi_stats st;
    if (nepmi_engine_describe(p->engine(), forms, (int)sizeof(forms)) >= 0 && nepmi_engine_stats(p->engine(), 0, &st) >= 0)
      std::printf("    (libnepmi: %s; list rebuilds so far: %lld)\n", forms, (long long)st.num_rebuild);
  }
  for (auto& d : dump_xyzs)
    if (d.fid) {
      std::fclose(d.fid);
      d.fid = nullptr;
    }
}

// ---- one process per GPU ---------------------------------------------------------------------------------------

// `potential <file> [x|y|z]` with more than one rank: the transport, the process grid and the decomposed driver;
// every rank has read the whole model.xyz and contributes a slice of the atoms (the driver migrates them to their
// owners).  With a direction the grid is one-dimensional along it (the reference's slab partition, force.cu:122-160),
// without one it is as cubic as the number of ranks allows.
void Run::setup_dist(const std::vector<std::string>& p)
{
  if (force.potentials.size() > 0 || dist_)
    input_error("Several potentials are not available in multi-GPU runs.");
  std::ifstream in(p[1]);
  std::string name;
  if (!(in >> name))
    input_error("Failed to open " + p[1] + ".");
  if (name.rfind("nep", 0) != 0)
    input_error("multi-GPU runs carry NEP potentials only.");
  dist_model_ = nepmi_model_load(p[1].c_str());
  if (!dist_model_)
    input_error(nepmi_last_error());
  const int P = par_.world;
  int grid[3] = {1, 1, 1};
  if (p.size() == 3) {
    grid[p[2] == "x" ? 0 : (p[2] == "y" ? 1 : 2)] = P;
  } else { // as cubic as possible: every peer of a 2x2x2 grid is a direct xGMI link on an 8-GPU node
    int best[3] = {P, 1, 1};
    for (int a = 1; a <= P; ++a)
      for (int b = 1; a * b <= P; ++b)
        if (P % (a * b) == 0) {
          int g[3] = {a, b, P / (a * b)};
          std::sort(g, g + 3);
          if (g[2] - g[0] < best[0] - best[2]) {
            best[0] = g[2];
            best[1] = g[1];
            best[2] = g[0];
          }
        }
    for (int d = 0; d < 3; ++d)
      grid[d] = best[d];
  }
  // rendezvous over TCP; RCCL (one GPU per rank) gets its id through it
  die_on(nepmi_transport_tcp(par_.master_addr.c_str(), par_.master_port, par_.rank, P, &boot_), "transport (tcp)");
  have_boot_ = true;
  const nepmi_transport* tr = &boot_;
  if (par_.use_rccl) {
    int id[NEPMI_RCCL_ID_BYTES / 4];
    std::memset(id, 0, sizeof id);
    if (par_.rank == 0)
      die_on(nepmi_transport_rccl_id((char*)id), "rccl id");
    boot_.allreduce(boot_.ctx, id, NEPMI_RCCL_ID_BYTES / 4, 1, 0, nullptr); // the other ranks contribute zeros
    die_on(nepmi_transport_rccl((const char*)id, par_.rank, P, &rccl_), "transport (rccl)");
    have_rccl_ = true;
    tr = &rccl_;
  }
  const int pbc[3] = {box.pbc_x, box.pbc_y, box.pbc_z};
  dist_ = nepmi_dist_create(dist_model_, tr, box.cpu_h, pbc, grid, nullptr);
  if (!dist_)
    input_error(nepmi_last_error());
  std::printf("Use %d GPUs: process grid %d x %d x %d, ghost exchange over %s.\n", P, grid[0], grid[1], grid[2],
              par_.use_rccl ? "RCCL" : "TCP sockets");
  nepmi_dist_info info;
  nepmi_dist_get_info(dist_, &info);
  std::printf(info.reverse_ghosts ? "    ghost shell rc + skin, the ghosts' partial forces return to their owners (two exchanges per step).\n"
                                  : "    ghost shell 2 (rc + skin), descriptors of the inner ring recomputed (one exchange per step).\n");
}

// this rank's share of the atoms goes to the driver at the first run (the velocity keyword may follow potential)
void Run::dist_setup_atoms()
{
  // this rank's slice of the atoms (any split will do: the driver migrates)
  const int P = par_.world;
  const int N = atom.number_of_atoms;
  std::vector<int> ty;
  std::vector<double> ms, xs, vs;
  std::vector<int64_t> ids;
  for (int i = par_.rank; i < N; i += P)
    ids.push_back(i);
  const size_t n = ids.size();
  ty.resize(n);
  ms.resize(n);
  xs.resize(3 * n);
  vs.resize(3 * n);
  for (size_t q = 0; q < n; ++q) {
    const int i = (int)ids[q];
    ty[q] = atom.cpu_type[i];
    ms[q] = atom.cpu_mass[i];
    for (int d = 0; d < 3; ++d) {
      xs[q + d * n] = atom.cpu_position_per_atom[i + (size_t)d * N];
      vs[q + d * n] = atom.cpu_velocity_per_atom[i + (size_t)d * N];
    }
  }
  GPU_Vector<int> d_t(n ? n : 1);
  GPU_Vector<double> d_m(n ? n : 1), d_x(3 * n ? 3 * n : 1), d_v(3 * n ? 3 * n : 1);
  GPU_Vector<int64_t> d_i(n ? n : 1);
  if (n) {
    d_t.copy_from_host(ty.data());
    d_m.copy_from_host(ms.data());
    d_x.copy_from_host(xs.data());
    d_v.copy_from_host(vs.data());
    d_i.copy_from_host(ids.data());
  }
  die_on(nepmi_dist_setup(dist_, (int64_t)n, d_t.data(), d_m.data(), d_x.data(), d_v.data(), d_i.data()), "dist setup");
  dist_ready_ = true;
}

void Run::perform_a_run_dist()
{
  if (!dist_)
    input_error("No potential is defined before run.");
  if (!dist_ready_)
    dist_setup_atoms();
  if (observer.active)
    input_error("dump_observer is not available in multi-GPU runs.");
  if (active_.active)
    input_error("active is not available in multi-GPU runs.");
  if (correct_interval_ > 0)
    input_error("correct_velocity is not available in multi-GPU runs.");
  for (const auto& d : dump_xyzs)
    if (d.has_unwrapped_position)
      input_error("unwrapped positions are not tracked in multi-GPU runs.");
  const int N = atom.number_of_atoms;
  const bool root = par_.rank == 0;
  const int ens = ensemble == "nve" ? 0 : ensemble == "nvt_ber" ? 1 : ensemble == "nvt_nhc" ? 2 : ensemble == "nvt_bdp" ? 3
                  : ensemble == "nvt_lan" ? 4 : 5;
  if (root && dump_thermo_interval > 0) {
    FILE* fid = std::fopen("thermo.out", "a");
    std::fprintf(fid, "# dump_thermo %d\n# format_version 1\n# num_atoms %d\n# dt_output %.10e fs\n", dump_thermo_interval, N,
                 time_step * dump_thermo_interval * TIME_UNIT_CONVERSION);
    std::fprintf(fid, "# columns T KE PE sxx syy szz syz sxz sxy ax ay az bx by bz cx cy cz\n");
    std::fclose(fid);
  }
  // temperature-dependent NEP: Force::temperature = T1 advanced once by the initial force call (run.cu:679-681, :232-241);
  // no effect on plain models
  die_on(nepmi_engine_set_temperature(nepmi_dist_engine(dist_),
                                      temperature1 + (temperature2 - temperature1) / number_of_steps), "set_temperature");
  die_on(nepmi_dist_compute(dist_), "dist compute"); // the initial force
  hip_check(hipDeviceSynchronize(), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  die_on(nepmi_dist_reset_thermostat(dist_), "reset_thermostat");
  if (ens == 3) {
    // every rank must draw the same noise: the seed comes from rank 0's clock
    int64_t seed = !root ? 0
                         : (std::getenv("GPUMD_MI_DEBUG") ? (int64_t)12345678
                                                          : (int64_t)std::chrono::system_clock::now().time_since_epoch().count());
    boot_.allreduce(boot_.ctx, &seed, 1, 2, 0, nullptr);
    die_on(nepmi_dist_bdp_seed(dist_, (uint64_t)seed), "bdp_seed");
    std::printf("    BDP noise seed = %llu.\n", (unsigned long long)((std::mt19937::result_type)seed));
  }
  if (ens == 4 || ens == 5) {
    // Ensemble_LAN's / Ensemble_BAO's constructor seeds the per-atom generators with rand() (ensemble_lan.cu:39): rank 0 draws, every rank uses it
    int64_t seed = root ? (int64_t)host_rand() : 0;
    boot_.allreduce(boot_.ctx, &seed, 1, 2, 0, nullptr);
    die_on(nepmi_dist_lan_seed(dist_, (int)seed), "lan_seed");
    std::printf("    Langevin generator seed = %d.\n", (int)seed);
  }
  auto temperature_at = [&](int step) { return temperature1 + (temperature2 - temperature1) * (double(step) / number_of_steps); };
  auto next_output = [&](int done) {
    int nxt = number_of_steps;
    auto upto = [&](int interval) {
      if (interval > 0) {
        const int k = (done / interval + 1) * interval;
        nxt = k < nxt ? k : nxt;
      }
    };
    upto(dump_thermo_interval);
    upto(dump_restart_interval);
    for (const auto& d : dump_xyzs)
      upto(d.interval);
    if (number_of_steps >= 10)
      upto(number_of_steps / 10);
    return nxt;
  };
  int done = 0;
  while (done < number_of_steps) {
    const int upto = next_output(done);
    double th[8];
    die_on(nepmi_dist_run(dist_, ens, time_step, upto - done, temperature_at(done), temperature_at(upto), temperature_coupling,
                          upto - done, th),
           "dist run");
    global_time += time_step * (upto - done);
    done = upto;
    const int step = done - 1;
    bool need_atoms = dump_restart_interval > 0 && (step + 1) % dump_restart_interval == 0;
    for (const auto& d : dump_xyzs)
      need_atoms = need_atoms || (step + 1) % d.interval == 0;
    if (need_atoms) // every atom to rank 0, in file order: the dumps below are the single-GPU code
      die_on(nepmi_dist_gather_global(dist_, 0, root ? atom.position_per_atom.data() : nullptr,
                                      root ? atom.velocity_per_atom.data() : nullptr, root ? atom.force_per_atom.data() : nullptr,
                                      root ? atom.potential_per_atom.data() : nullptr, root ? atom.virial_per_atom.data() : nullptr),
             "gather");
    if (root) {
      thermo.copy_from_host(th, 8);
      dump_thermo(step);
      for (auto& d : dump_xyzs)
        dump_xyz(d, step);
      dump_restart(step);
      if (number_of_steps >= 10 && (step + 1) % (number_of_steps / 10) == 0)
        std::printf("    %d steps completed.\n", step + 1);
    }
  }
  hip_check(hipDeviceSynchronize(), "sync");
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  nepmi_dist_info info;
  nepmi_dist_get_info(dist_, &info);
  std::printf("Time used for this run = %g second.\n", sec);
  std::printf("Speed of this run = %g atom*step/second.\n", (double)N * number_of_steps / sec);
  std::printf("    (%d GPUs; this rank: %lld owned + %lld ghost atoms, %lld decompositions)\n", par_.world,
              (long long)info.n_owned, (long long)(info.n_local - info.n_owned), (long long)info.num_decompositions);
  for (auto& d : dump_xyzs)
    if (d.fid) {
      std::fclose(d.fid);
      d.fid = nullptr;
    }
}

} // namespace gmi
