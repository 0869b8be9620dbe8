// Run: the run.in interpreter and the MD loop (src/main_gpumd/run.{cu,cuh}), restricted to the
// keywords of the hot path: potential, replicate, velocity, ensemble nve, time_step, dump_thermo,
// dump_xyz (all options but the NEP-charge quantities), dump_restart, run.
#pragma once
#include "force.h"

namespace gmi {

struct DumpXyz {
  int interval = 0;
  std::string filename;
  int precision = 1; // 1 single (%.9g, the default: dump_xyz.cuh:67), 2 double (%.17g)   -- dump_xyz.cu:163-165
  bool has_mass = false, has_charge = false, has_velocity = false, has_force = false, has_potential = false,
       has_unwrapped_position = false, has_virial = false, has_group_labels = false; // parse_utilities.cu:97-147
  bool separated = false;   // file name ended in '*': one file per frame, <name><step>  (dump_xyz.cu:104-110)
  int grouping_method = -1; // `group <method> <id>`: only that group's atoms are written (dump_xyz.cu:368-372)
  int group_id = 0;
  FILE* fid = nullptr;
};

// Dump_Observer (src/measure/dump_observer.cuh): `dump_observer <mode> <thermo> <exyz> <has_vel> <has_force>`
struct DumpObserver {
  bool active = false;
  std::string mode = "observe"; // observe: every potential re-evaluated at the output steps; average: the run's own
  int interval_thermo = 1, interval_exyz = 1, has_velocity = 0, has_force = 0;
  std::vector<FILE*> exyz_files, thermo_files; // observer<i>.xyz / observer<i>.out (no index when there is one)
};

// Active (src/measure/active.cuh): `active <interval> <has_velocity> <has_force> <has_uncertainty> <threshold>` -- committee
// uncertainty over the `potential` lines of run.in, the run itself follows the first one
struct ActiveLearning {
  bool active = false;
  int interval = 1, has_velocity = 0, has_force = 0, has_uncertainty = 0;
  double threshold = 0.0;
  FILE* exyz_file = nullptr; // active.xyz
  FILE* out_file = nullptr;  // active.out
};

// One process per GPU (torchrun / mpirun style launch: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT or
// OMPI_COMM_WORLD_*): the run is domain-decomposed by libnepmi's nepmi_dist_* driver (RCCL over xGMI when every
// rank has its own GPU, TCP sockets otherwise); rank 0 writes the output files.
struct Parallel {
  int rank = 0, world = 1, local_rank = 0;
  bool use_rccl = false;
  std::string master_addr = "127.0.0.1";
  int master_port = 29400;
};

class Run
{
public:
  Run(bool check_only, const Parallel& par);
  ~Run();
  void execute_run_in();

private:
  void parse_one_keyword(const std::vector<std::string>& tokens);
  void perform_a_run();
  void perform_a_run_dist();
  void run_segment(int steps, double t_a, double t_b);
  void correct_velocity_now();
  void setup_dist(const std::vector<std::string>& potential_line);
  void dist_setup_atoms();
  void find_thermo();
  void dump_thermo(int step);
  void dump_xyz(DumpXyz& d, int step);
  void dump_restart(int step);
  void dump_observer_open();
  void dump_observer_process(int step);
  void dump_observer_write(int step, int file_index);
  void dump_observer_close();
  void active_open();
  void active_process(int step);
  void active_close();

  bool check_only_;
  Parallel par_;
  nepmi_transport boot_{};      // TCP: the rendezvous, and the transport itself when RCCL is not used
  nepmi_transport rccl_{};
  bool have_boot_ = false, have_rccl_ = false;
  nepmi_model* dist_model_ = nullptr;
  nepmi_dist* dist_ = nullptr;
  bool dist_ready_ = false;
  double* nhc_state_ = nullptr; // stepwise path (several potentials averaged): the chain of this run
  int correct_interval_ = 0, correct_group_method_ = -1; // correct_velocity (run.cu:610-647)
  Box box;
  Atom atom;
  std::vector<Group> groups;
  Force force;
  bool has_velocity_in_xyz = false;
  bool gpu_allocated = false;
  std::string ensemble = "nve";
  double temperature1 = 300.0, temperature2 = 300.0, temperature_coupling = 100.0; // nvt_ber, nvt_nhc
  double time_step = 1.0 / TIME_UNIT_CONVERSION;
  double global_time = 0.0;
  int number_of_steps = 0;
  double initial_temperature = 300.0;
  int dump_thermo_interval = 0;
  int dump_restart_interval = 0;
  std::vector<DumpXyz> dump_xyzs;
  DumpObserver observer;
  ActiveLearning active_;
  GPU_Vector<double> thermo; // 8 doubles
  std::vector<std::string> elements;
  std::string potential_file;
};

} // namespace gmi
