// gpumd-mi: stand-alone C++ host of the MI355X NEP force engine.  Like the reference's `gpumd`
// (src/main_gpumd/main.cu:29-66) it takes no input arguments and expects run.in + model.xyz in
// the working directory.  `--check-input` parses the inputs without touching a GPU.
//
// Several GPUs: one process per GPU, launched torchrun- or mpirun-style (RANK / WORLD_SIZE / LOCAL_RANK /
// MASTER_ADDR / MASTER_PORT, or OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK).  Every rank reads the inputs, rank 0
// writes the outputs; the step is domain-decomposed by libnepmi (nepmi_dist_*).  NEPMI_TRANSPORT=rccl|tcp
// chooses the ghost exchange (default: RCCL when every rank has a GPU of its own, else TCP through host memory).
#include "run.h"

#include <chrono>
#include <cstdlib>
#include <cstring>

static int env_int(const char* a, const char* b, int def)
{
  const char* v = std::getenv(a);
  if (!v && b)
    v = std::getenv(b);
  return v ? std::atoi(v) : def;
}

int main(int argc, char* argv[])
{
  bool check_only = false;
  for (int k = 1; k < argc; ++k)
    if (std::strcmp(argv[k], "--check-input") == 0)
      check_only = true;
  if (argc == 4 && std::strcmp(argv[1], "--rand-check") == 0) { // the velocity stream's generator, for the tests
    const int n = std::atoi(argv[2]);
    const unsigned seed = (unsigned)std::strtoul(argv[3], nullptr, 10);
    for (int k = 0; k < n; ++k)
      std::printf("%d\n", gmi::host_rand_for_tests(seed, k == 0));
    return 0;
  }
  gmi::Parallel par;
  par.rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", 0);
  par.world = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
  par.local_rank = env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", par.rank);
  if (const char* a = std::getenv("MASTER_ADDR"))
    par.master_addr = a;
  par.master_port = env_int("MASTER_PORT", nullptr, 29400);
  if (par.world < 1 || par.rank < 0 || par.rank >= par.world) {
    std::printf("Error: RANK / WORLD_SIZE are inconsistent.\n");
    return 1;
  }
  if (check_only)
    par.world = 1, par.rank = 0;
  if (par.rank != 0) // rank 0 speaks for the run
    if (!std::freopen("/dev/null", "w", stdout))
      return 1;
  std::printf("***************************************************************\n");
  std::printf("*   gpumd-mi: MI355X-native NEP force engine + MD stepper     *\n");
  std::printf("*   (reads GPUMD's run.in / model.xyz / nep.txt)              *\n");
  std::printf("***************************************************************\n");
  if (!check_only) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
      std::printf("Error: no HIP device; gpumd-mi needs an MI355X (use --check-input to parse only).\n");
      return 1;
    }
    const int dev = par.local_rank % n;
    gmi::hip_check(hipSetDevice(dev), "hipSetDevice");
    hipDeviceProp_t prop;
    gmi::hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    std::printf("GPU %d: %s (%s), %d CUs, %.1f GB\n", dev, prop.name, prop.gcnArchName, prop.multiProcessorCount,
                prop.totalGlobalMem / 1073741824.0);
    const char* tr = std::getenv("NEPMI_TRANSPORT");
    par.use_rccl = par.world > 1 && (tr ? std::strcmp(tr, "rccl") == 0 : n >= par.world);
  }
  const auto t0 = std::chrono::steady_clock::now();
  {
    gmi::Run run(check_only, par);
    run.execute_run_in();
  }
  std::printf("Time used = %g s.\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
