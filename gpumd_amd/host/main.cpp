// gpumd-mi: stand-alone C++ host of the MI355X NEP force engine.  Like the reference's `gpumd`
// (src/main_gpumd/main.cu:29-66) it takes no input arguments and expects run.in + model.xyz in
// the working directory.  `--check-input` parses the inputs without touching a GPU.
#include "run.h"

#include <chrono>
#include <cstring>

int main(int argc, char* argv[])
{
  bool check_only = false;
  for (int k = 1; k < argc; ++k)
    if (std::strcmp(argv[k], "--check-input") == 0)
      check_only = true;
  std::printf("***************************************************************\n");
  std::printf("*   gpumd-mi: MI355X-native NEP force engine + NVE stepper    *\n");
  std::printf("*   (reads GPUMD's run.in / model.xyz / nep.txt)              *\n");
  std::printf("***************************************************************\n");
  if (!check_only) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
      std::printf("Error: no HIP device; gpumd-mi needs an MI355X (use --check-input to parse only).\n");
      return 1;
    }
    hipDeviceProp_t prop;
    gmi::hip_check(hipGetDeviceProperties(&prop, 0), "hipGetDeviceProperties");
    std::printf("GPU 0: %s (%s), %d CUs, %.1f GB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                prop.totalGlobalMem / 1073741824.0);
  }
  const auto t0 = std::chrono::steady_clock::now();
  gmi::Run run(check_only);
  run.execute_run_in();
  std::printf("Time used = %g s.\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
