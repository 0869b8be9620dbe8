// TEST INFRASTRUCTURE ONLY: stands in for src/utilities/gpu_macro.cuh (CUDA/HIP runtime macro table) when reference
// HOST code is compiled in place by oracle/Makefile's _ref target; the host code compiled there uses none of it.
#pragma once
