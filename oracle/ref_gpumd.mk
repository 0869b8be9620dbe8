# TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
#
#   make -f ref_gpumd.mk -j8      -> _ref/gpumd_ref : the REFERENCE's own `gpumd` executable, compiled for gfx950 from
#                                    the sources where they lie under $(REFERENCE)/src (nothing is copied; objects go to
#                                    the git-ignored _ref/gpumd_obj).  It is the comparator of profiles/ref_*: the same
#                                    run.in / model.xyz / nep.txt through the reference's HIP build (-DUSE_HIP, its
#                                    gpu_macro.cuh) and through gpumd-mi on the same MI355X -- a same-box throughput baseline
#                                    and an MD-level parity source (thermo.out).  Only possible where /root/reference
#                                    exists (this container); the binary travels to the GPU box like our own .so files.
#
# Flags follow the reference's src/makefile.hip (-O3 --offload-arch -DUSE_HIP; hipblas, hipsolver, hipfft); this is our
# own few lines, the reference's build system is not run and nothing is written under $(REFERENCE).
REFERENCE ?= /root/reference
SRC = $(REFERENCE)/src
OBJDIR = _ref/gpumd_obj
DIRS = main_gpumd minimize phonon integrate mc force measure model utilities
SOURCES = $(foreach d,$(DIRS),$(wildcard $(SRC)/$(d)/*.cu))
OBJECTS = $(patsubst $(SRC)/%.cu,$(OBJDIR)/%.o,$(SOURCES))
HIPCC ?= /opt/rocm/bin/hipcc
# -DDEBUG only fixes the PRNG seed (src/utilities/main_common.cu:30-35), as in the build that produced the reference's
# own regression goldens (tests/gpumd/*): velocities are then the same from run to run and gpumd-mi reproduces them.
CFLAGS = -O3 --offload-arch=gfx950 -DUSE_HIP -DDEBUG -w -I$(SRC)

_ref/gpumd_ref: $(OBJECTS)
	$(HIPCC) --offload-arch=gfx950 $^ -o $@ -L/opt/rocm/lib -lhipblas -lhipsolver -lhipfft

$(OBJDIR)/%.o: $(SRC)/%.cu
	@mkdir -p $(dir $@)
	$(HIPCC) $(CFLAGS) -x hip -c $< -o $@

clean:
	rm -rf $(OBJDIR) _ref/gpumd_ref

.PHONY: clean
