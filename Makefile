# Builds the MI355X linearization library and (optionally) the test-only oracle.
#
#   make            -> moptimizer_0_amd/lib/libmoptimizer_hip.so   (gfx950 code object + C ABI)
#   make oracle     -> oracle/_build/{liboracle.so,replay_reference_tests}   (CPU checker)
#   make cpptests   -> tests/cpp/_build/*                           (C++ drop-in programs)
#
# hipcc cross-compiles for gfx950 without a GPU present.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
ROCM       ?= /opt/rocm
HIPFLAGS   ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
INCLUDES    = -Iinclude -Imoptimizer_0_amd/csrc
CSRC        = moptimizer_0_amd/csrc
OBJDIR      = build/obj
LIBDIR      = moptimizer_0_amd/lib
LIB         = $(LIBDIR)/libmoptimizer_hip.so

PUBLIC_HEADERS = include/moptimizer_hip.h include/moptimizer_amd/so3.hpp $(CSRC)/sweep.hpp $(CSRC)/fd_device.hpp \
                 $(CSRC)/sweep_device.hpp $(CSRC)/jit_model.hpp $(CSRC)/cost_state.hpp $(CSRC)/lm_device.hpp $(CSRC)/aql.hpp

all: $(LIB)

$(OBJDIR) $(LIBDIR):
	mkdir -p $@

# leading scalar kernel arguments preloaded into SGPRs at wave launch (gfx940+)
$(OBJDIR)/sweep_kernels.o: $(CSRC)/sweep_kernels.hip $(PUBLIC_HEADERS) | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(INCLUDES) -mllvm -amdgpu-kernarg-preload-count=4 -c $< -o $@

# forward-difference sweeps: without the SLP vectorizer (fd_kernels.hip says why)
$(OBJDIR)/fd_kernels.o: $(CSRC)/fd_kernels.hip $(PUBLIC_HEADERS) | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(INCLUDES) -mllvm -amdgpu-kernarg-preload-count=4 -fno-slp-vectorize -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(PUBLIC_HEADERS) | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(INCLUDES) -x hip -c $< -o $@

$(OBJDIR)/icp_grid.o: $(CSRC)/icp_grid.hip $(PUBLIC_HEADERS) | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(INCLUDES) -c $< -o $@

$(OBJDIR)/lm_kernels.o: $(CSRC)/lm_kernels.hip $(PUBLIC_HEADERS) | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(INCLUDES) -c $< -o $@

OBJS = $(OBJDIR)/sweep_kernels.o $(OBJDIR)/fd_kernels.o $(OBJDIR)/icp_grid.o $(OBJDIR)/c_abi.o $(OBJDIR)/icp.o \
       $(OBJDIR)/group.o $(OBJDIR)/aql.o $(OBJDIR)/jit_model.o $(OBJDIR)/device_pool.o $(OBJDIR)/combine.o $(OBJDIR)/lm.o $(OBJDIR)/lm_kernels.o

$(LIB): $(OBJS) | $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -L$(ROCM)/lib -lrccl -lhiprtc -lhsa-runtime64 -lpthread -lrt \
	    -Wl,-rpath,$(ROCM)/lib -Wl,--no-undefined

oracle:
	$(MAKE) -C oracle all

cpptests: $(LIB)
	$(MAKE) -C tests/cpp all

clean:
	rm -rf build $(LIBDIR) tests/cpp/_build
	$(MAKE) -C oracle clean

.PHONY: all oracle cpptests clean
