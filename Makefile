# Builds the MI355X engine: edlib_amd/libedlib.so (the drop-in shared library,
# exporting the five edlib.h symbols plus include/edlib_amd.h) for gfx950.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := edlib_amd/csrc
OBJDIR  := build/obj
# -disable-promote-alloca-to-vector: keep the per-lane word arrays (Pv[], Mv[], Peq rows) as separate
# VGPRs; as <N x i32> tuples every partial update (banded kernel) costs a whole-tuple copy.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Wall -Wno-unused-function \
            -mllvm -disable-promote-alloca-to-vector

SRCS := $(CSRC)/reads_kernels.hip $(CSRC)/reads_kernels_long.hip $(CSRC)/pair_kernels.hip $(CSRC)/wide_kernels.hip $(CSRC)/ring32_kernels.hip $(CSRC)/lanepair_kernels.hip $(CSRC)/lanepair_kernels42.hip $(CSRC)/lanepair_kernels24.hip $(CSRC)/flat_results.hip $(CSRC)/runtime.hip $(CSRC)/engine.hip $(CSRC)/engine_reads.hip $(CSRC)/engine_pairs.hip $(CSRC)/engine_paths.hip $(CSRC)/engine_flat.hip $(CSRC)/long_reads.hip $(CSRC)/one_pair.hip $(CSRC)/api.hip
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))

all: edlib_amd/libedlib.so build/edlib-aligner-batch build/latency build/cu_hog build/libcu_hog.so

# the lane-per-pair scans (a ladder of unrolled loops per window height: minutes each) depend on their own headers only
$(OBJDIR)/lanepair_kernels.o $(OBJDIR)/lanepair_kernels42.o $(OBJDIR)/lanepair_kernels24.o: $(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/lanepair.hpp $(CSRC)/lanepair_core.hpp $(CSRC)/lanepair_asm.hpp $(CSRC)/lanepair_kernels.hpp
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/edlib.h include/edlib_amd.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

edlib_amd/libedlib.so: $(OBJS) $(CSRC)/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/exports.map -o $@ $(OBJS)

# batch-aware CLI with the reference CLI's flags and output
build/edlib-aligner-batch: apps/aligner_batch.cpp edlib_amd/libedlib.so include/edlib.h include/edlib_amd.h
	@mkdir -p build
	g++ -O2 -std=c++14 -Iinclude apps/aligner_batch.cpp -Ledlib_amd -l:libedlib.so -Wl,-rpath,'$$ORIGIN/../edlib_amd' -o $@

# microseconds per edlibAlign() call for any library with the edlib C ABI
build/latency: tools/latency.cpp
	@mkdir -p build
	g++ -O2 -std=c++14 -pthread tools/latency.cpp -ldl -o $@

# a second process that holds most wave slots of the device for a few seconds (tests/test_gpu_wide.py)
build/cu_hog: tools/cu_hog.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O2 tools/cu_hog.hip -o $@
build/libcu_hog.so: tools/cu_hog.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O2 -DCU_HOG_LIBRARY -shared -fPIC tools/cu_hog.hip -o $@

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf build edlib_amd/libedlib.so
.PHONY: all oracle clean
