# Top-level build: libptgpu.so (HIP, gfx950), libpthost.so + pt (C++ host), oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
PKG = path-tracer_amd
LIB = $(PKG)/lib
# -ffp-contract=off: hit records must be bit-identical to the CPU restatement; fma only where written.
HIPFLAGS = -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC -pthread -Wall -Wno-unused-function
CSRC = $(PKG)/csrc
GPU_DEPS = $(wildcard $(CSRC)/*.h $(CSRC)/*.inc) include/ptgpu.h

all: $(LIB)/libptgpu.so $(LIB)/libpthost.so $(LIB)/pt $(LIB)/calib_fetch $(LIB)/calib_atomics $(LIB)/rgb2spec_opt oracle

$(LIB)/libptgpu.so: $(CSRC)/ptgpu.hip $(CSRC)/bvh_build.cpp $(CSRC)/sdtree_host.cpp $(GPU_DEPS)
	mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/ptgpu.hip $(CSRC)/bvh_build.cpp $(CSRC)/sdtree_host.cpp

$(LIB)/libpthost.so: $(PKG)/host/pthost.cpp $(PKG)/host/rgb2spec_fit.h include/pthost.h include/ptgpu.h $(CSRC)/cie_data.inc
	mkdir -p $(LIB)
	$(CXX) -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -shared -o $@ $(PKG)/host/pthost.cpp -ldl -lz

$(LIB)/pt: $(PKG)/host/pt_main.cpp $(LIB)/libpthost.so $(LIB)/libptgpu.so
	$(CXX) -O2 -std=c++17 -Wall -o $@ $(PKG)/host/pt_main.cpp -L$(LIB) -lpthost -lptgpu -Wl,-rpath,'$$ORIGIN'

# Regenerates the RGB-to-spectrum coefficient table the reference embeds (srgb.coeffs, missing from its tree) and the excerpt
# shipped in $(PKG)/data (tools/rgb2spec_opt.cpp; build container only): `make rgb2spec-table`
$(LIB)/rgb2spec_opt: tools/rgb2spec_opt.cpp $(PKG)/host/rgb2spec_fit.h $(CSRC)/cie_data.inc
	mkdir -p $(LIB)
	$(CXX) -O2 -std=c++17 -pthread -o $@ tools/rgb2spec_opt.cpp
rgb2spec-table: $(LIB)/rgb2spec_opt
	$(LIB)/rgb2spec_opt --full $(LIB)/srgb.coeffs --slab $(PKG)/data/srgb_z05.coeffs

# FETCH_SIZE calibration micro-benchmark (tools/gpu_calib_fetch.sh); a measuring tool, not part of the product
$(LIB)/calib_atomics: tools/calib_atomics.hip
	mkdir -p $(LIB)
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -o $@ tools/calib_atomics.hip
$(LIB)/calib_fetch: tools/calib_fetch.hip
	mkdir -p $(LIB)
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -o $@ tools/calib_fetch.hip

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) oracle/_build

.PHONY: all oracle clean rgb2spec-table
