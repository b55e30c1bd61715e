# Builds the three native artefacts:
#   bulletproofs_amd/lib/libbpgpu.so   the product: HIP kernels + C ABI (include/bpgpu.h), gfx950
#   bulletproofs_amd/lib/libbpgpu.a    the same object as a static archive (for the Rust build.rs of INTEGRATION.md)
#   bulletproofs_amd/lib/libarkbp.so   host side above the ABI: C++ mirror of util::msm, Bp, Bp2aryStep, FiatShamirRng
#   bulletproofs_amd/lib/libbpchk.so   test-only host build of the device arithmetic headers
#   oracle/_build/libpasta_oracle.so   test-only CPU restatement of the reference algorithms (Oracle B)
#   bulletproofs_amd/bin/{msm_bench,test_ipa}   the reference's command-line drivers for this path, over the library
HIPCC ?= hipcc
CXX ?= g++
CC ?= gcc
CSRC := bulletproofs_amd/csrc
LIBDIR := bulletproofs_amd/lib

BINDIR := bulletproofs_amd/bin
ROCM_PATH ?= /opt/rocm
all: $(LIBDIR)/libbpgpu.so $(LIBDIR)/libbpgpu.a $(BINDIR)/msm_bench_static $(LIBDIR)/libarkbp.so $(LIBDIR)/libbpchk.so oracle/_build/libpasta_oracle.so $(BINDIR)/msm_bench $(BINDIR)/test_ipa

# one device+host compile; the shared library (tests, bench, C++ host side) and the static archive (what the
# reference's build.rs links: north-star "hipcc-built static lib") are two packagings of the same object
$(LIBDIR)/bpgpu.o: $(CSRC)/bpgpu.hip $(CSRC)/kernels.cuh $(CSRC)/curve.cuh $(CSRC)/field.cuh $(CSRC)/host_field.h $(CSRC)/pasta_consts.h include/bpgpu.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -c -fPIC $(HIPFLAGS) $< -o $@

$(LIBDIR)/libbpgpu.so: $(LIBDIR)/bpgpu.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $< -o $@

$(LIBDIR)/libbpgpu.a: $(LIBDIR)/bpgpu.o
	rm -f $@ && ar rcs $@ $<

# link check of the archive the way a Rust build.rs would (static bpgpu + dylib amdhip64 + C++ runtime)
$(BINDIR)/msm_bench_static: examples/msm_bench.cpp bulletproofs_amd/host/ark_bp.hpp $(CSRC)/host_field.h $(CSRC)/pasta_consts.h include/bpgpu.h $(LIBDIR)/libbpgpu.a
	@mkdir -p $(BINDIR)
	$(CXX) -O2 -std=c++17 $< $(LIBDIR)/libbpgpu.a -o $@ -L$(ROCM_PATH)/lib -lamdhip64 -lpthread -Wl,-rpath,$(ROCM_PATH)/lib

# host side above the C ABI (C++ mirror of the reference's Rust interface for this path)
$(LIBDIR)/libarkbp.so: bulletproofs_amd/host/ark_bp_capi.cpp bulletproofs_amd/host/ark_bp.hpp $(CSRC)/host_field.h $(CSRC)/pasta_consts.h include/bpgpu.h $(LIBDIR)/libbpgpu.so
	$(CXX) -O2 -std=c++17 -shared -fPIC $< -o $@ -L$(LIBDIR) -lbpgpu -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/libbpchk.so: $(CSRC)/host_check.cpp $(CSRC)/curve.cuh $(CSRC)/field.cuh $(CSRC)/pasta_consts.h
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -DBP_CHECKED -shared -fPIC -x c++ $< -o $@

# x86-64-v3 (AVX2 + BMI2: every host an MI355X sits in), not -march=native: the library is built here and runs on the GPU box
oracle/_build/libpasta_oracle.so: oracle/pasta_oracle.c
	@mkdir -p oracle/_build
	$(CC) -O3 -march=x86-64-v3 -std=c11 -shared -fPIC -pthread $< -o $@

# the reference's two command-line drivers for this path (examples/msm_bench.rs; test_ipa of src/lib.rs:110-135)
$(BINDIR)/%: examples/%.cpp bulletproofs_amd/host/ark_bp.hpp $(CSRC)/host_field.h $(CSRC)/pasta_consts.h include/bpgpu.h $(LIBDIR)/libbpgpu.so
	@mkdir -p $(BINDIR)
	$(CXX) -O2 -std=c++17 $< -o $@ -L$(LIBDIR) -lbpgpu -Wl,-rpath,'$$ORIGIN/../lib'

clean:
	rm -f $(LIBDIR)/*.so $(LIBDIR)/*.a $(LIBDIR)/*.o oracle/_build/*.so $(BINDIR)/*
.PHONY: all clean
