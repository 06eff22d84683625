# Builds every native artefact in-tree (the .so files travel to the GPU box with the snapshot).
#   make            -> sift4g_amd/lib/libs4g_hip.so (HIP, gfx950), sift4g_amd/lib/libswsharp.so (C++ host API),
#                      oracle/libsw_oracle.so (test infrastructure)
#   make dropin     -> build/sift4g : the reference's own sift4g sources compiled IN PLACE from /root/reference
#                      against include/swsharp/*.h and linked with our libraries (only where /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
LIBDIR  := sift4g_amd/lib
CSRC    := sift4g_amd/csrc
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -fvisibility=hidden -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Iinclude

all: $(LIBDIR)/libs4g_hip.so oracle

$(LIBDIR)/libs4g_hip.so: $(CSRC)/s4g_hip.hip $(CSRC)/sw_kernels.hip.h $(CSRC)/matrices.h include/s4g_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/s4g_hip.hip

oracle:
	$(MAKE) -s -C oracle

isa: $(CSRC)/s4g_hip.hip $(CSRC)/sw_kernels.hip.h
	@mkdir -p build/isa
	cd build/isa && $(HIPCC) $(HIPFLAGS) -c -save-temps -Rpass-analysis=kernel-resource-usage ../../$(CSRC)/s4g_hip.hip -o s4g_hip.o

clean:
	rm -rf build $(LIBDIR)/*.so
	$(MAKE) -s -C oracle clean

.PHONY: all oracle isa clean
