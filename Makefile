# Builds every native artefact in-tree (the .so files travel to the GPU box with the snapshot).
#   make            -> sift4g_amd/lib/libs4g_hip.so (HIP, gfx950), sift4g_amd/lib/libswsharp.so (C++ host API),
#                      oracle/libsw_oracle.so (test infrastructure)
#   make dropin-batched / dropin-fullscan / dropin-devsearch -> the same CLI with some of the reference's source files replaced (see below)
#   make dropin     -> build/sift4g : the reference's own sift4g sources compiled IN PLACE from /root/reference
#                      against include/swsharp/*.h and linked with our libraries (only where /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
# gfx950 (MI355X) ONLY: the kernels use what that target has and others lack -- 160 KB of LDS per workgroup (topk_rows declares 65 KB
# of static LDS, over the 64 KB limit of gfx90a / gfx942), v_pk_maximum3_f16, the packed kernels' denormal mode.  Another ARCH would
# stop somewhere inside the build with an unhelpful message; say it here instead.
ifneq ($(ARCH),gfx950)
$(error this library is written for gfx950 (MI355X) only: ARCH=$(ARCH) is not supported -- see DESIGN.md "No compatibility layers")
endif
LIBDIR  := sift4g_amd/lib
CSRC    := sift4g_amd/csrc
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -fvisibility=hidden -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Iinclude

SWSRC   := $(CSRC)/swsharp/core.cpp $(CSRC)/swsharp/evalue.cpp $(CSRC)/swsharp/align.cpp $(CSRC)/swsharp/output.cpp
REF     ?= /root/reference

all: $(LIBDIR)/libs4g_hip.so $(LIBDIR)/libswsharp.so build/shotgun_demo build/gather_demo build/pin_dump_ours oracle

build/shotgun_demo: tools/shotgun_demo.cpp $(LIBDIR)/libswsharp.so
	@mkdir -p build
	$(CXX) -O2 -std=c++17 -Iinclude -o $@ tools/shotgun_demo.cpp -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

# this repo's half of the pinning recipe (oracle/Makefile target _ref, oracle/pin_against_swsharp.py): oracle/pin_dump.cpp -- a program
# written against the swsharp API the reference uses -- on OUR headers and libraries; the other half is the same source on upstream swsharp
pin-dump: build/pin_dump_ours
build/pin_dump_ours: oracle/pin_dump.cpp $(LIBDIR)/libswsharp.so
	@mkdir -p build
	$(CXX) -O2 -std=c++11 -Wno-write-strings -Iinclude -o $@ oracle/pin_dump.cpp -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

# the one-process-per-GPU layout in plain C++ over the C ABI: N ranks (child processes), scan + s4g_gather_topk (RCCL), self-checking
build/gather_demo: tools/gather_demo.cpp include/s4g_hip.h $(LIBDIR)/libs4g_hip.so
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Iinclude -o $@ tools/gather_demo.cpp -L$(LIBDIR) -ls4g_hip -lpthread -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

# libs4g_hip.so: one object per translation unit (make -j builds them side by side), every unit depends on every header
HIPSRC  := $(CSRC)/s4g_core.hip $(CSRC)/s4g_scan.hip $(CSRC)/s4g_pairs.hip $(CSRC)/s4g_trace.hip $(CSRC)/s4g_diag.hip $(CSRC)/s4g_lists.hip $(CSRC)/s4g_prefilter.hip $(CSRC)/s4g_chains.hip $(CSRC)/s4g_stream.hip $(CSRC)/s4g_gather.hip
HIPOBJ  := $(patsubst $(CSRC)/%.hip,build/obj/%.o,$(HIPSRC))
HIPHDR  := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) include/s4g_hip.h

build/obj/%.o: $(CSRC)/%.hip $(HIPHDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIBDIR)/libs4g_hip.so: $(HIPOBJ) tools/check_denorm.py
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@.tmp $(HIPOBJ) -ldl
	@# the packed kernels' integer max3 needs f16 denormals preserved: refuse a build whose kernel descriptors say otherwise
	python3 tools/check_denorm.py $@.tmp || { rm -f $@.tmp; exit 1; }
	mv $@.tmp $@

$(LIBDIR)/libswsharp.so: $(SWSRC) $(CSRC)/swsharp/impl.hpp $(CSRC)/matrices.h $(CSRC)/swsharp/libswsharp.map include/swsharp/swsharp.h include/swsharp/evalue.h $(LIBDIR)/libs4g_hip.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(SWSRC) -L$(LIBDIR) -ls4g_hip -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,--version-script=$(CSRC)/swsharp/libswsharp.map

# Drop-in link test: the reference's own, unmodified sift4g sources compiled where they lie, against OUR
# swsharp-compatible headers and libraries.  Not an oracle (it runs our scorer); needs a GPU to run.
dropin: $(LIBDIR)/libswsharp.so
	@test -d $(REF)/sift4g/src || { echo "no reference tree at $(REF): skipping drop-in build"; exit 0; }
	@mkdir -p build/dropin
	$(CXX) -std=c++11 -O2 -Wno-write-strings -Iinclude -o build/dropin/sift4g $(REF)/sift4g/src/*.cpp \
	    -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

# The same with ONE source file of the reference replaced: its alignment wrapper (database_alignment.cpp: one scorer call per
# query and chunk) by sift4g_amd/csrc/dropin/batched_database_alignment.cpp (one scoring and one traceback submission per
# chunk for all queries).  Everything else -- main, prefilter, selection, SIFT -- is the reference's own code, unmodified.
dropin-batched: $(LIBDIR)/libswsharp.so sift4g_amd/csrc/dropin/batched_database_alignment.cpp
	@test -d $(REF)/sift4g/src || { echo "no reference tree at $(REF): skipping drop-in build"; exit 0; }
	@mkdir -p build/dropin
	$(CXX) -std=c++11 -O2 -Wno-write-strings -Iinclude -I$(REF)/sift4g/src -o build/dropin/sift4g_batched \
	    $(filter-out %/database_alignment.cpp,$(wildcard $(REF)/sift4g/src/*.cpp)) sift4g_amd/csrc/dropin/batched_database_alignment.cpp \
	    -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

# The full-scan build: TWO source files of the reference replaced -- the k-mer prefilter by a pass that only
# counts the database's residues (count_database_search.cpp) and the alignment wrapper by one shotgunDatabaseEValues call per
# chunk over ALL its sequences (fullscan_database_alignment.cpp).  main, selection and SIFT are the reference's own code.
dropin-fullscan: $(LIBDIR)/libswsharp.so sift4g_amd/csrc/dropin/fullscan_database_alignment.cpp sift4g_amd/csrc/dropin/count_database_search.cpp
	@test -d $(REF)/sift4g/src || { echo "no reference tree at $(REF): skipping drop-in build"; exit 0; }
	@mkdir -p build/dropin
	$(CXX) -std=c++11 -O2 -Wno-write-strings -Iinclude -I$(REF)/sift4g/src -o build/dropin/sift4g_fullscan \
	    $(filter-out %/database_alignment.cpp %/database_search.cpp,$(wildcard $(REF)/sift4g/src/*.cpp)) \
	    sift4g_amd/csrc/dropin/fullscan_database_alignment.cpp sift4g_amd/csrc/dropin/count_database_search.cpp \
	    -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

# The device-prefilter build: the k-mer prefilter's source files (database_search.cpp, hash.cpp) replaced by
# device_database_search.cpp (index, hit lists, LIS and best lists on the device: s4g_prefilter_*) and the alignment wrapper by
# the batched one.  main, selection and SIFT are the reference's own code.
dropin-devsearch: $(LIBDIR)/libswsharp.so sift4g_amd/csrc/dropin/device_database_search.cpp sift4g_amd/csrc/dropin/batched_database_alignment.cpp
	@test -d $(REF)/sift4g/src || { echo "no reference tree at $(REF): skipping drop-in build"; exit 0; }
	@mkdir -p build/dropin
	$(CXX) -std=c++11 -O2 -Wno-write-strings -Iinclude -I$(REF)/sift4g/src -o build/dropin/sift4g_devsearch \
	    $(filter-out %/database_alignment.cpp %/database_search.cpp %/hash.cpp,$(wildcard $(REF)/sift4g/src/*.cpp)) \
	    sift4g_amd/csrc/dropin/batched_database_alignment.cpp sift4g_amd/csrc/dropin/device_database_search.cpp \
	    -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

# The reference's own prefilter (database_search.cpp, hash.cpp, utils.cpp: compiled where they lie) behind a main that prints
# the candidate lists: the generator of tests/golden/prefilter/ (tools/ref_prefilter_dump.cpp explains what it is and is not).
dropin-prefilter-dump: $(LIBDIR)/libswsharp.so tools/ref_prefilter_dump.cpp
	@test -d $(REF)/sift4g/src || { echo "no reference tree at $(REF): skipping drop-in build"; exit 0; }
	@mkdir -p build/dropin
	$(CXX) -std=c++11 -O2 -Wno-write-strings -Iinclude -I$(REF)/sift4g/src -o build/dropin/ref_prefilter_dump \
	    $(REF)/sift4g/src/database_search.cpp $(REF)/sift4g/src/hash.cpp $(REF)/sift4g/src/utils.cpp tools/ref_prefilter_dump.cpp \
	    -L$(LIBDIR) -lswsharp -ls4g_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

oracle:
	$(MAKE) -s -C oracle

isa: $(HIPSRC) $(HIPHDR)
	@mkdir -p build/isa
	cd build/isa && for f in $(HIPSRC); do $(HIPCC) $(HIPFLAGS) -c -save-temps -Rpass-analysis=kernel-resource-usage ../../$$f -o $$(basename $$f .hip).o || exit 1; done

clean:
	rm -rf build $(LIBDIR)/*.so
	$(MAKE) -s -C oracle clean

.PHONY: all oracle isa clean pf-timing pin-dump dropin dropin-batched dropin-fullscan dropin-devsearch dropin-prefilter-dump

# A measuring build of the library: pf_target_lis with shader-clock counters per section (kmer_prefilter.hip.h: PF_TIMING), printed per
# chunk on stderr.  S4G_LIB=build/pf_timing/libs4g_hip.so python tools/prefilter_call_probe.py
pf-timing: $(LIBDIR)/libs4g_hip.so
	@mkdir -p build/pf_timing
	$(HIPCC) $(HIPFLAGS) -DPF_TIMING -Iinclude -c $(CSRC)/s4g_prefilter.hip -o build/pf_timing/s4g_prefilter.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/pf_timing/libs4g_hip.so $(filter-out build/obj/s4g_prefilter.o,$(HIPOBJ)) build/pf_timing/s4g_prefilter.o -ldl
