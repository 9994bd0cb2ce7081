# Build everything in-tree.  `make` = host library + HIP library + CLI; `make oracle` = the test checker.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
CXXFLAGS = -O2 -std=c++17 -Wall -Wextra -fPIC -Iinclude
HIPFLAGS = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Wall -Wno-unused-result

LIBDIR = seeksv_amd/lib
HOST_SRC = seeksv_amd/host/bam_reader.cpp seeksv_amd/host/bam_partition.cpp seeksv_amd/host/getsv_plan.cpp
HIP_SRC  = seeksv_amd/csrc/seeksv_hip.hip
HIP_DEPS = $(wildcard seeksv_amd/csrc/*.h) $(wildcard seeksv_amd/csrc/*.hip) include/seeksv_hip.h

all: host hip synth cli

host: $(LIBDIR)/libseeksv_host.so
hip: $(LIBDIR)/libseeksv_hip.so
synth: $(LIBDIR)/libseeksv_synth.so $(LIBDIR)/libseeksv_synth_cpu.so
cli: seeksv_amd/bin/seeksv

$(LIBDIR)/libseeksv_host.so: $(HOST_SRC) $(wildcard seeksv_amd/host/*.h) include/seeksv_host.h include/seeksv_hip.h
	mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lz -lpthread

$(LIBDIR)/libseeksv_hip.so: $(HIP_DEPS)
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

# the `seeksv` command line (getclip / getsv) over the two libraries
seeksv_amd/bin/seeksv: seeksv_amd/host/seeksv_cli.cpp seeksv_amd/host/junction_stage.cpp seeksv_amd/host/junction_stage.h seeksv_amd/host/somatic_stage.cpp seeksv_amd/host/somatic_stage.h $(LIBDIR)/libseeksv_host.so $(LIBDIR)/libseeksv_hip.so
	mkdir -p seeksv_amd/bin
	$(CXX) -O2 -std=c++17 -Wall -Wextra -Iinclude -o $@ seeksv_amd/host/seeksv_cli.cpp seeksv_amd/host/junction_stage.cpp seeksv_amd/host/somatic_stage.cpp -L$(LIBDIR) -lseeksv_host -lseeksv_hip -lz \
		-Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,$(ROCM)/lib

# synthetic BAM-record generator: the same source compiled for the GPU (bench) and for the CPU (tests, cpu baseline)
$(LIBDIR)/libseeksv_synth.so: seeksv_amd/csrc/synth.hip seeksv_amd/csrc/synth_core.h
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ seeksv_amd/csrc/synth.hip

$(LIBDIR)/libseeksv_synth_cpu.so: seeksv_amd/csrc/synth_cpu.cpp seeksv_amd/csrc/synth_core.h
	mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ seeksv_amd/csrc/synth_cpu.cpp

oracle:
	$(MAKE) -C oracle liboracle.so

oracle-ref:
	$(MAKE) -C oracle ref

# Sanitizer pass over everything that runs on the CPU (GPU AddressSanitizer is not available on the pool): host library, CPU build of the
# generator, the oracle and the CLI's host stages under ASan + UBSan, then the CPU test suite against that build.
ASAN_DIR ?= /tmp/seeksv_asan
ASAN_FLAGS = -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined
asan: $(LIBDIR)/libseeksv_hip.so
	mkdir -p $(ASAN_DIR)
	$(CXX) $(ASAN_FLAGS) -std=c++17 -Wall -Wextra -fPIC -Iinclude -shared -o $(ASAN_DIR)/libseeksv_host.so $(HOST_SRC) -lz -lpthread
	$(CXX) $(ASAN_FLAGS) -std=c++17 -Wall -Wextra -fPIC -Iinclude -shared -o $(ASAN_DIR)/libseeksv_synth_cpu.so seeksv_amd/csrc/synth_cpu.cpp
	gcc $(ASAN_FLAGS) -std=c11 -Wall -Wextra -fPIC -shared -o $(ASAN_DIR)/liboracle.so oracle/seeksv_oracle.c -lm
	cp $(LIBDIR)/libseeksv_hip.so $(ASAN_DIR)/
	$(CXX) $(ASAN_FLAGS) -std=c++17 -Wall -Wextra -Iinclude -o $(ASAN_DIR)/seeksv seeksv_amd/host/seeksv_cli.cpp seeksv_amd/host/junction_stage.cpp seeksv_amd/host/somatic_stage.cpp \
		-L$(ASAN_DIR) -lseeksv_host -lseeksv_hip -lz -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,$(ROCM)/lib
	$(CXX) $(ASAN_FLAGS) -std=c++17 -Iseeksv_amd/csrc tests/native/inflate_check.cpp -lz -o $(ASAN_DIR)/inflate_check && $(ASAN_DIR)/inflate_check
	LD_PRELOAD=$$(gcc -print-file-name=libasan.so):$$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 SSV_CLEAN_EXIT=1 \
		SSV_TEST_CXXFLAGS="$(ASAN_FLAGS)" SSV_LIBDIR=$(abspath $(ASAN_DIR)) SSV_ORACLE_SO=$(abspath $(ASAN_DIR))/liboracle.so SSV_CLI=$(abspath $(ASAN_DIR))/seeksv \
		python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf $(LIBDIR) && $(MAKE) -C oracle clean

.PHONY: all host hip synth cli oracle oracle-ref asan clean
