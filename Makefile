# Builds the gfx950 engine (libbpgpu.so), the oracle's C restatement and the
# host-simulation test harness.  `make -j8`.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function
CSRC     := bulletproofs_amd/csrc
OBJDIR   := build
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
HEADERS  := $(wildcard $(CSRC)/*.h) include/bpgpu.h

all: bulletproofs_amd/libbpgpu.so tests/hostsim/libhostsim.so benchmarks/libbpbench.so oracle

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HEADERS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

bulletproofs_amd/libbpgpu.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS)

tests/hostsim/libhostsim.so: tests/hostsim/hostsim.cpp $(HEADERS)
	g++ -O2 -std=c++17 -shared -fPIC -o $@ $<

benchmarks/libbpbench.so: benchmarks/proofgen.hip bulletproofs_amd/libbpgpu.so $(HEADERS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $< -Lbulletproofs_amd -lbpgpu -Wl,-rpath,'$$ORIGIN/../bulletproofs_amd'

benchmarks/microbench: benchmarks/microbench.hip $(HEADERS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -o $@ $<

oracle:
	@if [ -f oracle/Makefile ]; then $(MAKE) -C oracle; fi

clean:
	rm -rf $(OBJDIR) bulletproofs_amd/libbpgpu.so tests/hostsim/libhostsim.so benchmarks/microbench benchmarks/libbpbench.so

.PHONY: all clean oracle
