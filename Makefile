# Builds every native piece in-tree:
#   raytracer_amd/librt_host.so        host loader + PNG writer (g++)
#   raytracer_amd/libraytracer_amd.so  HIP kernels + C ABI for gfx950 (hipcc)
#   oracle/librt_oracle.so, oracle/_ref/ref_render   test infrastructure (see oracle/Makefile)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
PKG := raytracer_amd
HOST_SRC := $(PKG)/host/scene_loader.cpp $(PKG)/host/png_writer.cpp
HIP_SRC := $(wildcard $(PKG)/csrc/*.hip)
HIP_HDR := $(wildcard $(PKG)/csrc/*.h) include/rt_abi.h

all: host hip cli oracle

host: $(PKG)/librt_host.so
hip: $(PKG)/libraytracer_amd.so
cli: $(PKG)/rt_render

$(PKG)/librt_host.so: $(HOST_SRC) include/rt_host.h include/rt_abi.h
	$(CXX) -O2 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -o $@ $(HOST_SRC) -lz -pthread

# -ffp-contract=off: the reference-precision device code must not fuse a*b+c;
# the FP32 filter asks for its FMAs explicitly.
$(PKG)/libraytracer_amd.so: $(HIP_SRC) $(HIP_HDR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude -I$(PKG)/csrc \
	  $(EXTRA) -o $@ $(HIP_SRC)

# The host driver (the reference's main): links both libraries, found at run time next to itself.
$(PKG)/rt_render: $(PKG)/host/rt_cli.cpp $(PKG)/librt_host.so $(PKG)/libraytracer_amd.so include/rt_abi.h include/rt_host.h
	$(CXX) -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $@ $(PKG)/host/rt_cli.cpp -L$(PKG) -lrt_host -lraytracer_amd \
	  -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(PKG)/*.so $(PKG)/rt_render oracle/*.so
	rm -rf asan_build

# instruction-rate microbenchmarks behind the numbers quoted for the filter kernel
stream_queue_probe:
	$(HIPCC) --offload-arch=gfx950 -O2 -o tools/stream_queue_probe tools/stream_queue_probe.hip

microbench:
	for b in microbench microbench2 microbench3 microbench4 microbench5; do $(HIPCC) --offload-arch=gfx950 -O3 -o tools/$$b tools/$$b.hip || exit 1; done

# Host-side AddressSanitizer build of both libraries and the driver into asan_build/
# (device code is not instrumented).  tools/asan_cli.sh runs it; delete the directory
# afterwards, it is uploaded with every GPU call while it exists.
ASANXX ?= /opt/rocm/lib/llvm/bin/amdclang++
asan:
	mkdir -p asan_build
	$(HIPCC) --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Xarch_host -fsanitize=address \
	  -Iinclude -I$(PKG)/csrc -o asan_build/libraytracer_amd.so $(HIP_SRC)
	$(ASANXX) -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address -fPIC -shared -Iinclude -o asan_build/librt_host.so $(HOST_SRC) -lz -pthread
	$(ASANXX) -O1 -g -std=c++17 -pthread -fsanitize=address -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o asan_build/rt_render $(PKG)/host/rt_cli.cpp \
	  -Lasan_build -lrt_host -lraytracer_amd -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

.PHONY: all host hip cli oracle clean microbench asan stream_queue_probe

# A build of the device library with extra flags under variants/ (A/B measurements, diagnostic builds):
#   make variant NAME=phase FLAGS=-DRT_PHASE_CLOCK  ->  variants/libraytracer_amd_phase.so  (RAYTRACER_AMD_LIB selects it)
variant:
	mkdir -p variants
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude -I$(PKG)/csrc \
	  $(FLAGS) -o variants/libraytracer_amd_$(NAME).so $(HIP_SRC)
.PHONY: variant
