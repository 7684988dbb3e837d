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

all: host hip oracle

host: $(PKG)/librt_host.so
hip: $(PKG)/libraytracer_amd.so

$(PKG)/librt_host.so: $(HOST_SRC) include/rt_host.h include/rt_abi.h
	$(CXX) -O2 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude -o $@ $(HOST_SRC) -lz

# -ffp-contract=off: the reference-precision device code must not fuse a*b+c;
# the FP32 filter asks for its FMAs explicitly.
$(PKG)/libraytracer_amd.so: $(HIP_SRC) $(HIP_HDR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude -I$(PKG)/csrc \
	  -o $@ $(HIP_SRC)

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(PKG)/*.so oracle/*.so

.PHONY: all host hip oracle clean
