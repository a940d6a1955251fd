# Top-level build: the product library (HIP, gfx950 only) and, separately, the checker under oracle/.
#
#   make lib      deep-space-ray-tracer_amd/libdsrt_hip.so   (kernels + C ABI + host scene builder)
#   make tools    deep-space-ray-tracer_amd/dsrt_render       (CLI frame driver, mirrors src/main.cpp's flags)
#   make oracle   oracle/libdsrt_oracle*.so and, where /root/reference exists, oracle/_ref/ref_host
#   make all      everything
#
# hipcc cross-compiles for gfx950 without a GPU present.  -ffp-contract=off and the absence of any fast-math
# flag are part of the numerical contract, on both the device and the host side.

PKG      := deep-space-ray-tracer_amd
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
BUILD    := build

HOST_SRC := $(PKG)/host/obj_mesh.cpp $(PKG)/host/image_io.cpp $(PKG)/host/jpeg_decode.cpp $(PKG)/host/bmp_tga_decode.cpp $(PKG)/host/scene_flatten.cpp $(PKG)/host/bvh_median.cpp $(PKG)/host/bvh_sah.cpp $(PKG)/host/pose_camera.cpp
HIP_SRC  := $(PKG)/csrc/render_kernel.hip $(PKG)/csrc/device_api.hip $(PKG)/csrc/microbench.hip $(PKG)/csrc/multi_gpu.hip $(PKG)/csrc/bvh_lbvh.hip
HOST_OBJ := $(patsubst $(PKG)/host/%.cpp,$(BUILD)/host_%.o,$(HOST_SRC))
HIP_OBJ  := $(patsubst $(PKG)/csrc/%.hip,$(BUILD)/hip_%.o,$(HIP_SRC))
HEADERS  := $(wildcard include/*.h) $(wildcard $(PKG)/host/*.hpp) $(wildcard $(PKG)/csrc/*.h)

CXXFLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter
HIPFLAGS := -std=c++17 -O3 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-parameter

all: lib tools oracle

lib: $(PKG)/libdsrt_hip.so

$(BUILD)/host_%.o: $(PKG)/host/%.cpp $(HEADERS)
	@mkdir -p $(BUILD)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(BUILD)/hip_%.o: $(PKG)/csrc/%.hip $(HEADERS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# csrc/render_kernel.hip is compiled twice: as it is, and with -DDSRT_DEVICE_LIBM (the kernels of DsrtRenderDesc.math_mode 1, in namespace dsrt::devlibm)
$(BUILD)/hip_render_kernel_devlibm.o: $(PKG)/csrc/render_kernel.hip $(HEADERS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -DDSRT_DEVICE_LIBM -c $< -o $@

$(PKG)/libdsrt_hip.so: $(HOST_OBJ) $(HIP_OBJ) $(BUILD)/hip_render_kernel_devlibm.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -lz -lrccl

tools: $(PKG)/dsrt_render $(PKG)/main_flow_driver

$(PKG)/dsrt_render: $(PKG)/tools/dsrt_render.cpp $(PKG)/libdsrt_hip.so $(HEADERS)
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(PKG) -ldsrt_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib

$(PKG)/main_flow_driver: $(PKG)/tools/main_flow_driver.cpp $(PKG)/libdsrt_hip.so $(HEADERS)
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(PKG) -ldsrt_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(BUILD) $(PKG)/libdsrt_hip.so $(PKG)/dsrt_render $(PKG)/main_flow_driver
	$(MAKE) -C oracle clean

.PHONY: all lib tools oracle clean
