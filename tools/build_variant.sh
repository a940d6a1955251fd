#!/bin/bash
# Development aid: a second libdsrt_hip.so that differs from the tree's in extra flags for csrc/render_kernel.hip (both compilations), for interleaved A/B
# runs with tools/ab_lib.py.  Usage: tools/build_variant.sh <name> [flags...]   ->  variants_tmp/libdsrt_<name>.so   (needs `make lib` to have run)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants_tmp
F="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-parameter"
/opt/rocm/bin/hipcc $F "$@" -c deep-space-ray-tracer_amd/csrc/render_kernel.hip -o variants_tmp/rk_$name.o &
/opt/rocm/bin/hipcc $F "$@" -DDSRT_DEVICE_LIBM -c deep-space-ray-tracer_amd/csrc/render_kernel.hip -o variants_tmp/rk_${name}_devlibm.o &
wait
others=$(ls build/host_*.o build/hip_*.o | grep -v "hip_render_kernel")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants_tmp/libdsrt_$name.so $others variants_tmp/rk_$name.o variants_tmp/rk_${name}_devlibm.o -lz -lrccl
echo variants_tmp/libdsrt_$name.so
