#!/bin/bash
# Development aid: register / spill / scratch figures of every kernel in csrc/render_kernel.hip, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage), for both compilations (default and -DDSRT_DEVICE_LIBM).  Usage: tools/kernel_resources.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
for variant in "" "-DDSRT_DEVICE_LIBM"; do
  echo "== render_kernel.hip $variant"
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $variant "$@" \
      -Rpass-analysis=kernel-resource-usage -c deep-space-ray-tracer_amd/csrc/render_kernel.hip -o /dev/null 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} /TotalSGPRs:/ {s=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {sc=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF}
       /LDS Size/ {printf "%s VGPR %s SGPR %s sgpr_spill %s vgpr_spill %s scratch %s LDS %s\n", name, v, s, ss, vs, sc, $NF}' | c++filt | sed 's/dsrt:://g; s/(RenderArgs)//; s/void //'
done
