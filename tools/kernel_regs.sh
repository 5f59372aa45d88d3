#!/bin/bash
# Per-kernel register / spill report of one .hip file (cross-compiles for gfx950, no GPU needed).
f=${1:?usage: kernel_regs.sh file.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /SGPRs Spill:/ {ss=$(NF-1)} /VGPRs Spill:/ {vs=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {print name, "vgpr="v, "agpr="ag, "occ="oc, "sspill="ss, "vspill="vs}' | c++filt | sed 's/amber:://'
