#!/bin/bash
# VGPRs / spills / scratch of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage): usage tools/kernel_regs.sh <file.hip> [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kernel_regs_$$.o 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill:/ {sp=$(NF-1)} /LDS Size/ {print name, "vgpr=" v, "vgpr_spill=" sp, "scratch=" s}' |
  c++filt | sed 's/void mb:://; s/(mb::GemmArgs, int, int)//'
rm -f /tmp/kernel_regs_$$.o
