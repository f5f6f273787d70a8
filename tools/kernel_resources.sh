#!/bin/bash
# usage: tools/kernel_resources.sh [extra hipcc flags] — VGPRs / scratch / spills / occupancy / LDS of every kernel of kernels.hip as the
# Makefile compiles it (LLVM's kernel-resource-usage remarks; no GPU needed)
cd "$(dirname "$0")/../platinum_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage "$@" -c kernels.hip -o /tmp/kres.o 2>&1 |
  awk '/remark: Function Name:/ {name=$5} /remark:     VGPRs:/ {v=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /VGPRs Spill/ {sp=$5} /LDS Size/ {print name, "VGPRs", v, "scratch", s, "spill", sp, "occ", o, "LDS", $6}' | c++filt | sed -e 's/pt:://g' -e 's/(.*)//' | sort -u
