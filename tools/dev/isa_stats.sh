#!/bin/bash
# dev tool: compile one csrc/*.hip with -save-temps into /tmp/isa_<name>/ and print per-kernel VGPR / spill / LDS figures
#   tools/dev/isa_stats.sh match [kernel-name-filter]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
n=$1; f=${2:-.}
O=/tmp/isa_$n; mkdir -p $O
(cd $O && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -I$R/cvpce_amd/csrc -c $R/cvpce_amd/csrc/$n.hip -o $O/$n.o -save-temps=obj 2>&1 | grep -E "error|warning: v" || true)
S=$O/$n-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.agpr_count" $S | paste - - - - - | grep -E "$f" | sed 's/\s\+/ /g'
echo "asm: $S"
