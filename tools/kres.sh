#!/bin/bash
# register / scratch / occupancy of every kernel of one source file:  tools/kres.sh tf_kernels.hip [extra hipcc flags]
cd "$(dirname "$0")/../texturefusion_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
  -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
  awk '/Function Name:/{n=$NF} / VGPRs:/{v=$NF} /TotalSGPRs:/{s=$NF} /ScratchSize/{sc=$NF} /Occupancy/{o=$NF} /SGPRs Spill/{ss=$NF} /LDS Size/{print n, "VGPR", v, "SGPR", s, "(spilled " ss ")", "scratch", sc, "occ", o, "LDS", $NF}' | sort -u
