#!/bin/bash
# register / spill / scratch / LDS figures of every kernel of one translation unit: scripts/exp/regs.sh attn [extra hipcc flags]
f=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/hual_regs && cd /tmp/hual_regs || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I$R/include -x hip -c $R/hual_amd/csrc/$f.hip -o $f.o -save-temps=obj "$@" 2>/dev/null
awk '/^[ ]+\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{s=$2} /\.sgpr_spill_count:/{ss=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{printf "%-90s vgpr %3s vspill %3s sspill %3s scratch %4s lds %6s\n", substr(n,1,90), v, s, ss, p, l}' $f-hip-amdgcn-amd-amdhsa-gfx950.s
