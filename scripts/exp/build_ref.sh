#!/bin/bash
# build the library of another git revision for same-box A/B timelines (scripts/exp/tl_variant.sh):
#   scripts/exp/build_ref.sh <git-ref> <out.so>       e.g.  scripts/exp/build_ref.sh HEAD build_exp/base.so
set -e
ref=$1; out=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d /tmp/hual_ref.XXXXXX)
git -C $R archive $ref hual_amd/csrc include | tar -x -C $tmp
mkdir -p $tmp/obj $(dirname $R/$out)
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $EXTRA_FLAGS"
pids=()
for f in $tmp/hual_amd/csrc/*.hip $tmp/hual_amd/csrc/*.cpp; do
  x=""; [[ $f == *.hip ]] && x="-x hip"
  # per-file flags of hual_amd/build.py FILE_FLAGS (an A/B must not differ by them)
  [[ $(basename $f) == convblock.hip ]] && x="$x -mllvm -amdgpu-sched-strategy=max-ilp"
  /opt/rocm/bin/hipcc $FLAGS $x -c $f -o $tmp/obj/$(basename $f).o 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$out $tmp/obj/*.o
rm -rf $tmp
echo built $out from $ref
