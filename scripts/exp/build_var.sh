#!/bin/bash
# build a variant of the CURRENT tree with extra -D switches: scripts/exp/build_var.sh <out.so> DEF1 [DEF2 ...]
set -e
out=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d /tmp/hual_var.XXXXXX)
mkdir -p $tmp/obj $(dirname $R/$out)
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -I$R/include $EXTRA_FLAGS"
for d in "$@"; do FLAGS="$FLAGS -D$d"; done
pids=()
for f in $R/hual_amd/csrc/*.hip $R/hual_amd/csrc/*.cpp; do
  x=""; [[ $f == *.hip ]] && x="-x hip"
  /opt/rocm/bin/hipcc $FLAGS $x -c $f -o $tmp/obj/$(basename $f).o 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$out $tmp/obj/*.o
rm -rf $tmp
echo built $out with "$@"
