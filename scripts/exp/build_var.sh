#!/bin/bash
# build a variant of the CURRENT tree with extra -D switches: scripts/exp/build_var.sh <out.so> DEF1 [DEF2 ...]
# (through hual_amd.build: the same global AND per-file flags as the in-tree library - an A/B against "cur" differs by the switches only)
set -e
out=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
args=()
for d in "$@"; do args+=(--define "$d"); done
cd $R && python -m hual_amd.build --out $out "${args[@]}" > /dev/null
echo built $out with "$@"
