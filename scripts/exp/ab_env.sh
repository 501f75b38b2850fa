#!/bin/bash
# same-box A/B of a path switched by an environment variable: scripts/exp/ab_env.sh <tag> <VAR> [bench args]   (VAR=1: the old path)
tag=$1; var=$2; shift; shift
R=$GRAFT_REPO_ROOT
env $var=1 bash $R/scripts/exp/tl_shape.sh ${tag}_old "$@" > /dev/null || exit 2
bash $R/scripts/exp/tl_shape.sh ${tag}_new "$@" > /dev/null || exit 2
env $var=1 bash $R/scripts/exp/tl_shape.sh ${tag}_old2 "$@" > /dev/null || exit 2
bash $R/scripts/exp/tl_shape.sh ${tag}_new2 "$@" > /dev/null || exit 2
for v in old new old2 new2; do echo "$tag $v: $(grep totals $R/gpurun_out/${tag}_${v}_step_timeline.txt | cut -c1-45) | $(head -1 $R/gpurun_out/${tag}_${v}_step_timeline.txt | cut -c1-22) | $(grep -h 'conv_block_fwd\|ln_proj_kernel\|da_post' $R/gpurun_out/${tag}_${v}_step_timeline.txt | grep ' x[0-9]' | awk '{printf "%s %s %s us; ", substr($1,1,28), $2, $3}')"; done
