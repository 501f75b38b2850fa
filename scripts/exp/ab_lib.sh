#!/bin/bash
# same-box A/B of the in-tree library against hual_amd/variants/base.so (built from the commit in front): scripts/exp/ab_lib.sh <tag> <kernel regex>
tag=$1; pat=$2
R=$GRAFT_REPO_ROOT
HUAL_LIB_PATH=$R/hual_amd/variants/base.so bash $R/scripts/exp/tl_shape.sh ${tag}_old > /dev/null || exit 2
bash $R/scripts/exp/tl_shape.sh ${tag}_new > /dev/null || exit 2
HUAL_LIB_PATH=$R/hual_amd/variants/base.so bash $R/scripts/exp/tl_shape.sh ${tag}_old2 > /dev/null || exit 2
bash $R/scripts/exp/tl_shape.sh ${tag}_new2 > /dev/null || exit 2
for v in old new old2 new2; do echo "$tag $v: $(grep totals $R/gpurun_out/${tag}_${v}_step_timeline.txt | cut -c1-45)"; grep -h "$pat" $R/gpurun_out/${tag}_${v}_step_timeline.txt | grep ' x[0-9]'; done
