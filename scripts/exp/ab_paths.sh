R=$GRAFT_REPO_ROOT
for cfg in "c2 " "c4 --batch 32 --T 256"; do
  set -- $cfg; tag=$1; shift
  HUAL_CQ_NO_WIDE=1 HUAL_ATTN_NO_BIG=1 bash $R/scripts/exp/tl_shape.sh ${tag}_old "$@" > /dev/null
  bash $R/scripts/exp/tl_shape.sh ${tag}_new "$@" > /dev/null
  HUAL_CQ_NO_WIDE=1 HUAL_ATTN_NO_BIG=1 bash $R/scripts/exp/tl_shape.sh ${tag}_old2 "$@" > /dev/null
  bash $R/scripts/exp/tl_shape.sh ${tag}_new2 "$@" > /dev/null
  for v in old new old2 new2; do echo "$tag $v: $(grep totals $R/gpurun_out/${tag}_${v}_step_timeline.txt | cut -c1-45) | $(grep -h 'cq_\|tri_prep' $R/gpurun_out/${tag}_${v}_step_timeline.txt | grep ' x1 ' | awk '{printf "%s %s us; ", $1, $3}')$(grep -h 'attn_bwd' $R/gpurun_out/${tag}_${v}_step_timeline.txt | grep ' x[0-9]' | awk '{printf "%s %s %s us; ", $1, $2, $3}')"; done
done
