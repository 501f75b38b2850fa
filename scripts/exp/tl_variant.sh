#!/bin/bash
# timeline of the bench step with a variant library: scripts/exp/tl_variant.sh <tag> <lib.so> [bench args]
tag=$1; libv=$2; shift; shift
R=$GRAFT_REPO_ROOT
export HUAL_LIB_PATH=$R/$libv      # hual_amd/lib.py loads this file instead of the in-tree library (nothing is overwritten)
out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_$tag -o p --output-format csv -- python3 $R/bench.py --steps 60 --warmup 5 --prewarm 60 --no-cpu-baseline --no-roofline --no-epoch-loop "$@" > /dev/null 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 2; }
tr=$(ls $out/prof_$tag/*/*kernel_trace.csv $out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/step_timeline.py "$tr" > $out/${tag}_step_timeline.txt 2>&1
rm -rf $out/prof_$tag
echo "== $tag"; grep -E "attn|totals" $out/${tag}_step_timeline.txt | tail -4
