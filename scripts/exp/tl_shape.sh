#!/bin/bash
# step timeline at another shape: scripts/exp/tl_shape.sh <tag> <bench args...>   (e.g. c4 --batch 32 --T 256)
tag=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_$tag -o p --output-format csv -- python3 $R/bench.py --steps 60 --warmup 5 --prewarm 60 --no-cpu-baseline --no-roofline --no-epoch-loop "$@" > $out/${tag}_bench.json 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 2; }
tr=$(ls $out/prof_$tag/*/*kernel_trace.csv $out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/step_timeline.py "$tr" > $out/${tag}_step_timeline.txt 2>&1
rm -rf $out/prof_$tag
sed -n '/^--- totals/,$p' $out/${tag}_step_timeline.txt | head -40
