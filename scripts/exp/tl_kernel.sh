#!/bin/bash
# per-launch durations of the kernels matching <pattern> in the bench step, with a variant library: tl_kernel.sh <tag> <lib.so|cur> <pattern> [bench args]
tag=$1; libv=$2; pat=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT
[ "$libv" = cur ] && libv=hual_amd/libhual_seqpan.so
export HUAL_LIB_PATH=$R/$libv
out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_$tag -o p --output-format csv -- python3 $R/bench.py --steps 60 --warmup 5 --prewarm 60 --no-cpu-baseline --no-roofline --no-epoch-loop "$@" > /dev/null 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 2; }
tr=$(ls $out/prof_$tag/*/*kernel_trace.csv $out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/scripts/step_timeline.py "$tr" > $out/${tag}_step_timeline.txt 2>&1
rm -rf $out/prof_$tag
echo "== $tag"; grep -E "$pat|totals" $out/${tag}_step_timeline.txt
