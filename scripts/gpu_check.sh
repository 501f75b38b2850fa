#!/bin/bash
# One GPU-box call: parity tests of the model path, then a kernel-trace timeline of the bench step.
#   scripts/gpu_check.sh <tag> [pytest args]
set -o pipefail
tag=${1:-chk}; shift
out=gpurun_out
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py "$@" -x -q > $out/${tag}_tests.log 2>&1
rc=$?
tail -5 $out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$out/tl_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/tl_$tag -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/$out/${tag}_bench.err
cd $GRAFT_REPO_ROOT
python scripts/step_timeline.py $(ls $out/tl_$tag/*/*kernel_trace.csv $out/tl_$tag/*kernel_trace.csv 2>/dev/null | head -1) > $out/${tag}_timeline.txt 2>&1
rm -rf $out/tl_$tag
cut -c1-200 $out/${tag}_bench.json
grep -A60 "^--- totals" $out/${tag}_timeline.txt | head -50
