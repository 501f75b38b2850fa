#!/bin/bash
# One GPU-box call that produces the round's measurement artefacts under gpurun_out/<tag>_*:
#   1. plain default bench (incl. the epoch-loop leg)         -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command without the epoch-loop leg (other shapes, assembly launches: they would
#      mix into the timeline)                                 -> <tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json,
#                                                               <tag>_step_timeline.txt
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)    -> <tag>_pmc_traffic.json
#   4. SQ counters of the dense / dW / attention kernels -> <tag>_pmc_sq_counters.txt
# usage: scripts/profile_round.sh <tag>
set -o pipefail
tag=${1:-rX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
mkdir -p $out
cd $R
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
echo "[1] bench: $(cut -c1-160 $out/${tag}_bench.json)"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o p --output-format csv -- python3 $R/bench.py --no-epoch-loop > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err || { tail -5 $out/${tag}_rocprof.err; exit 2; }
st=$(ls $out/prof_$tag/*/*kernel_stats.csv $out/prof_$tag/*kernel_stats.csv 2>/dev/null | head -1)
tr=$(ls $out/prof_$tag/*/*kernel_trace.csv $out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1)
cp "$st" $out/${tag}_bench_kernel_stats.csv
python $R/scripts/step_timeline.py "$tr" --stats $out/${tag}_bench_kernel_stats_workload.csv > $out/${tag}_step_timeline.txt 2>&1
rm -rf $out/prof_$tag
echo "[2] rocprof stats: $(head -3 $out/${tag}_bench_kernel_stats.csv | cut -c1-200)"
short="--steps 5 --warmup 2 --prewarm 20 --no-cpu-baseline --no-roofline --no-epoch-loop --no-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o c --output-format csv -- python3 $R/bench.py $short > /dev/null 2> $out/${tag}_pmc_$c.err || { tail -5 $out/${tag}_pmc_$c.err; exit 3; }
done
f=$(ls $out/pmc_FETCH_SIZE/*/*counter_collection.csv $out/pmc_FETCH_SIZE/*counter_collection.csv 2>/dev/null | head -1)
w=$(ls $out/pmc_WRITE_SIZE/*/*counter_collection.csv $out/pmc_WRITE_SIZE/*counter_collection.csv 2>/dev/null | head -1)
python $R/scripts/pmc_traffic.py "$f" "$w" $out/${tag}_pmc_traffic.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
echo "[3] pmc traffic done"
rm -rf $out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_sq -o s --output-format csv -- python3 $R/bench.py $short > /dev/null 2> $out/${tag}_pmc_sq.err || { tail -5 $out/${tag}_pmc_sq.err; exit 4; }
q=$(ls $out/pmc_sq/*/*counter_collection.csv $out/pmc_sq/*counter_collection.csv 2>/dev/null | head -1)
{ echo "# shape: B64 T128 L20 C8 vdim1024 drop0.2 f32"; for k in dw_f16 conv_block_fwd conv_block_bwd ln_proj_kernel ln_proj_bwd da_post da_mid_bwd attn_fwd attn_bwd feature_ksplit mproj_kernel mproj_pair cq_fwd_wide cq_bwd_wide; do python $R/scripts/pmc_summary.py "$q" $k; done; } > $out/${tag}_pmc_sq_counters.txt
rm -rf $out/pmc_sq
echo "[4] sq counters done"
