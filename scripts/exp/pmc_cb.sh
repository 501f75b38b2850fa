#!/bin/bash
# SQ counters of the fused conv_block kernels (one PMC pass, eager steps)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
short="--steps 5 --warmup 2 --prewarm 20 --no-cpu-baseline --no-roofline --no-graph"
rm -rf $out/pmc_cb
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace -d $out/pmc_cb -o s --output-format csv -- python3 $R/bench.py $short > /dev/null 2> $out/pmc_cb.err || { tail -5 $out/pmc_cb.err; exit 4; }
q=$(ls $out/pmc_cb/*/*counter_collection.csv $out/pmc_cb/*counter_collection.csv 2>/dev/null | head -1)
for k in da_post_kernel da_mid_bwd_kernel ln_proj_kernel ln_proj_bwd_kernel conv_block_fwd; do python $R/scripts/pmc_summary.py "$q" "$k"; done > $out/pmc_cb.txt
rm -rf $out/pmc_cb
cat $out/pmc_cb.txt
