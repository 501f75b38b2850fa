#!/bin/bash
# VALU issue share of every kernel of the step: SQ_ACTIVE_INST_VALU (cycles a wave has a VALU instruction executing) against
# SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE; SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_INSTS_VMEM per wave.  One PMC pass, eager steps.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
short="--steps 5 --warmup 2 --prewarm 20 --no-cpu-baseline --no-roofline --no-graph"
rm -rf $out/pmc_valu
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_valu -o s --output-format csv -- python3 $R/bench.py $short > /dev/null 2> $out/pmc_valu.err || { tail -5 $out/pmc_valu.err; exit 4; }
q=$(ls $out/pmc_valu/*/*counter_collection.csv $out/pmc_valu/*counter_collection.csv 2>/dev/null | head -1)
python3 $R/scripts/exp/pmc_valu.py "$q" > $out/${1:-r4}_pmc_valu.txt
rm -rf $out/pmc_valu
cat $out/${1:-r4}_pmc_valu.txt
