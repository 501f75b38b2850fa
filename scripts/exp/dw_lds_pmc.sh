# GPU box: LDS counters of the weight-gradient kernel (HUAL_DW_DBG as exported by the caller)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
rm -rf $out/pmc_lds
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $out/pmc_lds -o c --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/pmc_lds.err || { tail -5 $out/pmc_lds.err; exit 3; }
f=$(ls $out/pmc_lds/*/*counter_collection.csv $out/pmc_lds/*counter_collection.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'dw_bf16' in k or 'da_post' in k or 'conv_block_fwd' in k:
        acc[k.split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-24s mean %14.1f  n %d' % (c, sum(v) / len(v), len(v)))
PY
rm -rf $out/pmc_lds
