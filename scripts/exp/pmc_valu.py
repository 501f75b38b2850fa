"""per kernel: mean counters per dispatch and the derived VALU figures (scripts/exp/pmc_valu.sh)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print('# B64 T128 L20 vdim1024 drop0.2, eager steps; per dispatch means.  valu_cyc/simd = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs (the counter')
print('# ticks once per 4 cycles per wave on this part if it tracks SQ_BUSY_CYCLES granularity - compare with gui cycles for the share)')
print('%-44s %6s %9s %11s %11s %9s %9s %9s %8s' % ('kernel', 'n', 'gui_cyc', 'act_valu', 'insts_valu', 'valu/wave', 'salu/wave', 'lds/wave', 'share'))
out = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if 'SQ_WAVES' not in m or m['SQ_WAVES'] == 0:
        continue
    gui = m.get('GRBM_GUI_ACTIVE', 0.0)
    av = m.get('SQ_ACTIVE_INST_VALU', 0.0)
    share = av / (gui * 1024.0) if gui else 0.0      # 1024 SIMDs
    out.append((gui, '%-44s %6d %9.0f %11.0f %11.0f %9.0f %9.0f %9.0f %8.3f' % (k[:44], len(c['SQ_WAVES']), gui, av, m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_VALU', 0) / m['SQ_WAVES'],
                m.get('SQ_INSTS_SALU', 0) / m['SQ_WAVES'], m.get('SQ_INSTS_LDS', 0) / m['SQ_WAVES'], share)))
for g, l in sorted(out, reverse=True):
    print(l)
