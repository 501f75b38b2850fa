"""VALU issue cycles of one kernel attributed to source lines: hipcc -S -gline-tables-only listing, kernel substring.
   isa_by_line.py file.s kernel_substring [loop_trip_count]
Quarter-rate integer multiplies count 16 cycles, transcendentals 8, everything else 4 (wave64 on a 16-lane SIMD).  The largest
backward branch of the kernel is taken as its main loop: instructions inside it are weighted by loop_trip_count (default 4: the layer
loop of the conv_block kernels), so the table estimates DYNAMIC issue cycles per wave."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
trips = int(sys.argv[3]) if len(sys.argv) > 3 else 4
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
labels = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m:
        labels[m.group(1)] = i
loop = (0, 0)
for i in range(start, end):
    m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', lines[i]) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > loop[1] - loop[0]:
        loop = (labels[m.group(1)], i)
cur = None
cnt, cyc = collections.Counter(), collections.Counter()
tot_in = tot_out = mfma_in = 0
for i in range(start, end):
    t = lines[i].strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    if not t or t.startswith(('.', ';', '_')) or t.endswith(':'):
        continue
    op = t.split()[0]
    inl = loop[0] <= i <= loop[1]
    if 'mfma' in op and inl:
        mfma_in += 1
    if op.startswith('v_') and 'mfma' not in op:
        c = 16 if ('mad_u64' in op or 'mul_lo' in op or 'mul_hi' in op) else (8 if op in ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32') else 4)
        w = trips if inl else 1
        cnt[cur] += w
        cyc[cur] += c * w
        if inl:
            tot_in += c
        else:
            tot_out += c
print('main loop: listing lines %d..%d; VALU issue cycles per trip %d (x%d), outside the loop %d; MFMA per trip %d' % (loop[0], loop[1], tot_in, trips, tot_out, mfma_in))
print('dynamic VALU issue cycles per wave ~ %d' % (tot_in * trips + tot_out))
byfile = collections.Counter()
for (f, ln), c in cyc.items():
    byfile[f] += c
print('by file', byfile.most_common())
for k, v in sorted(cyc.items(), key=lambda x: -x[1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print('%-18s %5d  instr %4d  cycles %5d' % (k[0], k[1], cnt[k], v))
