"""quarter-rate integer multiplies (v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32: 16 issue cycles per wave64 against 4 for a plain VALU
instruction) of every kernel in hipcc -S -gline-tables-only listings, by source line: isa_quarter_rate.py a_g.s b_g.s ..."""
import re, sys, collections
for fn in sys.argv[1:]:
    lines = open(fn).read().split('\n')
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    kern, cur = None, None
    per = collections.defaultdict(collections.Counter)
    valu = collections.Counter()
    for l in lines:
        if l.startswith('_Z') and l.rstrip().split(';')[0].strip().endswith(':'):
            kern = l.split(':')[0]
            continue
        t = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
        if m:
            cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            continue
        if kern is None or not t or t.startswith(('.', ';', '_')):
            continue
        op = t.split()[0]
        if op.startswith('v_') and 'mfma' not in op:
            valu[kern] += 1
        if op in ('v_mad_u64_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_i64_i32', 'v_mul_hi_i32'):
            per[kern][cur] += 1
    for k in per:
        n = sum(per[k].values())
        m = re.match(r'_Z(\d+)', k)
        name = k[len(m.group(0)):len(m.group(0)) + int(m.group(1))] + k[len(m.group(0)) + int(m.group(1)):][:24] if m else k
        print('%-60s VALU %5d  quarter-rate %4d (= %d plain instructions)' % (name[:60], valu[k], n, 4 * n))
        for (f, ln), c in per[k].most_common(8):
            print('      %-16s %5d  x%d' % (f, ln, c))
