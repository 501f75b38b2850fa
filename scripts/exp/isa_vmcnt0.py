"""every `s_waitcnt vmcnt(0)` (and vmcnt(1)) of the kernels in hipcc -S listings, with the number of vector-memory loads / stores issued
before it: a full drain in the middle of a kernel waits for EVERY outstanding load AND store - a memory round trip nothing hides.
usage: isa_vmcnt0.py a.s b.s ... [kernel substring]"""
import re, sys
files = [f for f in sys.argv[1:] if f.endswith('.s')]
key = [a for a in sys.argv[1:] if not a.endswith('.s')]
key = key[0] if key else ''
for fn in files:
    lines = open(fn).read().split('\n')
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, start, nl, ns, sites, total = m.group(1), i, 0, 0, [], 0
            continue
        if name is None:
            continue
        t = l.strip()
        op = t.split()[0] if t else ''
        if re.match(r'(global|buffer|flat|scratch)_load', op):
            nl += 1
        if re.match(r'(global|buffer|flat|scratch)_(store|atomic)', op):
            ns += 1
        mm = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
        if mm and int(mm.group(1)) <= 1:
            sites.append((i - start, int(mm.group(1)), nl, ns))
        if l.startswith('.Lfunc_end'):
            if key in name and sites:
                d = re.match(r'_Z(\d+)', name)
                short = name[len(d.group(0)):len(d.group(0)) + int(d.group(1))] + name[len(d.group(0)) + int(d.group(1)):][:16]
                n = i - start
                mid = [s for s in sites if s[0] < n - 60]
                print('%-52s lines %5d  vmcnt(<=1) sites %2d (not at the end: %2d)  %s' % (short, n, len(sites), len(mid),
                      ' '.join('%d:v%d[l%d,s%d]' % s for s in mid[:10])))
            name = None
