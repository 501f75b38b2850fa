"""Static scan of a kernel's ISA for exposed memory round trips: a vector load whose result is waited for (s_waitcnt vmcnt(<=1)) within a
few instructions of its issue - nothing is in flight under it.  usage: python scripts/exp/isa_exposed_waits.py <file.hip> <kernel substring>
Prints every such site with the instructions between the load and the wait (the consumer follows the wait)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, key = sys.argv[1], sys.argv[2]
out = '/tmp/isa_scan.s'
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + ROOT + '/include', '-I' + ROOT + '/hual_amd/csrc', '-S',
                       '--cuda-device-only', os.path.join(ROOT, 'hual_amd/csrc', src), '-o', out], stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
name, start = None, 0
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name, start, last = m.group(1), i, None
        continue
    if name is None or key not in name:
        continue
    t = l.strip()
    if re.match(r'(global|buffer|flat)_load', t) and 'lds' not in t.split()[0]:
        last = i
    mm = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
    if mm and last is not None and i - last <= 25 and int(mm.group(1)) <= 1:
        n = i - start
        print('--- %s  line %d' % (name[:50], n))
        for k in range(last, min(i + 4, len(lines))):
            print('   ', lines[k].strip()[:110])
        last = None
    if 's_endpgm' in t:
        name = None
