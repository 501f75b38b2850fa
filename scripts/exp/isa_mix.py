"""instruction mix of one kernel in a hipcc -S listing: isa_mix.py file.s kernel_substring"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
c = collections.Counter()
for l in lines[start:end]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '_')) or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
g = lambda f: sum(v for k, v in c.items() if f(k))
print('total', sum(c.values()), 'valu', g(lambda k: k.startswith('v_') and 'mfma' not in k), 'mfma', g(lambda k: 'mfma' in k),
      'ds', g(lambda k: k.startswith('ds_')), 'global', g(lambda k: k.startswith('global')), 'salu', g(lambda k: k.startswith('s_')))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print('%-28s %d' % (k, v))
