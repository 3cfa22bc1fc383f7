"""Static instruction mix of the kernels in a device assembly listing (hipcc -S --cuda-device-only): VALU by kind,
LDS, vector memory, waits.  usage: isa_mix.py file.s substring [substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(r'^(_Z\w+):\s*;', re.M)
pos = [(m.start(), m.group(1)) for m in pat.finditer(txt)]
for i, (p0, name) in enumerate(pos):
    if not all(s in name for s in sys.argv[2:]):
        continue
    end = txt.find('.Lfunc_end', p0)
    body = txt[p0:end]
    c = collections.Counter()
    n = 0
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in '.;_' or l.endswith(':'):
            continue
        x = l.split()[0]
        n += 1
        if x.startswith('v_pk'):
            c['v_pk'] += 1
        elif re.match(r'v_(fma|mul_f|add_f|sub_f|fmac|mac|mad)', x):
            c['v_fp'] += 1
        elif x.startswith('v_'):
            c['v_other'] += 1
        elif x.startswith('ds_'):
            c['ds'] += 1
        elif x.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            c['vmem'] += 1
        elif x.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif x.startswith('s_barrier'):
            c['barrier'] += 1
        elif x.startswith('s_'):
            c['salu'] += 1
    print(name[:60], '...', name[-30:], n, dict(c))
