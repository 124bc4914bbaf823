#!/usr/bin/env python3
"""Instruction mix per kernel from a `hipcc -S --cuda-device-only` listing."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "ntt_"
for m in re.finditer(r'^(_ZN3pha\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not re.search(pat, name):
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        c[line.split()[0]] += 1
    mul = sum(v for k, v in c.items() if k.startswith(('v_mad_u64', 'v_mul_lo', 'v_mul_hi')))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(name[:70], 'total', sum(c.values()), 'valu', valu, 'mul', mul, 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
          'vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))), 'salu', sum(v for k, v in c.items() if k.startswith('s_')))
    print('   ', c.most_common(16))
