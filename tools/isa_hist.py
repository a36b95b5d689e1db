#!/usr/bin/env python3
"""Opcode histogram of one kernel in a hipcc -S file. usage: isa_hist.py file.s mangled_substring [topN]"""
import collections, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
import re
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):\s*(;.*)?$', txt, re.M)
if not m: raise SystemExit('kernel not found')
start = m.start()
end = txt.index('s_endpgm', start)
ops = collections.Counter()
for l in txt[start:end].splitlines():
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'): continue
    ops[l.split()[0]] += 1
valu = sum(v for k, v in ops.items() if k.startswith('v_'))
print('total', sum(ops.values()), 'valu', valu, 'pk', sum(v for k,v in ops.items() if k.startswith('v_pk')), 'mov', ops['v_mov_b32_e32']+ops['v_pk_mov_b32']+ops['v_mov_b64_e32'], 'nop', ops['s_nop'], 'lds', sum(v for k,v in ops.items() if k.startswith('ds_')))
print(' '.join('%s:%d' % kv for kv in ops.most_common(top)))
