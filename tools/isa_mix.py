"""Instruction mix per kernel from the gfx950 assembly (metalbt709decoder_amd/build/*.s).
usage: python tools/isa_mix.py [file.s]"""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "metalbt709decoder_amd/build/bt709_kernels.s"
s = open(path).read()
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(parts), 2):
    name = parts[i]
    body = parts[i + 1].split('s_endpgm')[0]
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for k in ins:
        if k.startswith('v_'):
            c['valu'] += 1
        elif k.startswith(('ds_', 'global_', 'buffer_')):
            c[k] += 1
        elif k.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif k.startswith('s_'):
            c['salu'] += 1
    print(name[:70], len(ins), dict(c))
