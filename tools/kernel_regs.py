#!/usr/bin/env python3
"""Register / spill / scratch summary of every kernel in hipcc -S output.  usage: kernel_regs.py file.s [filter]"""
import re
import sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    name, body = m.group(1), m.group(2)
    if not name.startswith('_Z') or flt not in name:
        continue
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [0, '?'])[1]
    short = re.sub(r'^_Z\d+', '', name)
    short = re.sub(r'(EEv|l[PK]|v[lP]).*$', '', short)
    print('%-60s vgpr %3s agpr %3s spill %3s sgpr_spill %3s scratch %4s' % (short[:60], g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
