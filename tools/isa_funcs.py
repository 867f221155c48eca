"""Per-kernel comparison of two `hipcc -S --cuda-device-only` listings: kernels are matched by mangled name (optionally after a regex
substitution on the names of listing A), and compared instruction for instruction.
usage: python tools/isa_funcs.py a.s b.s [name_regex replacement]"""
import re
import sys


def funcs(p):
    d, cur = {}, None
    for l in open(p):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1); d[cur] = []; continue
        if l.startswith('.Lfunc_end'):
            cur = None
        if cur:
            t = l.strip()
            if t and not t.startswith((';', '.')):
                d[cur].append(re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'_Z\w+', 'SYM', t.split(';')[0].rstrip())))
    return d


a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
if len(sys.argv) > 4:
    a = {re.sub(sys.argv[3], sys.argv[4], k): v for k, v in a.items()}
same = diff = 0
for k in sorted(set(a) | set(b)):
    if k not in a or k not in b:
        print('ONLY IN', 'A' if k in a else 'B', k[:110]); continue
    if a[k] == b[k]:
        same += 1
    else:
        diff += 1
        nm = lambda L: sum(t.startswith('v_mfma') for t in L)   # noqa: E731
        print('DIFF %-100s %6d -> %6d instructions, %d -> %d MFMAs' % (k[:100], len(a[k]), len(b[k]), nm(a[k]), nm(b[k])))
print('%d kernels identical, %d differ' % (same, diff))
sys.exit(1 if diff else 0)
