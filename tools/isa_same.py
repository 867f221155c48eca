"""Compare two `hipcc -S --cuda-device-only` listings instruction for instruction (comments, .file/.ident and the per-source
__hip_cuid_* symbol ignored).  usage: python tools/isa_same.py a.s b.s   -> exit 0 when the device code is the same."""
import difflib
import re
import sys


def norm(path):
    out = []
    for l in open(path):
        t = l.strip()
        if not t or t.startswith((';', '.file', '.ident')):
            continue
        t = re.sub(r'__hip_cuid_\w+', '__hip_cuid', t.split(';')[0].rstrip())
        out.append(t)
    return out


a, b = norm(sys.argv[1]), norm(sys.argv[2])
if a == b:
    print('same device code: %d lines' % len(a))
    sys.exit(0)
d = list(difflib.unified_diff(a, b, lineterm='', n=0))
print('DIFFERENT: %d vs %d lines, %d diff lines' % (len(a), len(b), len(d)))
print('\n'.join(d[:60]))
sys.exit(1)
