"""Per-basic-block instruction-class histogram of one kernel in a hipcc -S listing.
usage: python tools/isa_hist.py file.s kernel_substring [min_instrs]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'acc_mov'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
blocks, cur, name = [], collections.Counter(), 'entry'
ops = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\w+):', l)
    if m:
        blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1); continue
    t = l.strip()
    if not t or t.startswith((';', '.')): continue
    op = t.split()[0]
    cur[cls(op)] += 1
    ops[name][op] += 1
    if op.startswith(('s_cbranch', 's_branch')): cur['->' + t.split()[-1]] += 1
blocks.append((name, cur))
for n, c in blocks:
    tot = sum(v for k, v in c.items() if not k.startswith('->'))
    if tot < mn: continue
    print(n, tot, dict(c))
    if '-v' in sys.argv:
        print('    ', ops[n].most_common(25))
