"""Resolve a fixed set of preprocessor switches in a source file and delete the dead branches (a minimal `unifdef` + constant
substitution; round 5's pruning of csrc/mlp.hip / mlp_bf16.hip).

usage: python tools/prune_macros.py FILE NAME=VALUE [NAME=VALUE ...] [-U NAME ...]     (rewrites FILE in place)
  NAME=VALUE   the switch is fixed at VALUE: `#ifndef NAME / #define NAME ... / #endif` blocks and bare `#define NAME ...` lines are
               removed, conditionals that depend ONLY on fixed / undefined names are resolved, and every use of NAME in code is
               replaced by VALUE
  -U NAME      the switch is never defined: `#ifdef NAME` branches are deleted, `#ifndef NAME` ones kept
Conditionals that mention any other identifier are left alone (their bodies are still processed).  Check the result with an ISA diff of
the default build (`hipcc -S --cuda-device-only`): the pruned file must compile to the same instructions."""
import re
import sys


def main():
    path = sys.argv[1]
    fixed, undef = {}, set()
    args = sys.argv[2:]
    i = 0
    while i < len(args):
        if args[i] == '-U':
            undef.add(args[i + 1]); i += 2
        else:
            k, v = args[i].split('='); fixed[k] = v; i += 1
    known = set(fixed) | undef
    ident = re.compile(r'[A-Za-z_]\w*')

    def evaluate(expr):
        """-> True / False, or None when the expression mentions an identifier we do not control."""
        expr = expr.split('//')[0].strip()
        e = re.sub(r'defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)',
                   lambda m: ('@1' if (m.group(1) or m.group(2)) in fixed else '@0') if (m.group(1) or m.group(2)) in known else 'UNKNOWN_IDENT',
                   expr)
        for name in ident.findall(e):
            if name == 'UNKNOWN_IDENT' or (name not in known):
                return None
        e = ident.sub(lambda m: fixed.get(m.group(0), '0'), e).replace('@', '')
        e = e.replace('&&', ' and ').replace('||', ' or ').replace('!=', '@NE@').replace('!', ' not ').replace('@NE@', '!=')
        return bool(eval(e, {'__builtins__': {}}, {}))

    lines = open(path).read().split('\n')
    out = []
    # stack entries: [kind, emitting_parent, state]; kind 'resolved': state = 'taken' | 'seeking' | 'done'; kind 'kept': passthrough
    stack = []

    def emitting():
        return all(s[3] for s in stack)

    n = 0
    while n < len(lines):
        line = lines[n]
        s = line.strip()
        m = re.match(r'#\s*(ifdef|ifndef|if|elif|else|endif|define)\b(.*)', s)
        if not m:
            if emitting():
                if fixed and not s.startswith('#'):
                    code, sep, comment = line.partition('//')
                    code = re.sub(r'\b(' + '|'.join(map(re.escape, fixed)) + r')\b', lambda mm: fixed[mm.group(1)], code)
                    line = code + sep + comment
                out.append(line)
            n += 1
            continue
        d, rest = m.group(1), m.group(2).strip()
        if d == 'define':
            name = ident.match(rest).group(0)
            if name in known:
                while line.rstrip().endswith('\\'):      # a multi-line definition of a pruned macro
                    n += 1; line = lines[n]
                n += 1
                continue
            if emitting():
                out.append(line)
            n += 1
            continue
        if d in ('ifdef', 'ifndef', 'if'):
            if d == 'if':
                val = evaluate(rest)
            else:
                name = ident.match(rest).group(0)
                val = None if name not in known else ((name in fixed) == (d == 'ifdef'))
                # `#ifndef NAME / #define NAME v / #endif` default-definition block of a fixed switch: drop it whole
                if d == 'ifndef' and name in fixed:
                    val = False
            if val is None:
                stack.append(['kept', None, None, emitting()])
                if emitting():
                    out.append(line)
                stack[-1][3] = True if emitting() or True else False
                stack[-1][3] = all(s_[3] for s_ in stack[:-1])
            else:
                parent = emitting()
                stack.append(['resolved', parent, 'taken' if val else 'seeking', parent and val])
        elif d == 'elif':
            top = stack[-1]
            if top[0] == 'kept':
                if emitting():
                    out.append(line)
            else:
                if top[2] == 'taken':
                    top[2], top[3] = 'done', False
                elif top[2] == 'seeking':
                    val = evaluate(rest)
                    if val is None:
                        raise SystemExit('%s:%d: #elif with foreign identifiers after a resolved #if is not supported' % (path, n + 1))
                    if val:
                        top[2], top[3] = 'taken', top[1]
        elif d == 'else':
            top = stack[-1]
            if top[0] == 'kept':
                if emitting():
                    out.append(line)
            else:
                if top[2] == 'taken':
                    top[2], top[3] = 'done', False
                elif top[2] == 'seeking':
                    top[2], top[3] = 'taken', top[1]
        elif d == 'endif':
            top = stack.pop()
            if top[0] == 'kept' and emitting():
                out.append(line)
        n += 1
    assert not stack, 'unbalanced conditionals'
    open(path, 'w').write('\n'.join(out))


if __name__ == '__main__':
    main()
