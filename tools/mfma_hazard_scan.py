#!/usr/bin/env python3
"""Static scan of gfx950 assembly (hipcc -S) for MFMA -> VALU register hazards.

For every v_mfma_* it walks the fall-through successors and reports VALU / DS / VMEM instructions that read or
write a register of the MFMA's destination block within `need` wait states (an instruction = 1 state, s_nop N =
N + 1).  The CDNA3/4 ISA requires software wait states between an XDL op's VGPR write and a dependent VALU
read / overwrite (8-pass ops: 11; LLVM's GCNHazardRecognizer, GFX940 tables); hipcc is supposed to insert them.
Used to look for the cause of the SLP-vectoriser corruption described in DESIGN.md section 9.

usage: mfma_hazard_scan.py file.s [need=11]"""
import re
import sys

RNG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs(tok):
    out = set()
    for m in RNG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def main():
    path = sys.argv[1]
    need = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    lines = [l.rstrip('\n') for l in open(path)]
    ins = []   # (lineno, mnemonic, operand string)
    kern = None
    for no, l in enumerate(lines, 1):
        s = l.split(';')[0].strip()
        if not s:
            continue
        if s.endswith(':') and not s.startswith('.'):
            kern = s[:-1]
        if s.startswith('.') or s.endswith(':'):
            ins.append((no, 'label', s, kern))
            continue
        parts = s.split(None, 1)
        ins.append((no, parts[0], parts[1] if len(parts) > 1 else '', kern))
    bad = 0
    for i, (no, mn, ops, kern) in enumerate(ins):
        if not mn.startswith('v_mfma'):
            continue
        dst = regs(ops.split(',')[0])
        states = 0
        j = i + 1
        while j < len(ins) and states < need:
            no2, mn2, ops2, _ = ins[j]
            j += 1
            if mn2 == 'label':
                continue
            if mn2.startswith('s_branch') or mn2 == 's_endpgm' or mn2.startswith('s_setpc'):
                break
            if mn2.startswith('v_mfma'):
                # back-to-back dependent MFMAs have their own (hardware interlocked) rules
                states += 1
                continue
            if mn2 == 's_nop':
                states += int(ops2.strip() or 0) + 1
                continue
            touched = regs(ops2)
            if (mn2.startswith('v_') or mn2.startswith('ds_') or mn2.startswith('buffer_') or mn2.startswith('global_')) and touched & dst:
                print(f'{path}:{no2}: {mn2} {ops2}   <- {states} states after line {no} {mn} {ops.split(",")[0]}   [{kern}]')
                bad += 1
                break
            states += 1
    print(f'{path}: {bad} suspicious MFMA->VALU pairs (need {need})')


if __name__ == '__main__':
    main()
