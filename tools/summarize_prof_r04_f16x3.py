#!/usr/bin/env python3
"""Turns the outputs of tools/collect_profiles_r04_f16x3.sh (gpurun_out/f16x3) into the committed summaries of the f16x3 mode:

  profiles/r04_f16x3_steps_kernel_stats.csv   rocprofv3 --kernel-trace --stats of 3 + 6 optimisation steps (tools/prof_r03.py steps f16x3 6)
  profiles/r04_f16x3_bench_line.json          the bench line of `python bench.py` on the same box (headline bf16x6 + the f16x3_mode sibling block)
  profiles/r04_f16x3_profile.md               per-kernel table, PMC HBM bytes per step, SQ counters of the stand-alone fine-pass launches,
                                              G22 / fp64-distance lines of the GPU suite's log
(the functions are tools/summarize_prof_r04.py's, pointed at the other directory)"""
import json
import os
import re
import shutil

import summarize_prof_r04 as S

ROOT = S.ROOT
S.SRC = os.path.join(ROOT, 'gpurun_out', 'f16x3')
MODE = 'f16x3'
PEAK = 2500.0 / 3


def sq2():
    import collections
    import csv
    acc = collections.defaultdict(dict)
    for i in (1, 2):
        rows = list(csv.DictReader(open(os.path.join(S.SRC, 'sq_%s_%d' % (MODE, i), 'pmc_counter_collection.csv'))))
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        mlp = [r for r in rows if 'mlp_' in r['Kernel_Name']]
        ids = []
        for r in mlp:
            if r['Dispatch_Id'] not in ids:
                ids.append(r['Dispatch_Id'])
        tail = ids[-15:]
        label = {tail[0]: 'forward, no save', tail[1]: 'forward, saving', tail[2]: 'dX'}
        for r in mlp:
            if r['Dispatch_Id'] in label:
                acc[label[r['Dispatch_Id']]][r['Counter_Name']] = float(r['Counter_Value'])
                acc[label[r['Dispatch_Id']]]['_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 if 'End_Timestamp' in r else float('nan')
    return acc


def main():
    dst = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(S.SRC, 'steps_' + MODE, 'steps_kernel_stats.csv'), os.path.join(dst, 'r04_f16x3_steps_kernel_stats.csv'))
    j = json.loads(open(os.path.join(S.SRC, 'bench_line.json')).read())
    json.dump(j, open(os.path.join(dst, 'r04_f16x3_bench_line.json'), 'w'), indent=1)
    b = j['f16x3_mode']
    acc, span = S.step_times(MODE)
    st = S.step_traffic(MODE)
    ktot = sum(sum(v) for v in acc.values()) / 1e3 / S.N_TRACE
    md = ['# r04 -- the f16x3 mode on the GPU: step profile, HBM bytes, SQ counters, parity lines', '',
          'Collected by `tools/collect_profiles_r04_f16x3.sh` (one gpurun call), summarised by `tools/summarize_prof_r04_f16x3.py`.', '',
          '## bench.py on that box (`profiles/r04_f16x3_bench_line.json`)', '',
          'headline (bf16x6, `value`): **%.0f rays/s, %.2f ms / step**, roofline `frac` %.3f of %.1f.  `f16x3_mode` sibling, same protocol: **%.0f rays/s, %.2f ms / step**; '
          'saving forward (fine pass) %.3f ms = %.1f algorithmic TFLOP/s = **%.3f of %.1f** (2500 / 3 products), forward without saving %.3f ms = %.1f TFLOP/s = %.3f, '
          'backward (dX + 12 dW launches) %.2f ms.' % (
              j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['peak'], b['init_state']['value'], b['init_state']['ms_per_step'],
              b['roofline']['launches'][0]['avg_launch_ms'], b['roofline']['launches'][0]['achieved'], b['roofline']['launches'][0]['frac'], b['roofline']['peak'],
              b['roofline']['launches'][1]['avg_launch_ms'], b['roofline']['launches'][1]['achieved'], b['roofline']['launches'][1]['frac'],
              b['roofline']['launches'][2]['avg_launch_ms']), '',
          '## 6 optimisation steps under `rocprofv3 --kernel-trace --stats` (`python tools/prof_r03.py steps f16x3 6`)', '',
          'Kernel time **%.2f ms per step** (first kernel start to last kernel end: %.2f); HBM traffic **%.1f GB per step** (fetch %.1f + write %.1f; PMC FETCH_SIZE / WRITE_SIZE in '
          'separate runs, FETCH doubled per MI355X_MICROARCH.md; bf16x6: 51.2 GB).' % (
              ktot, span, st['hbm_bytes_per_step'] / 1e9, st['fetch_bytes_per_step'] / 1e9, st['write_bytes_per_step'] / 1e9), '',
          '| kernel | launches / step | ms / step | fine-pass launch us | algorithmic TFLOP/s | frac of its roofline |', '|---|---|---|---|---|---|']
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / S.N_TRACE < 20:
            continue
        big = sorted(v)[len(v) // 2:]
        us = sum(big) / len(big)
        tf = ' | '
        if name.startswith('mlp_fwd'):
            f = S.FWD_FLOP
            tf = '%.1f | %.3f of 833.3' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / PEAK)
        elif name.startswith('mlp_bwd_dx'):
            f = 786432 * 2 * 557696
            tf = '%.1f | %.3f of 833.3' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / PEAK)
        elif 'dw' in name and '2, 2, 4, true, false' in name:
            f = 786432 * 2 * 65536
            tf = '%.1f | %.3f of 416.7 (bf16x6 kernel)' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / (2500.0 / 6))
        md.append('| `%s` | %.1f | %.3f | %.1f | %s |' % (name[:60], len(v) / S.N_TRACE, sum(v) / 1e3 / S.N_TRACE, us, tf))
    s = sq2()
    cols = [c for c in ('forward, no save', 'forward, saving', 'dX') if c in s]
    counters = sorted({c for d in s.values() for c in d if not c.startswith('_')})
    md += ['', '## SQ counters, stand-alone fine-pass launches (786 432 points; `tools/prof_r03.py kernels f16x3 1`, two `--pmc` passes)', '',
           '| counter | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
    for c in counters:
        md.append('| %s | ' % c + ' | '.join('%.4g' % s[k].get(c, float('nan')) for k in cols) + ' |')
    share = {k: s[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * s[k]['SQ_WAVE_CYCLES'] / 2.0) for k in cols
             if 'SQ_VALU_MFMA_BUSY_CYCLES' in s[k] and 'SQ_WAVE_CYCLES' in s[k]}
    md.append('| **matrix pipe busy, share of SIMD time** | ' + ' | '.join(('**%.0f %%**' % (100 * share[k])) if k in share else '' for k in cols) + ' |')
    md.append('| non-MFMA VALU per MFMA | ' + ' | '.join('%.2f' % ((s[k]['SQ_INSTS_VALU'] - s[k]['SQ_INSTS_MFMA']) / s[k]['SQ_INSTS_MFMA']) if 'SQ_INSTS_VALU' in s[k] else '' for k in cols) + ' |')
    md += ['', '(bf16x6, `profiles/r04_sq_counters.md`: 3.42e8 MFMAs of 16 x 16 x 32 per forward launch, 68 / 65 / 65 % busy, 2.6 - 2.7 non-MFMA VALU per MFMA; '
           'f16x3 issues half the MFMAs.)', '']
    log = open(os.path.join(S.SRC, 'gputest.log')).read()
    md += ['## parity lines of the GPU suite on that box (`pytest tests -m gpu -q -s`)', '', '```']
    for line in log.split('\n'):
        line = line.lstrip('.')
        if line.startswith('G22 paired') or line.startswith('rms logit error') or line.startswith('f16x3 packed') or re.search(r'\d+ passed', line):
            md.append(line[:400])
    md += ['```', '']
    open(os.path.join(dst, 'r04_f16x3_profile.md'), 'w').write('\n'.join(md) + '\n')
    json.dump({'_how': 'as profiles/r04_pmc_traffic.json, math mode f16x3', 'f16x3': {
        'step_traffic': {k: st[k] for k in ('fetch_bytes_per_step', 'write_bytes_per_step', 'hbm_bytes_per_step')},
        'per_kernel_per_step': st['kernels'], 'matrix_pipe_busy': share}}, open(os.path.join(dst, 'r04_f16x3_pmc_traffic.json'), 'w'), indent=1)
    print('\n'.join(md))


if __name__ == '__main__':
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
