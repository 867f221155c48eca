#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh <round> (gpurun_out/prof_<round>) into the committed summaries of that round (python tools/summarize_prof.py r06):

  profiles/<round>_bench_kernel_stats.csv              rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-iters 0`
  profiles/<round>_steps_<mode>_kernel_stats.csv       ... of 3 + 6 optimisation steps of the headline protocol in math mode <mode> (tools/prof_r03.py)
  profiles/<round>_pmc_traffic.json                    HBM bytes per launch of the MLP kernels and per optimisation step, per math mode, from the
                                                   PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only; counters are in KB;
                                                   FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads)
  profiles/<round>_sq_counters.md, profiles/<round>_summary.md   tables"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else 'r06'
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_' + RND)
DST = os.path.join(ROOT, 'profiles')
MODES = tuple(os.environ.get('MODES', 'bf16x6 fp32 bf16x3').split())
N_PMC, N_TRACE, WARM = 4, 6, 3
PEAK = {'bf16x6': 2500.0 / 6, 'fp32': 157.3, 'bf16x3': 2500.0 / 3}
FWD_FLOP = 786432 * 2 * 593408


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '')


def timed_rows(rows):
    """rows of the TIMED steps: from the (WARM + 1)-th pack_rays_kernel on."""
    idx = [i for i, r in enumerate(rows) if 'pack_rays_kernel' in r['Kernel_Name']]
    return rows[idx[WARM]:]


def step_traffic(mode):
    out = {'kernels': collections.OrderedDict()}
    tot = {}
    for counter, key, scale in (('FETCH_SIZE', 'fetch_bytes', 2.0 * 1024), ('WRITE_SIZE', 'write_bytes', 1024.0)):
        rows = [r for r in csv.DictReader(open(os.path.join(SRC, 'pmc_%s_%s' % (mode, counter), 'pmc_counter_collection.csv')))
                if r['Counter_Name'] == counter]
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        rows = timed_rows(rows)
        acc = collections.OrderedDict()
        for r in rows:
            acc.setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']) * scale)
        for k, v in acc.items():
            d = out['kernels'].setdefault(k, {'launches_per_step': len(v) / N_PMC})
            d[key + '_per_step'] = sum(v) / N_PMC
            d[key + '_max_launch'] = max(v)
        tot[key] = sum(sum(v) for v in acc.values()) / N_PMC
    out['fetch_bytes_per_step'], out['write_bytes_per_step'] = tot['fetch_bytes'], tot['write_bytes']
    out['hbm_bytes_per_step'] = tot['fetch_bytes'] + tot['write_bytes']
    return out


def step_times(mode):
    rows = list(csv.DictReader(open(os.path.join(SRC, 'steps_' + mode, 'steps_kernel_trace.csv'))))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    rows = timed_rows(rows)
    acc = collections.OrderedDict()
    for r in rows:
        acc.setdefault(short(r['Kernel_Name']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6 / N_TRACE
    return acc, span


def sq(mode):
    acc = collections.defaultdict(dict)
    for i in (1, 2, 3):
        rows = list(csv.DictReader(open(os.path.join(SRC, 'sq_%s_%d' % (mode, i), 'pmc_counter_collection.csv'))))
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        mlp = [r for r in rows if 'mlp_' in r['Kernel_Name']]
        ids = []
        for r in mlp:
            if r['Dispatch_Id'] not in ids:
                ids.append(r['Dispatch_Id'])
        # the stand-alone block is the tail: forward without saving, saving forward, dX, then the dW launches (bf16x6 since r06: the pe job,
        # the trunk launch of the eight 256 x 256 jobs, the view job; the other modes: twelve launches)
        name_of = {r['Dispatch_Id']: r['Kernel_Name'] for r in mlp}
        dx = [d for d in ids if 'mlp_bwd_dx' in name_of[d]][-1]
        k = ids.index(dx)
        tail = ids[k - 2:]
        label = {tail[0]: 'forward, no save', tail[1]: 'forward, saving', tail[2]: 'dX'}
        trunk = [d for d in tail[3:] if 'trunk' in name_of[d]]
        big = [d for d in tail[3:] if '<4, 2, 2, 4, true, false' in name_of[d]]
        if trunk:
            label[trunk[0]] = 'dW trunk launch (8 jobs of 256x256)'
        elif big:
            label[big[0]] = 'dW 256x256 job'
        for r in mlp:
            if r['Dispatch_Id'] in label:
                acc[label[r['Dispatch_Id']]][r['Counter_Name']] = float(r['Counter_Value'])
    return acc


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, 'bench', 'bench_kernel_stats.csv'), os.path.join(DST, RND + '_bench_kernel_stats.csv'))
    line = [l for l in open(os.path.join(SRC, 'bench.log')) if l.startswith('{"metric"')]
    j = json.loads(line[-1])
    json.dump(j, open(os.path.join(DST, RND + '_bench_line_under_rocprof.json'), 'w'), indent=1)
    traffic = {'_how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate run, --pmc WRITE_SIZE) -- python tools/prof_r03.py steps <mode> 4: '
                       'four optimisation steps (after three warm-up steps) of the SURVEY 8(d) throughput protocol (4096 rays x (64+128) samples, random-init '
                       'nets, U[0,1) targets, plain backward); counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports '
                       '1/2 of wide coalesced reads); per-step = sum over every kernel of the step / 4; kernels[...] = the LARGEST launch of that '
                       'kernel in a step = its fine pass (786 432 points)'}
    md = ['# Round %s rocprofv3 summary (MI355X, 1 GPU)' % RND[1:].lstrip('0'), '',
          'Collected by `tools/collect_profiles.sh %s` (through gpurun), summarised by `tools/summarize_prof.py %s`.' % (RND, RND), '',
          '## bench.py under the profiler', '',
          '`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-iters 0`', '',
          'bench line of that run (`profiles/%s_bench_line_under_rocprof.json`)' % RND + ': **%.0f rays/s, %.2f ms/step** '
          'in the headline mode (%s); sibling legs of the same process, ms per step: %s; bf16x6 on the trained sparse scene, compacted backward: %s.' % (
              j['value'], j['ms_per_step'], j['config']['math_mode'], json.dumps((j.get('siblings') or {}).get('ms_per_step')),
              json.dumps((j.get('siblings') or {}).get('sparse_scene_bf16x6'))), '',
          'roofline leg (HIP events inside bench.py): `%s` %.3f ms/launch = %.1f TFLOP/s algorithmic = frac %.3f of %.1f.' % (
              j['roofline']['kernel'], j['roofline']['avg_launch_ms'], j['roofline']['achieved'], j['roofline']['frac'], j['roofline']['peak']), '']
    rows = list(csv.DictReader(open(os.path.join(SRC, 'bench', 'bench_kernel_stats.csv'))))
    md += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in rows[:14]:
        md.append('| `%s` | %s | %.2f | %.1f | %s |' % (short(r['Name'])[:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
    md += ['', '(The bench process runs every leg -- headline, sustained, the other two modes, the sparse scene, the drop-in route, the PSNR runs, '
           'inference -- so this table mixes them; the per-mode step traces below are the clean view.)', '']
    sqmd = ['# ' + RND + ' -- SQ counters of the MLP kernels, stand-alone fine-pass launches (786 432 points), per math mode', '',
            'Source: `tools/collect_profiles.sh` part 4 (rocprofv3 `--pmc`, three passes of three SQ counters over `tools/prof_r03.py kernels <mode> 1`; '
            'sums over the chip, one launch; WAVE / BUSY counters are in quad-cycles).  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves per SIMD), '
            'two waves per SIMD in every kernel here.', '']
    for mode in MODES:
        shutil.copy(os.path.join(SRC, 'steps_' + mode, 'steps_kernel_stats.csv'), os.path.join(DST, RND + '_steps_%s_kernel_stats.csv' % mode))
        st = step_traffic(mode)
        acc, span = step_times(mode)
        traffic[mode] = {'kernels': {}, 'step_traffic': {k: st[k] for k in ('fetch_bytes_per_step', 'write_bytes_per_step', 'hbm_bytes_per_step')},
                         'per_kernel_per_step': st['kernels']}
        for name, d in st['kernels'].items():
            if name.startswith('mlp_fwd') or name.startswith('mlp_bwd_dx'):
                traffic[mode]['kernels'][name] = {'fetch_bytes': d.get('fetch_bytes_max_launch', 0.0), 'write_bytes': d.get('write_bytes_max_launch', 0.0),
                                                  'hbm_bytes': d.get('fetch_bytes_max_launch', 0.0) + d.get('write_bytes_max_launch', 0.0)}
        ktot = sum(sum(v) for v in acc.values()) / 1e3 / N_TRACE
        md += ['## math mode `%s`: %d optimisation steps of the headline protocol (`python tools/prof_r03.py steps %s %d`)' % (mode, N_TRACE, mode, N_TRACE), '',
               'Kernel time **%.2f ms per step** (first kernel start to last kernel end: %.2f ms per step); HBM traffic **%.1f GB per step** '
               '(fetch %.1f + write %.1f, PMC).' % (ktot, span, st['hbm_bytes_per_step'] / 1e9, st['fetch_bytes_per_step'] / 1e9, st['write_bytes_per_step'] / 1e9), '',
               '| kernel | launches / step | ms / step | largest launch us | algorithmic TFLOP/s of it | frac of %.1f |' % PEAK[mode], '|---|---|---|---|---|---|']
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            if sum(v) / N_TRACE < 20:
                continue
            big = sorted(v)[len(v) // 2:]   # the larger half = the fine-pass launches
            us = sum(big) / len(big)
            tf = ''
            if name.startswith('mlp_fwd'):
                tf = '%.1f | %.3f' % (FWD_FLOP / (us * 1e-6) / 1e12, FWD_FLOP / (us * 1e-6) / 1e12 / PEAK[mode])
            elif name.startswith('mlp_bwd_dx'):
                f = 786432 * 2 * 557696
                tf = '%.1f | %.3f' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / PEAK[mode])
            elif 'dw' in name and '2, 2, 4, true, false' in name:
                f = 786432 * 2 * 65536
                tf = '%.1f | %.3f' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / PEAK[mode])
            elif 'dw6_trunk' in name:      # eight 256 x 256 jobs in one launch
                f = 8 * 786432 * 2 * 65536
                tf = '%.1f | %.3f' % (f / (us * 1e-6) / 1e12, f / (us * 1e-6) / 1e12 / PEAK[mode])
            else:
                tf = ' | '
            md.append('| `%s` | %.1f | %.3f | %.1f | %s |' % (name[:60], len(v) / N_TRACE, sum(v) / 1e3 / N_TRACE, us, tf))
        md.append('')
        s = sq(mode)
        cols = [c for c in ('forward, no save', 'forward, saving', 'dX', 'dW 256x256 job', 'dW trunk launch (8 jobs of 256x256)') if c in s]
        counters = sorted({c for d in s.values() for c in d})
        sqmd += ['## `%s`' % mode, '', '| counter | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
        for c in counters:
            sqmd.append('| %s | ' % c + ' | '.join('%.4g' % s[k].get(c, float('nan')) for k in cols) + ' |')
        share = {k: s[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * s[k]['SQ_WAVE_CYCLES'] / 2.0) for k in cols
                 if 'SQ_VALU_MFMA_BUSY_CYCLES' in s[k] and 'SQ_WAVE_CYCLES' in s[k]}
        sqmd.append('| **matrix pipe busy, share of SIMD time** | ' + ' | '.join(('**%.0f %%**' % (100 * share[k])) if k in share else '' for k in cols) + ' |')
        sqmd.append('')
        traffic[mode]['matrix_pipe_busy'] = share
    json.dump(traffic, open(os.path.join(DST, RND + '_pmc_traffic.json'), 'w'), indent=1)
    open(os.path.join(DST, RND + '_summary.md'), 'w').write('\n'.join(md) + '\n')
    open(os.path.join(DST, RND + '_sq_counters.md'), 'w').write('\n'.join(sqmd) + '\n')
    print('\n'.join(md[-60:]))
    print('\n'.join(sqmd))


if __name__ == '__main__':
    main()
