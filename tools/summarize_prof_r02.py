#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles_r02.sh (gpurun_out/prof_r2) into the committed round-2 summaries:

  profiles/r02_bench_kernel_stats.csv            rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 ...`
  profiles/r02_steps_{compacted,plain}_kernel_stats.csv   ... of 10 steady-state optimisation steps (tools/prof_step.py)
  profiles/r02_pmc_traffic.json                  HBM bytes per launch of the MLP kernels and per optimisation step, from the PMC
                                                 passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only; counters
                                                 are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of
                                                 wide coalesced reads)
  profiles/r02_sq_counters.md, r02_summary.md    tables"""
import collections
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_r2')
DST = os.path.join(ROOT, 'profiles')
N_PMC_STEPS = 4


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '')


def read_counter(path, counter):
    """-> [(kernel, value)] in dispatch order, from the first pack_rays_kernel on (= the optimisation steps)."""
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    first = next(i for i, r in enumerate(rows) if 'pack_rays_kernel' in r['Kernel_Name'])
    return [(short(r['Kernel_Name']), float(r['Counter_Value'])) for r in rows[first:]]


def step_traffic(mode):
    out = {'kernels': collections.OrderedDict()}
    tot = {}
    for counter, key, scale in (('FETCH_SIZE', 'fetch_bytes', 2.0 * 1024), ('WRITE_SIZE', 'write_bytes', 1024.0)):
        rows = read_counter(os.path.join(SRC, 'pmc_c%d_%s' % (mode, counter), 'pmc_counter_collection.csv'), counter)
        acc = collections.OrderedDict()
        for k, v in rows:
            acc.setdefault(k, []).append(v * scale)
        for k, v in acc.items():
            d = out['kernels'].setdefault(k, {'launches_per_step': len(v) / N_PMC_STEPS})
            d[key + '_per_step'] = sum(v) / N_PMC_STEPS
            d[key + '_max_launch'] = max(v)
        tot[key] = sum(v * scale for _, v in rows) / N_PMC_STEPS
    out['fetch_bytes_per_step'], out['write_bytes_per_step'] = tot['fetch_bytes'], tot['write_bytes']
    out['hbm_bytes_per_step'] = tot['fetch_bytes'] + tot['write_bytes']
    return out


def stats_table(path, n=16):
    rows = list(csv.DictReader(open(path)))
    md = ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in rows[:n]:
        md.append('| `%s` | %s | %.2f | %.1f | %s |' % (short(r['Name'])[:64], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                        float(r['AverageNs']) / 1e3, r['Percentage']))
    return md, rows


def fine_launch_us(trace_path, kernel_sub, pick):
    """average duration of the fine-pass launches of a kernel in a step trace (the larger of the two launches per step)."""
    rows = [r for r in csv.DictReader(open(trace_path)) if kernel_sub in r['Kernel_Name']]
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows)
    half = d[len(d) // 2:] if pick == 'large' else d[:len(d) // 2]
    return sum(half) / max(1, len(half)), len(half)


def sq_tables():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for i in range(1, 6):
        p = os.path.join(SRC, 'sq%d' % i, 'pmc_counter_collection.csv')
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        # stand-alone launches of tools/prof_step.py kernels: the last 5 MLP-forward/backward groups; identify by order
        seq = [r for r in rows if 'mlp_' in r['Kernel_Name']]
        names = {}
        for r in seq:
            names.setdefault(r['Dispatch_Id'], short(r['Kernel_Name']))
        order = sorted(names, key=int)
        # the stand-alone block is the tail: fwd nosave, fwd save, bwd(plain: dx + 12 dw), fwd live, bwd live (dx + 12 dw)
        tail = order[-(1 + 1 + 13 + 1 + 13):]
        label = {}
        label[tail[0]] = 'forward, no save (786 432 points)'
        label[tail[1]] = 'forward, saving (786 432 points)'
        label[tail[2]] = 'dX (786 432 points)'
        label[tail[15]] = 'forward, saving, live list'
        label[tail[16]] = 'dX, live list'
        dw_plain = [d for d in tail[3:15] if '<4, 2, 2, 4, true, false>' in names[d]]
        dw_live = [d for d in tail[17:29] if '<4, 2, 2, 4, true, false>' in names[d]]
        if dw_plain:
            label[dw_plain[0]] = 'dW 256x256 job (786 432 points)'
        if dw_live:
            label[dw_live[0]] = 'dW 256x256 job, live list'
        for r in rows:
            if r['Dispatch_Id'] in label:
                acc[label[r['Dispatch_Id']]][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, 'bench', 'bench_kernel_stats.csv'), os.path.join(DST, 'r02_bench_kernel_stats.csv'))
    shutil.copy(os.path.join(SRC, 'steps_c1', 'steps_kernel_stats.csv'), os.path.join(DST, 'r02_steps_compacted_kernel_stats.csv'))
    shutil.copy(os.path.join(SRC, 'steps_c0', 'steps_kernel_stats.csv'), os.path.join(DST, 'r02_steps_plain_kernel_stats.csv'))
    comp, plain = step_traffic(1), step_traffic(0)
    traffic = {'_how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate run, --pmc WRITE_SIZE) -- python tools/prof_step.py '
                       'steps <state> 4 with FASTNERF_COMPACT=1 / 0: four optimisation steps (4096 rays x (64+128) samples) of nets '
                       'trained for 600 steps on the analytic solid-body scene of bench.py; counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md '
                       '(gfx950 reports 1/2 of wide coalesced reads); per-step = sum over every kernel of the step / 4',
               'kernels': {}, 'step_traffic': {
                   'compacted': {k: comp[k] for k in ('fetch_bytes_per_step', 'write_bytes_per_step', 'hbm_bytes_per_step')},
                   'plain': {k: plain[k] for k in ('fetch_bytes_per_step', 'write_bytes_per_step', 'hbm_bytes_per_step')},
                   'ratio_compacted_over_plain': comp['hbm_bytes_per_step'] / plain['hbm_bytes_per_step'],
                   'round1_plain_for_reference': 'about 42 GB per step (VERDICT r1, from profiles/r01_pmc_traffic.json)'},
               'per_kernel_per_step': {'compacted': comp['kernels'], 'plain': plain['kernels']}}
    # per-launch traffic of the fine-pass MLP launches (what bench.py's roofline.traffic reads): the larger launch per step
    for name, d in comp['kernels'].items():
        if 'mlp_fwd_bf16_kernel' in name or 'mlp_bwd_dx' in name:
            traffic['kernels']['void ' + name if not name.startswith('mlp_bwd') else name] = {
                'fetch_bytes': d.get('fetch_bytes_max_launch', 0.0), 'write_bytes': d.get('write_bytes_max_launch', 0.0),
                'hbm_bytes': d.get('fetch_bytes_max_launch', 0.0) + d.get('write_bytes_max_launch', 0.0),
                'note': 'largest launch of the step (fine pass) in the compacted step'}
    json.dump(traffic, open(os.path.join(DST, 'r02_pmc_traffic.json'), 'w'), indent=1)

    line = [l for l in open(os.path.join(SRC, 'bench.log')) if l.startswith('{"metric"')]
    j = json.loads(line[-1])
    md = ['# Round 2 rocprofv3 summary (MI355X, 1 GPU)', '',
          'Collected by `tools/collect_profiles_r02.sh` (through gpurun), summarised by `tools/summarize_prof_r02.py`.', '',
          '## bench.py under the profiler', '',
          '`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --sustained-steps 50 --no-cpu-baseline`', '',
          'bench line: **%.0f rays/s, %.2f ms/step** (nets trained on the analytic solid-body scene, stationary live fraction; compacted backward, live fraction '
          'fine %.3f / coarse %.3f); same state with the plain backward %.0f rays/s (%.2f ms); round-1 protocol (random init, noise '
          'targets) %.0f rays/s (%.2f ms); the blobs with their Gaussian tails after 300 steps %.0f rays/s (%.2f ms).' % (
              j['value'], j['ms_per_step'], j['live_fraction']['fine'], j['live_fraction']['coarse'],
              j['steady_state_plain']['value'], j['steady_state_plain']['ms_per_step'],
              j['init_state']['value'], j['init_state']['ms_per_step'],
              j['gaussian_tails_scene']['value'], j['gaussian_tails_scene']['ms_per_step']), '',
          'roofline leg (HIP events inside bench.py): `%s` %.3f ms/launch = %.1f TFLOP/s algorithmic = frac %.3f of 2500/3.' % (
              j['roofline']['kernel'], j['roofline']['avg_launch_ms'], j['roofline']['achieved'], j['roofline']['frac']), '']
    t, rows = stats_table(os.path.join(SRC, 'bench', 'bench_kernel_stats.csv'))
    md += t + ['']
    md += ['(The bench process runs every leg: 600 + 5 + 20 + 50 steps on the solid-body scene, 23 plain steps on the trained nets, 323 steps on the Gaussian-tail scene, 23 steps of '
           'the round-1 protocol, 23 steps of the exact-fp32 mode, the stand-alone roofline launches and the inference leg -- the table '
           'mixes them; the two step traces below are the clean per-step view.)', '']
    for tag, mode in (('compacted', 1), ('plain', 0)):
        md += ['## 10 steady-state optimisation steps, %s backward (`FASTNERF_COMPACT=%d python tools/prof_step.py steps <state> 10`)' % (tag, mode), '']
        t, rows = stats_table(os.path.join(SRC, 'steps_c%d' % mode, 'steps_kernel_stats.csv'), 22)
        ours = [r for r in rows if any(s in r['Name'] for s in ('mlp_', 'breduce', 'head_grads', 'bpack', 'cp_', 'raw2outputs', 'sample_',
                                                                 'mse_leafmax', 'adam_kernel', 'pack_rays'))]
        tot = sum(float(r['TotalDurationNs']) for r in ours) / 1e6 / 10
        md += t + ['', 'Sum of the step kernels: **%.2f ms per step** of kernel time.' % tot, '']
    tr = os.path.join(SRC, 'steps_c1', 'steps_kernel_trace.csv')
    a, n = fine_launch_us(tr, 'mlp_fwd_bf16_kernel<false, false>', 'large')
    ks = os.path.join(SRC, 'kernels', 'kernels_kernel_trace.csv')
    if os.path.exists(ks):
        # stand-alone launches of the same kernel doing ALL the work (tools/prof_step.py kernels: 5 reps after one forward of the
        # whole renderer, whose two launches come first)
        rows = [r for r in csv.DictReader(open(ks)) if 'mlp_fwd_bf16_kernel<false, false>' in r['Kernel_Name']]
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows][2:]
        md += ['Dominant kernel, stand-alone fine-pass launch with every tile doing all its work (`tools/prof_step.py kernels` under '
               '`rocprofv3 --kernel-trace`): `mlp_fwd_bf16_kernel<false, false>` **%.1f us** average over %d launches; the roofline leg '
               'of bench.py times the same launch with HIP events: **%.1f us** (ratio %.2f: five launches of a cold process under the '
               'profiler against the warmed-up bench process).' % (sum(d) / len(d), len(d), 1e3 * j['roofline']['avg_launch_ms'],
                                                                  sum(d) / len(d) / (1e3 * j['roofline']['avg_launch_ms'])), '']
    md += ['Inside the compacted step the fine-pass launch of that kernel takes %.1f us on average (%d launches): there, tiles '
           'without a live sample skip their colour branch (FN_FWD_SKIP_DEAD_RGB, 17 %% of a tile\'s MACs; exact: those samples have '
           'weight 0).' % (a, n), '']
    md += ['## HBM traffic per optimisation step (PMC)', '',
           '| backward | fetch GB | write GB | total GB / step |', '|---|---|---|---|']
    for tag, d in (('compacted', comp), ('plain', plain)):
        md.append('| %s | %.2f | %.2f | **%.2f** |' % (tag, d['fetch_bytes_per_step'] / 1e9, d['write_bytes_per_step'] / 1e9, d['hbm_bytes_per_step'] / 1e9))
    md += ['', 'Compacted / plain = %.3f.  Per kernel and step (compacted):' % traffic['step_traffic']['ratio_compacted_over_plain'], '',
           '| kernel | launches / step | fetch MB | write MB |', '|---|---|---|---|']
    for k, d in comp['kernels'].items():
        if d.get('fetch_bytes_per_step', 0) + d.get('write_bytes_per_step', 0) > 5e6:
            md.append('| `%s` | %.1f | %.1f | %.1f |' % (k[:64], d['launches_per_step'], d.get('fetch_bytes_per_step', 0) / 1e6, d.get('write_bytes_per_step', 0) / 1e6))
    open(os.path.join(DST, 'r02_summary.md'), 'w').write('\n'.join(md) + '\n')

    sq = sq_tables()
    cols = ['forward, no save (786 432 points)', 'forward, saving (786 432 points)', 'forward, saving, live list',
            'dX (786 432 points)', 'dX, live list', 'dW 256x256 job (786 432 points)', 'dW 256x256 job, live list']
    cols = [c for c in cols if c in sq]
    counters = sorted({c for d in sq.values() for c in d})
    md = ['# r02 -- SQ counters of the split-bf16 MLP kernels (stand-alone fine-pass launches on the trained fine net)', '',
          'Source: `tools/collect_profiles_r02.sh` part 4 (rocprofv3 `--pmc`, five passes of three SQ counters over '
          '`tools/prof_step.py kernels`; sums over the chip, one launch; ACTIVE / WAIT / WAVE counters are in quad-cycles).', '',
          '| counter | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
    for c in counters:
        md.append('| %s | ' % c + ' | '.join('%.4g' % sq[k].get(c, float('nan')) for k in cols) + ' |')

    def share(k):
        d = sq[k]
        wps = 2.0                                       # waves per SIMD: fwd / dX run two 4-wave workgroups per CU, dW one 8-wave workgroup
        return d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * d['SQ_WAVE_CYCLES'] / wps)
    md.append('| **matrix pipe busy, share of SIMD time** (MFMA_BUSY / (4 x WAVE_CYCLES / waves per SIMD)) | ' +
              ' | '.join('**%.0f %%**' % (100 * share(k)) for k in cols) + ' |')
    md.append('| other instructions issued per MFMA | ' + ' | '.join(
        '%.1f' % ((sq[k]['SQ_INSTS_VALU'] + sq[k].get('SQ_INSTS_LDS', 0) + sq[k].get('SQ_INSTS_VMEM_RD', 0) + sq[k].get('SQ_INSTS_VMEM_WR', 0)
                   - sq[k]['SQ_INSTS_MFMA']) / sq[k]['SQ_INSTS_MFMA']) for k in cols) + ' |')
    full = {'forward, saving, live list': 'forward, saving (786 432 points)', 'dX, live list': 'dX (786 432 points)',
            'dW 256x256 job, live list': 'dW 256x256 job (786 432 points)'}
    md.append('| share of the full launch (MFMA count: the live fraction of the trained state) | ' + ' | '.join(
        ('%.1f %%' % (100 * sq[k]['SQ_INSTS_MFMA'] / sq[full[k]]['SQ_INSTS_MFMA'])) if k in full and full[k] in sq else '100 %' for k in cols) + ' |')
    md.append('| cycles the pipe is held per MFMA | ' + ' | '.join('%.1f' % (sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / sq[k]['SQ_INSTS_MFMA']) for k in cols) + ' |')
    open(os.path.join(DST, 'r02_sq_counters.md'), 'w').write('\n'.join(md) + '\n')
    print(open(os.path.join(DST, 'r02_summary.md')).read()[:6000])
    print(open(os.path.join(DST, 'r02_sq_counters.md')).read())


if __name__ == '__main__':
    main()
