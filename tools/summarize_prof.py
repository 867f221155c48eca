"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_r1) into the committed summaries
under profiles/: per-mode kernel stats (csv + markdown) and HBM traffic per launch from the two PMC passes
(FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads; counters are in KB)."""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_r1')
DST = os.path.join(ROOT, 'profiles')


def short(name):
    return re.sub(r'\(.*', '', name)


def pmc(mode):
    out = {}
    for counter, key, scale in (('FETCH_SIZE', 'fetch_bytes', 2.0 * 1024), ('WRITE_SIZE', 'write_bytes', 1024.0)):
        f = os.path.join(SRC, 'pmc_%s_%s' % (mode, counter), 'pmc_counter_collection.csv')
        acc = {}
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            if 'mlp_' not in k and 'head_grads' not in k and 'reduce' not in k:
                continue
            acc.setdefault(k, []).append(float(r['Counter_Value']) * scale)
        for k, v in acc.items():
            # launches alternate fine-pass sizes only (prof_kernels.py): average per launch
            out.setdefault(k, {})[key] = sum(v) / len(v)
    for k, d in out.items():
        d['hbm_bytes'] = d.get('fetch_bytes', 0.0) + d.get('write_bytes', 0.0)
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    md = ['# Round 1 rocprofv3 summary (MI355X, 1 GPU)', '',
          'Collected by `tools/collect_profiles.sh` (run through gpurun), summarised by `tools/summarize_prof.py`.', '',
          'Command per math mode: `FASTNERF_MATH=<mode> rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py '
          '--steps 10 --warmup 3 --no-cpu-baseline` (13 optimisation steps of 4096 rays x (64+128) samples plus the 12 '
          'stand-alone fine-pass forward launches of the roofline leg).', '']
    traffic = {'_how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate pass, --pmc WRITE_SIZE) -- python '
                       'tools/prof_kernels.py 2; per launch on the fine pass (4096 rays x 192 samples = 786432 points); '
                       'counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide '
                       'coalesced reads)', 'kernels': {}}
    for mode in ('bf16x3', 'fp32'):
        f = os.path.join(SRC, 'bench_' + mode, 'bench_kernel_stats.csv')
        shutil.copy(f, os.path.join(DST, 'r01_bench_kernel_stats_%s.csv' % mode))
        shutil.copy(os.path.join(SRC, 'bench_' + mode, 'bench_domain_stats.csv'),
                    os.path.join(DST, 'r01_bench_domain_stats_%s.csv' % mode))
        rows = list(csv.DictReader(open(f)))
        line = [l for l in open(os.path.join(SRC, 'bench_%s.log' % mode)) if l.startswith('{"metric"')]
        md += ['## math mode `%s`' % mode, '']
        if line:
            j = json.loads(line[-1])
            md += ['bench line under the profiler: %.0f rays/s, %.2f ms/step; roofline leg `%s`: %.3f ms/launch, %.1f %s (frac %.3f)'
                   % (j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['avg_launch_ms'],
                      j['roofline']['achieved'], j['roofline']['unit'], j['roofline']['frac']), '']
        md += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
        for r in rows[:18]:
            md.append('| `%s` | %s | %.2f | %.1f | %s |' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                         float(r['AverageNs']) / 1e3, r['Percentage']))
        md.append('')
        t = pmc(mode)
        traffic['kernels'].update(t)
        md += ['HBM traffic per fine-pass launch (PMC):', '', '| kernel | fetch MB | write MB |', '|---|---|---|']
        for k, d in t.items():
            md.append('| `%s` | %.1f | %.1f |' % (k[:70], d.get('fetch_bytes', 0) / 1e6, d.get('write_bytes', 0) / 1e6))
        md.append('')
    json.dump(traffic, open(os.path.join(DST, 'r01_pmc_traffic.json'), 'w'), indent=1)
    open(os.path.join(DST, 'r01_summary.md'), 'w').write('\n'.join(md) + '\n')
    for old in ('r01_bench_kernel_stats.csv', 'r01_bench_domain_stats.csv'):
        p = os.path.join(DST, old)
        if os.path.exists(p):
            os.remove(p)


if __name__ == '__main__':
    main()
