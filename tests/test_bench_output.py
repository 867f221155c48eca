"""bench.py's output contract on the CPU (no GPU, no kernels): the headline line a harness parses from a TAIL WINDOW of the combined
output must be short, complete and last; `--gpus N` without a launcher must re-execute under torch.distributed.run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B   # noqa: E402

CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _worst_case_record(world=8):
    long = 'x' * 900
    launches = [{'kernel': 'mlp_fwd_kernel<true, false, 1>' + long, 'what': long, 'points': 786432, 'avg_launch_ms': 4.646274566650391,
                 'flop_per_launch': 933350080512, 'achieved': 200.88138725406282, 'frac': 0.4821153294097507} for _ in range(3)]
    roof = {'bound': 'mfma', 'kernel': 'mlp_fwd_kernel<true, false, 1> (fine pass; ' + long + ')', 'achieved': 200.88138725406282,
            'peak': 416.6666666666667, 'unit': 'TFLOP/s', 'frac': 0.4821153294097507, 'peak_note': long, 'traffic': 9676262304.0,
            'traffic_unit': long, 'traffic_source': long, 'step_traffic': {'hbm_bytes_per_step': 51249650632.0},
            'avg_launch_ms': 4.646274566650391, 'flop_per_launch': 933350080512, 'launches': launches, 'frac_of_measured_peak': 0.6441947212850758}
    cpu = {'unit': 'rays/s', 'cores': 32, 'kind': 'port', 'cpu': 'AMD EPYC 9575F 64-Core Processor', 'cores_available': 256,
           'value': 326.8388771374388, 'rays_per_s_n1024': 326.8388771374388, 'rays_per_s_n4096': 338.87110767670015, 'more_threads': long,
           'sample': long}
    return {
        'metric': 'training rays/sec (Lego-like 800x800, 64+128 samples) + PSNR@N-iters', 'value': 220123.456789, 'unit': 'rays/s',
        'n_gpus': world, 'steps': 20, 'warmup': 5, 'ms_per_step': 18.612345678, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'math_mode': 'bf16x6: ' + long, 'data': long,
        'config': {'workload': long, 'rays_per_gpu_per_step': 4096, 'rays_per_step': 4096 * world, 'parallelism': 'dp%d' % world},
        'final_loss': [0.08322971314191818, 0.08326460421085358], 'backward': 'plain (every sample)',
        'per_rank_ms_per_step': [18.612345678] * world,
        'allreduce_ms': {'fine_half': 0.123456789, 'coarse_half': 0.123456789, 'whole_buffer': 0.23456789, 'overlapped_with_coarse_backward': True},
        'collective': 'torch.distributed/nccl', 'step_tflops_per_gpu': 196.6123, 'step_frac_of_peak': 0.4718,
        'sustained': {'ms_per_step': 18.7, 'value': 1.0, 'unit': 'rays/s', 'steps': 150},
        'roofline': roof, 'cpu_baseline': cpu,
        'psnr': {'iters': 200, 'rays_per_iter': 256, 'gpu_train_db': 26.730632431504313, 'gpu_held_out_db': 27.787402617614918,
                 'cpu_train_db': 26.00215655553397, 'cpu_held_out_db': 27.55402213826612, 'lockstep_delta_db': -7.281735352293595e-05,
                 'lockstep_max_rel_loss_diff': 0.0004934942203400715, 'paired_ensemble': B.PAIRED_PSNR_NOTE},
        'siblings_summary': {'ms_per_step': {'fp32': 30.713, 'bf16x3': 12.582, 'drop_in_bf16x6': 18.999},
                             'sparse_scene_bf16x6': {'ms_per_step': 9.123, 'rays_per_s': 449000, 'live_fine': 0.1821, 'plain_ms': 19.012},
                             'configs_rays_per_s': {'configs[2]': 230377, 'configs[3]': 289625, 'configs[4]': 102061},
                             'inference_rays_per_s': 777496,
                             'batch_ms': {'512': 2.539, '1024': 4.706, '1920': 8.640, '4096': 17.891},
                             'batch_frac_of_4096_rate': {'512': 0.881, '1024': 0.950, '1920': 0.971, '4096': 1.0}, 'predicted_strong_8': 6.78},
        'scaling_weak': None if world == 1 else {'rays_per_gpu_per_step': 4096, 'rays_per_step': 4096 * world, 'ms_per_step': 18.612345678,
                                                 'value': 1760987.654321, 'unit': 'rays/s', 'steps': 20, 'warmup': 5, 'note': long},
        'scaling_strong': None if world == 1 else {'rays_per_gpu_per_step': 4096 // world, 'rays_per_step': 4096, 'ms_per_step': 2.712345678,
                                                   'value': 1510123.456789, 'unit': 'rays/s', 'steps': 20, 'warmup': 5, 'note': long},
        'errors': ['configs[2]: RuntimeError: ' + long] * 6,
        'full_record': ['bench_full.json', 'gpurun_out/bench_full.json'],
        'other_configs': {'configs[2]_quadtree': {'workload': long * 8}}, 'psnr_vs_cpu': {'gpu': {'bf16x6': {}}, 'note': long * 4},
    }


def test_headline_is_short_and_complete():
    for world in (1, 8):
        out = _worst_case_record(world)
        h = B.headline_record(out)
        line = json.dumps(h)
        assert len(line) <= B.HEADLINE_MAX_BYTES, len(line)
        for k in CONTRACT_KEYS:
            assert k in h, k
        assert h['n_gpus'] == world and h['dtype'] == 'f32' and h['value'] == out['value']
        if world > 1:      # --scaling both (the default on several GPUs): both curves' points of this N ride in the ONE line, parseable
            back = json.loads(line)
            assert back['scaling'] == 'weak' and back['scaling_weak']['value'] == out['scaling_weak']['value']
            assert back['scaling_strong']['rays_per_gpu_per_step'] == 4096 // world and back['scaling_strong']['rays_per_step'] == 4096
            assert back['scaling_strong']['value'] == out['scaling_strong']['value'] and back['scaling_strong']['ms_per_step'] > 0
        else:
            assert 'scaling_weak' not in h and 'scaling_strong' not in h
        r = h['roofline']
        assert r['bound'] == 'mfma' and r['frac'] == out['roofline']['frac'] and r['achieved'] > 0 and r['peak'] > 0
        assert r['traffic'] == out['roofline']['traffic'] and r['avg_launch_ms'] > 0 and r['flop_per_launch'] > 0
        c = h['cpu_baseline']
        assert c['value'] == out['cpu_baseline']['value'] and c['cores'] == 32 and c['kind'] == 'port' and len(c['sample']) <= 200
        assert abs(h['vs_cpu_baseline'] - out['value'] / out['cpu_baseline']['value']) < 1e-9
        # the sibling blocks never ride in the line
        for k in ('other_configs', 'psnr_vs_cpu', 'fp32_mfma_mode', 'split_bf16_mode', 'drop_in_route', 'inference'):
            assert k not in h


def test_headline_survives_a_tail_window():
    """What the round-4 harness did: keep the last ~8 KB of stdout + '---- stderr ----' + stderr, parse the last JSON line of stdout."""
    h = B.headline_record(_worst_case_record())
    stdout = 'noise from a library\n' * 50 + json.dumps(h) + '\n'
    stderr = '/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n' * 8
    combined = stdout + '\n---- stderr ----\n' + stderr
    tail = combined[-8192:]
    lines = [l for l in tail.split('\n---- stderr ----')[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[-1])
    assert j['roofline']['frac'] > 0 and j['cpu_baseline']['value'] > 0 and j['config']['parallelism'] == 'dp8'


def test_headline_without_optional_blocks():
    out = _worst_case_record(2)
    out.update(roofline=None, cpu_baseline=None, psnr=None, siblings_summary=None, sustained=None, errors=[])
    h = B.headline_record(out)
    assert h['roofline'] is None and h['cpu_baseline'] is None and h['psnr'] is None and h['errors'] is None
    assert len(json.dumps(h)) < 2048


def test_bare_multi_gpu_invocation_relaunches_under_torchrun(monkeypatch):
    seen = {}

    def fake_execv(exe, cmd):
        seen['exe'], seen['cmd'] = exe, cmd
        raise SystemExit(0)
    monkeypatch.setattr(os, 'execv', fake_execv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1'])
    try:
        B.main()
    except SystemExit:
        pass
    cmd = seen['cmd']
    assert seen['exe'] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[k + 1:] == ['--gpus', '8', '--steps', '2', '--warmup', '1']


def test_write_full_record(tmp_path):
    out = _worst_case_record()
    p = str(tmp_path / 'full.json')
    B.write_full_record(out, p)
    assert json.load(open(p))['other_configs']['configs[2]_quadtree']['workload'].startswith('x')
