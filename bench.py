#!/usr/bin/env python3
"""bench.py -- training rays/s (+ PSNR at equal iterations) of the MI355X-native NeRF inner loop (BASELINE.json metric).

A "step" = one full optimisation step of the reference's loop (run_nerf.py:479-508): ray packing -> stratified + hierarchical
sampling -> PE -> coarse / fine MLP -> compositing -> 2 x MSE + per-(image, leaf) error table (atomicMax) -> backward ->
[RCCL all-reduce] -> Adam + LR decay, on BASELINE.json configs[1] ("nerf-ours Lego full 800x800, 4096 rays, 64+128 samples").

`value` follows SURVEY 8(d) to the letter: 100 pose_spherical(-180 + 3.6 k, -30, 4) cameras, 800 x 800, focal 1111.11, near 2 /
far 6; every batch = 4096 rays drawn uniformly from all 64 M pixels, targets U[0,1)^3 (values do not affect timing), nets at
their default initialisation (seed 0), perturb = 1, white background, leaf tags of a depth-5 quadtree; fp32-WIDTH arithmetic
and the PLAIN backward (every sample goes through loss.backward(); FASTNERF_COMPACT is forced to 0 for this leg).  Inputs are
resident in HBM before the timed region.  The arithmetic is the `bf16x6` mode of csrc/mlp_*.hip: every fp32 operand decomposed
EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits), every product the sum of its six piece products of weight >= 2^-16
with fp32 accumulation on v_mfma_f32_16x16x32_bf16 -- the dropped terms are <= 2^-24 of the product, fp32's own rounding; measured
against fp64 its logits / head-layer gradients are as close as the fp32-MFMA kernels' (tests/test_gpu_mlp.py::
test_bf16x6_decomposition_is_exact_and_products_have_fp32_width, profiles/r03_precision_vs_fp64.md).

Everything else rides in the same JSON line as named sibling blocks and never feeds `value`:
  fp32_mfma_mode     the same protocol on v_mfma_f32_32x32x2_f32 (an fp32 FMA chain bit for bit; 157 TFLOP/s ceiling);
  split_bf16_mode    the same protocol in the split-bf16 mode (3 bf16 MFMA products per fp32 product, 16-bit operands: faster and
                     NARROWER than fp32, hence not the headline), plus that mode on a trained sparse scene with the exact
                     zero-gradient compaction (what a converged Lego-like field looks like to the backward);
  drop_in_route      INTEGRATION option A, the reference's loop verbatim (render(); loss.backward(); torch.optim.Adam.step());
  psnr_vs_cpu        the metric's second half: PSNR after N iterations on the analytic scene, GPU (all three modes) vs the CPU
                     oracle on identical batches / injected randoms (SURVEY 8d "PSNR runs"): free runs + the lockstep replay;
  other_configs      BASELINE configs[2] (quadtree train() end to end), [3] (LLFF / NDC, 64+64, sigma noise), [4] (nerf++ cascade,
                     1920 rays) at the headline arithmetic on ONE GPU, each with the roofline of its dominant launch (timed live) and a
                     short CPU-oracle baseline at the same shape;
  inference          render()-style rays/s;   cpu_baseline   the CPU oracle timed on this box's host cores.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
  N>1: either under the launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) or BARE
  (python bench.py --gpus N ...): without WORLD_SIZE in the environment the script re-executes itself under that launcher.

OUTPUT CONTRACT.  Rank 0 prints exactly ONE line on stdout, LAST: the headline record (headline_record(): <= 4 KB -- metric, value, unit,
n_gpus, steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline, psnr, a few scalars of the siblings).  The full record with
every sibling block is written to bench_full.json (--full-out; also copied under gpurun_out/ when that directory exists) BEFORE that
line, never to stdout / stderr (a harness that keeps only a tail window of the combined output must still find the headline).
The GPU legs import nothing from oracle/; only the CPU legs (cpu_baseline, psnr_vs_cpu.cpu) do.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS, N_SAMPLES, N_IMPORTANCE = 4096, 64, 128
SCENE_CUTOFF = 1.5      # sparse-scene sibling leg: density exactly zero beyond 1.5 standard deviations of a blob's centre
S1 = N_SAMPLES + N_IMPORTANCE
MAC_PER_POINT = 593408
FWD_FLOP_PER_POINT = 2 * MAC_PER_POINT     # 1.186816 MFLOP
BWD_FLOP_PER_POINT = 2 * (2 * MAC_PER_POINT - 35712)   # dX (without the input-side blocks) + dW
TRAIN_FLOP_PER_RAY = 2 * (3 * MAC_PER_POINT - 35712) * (N_SAMPLES + S1)  # 893.2 MFLOP
FP32_MFMA_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0             # MI355X_MICROARCH.md: dense bf16 matrix peak
PROFILE_ROUND = 'r06'
MAIN_MODE = 'bf16x6'                        # the headline's arithmetic: fp32-width products on the bf16 matrix cores (docstring)
MODE_PEAK = {'fp32': (FP32_MFMA_PEAK_TFLOPS, 1.0, 'dense fp32 MFMA (v_mfma_f32_32x32x2_f32)', 'mlp_fwd_kernel', ', 0'),
             'bf16x6': (BF16_MFMA_PEAK_TFLOPS / 6.0, 6.0, 'dense bf16 MFMA 2500 TFLOP/s / 6 piece products per fp32 product', 'mlp_fwd_kernel', ', 1'),
             'bf16x3': (BF16_MFMA_PEAK_TFLOPS / 3.0, 3.0, 'dense bf16 MFMA 2500 TFLOP/s / 3 split terms per product', 'mlp_fwd_bf16_kernel', '')}
H = W = 800
FOCAL = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
PSNR_ITERS, PSNR_RAYS, PSNR_HELD_OUT = 200, 256, 2048   # (256 rays per iteration: the paired protocol's batch, oracle/psnr_protocol.py)
MAC_PER_POINT_BG = MAC_PER_POINT + 2 * 21 * 256   # nerf++ background MLPNet: 84 instead of 63 input channels into layers 0 and 5
# measured ceiling of a bare v_mfma_f32_32x32x16_bf16 stream on uniform(-1, 1) data (tools/micro/mfma_power.hip, gap_probe.hip:
# 1848 - 1871 TFLOP/s issued at the 1.79 GHz the power management grants it) -- a REPO CONSTANT from earlier runs, not measured here
BF16_MFMA_REAL_DATA_TFLOPS = 1871.0
# the same kind of ceiling for the instruction the bf16x6 forward / dX actually issue since round 4, v_mfma_f32_16x16x32_bf16 with two waves per
# SIMD on random bf16 data: 2244 TFLOP/s issued against 1991 - 2002 for the 32x32x16 shape in the same call (tools/micro/mfma_shapes.hip,
# profiles/r04_power_limit.md section 5) -- also a REPO CONSTANT.  frac_of_measured_peak keeps the 32x32x16 figure (the dW kernels' shape and the
# comparison of rounds 3 - 5); frac_of_measured_peak_16x16x32 is the stricter one for the dominant kernel
BF16_MFMA_16X16X32_RANDOM_DATA_TFLOPS = 2244.0
PAIRED_PSNR_NOTE = 'tests/test_gpu_train.py::test_psnr_paired_study_g22 (profiles/r05_psnr_paired.md), null: profiles/r06_psnr_null.md'


HEADLINE_MAX_BYTES = 4096


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + '...'


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d}


def headline_record(out):
    """The ONE line of stdout: the contract's keys + roofline + cpu_baseline + psnr + one scalar per sibling, cut from the full
    record `out` (pure function of it; tests/test_bench_output.py holds it to HEADLINE_MAX_BYTES on a worst-case record)."""
    roof, cpu, cfg = out.get('roofline'), out.get('cpu_baseline'), out['config']
    h = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                             'vs_baseline', 'dtype')}
    h['data'] = _short(out['data'], 160)
    h['config'] = {'workload': _short(cfg['workload'], 330), 'rays_per_gpu_per_step': cfg['rays_per_gpu_per_step'],
                   'rays_per_step': cfg['rays_per_step'], 'parallelism': cfg['parallelism'], 'math_mode': out['math_mode'].split(':')[0],
                   'backward': out['backward']}
    if roof is not None:
        r = _pick(roof, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'flop_per_launch'))
        r['kernel'] = _short(roof['kernel'], 120)
        r['peak_note'] = _short(roof.get('peak_note', ''), 80)
        r['traffic_source'] = _short(roof.get('traffic_source', ''), 110)
        if roof.get('step_traffic'):
            r['step_hbm_bytes'] = roof['step_traffic'].get('hbm_bytes_per_step')
        for k in ('frac_of_measured_peak', 'frac_of_measured_peak_16x16x32'):
            if k in roof:
                r[k] = roof[k]
        r['launches_ms'] = {_short(x['kernel'], 40): round(x['avg_launch_ms'], 4) for x in roof.get('launches', [])[:3]}
        h['roofline'] = r
    else:
        h['roofline'] = None
    if cpu is not None:
        c = _pick(cpu, ('value', 'unit', 'cores', 'kind', 'cpu', 'cores_available', 'rays_per_s_n4096'))
        c['sample'] = _short(cpu.get('sample', ''), 200)
        h['cpu_baseline'] = c
        h['vs_cpu_baseline'] = out['value'] / cpu['value'] if cpu.get('value') else None
    else:
        h['cpu_baseline'] = None
    h['psnr'] = out.get('psnr')
    h.update({k: out.get(k) for k in ('final_loss', 'per_rank_ms_per_step', 'allreduce_ms', 'collective', 'step_tflops_per_gpu',
                                      'step_frac_of_peak')})
    if out.get('sustained'):
        h['sustained_ms_per_step'] = out['sustained']['ms_per_step']
    for k in ('scaling_weak', 'scaling_strong'):      # --scaling both: the two curves' points of this N in the one line
        if out.get(k):
            h[k] = _pick(out[k], ('rays_per_gpu_per_step', 'rays_per_step', 'ms_per_step', 'value'))
    h['siblings'] = out.get('siblings_summary')
    h['errors'] = [_short(e, 160) for e in out.get('errors', [])][:4] or None
    h['full_record'] = out.get('full_record')
    line = json.dumps(h)
    if len(line) > HEADLINE_MAX_BYTES:      # never let an over-long optional field break the contract: drop optional blocks, largest first
        for k in ('siblings', 'allreduce_ms', 'per_rank_ms_per_step', 'errors', 'psnr'):
            h[k] = None
            line = json.dumps(h)
            if len(line) <= HEADLINE_MAX_BYTES:
                break
    return h


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run (one rank per GPU, 127.0.0.1
    rendezvous, a free port) -- the command the contract names; stdout / stderr are inherited, so rank 0's single line is this
    process's single line."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def write_full_record(out, path):
    """bench_full.json (every sibling block), plus a copy under gpurun_out/ so that it comes back from a GPU box."""
    written = []
    for pth in [path] + ([os.path.join(ROOT, 'gpurun_out', os.path.basename(path))] if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else []):
        try:
            with open(pth, 'w') as f:
                json.dump(out, f, indent=1)
            written.append(os.path.relpath(pth, ROOT))
        except OSError:
            pass
    return written


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return avail, max(1, min(avail, 32))   # torch's CPU GEMMs peak at 32 threads on the bench boxes (tools/cpu_threads_probe.py)


def cpu_baseline(protocol='full'):
    """The CPU oracle (a port of the reference's step, validated against it by tests/) timed on the host cores of this
    box, SURVEY 8(d): the identical step (64+128 samples, fp32) at N = 1024 rays, 3 warm-up + 10 timed steps, median, with 32
    threads (where torch's CPU GEMMs peak on this box: 64 / 128 / 256 threads measured 0.47x / 0.24x / 0.003x in round 2,
    tools/cpu_threads_probe.py -- the one deviation from 8(d)'s "all cores", which would time a slower baseline); one step at
    N = 4096 as a confirmation.  About 55 s.  `short` = 1 + 3 steps of 256 rays."""
    from oracle import nerf_oracle as O
    avail, t32 = cpu_threads()
    gen = torch.Generator().manual_seed(0)
    sdc, sdf = O.init_nerf_params(gen), O.init_nerf_params(gen)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    K = O.intrinsics(800, 800, 1111.111)
    ro, rd = O.get_rays(800, 800, K, c2w)

    def run(n, threads, warm, timed):
        torch.set_num_threads(threads)
        sel = torch.randint(0, 640000, (n,), generator=gen)
        rb = O.make_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], 2.0, 6.0)
        tgt = torch.rand(n, 3, generator=gen)
        times = []
        for it in range(warm + timed):
            t_rand, u = torch.rand(n, N_SAMPLES, generator=gen), torch.rand(n, N_IMPORTANCE, generator=gen)
            t0 = time.time()
            O.train_step(sdc, sdf, opt, rb, tgt, N_SAMPLES, N_IMPORTANCE, True, t_rand=t_rand, u=u)
            times.append(time.time() - t0)
        return n / float(np.median(times[warm:]))

    base = {'unit': 'rays/s', 'cores': t32, 'kind': 'port', 'cpu': cpu_model(), 'cores_available': avail}
    if protocol == 'short':
        base.update(value=run(256, t32, 1, 3),
                    sample=f'1 warm-up + 3 timed steps of 256 rays x (64+128) samples, median, torch CPU fp32, {t32} threads '
                           '(oracle/nerf_oracle.py train_step)')
        return base
    v1024 = run(1024, t32, 3, 10)
    v4096 = run(4096, t32, 0, 1)
    base.update(value=v1024, rays_per_s_n1024=v1024, rays_per_s_n4096=v4096,
                more_threads='measured slower in round 2 on this CPU model: 64 threads 0.47x, 128 threads 0.24x, all 256 hardware '
                             'threads ~1 ray/s (tools/cpu_threads_probe.py); not repeated per run',
                sample=f'SURVEY 8(d): N=1024 rays x (64+128) samples per step, 3 warm-up + 10 timed steps, median, torch CPU fp32 '
                       f'with {t32} threads; one step at N=4096 (oracle/nerf_oracle.py train_step)')
    return base


def cpu_baseline_llff():
    """configs[3] shape on the host cores: NDC rays of a 1008 x 756 forward-facing camera, 64 + 64 samples, raw_noise_std = 1
    (oracle/nerf_oracle.py train_step), 1 warm-up + 3 timed steps of 1024 rays, median."""
    from oracle import nerf_oracle as O
    _, t32 = cpu_threads()
    torch.set_num_threads(t32)
    gen = torch.Generator().manual_seed(0)
    sdc, sdf = O.init_nerf_params(gen), O.init_nerf_params(gen)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    Hl, Wl, fl = 756, 1008, 815.13
    ro, rd = O.get_rays(Hl, Wl, O.intrinsics(Hl, Wl, fl), torch.eye(4)[:3, :4])
    n = 1024
    sel = torch.randint(0, Hl * Wl, (n,), generator=gen)
    rb = O.make_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], 0.0, 1.0, Hl, Wl, fl, ndc=True)
    tgt = torch.rand(n, 3, generator=gen)
    times = []
    for it in range(4):
        t_rand, u = torch.rand(n, 64, generator=gen), torch.rand(n, 64, generator=gen)
        n0, n1 = torch.randn(n, 64, generator=gen), torch.randn(n, 128, generator=gen)
        t0 = time.time()
        O.train_step(sdc, sdf, opt, rb, tgt, 64, 64, False, t_rand=t_rand, u=u, noise0=n0, noise1=n1)
        times.append(time.time() - t0)
    return {'value': n / float(np.median(times[1:])), 'unit': 'rays/s', 'cores': t32, 'kind': 'port',
            'sample': '1 warm-up + 3 timed steps of 1024 NDC rays x (64+64) samples with sigma noise, median (oracle/nerf_oracle.py)'}


def cpu_baseline_nerfpp():
    """configs[4] shape on the host cores: one two-level cascade batch (64 / 128 samples: 512 MLP evaluations per ray through the fg
    and the 4-D inverted-sphere bg net) + Adam per level (oracle/nerfpp_oracle.py cascade_step), 1 warm-up + 3 timed batches of
    240 rays, median."""
    from oracle import nerf_oracle as O
    from oracle import nerfpp_oracle as OP
    _, t32 = cpu_threads()
    torch.set_num_threads(t32)
    gen = torch.Generator().manual_seed(0)

    def init(input_ch):
        sd = {}
        for name, shp in OP.mlpnet_param_shapes(input_ch):
            fan_in = shp[1] if len(shp) == 2 else 256
            sd[name] = (torch.rand(shp, generator=gen) * 2 - 1) / fan_in ** 0.5
        return sd
    levels = [(init(63), init(84)) for _ in range(2)]
    opts = [O.Adam(list(fg.values()) + list(bg.values()), lr=5e-4) for fg, bg in levels]
    n = 240
    ro = (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
    rd = torch.randn(n, 3, generator=gen)
    tgt = torch.rand(n, 3, generator=gen)
    times = []
    for it in range(4):
        rand = [{'fg_t': torch.rand(n, 64, generator=gen), 'bg_t': torch.rand(n, 64, generator=gen)},
                {'fg_u': torch.rand(n, 128, generator=gen), 'bg_u': torch.rand(n, 128, generator=gen)}]
        t0 = time.time()
        outs = OP.cascade_step(levels, ro, rd, tgt, [64, 128], rand)
        for m, o in enumerate(outs):
            opts[m].step(o[1])
        times.append(time.time() - t0)
    return {'value': n / float(np.median(times[1:])), 'unit': 'rays/s', 'cores': t32, 'kind': 'port',
            'sample': '1 warm-up + 3 timed cascade batches of 240 rays (2 levels x (fg + bg), 64 / 128 samples) + Adam, median '
                      '(oracle/nerfpp_oracle.py)'}


# ---------------------------------------------------------------------------------------------------------------------
# PSNR at equal iterations, CPU side (the only other place that touches oracle/): a worker process, so that it runs beside
# the GPU legs
# ---------------------------------------------------------------------------------------------------------------------
def psnr_of(losses, last):
    return float(-10.0 * np.log10(np.mean(np.asarray(losses[-last:], dtype=np.float64))))


def psnr_cpu_worker(in_path, out_path):
    """Free run of the CPU oracle on the PSNR protocol's inputs; records its state (weights, Adam moments) BEFORE every step
    in <out_path>.states (float32 [iters + 1, 3, n_params]) for the GPU side's lockstep replay."""
    from oracle import nerf_oracle as O
    d = torch.load(in_path)
    _, t32 = cpu_threads()
    try:   # this run happens BESIDE the GPU legs: keep it off the cores the launching process uses (upper half of the CPU list, low priority)
        cpus = sorted(os.sched_getaffinity(0))
        if len(cpus) >= 4 * t32:
            os.sched_setaffinity(0, cpus[len(cpus) // 2:])
        os.nice(10)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(t32)
    sdc, sdf = d['sdc'], d['sdf']
    params = list(sdc.values()) + list(sdf.values())
    opt = O.Adam(params, lr=5e-4)
    n_par = sum(p.numel() for p in params)
    states = np.lib.format.open_memmap(out_path + '.states', mode='w+', dtype=np.float32, shape=(d['iters'] + 1, 3, n_par))

    def snapshot(i):
        for j, ts in enumerate((params, opt.m, opt.v)):
            states[i, j] = torch.cat([t.detach().reshape(-1) for t in ts]).numpy()
    losses, times = [], []
    for it in range(d['iters']):
        snapshot(it)
        opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4   # pre-increment rule (run_nerf.py:498-508)
        rb = O.make_ray_batch(d['ro'][it], d['rd'][it], 2.0, 6.0)
        t0 = time.time()
        l1, l0, _, _ = O.train_step(sdc, sdf, opt, rb, d['tgt'][it], N_SAMPLES, N_IMPORTANCE, True, t_rand=d['t_rand'][it], u=d['u'][it])
        times.append(time.time() - t0)
        losses.append(float(l1))
    snapshot(d['iters'])
    states.flush()
    with torch.no_grad():   # held-out rays, perturb = 0 (render_kwargs_test)
        rb = O.make_ray_batch(d['ho_ro'], d['ho_rd'], 2.0, 6.0)
        mse = 0.0
        for s in range(0, rb.shape[0], 512):
            ret = O.render_rays(rb[s:s + 512], sdc, sdf, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)
            mse += float(((ret['rgb_map'] - d['ho_tgt'][s:s + 512]) ** 2).sum())
        mse /= rb.shape[0] * 3
    json.dump({'losses': losses, 'held_out_mse': mse, 'median_step_s': float(np.median(times)), 'threads': t32,
               'rays_per_s': d['ro'].shape[1] / float(np.median(times))}, open(out_path, 'w'))


def psnr_inputs(fastnerf, dev, iters, args, poses, K, draw_pixels):
    """SURVEY 8(d) "PSNR runs": per iteration PSNR_RAYS uniformly drawn rays of the 100 cameras with the analytic scene's
    colours as targets, jitter streams t_rand / u from seed 2, default-init nets (seed 0) -- one set, handed to both sides."""
    from fastnerf import ops, synthetic
    g = torch.Generator().manual_seed(2)
    ros, rds, tgts = [], [], []
    for it in range(iters):
        ro, rd = ops.gen_rays_pixels(draw_pixels(g, PSNR_RAYS).to(dev), poses, K)
        ros.append(ro); rds.append(rd); tgts.append(synthetic.render_rays(ro, rd, cutoff=0.0))
    ho_ro, ho_rd = ops.gen_rays_pixels(draw_pixels(torch.Generator().manual_seed(7), PSNR_HELD_OUT).to(dev), poses, K)
    torch.manual_seed(0)
    k0 = fastnerf.run_nerf.create_nerf(args, device=dev)[0]
    return {'iters': iters, 'ro': torch.stack(ros).cpu(), 'rd': torch.stack(rds).cpu(), 'tgt': torch.stack(tgts).cpu(),
            't_rand': torch.rand(iters, PSNR_RAYS, N_SAMPLES, generator=g), 'u': torch.rand(iters, PSNR_RAYS, N_IMPORTANCE, generator=g),
            'ho_ro': ho_ro.cpu(), 'ho_rd': ho_rd.cpu(), 'ho_tgt': synthetic.render_rays(ho_ro, ho_rd, cutoff=0.0).cpu(),
            'sdc': {k: v.detach().cpu().clone() for k, v in k0['network_fn'].state_dict().items()},
            'sdf': {k: v.detach().cpu().clone() for k, v in k0['network_fine'].state_dict().items()}}


def psnr_gpu_free(fastnerf, dd, new_trainer, K, mode, jitter_ulp_seed=None):
    """One free GPU run of the protocol in `mode`.  jitter_ulp_seed: perturb every initial weight by a random -1 / 0 / +1 ulp
    (the ensemble that measures how far apart two runs of the SAME arithmetic class end up: trajectories are chaotic)."""
    from fastnerf import ops
    ops.set_math(mode)
    fastnerf.render.set_compact('0')
    tr, _, kt, _ = new_trainer()
    if jitter_ulp_seed is not None:
        g = torch.Generator(device=tr.flat.device).manual_seed(jitter_ulp_seed)
        with torch.no_grad():
            bits = tr.flat.view(torch.int32)
            bits += torch.randint(-1, 2, bits.shape, generator=g, device=bits.device, dtype=torch.int32)
        tr.repack()
    ls = []
    for it in range(dd['iters']):
        ls.append(tr.step(dd['ro'][it], dd['rd'][it], dd['tgt'][it], t_rand=dd['t_rand'][it], u=dd['u'][it])[0][0])
    ls = torch.stack(ls).cpu().numpy().tolist()
    with torch.no_grad():
        rgb = fastnerf.render.render(H, W, K, chunk=PSNR_HELD_OUT, rays=(dd['ho_ro'], dd['ho_rd']), near=2.0, far=6.0, **kt)[0]
        mse = float(torch.mean((rgb - dd['ho_tgt']) ** 2))
    return {'train_psnr_db': psnr_of(ls, 20), 'held_out_psnr_db': -10.0 * math.log10(mse), 'first_loss': ls[0], 'last_loss': ls[-1]}, ls


def psnr_gpu_lockstep(fastnerf, dd, new_trainer, states, cpu_losses, mode):
    """The GPU step from the CPU oracle's state, iteration by iteration (weights and Adam moments taken over before every step):
    the same PSNR number without the trajectory divergence -- what is left is the per-step arithmetic difference."""
    from fastnerf import ops
    ops.set_math(mode)
    fastnerf.render.set_compact('0')
    tr, _, _, _ = new_trainer()
    dev = tr.flat.device
    ls, upd = [], []
    for it in range(dd['iters']):
        st = torch.from_numpy(np.ascontiguousarray(states[it])).to(dev)
        with torch.no_grad():
            tr.flat.copy_(st[0]); tr.m.copy_(st[1]); tr.v.copy_(st[2])
        tr.adam_t = it
        tr.lr = 5e-4 * (0.1 ** ((it - 1) / (500 * 1000))) if it > 0 else 5e-4
        tr.repack()
        ls.append(tr.step(dd['ro'][it], dd['rd'][it], dd['tgt'][it], t_rand=dd['t_rand'][it], u=dd['u'][it], decay=False)[0][0])
        if it % 10 == 5:   # the update itself against the CPU's (Adam's g / (|g| + 1e-8) amplifies noise on ~zero gradients: L2, not max)
            nxt = torch.from_numpy(np.ascontiguousarray(states[it + 1][0])).to(dev)
            upd.append(float((tr.flat - nxt).norm() / (nxt - st[0]).norm()))
    ls = torch.stack(ls).cpu().numpy()
    cl = np.asarray(cpu_losses)
    return {'train_psnr_db': psnr_of(ls.tolist(), 20), 'max_rel_loss_diff': float(np.max(np.abs(ls - cl) / cl)),
            'median_rel_loss_diff': float(np.median(np.abs(ls - cl) / cl)),
            'max_rel_update_diff_l2': max(upd) if upd else None}


def time_launch(fn_, reps):
    """Average duration (ms) of reps back-to-back calls of fn_ on torch's current stream (= the stream the kernels are
    launched on: ops.* pass torch.cuda.current_stream() through the C ABI), HIP events."""
    for _ in range(2):
        fn_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn_()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scaling', choices=['weak', 'strong', 'both', 'auto'], default='auto',
                    help='weak: 4096 rays per GPU per step (`value`); strong: 4096 rays per step split over the GPUs (`value`); both: `value` '
                         'is the weak number and the line also carries scaling_weak / scaling_strong blocks (one invocation, both curves); '
                         'auto (default): weak on one GPU, both on several')
    ap.add_argument('--sustained-steps', type=int, default=150)
    ap.add_argument('--scene-steps', type=int, default=600, help='untimed optimisation steps of the sparse-scene sibling leg')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip both CPU legs (cpu_baseline, psnr_vs_cpu.cpu)')
    ap.add_argument('--no-siblings', action='store_true', help='headline + roofline only')
    ap.add_argument('--cpu-protocol', choices=['full', 'short'], default='full')
    ap.add_argument('--psnr-iters', type=int, default=PSNR_ITERS)
    ap.add_argument('--psnr-cpu-worker', nargs=2, metavar=('IN', 'OUT'), help=argparse.SUPPRESS)
    ap.add_argument('--full-out', default=os.path.join(ROOT, 'bench_full.json'), help='where the full record (all sibling blocks) goes')
    a = ap.parse_args()
    if a.psnr_cpu_worker:
        return psnr_cpu_worker(*a.psnr_cpu_worker)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(a.gpus, sys.argv[1:])
    warnings.filterwarnings('ignore')     # stderr shares the harness's tail window with the headline line: keep it empty

    import fastnerf
    from fastnerf import ops, parallel, synthetic
    from fastnerf.synthetic import pose_spherical
    rank, world, local = parallel.init_from_env('cuda')
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    local = local % max(1, torch.cuda.device_count())   # (only differs in single-GPU plumbing tests of the N>1 path)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if a.scaling == 'auto':
        a.scaling = 'both' if world > 1 else 'weak'
    both = a.scaling == 'both'
    # weak is the contract's default and the reference's own practice: nerf++-ours/ddp_train_nerf.py:201-202 quotes its rates at a FIXED
    # batch per GPU (1920 rays on 2 GPUs, 2880 on 3)
    n_local = N_RAYS // world if a.scaling == 'strong' else N_RAYS      # rays per rank per step (of the leg that is `value`)
    n_step = n_local * world                                          # rays per step, whole job
    siblings = rank == 0 and world == 1 and not a.no_siblings
    psnr_leg = rank == 0 and world == 1 and a.psnr_iters > 0

    args = fastnerf.run_nerf.make_args(N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, perturb=1.0, white_bkgd=True,
                                       no_reload=True, lrate=5e-4, lrate_decay=500, N_rand=n_local)
    K = np.array([[FOCAL, 0, 0.5 * W], [0, FOCAL, 0.5 * H], [0, 0, 1]])
    n_img, max_leaves = 100, 256
    poses = torch.stack([pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(n_img)], 0).to(dev)

    def draw_pixels(gen, n):
        return torch.stack([torch.randint(0, n_img, (n,), generator=gen), torch.randint(0, H, (n,), generator=gen),
                            torch.randint(0, W, (n,), generator=gen)], 1).int()

    # ---- PSNR protocol inputs first, so that the CPU side can start right away and run beside the GPU legs ----------------
    psnr_proc = psnr_in = psnr_out = None
    psnr_data = None
    if psnr_leg:
        psnr_data = psnr_inputs(fastnerf, dev, a.psnr_iters, args, poses, K, draw_pixels)
        if not a.no_cpu_baseline:
            tmp = tempfile.mkdtemp(prefix='fastnerf_psnr_')
            psnr_in, psnr_out = os.path.join(tmp, 'in.pt'), os.path.join(tmp, 'out.json')
            torch.save(psnr_data, psnr_in)
            psnr_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--psnr-cpu-worker', psnr_in, psnr_out],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=ROOT)

    # ---- throughput batches: uniformly drawn rays, U[0,1) targets (+ the analytic scene's colours for the sibling legs) ----
    gen = torch.Generator().manual_seed(1000 + rank)
    n_batches = 64          # 262 144 distinct rays per rank at 4096 rays per step, cycled
    batches = []            # (rays_o, rays_d, {targets}, leaf tag)
    for _ in range(n_batches):
        pix = draw_pixels(gen, n_local)
        ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
        # (image, leaf) tags of a depth-5 quadtree: 16 x 16 leaves of 50 x 50 pixels, DFS order irrelevant for timing
        tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
        tg = {'noise': torch.rand(n_local, 3, generator=gen).to(dev)}
        if siblings:
            tg['solid'] = synthetic.render_rays(ro, rd, cutoff=SCENE_CUTOFF).contiguous()
        batches.append((ro, rd, tg, tag))
    table = torch.zeros(n_img * max_leaves, device=dev, dtype=torch.int32)
    n_global = n_step if world > 1 else None

    def new_trainer():
        torch.manual_seed(0)   # identical initial weights on every rank, and for every leg
        k_train, k_test, _, _, grad_vars, _ = fastnerf.run_nerf.create_nerf(args, device=dev)
        return fastnerf.run_nerf.Trainer(k_train, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500), k_train, k_test, grad_vars

    def step(trainer, i, scene='noise'):
        ro, rd, tgts, tag = batches[i % n_batches]
        return trainer.step(ro, rd, tgts[scene], leaf_tag=tag, table=table, max_leaves=max_leaves, n_global=n_global)

    def timed(step_fn, first, warm, steps):
        """W warm-up + K timed steps, barrier + synchronize on both sides, MAX over ranks -> (seconds, last loss, local s)."""
        for i in range(warm):
            step_fn(first + i)
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss2 = step_fn(first + warm + i)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0
        parallel.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t[0]), loss2, t_local

    def leg(seconds, steps, **kw):
        d = {'ms_per_step': 1e3 * seconds / steps, 'value': n_step * steps / seconds, 'unit': 'rays/s', 'steps': steps}
        d.update(kw)
        return d

    def mlp_roofline(trainer, mode):
        """HIP-event timing of the MLP launches of one step's FINE pass (4096 x 192 points per rank at weak scaling); the
        launch the step spends the most time in is `roofline`.  achieved = algorithmic FLOPs of the launch / its duration."""
        ro, rd = batches[0][0], batches[0][1]
        rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
        n = ro.shape[0]
        z = torch.sort(torch.rand(n, S1, device=dev) * 4 + 2, -1).values
        P = n * S1
        act = torch.empty(ops.act_floats(P), device=dev)
        dact = torch.empty(ops.dact_floats(P), device=dev)
        partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
        gtmp = torch.empty(ops.NET_PARAMS, device=dev)
        raw = torch.empty(n, S1, 4, device=dev)
        draw = torch.randn(n, S1, 4, device=dev) * 1e-4
        reps = max(3, min(a.steps, 10))
        peak, n_prod, peak_note, base, targ = MODE_PEAK[mode]
        split = mode == 'bf16x3'
        ms_save = time_launch(lambda: ops.mlp_fwd(rays11, z, trainer.net_f.flat, trainer.pf[0], act=act, raw=raw), reps)
        ms_inf = time_launch(lambda: ops.mlp_fwd(rays11, z, trainer.net_f.flat, trainer.pf[0], raw=raw), reps)
        ms_bwd = time_launch(lambda: ops.mlp_bwd(draw, act, trainer.net_f.flat, trainer.pf[1], dact, partial, gtmp), reps)
        rows = [{'kernel': base + '<true, false%s>' % targ, 'what': 'training forward, saves activations, %d points' % P,
                 'points': P, 'avg_launch_ms': ms_save, 'flop_per_launch': P * FWD_FLOP_PER_POINT},
                {'kernel': base + '<false, false%s>' % targ, 'what': 'forward without saving (inference), %d points' % P,
                 'points': P, 'avg_launch_ms': ms_inf, 'flop_per_launch': P * FWD_FLOP_PER_POINT},
                {'kernel': 'backward of the pass: dX chain + 12 dW launches + head gradients + partial reduction (15 launches)',
                 'what': 'mlp_bwd, %d points' % P, 'points': P, 'avg_launch_ms': ms_bwd, 'flop_per_launch': P * BWD_FLOP_PER_POINT}]
        for r in rows:
            r['achieved'] = r['flop_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e12
            r['frac'] = r['achieved'] / peak
        dom = rows[0]    # the single launch the plain step spends the most time in (the backward row is 15 launches)
        traffic = step_traffic = None
        pmc_file = None
        for rnd in (PROFILE_ROUND, 'r05', 'r04'):   # HBM bytes from the committed PMC passes (separate rocprofv3 --pmc runs, tools/collect_profiles_*.sh)
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', rnd + '_pmc_traffic.json')))
                traffic = pmc[mode]['kernels'][dom['kernel']]['hbm_bytes']
                step_traffic = pmc[mode]['step_traffic']
                pmc_file = 'profiles/%s_pmc_traffic.json' % rnd
                break
            except Exception:
                pass
        roof = {'bound': 'mfma', 'kernel': dom['kernel'] + ' (fine pass; ' + dom['what'] + ')', 'achieved': dom['achieved'],
                'peak': peak, 'unit': 'TFLOP/s', 'frac': dom['frac'],
                'peak_note': peak_note, 'measured_in_this_run': ['achieved', 'frac', 'avg_launch_ms', 'launches'],
                'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC, %s)' % pmc_file,
                'traffic_source': 'REPO CONSTANT from %s (separate rocprofv3 --pmc passes of this workload), not collected in this run' % pmc_file,
                'step_traffic': step_traffic, 'avg_launch_ms': dom['avg_launch_ms'], 'flop_per_launch': dom['flop_per_launch'],
                'mfma_tflops_issued': n_prod * dom['achieved'], 'launches': rows}
        if mode != 'fp32':
            # tools/micro/mfma_power.hip: a saturated v_mfma_f32_32x32x16_bf16 stream holds 2.24-2.38 GHz on constant operands
            # but only 1.79 GHz = 1871 TFLOP/s on uniform(-1,1) bf16 data (power management)
            roof['peak_measured_real_data'] = BF16_MFMA_REAL_DATA_TFLOPS / n_prod
            roof['peak_measured_real_data_source'] = 'REPO CONSTANT (tools/micro/mfma_power.hip on an earlier box), not measured in this run'
            roof['frac_of_measured_peak'] = dom['achieved'] / (BF16_MFMA_REAL_DATA_TFLOPS / n_prod)
            if mode == 'bf16x6':
                roof['frac_of_measured_peak_16x16x32'] = dom['achieved'] / (BF16_MFMA_16X16X32_RANDOM_DATA_TFLOPS / n_prod)
        return roof

    def launch_roofline(kernel, what, fn_, flop, mode):
        """roofline block of ONE launch timed live with HIP events (the dominant kernel of a sibling config)."""
        peak = MODE_PEAK[mode][0]
        ms = time_launch(fn_, 5)
        ach = flop / (ms * 1e-3) / 1e12
        return {'bound': 'mfma', 'kernel': kernel + ' (' + what + ')', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                'avg_launch_ms': ms, 'flop_per_launch': flop, 'traffic': None, 'peak_note': MODE_PEAK[mode][2]}

    # =====================================================================================================================
    # BASELINE configs[2], [3], [4] at the headline arithmetic (siblings: never `value`)
    # =====================================================================================================================
    def config2_quadtree():
        """configs[2]: the whole epoch loop of run_nerf.train() with quadtree ray selection -- per-epoch one-launch ray generation from
        the leaf plans, fused steps feeding the on-device leaf-error table, tree adjustment -- END TO END incl. all host work."""
        ops.set_math(MAIN_MODE)
        fastnerf.render.set_compact('0')
        Hq = Wq = 400
        imgs, qposes, focal = synthetic.make_dataset(n_images=10, H=Hq, W=Wq, device='cuda')
        torch.manual_seed(0); np.random.seed(0)
        qa = fastnerf.run_nerf.make_args(N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, perturb=1.0, white_bkgd=True, no_reload=True,
                                         N_rand=N_RAYS, n_epoch=3, init_level=2, subdivide_every=1, subdivide_thres=0.02, lrate=5e-4,
                                         lrate_decay=500)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, qtr, mgr, hist = fastnerf.run_nerf.train(imgs, qposes, Hq, Wq, focal, qa, log=lambda *_: None, compat_rng=False)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        epochs = [{'epoch': ep, 'steps': it, 'seconds': sec, 'rays_per_s': it * N_RAYS / sec, 'psnr_db': float(psnr)} for ep, it, _, psnr, sec in hist]
        rays = sum(e['steps'] for e in epochs) * N_RAYS
        secs = sum(e['seconds'] for e in epochs)
        return {'workload': 'BASELINE configs[2]: nerf-ours with quadtree adaptive ray selection, 10 analytic views of 400x400, init_level 2, '
                            'subdivide every epoch (thres 0.02), 4096 rays x (64+128) samples per step, 3 epochs incl. centre-crop warm-up, '
                            'device ray generation (compat_rng=False), plain backward', 'math_mode': MAIN_MODE,
                'value': rays / secs, 'unit': 'rays/s', 'what': 'rays trained per second of epoch wall time: ray generation, fused steps, leaf '
                'table, tree adjustment, all host work included', 'epochs': epochs, 'wall_seconds_incl_warmup_and_setup': wall,
                'leaves_max': int(mgr.max_leaves()), 'fraction_of_headline_step_rate': (rays / secs) / (n_step * a.steps / dt),
                'roofline': dict(roof, note='the step is the headline step (same kernels, same 4096 x 192 fine pass): its roofline block applies')}

    def config3_llff():
        """configs[3] shape on one GPU: forward-facing cameras of 1008 x 756, NDC rays, 64 + 64 samples, raw_noise_std = 1."""
        ops.set_math(MAIN_MODE)
        fastnerf.render.set_compact('0')
        Hl, Wl, fl = 756, 1008, 815.13
        Kl = np.array([[fl, 0, 0.5 * Wl], [0, fl, 0.5 * Hl], [0, 0, 1]])
        torch.manual_seed(0)
        la = fastnerf.run_nerf.make_args(N_importance=64, N_samples=64, perturb=1.0, raw_noise_std=1.0, no_reload=True, dataset_type='llff',
                                         lrate=5e-4, lrate_decay=250)
        lk = fastnerf.run_nerf.create_nerf(la, device=dev)[0]
        lposes = torch.eye(4)[None, :3, :4].repeat(20, 1, 1)
        lposes[:, 0, 3] = torch.linspace(-0.3, 0.3, 20)
        g = torch.Generator().manual_seed(1)
        lb = []
        for _ in range(8):
            pix = torch.stack([torch.randint(0, 20, (N_RAYS,), generator=g), torch.randint(0, Hl, (N_RAYS,), generator=g),
                               torch.randint(0, Wl, (N_RAYS,), generator=g)], 1).int().to(dev)
            lb.append(ops.gen_rays_pixels(pix, lposes.to(dev), Kl) + (torch.rand(N_RAYS, 3, generator=g).to(dev),))
        ltr = fastnerf.run_nerf.Trainer(lk, Hl, Wl, Kl, 0.0, 1.0, lrate=5e-4, lrate_decay=250)
        assert ltr.ndc and ltr.raw_noise_std == 1.0
        tl_, ll_, _ = timed(lambda i: ltr.step(*lb[i % 8])[0], 0, 3, 20)
        P = N_RAYS * 128
        rays11 = ops.pack_rays(lb[0][0], lb[0][1], 0.0, 1.0)
        z = torch.sort(torch.rand(N_RAYS, 128, device=dev), -1).values
        act, raw = torch.empty(ops.act_floats(P), device=dev), torch.empty(N_RAYS, 128, 4, device=dev)
        blk = leg(tl_, 20, warmup=3, backward='plain', final_loss=[float(x) for x in ll_.tolist()])   # (siblings run at world == 1)
        blk.update(workload='BASELINE configs[3] shape, ONE GPU (the config names 8): LLFF-like forward-facing cameras 1008x756, NDC rays, 4096 '
                            'rays x (64+64) samples per step, raw_noise_std=1, U[0,1) targets, random-init nets, plain backward',
                   math_mode=MAIN_MODE,
                   roofline=launch_roofline('mlp_fwd_kernel<true, false, 1>', 'fine pass, saves activations, %d points' % P,
                                            lambda: ops.mlp_fwd(rays11, z, ltr.net_f.flat, ltr.pf[0], act=act, raw=raw), P * FWD_FLOP_PER_POINT, MAIN_MODE))
        return blk

    def config4_nerfpp():
        """configs[4] shape on one GPU: the nerf++ cascade batch (2 levels x (fg + bg) nets, 64 / 128 samples) at the config's 1920 rays."""
        from fastnerf import nerfpp
        ops.set_math(MAIN_MODE)
        torch.manual_seed(0)
        nets = [nerfpp.NerfNet(device=dev) for _ in range(2)]
        ptr_ = nerfpp.CascadeTrainer(nets, cascade_samples=(64, 128))
        g = torch.Generator().manual_seed(1)
        n = 1920
        pb_ = [(((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev), torch.randn(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev))
               for _ in range(8)]
        tp_, lp_, _ = timed(lambda i: ptr_.step(*pb_[i % 8])[0], 0, 3, 10)
        P = n * 192
        rays11 = ops.pack_rays(pb_[0][0], pb_[0][1], 0.0, 0.0)
        zb = torch.sort(torch.rand(n, 192, device=dev), -1).values
        bg = nets[1].bg_net
        act = torch.empty(ops.act_floats(P, 2), device=dev)
        blk = {'ms_per_step': 1e3 * tp_ / 10, 'value': n * 10 / tp_, 'unit': 'rays/s', 'steps': 10, 'warmup': 3,
               'final_loss': [float(x) for x in lp_.tolist()],
               'workload': 'BASELINE configs[4] shape, ONE GPU (the config names 8): nerf++-ours inverted-sphere fg / bg dual MLP, cascade of 2 '
                           'levels (64, then 64+128 samples per net): 512 MLP evaluations per ray, 1920 rays per batch (the config\'s batch), random '
                           'cameras inside the unit sphere, U[0,1) targets, Adam per level (CascadeTrainer: the engine of the sharded ddp_train_nerf)',
               'math_mode': MAIN_MODE,
               'roofline': launch_roofline('mlp_fwd_kernel<true, true, 1>', 'level-1 background net, 4-D inverted-sphere encoding, saves activations, '
                                           '%d points' % P,
                                           lambda: ops.mlp_fwd(rays11, zb, bg.flat, bg.packed(refresh=False)[0], act=act, kind=2),
                                           P * 2 * MAC_PER_POINT_BG, MAIN_MODE)}
        return blk

    # =====================================================================================================================
    # headline: fp32-width arithmetic (MAIN_MODE), SURVEY 8(d) protocol, plain backward
    # =====================================================================================================================
    main_mode, main_compact = ops.get_math(), fastnerf.render.get_compact()
    ops.set_math(MAIN_MODE)
    fastnerf.render.set_compact('0')
    tr, _, kte, _ = new_trainer()
    dt, loss2, t_local = timed(lambda i: step(tr, i)[0], 0, a.warmup, a.steps)
    per_rank_ms = [1e3 * t_local / a.steps]
    allreduce_ms = None
    if world > 1:
        tl = torch.tensor([1e3 * t_local / a.steps], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(gathered, tl)
        per_rank_ms = [float(g[0]) for g in gathered]
        # the step's data-path collectives, timed alone: all-reduce(SUM) of each net's half of the flat gradient (2.38 MB each;
        # in the step the fine net's runs beside the coarse pass's backward)
        Nn = ops.NET_PARAMS
        allreduce_ms = {'fine_half': time_launch(lambda: parallel.all_reduce_sum(tr.grad[Nn:]), 20),
                        'coarse_half': time_launch(lambda: parallel.all_reduce_sum(tr.grad[:Nn]), 20),
                        'whole_buffer': time_launch(lambda: parallel.all_reduce_sum(tr.grad), 20),
                        'overlapped_with_coarse_backward': bool(tr.overlap_allreduce)}
    sustained = None
    if a.sustained_steps > 0:   # power-managed clocks settle within seconds: the same stream of steps for a few seconds more
        ts, _, _ = timed(lambda i: step(tr, i)[0], a.warmup + a.steps, 0, a.sustained_steps)
        sustained = leg(ts, a.sustained_steps, wall_seconds=ts)
    # ---- --scaling both: the OTHER curve in the same invocation (every rank takes part: the steps hold collectives) ------------------
    scaling_blocks = None
    if both:
        def scaling_leg(n_loc):
            trs_ = new_trainer()[0]
            ng = n_loc * world

            def st(i):
                ro, rd, tgts, tag = batches[i % n_batches]
                return trs_.step(ro[:n_loc], rd[:n_loc], tgts['noise'][:n_loc], leaf_tag=tag[:n_loc].contiguous(), table=table,
                                 max_leaves=max_leaves, n_global=ng if world > 1 else None)[0]
            t_, l_, tl_ = timed(st, 0, a.warmup, a.steps)
            return {'rays_per_gpu_per_step': n_loc, 'rays_per_step': ng, 'ms_per_step': 1e3 * t_ / a.steps, 'value': ng * a.steps / t_,
                    'unit': 'rays/s', 'steps': a.steps, 'warmup': a.warmup, 'rank0_ms_per_step': 1e3 * tl_ / a.steps,
                    'final_loss': [float(x) for x in l_.tolist()]}
        weak_blk = {'rays_per_gpu_per_step': n_local, 'rays_per_step': n_step, 'ms_per_step': 1e3 * dt / a.steps, 'value': n_step * a.steps / dt,
                    'unit': 'rays/s', 'steps': a.steps, 'warmup': a.warmup, 'rank0_ms_per_step': 1e3 * t_local / a.steps,
                    'final_loss': [float(x) for x in loss2.tolist()], 'note': 'this leg is `value`'}
        strong_blk = scaling_leg(max(1, N_RAYS // world))
        strong_blk['note'] = '4096 rays per step split over the ranks (BASELINE configs[1] as ONE job): value / (the N=1 value) is the strong-scaling factor'
        scaling_blocks = {'weak': weak_blk, 'strong': strong_blk}
    roof = mlp_roofline(tr, MAIN_MODE) if rank == 0 else None
    step_tflops = n_step * a.steps / dt * TRAIN_FLOP_PER_RAY / 1e12 / world

    # =====================================================================================================================
    # siblings (1 GPU, rank 0): never part of `value`; each leg guarded -- a failing sibling is recorded, the headline still prints
    # =====================================================================================================================
    errors = []

    def guarded(name, fn_):
        try:
            return fn_()
        except Exception as e:     # noqa: BLE001  (recorded in the record, never swallowed silently)
            import traceback
            errors.append('%s: %s: %s | %s' % (name, type(e).__name__, e, traceback.format_exc().strip().splitlines()[-2].strip()))
            return None

    def mode_leg(mode, **kw):
        """The headline protocol (random init, U[0,1) targets, plain backward) in another arithmetic."""
        ops.set_math(mode)
        fastnerf.render.set_compact('0')
        trm, _, ktm, _ = new_trainer()
        tm_, lm_, _ = timed(lambda i: step(trm, i)[0], 0, 3, 20)
        blk = {'math_mode': mode, 'init_state': leg(tm_, 20, warmup=3, backward='plain', final_loss=[float(x) for x in lm_.tolist()],
                                                    what='the headline protocol in this mode'),
               'roofline': mlp_roofline(trm, mode)}
        blk.update(kw)
        return blk, trm, ktm

    def sparse_scene_leg(mode):
        """`mode` on a trained sparse scene with the exact zero-gradient compaction (what loss.backward() sees after the first epochs,
        run_nerf.py:493), and the plain backward on the same weights."""
        ops.set_math(mode)
        fastnerf.render.set_compact('auto')
        trs, _, _, _ = new_trainer()
        for i in range(a.scene_steps):
            step(trs, i, 'solid')
        tsol, lsol, _ = timed(lambda i: step(trs, i, 'solid')[0], a.scene_steps, 3, 20)
        live_frac, was_live = None, trs.last_step_live
        if was_live:
            c = trs.live_counts.cpu().tolist()
            live_frac = {'fine': c[0] / max(1, c[1]), 'coarse': c[2] / max(1, c[3])}
        fastnerf.render.set_compact('0')
        tpl, _, _ = timed(lambda i: step(trs, i, 'solid')[0], 7, 6, 20)
        return leg(
            tsol, 20, warmup=3, math_mode=mode, backward='compacted' if was_live else 'plain', live_fraction=live_frac,
            after_optimisation_steps=a.scene_steps + 3, final_loss=[float(x) for x in lsol.tolist()],
            what='three solid analytic bodies on white (density exactly zero beyond %.1f sigma, ~27 %% of the pixels covered): nets '
                 'trained inside the run, backward over the samples with a non-zero gradient only (exact; DESIGN 4a).  Scene and '
                 'trajectory dependent: a field without exactly-empty space runs at `same_state_plain_backward`' % SCENE_CUTOFF,
            same_state_plain_backward=leg(tpl, 20, warmup=6))

    def drop_in_leg(mode):
        """INTEGRATION option A: the reference's own loop on the drop-in surface (run_nerf.py:479-508)."""
        ops.set_math(mode)
        fastnerf.render.set_compact('0')
        torch.manual_seed(0)
        k_train, _, _, _, grad_vars, optimizer = fastnerf.run_nerf.create_nerf(args, device=dev)
        k_train.update(near=2.0, far=6.0)
        state = {'it': 0}

        def one(i):
            ro, rd, tgts, _ = batches[i % n_batches]
            rgb, disp, acc, extras = fastnerf.render.render(H, W, K, chunk=args.chunk, rays=(ro, rd), retraw=True, **k_train)
            optimizer.zero_grad()
            img_loss = fastnerf.run_nerf_helpers.img2mse(rgb, tgts['noise'])
            loss = img_loss + fastnerf.run_nerf_helpers.img2mse(extras['rgb0'], tgts['noise'])
            loss.backward()
            optimizer.step()
            new_lrate = 5e-4 * (0.1 ** (state['it'] / (500 * 1000)))
            for pg in optimizer.param_groups:
                pg['lr'] = new_lrate
            state['it'] += 1
            return img_loss.detach()
        td, ld, _ = timed(one, 0, 3, 20)
        return leg(td, 20, warmup=3, final_img_loss=float(ld))

    def inference_leg(test_kwargs):
        """render_path-style, perturb=0, no saved activations."""
        n_inf = 32768
        pix = draw_pixels(torch.Generator().manual_seed(7), n_inf).to(dev)
        ro_i, rd_i = ops.gen_rays_pixels(pix, poses, K)
        inf = {'what': 'render() of 32768 rays, 64+128 samples, perturb=0 (render_kwargs_test), random-init nets, 1 GPU'}
        for mode, kt in test_kwargs.items():
            if kt is None:
                continue
            ops.set_math(mode)
            with torch.no_grad():
                for _ in range(2):
                    fastnerf.render.render(H, W, K, chunk=n_inf, rays=(ro_i, rd_i), near=2.0, far=6.0, **kt)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    fastnerf.render.render(H, W, K, chunk=n_inf, rays=(ro_i, rd_i), near=2.0, far=6.0, **kt)
                torch.cuda.synchronize()
                dt_i = (time.perf_counter() - t1) / 5
            inf[mode] = {'value': n_inf / dt_i, 'unit': 'rays/s', 'ms_per_call': 1e3 * dt_i}
        return inf

    def batch_size_leg():
        """The headline step (same arithmetic, plain backward, leaf table on) at the reference's OWN batch sizes and at an 8-way strong-scaling
        shard of configs[1]: every shipped config trains at N_rand 1024 - 1920 (nerf-ours/configs/lego.txt:16 = 1920, fern.txt:9 = 1536, nine
        configs at 1024), none at 4096; 4096 / 8 = 512 is what one rank of `--gpus 8 --scaling strong` steps.  The first N rays of the headline's
        batches; best of three rounds of 30 steps per size (HIP events), sizes interleaved so that clock drift hits them alike."""
        ops.set_math(MAIN_MODE)
        fastnerf.render.set_compact('0')
        sizes = [n for n in (512, 1024, 1920, 4096) if n <= n_local]
        trs = {n: new_trainer()[0] for n in sizes}

        def one(n, i):
            ro, rd, tgts, tag = batches[i % n_batches]
            return trs[n].step(ro[:n], rd[:n], tgts['noise'][:n], leaf_tag=tag[:n].contiguous(), table=table, max_leaves=max_leaves)
        best = {n: float('inf') for n in sizes}
        for _ in range(3):
            for n in sizes:
                it = [0]

                def f():
                    one(n, it[0]); it[0] += 1
                best[n] = min(best[n], time_launch(f, 30))
        top = sizes[-1] / best[sizes[-1]]
        blk = {'what': 'the headline protocol at other batch sizes on ONE GPU: ms / step, k rays/s, fraction of the %d-ray rate (best of 3 x 30 steps)' % sizes[-1],
               'sizes': {str(n): {'ms_per_step': best[n], 'rays_per_s': 1e3 * n / best[n], 'frac_of_%d_rate' % sizes[-1]: (n / best[n]) / top} for n in sizes}}
        if 512 in best and 4096 in best:
            ar = 0.1    # ms; ASSUMED: the exposed coarse-half all-reduce of 2.38 MB over xGMI (latency-bound: 7 links, direct reduce-scatter + all-gather
            #             ~0.05 ms, a ring ~0.14 ms); the fine half rides under the coarse backward.  No multi-GPU box has run this.
            blk['predicted_strong_8'] = {
                'value': (4096.0 / (best[512] + ar)) / top, 'label': 'PREDICTION from 1-GPU timings, not a measurement',
                'formula': '8 * 512 / (t_512 + exposed half-buffer all-reduce) / (4096 / t_4096)', 't_512_ms': best[512], 't_4096_ms': best[4096],
                'assumed_exposed_allreduce_ms': ar, 'without_collective': (4096.0 / best[512]) / top}
        return blk

    split_block = fp32_block = drop_in = infer = psnr_block = cfg_blocks = main_sparse = batch_sizes = None
    kte_b = kte_32 = None
    dd = None
    if siblings:
        batch_sizes = guarded('batch_sizes', batch_size_leg)
        r_ = guarded('fp32_mfma_mode', lambda: mode_leg(
            'fp32', dtype='f32: v_mfma_f32_32x32x2_f32, an fp32 FMA chain (157.3 TFLOP/s ceiling)'))
        if r_ is not None:
            fp32_block, _, kte_32 = r_
            fp32_block['init_state']['step_frac_of_fp32_mfma_peak'] = (fp32_block['init_state']['value'] * TRAIN_FLOP_PER_RAY / 1e12
                                                                       / FP32_MFMA_PEAK_TFLOPS)
        r_ = guarded('split_bf16_mode', lambda: mode_leg(
            'bf16x3', dtype='split-bf16 x3: every fp32 product as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation; '
                            'operands carry 16 significand bits -- NARROWER than fp32, hence a sibling'))
        if r_ is not None:
            split_block, _, kte_b = r_
            split_block['init_state']['step_frac_of_bf16_mfma_peak_x3'] = (3.0 * split_block['init_state']['value'] * TRAIN_FLOP_PER_RAY / 1e12
                                                                           / BF16_MFMA_PEAK_TFLOPS)
            split_block['sparse_scene'] = guarded('split_bf16_mode.sparse_scene', lambda: sparse_scene_leg('bf16x3'))
        # ---- the headline arithmetic on the trained sparse scene: the step loss.backward() sees after the first epochs ------------
        main_sparse = guarded(MAIN_MODE + '.sparse_scene', lambda: sparse_scene_leg(MAIN_MODE))

        def all_drop_in():
            dmain, d32, d16 = drop_in_leg(MAIN_MODE), drop_in_leg('fp32'), drop_in_leg('bf16x3')
            frac = {MAIN_MODE: dmain['value'] / (n_step * a.steps / dt)}
            if fp32_block is not None:
                frac['fp32'] = d32['value'] / fp32_block['init_state']['value']
            if split_block is not None:
                frac['bf16x3'] = d16['value'] / split_block['init_state']['value']
            return {'what': 'the reference\'s loop verbatim on the drop-in surface: render(retraw=True) -> img2mse x 2 -> loss.backward() -> '
                            'torch.optim.Adam(48 tensors).step() -> lr rule; same protocol and batches as `value`',
                    MAIN_MODE: dmain, 'fp32': d32, 'bf16x3': d16, 'fraction_of_fused_trainer': frac}
        drop_in = guarded('drop_in_route', all_drop_in)

    # ---- PSNR at equal iterations: GPU side (free runs) on the inputs the CPU worker got -------------------------------------------
    psnr_modes = (MAIN_MODE, 'fp32', 'bf16x3') if siblings else (MAIN_MODE,)
    if psnr_data is not None:
        def psnr_gpu():
            blk = {'iters': psnr_data['iters'], 'rays_per_iter': PSNR_RAYS, 'samples': '64+128', 'cameras': '100 x 800x800',
                   'scene': 'three analytic Gaussian density blobs on white (fastnerf.synthetic), identical batches, injected '
                            't_rand / u (seed 2) and initial weights (seed 0) on both sides',
                   'train_psnr_window': 20, 'held_out_rays': PSNR_HELD_OUT, 'gpu': {}}
            for mode in psnr_modes:
                blk['gpu'][mode] = psnr_gpu_free(fastnerf, dd, new_trainer, K, mode)[0]
            blk['free_runs_note'] = (
                'single free runs from one initialisation: trajectories are chaotic, and the level reached after 200 iterations depends on the '
                'initial weights by ~3 dB; the falsifiable free-run statement is the PAIRED test over the initialisation seeds of the committed '
                'CPU ensemble G22 (tests/test_gpu_train.py::test_psnr_paired_with_the_cpu_ensemble_g22); the per-step statement is `lockstep`')
            return blk
        dd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in psnr_data.items()}
        psnr_block = guarded('psnr_vs_cpu.gpu', psnr_gpu)

    if siblings:
        # ---- the other BASELINE configs at the headline arithmetic --------------------------------------------------------------
        # (these loops have host work in their timed regions: the CPU PSNR worker -- our own child process -- is paused for their ~30 s; on boxes with
        #  fewer than 128 CPUs it cannot be kept off the launching process's cores, and the nerf++ batch then measured 30 instead of 18.4 ms)
        import signal
        if psnr_proc is not None and psnr_proc.poll() is None:
            psnr_proc.send_signal(signal.SIGSTOP)
        try:
            cfg_blocks = {k: v for k, v in (('configs[2]_quadtree', guarded('configs[2]', config2_quadtree)),
                                            ('configs[3]_llff_ndc', guarded('configs[3]', config3_llff)),
                                            ('configs[4]_nerfpp', guarded('configs[4]', config4_nerfpp))) if v is not None}
        finally:
            if psnr_proc is not None and psnr_proc.poll() is None:
                psnr_proc.send_signal(signal.SIGCONT)
        infer = guarded('inference', lambda: inference_leg({MAIN_MODE: kte, 'fp32': kte_32, 'bf16x3': kte_b}))
    ops.set_math(main_mode)
    fastnerf.render.set_compact(main_compact)

    if rank == 0:
        # ---- CPU legs: wait for the PSNR worker (it ran beside the GPU legs), then time the step alone ----------------------
        cpu = None
        psnr_head = None

        def psnr_cpu_side():
            _, err = psnr_proc.communicate()
            if psnr_proc.returncode != 0:
                psnr_block['cpu'] = {'error': err.decode()[-400:]}
                return
            r = json.load(open(psnr_out))
            psnr_block['cpu'] = {'train_psnr_db': psnr_of(r['losses'], 20), 'held_out_psnr_db': -10.0 * math.log10(r['held_out_mse']),
                                 'first_loss': r['losses'][0], 'last_loss': r['losses'][-1], 'threads': r['threads'],
                                 'rays_per_s_while_gpu_legs_ran': r['rays_per_s'],
                                 'what': 'oracle/nerf_oracle.py train_step / render_rays (torch CPU fp32), the same inputs'}
            psnr_block['delta_db'] = {m: {k: psnr_block['gpu'][m][k] - psnr_block['cpu'][k] for k in ('train_psnr_db', 'held_out_psnr_db')}
                                      for m in psnr_block['gpu']}
            states = np.load(psnr_out + '.states', mmap_mode='r')
            psnr_block['lockstep'] = {
                'what': 'the GPU step taken from the CPU run\'s state before every iteration (weights + Adam moments): the same PSNR '
                        'window without trajectory divergence; delta_db = GPU - CPU'}
            for mode in psnr_modes:
                r_ = psnr_gpu_lockstep(fastnerf, dd, new_trainer, states, r['losses'], mode)
                r_['delta_db'] = r_['train_psnr_db'] - psnr_block['cpu']['train_psnr_db']
                psnr_block['lockstep'][mode] = r_
            del states
        if psnr_proc is not None:
            if psnr_block is not None:
                guarded('psnr_vs_cpu.cpu', psnr_cpu_side)
            elif psnr_proc.poll() is None:
                psnr_proc.kill()
            ops.set_math(main_mode)
            fastnerf.render.set_compact(main_compact)
            for p in (psnr_in, psnr_out, psnr_out + '.states'):
                if p and os.path.exists(p):
                    os.remove(p)
        if psnr_block is not None:    # the compact block of the headline line
            g_ = psnr_block['gpu'].get(MAIN_MODE, {})
            psnr_head = {'iters': psnr_block['iters'], 'rays_per_iter': PSNR_RAYS, 'gpu_train_db': g_.get('train_psnr_db'),
                         'gpu_held_out_db': g_.get('held_out_psnr_db')}
            if 'train_psnr_db' in psnr_block.get('cpu', {}):
                psnr_head.update(cpu_train_db=psnr_block['cpu']['train_psnr_db'], cpu_held_out_db=psnr_block['cpu']['held_out_psnr_db'])
                ls_ = (psnr_block.get('lockstep') or {}).get(MAIN_MODE)      # absent when the lockstep replay failed (recorded in `errors`)
                if isinstance(ls_, dict) and 'delta_db' in ls_:
                    psnr_head.update(lockstep_delta_db=ls_['delta_db'], lockstep_max_rel_loss_diff=ls_.get('max_rel_loss_diff'))
            psnr_head['paired_ensemble'] = PAIRED_PSNR_NOTE
        if world == 1 and not a.no_cpu_baseline:
            cpu = guarded('cpu_baseline', lambda: cpu_baseline(a.cpu_protocol))
            if cpu is not None and cfg_blocks:
                def cfg_cpu():
                    if 'configs[2]_quadtree' in cfg_blocks:
                        cfg_blocks['configs[2]_quadtree']['cpu_baseline'] = dict(
                            cpu, note='the step of configs[2] is configs[1]\'s (the quadtree only chooses the rays): the same CPU figure; the '
                                      'reference\'s host-side quadtree work at the shipped scale is timed by tools/bench_quadtree_full.py '
                                      '(profiles/%s_quadtree_full.json)' % PROFILE_ROUND)
                    if 'configs[3]_llff_ndc' in cfg_blocks:
                        cfg_blocks['configs[3]_llff_ndc']['cpu_baseline'] = cpu_baseline_llff()
                    if 'configs[4]_nerfpp' in cfg_blocks:
                        cfg_blocks['configs[4]_nerfpp']['cpu_baseline'] = cpu_baseline_nerfpp()
                    for b in cfg_blocks.values():
                        if 'cpu_baseline' in b:
                            b['vs_cpu_baseline'] = b['value'] / b['cpu_baseline']['value']
                guarded('other_configs.cpu_baseline', cfg_cpu)
        rays_per_s = n_step * a.steps / dt

        def ms_of(blk):
            return None if blk is None else round(blk['init_state']['ms_per_step'], 3)
        summary = None
        if siblings:
            summary = {'ms_per_step': {'fp32': ms_of(fp32_block), 'bf16x3': ms_of(split_block),
                                       'drop_in_' + MAIN_MODE: None if drop_in is None else round(drop_in[MAIN_MODE]['ms_per_step'], 3)},
                       'sparse_scene_' + MAIN_MODE: None if main_sparse is None else {
                           'ms_per_step': round(main_sparse['ms_per_step'], 3), 'rays_per_s': round(main_sparse['value']),
                           'live_fine': None if not main_sparse['live_fraction'] else round(main_sparse['live_fraction']['fine'], 4),
                           'plain_ms': round(main_sparse['same_state_plain_backward']['ms_per_step'], 3)},
                       'configs_rays_per_s': None if not cfg_blocks else {k.split('_')[0]: round(v['value']) for k, v in cfg_blocks.items()},
                       'inference_rays_per_s': None if infer is None or MAIN_MODE not in infer else round(infer[MAIN_MODE]['value']),
                       # the step at the reference's own batch sizes / an 8-way strong-scaling shard: ms per step, and the predicted (NOT measured)
                       # 8-GPU strong-scaling factor 8 * 512 / (t_512 + 0.1 ms of exposed all-reduce) over the 4096-ray rate
                       'batch_ms': None if batch_sizes is None else {k: round(v['ms_per_step'], 3) for k, v in batch_sizes['sizes'].items()},
                       'batch_frac_of_4096_rate': None if batch_sizes is None else {
                           k: round(v.get('frac_of_4096_rate', float('nan')), 3) for k, v in batch_sizes['sizes'].items()},
                       'predicted_strong_8': None if batch_sizes is None or 'predicted_strong_8' not in batch_sizes else round(
                           batch_sizes['predicted_strong_8']['value'], 2)}
        out = {
            'metric': 'training rays/sec (Lego-like 800x800, 64+128 samples) + PSNR@N-iters', 'value': rays_per_s, 'unit': 'rays/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak' if both else a.scaling, 'vs_baseline': None,
            'scaling_weak': None if scaling_blocks is None else scaling_blocks['weak'],
            'scaling_strong': None if scaling_blocks is None else scaling_blocks['strong'],
            'affinity': parallel.affinity_report(),
            'dtype': 'f32', 'math_mode': MAIN_MODE + ': fp32-width products on the bf16 matrix cores -- operands decomposed exactly into three bf16 '
                                         'pieces, six piece products per fp32 product (weight >= 2^-16; dropped <= 2^-24), fp32 accumulation; '
                                         'activations, gradients, parameters and optimiser state stay fp32',
            'data': 'synthetic (SURVEY 8d: 100 pose_spherical cameras of 800x800, uniformly drawn rays, U[0,1) targets, default-init nets seed 0)',
            'config': {'workload': 'nerf-ours Lego full 800x800 (BASELINE configs[1]), SURVEY 8(d) throughput protocol: %d uniformly drawn rays '
                                   'per GPU per step, 64+128 samples, use_viewdirs, white_bkgd, perturb=1, U[0,1) targets, random-init nets, '
                                   'leaf-error table on, fp32-width bf16x6 kernels, plain backward (FASTNERF_COMPACT=0)' % n_local,
                       'rays_per_gpu_per_step': n_local, 'rays_per_step': n_step, 'parallelism': f'dp{world}'},
            'device': {'name': torch.cuda.get_device_name(dev), 'compute_units': torch.cuda.get_device_properties(dev).multi_processor_count},
            'final_loss': [float(x) for x in loss2.tolist()], 'backward': 'plain (every sample)',
            'per_rank_ms_per_step': per_rank_ms, 'allreduce_ms': allreduce_ms, 'collective': parallel.collective_route(),
            'step_tflops_per_gpu': step_tflops, 'step_frac_of_peak': step_tflops / MODE_PEAK[MAIN_MODE][0],
            'step_frac_of_fp32_mfma_peak': step_tflops / FP32_MFMA_PEAK_TFLOPS,
            'step_tflops_note': 'rays/s x the algorithmic FLOPs of a step (893.2 MFLOP/ray, SURVEY 8d); the fp32-MFMA roofline of the step is '
                                '%.1f k rays/s/GPU' % (FP32_MFMA_PEAK_TFLOPS * 1e12 / TRAIN_FLOP_PER_RAY / 1e3),
            'sustained': sustained,
            'roofline': roof,
            'psnr': psnr_head,
            'psnr_vs_cpu': psnr_block,
            'fp32_mfma_mode': fp32_block,
            'split_bf16_mode': split_block,
            'batch_sizes': batch_sizes,
            MAIN_MODE + '_sparse_scene': main_sparse,
            'drop_in_route': drop_in,
            'other_configs': cfg_blocks,
            'inference': infer,
            'cpu_baseline': cpu,
            'siblings_summary': summary,
            'errors': errors,
        }
        out['full_record'] = write_full_record(out, a.full_out)
        sys.stderr.flush()
        print(json.dumps(headline_record(out)), flush=True)      # the ONE stdout line, last
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
