#!/usr/bin/env python3
"""bench.py -- training rays/s of the MI355X-native NeRF inner loop (BASELINE.json metric).

A "step" = one full optimisation step (ray packing -> stratified + hierarchical sampling -> PE ->
coarse/fine MLP -> compositing -> 2xMSE + per-(image, leaf) error table (atomicMax) -> backward ->
[RCCL all-reduce] -> Adam + LR decay) on 4096 rays x (64 + 128) samples per GPU, i.e. BASELINE.json
configs[1] ("nerf-ours Lego full 800x800, 4096 rays, 64+128 samples"): synthetic Lego-like cameras
(100 x pose_spherical, 800x800, focal 1111.11, near 2 / far 6), quadtree leaf tags of a depth-5 tree
(256 leaves per image), default-init weights (seed 0).  Inputs are resident in HBM before the timed region.

What the nets are trained on matters since round 2: the backward skips samples whose gradient is EXACTLY zero
(sigma <= 0: empty space of a radiance field), so throughput depends on where the field puts its density.
`value` is measured on the analytic Lego-like scene of fastnerf.synthetic: three SOLID coloured bodies (density exactly zero
beyond 1.5 standard deviations of each blob's centre) on a white background, covering ~30 % of the pixels like the Lego
bulldozer; targets by quadrature along each batch's rays.  The nets are first optimised from their random initialisation for
--scene-steps (600) untimed steps, then W warm-up and K timed steps follow on the same stream of batches.  On this scene the
fraction of samples that stay live is STATIONARY from step ~500 on (0.19 fine / 0.08 coarse through 6000 steps,
tools/live_trajectory.py), so `value` does not depend on when it is measured.  Reported next to it in the same line:
  * `steady_state_plain`  the same state with the compaction off = the floor, what a field without dead samples costs;
  * `gaussian_tails_scene`  the same three blobs WITHOUT the cut-off (round-2's first protocol): density that never vanishes.
    After 300 steps 53 % / 37 % of the samples are live; the fraction then RISES with training (0.72 at 2000 steps, 0.8-0.9 from
    3500 on: the nets learn the faint tails) and the policy falls back to the plain backward -- the long-run rate on that scene
    is the floor.  Real scenes sit between the two: how many samples die is a property of the scene and of the training
    trajectory (a field that explains empty space as thin white fog keeps them all);
  * `init_state`  the round-1 protocol: random-init nets, U[0,1) noise targets, no scene (84 % live, plain backward).
The GPU legs import nothing from oracle/; only `cpu_baseline` does.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import math
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS, N_SAMPLES, N_IMPORTANCE = 4096, 64, 128
SCENE_CUTOFF = 1.5      # solid bodies: density exactly zero beyond 1.5 standard deviations of a blob's centre
S1 = N_SAMPLES + N_IMPORTANCE
MAC_PER_POINT = 593408
FWD_FLOP_PER_POINT = 2 * MAC_PER_POINT     # 1.186816 MFLOP
TRAIN_FLOP_PER_RAY = 2 * (3 * MAC_PER_POINT - 35712) * (N_SAMPLES + S1)  # 893.2 MFLOP
FP32_MFMA_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0             # MI355X_MICROARCH.md: dense bf16 matrix peak
PROFILE_ROUND = 'r02'


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(protocol='full'):
    """The CPU oracle (a port of the reference's step, validated against it by tests/) timed on the host cores of this
    box, SURVEY §8(d) protocol: the identical step (64+128 samples, fp32) at N = 1024 rays, 3 warm-up + 10 timed steps,
    median, with 32 threads (where torch's CPU GEMMs peak on this box); one step at N = 4096 as a confirmation; a short
    scan of larger thread counts on 256 rays.  About 65 s in total.  `--cpu-protocol short` = 1 + 3 steps of 256 rays."""
    from oracle import nerf_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
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

    t32 = max(1, min(avail, 32))
    if protocol == 'short':
        v = run(256, t32, 1, 3)
        return {'value': v, 'unit': 'rays/s', 'cores': t32, 'kind': 'port', 'cpu': cpu_model(), 'cores_available': avail,
                'sample': f'1 warm-up + 3 timed steps of 256 rays x (64+128) samples, median, torch CPU fp32, {t32} threads '
                          '(oracle/nerf_oracle.py train_step)'}
    v32 = run(1024, t32, 3, 10)
    v4096 = run(4096, t32, 0, 1)
    # more threads only lose on this box (tools/cpu_threads_probe.py, round 2: N=1024 315 rays/s with 32 threads, 210 with 64,
    # 89 with 128; with all 256 hardware threads ONE step takes 52-62 s whatever the ray count -- 16 rays or 64 -- so that
    # setting is reported from the probe instead of spending three minutes on it in every bench run)
    scan = {}
    for th in (64, 128):
        if avail >= th:
            scan[f'rays_per_s_{th}_threads_n256'] = run(256, th, 1, 1)
    best, cores = v32, t32
    for k, v in scan.items():
        if v > best:
            best, cores = v, int(k.split('_')[3])
    out = {'value': best, 'unit': 'rays/s', 'cores': cores, 'kind': 'port', 'cpu': cpu_model(), 'cores_available': avail,
           'rays_per_s_32_threads_n1024': v32, 'rays_per_s_32_threads_n4096': v4096}
    out.update(scan)
    out['all_hardware_threads'] = ('256 threads: 52-62 s per step at 16 and at 64 rays (~1 ray/s), measured once with '
                                   'tools/cpu_threads_probe.py on the round-2 bench box; not repeated per run')
    out['sample'] = (f'SURVEY 8(d): N=1024 rays x (64+128) samples per step, 3 warm-up + 10 timed steps, median, torch CPU fp32 '
                     f'with {t32} threads; one step at N=4096; 1 + 1 steps of 256 rays with 64 and 128 threads '
                     '(oracle/nerf_oracle.py train_step); value = the best thread count')
    return out


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
    ap.add_argument('--sustained-steps', type=int, default=400)
    ap.add_argument('--scene-steps', type=int, default=600, help='untimed optimisation steps on the analytic scene before W + K')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-protocol', choices=['full', 'short'], default='full')
    a = ap.parse_args()

    import fastnerf
    from fastnerf import ops, parallel, synthetic
    from fastnerf.synthetic import pose_spherical
    rank, world, local = parallel.init_from_env('cuda')
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    local = local % max(1, torch.cuda.device_count())   # (only differs in single-GPU plumbing tests of the N>1 path)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    args = fastnerf.run_nerf.make_args(N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, perturb=1.0, white_bkgd=True,
                                       no_reload=True, lrate=5e-4, lrate_decay=500)
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    n_img, max_leaves = 100, 256
    poses = torch.stack([pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(n_img)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000 + rank)
    n_batches = 64          # 262 144 distinct rays per rank, cycled
    batches = []            # (rays_o, rays_d, scene colour, noise colour, leaf tag)
    for _ in range(n_batches):
        pix = torch.stack([torch.randint(0, n_img, (N_RAYS,), generator=gen), torch.randint(0, H, (N_RAYS,), generator=gen),
                           torch.randint(0, W, (N_RAYS,), generator=gen)], 1).int()
        ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
        # (image, leaf) tags of a depth-5 quadtree: 16 x 16 leaves of 50 x 50 pixels, DFS order irrelevant for timing
        tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
        batches.append((ro, rd, {'solid': synthetic.render_rays(ro, rd, cutoff=SCENE_CUTOFF).contiguous(),
                                 'soft': synthetic.render_rays(ro, rd, cutoff=0.0).contiguous(),
                                 'noise': torch.rand(N_RAYS, 3, generator=gen).to(dev)}, tag))
    # silhouette of the solid scene: share of this rank's rays that hit a body (colour differs from the white background)
    coverage = float(torch.cat([(b[2]['solid'] < 0.999).any(-1) for b in batches]).float().mean())
    table = torch.zeros(n_img * max_leaves, device=dev, dtype=torch.int32)
    n_global = N_RAYS * world if world > 1 else None

    def new_trainer():
        torch.manual_seed(0)   # identical initial weights on every rank, and for every leg
        k_train, k_test, _, _, _, _ = fastnerf.run_nerf.create_nerf(args, device=dev)
        return fastnerf.run_nerf.Trainer(k_train, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500), k_test

    n_opt = {}              # optimisation steps applied per trainer (for the PSNR@iterations figure)

    def step(trainer, i, scene='solid'):
        n_opt[id(trainer)] = n_opt.get(id(trainer), 0) + 1
        ro, rd, tgts, tag = batches[i % n_batches]
        return trainer.step(ro, rd, tgts[scene], leaf_tag=tag, table=table, max_leaves=max_leaves, n_global=n_global)

    def timed(trainer, first, warm, steps, scene='solid'):
        """W warm-up + K timed steps, barrier + synchronize on both sides, MAX over ranks -> (seconds, last loss, local s)."""
        for i in range(warm):
            step(trainer, first + i, scene)
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss2, _ = step(trainer, first + warm + i, scene)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0
        parallel.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t[0]), loss2, t_local

    # ---- the solid-body scene: untimed optimisation from random init, then W + K (`value`) ----
    tr, kte = new_trainer()
    for i in range(a.scene_steps):
        step(tr, i)
    dt, loss2, t_local = timed(tr, a.scene_steps, a.warmup, a.steps)
    per_rank_ms = [1e3 * t_local / a.steps]
    allreduce_ms = None
    if world > 1:
        tl = torch.tensor([1e3 * t_local / a.steps], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(gathered, tl)
        per_rank_ms = [float(g[0]) for g in gathered]
        # the step's only data-path collective, timed alone: all-reduce(SUM) of the flat gradient (4.77 MB)
        allreduce_ms = time_launch(lambda: parallel.all_reduce_sum(tr.grad), 20)
    live_frac = None
    if tr.last_step_live:
        c = tr.live_counts.cpu().tolist()
        live_frac = {'fine': c[0] / max(1, c[1]), 'coarse': c[2] / max(1, c[3])}
    backward_kind = 'compacted' if tr.last_step_live else 'plain'

    # ---- sustained leg: the same stream of steps for >= 400 more steps (power-managed clocks settle within seconds) ----
    sustained = None
    if a.sustained_steps > 0:
        ts, _, _ = timed(tr, a.scene_steps + a.warmup + a.steps, 0, a.sustained_steps)
        sustained = {'steps': a.sustained_steps, 'ms_per_step': 1e3 * ts / a.sustained_steps,
                     'value': N_RAYS * world * a.sustained_steps / ts, 'unit': 'rays/s', 'seconds': ts}

    # ---- the same state with the compaction switched off (every point goes through the backward) ----
    old_mode = fastnerf.render.get_compact()
    fastnerf.render.set_compact('0')
    try:
        tp, _, _ = timed(tr, 7, 6, 20)   # (the first plain steps allocate the 11 GB of saved activations: keep them out of the timed 20)
    finally:
        fastnerf.render.set_compact(old_mode)
    steady_plain = {'ms_per_step': 1e3 * tp / 20, 'value': N_RAYS * world * 20 / tp, 'unit': 'rays/s', 'steps': 20, 'warmup': 6,
                    'what': 'FASTNERF_COMPACT=0 on the trained nets: plain backward over every sample'}

    # ---- round-1 protocol: random-init nets, U[0,1) noise targets, no scene (W = 3, K = 20) ----
    tr_i, _ = new_trainer()
    ti, loss_i, _ = timed(tr_i, 0, 3, 20, scene='noise')
    tr_i.live.poll()
    init_state = {'ms_per_step': 1e3 * ti / 20, 'value': N_RAYS * world * 20 / ti, 'unit': 'rays/s', 'steps': 20, 'warmup': 3,
                  'backward': 'compacted' if tr_i.last_step_live else 'plain', 'live_fraction_measured': tr_i.live.frac,
                  'final_loss': [float(x) for x in loss_i.tolist()],
                  'what': 'random-init nets (seed 0), U[0,1) targets: the protocol of BENCH_r01'}
    del tr_i

    # ---- the same blobs with their Gaussian tails (no cut-off): 300 untimed steps, then W = 3, K = 20 ----
    tr_s, _ = new_trainer()
    soft_steps = min(300, a.scene_steps)
    for i in range(soft_steps):
        step(tr_s, i, 'soft')
    tsoft, loss_s, _ = timed(tr_s, soft_steps, 3, 20, scene='soft')
    soft_live = None
    if tr_s.last_step_live:
        c = tr_s.live_counts.cpu().tolist()
        soft_live = {'fine': c[0] / max(1, c[1]), 'coarse': c[2] / max(1, c[3])}
    soft_scene = {'ms_per_step': 1e3 * tsoft / 20, 'value': N_RAYS * world * 20 / tsoft, 'unit': 'rays/s', 'steps': 20, 'warmup': 3,
                  'after_optimisation_steps': soft_steps, 'backward': 'compacted' if tr_s.last_step_live else 'plain', 'live_fraction': soft_live,
                  'final_loss': [float(x) for x in loss_s.tolist()],
                  'what': 'the three blobs WITHOUT the cut-off (density never vanishes), state after %d steps; the live fraction ' % soft_steps +
                          'rises with training on this scene (0.72 at 2000 steps, 0.8-0.9 from 3500 on: tools/live_trajectory.py) '
                          'and the policy then uses the plain backward, so its long-run rate is `steady_state_plain`'}
    del tr_s

    def mlp_roofline(trainer, split, compact_frac):
        """HIP-event timing of the MLP launches of one step's FINE pass (786 432 points); the one the step spends the most
        time in is `roofline`.  achieved = algorithmic FLOPs of the launch / its average duration."""
        ro, rd = batches[0][0], batches[0][1]
        rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
        z = torch.sort(torch.rand(N_RAYS, S1, device=dev) * 4 + 2, -1).values
        P = N_RAYS * S1
        act = torch.empty(ops.act_floats(P), device=dev)
        raw = torch.empty(N_RAYS, S1, 4, device=dev)
        reps = max(3, min(a.steps, 10))
        peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
        base = 'mlp_fwd_bf16_kernel' if split else 'mlp_fwd_kernel'
        rows = []
        ms_save = time_launch(lambda: ops.mlp_fwd(rays11, z, trainer.net_f.flat, trainer.pf[0], act=act, raw=raw), reps)
        ms_inf = time_launch(lambda: ops.mlp_fwd(rays11, z, trainer.net_f.flat, trainer.pf[0], raw=raw), reps)
        rows.append({'kernel': base + '<true, false>', 'what': 'training forward, saves activations, %d points' % P,
                     'points': P, 'avg_launch_ms': ms_save})
        rows.append({'kernel': base + '<false, false>', 'what': 'forward without saving (inference; first pass of a compacted step), %d points' % P,
                     'points': P, 'avg_launch_ms': ms_inf})
        if compact_frac is not None:
            # the compacted step: inference forward on all points + saving forward on the live list
            draw = torch.randn(P, 4, device=dev)
            draw[torch.rand(P, device=dev) >= compact_frac] = 0
            idx, cnt = ops.compact_live(draw)
            k = int(cnt[0])
            ms_live = time_launch(lambda: ops.mlp_fwd_live(rays11, z, trainer.net_f.flat, trainer.pf[0], act, idx, cnt), reps)
            rows.append({'kernel': base + '<true, false>', 'what': 'training forward over the live list, %d of %d points' % (k, P),
                         'points': k, 'avg_launch_ms': ms_live})
            in_step = [rows[1], rows[2]]
        else:
            in_step = [rows[0]]
        for r in rows:
            r['flop_per_launch'] = r['points'] * FWD_FLOP_PER_POINT
            r['achieved'] = r['flop_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e12
            r['frac'] = r['achieved'] / peak
        dom = max(in_step, key=lambda r: r['avg_launch_ms'])
        traffic = step_traffic = None
        try:   # HBM bytes from the committed PMC passes (separate rocprofv3 --pmc runs, tools/collect_profiles.sh)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_pmc_traffic.json')))
            traffic = pmc['kernels']['void ' + dom['kernel']]['hbm_bytes']
            step_traffic = pmc.get('step_traffic')
        except Exception:
            pass
        roof = {'bound': 'mfma', 'kernel': dom['kernel'] + ' (fine pass; ' + dom['what'] + ')', 'achieved': dom['achieved'],
                'peak': peak, 'unit': 'TFLOP/s', 'frac': dom['frac'],
                'peak_note': ('dense bf16 MFMA 2500 TFLOP/s / 3 split terms per product' if split
                              else 'dense fp32 MFMA (v_mfma_f32_32x32x2_f32)'),
                'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC, profiles/%s_pmc_traffic.json)' % PROFILE_ROUND,
                'step_traffic': step_traffic, 'avg_launch_ms': dom['avg_launch_ms'], 'flop_per_launch': dom['flop_per_launch'],
                'mfma_tflops_issued': (3.0 if split else 1.0) * dom['achieved'], 'launches': rows}
        if split:
            # tools/micro/mfma_power.hip on the bench box: a saturated v_mfma_f32_32x32x16_bf16 stream (32.0 clk per
            # MFMA and SIMD) holds 2.24-2.38 GHz on constant operands but only 1.79 GHz = 1871 TFLOP/s on uniform(-1,1)
            # bf16 data (power management)
            roof['peak_measured_real_data'] = 1871.0 / 3.0
            roof['frac_of_measured_peak'] = dom['achieved'] / (1871.0 / 3.0)
        return roof

    roof = None
    if rank == 0:
        roof = mlp_roofline(tr, ops.get_math() == 'bf16x3', live_frac['fine'] if live_frac else None)

    # ---- the same step in the exact-fp32 math mode (1 GPU only; not `value`): its own roofline block ---------------
    alt = None
    if rank == 0 and world == 1 and ops.get_math() != 'fp32':
        main_mode = ops.get_math()
        ops.set_math('fp32')
        try:
            tr32, _ = new_trainer()
            n32 = max(20, a.steps)
            t32, l32, _ = timed(tr32, 0, 3, n32, scene='noise')   # round-1 protocol: random init, noise targets
            dt32 = t32 / n32
            # steady state: the trained state of the main leg (same weights, Adam moments, LR position), W = 3, K = n32
            with torch.no_grad():
                tr32.flat.copy_(tr.flat); tr32.m.copy_(tr.m); tr32.v.copy_(tr.v)
            tr32.adam_t, tr32.global_iter, tr32.lr = tr.adam_t, tr.global_iter, tr.lr
            tr32.repack()
            tr32.live = fastnerf.render.LivePolicy()
            ts32, ls32, _ = timed(tr32, a.scene_steps, 3, n32)
            c32 = tr32.live_counts.cpu().tolist() if tr32.last_step_live else None
            alt = {'math_mode': 'fp32', 'dtype': 'f32', 'value': N_RAYS * n32 / ts32, 'unit': 'rays/s', 'ms_per_step': 1e3 * ts32 / n32,
                   'steps': n32, 'warmup': 3, 'final_loss': [float(x) for x in ls32.tolist()],
                   'what': 'steady state of the trained scene (the main leg\'s weights), exact-fp32 MFMA kernels',
                   'backward': 'compacted' if tr32.last_step_live else 'plain',
                   'live_fraction': None if c32 is None else {'fine': c32[0] / max(1, c32[1]), 'coarse': c32[2] / max(1, c32[3])},
                   'step_frac_of_fp32_mfma_peak': N_RAYS * n32 / ts32 * TRAIN_FLOP_PER_RAY / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                   'init_state': {'value': N_RAYS / dt32, 'ms_per_step': 1e3 * dt32, 'final_loss': [float(x) for x in l32.tolist()],
                                  'step_frac_of_fp32_mfma_peak': N_RAYS / dt32 * TRAIN_FLOP_PER_RAY / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  'what': 'random init, U[0,1) targets, plain backward: the protocol of BENCH_r01'},
                   'roofline': mlp_roofline(tr32, False, c32[0] / max(1, c32[1]) if c32 else None)}
            del tr32
        finally:
            ops.set_math(main_mode)

    # ---- inference rays/s (render_path-style, perturb=0, no saved activations), rank 0 only -------------
    infer = None
    if rank == 0:
        n_inf = 32768
        g2 = torch.Generator().manual_seed(7)
        pix = torch.stack([torch.randint(0, n_img, (n_inf,), generator=g2), torch.randint(0, H, (n_inf,), generator=g2),
                           torch.randint(0, W, (n_inf,), generator=g2)], 1).int().to(dev)
        ro_i, rd_i = ops.gen_rays_pixels(pix, poses, K)
        with torch.no_grad():
            for _ in range(2):
                fastnerf.render.render(H, W, K, chunk=n_inf, rays=(ro_i, rd_i), near=2.0, far=6.0, **kte)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_rep = 5
            for _ in range(n_rep):
                fastnerf.render.render(H, W, K, chunk=n_inf, rays=(ro_i, rd_i), near=2.0, far=6.0, **kte)
            torch.cuda.synchronize()
            dt_i = (time.perf_counter() - t1) / n_rep
            rgb_i = fastnerf.render.render(H, W, K, chunk=n_inf, rays=(ro_i, rd_i), near=2.0, far=6.0, **kte)[0]
            mse_i = float(torch.mean((rgb_i - synthetic.render_rays(ro_i, rd_i, cutoff=SCENE_CUTOFF)) ** 2))
        infer = {'value': n_inf / dt_i, 'unit': 'rays/s', 'rays_per_call': n_inf, 'ms_per_call': 1e3 * dt_i,
                 'what': 'render() of 32768 rays, 64+128 samples, perturb=0 (render_kwargs_test), 1 GPU',
                 # the metric's second half: PSNR of the nets this run trained, on rays that were never in a batch
                 'psnr_db': -10.0 * math.log10(max(mse_i, 1e-12)), 'psnr_after_optimisation_steps': n_opt.get(id(tr), 0),
                 'psnr_what': 'held-out rays of the analytic scene (32768 random pixels of the 100 views, seed 7) against its '
                              'quadrature colours; %d rays per step per GPU' % N_RAYS}

    if rank == 0:
        rays_per_s = N_RAYS * world * a.steps / dt
        step_tflops = rays_per_s * TRAIN_FLOP_PER_RAY / 1e12 / world
        out = {
            'metric': 'training rays/sec (Lego-like 800x800, 64+128 samples)', 'value': rays_per_s, 'unit': 'rays/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (split-bf16 x3 on the bf16 matrix cores, fp32 accumulate)' if ops.get_math() == 'bf16x3' else 'f32',
            'math_mode': ops.get_math(),
            'data': 'synthetic (three solid analytic bodies on white, 100 pose_spherical cameras; nets trained from random init inside the run)',
            'config': {'workload': 'nerf-ours Lego full 800x800, 4096 rays/GPU/step, 64+128 samples, use_viewdirs, '
                                   'white_bkgd, perturb=1, leaf-error table on (BASELINE configs[1]); nets trained on the analytic '
                                   'solid-body scene (%d untimed optimisation steps from random init, then W + K); the share of '
                                   'live samples is stationary on this scene from step ~500 on' % a.scene_steps,
                       'rays_per_gpu_per_step': N_RAYS, 'parallelism': f'dp{world}', 'scene_steps': a.scene_steps},
            'scene': {'bodies': 'three blobs of fastnerf.synthetic, density exactly zero beyond %.1f standard deviations of each centre' % SCENE_CUTOFF,
                      'silhouette_coverage': coverage,
                      'note': 'throughput depends on the share of samples with an exactly-zero gradient (sigma <= 0), a property of the '
                              'scene and of the training trajectory: this scene (coverage like the Lego bulldozer, empty space around '
                              'it) keeps 0.19 fine / 0.08 coarse live from step ~500 through 6000 (tools/live_trajectory.py); '
                              '`gaussian_tails_scene` and `steady_state_plain` are the adverse cases'},
            'device': {'name': torch.cuda.get_device_name(dev), 'compute_units': torch.cuda.get_device_properties(dev).multi_processor_count,
                       'note': 'the chip is power-managed under these kernels (DESIGN section 9): the same tree measured 720 k .. 757 k rays/s on '
                               'different boxes of the pool (one box of an earlier session ran every leg 8 % slower), every leg moving together'},
            'final_loss': [float(x) for x in loss2.tolist()],
            'backward': ('compacted: samples with an exactly-zero gradient skipped (FASTNERF_COMPACT=%s)' % fastnerf.render.get_compact())
            if backward_kind == 'compacted' else 'plain (every sample)',
            'live_fraction': live_frac,
            'steady_state_plain': steady_plain, 'speedup_vs_plain_backward': steady_plain['ms_per_step'] / (1e3 * dt / a.steps),
            'gaussian_tails_scene': soft_scene,
            'init_state': init_state,
            'per_rank_ms_per_step': per_rank_ms, 'allreduce_ms': allreduce_ms,
            'step_tflops_note': 'rays/s x the algorithmic FLOPs of an UNCOMPACTED step (893.2 MFLOP/ray): with samples skipped this is '
                                'an effective rate, not what the matrix cores executed -- see `roofline` for executed work per launch',
            'step_tflops_per_gpu': step_tflops, 'step_frac_of_fp32_mfma_peak': step_tflops / FP32_MFMA_PEAK_TFLOPS,
            'step_frac_of_bf16_mfma_peak_x3': 3.0 * step_tflops / BF16_MFMA_PEAK_TFLOPS,
            'sustained': sustained,
            'roofline': roof,
            'inference': infer,
            'exact_fp32_mode': alt,
            'cpu_baseline': None if (a.no_cpu_baseline or world > 1) else cpu_baseline(a.cpu_protocol),
        }
        print(json.dumps(out))
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
