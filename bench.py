#!/usr/bin/env python3
"""bench.py -- training rays/s of the MI355X-native NeRF inner loop (BASELINE.json metric).

A "step" = one full optimisation step (ray packing -> stratified + hierarchical sampling -> PE ->
coarse/fine MLP -> compositing -> 2xMSE + leaf-error table -> backward -> [RCCL all-reduce] ->
Adam + LR decay) on 4096 rays x (64 + 128) samples per GPU, i.e. BASELINE.json configs[1]
("nerf-ours Lego full 800x800, 4096 rays, 64+128 samples"), synthetic Lego-like cameras
(100 x pose_spherical, 800x800, focal 1111.11, near 2 / far 6), U[0,1) targets, default-init
weights (seed 0).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS, N_SAMPLES, N_IMPORTANCE = 4096, 64, 128
MAC_PER_POINT = 593408
FWD_FLOP_PER_POINT = 2 * MAC_PER_POINT     # 1.186816 MFLOP
TRAIN_FLOP_PER_RAY = 2 * (3 * MAC_PER_POINT - 35712) * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)  # 893.2 MFLOP
FP32_MFMA_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0             # MI355X_MICROARCH.md: dense bf16 matrix peak


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (a port of the reference's step, validated against it by tests/) timed on the
    host cores of this box on a bounded sample of the same workload: full 64+128 samples per ray,
    fewer rays per step."""
    from oracle import nerf_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))   # torch's CPU GEMMs stop scaling (and start thrashing) beyond this
    torch.set_num_threads(cores)
    n = 256
    gen = torch.Generator().manual_seed(0)
    sdc, sdf = O.init_nerf_params(gen), O.init_nerf_params(gen)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    K = O.intrinsics(800, 800, 1111.111)
    ro, rd = O.get_rays(800, 800, K, c2w)
    sel = torch.randint(0, 640000, (n,), generator=gen)
    rb = O.make_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], 2.0, 6.0)
    tgt = torch.rand(n, 3, generator=gen)
    times = []
    t_start = time.time()
    for it in range(16):   # ~0.8 s per step on 32 cores: 10-15 s of CPU work, bounded by seconds_budget
        t_rand, u = torch.rand(n, N_SAMPLES, generator=gen), torch.rand(n, N_IMPORTANCE, generator=gen)
        t0 = time.time()
        O.train_step(sdc, sdf, opt, rb, tgt, N_SAMPLES, N_IMPORTANCE, True, t_rand=t_rand, u=u)
        times.append(time.time() - t0)
        if it >= 1 and time.time() - t_start > seconds_budget:
            break
    med = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {'value': n / med, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{len(times) - 1} timed steps of {n} rays x (64+128) samples, torch CPU fp32, '
                      f'{cores} threads of {avail} available (oracle/nerf_oracle.py train_step)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    import fastnerf
    from fastnerf import ops, parallel
    from oracle import nerf_oracle as O
    rank, world, local = parallel.init_from_env('cuda')
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    local = local % max(1, torch.cuda.device_count())   # (only differs in single-GPU plumbing tests of the N>1 path)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    torch.manual_seed(0)   # identical initial weights on every rank
    args = fastnerf.run_nerf.make_args(N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, perturb=1.0, white_bkgd=True,
                                       no_reload=True, lrate=5e-4, lrate_decay=500)
    ktr, kte, _, _, _, _ = fastnerf.run_nerf.create_nerf(args, device=dev)
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([O.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000 + rank)
    n_batches = 8
    batches = []
    for _ in range(n_batches):
        pix = torch.stack([torch.randint(0, 100, (N_RAYS,), generator=gen), torch.randint(0, H, (N_RAYS,), generator=gen),
                           torch.randint(0, W, (N_RAYS,), generator=gen)], 1).int().to(dev)
        ro, rd = ops.gen_rays_pixels(pix, poses, K)
        batches.append((ro, rd, torch.rand(N_RAYS, 3, generator=gen).to(dev)))
    tr = fastnerf.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    n_global = N_RAYS * world if world > 1 else None

    def step(i):
        ro, rd, tgt = batches[i % n_batches]
        return tr.step(ro, rd, tgt, n_global=n_global)

    for i in range(a.warmup):
        step(i)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss2, _ = step(a.warmup + i)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t[0])

    # ---- dominant kernel, measured live with HIP events on the launch stream ----------------
    # mlp_fwd_kernel<SAVE=true> over the fine pass: P = 4096*192 points in one launch
    roof = None
    if rank == 0:
        ro, rd, tgt = batches[0]
        rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
        z = torch.sort(torch.rand(N_RAYS, N_SAMPLES + N_IMPORTANCE, device=dev) * 4 + 2, -1).values
        P = N_RAYS * (N_SAMPLES + N_IMPORTANCE)
        act = torch.empty(ops.act_floats(P), device=dev)
        raw = torch.empty(N_RAYS, N_SAMPLES + N_IMPORTANCE, 4, device=dev)
        for _ in range(2):
            ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw)
        reps = max(3, min(a.steps, 10))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = P * FWD_FLOP_PER_POINT
        achieved = flops / (ms * 1e-3) / 1e12
        split = ops.get_math() == 'bf16x3'
        kname = 'void mlp_fwd_bf16_kernel<true, false>' if split else 'void mlp_fwd_kernel<true, false>'
        traffic = None   # HBM bytes per launch from the committed PMC passes (separate rocprofv3 runs)
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
            traffic = pmc['kernels'][kname]['hbm_bytes']
        except Exception:
            pass
        if split:
            # every algorithmic multiply-add is issued as 3 bf16 MFMA terms (hi*hi + hi*lo + lo*hi), so the roofline of
            # the algorithmic FLOPs is the dense bf16 MFMA peak / 3
            peak = BF16_MFMA_PEAK_TFLOPS / 3.0
        else:
            peak = FP32_MFMA_PEAK_TFLOPS
        roof = {'bound': 'mfma', 'kernel': kname[5:] + ' (fine pass, 786432 points/launch)',
                'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                'peak_note': ('dense bf16 MFMA 2500 TFLOP/s / 3 split terms per product' if split
                              else 'dense fp32 MFMA (v_mfma_f32_32x32x2_f32)'),
                'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC, profiles/r01_pmc_traffic.json)',
                'avg_launch_ms': ms, 'flop_per_launch': flops, 'mfma_tflops_issued': (3.0 if split else 1.0) * achieved,
                'hbm_write_GBps': (ops.act_floats(P) * 4 / (ms * 1e-3) / 1e9)}
        if split:
            # tools/micro/mfma_power.hip on the bench box: a saturated v_mfma_f32_32x32x16_bf16 stream (32.0 clk per
            # MFMA and SIMD) holds 2.24-2.38 GHz on constant operands but only 1.79 GHz = 1871 TFLOP/s on uniform(-1,1)
            # bf16 data (power management); the kernel itself runs at 1.67 GHz (profiles/r01_sq_counters.md)
            roof['peak_measured_real_data'] = 1871.0 / 3.0
            roof['frac_of_measured_peak'] = achieved / (1871.0 / 3.0)
        del act

    # ---- the same step in the exact-fp32 math mode, for reference (1 GPU only; not `value`) ----------------------
    alt = None
    if rank == 0 and world == 1 and ops.get_math() != 'fp32':
        main_mode = ops.get_math()
        ops.set_math('fp32')
        try:
            torch.manual_seed(0)
            ktr32, _, _, _, _, _ = fastnerf.run_nerf.create_nerf(args, device=dev)
            tr32 = fastnerf.run_nerf.Trainer(ktr32, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
            for i in range(2):
                tr32.step(*batches[i % n_batches])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n32 = 5
            for i in range(n32):
                l32, _ = tr32.step(*batches[(2 + i) % n_batches])
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / n32
            alt = {'math_mode': 'fp32', 'dtype': 'f32', 'value': N_RAYS / dt32, 'unit': 'rays/s', 'ms_per_step': 1e3 * dt32,
                   'steps': n32, 'frac_of_fp32_mfma_peak': N_RAYS / dt32 * TRAIN_FLOP_PER_RAY / 1e12 / FP32_MFMA_PEAK_TFLOPS}
            del tr32, ktr32
        finally:
            ops.set_math(main_mode)

    # ---- inference rays/s (render_path-style, perturb=0, no saved activations), rank 0 only -------------
    infer = None
    if rank == 0:
        n_inf = 32768
        g2 = torch.Generator().manual_seed(7)
        pix = torch.stack([torch.randint(0, 100, (n_inf,), generator=g2), torch.randint(0, H, (n_inf,), generator=g2),
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
        infer = {'value': n_inf / dt_i, 'unit': 'rays/s', 'rays_per_call': n_inf, 'ms_per_call': 1e3 * dt_i,
                 'what': 'render() of 32768 rays, 64+128 samples, perturb=0 (render_kwargs_test), 1 GPU'}

    if rank == 0:
        rays_per_s = N_RAYS * world * a.steps / dt
        step_tflops = rays_per_s * TRAIN_FLOP_PER_RAY / 1e12 / world
        out = {
            'metric': 'training rays/sec (Lego-like 800x800, 64+128 samples)', 'value': rays_per_s, 'unit': 'rays/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (split-bf16 x3 on the bf16 matrix cores, fp32 accumulate)' if ops.get_math() == 'bf16x3' else 'f32',
            'math_mode': ops.get_math(), 'data': 'synthetic',
            'config': {'workload': 'nerf-ours Lego full 800x800, 4096 rays/GPU/step, 64+128 samples, use_viewdirs, '
                                   'white_bkgd, perturb=1 (BASELINE configs[1])',
                       'rays_per_gpu_per_step': N_RAYS, 'parallelism': f'dp{world}'},
            'final_loss': [float(x) for x in loss2.tolist()],
            'step_tflops_per_gpu': step_tflops, 'step_frac_of_fp32_mfma_peak': step_tflops / FP32_MFMA_PEAK_TFLOPS,
            'step_frac_of_bf16_mfma_peak_x3': 3.0 * step_tflops / BF16_MFMA_PEAK_TFLOPS,
            'roofline': roof,
            'inference': infer,
            'exact_fp32_mode': alt,
            'cpu_baseline': None if (a.no_cpu_baseline or world > 1) else cpu_baseline(),
        }
        print(json.dumps(out))
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
