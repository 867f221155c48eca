"""Is there a systematic GPU-vs-CPU shift in PSNR@200 iterations, or only chaotic scatter?  Small batches (128 rays) so that the
CPU side is affordable as an ENSEMBLE too.
  gpu : (on the GPU box) builds the inputs, saves them to gpurun_out/psnr_small_inputs.pt, runs 16 ulp-jittered fp32 runs + bf16x3.
  cpu K: (anywhere) runs CPU ensemble member K (K = 0: unjittered) on the saved inputs."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import bench as B
B.PSNR_RAYS = 128
B.PSNR_HELD_OUT = 1024
INP = os.path.join(ROOT, 'gpurun_out', 'psnr_small_inputs.pt')
if sys.argv[1] == 'gpu':
    import fastnerf as fn
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import conftest
    args, poses, K, draw_pixels, new_trainer = conftest.psnr_protocol(fn)
    dev = torch.device('cuda')
    data = B.psnr_inputs(fn, dev, 200, args, poses, K, draw_pixels)
    torch.save(data, INP)
    dd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
    res = {'fp32': [B.psnr_gpu_free(fn, dd, new_trainer, K, 'fp32', jitter_ulp_seed=None if j == 0 else 100 + j)[0] for j in range(16)],
           'bf16x3': [B.psnr_gpu_free(fn, dd, new_trainer, K, 'bf16x3', jitter_ulp_seed=None if j == 0 else 100 + j)[0] for j in range(16)]}
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'psnr_small_gpu.json'), 'w'))
    for k, v in res.items():
        for key in ('train_psnr_db', 'held_out_psnr_db'):
            x = np.array([e[key] for e in v]); print(k, key, 'mean %.3f std %.3f' % (x.mean(), x.std(ddof=1)), np.round(x, 2).tolist())
else:
    j = int(sys.argv[2])
    d = torch.load(INP)
    if j > 0:
        g = torch.Generator().manual_seed(500 + j)
        for sd in (d['sdc'], d['sdf']):
            for k in sd:
                bits = sd[k].view(torch.int32)
                bits += torch.randint(-1, 2, bits.shape, generator=g, dtype=torch.int32)
    tmp_in, out = '/tmp/psnr_small_in_%d.pt' % j, '/tmp/psnr_small_out_%d.json' % j
    torch.save(d, tmp_in)
    B.psnr_cpu_worker(tmp_in, out)
    r = json.load(open(out))
    print(j, 'train %.3f held-out %.3f' % (B.psnr_of(r['losses'], 20), -10 * np.log10(r['held_out_mse'])), flush=True)
    os.remove(tmp_in); os.remove(out + '.states')
