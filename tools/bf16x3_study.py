"""Accuracy study (CPU, oracle only): what would a split-bf16 MLP do to the rendered RGB?
Emulates every linear layer's GEMM with operands split into bf16 pieces and fp32 accumulation:
  bf16x1: a_hi*b_hi                      (plain bf16 MFMA)
  bf16x3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
  bf16x6: 3-way split (hi, mid, lo), all products down to 2^-24
Reports max |dRGB| vs the fp32 oracle on the G7 inputs (64 rays, 64+128 samples)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import nerf_oracle as O
torch.set_num_threads(8)

def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p); r = r - p
    return parts

def make_linear(mode):
    def lin(x, w, b):
        if mode == 'fp32':
            return torch.nn.functional.linear(x, w, b)
        n = {'bf16x1': 1, 'bf16x3': 2, 'bf16x6': 3}[mode]
        xs, ws = split(x, n), split(w, n)
        out = torch.zeros(x.shape[:-1] + (w.shape[0],))
        for i in range(n):
            for j in range(n):
                if i + j < n:          # keep products down to the smallest retained magnitude
                    out = out + xs[i] @ ws[j].t()
        return out + b
    return lin

g = np.load('/root/repo/tests/golden/g7_render.npz'); w = np.load('/root/repo/tests/golden/g7_weights.npz')
T = lambda a: torch.from_numpy(np.asarray(a))
sdc = {k[2:]: T(w[k]) for k in w.files if k.startswith('c.')}; sdf = {k[2:]: T(w[k]) for k in w.files if k.startswith('f.')}
rb = O.make_ray_batch(T(g['ro']), T(g['rd']), 2.0, 6.0)
ref = None
orig = torch.nn.functional.linear
for mode in ('fp32', 'bf16x1', 'bf16x3', 'bf16x6'):
    lin = make_linear(mode)
    import oracle.nerf_oracle as M
    class F: pass
    M.torch.nn.functional.linear = lin if mode != 'fp32' else orig
    with torch.no_grad():
        r = O.render_rays(rb, sdc, sdf, 64, 128, white_bkgd=True, t_rand=T(g['t_rand']), u=T(g['u']))
    M.torch.nn.functional.linear = orig
    if ref is None: ref = r
    print('%-7s max|dRGB fine| %.3e  max|dRGB coarse| %.3e  max|dz_samples| %.3e' % (
        mode, (r['rgb_map'] - ref['rgb_map']).abs().max(), (r['rgb0'] - ref['rgb0']).abs().max(),
        (r['z_samples'] - ref['z_samples']).abs().max()))
