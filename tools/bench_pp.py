"""nerf++ (BASELINE configs[4]) cascade step throughput on one GPU, every math mode (or those named after N): 2 levels x (fg + bg) nets,
64 / 128 samples, synthetic cameras inside the unit sphere.  Not the headline bench (bench.py is); DESIGN.md cites it."""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf as fn
from fastnerf import ops, nerfpp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
dev = torch.device('cuda')
for mode in (sys.argv[2:] or ['bf16x6', 'fp32', 'bf16x3']):
    ops.set_math(mode)
    torch.manual_seed(0)
    nets = [nerfpp.NerfNet(device=dev) for _ in range(2)]
    tr = nerfpp.CascadeTrainer(nets, cascade_samples=(64, 128))
    g = torch.Generator().manual_seed(1)
    ro = ((torch.rand(N, 3, generator=g) - 0.5) * 0.6).to(dev)
    rd = torch.randn(N, 3, generator=g).to(dev)
    tgt = torch.rand(N, 3, generator=g).to(dev)
    for _ in range(3): tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): loss, _ = tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print('%-7s %d rays: %.2f ms/step  %.0f rays/s  loss %s' % (mode, N, dt * 1e3, N / dt, [round(float(x), 5) for x in loss]))
