#!/usr/bin/env python3
"""Differential check of the nerf++ oracle (oracle/nerfpp_oracle.py cascade_step) against the REFERENCE's own train_step
(nerf++-ours/ddp_train_nerf.py:327-424) over random cascade sample counts and ray batches (build container only: needs
/root/reference; nothing here travels or is imported by tests).  Same weights, same torch.manual_seed draw stream: the rendered
colour of the last level and every gradient of both levels must agree to fp32 round-off.  Exit code 1 on mismatch.

Run:  python oracle/fuzz_pp_vs_reference.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from make_golden import install_stubs  # noqa: E402
from make_golden_pp import REF, Args  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)
    import ddp_train_nerf as D
    import ddp_model as M
    from oracle import nerfpp_oracle as PP
    rng = np.random.RandomState(5)
    bad = 0
    for ci in range(n_cases):
        S0 = int(rng.choice([3, 8, 17, 32, 64]))
        S1 = int(rng.choice([1, 5, 16, 33, 64, 128]))
        N = int(rng.choice([5, 12, 24]))
        g = torch.Generator().manual_seed(ci)
        torch.manual_seed(10 + ci)
        nets = [M.NerfNetWithAutoExpo(Args, optim_autoexpo=False) for _ in range(2)]
        optims = [torch.optim.Adam(nt.parameters(), lr=Args.lrate) for nt in nets]
        sds = [({k[len('fg_net.'):]: v.detach().clone() for k, v in nt.nerf_net.state_dict().items() if k.startswith('fg_net.')},
                {k[len('bg_net.'):]: v.detach().clone() for k, v in nt.nerf_net.state_dict().items() if k.startswith('bg_net.')})
               for nt in nets]
        ro = (torch.rand(N, 3, generator=g) - 0.5) * 0.9
        rd = torch.randn(N, 3, generator=g)
        tgt = torch.rand(N, 3, generator=g)
        models = {'cascade_level': 2, 'cascade_samples': [S0, S1], 'net_0': nets[0], 'net_1': nets[1], 'optim_0': optims[0],
                  'optim_1': optims[1]}
        Args.batch_size = N
        torch.manual_seed(7)
        try:
            rgb_pred = D.train_step(models, ro, rd, tgt, Args)
        except Exception as e:
            print(f'case {ci} S=({S0},{S1}) N={N}: reference raises {type(e).__name__}: skipped')
            continue
        torch.manual_seed(7)
        rand = [{'fg_t': torch.rand(N, S0), 'bg_t': torch.rand(N, S0)}, {'fg_u': torch.rand(N, S1), 'bg_u': torch.rand(N, S1)}]
        outs = PP.cascade_step(sds, ro, rd, tgt, [S0, S1], rand)
        e_rgb = float((outs[-1][2]['rgb'] - rgb_pred).abs().max())
        gw = 0.0
        for m, nt in enumerate(nets):
            ref_g = [p.grad for _, p in nt.nerf_net.named_parameters()]
            for a, b in zip(outs[m][1], ref_g):
                gw = max(gw, float((a - b).abs().max() / max(float(b.abs().max()), 1e-8)))
        ok = e_rgb < 2e-6 and gw < 2e-4
        bad += not ok
        print(f'case {ci} S=({S0},{S1}) N={N}: rgb {e_rgb:.1e} grads {gw:.1e} ' + ('OK' if ok else 'MISMATCH'))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
