#!/usr/bin/env python3
"""G21: the layout of the checkpoints nerf++-ours writes (ddp_train_nerf.py:306-314: `model_{epoch:04d}.pth` = net_<m> / optim_<m>
state_dicts), recorded from the REFERENCE itself (build container only): names and shapes of every tensor of
nn.DataParallel(NerfNetWithAutoExpo).state_dict(), and the structure of torch.optim.Adam(net.parameters()).state_dict() after one
step -- no weights (a real checkpoint is 29 MB).  Data only -> tests/golden/g21_pp_ckpt_layout.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402

REF = '/root/reference/nerf++-ours'


class Args:
    netdepth = 8; netwidth = 256; max_freq_log2 = 10; max_freq_log2_viewdirs = 4; use_viewdirs = True
    batch_size = 64; lrate = 5e-4; lambda_autoexpo = 1.0; optim_autoexpo = False


def main():
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)
    import ddp_model as M
    torch.manual_seed(777)
    net = torch.nn.DataParallel(M.NerfNetWithAutoExpo(Args, optim_autoexpo=False))
    opt = torch.optim.Adam(net.parameters(), lr=Args.lrate)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    sd, osd = net.state_dict(), opt.state_dict()
    names = list(sd.keys())
    out = {'names': np.array(names), 'shapes': np.array([';'.join(str(d) for d in sd[k].shape) for k in names]),
           'optim_group_keys': np.array(sorted(osd['param_groups'][0].keys())),
           'optim_params': np.array(osd['param_groups'][0]['params'], dtype=np.int64),
           'optim_state_keys': np.array(sorted(osd['state'][0].keys())),
           'optim_state_len': np.int64(len(osd['state'])),
           # the first weights under torch.manual_seed(777): pins the construction order (same seed -> same init here)
           'first_weight_head': sd[names[0]].reshape(-1)[:16].numpy().copy(),
           'last_bias': sd[names[-1]].reshape(-1).numpy().copy()}
    np.savez(os.path.join(OUT, 'g21_pp_ckpt_layout.npz'), **out)
    print('wrote g21', len(names), 'tensors')


if __name__ == '__main__':
    main()
