"""Drop-in check of INTEGRATION.md option A, run as a script by tests/test_gpu_reference_loop.py (start, then fresh-process resumes).

What it proves: a caller that uses the package the way nerf-ours' driver uses its own modules -- create_nerf's 6-tuple and
render_kwargs, `render(..., retraw=True)`, `img2mse`, `loss.backward()`, a torch.optim.Adam over grad_vars with the decayed learning
rate, QuadTreeManager.gen_rays_v3_multiThread / adjust_tree_multiThread, `{epoch:03d}.tar` + `treeDivide_{epoch:04d}.pkl` -- gets a
working, resumable training run, and that the files it writes and the files the fused `train()` writes are interchangeable.

Layout of this script (its own, not the reference's): a `Run` object holds the state; `Run.fit` is the only place that
optimises (one list of (rays_o, rays_d, colours) in, per-batch losses and predictions out); `PLAN` below is the table of
stages a fresh or resumed run goes through.  Reference lines the stages answer to: checkpoint discovery run_nerf.py:109-127,
tree reload :337-345, centre-crop warm-up :367-423 (the pick count is clamped to the crop: the test images are smaller than
N_rand * 500 / n_images, where the reference's draw without replacement would raise), epoch loop :436-530, saves :532-544.
Prints one `LOOPLOG {json}` line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastnerf                                                           # noqa: E402
from fastnerf.render import render                                        # noqa: E402
from fastnerf.run_nerf_helpers import img2mse, mse2psnr                   # noqa: E402
from fastnerf.tree import QuadTreeManager, get_children                   # noqa: E402

H = W = 32
NEAR, FAR = 2., 6.
DEV = torch.device('cuda')


def slices(n, size):
    return [slice(lo, min(lo + size, n)) for lo in range(0, n, size)]


class Run:
    def __init__(self, cli):
        self.args = a = fastnerf.run_nerf.make_args(
            N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, N_rand=256, n_epoch=cli.n_epoch, init_level=2,
            subdivide_every=1, subdivide_thres=0.05, lrate=5e-4, lrate_decay=500, basedir=cli.basedir, expname='loop', no_reload=False)
        self.dir = os.path.join(a.basedir, a.expname)
        os.makedirs(self.dir, exist_ok=True)
        images, poses, focal = fastnerf.synthetic.make_dataset(n_images=4, H=H, W=W)
        self.K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        # the newest .tar of basedir/expname, if any, is loaded by create_nerf itself (nets strict, Adam state)
        self.kw_train, self.kw_test, self.epoch, self.iters, self.grad_vars, self.opt = fastnerf.run_nerf.create_nerf(a, device=DEV)
        for kw in (self.kw_train, self.kw_test):
            kw.update(near=NEAR, far=FAR)
        self.mgr = QuadTreeManager(H, W, self.K, images, poses, mseThres=0.0, max_depth=a.init_level)
        self.log = {'resumed_from': self.epoch, 'iter_at_start': self.iters, 'epochs': []}

    # ---- the one optimisation routine ------------------------------------------------------------------------------
    def fit(self, rays_o, rays_d, colours):
        """Every batch of N_rand rays once: render -> fine + coarse mse -> backward -> Adam -> learning-rate rule.
        -> (losses, [gt], [pred]) of the fine pass, on the host."""
        a, losses, gts, preds = self.args, [], [], []
        for s in slices(rays_o.shape[0], a.N_rand):
            gt = colours[s].to(DEV)
            rgb, _, _, extras = render(H, W, self.K, chunk=a.chunk, rays=torch.stack([rays_o[s].to(DEV), rays_d[s].to(DEV)], 0),
                                       retraw=True, **self.kw_train)
            assert extras['raw'].shape[-1] == 4 and 'rgb0' in extras
            fine, coarse = img2mse(rgb, gt), img2mse(extras['rgb0'], gt)
            self.opt.zero_grad()
            (fine + coarse).backward()
            self.opt.step()
            for group in self.opt.param_groups:                      # rate for the NEXT step, from the count BEFORE its increment
                group['lr'] = a.lrate * 0.1 ** (self.iters / (a.lrate_decay * 1000))
            self.iters += 1
            losses.append(float(fine))
            gts.append(gt.detach().cpu())
            preds.append(rgb.detach().cpu())
        return losses, gts, preds

    # ---- stages ----------------------------------------------------------------------------------------------------
    def restore_trees(self):
        path = fastnerf.run_nerf.tree_pkl_path(self.args, self.epoch)
        self.log['loaded_tree'] = os.path.exists(path)
        if self.log['loaded_tree']:
            self.mgr.load_trees(path)
            # the assignments a reference-shaped caller makes after unpickling go through the manager's settable views
            self.mgr.quadTrees = self.mgr.quadTrees
            self.mgr.childrens = [get_children(t.root) for t in self.mgr.quadTrees]
            self.mgr.cur_level = self.epoch
        self.log['leaves_at_start'] = self.leaves()

    def warmup(self):
        if self.epoch != 0:
            return                                                    # a resumed run does not repeat it
        rows = torch.arange(H // 2 - H // 4, H // 2 + H // 4)
        cols = torch.arange(W // 2 - W // 4, W // 2 + W // 4)
        grid = torch.cartesian_prod(rows, cols)
        take = min(int(self.args.N_rand * 500 / self.mgr.n_images), grid.shape[0])
        pick = grid[torch.from_numpy(np.random.choice(grid.shape[0], size=[take], replace=False))]
        r, c = pick[:, 0], pick[:, 1]
        per_image = [(self.mgr.origins[i][r, c], self.mgr.dirs[i][r, c], self.mgr.images[i][r, c]) for i in range(self.mgr.n_images)]
        losses, _, _ = self.fit(*(torch.cat(parts, 0) for parts in zip(*per_image)))
        self.log['warmup_loss'] = losses[-1]

    def one_epoch(self, e):
        a, m = self.args, self.mgr
        final = e == a.n_epoch
        if final:
            m.epoch_size = m.n_images * m.h * m.w
        rays = m.gen_rays_v3_multiThread(down_scale=1, prob=False, last_epoch=final) if final else \
            m.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=a.randSamp_perc, last_epoch=False)
        losses, gts, preds = self.fit(*rays)
        before = sum(len(c) for c in m.childrens)
        if a.subdivide_every > 0 and e % a.subdivide_every == 0 and e < a.n_epoch - 1:
            m.adjust_tree_multiThread(torch.cat(gts, 0), torch.cat(preds, 0), thres=a.subdivide_thres, debug=False)
        self.epoch = e
        self.save()
        self.log['epochs'].append({
            'epoch': e, 'iters': len(losses), 'rays': int(rays[0].shape[0]), 'loss_first': losses[0],
            'loss_last': float(np.mean(losses[-3:])), 'leaves_before': before, 'leaves_after': sum(len(c) for c in m.childrens),
            'cur_level': m.cur_level, 'lr': self.opt.param_groups[0]['lr'], 'psnr': float(mse2psnr(torch.tensor(losses[-1])))})

    def save(self):
        ref_names = fastnerf.run_nerf.reference_state_dict          # the `module.` prefix the reference's strict load expects
        torch.save({'global_epoch': self.epoch, 'global_iter': self.iters,
                    'network_fn_state_dict': ref_names(self.kw_train['network_fn']),
                    'network_fine_state_dict': ref_names(self.kw_train['network_fine']),
                    'optimizer_state_dict': self.opt.state_dict()}, os.path.join(self.dir, '{:03d}.tar'.format(self.epoch)))
        self.mgr.save_trees(fastnerf.run_nerf.tree_pkl_path(self.args, self.epoch))

    def leaves(self):
        return [self.mgr.leaves(i).tolist() for i in range(self.mgr.n_images)]


PLAN = ('restore_trees', 'warmup', 'epochs')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--basedir', required=True)
    ap.add_argument('--stop_after', type=int, default=0)
    ap.add_argument('--n_epoch', type=int, default=4)
    ap.add_argument('--seed', type=int, default=0)
    cli = ap.parse_args()
    torch.manual_seed(cli.seed)
    np.random.seed(cli.seed)
    run = Run(cli)
    for stage in PLAN:
        if stage != 'epochs':
            getattr(run, stage)()
            continue
        for e in range(run.epoch + 1, run.args.n_epoch + 1):
            run.one_epoch(e)
            if cli.stop_after and e >= cli.stop_after:
                break
    run.log.update(leaves_at_end=run.leaves(), global_iter=run.iters,
                   adam_step=int(float(run.opt.state_dict()['state'][0]['step'])))
    print('LOOPLOG ' + json.dumps(run.log))


if __name__ == '__main__':
    main()
