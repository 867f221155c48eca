"""The reference's training loop SHAPE on the product imports (INTEGRATION.md option A) -- run as a script by
tests/test_gpu_reference_loop.py, twice (start, fresh-process resume).

The body restates, statement by statement, what nerf-ours/run_nerf.py does around the hot path:
  :76-99   create_nerf's model wrapping (nn.DataParallel) + torch.optim.Adam over grad_vars
  :109-127 checkpoint discovery / reload
  :337-345 QuadTreeManager(...) + treeDivide_{global_epoch:04d}.pkl reload through `quadTrees`, `get_children`, `cur_level`
  :367-423 center-crop warm-up (the coordinate count is clamped to the crop: the tiny test images are smaller than
           N_rand*500/n_images, where the reference's np.random.choice(replace=False) would raise)
  :436-530 epoch loop: gen_rays_v3_multiThread, render(...), img2mse x2, loss.backward(), optimizer.step(), LR rule,
           rgb_gt_collect / rgb_pred_collect, adjust_tree_multiThread
  :532-544 torch.save of the .tar + pickle.dump(treeManager.quadTrees) -- here through the manager's save_trees, which
           writes the same `tree.QuadTree` pickle the reference's dump produces
Prints one JSON line with what the test asserts on."""
import argparse
import json
import os
import pickle
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastnerf                                                   # noqa: E402
from fastnerf.run_nerf_helpers import *                           # noqa: E402,F401,F403
from fastnerf.model import NeRF                                   # noqa: E402
from fastnerf.render import render_rays, render_path, render      # noqa: E402,F401
from fastnerf.tree import QuadTreeManager, get_children           # noqa: E402

device = torch.device('cuda')


def create_nerf(args):
    """run_nerf.py:67-153 (the reference's own function body shape, with the product NeRF)."""
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    model = nn.DataParallel(model, device_ids=[0])
    grad_vars = list(model.parameters())
    model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch, skips=skips,
                      input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    model_fine = nn.DataParallel(model_fine, device_ids=[0])
    grad_vars += list(model_fine.parameters())
    network_query_fn = lambda inputs, viewdirs, network_fn: fastnerf.run_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=args.netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start_epoch, start_iter = 0, 0
    basedir, expname = args.basedir, args.expname
    os.makedirs(os.path.join(basedir, expname), exist_ok=True)
    ckpts = [os.path.join(basedir, expname, f) for f in sorted(os.listdir(os.path.join(basedir, expname))) if 'tar' in f]
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], weights_only=False)
        start_epoch, start_iter = ckpt['global_epoch'], ckpt['global_iter']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])            # strict, DataParallel-prefixed keys
        model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    render_kwargs_train = {'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
                           'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
                           'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd,
                           'raw_noise_std': args.raw_noise_std, 'ndc': False, 'lindisp': args.lindisp}
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start_epoch, start_iter, grad_vars, optimizer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--basedir', required=True)
    ap.add_argument('--stop_after', type=int, default=0)
    ap.add_argument('--n_epoch', type=int, default=4)
    ap.add_argument('--seed', type=int, default=0)
    cli = ap.parse_args()
    args = fastnerf.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, N_rand=256,
                                       n_epoch=cli.n_epoch, init_level=2, subdivide_every=1, subdivide_thres=0.05,
                                       lrate=5e-4, lrate_decay=500, basedir=cli.basedir, expname='loop', no_reload=False)
    torch.manual_seed(cli.seed)
    np.random.seed(cli.seed)
    images, poses, focal = fastnerf.synthetic.make_dataset(n_images=4, H=32, W=32)
    H = W = 32
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    near, far = 2., 6.
    render_kwargs_train, render_kwargs_test, global_epoch, global_iter, grad_vars, optimizer = create_nerf(args)
    resumed_from = global_epoch
    bds_dict = {'near': near, 'far': far}
    render_kwargs_train.update(bds_dict)
    render_kwargs_test.update(bds_dict)
    N_rand = args.N_rand
    basedir, expname = args.basedir, args.expname

    treeManager = QuadTreeManager(H, W, K, images, poses, mseThres=0.0, max_depth=args.init_level)
    tree_pkl_filename = os.path.join(args.basedir, args.expname, 'treeDivide_{:04d}.pkl'.format(global_epoch))
    loaded_tree = False
    if os.path.exists(tree_pkl_filename):
        treeManager.load_trees(tree_pkl_filename)                                   # f.read + class mapping
        treeManager.quadTrees = treeManager.quadTrees
        treeManager.childrens = [get_children(treeManager.quadTrees[i].root) for i in range(treeManager.n_images)]
        treeManager.cur_level = global_epoch
        loaded_tree = True
    leaves_at_start = [treeManager.leaves(i).tolist() for i in range(treeManager.n_images)]

    log = {'resumed_from': resumed_from, 'loaded_tree': loaded_tree, 'leaves_at_start': leaves_at_start,
           'iter_at_start': global_iter, 'epochs': []}

    if global_epoch == 0:
        dH, dW = H // 4, W // 4
        coords = torch.stack(torch.meshgrid(torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH),
                                            torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW), indexing='ij'), -1).reshape(-1, 2)
        randNum = min(int(N_rand * 500 / treeManager.n_images), coords.shape[0])
        select_inds = np.random.choice(coords.shape[0], size=[randNum], replace=False)
        select_coords = coords[select_inds].long()
        rays_o, rays_d, target_rgb = [], [], []
        origins, dirs = treeManager.origins, treeManager.dirs                         # [n,H,W,3] like tree.py:179-180
        for i in range(treeManager.n_images):
            rays_o.append(origins[i][select_coords[:, 0], select_coords[:, 1]])
            rays_d.append(dirs[i][select_coords[:, 0], select_coords[:, 1]])
            target_rgb.append(treeManager.images[i][select_coords[:, 0], select_coords[:, 1]])
        rays_o, rays_d, target_rgb = torch.cat(rays_o, 0), torch.cat(rays_d, 0), torch.cat(target_rgb, 0)
        epoch_size, batch_begin, batch_end = rays_o.shape[0], 0, 0
        while batch_end < epoch_size:
            batch_end = min(batch_begin + N_rand, epoch_size)
            batch_rays = torch.stack([rays_o[batch_begin:batch_end].to(device), rays_d[batch_begin:batch_end].to(device)], 0)
            target_s = target_rgb[batch_begin:batch_end].to(device)
            rgb, disp, acc, extras = render(H, W, K, chunk=args.chunk, rays=batch_rays, retraw=True, **render_kwargs_train)
            optimizer.zero_grad()
            img_loss = img2mse(rgb, target_s)
            loss = img_loss + img2mse(extras['rgb0'], target_s)
            loss.backward()
            optimizer.step()
            batch_begin = batch_end
        log['warmup_loss'] = float(img_loss)

    for epoch_id in range(global_epoch + 1, args.n_epoch + 1):
        if epoch_id == args.n_epoch:
            treeManager.epoch_size = treeManager.n_images * treeManager.h * treeManager.w
            rays_o, rays_d, target_rgb = treeManager.gen_rays_v3_multiThread(down_scale=1, prob=False, last_epoch=True)
        else:
            rays_o, rays_d, target_rgb = treeManager.gen_rays_v3_multiThread(down_scale=1, prob=False,
                                                                             randSamp_proc=args.randSamp_perc, last_epoch=False)
        epoch_size, batch_begin, batch_end, it = rays_o.shape[0], 0, 0, 0
        rgb_gt_collect, rgb_pred_collect, losses = [], [], []
        while batch_end < epoch_size:
            batch_end = min(batch_begin + N_rand, epoch_size)
            batch_origins = rays_o[batch_begin:batch_end].to(device)
            batch_dirs = rays_d[batch_begin:batch_end].to(device)
            target_s = target_rgb[batch_begin:batch_end].to(device)
            batch_rays = torch.stack([batch_origins, batch_dirs], 0)
            rgb, disp, acc, extras = render(H, W, K, chunk=args.chunk, rays=batch_rays, retraw=True, **render_kwargs_train)
            optimizer.zero_grad()
            img_loss = img2mse(rgb, target_s)
            trans = extras['raw'][..., -1]
            loss = img_loss
            psnr = mse2psnr(img_loss.cpu())
            if 'rgb0' in extras:
                img_loss0 = img2mse(extras['rgb0'], target_s)
                loss = loss + img_loss0
                psnr0 = mse2psnr(img_loss0.cpu())
            loss.backward()
            optimizer.step()
            decay_rate = 0.1
            decay_steps = args.lrate_decay * 1000
            new_lrate = args.lrate * (decay_rate ** (global_iter / decay_steps))
            for param_group in optimizer.param_groups:
                param_group['lr'] = new_lrate
            rgb_gt_collect.append(target_s.cpu().detach())
            rgb_pred_collect.append(rgb.cpu().detach())
            losses.append(float(img_loss))
            global_iter += 1
            it += 1
            batch_begin = batch_end
        n_before = sum(len(c) for c in treeManager.childrens)
        if args.subdivide_every > 0 and epoch_id % args.subdivide_every == 0 and epoch_id < args.n_epoch - 1:
            rgb_gt = torch.cat(rgb_gt_collect, 0)
            rgb_pred = torch.cat(rgb_pred_collect, 0)
            treeManager.adjust_tree_multiThread(rgb_gt, rgb_pred, thres=args.subdivide_thres, debug=False)
        path = os.path.join(basedir, expname, '{:03d}.tar'.format(epoch_id))
        torch.save({'global_epoch': epoch_id, 'global_iter': global_iter,
                    'network_fn_state_dict': render_kwargs_train['network_fn'].state_dict(),
                    'network_fine_state_dict': render_kwargs_train['network_fine'].state_dict(),
                    'optimizer_state_dict': optimizer.state_dict()}, path)
        tree_pkl_filename = os.path.join(args.basedir, args.expname, 'treeDivide_{:04d}.pkl'.format(epoch_id))
        treeManager.save_trees(tree_pkl_filename)
        log['epochs'].append({'epoch': epoch_id, 'iters': it, 'rays': int(epoch_size), 'loss_first': losses[0],
                              'loss_last': float(np.mean(losses[-3:])), 'leaves_before': n_before,
                              'leaves_after': sum(len(c) for c in treeManager.childrens), 'cur_level': treeManager.cur_level,
                              'lr': optimizer.param_groups[0]['lr'], 'psnr': float(psnr[0])})
        if cli.stop_after and epoch_id >= cli.stop_after:
            break
    log['leaves_at_end'] = [treeManager.leaves(i).tolist() for i in range(treeManager.n_images)]
    log['global_iter'] = global_iter
    log['adam_step'] = int(float(optimizer.state_dict()['state'][0]['step']))
    print('LOOPLOG ' + json.dumps(log))


if __name__ == '__main__':
    main()
