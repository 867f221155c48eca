"""BASELINE configs[4] as a LOOP: the caller surface of nerf++-ours/ddp_train_nerf.py on the product -- create_nerf (:136-184),
train_step (:327-424) and the epoch loop with the quadtree fork in it (:187-324: MEAN split rule, prob=True picks with
rand = randSamp_perc = 0.7, last epoch uniform over every pixel), `model_{epoch:04d}.pth` checkpoints in the reference's layout
(G21, recorded from the reference) written, resumed from, and read from a file laid out like the reference's own."""
import os
import types
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def _samplers(fn, n=3, H=24, W=32):
    """Cameras INSIDE the unit sphere (radius 0.5) looking at the analytic scene scaled into it; img / rays like a RaySamplerSingleImage."""
    out = []
    focal = 0.5 * W / np.tan(0.5 * 0.9)
    for i in range(n):
        c2w = fn.synthetic.pose_spherical(40.0 * i, -20.0, 0.5)[:3, :4]
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        ro, rd = fn.run_nerf_helpers.get_rays(H, W, K, c2w)
        img = fn.synthetic.render_rays((ro.reshape(-1, 3) * 8.0).cuda(), rd.reshape(-1, 3).cuda(), near=0.5, far=7.5).cpu()
        rs = types.SimpleNamespace(H=H, W=W, img=img.numpy().astype(np.float32), rays_o=ro.reshape(-1, 3).cpu().numpy(),
                                   rays_d=rd.reshape(-1, 3).cpu().numpy())
        out.append(rs)
    return out


def _args(fn, basedir, **kw):
    a = dict(cascade_level=2, cascade_samples='16,16', batch_size=256, lrate=5e-4, n_epoch=4, init_level=2, subdivide_every=1,
             subdivide_thres=0.02, randSamp_perc=0.7, rays_downscale=1, basedir=basedir, expname='pp', no_reload=False,
             ckpt_path=None, optim_autoexpo=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_surface_callables_against_g10(fn, golden_dir):
    """perturb_samples / sample_pdf / depth2pts_outside as callables with the reference's signatures (they existed as
    kernels only)."""
    g = np.load(os.path.join(golden_dir, 'g10_pp_ops.npz'))
    ray_o, ray_d, depth = (torch.from_numpy(g[k]).cuda() for k in ('ray_o', 'ray_d', 'depth'))
    pts, dr = fn.nerfpp.depth2pts_outside(ray_o[:, None].expand(-1, 16, 3), ray_d[:, None].expand(-1, 16, 3), depth)
    assert pts.shape == (40, 16, 4) and np.abs(pts.cpu().numpy() - g['pts']).max() < 2e-6
    assert np.abs(dr.cpu().numpy() - g['depth_real']).max() < 2e-5 * np.abs(g['depth_real']).max()
    bins, w = torch.from_numpy(g['bins']).cuda(), torch.from_numpy(g['w']).cuda()
    s_det = fn.nerfpp.sample_pdf(bins, w, 128, det=True)
    d = np.abs(s_det.cpu().numpy() - g['s_det'])
    assert (d > 2e-5).mean() < 0.02 and d.max() < 0.1                       # (ill-conditioned at bin edges)
    s_u = fn.ops.pp_sample_pdf(bins, w, 128, u=torch.from_numpy(g['u']).cuda())
    d = np.abs(s_u.cpu().numpy() - g['s_u'])
    assert (d > 2e-5).mean() < 0.02 and d.max() < 0.1
    s_r = fn.nerfpp.sample_pdf(bins[None].expand(2, -1, -1), w[None].expand(2, -1, -1), 32)     # leading dims, random draws
    assert s_r.shape == (2, 40, 32) and float(s_r.min()) >= float(bins.min()) - 1e-5 and float(s_r.max()) <= float(bins.max()) + 1e-4
    with pytest.raises(ValueError):
        fn.nerfpp.sample_pdf(bins, w[:, :-1], 8)
    # perturb_samples: inside the mid-point intervals, endpoints kept inside [z0, z_last]; injected draws reproduce the formula
    z = torch.sort(torch.rand(50, 20), -1).values.cuda()
    t = torch.rand(50, 20).cuda()
    got = fn.ops.pp_perturb_samples(z, t_rand=t)
    mids = .5 * (z[:, 1:] + z[:, :-1])
    upper, lower = torch.cat([mids, z[:, -1:]], -1), torch.cat([z[:, :1], mids], -1)
    assert torch.equal(got, lower + (upper - lower) * t)
    torch.manual_seed(1)
    a = fn.nerfpp.perturb_samples(z.reshape(5, 10, 20))
    torch.manual_seed(1)
    b = fn.nerfpp.perturb_samples(z.reshape(5, 10, 20))
    assert a.shape == (5, 10, 20) and torch.equal(a, b) and ((a.reshape(50, 20) >= lower) & (a.reshape(50, 20) <= upper)).all()


def test_checkpoint_layout_is_the_references(fn, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, 'g21_pp_ckpt_layout.npz'))
    args = _args(fn, str(tmp_path))
    start, models = fn.nerfpp.create_nerf(0, args)
    assert start == 0 and list(models.keys()) == ['cascade_level', 'cascade_samples', 'net_0', 'optim_0', 'net_1', 'optim_1']
    sd = models['net_0'].reference_state_dict()
    assert list(sd.keys()) == [str(k) for k in g['names']]
    assert [';'.join(str(d) for d in v.shape) for v in sd.values()] == [str(s) for s in g['shapes']]
    # same construction order and seed as the reference's create_nerf -> the same initial weights
    assert np.array_equal(sd[str(g['names'][0])].reshape(-1)[:16].cpu().numpy(), g['first_weight_head'])
    osd = models['optim_0'].state_dict()
    assert sorted(osd['param_groups'][0].keys()) == [str(k) for k in g['optim_group_keys']]
    assert osd['param_groups'][0]['params'] == g['optim_params'].tolist()
    # a file laid out like the reference's (DataParallel names, torch Adam state over 48 tensors) loads through create_nerf
    ref = OrderedDict()
    gen = torch.Generator().manual_seed(3)
    for m in range(2):
        ref['net_%d' % m] = OrderedDict((str(n), torch.randn([int(d) for d in str(s).split(';')], generator=gen) * 0.05)
                                        for n, s in zip(g['names'], g['shapes']))
        ps = [torch.nn.Parameter(v.clone()) for v in ref['net_%d' % m].values()]
        opt = torch.optim.Adam(ps, lr=5e-4)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=gen) * 1e-3
        opt.step()
        ref['optim_%d' % m] = opt.state_dict()
        assert sorted(ref['optim_%d' % m]['state'][0].keys()) == [str(k) for k in g['optim_state_keys']]
    os.makedirs(os.path.join(str(tmp_path), 'pp'))
    torch.save(ref, os.path.join(str(tmp_path), 'pp', 'model_0007.pth'))
    start, models = fn.nerfpp.create_nerf(0, args)
    assert start == 7
    for m in range(2):
        got = models['net_%d' % m].reference_state_dict()
        for k, v in ref['net_%d' % m].items():
            assert torch.equal(got[k].cpu(), v), k
        st = models['optim_%d' % m].state_dict()['state']
        assert len(st) == 48 and torch.equal(st[5]['exp_avg'].cpu(), ref['optim_%d' % m]['state'][5]['exp_avg'])


@pytest.mark.parametrize('fused', [True, False])
def test_epoch_loop_with_the_quadtree_fork_and_resume(fn, tmp_path, fused):
    """fused=True: the data-parallel engine (CascadeTrainer, on-device SUM / COUNT tables; one rank here);
    fused=False: the reference-shaped route (autograd + torch.optim.Adam, predictions collected on the host)."""
    samplers = _samplers(fn)
    args = _args(fn, str(tmp_path), fused=fused)
    torch.manual_seed(0)
    np.random.seed(0)
    logs = []
    models, tree, rec = fn.nerfpp.ddp_train_nerf(args, samplers, log=logs.append, stop_after=2)
    d = os.path.join(str(tmp_path), 'pp')
    assert sorted(os.listdir(d)) == ['model_0001.pth', 'model_0002.pth'] and [r['epoch'] for r in rec] == [1, 2]
    assert rec[0]['leaves_before'] == 3 * 4 and rec[0]['leaves_after'] > rec[0]['leaves_before']      # MEAN rule split some leaves
    assert rec[1]['cur_level'] == 4 and all(np.isfinite(r['mse']) for r in rec)
    assert rec[0]['rays'] == 3 * 24 * 32                           # down_scale 1: as many picks as pixels (int(area) per finest leaf)
    ck = torch.load(os.path.join(d, 'model_0002.pth'), weights_only=False)
    assert list(ck.keys()) == ['net_0', 'optim_0', 'net_1', 'optim_1']
    assert next(iter(ck['net_1'])).startswith('module.nerf_net.fg_net.base_layers.0.0.weight')
    steps = int(float(ck['optim_0']['state'][0]['step']))
    assert steps == sum(-(-r['rays'] // 256) for r in rec)         # one Adam step per batch and level
    w_end = models['net_1'].nerf_net.flat.clone()
    # ---- resume: create_nerf picks model_0002.pth up; epochs 3 (prob picks) and 4 (last epoch: every pixel, uniform) follow ----
    models2, tree2, rec2 = fn.nerfpp.ddp_train_nerf(args, samplers, log=logs.append)
    assert [r['epoch'] for r in rec2] == [3, 4]
    assert rec2[1]['rays'] == 3 * 24 * 32 and rec2[1]['leaves_after'] == rec2[1]['leaves_before']   # no subdivision in the last two epochs
    assert sorted(os.listdir(d))[-1] == 'model_0004.pth'
    ck4 = torch.load(os.path.join(d, 'model_0004.pth'), weights_only=False)
    assert int(float(ck4['optim_1']['state'][0]['step'])) == steps + sum(-(-r['rays'] // 256) for r in rec2)
    assert not torch.equal(models2['net_1'].nerf_net.flat, w_end)
    assert rec2[-1]['mse'] < rec[0]['mse']                         # it trains
    # the evaluation path renders with the resumed nets
    models2['cascade_level'], models2['cascade_samples'] = 2, [16, 16]
    rs = types.SimpleNamespace(H=24, W=32, get_all=lambda: {
        'ray_o': torch.from_numpy(samplers[0].rays_o), 'ray_d': torch.from_numpy(samplers[0].rays_d),
        'min_depth': 1e-4 * torch.ones(24 * 32)})
    ret = fn.nerfpp.render_single_image(models2, rs, 300)
    assert ret[1]['rgb'].shape == (24, 32, 3) and torch.isfinite(ret[1]['rgb']).all()
    mse = float(((ret[1]['rgb'].reshape(-1, 3) - torch.from_numpy(samplers[0].img)) ** 2).mean())
    assert mse < 0.1


def test_fused_and_reference_shaped_routes_split_the_same_leaves(fn, tmp_path):
    """One epoch from the same seeds on both routes: same picks, same per-batch arithmetic up to summation order -> the MEAN rule
    (device SUM / COUNT tables vs host predictions) splits the same leaves; the per-leaf means stay clear of the threshold."""
    samplers = _samplers(fn)
    leaves, mses = [], []
    for fused in (True, False):
        args = _args(fn, str(tmp_path / ('f%d' % fused)), fused=fused, n_epoch=4)
        torch.manual_seed(0)
        np.random.seed(0)
        _, tree, rec = fn.nerfpp.ddp_train_nerf(args, samplers, log=lambda *_: None, stop_after=1)
        leaves.append([np.asarray(tree.leaves(i)).copy() for i in range(tree.n_images)])
        mses.append(rec[0]['mse'])
    # (the two routes draw their jitter from different streams, so predictions differ at the 1e-2 level; the split decisions of
    #  this scene are far from the threshold)
    assert abs(mses[0] - mses[1]) < 0.2 * max(mses)
    assert all(np.array_equal(a, b) for a, b in zip(*leaves))


_PP_WORKER = r"""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import fastnerf
from fastnerf import parallel
from test_gpu_nerfpp_loop import _samplers, _args
rank, world, local = parallel.init_from_env('cuda')
torch.cuda.set_device(0 if torch.cuda.device_count() < world else local)
samplers = _samplers(fastnerf)
args = _args(fastnerf, %(base)r + '/w%%d' %% world, fused=True, perturb=0, batch_size=250, n_epoch=5, subdivide_thres=0.03)
# (1) one batch through CascadeTrainer.step without the update: the all-reduced gradient of rows r::world with n_global
_, m0 = fastnerf.nerfpp.create_nerf(rank, args)
ct = fastnerf.nerfpp.CascadeTrainer([m0['net_0'], m0['net_1']], [16, 16], perturb=False)
gen = torch.Generator().manual_seed(2)
N = 301                                           # uneven shards
ro = torch.from_numpy(np.concatenate([s.rays_o for s in samplers]))[torch.randperm(3 * 24 * 32, generator=gen)[:N]].cuda()
rd = torch.from_numpy(np.concatenate([s.rays_d for s in samplers]))[torch.randperm(3 * 24 * 32, generator=gen)[:N]].cuda()
tg = torch.rand(N, 3, generator=gen).cuda()
sl = slice(rank, N, world)
ct.step(ro[sl].contiguous(), rd[sl].contiguous(), tg[sl].contiguous(), n_global=N, update=False)
grad1 = [n.flat_grad.cpu().clone() for n in ct.nets]
# (2) the loop
torch.manual_seed(11); np.random.seed(11)
models, tree, rec = fastnerf.nerfpp.ddp_train_nerf(args, samplers, log=lambda *_: None, stop_after=3)
tr = fastnerf.nerfpp.ddp_train_nerf.last_trainer
out = {'leaves': [np.asarray(tree.leaves(i)).copy() for i in range(tree.n_images)], 'rec': rec, 'grad1': grad1,
       'flat': [n.flat.cpu() for n in tr.nets], 'm': [x.cpu() for x in tr.m], 't': list(tr.t)}
torch.save(out, %(base)r + '/res_w%%d_r%%d.pt' %% (world, rank))
if world > 1:
    parallel.barrier(); parallel.shutdown_cabi(); torch.distributed.destroy_process_group()
"""


def test_two_rank_loop_equals_single_rank(tmp_path):
    """Config 5 data-parallel: two ranks (gloo on a 1-GPU box, RCCL otherwise) run ddp_train_nerf on rows r::2 of
    every batch (batch 250: uneven shards never occur, 768 rays per epoch: a ragged last batch does); deterministic depths
    (perturb=0) make the run comparable with the single-rank run on the union.  Replicas bit-identical; leaf lists identical to the
    single-rank run after three epochs with two subdivisions; per-leaf SUM / COUNT tables are exact, so the decision only differs if
    a mean sits within the 1e-6 that summation grouping moves the predictions -- the test checks it does not."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = str(tmp_path / 'pp_worker.py')
    with open(script, 'w') as f:
        f.write(_PP_WORKER % {'root': root, 'base': str(tmp_path)})
    r1 = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env['FASTNERF_DIST_BACKEND'] = 'gloo'
    r2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                         '127.0.0.1', '--master-port', '29561', script], env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    one = torch.load(str(tmp_path / 'res_w1_r0.pt'), weights_only=False)
    ra, rb = (torch.load(str(tmp_path / ('res_w2_r%d.pt' % r)), weights_only=False) for r in (0, 1))
    # replicas: bit-identical parameters, moments, step counts, trees and records
    for k in ('flat', 'm'):
        assert all(torch.equal(x, y) for x, y in zip(ra[k], rb[k])), k
    assert ra['t'] == rb['t'] == one['t'] and ra['rec'] == rb['rec']
    assert all(np.array_equal(x, y) for x, y in zip(ra['leaves'], rb['leaves']))
    # against the single rank on the union: same trees, parameters within summation-order noise
    assert [r['leaves_after'] for r in ra['rec']] == [r['leaves_after'] for r in one['rec']]
    assert all(np.array_equal(x, y) for x, y in zip(ra['leaves'], one['leaves']))
    assert one['rec'][0]['leaves_after'] > one['rec'][0]['leaves_before']
    # one batch, no update: the all-reduced gradient of the shards IS the single-rank gradient (summation grouping only) ...
    for x, y in zip(ra['grad1'], one['grad1']):
        assert float((x - y).abs().max()) < 5e-6 * float(y.abs().max())
    assert all(torch.equal(x, y) for x, y in zip(ra['grad1'], rb['grad1']))
    # ... whereas 36 Adam steps later the two runs have decorrelated entry by entry (Adam's lr * g / (|g| + eps) turns 1e-9 of
    # gradient noise on |g| ~ 1e-8 entries into full-size steps); what is compared is the fit they reach
    assert abs(ra['rec'][-1]['mse'] - one['rec'][-1]['mse']) < 0.1 * one['rec'][-1]['mse']
    for x, y in zip(ra['flat'], one['flat']):
        assert float((x - y).norm() / y.norm()) < 0.2
