"""The one-call optimisation step (fastnerf_train_step, run_nerf.py:479-508 of the reference in one C-ABI call) against the
call-by-call sequencing of the same entry points: bit-identical, in both math modes and with both backward kinds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def _trainer(fn, fused, N_importance=24, raw_noise_std=0.):
    torch.manual_seed(3)
    args = fn.run_nerf.make_args(N_importance=N_importance, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True,
                                 lrate=5e-4, lrate_decay=500, raw_noise_std=raw_noise_std)
    ktr = fn.run_nerf.create_nerf(args)[0]
    K = np.array([[40.0, 0, 16.0], [0, 40.0, 16.0], [0, 0, 1]])
    tr = fn.run_nerf.Trainer(ktr, 32, 32, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    assert tr.fused
    tr.fused = fused
    return tr


def _batches(fn, n_steps, n):
    g = torch.Generator().manual_seed(11)
    c2w = fn.synthetic.pose_spherical(20.0, -30.0, 4.0)[:3, :4]
    K = np.array([[40.0, 0, 16.0], [0, 40.0, 16.0], [0, 0, 1]])
    ro, rd = fn.run_nerf_helpers.get_rays(32, 32, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    out = []
    for _ in range(n_steps):
        sel = torch.randint(0, 1024, (n,), generator=g).cuda()
        tag = torch.stack([torch.zeros(n, dtype=torch.int32), torch.randint(0, 16, (n,), generator=g).int()], 1).cuda()
        out.append((ro[sel].contiguous(), rd[sel].contiguous(), torch.rand(n, 3, generator=g).cuda(), tag))
    return out


@pytest.mark.parametrize('compact', ['0', '1'])
@pytest.mark.parametrize('n_imp', [24, 0])
def test_fused_step_equals_call_by_call(fn, math_mode, compact, n_imp):
    old = fn.render.get_compact()
    fn.render.set_compact(compact)
    try:
        res = []
        for fused in (True, False):
            tr = _trainer(fn, fused, N_importance=n_imp)
            table = torch.zeros(16, device='cuda', dtype=torch.int32)
            torch.manual_seed(5)            # the device-side Philox streams are keyed from torch's CPU generator
            losses, rgbs = [], []
            for ro, rd, tgt, tag in _batches(fn, 4, 200):
                loss2, out = tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=16)
                losses.append(loss2.clone())
                rgbs.append(out['rgb_map'].clone())
                assert ('rgb0' in out) == (n_imp > 0)
            assert tr.last_step_live == (compact == '1')
            res.append((torch.stack(losses), torch.stack(rgbs), tr.flat.clone(), tr.m.clone(), tr.v.clone(), table.clone(),
                        tr.pc[0].clone(), tr.lr, tr.adam_t))
        a, b = res
        for x, y in zip(a[:7], b[:7]):
            assert torch.equal(x, y)
        assert a[7] == b[7] and a[8] == b[8] == 4
    finally:
        fn.render.set_compact(old)


def test_fused_step_with_sigma_noise_and_injected_randoms(fn, math_mode):
    res = []
    for fused in (True, False):
        tr = _trainer(fn, fused, raw_noise_std=1.0)
        torch.manual_seed(9)
        g = torch.Generator().manual_seed(2)
        (ro, rd, tgt, tag), = _batches(fn, 1, 96)
        t_rand, u = torch.rand(96, 16, generator=g).cuda(), torch.rand(96, 24, generator=g).cuda()
        loss2, out = tr.step(ro, rd, tgt, t_rand=t_rand, u=u, n_global=192)
        res.append((loss2.clone(), out['rgb_map'].clone(), out['z_vals'].clone(), tr.grad.clone(), tr.flat.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_step_outputs_survive_the_next_step(fn):
    """Callers keep `loss2` / `out` of earlier steps (e.g. to average losses at the end of an epoch): every step writes
    into a fresh block."""
    tr = _trainer(fn, True)
    kept = []
    for ro, rd, tgt, tag in _batches(fn, 3, 64):
        loss2, out = tr.step(ro, rd, tgt)
        kept.append((loss2, loss2.clone(), out['rgb_map'], out['rgb_map'].clone()))
    torch.cuda.synchronize()
    for l, lc, r, rc in kept:
        assert torch.equal(l, lc) and torch.equal(r, rc)
