"""The fp32-MFMA MLP kernels vs the oracle (torch CPU fp32): forward on explicit points,
backward w.r.t. all 24 parameter tensors, weight packing, tail tiles, linearity."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


@pytest.fixture(scope='module')
def weights(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    return {k[2:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith('c.')}


def flat_of(sd):
    return torch.cat([sd[n].reshape(-1) for n, _ in O.nerf_param_shapes()])


def rays_for_points(pts, viewdirs):
    n = pts.shape[0]
    r = torch.zeros(n, 11)
    r[:, 0:3] = pts
    r[:, 8:11] = viewdirs
    return r


def test_pack_roundtrip(fn, weights):
    flat = flat_of(weights).cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    pf, pb = pf.cpu().numpy(), pb.cpu().numpy()
    W1 = weights['pts_linears.1.weight'].numpy()
    # fwd fragment order: [(nt*KS+ks)*64 + l][t] = W[nt*32+(l&31)][ks*8+(l>>5)*4+t]
    KS = 32
    blk = pf[256 * 64:256 * 64 + 65536].reshape(8, KS, 64, 4)
    for nt, ks, l, t in ((0, 0, 0, 0), (3, 5, 37, 2), (7, 31, 63, 3)):
        assert blk[nt, ks, l, t] == W1[nt * 32 + (l & 31), ks * 8 + (l >> 5) * 4 + t]
    # layer 0: column 63 of the padded K is zero
    b0 = pf[:256 * 64].reshape(8, 8, 64, 4)
    assert b0[2, 7, 40, 3] == 0.0 and b0[2, 7, 40, 2] == weights['pts_linears.0.weight'].numpy()[2 * 32 + 8, 62]
    # transposed: feature layer, [(jt*KS+ks)*64+l][t] = W[ks*8+(l>>5)*4+t][jt*32+(l&31)]
    Wf = weights['feature_linear.weight'].numpy()
    bt = pb[128 * 256:128 * 256 + 65536].reshape(8, 32, 64, 4)
    for jt, ks, l, t in ((0, 0, 0, 0), (5, 9, 50, 1)):
        assert bt[jt, ks, l, t] == Wf[ks * 8 + (l >> 5) * 4 + t, jt * 32 + (l & 31)]


@pytest.mark.parametrize('P', [1, 127, 128, 129, 1000])
def test_mlp_forward_points(fn, weights, P):
    gen = torch.Generator().manual_seed(P)
    pts = (torch.rand(P, 3, generator=gen) * 2 - 1) * 4.0
    vd = torch.randn(P, 3, generator=gen)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        ref = O.run_network(weights, pts[:, None, :], vd)[:, 0]
    flat = flat_of(weights).cuda()
    pf, _ = fn.ops.mlp_pack(flat)
    rays = rays_for_points(pts, vd).cuda()
    raw = fn.ops.mlp_fwd(rays, torch.zeros(P, 1).cuda(), flat, pf)[:, 0]
    err = (raw.cpu() - ref).abs().max().item()
    assert err < 2e-5, err   # logits are O(0.1..1); fp32 accumulation-order differences only
    # training variant writes the same raw and consistent activations
    act = torch.empty(fn.ops.act_floats(P)).cuda()
    raw2 = fn.ops.mlp_fwd(rays, torch.zeros(P, 1).cuda(), flat, pf, act=act)[:, 0]
    assert torch.equal(raw, raw2)
    pe = act[:P * 64].view(P, 64).cpu()
    assert (pe[:, 63] == 0).all()
    assert (pe[:, :63] - O.posenc(pts, 10)).abs().max() < 2e-6


def test_mlp_backward_vs_autograd(fn, weights):
    gen = torch.Generator().manual_seed(77)
    n, S = 9, 50          # P = 450: 3 full tiles + a tail
    ro = torch.randn(n, 3, generator=gen) * 0.5
    rd = torch.randn(n, 3, generator=gen)
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values
    cot = torch.randn(n, S, 4, generator=gen)
    sd = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    out = O.run_network(sd, pts, rb[:, 8:11])
    grads_ref = torch.autograd.grad((out * cot).sum(), list(sd.values()))
    flat = flat_of(weights).cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    P = n * S
    act = torch.empty(fn.ops.act_floats(P)).cuda()
    raw = fn.ops.mlp_fwd(rb.cuda(), z.cuda(), flat, pf, act=act)
    assert (raw.cpu() - out.detach()).abs().max() < 2e-5
    dact = torch.empty(P * fn.ops.DACT_FLOATS).cuda()
    partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
    grads = torch.full((fn.ops.NET_PARAMS,), float('nan')).cuda()
    fn.ops.mlp_bwd(cot.cuda(), act, flat, pb, dact, partial, grads)
    grads = grads.cpu()
    assert torch.isfinite(grads).all()
    off = 0
    for (name, shape), gr in zip(O.nerf_param_shapes(), grads_ref):
        k = gr.numel()
        got = grads[off:off + k].view(shape)
        scale = max(1.0, gr.abs().max().item())
        err = (got - gr).abs().max().item()
        assert err < 2e-5 * scale, (name, err, scale)
        off += k
    # linearity in the cotangent (size-independent property)
    grads2 = torch.empty(fn.ops.NET_PARAMS).cuda()
    fn.ops.mlp_bwd((cot * 2).cuda(), act, flat, pb, dact, partial, grads2)
    assert (grads2.cpu() - 2 * grads).abs().max() < 1e-4 * max(1.0, grads.abs().max().item())


def test_run_network_signature(fn, weights):
    net = fn.model.NeRF()
    net.load_state_dict({'module.' + k: v for k, v in weights.items()})
    gen = torch.Generator().manual_seed(3)
    pts = torch.randn(5, 7, 3, generator=gen)
    vd = torch.randn(5, 3, generator=gen); vd = vd / vd.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        ref = O.run_network(weights, pts, vd)
    out = fn.run_nerf.run_network(pts.cuda(), vd.cuda(), net)
    assert out.shape == (5, 7, 4) and (out.cpu() - ref).abs().max() < 2e-5
    # the nn.Module forward on embedded inputs agrees too (third-party convenience path)
    emb = torch.cat([O.posenc(pts.reshape(-1, 3), 10), O.posenc(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    with torch.no_grad():
        out2 = net(emb.cuda()).cpu().reshape(5, 7, 4)
    assert (out2 - ref).abs().max() < 1e-4
    assert list(net.state_dict().keys()) == [n for n, _ in O.nerf_param_shapes()]
