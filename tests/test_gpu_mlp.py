"""The MLP kernels (fp32-MFMA and split-bf16 math modes) vs the oracle (torch CPU fp32): forward on
explicit points, backward w.r.t. all 24 parameter tensors, weight packing, saved-tensor layout, tail tiles,
linearity."""
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


def bf16_pair_to_f32(u16_hi, u16_lo):
    hi = (u16_hi.astype(np.uint32) << 16).view(np.float32)
    lo = (u16_lo.astype(np.uint32) << 16).view(np.float32)
    return hi + lo


def decode_kfrag(buf_u16, base_u4, CT, ntiles, permuted):
    """Test-side restatement of the K-fragment layout (csrc/mlp_bf16.hip header): returns [ntiles*64, CT*32]."""
    n = ntiles * CT * 4 * 2 * 64 * 8
    a = buf_u16[base_u4 * 8: base_u4 * 8 + n].reshape(ntiles, CT, 4, 2, 2, 32, 8)   # tile ct ks part kb c e
    v = bf16_pair_to_f32(a[:, :, :, 0], a[:, :, :, 1])                                # tile ct ks kb c e
    v = v.transpose(0, 2, 3, 5, 1, 4).reshape(ntiles * 64, CT * 32)                   # (tile ks kb e) (ct c)
    if permuted:
        q = np.arange(CT * 32)
        n_of_q = (q & ~63) + 2 * (q & 31) + ((q >> 5) & 1)
        out = np.empty_like(v)
        out[:, n_of_q] = v
        v = out
    return v


def test_pack_roundtrip_bf16(fn, weights):
    fn.ops.set_math('bf16x3')
    flat = flat_of(weights).cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    pf = pf.cpu().numpy().view(np.uint16)
    pb = pb.cpu().numpy().view(np.uint16)
    W1 = weights['pts_linears.1.weight'].numpy()
    # layer 1 block: [nt_g][ks][part][lane][8], value = W[n][ks*16 + (lane>>5)*8 + e], n = (nt_g>>1)*64 + 2*(lane&31) + (nt_g&1)
    off = (8 * 4 * 2 * 64) * 8   # layer 0: 8 column tiles x 4 k-steps
    blk = pf[off: off + 8 * 16 * 2 * 64 * 8].reshape(8, 16, 2, 64, 8)
    for nt, ks, l, e in ((0, 0, 0, 0), (3, 5, 37, 2), (7, 15, 63, 7)):
        n = (nt >> 1) * 64 + 2 * (l & 31) + (nt & 1)
        got = bf16_pair_to_f32(blk[nt, ks, 0, l, e:e + 1], blk[nt, ks, 1, l, e:e + 1])[0]
        want = W1[n, ks * 16 + (l >> 5) * 8 + e]
        assert abs(got - want) <= 2.0 ** -16 * abs(want)
    # transposed feature layer (second backward block, after Vt with K = 128)
    Wf = weights['feature_linear.weight'].numpy()
    off = (8 * 8 * 2 * 64) * 8
    bt = pb[off: off + 8 * 16 * 2 * 64 * 8].reshape(8, 16, 2, 64, 8)
    for nt, ks, l, e in ((0, 0, 0, 0), (5, 9, 50, 1)):
        n = (nt >> 1) * 64 + 2 * (l & 31) + (nt & 1)
        got = bf16_pair_to_f32(bt[nt, ks, 0, l, e:e + 1], bt[nt, ks, 1, l, e:e + 1])[0]
        want = Wf[ks * 16 + (l >> 5) * 8 + e, n]
        assert abs(got - want) <= 2.0 ** -16 * abs(want)
    fn.ops.set_math(os.environ.get('FASTNERF_MATH', 'bf16x6'))


def test_pack_roundtrip(fn, weights):
    fn.ops.set_math('fp32')
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
    fn.ops.set_math(os.environ.get('FASTNERF_MATH', 'bf16x6'))


@pytest.mark.parametrize('P', [1, 127, 128, 129, 1000])
def test_mlp_forward_points(fn, weights, P, math_mode):
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
    if math_mode in ('fp32', 'bf16x6'):   # (bf16x6 keeps the exact-fp32 kernels' buffers)
        pe = act[:P * 64].view(P, 64).cpu()
    else:   # K-fragment tensors: pe (natural channel order), h0 (wave-permuted order)
        nt = (P + 63) // 64
        u16 = act.cpu().numpy().view(np.uint16)
        # buffer order: h0..h7 | feat | vpe | hv | sign bits | pe (csrc/mlp_bf16.hip ba_*)
        pe_off = nt * (9 * 4096 + 512 + 2048 + 1024 + 64)
        pe = torch.from_numpy(decode_kfrag(u16, pe_off, 2, nt, False)[:P])
        h0 = torch.from_numpy(decode_kfrag(u16, 0, 8, nt, True)[:P])
        h0_ref = torch.relu(O.posenc(pts, 10) @ weights['pts_linears.0.weight'].T + weights['pts_linears.0.bias'])
        assert (h0 - h0_ref).abs().max() < 2e-5
    assert (pe[:, 63] == 0).all()
    # fp32 mode stores fp32; split mode stores (hi, lo) bf16 pairs = 16 significand bits (|x| <= 4 here)
    assert (pe[:, :63] - O.posenc(pts, 10)).abs().max() < (2e-6 if math_mode in ('fp32', 'bf16x6') else 4 * 2.0 ** -16)


def test_mlp_backward_vs_autograd(fn, weights, math_mode):
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
    dact = torch.empty(fn.ops.dact_floats(P)).cuda()
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


@pytest.mark.parametrize('n,S', [(1, 1), (1, 16), (1, 17), (17, 241), (241, 291), (1367, 193)])
def test_mlp_backward_ragged_point_counts(fn, weights, math_mode, n, S):
    """The gradient kernels at point counts around their staging granularity (16-point k-steps of the bf16x6 dW kernel, 128-point
    tiles) and at counts large enough that every workgroup runs its steady-state loop (1367 x 193 = 263 831 points: 65 k-steps per
    workgroup, the last one partial).
      * a handful of points: the same gradients as the fp32-MFMA mode (pinned against autograd by test_mlp_backward_vs_autograd);
      * many points: ADDITIVITY over a split of the rays, G(all) = G(first part) + G(rest), which only the grouping of the fp32
        partial sums may break (a point's forward, and with it every ReLU decision, does not depend on its batch).  Against the
        other math mode only a loose bound holds there: among 5e8 pre-activations a few dozen sit within rounding of zero, and a
        ReLU that flips between two roundings moves a gradient entry by 1e-3 of the tensor's maximum."""
    gen = torch.Generator().manual_seed(1234 + n)
    ro = (torch.randn(n, 3, generator=gen) * 0.5).cuda()
    rd = torch.randn(n, 3, generator=gen).cuda()
    rb = fn.ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
    cot = torch.randn(n, S, 4, generator=gen).cuda()
    flat = flat_of(weights).cuda()

    def grads_in(mode, sl=slice(None)):
        fn.ops.set_math(mode)
        try:
            r, zz, c = rb[sl].contiguous(), z[sl].contiguous(), cot[sl].contiguous()
            P = r.shape[0] * S
            pf, pb = fn.ops.mlp_pack(flat)
            act = torch.empty(fn.ops.act_floats(P)).cuda()
            fn.ops.mlp_fwd(r, zz, flat, pf, act=act)
            dact = torch.empty(fn.ops.dact_floats(P)).cuda()
            partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
            g = torch.full((fn.ops.NET_PARAMS,), float('nan')).cuda()
            fn.ops.mlp_bwd(c, act, flat, pb, dact, partial, g)
            return g.double().cpu()
        finally:
            fn.ops.set_math(math_mode)

    def compare(got, ref, tol, what):
        off = 0
        for name, shape in O.nerf_param_shapes():
            k = int(np.prod(shape))
            a, b = got[off:off + k], ref[off:off + k]
            scale = max(1e-6, b.abs().max().item())
            assert (a - b).abs().max().item() <= tol * scale, (what, name, n, S, (a - b).abs().max().item(), scale)
            off += k

    got = grads_in(math_mode)
    assert torch.isfinite(got).all()
    if n * S <= 64:
        if math_mode != 'fp32':
            compare(got, grads_in('fp32'), 2e-5 if math_mode in ('bf16x6',) else 2e-2, 'vs fp32-MFMA')
        return
    cut = n // 2 + 1
    parts = grads_in(math_mode, slice(0, cut)) + grads_in(math_mode, slice(cut, n))
    compare(got, parts, 5e-6 if math_mode != 'bf16x3' else 2e-5, 'additivity')
    if math_mode != 'fp32':
        ref = grads_in('fp32')
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel < (2e-3 if math_mode in ('bf16x6',) else 2e-2), ('vs fp32-MFMA, relative L2', rel)


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


def test_mlp_edge_cases(fn, weights, math_mode):
    """Empty batches, error reporting through the C ABI (no abort), mode-specific opaque buffers."""
    flat = flat_of(weights).cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    rays = torch.zeros(0, 11).cuda()
    raw = fn.ops.mlp_fwd(rays, torch.zeros(0, 5).cuda(), flat, pf)
    assert raw.shape == (0, 5, 4)
    # a bad net kind is an error code + message, not a crash
    lib = fn._lib.lib()
    a = (7, 1, 1, fn._lib.ptr(torch.zeros(1, 11).cuda()), fn._lib.ptr(torch.zeros(1, 1).cuda()), fn._lib.ptr(flat), fn._lib.ptr(pf),
         fn._lib.ptr(torch.zeros(1, 1, 4).cuda()), None)
    if math_mode in ('bf16x6',):
        rc = lib.fastnerf_mlp_x6_fwd(*a, 0, None)
    else:
        rc = (lib.fastnerf_mlp_bf16_fwd if math_mode == 'bf16x3' else lib.fastnerf_mlp_fwd_ex)(*a, None)
    assert rc != 0 and b'kind' in lib.fastnerf_last_error()
    # packed weights of the other math mode are rejected by the wrappers
    other = 'fp32' if math_mode == 'bf16x3' else 'bf16x3'
    fn.ops.set_math(other)
    try:
        with pytest.raises(AssertionError):
            fn.ops.mlp_fwd(torch.zeros(1, 11).cuda(), torch.zeros(1, 1).cuda(), flat, pf)
    finally:
        fn.ops.set_math(math_mode)


def test_large_batch_consistency(fn, weights):
    """Several tiles per persistent workgroup (the size class where a timing-dependent corruption of the training
    forward once showed up): inference and training forwards agree bit for bit, run to run, and the two math
    modes agree to fp32-rounding class; gradients are deterministic."""
    flat = flat_of(weights).cuda()
    gen = torch.Generator().manual_seed(123)
    n, S = 2048, 96                       # 196 608 points = 3072 tiles of 64
    ro = torch.randn(n, 3, generator=gen) * 0.3
    rd = torch.randn(n, 3, generator=gen)
    rays = torch.from_numpy(O.make_ray_batch(ro, rd, 2.0, 6.0).numpy()).cuda()
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
    cot = torch.randn(n, S, 4, generator=gen).cuda()
    out = {}
    old = fn.ops.get_math()
    try:
        for mode in ('fp32', 'bf16x3'):
            fn.ops.set_math(mode)
            pf, pb = fn.ops.mlp_pack(flat)
            act = torch.empty(fn.ops.act_floats(n * S)).cuda()
            dact = torch.empty(fn.ops.dact_floats(n * S)).cuda()
            partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
            r_inf = fn.ops.mlp_fwd(rays, z, flat, pf).clone()
            grads = []
            for rep in range(3):
                r_sav = fn.ops.mlp_fwd(rays, z, flat, pf, act=act)
                assert torch.equal(r_inf, r_sav), (mode, rep, (r_inf - r_sav).abs().max().item())
                g = torch.empty(fn.ops.NET_PARAMS).cuda()
                fn.ops.mlp_bwd(cot, act, flat, pb, dact, partial, g)
                grads.append(g.clone())
            assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2]), mode
            out[mode] = (r_inf, grads[0])
            del act, dact
    finally:
        fn.ops.set_math(old)
    d = (out['fp32'][0] - out['bf16x3'][0]).abs().max().item()
    assert d < 2e-5, d
    ga, gb = out['fp32'][1], out['bf16x3'][1]
    assert (ga - gb).abs().max().item() < 2e-2 * ga.abs().max().item()   # ReLU-mask-flip floor, DESIGN section 4


def test_training_forward_stress_at_bench_size(fn, weights, math_mode):
    """the training forward once showed a timing-dependent corruption (bias missing in lanes 48..63 of one
    accumulator register on ~0.1 % of the points) that disappeared with -fno-slp-vectorize and was never root-caused; at
    HEAD it does not reproduce with SLP on either (tools/slp_bisect.py: 0 bad points in 5 x 40 launches of 786 432 points,
    profiles/r02_slp_bisect.md).  This is the witness that caught it, at the bench size (12 288 tiles, 24 per persistent
    workgroup), repeated: the saving forward must reproduce the non-saving forward bit for bit on every launch, and the
    saved layer-7 activations must decode to the values that produce those outputs."""
    flat = flat_of(weights).cuda()
    gen = torch.Generator().manual_seed(321)
    n, S = 4096, 192
    ro = torch.randn(n, 3, generator=gen) * 0.3
    rd = torch.randn(n, 3, generator=gen)
    rays = torch.from_numpy(O.make_ray_batch(ro, rd, 2.0, 6.0).numpy()).cuda()
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    act = torch.empty(fn.ops.act_floats(n * S)).cuda()
    ref = fn.ops.mlp_fwd(rays, z, flat, pf).clone()
    assert torch.isfinite(ref).all()
    for rep in range(12):
        got = fn.ops.mlp_fwd(rays, z, flat, pf, act=act)
        bad = int((got != ref).any(-1).sum())
        assert bad == 0, 'launch %d: %d of %d points differ between the saving and the non-saving forward' % (rep, bad, n * S)
        assert torch.equal(fn.ops.mlp_fwd(rays, z, flat, pf), ref)
    if math_mode == 'bf16x3':
        # alpha = h7 . w_alpha + b_alpha recomputed from the SAVED layer-7 activations of 64 tiles spread over the launch
        nt = n * S // 64
        lo = 7 * nt * 4096                                     # ba_h(nt, 7) in 16-byte units: only the h7 block is copied
        buf = act[lo * 4:(lo + nt * 4096) * 4].view(torch.int16).cpu().numpy().view(np.uint16)
        base = 0
        wa = weights['alpha_linear.weight'].numpy().reshape(-1).astype(np.float64)
        ba = float(weights['alpha_linear.bias'])
        for tile in list(range(0, nt, nt // 60))[:60] + [nt - 1]:
            h7 = decode_kfrag(buf, base + tile * 4096, 8, 1, True).astype(np.float64)       # [64, 256]
            alpha = h7 @ wa + ba
            want = ref.reshape(-1, 4)[tile * 64:(tile + 1) * 64, 3].cpu().numpy()
            assert np.abs(alpha - want).max() < 2e-5 * max(1.0, np.abs(want).max()), tile


def test_tile_scheduler_back_to_back_and_streams(fn, weights):
    """The persistent forward / dX kernels draw their tiles from a self-resetting ticket counter (one counter pair per
    launch from a pool): many launches of very different sizes, back to back and interleaved on two streams, must
    reproduce the single-launch results bit for bit (a stale or shared counter would skip or repeat tiles)."""
    old = fn.ops.get_math()
    fn.ops.set_math('bf16x3')
    try:
        flat = flat_of(weights).cuda()
        pf, pb = fn.ops.mlp_pack(flat)
        gen = torch.Generator().manual_seed(5)
        sizes = [(1, 1), (1, 63), (1, 64), (5, 13), (64, 64), (257, 192), (700, 100)]   # (rays, samples): 1 .. 70 000 points
        cases = []
        for n, S in sizes:
            ro = torch.randn(n, 3, generator=gen) * 0.3
            rd = torch.randn(n, 3, generator=gen)
            rays = torch.from_numpy(O.make_ray_batch(ro, rd, 2.0, 6.0).numpy()).cuda()
            z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
            cot = torch.randn(n, S, 4, generator=gen).cuda()
            act = torch.empty(fn.ops.act_floats(n * S)).cuda()
            raw = fn.ops.mlp_fwd(rays, z, flat, pf, act=act).clone()
            dact = torch.empty(fn.ops.dact_floats(n * S)).cuda()
            partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
            grads = torch.zeros(fn.ops.NET_PARAMS).cuda()
            fn.ops.mlp_bwd(cot, act, flat, pb, dact, partial, grads)
            cases.append((rays, z, cot, raw, grads.clone()))
        torch.cuda.synchronize()
        order = torch.randint(0, len(cases), (120,), generator=gen).tolist()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        results = []
        for k, ci in enumerate(order):
            rays, z, cot, raw_ref, g_ref = cases[ci]
            with torch.cuda.stream(streams[k & 1]):
                n, S = z.shape
                act = torch.empty(fn.ops.act_floats(n * S)).cuda()
                raw = fn.ops.mlp_fwd(rays, z, flat, pf, act=act)
                dact = torch.empty(fn.ops.dact_floats(n * S)).cuda()
                partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
                grads = torch.zeros(fn.ops.NET_PARAMS).cuda()
                fn.ops.mlp_bwd(cot, act, flat, pb, dact, partial, grads)
                results.append((ci, raw, grads, act, dact, partial))
        torch.cuda.synchronize()
        for ci, raw, grads, *_ in results:
            assert torch.equal(raw, cases[ci][3]), ci
            assert torch.equal(grads, cases[ci][4]), ci
    finally:
        fn.ops.set_math(old)


def test_bf16x6_decomposition_is_exact_and_products_have_fp32_width(fn, weights):
    """The bf16x6 mode's claim (csrc/mlp_*.hip, MM_X6): every fp32 operand is decomposed EXACTLY into three bf16 pieces, so a product
    evaluated from the six leading piece products is as accurate as fp32's own rounding of it.
    (a) the packed weight planes of every layer sum back to the fp32 weights bit for bit (h + m + l in fp64, rounded to fp32);
    (b) against an fp64 evaluation of the network on the same fp32 inputs, the logits of the bf16x6 kernels are as close as the
        exact-fp32 MFMA kernels' (both are dominated by fp32 accumulation and sin / cos rounding), while the two-piece split-bf16
        mode -- 16-bit operands -- sits measurably further away: that mode is narrower than fp32, this one is not."""
    old = fn.ops.get_math()
    try:
        flat = flat_of(weights).cuda()
        fn.ops.set_math('bf16x6')
        pf, pb = fn.ops.mlp_pack(flat)
        u16 = pf.cpu().numpy().view(np.uint16)             # uint4 units of 8 bf16: [tile][ks][plane][lane][8]

        def bf(x):
            return (x.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        off = 0
        for l in range(10):
            name = ('pts_linears.%d.weight' % l) if l < 8 else ('feature_linear.weight' if l == 8 else 'views_linears.0.weight')
            w = weights[name].numpy()
            N = w.shape[0]
            kp = 64 if l == 0 else (320 if l == 5 else (288 if l == 9 else 256))
            # fragment order of v_mfma_f32_16x16x32_bf16 (csrc/mlp_*.hip): column tiles of 16, k-steps of 32
            n_u4 = (N // 16) * (kp // 32) * 3 * 64
            blk = u16[off * 8:(off + n_u4) * 8].reshape(N // 16, kp // 32, 3, 64, 8)
            tot = bf(blk[:, :, 0]) + bf(blk[:, :, 1]) + bf(blk[:, :, 2])          # [tile][ks][lane][8]
            # lane l of (tile, ks): n = tile*16 + (l & 15), k' = ks*32 + (l >> 4)*8 + 0..7
            got = tot.reshape(N // 16, kp // 32, 4, 16, 8).transpose(0, 3, 1, 2, 4).reshape(N, kp)
            if l == 0:
                ref = np.concatenate([w, np.zeros((N, 1), np.float32)], 1)
            elif l == 5:
                ref = np.concatenate([w[:, :63], np.zeros((N, 1), np.float32), w[:, 63:]], 1)
            elif l == 9:
                ref = np.concatenate([w, np.zeros((N, 5), np.float32)], 1)
            else:
                ref = w
            assert np.array_equal(got.astype(np.float32), ref) and np.array_equal(got, ref.astype(np.float64)), name
            off += n_u4
        assert off * 4 == pf.numel()
        # (b) logits against fp64
        gen = torch.Generator().manual_seed(4)
        P = 4096
        pts = (torch.rand(P, 3, generator=gen) * 2 - 1) * 1.5
        vd = torch.randn(P, 3, generator=gen)
        vd = vd / vd.norm(dim=-1, keepdim=True)
        rays = rays_for_points(pts, vd).cuda()
        sd64 = {k: v.double() for k, v in weights.items()}
        x = torch.cat([O.posenc(pts.double(), 10), O.posenc(rays[:, 8:11].cpu().double(), 4)], -1)
        ref = O.nerf_forward(sd64, x)
        err = {}
        for mode in fn.ops.MATH_MODES:
            fn.ops.set_math(mode)
            p1, _ = fn.ops.mlp_pack(flat)
            raw = fn.ops.mlp_fwd(rays, torch.zeros(P, 1).cuda(), flat, p1)[:, 0].cpu().double()
            err[mode] = float((raw - ref).pow(2).mean().sqrt())
        print('rms logit error vs fp64:', err)
        assert err['bf16x6'] <= 1.25 * err['fp32'] + 1e-9, err
        assert err['bf16x3'] >= 1.5 * err['fp32'], err
    finally:
        fn.ops.set_math(old)
