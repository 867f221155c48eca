"""`torch.ops.fastnerf.*`: the C-ABI entry points registered as PyTorch custom ops (fastnerf/torch_ops.py) -- schemas and
fake-tensor shapes on CPU; values and the registered autograd formula on the GPU."""
import os

import numpy as np
import pytest
import torch


def test_registered_schemas_and_fake_shapes():
    import fastnerf
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in fastnerf.torch_ops.OPS:
        op = getattr(torch.ops.fastnerf, name)
        assert str(op.default._schema).startswith('fastnerf::' + name + '(')
    assert 'Tensor(a0!) params' in str(torch.ops.fastnerf.adam_step.default._schema)       # in-place ops declare it
    with FakeTensorMode():
        z, raw, r11 = torch.empty(5, 7), torch.empty(5, 7, 4), torch.empty(5, 11)
        out = torch.ops.fastnerf.raw2outputs(raw, z, r11, None, True)
        assert [tuple(o.shape) for o in out] == [(5, 3), (5,), (5,), (5, 7), (5,)]
        assert torch.ops.fastnerf.sample_pdf_merge(z, z, 3, True, None, 0)[0].shape == (5, 10)
        assert torch.ops.fastnerf.posenc(torch.empty(9, 3), 10).shape == (9, 63)
        assert torch.ops.fastnerf.mlp_fwd(r11, z, torch.empty(3), torch.empty(3)).shape == (5, 7, 4)
    with pytest.raises(RuntimeError):
        torch.ops.fastnerf.posenc(torch.zeros(2, 3), 4)          # no CPU kernel behind the op either


@pytest.mark.gpu
def test_custom_ops_on_the_gpu(golden_dir):
    import fastnerf
    g = np.load(os.path.join(golden_dir, 'g5_raw2out_S64_wb1.npz'))
    raw = torch.from_numpy(g['raw']).cuda().requires_grad_(True)
    z = torch.from_numpy(g['z']).cuda()
    r11 = torch.zeros(64, 11).cuda()
    r11[:, 3:6] = torch.from_numpy(g['rd']).cuda()
    rgb, disp, acc, w, depth = torch.ops.fastnerf.raw2outputs(raw, z, r11, None, True)
    assert np.abs(rgb.detach().cpu().numpy() - g['rgb']).max() < 2e-6
    assert np.abs(w.detach().cpu().numpy() - g['weights']).max() < 2e-6
    (rgb * torch.from_numpy(g['cot']).cuda()).sum().backward()                  # the registered autograd formula
    assert np.abs(raw.grad.cpu().numpy() - g['graw']).max() < 2e-6 * max(1.0, np.abs(g['graw']).max())
    x = torch.rand(100, 3).cuda()
    assert torch.equal(torch.ops.fastnerf.posenc(x, 10), fastnerf.ops.posenc(x, 10))
    draw = torch.randn(3000, 4).cuda()
    draw[::3] = 0
    idx, cnt = torch.ops.fastnerf.compact_live(draw)
    assert cnt.cpu().tolist() == [2000, 3000] and int(idx[0]) == 1
    p, gr, m, v = (torch.randn(1000).cuda() for _ in range(4))
    v.abs_()
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    torch.ops.fastnerf.adam_step(p, gr, m, v, 5e-4, 3, 0.9, 0.999, 1e-8)
    fastnerf.ops.adam_step(p2, gr, m2, v2, 5e-4, 3, 0.9, 0.999, 1e-8)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and not torch.equal(p, gr)
    torch.library.opcheck(torch.ops.fastnerf.sample_coarse.default, (r11, 16, False, False, None, 0), test_utils=('test_schema', 'test_faketensor'))


@pytest.mark.gpu
def test_mlp_fwd_op_checks_the_math_mode_and_r2o_backward_refuses_other_gradients():
    """weights packed under one math mode and used under the other through the dispatcher are an error (the
    dispatcher hands the op fresh tensor objects: the tag lives in ops.mlp_pack's registry); the registered autograd formula
    of raw2outputs implements d/d(raw) through the colour map only and says so when asked for more."""
    import fastnerf
    from fastnerf import ops
    old = ops.get_math()
    try:
        ops.set_math('fp32')
        torch.manual_seed(0)
        net = fastnerf.model.NeRF()
        pf, _ = net.packed()
        r11 = torch.zeros(8, 11).cuda()
        r11[:, 3:6] = torch.randn(8, 3).cuda()
        r11[:, 8:11] = torch.nn.functional.normalize(r11[:, 3:6], dim=-1)
        z = torch.rand(8, 4).cuda() + 2
        raw = torch.ops.fastnerf.mlp_fwd(r11, z, net.flat, pf)
        assert torch.equal(raw, ops.mlp_fwd(r11, z, net.flat, pf))
        ops.set_math('bf16x3')
        with pytest.raises(AssertionError):
            torch.ops.fastnerf.mlp_fwd(r11, z, net.flat, pf)      # fp32-packed weights under the split-bf16 mode
        with pytest.raises(AssertionError):
            torch.ops.fastnerf.mlp_fwd(r11, z, net.flat, torch.empty_like(pf))   # a buffer mlp_pack never produced
    finally:
        ops.set_math(old)
    raw = torch.randn(8, 4, 4).cuda().requires_grad_(True)
    rgb, disp, acc, w, depth = torch.ops.fastnerf.raw2outputs(raw, z, r11, None, False)
    with pytest.raises(NotImplementedError):
        (rgb.sum() + depth.sum()).backward()


@pytest.mark.gpu
def test_packed_tag_does_not_outlive_the_packed_tensor():
    """A block the caching allocator hands to another tensor after the packed weights died carries no math tag."""
    import gc
    import fastnerf
    from fastnerf import ops
    torch.manual_seed(0)
    flat = fastnerf.model.NeRF().flat
    pf, pb = ops.mlp_pack(flat)
    key = (pf.device.index, pf.data_ptr(), pf.numel())
    assert ops.packed_tag(pf) == ops.get_math() and key in ops._PACK_TAGS
    del pf, pb
    gc.collect()
    assert key not in ops._PACK_TAGS
    again = torch.empty(key[2], device='cuda')          # most likely the very block that was just freed
    assert ops.packed_tag(again) is None
