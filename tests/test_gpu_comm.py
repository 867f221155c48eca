"""The exchange steps behind the C ABI (csrc/comm.cpp): RCCL communicator life cycle, all-reduce(SUM) of the gradient buffer with
the global-batch scale, all-reduce(MAX) of the leaf table, on the caller's stream and on a side stream; leaf-table reset / read.
One GPU per box here: the communicator has ONE rank (RCCL refuses two ranks on a device), so what is pinned is the call contract --
identity of a one-rank reduction, the scale, stream ordering, error reporting -- while the multi-rank arithmetic is RCCL's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def test_comm_one_rank_contract(fn):
    P = fn.parallel
    cid = P.CabiComm.unique_id()
    assert isinstance(cid, bytes) and len(cid) == 128 and cid != P.CabiComm.unique_id()
    comm = P.CabiComm(0, 1, cid)
    try:
        g = torch.randn(2 * fn.ops.NET_PARAMS, device='cuda')
        ref = g.clone()
        assert torch.equal(comm.all_reduce_sum(g), ref)                      # SUM over one rank
        comm.all_reduce_sum(g, scale=0.5)
        assert torch.equal(g, ref * 0.5)                                     # the global-batch mean's 1 / world
        # a slice of the buffer on a side stream, ordered behind the producer and ahead of the consumer
        h = torch.zeros_like(ref)
        h.copy_(ref)
        h.mul_(3.0)
        work = comm.all_reduce_sum_async(h[fn.ops.NET_PARAMS:], scale=2.0)
        h[:fn.ops.NET_PARAMS].add_(1.0)                                       # "the coarse net's backward" beside it
        work.wait()
        out = h.clone()
        assert torch.equal(out[fn.ops.NET_PARAMS:], ref[fn.ops.NET_PARAMS:] * 6.0)
        assert torch.equal(out[:fn.ops.NET_PARAMS], ref[:fn.ops.NET_PARAMS] * 3.0 + 1.0)
        t = (torch.rand(100 * 256, device='cuda') * 3).view(torch.int32)
        keep = t.clone()
        assert torch.equal(comm.all_reduce_leaf_table(t), keep)              # MAX over one rank
        empty = torch.empty(0, device='cuda')
        assert comm.all_reduce_sum(empty).numel() == 0
        sums = torch.rand(3 * 64, device='cuda', dtype=torch.float64)
        counts = torch.randint(0, 1000, (3 * 64,), device='cuda', dtype=torch.int32)
        ks, kc = sums.clone(), counts.clone()
        comm.all_reduce_leaf_sumcount(sums, counts)                          # SUM over one rank (nerf++ fork, MEAN rule)
        assert torch.equal(sums, ks) and torch.equal(counts, kc)
    finally:
        comm.destroy()
    comm.destroy()                                                            # idempotent


def test_comm_errors_are_codes_not_aborts(fn):
    lib = fn._lib.lib()
    rc = lib.fastnerf_comm_init(None, b'\0' * 128, 0, 1)
    assert rc != 0 and b'fastnerf_comm_init' in lib.fastnerf_last_error()
    h = fn._lib.P()
    rc = lib.fastnerf_comm_init(fn._lib.C.byref(h), b'\0' * 128, 2, 2)
    assert rc != 0 and b'rank' in lib.fastnerf_last_error()
    rc = lib.fastnerf_allreduce_grads(None, None, 8, 1.0, None)
    assert rc != 0 and b'fastnerf_allreduce_grads' in lib.fastnerf_last_error()


def test_leaf_table_reset_and_read(fn):
    vals = torch.rand(7, 64, device='cuda') * 2
    table = vals.view(torch.int32).clone()
    host = fn.ops.leaf_table_read(table)
    assert host.dtype == torch.float32 and host.device.type == 'cpu' and torch.equal(host, vals.cpu())
    fn.ops.leaf_table_reset(table)
    assert int(table.abs().max()) == 0 and float(fn.ops.leaf_table_read(table).abs().max()) == 0.0
    # the table the loss kernel fills is the one these read: max |gt - pred| per (image, leaf)
    n = 512
    gen = torch.Generator().manual_seed(0)
    pred = torch.rand(n, 3, generator=gen).cuda()
    gt = torch.rand(n, 3, generator=gen).cuda()
    tag = torch.stack([torch.randint(0, 7, (n,), generator=gen), torch.randint(0, 64, (n,), generator=gen)], 1).int().cuda()
    fn.ops.mse_leafmax(pred, None, gt, want_grads=False, leaf_tag=tag, max_leaves=64, table=table.view(-1))
    got = fn.ops.leaf_table_read(table)
    err = (gt - pred).abs().max(-1).values.cpu()
    exp = torch.zeros(7, 64)
    for i in range(n):
        a, b = int(tag[i, 0]), int(tag[i, 1])
        exp[a, b] = max(exp[a, b], float(err[i]))
    assert torch.equal(got, exp)


def test_leaf_sumcount_is_exact_and_shard_independent(fn):
    """The nerf++ fork's SUM / COUNT tables (fastnerf_leaf_sumcount): a ray's term is a multiple of 2^-30, so the fp64 atomics are
    exact -- the table equals the oracle's bit for bit, does not depend on the order of the rays, and the SUM of the tables of any
    sharding (what the all-reduce forms) IS the single-rank table."""
    from oracle import nerf_oracle as O
    n, ni, ml = 20000, 5, 16
    gen = torch.Generator().manual_seed(3)
    pred, gt = torch.rand(n, 3, generator=gen), torch.rand(n, 3, generator=gen)
    tag = torch.stack([torch.randint(0, ni, (n,), generator=gen), torch.randint(0, ml, (n,), generator=gen)], 1).int()

    def tables(idx):
        s = torch.zeros(ni * ml, device='cuda', dtype=torch.float64)
        c = torch.zeros(ni * ml, device='cuda', dtype=torch.int32)
        if len(idx):
            fn.ops.leaf_sumcount(pred[idx].cuda(), gt[idx].cuda(), tag[idx].cuda().contiguous(), ml, s, c)
        return s, c
    s0, c0 = tables(torch.arange(n))
    so, co = O.leaf_loss_sumcount(gt, pred, tag.long(), ni, ml)
    assert torch.equal(s0.cpu(), so) and torch.equal(c0.cpu(), co)
    s1, c1 = tables(torch.randperm(n, generator=gen))
    assert torch.equal(s0, s1) and torch.equal(c0, c1)
    for world in (2, 8):
        parts = [tables(torch.arange(r, n, world)) for r in range(world)]
        s = torch.stack([p[0] for p in parts]).sum(0)
        c = torch.stack([p[1] for p in parts]).sum(0, dtype=torch.int32)
        assert torch.equal(s, s0) and torch.equal(c, c0)
    # the mean it feeds differs from the unrounded fp64 mean by < 2^-31 per ray
    exact = torch.zeros(ni * ml, dtype=torch.float64).index_add_(0, tag[:, 0].long() * ml + tag[:, 1].long(), (gt - pred).abs().double().sum(-1))
    assert float(((s0.cpu() - exact) / c0.cpu().clamp(min=1)).abs().max()) < 2.0 ** -31
