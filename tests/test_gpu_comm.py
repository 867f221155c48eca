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
