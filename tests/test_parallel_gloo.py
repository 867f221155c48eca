"""Data-parallel logic on CPU (gloo), world sizes 2 and 8: ray sharding (rows r::world, uneven shards, an EMPTY shard) + one
all-reduce(SUM) of the flat gradient (pre-scaled by n_local/N_global) reproduces the full-batch gradient; the two half-buffer
collectives equal the whole-buffer one; the leaf table all-reduce(MAX) on float bit patterns and the nerf++ fork's SUM / COUNT
tables (multiples of 2^-30 in fp64) reproduce the single-rank tables EXACTLY, and so do the split decisions taken from them."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import fastnerf
    from fastnerf import parallel
    from oracle import nerf_oracle as O
    rk, ws, _ = parallel.init_from_env('cpu')
    assert (rk, ws) == (rank, world) and parallel.world_size() == world and parallel.rank() == rank
    gen = torch.Generator().manual_seed(0)          # same data on every rank
    sdc, sdf = O.init_nerf_params(gen), O.init_nerf_params(gen)
    N, S, Ni = (12 if world == 2 else 13), 8, 8      # 13 rays on 8 ranks: shards of 2, 2, 2, 2, 2, 1, 1, 1
    ro = torch.randn(N, 3, generator=gen) * 0.3
    rd = torch.randn(N, 3, generator=gen)
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    tgt = torch.rand(N, 3, generator=gen)
    t_rand, u = torch.rand(N, S, generator=gen), torch.rand(N, Ni, generator=gen)
    tag = torch.stack([torch.randint(0, 2, (N,), generator=gen), torch.randint(0, 4, (N,), generator=gen)], 1)

    def grads_of(sl, scale):
        params = list(sdc.values()) + list(sdf.values())
        for p in params:
            p.requires_grad_(True)
        ret = O.render_rays(rb[sl], sdc, sdf, S, Ni, white_bkgd=True, t_rand=t_rand[sl], u=u[sl])
        loss = (O.img2mse(ret['rgb_map'], tgt[sl]) + O.img2mse(ret['rgb0'], tgt[sl])) * scale
        g = torch.autograd.grad(loss, params)
        for p in params:
            p.requires_grad_(False)
        return torch.cat([x.reshape(-1) for x in g]), ret['rgb_map'].detach()

    full, rgb_full = grads_of(slice(0, N), 1.0)
    sl = parallel.shard(N)
    n_local = len(range(N)[sl])
    local, rgb_local = grads_of(sl, n_local / N)     # what mse_leafmax's grad_scale does on the device
    halves = local.clone()
    parallel.all_reduce_sum(local)
    # what Trainer.step does: the fine net's half while the coarse pass's backward still runs, then the coarse half
    h = halves.numel() // 2
    w1 = parallel.all_reduce_sum_async(halves[h:])
    w0 = parallel.all_reduce_sum_async(halves[:h])
    parallel.wait_all(w1, w0)
    # two ranks: a + b is one addition whichever way the buffer is cut -> bit-equal.  Eight ranks: the collective's reduction order
    # depends on the buffer length (ring segments), so halves and whole agree to rounding only -- every RANK still holds the same
    # bits (checked below through the replicas' digests), which is what keeps the replicas identical
    if world == 2:
        assert torch.equal(halves, local)
    else:
        assert (halves - local).abs().max().item() <= 1e-6 * local.abs().max().item()
    digest = torch.tensor([float(halves.double().sum()), float(local.double().sum())], dtype=torch.float64)
    lo, hi = digest.clone(), digest.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi)                       # every rank reduced to the same bits
    err = (local - full).abs().max().item() / full.abs().max().item()
    # leaf table: per-rank segmented max -> all-reduce(MAX) on the int32 bit patterns
    tab = O.leaf_loss_max(tgt[sl], rgb_local, tag[sl], 2, 4).view(-1).view(torch.int32).clone()
    parallel.all_reduce_max_int(tab)
    tab_full = O.leaf_loss_max(tgt, rgb_full, tag, 2, 4).view(-1)
    same = torch.equal(tab.view(torch.float32), tab_full)
    # nerf++ fork, MEAN rule: per-rank SUM / COUNT tables -> all-reduce(SUM); exact, so the bits equal the single-rank tables ...
    sums, counts = O.leaf_loss_sumcount(tgt[sl], rgb_local, tag[sl], 2, 4)
    parallel.all_reduce_leaf_sumcount(sums, counts)
    sums_full, counts_full = O.leaf_loss_sumcount(tgt, rgb_full, tag, 2, 4)
    same = same and torch.equal(sums, sums_full) and torch.equal(counts, counts_full) and int(counts.sum()) == N
    # ... and the native quadtree manager (host C++, the product's) splits the same leaves from them on every rank
    from fastnerf.tree import QuadTreeManager
    def leaves_after(s_, c_):
        mgr = QuadTreeManager(8, 8, np.eye(3), torch.zeros(2, 8, 8, 3), torch.eye(4)[None, :3, :4].repeat(2, 1, 1), 0.0, 2,
                              device='cpu', criterion='mean')
        assert mgr.max_leaves() == 4
        S2, C2 = s_.view(2, 4).clone(), c_.view(2, 4).clone()
        thres = float((S2.sum() / (3 * C2.sum().clamp(min=1))))          # the global mean: some leaves above, some below
        mgr.adjust_tree_from_sumcount(S2, C2, thres)
        return [np.asarray(mgr.leaves(i)).tolist() for i in range(2)]
    lv = leaves_after(sums, counts)
    same = same and lv == leaves_after(sums_full, counts_full) and sum(len(x) for x in lv) > 8
    # a rank whose shard is EMPTY (N = 1 on world 2: rank 1; several ranks on world 8) still joins and contributes zeros
    sl1 = parallel.shard(1)
    z = torch.ones(3) * (len(range(1)[sl1]))
    parallel.all_reduce_sum(z)
    same = same and z.tolist() == [1.0, 1.0, 1.0]
    # epoch seed broadcast (train() calls it before every host-side random decision): ranks start from DIFFERENT generator
    # states and end up drawing identical torch / numpy numbers
    torch.manual_seed(1000 + 17 * rank)
    np.random.seed(5 + rank)
    seed = parallel.sync_seed()
    draws = (torch.randint(0, 10 ** 9, (4,)).tolist(), np.random.randint(0, 10 ** 9, 4).tolist(), int(seed))
    # the epoch's rows dealt out batch by batch (run_nerf.train with the device generator: every rank generates ONLY its rows): the ranks'
    # global-row lists, gathered, are a partition of the epoch, each equals the strided slices of the unsharded loop, and the C ABI's host
    # arithmetic (fastnerf_epoch_shard_rows) counts them -- an epoch of 10 007 rows in batches of 1920 (lego.txt:16; not a multiple of 8)
    from fastnerf import _lib
    Nrows, batch = 10007, 1920
    mine = parallel.shard_global_rows(Nrows, batch)
    want = np.concatenate([np.arange(b0 + rank, min(b0 + batch, Nrows), world) for b0 in range(0, Nrows, batch)])
    same = same and np.array_equal(mine, want) and len(mine) == parallel.shard_count(Nrows, batch) == int(
        _lib.lib().fastnerf_epoch_shard_rows(Nrows, batch, rank, world))
    box = [None] * world
    torch.distributed.all_gather_object(box, mine.tolist())
    same = same and sorted(i for part in box for i in part) == list(range(Nrows))
    parallel.barrier()
    q.put((rank, err, bool(same), n_local, draws))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_gradient_allreduce_and_tables(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    N = 12 if world == 2 else 13
    for rank, err, same, n_local, draws in res:
        assert err < 1e-5, (rank, err)      # summation order only
        assert same, rank
        assert n_local == len(range(rank, N, world))
    assert sum(r[3] for r in res) == N and min(r[3] for r in res) == (6 if world == 2 else 1)   # uneven shards on 8 ranks
    assert all(r[4] == res[0][4] for r in res)           # same seed, same torch and numpy draws on every rank


def test_shard_covers_batch():
    sys.path.insert(0, ROOT)
    from fastnerf import parallel
    for n in (1, 7, 4096, 4097):
        for world in (1, 2, 8):
            idx = sorted(i for r in range(world) for i in range(n)[parallel.shard(n, r, world)])
            assert idx == list(range(n))


def test_affinity_helpers_and_shard_arithmetic():
    """The host-side pieces of the multi-GPU hardening that need no GPU: sysfs cpulist parsing, the even split of a NUMA node's CPUs among
    the GPUs that share it, and the shard row arithmetic against brute force (incl. batches smaller than the world)."""
    sys.path.insert(0, ROOT)
    from fastnerf import _lib, parallel
    assert parallel.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and parallel.parse_cpulist('') == []
    node = list(range(0, 32)) + list(range(128, 160))      # 32 cores + their SMT siblings
    parts = [parallel.cpus_for_rank(node, 4, k) for k in range(4)]
    assert sorted(c for p in parts for c in p) == node and all(len(p) == 16 for p in parts)
    assert parts[1] == list(range(8, 16)) + list(range(136, 144))
    assert parallel.cpus_for_rank([3, 1, 2], 1, 0) == [1, 2, 3] and parallel.cpus_for_rank([0, 1], 4, 3) == [0, 1]
    assert parallel.affinity_report() is None and parallel.select_device(5) == 5 % max(1, torch.cuda.device_count())
    for n, batch, world in ((0, 4, 2), (5, 1920, 8), (10007, 1920, 8), (4096, 4096, 8), (4097, 1024, 3), (100, 3, 8)):
        got = []
        for rk in range(world):
            rows = parallel.shard_global_rows(n, batch, rk, world)
            assert len(rows) == parallel.shard_count(n, batch, rk, world) == int(_lib.lib().fastnerf_epoch_shard_rows(n, batch, rk, world))
            got += rows.tolist()
        assert sorted(got) == list(range(n))
    assert int(_lib.lib().fastnerf_epoch_shard_rows(10, 4, 2, 2)) == -1      # row0 must be below the stride
