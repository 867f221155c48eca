"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on
ROCm) across xGMI; gloo on CPU for the world_size-2 tests.

The reference's only multi-GPU strategy is single-process nn.DataParallel around the MLP
(run_nerf.py:82,90).  Here every rank renders its shard of the ray batch end to end
and the ONLY exchange per optimiser step is one all-reduce(SUM) of the flat gradient buffer
(2 x 595,844 fp32 = 4.77 MB: latency-bound on xGMI, so one fused buffer instead of 48 tensors),
plus one all-reduce(MAX) of the per-(image, leaf) error table per subdivide epoch (max is
associative, so the result is bit-identical to the single-GPU table)."""
import atexit
import os

import torch
import torch.distributed as dist


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpus_for_rank(node_cpus, n_sharing, position):
    """The contiguous slice number `position` of `n_sharing` equal slices of a NUMA node's CPU list (the GPUs that hang off one node share
    its cores evenly; SMT siblings are usually listed in the upper half, so a contiguous slice of each half stays on whole cores)."""
    node_cpus = sorted(node_cpus)
    if n_sharing <= 1 or len(node_cpus) < 2 * n_sharing:
        return node_cpus
    half = len(node_cpus) // 2
    out = []
    for part in (node_cpus[:half], node_cpus[half:]):
        k = len(part) // n_sharing
        out += part[position * k:(position + 1) * k]
    return out or node_cpus


def gpu_local_cpus(index):
    """CPUs of the NUMA node GPU `index` hangs off (local_cpulist of its PCI function in sysfs -- the physical bus address, so the answer
    does not depend on HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumbering), or None where sysfs / the property is unavailable."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open('/sys/bus/pci/devices/%s/local_cpulist' % bdf) as f:
            return parse_cpulist(f.read()) or None
    except Exception:      # noqa: BLE001  (an affinity hint, never a failure)
        return None


_AFFINITY = None   # what pin_to_gpu_numa_node did in this process (bench.py reports it)


def pin_to_gpu_numa_node(index):
    """One process per GPU: keep the rank's host threads (the step's enqueue loop, the quadtree's host side, the data-parallel progress thread)
    on the cores next to ITS GPU.  The GPUs sharing a NUMA node split its CPU list evenly, by device order.  FASTNERF_AFFINITY=0 disables;
    anything the kernel refuses is ignored.  Returns the CPU list set, or None."""
    global _AFFINITY
    if os.environ.get('FASTNERF_AFFINITY', 'auto') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    mine = gpu_local_cpus(index)
    if not mine:
        return None
    sharing = [i for i in range(torch.cuda.device_count()) if gpu_local_cpus(i) == mine]
    cpus = cpus_for_rank(mine, len(sharing), sharing.index(index) if index in sharing else 0)
    try:
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed] or sorted(allowed)
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    _AFFINITY = {'device': int(index), 'numa_cpus': len(mine), 'gpus_on_node': len(sharing), 'pinned_cpus': len(cpus), 'first_cpu': cpus[0]}
    return cpus


def affinity_report():
    return _AFFINITY


def select_device(local_rank):
    """The device of a rank: LOCAL_RANK modulo the devices this process can see.  A launcher that masks one GPU per rank
    (HIP_VISIBLE_DEVICES=k) and one that shows all eight both work: the index is taken among the VISIBLE devices."""
    return int(local_rank) % max(1, torch.cuda.device_count())


def init_from_env(device_type=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        use_cuda = torch.cuda.is_available() if device_type is None else device_type == 'cuda'
        if use_cuda:
            torch.cuda.set_device(select_device(local))
            if world <= torch.cuda.device_count():      # (ranks sharing a device -- the 1-GPU plumbing tests -- are not pinned apart)
                pin_to_gpu_numa_node(select_device(local))
        # FASTNERF_DIST_BACKEND=gloo: plumbing tests of the multi-process path on a box with fewer GPUs than ranks
        backend = os.environ.get('FASTNERF_DIST_BACKEND', 'nccl' if use_cuda else 'gloo')
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    _maybe_init_cabi(rk, world)
    return rk, world, local


class CabiComm:
    """One RCCL communicator behind the C ABI (csrc/comm.cpp: fastnerf_comm_* / fastnerf_allreduce_*): the exchange steps of
    the data-parallel path for a host without torch.distributed.  Collectives are enqueued on a HIP stream, never waited for
    on the host."""

    def __init__(self, rank, world, comm_id):
        from . import _lib
        assert len(comm_id) == 128
        self._lib = _lib
        self.rank, self.world = int(rank), int(world)
        h = _lib.P()
        _lib.check(_lib.lib().fastnerf_comm_init(_lib.C.byref(h), comm_id, self.rank, self.world), 'fastnerf_comm_init')
        self._h = h
        self._side = None

    @staticmethod
    def unique_id():
        from . import _lib
        buf = _lib.C.create_string_buffer(128)
        _lib.check(_lib.lib().fastnerf_comm_unique_id(buf), 'fastnerf_comm_unique_id')
        return buf.raw

    def all_reduce_sum(self, flat, scale=1.0, stream=None):
        L = self._lib
        L.require_gpu(flat)
        if flat.dtype != torch.float32 or not flat.is_contiguous():      # (an assert would vanish under python -O and reduce garbage)
            raise TypeError('fastnerf_allreduce_grads takes a contiguous float32 buffer, got %s%s' % (
                flat.dtype, '' if flat.is_contiguous() else ' (non-contiguous)'))
        st = L.stream() if stream is None else stream.cuda_stream
        L.check(L.lib().fastnerf_allreduce_grads(self._h, L.ptr(flat), flat.numel(), float(scale), st), 'fastnerf_allreduce_grads')
        return flat

    def all_reduce_sum_async(self, flat, scale=1.0):
        """The collective on a side stream behind everything enqueued on the current stream so far; .wait() orders the
        current stream behind it (same contract as torch.distributed's async work handle)."""
        if self._side is None:
            self._side = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        self.all_reduce_sum(flat, scale, stream=self._side)
        flat.record_stream(self._side)
        side = self._side

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream().wait_stream(side)
        return _Work()

    def all_reduce_leaf_table(self, table_i32):
        L = self._lib
        L.require_gpu(table_i32)
        assert table_i32.dtype == torch.int32 and table_i32.is_contiguous()
        L.check(L.lib().fastnerf_allreduce_leaf_table(self._h, L.ptr(table_i32), table_i32.numel(), L.stream()),
                'fastnerf_allreduce_leaf_table')
        return table_i32

    def all_reduce_leaf_sumcount(self, sums_f64, counts_i32):
        L = self._lib
        L.require_gpu(sums_f64, counts_i32)
        assert sums_f64.dtype == torch.float64 and counts_i32.dtype == torch.int32 and sums_f64.numel() == counts_i32.numel()
        assert sums_f64.is_contiguous() and counts_i32.is_contiguous()
        L.check(L.lib().fastnerf_allreduce_leaf_sumcount(self._h, L.ptr(sums_f64), L.ptr(counts_i32), sums_f64.numel(), L.stream()),
                'fastnerf_allreduce_leaf_sumcount')
        return sums_f64, counts_i32

    def destroy(self):
        if self._h is not None:
            self._lib.check(self._lib.lib().fastnerf_comm_destroy(self._h), 'fastnerf_comm_destroy')
            self._h = None


_CABI = None   # FASTNERF_COLLECTIVE=cabi: the collectives go through the C ABI instead of torch.distributed


_CABI_NOTE = None   # why FASTNERF_COLLECTIVE=cabi was NOT honoured (shown by collective_route(), printed to stderr once)


def collective_route():
    """Which implementation the data-path collectives of this process go through (bench.py reports it)."""
    if world_size() <= 1:
        return 'none (single process)'
    if _CABI is not None:
        return 'cabi: RCCL behind the C ABI (csrc/comm.cpp)'
    backend = dist.get_backend()
    return 'torch.distributed/' + backend + ('' if _CABI_NOTE is None else ' -- FASTNERF_COLLECTIVE=cabi NOT honoured: ' + _CABI_NOTE)


def _maybe_init_cabi(rk, world):
    """torch.distributed only carries the 128-byte RCCL id from rank 0 to the others (any backend)."""
    global _CABI, _CABI_NOTE
    if _CABI is not None or os.environ.get('FASTNERF_COLLECTIVE', 'torch') != 'cabi':
        return
    why = None
    if not torch.cuda.is_available():
        why = 'no GPU in this process'
    elif world > torch.cuda.device_count():
        # RCCL refuses two ranks of one communicator on the same device: the collectives stay on torch.distributed (whose backend the
        # caller chose) -- said loudly, never silently
        why = '%d ranks share %d device(s); RCCL needs one device per rank' % (world, torch.cuda.device_count())
    if why is not None:
        _CABI_NOTE = why
        import sys
        print('[fastnerf.parallel] rank %d: FASTNERF_COLLECTIVE=cabi requested but %s -> collectives through torch.distributed' % (rk, why),
              file=sys.stderr, flush=True)
        return
    if world > 1:
        box = [CabiComm.unique_id() if rk == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm_id = box[0]
    else:
        comm_id = CabiComm.unique_id()
    _CABI = CabiComm(rk, world, comm_id)
    atexit.register(shutdown_cabi)


def shutdown_cabi():
    """Destroy the C-ABI communicator (idempotent; registered with atexit, call it before dist.destroy_process_group())."""
    global _CABI
    if _CABI is not None:
        try:
            torch.cuda.synchronize()
            _CABI.destroy()
        finally:
            _CABI = None


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _cabi_takes(flat):
    """The C-ABI collective is the gradient exchange: contiguous fp32 device buffers.  Anything else (the fp64 scalar of a logged
    mse, a CPU tensor) goes through torch.distributed, which is initialised in every multi-rank run."""
    return _CABI is not None and flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()


def all_reduce_sum(flat):
    if world_size() > 1:
        if _cabi_takes(flat):
            return _CABI.all_reduce_sum(flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_reduce_sum_async(flat):
    """Start all-reduce(SUM) of a contiguous slice of the gradient buffer and return its work handle: with the RCCL backend
    the collective runs on the process group's own stream behind everything enqueued on the current stream so far, beside
    whatever the caller enqueues next; `wait_all` orders the current stream behind it."""
    if world_size() > 1:
        if _cabi_takes(flat):
            return _CABI.all_reduce_sum_async(flat)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def wait_all(*works):
    for w in works:
        if w is not None:
            w.wait()


def all_reduce_max_int(table_i32):
    """MAX over ranks of the leaf-error table stored as the int32 bit patterns of non-negative
    floats (monotone in the float value) -> exact and order independent."""
    if world_size() > 1:
        if _CABI is not None and table_i32.is_cuda:
            return _CABI.all_reduce_leaf_table(table_i32)
        dist.all_reduce(table_i32, op=dist.ReduceOp.MAX)
    return table_i32


def all_reduce_leaf_sumcount(sums_f64, counts_i32):
    """SUM over ranks of the nerf++ fork's per-(image, leaf) fp64 error sums and int32 ray counts (MEAN split rule,
    nerf++-ours/tree.py:609-632).  ops.leaf_sumcount accumulates multiples of 2^-30: fp64 addition of them is exact, so the
    reduced tables are bit-identical to a single rank's over all rays, whatever the reduction order."""
    if world_size() > 1:
        if _CABI is not None and sums_f64.is_cuda:
            return _CABI.all_reduce_leaf_sumcount(sums_f64, counts_i32)
        dist.all_reduce(sums_f64, op=dist.ReduceOp.SUM)
        dist.all_reduce(counts_i32, op=dist.ReduceOp.SUM)
    return sums_f64, counts_i32


def shard(n, rk=None, world=None):
    """Interleaved shard rows rk::world of a batch of n rays."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    return slice(rk, n, world)


def shard_count(n_rows, batch, rk=None, world=None):
    """Rows of an epoch of n_rows rows that rank rk steps when every batch of `batch` consecutive rows is dealt out rk :: world (the host
    mirror of fastnerf_epoch_shard_rows)."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    per = (batch - rk + world - 1) // world if batch > rk else 0
    full, tail = divmod(int(n_rows), int(batch))
    return full * per + ((tail - rk + world - 1) // world if tail > rk else 0)


def shard_global_rows(n_rows, batch, rk=None, world=None):
    """The epoch rows behind a rank's local rows, in local order (int64 numpy): what QuadTreeManager.gen_rays_device(shard=...) generates
    and what run_nerf.train's `rays[b0 + rank : b1 : world]` slices select, batch after batch."""
    import numpy as np
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    out = [np.arange(b0 + rk, min(b0 + batch, n_rows), world, dtype=np.int64) for b0 in range(0, int(n_rows), int(batch))]
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def barrier():
    if world_size() > 1:
        dist.barrier()


def sync_seed():
    """Rank 0 draws a seed from its torch CPU generator and every rank re-seeds torch (CPU + current GPU) and
    numpy with it, so that RNG-driven host decisions made redundantly on every rank (the epoch's quadtree pixel
    picks, the warm-up coordinates) are identical.  No-op for a single process.  Returns the seed (or None)."""
    if world_size() <= 1:
        return None
    import numpy as np
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(dev)
    dist.broadcast(t, src=0)
    seed = int(t.item())
    torch.manual_seed(seed)
    np.random.seed(seed)
    return seed
