"""One process per GPU: sharding of the naturally independent units of the forward
solver over the ranks of a node.

 * day kernels (`prob_mass`) are independent per day -- the reference maps them over a
   process pool (Run.py:422-425).  Here: broadcast wind + parameters, every rank builds
   days[rank::world] on its GPU, all-gather the COO kernels.
 * whole simulations (MCMC chains, ensemble members) are independent -- round-robin
   over ranks, gather the results on rank 0.
 * the day chain of ONE simulation is a sequential recurrence (the boundary flag decides
   each day whether the state is truncated) and stays on one GPU: replicas only -- except while
   no day raises the flag: then it is a product of spectra and splits over the ranks by days
   with ONE all-gather of a spectrum per rank (`chain_prefix_split`).

Collectives: `torch.distributed` broadcast / all_gather / gather only ("nccl" = RCCL over
xGMI on the GPU box, "gloo" on CPU for the tests).  Messages are small (<~1 MB per day
kernel), nothing here is all-reduce shaped.
"""
import os

import numpy as np
from scipy import sparse


def _dist():
    import torch.distributed as dist
    return dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world)."""
    import torch
    dist = _dist()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            os.environ.setdefault('PARASITOID_DEVICE', str(local))
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def rank_world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _device():
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def shard(items, rank=None, world=None):
    """Round-robin share of `items` for this rank (item i -> rank i % world)."""
    if rank is None:
        rank, world = rank_world()
    return list(items)[rank::world]


def owner(i, world=None):
    if world is None:
        world = rank_world()[1]
    return i % world


# ------------------------------------------------------------------ collectives

def broadcast_array(arr, src=0):
    """Broadcast a numpy array (shape/dtype included) from rank `src`."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return arr
    dev = _device()
    meta = [None]
    if rank == src:
        arr = np.ascontiguousarray(arr)
        meta = [(arr.shape, arr.dtype.str)]
    dist.broadcast_object_list(meta, src=src)
    shape, dt = meta[0]
    if rank == src:
        buf = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev)
    else:
        buf = torch.empty(int(np.prod(shape)) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy().view(np.dtype(dt)).reshape(shape)


def broadcast_wind(wind_data, days, src=0):
    """Broadcast the interpolated wind dict {day: float64[T,3]} as one [ndays,T,3] block."""
    rank, world = rank_world()
    if world == 1:
        return wind_data, days
    if rank == src:
        block = np.stack([wind_data[d] for d in days])
        keys = np.array(days, dtype=np.int64)
    else:
        block = keys = None
    block = broadcast_array(block, src)
    keys = broadcast_array(keys, src)
    return {int(k): block[i] for i, k in enumerate(keys)}, [int(k) for k in keys]


def _pack(mats):
    """list of coo -> (header int64[n,3], row int32, col int32, val float64)"""
    hdr = np.array([(m.shape[0], m.shape[1], m.nnz) for m in mats], dtype=np.int64).reshape(-1, 3)
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
    return (hdr, cat([m.row for m in mats], np.int32), cat([m.col for m in mats], np.int32),
            cat([m.data for m in mats], np.float64))


def _unpack(hdr, row, col, val):
    out, o = [], 0
    for s0, s1, n in hdr:
        out.append(sparse.coo_matrix((val[o:o + n], (row[o:o + n], col[o:o + n])),
                                     shape=(int(s0), int(s1))))
        o += int(n)
    return out


def all_gather_arrays(arr):
    """all_gather of a 1-D/2-D numpy array whose leading length differs per rank."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return [arr]
    dev = _device()
    arr = np.ascontiguousarray(arr)
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    tail = arr.shape[1:]
    pad = np.zeros((mx,) + tail, dtype=arr.dtype)
    pad[:arr.shape[0]] = arr
    buf = torch.from_numpy(pad.view(np.uint8).reshape(-1)).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o.cpu().numpy().view(arr.dtype).reshape((mx,) + tail)[:sizes[r]]
            for r, o in enumerate(outs)]


def all_gather_sparse(local_mats, total):
    """Every rank holds the matrices of items rank, rank+world, ...; returns the full list
    of `total` coo matrices on every rank (all_gather of the packed COO triplets)."""
    rank, world = rank_world()
    if world == 1:
        return list(local_mats)
    parts = [all_gather_arrays(a) for a in _pack([sparse.coo_matrix(m) for m in local_mats])]
    per_rank = [_unpack(parts[0][r], parts[1][r], parts[2][r], parts[3][r]) for r in range(world)]
    return [per_rank[i % world][i // world] for i in range(total)]


def gather_objects(obj, dst=0):
    """Gather small python objects (per-member results) on rank dst."""
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return [obj]
    out = [None] * world if rank == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


# --------------------------------------------------------------- sharded drivers

def prob_mass_sharded(days, wind_data, model_params, start_times=None, build=None):
    """Day kernels for `days`, built days[rank::world] per GPU and all-gathered so that
    every rank ends with the full pmf_list (what Run.main's pool.starmap returns).

    `build(days, wind_data, *model_params, start_times=...)` defaults to the device
    `ParasitoidModel.prob_mass_batch`; the tests inject a CPU stand-in to exercise the
    exchange under gloo."""
    rank, world = rank_world()
    if build is None:
        from . import ParasitoidModel as PM
        build = PM.prob_mass_batch
    if start_times is None:
        start_times = [None] * len(days)
    mine = list(range(len(days)))[rank::world]
    local = build([days[i] for i in mine], wind_data, *model_params,
                  start_times=[start_times[i] for i in mine]) if mine else []
    return all_gather_sparse(local, len(days))


def all_gather_device(t):
    """all_gather of a 1-D torch tensor whose length differs per rank; the pieces stay on `t`'s device.
    RCCL ("nccl") moves them GPU to GPU over xGMI; gloo (the CPU rehearsal backend, which has no
    all_gather for device tensors) stages through the host."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return [t]
    via_host = t.is_cuda and dist.get_backend() != 'nccl'
    w = t.cpu() if via_host else t
    n = torch.tensor([w.numel()], dtype=torch.int64, device=w.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=w.dtype, device=w.device)
    pad[:w.numel()] = w
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    outs = [o[:sizes[r]] for r, o in enumerate(outs)]
    return [o.to(t.device) for o in outs] if via_host else outs


def _device_export(model, days, model_params, start_times):
    """build `days` on this rank's GPU and copy their COO triplets into torch tensors on the same GPU
    (ps_model_export_device: device to device) -> (kshape list, nnz list, row, col, val)"""
    import torch
    from . import _lib as L
    dev = torch.device('cuda', torch.cuda.current_device())
    if not days:
        z = lambda dt: torch.zeros(0, dtype=dt, device=dev)
        return [], [], z(torch.int32), z(torch.int32), z(torch.float64)
    kshape, nnz, _, _ = model.build(days, *model_params, start_times)
    for i in range(len(days)):
        model.check(i)
    tot = int(nnz.sum())
    row = torch.empty(max(tot, 1), dtype=torch.int32, device=dev)
    col = torch.empty(max(tot, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(tot, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    L.check(model._lib.ps_model_export_device(model._h, 0, len(days), row.data_ptr(), col.data_ptr(),
                                              val.data_ptr(), tot))
    return [int(k) for k in kshape], [int(n) for n in nnz], row[:tot], col[:tot], val[:tot]


def prob_mass_sharded_device(model, days, model_params, start_times=None, export=None):
    """Day kernels for `days` without a host round trip: every rank builds days[rank::world] with its
    device model, the COO triplets are all-gathered GPU to GPU (RCCL over xGMI under "nccl") and put in
    day order on every rank's device.  What Run.main's `pool.starmap(prob_mass, ...)` returns
    (Run.py:412-425), resident where the chain will read it.

    -> dict(kshape=int32[nd], off=int64[nd + 1] (numpy, host: sizes only),
            row=int32[...], col=int32[...], val=float64[...] (torch tensors on this rank's device))
    Hand it to `HipSolve.from_device_kernels` / `set_kernels_device`.
    `export(days, model_params, start_times) -> (kshape, nnz, row, col, val)` replaces the device
    builder (the CPU rehearsal of the exchange under gloo)."""
    import torch
    rank, world = rank_world()
    nd = len(days)
    if start_times is None:
        start_times = [None] * nd
    mine = list(range(nd))[rank::world]
    my_days, my_st = [days[i] for i in mine], [start_times[i] for i in mine]
    if export is None:
        ks, nz, row, col, val = _device_export(model, my_days, model_params, my_st)
    else:
        ks, nz, row, col, val = export(my_days, model_params, my_st)
    meta = gather_all_objects((ks, nz))
    rows, cols, vals = all_gather_device(row), all_gather_device(col), all_gather_device(val)
    kshape = np.zeros(nd, dtype=np.int32)
    off = np.zeros(nd + 1, dtype=np.int64)
    pieces = []
    for i in range(nd):
        r, k = i % world, i // world
        lo = int(sum(meta[r][1][:k]))
        n = int(meta[r][1][k])
        kshape[i] = meta[r][0][k]
        off[i + 1] = off[i] + n
        pieces.append((r, lo, lo + n))
    cat = lambda parts: torch.cat([parts[r][lo:hi] for r, lo, hi in pieces]) if pieces else parts[0][:0]
    return dict(kshape=kshape, off=off, row=cat(rows).contiguous(), col=cat(cols).contiguous(),
                val=cat(vals).contiguous())


def gather_all_objects(obj):
    """all_gather of small python objects (sizes, shapes): every rank gets the list in rank order"""
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def run_members(members, fn, dst=0):
    """Round-robin independent simulations (MCMC chains, ensemble members): member i runs
    on rank i % world; results (small python objects) are gathered on rank dst in member
    order.  Returns the ordered list on dst, None elsewhere."""
    rank, world = rank_world()
    mine = list(range(len(members)))[rank::world]
    local = [(i, fn(members[i])) for i in mine]
    parts = gather_objects(local, dst)
    if rank != dst:
        return None
    flat = dict(kv for part in parts for kv in part)
    return [flat[i] for i in range(len(members))]


# ------------------------------------------------------------------ one simulation over G GPUs
def split_days(nd, world):
    """contiguous blocks of days, one per rank, sizes differing by at most one -> [(first, count)]"""
    base, rem = divmod(int(nd), int(world))
    out, first = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((first, c))
        first += c
    return out


class DeviceBlockOps:
    """The two halves of a rank's work in `chain_prefix_split` on a `HipSolve` (fast mode, state set, all day
    kernels uploaded): the block's running products stay in the solver, its total travels as a float64 tensor
    on this rank's device (torch imported before the library touched the GPU)."""

    def __init__(self, solver, negval=1e-8, scale=1.0, renorm=True):
        self.s, self.negval, self.scale, self.renorm = solver, negval, scale, renorm

    def prefix(self, first, count):
        import torch
        ptr, nbytes = self.s.block_prefix(first, count)
        t = torch.empty(nbytes // 8, dtype=torch.float64, device=torch.device('cuda', torch.cuda.current_device()))
        torch.cuda.synchronize()
        self.s.device_copy(t.data_ptr(), ptr, nbytes)
        return t

    def finish(self, first, count, prev):
        return self.s.block_finish(first, count, [t.data_ptr() for t in prev], self.negval, self.scale, self.renorm)


def all_ok(ok):
    """True on every rank iff `ok` was true on every rank (one tiny all-reduce): what a rank asks before it enters
    a collective that a failed rank would never reach."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    if world == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def chain_prefix_split(ops, nd):
    """ONE simulation's flag-free day chain over the ranks (no counterpart in the reference,
    whose CalcSol.py:140-201 loop is sequential): rank g owns the g-th contiguous block of the nd days.
        total = ops.prefix(first, count)        running products of the block's kernel spectra, their last one
        all_gather(total)                       the only exchange: one spectrum per rank, GPU to GPU under RCCL
        ops.finish(first, count, totals[:g])    the block's day records from state x earlier totals x own products
    -> (first, count, flagged): this rank's block, and whether ANY rank saw the boundary flag -- then the split
    does not apply (a flagged day is truncated and re-transformed before the next one) and every rank knows to
    take the sequential route (`run_chain` on one of them).  nd >= world (every rank owns a day)."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    if nd < world:
        raise ValueError('chain_prefix_split: %d days for %d ranks' % (nd, world))
    first, count = split_days(nd, world)[rank]
    # a rank whose own work fails still takes part in every collective up to the point where all ranks know
    # (all_ok) and raise together -- nobody is left waiting in the all-gather
    err, total = None, None
    try:
        total = ops.prefix(first, count)
    except Exception as e:
        err = e
    if not all_ok(err is None):
        raise RuntimeError('chain_prefix_split: the block products failed on a rank%s'
                           % ('' if err is None else ' (this one: %s: %s)' % (type(err).__name__, err)))
    if world > 1:
        via_host = total.is_cuda and dist.get_backend() != 'nccl'      # gloo rehearsal: no device collectives
        w = total.cpu() if via_host else total
        outs = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(outs, w)
        prev = [o.to(total.device) if via_host else o for o in outs[:rank]]
        if total.is_cuda:
            # under "nccl" the call returns once the collective is ENQUEUED on torch's stream; the library reads
            # the gathered totals on the solver's own stream, which knows nothing of that one
            torch.cuda.synchronize()
    else:
        prev = []
    flagged = True
    try:
        flagged = bool(ops.finish(first, count, prev))
    except Exception as e:
        err = e
    if not all_ok(err is None):
        raise RuntimeError('chain_prefix_split: the block records failed on a rank%s'
                           % ('' if err is None else ' (this one: %s: %s)' % (type(err).__name__, err)))
    if world > 1:
        f = torch.tensor([int(flagged)], dtype=torch.int32, device=_device())
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        flagged = bool(int(f.item()))
    return first, count, flagged


def chain_prefix_split_local(solvers, nd, negval=1e-8, scale=1.0, renorm=True):
    """The same split with the "ranks" in ONE process: len(solvers) solvers on one GPU (same state, same
    kernels), the block totals handed over as device pointers instead of through a collective -- what the tests
    and `bench_extras.prefix_split_record` run on the one-GPU box.  -> ([(first, count)], flagged)"""
    blocks = split_days(nd, len(solvers))
    totals = [s.block_prefix(f, c)[0] for s, (f, c) in zip(solvers, blocks)]
    flagged = False
    # last block first: block_finish turns a solver's products into its days' spectra IN PLACE, and the pointers
    # in `totals` are those very buffers (the collective of chain_prefix_split hands out copies instead)
    for g in reversed(range(len(solvers))):
        f, c = blocks[g]
        flagged = solvers[g].block_finish(f, c, totals[:g], negval, scale, renorm) or flagged
    return blocks, flagged
