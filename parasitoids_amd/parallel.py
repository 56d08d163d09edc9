"""One process per GPU: sharding of the naturally independent units of the forward
solver over the ranks of a node.

 * day kernels (`prob_mass`) are independent per day -- the reference maps them over a
   process pool (Run.py:422-425).  Here: broadcast wind + parameters, every rank builds
   days[rank::world] on its GPU, all-gather the COO kernels.
 * whole simulations (MCMC chains, ensemble members) are independent -- round-robin
   over ranks, gather the results on rank 0.
 * the day chain of ONE simulation is a sequential recurrence (the boundary flag decides
   each day whether the state is truncated) and stays on one GPU: replicas only.

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
