"""world_size-2 gloo tests (CPU) of the N>1 path: day sharding with broadcast +
all-gather of COO kernels, member round-robin with gather.
The per-day compute is the CPU oracle here (test stand-in for the device builder):
what is under test is the exchange, which is identical under RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, golden_dir, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    from parasitoids_amd import parallel
    from parasitoids_amd import ParasitoidModel as PM
    from oracle import model as OM
    from helpers import HP, DP, DLP, MU_R, NPER
    r, w = parallel.init('gloo')
    assert (r, w) == (rank, world)
    # only rank 0 reads the wind file; everyone gets it by broadcast
    if rank == 0:
        wd, days = PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '00:00')
    else:
        wd, days = None, None
    wd, days = parallel.broadcast_wind(wd, days)

    def build(ds, wind, *params, start_times=None):
        return [OM.prob_mass(d, wind, *params, start_time=s) for d, s in zip(ds, start_times)]

    params = (HP, DP, DLP, MU_R, NPER, 10000.0, 32)
    pmfs = parallel.prob_mass_sharded(days[:5], wd, params, build=build)
    members = [dict(mu_r=1.0 + 0.1 * i) for i in range(5)]
    res = parallel.run_members(members, lambda m: (rank, round(m['mu_r'] * 10)))
    import torch
    import torch.distributed as dist
    # the device-resident exchange (prob_mass_sharded_device) with a CPU stand-in for the exporter: each
    # rank contributes the triplets of ITS days only, the result is every day in order on every rank
    by_day = dict(zip(days[:5], pmfs))

    def export(ds, model_params, sts):
        mats = [by_day[d].tocoo() for d in ds]
        cat = lambda xs, dt: torch.from_numpy(np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
        return ([m.shape[0] for m in mats], [m.nnz for m in mats], cat([m.row for m in mats], np.int32),
                cat([m.col for m in mats], np.int32), cat([m.data for m in mats], np.float64))

    g = parallel.prob_mass_sharded_device(None, days[:5], params, export=export)
    dev = dict(kshape=g['kshape'].tolist(), off=g['off'].tolist(), row=g['row'].numpy(), col=g['col'].numpy(),
               val=g['val'].numpy())
    out = dict(days=days, nnz=[p.nnz for p in pmfs], sums=[float(p.sum()) for p in pmfs], dev=dev,
               shapes=[p.shape for p in pmfs], first=pmfs[0].toarray(), last=pmfs[4].toarray(),
               wind_sum=float(sum(v.sum() for v in wd.values())), res=res)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_day_sharding_and_member_gather(golden_dir):
    import torch.multiprocessing as mp
    from oracle import model as OM
    from helpers import HP, DP, DLP, MU_R, NPER
    from parasitoids_amd import ParasitoidModel as PM
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wd, days = PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '00:00')
    ref = [OM.prob_mass(d, wd, HP, DP, DLP, MU_R, NPER, 10000.0, 32) for d in days[:5]]
    for rank in (0, 1):
        o = got[rank]
        assert o['days'] == days
        assert o['wind_sum'] == float(sum(v.sum() for v in wd.values()))
        assert o['nnz'] == [p.nnz for p in ref]
        assert o['shapes'] == [p.shape for p in ref]
        assert np.array_equal(o['first'], ref[0].toarray())      # built on rank 0
        assert np.array_equal(o['last'], ref[4].toarray())       # built on rank 0 (4 % 2)
        d = o['dev']                                             # the device-style exchange: all days, in order
        assert d['kshape'] == [p.shape[0] for p in ref]
        assert d['off'] == [0] + list(np.cumsum([p.nnz for p in ref]))
        assert np.array_equal(d['row'], np.concatenate([p.row for p in ref]))
        assert np.array_equal(d['col'], np.concatenate([p.col for p in ref]))
        assert np.array_equal(d['val'], np.concatenate([p.data for p in ref]))
    assert got[1]['res'] is None
    assert got[0]['res'] == [(i % 2, 10 + i) for i in range(5)]  # member i ran on rank i % 2


def test_shard_and_owner():
    from parasitoids_amd import parallel
    assert parallel.shard(range(10), 1, 4) == [1, 5, 9]
    assert parallel.shard(range(3), 3, 4) == []
    assert [parallel.owner(i, 8) for i in (0, 7, 8, 513)] == [0, 7, 0, 1]
    import scipy.sparse as sp
    m = [sp.random(7, 7, 0.3, format='coo', random_state=i) for i in range(3)]
    back = parallel._unpack(*parallel._pack(m))
    for a, b in zip(m, back):
        assert (a != b).nnz == 0
