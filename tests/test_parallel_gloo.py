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


class _NumpyBlockOps:
    """CPU stand-in for parallel.DeviceBlockOps: the same two halves with numpy FFTs on a small torus (the
    exchange and the bookkeeping of chain_prefix_split are what is under test)."""

    def __init__(self, state, kernels):
        import torch
        self.torch = torch
        self.A0 = np.fft.fft2(state)
        self.K = [np.fft.fft2(k) for k in kernels]
        self.fields = {}
        self.prev_seen = None

    def prefix(self, first, count):
        run, self.L = None, []
        for d in range(first, first + count):
            run = self.K[d] if run is None else run * self.K[d]
            self.L.append(run)
        return self.torch.from_numpy(np.ascontiguousarray(self.L[-1]).view(np.float64).ravel().copy())

    def finish(self, first, count, prev):
        P = self.A0
        self.prev_seen = len(prev)
        for t in prev:
            P = P * t.numpy().view(np.complex128).reshape(self.A0.shape)
        for i in range(count):
            self.fields[first + i] = np.fft.ifft2(P * self.L[i]).real
        return False


def _split_worker(rank, world, port, q, fail_rank=None):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    from parasitoids_amd import parallel
    import torch.distributed as dist
    parallel.init('gloo')
    rng = np.random.default_rng(5)
    P, nd = 24, 7
    state = np.zeros((P, P)); state[3, 4] = 1.0
    kernels = []
    for _ in range(nd):
        k = np.zeros((P, P)); k[:3, :3] = rng.random((3, 3)); kernels.append(k / k.sum())
    ops = _NumpyBlockOps(state, kernels)
    if fail_rank is not None:
        if rank == fail_rank:
            ops.prefix = lambda first, count: 1 / 0        # this rank's block products fail
        try:
            parallel.chain_prefix_split(ops, nd)
            q.put((rank, 'no error'))
        except RuntimeError as e:
            q.put((rank, str(e)))
        dist.barrier()
        dist.destroy_process_group()
        return
    first, count, flagged = parallel.chain_prefix_split(ops, nd)
    q.put((rank, dict(first=first, count=count, flagged=flagged, prev=ops.prev_seen, fields=ops.fields)))
    dist.barrier()
    dist.destroy_process_group()


def test_prefix_split_exchange_two_ranks():
    """SURVEY 8e row 2 under gloo, world size 2: rank 0 owns days 0-3, rank 1 days 4-6 and receives exactly one
    block total (rank 0's); the union of their fields is the sequential chain."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (got[0]['first'], got[0]['count'], got[0]['prev']) == (0, 4, 0)
    assert (got[1]['first'], got[1]['count'], got[1]['prev']) == (4, 3, 1)
    assert got[0]['flagged'] is False and got[1]['flagged'] is False
    rng = np.random.default_rng(5)
    P, nd = 24, 7
    A = np.zeros((P, P)); A[3, 4] = 1.0
    fields = {**got[0]['fields'], **got[1]['fields']}
    assert sorted(fields) == list(range(nd))
    for d in range(nd):
        k = np.zeros((P, P)); k[:3, :3] = rng.random((3, 3)); k /= k.sum()
        A = np.fft.ifft2(np.fft.fft2(A) * np.fft.fft2(k)).real          # the sequential chain, day by day
        np.testing.assert_allclose(fields[d], A, rtol=0, atol=1e-14)


def test_prefix_split_failing_rank_raises_everywhere():
    """A rank whose block products fail does not leave the others in the all-gather: every rank raises."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q, 1)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 'failed on a rank' in got[0] and 'this one' not in got[0]
    assert 'failed on a rank' in got[1] and 'ZeroDivisionError' in got[1]


def test_split_days():
    from parasitoids_amd import parallel
    assert parallel.split_days(30, 8) == [(0, 4), (4, 4), (8, 4), (12, 4), (16, 4), (20, 4), (24, 3), (27, 3)]
    assert parallel.split_days(5, 1) == [(0, 5)]
    assert sum(c for _, c in parallel.split_days(17, 6)) == 17
