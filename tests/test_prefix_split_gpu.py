"""One simulation's flag-free day chain split over "ranks" by days (SURVEY.md 8e row 2:
ps_chain_block_prefix / ps_chain_block_finish, parallel.chain_prefix_split).  The ranks are solvers of one
process here (parallel.chain_prefix_split_local: block totals handed over as device pointers; the collective
itself is covered by tests/test_parallel_gloo.py).  The split re-associates the spectral products, so its
fields are compared with the sequential chain's -- and with the oracle's -- to a tolerance, not bit for bit."""
import numpy as np
import pytest
from scipy import sparse

from oracle import calcsol as OC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip_lib():
    from parasitoids_amd import hip_lib
    return hip_lib


def _stack(R, K, nd, seed=7):
    from parasitoids_amd import synthetic
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=seed, sigma=(6.0, 12.0), shift=10)
    return kernels


def test_split_chain_matches_the_sequential_chain_and_the_oracle(hip_lib, monkeypatch):
    from parasitoids_amd import parallel
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 12
    N = 2 * R + 1
    kernels = _stack(R, K, nd + 1)           # one more day than the split covers: the chain goes on from its end
    state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
    ref, trace = [state], {}
    OC.get_solutions(ref, [None] + kernels, list(range(nd + 2)), nd + 2, N, np.array([K, K]), trace=trace)
    assert not any(trace['flags'])
    seq = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
    assert seq.fft_len == 1008
    seq.set_kernels(kernels)
    seq.run_chain(renorm=True)
    seq_fields = [seq.dense(0, d) for d in range(nd + 1)]
    seq_stats = seq.chain_stats(0, nd + 1)
    seq.close()
    for G in (1, 2, 3, 5):
        solvers = [hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True) for _ in range(G)]
        for s in solvers:
            s.set_kernels(kernels)
        blocks, flagged = parallel.chain_prefix_split_local(solvers, nd)
        assert blocks == parallel.split_days(nd, G) and sum(c for _, c in blocks) == nd
        assert not flagged
        for s, (first, count) in zip(solvers, blocks):
            st = s.chain_stats(first, count)
            for i in range(count):
                d = first + i
                got = s.dense(0, d)
                scale = max(1.0, float(np.abs(seq_fields[d]).max()))
                assert np.abs(got - seq_fields[d]).max() <= 1e-14 * scale, (G, d)       # rounding only
                np.testing.assert_allclose(got, trace['raw'][d], rtol=0, atol=1e-12)       # the oracle's raw field
                assert not st[i].flag
                assert abs(st[i].sum - seq_stats[d].sum) <= 1e-13 and abs(st[i].nnz - seq_stats[d].nnz) <= 2
                sol = s.chain_solution(d, st[i])
                assert abs(sol.tocsr() - ref[d + 1].tocsr()).max() < 1e-12
        # the last "rank"'s solver holds the state after the last day of the split: the chain goes on from there
        last = solvers[-1]
        last.run_chain(nd, 1, renorm=True)
        assert np.abs(last.dense(0, nd) - seq_fields[nd]).max() <= 1e-14 * max(1.0, float(np.abs(seq_fields[nd]).max()))
        for s in solvers:
            s.close()


def test_a_flag_voids_the_split_and_the_solver_still_runs_the_chain(hip_lib, monkeypatch):
    from parasitoids_amd import parallel
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 10
    N = 2 * R + 1
    kernels = _stack(R, K, nd)
    edge = sparse.coo_matrix(([1.0], ([760], [760])), shape=(N, N))        # mass next to the boundary: flags
    ref, trace = [edge], {}
    OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, np.array([K, K]), trace=trace)
    assert any(trace['flags'])
    solvers = [hip_lib.HipSolve(edge, [K, K], mode='fast', chain_only=True) for _ in range(2)]
    for s in solvers:
        s.set_kernels(kernels)
    _, flagged = parallel.chain_prefix_split_local(solvers, nd)
    assert flagged
    s = solvers[0]                           # the sequential route on one of them
    s.set_state(edge)
    s.run_chain(renorm=True)
    st = s.chain_stats(0, nd)
    for d in range(nd):
        np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=1e-12)
        assert bool(st[d].flag) == bool(trace['flags'][d])
    for s in solvers:
        s.close()


def test_block_calls_refuse_what_they_cannot_do(hip_lib, monkeypatch):
    from parasitoids_amd import _lib as L
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 4
    N = 2 * R + 1
    kernels = _stack(R, K, nd)
    state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
    s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
    s.set_kernels(kernels)
    with pytest.raises(L.HipError) as e:
        s.block_finish(0, nd)                                   # nothing prepared
    assert e.value.code == L.PS_ERR_STATE
    with pytest.raises(L.HipError) as e:
        s.block_prefix(2, nd)                                   # days beyond the uploaded kernels
    assert e.value.code == L.PS_ERR_STATE
    ptr, nbytes = s.block_prefix(0, nd)
    assert ptr and nbytes == s.fft_len * (((s.fft_len // 2 + 1) + 7) // 8 * 8) * 16
    with pytest.raises(L.HipError):
        s.block_finish(1, nd - 1)                               # another block than the prepared one
    assert s.block_finish(0, nd) is False
    s.close()
    x = hip_lib.HipSolve(state, [K, K], mode='exact', chain_only=True)
    x.set_kernels(kernels)
    with pytest.raises(L.HipError) as e:
        x.block_prefix(0, nd)
    assert e.value.code == L.PS_ERR_UNSUPPORTED
    x.close()


_DEVICE_OPS = r'''
import os, sys
import torch                                   # first: see INTEGRATION.md (torch brings its own HIP runtime)
assert torch.cuda.is_available()
root = sys.argv[1]
sys.path.insert(0, root)
os.environ['PS_TPIPE'] = '1'
import numpy as np
from scipy import sparse
from parasitoids_amd import hip_lib, parallel, synthetic
R, K, nd = 400, 401, 8
N = 2 * R + 1
_, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
seq = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
seq.set_kernels(kernels); seq.run_chain(renorm=True)
s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
s.set_kernels(kernels)
ops = parallel.DeviceBlockOps(s)
first, count, flagged = parallel.chain_prefix_split(ops, nd)          # world size 1: the whole chain is one block
assert (first, count, flagged) == (0, nd, False)
for d in range(nd):
    assert np.abs(s.dense(0, d) - seq.dense(0, d)).max() <= 1e-16
# the block total as the collective would carry it: a float64 tensor on the device, the layout's size
t = ops.prefix(0, 3)
assert t.is_cuda and t.dtype == torch.float64 and t.numel() * 8 == s.block_prefix(0, 3)[1]
assert bool(torch.isfinite(t[: 2 * (s.fft_len // 2 + 1) * s.fft_len]).all())
print('DEVICE_OPS_OK')
'''


def test_chain_prefix_split_on_device_tensors():
    """parallel.chain_prefix_split with DeviceBlockOps in a process that imports torch first (as a
    torch.distributed driver does): world size 1 -- one block, no exchange -- equals the sequential chain; the
    block total travels as a device tensor of the spectrum's size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, '-c', _DEVICE_OPS, root], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and 'DEVICE_OPS_OK' in p.stdout, p.stderr[-3000:]
