"""GPU parity tests of the day-chain solver (through the C ABI) against the oracle
and the reference's golden vectors.  Mirrors the reference's own
tests/test_CalcSol.py (test_cuda_convolve, test_cuda_back_solve) and adds the
chain fixtures G6-G8.  Tolerances are fp64: 1e-12 absolute on probability
fields (values <= 1), relative 1e-12 on population fields."""
import numpy as np
import pytest
from scipy import signal, sparse

from oracle import calcsol as OC
from helpers import coo_from, recentre, assert_summary

pytestmark = pytest.mark.gpu

ATOL = 1e-12


@pytest.fixture(scope='module')
def hip_lib():
    from parasitoids_amd import hip_lib
    return hip_lib


@pytest.fixture(scope='module')
def two_arrays():
    A = np.outer(range(10), range(1, 11))
    B = np.outer(range(4, -1, -1), range(8, -1, -2))
    return (A, B)


@pytest.fixture(scope='module')
def many_arrays(golden):
    g = golden('g8_back_solve')
    return tuple(g['toy_' + k] for k in 'ABCD')


def test_hip_convolve(hip_lib, two_arrays):
    '''reference tests/test_CalcSol.py:100-113 (test_cuda_convolve)'''
    A, B = two_arrays
    max_shape = np.array(A.shape) + 6
    solver = hip_lib.HipSolve(sparse.coo_matrix(A), max_shape)
    solver.fftconv2(sparse.csr_matrix(B))
    C = solver.get_cursol(A.shape)
    assert np.allclose(C.toarray(), signal.fftconvolve(A, B, 'same'), rtol=1e-12, atol=1e-10)
    assert np.all(A == np.outer(range(10), range(1, 11)))
    assert np.all(B == np.outer(range(4, -1, -1), range(8, -1, -2)))


def test_hip_back_solve(hip_lib, many_arrays, golden):
    '''reference tests/test_CalcSol.py:141-171 (test_cuda_back_solve), fp64 tolerance'''
    A, B, C, D = many_arrays
    solver = hip_lib.HipSolve(sparse.coo_matrix(C), A.shape)
    solver.fftconv2(sparse.csr_matrix(D))
    bck = solver.back_solve([sparse.csr_matrix(A), sparse.csr_matrix(B)], A.shape)
    g = golden('g8_back_solve')
    # the reference's CPU back_solve returns unthresholded fields; ours drops < 1e-8
    for got, ref in ((bck[0], g['toy_bck0']), (bck[1], g['toy_bck1'])):
        ref = np.where(ref < 1e-8, 0.0, ref)
        np.testing.assert_allclose(got.toarray(), ref, rtol=0, atol=1e-11)


def test_spectrum_roundtrip(hip_lib, two_arrays):
    '''CalcSol.fft2 parity at the spectrum level (CalcSol.py:11-24)'''
    A, B = two_arrays
    solver = hip_lib.HipSolve(sparse.coo_matrix(A), B.shape)
    ref = OC.fft2(sparse.coo_matrix(A), np.array(B.shape))
    got = solver.get_spectrum()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10)
    OC.fftconv2(ref, sparse.csr_matrix(B))
    solver.fftconv2(sparse.csr_matrix(B))
    np.testing.assert_allclose(solver.get_spectrum(), ref, rtol=0, atol=1e-8)


@pytest.mark.parametrize('R,mode', [(128, 'exact'), (200, 'exact'), (128, 'fast'), (200, 'fast'),
                                    (128, 'fold'), (200, 'fold'), (128, 'auto'), (200, 'auto')])
def test_get_solutions_chain(hip_lib, golden, R, mode):
    '''G6: CalcSol.get_solutions on Kalbar kernels.  R=128: P=364=4*7*13 (no flags);
    R=200: P=573=3*191 (generic radix, flags fire on 16 of 17 days).'''
    g = golden('g6_solutions')
    tag = 'r%d' % R
    nd = int(g[tag + '_ndays'])
    pmfs = [coo_from(g, '%s_pmf%d' % (tag, i)) for i in range(nd)]
    ms = g[tag + '_max_shape']
    N = 2 * R + 1
    first = recentre(pmfs[0], R)
    # oracle with the unthresholded states
    trace = {}
    modelsol = [first]
    OC.get_solutions(modelsol, pmfs, list(range(nd)), nd, N, ms, trace=trace)

    solver = hip_lib.HipSolve(first, ms, mode=mode, chain_only=(mode == 'auto'))
    assert solver.pad_shape == (N + ms[0] // 2, N + ms[1] // 2)
    if mode == 'auto':       # PS_MODE_AUTO: fast torus while clean, folded reference torus after
        assert solver.mode == 'auto' and solver.fft_len >= N + ms[0] // 2
    if mode == 'fold':       # linear convolution on a fast size, folded back onto the reference torus
        assert solver.fft_len >= N + 3 * (ms[0] // 2) and solver.mode == 'fold'
    solver.set_kernels(pmfs[1:])
    solver.run_chain(0, nd - 1, negval=1e-8, scale=1.0, renorm=True)
    stats = solver.chain_stats(0, nd - 1)
    exact = mode in ('exact', 'fold', 'auto')         # all on the reference torus
    if mode == 'auto':
        # the hand-over happens on the first day with anything above 1e-15 outside the domain:
        # at or before the first flagged day, and the later days ran on the fold path
        f, fold_fft = solver.auto_info()
        flags = [bool(v) for v in g[tag + '_flags'][:nd - 1]]
        if any(flags):
            assert 0 <= f <= flags.index(True)
            route = solver.auto_route(0, nd - 1)
            assert np.all(route[:f] == 0) and np.all(route[f:] >= 1)
            # flagged days past the prefix ran on the wide fast torus or (right after dusty days) in the fold child
            if np.any(route == 2):
                assert fold_fft >= N + 3 * (ms[0] // 2)
    tol = ATOL if exact else 5e-8             # fast mode: pad-region semantics differ
    pos = g[tag + '_pos']
    for n in range(nd - 1):
        raw = solver.dense(0, n)
        np.testing.assert_allclose(raw, trace['raw'][n], rtol=0, atol=tol)
        np.testing.assert_allclose(raw[pos[:, 0], pos[:, 1]], g['%s_rawsamp%d' % (tag, n + 1)],
                                   rtol=0, atol=tol)
        if exact:
            assert bool(stats[n].flag) == bool(g[tag + '_flags'][n])
            sol = solver.chain_solution(n, stats[n])
            assert_summary(g, '%s_sum%d' % (tag, n + 1), sol, pos, rtol=1e-11, atol=1e-12,
                           nnz_slack=2)
            ref = modelsol[n + 1].tocsr()
            assert abs(sol.tocsr() - ref).max() < 1e-12
            assert abs(sol.sum() - 1.0) < 1e-12


def test_get_cursol_matches_chain(hip_lib, golden):
    '''per-call API (fftconv2 + get_cursol) == fused chain, bit for bit'''
    g = golden('g6_solutions')
    R, nd = 128, 4
    pmfs = [coo_from(g, 'r128_pmf%d' % i) for i in range(nd)]
    ms = g['r128_max_shape']
    first = recentre(pmfs[0], R)
    a = hip_lib.HipSolve(first, ms)
    a.set_kernels(pmfs[1:])
    a.run_chain(renorm=True)
    b = hip_lib.HipSolve(first, ms)
    for n in range(nd - 1):
        b.fftconv2(pmfs[n + 1].tocsr())
        sol = b.get_cursol([2 * R + 1] * 2)
        ref = a.dense(0, n)
        got = b.dense(0, 0)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-15)
        assert sol.nnz == a.chain_stats(n, 1)[0].nnz


def test_errors(hip_lib, two_arrays):
    A, B = two_arrays
    solver = hip_lib.HipSolve(sparse.coo_matrix(A), np.array(A.shape) + 6)
    with pytest.raises(Exception):
        solver.fftconv2(sparse.csr_matrix(np.ones((4, 4))))   # even kernel, CalcSol.py:58
    with pytest.raises(ValueError):
        solver.get_cursol((11, 11))


def test_error_classes_and_auto_mode(hip_lib):
    '''C-ABI error classes surface as HipError codes; auto mode falls back to the fast
    size only when the reference pad cannot be planned (prime factor > 1024).'''
    from parasitoids_amd import _lib as L
    import ctypes as C
    N = 1001
    one = sparse.coo_matrix(([1.0], ([500], [500])), shape=(N, N))
    # P = 1001 + 128 = 1129 is prime: exact is unsupported, auto picks the fast size
    with pytest.raises(L.HipError) as ei:
        hip_lib.HipSolve(one, [257, 257], mode='exact')
    assert ei.value.code == L.PS_ERR_UNSUPPORTED
    s = hip_lib.HipSolve(one, [257, 257], mode='auto')
    assert s.mode == 'fast' and s.fft_len >= 1129 and s.pad_shape == (1129, 1129)
    with pytest.raises(L.HipError) as ei:
        s.run_chain(0, 1)                                  # nothing uploaded
    assert ei.value.code == L.PS_ERR_STATE
    with pytest.raises(L.HipError) as ei:
        s.set_kernels([sparse.coo_matrix(([1.0], ([0], [0])), shape=(259, 259))])   # > max_shape
    assert ei.value.code == L.PS_ERR_BAD_SHAPE
    with pytest.raises(L.HipError) as ei:
        s.dense(L.REC_BACK, 3)                              # record does not exist
    assert ei.value.code == L.PS_ERR_STATE
    s.set_kernels([sparse.coo_matrix(([0.5, 0.5], ([128, 128], [128, 130])), shape=(257, 257))])
    s.run_chain(renorm=True)
    st = s.chain_stats(0, 1)[0]
    row = np.empty(1, dtype=np.int32); col = np.empty(1, dtype=np.int32); val = np.empty(1)
    nnz = C.c_int64()
    rc = L.load().ps_record_fetch_coo(s._h, 0, 0, 1e-8, 1.0, st.delta, 1.0, L.p_i32(row), L.p_i32(col),
                                      L.p_f64(val), 1, C.byref(nnz))
    assert rc == L.PS_ERR_BAD_ARG and nnz.value == 2        # capacity too small, count reported
    got = s.chain_solution(0, st)
    assert got.nnz == 2 and abs(got.sum() - 1.0) < 1e-15
    assert sorted(zip(got.row.tolist(), got.col.tolist())) == [(500, 500), (500, 502)]
    v = s.gather(0, 0, [500, 500, 0], [500, 501, 0])
    assert abs(v[0] - 0.5) < 1e-15 and v[1] == 0.0 and v[2] == 0.0
    s.close()
    a = hip_lib.HipSolve(sparse.coo_matrix(([1.0], ([16], [16])), shape=(33, 33)), [9, 9], mode='auto')
    assert a.mode == 'exact' and a.fft_len == 37               # 37 is prime but <= 1024: planned
    a.close()


def test_flag_speculation_is_exact(hip_lib, monkeypatch):
    '''ps_chain_run enqueues windows of days without the flag-conditional re-FFT launches and
    checks the pad maxima afterwards; when a flag turns up inside a window the days after it
    are redone.  The result must be bit-identical to the non-speculative chain, whichever
    day the first flag falls on.'''
    from parasitoids_amd import synthetic
    R, K, nd = 150, 101, 12
    N = 2 * R + 1
    for start in (150, 230, 262):          # first flag late / mid-window / early
        _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(3.0, 6.0), shift=4)
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        runs = []
        for spec in (True, False):
            if spec:
                monkeypatch.delenv('PS_NO_SPECULATION', raising=False)
            else:
                monkeypatch.setenv('PS_NO_SPECULATION', '1')
            s = hip_lib.HipSolve(state, [K, K], mode='exact')
            s.set_kernels(kernels)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            runs.append(([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st]))
            s.close()
        flags = [f for f, _, _, _ in runs[1][1]]
        assert runs[0][1] == runs[1][1]
        for a, b in zip(runs[0][0], runs[1][0]):
            assert np.array_equal(a, b)
        if start != 150:
            assert any(flags) and not flags[0]      # the first flag falls inside a later window


def test_random_small_chains_against_oracle(hip_lib):
    '''Forty random small problems (domain 21..301, kernel shape 3..N, 2..5 days, kernels with
    random support, sometimes mass leaving the domain) through the whole chain in exact mode
    against the oracle: raw fields, flags, thresholded solutions.  Sizes are whatever the
    random draw gives -- odd primes, powers of two plus one, pads with generic radices.'''
    rng = np.random.default_rng(20260104)
    checked_flags = 0
    for case in range(40):
        R = int(rng.integers(10, 151))
        N = 2 * R + 1
        K = 2 * int(rng.integers(1, min(R, 60) + 1)) + 1
        nd = int(rng.integers(2, 6))
        kernels = []
        for _ in range(nd):
            k = rng.random((K, K)) * (rng.random((K, K)) < rng.uniform(0.05, 0.6))
            k[K // 2, K // 2] += 0.5
            k /= k.sum()
            kernels.append(sparse.coo_matrix(k))
        npts = int(rng.integers(1, 6))
        rows = rng.integers(0, N, npts); cols = rng.integers(0, N, npts)
        vals = rng.random(npts); vals /= vals.sum()
        state = sparse.coo_matrix((vals, (rows, cols)), shape=(N, N))
        state.sum_duplicates()
        ms = np.array([K, K])
        P = N + K // 2
        ref = [state]
        trace = {}
        OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, ms, trace=trace)
        # direct transform on P / linear convolution folded onto P / fast torus while clean, then fold
        for mode in ('exact', 'fold', 'auto'):
            s = hip_lib.HipSolve(state, ms, mode=mode, chain_only=(mode == 'auto'))
            s.set_kernels(kernels)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            for d in range(nd):
                np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=1e-13,
                                           err_msg='%s case %d N=%d K=%d P=%d day %d' % (mode, case, N, K, P, d))
                assert bool(st[d].flag) == bool(trace['flags'][d]), (mode, case, N, K, d)
                got = s.chain_solution(d, st[d]).tocsr()
                assert abs(got - ref[d + 1].tocsr()).max() < 1e-12
            s.close()
        checked_flags += int(any(trace['flags']))
    assert checked_flags >= 5


def test_auto_mode_routes(hip_lib):
    """PS_MODE_AUTO: a chain whose mass stays away from the boundary never leaves the fast torus
    (and equals the exact torus to round-off); one that reaches it hands over to the fold path
    on the first unclean day; a second run of the same solver goes straight there (hint) with
    identical results; stats of clean and folded days come back through one call."""
    from parasitoids_amd import synthetic
    R, K, nd = 150, 101, 10
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(3.0, 6.0), shift=4)
    ms = np.array([K, K])
    for start, expect_fold in ((150, False), (262, True)):
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        ref = [state]
        trace = {}
        OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, ms, trace=trace)
        s = hip_lib.HipSolve(state, ms, mode='auto', chain_only=True)
        assert s.mode == 'auto'
        s.set_kernels(kernels)
        outs = []
        for rep in range(3):
            if rep:
                s.set_state(state)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            f, fold_fft = s.auto_info()
            if expect_fold:
                assert 0 <= f <= [bool(v) for v in trace['flags']].index(True)
                route = s.auto_route(0, nd)
                assert np.all(route[:f] == 0) and np.all(route[f:] >= 1)
                if np.any(route == 2):
                    assert fold_fft >= N + 3 * (K // 2)
            else:
                assert f == -1 and fold_fft == 0 and not any(trace['flags'])
            for d in range(nd):
                np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=1e-13)
                assert bool(st[d].flag) == bool(trace['flags'][d])
                got = s.chain_solution(d, st[d]).tocsr()
                assert abs(got - ref[d + 1].tocsr()).max() < 1e-12
            outs.append([s.dense(0, d) for d in range(nd)])
        for d in range(nd):      # run 2 and 3 both start from the remembered hint: identical
            assert np.array_equal(outs[1][d], outs[2][d])
        # a split run: continuing behind a hand-over needs a fresh state in this mode (the header says
        # so); behind a clean first part it simply continues
        s.set_state(state)
        s.run_chain(0, nd // 2, renorm=True)
        if expect_fold and s.auto_info()[0] >= 0:
            from parasitoids_amd import _lib as L
            with pytest.raises(L.HipError, match='set the state before every chain run') as ei:
                s.run_chain(nd // 2, nd - nd // 2, renorm=True)
            assert ei.value.code == L.PS_ERR_STATE
        else:
            s.run_chain(nd // 2, nd - nd // 2, renorm=True)
            for d in range(nd):
                np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=1e-13)
        s.close()


def test_flag_speculation_is_exact_in_the_full_column_pipeline(hip_lib, monkeypatch):
    '''The same property as test_flag_speculation_is_exact on a register-resident FFT size in fast
    mode with the full-column pipeline forced: un-flagged days run in chained groups (state column
    parked in LDS), a flag inside a group rolls the chain back to single-day passes with the
    predicated full-column re-FFT -- bit-identical to the non-speculative chain, wherever the
    first flag falls.'''
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 12
    N = 2 * R + 1
    for start in (400, 730, 785):          # no flag at all / first flag late / early
        _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        runs = []
        for spec in (True, False):
            if spec:
                monkeypatch.delenv('PS_NO_SPECULATION', raising=False)
            else:
                monkeypatch.setenv('PS_NO_SPECULATION', '1')
            s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
            assert s.fft_len == 1008
            s.set_kernels(kernels)
            s.prof_enable(True, every=1)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            prof = s.prof_read()
            assert s.full_column
            if spec and start == 400:
                assert prof['col_inv_a_x2'][1] + prof['col_inv_a_x4'][1] + prof['col_inv_a_x8'][1] >= 2   # chained groups ran
            runs.append(([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st]))
            s.close()
        flags = [f for f, _, _, _ in runs[1][1]]
        assert runs[0][1] == runs[1][1]
        for a, b in zip(runs[0][0], runs[1][0]):
            assert np.array_equal(a, b)
        if start == 400:
            assert not any(flags)
        else:
            assert any(flags)
        # and against the oracle (fast torus: the domain part agrees to the fast-mode tolerance)
        ref, trace = [state], {}
        OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, np.array([K, K]), trace=trace)
        for d in range(nd):
            assert np.abs(runs[0][0][d] - trace['raw'][d]).max() < 5e-8
            assert bool(flags[d]) == bool(trace['flags'][d])


def test_deferred_column_retransform_is_bit_identical(hip_lib, monkeypatch):
    """After a flagged day the full-column pipeline transforms only the ROWS of the truncated field;
    the column half runs inside the next day's pass (k_colfull_day ALT), the last day's as a pass of
    its own when the run ends.  Same bits as the separate re-transform (PS_NO_DEFER_REFFT=1), half
    the predicated launches, and a run split in two (the pending column half is resolved between
    them) continues to the same fields."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    monkeypatch.setenv('PS_NO_SPECULATION', '1')
    R, K, nd = 400, 401, 12
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
    state = sparse.coo_matrix(([1.0], ([785], [785])), shape=(N, N))
    runs = {}
    for tag in ('deferred', 'separate', 'split'):
        if tag == 'separate':
            monkeypatch.setenv('PS_NO_DEFER_REFFT', '1')
        else:
            monkeypatch.delenv('PS_NO_DEFER_REFFT', raising=False)
        s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
        assert s.fft_len == 1008
        s.set_kernels(kernels)
        s.prof_enable(True, every=1)
        if tag == 'split':
            s.run_chain(0, 5, renorm=True)
            s.run_chain(5, nd - 5, renorm=True)
        else:
            s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        prof = s.prof_read()
        assert s.full_column
        runs[tag] = ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st],
                     prof['refft_pred'][1])
        s.close()
    assert sum(f for f, _, _, _ in runs['separate'][1]) >= 3          # flags did fire
    assert runs['separate'][2] == 2 * nd and runs['deferred'][2] == nd + 1 and runs['split'][2] == nd + 2
    for tag in ('deferred', 'split'):
        assert runs[tag][1] == runs['separate'][1]
        for a, b in zip(runs[tag][0], runs['separate'][0]):
            assert np.array_equal(a, b)


def test_auto_mode_wide_helper_is_exact(hip_lib, golden, monkeypatch):
    """PS_MODE_AUTO past the clean prefix: flagged and clean days on the wide fast torus
    (N + 2M), dusty days in the fold child, hand-overs in both directions -- forced on for small
    domains here (PS_WIDE_MIN_N=0; by default only domains >= 1500 use the wide helper).  G6 at
    R = 200 (16 of 17 days flagged) and synthetic chains whose first flag comes early / late:
    raw fields, flags and thresholded solutions against the oracle, and the route shows that
    both helpers ran."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_WIDE_MIN_N', '0')
    seen = set()
    # --- G6, R = 200
    g = golden('g6_solutions')
    nd = int(g['r200_ndays'])
    pmfs = [coo_from(g, 'r200_pmf%d' % i) for i in range(nd)]
    ms = g['r200_max_shape']
    N = 401
    first = recentre(pmfs[0], 200)
    trace, modelsol = {}, [first]
    OC.get_solutions(modelsol, pmfs, list(range(nd)), nd, N, ms, trace=trace)
    s = hip_lib.HipSolve(first, ms, mode='auto', chain_only=True)
    for rep in range(3):                      # second run: the hint path starts where the first did; third: route hints
        if rep:
            s.set_state(first)
        s.set_kernels(pmfs[1:])
        s.run_chain(0, nd - 1, renorm=True)
        st = s.chain_stats(0, nd - 1)
        route = s.auto_route(0, nd - 1)
        seen |= set(int(v) for v in route)
        for n in range(nd - 1):
            np.testing.assert_allclose(s.dense(0, n), trace['raw'][n], rtol=0, atol=ATOL)
            assert bool(st[n].flag) == bool(g['r200_flags'][n])
            sol = s.chain_solution(n, st[n])
            assert abs(sol.tocsr() - modelsol[n + 1].tocsr()).max() < 1e-12
    s.close()
    # --- synthetic: mass starting next to the edge
    R, K, nd = 150, 101, 12
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(3.0, 6.0), shift=4)
    # ONE solver for the three states, each twice in a row: what a run remembers of the one before
    # (first unclean day, first helper) belongs to a different state half of the time.  Exact either way.
    s = None
    for start in (230, 262, 285):
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        ref, trace = [state], {}
        OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, np.array([K, K]), trace=trace)
        if s is None:
            s = hip_lib.HipSolve(state, [K, K], mode='auto', chain_only=True)
            s.set_kernels(kernels)
        routes = []
        for rep in range(3):                   # the third run also follows the hints of the second (narrow helper)
            s.set_state(state)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            routes.append(s.auto_route(0, nd))
            seen |= set(int(v) for v in routes[-1])
            for d in range(nd):
                np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=1e-13)
                assert bool(st[d].flag) == bool(trace['flags'][d]), (start, rep, d)
                assert abs(s.chain_solution(d, st[d]).tocsr() - ref[d + 1].tocsr()).max() < 1e-12
    # the per-owner kernel table of a profiled run: every helper that took days of the last run shows
    # launches, and what the helpers add up to is most of the chain (bench.py's helper_kernels)
    s.prof_enable(True)
    s.set_state(state)
    s.run_chain(renorm=True)
    route = s.auto_route(0, nd)
    assert np.array_equal(route, routes[-1])                     # profiling changes no route
    for o, name in enumerate(s.PROF_OWNERS):
        t = s.prof_owner(name)
        assert t == s.prof_owner(o)
        if o and (route == o).any():
            assert s.helper_fft_len(name) > 0
            assert sum(v['launches'] for v in t.values()) > 0 and sum(v['ms'] for v in t.values()) > 0, (name, t)
        if o and not s.helper_fft_len(name):
            assert t == {}
    s.prof_enable(False)
    s.close()
    assert seen >= {0, 1, 2}, seen        # clean prefix, wide helper and fold child all took days


def test_persistent_row_kernel_is_bit_identical(hip_lib, monkeypatch):
    """k_row_inv_rsp (one workgroup per CU walking the row pairs, next pair's rows prefetched
    HBM -> LDS, all days of a chained group in one launch) performs the transform and the epilogue
    of k_row_inv_rs on the same values in the same order: records, flags and statistics must be
    BIT-identical with PS_RSP=0 (one-shot kernel), PS_RSP=1 and PS_NO_ROW_BATCH=1 (one launch per
    day), and with k_row_inv_rs2 (PS_ROW2=1: two roles per workgroup in anti-phase, the default from 4096
    points on) -- un-flagged chained groups (2, 4, 8 days) and a flagged chain (single days, pad rows
    that raise the flag, the Parseval skip of quiet pad-only pairs).  Every column pass of the pipeline
    (single day, chained, two-role chained, the tail columns) writes the intermediate in either layout --
    row pairs interleaved or row-major (PS_NO_PAIR_ROWS=1) -- and every row kernel reads either: the same
    values in other places, bit-identical results."""
    from parasitoids_amd import synthetic
    R, K, nd = 400, 401, 16
    N = 2 * R + 1
    for start in (400, 760):               # no flag (groups of 2, 4, 8, 2 days) / flags from the middle on
        _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        runs = {}
        for tag, env in (('oneshot', {'PS_RSP': '0'}), ('persistent', {'PS_RSP': '1', 'PS_ROW2': '0'}),
                         ('per_day', {'PS_RSP': '1', 'PS_NO_ROW_BATCH': '1', 'PS_ROW2': '0'}),
                         ('two_role', {'PS_RSP': '1', 'PS_ROW2': '1'}),          # k_row_inv_rs2: two roles in anti-phase
                         ('two_role_per_day', {'PS_RSP': '1', 'PS_ROW2': '1', 'PS_NO_ROW_BATCH': '1'}),
                         # the intermediate between the column and the row pass row-major again instead of row pairs
                         # interleaved (the default of the full-column pipeline): each of the three row kernels
                         ('oneshot_row_major', {'PS_RSP': '0', 'PS_NO_PAIR_ROWS': '1'}),
                         ('persistent_row_major', {'PS_RSP': '1', 'PS_ROW2': '0', 'PS_NO_PAIR_ROWS': '1'}),
                         ('two_role_row_major', {'PS_RSP': '1', 'PS_ROW2': '1', 'PS_NO_PAIR_ROWS': '1'})):
            for k in ('PS_RSP', 'PS_NO_ROW_BATCH', 'PS_ROW2', 'PS_NO_PAIR_ROWS'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv('PS_TPIPE', '1')
            s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
            assert s.fft_len == 1008 and s.full_column
            s.set_kernels(kernels)
            s.prof_enable(True, every=1)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            prof = s.prof_read()
            batched = prof['row_inv_x2'][1] + prof['row_inv_x4'][1] + prof['row_inv_x8'][1]
            if tag in ('persistent', 'two_role', 'persistent_row_major', 'two_role_row_major') and start == 400:
                assert batched >= 3            # the chained groups went through one row launch each
            if 'oneshot' in tag or 'per_day' in tag:
                assert batched == 0
            assert s.get_option('PS_NO_PAIR_ROWS') == float('row_major' in tag)
            runs[tag] = ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st])
            s.close()
        assert any(f for f, _, _, _ in runs['oneshot'][1]) == (start != 400)
        for tag in runs:
            assert runs[tag][1] == runs['oneshot'][1], tag
            for a, b in zip(runs[tag][0], runs['oneshot'][0]):
                assert np.array_equal(a, b), tag


def test_window_hint_changes_nothing(hip_lib, monkeypatch):
    """A solver whose previous chain raised no flag opens its next run with windows of up to 32
    days (one chained full-column pass + one row launch each: `col_inv_a_xn`, `row_inv_xn`) instead
    of 2, 4, 8, ...  Same records, flags and statistics bit for bit -- for a clean chain re-run, and
    for a re-run on a state that DOES raise a flag inside the first long window (the days behind
    it are redone one at a time, the hint is gone for the run after)."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    monkeypatch.setenv('PS_RSP', '1')      # the persistent (batched) row kernel also below 1536 points
    R, K, nd = 400, 401, 20
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
    centre = sparse.coo_matrix(([1.0], ([400], [400])), shape=(N, N))
    edge = sparse.coo_matrix(([1.0], ([770], [770])), shape=(N, N))

    def run(s, state):
        s.set_state(state)
        s.prof_enable(True, every=1)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        days = s.prof_days()
        return ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st],
                days['col_inv_a_xn'], days['row_inv_xn'])

    s = hip_lib.HipSolve(centre, [K, K], mode='fast', chain_only=True)
    assert s.fft_len == 1008 and s.full_column
    s.set_kernels(kernels)
    a = run(s, centre)                     # windows 2, 4, 8, 6: nothing hinted
    assert not any(f for f, _, _, _ in a[1])
    b = run(s, centre)                     # hinted: one 20-day window
    assert b[2] == nd and b[3] == nd, b[2:]
    assert a[1] == b[1]
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y)
    c = run(s, edge)                       # hinted again, but this state flags early
    assert any(f for f, _, _, _ in c[1])
    s.close()
    monkeypatch.setenv('PS_NO_WINDOW_HINT', '1')
    monkeypatch.setenv('PS_NO_SPECULATION', '1')
    s2 = hip_lib.HipSolve(edge, [K, K], mode='fast', chain_only=True)
    s2.set_kernels(kernels)
    d = run(s2, edge)
    s2.close()
    assert c[1] == d[1]
    for x, y in zip(c[0], d[0]):
        assert np.array_equal(x, y)


def test_fold_mode_fused_day_is_bit_identical(hip_lib, monkeypatch):
    """PS_MODE_FOLD on the full-column pipeline: the forward column transform of the state runs inside the
    day pass (k_colfull_day ALT on the row pass of the torus) and a flagged day's truncation inside the next
    row pass (k_row_fwd_rs reads the flag; the torus itself is truncated when the run ends).  Same bits as
    the separate launches (PS_NO_FOLD_FUSE=1), also for a run split in two, with flags on both sides of
    the split.  The inverse row pass that folds on chip (k_row_inv_fold, the default) adds rows first and
    columns second where k_fold adds element by element: equal to round-off, same flags and counts."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 10
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
    state = sparse.coo_matrix(([1.0], ([730], [730])), shape=(N, N))
    runs = {}
    for tag in ('fused', 'separate', 'split', 'no_alt', 'rows', 'rows_split'):
        monkeypatch.delenv('PS_NO_FOLD_FUSE', raising=False)
        monkeypatch.delenv('PS_NO_FOLD_ROWS', raising=False)
        monkeypatch.delenv('PS_NO_FOLD_ALT', raising=False)
        if tag == 'no_alt':        # sizes without an ALT day pass (9-wave transforms): forward column pass kept
            monkeypatch.setenv('PS_NO_FOLD_ALT', '1')
            monkeypatch.setenv('PS_NO_FOLD_ROWS', '1')
        if tag == 'separate':
            monkeypatch.setenv('PS_NO_FOLD_FUSE', '1')
        if tag in ('fused', 'split'):
            monkeypatch.setenv('PS_NO_FOLD_ROWS', '1')
        s = hip_lib.HipSolve(state, [K, K], mode='fold', chain_only=True)
        assert s.mode == 'fold' and s.full_column
        s.set_kernels(kernels)
        s.prof_enable(True, every=1)
        if tag.endswith('split'):
            s.run_chain(0, 4, renorm=True)
            s.run_chain(4, nd - 4, renorm=True)
        else:
            s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        prof = s.prof_read()
        runs[tag] = ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st],
                     prof['col_fwd_a'][1])
        s.close()
    flags = [f for f, _, _, _ in runs['separate'][1]]
    assert sum(flags[:4]) >= 1 and sum(flags[4:]) >= 1 and not all(flags), flags
    assert runs['separate'][2] == nd and all(runs[t][2] == 0 for t in ('fused', 'split', 'rows', 'rows_split'))
    assert runs['no_alt'][2] == nd
    for tag in ('fused', 'split', 'no_alt'):
        assert runs[tag][1] == runs['separate'][1]
        for a, b in zip(runs[tag][0], runs['separate'][0]):
            assert np.array_equal(a, b)
    for tag in ('rows', 'rows_split'):
        for (f, n, sm, dl), (f0, n0, sm0, dl0) in zip(runs[tag][1], runs['separate'][1]):
            assert (f, n) == (f0, n0) and abs(sm - sm0) < 1e-13 and abs(dl - dl0) < 1e-16
        for a, b in zip(runs[tag][0], runs['separate'][0]):
            assert np.abs(a - b).max() < 1e-16
    assert runs['rows'][1] == runs['rows_split'][1]
    for a, b in zip(runs['rows'][0], runs['rows_split'][0]):
        assert np.array_equal(a, b)


def test_flag_history_chains_the_quiet_stretches(hip_lib, monkeypatch):
    """A solver that has seen a flag stops speculating blindly; from its second run over the same days on it
    chains the stretches that raised no flag last time (verified like any speculation window) and gives
    every other day its predicated re-transform.  Same bits as one safe day at a time
    (PS_NO_FLAG_HISTORY=1) -- when the flags repeat, when the next state flags EARLIER than the history
    says (a flag inside a chained stretch: the rest of the run is redone the safe way), and when it flags
    later or not at all."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 16
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
    pt = lambda c: sparse.coo_matrix(([1.0], ([c], [c])), shape=(N, N))
    late, early, never = pt(700), pt(730), pt(400)

    def run(s, state):
        s.set_state(state)
        s.prof_enable(True, every=1)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        days = s.prof_days()
        chained = sum(days[k] for k in ('col_inv_a_x2', 'col_inv_a_x4', 'col_inv_a_x8', 'col_inv_a_xn'))
        return ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st], chained)

    order = (late, late, late, early, early, never, never, late)
    out = {}
    for tag in ('history', 'plain'):
        if tag == 'plain':
            monkeypatch.setenv('PS_NO_FLAG_HISTORY', '1')
        s = hip_lib.HipSolve(late, [K, K], mode='fast', chain_only=True)
        assert s.fft_len == 1008 and s.full_column
        s.set_kernels(kernels)
        out[tag] = [run(s, st) for st in order]
        s.close()
    flags = [[f for f, _, _, _ in r[1]] for r in out['plain']]
    first = [f.index(1) if 1 in f else nd for f in flags]
    assert first[3] < first[0] < nd and first[0] >= 3 and first[5] == nd, first      # early < late < never
    for i, (h, p) in enumerate(zip(out['history'], out['plain'])):
        assert h[1] == p[1], i
        for a, b in zip(h[0], p[0]):
            assert np.array_equal(a, b), i
    # run 0 speculates until its first flag; the plain solver never chains again, the other one chains the
    # leading quiet stretch of runs 1, 2 (flags repeat), 3 (flags earlier: stretch redone), 5, 6, 7
    assert all(r[2] == 0 for r in out['plain'][1:])
    chained = [r[2] for r in out['history']]
    assert chained[1] >= first[0] - 1 and chained[2] == chained[1], chained
    assert chained[6] == nd, chained                                     # history "no flag at all": one window
    assert chained[4] >= first[3] - 1, chained


def test_second_stream_kernel_transforms_change_nothing(hip_lib, monkeypatch):
    """Long chunks in the full-column pipeline transform only the kernels of the first windows
    ahead of the day passes; the rest run on a second stream behind them (ps_chain_run, PS_KT_SPLIT).
    Same records, flags and statistics, bit for bit, as with every transform up front -- for a
    clean 26-day chain, for one that raises its first flag before the second part is needed (the
    run rolls back to single days while the second stream is still working) and for one that raises
    it after; and twice in a row on the same solver (the second stream's slots are reused)."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    R, K, nd = 400, 401, 26
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=11, sigma=(6.0, 12.0), shift=6)
    seen_flags = []
    for start in (400, 600, 650, 775):
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        runs = {}
        for tag, split in (('upfront', '0'), ('split', '8')):
            monkeypatch.setenv('PS_KT_SPLIT', split)
            s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
            s.set_kernels(kernels)
            out = []
            for rep in range(2):
                s.set_state(state)
                s.run_chain(renorm=True)
                st = s.chain_stats(0, nd)
                out.append(([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st]))
            assert s.full_column
            s.close()
            runs[tag] = out
        for rep in range(2):
            assert runs['split'][rep][1] == runs['upfront'][rep][1]
            for a, b in zip(runs['split'][rep][0], runs['upfront'][rep][0]):
                assert np.array_equal(a, b)
        flags = [f for f, _, _, _ in runs['upfront'][0][1]]
        seen_flags.append(flags.index(1) if 1 in flags else -1)
    assert seen_flags[0] == -1 and any(f >= 8 for f in seen_flags) and any(0 <= f < 8 for f in seen_flags), seen_flags


def test_options_are_frozen_per_handle(hip_lib, monkeypatch):
    """The library reads its PS_* environment once, in ps_solver_create; after that the handle's own
    table is all it looks at and ps_solver_set_option changes it (the reference has
    one switch, globalvars.py:5).  An environment change behind a live solver does nothing; the same
    knob set through the call switches the route between two runs of ONE solver, bit-identically;
    creation-time knobs and unknown names are refused."""
    from parasitoids_amd import synthetic
    from parasitoids_amd._lib import HipError, PS_ERR_STATE, PS_ERR_BAD_ARG
    monkeypatch.setenv('PS_TPIPE', '1')
    monkeypatch.setenv('PS_RSP', '1')
    R, K, nd = 400, 401, 16
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7, sigma=(6.0, 12.0), shift=10)
    state = sparse.coo_matrix(([1.0], ([400], [400])), shape=(N, N))
    s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
    assert s.get_option('PS_RSP') == 1 and s.get_option('PS_TPIPE') == 1
    s.set_kernels(kernels)

    def run():
        s.set_state(state)
        s.prof_enable(True, every=1)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        prof = s.prof_read()
        batched = sum(prof[k][1] for k in ('row_inv_x2', 'row_inv_x4', 'row_inv_x8', 'row_inv_xn'))
        return [s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta) for x in st], batched

    s.set_option('PS_NO_WINDOW_HINT', 1)       # every run ramps 2, 4, 8, 2: comparable launch counts
    a = run()
    assert a[2] >= 3
    monkeypatch.setenv('PS_RSP', '0')          # behind the solver's back: no effect
    monkeypatch.setenv('PS_NO_ROW_BATCH', '1')
    b = run()
    assert b[2] == a[2]
    s.set_option('PS_NO_ROW_BATCH', 1)         # through the call: one row launch per day
    c = run()
    assert c[2] == 0
    s.set_option('PS_RSP', 0)                  # ... and the one-shot row kernel
    d = run()
    assert d[2] == 0 and s.get_option('PS_RSP') == 0
    for r in (b, c, d):
        assert r[1] == a[1]
        for x, y in zip(r[0], a[0]):
            assert np.array_equal(x, y)
    for key, code in (('PS_NO_RS', PS_ERR_STATE), ('PS_COL_L1', PS_ERR_STATE), ('PS_NO_SUCH_KNOB', PS_ERR_BAD_ARG)):
        with pytest.raises(HipError) as e:
            s.set_option(key, 1)
        assert e.value.code == code
    s.close()
    # a solver created now starts from the environment as it is now
    s2 = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
    assert s2.get_option('PS_RSP') == 0 and s2.get_option('PS_NO_ROW_BATCH') == 1
    s2.close()


def test_lazy_kernel_transforms_keep_one_format_per_chunk(hip_lib, monkeypatch):
    """an auto-mode front whose previous run handed over at day h transforms only
    the kernels up to day h + 1 up front and the rest when the run gets there (`lazy_tail`).  On the
    TILED pipeline the two halves of such a chunk must share one format -- `kt_direct` / `kt_live` are
    solver-wide -- even when the early days are compact (direct-sum first column sub-pass) and a late
    day is broad.  A size without register-resident kernels (PS_NO_RS), compact kernels on days 0-7 and
    broad ones behind: first run from the edge (hands over early), second from the centre (stays clean
    past the lazily transformed days).  Bit-identical to PS_NO_LAZY_KT=1, and right against the oracle."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_NO_RS', '1')
    R, K, nd = 700, 401, 12
    N = 2 * R + 1
    _, narrow, _ = synthetic.make_stack(R=R, K=K, ndays=8, seed=5, sigma=(2.5, 3.5), shift=3)
    _, broad, _ = synthetic.make_stack(R=R, K=K, ndays=nd - 8, seed=6, sigma=(45.0, 60.0), shift=3)
    kernels = narrow + broad
    pt = lambda c: sparse.coo_matrix(([1.0], ([c], [c])), shape=(N, N))
    edge, centre = pt(N - 1 - 40), pt(R)
    ms = np.array([K, K])
    ref, trace = [centre], {}
    OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, ms, trace=trace)
    out = {}
    for tag in ('lazy', 'upfront'):
        s = hip_lib.HipSolve(edge, ms, mode='auto', chain_only=True)
        assert s.mode == 'auto' and not s.full_column
        s.set_option('PS_DIRECT_MAX_TERMS', 4)     # narrow: 2 terms per output, broad: >= 8 (any column split)
        if tag == 'upfront':
            s.set_option('PS_NO_LAZY_KT', 1)
        s.set_kernels(kernels)
        s.run_chain(renorm=True)
        h = s.auto_info()[0]
        assert 1 <= h <= 5, h                       # handed over while the kernels were still the narrow ones
        s.set_state(centre)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        h2 = s.auto_info()[0]
        assert h2 == -1 or h2 >= 9, h2              # the front consumed lazily transformed (broad) days
        out[tag] = [s.dense(0, d) for d in range(nd)]
        for d in range(nd):
            np.testing.assert_allclose(out[tag][d], trace['raw'][d], rtol=0, atol=1e-13)
            assert abs(s.chain_solution(d, st[d]).tocsr() - ref[d + 1].tocsr()).max() < 1e-12
        s.close()
    for a, b in zip(out['lazy'], out['upfront']):
        assert np.array_equal(a, b)
