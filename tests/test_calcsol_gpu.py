"""GPU tests of the CalcSol / Run mirrors: the reference's own tests/test_CalcSol.py
(test_fftconv2, test_convolve_same, test_back_solve) re-run against the device functions,
and end-to-end runs (wind file -> device prob_mass -> device chain) against G6/G7/G8."""
import os

import numpy as np
import pytest
from scipy import sparse, signal, fft as sfft

from helpers import coo_from, assert_summary, recentre

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def CS():
    from parasitoids_amd import CalcSol
    return CalcSol


@pytest.fixture(scope='module')
def two_arrays():
    A = np.outer(range(10), range(1, 11))
    B = np.outer(range(4, -1, -1), range(8, -1, -2))
    return (A, B)


def test_fftconv2(CS, two_arrays):
    '''reference tests/test_CalcSol.py:75-83'''
    A, B = two_arrays
    A_hat = CS.fft2(sparse.coo_matrix(A), np.array(B.shape))
    before = A_hat.copy()
    CS.fftconv2(A_hat, sparse.csr_matrix(B))
    assert not np.all(A_hat == before)
    assert np.all(B == np.outer(range(4, -1, -1), range(8, -1, -2)))
    assert np.allclose(sfft.ifft2(A_hat)[:A.shape[0], :A.shape[1]].real,
                       signal.convolve2d(A, B, 'same'))


def test_convolve_same(CS, two_arrays):
    '''reference tests/test_CalcSol.py:85-98'''
    A, B = two_arrays
    fft_shape = np.array([A.shape[0] + 6, A.shape[1] + 6])
    A_hat = CS.fft2(sparse.coo_matrix(A), fft_shape)
    CS.fftconv2(A_hat, sparse.csr_matrix(B))
    C, flag = CS.ifft2(A_hat, A.shape)
    C = C.toarray()
    assert not np.iscomplexobj(C)
    assert np.allclose(C, signal.fftconvolve(A, B, 'same'))


def test_back_solve(CS, golden):
    '''reference tests/test_CalcSol.py:115-139'''
    g = golden('g8_back_solve')
    A, B, C, D = (g['toy_' + k] for k in 'ABCD')
    C_hat = CS.fft2(sparse.coo_matrix(C), A.shape)
    CS.fftconv2(C_hat, sparse.csr_matrix(D))
    bckCD = CS.back_solve([sparse.csr_matrix(A), sparse.csr_matrix(B)], C_hat, A.shape)
    B_hat = CS.fft2(sparse.coo_matrix(B), A.shape)
    CS.fftconv2(B_hat, sparse.csr_matrix(C))
    CS.fftconv2(B_hat, sparse.csr_matrix(D))
    BCD, flag = CS.ifft2(B_hat, B.shape)
    A_hat = CS.fft2(sparse.coo_matrix(A), A.shape)
    for X in (B, C, D):
        CS.fftconv2(A_hat, sparse.csr_matrix(X))
    ABCD, flag = CS.ifft2(A_hat, A.shape)
    assert np.allclose(bckCD[1].toarray(), BCD.toarray())
    assert np.allclose(bckCD[0].toarray(), ABCD.toarray())
    np.testing.assert_allclose(bckCD[0].toarray(), g['toy_bck0'], rtol=0, atol=1e-11)
    np.testing.assert_allclose(bckCD[1].toarray(), g['toy_bck1'], rtol=0, atol=1e-11)


def test_r_small_vals(CS):
    rng = np.random.default_rng(3)
    A = rng.random((40, 40)) * (rng.random((40, 40)) < 0.3)
    A[A < 0.05] *= 1e-9
    A /= A.sum()
    got = CS.r_small_vals(sparse.coo_matrix(A), prob_model=True)
    keep = A >= 1e-8
    ref = np.where(keep, A, 0.0)
    ref[keep] += (1 - ref.sum()) / keep.sum()
    np.testing.assert_allclose(got.toarray(), ref, rtol=0, atol=1e-16)
    assert got.nnz == keep.sum()
    got2 = CS.r_small_vals(sparse.coo_matrix(A))
    np.testing.assert_allclose(got2.toarray(), np.where(keep, A, 0.0), rtol=0, atol=0)


def _params(golden_dir, *args):
    from parasitoids_amd import Run
    p = Run.Params(config=None)
    p.cmd_line_chg(list(args))
    p.site_name = os.path.join(golden_dir, p.site_name)
    p.OUTPUT = False
    return Run, p


def test_run_config1_prob_model(golden, golden_dir):
    '''C1 / G6: Run.py --prob --kalbar ndays=6 domain_info=(10000.0,128), end to end on the
    device, against the reference's CPU solutions.'''
    Run, p = _params(golden_dir, '--kalbar', '--prob', 'ndays=6', 'domain_info=(10000.0,128)')
    modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    g = golden('g6_solutions')
    assert ndays == 6 and days[:6] == [13, 14, 15, 16, 17, 18]
    pos = g['r128_pos']
    for i, s in enumerate(modelsol):
        ref = coo_from(g, 'r128_sol%d' % i).tocsr()
        assert abs(s.tocsr() - ref).max() < 1e-12
        assert abs(s.sum() - 1.0) < 1e-12
        assert_summary(g, 'r128_sum%d' % i, s, pos, rtol=1e-10, atol=1e-12, nnz_slack=3)


def test_run_pop_model_kalbar(golden, golden_dir):
    '''G7: --pop --kalbar (r_dur=1), R=128, 6 days'''
    Run, p = _params(golden_dir, '--kalbar', '--pop', 'ndays=6', 'domain_info=(10000.0,128)')
    modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    g = golden('g7_populations')
    pos = g['r128_pos']
    for i, s in enumerate(modelsol):
        ref = coo_from(g, 'r128_pop%d' % i).tocsr()
        assert abs(s.tocsr() - ref).max() < 1e-7          # values up to 1.3e5 (rel 1e-12)
        assert_summary(g, 'r128_sum%d' % i, s, pos, rtol=1e-10, atol=1e-7, nnz_slack=3)


def test_run_pop_model_carnarvon_rdur5(golden, golden_dir):
    '''G8: --pop --carnarvon r_dur=5 at domain_info=(40000.0,200), 30 days: multi-day
    release with back_solve and the release-day weighted sums'''
    Run, p = _params(golden_dir, '--carnarvon', '--pop', 'domain_info=(40000.0,200)')
    assert p.r_dur == 5 and p.r_number == 40000 and p.r_start == 0.354
    modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    g = golden('g8_back_solve')
    assert ndays == 30
    pos = g['car_pos']
    for i, s in enumerate(modelsol):
        assert_summary(g, 'car_sum%d' % i, s, pos, rtol=1e-9, atol=1e-7, nnz_slack=4)
    for i in (0, 4, 5, 29):
        ref = coo_from(g, 'car_pop%d' % i).tocsr()
        assert abs(modelsol[i].tocsr() - ref).max() < 1e-7


def test_save_result_format(golden_dir, tmp_path):
    '''on-disk format of Run.py:490-516 (read back like Plot_Result.py:511-524)'''
    Run, p = _params(golden_dir, '--kalbar', '--prob', 'ndays=2', 'domain_info=(10000.0,64)')
    p.outfile = str(tmp_path / 'out' / 'kalbar_test')
    modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    Run.save_result(p, modelsol, days, ndays)
    z = np.load(p.outfile + '.npz')
    assert list(z['days']) == days[:2]
    dom = 129
    for n, day in enumerate(days[:2]):
        m = sparse.csr_matrix((z[str(day) + '_data'], z[str(day) + '_ind'],
                               z[str(day) + '_indptr']), shape=(dom, dom))
        assert abs(m - modelsol[n].tocsr()).max() == 0
    q = Run.Params(config=None)
    q.file_read_chg(p.outfile)
    assert tuple(q.domain_info) == (10000.0, 64) and q.ndays == 2


def test_fetch_csr_equals_coo_tocsr(golden):
    '''ps_record_fetch_csr: the CSR triplets from the device compaction are exactly what
    scipy builds from the COO result (the format of the reference's result files,
    Run.py:490-516), including empty rows and an all-zero record.'''
    from parasitoids_amd import hip_lib, _lib as L
    g = golden('g6_solutions')
    nd = 4
    pmfs = [coo_from(g, 'r128_pmf%d' % i) for i in range(nd)]
    s = hip_lib.HipSolve(recentre(pmfs[0], 128), g['r128_max_shape'])
    s.set_kernels(pmfs[1:])
    s.run_chain(renorm=True)
    st = s.chain_stats(0, nd - 1)
    for d in range(nd - 1):
        a = s.chain_solution(d, st[d]).tocsr()
        b = s.chain_solution(d, st[d], fmt='csr')
        assert b.format == 'csr' and b.has_sorted_indices
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
        assert np.array_equal(a.data, b.data)
    # threshold above every value: empty matrix, indptr all zero
    e = s._fetch(L.REC_CHAIN, 0, 10.0, 1.0, 0.0, 1.0, 0, 'csr')
    assert e.nnz == 0 and e.shape == (257, 257) and not e.indptr.any()
    s.close()


@pytest.mark.parametrize('rad_dist', [10000.0, 40000.0])
def test_baseline_config2_population_model_1024(golden_dir, rad_dist):
    """BASELINE.json configs[1]: Run.py --pop, 1024 x 1024 grid (R = 512, N = 1025), 30 days,
    fp64 -- the device chain (default 'auto' mode: the reference's own torus) against the
    oracle's get_populations on the SAME day kernels (device prob_mass, whose parity at this
    grid has its own test against the reference: g5b).  At the default rad_dist = 10 km (SURVEY 8d
    C2: `--carnarvon --pop r_dur=1`) the boundary flag fires and mass leaves the domain -- day 30
    holds about 27 211 of the 40 000 wasps; 40 km is the flag-free variant.  Values reach
    r_number = 40 000, so 1e-7 absolute is 2.5e-12 relative."""
    from oracle import calcsol as OC
    from parasitoids_amd import ParasitoidModel as PM
    Run, p = _params(golden_dir, '--carnarvon', '--pop', 'r_dur=1', 'domain_info=(%r,512)' % rad_dist)
    modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    assert ndays == 30 and modelsol[0].shape == (1025, 1025)
    wind_data, days2 = PM.get_wind_data(*p.get_wind_params())
    starts = [p.r_start] + [None] * (ndays - 1)
    pmf_list = PM.prob_mass_batch(days2[:ndays], wind_data, *p.get_model_params(), start_times=starts)
    max_shape = np.array([0, 0])
    for pmf in pmf_list:
        max_shape = np.maximum(max_shape, pmf.shape)
    r_spread = [recentre(pmf_list[0], 512).tocsr()]
    ref = OC.get_populations(r_spread, pmf_list, days2, ndays, 1025, max_shape, 1, p.r_number,
                             p.r_mthd())
    assert len(ref) == len(modelsol) == 30
    for i, (a, b) in enumerate(zip(modelsol, ref)):
        d = abs(a.tocsr() - b.tocsr())
        assert (d.max() if d.nnz else 0.0) < 1e-7, i
        assert abs(a.sum() - b.sum()) < 1e-6 * max(1.0, abs(b.sum()))
    if rad_dist == 10000.0:
        assert 27000.0 < modelsol[-1].sum() < 27400.0, modelsol[-1].sum()    # mass has left (flags fired)
    else:
        assert modelsol[-1].sum() > 0.999 * p.r_number


@pytest.mark.parametrize('mode', ['fast', 'exact', 'auto'])
def test_multi_day_release_on_the_chain_api(golden_dir, mode):
    """get_populations with r_dur = 5 (the reference's default Carnarvon preset, Run.py:118) through
    ps_chain_run_release -- day kernels and release-day filters uploaded once, every day after the release
    enqueued without a host round trip (CalcSol.py:296-323, cuda_lib.py:145-221) -- against the oracle's
    get_populations on the same kernels, at a domain where the flag FIRES in the cohort and in the
    back-solves (10 km, R = 128).  'fast': the chain route, fast-torus tolerance.  'exact': the chain
    route on the reference's own torus (tiled pipeline), 1e-7.  'auto': the fast torus
    cannot certify this run (dust outside the domain), the exact per-call route takes over: 1e-7 on
    populations up to 8000 per cohort (1e-11 relative).  The flag-free 40 km run of G8
    (test_run_pop_model_carnarvon_rdur5) is the certified 'auto' chain route against the reference."""
    from oracle import calcsol as OC
    from parasitoids_amd import CalcSol, globalvars
    from parasitoids_amd import ParasitoidModel as PM
    Run, p = _params(golden_dir, '--carnarvon', '--pop', 'ndays=12', 'domain_info=(10000.0,128)')
    assert p.r_dur == 5
    old = globalvars.fft_mode
    globalvars.fft_mode = mode
    try:
        modelsol, days, ndays, _ = Run.run_model(p, verbose=False)
    finally:
        globalvars.fft_mode = old
    assert ndays == 12 and len(modelsol) == 12
    assert CalcSol.last_release_route == ('per-call' if mode == 'auto' else 'chain')
    wind_data, days2 = PM.get_wind_data(*p.get_wind_params())
    starts = [p.r_start] + [None] * (ndays - 1)
    pmf_list = PM.prob_mass_batch(days2[:ndays], wind_data, *p.get_model_params(), start_times=starts)
    max_shape = np.array([0, 0])
    for pmf in pmf_list:
        max_shape = np.maximum(max_shape, pmf.shape)
    r_spread = [recentre(pmf_list[d], 128).tocsr() for d in range(p.r_dur)]
    trace = {}
    ref = OC.get_populations(r_spread, pmf_list, days2, ndays, 257, max_shape, p.r_dur, p.r_number,
                             p.r_mthd(), trace=trace)
    tol = 5e-8 * p.r_number if mode == 'fast' else 1e-7       # fast torus: <= 1e-8 per un-flagged field
    for i, (a, b) in enumerate(zip(modelsol, ref)):
        d = abs(a.tocsr() - b.tocsr())
        assert (d.max() if d.nnz else 0.0) < tol, (mode, i)
    assert modelsol[-1].sum() < 0.999 * p.r_number            # mass has left: flags did fire


def test_multi_day_release_certified_chain_route(golden, golden_dir):
    """The G8 run (Carnarvon r_dur = 5, 40 km: nothing reaches the pad) takes the chain route in the
    default 'auto' mode and is certified exact; forced through the per-call route it gives the same
    populations to round-off."""
    from parasitoids_amd import CalcSol
    Run, p = _params(golden_dir, '--carnarvon', '--pop', 'ndays=10', 'domain_info=(40000.0,200)')
    a, _, _, _ = Run.run_model(p, verbose=False)
    assert CalcSol.last_release_route == 'chain'
    orig = CalcSol._release_on_chain
    CalcSol._release_on_chain = lambda *args: None
    try:
        b, _, _, _ = Run.run_model(p, verbose=False)
    finally:
        CalcSol._release_on_chain = orig
    assert CalcSol.last_release_route == 'per-call'
    g = golden('g8_back_solve')
    for i, (x, y) in enumerate(zip(a, b)):
        d = abs(x.tocsr() - y.tocsr())
        assert (d.max() if d.nnz else 0.0) < 1e-8, i
        assert_summary(g, 'car_sum%d' % i, x, g['car_pos'], rtol=1e-9, atol=1e-7, nnz_slack=4)


def test_multi_day_release_runs_chunk_by_chunk(golden_dir, monkeypatch):
    """ps_chain_run_release keeps the spectra of a chunk of day kernels plus the filters at a time; a run
    longer than a chunk (here PS_CHUNK_DAYS = 6: two days + four filters per chunk) gives the same
    populations bit for bit as the run whose 7 days fit one chunk."""
    from parasitoids_amd import CalcSol, globalvars
    Run, p = _params(golden_dir, '--carnarvon', '--pop', 'ndays=12', 'domain_info=(10000.0,128)')
    old = globalvars.fft_mode
    globalvars.fft_mode = 'fast'
    try:
        a, _, _, _ = Run.run_model(p, verbose=False)
        monkeypatch.setenv('PS_CHUNK_DAYS', '6')
        b, _, _, _ = Run.run_model(p, verbose=False)
    finally:
        globalvars.fft_mode = old
    assert CalcSol.last_release_route == 'chain' and len(a) == len(b) == 12
    for x, y in zip(a, b):
        assert (x != y).nnz == 0
