"""GPU parity tests of the per-day kernel construction (prob_mass on the device,
through the C ABI) against the reference's golden vectors G2/G3/G5 and the oracle.
Mirrors the reference's tests/test_ParsitoidModel.py (properties) and pins values."""
import math
import os
import warnings

import numpy as np
import pytest
from scipy import sparse

from oracle import model as OM
from helpers import HP, DP, DLP, MU_R, NPER, HP_T, DP_T, coo_from, check_digest

pytestmark = pytest.mark.gpu

# cell masses are differences of O(1) cdf values: absolute error ~1e-16 per BVU
VAL_ATOL = 5e-15


@pytest.fixture(scope='module')
def PM():
    from parasitoids_amd import ParasitoidModel
    return ParasitoidModel


@pytest.fixture(scope='module')
def kalbar(PM, golden_dir):
    return PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '00:00')


@pytest.fixture(scope='module')
def carnarvon(PM, golden_dir):
    return PM.get_wind_data(os.path.join(golden_dir, 'data', 'carnarvonearl'), 30, '00:30')


def test_h_flight_prob(PM, golden, kalbar, carnarvon):
    '''G3 + reference test_h_flight_prob properties (tests/test_ParsitoidModel.py:213-245)'''
    g = golden('g3_hprob_wind')
    for site, (wd, days) in (('kalbar', kalbar), ('carnarvonearl', carnarvon)):
        model = PM.WindModel(wd)
        for i, d in enumerate(days[:8]):
            h = model.h_flight_prob(d, *HP)
            np.testing.assert_allclose(h, g[site + '_h_def'][i], rtol=1e-13, atol=1e-18)
        for i, d in enumerate(days[:4]):
            h = model.h_flight_prob(d, *HP_T)
            np.testing.assert_allclose(h, g[site + '_h_test'][i], rtol=1e-13, atol=1e-18)
            f = PM.f_time_prob(h.size, *HP_T[3:])
            gw = PM.g_wind_prob(wd[d][:, 2], *HP_T[1:3])
            assert np.all(h >= 0) and h.sum() <= 1 and np.all(h >= f * gw * (1 - 1e-12))
        model.close()
    one = PM.h_flight_prob(kalbar[0][13], *HP)
    np.testing.assert_allclose(one, g['kalbar_h_def'][0], rtol=1e-13, atol=1e-18)


def test_get_mvn_cdf_values(PM, golden):
    '''G2 stamps + the properties of the reference test (tests/test_ParsitoidModel.py:247-296)'''
    g = golden('g2_stamps')
    for k in range(int(g['n'])):
        c = g['c%d' % k]
        ref = g['m%d' % k]
        mat = PM.get_mvn_cdf_values(c[0], np.array(c[1:3]), PM.Dmat(*c[3:6]))
        assert mat.shape == ref.shape
        assert np.abs(mat - ref).max() < VAL_ATOL
    S1 = np.array([[16., 8.], [8., 16.]])
    S2 = np.array([[100., -50.], [-50., 100.]])
    m1 = PM.get_mvn_cdf_values(2, np.zeros(2), S1)
    m2 = PM.get_mvn_cdf_values(2, np.zeros(2), S2)
    assert 0.99 < m1.sum() < 1 and 0.99 < m2.sum() < 1 and m2.size > m1.size
    cen = m1.shape[0] // 2
    assert m1[0:cen, 0:cen].sum() < m1[0:cen, cen + 1:].sum()
    assert m1.max() == m1[cen, cen]


def _check(g, name, r, tol=VAL_ATOL):
    r = r.tocoo()
    assert tuple(g[name + '_shape']) == r.shape
    assert r.nnz == len(g[name + '_val'])
    assert np.array_equal(g[name + '_row'], r.row)
    assert np.array_equal(g[name + '_col'], r.col)
    assert np.abs(r.data - g[name + '_val']).max() < tol
    assert math.isclose(r.data.sum(), 1.0, rel_tol=1e-12)


def test_prob_mass_kalbar_r128_batch(PM, golden, kalbar):
    '''G5: Kalbar days 13-18 at R=128 in ONE device batch vs the reference's outputs'''
    g = golden('g5_prob_mass')
    wd, days = kalbar
    wd_copy = {k: v.copy() for k, v in wd.items()}
    res = PM.prob_mass_batch(days[:6], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 128)
    for d, r in zip(days[:6], res):
        _check(g, 'kal128_d%d' % d, r)
    for k in wd:
        assert np.array_equal(wd[k], wd_copy[k])          # inputs unmodified
    # support half widths and losses agree with the oracle's per-period trace
    model = PM._model_for(wd)
    _, dbg = OM.prob_mass(days[0], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 128, return_debug=True)
    mine = model.debug(0)
    assert list(mine['H']) == dbg['H']
    np.testing.assert_allclose(mine['hprob'], dbg['hprob'], rtol=1e-13, atol=1e-18)
    assert abs(mine['loss'] - dbg['loss']) < 1e-15 and abs(mine['pmfsum'] - dbg['pmfsum']) < 1e-13


def test_prob_mass_single_call_signature(PM, golden, kalbar, carnarvon):
    '''reference one-day signature, incl. start_time and r_start'''
    g = golden('g5_prob_mass')
    wd, days = kalbar
    _check(g, 'kal128_d14', PM.prob_mass(14, wd, HP, DP, DLP, MU_R, NPER, 10000.0, 128))
    wc, dc = carnarvon
    _check(g, 'car128_start', PM.prob_mass(dc[0], wc, HP, DP, DLP, MU_R, NPER, 10000.0, 128, 0.354))


def test_prob_mass_r400(PM, golden, kalbar):
    g = golden('g5_prob_mass')
    wd, days = kalbar
    res = PM.prob_mass_batch(days[:2], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 400)
    for d, r in zip(days[:2], res):
        _check(g, 'kal400_d%d' % d, r)


def test_prob_mass_leaves_domain(PM, golden, kalbar):
    '''strong advection on a small domain: clipped windows, lost periods, one warning'''
    g = golden('g5_prob_mass')
    wd, days = kalbar
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        r = PM.prob_mass(days[1], wd, HP, DP, DLP, 6.0, NPER, 2000.0, 64)
        assert sum(issubclass(x.category, RuntimeWarning) for x in w) == 1
    _check(g, 'kal64_clip', r, tol=1e-13)


def test_prob_mass_reference_test_cases(PM, golden, carnarvon):
    '''the parameter sets of tests/test_ParsitoidModel.py:300-407'''
    g = golden('g5_prob_mass')
    wc, dc = carnarvon
    full = PM.prob_mass(1, wc, HP_T, DP_T, DP_T, 1, 6, 8000.0, 320)
    _check(g, 'test320_full', full)
    noon = PM.prob_mass(1, wc, HP_T, DP_T, DP_T, 1, 6, 8000.0, 320, 0.5)
    _check(g, 'test320_noon', noon)
    # noon release leaves more at the origin than the full day (reference :407)
    offset = 320 - full.shape[0] // 2
    first = sparse.coo_matrix((full.data, (full.row + offset, full.col + offset)),
                              shape=(641, 641)).tocsr()
    mid2 = noon.shape[0] // 2
    assert noon.tocsr()[mid2, mid2] > first[320, 320]
    # single-period TEST_RUN mode (:315-325): lands in the wind's quadrant
    sing = {1: g['test320_single_wind']}
    cpy = {1: sing[1].copy()}
    r = PM.prob_mass(1, sing, (1.0, 1.8, 6, -4., 2., 19., 2.), DP_T, DP_T, 0.1 / 24, 1, 8000.0, 320)
    _check(g, 'test320_single', r)
    assert np.array_equal(sing[1], cpy[1])


def test_prob_mass_bad_parameters(PM, kalbar):
    wd, days = kalbar
    with pytest.raises(Exception):
        PM.prob_mass(days[0], wd, HP, (-1.0, 1.0, 0.0), DLP, MU_R, NPER, 10000.0, 64)
    with pytest.raises(AssertionError):      # lam = 3 pushes hprob out of bounds
        PM.prob_mass(days[0], wd, (3.0,) + HP[1:], DP, DLP, MU_R, NPER, 10000.0, 64)


def test_prob_mass_random_parameters_against_oracle(PM, kalbar, carnarvon):
    '''Twelve random draws of the model parameters (flight/wind logistic parameters, diffusion
    with correlations of both signs, mu_r, n_periods, grid resolution, start time) on random
    days of both wind files against the oracle: identical COO pattern, values to 5e-15.'''
    rng = np.random.default_rng(314159)
    for case in range(12):
        wd, days = kalbar if case % 2 == 0 else carnarvon
        day = days[int(rng.integers(0, len(days) - 1))]
        hp = (float(rng.uniform(0.6, 1.0)), float(rng.uniform(0.5, 2.5)), float(rng.uniform(2.0, 6.0)),
              float(rng.uniform(5.0, 8.0)), float(rng.uniform(1.5, 4.0)), float(rng.uniform(18.0, 23.0)),
              float(rng.uniform(1.5, 4.0)))
        dp = (float(rng.uniform(80, 260)), float(rng.uniform(80, 220)), float(rng.uniform(-0.6, 0.6)))
        dlp = (float(rng.uniform(3, 25)), float(rng.uniform(3, 25)), float(rng.uniform(-0.5, 0.5)))
        mu_r = float(rng.uniform(0.6, 1.6))
        npd = int(rng.integers(5, 50))
        R = int(rng.integers(24, 64))
        st = None if case % 3 else float(rng.uniform(0.1, 0.6))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            ref = OM.prob_mass(day, wd, hp, dp, dlp, mu_r, npd, 10000.0, R, st)
            got = PM.prob_mass(day, wd, hp, dp, dlp, mu_r, npd, 10000.0, R, st)
        msg = 'case %d day %d R=%d' % (case, day, R)
        assert got.shape == ref.shape, msg
        a, b = got.tocsr(), ref.tocsr()
        a.sort_indices(); b.sort_indices()
        assert a.nnz == b.nnz and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), msg
        np.testing.assert_allclose(a.data, b.data, rtol=0, atol=VAL_ATOL, err_msg=msg)
        assert abs(got.sum() - 1.0) < 1e-12


def test_prob_mass_large_grids_against_reference(PM, golden, kalbar, carnarvon):
    '''G5b: the REFERENCE's prob_mass at the grid sizes of BASELINE configs 2-5 -- Kalbar R = 512
    (stamp 59^2), a prior-drawn ensemble member (config 5, member 0) on Carnarvon at R = 1024
    (117^2), Carnarvon at R = 2048 for a late release (235^2 stamps, one stamp over ~15 x 15 pmf
    tiles): identical COO pattern, sampled values to 5e-15.'''
    g = golden('g5b_prob_mass_large')
    wd, days = kalbar
    wc, dc = carnarvon
    assert int(g['kal512_day']) == days[0] and int(g['car1024_day']) == dc[3] and int(g['car2048_day']) == dc[0]
    lam, sx, sy, mu = (float(v) for v in g['member0'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        check_digest(g, 'kal512', PM.prob_mass(days[0], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 512))
        check_digest(g, 'car1024_member0',
                      PM.prob_mass(dc[3], wc, (lam,) + tuple(HP[1:]), (sx, sy, 0.253), DLP, mu, NPER, 10000.0, 1024))
        check_digest(g, 'car2048_late', PM.prob_mass(dc[0], wc, HP, DP, DLP, MU_R, NPER, 10000.0, 2048, 0.93))


def test_prob_mass_pair_list_overflow_is_redone(PM, kalbar, monkeypatch):
    '''The pair stage of a batch is enqueued with list capacities guessed from the previous batches of
    the same shape (no host sync behind the scan).  A batch with far more (tile, period) pairs than
    any before it does not fit: it must be detected and redone with exact sizes -- bit-identical to
    the always-exact path (PS_PM_SYNC=1) -- and so must the batches after it.'''
    wd, days = kalbar
    narrow = (DP[0] * 0.3, DP[1] * 0.3, DP[2])
    wide = (DP[0] * 2.0, DP[1] * 2.0, DP[2])
    seq = [narrow, narrow, wide, narrow, wide]

    def run():
        m = PM.WindModel(wd)
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            for dp in seq:
                m.build(days[:3], HP, dp, DLP, MU_R, NPER, 10000.0, 128)
                out.append([m.fetch(i) for i in range(3)])
        m.close()
        return out

    monkeypatch.delenv('PS_PM_SYNC', raising=False)
    got = run()
    monkeypatch.setenv('PS_PM_SYNC', '1')
    ref = run()
    assert sum(r.nnz for r in ref[2]) > 2 * sum(r.nnz for r in ref[0])     # the wide batch really is much bigger
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert x.shape == y.shape and x.nnz == y.nnz
            assert np.array_equal(x.row, y.row) and np.array_equal(x.col, y.col) and np.array_equal(x.data, y.data)


def test_prob_mass_day_does_not_depend_on_its_batch_beyond_round_off(PM, kalbar):
    '''with PS_PM_SEG = 8 (default) eight periods are summed per record before the
    ordered per-tile pass, and the runs start at multiples of 8 of the CHUNK-global pair index -- so how a
    tile's periods are grouped depends on the pair counts of the tiles and days ahead of it in the batch.
    The grouping changes the order of a few additions, nothing else: a day built alone and the same day
    built behind two others agree to 1e-15 with the same COO pattern; with PS_PM_SEG = 1 (strictly
    sequential sums, the reference's order) they are bit-identical.'''
    wd, days = kalbar
    for seg, exact in ((8, False), (1, True)):
        m = PM.WindModel(wd)
        m.set_option('PS_PM_SEG', seg)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            m.build([days[2]], HP, DP, DLP, MU_R, NPER, 10000.0, 128)
            alone = m.fetch(0)
            m.build(days[:3], HP, DP, DLP, MU_R, NPER, 10000.0, 128)
            batched = m.fetch(2)
        m.close()
        assert alone.shape == batched.shape and alone.nnz == batched.nnz
        assert np.array_equal(alone.row, batched.row) and np.array_equal(alone.col, batched.col)
        if exact:
            assert np.array_equal(alone.data, batched.data)
        else:
            assert np.abs(alone.data - batched.data).max() < 1e-15


def test_unrolled_pair_masses_are_bit_identical(PM, kalbar):
    '''k_pair_masses<false, 3>: the Gauss-Legendre node count as a compile-time constant and the
    device library's exp() written out stage by stage over the 6 exponentials of a corner
    (pm_exp_many: same constants, same operations, same order) -- against the run-time-count instance
    that calls exp() (PS_PM_NO_UNROLL): identical kernels bit for bit for |rho| < 0.3 (3 node pairs);
    0.3 <= |rho| < 0.75 (6 pairs) takes the run-time-count instance either way.'''
    wd, days = kalbar
    for rho in (0.253, -0.1, 0.5, -0.7):
        out = []
        for no_unroll in (0, 1):
            m = PM.WindModel(wd)
            m.set_option('PS_PM_NO_UNROLL', no_unroll)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', RuntimeWarning)
                m.build(days[:3], HP, (DP[0], DP[1], rho), DLP, MU_R, NPER, 10000.0, 128)
                out.append([m.fetch(i) for i in range(3)])
            m.close()
        for a, b in zip(*out):
            assert a.shape == b.shape and a.nnz == b.nnz
            assert np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col) and np.array_equal(a.data, b.data)
