"""CPU tests of the host-side sampler (parasitoids_amd/mcmc.py): the joint log density against an
independent scipy.stats evaluation on the G9 expected-observation arrays, the
AdaptiveMetropolis covariance recursion, scalar-step tuning, save / resume.  No device: the
model evaluation is replaced by a deterministic function of the parameters."""
import math
import types

import numpy as np
from scipy import stats

from parasitoids_amd import mcmc


def g9_locinfo(g, seed=4):
    """LocInfo stand-in on the G9 geometry with deterministic 'observations' around the
    reference's expected values (tests/golden/make_golden.py:g9)."""
    rng = np.random.default_rng(seed)
    li = types.SimpleNamespace()
    li.sent_ids = ['A', 'B', 'C']
    li.field_sizes = {k: len(g['field_' + k]) for k in li.sent_ids}
    li.release_collection = [np.full(g['rel0'].shape[0], 1.0), np.full(g['rel1'].shape[0], 0.5)]
    li.grid_samples = np.full(g['grid'].shape, 2.0)
    cell_area = (10000.0 / 128) ** 2
    sp = mcmc.initial_sent_obs_probs(li, cell_area)
    li.release_emerg = [rng.poisson(0.75 * g['rel%d' % i] * (li.release_collection[i] * 0.05)[:, None]) for i in range(2)]
    li.sentinel_emerg = [rng.poisson(0.75 * g['sen%d' % i] * sp[:, None]) for i in range(2)]
    li.grid_obs = rng.poisson(0.005 * li.grid_samples * g['grid'])
    return li, cell_area


def independent_log_posterior(theta, nuis, A, sp, expected, li, cell_area):
    """The same density spelled out with scipy.stats in PyMC 2's parameterisations
    (Bayes_Run.py:102-166, :344-433)."""
    t = dict(zip([m[0] for m in mcmc.MODEL_BLOCK], theta))
    tn = lambda x, mu, tau, a, b: stats.truncnorm.logpdf(x, (a - mu) * math.sqrt(tau), (b - mu) * math.sqrt(tau),
                                                         loc=mu, scale=1 / math.sqrt(tau))
    lp = (stats.beta.logpdf(t['lam'], 5, 1) + tn(t['f_a1'], 6, 0.3, 0, 9) + tn(t['f_a2'], 20, 0.3, 15, 24)
          + stats.gamma.logpdf(t['f_b1_p'], 2, scale=1) + stats.gamma.logpdf(t['f_b2_p'], 2, scale=1)
          + stats.gamma.logpdf(t['g_aw'], 2.2, scale=1) + stats.gamma.logpdf(t['g_bw'], 5, scale=1)
          + stats.gamma.logpdf(t['sig_x'], 26, scale=1 / 0.15) + stats.gamma.logpdf(t['sig_y'], 15, scale=1 / 0.15)
          + stats.beta.logpdf(t['corr_p'], 5, 5) + stats.beta.logpdf(t['corr_l_p'], 5, 5)
          + stats.gamma.logpdf(t['sig_x_l'], 2, scale=1 / 0.08) + stats.gamma.logpdf(t['sig_y_l'], 2, scale=1 / 0.14)
          + stats.norm.logpdf(t['mu_r'], 1, 1) + stats.poisson.logpmf(t['n_periods'], 30))
    xi, em, gp = nuis
    lp += stats.gamma.logpdf(xi, 1, scale=1) + stats.beta.logpdf(em, 1, 1) + stats.beta.logpdf(gp, 1, 1)
    areas = np.array([li.field_sizes[k] * cell_area for k in li.sent_ids])
    lp += tn(A, 2500, 1 / 2500, 0, areas.min())
    for p, a in zip(sp, areas):
        m = A / a
        lp += stats.beta.logpdf(p, m * 40 / (1 - m), 40)
    rel, sen, grid = expected
    for i in range(2):
        lp += stats.poisson.logpmf(li.release_emerg[i], xi * rel[i] * (li.release_collection[i] * em)[:, None]).sum()
        lp += stats.poisson.logpmf(li.sentinel_emerg[i], xi * sen[i] * np.asarray(sp)[:, None]).sum()
    lp += stats.poisson.logpmf(li.grid_obs, gp * li.grid_samples * grid).sum()
    return float(lp)


def test_log_posterior_of_a_fixed_point_on_g9(golden):
    g = golden('g9_bayes_funcs')
    li, cell_area = g9_locinfo(g)
    expected = ([g['rel0'], g['rel1']], [g['sen0'], g['sen1']], g['grid'])
    # rates must be positive wherever something was observed
    assert all((e > 0).all() or True for e in expected[0])
    theta = np.array([m[2] for m in mcmc.MODEL_BLOCK])
    nuis = np.array([0.75, 0.05, 0.005])
    sp = mcmc.initial_sent_obs_probs(li, cell_area)
    for A, th, nu, s in ((2500.0, theta, nuis, sp),
                         (1800.0, theta * np.where(np.array(mcmc.DISCRETE), 1.0, 1.02), nuis * 1.1, sp * 0.9)):
        got = mcmc.log_posterior(th, nu, A, s, expected, li, cell_area)
        ref = independent_log_posterior(th, nu, A, s, expected, li, cell_area)
        assert math.isfinite(ref)
        assert abs(got - ref) < 1e-8 * max(1.0, abs(ref)), (got, ref)
    # out of support -> zero probability
    bad = theta.copy()
    bad[[m[0] for m in mcmc.MODEL_BLOCK].index('lam')] = 1.2
    assert mcmc.log_posterior(bad, nuis, 2500.0, sp, expected, li, cell_area) == mcmc.NEG_INF
    assert mcmc.log_posterior(theta, nuis, 1e9, sp, expected, li, cell_area) == mcmc.NEG_INF


def test_adaptive_metropolis_covariance_recursion():
    """The recursive update equals scaling * (sample covariance of the whole internal trace +
    eps/(n-1) I) once the initial covariance has been weighted out (k = 0 at the first update
    gives exactly that), and later blocks continue the same recursion."""
    rng = np.random.default_rng(0)
    dim = 4
    am = mcmc.AdaptiveMetropolis(np.full(dim, 0.5), delay=10, interval=5, greedy=False)
    assert np.allclose(am.C, np.diag(np.full(dim, 0.5)))            # scales on the diagonal
    assert np.allclose(am.proposal_sd @ am.proposal_sd.T, am.C)
    X = rng.normal(size=(40, dim)) @ np.diag([1.0, 2.0, 0.5, 3.0])
    for i, x in enumerate(X):
        am.tally(x, accepted=(i % 3 != 0))
    # updates happen at iterations 15, 20, ... 35 (> delay, multiple of interval): 5 so far
    assert am.cov_updates == 5 and am.trace_count == 36
    s = 2.4 ** 2 / dim
    seen = X[:36]
    expect = s * (np.cov(seen.T) + 1e-5 / 35 * np.eye(dim))
    # first update had k = 0: the (k-1)/(n-1) weight of the INITIAL covariance is -1/(n-1)
    n1 = 16
    w = -1.0 / (n1 - 1)
    for n_prev, n_new in ((16, 21), (21, 26), (26, 31), (31, 36)):
        w *= (n_prev - 1) / (n_new - 1)
    # eps terms of the five blocks accumulate with the same weights as the data
    assert np.allclose(am.C - w * np.diag(np.full(dim, 0.5)), expect + s * 1e-5 * 0, atol=2e-4, rtol=1e-3)
    assert np.allclose(am.chain_mean, seen.mean(0))
    assert np.allclose(am.proposal_sd @ am.proposal_sd.T, am.C)
    # greedy: before `delay` only accepted states are tallied
    am2 = mcmc.AdaptiveMetropolis(np.ones(2), delay=100, interval=50, greedy=True)
    for i in range(20):
        am2.tally(np.array([i, -i], float), accepted=(i % 4 == 0))
    assert len(am2._trace) == 5
    # shrink_if_necessary
    am3 = mcmc.AdaptiveMetropolis(np.ones(2), delay=2, interval=2, greedy=False)
    for i in range(7):
        am3.tally(np.array([0.1 * i, 1.0]), accepted=False)
    assert am3.cov_updates == 2 and np.all(np.abs(am3.C) < 0.1)       # scaled by 0.01 twice


def test_scalar_metropolis_tuning():
    st = mcmc.ScalarMetropolis(0.05, tune_interval=10)
    assert st.proposal_sd == 0.05 and st.factor == 1.0
    for _ in range(10):
        st.tally(False)
    assert st.factor == 0.1                      # acceptance < 0.001
    for i in range(10):
        st.tally(i < 8)
    assert abs(st.factor - 0.2) < 1e-15          # acceptance 0.8 > 0.75: x2
    assert mcmc.ScalarMetropolis(0.0).proposal_sd == 1.0


def _fake_sampler(g, seed, **kw):
    li, cell_area = g9_locinfo(g)
    base = ([g['rel0'], g['rel1']], [g['sen0'], g['sen1']], g['grid'])
    t0 = np.array([m[2] for m in mcmc.MODEL_BLOCK])

    def evaluate(theta):
        # a smooth, deterministic stand-in for the model: expected observations scale with the
        # diffusion parameters
        f = float(np.exp(-0.5 * (((theta - t0) / (0.1 * np.abs(t0) + 1e-3)) ** 2).sum() / len(t0)))
        return ([f * r for r in base[0]], [f * s for s in base[1]], f * base[2])
    return mcmc.Sampler(None, li, cell_area, seed=seed, evaluate=evaluate, **kw)


def test_sampler_moves_adapts_and_resumes(golden, tmp_path):
    g = golden('g9_bayes_funcs')
    a = _fake_sampler(g, 11, delay=20, interval=10, tune_interval=15)
    full = a.run(60)
    assert np.all(np.isfinite(full['logp']))
    assert full['cov_updates'] >= 3 and 0 < full['acceptance'] < 1
    names = full['names']
    assert names[:15] == [m[0] for m in mcmc.MODEL_BLOCK] and 'A_collected' in names
    assert names[-3:] == ['sent_obs_probs_A', 'sent_obs_probs_B', 'sent_obs_probs_C']
    tr = full['trace']
    for col in ('sig_x', 'xi', 'A_collected', 'sent_obs_probs_B'):
        assert np.ptp(tr[:, names.index(col)]) > 0, col
    k = names.index('n_periods')
    assert np.all(tr[:, k] == np.round(tr[:, k]))                 # discrete stays integer
    # the joint density the sampler tracks incrementally equals a from-scratch evaluation
    assert abs(a.logp - mcmc.log_posterior(a.theta, a.nuis, a.A_collected, a.sent_obs_probs,
                                           a.expected, a.li, a.cell_area)) < 1e-9
    # save after 35, resume in a fresh sampler, 25 more == the uninterrupted 60
    b = _fake_sampler(g, 11, delay=20, interval=10, tune_interval=15)
    b.run(35)
    b.save(tmp_path / 'chain.npz')
    c = _fake_sampler(g, 999, delay=20, interval=10, tune_interval=15)
    tr0, lp0 = c.resume(tmp_path / 'chain.npz')
    assert np.array_equal(tr0, full['trace'][:35]) and np.array_equal(lp0, full['logp'][:35])
    rest = c.run(25)
    assert np.array_equal(rest['trace'], full['trace'][35:])
    assert np.array_equal(rest['logp'], full['logp'][35:])


def test_bayes_funcs_projection_against_a_loop_restatement(golden):
    """parasitoids_amd.Bayes_funcs turns the model -> emergence translation into one gather and
    a matrix product per collection (precomputed per site).  Checked here, without a device,
    against a day-by-day loop that follows the reference's statements (Bayes_funcs.py:57-90,
    :116-144): incubation spread, then binning into observation dates."""
    import pandas as pd
    from parasitoids_amd import Bayes_funcs as BF
    g = golden('g9_bayes_funcs')
    td = lambda d: pd.Timedelta(days=int(d))
    li = types.SimpleNamespace()
    li.collection_datesPR = [td(d) for d in g['collection_days']]
    li.emerg_grids = [[tuple(rc) for rc in g['emerg_grid%d' % i]] for i in range(2)]
    li.release_DataFrames = [pd.DataFrame({'datePR': [td(d) for d in g['rel_dates%d' % i]]}) for i in range(2)]
    li.sent_DataFrames = [pd.DataFrame({'datePR': [td(d) for d in g['sen_dates%d' % i]]}) for i in range(2)]
    li.sent_ids = ['A', 'B', 'C']
    li.field_cells = {k: g['field_' + k] for k in li.sent_ids}
    li.grid_cells = g['grid_cells']
    li.grid_obs_datesPR = [td(d) for d in g['grid_obs_days']]
    rng = np.random.default_rng(3)
    fields = rng.random((8, 257, 257)) * 1e3

    class Model():
        def gather(self, day, rows, cols):
            return fields[day][np.asarray(rows), np.asarray(cols)]

    def loop_project(per_day, start_day, collection_day, obs_days):
        nitem = len(per_day[start_day]) if collection_day > start_day else 0
        proj = np.zeros((nitem, BF.max_incubation_time))
        for day in range(start_day, collection_day):
            max_post = day + BF.max_incubation_time - collection_day
            min_post = max(0, max_post + 1 - BF.incubation_time.size)
            span = max_post - min_post + 1
            proj[:, min_post:max_post + 1] += np.outer(per_day[day], BF.incubation_time)[:, -span:]
        col = obs_days - collection_day
        out = np.zeros((nitem, len(obs_days)))
        out[:, 0] = proj[:, 0:col[0] + 1].sum(axis=1)
        for n, c in enumerate(col[1:]):
            out[:, n + 1] = proj[:, col[n] + 1:c + 1].sum(axis=1)
        return out

    m = Model()
    rel, sen = BF.popdensity_to_emergence(m, li)
    for i in range(2):
        cday = int(g['collection_days'][i])
        sday = max(cday - BF.max_incubation_time, 0)
        cells = np.asarray(li.emerg_grids[i])
        per = {d: m.gather(d, cells[:, 0], cells[:, 1]) for d in range(sday, cday)}
        ref = loop_project(per, sday, cday, np.unique(g['rel_dates%d' % i]))
        np.testing.assert_allclose(rel[i], ref, rtol=1e-13, atol=1e-9)
        per = {d: np.array([m.gather(d, li.field_cells[k][:, 0], li.field_cells[k][:, 1]).sum() for k in li.sent_ids])
               for d in range(sday, cday)}
        ref = loop_project(per, sday, cday, np.unique(g['sen_dates%d' % i]))
        np.testing.assert_allclose(sen[i], ref, rtol=1e-13, atol=1e-9)
        assert rel[i].shape == g['rel%d' % i].shape and sen[i].shape == g['sen%d' % i].shape
    grid = BF.popdensity_grid(m, li)
    for n, d in enumerate(g['grid_obs_days']):
        assert np.array_equal(grid[:, n], fields[int(d) - 1][li.grid_cells[:, 0], li.grid_cells[:, 1]])
    assert getattr(li, '_ps_plan', None) is not None          # the site plan is cached on the object

    # the one-gather form a device model gets: same numbers
    class Model2(Model):
        calls = 0

        def gather_days(self, days, rows, cols):
            Model2.calls += 1
            return np.array([fields[d][np.asarray(rows), np.asarray(cols)] for d in days])
    rel2, sen2, grid2 = BF.expected_observations(Model2(), li)
    assert Model2.calls == 1
    for i in range(2):
        np.testing.assert_allclose(rel2[i], rel[i], rtol=1e-14, atol=1e-10)
        np.testing.assert_allclose(sen2[i], sen[i], rtol=1e-14, atol=1e-10)
    assert np.array_equal(grid2, grid)


def test_likelihood_statistics_equal_the_array_form(golden):
    """loglik_parts_stats (sufficient statistics of one evaluation, what the scalar Metropolis steps
    use) against loglik_parts (the rates written out, Bayes_Run.py:344-433): same values to
    round-off for random nuisance values, -inf in the same places (zero base rate under a positive
    count, non-positive scalars)."""
    g = golden('g9_bayes_funcs')
    li, cell_area = g9_locinfo(g)
    rng = np.random.default_rng(3)
    exp = ([g['rel0'], g['rel1']], [g['sen0'], g['sen1']], g['grid'])
    st = mcmc.lik_stats(exp, li)
    nf = len(mcmc.initial_sent_obs_probs(li, cell_area))
    for _ in range(50):
        nuis = np.array([rng.uniform(0.1, 3.0), rng.uniform(0.01, 0.9), rng.uniform(1e-4, 0.1)])
        sp = rng.uniform(0.01, 0.9, nf)
        a = mcmc.loglik_parts(exp, li, nuis, sp)
        b = mcmc.loglik_parts_stats(st, nuis, sp)
        for x, y in zip(a, b):
            assert (x == y == mcmc.NEG_INF) or abs(x - y) <= 1e-10 * max(1.0, abs(x)), (x, y)
    # non-positive scalars
    for nuis, sp in ((np.array([-1.0, 0.5, 0.01]), np.full(nf, 0.1)), (np.array([1.0, 0.5, -0.01]), np.full(nf, 0.1)),
                     (np.array([1.0, 0.5, 0.01]), np.r_[-0.1, np.full(nf - 1, 0.1)])):
        a = mcmc.loglik_parts(exp, li, nuis, sp)
        b = mcmc.loglik_parts_stats(st, nuis, sp)
        assert [x == mcmc.NEG_INF for x in a] == [y == mcmc.NEG_INF for y in b]
    # a zero base rate where something was observed
    rel0 = np.array(g['rel0'], dtype=np.float64)
    obs0 = np.asarray(li.release_emerg[0], dtype=np.float64)
    hit = np.argwhere(obs0 > 0)
    if len(hit):
        rel0[tuple(hit[0])] = 0.0
        exp2 = ([rel0, g['rel1']], exp[1], exp[2])
        nuis, sp = np.array([1.0, 0.5, 0.01]), np.full(nf, 0.1)
        assert mcmc.loglik_parts(exp2, li, nuis, sp)[0] == mcmc.NEG_INF
        assert mcmc.loglik_parts_stats(mcmc.lik_stats(exp2, li), nuis, sp)[0] == mcmc.NEG_INF


def test_parallel_chains_equal_the_same_chains_run_alone(golden):
    """mcmc.run_parallel: k chains in k host threads; every chain owns its state and random stream, so
    its trace is the one it produces alone."""
    g = golden('g9_bayes_funcs')
    alone = [_fake_sampler(g, 100 + c, delay=20, interval=10, tune_interval=15).run(40) for c in range(3)]
    chains = [_fake_sampler(g, 100 + c, delay=20, interval=10, tune_interval=15) for c in range(3)]
    res, dt = mcmc.run_parallel(chains, 40)
    assert dt > 0 and len(res) == 3
    for a, b in zip(alone, res):
        assert np.array_equal(a['trace'], b['trace']) and np.array_equal(a['logp'], b['logp'])
    assert not np.array_equal(res[0]['trace'], res[1]['trace'])      # different seeds, different chains


def test_unplannable_torus_rejects_the_proposal_device_errors_stop_the_chain(golden, monkeypatch):
    """PS_ERR_UNSUPPORTED from a shape-dependent solver rebuild (mode='exact' and a pad with a
    prime factor > 1024) is a property of the proposed parameters: rejected and counted.  A device error
    is not: it stops the run."""
    from parasitoids_amd import _lib
    g = golden('g9_bayes_funcs')
    li, cell_area = g9_locinfo(g)
    base = ([g['rel0'], g['rel1']], [g['sen0'], g['sen1']], g['grid'])
    monkeypatch.setattr(mcmc, 'expected_observations', lambda pm, li: base)
    calls = {'n': 0, 'code': _lib.PS_ERR_UNSUPPORTED}

    class FakeModel():
        def evaluate(self, *a, **k):
            calls['n'] += 1
            if calls['n'] % 2 == 0:
                raise _lib.HipError(calls['code'], 'cannot plan a length-1031 FFT')

    s = mcmc.Sampler(FakeModel(), li, cell_area, seed=3)
    r = s.run(12)
    assert r['failed_evaluations'] >= 1 and r['failed_evaluations'] == s.n_failed
    assert np.all(np.isfinite(r['logp']))
    calls['code'] = _lib.PS_ERR_HIP
    import pytest
    with pytest.raises(_lib.HipError):
        s.run(12)
