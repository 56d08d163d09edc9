"""GPU parity of the Bayes_funcs mirrors (solution -> expected observations through device
gathers) against G9: the reference's Bayes_funcs run on the reference's Kalbar R=128
population solution with a synthetic LocInfo stand-in (tests/golden/make_golden.py:g9)."""
import os
import types

import numpy as np
import pandas as pd
import pytest

from helpers import HP, DP, DLP, MU_R, NPER

pytestmark = pytest.mark.gpu


def locinfo_from(g):
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
    li.card_obs_datesPR = [td(d) for d in g['card_obs_days']]
    li.card_obs = [np.zeros((4, int(n))) for n in g['card_obslen']]
    li.step_size = [int(v) for v in g['step_size']]
    return li


def test_bayes_funcs_against_reference(golden, golden_dir):
    from parasitoids_amd import ParasitoidModel as PM, Bayes_funcs as BF
    from parasitoids_amd.pop_model import PopModel
    g = golden('g9_bayes_funcs')
    li = locinfo_from(g)
    wd, days = PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '00:00')
    pm = PopModel(wd, days, domain_info=(10000.0, 128), r_number=130000, mode='exact')
    pm.evaluate(HP, DP, DLP, MU_R, NPER, ndays=6)
    rel, sen = BF.popdensity_to_emergence(pm, li)
    for i in range(2):
        assert rel[i].shape == g['rel%d' % i].shape and sen[i].shape == g['sen%d' % i].shape
        np.testing.assert_allclose(rel[i], g['rel%d' % i], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(sen[i], g['sen%d' % i], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(BF.popdensity_grid(pm, li), g['grid'], rtol=1e-10, atol=1e-9)
    card = BF.popdensity_card(pm, li, (10000.0, 128))
    for i in range(2):
        np.testing.assert_allclose(card[i], g['card%d' % i], rtol=1e-10, atol=1e-9)
    assert g['grid'].max() > 1.0 and g['sen0'].max() > 1.0     # the fixture is not trivially zero
    pm.close()


def test_metropolis_sampler_runs_and_is_reproducible():
    '''parasitoids_amd.mcmc: priors/likelihood plumbing around PopModel.  Two chains with the
    same seed give identical traces (the device path is deterministic); the log-posterior is
    finite; the sampler moves.'''
    import os
    import warnings
    from parasitoids_amd import ParasitoidModel as PM, mcmc
    from parasitoids_amd.pop_model import PopModel
    warnings.simplefilter('ignore', RuntimeWarning)
    root = os.path.dirname(os.path.abspath(__file__))
    wd, days = PM.get_wind_data(os.path.join(root, 'golden', 'data', 'kalbar'), 30, '00:00')
    traces = []
    for rep in range(2):
        pm = PopModel(wd, days, domain_info=(10000.0, 128), r_number=130000, mode='auto')
        li = mcmc.synthetic_locinfo(pm, 128, seed=9, ndays=8)
        assert sum(int(a.sum()) for a in li.release_emerg) > 0 and int(li.grid_obs.sum()) > 0
        chain = mcmc.Metropolis(pm, li, (10000.0 / 128) ** 2, seed=5, ndays=8)
        res = chain.run(12)
        assert np.all(np.isfinite(res['logp']))
        assert res['evaluations'] >= 12 and 0.0 <= res['acceptance'] <= 1.0
        traces.append(res['trace'])
        pm.close()
    assert np.array_equal(traces[0], traces[1])
    assert np.ptp(traces[0][:, -3]) > 0            # the nuisance parameters move



def test_metropolis_on_kalbar_field_data():
    '''the sampler on the real Kalbar observations (Data_Import.LocInfo on the CSV fixtures):
    finite log-posterior, deterministic, expected observations shaped like the data'''
    import os
    import warnings
    from parasitoids_amd import ParasitoidModel as PM, mcmc
    from parasitoids_amd.Data_Import import LocInfo
    from parasitoids_amd.pop_model import PopModel
    warnings.simplefilter('ignore', RuntimeWarning)
    root = os.path.dirname(os.path.abspath(__file__))
    wd, days = PM.get_wind_data(os.path.join(root, 'golden', 'data', 'kalbar'), 30, '00:00')
    R = 200
    li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, R))
    logps = []
    for rep in range(2):
        pm = PopModel(wd, days, domain_info=(10000.0, R), r_number=130000, mode='auto')
        chain = mcmc.Metropolis(pm, li, (10000.0 / R) ** 2, seed=3)
        rel, sen, grid = chain.expected
        assert rel[0].shape == li.release_emerg[0].shape and sen[0].shape == li.sentinel_emerg[0].shape
        assert grid.shape == li.grid_obs.shape and grid.sum() > 0
        res = chain.run(6)
        assert np.all(np.isfinite(res['logp']))
        logps.append(res['logp'])
        pm.close()
    assert np.array_equal(logps[0], logps[1])


def test_gather_days_equals_per_day_gather():
    """PopModel.gather_days (one ps_record_gather_multi call) against one ps_record_gather per
    day: identical values, including day 0 (the state record) and repeated cells."""
    from parasitoids_amd import hip_lib
    from parasitoids_amd import _lib as L
    from parasitoids_amd import synthetic
    R, K, nd = 64, 33, 5
    state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=11, sigma=(1.5, 3.0), shift=2.0)
    s = hip_lib.HipSolve(state, [K, K], mode='exact')
    s.set_kernels(kernels)
    s.run_chain(renorm=False)
    rng = np.random.default_rng(0)
    rows = rng.integers(R - 20, R + 20, 300)
    cols = rng.integers(R - 20, R + 20, 300)
    kinds = [L.REC_STATE] + [L.REC_CHAIN] * nd
    idxs = [0] + list(range(nd))
    multi = s.gather_multi(kinds, idxs, rows, cols, scale=1e4, negval=1e-8)
    assert multi.shape == (nd + 1, 300)
    for n, (k, i) in enumerate(zip(kinds, idxs)):
        assert np.array_equal(multi[n], s.gather(k, i, rows, cols, scale=1e4, negval=1e-8))
    assert multi[1:].max() > 0
    assert s.gather_multi([], [], rows, cols).shape == (0, 300)
    s.close()
