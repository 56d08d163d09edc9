"""GPU parity of the device-resident population-model evaluation (the body of the
reference's Bayes_Run.pop_model) against G7: get_populations on Kalbar, R=400, 18 days
(reference settings, Bayes_Run.py:91; P = 1121 = 19*59, exact torus)."""
import os

import numpy as np
import pytest

from helpers import HP, DP, DLP, MU_R, NPER, coo_from, assert_summary

pytestmark = pytest.mark.gpu


def test_pop_model_kalbar_r400(golden, golden_dir):
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    g = golden('g7_populations')
    wd, days = PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '00:00')
    pm = PopModel(wd, days, domain_info=(10000.0, 400), r_number=130000)
    stats = pm.evaluate(HP, DP, DLP, MU_R, NPER)
    assert len(stats) == 18
    assert int(pm.model.last['kshape'].max()) == int(g['r400_max_shape'][0])
    # the device-built day kernels equal the reference's
    for i in (0, 7, 17):
        ref = coo_from(g, 'r400_pmf%d' % i)
        got = pm.model.fetch(i)
        assert got.shape == ref.shape and got.nnz == ref.nnz
        assert np.abs(got.data - ref.data).max() < 5e-15
    pos = g['r400_pos']
    for d in range(18):
        pop = pm.population(d)
        assert_summary(g, 'r400_sum%d' % d, pop, pos, rtol=1e-9, atol=1e-7, nnz_slack=6)
        assert abs(stats[d][1] - float(g['r400_sum%d_sum' % d])) < 1e-6 * 130000
    # second evaluation with other parameters reuses the context; first one is reproducible
    s2 = pm.evaluate(HP, (150.0, 160.0, -0.1), DLP, 1.0, NPER)
    assert s2[5][0] != stats[5][0]
    s3 = pm.evaluate(HP, DP, DLP, MU_R, NPER)
    assert [a[0] for a in s3] == [a[0] for a in stats]
    assert max(abs(a[1] - b[1]) for a, b in zip(s3, stats)) == 0.0      # bitwise reproducible
    v = pm.gather(3, [400, 390], [400, 410])
    ref = pm.population(3)
    assert abs(v[0] - ref[400, 400]) < 1e-9 and abs(v[1] - ref[390, 410]) < 1e-9
    pm.close()


def test_pop_model_fast_mode_reuses_solvers_and_matches_exact():
    '''fast mode keeps one solver per FFT size class and reuses it when the kernel extent moves
    with the diffusion parameters; the populations must agree with exact-mode evaluations of
    the same parameters to the fast-mode tolerance (the tori differ only in sub-threshold dust).'''
    import warnings
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    from helpers import HP, DLP, MU_R, NPER
    warnings.simplefilter('ignore', RuntimeWarning)
    root = os.path.dirname(os.path.abspath(__file__))
    wd, days = PM.get_wind_data(os.path.join(root, 'golden', 'data', 'kalbar'), 30, '00:00')
    fast = PopModel(wd, days, domain_info=(10000.0, 200), r_number=130000, mode='fast')
    exact = PopModel(wd, days, domain_info=(10000.0, 200), r_number=130000, mode='exact')
    sizes = set()
    for sx, sy in ((171.82, 144.58), (150.0, 130.0), (185.0, 160.0), (171.82, 144.58)):
        dp = (sx, sy, 0.253)
        sf = fast.evaluate(HP, dp, DLP, MU_R, NPER, ndays=8)
        se = exact.evaluate(HP, dp, DLP, MU_R, NPER, ndays=8)
        sizes.add(fast.solver.fft_len)
        for (nf, tf), (ne, te) in zip(sf, se):
            assert abs(tf - te) <= 1e-6 * max(te, 1.0)
            assert abs(nf - ne) <= 0.002 * ne + 50        # entries right at the 1e-8 cut may flip
        a = fast.population(7).toarray()
        b = exact.population(7).toarray()
        assert np.abs(a - b).max() <= 5e-8 * 130000
    assert len(fast._solvers) <= 2 and len(sizes) <= 2     # kernel shapes moved, solvers did not
    fast.close(); exact.close()


def test_chains_side_by_side_on_one_gpu_are_bitwise_the_chains_alone():
    '''k chains in one process (mcmc.run_parallel: a host thread, a PopModel with its own
    model / solver handles and streams per chain) fill the card a single R = 400-class chain leaves idle.
    What runs next to a chain must not change it: traces bit-identical to the same seeds run one at a time.'''
    import warnings
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd import mcmc
    from parasitoids_amd.Data_Import import LocInfo
    from parasitoids_amd.pop_model import PopModel
    warnings.simplefilter('ignore', RuntimeWarning)
    R = 200
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, R))

    def chain(seed):
        pm = PopModel(wd, days, domain_info=(10000.0, R), r_number=130000, mode='auto')
        return pm, mcmc.Sampler(pm, li, (10000.0 / R) ** 2, seed=seed)

    alone = []
    for c in range(3):
        pm, s = chain(1000 + c)
        alone.append(s.run(25))
        pm.close()
    pairs = [chain(1000 + c) for c in range(3)]
    res, dt = mcmc.run_parallel([s for _, s in pairs], 25)
    for pm, _ in pairs:
        pm.close()
    for a, b in zip(alone, res):
        assert np.array_equal(a['trace'], b['trace']) and np.array_equal(a['logp'], b['logp'])
        assert b['evaluations_this_run'] == a['evaluations_this_run'] > 0


_DEVICE_EXCHANGE = r'''
import sys, warnings
import numpy as np
import torch                      # before the library: one HIP runtime per process (torch's wheel brings its own)
torch.cuda.init()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
from helpers import HP, DP, DLP, MU_R, NPER
from parasitoids_amd import ParasitoidModel as PM
from parasitoids_amd import hip_lib, parallel, _lib as L
warnings.simplefilter('ignore', RuntimeWarning)
R, nd = 200, 8
N = 2 * R + 1
wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
model = PM.WindModel(wd)
params = (HP, DP, DLP, MU_R, NPER, 10000.0, R)
g = parallel.prob_mass_sharded_device(model, days[:nd], params)
g['dom_len'] = N
assert g['row'].is_cuda and g['val'].dtype == torch.float64 and len(g['kshape']) == nd
for i in (0, 3, nd - 1):
    ref = model.fetch(i)
    o, e = int(g['off'][i]), int(g['off'][i + 1])
    assert ref.shape[0] == g['kshape'][i] and ref.nnz == e - o
    assert np.array_equal(g['row'][o:e].cpu().numpy(), ref.row) and np.array_equal(g['val'][o:e].cpu().numpy(), ref.data)
ms = int(g['kshape'].max())
a = hip_lib.HipSolve.from_device_kernels(g, 0, [ms, ms], mode='fast')
a.set_kernels_device(g, 1, nd - 1)
a.run_chain(0, nd - 1, renorm=False, scale=130000.0)
b = hip_lib.HipSolve.from_model(model, 0, [ms, ms], mode='fast', chain_only=True)
b.set_kernels_from_model(model, 1, nd - 1)
b.run_chain(0, nd - 1, renorm=False, scale=130000.0)
sa, sb = a.chain_stats(0, nd - 1), b.chain_stats(0, nd - 1)
for d in range(nd - 1):
    assert (sa[d].nnz, sa[d].sum, sa[d].flag) == (sb[d].nnz, sb[d].sum, sb[d].flag)
    assert np.array_equal(a.dense(L.REC_CHAIN, d), b.dense(L.REC_CHAIN, d))
a.close(); b.close(); model.close()
print('DEVICE_EXCHANGE_OK', int(g['off'][-1]))
'''


def test_device_resident_kernel_exchange_feeds_the_chain_bitwise():
    '''`parallel.prob_mass_sharded_device` keeps the day kernels on the device from the
    builder to the chain (ps_model_export_device -> RCCL all_gather -> ps_chain_set_kernels_device /
    ps_solver_set_state_device).  World size 1 here (the exchange across ranks is the gloo test's part):
    the chain fed that way equals the chain fed by the model hand-over, bit for bit, and the exported
    triplets are the ones `fetch` returns.  Runs in a process of its own that imports torch FIRST, the way
    a torch.distributed driver does: torch's wheel brings its own HIP runtime, and a process that has
    already initialised the system one through libparasitoid_hip.so finds no device through torch's.'''
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, '-c', _DEVICE_EXCHANGE, root], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert p.returncode == 0 and 'DEVICE_EXCHANGE_OK' in p.stdout, p.stderr[-3000:]
