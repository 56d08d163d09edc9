#!/usr/bin/env python3
"""Measured deviations of the device path from the reference's golden vectors and from the
oracle (what the -m gpu tests assert, as numbers).  Run on the GPU box:
    python tests/parity_report.py > profiles/rNN_parity_report.txt
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')

from helpers import HP, DP, DLP, MU_R, NPER, HP_T, DP_T, coo_from, recentre   # noqa: E402
from oracle import calcsol as OC, model as OM                                  # noqa: E402
from parasitoids_amd import ParasitoidModel as PM, hip_lib, synthetic          # noqa: E402

G = lambda n: np.load(os.path.join(ROOT, 'tests', 'golden', n + '.npz'))
DATA = os.path.join(ROOT, 'tests', 'golden', 'data')


def line(what, val, tol):
    print('%-78s %10.3e   (test tolerance %.0e)' % (what, val, tol))


def main():
    print('# parity report: max absolute deviation, device (C ABI) vs reference golden / oracle')
    g3, g2, g5, g6, g7 = G('g3_hprob_wind'), G('g2_stamps'), G('g5_prob_mass'), G('g6_solutions'), G('g7_populations')
    wd, days = PM.get_wind_data(os.path.join(DATA, 'kalbar'), 30, '00:00')
    model = PM.WindModel(wd)
    h = np.array([model.h_flight_prob(d, *HP) for d in days[:8]])
    line('h_flight_prob, 8 Kalbar days (relative, vs reference G3)',
         np.abs(h / g3['kalbar_h_def'] - 1).max(), 1e-13)
    model.close()
    e = 0.0
    for k in range(int(g2['n'])):
        c = g2['c%d' % k]
        e = max(e, np.abs(PM.get_mvn_cdf_values(c[0], c[1:3], PM.Dmat(*c[3:6])) - g2['m%d' % k]).max())
    line('get_mvn_cdf_values, 19 stamps incl. rho = 0.95 / -0.8 (vs reference G2)', e, 5e-15)
    res = PM.prob_mass_batch(days[:6], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 128)
    e, same = 0.0, True
    for d, r in zip(days[:6], res):
        n = 'kal128_d%d' % d
        same &= np.array_equal(g5[n + '_row'], r.row) and np.array_equal(g5[n + '_col'], r.col)
        e = max(e, np.abs(r.data - g5[n + '_val']).max())
    line('prob_mass Kalbar R=128, 6 days: values (COO pattern identical: %s)' % same, e, 5e-15)
    res = PM.prob_mass_batch(days[:2], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 400)
    e = max(np.abs(r.data - g5['kal400_d%d_val' % d]).max() for d, r in zip(days[:2], res))
    line('prob_mass Kalbar R=400, 2 days: values', e, 5e-15)
    for R, mode in ((128, 'exact'), (200, 'exact'), (200, 'fold'), (200, 'fast')):
        tag = 'r%d' % R
        nd = int(g6[tag + '_ndays'])
        pmfs = [coo_from(g6, '%s_pmf%d' % (tag, i)) for i in range(nd)]
        ms = g6[tag + '_max_shape']
        first = recentre(pmfs[0], R)
        trace = {}
        ref = [first]
        OC.get_solutions(ref, pmfs, list(range(nd)), nd, 2 * R + 1, ms, trace=trace)
        s = hip_lib.HipSolve(first, ms, mode=mode)
        s.set_kernels(pmfs[1:])
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd - 1)
        e = max(np.abs(s.dense(0, n) - trace['raw'][n]).max() for n in range(nd - 1))
        flags_ok = [bool(x.flag) for x in st] == [bool(f) for f in g6[tag + '_flags']]
        line('get_solutions R=%d %d days, %s (P=%d, FFT %d): raw states; flags identical: %s'
             % (R, nd, mode, s.pad_shape[0], s.fft_len, flags_ok), e, 5e-8 if mode == 'fast' else 1e-12)
        if mode != 'fast':
            e = max(abs(s.chain_solution(n, st[n]).tocsr() - ref[n + 1].tocsr()).max() for n in range(nd - 1))
            line('   ... thresholded + renormalised solutions vs reference', e, 1e-12)
        s.close()
    from parasitoids_amd.pop_model import PopModel
    pm = PopModel(wd, days, domain_info=(10000.0, 400), r_number=130000)
    stats = pm.evaluate(HP, DP, DLP, MU_R, NPER)
    e = max(abs(stats[d][1] - float(g7['r400_sum%d_sum' % d])) / 130000 for d in range(18))
    line('get_populations Kalbar R=400, 18 days (P=1121=19*59): daily totals, relative', e, 1e-6)
    pos = g7['r400_pos']
    e = 0.0
    for d in range(18):
        got = np.asarray(pm.population(d)[pos[:, 0], pos[:, 1]]).ravel()
        e = max(e, np.abs(got - g7['r400_sum%d_samp' % d]).max())
    line('   ... 4000 sampled cells per day, absolute (values up to 1.3e5)', e, 1e-7)
    pm.close()
    # full size: fast vs exact torus
    state, kernels, _ = synthetic.make_stack(R=2048, K=2049, ndays=2, seed=20240613)
    f = {}
    for mode in ('fast', 'exact'):
        s = hip_lib.HipSolve(state, [2049, 2049], mode=mode)
        s.set_kernels(kernels); s.run_chain(renorm=True); s.chain_stats(0, 2)
        f[mode] = s.dense(0, 1)
        s.close()
    line('N=4097 headline stack: fast (FFT 5184) vs exact reference torus (5121 = 9*569)',
         np.abs(f['fast'] - f['exact']).max(), 1e-13)


if __name__ == '__main__':
    main()
