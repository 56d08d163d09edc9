"""Sub-records of the bench.py JSON line that go beyond the synthetic headline stack:

  real_wind_record   BASELINE config 3 on real data: Carnarvon wind,
                     R = 2048 (N = 4097), 30 days, probability model, day kernels built by the
                     device `prob_mass`; device-only chain rate in 'fast' and 'auto' mode, with
                     the number of flagged days and what route the kernels took.
  bayes_record       the second BASELINE metric: MCMC samples/hour on the Kalbar data set
                     (in-repo AdaptiveMetropolis around the device-resident pop_model body).

Measurement code, not product code: lives next to bench.py and may time the CPU oracle as a
reported baseline.
"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# reference defaults, Run.py:68-83
HP = (1., 1.263, 3.913, 7.302, 2.614, 23.999, 2.350)
DP = (171.82, 144.58, 0.253)
DLP = (7.096, 7.260, 0.000)
MU_R = 1.179
NPER = 30


def _chain_rate(pm, nd, reps):
    """device-only chain rate of the last evaluated model: state + kernels are already on the
    device; time set_state + transform + nd-1 day steps + statistics, `reps` times"""
    from parasitoids_amd import _lib as L
    s = pm.solver
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.set_state_from_model(pm.model, 0)
        s.run_chain(0, nd - 1, negval=1e-8, scale=1.0, renorm=True)
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    stats = s.chain_stats(0, nd - 1)
    return dt, stats


def real_wind_case(rad_dist, R=2048, nd=30, mode='fast', device=None, reps=3, prof=True):
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        pm = PopModel(wd, days, domain_info=(float(rad_dist), R), mode=mode, device=device,
                      prob_model=True)
        t0 = time.perf_counter()
        pm.evaluate(HP, DP, DLP, MU_R, NPER, ndays=nd)       # builds kernels, runs the chain once
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        pm.evaluate(HP, DP, DLP, MU_R, NPER, ndays=nd)
        t_eval = time.perf_counter() - t0
    s = pm.solver
    ks = pm.model.last['kshape']
    if prof:
        s.prof_enable(True, every=1)
    dt, stats = _chain_rate(pm, nd, reps)
    kern = {}
    if prof:
        for k, (ms, cnt) in s.prof_read().items():
            if cnt:
                kern[k] = {'launches_per_chain': round(cnt / reps, 1), 'avg_ms': round(ms / cnt, 4),
                           'ms_per_chain': round(ms / reps, 3)}
        s.prof_enable(False)
    flags = [bool(x.flag) for x in stats]
    last = stats[-1]
    rec = {'rad_dist': rad_dist, 'mode': s.mode, 'fft_len': s.fft_len,
           'P_reference': int(2 * R + 1 + int(ks.max()) // 2),
           'kshape_min': int(ks.min()), 'kshape_max': int(ks.max()),
           'grid_days_per_s': round((nd - 1) / dt, 2), 'chain_ms': round(dt * 1e3, 3),
           'flagged_days': int(sum(flags)), 'days': nd - 1,
           'kernels_direct': bool(s.kernels_direct),
           'multi_day_launches': {k: v['launches_per_chain'] for k, v in kern.items() if k.startswith('col_inv_a_x')},
           'end_to_end_eval_s': round(t_eval, 4), 'first_eval_s': round(t_first, 3),
           'last_day_mass': round(last.sum + last.delta * last.nnz, 12), 'last_day_nnz': int(last.nnz),
           'kernels': kern}
    pm.close()
    return rec


def real_wind_record(device=None, R=2048, nd=30):
    """C3a (rad_dist 10 km: the wind carries mass to the pad, flags fire) and C3b (40 km: no
    flags) in fast and auto mode.  value = simulated day steps / device time of the chain
    (kernel COO already on the device, as in the headline metric)."""
    out = {'workload': 'Carnarvon wind (data/carnarvonearlwind.txt), R=%d (N=%d), %d days, probability '
                       'model, default parameters (Run.py:68-83); day kernels from the device prob_mass'
                       % (R, 2 * R + 1, nd), 'unit': 'grid-days/s'}
    for name, rd in (('c3a', 10000.0), ('c3b', 40000.0)):
        for mode in ('fast', 'auto'):
            try:
                out['%s_%s' % (name, mode)] = real_wind_case(rd, R, nd, mode, device)
            except Exception as e:
                out['%s_%s' % (name, mode)] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def bayes_record(device=None):
    return {'error': 'not wired yet'}


if __name__ == '__main__':
    import json
    which = sys.argv[1] if len(sys.argv) > 1 else 'real_wind'
    if which == 'real_wind':
        R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
        print(json.dumps(real_wind_record(R=R)))
    else:
        print(json.dumps(bayes_record()))
