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
           'auto_first_fold_day': s.auto_info()[0] if s.mode == 'auto' else None,
           'auto_fold_fft': s.auto_info()[1] if s.mode == 'auto' else None,
           'auto_route_days': ({k: int((s.auto_route(0, nd - 1) == v).sum()) for k, v in
                                (('front', 0), ('wide', 1), ('fold', 2))} if s.mode == 'auto' else None),
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


def _oracle_day(args):
    """worker of the all-cores CPU baseline: the oracle's prob_mass for one day (the reference
    maps exactly this over `Pool()`, Run.py:422-425 / Bayes_Run.py:298-306)"""
    day, rad_res = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import model as OM
    from parasitoids_amd import ParasitoidModel as PM
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    t0 = time.perf_counter()
    p = OM.prob_mass(day, wd, HP, DP, DLP, MU_R, NPER, 10000.0, rad_res)
    return time.perf_counter() - t0, p.shape[0], p.nnz


def bayes_cpu_baseline(rad_res, days, all_cores=True):
    """The oracle's model evaluation (18 x prob_mass + 17 chain days) on the host: prob_mass of
    ONE day and 2 chain days on one core, extrapolated (`cores: 1`); and, like the reference
    runs it, the 18 prob_mass calls spread over a process pool (`all_cores`)."""
    from oracle import calcsol as OC
    from oracle import model as OM
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.Run import recentre
    wd, _ = PM.get_wind_data('data/kalbar', 30, '00:00')
    t0 = time.perf_counter()
    p0 = OM.prob_mass(days[0], wd, HP, DP, DLP, MU_R, NPER, 10000.0, rad_res)
    t_pm = time.perf_counter() - t0
    N = 2 * rad_res + 1
    ms = np.array(p0.shape)
    hat = OC.fft2(recentre(p0, rad_res), ms)
    t0 = time.perf_counter()
    for _ in range(2):
        OC.fftconv2(hat, p0.tocsr())
        A, flag = OC.ifft2(hat, [N, N])
        OC.r_small_vals(A * 130000.0)
        if flag:
            hat = OC.fft2(A, ms)
    t_day = (time.perf_counter() - t0) / 2
    nd = len(days)
    rec = {'value': round(3600.0 / (nd * t_pm + (nd - 1) * t_day), 2), 'unit': 'samples/hour', 'cores': 1,
           'kind': 'port',
           'sample': 'oracle prob_mass of 1 day (%.1fs) and 2 chain days (%.2fs each) at R=%d, extrapolated to '
                     '%d + %d; one model evaluation per MCMC sample' % (t_pm, t_day, rad_res, nd, nd - 1)}
    if all_cores:
        import multiprocessing as mp
        ncpu = os.cpu_count() or 1
        nproc = min(nd, ncpu)
        ctx = mp.get_context('spawn')          # this process holds the GPU: never fork it
        t0 = time.perf_counter()
        with ctx.Pool(nproc) as pool:
            res = pool.map(_oracle_day, [(d, rad_res) for d in days])
        t_pool = time.perf_counter() - t0
        rec['all_cores'] = {'value': round(3600.0 / (t_pool + (nd - 1) * t_day), 2), 'unit': 'samples/hour',
                            'cores': nproc, 'host_cpus': ncpu,
                            'sample': 'the %d prob_mass days of one evaluation over a %d-process pool (%.1fs wall '
                                      'incl. start-up, slowest day %.1fs) + %d chain days on one core (%.2fs each)'
                                      % (nd, nproc, t_pool, max(r[0] for r in res), nd - 1, t_day)}
    return rec


def bayes_case(rad_res, mode, samples, burn, device=None, seed=1000):
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd import mcmc
    from parasitoids_amd.Data_Import import LocInfo
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        pm = PopModel(wd, days, domain_info=(10000.0, rad_res), r_number=130000, mode=mode, device=device)
        li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, rad_res))      # Run.py:129
        chain = mcmc.Sampler(pm, li, (10000.0 / rad_res) ** 2, seed=seed)
        chain.run(burn)
        res = chain.run(samples)
    rec = {'value': round(res['samples_per_hour'], 1), 'unit': 'samples/hour', 'rad_res': rad_res,
           'grid': '%d^2' % (2 * rad_res + 1), 'mode': pm.solver.mode, 'fft_len': pm.solver.fft_len,
           'samples': samples, 'burn': burn, 'ms_per_sample': round(1e3 * res['seconds'] / samples, 3),
           'acceptance': round(res['acceptance'], 3), 'evaluations': res['evaluations_this_run'],
           'failed_evaluations': res['failed_evaluations'],
           'logp_first_last': [round(float(res['logp'][0]), 3), round(float(res['logp'][-1]), 3)]}
    pm.close()
    return rec, days


def bayes_record(device=None, samples=150, burn=20, cpu=True):
    """BASELINE.json's second metric: Bayes_Run MCMC samples/hour on the Kalbar data -- one
    chain on one GPU with the in-repo sampler (parasitoids_amd/mcmc.py: AdaptiveMetropolis block
    + scalar Metropolis steps, the reference's priors and Poisson observation model, Kalbar wind
    and the Kalbar field observations through Data_Import.LocInfo).  One sample = one
    pop_model evaluation (18 x prob_mass + get_populations + observation gathers) + the scalar
    updates.  R = 400 is the reference's hard-coded grid (Bayes_Run.py:91), R = 512 the
    1024^2 grid BASELINE config 4 names."""
    out = {'metric': 'Bayes_Run MCMC samples/hour (Kalbar)', 'unit': 'samples/hour', 'chains': 1,
           'sampler': 'AdaptiveMetropolis(15 model parameters; scales, interval=500, delay=1000, '
                      'shrink_if_necessary) + scalar Metropolis on xi, em_obs_prob, grid_obs_prob, '
                      'A_collected, sent_obs_probs_* (Bayes_Run.py:102-196, :486-487)'}
    days = None
    for R in (400, 512):
        for mode in ('auto', 'fast'):
            key = 'r%d_%s' % (R, mode)
            try:
                out[key], days = bayes_case(R, mode, samples, burn, device)
            except Exception as e:
                out[key] = {'error': '%s: %s' % (type(e).__name__, e)}
    if 'value' in out.get('r400_auto', {}):
        out['value'] = out['r400_auto']['value']            # exact-torus results at the reference's grid
    if cpu and days is not None:
        for R in (400, 512):
            try:
                out['cpu_baseline_r%d' % R] = bayes_cpu_baseline(R, days, all_cores=True)
            except Exception as e:
                out['cpu_baseline_r%d' % R] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


if __name__ == '__main__':
    import json
    which = sys.argv[1] if len(sys.argv) > 1 else 'real_wind'
    if which == 'real_wind':
        R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
        print(json.dumps(real_wind_record(R=R)))
    else:
        print(json.dumps(bayes_record()))
