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
    dt, stats = _chain_rate(pm, nd, reps)      # the rate: no per-launch events in the stream
    kern, helpers = {}, {}
    if prof:                                   # the per-class table: the same chain again, with them
        s.prof_enable(True, every=1)
        _chain_rate(pm, nd, reps)
        for k, (ms, cnt) in s.prof_read().items():
            if cnt:
                kern[k] = {'launches_per_chain': round(cnt / reps, 1), 'avg_ms': round(ms / cnt, 4),
                           'ms_per_chain': round(ms / reps, 3)}
        if s.mode == 'auto':                   # the helpers' launches, per owner (most of an auto chain)
            for o in s.PROF_OWNERS[1:]:
                t = {k: {'launches_per_chain': round(v['launches'] / reps, 1), 'avg_ms': round(v['ms'] / v['timed'], 4),
                         'ms_per_chain': round(v['ms'] / reps, 3)} for k, v in s.prof_owner(o).items() if v['timed']}
                if t:
                    helpers[o] = {'fft_len': s.helper_fft_len(o), 'ms_per_chain': round(sum(v['ms_per_chain'] for v in t.values()), 3),
                                  'kernels': t}
        s.prof_enable(False)
    flags = [bool(x.flag) for x in stats]
    last = stats[-1]
    rec = {'rad_dist': rad_dist, 'mode': s.mode, 'fft_len': s.fft_len,
           'P_reference': int(2 * R + 1 + int(ks.max()) // 2),
           'kshape_min': int(ks.min()), 'kshape_max': int(ks.max()),
           'grid_days_per_s': round((nd - 1) / dt, 2), 'chain_ms': round(dt * 1e3, 3),
           'flagged_days': int(sum(flags)), 'days': nd - 1, 'flag_pattern': ''.join('1' if f else '0' for f in flags),
           'kernels_direct': bool(s.kernels_direct),
           'auto_first_fold_day': s.auto_info()[0] if s.mode == 'auto' else None,
           'auto_fold_fft': s.auto_info()[1] if s.mode == 'auto' else None,
           'auto_route_days': ({k: int((s.auto_route(0, nd - 1) == v).sum()) for k, v in
                                (('front', 0), ('wide', 1), ('fold', 2), ('narrow', 3))} if s.mode == 'auto' else None),
           'multi_day_launches': {k: v['launches_per_chain'] for k, v in kern.items() if k.startswith('col_inv_a_x')},
           'end_to_end_eval_s': round(t_eval, 4), 'first_eval_s': round(t_first, 3),
           'last_day_mass': round(last.sum + last.delta * last.nnz, 12), 'last_day_nnz': int(last.nnz),
           'kernels': kern}
    if helpers:
        rec['helper_kernels'] = helpers
        rec['kernel_ms_accounted'] = round(sum(v['ms_per_chain'] for v in kern.values())
                                           + sum(h['ms_per_chain'] for h in helpers.values()), 3)
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


def bayes_cpu_baseline(rad_res, days, all_cores=True, eval_fraction=None):
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
    ev1 = 3600.0 / (nd * t_pm + (nd - 1) * t_day)
    frac = eval_fraction if eval_fraction else 1.0
    rec = {'value': round(ev1, 2), 'unit': 'evaluations/hour', 'cores': 1,
           'kind': 'port',
           'samples_per_hour_at_the_gpu_chain_evaluation_fraction': round(ev1 / frac, 2),
           'evaluation_fraction': eval_fraction,
           'sample': 'oracle prob_mass of 1 day (%.1fs) and 2 chain days (%.2fs each) at R=%d, extrapolated to '
                     '%d + %d = one model evaluation; samples/hour = evaluations/hour / the fraction of the GPU '
                     'chain\'s samples that evaluated the model' % (t_pm, t_day, rad_res, nd, nd - 1)}
    if all_cores:
        import multiprocessing as mp
        ncpu = os.cpu_count() or 1
        nproc = min(nd, ncpu)
        ctx = mp.get_context('spawn')          # this process holds the GPU: never fork it
        t0 = time.perf_counter()
        with ctx.Pool(nproc) as pool:
            res = pool.map(_oracle_day, [(d, rad_res) for d in days])
        t_pool = time.perf_counter() - t0
        evp = 3600.0 / (t_pool + (nd - 1) * t_day)
        rec['all_cores'] = {'value': round(evp, 2), 'unit': 'evaluations/hour',
                            'samples_per_hour_at_the_gpu_chain_evaluation_fraction': round(evp / frac, 2),
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
    ev = res['evaluations_this_run']
    rec = {'value': round(res['samples_per_hour'], 1), 'unit': 'samples/hour', 'rad_res': rad_res,
           'grid': '%d^2' % (2 * rad_res + 1), 'mode': pm.solver.mode, 'fft_len': pm.solver.fft_len,
           'samples': samples, 'burn': burn, 'ms_per_sample': round(1e3 * res['seconds'] / samples, 3),
           'timed_window_s': round(res['seconds'], 3),
           'acceptance': round(res['acceptance'], 3), 'evaluations': ev,
           # a sample whose block proposal leaves the priors' support costs no model evaluation
           # (mcmc.Sampler.step; PyMC2 skips the deterministic for a -inf prior as well)
           'evaluations_per_hour': round(3600.0 * ev / res['seconds'], 1),
           'evaluation_fraction': round(ev / float(samples), 4),
           'failed_evaluations': res['failed_evaluations'],
           'logp_first_last': [round(float(res['logp'][0]), 3), round(float(res['logp'][-1]), 3)]}
    import hashlib
    rec['trace_sha256'] = hashlib.sha256(np.ascontiguousarray(res['trace']).tobytes()).hexdigest()
    pm.close()
    return rec, days


def bayes_multi_case(rad_res, mode, k, samples, burn, device=None, seed0=1000, single_sha=None):
    """k independent chains in ONE process on one GPU (mcmc.run_parallel: a host thread, a PopModel and
    its streams per chain; seeds seed0 .. seed0 + k - 1).  Aggregate samples and evaluations per hour over
    the wall time of the timed window; `chain0_sha256` is the digest of chain 0's trace, to be compared
    with the single-chain run of the same seed (`bitwise_equal_to_single_chain`)."""
    import hashlib
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd import mcmc
    from parasitoids_amd.Data_Import import LocInfo
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, rad_res))
    pms, chains = [], []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        for c in range(k):
            pm = PopModel(wd, days, domain_info=(10000.0, rad_res), r_number=130000, mode=mode, device=device)
            pms.append(pm)
            chains.append(mcmc.Sampler(pm, li, (10000.0 / rad_res) ** 2, seed=seed0 + c))
        mcmc.run_parallel(chains, burn)
        res, dt = mcmc.run_parallel(chains, samples)
    ev = sum(r['evaluations_this_run'] for r in res)
    sha = hashlib.sha256(np.ascontiguousarray(res[0]['trace']).tobytes()).hexdigest()
    rec = {'chains_per_gpu': k, 'value': round(3600.0 * k * samples / dt, 1), 'unit': 'samples/hour (sum over the chains)',
           'evaluations_per_hour': round(3600.0 * ev / dt, 1), 'timed_window_s': round(dt, 3),
           'ms_per_sample_aggregate': round(1e3 * dt / (k * samples), 3),
           'evaluation_fraction': round(ev / float(k * samples), 4),
           'per_chain_samples_per_hour': [round(r['samples_per_hour'], 1) for r in res],
           'per_chain_acceptance': [round(r['acceptance'], 3) for r in res],
           'seeds': [seed0 + c for c in range(k)], 'samples_per_chain': samples, 'burn': burn,
           'mode': pms[0].solver.mode, 'fft_len': pms[0].solver.fft_len, 'chain0_sha256': sha}
    if single_sha is not None:
        rec['bitwise_equal_to_single_chain'] = bool(sha == single_sha)
    for pm in pms:
        pm.close()
    return rec


def bayes_record(device=None, samples=600, burn=50, cpu=True):
    """BASELINE.json's second metric: Bayes_Run MCMC samples/hour on the Kalbar data -- one
    chain on one GPU with the in-repo sampler (parasitoids_amd/mcmc.py: AdaptiveMetropolis block
    + scalar Metropolis steps, the reference's priors and Poisson observation model, Kalbar wind
    and the Kalbar field observations through Data_Import.LocInfo).  One sample = one block
    proposal -- a pop_model evaluation (18 x prob_mass + get_populations + observation gathers)
    UNLESS the proposal leaves the priors' support, which costs none -- + the scalar updates;
    `evaluations_per_hour` / `evaluation_fraction` say how many of the timed samples evaluated the
    model, and the CPU baselines are stated in evaluations/hour with the samples/hour they would
    give at that same fraction.  R = 400 is the reference's hard-coded grid (Bayes_Run.py:91),
    R = 512 the 1024^2 grid BASELINE config 4 names.  The timed window is `samples` samples after
    `burn` (AdaptiveMetropolis: delay = 1000, so the window samples with the initial proposal
    scales, like the first 1000 iterations of the reference's run)."""
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
    # the same chain next to k - 1 others on the one GPU (seeds 1000 .. 1000 + k - 1; chain 0 = the run above)
    for k in (2, 4, 8):
        key = 'r400_auto_x%d' % k
        try:
            out[key] = bayes_multi_case(400, 'auto', k, samples, burn, device,
                                        single_sha=out.get('r400_auto', {}).get('trace_sha256'))
        except Exception as e:
            out[key] = {'error': '%s: %s' % (type(e).__name__, e)}
    best = max((out[k] for k in ('r400_auto', 'r400_auto_x2', 'r400_auto_x4', 'r400_auto_x8') if 'evaluations_per_hour' in out.get(k, {})),
               key=lambda r: r['evaluations_per_hour'], default=None)
    if best is not None:
        out['per_gpu'] = {'evaluations_per_hour': best['evaluations_per_hour'], 'samples_per_hour': best['value'],
                          'chains_per_gpu': best.get('chains_per_gpu', 1),
                          'note': 'most evaluations/hour one GPU delivers: independent chains side by side in one '
                                  'process, each bit-identical to running alone'}
    if 'value' in out.get('r400_auto', {}):
        out['value'] = out['r400_auto']['value']            # exact-torus results at the reference's grid
    if 'evaluations_per_hour' in out.get('r400_auto', {}):
        out['evaluations_per_hour'] = out['r400_auto']['evaluations_per_hour']
    if cpu and days is not None:
        for R in (400, 512):
            try:
                frac = out.get('r%d_auto' % R, {}).get('evaluation_fraction')
                out['cpu_baseline_r%d' % R] = bayes_cpu_baseline(R, days, all_cores=True, eval_fraction=frac)
            except Exception as e:
                out['cpu_baseline_r%d' % R] = {'error': '%s: %s' % (type(e).__name__, e)}
    try:
        out['prob_mass_roofline'] = prob_mass_roofline(device=device)
    except Exception as e:
        out['prob_mass_roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


# estimated fp64 operations per rectangle probability of the device pipeline (model_kernels.h):
# one bivariate-normal corner value -- Genz BVU, |rho| < 0.3: 3 Gauss-Legendre points = 6 exponentials
# of ~30 operations each (argument, exp, weighted accumulate) + 7 around them -- per cell (the
# (2H+2)^2 corner grid of a (2H+1)^2 stamp), + 5 for the cell mass (three differences, x hprob, the
# ordered add).  The Phi (erfc) values are per row / column edge of the corner grid, not per corner.
EXP_PER_CORNER = 6
FLOPS_PER_RECT = 6 * 30 + 7 + 5
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md


def prob_mass_roofline(device=None, rad_res=400, reps=10):
    """SURVEY 8d's roofline for the prob_mass half of a Bayes evaluation (compute-bound: fp64 VALU +
    transcendentals, not HBM): the 18 Kalbar days at the reference's grid, default parameters --
    rectangle probabilities per second (one per cell of every period's stamp, what the reference
    spends 97 % of prob_mass on: ParasitoidModel.py:311-380), estimated fp64 rate against the
    vector peak, and the HBM bytes of the same batch from the committed rocprofv3 PMC summary."""
    import glob
    import json
    import re
    from parasitoids_amd import ParasitoidModel as PM
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    model = PM.WindModel(wd, device)
    args = (days, HP, DP, DLP, MU_R, NPER, 10000.0, rad_res)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        model.build(*args)
        t0 = time.perf_counter()
        for _ in range(reps):
            model.build(*args)                       # returns after the batch's statistics are on the host
        dt = (time.perf_counter() - t0) / reps
    rect = corners = periods = 0
    for i in range(len(days)):
        dbg = model.debug(i)
        live = dbg['hprob'] > 0
        H = dbg['H'][live].astype(np.int64)
        rect += int(((2 * H + 1) ** 2).sum())
        corners += int(((2 * H + 2) ** 2).sum())
        periods += int(live.sum())
    model.close()
    tf = rect * FLOPS_PER_RECT / dt / 1e12
    rec = {'workload': 'prob_mass of the %d Kalbar days, R = %d, default parameters (one Bayes evaluation\'s kernels)'
                       % (len(days), rad_res),
           'ms_per_batch': round(dt * 1e3, 3), 'periods': periods, 'rect_probs': rect, 'corner_values': corners,
           'rect_probs_per_s': round(rect / dt, 1), 'bound': 'fp64 VALU + transcendentals',
           'exp_per_corner': EXP_PER_CORNER, 'est_flops_per_rect_prob': FLOPS_PER_RECT,
           'achieved': round(tf, 2), 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s (estimated)',
           'frac': round(tf / FP64_VALU_PEAK_TFLOPS, 4),
           'note': 'wall time of the whole batch (periods, pair lists, corner values, ordered accumulate, '
                   'threshold, compaction, statistics to the host); the corner values (k_pair_masses) are '
                   'about half of it, see profiles/'}
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*prob_mass_hbm_traffic.json')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])
    if files:
        try:
            pm = json.load(open(files[-1]))
            rec['hbm'] = {'source': 'profiles/%s (committed rocprofv3 --pmc summary of the same batch)' % os.path.basename(files[-1])}
            rec['hbm'].update(pm.get('summary', {}))
            tot = pm.get('summary', {}).get('total_bytes_per_batch')
            if tot:
                rec['hbm']['GBps_at_measured_time'] = round(tot / dt / 1e9, 1)
                rec['hbm']['frac_of_8000'] = round(tot / dt / 1e9 / 8000.0, 4)
        except Exception as e:       # a malformed summary must not cost the record
            rec['hbm'] = {'error': str(e)}
    return rec


# --------------------------------------------------------------------------- multi-day release (r_dur = 5)
def release_case(R, mode, rad_dist=10000.0, nd=30, reps=3):
    """`Run.py --carnarvon --pop` -- the reference's default Carnarvon preset, r_dur = 5 (Run.py:118) -- on the
    chain API (ps_chain_run_release): end to end through Run.run_model (kernels from the device prob_mass,
    CSR export of every day included), and the device part alone: the 25 days after the release, each one
    cohort step + 4 back-solves + the weighted population, enqueued as ONE run."""
    from parasitoids_amd import CalcSol, globalvars, hip_lib, Run
    from parasitoids_amd import ParasitoidModel as PM
    p = Run.Params(config=None)
    p.cmd_line_chg(['--carnarvon', '--pop', 'domain_info=(%r,%d)' % (rad_dist, R), 'ndays=%d' % nd])
    old = globalvars.fft_mode
    globalvars.fft_mode = mode
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            t0 = time.perf_counter()
            Run.run_model(p, verbose=False)          # first call of the process: device allocations, first touch
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            modelsol, days, ndays, tm = Run.run_model(p, verbose=False)
            t_all = time.perf_counter() - t0
            route = CalcSol.last_release_route
            wind_data, days2 = PM.get_wind_data(*p.get_wind_params())
            starts = [p.r_start] + [None] * (ndays - 1)
            pmf_list = PM.prob_mass_batch(days2[:ndays], wind_data, *p.get_model_params(), start_times=starts)
    finally:
        globalvars.fft_mode = old
    N = 2 * R + 1
    ms = np.max([q.shape for q in pmf_list], axis=0)
    r_spread = [Run.recentre(pmf_list[d], R).tocsr() for d in range(p.r_dur)]
    rec = {'rad_res': R, 'grid': '%d^2' % N, 'rad_dist': rad_dist, 'r_dur': p.r_dur, 'days': ndays, 'mode_asked': mode,
           'route': route, 'first_call_s': round(t_first, 3), 'end_to_end_s': round(t_all, 3), 'prob_mass_s': round(tm['prob_mass_s'], 3),
           'get_populations_s': round(tm['solver_s'], 3), 'last_day_total': round(float(modelsol[-1].sum()), 3)}
    s = hip_lib.HipSolve(r_spread[-1], ms, mode='fast' if mode == 'fast' else mode, chain_only=True)
    try:
        if s.mode != 'fold' and s.set_release(pmf_list[p.r_dur:ndays], r_spread[:-1]):
            nk = ndays - p.r_dur
            w = [p.r_mthd()(d + 1) * p.r_number for d in range(p.r_dur)]
            ok = True
            for i in range(reps + 1):
                if i == 1:
                    s.sync()
                    t0 = time.perf_counter()
                s.set_state(r_spread[-1])
                ok = s.run_release(0, nk, w) and ok
            s.sync()
            dt = (time.perf_counter() - t0) / reps
            rec['device_chain'] = {'ms': round(dt * 1e3, 3), 'days': nk, 'grid_days_per_s': round(nk / dt, 1),
                                   'conv_steps_per_s': round(nk * p.r_dur / dt, 1), 'fft_len': s.fft_len,
                                   'mode': s.mode, 'certified_exact': bool(ok) if s.mode == 'auto' else None}
    finally:
        s.close()
    return rec


def release_record():
    out = {'workload': 'Run.py --carnarvon --pop (r_dur = 5, 40 000 wasps, uniform emergence), 30 days; one grid-day = '
                       'one cohort step + 4 back-solves + the weighted population (CalcSol.py:308-323)'}
    for R, mode in ((512, 'fast'), (512, 'auto'), (2048, 'fast')):
        try:
            out['r%d_%s' % (R, mode)] = release_case(R, mode)
        except Exception as e:
            out['r%d_%s' % (R, mode)] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def prefix_split_record(G=8, R=2048, K=2049, nd=30, device=None):
    """SURVEY 8e row 2 on the headline stack, the G "ranks" as G solvers of this one GPU
    (parallel.chain_prefix_split_local): what every rank computes -- its block's kernel transforms and running
    products, then its days' records from state x earlier totals x own products -- timed rank by rank, next to
    the sequential chain of one solver.  The exchange itself (an all-gather of one spectrum per rank) cannot be
    measured on one card: its volume is reported."""
    from parasitoids_amd import hip_lib, parallel, synthetic
    state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=20240613)
    seq = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True, device=device)
    seq.set_kernels(kernels)
    for _ in range(2):
        seq.set_state(state)
        seq.run_chain(renorm=True)
    seq.sync()
    t0 = time.perf_counter()
    seq.set_state(state)
    seq.run_chain(renorm=True)
    seq.sync()
    t_seq = time.perf_counter() - t0
    blocks = parallel.split_days(nd, G)
    solvers = [hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True, device=device) for _ in range(G)]
    for s in solvers:
        s.set_kernels(kernels)
    tp, tf, totals = [0.0] * G, [0.0] * G, [None] * G
    for rep in range(2):                      # the second round is the one reported (buffers exist)
        for g, (s, (f, c)) in enumerate(zip(solvers, blocks)):
            s.set_state(state)
            s.sync()
            t0 = time.perf_counter()
            totals[g], nbytes = s.block_prefix(f, c)
            tp[g] = time.perf_counter() - t0
        flagged = False
        for g in reversed(range(G)):          # see chain_prefix_split_local
            f, c = blocks[g]
            s = solvers[g]
            t0 = time.perf_counter()
            flagged = s.block_finish(f, c, totals[:g]) or flagged
            s.sync()
            tf[g] = time.perf_counter() - t0
    worst = 0.0
    for s, (f, c) in zip(solvers, blocks):
        for d in (f, f + c - 1):
            worst = max(worst, float(np.abs(s.dense(0, d) - seq.dense(0, d)).max()))
    rec = {'workload': 'headline stack (N = %d, %d days, FFT %d) as %d blocks of days on ONE GPU, one solver per block'
                       % (2 * R + 1, nd, seq.fft_len, G),
           'blocks': blocks, 'flagged': bool(flagged),
           'per_rank_prefix_ms': [round(t * 1e3, 3) for t in tp], 'per_rank_finish_ms': [round(t * 1e3, 3) for t in tf],
           'slowest_rank_ms': round(max(a + b for a, b in zip(tp, tf)) * 1e3, 3),
           'sequential_chain_ms': round(t_seq * 1e3, 3),
           'exchange_bytes_into_each_rank': int((G - 1) * nbytes), 'block_total_bytes': int(nbytes),
           'max_abs_vs_sequential': worst,
           'note': 'per-rank compute only; the all-gather of the block totals (RCCL over xGMI) is not in these times'}
    for s in solvers:
        s.close()
    seq.close()
    return rec


# --------------------------------------------------------------------------- N > 1: configs 4 and 5
def _ensemble_runner(device, rad_res, ndays):
    """member -> result dict on this rank's GPU (BASELINE config 5: probability model, Carnarvon wind)"""
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    model = PopModel(wd, days, domain_info=(10000.0, rad_res), r_start=0.354, mode='fast', prob_model=True,
                     device=device)
    g, f = (1.263, 3.913), (7.302, 2.614, 23.999, 2.350)

    def run(mem):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            st = model.evaluate((mem['lam'],) + g + f, (mem['sig_x'], mem['sig_y'], 0.253), (7.096, 7.260, 0.0),
                                mem['mu_r'], 30, ndays=ndays)
        tot = model.moments(ndays - 1)[0]
        return {'total': float(tot), 'nnz_last': int(st[-1][0])}
    return run, model


def multi_gpu_record(rank, world, device=None, rehearse=False, members_per_rank=4, rad_res=1024, ndays=30,
                     chain_samples=300, chain_burn=30, fail_stage=None):
    """What the N-rank job is for besides replica stacks: BASELINE config 5 -- ensemble
    members round-robin over the ranks (`parallel.run_members`, results gathered on rank 0) -- and
    config 4 -- one independent MCMC chain per rank, chain c seeded 1000 + c.  No data-path
    collective; the gathers carry small python objects.  Returns the record on rank 0, None elsewhere.
    `rehearse`: the same control flow with stand-in evaluation functions (CPU test, gloo)."""
    from parasitoids_amd import parallel
    from parasitoids_amd.synthetic import ensemble_members
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()

    out = {'n_gpus': world}
    # Every rank takes part in EVERY collective below whatever happens to its own work: a
    # stage that fails on one rank turns into an error object that travels through the same gathers,
    # instead of that rank skipping ahead to the final barrier while the others wait in this one.
    errors = []

    def guarded(stage, fn, fallback):
        try:
            return fn()
        except Exception as e:
            errors.append('%s on rank %d: %s: %s' % (stage, rank, type(e).__name__, e))
            return fallback

    # ---- config 5
    members = ensemble_members(members_per_rank * world)
    model = None
    if rehearse:
        run = lambda mem: {'total': 1.0, 'nnz_last': int(1000 * mem['lam'])}
    else:
        run, model = guarded('ensemble setup', lambda: _ensemble_runner(device, rad_res, ndays), (None, None))
        if run is not None:
            guarded('ensemble warm-up', lambda: run(members[rank]), None)   # solvers, plans, buffers
    if fail_stage == 'ensemble' and rank == world - 1:
        run = None                              # test hook: this rank's members fail
    run_safe = lambda mem: guarded('ensemble member', lambda: run(mem), {'error': True})
    barrier()
    t0 = time.perf_counter()
    res = parallel.run_members(members, run_safe)
    dt_own = time.perf_counter() - t0           # this rank's members (rank 0: + the gather)
    own = len(members[rank::world])
    times = parallel.gather_objects((rank, own, dt_own, list(errors)))
    if model is not None:
        guarded('ensemble close', model.close, None)
    if rank == 0 and any(e for _, _, _, e in times):
        out['ensemble'] = {'error': [m for _, _, _, e in times for m in e]}
    elif rank == 0:
        times = [t[:3] for t in times]
        dt = max(t for _, _, t in times)
        out['ensemble'] = {'workload': 'BASELINE config 5: %d members (lambda, sigma, mu_r from the priors) x %d^2 grid x '
                                       '%d Carnarvon days, probability model, member i on rank i mod %d'
                                       % (len(members), 2 * rad_res + 1, ndays, world),
                           'value': round(len(members) / dt, 3), 'unit': 'members/s',
                           'member_grid_days_per_s': round(len(members) * (ndays - 1) / dt, 1),
                           'seconds': round(dt, 3),
                           'per_rank_members_per_s': [round(n / t, 3) for _, n, t in sorted(times)],
                           'members_gathered': len(res),
                           'all_members_conserve_mass': bool(all(abs(r['total'] - 1.0) < 1e-6 or r['total'] < 1.0 + 1e-9 for r in res))}
    # ---- one simulation's kernel construction sharded over the ranks (Run.py:412-425): days
    # round-robin, COO triplets all-gathered device to device (RCCL over xGMI), every rank ends with
    # all kernels on its GPU.  Timed next to one rank building all days itself.
    del errors[:]
    shard = None
    if not rehearse:
        def sharded():
            import torch
            from parasitoids_amd import ParasitoidModel as PM
            wd, days = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
            m = PM.WindModel(wd, device=device)
            prm = (HP, DP, DLP, MU_R, NPER, 10000.0, rad_res)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', RuntimeWarning)
                parallel.prob_mass_sharded_device(m, days[:ndays], prm)        # warm-up (lists, buffers)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g = parallel.prob_mass_sharded_device(m, days[:ndays], prm)
                torch.cuda.synchronize()
                t_sh = time.perf_counter() - t0
                t0 = time.perf_counter()
                m.build(days[:ndays], *prm)
                t_one = time.perf_counter() - t0
            m.close()
            return {'sharded_s': t_sh, 'one_gpu_s': t_one, 'entries': int(g['off'][-1]), 'days': ndays}
        shard = guarded('sharded prob_mass', sharded, None)
    barrier()
    shards = parallel.gather_objects((rank, shard, list(errors)))
    if rank == 0 and not rehearse:
        if any(sh is None for _, sh, _ in shards):
            out['sharded_prob_mass'] = {'error': [m for _, _, e in shards for m in e]}
        else:
            t_sh = max(sh['sharded_s'] for _, sh, _ in shards)
            t_one = max(sh['one_gpu_s'] for _, sh, _ in shards)
            out['sharded_prob_mass'] = {
                'workload': '%d Carnarvon day kernels at %d^2, days round-robin over %d ranks, triplets all-gathered '
                            'device to device' % (ndays, 2 * rad_res + 1, world),
                'seconds': round(t_sh, 4), 'one_gpu_builds_all_days_s': round(t_one, 4),
                'speedup': round(t_one / t_sh, 2), 'coo_entries': shards[0][1]['entries'],
                'bytes_gathered_per_rank': shards[0][1]['entries'] * 16}
    # ---- one flag-free simulation split over the ranks by days: the headline stack, every rank
    # its block of days, ONE all-gather of a spectrum per rank in between -- next to the same chain run sequentially
    # by every rank on its own.  (chain_prefix_split keeps its own collectives safe against a failing rank.)
    del errors[:]
    split = None
    if not rehearse:
        Rs, Ks, nds = 2048, 2049, 30

        def split_setup():
            from parasitoids_amd import hip_lib, synthetic
            state, kernels, _ = synthetic.make_stack(R=Rs, K=Ks, ndays=nds, seed=20240613)
            s = hip_lib.HipSolve(state, [Ks, Ks], mode='fast', chain_only=True, device=device)
            s.set_kernels(kernels)
            for _ in range(2):                       # the sequential chain, warm and hinted
                s.set_state(state)
                s.run_chain(renorm=True)
            s.sync()
            t0 = time.perf_counter()
            s.set_state(state)
            s.run_chain(renorm=True)
            s.sync()
            return s, state, time.perf_counter() - t0
        setup = guarded('prefix split setup', split_setup, None)
        if parallel.all_ok(setup is not None and world <= nds):
            s_split, state_s, t_seq = setup

            def split_run():
                ops = parallel.DeviceBlockOps(s_split)
                ts, blk = [], None
                for rep in range(3):
                    s_split.set_state(state_s)
                    s_split.sync()
                    barrier()
                    t0 = time.perf_counter()
                    blk = parallel.chain_prefix_split(ops, nds)
                    s_split.sync()
                    ts.append(time.perf_counter() - t0)
                return {'split_s': min(ts[1:]), 'sequential_s': t_seq, 'first': blk[0], 'count': blk[1],
                        'flagged': blk[2], 'fft_len': s_split.fft_len}
            split = guarded('prefix split', split_run, None)
        if setup is not None:
            guarded('prefix split close', setup[0].close, None)
    barrier()
    splits = parallel.gather_objects((rank, split, list(errors)))
    if rank == 0 and not rehearse:
        if any(sp is None for _, sp, _ in splits):
            out['prefix_split'] = {'error': [m for _, _, e in splits for m in e] or ['skipped: more ranks than days']}
        else:
            t_sp = max(sp['split_s'] for _, sp, _ in splits)
            t_sq = max(sp['sequential_s'] for _, sp, _ in splits)
            out['prefix_split'] = {
                'workload': 'headline stack (N = %d, %d days, FFT %d): ONE simulation, rank g runs the g-th block of '
                            'days, one all-gather of a %d-byte spectrum per rank in between'
                            % (2 * Rs + 1, nds, splits[0][1]['fft_len'], 16 * splits[0][1]['fft_len'] * (((splits[0][1]['fft_len'] // 2 + 1) + 7) // 8 * 8)),
                'seconds': round(t_sp, 5), 'sequential_chain_on_one_rank_s': round(t_sq, 5),
                'speedup': round(t_sq / t_sp, 3), 'grid_days_per_s': round(nds / t_sp, 1),
                'blocks': [(sp['first'], sp['count']) for _, sp, _ in sorted(splits)],
                'flagged': bool(any(sp['flagged'] for _, sp, _ in splits))}
    # ---- config 4
    seed = 1000 + rank
    del errors[:]
    if rehearse:
        rec = {'value': 1.0e6 + seed, 'evaluations_per_hour': 0.8e6, 'evaluation_fraction': 0.8, 'acceptance': 0.4,
               'timed_window_s': 1.0, 'samples': chain_samples}
        if fail_stage == 'bayes' and rank == world - 1:
            rec = guarded('chain', lambda: 1 / 0, None)
    else:
        rec = guarded('chain', lambda: bayes_case(400, 'auto', chain_samples, chain_burn, device, seed=seed)[0], None)
    barrier()
    chains = parallel.gather_objects((rank, seed, rec, list(errors)))
    if rank == 0 and any(r is None for _, _, r, _ in chains):
        out['bayes'] = {'error': [m for _, _, _, e in chains for m in e]}
        return out
    if rank == 0:
        chains = sorted(c[:3] for c in chains)
        out['bayes'] = {'workload': 'BASELINE config 4: %d independent chains, one per rank, chain c seeded 1000 + c, '
                                    'Kalbar, R = 400 (the reference\'s grid), exact-torus results' % world,
                        'value': round(sum(r['value'] for _, _, r in chains), 1), 'unit': 'samples/hour (sum over chains)',
                        'evaluations_per_hour': round(sum(r['evaluations_per_hour'] for _, _, r in chains), 1),
                        'seeds': [sd for _, sd, _ in chains],
                        'per_rank_samples_per_hour': [r['value'] for _, _, r in chains],
                        'per_rank_acceptance': [r['acceptance'] for _, _, r in chains],
                        'samples_per_chain': chain_samples}
        return out
    return None


if __name__ == '__main__':
    import json
    which = sys.argv[1] if len(sys.argv) > 1 else 'real_wind'
    if which == 'real_wind':
        R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
        print(json.dumps(real_wind_record(R=R)))
    elif which == 'prob_mass':
        print(json.dumps(prob_mass_roofline()))
    elif which == 'release':
        print(json.dumps(release_record()))
    elif which == 'prefix_split':
        print(json.dumps(prefix_split_record()))
    else:
        print(json.dumps(bayes_record()))
