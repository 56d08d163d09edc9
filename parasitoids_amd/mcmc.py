"""The sampler around the device-resident population model -- the caller that defines "MCMC
samples/hour" (SURVEY section 8f-1).  It restates the *model block* and the *step methods* of
the reference's Bayes_Run.py in plain numpy; PyMC 2 itself is absent here and out of scope:

  priors                      Bayes_Run.py:102-166  (PyMC 2 parameterisations: Gamma(alpha, rate),
                              Normal/TruncatedNormal(mu, tau = precision), Beta(a, b)), including
                              `A_collected` (:149-152) and the per-field `sent_obs_probs` Beta
                              nodes whose first parameter depends on it (:157-164)
  block of model parameters   Bayes_Run.py:176-196, :486-487  `AdaptiveMetropolis(stoc_vars,
                              scales=step_scales, interval=500, shrink_if_necessary=True)`: one
                              model evaluation per iteration -> class AdaptiveMetropolis below
                              (Haario et al. 2001 with PyMC 2.3's delay / interval / greedy /
                              shrink rules, restated from the published algorithm)
  the other stochastics       xi, em_obs_prob, grid_obs_prob, A_collected, sent_obs_probs_*: PyMC 2
                              assigns each its default scalar `Metropolis` (normal proposal,
                              sd = |initial value| x an acceptance-tuned factor) -> ScalarMetropolis
  model evaluation            Bayes_Run.py:204-336  -> pop_model.PopModel.evaluate (on the GPU)
  expected observations       Bayes_funcs.py        -> parasitoids_amd.Bayes_funcs (device gathers)
  Poisson observation model   Bayes_Run.py:344-433  (release-field emergence, sentinel-field
                              emergence, grid sampling)
  chain persistence / resume  Bayes_Run.py:484-485, :513-516 (hdf5 via PyTables there) ->
                              Sampler.save / Sampler.resume on an .npz file

One MCMC iteration = every step method once, like `pymc.MCMC.sample`: the block costs one model
evaluation, the scalar steps reuse the cached expected observations.

`locinfo` is `Data_Import.LocInfo` or any object with the attributes read by Bayes_funcs plus the
observation arrays (`release_emerg`, `sentinel_emerg`, `grid_obs`, `grid_samples`,
`release_collection`, `field_sizes`).  `synthetic_locinfo` builds such an object with observations
drawn from the model itself.
"""
import math
import time
import types

import numpy as np
from scipy.special import gammaln

from . import Bayes_funcs as BF
from . import _lib

# library return codes that mean "the model rejects these parameters" (ParasitoidModel.py:528-537,
# :568-599 raise AssertionError for them; a kernel that outgrows the pad is a shape error)
# PS_ERR_UNSUPPORTED: the proposal's kernel extent asks for a torus this solver mode cannot plan
# (mode='exact' and a pad with a prime factor > 1024, or any mode beyond the size limit) -- a property
# of the PARAMETERS in that mode, so the proposal is rejected and counted (`failed_evaluations`), the
# chain goes on; 'auto' / 'fast' never produce it below the size limit.  Device errors (PS_ERR_HIP,
# PS_ERR_OOM, PS_ERR_STATE) are not in this list: they stop the run with their cause.
_PARAMETER_ERRORS = (_lib.PS_ERR_HPROB_BOUNDS, _lib.PS_ERR_PMF_NEGATIVE, _lib.PS_ERR_FLIGHT_PROB,
                     _lib.PS_ERR_BAD_SHAPE, _lib.PS_ERR_EMPTY, _lib.PS_ERR_UNSUPPORTED)

NEG_INF = float('-inf')


# ------------------------------------------------------------------ log densities
def _lg_gamma(x, alpha, rate):
    if x <= 0:
        return NEG_INF
    return alpha * math.log(rate) - math.lgamma(alpha) + (alpha - 1) * math.log(x) - rate * x


def _lg_beta(x, a, b):
    if x <= 0 or x >= 1 or a <= 0 or b <= 0:
        return NEG_INF
    return (math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
            + (a - 1) * math.log(x) + (b - 1) * math.log1p(-x))


def _lg_normal(x, mu, tau):
    return 0.5 * math.log(tau / (2 * math.pi)) - 0.5 * tau * (x - mu) ** 2


def _lg_truncnormal(x, mu, tau, a, b):
    if x < a or x > b:
        return NEG_INF
    sd = 1.0 / math.sqrt(tau)
    z = 0.5 * (math.erf((b - mu) / (sd * math.sqrt(2))) - math.erf((a - mu) / (sd * math.sqrt(2))))
    return _lg_normal(x, mu, tau) - math.log(z)


def _lg_poisson(k, mu):
    if k < 0 or k != int(k):
        return NEG_INF
    return k * math.log(mu) - mu - math.lgamma(k + 1)


# name, log prior, initial value, AdaptiveMetropolis scale (Bayes_Run.py:102-131, :188-196)
MODEL_BLOCK = [
    ('g_aw', lambda v: _lg_gamma(v, 2.2, 1), 1.0, 0.04),
    ('g_bw', lambda v: _lg_gamma(v, 5, 1), 3.8, 0.08),
    ('f_a1', lambda v: _lg_truncnormal(v, 6, 0.3, 0, 9), 6.0, 0.25),
    ('f_b1_p', lambda v: _lg_gamma(v, 2, 1), 1.5, 0.05),
    ('f_a2', lambda v: _lg_truncnormal(v, 20, 0.3, 15, 24), 20.0, 0.25),
    ('f_b2_p', lambda v: _lg_gamma(v, 2, 1), 1.5, 0.05),
    ('sig_x', lambda v: _lg_gamma(v, 26, 0.15), 180.0, 2.0),
    ('sig_y', lambda v: _lg_gamma(v, 15, 0.15), 150.0, 2.0),
    ('corr_p', lambda v: _lg_beta(v, 5, 5), 0.5, 0.0005),
    ('sig_x_l', lambda v: _lg_gamma(v, 2, 0.08), 10.0, 2.0),
    ('sig_y_l', lambda v: _lg_gamma(v, 2, 0.14), 10.0, 2.0),
    ('corr_l_p', lambda v: _lg_beta(v, 5, 5), 0.5, 0.0005),
    ('lam', lambda v: _lg_beta(v, 5, 1), 0.95, 0.0005),
    ('n_periods', lambda v: _lg_poisson(v, 30), 30.0, 1.0),
    ('mu_r', lambda v: _lg_normal(v, 1.0, 1.0), 1.0, 0.005),
]
# observation-model scalars (Bayes_Run.py:132-141): name, log prior, initial value
NUISANCE = [
    ('xi', lambda v: _lg_gamma(v, 1, 1), 0.75),
    ('em_obs_prob', lambda v: _lg_beta(v, 1, 1), 0.05),
    ('grid_obs_prob', lambda v: _lg_beta(v, 1, 1), 0.005),
]
# likelihood parts (release, sentinel, grid) each nuisance parameter enters (loglik_parts)
NUISANCE_PARTS = [(True, True, False), (True, False, False), (False, False, True)]
SENT_BETA = 40.0            # Bayes_Run.py:158
A_COLLECTED_INIT = 2500.0   # Bayes_Run.py:149-152: TruncatedNormal(2500, tau=1/2500, 0, min field area)
DISCRETE = [m[0] == 'n_periods' for m in MODEL_BLOCK]


def model_args(theta):
    """Sampled block -> prob_mass arguments (Bayes_Run.py:216-232; the deterministic f_b = p + 1
    and corr = 2 p - 1 of :106-113, :119-127)."""
    t = dict(zip([m[0] for m in MODEL_BLOCK], theta))
    hparams = (t['lam'], t['g_aw'], t['g_bw'], t['f_a1'], t['f_b1_p'] + 1, t['f_a2'], t['f_b2_p'] + 1)
    Dparams = (t['sig_x'], t['sig_y'], 2 * t['corr_p'] - 1)
    Dlparams = (t['sig_x_l'], t['sig_y_l'], 2 * t['corr_l_p'] - 1)
    return hparams, Dparams, Dlparams, t['mu_r'], int(round(t['n_periods']))


def field_areas(locinfo, cell_area):
    """area in m^2 of every sentinel field, in `sent_ids` order"""
    return np.array([locinfo.field_sizes[k] * cell_area for k in locinfo.sent_ids], dtype=np.float64)


def initial_sent_obs_probs(locinfo, cell_area):
    return 0.1 * 3600.0 / field_areas(locinfo, cell_area)           # Bayes_Run.py:164


# ------------------------------------------------------------------ likelihood
def poisson_loglik(obs, rate, lgam=None):
    """sum over cells of log Poisson(obs | rate); a zero rate only explains zero counts.
    `lgam` = sum of gammaln(obs + 1) when the caller has it (it depends on the data only)."""
    obs = np.asarray(obs, dtype=np.float64)
    rate = np.asarray(rate, dtype=np.float64)
    if not np.all(np.isfinite(rate)) or rate.min(initial=0.0) < 0:
        return NEG_INF
    pos = rate > 0
    if not pos.all():
        if np.any(obs[~pos] > 0):
            return NEG_INF
        g = float(gammaln(obs[pos] + 1).sum())
        return float((obs[pos] * np.log(rate[pos]) - rate[pos]).sum()) - g
    if lgam is None:
        lgam = float(gammaln(obs + 1).sum())
    return float((obs * np.log(rate) - rate).sum()) - lgam


def observation_cache(locinfo):
    """The data side of the likelihood, prepared once per site object: observation arrays as
    float64, collection efforts, grid sampling effort and the sums of log(obs!)."""
    c = getattr(locinfo, '_ps_obs', None)
    if c is not None:
        return c
    f = lambda a: np.asarray(a, dtype=np.float64)
    c = {'rel': [f(a) for a in locinfo.release_emerg], 'sen': [f(a) for a in locinfo.sentinel_emerg],
         'grid': f(locinfo.grid_obs), 'effort': [f(a) for a in locinfo.release_collection],
         'samples': f(locinfo.grid_samples)}
    c['rel_lg'] = [float(gammaln(a + 1).sum()) for a in c['rel']]
    c['sen_lg'] = [float(gammaln(a + 1).sum()) for a in c['sen']]
    c['grid_lg'] = float(gammaln(c['grid'] + 1).sum())
    try:
        locinfo._ps_obs = c
    except AttributeError:
        pass
    return c


def expected_observations(pop_model, locinfo):
    """(release_emerg, sentinel_emerg, grid_counts) of Bayes_Run.py:325-336 (one device gather)."""
    return BF.expected_observations(pop_model, locinfo)


def loglik_parts(expected, locinfo, nuis, sent_obs_probs):
    """(release, sentinel, grid) log likelihoods, Bayes_Run.py:344-433: Poisson rates
    xi*emerg*effort*em_obs_prob (release grids), xi*emerg*sent_obs_prob[field] (sentinel fields),
    grid_obs_prob*samples*density (grid)."""
    rel, sen, grid = expected
    xi, em_p, grid_p = nuis
    c = observation_cache(locinfo)
    ll_rel = 0.0
    for ii, e in enumerate(rel):
        ll_rel += poisson_loglik(c['rel'][ii], xi * e * (c['effort'][ii] * em_p)[:, None], c['rel_lg'][ii])
    ll_sen = 0.0
    sp = np.asarray(sent_obs_probs, dtype=np.float64)[:, None]
    for ii, e in enumerate(sen):
        ll_sen += poisson_loglik(c['sen'][ii], xi * e * sp, c['sen_lg'][ii])
    ll_grid = poisson_loglik(c['grid'], grid_p * c['samples'] * grid, c['grid_lg'])
    return ll_rel, ll_sen, ll_grid


def lik_stats(expected, locinfo):
    """Sufficient statistics of the observation likelihood for ONE model evaluation.  Every Poisson
    rate of loglik_parts is (a product of scalar nuisance parameters) x (an array fixed by the
    evaluation): sum(obs log(s b) - s b) = log(s) sum(obs) + sum(obs log b) - s sum(b).  The scalar
    Metropolis steps (xi, em_obs_prob, grid_obs_prob, sent_obs_probs_k: 6-7 per sample, no model
    evaluation) then cost a handful of scalar operations instead of a pass over the arrays --
    0.66 -> 0.2 ms of host time per sample, which the GPU spends idle.
    -> dict with (S, A, B, bad) per release grid, per sentinel field (arrays over the fields) and
    for the grid counts; `bad` = some cell with a zero base rate has a positive count."""
    rel, sen, grid = expected
    c = observation_cache(locinfo)

    def sab(obs, base, axis=None):
        pos = base > 0
        bad = bool(np.any((~pos) & (obs > 0))) or not bool(np.all(np.isfinite(base))) or bool(np.any(base < 0))
        logb = np.log(np.where(pos, base, 1.0))
        return obs.sum(axis=axis), (obs * logb).sum(axis=axis), base.sum(axis=axis), bad

    # aggregated: the release grids share one scalar (xi em_obs_prob), the sentinel arrays share
    # the per-field scalars (xi sent_obs_prob_k); plain floats, the consumers are scalar code
    rS = rA = rB = 0.0
    rbad = False
    for ii, e in enumerate(rel):
        S, A, B, bad = sab(c['rel'][ii], np.asarray(e, dtype=np.float64) * c['effort'][ii][:, None])
        rS, rA, rB, rbad = rS + float(S), rA + float(A), rB + float(B), rbad or bad
    sS = sA = sB = None
    sbad = False
    for ii, e in enumerate(sen):
        # one sent_obs_prob per field = per row of the sentinel arrays
        S, A, B, bad = sab(c['sen'][ii], np.asarray(e, dtype=np.float64), axis=1)
        sS, sA, sB = (S, A, B) if sS is None else (sS + S, sA + A, sB + B)
        sbad = sbad or bad
    gS, gA, gB, gbad = sab(c['grid'], c['samples'] * np.asarray(grid, dtype=np.float64))
    return {'rel': (rS, rA, rB, rbad),
            'sen': ([float(v) for v in sS], [float(v) for v in sA], [float(v) for v in sB], sbad) if sS is not None
            else ([], [], [], False),
            'grid': (float(gS), float(gA), float(gB), gbad),
            'lg': (sum(c['rel_lg']), sum(c['sen_lg']), c['grid_lg'])}


def loglik_parts_stats(st, nuis, sent_obs_probs, which=(True, True, True), prev=None):
    """loglik_parts from the statistics of lik_stats (same values to round-off).  `which` selects
    the parts to compute (release, sentinel, grid); the others are taken from `prev`."""
    xi, em_p, grid_p = float(nuis[0]), float(nuis[1]), float(nuis[2])
    out = [None, None, None] if prev is None else list(prev)
    if which[0]:
        S, A, B, bad = st['rel']
        s = xi * em_p
        ll = NEG_INF
        if not bad:
            if s > 0:
                ll = math.log(s) * S + A - s * B - st['lg'][0]
            elif s == 0 and S == 0:
                ll = -st['lg'][0]
        out[0] = ll
    if which[1]:
        S, A, B, bad = st['sen']
        ll = NEG_INF
        if not bad:
            ll = -st['lg'][1]
            for k in range(len(S)):
                r = xi * float(sent_obs_probs[k])
                if r > 0:
                    ll += math.log(r) * S[k] + A[k] - r * B[k]
                elif r < 0 or S[k] > 0:     # a zero rate explains only zero counts
                    ll = NEG_INF
                    break
        out[1] = ll
    if which[2]:
        S, A, B, bad = st['grid']
        ll = NEG_INF
        if not bad:
            if grid_p > 0:
                ll = math.log(grid_p) * S + A - grid_p * B - st['lg'][2]
            elif grid_p == 0 and S == 0:
                ll = -st['lg'][2]
        out[2] = ll
    return tuple(out)


def observation_loglik(expected, locinfo, nuis, sent_obs_probs):
    return sum(loglik_parts(expected, locinfo, nuis, sent_obs_probs))


def collection_logprior(A_collected, sent_obs_probs, areas):
    """log p(A_collected) + sum_k log p(sent_obs_probs_k | A_collected), Bayes_Run.py:149-164:
    the Beta's mean is A_collected / field area with its second parameter fixed at 40."""
    lp = _lg_truncnormal(A_collected, 2500.0, 1.0 / 2500.0, 0.0, float(areas.min()))
    if lp == NEG_INF:
        return NEG_INF
    for p, area in zip(sent_obs_probs, areas):
        m = A_collected / area
        if not 0.0 < m < 1.0:
            return NEG_INF
        lp += _lg_beta(p, m * SENT_BETA / (1.0 - m), SENT_BETA)
    return lp


def log_prior(theta, nuis, A_collected, sent_obs_probs, areas):
    lp = sum(m[1](v) for m, v in zip(MODEL_BLOCK, theta))
    lp += sum(m[1](v) for m, v in zip(NUISANCE, nuis))
    return lp + collection_logprior(A_collected, sent_obs_probs, areas)


def log_posterior(theta, nuis, A_collected, sent_obs_probs, expected, locinfo, cell_area):
    """Joint log density of every stochastic of the reference's model at one point, given the
    expected observations of the model evaluation at `theta` (a pure host function)."""
    lp = log_prior(theta, nuis, A_collected, sent_obs_probs, field_areas(locinfo, cell_area))
    if lp == NEG_INF:
        return NEG_INF
    return lp + observation_loglik(expected, locinfo, nuis, sent_obs_probs)


# ------------------------------------------------------------------ step methods
class AdaptiveMetropolis():
    """Block random-walk Metropolis whose proposal covariance is re-estimated from the chain
    (Haario, Saksman & Tamminen 2001), with the rules of PyMC 2.3's step method of that name as
    the reference configures it (Bayes_Run.py:486-487):

      * initial covariance = diag(scales) (PyMC's `cov_from_scales` puts the scales on the
        diagonal of the covariance); proposals are value + chol(C) z, discrete entries rounded;
      * `greedy`: until `delay` iterations have passed only accepted states enter the internal
        trace, afterwards every state does;
      * from iteration `delay` on, every `interval` iterations:
        C <- (k-1)/(n-1) C + s/(n-1) (k m m' + X'X - n m_new m_new' + eps I),  s = 2.4^2/dim,
        eps = 1e-5, k / n the trace lengths before / after the new block X;
      * `shrink_if_necessary`: acceptance below 1e-3 scales C by 0.01, below 1e-2 by 0.25;
      * a covariance that is not positive definite keeps the previous proposal.
    """

    def __init__(self, scales, discrete=None, delay=1000, interval=500, greedy=True,
                 shrink_if_necessary=True):
        scales = np.asarray(scales, dtype=np.float64)
        self.dim = scales.size
        self.discrete = np.zeros(self.dim, bool) if discrete is None else np.asarray(discrete, bool)
        self.C = np.diag(scales)
        self.proposal_sd = np.linalg.cholesky(self.C)
        self.delay, self.interval = int(delay), int(interval)
        self.greedy0 = bool(greedy)
        self.shrink_if_necessary = bool(shrink_if_necessary)
        self.accepted = 0
        self.rejected = 0
        self.current_iter = 0
        self.trace_count = 0
        self.chain_mean = np.zeros(self.dim)
        self._trace = []
        self.cov_updates = 0

    def propose(self, value, rng):
        jump = self.proposal_sd @ rng.normal(size=self.dim)
        jump[self.discrete] = np.round(jump[self.discrete])
        return value + jump

    def tally(self, value, accepted):
        """bookkeeping after the accept/reject decision of one iteration (`value` = the state
        the chain is in now)"""
        if accepted:
            self.accepted += 1
        else:
            self.rejected += 1
        greedy = self.greedy0 and self.current_iter < self.delay
        if accepted or not greedy:
            self._trace.append(np.array(value, dtype=np.float64))
        if self.current_iter > self.delay and self.current_iter % self.interval == 0:
            self.update_cov()
        self.current_iter += 1

    def update_cov(self):
        if not self._trace:
            return
        chain = np.asarray(self._trace)
        scaling = 2.4 ** 2 / self.dim
        eps = 1.0e-5
        k = self.trace_count
        n = k + len(chain)
        new_mean = k * self.chain_mean / n + chain.sum(0) / n
        if n > 1:
            t0 = k * np.outer(self.chain_mean, self.chain_mean)
            t1 = chain.T @ chain
            t2 = n * np.outer(new_mean, new_mean)
            t3 = eps * np.eye(self.dim)
            self.C = (k - 1) / (n - 1.0) * self.C + scaling / (n - 1.0) * (t0 + t1 - t2 + t3)
        self.chain_mean = new_mean
        if self.shrink_if_necessary:
            rate = self.accepted / max(1, self.accepted + self.rejected)
            if rate < 0.001:
                self.C *= 0.01
            elif rate < 0.01:
                self.C *= 0.25
        try:
            self.proposal_sd = np.linalg.cholesky(self.C)
        except np.linalg.LinAlgError:
            pass
        self.trace_count = n
        self._trace = []
        self.cov_updates += 1

    def state(self):
        return {'am_C': self.C, 'am_sd': self.proposal_sd, 'am_mean': self.chain_mean,
                'am_trace': np.asarray(self._trace).reshape(-1, self.dim),
                'am_counts': np.array([self.accepted, self.rejected, self.current_iter,
                                       self.trace_count, self.cov_updates])}

    def load_state(self, st):
        self.C, self.proposal_sd, self.chain_mean = st['am_C'].copy(), st['am_sd'].copy(), st['am_mean'].copy()
        self._trace = [r.copy() for r in st['am_trace']]
        self.accepted, self.rejected, self.current_iter, self.trace_count, self.cov_updates = (
            int(v) for v in st['am_counts'])


class ScalarMetropolis():
    """PyMC 2's default step method for a continuous scalar stochastic: normal proposal with
    sd = |initial value| * adaptive_scale_factor, the factor re-tuned from the acceptance rate of
    the last `tune_interval` steps (PyMC 2 `Metropolis.tune`)."""

    def __init__(self, value, tune_interval=1000):
        self.proposal_sd = abs(value) if value != 0 else 1.0
        self.factor = 1.0
        self.tune_interval = int(tune_interval)
        self.accepted = 0
        self.rejected = 0
        self._acc = 0
        self._n = 0

    def propose(self, value, rng):
        return value + rng.normal(0.0, self.proposal_sd * self.factor)

    def tally(self, accepted):
        self.accepted += int(accepted)
        self.rejected += int(not accepted)
        self._acc += int(accepted)
        self._n += 1
        if self._n >= self.tune_interval:
            r = self._acc / self._n
            if r < 0.001:
                self.factor *= 0.1
            elif r < 0.05:
                self.factor *= 0.5
            elif r < 0.2:
                self.factor *= 0.9
            elif r > 0.95:
                self.factor *= 10.0
            elif r > 0.75:
                self.factor *= 2.0
            elif r > 0.5:
                self.factor *= 1.1
            self._acc = self._n = 0


# ------------------------------------------------------------------ sampler
class Sampler():
    """The reference's Bayes_Run model: AdaptiveMetropolis on the 15 model parameters (one
    pop_model evaluation per iteration) + scalar Metropolis on xi, em_obs_prob, grid_obs_prob,
    A_collected and every sent_obs_probs_k.  `adaptive=False` freezes the block proposal at its
    initial diag(scales) (plain Metropolis)."""

    def __init__(self, pop_model, locinfo, cell_area, seed=0, ndays=None, adaptive=True,
                 delay=1000, interval=500, tune_interval=1000, evaluate=None):
        self.pm = pop_model
        self.li = locinfo
        self.cell_area = float(cell_area)
        self.ndays = ndays
        self.rng = np.random.default_rng(seed)
        self.areas = field_areas(locinfo, cell_area)
        self.theta = np.array([m[2] for m in MODEL_BLOCK], dtype=np.float64)
        self.nuis = np.array([m[2] for m in NUISANCE], dtype=np.float64)
        self.A_collected = min(A_COLLECTED_INIT, 0.5 * float(self.areas.min()))
        self.sent_obs_probs = initial_sent_obs_probs(locinfo, cell_area)
        self.block = AdaptiveMetropolis([m[3] for m in MODEL_BLOCK], DISCRETE,
                                        delay=delay if adaptive else 1 << 60, interval=interval)
        self.scalars = ([ScalarMetropolis(v, tune_interval) for v in self.nuis]
                        + [ScalarMetropolis(self.A_collected, tune_interval)]
                        + [ScalarMetropolis(v, tune_interval) for v in self.sent_obs_probs])
        self._evaluate_fn = evaluate          # tests: expected observations without a device
        self.n_eval = 0
        self.n_failed = 0
        self.iteration = 0
        self.expected = self._evaluate(self.theta)
        if self.expected is None:
            raise ValueError('the initial parameters do not evaluate')
        self._refresh()
        if not math.isfinite(self.logp):
            raise ValueError('the initial point has zero probability')

    # bookkeeping of the pieces of the joint density
    def _refresh(self):
        self.lp_model = sum(m[1](v) for m, v in zip(MODEL_BLOCK, self.theta))
        self.lp_nuis = [m[1](v) for m, v in zip(NUISANCE, self.nuis)]
        self.lp_coll = collection_logprior(self.A_collected, self.sent_obs_probs, self.areas)
        self.stats = lik_stats(self.expected, self.li)
        self.ll = list(loglik_parts_stats(self.stats, self.nuis, self.sent_obs_probs))

    @property
    def logp(self):
        return self.lp_model + sum(self.lp_nuis) + self.lp_coll + sum(self.ll)

    @property
    def accepted(self):
        return self.block.accepted

    @property
    def proposed(self):
        return self.block.accepted + self.block.rejected

    def _evaluate(self, theta):
        """one model evaluation on the GPU + the gathers; None if the parameters are
        rejected by the model's own checks (ParasitoidModel.py:528-537, :568-599)"""
        self.n_eval += 1
        if self._evaluate_fn is not None:
            return self._evaluate_fn(theta)
        try:
            self.pm.evaluate(*model_args(theta), ndays=self.ndays, want_stats=False)
        except (AssertionError, ValueError):
            self.n_failed += 1
            return None
        except _lib.HipError as e:
            # only the library's parameter / shape checks count as a rejected proposal; a device
            # error (PS_ERR_HIP, out of memory, a missing state) must stop the run with its cause
            if e.code not in _PARAMETER_ERRORS:
                raise
            self.n_failed += 1
            return None
        return expected_observations(self.pm, self.li)

    def _accept(self, new, old):
        return math.log(self.rng.random()) < new - old

    def step(self):
        # --- AdaptiveMetropolis block: the 15 model parameters, one evaluation
        prop = self.block.propose(self.theta, self.rng)
        lp_model = sum(m[1](v) for m, v in zip(MODEL_BLOCK, prop))
        ok = False
        if lp_model > NEG_INF:
            exp = self._evaluate(prop)
            if exp is not None:
                st = lik_stats(exp, self.li)
                ll = list(loglik_parts_stats(st, self.nuis, self.sent_obs_probs))
                if self._accept(lp_model + sum(ll), self.lp_model + sum(self.ll)):
                    self.theta, self.expected, self.lp_model, self.ll = prop, exp, lp_model, ll
                    self.stats = st
                    ok = True
        self.block.tally(self.theta, ok)
        # --- scalar steps: no model evaluation
        for i, m in enumerate(NUISANCE):
            st = self.scalars[i]
            nu = self.nuis.copy()
            nu[i] = st.propose(nu[i], self.rng)
            lp = m[1](nu[i])
            ok = False
            if lp > NEG_INF:
                ll = list(loglik_parts_stats(self.stats, nu, self.sent_obs_probs, NUISANCE_PARTS[i], self.ll))
                if self._accept(lp + sum(ll), self.lp_nuis[i] + sum(self.ll)):
                    self.nuis, self.lp_nuis[i], self.ll = nu, lp, ll
                    ok = True
            st.tally(ok)
        st = self.scalars[len(NUISANCE)]
        A = st.propose(self.A_collected, self.rng)
        lp = collection_logprior(A, self.sent_obs_probs, self.areas)
        ok = lp > NEG_INF and self._accept(lp, self.lp_coll)
        if ok:
            self.A_collected, self.lp_coll = A, lp
        st.tally(ok)
        for k in range(len(self.sent_obs_probs)):
            st = self.scalars[len(NUISANCE) + 1 + k]
            sp = self.sent_obs_probs.copy()
            sp[k] = st.propose(sp[k], self.rng)
            lp = collection_logprior(self.A_collected, sp, self.areas)
            ok = False
            if lp > NEG_INF:
                ll_sen = loglik_parts_stats(self.stats, self.nuis, sp, (False, True, False), self.ll)[1]
                if self._accept(lp + ll_sen, self.lp_coll + self.ll[1]):
                    self.sent_obs_probs, self.lp_coll = sp, lp
                    self.ll[1] = ll_sen
                    ok = True
            st.tally(ok)
        self.iteration += 1

    # trace columns
    def names(self):
        return ([m[0] for m in MODEL_BLOCK] + [m[0] for m in NUISANCE] + ['A_collected']
                + ['sent_obs_probs_{}'.format(k) for k in self.li.sent_ids])

    def point(self):
        return np.concatenate([self.theta, self.nuis, [self.A_collected], self.sent_obs_probs])

    def run(self, nsamples):
        trace = np.empty((nsamples, len(self.names())))
        logp = np.empty(nsamples)
        ev0 = self.n_eval
        t0 = time.perf_counter()
        for n in range(nsamples):
            self.step()
            trace[n] = self.point()
            logp[n] = self.logp
        dt = time.perf_counter() - t0
        self.last = {'trace': trace, 'logp': logp}
        return {'trace': trace, 'logp': logp, 'seconds': dt,
                'samples_per_hour': 3600.0 * nsamples / dt if dt > 0 else float('inf'),
                'acceptance': self.accepted / max(1, self.proposed),
                'evaluations': self.n_eval, 'evaluations_this_run': self.n_eval - ev0,
                'failed_evaluations': self.n_failed, 'cov_updates': self.block.cov_updates,
                'names': self.names()}

    # ---- persistence (Bayes_Run.py:484-485 new chain into a database, :513-516 resume)
    def save(self, fname, trace=None, logp=None):
        """Write the chain so far and the complete sampler state; `resume` continues from it as
        if the run had never stopped (same random stream)."""
        old = {}
        if trace is None:
            trace, logp = self.last['trace'], self.last['logp']
        st = self.block.state()
        sc = np.array([[s.proposal_sd, s.factor, s.accepted, s.rejected, s._acc, s._n] for s in self.scalars])
        np.savez(fname, trace=trace, logp=logp, names=np.array(self.names()), point=self.point(),
                 scalars=sc, iteration=self.iteration, n_eval=np.array([self.n_eval, self.n_failed]),
                 rng=np.array([repr(self.rng.bit_generator.state)]), **st, **old)

    def resume(self, fname):
        """Load a chain written by `save`: returns (trace, logp) so far and puts the sampler
        where that run stopped."""
        import ast
        f = np.load(fname if str(fname).endswith('.npz') else str(fname) + '.npz', allow_pickle=False)
        if list(f['names']) != self.names():
            raise ValueError('the chain file was written for a different model')
        pt = f['point']
        nb, nn = len(MODEL_BLOCK), len(NUISANCE)
        self.theta, self.nuis = pt[:nb].copy(), pt[nb:nb + nn].copy()
        self.A_collected, self.sent_obs_probs = float(pt[nb + nn]), pt[nb + nn + 1:].copy()
        self.block.load_state(f)
        for s, row in zip(self.scalars, f['scalars']):
            s.proposal_sd, s.factor = float(row[0]), float(row[1])
            s.accepted, s.rejected, s._acc, s._n = (int(v) for v in row[2:])
        self.iteration = int(f['iteration'])
        self.n_eval, self.n_failed = (int(v) for v in f['n_eval'])
        self.rng.bit_generator.state = ast.literal_eval(str(f['rng'][0]))
        self.n_eval -= 1
        self.expected = self._evaluate(self.theta)
        self._refresh()
        return f['trace'], f['logp']


def run_parallel(samplers, nsamples):
    """Run several independent chains of ONE process side by side, one host thread each (BASELINE
    config 4 keeps one chain per GPU; this fills a GPU that a single R = 400 chain leaves mostly
    idle).  Every sampler owns its PopModel -- its own model / solver handles and HIP streams
    (include/parasitoid_hip.h: handles are independent) -- and its own random stream, the library
    calls release the GIL, so the chains' launches interleave on the device and one chain's host work
    (proposal, likelihood) hides behind the others' kernels.  A chain's trace does not depend on what
    runs next to it: bit-identical to `sampler.run(nsamples)` on its own.
    -> (list of run() results in sampler order, wall seconds)"""
    import threading
    out = [None] * len(samplers)
    errs = []

    def work(i):
        try:
            out[i] = samplers[i].run(nsamples)
        except BaseException as e:      # re-raised in the caller's thread
            errs.append((i, e))

    threads = [threading.Thread(target=work, args=(i,), name='chain-%d' % i) for i in range(len(samplers))]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0][1]
    return out, dt


class Metropolis(Sampler):
    """Round-1 name of the sampler, kept for callers: the same model with the block proposal
    frozen at its initial covariance unless `adaptive=True`."""

    def __init__(self, pop_model, locinfo, cell_area, seed=0, ndays=None, adaptive=False, **kw):
        super().__init__(pop_model, locinfo, cell_area, seed=seed, ndays=ndays, adaptive=adaptive, **kw)


# ------------------------------------------------------------------ synthetic observations
def synthetic_locinfo(pop_model, rad_res, true_theta=None, true_nuis=None, seed=9, ndays=None):
    """A LocInfo stand-in with the geometry of a Kalbar-like campaign scaled to the domain
    (release-field emergence grids, three sentinel fields, a sampling grid) and observations
    drawn from the model itself at `true_theta` -- SYNTHETIC data, labelled as such wherever it
    is reported.  Dates are integers (days post release), which Bayes_funcs accepts."""
    rng = np.random.default_rng(seed)
    R = int(rad_res)
    s = R / 128.0                           # the G9 fixture geometry was laid out at R = 128

    def box(lo, hi, n):
        lo_, hi_ = int(R + (lo - 128) * s), max(int(R + (hi - 128) * s), int(R + (lo - 128) * s) + 1)
        return rng.integers(lo_, hi_, size=(n, 2))

    li = types.SimpleNamespace()
    li.collection_datesPR = [3, 6]
    li.emerg_grids = [[(int(r), int(c)) for r, c in box(118, 139, 12)],
                      [(int(r), int(c)) for r, c in box(110, 147, 9)]]
    li.release_DataFrames = [{'datePR': [22, 22, 24, 27]}, {'datePR': [25, 28, 28, 30]}]
    li.release_collection = [np.full(12, 1.0), np.full(9, 0.5)]
    li.sent_ids = ['A', 'B', 'C']
    li.field_cells = {'A': box(100, 157, 40), 'B': box(60, 200, 25), 'C': box(120, 137, 60)}
    li.field_sizes = {k: len(v) for k, v in li.field_cells.items()}
    li.sent_DataFrames = [{'datePR': [23, 26]}, {'datePR': [26, 29, 31]}]
    li.grid_cells = box(100, 157, 30)
    li.grid_obs_datesPR = [2, 5, 6]
    li.grid_samples = np.full((30, 3), 1.0)
    theta = np.array([m[2] for m in MODEL_BLOCK]) if true_theta is None else np.asarray(true_theta, float)
    nuis = np.array([m[2] for m in NUISANCE]) if true_nuis is None else np.asarray(true_nuis, float)
    pop_model.evaluate(*model_args(theta), ndays=ndays)
    rel, sen, grid = expected_observations(pop_model, li)
    cell_area = (pop_model.rad_dist / pop_model.rad_res) ** 2
    sent_p = initial_sent_obs_probs(li, cell_area)
    xi, em_p, grid_p = nuis
    li.release_emerg = [rng.poisson(xi * e * (li.release_collection[i] * em_p)[:, None]) for i, e in enumerate(rel)]
    li.sentinel_emerg = [rng.poisson(xi * e * sent_p[:, None]) for e in sen]
    li.grid_obs = rng.poisson(grid_p * li.grid_samples * grid)
    return li
