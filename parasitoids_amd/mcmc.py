"""A minimal Metropolis sampler around the device-resident population model -- the caller
that defines "MCMC samples/hour" (SURVEY section 8f-1).  It restates the *model block* of the
reference's Bayes_Run.py in plain numpy instead of PyMC 2 (absent here and out of scope):

  priors                      Bayes_Run.py:102-138  (PyMC 2 parameterisations: Gamma(alpha, rate),
                              Normal/TruncatedNormal(mu, tau = precision), Beta(a, b))
  block of model parameters   Bayes_Run.py:176-196  (one model evaluation per proposal, the
                              reference's initial step scales; plain Metropolis, not the
                              adaptive variant)
  model evaluation            Bayes_Run.py:204-336  -> pop_model.PopModel.evaluate (on the GPU)
  expected observations       Bayes_funcs.py        -> parasitoids_amd.Bayes_funcs (device gathers)
  Poisson observation model   Bayes_Run.py:344-433  (release-field emergence, sentinel-field
                              emergence, grid sampling)

Simplifications, all on the cheap side of the model evaluation: the per-field sentinel
observation probabilities are held at the reference's initial values (0.1*3600/field area,
:154-157) instead of being sampled, `A_collected` is not sampled, and the three scalar
nuisance parameters (xi, em_obs_prob, grid_obs_prob) get their own Metropolis steps that
reuse the cached expected observations (no model evaluation).

`locinfo` is the reference's Data_Import.LocInfo or any object with the attributes read by
Bayes_funcs plus the observation arrays (`release_emerg`, `sentinel_emerg`, `grid_obs`,
`grid_samples`, `release_collection`, `field_sizes`).  `synthetic_locinfo` builds such an
object with observations drawn from the model itself (no xlsx reader in this image).
"""
import math
import time
import types

import numpy as np
from scipy.special import gammaln

from . import Bayes_funcs as BF

NEG_INF = float('-inf')


# ------------------------------------------------------------------ log densities
def _lg_gamma(x, alpha, rate):
    if x <= 0:
        return NEG_INF
    return alpha * math.log(rate) - math.lgamma(alpha) + (alpha - 1) * math.log(x) - rate * x


def _lg_beta(x, a, b):
    if x <= 0 or x >= 1:
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


# name, log prior, initial value, step scale (Bayes_Run.py:102-131, :188-196)
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
NUISANCE = [
    ('xi', lambda v: _lg_gamma(v, 1, 1), 0.75, 0.05),
    ('em_obs_prob', lambda v: _lg_beta(v, 1, 1), 0.05, 0.005),
    ('grid_obs_prob', lambda v: _lg_beta(v, 1, 1), 0.005, 0.0005),
]


def model_args(theta):
    """Sampled block -> prob_mass arguments (Bayes_Run.py:216-232; the deterministic f_b = p + 1
    and corr = 2 p - 1 of :106-113, :119-127)."""
    t = dict(zip([m[0] for m in MODEL_BLOCK], theta))
    hparams = (t['lam'], t['g_aw'], t['g_bw'], t['f_a1'], t['f_b1_p'] + 1, t['f_a2'], t['f_b2_p'] + 1)
    Dparams = (t['sig_x'], t['sig_y'], 2 * t['corr_p'] - 1)
    Dlparams = (t['sig_x_l'], t['sig_y_l'], 2 * t['corr_l_p'] - 1)
    return hparams, Dparams, Dlparams, t['mu_r'], int(round(t['n_periods']))


# ------------------------------------------------------------------ likelihood
def poisson_loglik(obs, rate):
    """sum over cells of log Poisson(obs | rate); a zero rate only explains zero counts."""
    obs = np.asarray(obs, dtype=np.float64)
    rate = np.asarray(rate, dtype=np.float64)
    if np.any((rate <= 0) & (obs > 0)) or np.any(rate < 0) or not np.all(np.isfinite(rate)):
        return NEG_INF
    pos = rate > 0
    return float((obs[pos] * np.log(rate[pos]) - rate[pos] - gammaln(obs[pos] + 1)).sum())


def expected_observations(pop_model, locinfo):
    """(release_emerg, sentinel_emerg, grid_counts) of Bayes_Run.py:325-336."""
    rel, sen = BF.popdensity_to_emergence(pop_model, locinfo)
    grid = BF.popdensity_grid(pop_model, locinfo)
    return rel, sen, grid


def observation_loglik(expected, locinfo, nuis, sent_obs_probs):
    """Bayes_Run.py:344-433: Poisson rates xi*emerg*beta (release grids, with collection
    effort; sentinel fields with their per-field probability) and beta*samples*density (grid)."""
    rel, sen, grid = expected
    xi, em_p, grid_p = nuis
    ll = 0.0
    for ii, e in enumerate(rel):
        effort = np.asarray(locinfo.release_collection[ii], dtype=np.float64)
        ll += poisson_loglik(locinfo.release_emerg[ii], xi * e * (effort * em_p)[:, None])
    for ii, e in enumerate(sen):
        ll += poisson_loglik(locinfo.sentinel_emerg[ii], xi * e * np.asarray(sent_obs_probs)[:, None])
    ll += poisson_loglik(locinfo.grid_obs, grid_p * np.asarray(locinfo.grid_samples) * grid)
    return ll


# ------------------------------------------------------------------ sampler
class Metropolis():
    def __init__(self, pop_model, locinfo, cell_area, seed=0, ndays=None):
        self.pm = pop_model
        self.li = locinfo
        self.ndays = ndays
        self.rng = np.random.default_rng(seed)
        self.theta = np.array([m[2] for m in MODEL_BLOCK], dtype=np.float64)
        self.nuis = np.array([m[2] for m in NUISANCE], dtype=np.float64)
        self.sent_obs_probs = np.array([0.1 * 3600.0 / (locinfo.field_sizes[k] * cell_area)
                                        for k in locinfo.sent_ids])
        self.n_eval = 0
        self.n_failed = 0
        self.accepted = 0
        self.proposed = 0
        self.expected = self._evaluate(self.theta)
        if self.expected is None:
            raise ValueError('the initial parameters do not evaluate')
        self.lp = self._prior(self.theta, MODEL_BLOCK) + self._prior(self.nuis, NUISANCE)
        self.ll = observation_loglik(self.expected, self.li, self.nuis, self.sent_obs_probs)

    @staticmethod
    def _prior(vals, table):
        return sum(m[1](v) for m, v in zip(table, vals))

    def _evaluate(self, theta):
        """one model evaluation on the GPU + the gathers; None if the parameters are
        rejected by the model's own checks (ParasitoidModel.py:528-537, :568-599)"""
        self.n_eval += 1
        try:
            self.pm.evaluate(*model_args(theta), ndays=self.ndays)
        except (AssertionError, ValueError, RuntimeError):
            self.n_failed += 1
            return None
        return expected_observations(self.pm, self.li)

    def step(self):
        # block update of the 15 model parameters: one evaluation
        prop = self.theta + self.rng.normal(0.0, 1.0, self.theta.size) * np.array([m[3] for m in MODEL_BLOCK])
        k = [m[0] for m in MODEL_BLOCK].index('n_periods')
        prop[k] = round(prop[k])
        self.proposed += 1
        lp_model = self._prior(prop, MODEL_BLOCK)
        if lp_model > NEG_INF:
            exp = self._evaluate(prop)
            if exp is not None:
                ll = observation_loglik(exp, self.li, self.nuis, self.sent_obs_probs)
                lp = lp_model + self._prior(self.nuis, NUISANCE)
                if math.log(self.rng.random()) < (lp + ll) - (self.lp + self.ll):
                    self.theta, self.expected, self.lp, self.ll = prop, exp, lp, ll
                    self.accepted += 1
        # scalar updates of the observation-model parameters: no model evaluation
        for i, m in enumerate(NUISANCE):
            nu = self.nuis.copy()
            nu[i] += self.rng.normal(0.0, m[3])
            lp = self._prior(self.theta, MODEL_BLOCK) + self._prior(nu, NUISANCE)
            if lp == NEG_INF:
                continue
            ll = observation_loglik(self.expected, self.li, nu, self.sent_obs_probs)
            if math.log(self.rng.random()) < (lp + ll) - (self.lp + self.ll):
                self.nuis, self.lp, self.ll = nu, lp, ll

    def run(self, nsamples):
        trace = np.empty((nsamples, self.theta.size + self.nuis.size))
        logp = np.empty(nsamples)
        t0 = time.perf_counter()
        for n in range(nsamples):
            self.step()
            trace[n] = np.concatenate([self.theta, self.nuis])
            logp[n] = self.lp + self.ll
        dt = time.perf_counter() - t0
        return {'trace': trace, 'logp': logp, 'seconds': dt,
                'samples_per_hour': 3600.0 * nsamples / dt,
                'acceptance': self.accepted / max(1, self.proposed),
                'evaluations': self.n_eval, 'failed_evaluations': self.n_failed,
                'names': [m[0] for m in MODEL_BLOCK] + [m[0] for m in NUISANCE]}


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
    sent_p = np.array([0.1 * 3600.0 / (li.field_sizes[k] * cell_area) for k in li.sent_ids])
    xi, em_p, grid_p = nuis
    li.release_emerg = [rng.poisson(xi * e * (li.release_collection[i] * em_p)[:, None]) for i, e in enumerate(rel)]
    li.sentinel_emerg = [rng.poisson(xi * e * sent_p[:, None]) for e in sen]
    li.grid_obs = rng.poisson(grid_p * li.grid_samples * grid)
    return li
