"""Device-resident evaluation of the population model for given parameters -- the
compute body of the reference's `Bayes_Run.pop_model` (Bayes_Run.py:204-336): scatter the
sampled parameters, build one probability-mass kernel per day (`prob_mass`), re-centre the
release-day kernel, run `get_populations`.  Here the day kernels never leave the GPU: the
model batch hands its COO pool straight to the solver.

The PyMC sampler, the priors and the observation likelihood of Bayes_Run are not part of
this package; "MCMC samples/hour" is measured as evaluations of this
body per hour (bench_bayes.py).
"""
import numpy as np

from . import _lib as L
from . import ParasitoidModel as PM


class PopModel():
    def __init__(self, wind_data, days=None, domain_info=(10000.0, 400), r_number=130000,
                 r_start=None, mode='auto', device=None, max_solvers=4, prob_model=False):
        '''wind_data: dict day -> [T,3] (PM.get_wind_data); r_dur = 1 releases (Kalbar,
        Run.py:126-138).  mode: 'exact' | 'fast' | 'auto' (DESIGN.md section 5).'''
        from . import hip_lib
        self._hip = hip_lib
        self.model = PM.WindModel(wind_data, device=device)
        self.days = list(self.model.days if days is None else days)
        self.rad_dist, self.rad_res = float(domain_info[0]), int(domain_info[1])
        # prob_model: the probability chain of get_solutions (renormalised pmf per day,
        # CalcSol.py:191-201) instead of the population chain of get_populations
        self.prob_model = prob_model
        self.r_number = 1.0 if prob_model else r_number
        self.r_start = r_start
        self.mode = mode
        self.device = device
        self._solvers = {}
        self._max_solvers = max_solvers
        self.solver = None
        self.stats = None

    def close(self):
        for s in self._solvers.values():
            s.close()
        self._solvers = {}
        self.model.close()

    def _solver_for(self, max_shape):
        """A solver whose torus holds kernels up to `max_shape`.  In exact mode the torus is the
        reference's N + max_shape//2, so solvers are keyed by max_shape.  In fast mode any
        torus >= that works: the solver is created for the largest max_shape that still maps
        to the same FFT size, so one solver (plans + ~GBs of buffers) serves every evaluation
        whose kernels fit -- the kernel extent moves with the sampled diffusion parameters."""
        key = int(max_shape)
        N = 2 * self.rad_res + 1
        if self.mode == 'fast':
            lib = L.load()
            pf = int(lib.ps_fast_size(N, key))
            key = 2 * (pf - N) + 1
            if lib.ps_fast_size(N, key) != pf:      # never leave the size class
                key = int(max_shape)
            fits = [k for k in self._solvers if k >= int(max_shape) and k <= key]
            if fits:
                key = min(fits)
        s = self._solvers.pop(key, None)
        if s is None and self.mode in ('auto', 'fold'):
            # a cached fold-mode solver whose FFT size has room for this torus is re-targeted
            # (same plans and buffers) instead of building a new one for every kernel shape
            for k in list(self._solvers):
                c = self._solvers[k]
                # fold: the FFT holds the whole linear convolution; auto: its fast-torus front
                # only has to hold the reference torus (the fold child follows or is rebuilt)
                need = N + (3 if c.mode == 'fold' else 1) * (key // 2)
                if c.mode in ('fold', 'auto') and need <= c.fft_len <= 1.15 * need:
                    s = self._solvers.pop(k)
                    s.retarget(key)
                    break
        if s is None:
            if len(self._solvers) >= self._max_solvers:
                self._solvers.pop(next(iter(self._solvers))).close()
            s = self._hip.HipSolve.from_model(self.model, 0, [key, key], mode=self.mode,
                                              device=self.device, chain_only=True)
        else:
            s.set_state_from_model(self.model, 0)
        self._solvers[key] = s          # most recently used last
        return s

    def evaluate(self, hparams, Dparams, Dlparams, mu_r, n_periods, ndays=None, want_stats=True):
        '''One model evaluation; results stay on the device.  Returns the per-day
        statistics [(nnz, total population above 1e-8), ...], day 0 first -- or None with
        want_stats=False: the chain is then only enqueued (no host synchronisation; a sampler
        that reads nothing but point gathers does not need the statistics, `self.stats` and
        `population()` fetch them on demand).'''
        nd = len(self.days) if ndays is None else ndays
        starts = [self.r_start] + [None] * (nd - 1)
        kshape, nnz, warned, status = self.model.build(
            self.days[:nd], hparams, Dparams, Dlparams, mu_r, n_periods, self.rad_dist,
            self.rad_res, starts)
        for i in range(nd):
            self.model.check(i)
        solver = self._solver_for(int(kshape.max()))
        self.solver = solver
        scale = float(self.r_number)          # dist(1) = 1 for a one-day release
        self._nd = nd
        self._stats = None
        if nd > 1:
            solver.set_kernels_from_model(self.model, 1, nd - 1)
            solver.run_chain(0, nd - 1, negval=1e-8, scale=scale, renorm=self.prob_model)
        if not want_stats:
            return None
        st0 = solver.record_stats(L.REC_STATE, 0, 1e-8, 1.0, False)
        return [(st0.nnz, st0.sum * scale)] + [(s.nnz, s.sum) for s in self.stats]

    @property
    def stats(self):
        '''per-day statistics of the chain days of the last evaluation (synchronises)'''
        if self._stats is None and self.solver is not None and getattr(self, '_nd', 0) > 1:
            self._stats = self.solver.chain_stats(0, self._nd - 1)
        return self._stats if self._stats is not None else []

    @stats.setter
    def stats(self, value):
        self._stats = value

    def population(self, day):
        '''Day `day` (0 = release day) of the last evaluation as a csr matrix, the value
        `get_populations` returns (CalcSol.py:236-237, :322-323).'''
        solver = self.solver
        scale = float(self.r_number)
        if day == 0:
            st = solver.record_stats(L.REC_STATE, 0, 1e-8, 1.0, False)
            return solver._fetch(L.REC_STATE, 0, 1e-8, 1.0, 0.0, scale, st.nnz, 'csr')
        st = self.stats[day - 1]
        return solver._fetch(L.REC_CHAIN, day - 1, 1e-8, scale, st.delta, 1.0, st.nnz, 'csr')

    def moments(self, day):
        '''(total, mean_row, mean_col, var_row, var_col) of one day's raw field, in cells.'''
        solver = self.solver
        kind, idx = (L.REC_STATE, 0) if day == 0 else (L.REC_CHAIN, day - 1)
        raw = solver.dense(kind, idx)
        ix = np.arange(raw.shape[0], dtype=np.float64)
        tot = raw.sum()
        r, c = raw.sum(1), raw.sum(0)
        mr, mc = (r * ix).sum() / tot, (c * ix).sum() / tot
        return (tot * self.r_number, mr, mc, (r * (ix - mr) ** 2).sum() / tot,
                (c * (ix - mc) ** 2).sum() / tot)

    def gather(self, day, rows, cols):
        '''Population density at the given cells of one day (what popdensity_grid /
        popdensity_to_emergence read from the daily solutions, Bayes_funcs.py:20-179).'''
        return self.gather_days([day], rows, cols)[0]

    def gather_days(self, days, rows, cols):
        '''gather() for several days at once -> [len(days), len(rows)] (one device call): the
        values `population(day)` holds at those cells -- thresholded at 1e-8 and, for the
        probability model, with the day's renormalisation delta added to every kept entry
        (CalcSol.py:134-135).'''
        days = list(days)
        kinds = [L.REC_STATE if d == 0 else L.REC_CHAIN for d in days]
        idxs = [0 if d == 0 else d - 1 for d in days]
        out = self.solver.gather_multi(kinds, idxs, rows, cols, scale=float(self.r_number), negval=1e-8)
        if self.prob_model and self.stats:
            for n, d in enumerate(days):
                if d > 0:
                    out[n] = np.where(out[n] != 0.0, out[n] + self.stats[d - 1].delta, 0.0)
        return out
