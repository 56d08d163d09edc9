"""Drift-diffusion model pieces on the MI355X -- counterpart of the reference's
`ParasitoidModel.py`, same function names and argument meaning.

Wind file parsing/interpolation stays on the host (it is I/O, done once per run,
ParasitoidModel.py:64-227).  Everything on the hot path -- `h_flight_prob`,
`get_mvn_cdf_values` and `prob_mass` -- runs on the device through
libparasitoid_hip.so; nothing here falls back to a CPU implementation.

`prob_mass` keeps the reference's one-day signature (it is what
`pool.starmap(PM.prob_mass, ...)` calls, Run.py:423).  `prob_mass_batch` builds
many days in one launch sequence, which is what the GPU wants; `Run.main` uses it.
"""
import ctypes as C
import os
import warnings
from math import floor

import numpy as np
from scipy import sparse

from . import _lib as L

# --------------------------------------------------------------------------- wind

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))


def data_path(site_name, suffix='wind.txt'):
    """`site_name` as the reference spells it ('data/kalbar', relative to the directory the
    reference is run from, Run.py:96-138): used as given when that file exists, otherwise
    resolved against this package, which ships the reference's data files under
    parasitoids_amd/data/."""
    if os.path.exists(site_name + suffix) or os.path.isabs(site_name):
        return site_name
    cand = os.path.join(PACKAGE_DIR, site_name)
    return cand if os.path.exists(cand + suffix) else site_name


def read_wind_file(site_name):
    """Reads `<site_name>wind.txt` (columns: day windx windy [% comment]).

    Returns (wind_data, days): dict day -> float64[n,3] (windx, windy, windr) and the
    sorted list of days.  Components with magnitude < 1e-4 are zeroed
    (ParasitoidModel.py:64-126)."""
    rows = []
    with open(data_path(site_name) + 'wind.txt') as fobj:
        for line in fobj:
            parts = line.split()
            if parts:
                rows.append((int(parts[0]), float(parts[1]), float(parts[2])))
    arr = np.array(rows, dtype=np.float64)
    day_col = arr[:, 0].astype(int)
    wx = np.where(np.abs(arr[:, 1]) < 10e-5, 0.0, arr[:, 1])
    wy = np.where(np.abs(arr[:, 2]) < 10e-5, 0.0, arr[:, 2])
    wr = np.sqrt(wx**2 + wy**2)
    wr = np.where(np.abs(wr) < 10e-5, 0.0, wr)
    days = sorted(set(day_col.tolist()))
    wind_data = {d: np.column_stack((wx, wy, wr))[day_col == d] for d in days}
    return wind_data, days


def get_wind_data(site_name, interp_num, start_time):
    """Linear interpolation of the 30-minute wind samples to `interp_num` points per
    interval with the '00:00' / '00:30' day conventions (ParasitoidModel.py:136-227).

    Returns (wind_data, days): dict day -> float64[48*interp_num, 3]."""
    raw, days = read_wind_file(site_name)
    tp = raw[days[0]].shape[0]
    s = np.linspace(0, 1, interp_num + 1)[:-1][:, None]        # (interp_num, 1)
    d = 1 - s

    def blend(a, b):
        # (k, 3), (k, 3) -> (k*interp_num, 3): (1-s)*a + s*b for every sample pair
        return (d[None, :, :] * a[:, None, :] + s[None, :, :] * b[:, None, :]).reshape(-1, 3)

    def windr(iw):
        iw[:, 2] = np.sqrt(iw[:, 0]**2 + iw[:, 1]**2)

    out = {}
    if start_time == '00:00':
        for n, day in enumerate(days):
            iw = np.zeros((tp * interp_num, 3))
            iw[:(tp - 1) * interp_num] = blend(raw[day][:-1], raw[day][1:])
            if n < len(days) - 1:
                iw[(tp - 1) * interp_num:] = blend(raw[day][-1:], raw[day + 1][:1])
                windr(iw)
            else:
                windr(iw)
                iw[(tp - 1) * interp_num:] = raw[day][-1]    # repeat the last sample
            out[day] = iw
    elif start_time == '00:30':
        for n, day in enumerate(days):
            iw = np.zeros((tp * interp_num, 3))
            if n == 0:
                iw[:interp_num] = raw[day][0]                 # repeat the first sample backwards
            else:
                iw[:interp_num] = blend(raw[day - 1][-1:], raw[day][:1])
            iw[interp_num:] = blend(raw[day][:-1], raw[day][1:])
            windr(iw)
            out[day] = iw
    else:
        raise ValueError("start_time must be either '00:00' or '00:30'")
    return out, days


def emergence_data(site_name):
    """Observed emergence counts `<site_name>emergence.txt` -> {field: {day: count}}
    (ParasitoidModel.py:28-60)."""
    with open(data_path(site_name, 'emergence.txt') + 'emergence.txt') as fobj:
        fields = fobj.readline().split()[1:]
        em = {f: {} for f in fields}
        for line in fobj:
            parts = line.split()
            if not parts:
                continue
            date = int(parts[0])
            for f, v in zip(fields, parts[1:]):
                em[f][date] = int(v)
    return em


# ------------------------------------------------------------ closed-form pieces

def g_wind_prob(windr, aw, bw):
    """Take-off scaling by wind speed, a decreasing logistic (ParasitoidModel.py:231-240).
    Host helper for inspection/plots; the device evaluates it inside `h_flight_prob`."""
    return 1.0 / (1. + np.exp(bw * (windr - aw)))


def f_time_prob(n, a1, b1, a2, b2):
    """Take-off pmf over n equally spaced times of day (ParasitoidModel.py:243-267).
    Host helper for inspection/plots; the device evaluates it inside `h_flight_prob`."""
    t = np.linspace(0, 24 - 24. / n, n)
    lik = np.fmax(1.0 / (1. + np.exp(-b1 * (t - a1))) - 1.0 / (1. + np.exp(-b2 * (t - a2))), 0.0)
    return lik / lik.sum()


def Dmat(sig_x, sig_y, rho):
    """2x2 diffusion covariance (ParasitoidModel.py:269-280)."""
    assert sig_x > 0, 'sig_x must be positive'
    assert sig_y > 0, 'sig_y must be positive'
    assert -1 <= rho <= 1, 'correlation must be between -1 and 1'
    return np.array([[sig_x**2, rho * sig_x * sig_y], [rho * sig_x * sig_y, sig_y**2]])


# ------------------------------------------------------------------ device model

class WindModel():
    """Wind data resident on one GPU + the kernels built from it."""

    def __init__(self, wind_data, device=None):
        L.require_device()
        self._lib = L.load()
        self._h = L._VP()
        self.days = sorted(wind_data.keys())
        first = np.asarray(wind_data[self.days[0]])
        self.test_run = first.ndim == 1
        if self.test_run:
            arr = np.stack([np.asarray(wind_data[d], dtype=np.float64).reshape(1, 3)
                            for d in self.days])
        else:
            arr = np.stack([np.asarray(wind_data[d], dtype=np.float64) for d in self.days])
        self.T = arr.shape[1]
        self._index = {d: i for i, d in enumerate(self.days)}
        dev = L.default_device() if device is None else device
        L.check(self._lib.ps_model_create(C.byref(self._h), dev))
        arr = L.f64(arr)
        keys = L.i32(self.days)
        L.check(self._lib.ps_model_set_wind(self._h, L.p_f64(arr), L.p_i32(keys), len(self.days),
                                            self.T, int(self.test_run)))
        self.last = None

    def set_option(self, key, value):
        '''PS_PM_SEG / PS_PM_SYNC of this model handle (read from the environment when it was created)'''
        L.check(self._lib.ps_model_set_option(self._h, key.encode(), float(value)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.ps_model_destroy(self._h)
            self._h = L._VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def h_flight_prob(self, day, lam, aw, bw, a1, b1, a2, b2):
        hp = L.f64([lam, aw, bw, a1, b1, a2, b2])
        out = np.empty(self.T)
        L.check(self._lib.ps_model_hflight(self._h, self._index[day], L.p_f64(hp), L.p_f64(out)))
        return out

    def build(self, days, hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res,
              start_times=None):
        """Run the device pipeline for `days`; results stay on the device.
        Returns per-day (kshape, nnz, warned, status) arrays."""
        nd = len(days)
        idx = L.i32([self._index[d] for d in days])
        st = np.full(nd, -1.0)
        if start_times is not None:
            for i, v in enumerate(start_times):
                if v is not None:
                    st[i] = v
        st = L.f64(st)
        hp, dp, dl = L.f64(hparams), L.f64(Dparams), L.f64(Dlparams)
        kshape = np.zeros(nd, dtype=np.int32)
        nnz = np.zeros(nd, dtype=np.int64)
        warned = np.zeros(nd, dtype=np.int32)
        status = np.zeros(nd, dtype=np.int32)
        L.check(self._lib.ps_model_prob_mass(
            self._h, nd, L.p_i32(idx), L.p_f64(st), L.p_f64(hp), L.p_f64(dp), L.p_f64(dl),
            float(mu_r), int(n_periods), float(rad_dist), int(rad_res),
            L.p_i32(kshape), L.p_i64(nnz), L.p_i32(warned), L.p_i32(status)))
        self.last = dict(days=list(days), kshape=kshape, nnz=nnz, warned=warned, status=status,
                         args=(hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res))
        return kshape, nnz, warned, status

    def check(self, i):
        """Raise / warn like the reference for day i of the last batch."""
        last = self.last
        day = last['days'][i]
        hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res = last['args']
        st = int(last['status'][i])
        if st != 0:
            what = {L.PS_ERR_HPROB_BOUNDS: 'hprob out of bounds',
                    L.PS_ERR_PMF_NEGATIVE: 'pmf.min() less than zero',
                    L.PS_ERR_FLIGHT_PROB: 'flight prob > 1 or negative loss',
                    L.PS_ERR_EMPTY: 'no probability mass above 1e-8'}.get(st, 'error %d' % st)
            raise AssertionError(
                what, 'day={}'.format(day), 'hparams={}'.format(hparams),
                'Dparams={}'.format(Dparams), 'Dlparams={}'.format(Dlparams),
                'mu_r={}'.format(mu_r), 'n_periods={}'.format(n_periods),
                'rad_dist={}'.format(rad_dist), 'rad_res={}'.format(rad_res))
        if last['warned'][i]:
            warnings.warn('Index error in calculating prob_mass.\nDay: {}\n'.format(day) +
                          'Wind advection during this period appears to be greater'
                          ' than the size of the domain.\n'
                          'Wasps flying during this time will be considered lost.',
                          RuntimeWarning)

    def fetch(self, i):
        """Day i of the last batch as a scipy coo matrix (shape K x K)."""
        self.check(i)
        n = int(self.last['nnz'][i])
        K = int(self.last['kshape'][i])
        row = np.empty(max(n, 1), dtype=np.int32)
        col = np.empty(max(n, 1), dtype=np.int32)
        val = np.empty(max(n, 1), dtype=np.float64)
        L.check(self._lib.ps_model_fetch_coo(self._h, i, L.p_i32(row), L.p_i32(col), L.p_f64(val),
                                             max(n, 1)))
        return sparse.coo_matrix((val[:n], (row[:n], col[:n])), shape=(K, K))

    def debug(self, i):
        hprob = np.empty(self.T)
        Hs = np.empty(self.T, dtype=np.int32)
        loss = C.c_double()
        pmfsum = C.c_double()
        L.check(self._lib.ps_model_fetch_debug(self._h, i, L.p_f64(hprob), L.p_i32(Hs),
                                               C.byref(loss), C.byref(pmfsum)))
        return dict(hprob=hprob, H=Hs, loss=loss.value, pmfsum=pmfsum.value)

    def mvn_cdf_values(self, cell_length, mu, S):
        S = np.asarray(S, dtype=np.float64)
        sx, sy = np.sqrt(S[0, 0]), np.sqrt(S[1, 1])
        rho = S[0, 1] / sx / sy
        H = C.c_int32()
        cap = 64 * 64
        while True:
            out = np.empty(cap)
            L.check(self._lib.ps_model_mvn_cdf_values(self._h, float(cell_length), float(mu[0]),
                                                      float(mu[1]), float(sx), float(sy), float(rho),
                                                      C.byref(H), L.p_f64(out), cap))
            side = 2 * H.value + 1
            if side * side <= cap:
                return out[:side * side].reshape(side, side).copy()
            cap = side * side


_cache = {'key': None, 'model': None}


def _model_for(wind_data):
    """One device-resident copy of the wind per process, re-uploaded when it changes."""
    days = sorted(wind_data.keys())
    first = np.asarray(wind_data[days[0]])
    last = np.asarray(wind_data[days[-1]])
    key = (id(wind_data), len(days), first.shape, float(first.sum()), float(last.sum()))
    if _cache['key'] != key:
        if _cache['model'] is not None:
            _cache['model'].close()
        _cache['model'] = WindModel(wind_data)
        _cache['key'] = key
    return _cache['model']


def h_flight_prob(day_wind, lam, aw, bw, a1, b1, a2, b2):
    """Probability of take-off at each time of the day, lam*(f*g + carry-over)
    (ParasitoidModel.py:282-309), evaluated on the device."""
    day_wind = np.asarray(day_wind, dtype=np.float64)
    model = WindModel({0: day_wind})
    try:
        return model.h_flight_prob(0, lam, aw, bw, a1, b1, a2, b2)
    finally:
        model.close()


def get_mvn_cdf_values(cell_length, mu, S):
    """Cell masses of N(mu, S) on the cell lattice out to the support that holds
    all but 1e-3 of the mass (ParasitoidModel.py:311-380), evaluated on the device."""
    model = _cache['model']
    own = model is None
    if own:
        model = WindModel({0: np.zeros((1, 3))})
    try:
        return model.mvn_cdf_values(cell_length, mu, S)
    finally:
        if own:
            model.close()


def prob_mass_batch(days, wind_data, hparams, Dparams, Dlparams, mu_r, n_periods,
                    rad_dist, rad_res, start_times=None, model=None):
    """`prob_mass` for a list of days in one device batch -> list of coo matrices."""
    model = _model_for(wind_data) if model is None else model
    model.build(days, hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res, start_times)
    return [model.fetch(i) for i in range(len(days))]


def prob_mass(day, wind_data, hparams, Dparams, Dlparams, mu_r, n_periods,
              rad_dist, rad_res, start_time=None):
    """Returns the probability mass function of one day's spread from the origin as a
    shrunk sparse (coo) array, ParasitoidModel.py:384-613.

    Arguments:
        - day -- day as specified in wind data
        - wind_data -- dictionary of wind data (units: m/s)
        - hparams -- (lam,aw,bw,a1,b1,a2,b2)
        - Dparams -- in-flow diffusion (sig_x,sig_y,rho)
        - Dlparams -- out-of-flow diffusion (sig_x,sig_y,rho)
        - mu_r -- scaling of flight advection to wind advection
        - n_periods -- number of time periods in one flight (int)
        - rad_dist -- distance from release point to side of the domain (m)
        - rad_res -- number of cells from center to side of the domain
        - start_time -- (optional) release time as a fraction of the day"""
    return prob_mass_batch([day], wind_data, hparams, Dparams, Dlparams, mu_r, n_periods,
                           rad_dist, rad_res, [start_time])[0]
