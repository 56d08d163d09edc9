"""Oracle for the per-day kernel construction (reference: ParasitoidModel.py).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Every function cites the reference lines it restates.  The bivariate-normal
rectangle probability restates the *published* algorithm behind
`scipy.stats.mvn.mvnun` for d=2 (Alan Genz, MVNDST: routines MVNDNT -> BVNMVN
-> BVU; Drezner & Wesolowsky 1990 / Genz 2004 Gauss-Legendre form).  The
Fortran is a SciPy dependency (scipy 1.15.3 in the build container,
`scipy/stats/_mvn*.so`), not part of /root/reference; parity is pinned by the
golden vectors G1/G2 captured from that module (tests/golden/make_golden.py).
"""
import math
import warnings

import numpy as np
from scipy import sparse
from scipy.special import ndtr

# ----------------------------------------------------------------------------
# wind I/O  (ParasitoidModel.py:64-227)
# ----------------------------------------------------------------------------


def read_wind_file(site_name):
    """ParasitoidModel.py:64-126. text 'day wx wy' -> {day: float64[n,3]}, days."""
    wind = {}
    days = []
    with open(site_name + 'wind.txt') as fobj:
        for line in fobj:
            parts = line.split()
            day = int(parts[0])
            wx = float(parts[1])
            wy = float(parts[2])
            if abs(wx) < 10e-5:
                wx = 0
            if abs(wy) < 10e-5:
                wy = 0
            wr = np.sqrt(wx**2 + wy**2)
            if abs(wr) < 10e-5:
                wr = 0
            if day not in wind:
                wind[day] = []
                days.append(day)
            wind[day].append((wx, wy, wr))
    for day in wind:
        wind[day] = np.array(wind[day], dtype=np.float64)
    days.sort()
    return wind, days


def get_wind_data(site_name, interp_num, start_time):
    """ParasitoidModel.py:136-227. Linear interpolation x interp_num."""
    raw, days = read_wind_file(site_name)
    tp = raw[days[0]].shape[0]
    s = np.linspace(0, 1, interp_num + 1)[:-1]
    sm = np.tile(s, (3, 1)).T
    smd = 1 - sm
    out = {}

    def fill(iw, day, shift):
        for k in range(tp - 1):
            iw[(k + shift) * interp_num:(k + shift + 1) * interp_num, :] = (
                smd * raw[day][k, :] + sm * raw[day][k + 1, :])

    if start_time == '00:00':
        for day in days[:-1]:
            iw = np.zeros((tp * interp_num, 3))
            fill(iw, day, 0)
            iw[(tp - 1) * interp_num:, :] = (smd * raw[day][-1, :] +
                                             sm * raw[day + 1][0, :])
            iw[:, 2] = np.sqrt(iw[:, 0]**2 + iw[:, 1]**2)
            out[day] = iw
        day = days[-1]
        iw = np.zeros((tp * interp_num, 3))
        fill(iw, day, 0)
        iw[:, 2] = np.sqrt(iw[:, 0]**2 + iw[:, 1]**2)
        iw[(tp - 1) * interp_num:, :] = raw[day][-1, :]
        out[day] = iw
    elif start_time == '00:30':
        day = days[0]
        iw = np.zeros((tp * interp_num, 3))
        iw[:interp_num, :] = raw[day][0, :]
        fill(iw, day, 1)
        iw[:, 2] = np.sqrt(iw[:, 0]**2 + iw[:, 1]**2)
        out[day] = iw
        for day in days[1:]:
            iw = np.zeros((tp * interp_num, 3))
            iw[:interp_num, :] = (smd * raw[day - 1][-1, :] +
                                  sm * raw[day][0, :])
            fill(iw, day, 1)
            iw[:, 2] = np.sqrt(iw[:, 0]**2 + iw[:, 1]**2)
            out[day] = iw
    else:
        raise ValueError("start_time must be either '00:00' or '00:30'")
    return out, days


# ----------------------------------------------------------------------------
# take-off probability  (ParasitoidModel.py:231-309)
# ----------------------------------------------------------------------------


def g_wind_prob(windr, aw, bw):
    """ParasitoidModel.py:231-240."""
    return 1.0 / (1. + np.exp(bw * (windr - aw)))


def f_time_prob(n, a1, b1, a2, b2):
    """ParasitoidModel.py:243-267."""
    t = np.linspace(0, 24 - 24. / n, n)
    lik = np.fmax(1.0 / (1. + np.exp(-b1 * (t - a1))) -
                  1.0 / (1. + np.exp(-b2 * (t - a2))), np.zeros_like(t))
    return lik / lik.sum()


def Dmat(sig_x, sig_y, rho):
    """ParasitoidModel.py:269-280."""
    assert sig_x > 0, 'sig_x must be positive'
    assert sig_y > 0, 'sig_y must be positive'
    assert -1 <= rho <= 1, 'correlation must be between -1 and 1'
    return np.array([[sig_x**2, rho * sig_x * sig_y],
                     [rho * sig_x * sig_y, sig_y**2]])


def h_flight_prob(day_wind, lam, aw, bw, a1, b1, a2, b2):
    """ParasitoidModel.py:282-309."""
    n = day_wind.shape[0]
    try:
        windr = day_wind[:, 2]
    except IndexError:
        windr = day_wind[2]
        n = 1
    f = f_time_prob(n, a1, b1, a2, b2)
    g = g_wind_prob(windr, aw, bw)
    t = np.linspace(1, n, n)
    integral_avg = f * g / t / np.max(f) * np.cumsum(
        (1 - np.cumsum(f)) * (f - f * g))
    return lam * (f * g + integral_avg)


# ----------------------------------------------------------------------------
# bivariate normal rectangle probability (Genz MVNDST, d = 2)
# ----------------------------------------------------------------------------

# Gauss-Legendre half-rules (weights, abscissae) for N = 6, 12, 20, as published
# with Genz's BVU/BVND.
_GL = {
    1: (np.array([0.1713244923791705, 0.3607615730481384, 0.4679139345726904]),
        np.array([-0.9324695142031522, -0.6612093864662647,
                  -0.2386191860831970])),
    2: (np.array([0.4717533638651177e-01, 0.1069393259953183,
                  0.1600783285433464, 0.2031674267230659, 0.2334925365383547,
                  0.2491470458134029]),
        np.array([-0.9815606342467191, -0.9041172563704750,
                  -0.7699026741943050, -0.5873179542866171,
                  -0.3678314989981802, -0.1252334085114692])),
    3: (np.array([0.1761400713915212e-01, 0.4060142980038694e-01,
                  0.6267204833410906e-01, 0.8327674157670475e-01,
                  0.1019301198172404, 0.1181945319615184, 0.1316886384491766,
                  0.1420961093183821, 0.1491729864726037, 0.1527533871307259]),
        np.array([-0.9931285991850949, -0.9639719272779138,
                  -0.9122344282513259, -0.8391169718222188,
                  -0.7463319064601508, -0.6360536807265150,
                  -0.5108670019508271, -0.3737060887154196,
                  -0.2277858511416451, -0.7652652113349733e-01])),
}
_TWOPI = 6.283185307179586


def gl_rule(r):
    """Rule selection of BVU: |r|<0.3 -> 6 pt, <0.75 -> 12 pt, else 20 pt."""
    if abs(r) < 0.3:
        return _GL[1]
    if abs(r) < 0.75:
        return _GL[2]
    return _GL[3]


def bvu(sh, sk, r):
    """P(X > sh, Y > sk) for a standard bivariate normal with correlation r.

    Restates Genz's BVU (MVNDST).  `sh`, `sk` broadcastable arrays, `r` scalar.
    """
    h = np.asarray(sh, dtype=np.float64)
    k = np.asarray(sk, dtype=np.float64)
    h, k = np.broadcast_arrays(h, k)
    w, x = gl_rule(r)
    hk = h * k
    if abs(r) < 0.925:
        hs = (h * h + k * k) / 2
        asr = math.asin(r)
        bvn = np.zeros_like(hk)
        for wi, xi in zip(w, x):
            sn = math.sin(asr * (xi + 1) / 2)
            bvn = bvn + wi * np.exp((sn * hk - hs) / (1 - sn * sn))
            sn = math.sin(asr * (-xi + 1) / 2)
            bvn = bvn + wi * np.exp((sn * hk - hs) / (1 - sn * sn))
        return bvn * asr / (2 * _TWOPI) + ndtr(-h) * ndtr(-k)
    # |r| >= 0.925 branch
    k = np.array(k, copy=True)
    hk = np.array(hk, copy=True)
    if r < 0:
        k = -k
        hk = -hk
    bvn = np.zeros_like(hk)
    if abs(r) < 1:
        as_ = (1 - r) * (1 + r)
        a = math.sqrt(as_)
        bs = (h - k)**2
        c = (4 - hk) / 8
        d = (12 - hk) / 16
        with np.errstate(under='ignore', over='ignore', invalid='ignore',
                         divide='ignore'):
            bvn = a * np.exp(-(bs / as_ + hk) / 2) * (
                1 - c * (bs - as_) * (1 - d * bs / 5) / 3 + c * d * as_ * as_ / 5)
            b = np.sqrt(bs)
            corr = np.exp(-hk / 2) * math.sqrt(_TWOPI) * ndtr(-b / a) * b * (
                1 - c * bs * (1 - d * bs / 5) / 3)
            bvn = np.where(hk > -160, bvn - corr, bvn)
            a2 = a / 2
            for wi, xi in zip(w, x):
                xs = (a2 + a2 * xi)**2
                rs = math.sqrt(1 - xs)
                bvn = bvn + a2 * wi * (
                    np.exp(-bs / (2 * xs) - hk / (1 + rs)) / rs -
                    np.exp(-(bs / xs + hk) / 2) * (1 + c * xs * (1 + d * xs)))
                xs = as_ * (-xi + 1)**2 / 4
                rs = math.sqrt(1 - xs)
                bvn = bvn + a2 * wi * np.exp(-(bs / xs + hk) / 2) * (
                    np.exp(-hk * (1 - rs) / (2 * (1 + rs))) / rs -
                    (1 + c * xs * (1 + d * xs)))
            bvn = -bvn / _TWOPI
    if r > 0:
        bvn = bvn + ndtr(-np.maximum(h, k))
    if r < 0:
        bvn = -bvn + np.maximum(0.0, ndtr(-h) - ndtr(-k))
    return bvn


def mvn_rect(low, upp, mu, S):
    """Rectangle probability of N(mu,S) on [low,upp] (d=2), as `mvnun` computes
    it: standardise by the marginal std devs, then BVNMVN with both limits
    finite = BVU(l1,l2) - BVU(u1,l2) - BVU(l1,u2) + BVU(u1,u2).

    low/upp: arrays [...,2]; returns array [...].
    """
    low = np.asarray(low, dtype=np.float64)
    upp = np.asarray(upp, dtype=np.float64)
    sd0 = math.sqrt(S[0][0])
    sd1 = math.sqrt(S[1][1])
    r = S[0][1] / sd0 / sd1
    l0 = (low[..., 0] - mu[0]) / sd0
    l1 = (low[..., 1] - mu[1]) / sd1
    u0 = (upp[..., 0] - mu[0]) / sd0
    u1 = (upp[..., 1] - mu[1]) / sd1
    return bvu(l0, l1, r) - bvu(u0, l1, r) - bvu(l0, u1, r) + bvu(u0, u1, r)


def get_mvn_cdf_values(cell_length, mu, S):
    """ParasitoidModel.py:311-380.  Same cells, same shell order, same running
    sum (sequential float64 adds) for the 1e-3 support rule."""
    cdf_eps = 0.001
    r = cell_length / 2
    cl = np.array([cell_length, cell_length])
    vals = {}
    low = np.array([-r, -r])
    upp = np.array([r, r])
    v = float(mvn_rect(low, upp, mu, S))
    vals[(0, 0)] = v
    val_sum = v
    h = 0
    while 1 - val_sum >= cdf_eps:
        h += 1
        cells = []
        for ii in (-h, h):
            for jj in (-h, h):
                cells.append((ii, jj))
        for ii in (-h, h):
            for jj in range(-h + 1, h):
                cells.append((ii, jj))
                cells.append((jj, ii))
        idx = np.array(cells, dtype=np.float64)
        lo = idx * cell_length - r
        up = lo + cl
        v = mvn_rect(lo, up, mu, S)
        for c, vv in zip(cells, v):
            vals[c] = vv
        # sequential accumulation in the reference's order (:359,:369,:373)
        val_sum = np.cumsum(np.concatenate(([val_sum], v)))[-1]
    return np.array([[vals[(x, y)] for x in range(-h, h + 1)]
                     for y in range(h, -h - 1, -1)])


# ----------------------------------------------------------------------------
# small-value removal (CalcSol.py:112-136) -- shared by both oracle modules
# ----------------------------------------------------------------------------


def r_small_vals(A, prob_model=False, negval=1e-8):
    """CalcSol.py:112-136 (the Python loop is a mask `not (val < negval)`)."""
    if not sparse.isspmatrix_coo(A):
        A = sparse.coo_matrix(A)
    mask = ~(A.data < negval)
    A_red = sparse.coo_matrix((A.data[mask], (A.row[mask], A.col[mask])),
                              A.shape)
    if prob_model:
        A_red.data += (1 - A_red.data.sum()) / A_red.data.size
    return A_red


# ----------------------------------------------------------------------------
# per-day probability mass kernel (ParasitoidModel.py:384-613)
# ----------------------------------------------------------------------------


def period_mu_v(day, wind_data, t_indx, n_periods, periods, test_run):
    """Advection velocity for one period, m/s  (ParasitoidModel.py:439-465)."""
    day_wind = wind_data[day]
    if (not test_run) and n_periods > 1:
        if t_indx + n_periods - 1 < periods:
            return np.sum(day_wind[t_indx:t_indx + n_periods, 0:2], 0) / n_periods
        if day + 1 in wind_data:
            if t_indx != periods - 1:
                mu_v = np.sum(day_wind[t_indx:, 0:2], 0)
            else:
                mu_v = np.array(day_wind[-1, 0:2])
            wrap = n_periods - (periods - t_indx)
            if wrap != 1:
                mu_v += np.sum(wind_data[day + 1][:wrap, 0:2], 0)
            else:
                mu_v += wind_data[day + 1][0, 0:2]
            mu_v /= n_periods
            return mu_v
        if t_indx != periods - 1:
            return np.sum(day_wind[t_indx:, 0:2], 0) / (periods - t_indx)
        return np.array(day_wind[-1, 0:2])
    if not test_run:
        return np.array(day_wind[t_indx, 0:2])
    return np.array(day_wind[0:2])


def prob_mass(day, wind_data, hparams, Dparams, Dlparams, mu_r, n_periods,
              rad_dist, rad_res, start_time=None, return_debug=False):
    """ParasitoidModel.py:384-613 (same statement order, silent)."""
    dom_len = rad_res * 2 + 1
    cell_dist = rad_dist / rad_res
    pmf = np.zeros((dom_len, dom_len))
    day_wind = wind_data[day]
    hprob = h_flight_prob(day_wind, *hparams)
    S = Dmat(*Dparams)
    Sl = Dmat(*Dlparams)
    loss = 0.0
    if day_wind.ndim > 1:
        periods = day_wind.shape[0]
        test_run = False
    else:
        periods = 1
        test_run = True
        hprob = np.atleast_1d(hprob)
    start_indx = 0 if start_time is None else math.floor(start_time * periods)
    warned = False
    dbg = {'H': [], 'cent': []}
    for t in range(start_indx, periods):
        mu_v = period_mu_v(day, wind_data, t, n_periods, periods, test_run)
        mu_v = mu_v * (3600 * 24 * (n_periods / periods))
        mu_v = mu_v * mu_r
        cdf_mu = mu_v - np.round(mu_v / cell_dist) * cell_dist
        cdf_mat = get_mvn_cdf_values(cell_dist, cdf_mu, S)
        col_offset = int(np.round(mu_v[0] / cell_dist))
        row_offset = int(np.round(-mu_v[1] / cell_dist))
        row_cent = rad_res + row_offset
        col_cent = rad_res + col_offset
        norm_r = int(cdf_mat.shape[0] / 2)
        dbg['H'].append(norm_r)
        dbg['cent'].append((row_cent, col_cent))
        row_min, col_min = row_cent - norm_r, col_cent - norm_r
        row_max, col_max = row_cent + norm_r, col_cent + norm_r
        rs, re = 0, cdf_mat.shape[0]
        cs, ce = 0, cdf_mat.shape[1]
        if row_max + 1 > dom_len:
            re = max(0, re - (row_max + 1 - dom_len))
            row_max = dom_len - 1
        if col_max + 1 > dom_len:
            ce = max(ce - (col_max + 1 - dom_len), 0)
            col_max = dom_len - 1
        if row_min < 0:
            rs = max(rs - row_min, 0)
            row_min = 0
        if col_min < 0:
            cs = max(cs - col_min, 0)
            col_min = 0
        assert -1e-9 <= hprob[t] <= 1.000000001, \
            'hprob out of bounds at t_indx {}'.format(t)
        try:
            pmf[row_min:row_max + 1, col_min:col_max + 1] += (
                hprob[t] * cdf_mat[rs:re, cs:ce])
            if rs > 0 or re < cdf_mat.shape[0] or cs > 0 or \
                    ce < cdf_mat.shape[1]:
                loss += (1 - cdf_mat[rs:re, cs:ce].sum()) * hprob[t]
        except ValueError:
            if not warned:
                warnings.warn('Index error in calculating prob_mass.',
                              RuntimeWarning)
                warned = True
            loss += hprob[t]

    pmfsum = pmf.sum()
    total = pmfsum + loss
    assert loss >= 0.0, 'negative loss'
    assert pmf.min() >= -1e-8, 'pmf.min() less than zero, first block'
    assert pmfsum <= 1.00001, 'flight prob > 1, first block'
    if total < 0.99999:
        cdf_mat = get_mvn_cdf_values(cell_dist, np.array([0., 0.]), Sl)
        nr = int(cdf_mat.shape[0] / 2)
        pmf[rad_res - nr:rad_res + nr + 1, rad_res - nr:rad_res + nr + 1] += \
            (1 - total) * cdf_mat
        total = pmf.sum() + loss
        assert pmf.min() >= -1e-8, 'pmf.min() less than zero'
        assert total <= 1.00001, 'flight prob > 1'
    coo = r_small_vals(sparse.coo_matrix(pmf), prob_model=True)
    I, J, V = coo.row, coo.col, coo.data
    rad = int(max(np.fabs(I - rad_res).max(), np.fabs(J - rad_res).max()))
    out = sparse.coo_matrix((V, (I - rad_res + rad, J - rad_res + rad)),
                            shape=(rad * 2 + 1, rad * 2 + 1))
    if return_debug:
        dbg['hprob'] = hprob
        dbg['loss'] = loss
        dbg['pmfsum'] = pmfsum
        return out, dbg
    return out
