#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Run in the build container only (it imports /root/reference, which does not
exist on the GPU box):

    python tests/golden/make_golden.py [group ...]     # groups: g1 .. g9, g5b, g6b

Three harness-side shims, none of which touch the
reference: (1) scipy 1.15 no longer exposes `scipy.stats.mvn`; the same Genz
Fortran wrapper lives at `scipy.stats._mvn`, registered here under the old
name; (2) stdout of `prob_mass` (1440 prints per day) is discarded; (3) the
reference is run from a scratch cwd with a `data/` symlink.

Fixtures are data only: inputs and the reference's outputs.
"""
import contextlib
import io
import os
import sys
import types
import warnings
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def _shim():
    import scipy.stats
    import scipy.stats._mvn as _mvn
    m = types.ModuleType('scipy.stats.mvn')
    m.mvnun = _mvn.mvnun
    sys.modules['scipy.stats.mvn'] = m
    scipy.stats.mvn = m
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    scratch = '/tmp/parasitoid_golden_cwd'
    os.makedirs(scratch, exist_ok=True)
    if not os.path.exists(os.path.join(scratch, 'data')):
        os.symlink(os.path.join(REF, 'data'), os.path.join(scratch, 'data'))
    os.chdir(scratch)


_shim()
import CalcSol as CS            # noqa: E402  (the reference)
import ParasitoidModel as PM    # noqa: E402  (the reference)
from scipy import sparse        # noqa: E402

warnings.simplefilter('ignore')

# default model parameters, Run.py:68-83
HP = (1., 1.263, 3.913, 7.302, 2.614, 23.999, 2.350)
DP = (171.82, 144.58, 0.253)
DLP = (7.096, 7.260, 0.000)
MU_R = 1.179
NPER = 30
# reference test parameters, tests/test_ParsitoidModel.py:24-56
HP_T = (1.0, 1.8, 6, 7., 2., 19., 2.)
DP_T = (4.0, 4.0, 0.)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _pm(args):
    return quiet(PM.prob_mass, *args)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print('wrote', path, os.path.getsize(path + '.npz') // 1024, 'KiB')


def coo_pack(prefix, M, out):
    M = M.tocoo()
    out[prefix + '_row'] = M.row.astype(np.int32)
    out[prefix + '_col'] = M.col.astype(np.int32)
    out[prefix + '_val'] = M.data.astype(np.float64)
    out[prefix + '_shape'] = np.array(M.shape, dtype=np.int64)


def sample_positions(n, count, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, n, size=(count, 2))


def summarize(prefix, M, out, pos):
    """nnz, sum, weighted checksum and sampled entries of a big sparse field."""
    C = M.tocsr()
    D = M.tocoo()
    out[prefix + '_nnz'] = np.int64(D.nnz)
    out[prefix + '_sum'] = np.float64(D.data.sum())
    wt = 1.0 + ((D.row.astype(np.int64) * 31 + D.col.astype(np.int64) * 17) % 97)
    out[prefix + '_wsum'] = np.float64((D.data * wt).sum())
    out[prefix + '_samp'] = np.asarray(C[pos[:, 0], pos[:, 1]]).ravel()


# ---------------------------------------------------------------------------

def g1():
    """mvnun rectangle probabilities."""
    rng = np.random.default_rng(1)
    n = 2400
    rhos = np.array([0.0, 0.253, -0.253, 0.5, -0.5, 0.9, -0.9, 0.3, 0.75,
                     0.93, -0.93, 0.99, -0.99, 0.9249, 0.925])
    low = np.empty((n, 2)); upp = np.empty((n, 2)); mu = np.empty((n, 2))
    S = np.empty((n, 2, 2)); val = np.empty(n)
    for i in range(n):
        sx, sy = rng.uniform(3, 200, 2)
        rho = rhos[i % len(rhos)] if i % 3 else rng.uniform(-0.999, 0.999)
        S[i] = PM.Dmat(sx, sy, rho)
        mu[i] = rng.uniform(-20, 20, 2)
        width = rng.choice([2.0, 4.8828125, 19.53125, 25.0, 78.125])
        # centres out to ~7 sigma so tails down to ~1e-12 are covered
        cx = rng.normal(0, 2.5) * sx
        cy = rng.normal(0, 2.5) * sy
        low[i] = (cx - width / 2, cy - width / 2)
        upp[i] = low[i] + width
        v, inform = PM.mvn.mvnun(low[i], upp[i], mu[i], S[i])
        assert inform == 0
        val[i] = v
    save('g1_mvnun', low=low, upp=upp, mu=mu, S=S, val=val)


def g2():
    """get_mvn_cdf_values stamps."""
    out = {}
    cases = []
    k = 0
    for c in (78.125, 25.0, 19.53125):
        for fx, fy in ((0, 0), (0.5, -0.5), (-0.25, 0.3), (0.49, 0.49)):
            cases.append((c, (fx * c, fy * c), DP))
    cases.append((9.765625, (1.0, -2.0), DP))
    cases.append((25.0, (0., 0.), DLP))
    cases.append((78.125, (0., 0.), DLP))
    cases.append((2.0, (0., 0.), (4., 4., 0.5)))
    cases.append((2.0, (0., 0.), (10., 10., -0.5)))
    cases.append((25.0, (3., -7.), (120., 90., 0.95)))
    cases.append((25.0, (3., -7.), (120., 90., -0.8)))
    for c, mu, dp in cases:
        mat = PM.get_mvn_cdf_values(c, np.array(mu), PM.Dmat(*dp))
        out['c%d' % k] = np.array([c, mu[0], mu[1], *dp])
        out['m%d' % k] = mat
        k += 1
    out['n'] = np.int64(k)
    save('g2_stamps', **out)


def g3():
    """h_flight_prob + wind interpolation (G3, G4)."""
    out = {}
    for site, st in (('kalbar', '00:00'), ('carnarvonearl', '00:30')):
        wd, days = PM.get_wind_data('data/' + site, 30, st)
        out[site + '_days'] = np.array(days)
        out[site + '_wind_sum'] = np.array([wd[d].sum(0) for d in days])
        out[site + '_wind_first'] = wd[days[0]]
        out[site + '_wind_last'] = wd[days[-1]]
        out[site + '_wind_mid'] = wd[days[3]]
        out[site + '_h_def'] = np.array(
            [PM.h_flight_prob(wd[d], *HP) for d in days[:8]])
        out[site + '_h_test'] = np.array(
            [PM.h_flight_prob(wd[d], *HP_T) for d in days[:4]])
    save('g3_hprob_wind', **out)


def g5():
    """prob_mass COO kernels."""
    out = {}
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    args = [(d, wd, HP, DP, DLP, MU_R, NPER, 10000.0, 128) for d in days[:6]]
    args += [(d, wd, HP, DP, DLP, MU_R, NPER, 10000.0, 400) for d in days[:2]]
    wc, dc = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    args.append((dc[0], wc, HP, DP, DLP, MU_R, NPER, 10000.0, 128, 0.354))
    # strong advection, small domain: clipped + fully-outside periods
    args.append((days[1], wd, HP, DP, DLP, 6.0, NPER, 2000.0, 64))
    # reference-test parameter set, full day and noon release
    args.append((1, wc, HP_T, DP_T, DP_T, 1, 6, 8000.0, 320))
    args.append((1, wc, HP_T, DP_T, DP_T, 1, 6, 8000.0, 320, 0.5))
    with Pool(8) as pool:
        res = pool.map(_pm, args)
    names = ['kal128_d%d' % d for d in days[:6]] + \
            ['kal400_d%d' % d for d in days[:2]] + \
            ['car128_start', 'kal64_clip', 'test320_full', 'test320_noon']
    for nm, r in zip(names, res):
        coo_pack(nm, r, out)
    # single-period TEST_RUN case, tests/test_ParsitoidModel.py:315-325
    sing = {1: wc[1][24 * 30, :]}
    hp1 = (1.0, 1.8, 6, -4., 2., 19., 2.)
    r = quiet(PM.prob_mass, 1, sing, hp1, DP_T, DP_T, 0.1 / 24, 1, 8000.0, 320)
    coo_pack('test320_single', r, out)
    out['test320_single_wind'] = sing[1]
    save('g5_prob_mass', **out)


def coo_digest(prefix, M, out, every):
    """A big COO kernel as shape, nnz, sum, a SHA-256 over its (row, col) pattern in the
    reference's entry order and every `every`-th entry (index, row, col, value) -- a full dump
    of a 1600^2 kernel would be several MB."""
    import hashlib
    M = M.tocoo()
    row, col = M.row.astype(np.int32), M.col.astype(np.int32)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(row).tobytes())
    h.update(np.ascontiguousarray(col).tobytes())
    out[prefix + '_shape'] = np.array(M.shape, dtype=np.int64)
    out[prefix + '_nnz'] = np.int64(M.nnz)
    out[prefix + '_sum'] = np.float64(M.data.sum())
    out[prefix + '_pattern_sha256'] = np.frombuffer(h.digest(), dtype=np.uint8).copy()
    idx = np.arange(0, M.nnz, every, dtype=np.int64)
    out[prefix + '_samp_idx'] = idx
    out[prefix + '_samp_row'] = row[idx]
    out[prefix + '_samp_col'] = col[idx]
    out[prefix + '_samp_val'] = M.data[idx].astype(np.float64)
    # the entry with the largest value and the bounding rows / columns pin the shrink (ParasitoidModel.py:606-613)
    out[prefix + '_argmax'] = np.int64(np.argmax(M.data))


def g5b():
    """prob_mass at the grid sizes of BASELINE configs 2-5 (the tile / pair / ordered
    accumulate pipeline above R = 400 was only pinned through sum = 1 properties): Kalbar at
    R = 512 (config 4's grid), a prior-drawn ensemble member of config 5 on Carnarvon at R = 1024,
    and Carnarvon at R = 2048 (config 3) for a late release (about 100 periods)."""
    out = {}
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    wc, dc = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    # member 0 of scripts/run_ensemble.py:draw_members(512) -- lambda ~ Beta(5,1), sigma_x ~ Gamma(26, rate .15),
    # sigma_y ~ Gamma(15, rate .15), mu_r ~ N(1,1) > 0 (Bayes_Run.py:102,:116-117,:129)
    rng = np.random.default_rng(512)
    lam = rng.beta(5, 1, 512)
    sx = rng.gamma(26, 1 / 0.15, 512)
    sy = rng.gamma(15, 1 / 0.15, 512)
    mu = rng.normal(1, 1, 4 * 512)
    mu = mu[mu > 0][:512]
    hp5 = (float(lam[0]),) + HP[1:]
    dp5 = (float(sx[0]), float(sy[0]), 0.253)
    out['member0'] = np.array([lam[0], sx[0], sy[0], mu[0]])
    args = [(days[0], wd, HP, DP, DLP, MU_R, NPER, 10000.0, 512),
            (dc[3], wc, hp5, dp5, DLP, float(mu[0]), NPER, 10000.0, 1024),
            (dc[0], wc, HP, DP, DLP, MU_R, NPER, 10000.0, 2048, 0.93)]
    out['kal512_day'] = np.int64(days[0])
    out['car1024_day'] = np.int64(dc[3])
    out['car2048_day'] = np.int64(dc[0])
    with Pool(3) as pool:
        res = pool.map(_pm, args)
    for nm, r, every in zip(('kal512', 'car1024_member0', 'car2048_late'), res, (7, 23, 23)):
        coo_digest(nm, r, out, every)
        print(nm, r.shape, r.nnz, r.data.sum())
    save('g5b_prob_mass_large', **out)


def _sol_digest(prefix, M, out, count):
    """A full-size solution day as nnz, sum, pattern SHA-256 and about `count` evenly spaced
    entries (coo_digest's form with the spacing derived from nnz)."""
    every = max(1, M.nnz // count)
    coo_digest(prefix, M, out, every)


def g6b():
    """BASELINE config 3 on real wind at full size: the reference's `prob_mass` for
    30 Carnarvon days at R = 2048 and its `get_solutions` (CalcSol.py:140-201, CPU branch) at
    rad_dist 10 km (C3a: flags fire) and 40 km (C3b).  Stored per kernel and per solution day as
    digests (shape, nnz, sum, SHA-256 of the (row, col) pattern in entry order, ~400 evenly spaced
    entries), plus per chain day the flag `ifft2` returned and samples / sum / min of the raw field
    it was computed from (recorded by wrapping the reference's own `ifft2`; nothing is replayed).
    The release day is a full-day kernel (no start time), as in bench_extras.real_wind_case."""
    import time
    out = {}
    wc, dc = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    R, nd = 2048, 30
    N = 2 * R + 1
    out['days'] = np.array(dc[:nd], dtype=np.int64)
    pos = np.vstack([sample_positions(N, 3000, 61),
                     R + np.random.default_rng(62).integers(-300, 301, size=(1000, 2))])
    out['pos'] = pos
    for tag, rd in (('c3b', 40000.0), ('c3a', 10000.0)):
        t0 = time.time()
        args = [(d, wc, HP, DP, DLP, MU_R, NPER, rd, R) for d in dc[:nd]]
        with Pool(8) as pool:
            pmfs = pool.map(_pm, args, chunksize=1)
        print(tag, 'prob_mass', round(time.time() - t0), 's', flush=True)
        ms = _max_shape(pmfs)
        out[tag + '_max_shape'] = ms
        for i, p in enumerate(pmfs):
            _sol_digest('%s_pmf%d' % (tag, i), p, out, 400)
        raw = []
        real_ifft2 = CS.ifft2

        def spy(A_hat, dom_shape):
            A, flag = real_ifft2(A_hat, dom_shape)
            C = A.tocsr()
            raw.append((bool(flag), np.asarray(C[pos[:, 0], pos[:, 1]]).ravel(),
                        float(A.data.sum()), float(A.data.min()), float(A.data.max())))
            return A, flag

        CS.ifft2 = spy
        try:
            modelsol = [_recentre(pmfs[0], R)]
            t0 = time.time()
            quiet(CS.get_solutions, modelsol, pmfs, dc, nd, N, ms)
        finally:
            CS.ifft2 = real_ifft2
        print(tag, 'get_solutions', round(time.time() - t0), 's', 'flags',
              ''.join('1' if r[0] else '0' for r in raw), flush=True)
        assert len(raw) == nd - 1 and len(modelsol) == nd
        out[tag + '_flags'] = np.array([r[0] for r in raw])
        out[tag + '_rawsamp'] = np.array([r[1] for r in raw])
        out[tag + '_rawsum'] = np.array([r[2] for r in raw])
        out[tag + '_rawmin'] = np.array([r[3] for r in raw])
        out[tag + '_rawmax'] = np.array([r[4] for r in raw])
        for i, s in enumerate(modelsol):
            _sol_digest('%s_sol%d' % (tag, i), s, out, 400)
            C = s.tocsr()
            out['%s_solsamp%d' % (tag, i)] = np.asarray(C[pos[:, 0], pos[:, 1]]).ravel()
        # checkpoint after each variant: the run takes tens of CPU-minutes
        save('g6b_config3_full', **_g6b_compact(out))


def _g6b_compact(out):
    """Keep the committed fixture under 1 MB: 375 of the 3000 random and 125 of the 1000 central sample
    positions, every second digest entry (the digests' nnz / sum / SHA-256 cover every entry anyway)."""
    keep = np.r_[0:375, 3000:3125]
    res = {}
    for k, v in out.items():
        if k == 'pos':
            v = v[keep]
        elif k.endswith('_rawsamp'):
            v = v[:, keep]
        elif '_solsamp' in k:
            v = v[keep]
        elif k.endswith(('_samp_idx', '_samp_row', '_samp_col', '_samp_val')):
            v = v[::2]
        res[k] = v
    return res


def _kalbar_pmfs(R, nd):
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    args = [(d, wd, HP, DP, DLP, MU_R, NPER, 10000.0, R) for d in days[:nd]]
    with Pool(8) as pool:
        return pool.map(_pm, args), days


def _recentre(p, R):
    off = R - p.shape[0] // 2
    n = 2 * R + 1
    return sparse.coo_matrix((p.data, (p.row + off, p.col + off)), shape=(n, n))


def _max_shape(pmfs):
    ms = np.array([0, 0])
    for p in pmfs:
        ms = np.maximum(ms, p.shape)
    return ms


def g6():
    """get_solutions, config 1 (Kalbar, R=128, 6 days) + R=200 flag sequence."""
    out = {}
    for R, nd, full in ((128, 6, True), (200, 18, False)):
        pmfs, days = _kalbar_pmfs(R, nd)
        N = 2 * R + 1
        ms = _max_shape(pmfs)
        tag = 'r%d' % R
        out[tag + '_max_shape'] = ms
        out[tag + '_ndays'] = np.int64(nd)
        for i, p in enumerate(pmfs):
            coo_pack('%s_pmf%d' % (tag, i), p, out)
        modelsol = [_recentre(pmfs[0], R)]
        quiet(CS.get_solutions, modelsol, pmfs, days, nd, N, ms)
        pos = sample_positions(N, 4000, 7)
        out[tag + '_pos'] = pos
        for i, s in enumerate(modelsol):
            if full:
                coo_pack('%s_sol%d' % (tag, i), s, out)
            summarize('%s_sum%d' % (tag, i), s, out, pos)
        # replay with the reference's own primitives: flags and raw fields
        hat = CS.fft2(_recentre(pmfs[0], R), ms)
        flags = []
        for n in range(1, nd):
            CS.fftconv2(hat, pmfs[n].tocsr())
            A, flag = CS.ifft2(hat, [N, N])
            flags.append(flag)
            Ad = A.toarray()
            out['%s_rawsamp%d' % (tag, n)] = Ad[pos[:, 0], pos[:, 1]]
            out['%s_rawsum%d' % (tag, n)] = np.float64(Ad.sum())
            out['%s_rawmin%d' % (tag, n)] = np.float64(Ad.min())
            if flag:
                hat = CS.fft2(A, ms)
        out[tag + '_flags'] = np.array(flags)
    save('g6_solutions', **out)


def g7():
    """get_populations, Kalbar r_dur=1: R=128 6 days (full), R=400 18 days."""
    out = {}
    dist = lambda day: 1.0                                      # uniform, r_dur=1
    for R, nd, full in ((128, 6, True), (400, 18, False)):
        pmfs, days = _kalbar_pmfs(R, nd)
        N = 2 * R + 1
        ms = _max_shape(pmfs)
        tag = 'r%d' % R
        out[tag + '_max_shape'] = ms
        out[tag + '_ndays'] = np.int64(nd)
        if R == 400:
            for i, p in enumerate(pmfs):
                coo_pack('%s_pmf%d' % (tag, i), p, out)
        r_spread = [_recentre(pmfs[0], R).tocsr()]
        pop = quiet(CS.get_populations, r_spread, pmfs, days, nd, N, ms,
                    1, 130000, dist)
        pos = sample_positions(N, 4000, 11)
        out[tag + '_pos'] = pos
        for i, s in enumerate(pop):
            if full:
                coo_pack('%s_pop%d' % (tag, i), s, out)
            summarize('%s_sum%d' % (tag, i), s, out, pos)
    save('g7_populations', **out)


def g8():
    """back_solve: toy arrays (tests/test_CalcSol.py:41-62, :115-139) and
    Carnarvon --pop r_dur=5 at domain_info=(40000.0,200), 30 days."""
    out = {}
    Ad = np.outer(range(5), np.arange(.1, .6, .1))
    Bd = np.outer(np.arange(0, 2.5, 0.5), np.ones(5))
    Cd = np.outer(range(5, 0, -1), np.arange(.1, .6, .1))
    Dd = np.outer(np.arange(1, 0, -.2), np.arange(0, 2.5, 0.5))
    mats = []
    for X in (Ad, Bd, Cd, Dd):
        Z = np.zeros((55, 55)); Z[25:30, 25:30] = X
        mats.append(Z)
    A, B, C, D = mats
    C_hat = CS.fft2(sparse.coo_matrix(C), A.shape)
    CS.fftconv2(C_hat, sparse.csr_matrix(D))
    bck = CS.back_solve([sparse.csr_matrix(A), sparse.csr_matrix(B)],
                        C_hat, A.shape)
    out['toy_bck0'] = bck[0].toarray()
    out['toy_bck1'] = bck[1].toarray()
    for nm, X in zip('ABCD', mats):
        out['toy_' + nm] = X

    wc, dc = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
    R, nd, r_dur, r_number = 200, 30, 5, 40000
    args = [(dc[0], wc, HP, DP, DLP, MU_R, NPER, 40000.0, R, 0.354)]
    args += [(d, wc, HP, DP, DLP, MU_R, NPER, 40000.0, R) for d in dc[1:nd]]
    with Pool(8) as pool:
        pmfs = pool.map(_pm, args)
    N = 2 * R + 1
    ms = _max_shape(pmfs)
    out['car_max_shape'] = ms
    for i, p in enumerate(pmfs):
        coo_pack('car_pmf%d' % i, p, out)
    r_spread = [_recentre(pmfs[i], R).tocsr() for i in range(r_dur)]
    dist = lambda day: 1. / r_dur
    pop = quiet(CS.get_populations, r_spread, pmfs, dc, nd, N, ms, r_dur,
                r_number, dist)
    pos = sample_positions(N, 4000, 13)
    out['car_pos'] = pos
    for i, s in enumerate(pop):
        summarize('car_sum%d' % i, s, out, pos)
    for i in (0, 4, 5, 29):
        coo_pack('car_pop%d' % i, pop[i], out)
    save('g8_back_solve', **out)


def fake_locinfo():
    """A small stand-in for Data_Import.LocInfo with exactly the attributes Bayes_funcs reads
    (the real one needs the xlsx sheets and openpyxl).  Shared with tests/helpers.py through
    the arrays stored in the fixture."""
    import types
    import pandas as pd
    td = lambda d: pd.Timedelta(days=int(d))
    rng = np.random.default_rng(9)
    li = types.SimpleNamespace()
    li.collection_datesPR = [td(3), td(6)]
    li.emerg_grids = [[(int(r), int(c)) for r, c in rng.integers(118, 139, size=(12, 2))],
                      [(int(r), int(c)) for r, c in rng.integers(110, 147, size=(9, 2))]]
    li.release_DataFrames = [
        pd.DataFrame({'datePR': [td(22), td(22), td(24), td(27)]}),
        pd.DataFrame({'datePR': [td(25), td(28), td(28), td(30)]})]
    li.sent_ids = ['A', 'B', 'C']
    li.field_cells = {'A': rng.integers(100, 157, size=(40, 2)),
                      'B': rng.integers(60, 200, size=(25, 2)),
                      'C': rng.integers(120, 137, size=(60, 2))}
    li.sent_DataFrames = [pd.DataFrame({'datePR': [td(23), td(26)]}),
                          pd.DataFrame({'datePR': [td(26), td(29), td(31)]})]
    li.grid_cells = rng.integers(100, 157, size=(30, 2))
    li.grid_obs_datesPR = [td(2), td(5), td(6)]
    li.card_obs_datesPR = [td(3), td(6)]
    li.card_obs = [np.zeros((4, 6)), np.zeros((4, 9))]
    li.step_size = [10, 25]
    return li


def g9():
    """Bayes_funcs: popdensity_to_emergence / popdensity_grid / popdensity_card on the
    reference's Kalbar R=128 population solution with a synthetic LocInfo."""
    import Bayes_funcs as BF
    pmfs, days = _kalbar_pmfs(128, 6)
    N, R = 257, 128
    ms = _max_shape(pmfs)
    pop = quiet(CS.get_populations, [_recentre(pmfs[0], R).tocsr()], pmfs, days, 6, N, ms,
                1, 130000, lambda day: 1.0)
    li = fake_locinfo()
    rel, sen = BF.popdensity_to_emergence(pop, li)
    grid = BF.popdensity_grid(pop, li)
    card = BF.popdensity_card(pop, li, (10000.0, 128))
    out = {}
    for i, a in enumerate(rel): out['rel%d' % i] = a
    for i, a in enumerate(sen): out['sen%d' % i] = a
    out['grid'] = grid
    for i, a in enumerate(card): out['card%d' % i] = a
    # the LocInfo stand-in itself (so the GPU test rebuilds the identical object)
    out['collection_days'] = np.array([t.days for t in li.collection_datesPR])
    for i, g in enumerate(li.emerg_grids): out['emerg_grid%d' % i] = np.array(g)
    for i, d in enumerate(li.release_DataFrames): out['rel_dates%d' % i] = np.array([t.days for t in d['datePR']])
    for i, d in enumerate(li.sent_DataFrames): out['sen_dates%d' % i] = np.array([t.days for t in d['datePR']])
    for k in li.sent_ids: out['field_' + k] = li.field_cells[k]
    out['grid_cells'] = li.grid_cells
    out['grid_obs_days'] = np.array([t.days for t in li.grid_obs_datesPR])
    out['card_obs_days'] = np.array([t.days for t in li.card_obs_datesPR])
    out['card_obslen'] = np.array([a.shape[1] for a in li.card_obs])
    out['step_size'] = np.array(li.step_size)
    save('g9_bayes_funcs', **out)


GROUPS = {'g1': g1, 'g2': g2, 'g3': g3, 'g5': g5, 'g5b': g5b, 'g6': g6, 'g6b': g6b, 'g7': g7, 'g8': g8, 'g9': g9}

if __name__ == '__main__':
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        print('==', g)
        GROUPS[g]()
