"""Synthetic day-kernel stacks for benchmarks and size-independent parity checks
(SURVEY.md section 8d, config C3): unit mass at the centre of an N x N domain and
`ndays` axis-aligned Gaussian kernels of shape K x K, entries below 1e-8 dropped
and renormalised to sum 1, stored as COO like `prob_mass` output."""
import numpy as np
from scipy import sparse


def gaussian_kernel(K, sx, sy, dx, dy, cut=1e-8):
    m = K // 2
    x = np.arange(K) - m
    gx = np.exp(-0.5 * ((x - dx) / sx) ** 2)
    gy = np.exp(-0.5 * ((x + dy) / sy) ** 2)      # row index grows downwards: +dy is up
    gx /= gx.sum()
    gy /= gy.sum()
    # only the block that can exceed the cut
    cols = np.nonzero(gx * gy.max() >= cut)[0]
    rows = np.nonzero(gy * gx.max() >= cut)[0]
    blk = np.outer(gy[rows], gx[cols])
    r, c = np.nonzero(blk >= cut)
    v = blk[r, c]
    v = v / v.sum()
    return sparse.coo_matrix((v, (rows[r], cols[c])), shape=(K, K))


def make_stack(R=2048, K=2049, ndays=30, seed=20240613, sigma=(20.0, 60.0), shift=64.0):
    """-> (state coo N x N, [kernel coo K x K] * ndays, params list)"""
    N = 2 * R + 1
    rng = np.random.default_rng(seed)
    kernels, params = [], []
    for _ in range(ndays):
        sx, sy = rng.uniform(sigma[0], sigma[1], 2)
        dx, dy = rng.uniform(-shift, shift, 2)
        kernels.append(gaussian_kernel(K, sx, sy, dx, dy))
        params.append((sx, sy, dx, dy))
    state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
    return state, kernels, params


def moments(M):
    """(sum, mean_row, mean_col, var_row, var_col) of a sparse/dense field"""
    M = sparse.coo_matrix(M)
    s = M.data.sum()
    mr = (M.data * M.row).sum() / s
    mc = (M.data * M.col).sum() / s
    vr = (M.data * (M.row - mr) ** 2).sum() / s
    vc = (M.data * (M.col - mc) ** 2).sum() / s
    return s, mr, mc, vr, vc


def ensemble_members(m, seed=512):
    """BASELINE config 5: `m` samples of (lambda, sigma_x, sigma_y, mu_r) from the reference's
    priors -- lambda ~ Beta(5, 1), sigma_x ~ Gamma(26, rate 0.15), sigma_y ~ Gamma(15, rate 0.15),
    mu_r ~ N(1, 1) truncated > 0 (Bayes_Run.py:102, :116-117, :129), `default_rng(512)`."""
    rng = np.random.default_rng(seed)
    lam = rng.beta(5, 1, m)
    sx = rng.gamma(26, 1 / 0.15, m)
    sy = rng.gamma(15, 1 / 0.15, m)
    mu = rng.normal(1, 1, 4 * m)
    mu = mu[mu > 0][:m]
    return [dict(lam=float(lam[i]), sig_x=float(sx[i]), sig_y=float(sy[i]), mu_r=float(mu[i]))
            for i in range(m)]
