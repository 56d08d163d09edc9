"""Shared helpers for the parity tests (fixtures -> scipy objects, parameters)."""
import numpy as np
from scipy import sparse

# default model parameters, reference Run.py:68-83
HP = (1., 1.263, 3.913, 7.302, 2.614, 23.999, 2.350)
DP = (171.82, 144.58, 0.253)
DLP = (7.096, 7.260, 0.000)
MU_R = 1.179
NPER = 30
# reference test parameters, tests/test_ParsitoidModel.py:24-56
HP_T = (1.0, 1.8, 6, 7., 2., 19., 2.)
DP_T = (4.0, 4.0, 0.)


def coo_from(g, prefix):
    shape = tuple(int(v) for v in g[prefix + '_shape'])
    return sparse.coo_matrix((g[prefix + '_val'],
                              (g[prefix + '_row'], g[prefix + '_col'])), shape=shape)


def recentre(p, R):
    """Run.py:454-458: shrunk kernel -> N x N solution."""
    off = R - p.shape[0] // 2
    n = 2 * R + 1
    return sparse.coo_matrix((p.data, (p.row + off, p.col + off)), shape=(n, n))


def wsum(M):
    """Position-weighted checksum used by make_golden.summarize."""
    D = M.tocoo()
    wt = 1.0 + ((D.row.astype(np.int64) * 31 + D.col.astype(np.int64) * 17) % 97)
    return float((D.data * wt).sum())


def assert_summary(g, prefix, M, pos, rtol=1e-12, atol=1e-12, nnz_slack=0):
    C = M.tocsr()
    D = M.tocoo()
    assert abs(D.nnz - int(g[prefix + '_nnz'])) <= nnz_slack
    assert np.isclose(D.data.sum(), float(g[prefix + '_sum']), rtol=rtol, atol=atol)
    assert np.isclose(wsum(M), float(g[prefix + '_wsum']), rtol=max(rtol, 1e-11), atol=atol)
    samp = np.asarray(C[pos[:, 0], pos[:, 1]]).ravel()
    np.testing.assert_allclose(samp, g[prefix + '_samp'], rtol=rtol, atol=atol)


def check_digest(g, name, got, tol=5e-15):
    """device COO kernel against a g5b fixture: shape, nnz, the (row, col) pattern in entry order
    (SHA-256), every sampled entry, the sum"""
    import hashlib
    got = got.tocoo()
    assert tuple(got.shape) == tuple(g[name + '_shape']), (name, got.shape)
    assert got.nnz == int(g[name + '_nnz']), (name, got.nnz, int(g[name + '_nnz']))
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(got.row.astype(np.int32)).tobytes())
    h.update(np.ascontiguousarray(got.col.astype(np.int32)).tobytes())
    assert h.digest() == g[name + '_pattern_sha256'].tobytes(), name + ': COO pattern differs'
    idx = g[name + '_samp_idx']
    assert np.array_equal(got.row[idx], g[name + '_samp_row']) and np.array_equal(got.col[idx], g[name + '_samp_col'])
    np.testing.assert_allclose(got.data[idx], g[name + '_samp_val'], rtol=0, atol=tol, err_msg=name)
    assert abs(got.data.sum() - float(g[name + '_sum'])) < 1e-12
    assert int(np.argmax(got.data)) == int(g[name + '_argmax'])
