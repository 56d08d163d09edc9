"""A/B of the two-role inverse row kernel (PS_ROW2): bit-identity and time.
    python scripts/try_row2.py small|headline"""
import os
import sys
import time

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PS_TPIPE'] = '1'
os.environ['PS_RSP'] = '1'
from parasitoids_amd import hip_lib, synthetic   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'small'
if which == 'small':
    cases = [(400, 401, 16, 400), (400, 401, 16, 760), (1100, 801, 10, 1100)]
else:
    cases = [(2048, 2049, 30, 2048)]
for R, K, nd, start in cases:
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=7 if R < 2000 else 20240613,
                                         sigma=(6.0, 12.0) if R < 2000 else (20.0, 60.0), shift=10 if R < 2000 else 64)
    state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
    out = {}
    for row2 in (0, 1):
        s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
        s.set_option('PS_ROW2', row2)
        s.set_kernels(kernels)
        ts = []
        for rep in range(4):
            s.set_state(state)
            s.sync()
            t0 = time.perf_counter()
            s.run_chain(renorm=True)
            s.sync()
            ts.append(time.perf_counter() - t0)
        st = s.chain_stats(0, nd)
        days = [0, nd // 2, nd - 1] if R >= 2000 else range(nd)
        out[row2] = ([s.dense(0, d) for d in days], [(x.flag, x.nnz, x.sum, x.delta, x.padmax) for x in st], min(ts[1:]))
        print('R', R, 'fft', s.fft_len, 'row2', row2, 'ms', round(min(ts[1:]) * 1e3, 3), flush=True)
        s.close()
    same = out[0][1] == out[1][1] and all(np.array_equal(a, b) for a, b in zip(out[0][0], out[1][0]))
    print('R', R, 'start', start, 'bit-identical:', same, 'flags', sum(x[0] for x in out[0][1]), flush=True)
