"""A/B of the intermediate's layout between the inverse column and row passes (row pairs interleaved, the
default, against row-major: PS_NO_PAIR_ROWS=1): per-class times of the hinted 30-day stack, bit-identity.
    python scripts/try_pair_rows.py [K ...]      K = 2049 -> 5184 points, 2045 -> 5120"""
import os
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parasitoids_amd import hip_lib, synthetic   # noqa: E402

R, nd = 2048, 30
N = 2 * R + 1
for K in [int(v) for v in sys.argv[1:]] or [2049]:
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=20240613, sigma=(20.0, 60.0), shift=64)
    state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
    out = {}
    for tag, opts in (('pairs', {}), ('row_major', {'PS_NO_PAIR_ROWS': 1}),
                      ('pairs_single_role', {'PS_DUAL_MIN_DAYS': 0}), ('row_major_single_role', {'PS_NO_PAIR_ROWS': 1, 'PS_DUAL_MIN_DAYS': 0})):
        s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_kernels(kernels)
        for rep in range(2):
            s.set_state(state); s.run_chain(renorm=True); s.sync()
        s.prof_enable(True)
        for rep in range(5):
            s.set_state(state); s.run_chain(renorm=True); s.sync()
        t, d = s.prof_read(), s.prof_days()
        st = s.chain_stats(0, nd)
        out[tag] = ([s.dense(0, i) for i in (0, nd // 2, nd - 1)], [(x.flag, x.nnz, x.sum, x.delta, x.padmax) for x in st])
        tab = {k: round(ms / c, 4) for k, (ms, c) in t.items() if c}
        print('K', K, 'fft', s.fft_len, tag, tab, 'sum', round(sum(ms for ms, c in t.values()) / 5, 3), flush=True)
        s.close()
    ref = out['row_major_single_role']
    for tag, o in out.items():
        same = o[1] == ref[1] and all(np.array_equal(a, b) for a, b in zip(o[0], ref[0]))
        print('K', K, tag, 'bit-identical to row_major_single_role:', same, flush=True)
