"""Every register-resident FFT size (csrc/fft_rs_sizes.h) once: a 7-day chain run twice (the second run opens with
one 7-day window: the chained / two-role column kernels where the size has them, the batched row launch) with the
intermediate between column and row pass pair-interleaved (default) and row-major (PS_NO_PAIR_ROWS) -- fields,
statistics and flags must be bit-identical.  One template instance per size and kernel: this walks them all.
    python scripts/sweep_pair_rows.py [first_index [count]]"""
import os
import re
import sys
import time

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PS_TPIPE'] = '1'
os.environ['PS_RSP'] = '1'
sizes = sorted(16 * int(a) * int(b) for a, b in
               re.findall(r'X\((\d+), (\d+)\)', open(os.path.join(ROOT, 'parasitoids_amd', 'csrc', 'fft_rs_sizes.h')).read()))
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else len(sizes)
from parasitoids_amd import hip_lib, synthetic   # noqa: E402

bad = []
for L in sizes[lo:lo + cnt]:
    K = 2 * (L // 7) + 1
    R = (L - K // 2 - 1) // 2
    N = 2 * R + 1
    nd = 7
    os.environ['PS_FAST_SIZE'] = str(L)
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=L, sigma=(L / 260.0, L / 90.0), shift=L / 80.0)
    out = {}
    t0 = time.perf_counter()
    for tag, opts in (('pairs', {}), ('row_major', {'PS_NO_PAIR_ROWS': 1})):
        for start in (R, N - 1 - max(3, L // 200)):        # centre: no flag; next to the edge: flags
            state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
            s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
            assert s.fft_len == L and s.full_column, (L, s.fft_len)
            for k, v in opts.items():
                s.set_option(k, v)
            s.set_kernels(kernels)
            for rep in range(2):
                s.set_state(state)
                s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            out[tag, start] = ([s.dense(0, d) for d in (0, 3, nd - 1)],
                               [(x.flag, x.nnz, x.sum, x.delta, x.padmax) for x in st])
            s.close()
    ok = True
    for start in (R, N - 1 - max(3, L // 200)):
        a, b = out['pairs', start], out['row_major', start]
        ok = ok and a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
        ok = ok and all(np.isfinite(x).all() for x in a[0])
    # sanity of the run itself: the centre start keeps its mass and raises no flag
    centre = out['pairs', R]
    ok = ok and not any(f for f, *_ in centre[1]) and abs(float(centre[0][0].sum()) - 1.0) < 1e-9
    flags = sum(f for f, *_ in out['pairs', N - 1 - max(3, L // 200)][1])
    print('L', L, 'N', N, 'K', K, 'bit-identical' if ok else 'DIFFERENT', 'flags(edge run)', flags,
          's', round(time.perf_counter() - t0, 1), flush=True)
    if not ok:
        bad.append(L)
print('sizes', len(sizes[lo:lo + cnt]), 'different:', bad, flush=True)
sys.exit(1 if bad else 0)
