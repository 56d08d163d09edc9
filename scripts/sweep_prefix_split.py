"""The day-split of one flag-free simulation (parallel.chain_prefix_split_local, three blocks) against the
sequential chain at every register-resident FFT size: batched forward / inverse column launches and the
element-wise products of ps_chain_block_prefix / _finish, one template instance per size.
    python scripts/sweep_prefix_split.py"""
import os
import re
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PS_TPIPE'] = '1'
sizes = sorted(16 * int(a) * int(b) for a, b in
               re.findall(r'X\((\d+), (\d+)\)', open(os.path.join(ROOT, 'parasitoids_amd', 'csrc', 'fft_rs_sizes.h')).read()))
from parasitoids_amd import hip_lib, parallel, synthetic   # noqa: E402

bad = []
for L in sizes:
    K = 2 * (L // 7) + 1
    R = (L - K // 2 - 1) // 2
    N = 2 * R + 1
    nd = 7
    os.environ['PS_FAST_SIZE'] = str(L)
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=L, sigma=(L / 260.0, L / 90.0), shift=L / 80.0)
    state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
    seq = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
    assert seq.fft_len == L
    seq.set_kernels(kernels)
    seq.run_chain(renorm=True)
    solvers = [hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True) for _ in range(3)]
    for s in solvers:
        s.set_kernels(kernels)
    blocks, flagged = parallel.chain_prefix_split_local(solvers, nd)
    worst = 0.0
    for s, (f, c) in zip(solvers, blocks):
        for d in range(f, f + c):
            ref = seq.dense(0, d)
            worst = max(worst, float(np.abs(s.dense(0, d) - ref).max()) / float(np.abs(ref).max()))
    ok = (not flagged) and worst <= 1e-13
    print('L', L, 'N', N, 'blocks', blocks, 'max |split - sequential| / max', '%.2e' % worst, 'ok' if ok else 'BAD', flush=True)
    if not ok:
        bad.append(L)
    for s in solvers + [seq]:
        s.close()
print('sizes', len(sizes), 'bad:', bad, flush=True)
sys.exit(1 if bad else 0)
