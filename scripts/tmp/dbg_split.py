import os, sys
os.environ['PS_TPIPE'] = '1'
sys.path.insert(0, '/root/repo')
import numpy as np
from scipy import sparse
from parasitoids_amd import hip_lib, synthetic, parallel
R, K, nd = 400, 401, 12
N = 2 * R + 1
_, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd + 1, seed=7, sigma=(6.0, 12.0), shift=10)
state = sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N))
seq = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
seq.set_kernels(kernels); seq.run_chain(0, nd, renorm=True)
sf = [seq.dense(0, d) for d in range(nd)]
print('seq padmax', [x.padmax for x in seq.chain_stats(0, nd)], 'full_column', seq.full_column)
s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
s.set_kernels(kernels)
ptr, nb = s.block_prefix(0, nd)
fl = s.block_finish(0, nd)
st = s.chain_stats(0, nd)
print('flagged', fl, 'padmax', [x.padmax for x in st])
for d in range(nd):
    g = s.dense(0, d)
    print(d, 'max', sf[d].max(), g.max(), 'diff', np.abs(g - sf[d]).max(), 'sum', g.sum(), sf[d].sum(), 'argmax', np.unravel_index(g.argmax(), g.shape), np.unravel_index(sf[d].argmax(), sf[d].shape))
s.close()
for G in (2, 3):
    solvers = [hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True) for _ in range(G)]
    for x in solvers:
        x.set_kernels(kernels)
    blocks, flagged = parallel.chain_prefix_split_local(solvers, nd)
    print('G', G, blocks, 'flagged', flagged)
    for x, (f, c) in zip(solvers, blocks):
        st = x.chain_stats(f, c)
        for i in range(c):
            g = x.dense(0, f + i)
            print('  day', f + i, 'padmax', st[i].padmax, 'diff', np.abs(g - sf[f + i]).max(), 'max', g.max())
    for x in solvers:
        x.close()
