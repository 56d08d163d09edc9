#!/usr/bin/env python3
"""Host side of a headline stack: how long set_state and run_chain take to return, and how long the
device idles between two stacks (wall time of K stacks minus K x the device time of one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from parasitoids_amd import hip_lib, synthetic

R, K, nd = 2048, 2049, 30
state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=20240613)
s = hip_lib.HipSolve(state, [K, K], mode='fast', chain_only=True)
s.set_kernels(kernels)
for _ in range(3):
    s.set_state(state); s.run_chain(0, nd, renorm=True)
s.sync()
ts, tr = [], []
t00 = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter(); s.set_state(state); t1 = time.perf_counter(); s.run_chain(0, nd, renorm=True); t2 = time.perf_counter()
    ts.append(t1 - t0); tr.append(t2 - t1)
s.sync()
tot = time.perf_counter() - t00
print('set_state returns after %.0f us (min %.0f), run_chain after %.0f us; %.3f ms per stack' %
      (1e6 * np.median(ts), 1e6 * min(ts), 1e6 * np.median(tr), 1e3 * tot / 10))
