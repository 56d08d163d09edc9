"""cProfile of the host side of a short MCMC run (Kalbar, R = 400): where the time between the
device's kernels goes.  usage: profile_bayes_host.py [mode] [samples]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fast'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
B.bayes_case(400, mode, 30, 5)            # warm
pr = cProfile.Profile()
pr.enable()
rec, _ = B.bayes_case(400, mode, n, 10)
pr.disable()
print(rec['ms_per_sample'])
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
