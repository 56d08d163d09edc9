#!/usr/bin/env python3
"""One MCMC chain on the Kalbar data (bench_extras.bayes_case) for profiling:
    python scripts/time_bayes_case.py [R] [mode] [samples]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mode = sys.argv[2] if len(sys.argv) > 2 else 'auto'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rec, _ = B.bayes_case(R, mode, n, 10)
print(json.dumps(rec))
