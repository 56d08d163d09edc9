#!/usr/bin/env python3
"""BASELINE config 3a (Carnarvon, R = 2048, rad_dist 10 km: flags on 19 of 29 days) in fast mode on
the FFT sizes given on the command line (PS_FAST_SIZE forces the fast torus): which size the
flagged chain runs best on.  usage: run_c3a_sizes.py 5600 5760 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import json, sys; sys.path.insert(0, %r); import bench_extras as B; "
        "r = B.real_wind_case(10000.0, 2048, 30, 'fast'); "
        "print(json.dumps({k: r[k] for k in ('fft_len', 'grid_days_per_s', 'chain_ms', 'flagged_days', 'kernels')}))" % ROOT)
for size in sys.argv[1:] or ['0']:
    env = dict(os.environ)
    if size != '0':
        env['PS_FAST_SIZE'] = size
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, cwd=ROOT)
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
    try:
        d = json.loads(line)
        print(size, d['fft_len'], d['grid_days_per_s'], d['chain_ms'], d['flagged_days'],
              {k: v['ms_per_chain'] for k, v in d['kernels'].items()}, flush=True)
    except Exception:
        print(size, 'failed:', line, flush=True)
