#!/usr/bin/env python3
"""Wall time of `get_populations` with r_dur = 5 (Carnarvon preset: four release-day filters
back-solved on each of the 30 days, CalcSol.py:296-323) -- run once with PS_NO_FILTER_CACHE=1
and once without to see what caching the filters' spectra on the solver buys.

    python scripts/time_back_solve.py [R]   ->  one JSON line
"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.simplefilter('ignore')
from parasitoids_amd import Run   # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = Run.Params(config=None)
p.cmd_line_chg(['--carnarvon', '--pop', 'domain_info=(40000.0,%d)' % R])
p.OUTPUT = False
times = []
for rep in range(3):
    t0 = time.perf_counter()
    modelsol, days, ndays, t = Run.run_model(p, verbose=False)
    times.append(round(t['solver_s'], 4))
print(json.dumps({'workload': 'Run.py --carnarvon --pop (r_dur=5), domain_info=(40000,%d), %d days' % (R, ndays),
                  'filter_cache': os.environ.get('PS_NO_FILTER_CACHE') is None,
                  'get_populations_s': times, 'day30_total': round(float(modelsol[-1].sum()), 6)}))
