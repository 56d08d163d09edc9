"""One real-wind chain (Carnarvon, R = 2048, rad_dist 10 km) in a chosen mode, for kernel traces:
    python scripts/trace_c3a_auto.py [mode] [rad_dist]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'auto'
rd = float(sys.argv[2]) if len(sys.argv) > 2 else 10000.0
rec = B.real_wind_case(rd, 2048, 30, mode, None, reps=2, prof=False)
print(json.dumps({k: v for k, v in rec.items() if k != 'kernels'}))
