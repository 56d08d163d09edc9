"""prob_mass batch time (18 Kalbar days, R = 400) with the unrolled pair kernel and without:
    python scripts/time_prob_mass.py [rho ...]"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402
from parasitoids_amd import ParasitoidModel as PM   # noqa: E402

warnings.simplefilter('ignore', RuntimeWarning)
wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
for rho in [float(v) for v in sys.argv[1:]] or [0.253, 0.5]:
    rec = {'rho': rho}
    for tag, flag in (('unrolled', 0), ('runtime_count', 1)):
        m = PM.WindModel(wd)
        m.set_option('PS_PM_NO_UNROLL', flag)
        dp = (B.DP[0], B.DP[1], rho)
        for _ in range(3):
            m.build(days, B.HP, dp, B.DLP, B.MU_R, B.NPER, 10000.0, 400)
        t0 = time.perf_counter()
        for _ in range(10):
            m.build(days, B.HP, dp, B.DLP, B.MU_R, B.NPER, 10000.0, 400)
        rec[tag + '_ms'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        m.close()
    print(json.dumps(rec), flush=True)
