"""Per-day anatomy of the real-wind chain (Carnarvon, R = 2048): kernel shapes, pad maxima, flags,
and -- in auto mode -- which helper ran each day.  python scripts/diag_c3a.py [rad_dist]"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402
from parasitoids_amd import ParasitoidModel as PM   # noqa: E402
from parasitoids_amd.pop_model import PopModel   # noqa: E402

rd = float(sys.argv[1]) if len(sys.argv) > 1 else 10000.0
R, nd = 2048, 30
wd, days = PM.get_wind_data('data/carnarvonearl', 30, '00:30')
out = {}
for mode in ('fast', 'auto'):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        pm = PopModel(wd, days, domain_info=(rd, R), mode=mode, prob_model=True)
        pm.evaluate(B.HP, B.DP, B.DLP, B.MU_R, B.NPER, ndays=nd)
        pm.evaluate(B.HP, B.DP, B.DLP, B.MU_R, B.NPER, ndays=nd)
    s = pm.solver
    st = pm.stats
    ks = pm.model.last['kshape']
    rec = {'fft_len': s.fft_len, 'kshape': [int(k) for k in ks],
           'padmax': [float(x.padmax) for x in st], 'flag': [int(x.flag) for x in st],
           'nnz': [int(x.nnz) for x in st]}
    if mode == 'auto':
        rec['route'] = [int(v) for v in s.auto_route(0, nd - 1)]
        rec['auto_info'] = list(s.auto_info())
    t0 = time.perf_counter()
    dt, _ = B._chain_rate(pm, nd, 3)
    rec['chain_ms'] = round(dt * 1e3, 3)
    out[mode] = rec
    pm.close()
print(json.dumps(out))
