#!/usr/bin/env python3
"""BASELINE.json config 5: parameter-ensemble sweep -- M samples of (lambda, sigma_x, sigma_y,
mu_r) drawn from the reference's priors (Bayes_Run.py:102,:116-117,:129) x an N x N grid x
30 Carnarvon days, probability model, member i on GPU i mod world (round-robin, results
gathered on rank 0).

    python scripts/run_ensemble.py --members 512 --rad-res 1024
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_ensemble.py ...
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def draw_members(m, seed=512):
    from parasitoids_amd.synthetic import ensemble_members
    return ensemble_members(m, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=512)
    ap.add_argument('--rad-res', type=int, default=1024)
    ap.add_argument('--ndays', type=int, default=30)
    ap.add_argument('--mode', default='fast', choices=['auto', 'exact', 'fold', 'fast'])
    args = ap.parse_args()
    warnings.simplefilter('ignore', RuntimeWarning)
    from parasitoids_amd import parallel, ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    rank, world = parallel.init()
    if rank == 0:
        wd, days = PM.get_wind_data(os.path.join(ROOT, 'parasitoids_amd', 'data', 'carnarvonearl'),
                                    30, '00:30')
    else:
        wd, days = None, None
    wd, days = parallel.broadcast_wind(wd, days)
    model = PopModel(wd, days, domain_info=(10000.0, args.rad_res), r_start=0.354, mode=args.mode,
                     prob_model=True)
    g, f = (1.263, 3.913), (7.302, 2.614, 23.999, 2.350)

    def run(mem):
        hp = (mem['lam'], *g, *f)
        st = model.evaluate(hp, (mem['sig_x'], mem['sig_y'], 0.253), (7.096, 7.260, 0.0),
                            mem['mu_r'], 30, ndays=args.ndays)
        tot, mr, mc, vr, vc = model.moments(args.ndays - 1)
        return dict(nnz_last=int(st[-1][0]), kept_mass_last=float(st[-1][1]), total=float(tot),
                    mean=(float(mr), float(mc)), var=(float(vr), float(vc)))

    members = draw_members(args.members)
    t0 = time.time()
    res = parallel.run_members(members, run)
    dt = time.time() - t0
    if rank == 0:
        grid_days = args.members * (args.ndays - 1)
        print(json.dumps({'members': args.members, 'n_gpus': world, 'rad_res': args.rad_res,
                          'ndays': args.ndays, 'seconds': round(dt, 2),
                          'members_per_s': round(args.members / dt, 2),
                          'grid_days_per_s': round(grid_days / dt, 1),
                          'first': res[0], 'last': res[-1]}))
    model.close()


if __name__ == '__main__':
    main()
