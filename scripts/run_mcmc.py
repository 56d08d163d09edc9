#!/usr/bin/env python3
"""MCMC samples/hour with the in-repo Metropolis sampler (parasitoids_amd/mcmc.py): Kalbar
wind and Kalbar field observations (the reference's data files; the xlsx sheets as CSV
fixtures, parasitoids_amd/Data_Import.py), the reference's priors and Poisson observation
model, one chain on one GPU.  --synthetic draws the observations from the model itself on a
Kalbar-like geometry instead.

    python scripts/run_mcmc.py [--samples 200] [--rad-res 400] [--mode auto] [--seed 1000] [--synthetic]
"""
import argparse
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=200)
    ap.add_argument('--burn', type=int, default=20)
    ap.add_argument('--rad-res', type=int, default=400)
    ap.add_argument('--mode', default='auto', choices=['exact', 'fold', 'fast', 'auto'])
    ap.add_argument('--seed', type=int, default=1000)
    ap.add_argument('--synthetic', action='store_true',
                    help='observations drawn from the model on a Kalbar-like geometry instead of '
                         'the Kalbar field data (parasitoids_amd/data CSV exports)')
    args = ap.parse_args()
    warnings.simplefilter('ignore', RuntimeWarning)
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd import mcmc
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data(os.path.join(ROOT, 'parasitoids_amd', 'data', 'kalbar'), 30, '00:00')
    pm = PopModel(wd, days, domain_info=(10000.0, args.rad_res), r_number=130000, mode=args.mode)
    if args.synthetic:
        li = mcmc.synthetic_locinfo(pm, args.rad_res, seed=9)
    else:
        from parasitoids_amd.Data_Import import LocInfo
        li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, args.rad_res))   # Run.py:129
    cell_area = (10000.0 / args.rad_res) ** 2
    chain = mcmc.Metropolis(pm, li, cell_area, seed=args.seed)
    chain.run(args.burn)
    res = chain.run(args.samples)
    tr = res['trace']
    out = {'metric': 'MCMC samples/hour (Kalbar wind, %s observations)' % ('synthetic' if args.synthetic else 'Kalbar field'),
           'value': round(res['samples_per_hour'], 1), 'unit': 'samples/hour', 'n_gpus': 1,
           'samples': args.samples, 'ms_per_sample': round(1e3 * res['seconds'] / args.samples, 3),
           'acceptance': round(res['acceptance'], 3),
           'evaluations': res['evaluations'], 'failed_evaluations': res['failed_evaluations'],
           'config': {'workload': 'block Metropolis over 15 model parameters (one pop_model '
                                  'evaluation: 18 x prob_mass + get_populations + gathers) + 3 '
                                  'scalar nuisance updates per sample, R=%d, %s mode'
                                  % (args.rad_res, args.mode)},
           'posterior_mean': {n: round(float(v), 5) for n, v in zip(res['names'], tr.mean(0))},
           'logp_first_last': [round(float(res['logp'][0]), 3), round(float(res['logp'][-1]), 3)]}
    print(json.dumps(out))
    pm.close()


if __name__ == '__main__':
    main()
