#!/usr/bin/env python3
"""MCMC samples/hour with the in-repo Metropolis sampler (parasitoids_amd/mcmc.py): Kalbar
wind (the reference's data file), the reference's priors and Poisson observation model, one
chain on one GPU.  The OBSERVATIONS ARE SYNTHETIC -- drawn from the model at the reference's
initial parameter values on a Kalbar-like sampling geometry -- because the xlsx field data
cannot be read in this image; the cost per sample does not depend on that.

    python scripts/run_mcmc.py [--samples 200] [--rad-res 400] [--mode auto] [--seed 1000]
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
    ap.add_argument('--mode', default='auto', choices=['exact', 'fast', 'auto'])
    ap.add_argument('--seed', type=int, default=1000)
    args = ap.parse_args()
    warnings.simplefilter('ignore', RuntimeWarning)
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd import mcmc
    from parasitoids_amd.pop_model import PopModel
    wd, days = PM.get_wind_data(os.path.join(ROOT, 'tests', 'golden', 'data', 'kalbar'), 30, '00:00')
    pm = PopModel(wd, days, domain_info=(10000.0, args.rad_res), r_number=130000, mode=args.mode)
    li = mcmc.synthetic_locinfo(pm, args.rad_res, seed=9)
    cell_area = (10000.0 / args.rad_res) ** 2
    chain = mcmc.Metropolis(pm, li, cell_area, seed=args.seed)
    chain.run(args.burn)
    res = chain.run(args.samples)
    tr = res['trace']
    out = {'metric': 'MCMC samples/hour (Kalbar wind, synthetic observations)',
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
