#!/usr/bin/env python3
"""Second headline metric (BASELINE.json): Bayes_Run MCMC samples/hour on the Kalbar data,
measured as evaluations/hour of the model body of `Bayes_Run.pop_model` (18 x prob_mass +
get_populations, Bayes_Run.py:204-336) with the parameters moved by a random-walk proposal
every evaluation, one independent chain per GPU (BASELINE config 4; no data-path collective).

    python bench_bayes.py [--gpus N] [--evals K] [--warmup W] [--rad-res 400|512] [--mode exact|fast]

Prints one JSON line on rank 0.  The driver's headline bench is bench.py; this one exists
for the samples/hour figure and is cited in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def proposals(rng, n):
    """Random-walk proposals around the reference's default parameters (Run.py:68-83) with
    step sizes in the range of Bayes_Run's AdaptiveMetropolis scales (Bayes_Run.py:188-196)."""
    lam, aw, bw, a1, b1, a2, b2 = 1., 1.263, 3.913, 7.302, 2.614, 23.999, 2.350
    sx, sy, rho = 171.82, 144.58, 0.253
    slx, sly = 7.096, 7.260
    mu_r = 1.179
    out = []
    for _ in range(n):
        sx = abs(sx + rng.normal(0, 2.0)); sy = abs(sy + rng.normal(0, 2.0))
        rho = float(np.clip(rho + rng.normal(0, 0.01), -0.9, 0.9))
        mu_r = abs(mu_r + rng.normal(0, 0.01))
        aw = aw + rng.normal(0, 0.01); bw = abs(bw + rng.normal(0, 0.02))
        lam = float(np.clip(lam + rng.normal(0, 0.005), 0.5, 1.0))
        out.append(((lam, aw, bw, a1, b1, a2, b2), (sx, sy, rho), (slx, sly, 0.0), mu_r))
    return out


def cpu_sample(wd, days, rad_res):
    """Oracle on a bounded sample: prob_mass of ONE day + 2 chain days; extrapolated to one
    18-day evaluation on one core."""
    from oracle import model as OM, calcsol as OC
    from helpers import HP, DP, DLP, MU_R, NPER, recentre
    t0 = time.perf_counter()
    p0 = OM.prob_mass(days[0], wd, HP, DP, DLP, MU_R, NPER, 10000.0, rad_res)
    t_pm = time.perf_counter() - t0
    N = 2 * rad_res + 1
    ms = np.array(p0.shape)
    t0 = time.perf_counter()
    hat = OC.fft2(recentre(p0, rad_res), ms)
    for _ in range(2):
        OC.fftconv2(hat, p0.tocsr())
        A, flag = OC.ifft2(hat, [N, N])
        OC.r_small_vals(A * 130000.0)
        if flag:
            hat = OC.fft2(A, ms)
    t_day = (time.perf_counter() - t0) / 2
    per_eval = 18 * t_pm + 17 * t_day
    return {'value': 3600.0 / per_eval, 'unit': 'evaluations/hour', 'cores': 1, 'kind': 'port',
            'sample': 'oracle prob_mass of 1 day (%.1fs) and 2 chain days (%.2fs each) at R=%d, '
                      'extrapolated to 18 + 17' % (t_pm, t_day, rad_res)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--evals', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rad-res', type=int, default=400)
    ap.add_argument('--mode', default='auto', choices=['exact', 'fold', 'fast', 'auto'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rehearse', action='store_true',
                    help='launcher/rendezvous rehearsal without device work (CPU test, gloo)')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        # the driver calls `python bench_bayes.py --gpus N`: start the N ranks here, from a
        # parent that never touches the GPU (bench.launch_ranks)
        from bench import launch_ranks
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], script=__file__))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, args.gpus))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['PARASITOID_DEVICE'] = str(local)
    import torch
    import torch.distributed as dist
    backend = os.environ.get('BENCH_BACKEND', 'nccl')
    if args.rehearse:
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo', rank=rank, world_size=world)
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'Bayes_Run model evaluations/hour (Kalbar, 18 days)', 'value': None,
                              'n_gpus': dist.get_world_size() if world > 1 else 1, 'rehearsal': True}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if backend == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit('--gpus %d but only %d device(s) visible' % (world, torch.cuda.device_count()))
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
        os.environ['PARASITOID_DEVICE'] = str(local)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    import warnings
    warnings.simplefilter('ignore', RuntimeWarning)

    wd, days = PM.get_wind_data(os.path.join(ROOT, 'parasitoids_amd', 'data', 'kalbar'), 30, '00:00')
    pm = PopModel(wd, days, domain_info=(10000.0, args.rad_res), r_number=130000,
                  mode=args.mode, device=local)
    props = proposals(np.random.default_rng(1000 + rank), args.warmup + args.evals)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for hp, dp, dl, mu in props[:args.warmup]:
        pm.evaluate(hp, dp, dl, mu, 30)
    fence()
    t0 = time.perf_counter()
    t_pm = 0.0
    for hp, dp, dl, mu in props[args.warmup:]:
        stats = pm.evaluate(hp, dp, dl, mu, 30)      # chain_stats synchronises
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        out = {'metric': 'Bayes_Run model evaluations/hour (Kalbar, 18 days)',
               'value': round(world * args.evals / dt * 3600.0, 1), 'unit': 'evaluations/hour',
               'n_gpus': dist.get_world_size() if world > 1 else 1, 'evals': args.evals, 'warmup': args.warmup,
               'ms_per_eval': round(dt / args.evals * 1e3, 3), 'higher_is_better': True,
               'scaling': 'weak', 'dtype': 'f64', 'data': 'kalbarwind.txt (reference data)',
               'config': {'workload': 'pop_model body: 18 x prob_mass + get_populations, R=%d '
                                      '(N=%d), one chain per GPU, %s mode'
                                      % (args.rad_res, 2 * args.rad_res + 1, args.mode),
                          'fft_len': pm.solver.fft_len, 'last_day_total': round(stats[-1][1], 3)}}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_sample(wd, days, args.rad_res)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
