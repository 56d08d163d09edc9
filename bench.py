#!/usr/bin/env python3
"""Headline benchmark: grid-days/sec of the day-chain FFT convolution solver on a
nominal 4096^2 fp64 domain (BASELINE.json metric, config 3).

One "step" = one 30-day stack: state FFT, then per day  kernel scatter + forward
FFT -> spectral product -> inverse FFT -> threshold statistics / boundary flag ->
(flagged) truncate + re-FFT, all on the device with inputs (kernel COO triplets)
resident in HBM.  N > 1: one process per GPU, every rank runs its own replica
stack (weak scaling, no data-path collective).

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks
itself (one child process per GPU, RCCL rendezvous on 127.0.0.1); under
`torch.distributed.run` it is one of the ranks.  The launching parent never touches the GPU.

Prints ONE JSON line on rank 0.  Besides the contract's fields it carries
  parity            device vs CPU oracle on the first days of the benchmarked stack itself,
  exact_torus_mode  the same stack with exact reference-torus results ('auto' mode),
  real_wind         BASELINE config 3a/3b: Carnarvon wind, R=2048, 30 days, device chain rate,
  bayes             the second BASELINE metric: MCMC samples/hour on the Kalbar data.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROF_EVERY = 7          # co-prime with the 30 days of a stack: every day position gets sampled

# SURVEY 8d's normative model of an UNFUSED pipeline, W_day = 96 P^2 (P = reference torus
# N + K//2), split over this implementation's launch classes.  Reported per class as
# `model_GBps` and for the whole chain as `alg_GBps_whole_chain`; NOT what `roofline` uses.
MODEL_P2 = {
    'row_fwd': 16.0, 'col_fwd_a': 8.0, 'col_fwd_b': 8.0, 'col_inv_a': 40.0, 'col_inv_b': 8.0,
    'row_inv': 24.0, 'refft_pred': 0.0,
    'col_inv_a_x2': 80.0, 'col_inv_a_x4': 160.0, 'col_inv_a_x8': 320.0,
    'row_inv_x2': 48.0, 'row_inv_x4': 96.0, 'row_inv_x8': 192.0, 'col_tail': 0.0,
}
DAYS_PER_LAUNCH = {'col_inv_a_x2': 2, 'col_inv_a_x4': 4, 'col_inv_a_x8': 8,
                   'row_inv_x2': 2, 'row_inv_x4': 4, 'row_inv_x8': 8}
# `_xn`: any other number of chained days per launch (a solver whose previous run raised no flag opens
# with windows of up to 32 days: one 30-day window for this stack); the library counts the grid-days its timed
# launches covered (ps_prof_read_days) and main() fills in the average
VARIABLE_DAYS = ('col_inv_a_xn', 'row_inv_xn')
# kernel class -> (kernel symbol prefixes in the rocprofv3 PMC summaries under profiles/, days per
# launch or None).  The multi-day launches of the full-column pipeline are ONE kernel each; the
# summaries tell them apart by the bytes they write (scripts/hbm_traffic.py: size_rank).
PMC_NAME = {'row_inv': (('void k_row_inv',), None), 'col_inv_b': (('void k_col<1',), None),
            'col_inv_a': (('void k_col_fused<', 'void k_colfull_day<'), None)}
for _n in (2, 4, 8):
    PMC_NAME['col_inv_a_x%d' % _n] = (('void k_col_fused_multi<false, %d,' % _n, 'void k_colfull<'), _n)
    PMC_NAME['row_inv_x%d' % _n] = (('void k_row_inv_rsp<', 'void k_row_inv_rs2<'), _n)
PMC_NAME['col_inv_a_xn'] = (('void k_colfull<', 'void k_colfull_dual<'), 0)   # 0: the cluster with the most dispatches
PMC_NAME['row_inv_xn'] = (('void k_row_inv_rsp<', 'void k_row_inv_rs2<'), 0)


def launch_bytes(cls, N, fft_len, direct, kernel_rows_bytes=0.0, days=None):
    """ALGORITHMIC HBM bytes of ONE launch of this implementation's kernel class: what the
    launch has to read and write given that its intermediates stay in LDS --
    the compulsory traffic, which the PMC counters confirm (`roofline.traffic`).
    S = one half spectrum [fft_len][ld] complex128."""
    ld = (fft_len // 2 + 1 + 7) // 8 * 8
    S = fft_len * ld * 16.0
    F = N * N * 8.0
    nd = days if days else DAYS_PER_LAUNCH.get(cls, 1)
    if cls.startswith('col_inv_a'):
        # state in, state out, nd first-inverse-sub-pass outputs; the kernels' side is either
        # their live row-pass rows (direct sum / full-column pipeline) or nd intermediate spectra
        return (2 + nd) * S + (kernel_rows_bytes * nd if direct else nd * S)
    if cls.startswith('row_inv'):
        return nd * (S + F)
    return {'col_inv_b': 2 * S, 'col_fwd_a': 2 * S, 'col_fwd_b': 2 * S,
            'row_fwd': None, 'refft_pred': None}.get(cls)


def chain_compulsory_bytes(N, K, fl, nd, kernels, per_step, kern, num_cu=256):
    """Compulsory HBM bytes of ONE stack as this implementation launches it: the sum over
    launch classes of `launch_bytes` x launches per step, plus what the classes without a byte model move
    -- kernel staging (zeroed band, scatter, row pass of the live rows), `set_state` (record memset +
    the state's column pass) and the short launches for the columns taken out of a chained pass.
    `per_step`: {class: launches per step}; `kern`: the per-class table with `bytes_per_launch`."""
    ld = (fl // 2 + 1 + 7) // 8 * 8
    H = fl // 2 + 1
    S = fl * ld * 16.0
    F = N * N * 8.0
    parts = {}
    for k, n in per_step.items():
        b = kern.get(k, {}).get('bytes_per_launch')
        if b and n:
            parts[k] = n * b
    live = [(int(k.row.max()) - int(k.row.min()) + 1) if k.nnz else 0 for k in kernels]
    nnz = sum(int(k.nnz) for k in kernels)
    # staging block: zero the live band, scatter the triplets, row pass reads the band and writes live rows x H
    parts['kernel_staging'] = sum(live) * K * 8.0 * 2 + nnz * 24.0 + sum(live) * H * 16.0
    parts['set_state'] = F + S
    rem = H % num_cu
    if 0 < rem * 3 <= num_cu and per_step.get('col_tail'):
        parts['col_tail'] = 4.0 * nd * rem * fl * 16.0
    return sum(parts.values()), parts


def pmc_chain_traffic():
    """HBM bytes of one hinted stack from the latest committed PMC summary (`__chain__` entry written by
    scripts/hbm_traffic.py: the dispatches between two `set_state` markers), or (None, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic_pmc.json')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])
    if not files:
        return None, None
    for e in json.load(open(files[-1])):
        if e.get('kernel') == '__chain__':
            return (e['fetch_corrected_MB'] + e['write_size_MB']) * 1024 * 1024, os.path.basename(files[-1])
    return None, None


def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the latest committed rocprofv3 PMC
    summary (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE);
    None when no summary covers the kernel.  The PMC pass is a separate rocprofv3 run
    (scripts/hbm_traffic.py); it cannot be collected inside this process."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic_pmc.json')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])   # r01_v10 after r01_v9
    if not files or kernel_class not in PMC_NAME:
        return None, None
    names, days = PMC_NAME[kernel_class]
    best = None
    entries = json.load(open(files[-1]))
    pmc_traffic.provenance = next((e for e in entries if e.get('kernel') == '__provenance__'), None)
    for e in entries:
        if not e['kernel'].startswith(names) or e.get('write_size_MB') is None:
            continue
        if kernel_class.startswith('col_inv_a_x') and e['kernel'].startswith('void k_colfull<') and ', true, 0>' not in e['kernel']:
            continue      # the forward / inverse / product instances of k_colfull are other classes
        if days is None and e.get('size_groups', 1) != 1:
            continue      # a kernel whose dispatches differ in size: not a single-day class
        if days and e['kernel'].startswith(('void k_colfull<', 'void k_row_inv_rsp<', 'void k_row_inv_rs2<')):
            # the clusters of this kernel's dispatches by bytes written are its 2-, 4-, 8-day
            # launches in that order; a summary that does not hold all three cannot be attributed
            if e.get('size_groups') != 3 or e.get('size_rank') != (2, 4, 8).index(days):
                continue
        if best is None or e['dispatches'] > best['dispatches']:
            best = e
    if best is None:
        return None, None
    return (best['fetch_corrected_MB'] + best['write_size_MB']) * 1024 * 1024, os.path.basename(files[-1])


pmc_traffic.provenance = None


def lib_sha256():
    import hashlib
    try:
        from parasitoids_amd import _lib
        return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()
    except OSError:
        return None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rad-res', type=int, default=2048, help='R; domain N = 2R+1')
    ap.add_argument('--kshape', type=int, default=2049)
    ap.add_argument('--ndays', type=int, default=30)
    ap.add_argument('--mode', default='fast', choices=['fast', 'exact', 'fold', 'auto'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-days', type=int, default=6)   # ~10 s of one host core
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the exact_torus_mode / real_wind / bayes sub-records')
    ap.add_argument('--rehearse', action='store_true',
                    help='launcher/rendezvous rehearsal without device work (CPU test of the N>1 '
                         'control flow with BENCH_BACKEND=gloo); prints value null')
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- launcher
def launch_ranks(n, argv, script=None):
    """Start n ranks of this script, one per GPU, and wait for them.  Runs in a parent that has
    made no torch.cuda / HIP call (and imports neither): children are fresh interpreters, never
    an exec of this one.  Returns the exit code (non-zero if any rank failed)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0:
                rc = rc or code
                deadline = deadline or time.time() + 30     # a dead rank leaves the others in a barrier
        if deadline and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    return rc


# --------------------------------------------------------------------------- CPU oracle leg
def cpu_baseline(state, kernels, K, ndays_sample):
    """Oracle (CPU restatement of CalcSol.get_solutions) on a bounded sample of the
    same workload: state FFT + `ndays_sample` day steps, scipy.fft, one thread.
    Returns (record, raw oracle fields of those days) -- the fields are what `parity`
    compares the device records with."""
    from oracle import calcsol as OC
    N = state.shape[0]
    ms = np.array([K, K])
    t0 = time.perf_counter()
    hat = OC.fft2(state, ms)
    t_init = time.perf_counter() - t0
    fields = []
    dt = 0.0
    for n in range(ndays_sample):
        t0 = time.perf_counter()
        OC.fftconv2(hat, kernels[n].tocsr())
        A, flag = OC.ifft2(hat, [N, N])
        OC.r_small_vals(A, prob_model=True)
        if flag:
            hat = OC.fft2(A, ms)
        dt += time.perf_counter() - t0
        fields.append(A.toarray())
    rec = {'value': ndays_sample / dt, 'unit': 'grid-days/s', 'cores': 1, 'kind': 'port',
           'sample': '%d of the %d day steps of the same stack (oracle/calcsol.py, scipy.fft c2c '
                     'at P=%d, 1 thread; state FFT %.1fs not counted)'
                     % (ndays_sample, len(kernels), N + K // 2, t_init),
           'host_cpus': os.cpu_count()}
    return rec, fields


def cold_run_record(hip_lib, state, kernels, K, nd, device):
    """What a single `Run.main`-style run sees: the headline steps run on a warm
    solver whose previous run raised no flag (one 30-day window).  `first_run`: a fresh solver's first
    chain -- plans, buffer allocation, kernel upload excluded, windows ramp 2, 4, 8, 16; `unhinted`: the
    same ramp on warm buffers (PS_NO_WINDOW_HINT), median of 3; `hinted`: one-window runs of the same solver."""
    s = hip_lib.HipSolve(state, [K, K], mode='fast', device=device, chain_only=True)
    s.set_kernels(kernels)
    s.sync()

    def timed():
        t0 = time.perf_counter()
        s.set_state(state)
        s.run_chain(0, nd, negval=1e-8, scale=1.0, renorm=True)
        s.sync()
        return time.perf_counter() - t0

    first = timed()
    s.set_option('PS_NO_WINDOW_HINT', 1)
    unh = sorted(timed() for _ in range(3))[1]
    s.set_option('PS_NO_WINDOW_HINT', 0)
    timed()
    hin = sorted(timed() for _ in range(3))[1]
    s.close()
    r = lambda t: {'ms': round(t * 1e3, 3), 'grid_days_per_s': round(nd / t, 1)}
    return {'first_run': r(first), 'unhinted': r(unh), 'hinted': r(hin),
            'note': 'first_run includes first-touch allocation of the solver\'s buffers; value/ms_per_step are hinted runs'}


def api_get_solutions_record(state, kernels, K, nd):
    """`CalcSol.get_solutions` at the headline size through the package's reference-shaped API
    (CalcSol.py:140-201): solver construction, kernel upload from host COO, the chain, and the COO export of
    every day the reference returns (CalcSol.py:198) -- everything the timed steps leave out."""
    from parasitoids_amd import CalcSol, globalvars
    old = globalvars.fft_mode
    globalvars.fft_mode = 'fast'
    try:
        modelsol = [state]
        days = list(range(nd + 1))
        t0 = time.perf_counter()
        CalcSol.get_solutions(modelsol, [None] + list(kernels), days, nd + 1, state.shape[0], np.array([K, K]))
        dt = time.perf_counter() - t0
    finally:
        globalvars.fft_mode = old
    nnz = sum(int(m.nnz) for m in modelsol[1:])
    return {'seconds': round(dt, 3), 'grid_days_per_s': round(nd / dt, 1), 'days': nd, 'mode': 'fast',
            'coo_entries_returned': nnz, 'coo_bytes_d2h': nnz * 16,
            'note': 'includes HipSolve construction, H2D of the kernel triplets and the D2H COO export of all days'}


def max_abs_vs(solver, fields):
    return float(max(np.abs(solver.dense(0, d) - f).max() for d, f in enumerate(fields)))


# --------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, args.gpus))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['PARASITOID_DEVICE'] = str(local)

    import torch
    import torch.distributed as dist
    # BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get('BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if not args.rehearse:
        if backend == 'nccl' and ndev < world:
            raise SystemExit('--gpus %d but only %d device(s) visible' % (world, ndev))
        if backend != 'nccl':
            local = local % max(1, ndev)
            os.environ['PARASITOID_DEVICE'] = str(local)
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl' and not args.rehearse:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            backend = backend if backend != 'nccl' else 'gloo'
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit('process group has %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus))
    on_gpu = backend == 'nccl' and not args.rehearse

    def gather_times(dt):
        """max over ranks (the contract's clock) and every rank's own time"""
        if world == 1:
            return dt, [dt]
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if on_gpu else 'cpu')
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per = [float(x.item()) for x in allt]
        return max(per), per

    def rank_identities():
        """every rank's device as the runtime names it, on rank 0: the driver can see that the N ranks
        of an RCCL job sat on N different GPUs"""
        ident = {'rank': rank, 'local_device': local, 'backend': backend, 'pid': os.getpid()}
        if not args.rehearse:
            try:
                pr = torch.cuda.get_device_properties(local)
                ident['name'] = pr.name
                ident['uuid'] = str(getattr(pr, 'uuid', None))
                ident['pci_bus_id'] = getattr(pr, 'pci_bus_id', None)
            except Exception as e:
                ident['error'] = str(e)
        if world == 1:
            return [ident]
        allv = [None] * world
        dist.all_gather_object(allv, ident)
        return allv

    def gather_parity(v):
        """every rank's own device-vs-oracle figure on rank 0 (a rank on the wrong device, or with a
        different result, cannot hide behind rank 0's)"""
        if world == 1:
            return [v]
        t = torch.tensor([v], dtype=torch.float64, device='cuda' if on_gpu else 'cpu')
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        return [float(x.item()) for x in allt]

    R, K, nd = args.rad_res, args.kshape, args.ndays
    N = 2 * R + 1
    P = N + K // 2

    if args.rehearse:
        dist.barrier() if world > 1 else None
        dt, per = gather_times(1.0 + 0.0 * rank)
        idents = rank_identities()
        mg = None
        if world > 1:
            import bench_extras
            mg = bench_extras.multi_gpu_record(rank, world, rehearse=True,       # configs 4 / 5 control flow, stand-in work
                                               fail_stage=os.environ.get('BENCH_FAIL_STAGE'))
            par = gather_parity(1e-19 * (rank + 1))
        if rank == 0:
            print(json.dumps({'metric': 'grid-days/sec on 4096^2 fp64 domain', 'value': None,
                              'unit': 'grid-days/s', 'n_gpus': dist.get_world_size() if world > 1 else 1,
                              'steps': args.steps, 'warmup': args.warmup, 'rehearsal': True,
                              'ranks_seen': len(per), 'backend': backend,
                              'world_size': dist.get_world_size() if world > 1 else 1, 'ranks': idents,
                              'multi_gpu': mg, 'parity': {'per_rank_max_abs': par} if world > 1 else None}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from parasitoids_amd import hip_lib, synthetic

    state, kernels, params = synthetic.make_stack(R=R, K=K, ndays=nd, seed=20240613)  # same stack on every rank
    solver = hip_lib.HipSolve(state, [K, K], mode=args.mode, device=local, chain_only=True)
    solver.set_kernels(kernels)

    def step():
        solver.set_state(state)
        solver.run_chain(0, nd, negval=1e-8, scale=1.0, renorm=True)

    def fence():
        solver.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    # correctness guard outside the timed region: mass conserved, moments add up
    stats = solver.chain_stats(0, nd)
    raw = solver.dense(0, nd - 1)                 # unthresholded last-day field
    idx = np.arange(N, dtype=np.float64)
    s = raw.sum()
    rowm = raw.sum(1)
    mr = (rowm * idx).sum() / s
    vr = (rowm * (idx - mr) ** 2).sum() / s
    exp_mr = R + sum(synthetic.moments(k)[1] - K // 2 for k in kernels)
    exp_vr = sum(synthetic.moments(k)[3] for k in kernels)
    check = os.environ.get('BENCH_SKIP_CHECK') is None    # kernel-timing experiments with stubbed phases only
    flagged = any(st.flag for st in stats)
    if check:
        assert s < 1.0 + 1e-9 and abs(stats[-1].sum + stats[-1].delta * stats[-1].nnz - 1.0) < 1e-12
        # Mass can leave the domain without raising the flag (many pad cells below 1e-8 each);
        # when it all stayed inside, mean and variance must add up exactly
        if not flagged and abs(s - 1.0) < 1e-9:
            assert abs(mr - exp_mr) < 1e-6 and abs(vr / exp_vr - 1) < 1e-7, (mr, exp_mr, vr, exp_vr)
        elif (R, K, nd) == (2048, 2049, 30):
            raise AssertionError('headline stack lost mass: sum %r flagged %r' % (s, flagged))
    del raw

    # HIP events bracket every PROF_EVERY-th launch of each kernel class inside the timed
    # region (bracketing all ~250 launches per stack costs ~9 % of the throughput)
    solver.prof_enable(os.environ.get('BENCH_NO_PROF') is None, every=PROF_EVERY)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_own = time.perf_counter() - t0
    prof = solver.prof_read()
    prof_days = solver.prof_days()
    prof_launches = solver.prof_launches()
    solver.prof_enable(False)
    dt, per_rank = gather_times(dt_own)
    idents = rank_identities()
    rank_parity = None
    if world > 1 and not args.no_cpu_baseline:
        # every rank checks ITS stack against the oracle (2 day steps, ~5 s of one host core each)
        _, of = cpu_baseline(state, kernels, K, 2)
        rank_parity = gather_parity(max_abs_vs(solver, of))
        del of

    if rank == 0:
        nranks = dist.get_world_size() if world > 1 else 1
        grid_days = nranks * args.steps * nd
        value = grid_days / dt
        kern = {}
        model_p2 = dict(MODEL_P2)
        days_of = dict(DAYS_PER_LAUNCH)
        for k in VARIABLE_DAYS:
            if prof.get(k, (0, 0))[1]:
                days_of[k] = prof_days[k] / prof[k][1]
                model_p2[k] = (40.0 if k.startswith('col') else 24.0) * days_of[k]
        direct = bool(solver.kernels_direct) or bool(solver.full_column)
        if direct:
            # compact kernels: the first forward column sub-pass (col_fwd_a, 8 P^2 per grid-day)
            # is evaluated inside the fused launch
            for k, n in days_of.items():
                model_p2[k] += 8.0 * n
            model_p2['col_inv_a'] += 8.0
        fl = solver.fft_len
        # the row-pass rows of one day kernel that hold anything (half spectra), averaged over the days
        live = [(int(k.row.max()) - int(k.row.min()) + 1) if k.nnz else 0 for k in kernels]
        krows = float(np.mean(live)) * (fl // 2 + 1) * 16.0
        for k, (ms, cnt) in prof.items():
            if cnt:
                avg = ms / cnt
                e = {'avg_ms': round(avg, 4), 'timed_launches': cnt,
                     'model_GBps': round(model_p2[k] * P * P / (avg * 1e-3) / 1e9, 1)}
                b = launch_bytes(k, N, fl, direct, krows, days_of.get(k))
                if k in days_of:
                    e['days_per_launch'] = round(days_of[k], 2)
                if b:
                    e['bytes_per_launch'] = b
                    e['hbm_GBps'] = round(b / (avg * 1e-3) / 1e9, 1)
                kern[k] = e
        if not kern:      # BENCH_NO_PROF diagnostic run
            print(json.dumps({'value': round(value, 3), 'ms_per_step': round(dt / args.steps * 1e3, 3),
                              'note': 'HIP-event profiling disabled'}))
            return
        # total time per class: multi-day launches are all timed, the others every PROF_EVERY-th
        # (only classes with a byte model: the HIP-event durations of `row_fwd` include the wait of the
        # low-priority kernel-transform launches for idle CUs behind the day passes, not work)
        cand = [k for k in kern if kern[k].get('hbm_GBps')] or list(kern)
        dom = max(cand, key=lambda k: kern[k]['avg_ms'] * kern[k]['timed_launches']
                  * (1 if k in days_of else PROF_EVERY))
        ach = kern[dom].get('hbm_GBps')
        traffic, traffic_src = pmc_traffic(dom) if (R, K, nd) == (2048, 2049, 30) else (None, None)
        out = {
            'metric': 'grid-days/sec on 4096^2 fp64 domain',
            'value': round(value, 3),
            'unit': 'grid-days/s',
            'n_gpus': nranks, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C3 synthetic stack: N=%d (nominal %d^2), %d Gaussian day kernels '
                                   'K=%d, reference torus P=%d, FFT size %d (%s mode), prob model, '
                                   'one replica stack per GPU' % (N, 2 * R, nd, K, P, solver.fft_len,
                                                                  args.mode),
                       'dom_len': N, 'ndays': nd, 'kshape': K, 'P': P, 'fft_len': solver.fft_len,
                       'kernels_direct': bool(solver.kernels_direct), 'full_column_pipeline': bool(solver.full_column)},
            'per_rank_grid_days_per_s': [round(args.steps * nd / t, 2) for t in per_rank],
            'world_size': nranks, 'backend': backend if world > 1 else None, 'ranks': idents,
            'distinct_devices': len(set((i.get('uuid'), i.get('pci_bus_id'), i.get('local_device')) for i in idents)),
            # SURVEY 8d's normative whole-chain figure: the unfused model's 96 P^2 per grid-day
            # times the measured rate.  A model rate, not a bandwidth measurement.
            'alg_bytes_per_grid_day': 96.0 * P * P,
            'alg_GBps_whole_chain': round(value / nranks * 96.0 * P * P / 1e9, 1),
            # roofline of the dominant launch class: achieved = the bytes one launch of THIS
            # kernel has to move (launch_bytes) / its HIP-event duration; frac = achieved / peak
            # (<= 1); traffic = bytes the PMC counters saw for one launch (committed rocprofv3
            # summary, separate run), traffic_GBps = traffic / the duration measured here
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None,
                         'traffic': traffic,
                         'traffic_GBps': (round(traffic / (kern[dom]['avg_ms'] * 1e-3) / 1e9, 1)
                                          if traffic else None),
                         'traffic_frac': (round(traffic / (kern[dom]['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                          if traffic else None),
                         # SURVEY 8d's unfused model for the same launch (48 P^2 per grid-day of the fused day
                         # pass): what an implementation that streamed every intermediate would have to move
                         'survey_model_GBps': kern[dom]['model_GBps'],
                         'survey_model_frac': round(kern[dom]['model_GBps'] / HBM_PEAK_GBS, 4),
                         'bytes_per_launch': kern[dom].get('bytes_per_launch'),
                         'days_per_launch': days_of.get(dom, 1),
                         'avg_launch_ms': kern[dom]['avg_ms'],
                         'traffic_source': ('profiles/%s (committed rocprofv3 --pmc summary, not '
                                            'collected in this run)' % traffic_src) if traffic_src else None,
                         # which build the committed counters were collected from, and whether it is the
                         # library timed here (a stale summary shows as false)
                         'traffic_provenance': ({'git_head': pmc_traffic.provenance.get('git_head'),
                                                 'lib_sha256': pmc_traffic.provenance.get('lib_sha256'),
                                                 'this_lib_sha256': lib_sha256(),
                                                 'same_library': pmc_traffic.provenance.get('lib_sha256') == lib_sha256()}
                                                if (traffic_src and pmc_traffic.provenance) else None)},
            'kernels': kern,
        }
        # Whole-chain roofline: the compulsory bytes of the stack as launched / the
        # measured step time, next to the PMC bytes of one stack.  SURVEY 8d's 96 P^2 model above
        # (`alg_*`, `survey_model_*`) is superseded for this code: fusion keeps the state column in LDS
        # for all 30 days and never writes kernel spectra, so that model counts bytes nobody moves.
        per_step = {k: prof_launches.get(k, 0) / float(args.steps) for k in prof_launches}
        cb, cparts = chain_compulsory_bytes(N, K, fl, nd, kernels, per_step, kern)
        step_s = dt / args.steps
        ctraffic, csrc = pmc_chain_traffic() if (R, K, nd) == (2048, 2049, 30) else (None, None)
        out['roofline']['chain'] = {
            'bytes_per_step': cb, 'achieved': round(cb / step_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(cb / step_s / 1e9 / HBM_PEAK_GBS, 4),
            'bytes_by_class': {k: round(v) for k, v in cparts.items()},
            'launches_per_step': {k: round(v, 2) for k, v in per_step.items() if v},
            'traffic': ctraffic,
            'traffic_frac': round(ctraffic / step_s / 1e9 / HBM_PEAK_GBS, 4) if ctraffic else None,
            'traffic_source': ('profiles/%s' % csrc) if csrc else None,
            'hbm_roof_grid_days_per_s': round(nd / (cb / (HBM_PEAK_GBS * 1e9)), 1),
            'note': 'compulsory bytes of the stack as launched (state in/out per chained pass, one intermediate '
                    'and one N x N record per day, live kernel rows, staging, set_state); survey_model_* / alg_* '
                    'are SURVEY 8d\'s unfused 96 P^2 model, superseded'}
        if dom.startswith('col_inv_a') and solver.full_column:
            # the full-column day pass is two length-L complex transforms and one product per column
            # and day: what its time is spent on (nominal 5 L log2 L flops per transform); the fp64
            # vector peak is the guide's 78.6 TFLOP/s.  Neither roof is near: see DESIGN.md 4.1c.
            import math
            ncol = fl // 2 + 1
            flops = days_of.get(dom, 1) * ncol * (2 * 5.0 * fl * math.log2(fl) + 6.0 * fl)
            tf = flops / (kern[dom]['avg_ms'] * 1e-3) / 1e12
            out['roofline']['fp64_valu'] = {'flops_per_launch': flops, 'achieved': round(tf, 2), 'peak': 78.6,
                                            'unit': 'TFLOP/s', 'frac': round(tf / 78.6, 4)}
        if nranks == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'], ofields = cpu_baseline(state, kernels, K, args.cpu_days)
            # parity at the benchmarked configuration: the oracle's raw day fields against the
            # device records of the very stack that was timed (fast mode), outside the timed region
            par = {'days': len(ofields), 'tolerance': 1e-12,
                   'max_abs_%s' % args.mode: max_abs_vs(solver, ofields)}
            if args.mode == 'fast':
                # the same stack with exact reference-torus results ('auto' mode: direct
                # transform on P, or the folded linear convolution when P is awkward)
                solver.close()
                s2 = hip_lib.HipSolve(state, [K, K], mode='auto', device=local, chain_only=True)
                s2.set_kernels(kernels)
                for i in range(3):
                    if i == 1:
                        s2.sync(); t1 = time.perf_counter()
                    s2.set_state(state)
                    s2.run_chain(0, nd, negval=1e-8, scale=1.0, renorm=True)
                s2.sync()
                out['exact_torus_mode'] = {'value': round(2 * nd / (time.perf_counter() - t1), 3),
                                           'unit': 'grid-days/s', 'mode': s2.mode, 'fft_len': s2.fft_len}
                par['max_abs_exact_torus'] = max_abs_vs(s2, ofields)
                s2.close()
            par['ok'] = all(v < 1e-12 for k, v in par.items() if k.startswith('max_abs'))
            out['parity'] = par
            # the same figures where a reader of only `roofline` / `config` finds them
            out['roofline']['parity'] = dict(par)
            out['config']['parity_max_abs'] = max(v for k, v in par.items() if k.startswith('max_abs'))
            out['config']['parity_ok'] = par['ok']
            del ofields
            if args.mode == 'fast' and not args.no_extras:
                try:
                    out['cold_first_run'] = cold_run_record(hip_lib, state, kernels, K, nd, local)
                    out['api_get_solutions'] = api_get_solutions_record(state, kernels, K, nd)
                except Exception as e:          # a sub-record must not cost the headline line
                    out['cold_first_run'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if rank_parity is not None:
            out['parity'] = {'days': 2, 'tolerance': 1e-12, 'per_rank_max_abs': rank_parity,
                             'ok': all(v < 1e-12 for v in rank_parity)}
        if nranks == 1 and not args.no_extras and (R, K, nd) == (2048, 2049, 30):
            solver.close()
            import bench_extras
            for key, fn in (('real_wind', bench_extras.real_wind_record),
                            ('release', lambda device=None: bench_extras.release_record()),
                            ('prefix_split', lambda device=None: bench_extras.prefix_split_record(device=device)),
                            ('bayes', bench_extras.bayes_record)):
                try:
                    out[key] = fn(device=local)
                except Exception as e:          # a sub-record must not cost the headline line
                    out[key] = {'error': '%s: %s' % (type(e).__name__, e)}
    mg = None
    if world > 1 and not args.no_extras and (R, K, nd) == (2048, 2049, 30):
        # configs 4 and 5 over the ranks of this job (every rank takes part in the gathers)
        solver.close()
        import bench_extras
        try:
            mg = bench_extras.multi_gpu_record(rank, world, device=local)
        except Exception as e:
            mg = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0:
        if mg is not None:
            out['multi_gpu'] = mg
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
