"""CPU tests of the host-side logic of the product package: wind I/O (bit-exact
against the reference's golden arrays), the C-ABI library (loads, exports every
declared symbol), FFT program emulation, parameter handling."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wind_interpolation_bit_exact(golden, golden_dir):
    from parasitoids_amd import ParasitoidModel as PM
    g = golden('g3_hprob_wind')
    for site, st in (('kalbar', '00:00'), ('carnarvonearl', '00:30')):
        wd, days = PM.get_wind_data(os.path.join(golden_dir, 'data', site), 30, st)
        assert list(g[site + '_days']) == days
        assert np.array_equal(np.array([wd[d].sum(0) for d in days]), g[site + '_wind_sum'])
        assert np.array_equal(wd[days[0]], g[site + '_wind_first'])
        assert np.array_equal(wd[days[-1]], g[site + '_wind_last'])
        assert np.array_equal(wd[days[3]], g[site + '_wind_mid'])
        # reference test_get_wind_data properties (tests/test_ParsitoidModel.py:117-143)
        raw, days_raw = PM.read_wind_file(os.path.join(golden_dir, 'data', site))
        assert days_raw == days
        assert wd[days[0]].shape[0] == 30 * raw[days[0]].shape[0]
        for key in raw:
            assert np.all(np.sqrt(wd[key][:, 0]**2 + wd[key][:, 1]**2) == wd[key][:, 2])
    with pytest.raises(ValueError):
        PM.get_wind_data(os.path.join(golden_dir, 'data', 'kalbar'), 30, '01:00')


def test_library_exports_every_declared_symbol():
    from parasitoids_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'parasitoid_hip.h')).read()
    declared = set(re.findall(r'\b(ps_[a-z0-9_]+)\s*\(', header))
    declared -= {'ps_day_stats'}
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), 'missing export ' + name
        assert name in _lib.SIGNATURES, 'no ctypes signature for ' + name
    assert lib.ps_version() >= 100


def test_no_gpu_is_import_error():
    """Without a GPU the device modules must fail loudly (ImportError), never fall back."""
    from parasitoids_amd import _lib
    lib = _lib.load()
    if lib.ps_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(ImportError):
        import parasitoids_amd.hip_lib  # noqa: F401
    from parasitoids_amd import ParasitoidModel as PM
    with pytest.raises(ImportError):
        PM.prob_mass(1, {1: np.zeros((48, 3))}, (1,) * 7, (1, 1, 0), (1, 1, 0), 1, 1, 100.0, 4)


def test_fft_program_emulation():
    """The in-LDS FFT program (fft_core.h), emulated thread by thread on the host,
    against a long-double DFT: all register radices, generic primes, both layouts."""
    exe = os.path.join(ROOT, 'tests', 'host', '_build', 'fft_emul')
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(['g++', '-O2', '-std=c++17', '-o', exe,
                    os.path.join(ROOT, 'tests', 'host', 'fft_emul.cpp')], check=True)
    out = subprocess.run([exe, '1', '2', '3', '5', '7', '8', '9', '16', '72', '82', '364', '573',
                          '1121', '1155', '2592'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    worst = float(out.stdout.strip().split()[-1])
    assert worst < 5e-15
    # register-resident three-stage transforms (fft_rs.h): Stockham indexing, padded exchange
    out = subprocess.run([exe, 'rs'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert float(out.stdout.strip().split()[-1]) < 5e-15


def test_params_defaults_and_parsing(tmp_path):
    """Run.Params: defaults (reference Run.py:47-91), presets (:96-138), tuple orders
    (:374-384), command line / config handling (:218-352) incl. the reference's quirks."""
    from parasitoids_amd.Run import Params
    p = Params(config=None)
    assert p.dataset == 'kalbar' and p.site_name == 'data/kalbar' and p.start_time == '00:00'
    assert p.r_dur == 1 and p.r_number == 130000 and p.r_start is None
    assert p.domain_info == (10000.0, 400) and p.interp_num == 30 and p.n_periods == 30
    hparams, Dp, Dl, mu_r, npd, rad_dist, rad_res = p.get_model_params()
    assert hparams == (1., 1.263, 3.913, 7.302, 2.614, 23.999, 2.350)
    assert Dp == (171.82, 144.58, 0.253) and Dl == (7.096, 7.260, 0.000)
    assert (mu_r, npd, rad_dist, rad_res) == (1.179, 30, 10000.0, 400)
    assert p.get_wind_params() == ('data/kalbar', 30, '00:00')
    p.cmd_line_chg(['--carnarvon', '--pop', 'r_dur=1', 'domain_info=(8000.0,320)', 'ndays=3',
                    'Dparams=(4.0, 4.0, 0.)', 'r_number=7', '--no_output', 'cuda=False'])
    assert p.site_name == 'data/carnarvonearl' and p.start_time == '00:30'
    assert p.PROB_MODEL is False and p.r_dur == 1 and p.r_start == 0.354
    assert p.r_number == 40000                      # reference Run.py:294-295 never assigns it
    assert p.domain_info == (8000.0, 320) and p.ndays == 3 and p.Dparams == (4.0, 4.0, 0.0)
    assert p.OUTPUT is False and p.CUDA is False and '_pop' in p.outfile
    assert p.uniform(1) == 1.0 and p.r_mthd() == p.uniform
    with pytest.raises(ValueError):
        p.cmd_line_chg(['--bogus'])
    with pytest.raises(LookupError):
        p.cmd_line_chg(['bogus=1'])
    with pytest.raises(ValueError):
        p.cmd_line_chg(['ndays=abc'])
    cfg = tmp_path / 'config.txt'
    cfg.write_text('# local configuration\ndataset = carnarvon  # preset\nmu_r = 2.5\nplot = False\n')
    q = Params(config=str(cfg))
    assert q.dataset == 'carnarvon' and q.r_dur == 5 and q.mu_r == 2.5 and q.PLOT is False


def test_mcmc_log_densities():
    '''closed-form checks of the prior/likelihood helpers against scipy.stats'''
    from scipy import stats
    from parasitoids_amd import mcmc
    assert abs(mcmc._lg_gamma(3.1, 26, 0.15) - stats.gamma.logpdf(3.1, 26, scale=1 / 0.15)) < 1e-12
    assert abs(mcmc._lg_beta(0.3, 5, 1) - stats.beta.logpdf(0.3, 5, 1)) < 1e-12
    assert abs(mcmc._lg_normal(0.7, 1.0, 1.0) - stats.norm.logpdf(0.7, 1.0, 1.0)) < 1e-12
    sd = 1 / np.sqrt(0.3)
    assert abs(mcmc._lg_truncnormal(7.0, 6, 0.3, 0, 9)
               - stats.truncnorm.logpdf(7.0, (0 - 6) / sd, (9 - 6) / sd, loc=6, scale=sd)) < 1e-12
    assert mcmc._lg_truncnormal(9.5, 6, 0.3, 0, 9) == float('-inf')
    assert abs(mcmc._lg_poisson(28, 30) - stats.poisson.logpmf(28, 30)) < 1e-12
    obs = np.array([[0, 2], [5, 0]]); rate = np.array([[0.0, 1.5], [4.0, 0.2]])
    assert abs(mcmc.poisson_loglik(obs, rate) - stats.poisson.logpmf(obs, rate)[rate > 0].sum()) < 1e-12
    assert mcmc.poisson_loglik(np.array([1]), np.array([0.0])) == float('-inf')


def test_locinfo_loader_kalbar():
    """parasitoids_amd.Data_Import.LocInfo on the CSV/text fixtures of the Kalbar campaign:
    no golden exists (the reference's loader cannot run in this image), so the arrays are
    checked against direct reductions of the fixture files and against the geometry."""
    import csv
    import pandas as pd
    from parasitoids_amd.Data_Import import LocInfo, latlong_tocoord, DEFAULT_DATA_DIR
    R = 400
    li = LocInfo('kalbar', (-27.947131, 152.584171), (10000.0, R))
    N = 2 * R + 1

    def rows(name):
        with open(os.path.join(DEFAULT_DATA_DIR, name)) as f:
            return list(csv.DictReader(f))

    # fields: A is the release field, B..G the sentinels; every cell inside the domain
    assert sorted(li.field_cells) == list('ABCDEFG') and li.sent_ids == list('BCDEFG')
    for k, c in li.field_cells.items():
        assert c.ndim == 2 and c.shape[1] == 2 and c.min() >= 0 and c.max() < N
        assert li.field_sizes[k] == len(c) > 10
        # cell centres of a field lie inside the bounding box of its polygon
        xy = np.asarray(li.field_polys[k].vertices[:-1])      # Path: last vertex closes the polygon
        x = (c[:, 1] - R) * 25.0
        y = (R - c[:, 0]) * 25.0
        assert x.min() >= xy[:, 0].min() - 1e-9 and x.max() <= xy[:, 0].max() + 1e-9
        assert y.min() >= xy[:, 1].min() - 1e-9 and y.max() <= xy[:, 1].max() + 1e-9
    # the release point is inside field A
    assert any((c == [R, R]).all() for c in li.field_cells['A'])
    assert abs(latlong_tocoord((0, 0), 0.0, 1.0)[0] - 6378100 * np.pi / 180) < 1e-6
    # release grid: one cell per line of the text file, rotation keeps the distance to the origin
    assert li.grid_cells.shape == (75, 2) and li.grid_cells.min() >= 0 and li.grid_cells.max() < N
    d = np.hypot(li.grid_data['xcoord'], li.grid_data['ycoord'])
    assert 0 < d.max() < 500 and li.grid_samples.max() == 1.0 and li.grid_samples.min() > 0
    # sentinel emergence: E = females + males, every row of the sheet lands in the array
    sen = rows('kalbar_sentinels_raw.csv')
    assert li.sentinel_emerg[0].shape == (6, 10)
    assert li.sentinel_emerg[0].sum() == sum(int(r['Efemales']) + int(r['Emales']) for r in sen)
    by_field = {k: sum(int(r['Efemales']) + int(r['Emales']) for r in sen if r['Field ID (jpgs)'] == k)
                for k in li.sent_ids}
    assert [by_field[k] for k in li.sent_ids] == li.sentinel_emerg[0].sum(axis=1).tolist()
    assert li.collection_datesPR[0].days == 18
    # release-field emergence: points on the two axes through the release point are dropped
    rel = [r for r in rows('kalbar_releasefield_raw.csv')
           if float(r['ycoord']) - 200 != 0 and -float(r['xcoord']) + 300 != 0]
    assert li.release_emerg[0].sum() == sum(int(r['Efemales']) + int(r['Emales']) for r in rel)
    assert li.release_emerg[0].shape == (len(li.emerg_grids[0]), 10) == (15, 10)
    grid = {tuple(c) for c in li.grid_cells.tolist()}
    assert all(tuple(int(v) for v in g) in grid for g in li.emerg_grids[0])
    assert li.release_collection[0].max() == 1.0 and li.release_collection[0].min() > 0
    # grid counts: every non-zero observation of the sheet is found on a grid point
    obs = rows('kalbar_adult_counts_field_A.csv')
    assert li.grid_obs.shape == (75, 3) and [t.days for t in li.grid_obs_datesPR] == [2, 5, 8]
    assert li.grid_obs.sum() == sum(int(r['num hayati']) for r in obs)
    # cardinal directions
    assert [a.shape[0] for a in li.card_obs] == [4, 4] and [t.days for t in li.card_obs_datesPR] == [2, 8]
    for a, name in zip(li.card_obs, ('kalbar_cardinal_15mar05.csv', 'kalbar_cardinal_21mar05.csv')):
        assert a.sum() == sum(int(r['num adults']) for r in rows(name))


def test_bench_roofline_is_a_bandwidth_fraction():
    """`roofline.achieved` is built from the bytes a launch of the dominant kernel class has to
    move (state in/out, outputs, inverse-pass field) -- never more than the unfused model's
    share, so a launch cannot show more than the HBM peak by bookkeeping alone."""
    import bench
    N, fl = 4097, 5184
    ld = (fl // 2 + 1 + 7) // 8 * 8
    S = fl * ld * 16.0
    assert bench.launch_bytes('col_inv_a_x4', N, fl, True, 0.0) == 6 * S
    assert bench.launch_bytes('col_inv_a_x4', N, fl, False) == 10 * S
    assert bench.launch_bytes('col_inv_a', N, fl, False) == 4 * S
    assert bench.launch_bytes('row_inv', N, fl, True) == S + N * N * 8.0
    assert bench.launch_bytes('col_inv_b', N, fl, True) == 2 * S


def test_committed_bench_line_keeps_the_contract():
    """The newest committed `bench.py` line of round 2+ (profiles/rNN_vMM_bench.json, produced
    on the GPU box) carries every field the bench contract names, with consistent values."""
    import glob
    import json
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[2-9]*_bench.json')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])
    if not files:
        pytest.skip('no round-2 bench line committed yet')
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity'):
        assert k in d, k
    assert d['unit'] == 'grid-days/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'f64' and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['frac'] <= 1.0
    if r['traffic'] is not None:
        assert r['traffic_frac'] <= 1.0
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1
    assert d['parity']['ok'] is True
    # value and ms_per_step describe the same run: 30-day stacks
    assert abs(d['value'] - d['n_gpus'] * d['config']['ndays'] / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']


def test_hbm_traffic_splits_multi_day_launches_by_bytes_written(tmp_path):
    """scripts/hbm_traffic.py: dispatches of one kernel and grid whose WRITE_SIZE falls into
    separate clusters (the 2-, 4-, 8-day launches of the chained full-column pass) become separate
    entries with size_rank 0, 1, 2; FETCH_SIZE follows by dispatch order and is doubled."""
    import csv
    import json as _json
    import subprocess
    import sys
    hdr = ['Dispatch_Id', 'Grid_Size', 'Kernel_Name', 'LDS_Block_Size', 'Counter_Name', 'Counter_Value']
    days = [2, 4, 8, 8, 8, 2, 4, 8, 8, 8]
    for name, counter, per_day in (('f', 'FETCH_SIZE', 40e3), ('w', 'WRITE_SIZE', 200e3)):
        d = tmp_path / name
        d.mkdir()
        with open(d / 'x_counter_collection.csv', 'w', newline='') as fh:
            wr = csv.writer(fh)
            wr.writerow(hdr)
            for i, n in enumerate(days):
                for xcd in range(2):     # a counter arrives in several rows per dispatch
                    wr.writerow([10 + i, 1000, 'void k_colfull<16, 18, 18, true, 0>(ColFullArgs)', 0, counter,
                                 (n * per_day + 100e3) / 2])
            wr.writerow([99, 512, 'k_other(int)', 0, counter, 1024.0])
    out = tmp_path / 'o.json'
    subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'hbm_traffic.py'), str(tmp_path / 'f'),
                    str(tmp_path / 'w'), str(out)], check=True, capture_output=True)
    res = [e for e in _json.load(open(out)) if e['kernel'].startswith('void k_colfull<')]
    assert [e['size_rank'] for e in res] == [0, 1, 2] and all(e['size_groups'] == 3 for e in res)
    assert [e['dispatches'] for e in res] == [2, 2, 6]
    for e, n in zip(res, (2, 4, 8)):
        assert abs(e['write_size_MB'] - (n * 200e3 + 100e3) / 1024.0) < 1e-9
        assert abs(e['fetch_corrected_MB'] - 2 * (n * 40e3 + 100e3) / 1024.0) < 1e-9


def test_sentinel_field_sums_with_empty_fields():
    """A sentinel field polygon may contain no cell centre (coarse rad_res): the per-field sums
    must equal the reference's per-field slice sums (Bayes_funcs.py:116-144) wherever the empty
    fields sit -- last, in the middle, first, several in a row."""
    from parasitoids_amd import Bayes_funcs as BF
    rng = np.random.default_rng(11)
    for lens in ([3, 0], [3, 0, 2], [0, 4, 1], [2, 0, 0, 3, 0], [0, 0], [5], [1, 1, 0, 0]):
        n = sum(lens)
        vals = rng.random((4, n))
        b = np.cumsum([0] + lens)
        fr = dict(starts=b[:-1], empty=np.array([l == 0 for l in lens]))
        want = np.array([[v[b[i]:b[i + 1]].sum() for i in range(len(lens))] for v in vals])
        got = BF._field_sums(vals, fr)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-15, atol=0)
    assert BF._field_sums(np.zeros((0, 3)), dict(starts=np.array([0, 2]), empty=np.array([False, False]))).shape == (0, 2)


def test_two_role_pass_compiles_without_scratch(tmp_path):
    """k_colfull_dual lives at the 168 registers a 12-wave workgroup may use; a few spilled
    registers cost more than its second role gains.  Every instance the library
    enables (RsDual::ok) must compile without scratch -- a change of the butterflies or of the
    compiler that pushes one over the edge shows up here, not as a slow kernel on the GPU box."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    csrc = os.path.join(ROOT, 'parasitoids_amd', 'csrc')
    out = tmp_path / 'coldual.s'
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-munsafe-fp-atomics',
                    '-ffp-contract=on', '-I' + csrc, '-I' + os.path.join(ROOT, 'include'), '--cuda-device-only',
                    '-S', '-o', str(out), os.path.join(csrc, 'ps_coldual.hip')], check=True, capture_output=True)
    text = out.read_text()
    names = re.findall(r'\.name:\s+(\S*k_colfull_dual\S*)', text)
    spills = re.findall(r'\.vgpr_spill_count:\s+(\d+)', text)
    vgprs = re.findall(r'\.vgpr_count:\s+(\d+)', text)
    kern = [(n, int(v), int(sp)) for n, v, sp in zip(re.findall(r'\.name:\s+(\S+)', text), vgprs, spills) if 'k_colfull_dual' in n]
    assert len(kern) == len(names) >= 8
    for n, v, sp in kern:
        # the two radix-20 sizes the library enables spill 1-6 registers (measured: still well ahead of the
        # single-role pass -- 30-day launch at 5120 with 6 spilled: 3.45 against 4.42 ms); everything else none
        assert sp <= (6 if 'Li20E' in n else 0) and v <= 168, (n, v, sp)


def test_every_option_is_documented():
    """The knob table of the library (csrc/ps_config.h) and the table a caller reads list the
    same keys: an option added without a word on what it does fails here."""
    import re
    keys = re.findall(r'X\("(PS_[A-Z0-9_]+)"', open(os.path.join(ROOT, 'parasitoids_amd', 'csrc', 'ps_config.h')).read())
    assert len(keys) == len(set(keys)) >= 50
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    section = design[design.index('### 6.2 Tuning'):design.index('## 7. Multi-GPU')]
    missing = [k for k in keys if k not in section]
    assert not missing, missing
