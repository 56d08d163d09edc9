"""Per-element cost of the day-chain kernels over FFT sizes (occupancy study): runs bench.py's
synthetic stack at domain sizes whose fast torus is exactly a chosen size, in the tiled and in
the full-column pipeline, and prints ns per L^2 element for every kernel class."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [int(x) for x in (sys.argv[1:] or ['4096', '4608', '5120', '5184', '6144'])]
out = {}
for L in SIZES:
    R = int(L * 0.8) // 2
    N = 2 * R + 1
    K = 2 * (L - N) + 1
    for tp in ('0', '1'):
        env = dict(os.environ, PS_TPIPE=tp)
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-extras', '--no-cpu-baseline',
                            '--rad-res', str(R), '--kshape', str(K), '--steps', '3', '--warmup', '1'],
                           env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(L, tp, 'failed', p.stderr[-400:])
            continue
        d = json.loads(p.stdout.strip().splitlines()[-1])
        assert d['config']['fft_len'] == L, (L, d['config'])
        row = {'ms_per_step': d['ms_per_step'], 'ns_per_elem_day': d['ms_per_step'] * 1e6 / 30 / (L * L)}
        for k, v in d['kernels'].items():
            nd = v.get('days_per_launch') or (int(k.rsplit('_x', 1)[1]) if '_x' in k and k.rsplit('_x', 1)[1].isdigit() else 1)
            row[k] = round(v['avg_ms'] * 1e6 / nd / (L * L), 4)
        out['%d_tp%s' % (L, tp)] = row
        print(L, 'tpipe' if tp == '1' else 'tiled', json.dumps(row), flush=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'size_sweep.json'), 'w'), indent=1)
