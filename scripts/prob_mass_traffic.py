"""HBM traffic of the prob_mass batch (bench_extras.py prob_mass) from two rocprofv3 counter passes
(FETCH_SIZE, WRITE_SIZE; units and the gfx950 correction as in hbm_traffic.py): per kernel, the
bytes all its dispatches of ONE 18-day batch moved, and the total -- SURVEY 8d asks for this next
to the rect-probs/s figure to show the path is not bandwidth-limited.
usage: prob_mass_traffic.py FETCH_DIR WRITE_DIR NBATCHES OUT.json"""
import json
import sys
from collections import defaultdict

from hbm_traffic import read


def main(fd, wd, nb, out):
    nb = int(nb)
    f = read(fd, 'FETCH_SIZE')
    w = read(wd, 'WRITE_SIZE')
    per = defaultdict(lambda: {'dispatches_per_batch': 0.0, 'fetch_corrected_MB_per_batch': 0.0, 'write_MB_per_batch': 0.0})
    for (name, grid), vals in f.items():
        e = per[name]
        e['dispatches_per_batch'] += len(vals) / nb
        e['fetch_corrected_MB_per_batch'] += 2.0 * sum(vals) / 1024.0 / nb
    for (name, grid), vals in w.items():
        per[name]['write_MB_per_batch'] += sum(vals) / 1024.0 / nb
    kernels = [dict(kernel=k, **{a: round(b, 4) for a, b in v.items()}) for k, v in
               sorted(per.items(), key=lambda kv: -(kv[1]['fetch_corrected_MB_per_batch'] + kv[1]['write_MB_per_batch']))]
    tot = sum(k['fetch_corrected_MB_per_batch'] + k['write_MB_per_batch'] for k in kernels) * 1024 * 1024
    res = {'summary': {'total_bytes_per_batch': tot, 'batches_profiled': nb,
                       'largest': [{'kernel': k['kernel'][:48],
                                    'MB_per_batch': round(k['fetch_corrected_MB_per_batch'] + k['write_MB_per_batch'], 2)}
                                   for k in kernels[:4]]},
           'kernels': kernels}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['summary']))


if __name__ == '__main__':
    main(*sys.argv[1:5])
