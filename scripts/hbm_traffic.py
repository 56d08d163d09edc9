"""Build profiles/*_hbm_traffic_pmc.json from two rocprofv3 counter passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950).  FETCH_SIZE is reported in KB and counts 64 B per
128-B request on gfx950, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is KB.
Dispatches are grouped by (kernel, grid, LDS size): the multi-day launches of one kernel differ only
in their LDS size (PS_LDS_TAG, csrc/rs_cfg.h); `days` is the rank of the LDS size among the groups
of that kernel and grid (1, 2, 4, 8 days per launch).
usage: hbm_traffic.py FETCH_DIR WRITE_DIR OUT.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d, counter):
    acc = defaultdict(lambda: [0.0, set(), 0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                key = (row['Kernel_Name'].split('(')[0], int(row['Grid_Size']), int(row.get('LDS_Block_Size') or 0))
                acc[key][0] += float(row['Counter_Value'])
                acc[key][1].add(row['Dispatch_Id'])
                acc[key][2] = int(row['Grid_Size'])
    return acc


def main(fd, wd, out):
    f = read(fd, 'FETCH_SIZE')
    w = read(wd, 'WRITE_SIZE')
    res = []
    for key in sorted(f, key=lambda k: -len(f[k][1])):
        n = len(f[key][1])
        fetch_mb = f[key][0] / n / 1024.0          # KB -> MB per dispatch
        ldss = sorted({k[2] for k in f if k[:2] == key[:2]})
        e = {'kernel': key[0], 'grid': key[1], 'lds_block': key[2],
             'lds_rank': ldss.index(key[2]), 'lds_groups': len(ldss),
             'dispatches': n, 'fetch_size_MB': fetch_mb,
             'fetch_corrected_MB': 2.0 * fetch_mb,
             'write_size_MB': (w[key][0] / max(1, len(w[key][1])) / 1024.0) if key in w else None}
        res.append(e)
    json.dump(res, open(out, 'w'), indent=1)
    for e in res[:8]:
        print(e)


if __name__ == '__main__':
    main(*sys.argv[1:4])
