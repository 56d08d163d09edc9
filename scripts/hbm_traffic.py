"""Build profiles/*_hbm_traffic_pmc.json from two rocprofv3 counter passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950).  FETCH_SIZE is reported in KB and counts 64 B per
128-B request on gfx950, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is KB.
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
                key = (row['Kernel_Name'].split('(')[0], int(row['Grid_Size']))
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
        e = {'kernel': key[0], 'grid': key[1], 'dispatches': n, 'fetch_size_MB': fetch_mb,
             'fetch_corrected_MB': 2.0 * fetch_mb,
             'write_size_MB': (w[key][0] / max(1, len(w[key][1])) / 1024.0) if key in w else None}
        res.append(e)
    json.dump(res, open(out, 'w'), indent=1)
    for e in res[:8]:
        print(e)


if __name__ == '__main__':
    main(*sys.argv[1:4])
