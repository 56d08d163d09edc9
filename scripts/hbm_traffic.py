"""Build profiles/*_hbm_traffic_pmc.json from two rocprofv3 counter passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950).  FETCH_SIZE is reported in KB and counts 64 B per
128-B request on gfx950, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is KB.
Dispatches are grouped by (kernel, grid).  The multi-day launches of the full-column pipeline are
ONE kernel and one grid whatever the number of days (rocprofv3 reports neither the kernel arguments
nor the dynamic LDS size), so a group whose WRITE_SIZE values fall into separate clusters (more
than 25 % apart) is split into them; `size_rank` / `size_groups` say which cluster an entry is
(ascending bytes: 2, 4, 8 days per launch).  Both passes run the same deterministic command, so the
k-th dispatch of a kernel in the FETCH pass is the k-th in the WRITE pass.
usage: hbm_traffic.py FETCH_DIR WRITE_DIR OUT.json [PROVENANCE.txt]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d, counter):
    acc = defaultdict(dict)   # (kernel, grid) -> {dispatch id: value}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                key = (row['Kernel_Name'].split('(')[0], int(row['Grid_Size']))
                did = int(row['Dispatch_Id'])
                acc[key][did] = acc[key].get(did, 0.0) + float(row['Counter_Value'])
    return {k: [v[i] for i in sorted(v)] for k, v in acc.items()}   # in dispatch order


def read_ordered(d, counter):
    """[(dispatch id, kernel, value KB)] of the whole pass, in dispatch order"""
    acc = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                did = int(row['Dispatch_Id'])
                name = row['Kernel_Name'].split('(')[0]
                acc[did] = (name, acc.get(did, (name, 0.0))[1] + float(row['Counter_Value']))
    return [(i, acc[i][0], acc[i][1]) for i in sorted(acc)]


def chain_segments(d, counter, marker='k_set_int2'):
    """KB per stack: the dispatches between two `set_state` markers (ps_solver_set_state_coo launches
    k_set_int2 exactly once) -- everything one `set_state` + `run_chain` step of bench.py enqueues.  The
    first segment (solver construction) and the last (followed by the read-back) are left out."""
    segs, cur = [], None
    for _, name, v in read_ordered(d, counter):
        if name == marker:
            if cur is not None:
                segs.append(cur)
            cur = 0.0
        elif cur is not None:
            cur += v
    return segs[1:] if len(segs) > 2 else segs       # [construction | stack ...] ; the open last one is never appended


def clusters(vals):
    """indices of `vals` grouped into clusters of similar size (neighbours within 25 %), ascending"""
    order = sorted(range(len(vals)), key=lambda i: vals[i])
    out = [[order[0]]]
    for i in order[1:]:
        if vals[i] > 1.25 * vals[out[-1][-1]] and vals[i] > 1.0:
            out.append([])
        out[-1].append(i)
    return out


def provenance(path):
    """{'git_head': ..., 'lib_sha256': ..., 'date': ...} from collect_profiles.sh's provenance.txt"""
    rec = {'kernel': '__provenance__'}
    try:
        for line in open(path):
            k, _, v = line.strip().partition(' ')
            if k:
                rec[k] = v
    except OSError:
        pass
    return rec


def main(fd, wd, out, prov=None):
    f = read(fd, 'FETCH_SIZE')
    w = read(wd, 'WRITE_SIZE')
    res = []
    for key in sorted(f, key=lambda k: -len(f[k])):
        fv, wv = f[key], w.get(key)
        groups = [list(range(len(fv)))]
        if wv is not None and len(wv) == len(fv) and len(fv) > 1:
            groups = clusters(wv)
        for rank, idx in enumerate(groups):
            n = len(idx)
            fetch_mb = sum(fv[i] for i in idx) / n / 1024.0          # KB -> MB per dispatch
            e = {'kernel': key[0], 'grid': key[1], 'size_rank': rank, 'size_groups': len(groups),
                 'dispatches': n, 'fetch_size_MB': fetch_mb, 'fetch_corrected_MB': 2.0 * fetch_mb,
                 'write_size_MB': (sum(wv[i] for i in idx) / n / 1024.0) if wv is not None and len(wv) == len(fv)
                 else ((sum(wv) / len(wv) / 1024.0) if wv else None)}
            res.append(e)
    # one stack = one step of bench.py: whole-chain traffic next to bench.py's `roofline.chain`
    fs, ws = chain_segments(fd, 'FETCH_SIZE'), chain_segments(wd, 'WRITE_SIZE')
    if fs and ws:
        # the hinted stacks (one 30-day window) are the steps bench.py times: the last segments; median
        k = max(1, min(len(fs), len(ws)) - 1)
        med = lambda v: sorted(v)[len(v) // 2]
        res.append({'kernel': '__chain__', 'stacks': k, 'fetch_size_MB': med(fs[-k:]) / 1024.0,
                    'fetch_corrected_MB': 2.0 * med(fs[-k:]) / 1024.0, 'write_size_MB': med(ws[-k:]) / 1024.0,
                    'per_stack_fetch_corrected_MB': [round(2.0 * x / 1024.0, 1) for x in fs],
                    'per_stack_write_MB': [round(x / 1024.0, 1) for x in ws]})
    if prov:
        res.append(provenance(prov))     # which build these counters belong to (bench.py echoes it)
    json.dump(res, open(out, 'w'), indent=1)
    for e in res[:10]:
        print(e)


if __name__ == '__main__':
    main(*sys.argv[1:5])
