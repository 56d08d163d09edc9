"""Per-stack time breakdown from a rocprofv3 kernel trace of bench.py: totals per kernel,
normalised by the number of 30-day stacks (counted through the inverse row pass), plus idle time
between consecutive kernels.  usage: trace_breakdown.py DIR_OR_CSV [ndays]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(path, ndays=30):
    files = [path] if path.endswith('.csv') else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    ev = []
    for f in files:
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
    ev.sort()
    # steady state: from the first inverse row pass of the third stack on
    inv = [i for i, e in enumerate(ev) if 'k_row_inv' in e[2]]
    nst = len(inv) // ndays
    if nst < 3:
        print('too few stacks in the trace')
        return
    lo = inv[2 * ndays] - 8
    ev = ev[max(lo, 0):]
    stacks = sum(1 for e in ev if 'k_row_inv' in e[2]) / ndays
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for s, e, n in ev:
        tot[n] += e - s
        cnt[n] += 1
    span = ev[-1][1] - ev[0][0]
    busy = sum(tot.values())
    print('stacks %.2f  span %.3f ms/stack  busy %.3f ms/stack  idle %.1f %%' %
          (stacks, span / 1e6 / stacks, busy / 1e6 / stacks, 100 * (1 - busy / span)))
    for n in sorted(tot, key=lambda k: -tot[k]):
        print('%-48s %6.1f launches/stack %9.1f us/stack  %7.1f us each' %
              (n[-48:], cnt[n] / stacks, tot[n] / 1e3 / stacks, tot[n] / 1e3 / cnt[n]))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
