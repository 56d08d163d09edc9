"""Summarise a rocprofv3 --pmc CSV (counter_collection) per kernel: launches and the
mean of every counter per launch.  usage: pmc_summary.py DIR"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split('(')[0]
    return name[-60:]


def main(d):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        print('no counter_collection.csv under', d)
        return 1
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row['Kernel_Name'])
                acc[k][row['Counter_Name']] += float(row['Counter_Value'])
                launches[k].add(row['Dispatch_Id'])
    for k in sorted(acc, key=lambda k: -len(launches[k])):
        n = len(launches[k])
        vals = '  '.join('%s=%.4g' % (c, v / n) for c, v in sorted(acc[k].items()))
        print('%-62s n=%-5d %s' % (k, n, vals))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
