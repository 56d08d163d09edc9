"""Timeline of a window of a rocprofv3 kernel trace: start offset, duration and the idle gap before
every kernel.  usage: trace_timeline.py DIR_OR_CSV [first] [count]"""
import csv
import glob
import os
import sys


def main(path, first=-200, count=200):
    files = [path] if path.endswith('.csv') else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    ev = []
    for f in files:
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0],
                       r.get('LDS_Block_Size', '')))
    ev.sort()
    win = ev[first:][:count]
    t0 = win[0][0]
    prev_end = t0
    busy = gap_tot = 0
    for s, e, n, lds in win:
        gap = s - prev_end
        busy += e - s
        gap_tot += max(gap, 0)
        print('%9.1f us  dur %8.1f  gap %6.1f  lds %6s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, lds, n[-60:]))
        prev_end = max(prev_end, e)
    print('window %.1f us  busy %.1f  idle %.1f' % ((win[-1][1] - t0) / 1e3, busy / 1e3, gap_tot / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], *(int(x) for x in sys.argv[2:4]))
