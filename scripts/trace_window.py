import csv, glob, os, sys
from collections import defaultdict
path=sys.argv[1]
files = glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
ev=[]
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
ev.sort()
# the last chain: find last big gap > 2 ms? use last 47 ms window ending at last event
end=ev[-1][1]
# locate the start of the final chain: last k_set.. use time window of 48 ms before the end minus trailing copies
win=[e for e in ev if e[0] >= end-50_000_000]
tot=defaultdict(float); cnt=defaultdict(int)
for s,e,n in win:
    tot[n]+=e-s; cnt[n]+=1
span=win[-1][1]-win[0][0]; busy=sum(tot.values())
print('window %.2f ms busy %.2f ms'%(span/1e6,busy/1e6))
for n in sorted(tot,key=lambda k:-tot[k])[:28]:
    print('%-52s %5d  %9.1f us  %8.1f each'%(n[-52:],cnt[n],tot[n]/1e3,tot[n]/1e3/cnt[n]))
# gaps
prev=win[0][1]; gaps=[]
for s,e,n in win[1:]:
    if s>prev: gaps.append((s-prev,n))
    prev=max(prev,e)
gaps.sort(reverse=True)
print('idle total %.2f ms; largest gaps:'%(sum(g for g,_ in gaps)/1e6), [(round(g/1e3),n[-24:]) for g,n in gaps[:12]])
