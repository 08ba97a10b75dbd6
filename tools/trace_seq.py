"""Kernel sequence of one phase of a bench step from a rocprofv3 --kernel-trace CSV: start offset, duration, queue, name.
usage: trace_seq.py trace.csv <first-kernel substring> <last-kernel substring> [step index]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem_conv_fwd' in r['Kernel_Name']]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -3
a, b = idx[which], idx[which + 1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
i0 = next(i for i, r in enumerate(step) if sys.argv[2] in r['Kernel_Name'])
i1 = max(i for i, r in enumerate(step) if sys.argv[3] in r['Kernel_Name'])
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n.replace('void ', '')[:90]
for r in step[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), short(r['Kernel_Name'])))
