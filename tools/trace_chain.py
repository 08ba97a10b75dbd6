"""Kernel-by-kernel listing of one phase of a bench step from a rocprofv3 kernel trace: per queue, start offset,
duration and the idle gap since the previous kernel of the same queue.  trace_chain.py trace.csv [which_step] [t_lo_ms] [t_hi_ms]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('stem_conv_fwd')]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = idx[which], idx[which + 1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n.replace('void ', '').replace('sbl_mfma_gemm_kernel', 'G').replace('sbl_skinny_gemm_kernel', 'SK')[:60]
last_end = {}
gaps = collections.defaultdict(float); busy = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in step:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    q = r.get('Queue_Id', '?')
    if s / 1e3 < lo or s / 1e3 >= hi:
        last_end[q] = e
        continue
    gap = s - last_end.get(q, s)
    last_end[q] = e
    gaps[q] += max(gap, 0); busy[q] += e - s; cnt[q] += 1
    print("q%-3s %9.1f us  dur %7.1f  gap %6.1f  grid %-8s %s" % (q, s, e - s, gap, r.get('Grid_Size', r.get('Grid_Size_X', '')), short(r['Kernel_Name'])))
for q in busy:
    print("queue %s: %d kernels, busy %.2f ms, gaps %.2f ms" % (q, cnt[q], busy[q] / 1e3, gaps[q] / 1e3))
