"""Timeline report of one bench step from a rocprofv3 --kernel-trace CSV: phase boundaries, per-phase kernel mix."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem_conv_fwd' in r['Kernel_Name']]
if len(sys.argv) > 2 and sys.argv[2] == "median":
    # among the last 16 complete steps (bench.py replays every coin pattern twice at the end) the one whose kernel count is
    # closest to the middle of their range: a step with the expected number of own-arg-max coins
    cand = list(range(max(0, len(idx) - 17), len(idx) - 1))
    cnts = [idx[j + 1] - idx[j] for j in cand]
    mean = (min(cnts) + max(cnts)) / 2.0      # mid-range: the stratified patterns are symmetric about the expectation
    which = min(cand, key=lambda j: (abs(idx[j + 1] - idx[j] - mean), -j)) - len(idx)
else:
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = idx[which], idx[which + 1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cov = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce: cov += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
cov += ce - cs
starts = [int(rows[i]['Start_Timestamp']) for i in idx]
per = [(b_ - a_) / 1e6 for a_, b_ in zip(starts[:-1], starts[1:])][-4:]
print("step-to-step period over the last %d steps: mean %.2f ms (%s) - the teacher-forcing coin patterns differ in decoder work; the step below is one of them" % (len(per), sum(per) / len(per), " ".join("%.1f" % p for p in per)))
print("step wall %.2f ms, %d kernels, union busy %.2f ms, sum %.2f ms" % ((t1 - t0) / 1e6, len(step), cov / 1e6, sum(e - s for s, e in ev) / 1e6))
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n.replace('void ', '').replace('sbl_mfma_gemm_kernel', 'G').replace('sbl_skinny_gemm_kernel', 'SK')[:72]
def first(name): return next(((int(r['Start_Timestamp']) - t0) / 1e6 for r in step if name in r['Kernel_Name']), None)
def last(name): return next(((int(r['End_Timestamp']) - t0) / 1e6 for r in reversed(step) if name in r['Kernel_Name']), None)
marks = [("frontend fwd", 0.0), ("encoder fwd", last('avgpool_fwd') or first('avgpool')), ("decoder fwd", first('embed_pe')),
         ("decoder bwd", first('smoothed_ce_bwd') or first('smoothed_ce')), ("frontend bwd", first('avgpool_bwd')), ("tail", last('stem_wgrad')), ("end", (t1 - t0) / 1e6)]
print(marks)
for (nm, lo), (_, hi) in zip(marks[:-1], marks[1:]):
    if lo is None or hi is None: continue
    w = [r for r in step if lo <= (int(r['Start_Timestamp']) - t0) / 1e6 < hi]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in w:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        agg[short(r['Kernel_Name'])][0] += 1; agg[short(r['Kernel_Name'])][1] += d
    print("=== %s [%.2f, %.2f) = %.2f ms wall: %d kernels, sum %.2f ms" % (nm, lo, hi, hi - lo, len(w), sum(v[1] for v in agg.values()) / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 10]:
        print("   %-74s %5d %8.2f ms  avg %6.1f us" % (k, v[0], v[1] / 1e3, v[1] / v[0]))
