"""Sum of a PMC counter per kernel name from a rocprofv3 counter_collection.csv:  pmc_sum.py dir COUNTER"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2]:
        e = per[r["Kernel_Name"][:110]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%6d launches  avg %10.1f KB  %s" % (n, v / n, k))
