"""Average of every PMC counter per kernel name from a rocprofv3 counter_collection.csv:  pmc_avg.py dir [name-filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    if flt and flt not in r["Kernel_Name"]:
        continue
    e = per[r["Kernel_Name"][:100]][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k, cs in per.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print("   %-28s %14.0f (avg of %d)" % (c, v / n, n))
