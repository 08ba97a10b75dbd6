import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem_wgrad' in r['Kernel_Name']]
a = idx[-3]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a-6:a + 40]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]
    print("%9.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), n))
