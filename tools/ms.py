import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("   ms_per_step %.3f  clips/s %.1f" % (d["ms_per_step"], d["value"]))
