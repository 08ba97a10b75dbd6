"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
busy fraction = MFMA busy cycles summed over the 1024 SIMDs / (GPU-active cycles x 1024); GRBM_GUI_ACTIVE comes summed over
the 8 XCDs (333 us -> 6.29 M, i.e. 0.79 M cycles = 2.36 GHz per XCD), hence the divisor act x 128.
pmc_mfma_report.py dir out.csv"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: {"n": 0, "mfma": 0.0, "act": 0.0, "us": 0.0, "wave": 0.0, "wait_inst": 0.0, "wait_any": 0.0})
seen = set()
for r in csv.DictReader(open(f)):
    e = per[r["Kernel_Name"]]
    v = float(r["Counter_Value"])
    c = r["Counter_Name"]
    if c == "SQ_VALU_MFMA_BUSY_CYCLES":
        e["mfma"] += v
        e["n"] += 1
        e["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    elif c == "GRBM_GUI_ACTIVE": e["act"] += v
    elif c == "SQ_WAVE_CYCLES": e["wave"] += v
    elif c == "SQ_WAIT_INST_ANY": e["wait_inst"] += v
    elif c == "SQ_WAIT_ANY": e["wait_any"] += v
with open(sys.argv[2], "w") as out:
    out.write("kernel,launches,avg_duration_us,avg_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE,mfma_busy_fraction_of_1024_SIMDs,issue_stall_fraction_of_wave_cycles,wait_fraction_of_wave_cycles\n")
    for k, e in sorted(per.items(), key=lambda kv: -kv[1]["mfma"]):
        if e["n"] == 0 or e["mfma"] == 0: continue
        busy = e["mfma"] / (e["act"] * 128) if e["act"] else float("nan")
        out.write('"%s",%d,%.1f,%.0f,%.0f,%.3f,%.3f,%.3f\n' % (k, e["n"], e["us"] / e["n"], e["mfma"] / e["n"], e["act"] / e["n"], busy,
                                                         e["wait_inst"] / e["wave"] if e["wave"] else float("nan"), e["wait_any"] / e["wave"] if e["wave"] else float("nan")))
