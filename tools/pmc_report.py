"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: counter_collection.csv each) of the
bench step, with FETCH_SIZE scaled by factors CALIBRATED on kernels of known traffic (tools/pmc_calibrate.py run under the same
two passes): one factor for float4 streaming reads, one per operand-loader kind of the tile engine.
  pmc_report.py step_fetch_dir step_write_dir cal_fetch_dir cal_write_dir known.json out.csv calibration.txt"""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        e = per[r["Kernel_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
        if r.get("End_Timestamp"):
            e[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per


sf, sw, cf, cw, known_p, out_p, cal_p = sys.argv[1:8]
known = json.load(open(known_p))
cfe, cwr = load(cf, "FETCH_SIZE"), load(cw, "WRITE_SIZE")
factors, lines = {}, []
for name, k in known.items():
    def last(per):       # the LAST launch of the matching kernel is the flushed, steady one; the counter CSV sums them: use the mean
        m = [(kn, v) for kn, v in per.items() if re.search(k["kernel_re"], kn)]
        return (sum(v[1] for _, v in m) / max(1, sum(v[0] for _, v in m))) if m else None
    fr, fw = last(cfe), last(cwr)
    if fr:
        factors[name] = k["read"] / (fr * 1024.0)
        lines.append("%-18s known read %8.1f MB  FETCH_SIZE %8.1f MB  factor %.3f | known write %8.1f MB  WRITE_SIZE %8.1f MB  ratio %.3f"
                     % (name, k["read"] / 1e6, fr * 1024 / 1e6, factors[name], k["write"] / 1e6, (fw or 0) * 1024 / 1e6,
                        k["write"] / ((fw or float("nan")) * 1024.0)))
stream = [v for n_, v in factors.items() if n_.startswith("stream16")]
f_stream = sum(stream) / len(stream) if stream else 2.0
f_kc, f_mc = factors.get("tile_kc", 2.0), factors.get("tile_mc", 2.0)
lines.append("applied: streaming kernels x%.3f; tile-engine launches with k-contiguous operands x%.3f, m-contiguous x%.3f, mixed: mean"
             % (f_stream, f_kc, f_mc))
open(cal_p, "w").write("\n".join(lines) + "\n")


def factor(kernel):
    if "sbl_" in kernel and ("gemm" in kernel or "conv_pm" in kernel or "wgrad_group" in kernel):
        kc = len(re.findall(r"DenseKC|ConvGatherKC|ConvGatherPM|DenseKCTap", kernel))
        mc = len(re.findall(r"DenseMC|ConvGatherMC|SegMC", kernel))
        if "wgrad_group" in kernel:
            return f_mc
        return (kc * f_kc + mc * f_mc) / max(1, kc + mc)
    return f_stream


fe, wr = load(sf, "FETCH_SIZE"), load(sw, "WRITE_SIZE")
with open(out_p, "w") as out:
    out.write("kernel,launches,avg_FETCH_SIZE_KB_raw,fetch_factor,avg_FETCH_SIZE_KB_calibrated,avg_WRITE_SIZE_KB,avg_duration_us\n")
    for k, (n, v, us) in sorted(fe.items(), key=lambda kv: -kv[1][1]):
        w = wr.get(k, [1, 0.0, 0.0])
        f = factor(k)
        out.write('"%s",%d,%.1f,%.3f,%.1f,%.1f,%.2f\n' % (k, n, v / n, f, f * v / n, w[1] / max(w[0], 1), us / n))
