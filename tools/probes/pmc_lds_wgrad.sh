cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_lds -o r -- python3 tools/probes/patch_wgrad.py > gpurun_out/pmc_lds.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_lds/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQ_INSTS_LDS': n[k] += 1
for k, v in acc.items():
    if 'wgrad' in k or 'ConvGatherMC' in k:
        print(k, n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
PY
rm -rf gpurun_out/pmc_lds
