export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for g in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $g | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmc_picks/$n -o pmc -- python $R/scripts/time_picks2.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_picks/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "find_peaks" in row["Kernel_Name"]:
            acc["find_peaks"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc["find_peaks"].items()):
    # 4 inputs x 6 launches each: print per-input means (launch order: zeros, ramp, noise, envelope)
    n = len(v) // 4
    print("%-24s" % c, [round(sum(v[i*n:(i+1)*n]) / max(n,1)) for i in range(4)])
PY
