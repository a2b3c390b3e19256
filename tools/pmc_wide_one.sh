cd /tmp && export TMPDIR=/tmp
for shape in "272 194" "1088 194"; do
set -- $shape
tag=${1}x${2}
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcw_$tag -o w -- python3 $GRAFT_REPO_ROOT/tools/probe_wide_one.py $1 $2 3 2>&1 | tail -1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcw2_$tag -o w -- python3 $GRAFT_REPO_ROOT/tools/probe_wide_one.py $1 $2 3 2>&1 | tail -1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcw*_*")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "em_wide" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(d, {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
