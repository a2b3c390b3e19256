# GPU box: counters of the chain's kernels (one pass per group), summarised per kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chain
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o c -- python3 $GRAFT_REPO_ROOT/tools/probe_chain_sample.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -o c -- python3 $GRAFT_REPO_ROOT/tools/probe_chain_sample.py > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o c -- python3 $GRAFT_REPO_ROOT/tools/probe_chain_sample.py > $OUT/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o c -- python3 $GRAFT_REPO_ROOT/tools/probe_chain_sample.py > $OUT/w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o c -- python3 $GRAFT_REPO_ROOT/tools/probe_chain_sample.py > $OUT/s.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_chain/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "sb::" in k:
            acc[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/pmc_chain/summary.json", "w"), indent=1)
for k, cs in out.items():
    print(k, {c: ("%.3g" % v) for c, v in sorted(cs.items())})
for f in glob.glob("gpurun_out/pmc_chain/s/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12: print(row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
PY
