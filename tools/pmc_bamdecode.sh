cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bam_pmc; rm -rf $OUT; mkdir -p $OUT
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/$tag -o b -- python3 $GRAFT_REPO_ROOT/tools/bench_bamdecode.py 4e6 --no-cpu-baseline > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
from strawberry_amd import _lib
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/bam_pmc/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "sb::bam_" in k:
            acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
out["_build_id"] = _lib.load().sbgpu_build_id().decode()
out["_note"] = "tools/bench_bamdecode.py 4e6 (917 988 080 stream bytes): FETCH_SIZE / WRITE_SIZE in KB; HBM bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE"
json.dump(out, open("gpurun_out/bam_pmc/r04_bamdecode_pmc_summary.json", "w"), indent=1)
for k, cs in out.items():
    if k.startswith("_"): continue
    print(k, {c: "%.4g" % v["mean_per_dispatch"] for c, v in sorted(cs.items())})
PY
