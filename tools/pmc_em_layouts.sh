#!/bin/bash
# GPU box: instructions per algorithmic FMA of the tile kernels, by isoform count (= by register-tile layout).
# usage (repo root): bash tools/pmc_em_layouts.sh > profiles/r05_em_layout_instr.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_em_layouts
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc -o em -- python3 $REPO/tools/pmc_em_layouts.py $OUT/runs.json > $OUT/log.txt 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
runs = json.load(open(out + "/runs.json"))
disp = collections.OrderedDict()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "em_fused_kernel" in row["Kernel_Name"]:
            disp.setdefault(int(row["Dispatch_Id"]), {"kernel": row["Kernel_Name"].split("(")[0]})[row["Counter_Name"]] = float(row["Counter_Value"])
ids = sorted(disp)
# every subset launched its kinds' kernels 3 times, in order: walk the dispatches
print("# C3 split by isoform count; per subset: the em_fused_kernel launches of three solves (all kinds of the subset summed),")
print("# VALU wave-instructions per algorithmic FMA-equivalent (2 per element and iteration: E-step and M-step), the matrix-pipe")
print("# share, and how busy the vector pipe was per resident wave (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)")
print("%-8s %7s %9s %6s %12s %10s %9s %9s %9s %9s" % ("isoforms", "loci", "elements", "capped", "VALU instr", "per FMA", "MFMA/VALU", "SALU/VALU", "VALU busy", "waiting"))
pos = 0
for r in runs:
    nk = sum(1 for k in r["kinds"][:5] if k)
    take = ids[pos:pos + 3 * nk]
    pos += 3 * nk
    if not take:
        break
    tot = collections.Counter()
    for d in take:
        for k, v in disp[d].items():
            if k != "kernel":
                tot[k] += v
    fma = 2.0 * r["elem_iters"] * 3 / 64.0     # wave-instructions if every lane did useful FMAs
    name = "%d" % r["niso"][0] if r["niso"][0] == r["niso"][1] else "%d-%d" % tuple(r["niso"])
    print("%-8s %7d %9d %6d %12.0f %10.2f %9.3f %9.2f %9.2f %9.2f" % (name, r["loci"], r["elements"], r["capped"], tot["SQ_INSTS_VALU"],
          tot["SQ_INSTS_VALU"] / max(fma, 1), tot["SQ_INSTS_MFMA"] / max(tot["SQ_INSTS_VALU"], 1), tot["SQ_INSTS_SALU"] / max(tot["SQ_INSTS_VALU"], 1),
          tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1), tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1)))
PY
