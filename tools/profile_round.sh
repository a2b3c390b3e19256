#!/bin/bash
# GPU box: the round's evidence.  Usage (repo root): bash tools/profile_round.sh r01
# Everything judged is collected under gpurun_out/$R/summary/ -- copy that directory's files into profiles/.
R=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$R
SUM=$OUT/summary
mkdir -p $SUM
cd $REPO
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $SUM/${R}_pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats in their own runs (the bench's own default command: 200 steps as a run of batches); every PMC group in its own run, never with a trace domain
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -o em -- python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline > $OUT/stats_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -o em -- python3 $REPO/bench.py --workload c2 --no-cpu-baseline > $OUT/stats_c2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -o em -- python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -o em -- python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq_c3 -o em -- python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
# the matrix pipe: instructions, busy cycles (north star: "MFMA-busy where used" -- v_mfma_f64_4x4x4 is the cross-lane reducer of every EM kernel)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_mfma_c3 -o em -- python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2 -o em -- python3 $REPO/bench.py --workload c2 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch_c2.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2 -o em -- python3 $REPO/bench.py --workload c2 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write_c2.log 2>&1
cd $REPO
python3 - "$OUT" "$SUM" "$R" <<'PY'
import collections, csv, glob, json, shutil, sys
out, summ, r = sys.argv[1:4]
for w in ("c3", "c2"):
    for f in glob.glob("%s/stats_%s/**/*kernel_stats.csv" % (out, w), recursive=True):
        shutil.copy(f, "%s/%s_%s_kernel_stats.csv" % (summ, r, w))
for w in ("c3", "c2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/pmc_*_%s/**/*counter_collection.csv" % w, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "sb::" in k:
                acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    import ctypes
    L = ctypes.CDLL("strawberry_amd/lib/libsbgpu.so")
    L.sbgpu_build_id.restype = ctypes.c_char_p
    summary = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    summary["_build_id"] = L.sbgpu_build_id().decode()     # bench.py quotes the traffic only for this build
    json.dump(summary, open("%s/%s_%s_pmc_summary.json" % (summ, r, w), "w"), indent=1)
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chain -o ch -- python3 $REPO/bench.py --workload c3-chain --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats_chain.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${tag}_chain -o ch -- python3 $REPO/bench.py --workload c3-chain --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_${tag}_chain.log 2>&1
done
# the chain step's timeline on the GPU: every kernel with the idle gap before it (tools/chain_gpu_gaps.py)
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace_chain -o ch -- python3 $REPO/bench.py --workload c3-chain --no-cpu-baseline --steps 6 --warmup 2 > $OUT/trace_chain.log 2>&1
python3 $REPO/tools/chain_gpu_gaps.py $(find $OUT/trace_chain -name "ch_results.db" | head -1) > $SUM/${R}_c3chain_timeline.txt 2>&1
# the wide-locus kernel on C3-T: stats and counters
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3t -o w -- python3 $REPO/bench.py --workload c3t --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats_c3t.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${tag}_c3t -o w -- python3 $REPO/bench.py --workload c3t --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_${tag}_c3t.log 2>&1
done
cd $REPO
for f in $(find $OUT/stats_chain -name "*kernel_stats.csv"); do cp $f $SUM/${R}_c3chain_kernel_stats.csv; done
for f in $(find $OUT/stats_c3t -name "*kernel_stats.csv"); do cp $f $SUM/${R}_c3t_kernel_stats.csv; done
python3 - "$OUT" "$SUM" "$R" <<'PY'
import collections, csv, ctypes, glob, json, sys
out, summ, r = sys.argv[1:4]
L = ctypes.CDLL("strawberry_amd/lib/libsbgpu.so"); L.sbgpu_build_id.restype = ctypes.c_char_p
for w, name in (("chain", "c3chain"), ("c3t", "c3t")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/pmc_*_%s/**/*counter_collection.csv" % w, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "sb::" in k:
                acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summary = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    summary["_build_id"] = L.sbgpu_build_id().decode()
    json.dump(summary, open("%s/%s_%s_pmc_summary.json" % (summ, r, name), "w"), indent=1)
PY
# the bench lines come after the counter passes: bench.py reads roofline.traffic from profiles/*_pmc_summary.json
cp $SUM/${R}_c3_pmc_summary.json $SUM/${R}_c2_pmc_summary.json $SUM/${R}_c3chain_pmc_summary.json $SUM/${R}_c3t_pmc_summary.json $REPO/profiles/ 2>/dev/null
(timeout 900 python bench.py 2>/dev/null | tail -1) > $SUM/${R}_bench_c3.json
(timeout 600 python bench.py --workload c2 2>/dev/null) > $SUM/${R}_bench_c2.json
(timeout 900 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null) > $SUM/${R}_bench_c5.json
(timeout 900 python bench.py --workload c3t --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null) > $SUM/${R}_bench_c3t.json
(timeout 900 python bench.py --workload c3-chain --steps 10 --warmup 3 2>/dev/null) > $SUM/${R}_bench_c3chain.json
(SB_DIST_TIMEOUT_S=120 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 2>/dev/null | tail -1) > $SUM/${R}_bench_c3_2ranks_one_gpu.json
bash tools/profile_exonbin.sh $R > /dev/null 2>&1
cp $OUT/exonbin/summary.json $SUM/${R}_exonbin_summary.json
cp $OUT/exonbin/stats/eb_kernel_stats.csv $SUM/${R}_exonbin_kernel_stats.csv
bash tools/profile_binweight.sh $R > /dev/null 2>&1
cp $OUT/binweight/summary.json $SUM/${R}_binweight_summary.json
cp $OUT/binweight/bench_binweight.json $SUM/${R}_bench_binweight.json
bash tools/profile_binseq.sh $R > /dev/null 2>&1
cp $OUT/binseq/summary.json $SUM/${R}_binseq_summary.json
cp $OUT/binseq/stats/bs_kernel_stats.csv $SUM/${R}_binseq_kernel_stats.csv
# wide loci (> 64 isoforms): multi-workgroup kernel, by shape and on C3-T
(timeout 600 python tools/probe_c3t.py; timeout 300 python tools/probe_wide_loci.py; timeout 300 python tools/probe_wide_shapes.py; timeout 600 python tools/check_wide_shapes.py) 2>/dev/null > $SUM/${R}_wide_loci.txt
timeout 600 python tools/c5_sweep.py $SUM/${R}_c5_sweep.json > /dev/null 2>&1
timeout 120 ./tools/bin/microbench > $SUM/${R}_microbench.txt 2>&1
# round 4: the front end (records -> pairs -> unique hits, flat kernels) with its kernel stats; the PCIe-inclusive host entry;
# the drop-in under the reference's own driver, timed (oracle/_ref binaries)
(timeout 300 python tools/bench_frontend.py; timeout 300 python tools/bench_frontend.py 20000 1e7 0.5) 2>/dev/null > $SUM/${R}_frontend.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_frontend -o fe -- python3 $REPO/tools/bench_frontend.py > $OUT/stats_frontend.log 2>&1)
for f in $(find $OUT/stats_frontend -name "*kernel_stats.csv"); do cp $f $SUM/${R}_frontend_kernel_stats.csv; done
(timeout 600 python tools/bench_host_entry.py 60000 2e8 3 2>/dev/null | tail -1) > $SUM/${R}_host_entry.json
(timeout 1200 python tools/dropin_timing.py 2>/dev/null) > $SUM/${R}_dropin.txt
# loci of 65-128 segments: the 128-bit segment basis against the exon walk (kernel rows of two profiled runs)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ebbig_1 -o eb -- python3 $REPO/tools/bench_exonbin_big.py > $OUT/stats_ebbig_1.log 2>&1)
(echo "# 1500 loci of ~98 segments, 4.5 M hits (tools/bench_exonbin_big.py under rocprofv3 --kernel-trace --stats): kernel rows, 128-bit segment basis"; python3 tools/kernel_stats.py $OUT/stats_ebbig_1 6 | grep sb::) > $SUM/${R}_exonbin_big_loci.txt
# BAM records -> the read stream: the bench line and its kernel rows
(timeout 300 python tools/bench_bamdecode.py 4e6 2>/dev/null | tail -1) > $SUM/${R}_bamdecode.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bamdecode -o bd -- python3 $REPO/tools/bench_bamdecode.py 4e6 --no-cpu-baseline > $OUT/stats_bamdecode.log 2>&1)
for f in $(find $OUT/stats_bamdecode -name "*kernel_stats.csv"); do cp $f $SUM/${R}_bamdecode_kernel_stats.csv; done
# round 5: strong-scaling shards a rank at a time (EM kinds, plan settings of a small shard, the chain's shards); records -> theta
# (c3-front) with its kernel rows; instructions per algorithmic FMA of the tile kernels by layout; a full wide-locus tile's counters
((timeout 600 python tools/probe_strong_kinds.py; echo; echo "# the CHAIN's shards (tools/probe_strong_chain.py)"; timeout 900 python tools/probe_strong_chain.py) 2>/dev/null | grep -v amdgpu.ids) > $SUM/${R}_strong_shards.txt
((echo "# records -> theta (bench.py --workload c3-front --gpus N), a rank at a time (tools/probe_strong_front.py)"; timeout 1200 python tools/probe_strong_front.py 2>/dev/null | grep world) > $SUM/${R}_strong_shards_front.txt)
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_front_$R -o fr -- python3 $REPO/bench.py --workload c3-front --steps 2 --warmup 1 --no-cpu-baseline > /tmp/stats_front_$R.log 2>&1)
for f in $(find /tmp/stats_front_$R -name "*kernel_stats.csv"); do cp $f $SUM/${R}_c3front_kernel_stats.csv; done
# HBM traffic counters of the records -> theta pass (every kernel of one step; separate FETCH / WRITE runs)
for pmc in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 1500 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_${pmc}_front_$R -o fr -- python3 $REPO/bench.py --workload c3-front --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_${pmc}_front_$R.log 2>&1)
done
python3 - "$SUM" "$R" <<'PY'
import collections, csv, ctypes, glob, json, sys
summ, r = sys.argv[1:3]
L = ctypes.CDLL("strawberry_amd/lib/libsbgpu.so"); L.sbgpu_build_id.restype = ctypes.c_char_p
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_*_front_%s/**/*counter_collection.csv" % r, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "sb::" in k or "rocprim" in k:
            acc[k.split("(")[0][:160]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
summary["_build_id"] = L.sbgpu_build_id().decode()
json.dump(summary, open("%s/%s_c3front_pmc_summary.json" % (summ, r), "w"), indent=1)
PY
cp $SUM/${R}_c3front_pmc_summary.json $REPO/profiles/ 2>/dev/null
(timeout 1500 python bench.py --workload c3-front --steps 5 --warmup 2 2>/dev/null | tail -1) > $SUM/${R}_bench_c3front.json
# round 6: the records start in HOST memory (sbgpu_front_stream_*): C3's sample, and config 5's size (4e8 read pairs, two parts)
(timeout 900 python bench.py --workload c3-front --from-host --steps 3 2>/dev/null | tail -1) > $SUM/${R}_bench_c3front_fromhost.json
(SB_FRONT_FRAGS=4e8 timeout 1200 python bench.py --workload c3-front --from-host --steps 2 2>/dev/null | tail -1) > $SUM/${R}_bench_c5size_fromhost.json
bash tools/pmc_em_layouts.sh > $SUM/${R}_em_layout_instr.txt 2>&1
(bash tools/pmc_wide_one.sh 2>&1 | grep "^gpurun_out/pmcw") > $SUM/${R}_wide_tile_pmc.txt
# random stress on this build (tails; the library's build id on top)
(python -c "import sys; sys.path.insert(0, '.'); from strawberry_amd import _lib; print('libsbgpu build', _lib.load().sbgpu_build_id().decode())"; timeout 900 python tools/stress_em.py 48 2>&1 | tail -3; timeout 600 python tools/stress_exonbin.py 48 2>&1 | tail -2; timeout 600 python tools/stress_binseq.py 2>&1 | tail -2; timeout 600 python tools/stress_bamdecode.py 48 2>/dev/null | tail -2) > $SUM/${R}_stress.txt 2>&1
# what travels back is the summary (gpurun merges at most 64 MiB of gpurun_out/): the raw traces and counter files stay on the box
find $OUT -mindepth 1 -maxdepth 1 ! -name summary -exec rm -rf {} + 2>/dev/null
find $REPO/gpurun_out -mindepth 1 -maxdepth 1 ! -name $R ! -name "profile_round_*.log" -exec rm -rf {} + 2>/dev/null
du -sh $REPO/gpurun_out
cat $SUM/${R}_pytest_gpu.txt; cat $SUM/${R}_bench_c3.json; echo; cat $SUM/${R}_bench_c2.json; echo; ls -la $SUM
