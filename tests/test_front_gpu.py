"""Records -> theta on the device (strawberry_amd/front.py, bench.py --workload c3-front): the BAM records of a chain
sample's read pairs, resident in HBM, through sbgpu_bam_decode_device -> sbgpu_assign_reads_device ->
sbgpu_pair_mates_device -> sbgpu_collapse_pairs_device -> sbgpu_quantify_device must hand the chain the sample's
unique hits (less the pairs the reference's span filter drops): theta, status and iteration counts equal sbgpu_quantify_device's on those hits BIT FOR BIT (and through it the
oracle chain's and the reference program's, tests/test_chain_scale_gpu.py, bench.py's parity legs).  tests/test_front.py
checks the packed records themselves on the CPU (host decoder, oracle decoder, host pairing and collapse)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_records_to_theta_equals_the_chain_on_the_same_sample():
    from strawberry_amd import em, front
    ctx = em.default_context(0)
    q = front.FrontQuantifier(ctx, n_loci=2500, n_frags=2.5e6, seed=44)
    assert q.n_records == 2 * q.n_frags and q.n_frags > q.n_hits
    for _ in range(2):                       # (the second step reuses pooled scratch)
        q.theta[:] = -1
        q.step()
        assert q.counts["accepted_records"] == q.n_records
        assert set(q.stage_wall_ms) == set(front.FrontQuantifier.STAGES)
        # locus by locus against the chain on the sample's own unique hits: bit for bit, except where the reference's span
        # filter (alignments.cpp:666-682; the sample generator does not model it) dropped a pair -- a handful of loci
        c = q.compare_with_chain()
        assert c["ok"] and c["bitwise_equal_there"], c
        assert c["loci_with_the_samples_hits"] >= q.n_loci - 20 and c["unique_hits_lost_there"] <= c["pairs_dropped_by_the_span_filter"] <= 50, c
        assert q.counts["unique_hits"] == q.n_hits - c["unique_hits_lost_there"]
    assert (q.status[:q.n_loci] != 1).sum() > 2000 and q.iters[:q.n_loci].max() > 100
    q.close()


def test_bench_c3_front_shards_the_sample_over_two_ranks():
    """`bench.py --workload c3-front --gpus 2` (here: two ranks on the one GPU): ONE sample, locus l -- its cluster and its
    records -- on rank l mod 2, every stage on the rank's own records, nothing but the step's barrier between the ranks.
    Every rank's theta must be the chain's on its shard; the line carries the ranks' own times and stage times."""
    env = dict(os.environ, SB_FRONT_LOCI="3000", SB_FRONT_FRAGS="3e6", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3-front", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity_with_chain"]["ok"] and d["parity_with_chain"]["ranks_ok"] == 2
    assert d["loci_per_rank"] == [1500, 1500] and sum(d["records_per_rank"]) == d["config"]["records"] == 2 * d["config"]["read_pairs"]
    assert len(d["per_rank_ms"]) == 2 and set(d["per_rank_stage_ms"][1]) == {"bam_decode", "assign_reads", "pair_mates", "collapse_pairs", "quantify"}
    assert d["value"] > 0 and d["ms_per_step"] >= max(d["per_rank_ms"]) * 0.5


def test_default_bench_line_carries_the_front_leg_and_survives_its_failure():
    """`python bench.py` (the driver's command) times records -> theta as a leg of the default line ("front").  The leg must never
    cost the line: where it cannot run (here: a sample that cannot fit the device) the line comes out without its numbers."""
    def run(**kw):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", **kw)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-chain", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    d = run(SB_FRONT_LOCI="3000", SB_FRONT_FRAGS="3e6")
    f = d["front"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and f["ranks_that_ran"] == [0] and f["parity_with_chain"]["ok"]
    assert f["loci"] == 3000 and f["records"] > 5000000
    assert set(f["per_rank_stage_ms"][0]) == {"bam_decode", "assign_reads", "pair_mates", "collapse_pairs", "quantify"} and f["ms_per_step"] > 0
    d = run(SB_FRONT_LOCI="3000", SB_FRONT_FRAGS="1e12")
    assert d["value"] > 0 and d["front"]["ranks_that_ran"] == [] and "skipped" in d["front"]["note_rank0"] and "ms_per_step" not in d["front"]


def test_records_from_host_in_chunks_equal_the_resident_pass():
    """sbgpu_front_stream_begin / push / end (include/sbgpu.h): the record stream in HOST memory, pushed in chunks of whole
    records -- 40 and 7 chunks, so that clusters straddle chunk after chunk and records are decoded twice -- gives the unique
    hits, the law, theta, FPKM, Frac, keep and TPM of the resident pass over the whole sample, bit for bit; with the empirical law
    and with a given one; a chunk smaller than a cluster's records is refused (SBGPU_ESHAPE), not mis-served."""
    from strawberry_amd import _lib, em, front
    ctx = em.default_context(0)
    for empirical, pinned in ((True, True), (False, False)):    # (page-locked memory: the upload overlaps; pageable: staged by the runtime)
        q = front.FrontQuantifier(ctx, n_loci=2500, n_frags=2.5e6, seed=44, resident=True, empirical=empirical)
        q.step()
        size = {"theta": q.n_iso, "fpkm": q.n_iso, "frac": q.n_iso, "tpm": q.n_iso, "keep": q.n_iso, "status": q.n_loci, "iters": q.n_loci}
        want = {k: getattr(q, k)[:n].copy() for k, n in size.items()}
        want_law, want_off, want_tot = dict(q.law), q.front_hit_off.copy(), (q.total_fpkm, q.total_mapped_reads)
        n_bytes = q.n_bytes
        assert q.to_host(n_bytes // 40 + 4096, pinned=pinned)["pinned"] == pinned
        for chunk in (n_bytes // 40 + 4096, n_bytes // 7 + 4096):
            q.cut(chunk)
            for k in want:
                getattr(q, k)[:] = -1
            info = q.stream_step()
            assert info["records"] == q.n_records and info["chunks"] == len(q.h_chunks) and info["clusters_finished"] == q.n_loci
            np.testing.assert_array_equal(q.front_hit_off, want_off)
            for k, v in want.items():
                np.testing.assert_array_equal(getattr(q, k)[:size[k]], v, err_msg="%s (chunk %d)" % (k, chunk))
            assert (q.total_fpkm, q.total_mapped_reads) == want_tot
            for k in ("mean", "sd", "use_emp", "start_offset", "end_offset", "total_reads"):
                assert q.law[k] == want_law[k], k
            if empirical:
                np.testing.assert_array_equal(q.law["emp_hist"], want_law["emp_hist"])
            assert info["records_decoded_twice"] > 0 and 0 < info["least_free_device_bytes"] <= info["free_device_bytes_at_begin"]
        q.cut(1 << 16)          # a cluster's records must fit a chunk
        with pytest.raises(_lib.SbgpuError, match="exceed a chunk"):
            q.stream_step()
        q.close()


def test_span_filtered_loci_against_the_oracle():
    """Where the reference's span filter dropped pairs the front's unique hits differ from the sample's, and compare_with_chain can
    only count them: those clusters' pairs go through the oracle's collapse instead, and the chain on the oracle's hits must be the
    front's theta / FPKM on those loci, bit for bit (FrontQuantifier.check_filtered_loci; at full size inside bench.py --workload c3-front)."""
    from oracle import OracleLib
    from strawberry_amd import em, front
    ctx = em.default_context(0)
    q = front.FrontQuantifier(ctx, n_loci=2500, n_frags=2.5e6, seed=44, resident=True, empirical=True)
    q.step()
    r = q.check_filtered_loci(OracleLib())
    assert r["ok"] and r["loci"] >= 1 and r["loci_checked"] == r["loci"], r
    assert r["pairs_the_oracle_dropped"] == q.counts["pairs_dropped_by_the_span_filter"] > 0, (r, q.counts)
    q.close()
