"""bench.py's own launcher (CPU, gloo): `python bench.py --gpus N` without a torchrun environment starts N ranks
itself -- the driver's command line -- before anything touches a GPU; the ranks rendezvous on 127.0.0.1, go through
the run's barrier / max-over-ranks / sum-over-ranks calls, and rank 0's JSON line comes out of the parent's stdout.
SB_BENCH_DRY_RUN=1 leaves the GPU work out (there is no device here); everything else is the real code path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SB_BENCH_DRY_RUN="1", **kw)
    return env


def test_gpus_2_launches_two_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rank_sum"] == 1
    assert out["max_over_ranks"] == 2.0 and out["steps"] == 3 and out["warmup"] == 1
    # the headline at N > 1 is BASELINE config 3: ONE problem, loci sharded over the ranks
    assert out["scaling"] == "strong"
    # ... and the line explains itself: every rank's own time, who was slowest, the 1000-iteration loci per rank, and
    # the weak value beside the strong one (gathered through the run's own all-reduce)
    assert out["per_rank_ms"] == [1.0, 2.0] and out["slowest_rank"] == 1 and out["capped_loci_per_rank"] == [0, 10]
    assert "value_strong" in out and "value_weak" in out


def test_world_size_that_differs_from_gpus_fails_loudly():
    # a launcher that started one rank for --gpus 2: no silent single-rank run
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "1 rank" in r.stderr


def test_child_failure_is_the_parents_return_code():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "nonsense"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_default_step_counts_by_workload():
    """No --steps / --warmup: the EM workloads (a step is under a millisecond, a run of batches reaches its steady rate after ~20
    of them) time 200 steps after 20, the others 20 after 3; explicit flags -- the driver's `--steps 20 --warmup 5` -- are taken as
    they are, and the children of `--gpus 2` get the parent's resolved numbers."""
    def line(*args):
        r = subprocess.run([sys.executable, BENCH, *args], env=_env(), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out = line()
    assert (out["steps"], out["warmup"], out["workload"]) == (200, 20, "c3")
    out = line("--workload", "c3-chain")
    assert (out["steps"], out["warmup"]) == (20, 3)
    out = line("--gpus", "1", "--steps", "20", "--warmup", "5")
    assert (out["steps"], out["warmup"]) == (20, 5)
    out = line("--gpus", "2", "--workload", "c2")
    assert (out["steps"], out["warmup"], out["n_gpus"]) == (200, 20, 2)
