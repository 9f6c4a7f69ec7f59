"""CPU: `bench.py --gpus N` started WITHOUT a launcher must start the N ranks itself (child processes over
torch.distributed.run on 127.0.0.1), print one rank-0 JSON line with n_gpus = N and exit non-zero when a rank fails.
The rank protocol runs on gloo here (`--launcher-selftest`: no GPU, no libchq), the spawn path is the real one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest", *extra],
                          capture_output=True, text=True, timeout=300, env=env)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_n_without_world_size_launches_n_ranks():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    j = lines[0]
    assert j["n_gpus"] == 2 and len(j["per_gpu_rows_per_s"]) == 2
    assert j["rows_total"] == 1000 + 2000                 # SUM over the ranks
    assert j["value"] <= sum(j["per_gpu_rows_per_s"]) * 1.001   # aggregate uses the MAX elapsed time


def test_strong_scaling_is_the_default_and_splits_the_table_evenly():
    """"repeat config 2 at 2/4/8 GPUs (n split evenly)" -- the N > 1 default"""
    r = _run("--gpus", "2", "--rows", "1000001")
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)[0]
    assert j["scaling"] == "strong" and j["config"] == 2
    assert j["rows_planned_total"] == 1000001 and j["rows_summed"] == 1000001      # nothing lost to the division
    assert j["rows_this_rank"] == 500001                                           # the remainder goes to the lowest ranks


def test_weak_scaling_keeps_the_per_gpu_rows():
    r = _run("--gpus", "2", "--rows", "1000", "--scaling", "weak")
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)[0]
    assert j["scaling"] == "weak" and j["rows_this_rank"] == 1000 and j["rows_summed"] == 2000 == j["rows_planned_total"]


def test_config5_is_one_fixed_shard_per_gpu_and_gathers_to_rank0():
    """BASELINE config 5: 1.25 B rows per GPU (8 GPUs = the 10 B-row table); --gathered ships every peer's survivors to
    rank 0 (here: host batches over gloo through the same operators.distributed calls)"""
    r = _run("--gpus", "3", "--config", "5", "--gathered")
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)[0]
    assert j["config"] == 5 and j["scaling"] == "weak"
    assert j["rows_this_rank"] == 1_250_000_000 and j["rows_planned_total"] == 3 * 1_250_000_000 == j["rows_summed"]
    assert j["gathered_rows"] == sum(10 * rank + b for rank in (1, 2) for b in range(2))


def test_plan_rows_covers_every_mode():
    sys.path.insert(0, ROOT)
    import bench

    class A:
        config, rows, scaling = 2, None, "strong"
    assert [bench.plan_rows(A, 8, r)[0] for r in range(8)] == [125_000_000] * 8 and bench.plan_rows(A, 8, 0)[1:] == (1_000_000_000, "strong")
    assert bench.plan_rows(A, 1, 0) == (1_000_000_000, 1_000_000_000, "strong")
    A.rows = 10
    assert [bench.plan_rows(A, 4, r)[0] for r in range(4)] == [3, 3, 2, 2]
    A.scaling = "weak"
    assert bench.plan_rows(A, 4, 3) == (10, 40, "weak")
    A.config, A.rows = 5, None
    assert bench.plan_rows(A, 8, 5) == (1_250_000_000, 10_000_000_000, "weak")


def test_a_failing_rank_fails_the_run():
    r = _run("--gpus", "2", "--selftest-fail-rank", "1")
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_single_rank_needs_no_launcher():
    r = _run("--gpus", "1")
    assert r.returncode == 0 and _json_lines(r.stdout)[0]["n_gpus"] == 1


def test_pmc_summaries_are_tied_to_the_kernel_sources():
    """roofline.traffic is quoted from a committed rocprofv3 --pmc summary only while it carries the hash of the kernel
    sources it was measured on"""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 64
    for rnd in ("r2", "r3"):
        for name in ("bench_pmc_hbm.json", "config3_pmc.json"):
            path = os.path.join(ROOT, "profiles", rnd, name)
            if os.path.exists(path):
                assert "kernel_source_sha256" in json.load(open(path)), name
