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
    for name in ("bench_pmc_hbm.json", "config3_pmc.json"):
        path = os.path.join(ROOT, "profiles", "r2", name)
        if os.path.exists(path):
            assert "kernel_source_sha256" in json.load(open(path)), name
