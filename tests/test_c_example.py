"""The C ABI from plain C (examples/c_abi_demo.c, gcc, no Python / torch / C++ on the caller's side): every entry point
of the path -- chq_filter_record, chq_filter_records, chq_filter_records_coalesced, chq_filter_project_record -- over
hand-built Arrow C Data structs, checked inside the program against a scalar loop AND, by the digests it prints, against
tests/golden/c_abi_demo_expected.json: the oracle's results on the same generated data (scripts/make_c_abi_demo_fixture.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_demo")


def build():
    # never rebuild libchq.so from inside the test process (it is already mapped by the other tests): only when absent
    if not os.path.exists(os.path.join(ROOT, "chapterhouseqe_amd", "lib", "libchq.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "chapterhouseqe_amd", "csrc")], check=True)
    if not os.path.exists(EXE):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples")], check=True)


def test_demo_builds_and_fails_loudly_without_a_gpu():
    build()
    assert os.path.exists(EXE)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test below")
    r = subprocess.run([EXE, "1000", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "chq_ctx_create" in r.stderr   # no CPU fallback: the context cannot be created


@pytest.mark.gpu
@pytest.mark.parametrize("rows,batches", [(10_000, 64), (3, 5), (70_001, 7)])
def test_demo_matches_its_scalar_loop(rows, batches):
    build()
    r = subprocess.run([EXE, str(rows), str(batches)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches against the scalar loop: 0" in r.stdout
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "c_abi_demo_expected.json")))["cases"][f"{rows}x{batches}"]
    got = [line for line in r.stdout.splitlines() if line.startswith("digest ")]
    assert got == want, "\n".join(["demo:"] + got + ["oracle fixture:"] + want)


def test_the_committed_fixture_is_what_the_oracle_produces():
    """regenerate the fixture (oracle on the demo's LCG data, CPU only) and compare with the committed file"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_c_abi_demo_fixture.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == json.load(open(os.path.join(ROOT, "tests", "golden", "c_abi_demo_expected.json")))
