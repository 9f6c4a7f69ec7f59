"""CPU: the C-ABI library loads without a GPU and exports exactly what include/chq.h declares; contexts fail
loudly (no CPU fallback) when no device is usable."""
import os
import re

import pytest

from chapterhouseqe_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "chq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.chq_abi_version() == 1
    assert L.chq_status_name(21) == b"ArrowError::DivideByZero"
    assert L.chq_status_name(9) == b"ComputeValueError::UnsupportedTypeCoersionForOperationBetweenTypes"


def test_no_oracle_in_the_product():
    """The product path must never import, link or call the oracle."""
    for base, _, files in os.walk(os.path.join(ROOT, "chapterhouseqe_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(base, f), errors="replace").read().lower(), os.path.join(base, f)
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_context_creation_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chapterhouseqe_amd import ChqError, Context
    with pytest.raises(ChqError) as ei:
        Context(0)
    assert ei.value.code == 40   # CHQ_ERR_DEVICE


def test_expression_handles_build_without_a_gpu():
    from chapterhouseqe_amd.record_utils import _expr_to_c
    from chapterhouseqe_amd.sqlparse import parse_expr
    h = _expr_to_c(parse_expr("(a + 1.5) * b > c and s = 'x' or -d < 2"))
    assert h
    _lib.lib().chq_expr_free(h)
