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


def _split_params(text):
    """top-level comma split of a parameter list ("void" / "" = no parameters)"""
    text = text.strip()
    if text in ("", "void"):
        return []
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def test_rust_ffi_declares_every_export_with_the_same_arity():
    """rust/chq_sys.rs (the binding a maintainer drops into the reference) cannot be compiled here: keep it
    honest by parsing it -- every function of include/chq.h must be declared, with the same number of parameters, and
    nothing else."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "chq.h")).read(), flags=re.S)
    c_funcs = {m.group(1): len(_split_params(m.group(2)))
               for m in re.finditer(r"\b(chq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}
    rust = open(os.path.join(ROOT, "rust", "chq_sys.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)
    r_funcs = {m.group(1): len(_split_params(m.group(2)))
               for m in re.finditer(r"pub fn (chq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*(?:->[^;]*)?;", rust, flags=re.S)}
    assert sorted(c_funcs) == _declared_symbols()
    assert sorted(r_funcs) == sorted(c_funcs), (sorted(set(c_funcs) - set(r_funcs)), sorted(set(r_funcs) - set(c_funcs)))
    for name, n in c_funcs.items():
        assert r_funcs[name] == n, (name, n, r_funcs[name])
    # the repr(C) structs the functions take by pointer
    for struct in ("ArrowDeviceArray", "chq_select_item", "chq_alias_list", "chq_table_aliases", "chq_call_stats", "chq_column_desc", "chq_ipc_message"):
        assert re.search(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct " + struct + r"\b", rust), struct


def test_rust_task_builders_implement_the_plugin_trait():
    """the two GPU tasks are complete files: a struct, `impl TaskBuilder for ...` with the trait's seven parameters
    (operators/traits.rs:22-36) and a MessageConsumer"""
    for fname, builder in (("gpu_filter_task.rs", "GpuFilterTaskBuilder"), ("gpu_materialize_files_task.rs", "GpuMaterializeFilesTaskBuilder")):
        text = open(os.path.join(ROOT, "rust", fname)).read()
        assert f"pub struct {builder}" in text and f"impl TaskBuilder for {builder}" in text, fname
        m = re.search(r"fn build\(\s*&self,(.*?)\)\s*->", text, flags=re.S)
        assert m and len(_split_params(m.group(1))) == 7, fname
        assert "tt.spawn(" in text and "oneshot::channel()" in text and "RecordHandler::initiate(" in text, fname
        assert "// ..." not in text and "same `use` list" not in text, fname      # no elided parts
    assert "impl MessageConsumer for GpuTaskConsumer" in open(os.path.join(ROOT, "rust", "gpu_filter_task.rs")).read()


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
