"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/cfm_gfx950.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cfm_gfx950.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cfm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("cfm_sqeuclid_cost_f32", "cfm_sinkhorn_log_f32", "cfm_assign_exact_f32",
                 "cfm_plan_sample_perm", "cfm_plan_sample_dense", "cfm_sample_xt_ut_f32",
                 "cfm_mlp_forward_f32", "cfm_ode_dopri5_mlp_f32", "cfm_workspace_bytes"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_built):
    lib = ctypes.CDLL(lib_built.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/cfm_gfx950.h but not exported"


def test_python_binding_covers_header(lib_built):
    assert sorted(lib_built.SIGNATURES) == _declared()
    assert sorted(lib_built.exported_symbols()) == _declared()


def test_workspace_sizes(lib_built):
    lib = lib_built.load()
    assert lib.cfm_abi_version() == lib_built.ABI_VERSION == 2
    for op in (1, 2, 3, 4, 5, 6, 7):
        n = lib.cfm_workspace_bytes(op, 4096, 4096, 784)
        assert n > 0 and n % 256 == 0
    assert lib.cfm_workspace_bytes(99, 4, 4, 4) == 0
    # Sinkhorn scratch is O(B) potentials + strip partials, far below the B^2 matrix
    assert lib.cfm_workspace_bytes(1, 4096, 4096, 0) < 16 * 2**20
    # cost scratch (matrix-core form): the centre (d floats) and one norm per point
    assert 4 * (784 + 8192) <= lib.cfm_workspace_bytes(7, 4096, 4096, 784) < 64 * 2**10
    assert lib.cfm_workspace_bytes(7, 4096, 4096, 0) == 0
    # assignment scratch: O(B) state + 64 candidate (column, cost) pairs per row
    assert lib.cfm_workspace_bytes(2, 4096, 4096, 0) < 4 * 2**20


def test_code_object_is_gfx950_only(lib_built):
    blob = open(lib_built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert other not in blob


def test_tuning_header_lists_every_extra_export(lib_built):
    """The measurement / tuning exports are declared too (include/cfm_gfx950_tuning.h) — separately: they are not part
    of the operator ABI — and the binding's EXTRA_SIGNATURES is exactly that list."""
    src = open(os.path.join(ROOT, "include", "cfm_gfx950_tuning.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(cfm_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(lib_built.EXTRA_SIGNATURES)
    lib = ctypes.CDLL(lib_built.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert not set(names) & set(_declared())


def test_experiment_switches_are_off_in_the_product_build():
    """Every compile-time experiment of the kernels (prepared, not yet measured) defaults to 0 — the product
    library is built without any -D flag (csrc/build.sh) — except the two that only select WHICH experimental loop a
    COST_GLDS_V2 build takes."""
    csrc = os.path.join(ROOT, "conditional-flow-matching_amd", "csrc")
    want = {"gemm_core.h": {"GC_FAIR": "0", "GC_PIPE": "0", "GC_DBG": "0", "GC_FETCH_MODE": "0"},
            "cost.hip": {"COST_GLDS_V2": "0"}, "assign.hip": {"ASG_PREFETCH_CTL": "0"}, "gemm_glds.h": {"GL_DBG": "0"}}
    for fname, macros in want.items():
        src = open(os.path.join(csrc, fname)).read()
        for name, val in macros.items():
            m = re.search(r"#ifndef %s\s*\n#define %s (\S+)" % (name, name), src)
            assert m and m.group(1) == val, (fname, name, m and m.group(1))
    build = open(os.path.join(csrc, "build.sh")).read()
    assert "-D" not in build.replace("$CFM_EXTRA_FLAGS", "")
    r = __import__("subprocess").run(["bash", "-n", os.path.join(ROOT, "tools", "probe", "try_glds_v2.sh")], capture_output=True)
    assert r.returncode == 0, r.stderr
