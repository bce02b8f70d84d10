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


def test_library_exports_nothing_undeclared(lib_built):
    """-fvisibility=hidden (csrc/build.sh) + the visibility pragmas of the two headers: the dynamic symbol table's cfm_*
    functions are EXACTLY the declared ones (no *_internal helper, no stray setter leaves the library)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib_built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if " T " in ln})
    src = open(os.path.join(ROOT, "include", "cfm_gfx950_tuning.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    tuning = set(re.findall(r"\b(cfm_[a-z0-9_]+)\s*\(", src))
    assert exported == sorted(set(_declared()) | tuning), sorted(set(exported) ^ (set(_declared()) | tuning))
    assert "-fvisibility=hidden" in open(os.path.join(ROOT, "conditional-flow-matching_amd", "csrc", "build.sh")).read()


def test_python_binding_covers_header(lib_built):
    assert sorted(lib_built.SIGNATURES) == _declared()
    assert sorted(lib_built.exported_symbols()) == _declared()


def test_workspace_sizes(lib_built):
    lib = lib_built.load()
    assert lib.cfm_abi_version() == lib_built.ABI_VERSION == 3
    assert lib.cfm_workspace_bytes(9, 127, 128, 0) >= 2 * 4 * 127 * 128 and lib.cfm_workspace_bytes(9, 128, 127, 0) == lib.cfm_workspace_bytes(9, 127, 128, 0)
    for op in (1, 2, 3, 4, 5, 6, 7):
        n = lib.cfm_workspace_bytes(op, 4096, 4096, 784)
        assert n > 0 and n % 256 == 0
    assert lib.cfm_workspace_bytes(99, 4, 4, 4) == 0
    # Sinkhorn scratch is O(B) potentials + strip partials, far below the B^2 matrix
    assert lib.cfm_workspace_bytes(1, 4096, 4096, 0) < 16 * 2**20
    # cost scratch (matrix-core form): the centre (d floats), one norm per point and the centred clouds in the padded
    # layout of the direct-to-LDS engine ([(B0 + 1) + (B1 + 1)][d rounded up to 32] floats)
    need = 4 * (784 + 8192 + (4097 + 4097) * 800)
    assert need <= lib.cfm_workspace_bytes(7, 4096, 4096, 784) < need + 64 * 2**10
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


def test_no_compile_time_experiments_left_in_the_kernels():
    """Round 5 adopted or deleted every compile-time experiment of the kernels (profiles/r5_experiments.txt): what is
    left of the preprocessor in csrc/ is the instrumented build of the list solver (SP_PROFILE, tools/probe/build_prof.sh);
    the product library is built without any -D flag."""
    import glob
    csrc = os.path.join(ROOT, "conditional-flow-matching_amd", "csrc")
    conds = []
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        for ln in open(f):
            if re.match(r"#\s*(if|ifdef|ifndef)\b", ln):
                conds.append((os.path.basename(f), ln.strip()))
    assert all(c[1] == "#ifdef SP_PROFILE" for c in conds), conds
    assert len(conds) <= 8
    build = open(os.path.join(csrc, "build.sh")).read()
    assert "-D" not in build.replace("$CFM_EXTRA_FLAGS", "")
