"""Soak of the exact solver under the pipelined loop's contention (round 6): three threads solve batches of four problems on their
own streams beside the C3 model step for 20 s; no list-path solve may be refused by the fp64 certificate (the process-wide
dense-fallback counter of `cfm_assign_debug_fallback` stays where it was).  The first form of the seeded list solver lost a
pending column once in ~50 000 solves (a missing barrier behind the seed sweep) — this test caught that build in two of four runs at
B = 1024 (~100 000 solves in its 20 s).  The long form: tools/probe/fallback_hunt.py."""
import ctypes
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_list_path_solve_is_refused_under_contention():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "fallback_hunt.py"), "20", "1024", "784"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if "solves in batches of four" in l]
    assert line, out.stdout[-2000:]
    n = int(line[0].split()[0])
    assert n > 20000, line[0]                                   # the soak really ran
    assert "fallback counter (0, 0)" in line[0], out.stdout[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("HIT")]
