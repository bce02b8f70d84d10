"""Per-kernel register / scratch / MFMA bookkeeping from the compiler's own assembly (no GPU needed):
    python tools/isa_report.py [file.hip ...]        (default: every csrc/*.hip)
For each kernel: VGPRs (`vgpr` = the TOTAL of the unified file: architectural + accumulator registers; `agpr` = the
accumulator part of it; waves per SIMD = 512 // total rounded up to 8), scratch bytes per lane, spills, LDS bytes, number of MFMA instructions and of
v_accvgpr_read / v_accvgpr_write moves (accumulators shuttled between the two register files inside a loop are a
compiler artefact that costs issue slots: round 4 found 128 of them per K step in every dense product).
Measurement infrastructure."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "conditional-flow-matching_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result".split()


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def report(path, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", out, path],
                           capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-2000:]); return []
        txt = open(out).read()
    rows = []
    for name in re.findall(r"^(_Z\w+|\w+):\s*(?:;.*)?$", txt, re.M):
        k = txt.find(".name:           " + name + "\n")
        if k < 0:
            continue
        i = txt.index("\n" + name + ":")
        j = txt.find("s_endpgm", i)
        body = txt[i:j]
        # the kernel's metadata entry: keys are sorted, so .agpr_count / .group_segment_fixed_size come BEFORE .name
        m0 = txt.rfind("\n  - .", 0, k)
        m1 = txt.find("\n  - .", k)
        meta = txt[m0:m1 if m1 > 0 else k + 1600]
        g = lambda key: (re.search(r"\.%s:\s+(\d+)" % key, meta) or [None, "0"])[1]
        rows.append(dict(kernel=demangle(name)[:84], vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")),
                         scratch=int(g("private_segment_fixed_size")), spill=int(g("vgpr_spill_count")),
                         lds=int(g("group_segment_fixed_size")), mfma=body.count("v_mfma"),
                         acc_w=body.count("v_accvgpr_write"), acc_r=body.count("v_accvgpr_read"),
                         # m0 accesses that do NOT sit directly in front of a global_load_lds (gemm_glds.h writes m0 in
                         # inline assembly without a clobber: any other consumer of m0 in such a kernel would be at risk)
                         m0_w=len(re.findall(r"s_mov_b32 m0,", body)), m0_dma=len(re.findall(r"global_load_lds_dword", body)),
                         m0_other=len(re.findall(r"\bm0\b", body)) - len(re.findall(r"s_mov_b32 m0,", body))))
    return rows


def main(argv):
    files = [a for a in argv if not a.startswith("-")] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    extra = [a for a in argv if a.startswith("-")]
    for f in files:
        print(f"== {os.path.basename(f)}")
        for r in report(f, extra):
            print(f"  vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} scratch {r['scratch']:4d} spill {r['spill']:3d} lds {r['lds']:6d} "
                  f"mfma {r['mfma']:4d} acc_w {r['acc_w']:4d} acc_r {r['acc_r']:4d} m0 w/dma/other {r['m0_w']}/{r['m0_dma']}/{r['m0_other']}  {r['kernel']}")


if __name__ == "__main__":
    main(sys.argv[1:])
