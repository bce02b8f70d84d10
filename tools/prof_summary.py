"""Summarise rocprofv3 output directories into the small files kept under profiles/.
    python tools/prof_summary.py stats  <dir> <out.csv>        per-kernel calls / total / avg / min / max (kernel trace)
    python tools/prof_summary.py pmc    <dir> <out.csv>        per-kernel per-counter mean per call
    python tools/prof_summary.py util   <pmc.csv> <out.csv>    MFMA-busy per kernel from a pmc summary that holds
                                                               SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
    python tools/prof_summary.py asgjson <fetch.csv> <write.csv> <out.json>   HBM bytes per launch of the assignment
                                                               kernels (FETCH_SIZE x 2 + WRITE_SIZE; bench.py reads it)
Measurement infrastructure."""
import collections, csv, glob, os, sys


def _rows(d, pat):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def stats(d, out):
    agg = collections.defaultdict(list)
    for r in _rows(d, "*kernel_trace.csv"):
        agg[r["Kernel_Name"].split("(")[0][:80]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    with open(out, "w") as fh:
        fh.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f"\"{k}\",{len(v)},{sum(v)/1e3:.1f},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{max(v)/1e3:.2f},{100*sum(v)/tot:.2f}\n")


def pmc(d, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in _rows(d, "*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0][:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    with open(out, "w") as fh:
        fh.write("kernel,counter,calls,mean_per_call\n")
        for k, dct in sorted(agg.items()):
            for c, v in sorted(dct.items()):
                fh.write(f"\"{k}\",{c},{calls[(k, c)]},{v / calls[(k, c)]:.3f}\n")


def util(pmc_csv, out):
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy fraction of the
    # MFMA pipes = busy / (gui_active / 8 * 1024) = busy / (gui_active * 128)
    rows = collections.defaultdict(dict)
    with open(pmc_csv) as fh:
        for r in csv.DictReader(fh):
            rows[r["kernel"]][r["counter"]] = float(r["mean_per_call"])
    with open(out, "w") as fh:
        fh.write("kernel,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE_sum_over_8_XCDs,MfmaUtil_percent\n")
        for k, d in sorted(rows.items()):
            b, g = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
            if b > 0 and g > 0:
                fh.write(f"\"{k}\",{b:.0f},{g:.0f},{100 * b / (g * 128):.1f}\n")


def asgjson(fetch_csv, write_csv, out):
    import json
    def load(f, counter):
        d, calls = {}, {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["counter"] == counter:
                    d[r["kernel"]] = float(r["mean_per_call"]); calls[r["kernel"]] = int(r["calls"])
        return d, calls
    (fe, fcalls), (wr, _) = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in ("asg_auction", "asg_step", "asg_build", "asg_solve", "asg_small"):
        if k in fe:
            res[f"{k}_FETCH_SIZE_KiB_per_launch"] = round(fe[k], 3)
            res[f"{k}_WRITE_SIZE_KiB_per_launch"] = round(wr.get(k, 0.0), 3)
            res[f"{k}_launches"] = fcalls[k]
    if "asg_step" in fe:
        res["asg_step_hbm_bytes_per_launch"] = round((2.0 * fe["asg_step"] + wr.get("asg_step", 0.0)) * 1024.0, 3)
    # per SOLVE over the chip-wide kernels (asg_auction: the epsilon > 0 phases in one launch; asg_step: every other step):
    # the figure bench.py's roofline sets next to algorithmic_bytes_per_solve
    solves = fcalls.get("asg_init", 0)
    if solves:
        tot = sum((2.0 * fe[k] + wr.get(k, 0.0)) * 1024.0 * fcalls[k] for k in ("asg_auction", "asg_step") if k in fe)
        res["solves"] = solves
        res["asg_chip_wide_hbm_bytes_per_solve"] = round(tot / solves, 3)
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/asg_trace.py (8 C3 solves); FETCH_SIZE x2 "
                   "(gfx950 reports half the bytes of wide coalesced reads: calibrated on cost_tiled, which writes its 64 MiB output = "
                   "65536 KiB of WRITE_SIZE), WRITE_SIZE x1; means include no-op launches; per solve = sum over the launches of "
                   "asg_auction and asg_step / number of asg_init launches")
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


def skjson(prof_dir, out):
    """Sinkhorn HBM traffic per iteration from the per-config counter passes (sk_<CFG>_pmc_{FETCH,WRITE}_SIZE.csv) and the
    per-iteration kernel time from sk_<CFG>_kernel_stats.csv.  Matrix streaming: sk_col_pass + sk_col_finalize + sk_row_stream;
    variant B (C2 only): 2 x sk_pts_pass.  FETCH_SIZE x2 (gfx950), WRITE_SIZE x1."""
    import json, os
    def load(f, col="mean_per_call"):
        with open(f) as fh:
            return {r["kernel"]: float(r[col]) for r in csv.DictReader(fh)}
    stream = ("sk_col_finalize", "sk_col_pass", "void sk_row_stream<4>")
    res = {}
    for cfg in ("C5", "C2"):
        fe = load(os.path.join(prof_dir, f"sk_{cfg}_pmc_FETCH_SIZE.csv"))
        wr = load(os.path.join(prof_dir, f"sk_{cfg}_pmc_WRITE_SIZE.csv"))
        us = load(os.path.join(prof_dir, f"sk_{cfg}_kernel_stats.csv"), "avg_us")
        res[f"{cfg}_streaming_hbm_bytes_per_iteration"] = sum(2.0 * fe[k] + wr[k] for k in stream) * 1024.0
        res[f"{cfg}_streaming_FETCH_SIZE_KiB"] = {k: fe[k] for k in stream}
        res[f"{cfg}_streaming_WRITE_SIZE_KiB"] = {k: wr[k] for k in stream}
        res[f"{cfg}_streaming_us_per_iteration_kernel_trace"] = sum(us[k] for k in stream)
        pts = [k for k in fe if k.startswith("void sk_pts_pass")]
        if pts:
            res[f"{cfg}_points_hbm_bytes_per_iteration"] = 2.0 * (2.0 * fe[pts[0]] + wr[pts[0]]) * 1024.0
            res[f"{cfg}_points_us_per_iteration_kernel_trace"] = 2.0 * us[pts[0]]
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/sk_probe.py (one config per process); per "
                   "iteration = sk_col_pass + sk_col_finalize + sk_row_stream (matrix streaming) or 2 x sk_pts_pass (variant B); "
                   "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE x1")
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    fn = {"stats": stats, "pmc": pmc, "util": util, "asgjson": asgjson, "skjson": skjson}[sys.argv[1]]
    fn(*sys.argv[2:])
