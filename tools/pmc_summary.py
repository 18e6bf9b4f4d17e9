"""tools/pmc_summary.py <dir written by tools/pmc_profile.sh or tools/pmc_target.sh> [--kernels a,b] [--label TEXT] -> JSON on stdout:
per-launch counters of the selected kernels and the figures derived from them, as MI355X_MICROARCH.md prescribes (FETCH_SIZE /
WRITE_SIZE in KiB units; gfx950 reports wide coalesced reads at half their bytes: FETCH_SIZE x 2; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time).  Without --kernels: the headline kernel (render_kernel), flat layout of
rounds 1-2; with --kernels: {"kernels": [one record per distinct kernel name that matches]}."""
import argparse
import csv
import glob
import json
import os


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def summarise(root, match):
    counters, meta, durs = {}, {}, []
    for r in rows(os.path.join(root, "pmc*", "**", "*counter_collection.csv")):
        if not match(r["Kernel_Name"]):
            continue
        counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = {"kernel": r["Kernel_Name"], "vgpr_count": r.get("VGPR_Count"), "accum_vgpr_count": r.get("Accum_VGPR_Count"),
                "lds_block_size": r.get("LDS_Block_Size"), "scratch_size": r.get("Scratch_Size")}
    for r in rows(os.path.join(root, "pmc3", "**", "*kernel_trace.csv")):
        if match(r["Kernel_Name"]):
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    trace = []
    for r in rows(os.path.join(root, "trace", "**", "*kernel_stats.csv")):
        if match(r["Name"]):
            trace.append({"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6})
    c = {k: sum(v) / len(v) for k, v in counters.items()}
    durs.sort()
    ms = durs[len(durs) // 2] if durs else None
    out = dict(meta)
    out["kernel_trace_stats"] = trace
    out["kernel_ms_median_under_pmc"] = ms
    out["counters_per_launch"] = c
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        d.update(hbm_read_bytes_corrected_x2=int(rd), hbm_write_bytes=int(wr), hbm_bytes_per_launch=int(rd + wr))
        if ms:
            d["hbm_GBps"] = round((rd + wr) / (ms * 1e-3) / 1e9, 3)
    if ms and "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3)
        d["clock_GHz_from_GRBM_GUI_ACTIVE"] = round(clk / 1e9, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_pipe_busy_fraction_of_elapsed_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * clk * ms * 1e-3), 4)
    if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c and c["SQ_INSTS_MFMA"] > 0:
        d["mfma_insts"] = c["SQ_INSTS_MFMA"]
        d["valu_per_mfma_excl_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 3)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        for k, name in (("SQ_WAIT_ANY", "wait_any_fraction_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_fraction")):
            if k in c:
                d[name] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
    if "SQ_INSTS_VMEM" in c:
        d["vmem_insts"] = c["SQ_INSTS_VMEM"]
    out["derived"] = d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root", nargs="?", default="gpurun_out/pmc")
    ap.add_argument("--kernels", default=None, help="comma-separated substrings; one record per distinct matching kernel name")
    ap.add_argument("--label", default=None)
    a = ap.parse_args()
    if a.kernels is None:
        out = summarise(a.root, lambda n: "render_kernel" in n)
        out["workload"] = a.label or "bench.py default: 800x800 rays x 64 samples, V1, ERT off, one launch"
        if "hbm_bytes_per_launch" in out["derived"]:
            out["derived"]["algorithmic_bytes_per_launch"] = 640000 * 16
        print(json.dumps(out, indent=1))
        return
    subs = [k for k in a.kernels.split(",") if k]
    names = set()
    for r in rows(os.path.join(a.root, "pmc3", "**", "*kernel_trace.csv")):
        if any(k in r["Kernel_Name"] for k in subs):
            names.add(r["Kernel_Name"])
    recs = [summarise(a.root, lambda n, full=full: n == full) for full in sorted(names)]
    print(json.dumps({"workload": a.label, "kernels": recs}, indent=1))


if __name__ == "__main__":
    main()
