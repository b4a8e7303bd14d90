#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/collect_profiles.sh) into profiles/<tag>_*.

    python tools/summarize_profiles.py r01

Writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --stats table, stgcn kernels first),
profiles/<tag>_counters.json (per-kernel mean of every PMC counter, derived clock / MFMA-busy fraction and
HBM traffic per launch with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md §HBM) and
profiles/<tag>_bench.json (the bench line printed under the profiler).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda paths: sorted(paths, key=os.path.getmtime)[-1:]   # gpurun_out/ accumulates earlier runs' files
    stats = newest(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        head, body = rows[0], rows[1:]
        body.sort(key=lambda r: (0 if "stgcn" in r[0] else 1, -float(r[2])))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(head)
            w.writerows(body[:int(os.environ.get("STGCN_PROFILE_ROWS", "12"))])
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for f in newest(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "stgcn" not in name:
                    continue
                short = name.replace("void ", "").replace("stgcn::(anonymous namespace)::", "").replace("stgcn::", "")
                short = short.split("(")[0]            # kernel<template args>, namespaces stripped
                counters[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                counters[short]["_dur_ns_" + os.path.basename(d)].append(
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                counters[short]["_vgpr"] = [float(r["VGPR_Count"])]
                counters[short]["_lds_bytes"] = [float(r["LDS_Block_Size"])]
    out = {}
    for k, cs in counters.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read
            m["hbm_read_bytes_per_launch_x2_corrected"] = m.get("FETCH_SIZE", 0.0) * 1024 * 2
            m["hbm_write_bytes_per_launch"] = m.get("WRITE_SIZE", 0.0) * 1024
            m["hbm_traffic_bytes_per_launch"] = m["hbm_read_bytes_per_launch_x2_corrected"] + m["hbm_write_bytes_per_launch"]
        per_pass = [len(v) for c, v in cs.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES")]
        m["launches_seen"] = max(per_pass) if per_pass else 0
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            m["clock_GHz"] = cyc / m["_dur_ns_pmc_sq1"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        out[k] = m
    prog = os.path.join(src, "program.txt")
    if os.path.exists(prog):                         # tools/collect_counters.sh: any program, not only bench.py
        out["_meta"] = {"program": open(prog).read().strip(), "collected_by": "tools/collect_counters.sh " + tag}
    log = os.path.join(src, "trace.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                bench = json.loads(line)
                if "config" not in bench:            # another program's result line (tools/train_step.py ...): keep it as is
                    out.setdefault("_meta", {})["result_line"] = bench
                    continue
                json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
                # what bench.py checks before quoting `roofline.traffic` from this file: same workload, same kernel sources
                out["_meta"] = dict(out.get("_meta", {}), workload=bench["config"].get("workload_key"),
                                    csrc_digest=bench["config"].get("csrc_digest"))
                out["_meta"].setdefault("collected_by", "tools/collect_profiles.sh " + tag)
    json.dump(out, open(os.path.join(dst, f"{tag}_counters.json"), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
