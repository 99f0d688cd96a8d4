#!/usr/bin/env python3
"""Summarise rocprofv3 output directories: per-kernel time statistics (kernel trace) and per-kernel
mean counter values (PMC passes).  Usage: prof_summary.py DIR [DIR...] > summary.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for tag in ("pp::", "void "):
        name = name.replace(tag, "")
    return name[:60]


def counter_means(dirs):
    """{kernel: {counter: mean per dispatch}} over rocprofv3 counter_collection CSVs."""
    agg = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def traffic_json(out_path, bench_dirs, calib_dirs):
    """HBM traffic per launch of every pp kernel of the bench workload.  rocprofv3 reports FETCH_SIZE and
    WRITE_SIZE in KiB.  Corrections: the guide's gfx950 rule (FETCH_SIZE counts 128-B requests as 64 B:
    double it), cross-checked here -- and WRITE_SIZE calibrated -- on a 2 GiB streaming copy."""
    import json
    calib = counter_means(calib_dirs)
    n = float(2 << 30)
    ck = max(calib, key=lambda k: calib[k].get("FETCH_SIZE", 0) + calib[k].get("WRITE_SIZE", 0)) if calib else None
    f_raw = calib[ck].get("FETCH_SIZE", 0) * 1024 if ck else 0
    w_raw = calib[ck].get("WRITE_SIZE", 0) * 1024 if ck else 0
    fetch_factor = 2.0                                # MI355X_MICROARCH.md, HBM section
    write_factor = round(n / w_raw, 3) if w_raw else 1.0
    bench = counter_means(bench_dirs)
    kernels = {}
    for k, cs in bench.items():
        if not k.startswith("k_"):
            continue
        fr, wr = cs.get("FETCH_SIZE", 0) * 1024, cs.get("WRITE_SIZE", 0) * 1024
        kernels[k] = {"fetch_raw_bytes": round(fr), "write_raw_bytes": round(wr),
                      "hbm_bytes": round(fr * fetch_factor + wr * write_factor)}
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                     "`python bench.py --steps 10 --warmup 2 --no-cpu-baseline`; mean per dispatch",
           "fetch_factor": fetch_factor, "write_factor": write_factor,
           "calibration": {"kernel": ck, "copied_bytes": n, "fetch_raw_bytes": round(f_raw), "write_raw_bytes": round(w_raw),
                           "fetch_measured_factor": round(n / f_raw, 3) if f_raw else None},
           "kernels": kernels,
           # (a step's kernels: k_check_wo belongs to the bench's foreign-mirror leg, one launch that reads 7 GB)
           "pipeline_hbm_bytes": sum(v["hbm_bytes"] for k, v in kernels.items() if not k.startswith("k_check_wo"))}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic-json":
        a = sys.argv[2:]
        out, rest = a[0], a[1:]
        b, c = rest[rest.index("--bench") + 1:rest.index("--calib")], rest[rest.index("--calib") + 1:]
        traffic_json(out, b, c)
        return
    for d in sys.argv[1:]:
        print(f"== {d}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            dur = defaultdict(list)
            with open(path) as f:
                for row in csv.DictReader(f):
                    dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            total = sum(sum(v) for v in dur.values())
            print(f"-- kernel trace {os.path.basename(path)}: total {total / 1e6:.3f} ms")
            print(f"{'kernel':<62}{'calls':>7}{'total_ms':>11}{'avg_us':>11}{'min_us':>10}{'max_us':>10}{'pct':>7}")
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                print(f"{k:<62}{len(v):>7}{sum(v) / 1e6:>11.3f}{sum(v) / len(v) / 1e3:>11.2f}{min(v) / 1e3:>10.2f}"
                      f"{max(v) / 1e3:>10.2f}{100 * sum(v) / total:>7.1f}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            agg = defaultdict(lambda: defaultdict(list))
            meta = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    meta[k] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                               row.get("Grid_Size"), row.get("Workgroup_Size"))
            print(f"-- counters {os.path.basename(path)} (mean per dispatch)")
            for k in sorted(agg):
                print(f"{k}  vgpr={meta[k][0]} sgpr={meta[k][1]} lds={meta[k][2]} grid={meta[k][3]} wg={meta[k][4]}")
                for c, v in sorted(agg[k].items()):
                    print(f"    {c:<28}{sum(v) / len(v):>20.1f}   (n={len(v)})")


if __name__ == "__main__":
    main()
