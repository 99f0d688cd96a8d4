#!/usr/bin/env python3
"""Summarise rocprofv3 output directories: per-kernel time statistics (kernel trace) and per-kernel
mean counter values (PMC passes).  Usage: prof_summary.py DIR [DIR...] > summary.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for tag in ("pp::", "void "):
        name = name.replace(tag, "")
    return name[:60]


def main():
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
