#!/usr/bin/env python3
"""Experiment: where a step's time goes BETWEEN its kernels.  Reads a rocprofv3 --kernel-trace CSV (the directory given) and
prints, for the library's kernels of the bench's timed steps, every kernel's duration and the idle time in front of it (its
start minus the end of the kernel before it, whatever that was), averaged over the jobs.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python bench.py --steps 30 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout
    python tools/exp_gaps.py /tmp/gt"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            n = n[5:] if n.startswith("void ") else n
            head = n.split("(")[0].split("<")[0]
            n = head.split("::")[-1] + n[len(head):]   # (no namespaces in front of the kernel's own name)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
print(len(rows), "kernel launches in the trace")
ours = ("k_meta_init", "k_prepd", "k_prepg", "k_winplan", "k_tile_direct", "k_scan", "k_emit", "k_xmat", "k_exact")
# jobs: from a k_prepd to the next k_prepd
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_prepd")]
jobs = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
all_jobs = jobs
jobs = [j for j in jobs if all(any(x[2].startswith(o) for o in ours) or x[2].startswith("__amd_rocclr") for x in j)]  # steps only (no torch kernels between)
if not jobs and all_jobs:
    print("no clean job; kernels of the middle one:", [x[2][:30] for x in all_jobs[len(all_jobs) // 2]])
    sys.exit(0)
print(len(jobs), "jobs between two k_prepd launches with nothing but the library's kernels and copies")
agg = collections.OrderedDict()
spans = []
for j in jobs[len(jobs) // 4:]:
    prev_end = None
    for (s, e, n) in j:
        name = n.split("(")[0][:40]
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        if prev_end is not None:
            a[2] += (s - prev_end) / 1e3
        prev_end = e
    spans.append((j[-1][1] - j[0][0]) / 1e3)
nj = len(jobs) - len(jobs) // 4
print("kernel                                   per job   duration us   idle in front us")
for name, (c, d, g) in agg.items():
    print(f"{name:40s} {c / nj:7.2f} {d / c:12.2f} {g / c:14.2f}")
# from the last kernel of a job to the first of the next (host turnaround + launch latency)
turn = [(jobs[i + 1][0][0] - jobs[i][-1][1]) / 1e3 for i in range(len(jobs) // 4, len(jobs) - 1)]
print("first kernel to last kernel's end, mean us:", sum(spans) / len(spans))
if turn:
    print("last kernel's end to the next job's first kernel, mean us:", sum(turn) / len(turn), "min", min(turn))
