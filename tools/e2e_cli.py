#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI (SAM text in, FASTA out) against the CPU oracle's text path on
the same files.  Usage: e2e_cli.py [genome_bp] [coverage]   (writes under $TMPDIR)"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tmp = os.environ.get("TMPDIR", "/tmp")
t = time.time()
contig_off, bases, recs = synth.fast_records(seed=3, contig_lens=(G,), coverage=cov, read_len=150, indel_read_frac=0.01)
fa, sam = os.path.join(tmp, "e2e.fasta"), os.path.join(tmp, "e2e.sam")
# vectorised-ish SAM writer (NM is not checked against the sequence by either tool)
seq = recs["seq"].reshape(-1, 150)
with open(fa, "w") as f:
    f.write(">c0\n" + bytes(bases).decode() + "\n")
ops = "MIDNSHP=X"
with open(sam, "w") as f:
    f.write(f"@SQ\tSN:c0\tLN:{G}\n")
    cig, co, nc, rs = recs["cigar"], recs["cig_off"], recs["n_cig"], recs["ref_start"]
    lines = []
    for i in range(len(rs)):
        c = "".join(f"{int(x) >> 4}{ops[int(x) & 15]}" for x in cig[int(co[i]):int(co[i]) + int(nc[i])])
        lines.append(f"r{i}\t0\tc0\t{int(rs[i]) + 1}\t60\t{c}\t*\t0\t0\t{seq[i].tobytes().decode()}\t*\tNM:i:1\n")
        if len(lines) == 100000:
            f.write("".join(lines)); lines = []
    f.write("".join(lines))
print(f"generated {len(rs)} records, SAM {os.path.getsize(sam) / 1e6:.0f} MB in {time.time() - t:.1f} s", flush=True)

def run(cmd):
    t = time.time()
    r = subprocess.run(cmd, capture_output=True)
    return time.time() - t, r
os.environ["PP_DEVICE_INGEST"] = "1"
for rep in range(2):
    dt_dev, r3 = run([os.path.join(ROOT, "bin", "polypolish"), "polish", fa, sam])
print("device tokenizer:\n" + "\n".join(l for l in r3.stderr.decode().split("\n") if "[timing]" in l or "Time to run" in l))
os.environ["PP_DEVICE_INGEST"] = "0"
for rep in range(2):
    dt_gpu, r1 = run([os.path.join(ROOT, "bin", "polypolish"), "polish", fa, sam])
dt_cpu, r2 = run([os.path.join(ROOT, "oracle", "_build", "pp_oracle"), "polish", fa, sam])
same = r1.stdout == r2.stdout
err = r1.stderr.decode()
print("\n".join(l for l in err.split("\n") if "[timing]" in l or "Time to run" in l))
print(f"bin/polypolish (device tokenizer, default): {dt_dev:.2f} s ({dt_cpu / dt_dev:.1f}x)   identical FASTA: {r3.stdout == r2.stdout}")
print(f"bin/polypolish with PP_DEVICE_INGEST=0 (host ingest): {dt_gpu:.2f} s ({G / 1e6 / dt_gpu:.2f} Mbp/s)   oracle (1 core): {dt_cpu:.2f} s ({G / 1e6 / dt_cpu:.3f} Mbp/s)"
      f"   speed-up {dt_cpu / dt_gpu:.1f}x   identical FASTA: {same}   sha256 {hashlib.sha256(r1.stdout).hexdigest()[:16]}")
