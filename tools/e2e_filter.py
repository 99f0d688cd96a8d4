#!/usr/bin/env python3
"""End-to-end timing of `polypolish filter` (two SAMs in, two tagged SAMs out) against the CPU oracle's
filter on the same files.  Usage: e2e_filter.py [genome_bp] [coverage] [repeat_frac]  (writes under $TMPDIR)"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 200
multi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
tmp = os.environ.get("TMPDIR", "/tmp")
L = 150
n_pairs = G * cov // (2 * L)
rng = np.random.default_rng(5)
t = time.time()
ins = np.clip(np.rint(rng.normal(350, 35, n_pairs)), 160, 700).astype(np.int64)
start = rng.integers(0, G - 700, n_pairs)
flip = rng.random(n_pairs) < 0.5           # which mate is forward
odd = rng.random(n_pairs) < 0.002          # wrong orientation / huge insert
is_multi = rng.random(n_pairs) < multi     # read 1 has 3 alignments, one of them consistent with the mate
genome = rng.integers(0, 4, G, dtype=np.uint8)
gstr = np.frombuffer(b"ACGT", dtype=np.uint8)[genome].tobytes().decode()
fa = os.path.join(tmp, "flt.fasta")
with open(fa, "w") as f:
    f.write(">c0\n" + gstr + "\n")
paths = [os.path.join(tmp, f"flt_{i}.sam") for i in (1, 2)]
with open(paths[0], "w") as f1, open(paths[1], "w") as f2:
    for f in (f1, f2):
        f.write(f"@SQ\tSN:c0\tLN:{G}\n")
    b1, b2 = [], []
    for i in range(n_pairs):
        s1, s2 = int(start[i]), int(start[i] + ins[i] - L)
        fl1, fl2 = (99, 147) if not flip[i] else (83, 163)
        if flip[i]:
            s1, s2 = s2, s1
        if odd[i]:
            fl2 = fl1 & ~0x10 | (fl1 & 0x10)   # same strand as mate 1
            fl2 = (fl2 & ~0x40) | 0x80
        b1.append(f"r{i}\t{fl1}\tc0\t{s1 + 1}\t60\t150M\t=\t{s2 + 1}\t0\t{gstr[s1:s1 + 150]}\t*\tNM:i:0\n")
        if is_multi[i]:
            for j in range(2):
                far = int(rng.integers(0, G - 700))
                b1.append(f"r{i}\t{fl1 | 256}\tc0\t{far + 1}\t0\t150M\t=\t{s2 + 1}\t0\t*\t*\tNM:i:1\n")
        b2.append(f"r{i}\t{fl2}\tc0\t{s2 + 1}\t60\t150M\t=\t{s1 + 1}\t0\t{gstr[s2:s2 + 150]}\t*\tNM:i:0\n")
        if len(b1) >= 100000:
            f1.write("".join(b1)); f2.write("".join(b2)); b1, b2 = [], []
    f1.write("".join(b1)); f2.write("".join(b2))
print(f"generated {n_pairs} pairs, SAMs {sum(os.path.getsize(p) for p in paths) / 1e6:.0f} MB in {time.time() - t:.1f} s", flush=True)

def run(cmd):
    t = time.time()
    r = subprocess.run(cmd, capture_output=True)
    return time.time() - t, r
outs_g = [os.path.join(tmp, f"flt_gpu_{i}.sam") for i in (1, 2)]
outs_c = [os.path.join(tmp, f"flt_cpu_{i}.sam") for i in (1, 2)]
args = lambda o: ["filter", "--in1", paths[0], "--in2", paths[1], "--out1", o[0], "--out2", o[1]]
for rep in range(2):
    dt_gpu, r1 = run([os.path.join(ROOT, "bin", "polypolish")] + args(outs_g))
dt_cpu, r2 = run([os.path.join(ROOT, "oracle", "_build", "pp_oracle")] + args(outs_c))
def sha(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()[:16]
same = all(sha(a) == sha(b) for a, b in zip(outs_g, outs_c))
err = r1.stderr.decode()
print("\n".join(l for l in err.split("\n") if "[timing]" in l or "Time to run" in l or "threshold" in l or " fail" in l))
print(f"rc {r1.returncode}/{r2.returncode}  bin/polypolish filter: {dt_gpu:.2f} s   oracle (1 core): {dt_cpu:.2f} s   speed-up {dt_cpu / dt_gpu:.1f}x   "
      f"identical SAMs: {same}")

# the whole chain: filter then polish (two processes, tagged SAMs on disk) vs the fused command vs the oracle's chain
exe = os.path.join(ROOT, "bin", "polypolish")
for rep in range(2):
    t = time.time()
    ra = subprocess.run([exe] + args(outs_g), capture_output=True)
    rb = subprocess.run([exe, "polish", fa] + outs_g, capture_output=True)
    dt_two = time.time() - t
for rep in range(2):
    dt_fused, rf = run([exe, "filter-polish", "--in1", paths[0], "--in2", paths[1], fa])
t = time.time()
rc1 = subprocess.run([os.path.join(ROOT, "oracle", "_build", "pp_oracle")] + args(outs_c), capture_output=True)
rc2 = subprocess.run([os.path.join(ROOT, "oracle", "_build", "pp_oracle"), "polish", fa] + outs_c, capture_output=True)
dt_orc = time.time() - t
print(f"filter + polish, two processes: {dt_two:.2f} s   fused filter-polish: {dt_fused:.2f} s   oracle chain (1 core): {dt_orc:.2f} s   "
      f"identical FASTA: {rb.stdout == rc2.stdout and rf.stdout == rc2.stdout}   speed-up fused {dt_orc / dt_fused:.1f}x, two-step {dt_orc / dt_two:.1f}x")
