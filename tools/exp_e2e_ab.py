import os, sys, subprocess, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
tmp = "/tmp/e2e_ab"; os.makedirs(tmp, exist_ok=True)
job = bench.make_job(dev, contig_lens=[5_000_000], coverage=200, seed=4242, pairs=True, unaligned_frac=1e-3)
fa, sams = bench.write_sam_pair(job, tmp, qual=True)
del job
exe = "bin/polypolish"
for rep in range(3):
    for name, env in (("stream", {"PP_DEVICE_INGEST": "0"}), ("single", {"PP_DEVICE_INGEST": "0", "PP_STREAM_ADDS": "0"}), ("devtok", {})):
        t = time.perf_counter()
        r = subprocess.run([exe, "polish", fa] + sams, capture_output=True, env=dict(os.environ, PP_TIMING="1", **env))
        dt = time.perf_counter() - t
        print(rep, name, round(dt, 3), flush=True)
        if rep == 2:
            print("\n".join(l for l in r.stderr.decode().splitlines() if "[timing]" in l))
