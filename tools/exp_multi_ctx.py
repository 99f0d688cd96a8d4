"""The multi-GPU CLI path at a configuration's full size on whatever GPUs there are: files of bench.py --config C are
written, `bin/polypolish polish` runs once on one context and once on N contexts (PP_SHARE_GPU=N puts them all on GPU 0
when the box has only one; on a multi-GPU node PP_GPUS=N uses N devices), sha256 of the two FASTA outputs compared.
    python tools/exp_multi_ctx.py 4 8 [dir]      # config 4, 8 contexts"""
import hashlib, json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

config, n_ctx = int(sys.argv[1]), int(sys.argv[2])
tmp = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="pp_multi_")
os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda", 0)
lens, cov, repeat, label = bench.config_shape(config)
t = time.perf_counter()
job = bench.make_job(dev, contig_lens=lens, coverage=cov, seed=4242 + config, pairs=True, unaligned_frac=1e-3, repeat=repeat)
torch.cuda.synchronize()
fa, sams = bench.write_sam_pair(job, tmp)
G = job["G"]
del job
torch.cuda.empty_cache()
out = {"workload": label, "text_bytes": sum(os.path.getsize(p) for p in sams), "generated_s": round(time.perf_counter() - t, 1),
       "gpus_on_this_box": torch.cuda.device_count()}
exe = os.path.join(bench.ROOT, "bin", "polypolish")
def run(env_extra):
    env = dict(os.environ, **env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([exe, "polish", fa] + sams, capture_output=True, env=env)
    dt = time.perf_counter() - t0
    timing = [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]")]
    return r.returncode, hashlib.sha256(r.stdout).hexdigest(), round(dt, 3), timing, r.stderr.decode(errors="replace")[-300:] if r.returncode else ""
rc1, sha1, t1, tm1, e1 = run({"PP_DEVICE": "0", "PP_TIMING": "1"})
many = {"PP_GPUS": str(n_ctx)} if torch.cuda.device_count() >= n_ctx else {"PP_SHARE_GPU": str(n_ctx)}
rcn, shan, tn, tmn, en = run(dict(many, PP_TIMING="1"))
rch, shah, th, tmh, eh = run(dict(many, PP_DEVICE_INGEST="0"))
out.update({"one_context": {"rc": rc1, "wall_s": t1, "err": e1}, f"{n_ctx}_contexts_sharded_device_ingest": {"rc": rcn, "wall_s": tn, "err": en, "how": many},
            f"{n_ctx}_contexts_host_ingest_host_split": {"rc": rch, "wall_s": th, "err": eh},
            "identical_output": rc1 == rcn == rch == 0 and sha1 == shan == shah, "mbp": G / 1e6, "timing_one": tm1, "timing_many": tmn})
print(json.dumps(out))
if len(sys.argv) <= 3:
    for p in os.listdir(tmp):
        os.unlink(os.path.join(tmp, p))
    os.rmdir(tmp)
