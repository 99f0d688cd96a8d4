# Run ON THE GPU BOX: k_prepd's geometry (threads per workgroup x workgroups), bench.py's default job
for cfg in "1024 512" "1024 1024" "512 1024" "512 2048"; do
  set -- $cfg
  PP_PREPD_THREADS=$1 PP_PREPD_BLOCKS=$2 python bench.py --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 30 --warmup 5 $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threads=$1 blocks=$2', 'ms/step', d['ms_per_step'], d['kernel_ms_per_step'])"
done
