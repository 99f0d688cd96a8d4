#!/bin/bash
# One GPU call (gpurun: a fresh one-GPU MI355X box, this tree as it is) = a named list of steps.
#   tools/gpu_call.sh NAME [-t SECONDS] STEP...
# Every STEP is a shell command run from the repo root on the box, one after the other, each under its own `timeout`
# (STEP_TIMEOUT, default 900 s), stdout+stderr to gpurun_out/NAME/NN.log (merged back here when the call ends); the exit
# codes are collected in gpurun_out/NAME/status.txt.  Shorthands:
#   @tests            python -m pytest tests -m gpu -x -q
#   @bench[:ARGS]     python bench.py ARGS            (its JSON line to gpurun_out/NAME/bench*.json)
#   (profiles: 'bash tools/profile_round.sh TAG' as a step -- kernel trace + the two PMC passes + calibration)
# Example: tools/gpu_call.sh r4a @tests '@bench:--steps 20 --warmup 5' 'PP_TIMING=1 bin/polypolish polish a.fa x.sam > /dev/null'
set -u
name=$1; shift
total=2400
if [ "${1:-}" = "-t" ]; then total=$2; shift 2; fi
step_timeout=${STEP_TIMEOUT:-900}
out=gpurun_out/$name
script="ulimit -c 0; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd \$GRAFT_REPO_ROOT; : > $out/status.txt;"
i=0
for s in "$@"; do
    i=$((i + 1)); n=$(printf %02d $i)
    case "$s" in
        @tests) cmd="python -m pytest tests -m gpu -x -q" ;;
        @bench) cmd="python bench.py > $out/bench_$n.json" ;;
        @bench:*) cmd="python bench.py ${s#@bench:} > $out/bench_$n.json" ;;
        *) cmd="$s" ;;
    esac
    script+=" ( timeout $step_timeout bash -c $(printf %q "$cmd") ) > $out/$n.log 2>&1; echo \"$n rc=\$? $(printf %s "$s" | tr -c 'A-Za-z0-9_.,:=@/+-' ' ' | cut -c1-110)\" >> $out/status.txt;"
done
script+=" cat $out/status.txt"
exec /usr/local/graft/bin/gpurun --timeout "$total" -- "$script"
