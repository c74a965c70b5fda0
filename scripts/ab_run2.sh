#!/bin/bash
# on the GPU box: bench DDPG / SAC / NAF for every ab/<name>.so given on the command line (same box, same session)
#   REPS=2 ALGOS="ddpg sac naf" scripts/ab_run2.sh base a0 a1 ...   -> gpurun_out/ab_run2.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPS=${REPS:-1}
ALGOS=${ALGOS:-"ddpg sac naf"}
OUT=gpurun_out/${ABOUT:-ab_run2.txt}
for rep in $(seq 1 $REPS); do
  for n in "$@"; do
    line="$n rep$rep"
    for algo in $ALGOS; do
      if [ $algo = ddpg ]; then extra="--no-side-records"; else extra="--side-only $algo"; fi
      v=$(RLCONTROL_HIP_LIB=$PWD/ab/$n.so timeout -k 10 300 python bench.py --no-cpu-baseline $extra --updates-per-step 64 --steps 6 --warmup 2 2>gpurun_out/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
      line="$line $algo=$v"
    done
    echo "$line" | tee -a $OUT
  done
done
