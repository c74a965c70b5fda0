#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s20
rm -rf $OUT && mkdir -p $OUT
export RLCONTROL_HIP_LIB=$PWD/ab/base.so
timeout -k 10 400 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu -k "mfma and (ten_updates or k_updates or independent or philox_minibatches or learns_critic)" > $OUT/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc" | tee -a $OUT/gpu_tests.log
tail -5 $OUT/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
  for t4 in 1 0; do
    line="t4=$t4 rep$rep"
    for algo in ddpg sac naf; do
      if [ $algo = ddpg ]; then extra="--no-side-records"; else extra="--side-only $algo"; fi
      if [ $t4 = 0 ]; then export RLC_NO_TAIL4=1; else unset RLC_NO_TAIL4; fi
      v=$(timeout -k 10 300 python bench.py --no-cpu-baseline $extra --updates-per-step 64 --steps 6 --warmup 2 2>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
      line="$line $algo=$v"
    done
    echo "$line" | tee -a $OUT/ab.txt
  done
done
